// libbusca_hip.so, ReID extractor: every flavour of the ResNet-50 kernels and their C-ABI (units: busca_internal.hpp).  gfx950 only.
#include "busca_internal.hpp"

#include "dt_kernel.hip.inc"          // (vector typedefs and the small device helpers the conv kernels share with the Decision Transformer)
#include "reid_kernel.hip.inc"
#include "reid_gram.hip.inc"
#include "reid_halo.hip.inc"
#include "reid_kwave.hip.inc"
#include "reid_pipe.hip.inc"
#include "reid_f32.hip.inc"
#include "reid_x3.hip.inc"
#include "reid_x3p.hip.inc"

ReidState* reid_state_new() { return new ReidState(); }
void reid_state_delete(ReidState* r) { if (r) { reid_free(*r); delete r; } }
bool reid_state_loaded(const ReidState* r) { return r && r->loaded; }

// ReID schedule knobs by name ("reid_gram", "reid_halo", ...): the same fields the BUSCA_REID_* environment variables set when weights are
// loaded; through busca_set_option they change between two forwards of a loaded extractor (A/B runs, the tests that compare schedules).
static int* reid_option_field(ReidState& R, const std::string& n) {
    struct { const char* name; int* p; } tab[] = {
        {"reid_gram", &R.gram_mode}, {"reid_halo_min", &R.halo_min_blocks}, {"reid_halo_half", &R.halo_half_blocks},
        {"reid_halo_wpx", &R.halo_wpx}, {"reid_halo_wpx_min", &R.halo_wpx_min}, {"reid_gram_min", &R.gram_min_pixels}, {"reid_direct_rows", &R.direct_rows},
        {"reid_fuse_ds_layers", &R.fuse_ds_layers}, {"reid_fuse_c1_layers", &R.fuse_c1_layers}, {"reid_kwave_blocks", &R.kwave_blocks},
        {"reid_kwave_halo", &R.kwave_halo_blocks}, {"reid_kwave_nw", &R.kwave_nw}, {"reid_kwave_pt", &R.kwave_pt}, 
        {"reid_pipe_min", &R.pipe_min_tiles}, {"reid_pipe_half", &R.pipe_half_blocks},
        {"reid_x3_merge_layers", &R.x3_merge_layers}, {"reid_x3_half", &R.x3_half_blocks}, {"reid_x3_gram_min", &R.x3_gram_min}, {"reid_x3_merge_in_min", &R.x3_merge_in_min}, {"reid_x3_fuse_c1_min", &R.x3_fuse_c1_min}, {"reid_x3_narrow3", &R.x3_narrow3}, {"reid_x3_row3", &R.x3_row3}, {"reid_x3_ptail", &R.x3_ptail_min}};
    for (auto& e : tab) if (n == e.name) return e.p;
    return nullptr;
}
static bool* reid_option_flag(ReidState& R, const std::string& n) {
    struct { const char* name; bool* p; } tab[] = {{"reid_halo", &R.halo}, {"reid_fuse_c1", &R.fuse_c1}, {"reid_fuse_c1_small", &R.fuse_c1_small},
                                                   {"reid_stats2", &R.two_launch_stats}, {"reid_pipe_all", &R.pipe_all}, {"reid_x3_gram", &R.x3_gram}, {"reid_x3_merge_in", &R.x3_merge_in}, {"reid_x3_fuse_c1", &R.x3_fuse_c1}, {"reid_x3_stem_halo", &R.x3_stem_halo}, {"reid_x3_stem_u8", &R.x3_stem_u8}, {"reid_x3_stem_pool", &R.x3_stem_pool}};
    for (auto& e : tab) if (n == e.name) return e.p;
    return nullptr;
}

int reid_set_option(busca_ctx* c, const char* name, int32_t value) {
    const std::string n(name);
    if (!c->reid->loaded) return fail(c, BUSCA_ENOWEIGHTS, "busca_set_option('%s'): ReID schedule options belong to a loaded extractor (load weights first)", name);
    if (int* p = reid_option_field(*c->reid, n)) { *p = value; return BUSCA_OK; }
    if (bool* p = reid_option_flag(*c->reid, n)) { *p = value != 0; return BUSCA_OK; }
    return fail(c, BUSCA_EINVAL, "busca_set_option: unknown option '%s'", name);
}
int reid_get_option(busca_ctx* c, const char* name, int32_t* value) {
    const std::string n(name);
    if (int* p = reid_option_field(*c->reid, n)) { *value = *p; return BUSCA_OK; }
    if (bool* p = reid_option_flag(*c->reid, n)) { *value = *p ? 1 : 0; return BUSCA_OK; }
    return fail(c, BUSCA_EINVAL, "busca_get_option: unknown option '%s'", name);
}

#include "capi_reid.hip.inc"
