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

// ReID schedule knobs by name: the ones a test flips between two forwards of a loaded extractor (tests/test_reid_gpu.py: schedules must agree / be bit-identical).
// Every other schedule threshold is a BUSCA_REID_* environment variable read when weights are loaded (busca_reid_load_weights) - A/B runs only.
static int* reid_option_field(ReidState& R, const std::string& n) {
    struct { const char* name; int* p; } tab[] = {{"reid_gram", &R.gram_mode}, {"reid_x3_gram_min", &R.x3_gram_min}, {"reid_x3_merge_in_min", &R.x3_merge_in_min},
                                                  {"reid_x3_row3", &R.x3_row3}, {"reid_x3_ptail", &R.x3_ptail_min}};
    for (auto& e : tab) if (n == e.name) return e.p;
    return nullptr;
}
static bool* reid_option_flag(ReidState& R, const std::string& n) {
    struct { const char* name; bool* p; } tab[] = {{"reid_halo", &R.halo}, {"reid_fuse_c1", &R.fuse_c1}, {"reid_x3_fuse_c1", &R.x3_fuse_c1},
                                                   {"reid_x3_stem_halo", &R.x3_stem_halo}, {"reid_x3_stem_u8", &R.x3_stem_u8}, {"reid_x3_stem_pool", &R.x3_stem_pool}};
    for (auto& e : tab) if (n == e.name) return e.p;
    return nullptr;
}

int reid_set_option(busca_ctx* c, const char* name, int32_t value) {
    const std::string n(name);
    if (n == "reid_status") { if (c->reid->xerr) *c->reid->xerr = value; return BUSCA_OK; }      // 0 = the caller has read the status of its synchronised forwards and dealt with it
    if (!c->reid->loaded) return fail(c, BUSCA_ENOWEIGHTS, "busca_set_option('%s'): ReID schedule options belong to a loaded extractor (load weights first)", name);
    if (int* p = reid_option_field(*c->reid, n)) { *p = value; return BUSCA_OK; }
    if (bool* p = reid_option_flag(*c->reid, n)) { *p = value != 0; return BUSCA_OK; }
    return fail(c, BUSCA_EINVAL, "busca_set_option: unknown option '%s'", name);
}
int reid_get_option(busca_ctx* c, const char* name, int32_t* value) {
    const std::string n(name);
    // 0 ok, 2 = a split-fp16 (BUSCA_PREC_F16X3) forward since the status was last cleared staged an operand beyond |x| = 1023.5: its features are invalid (non-finite
    // statistics) - valid once the forwards' streams are synchronised
    if (n == "reid_status") { *value = c->reid->xerr ? *c->reid->xerr : 0; return BUSCA_OK; }
    if (int* p = reid_option_field(*c->reid, n)) { *value = *p; return BUSCA_OK; }
    if (bool* p = reid_option_flag(*c->reid, n)) { *value = *p ? 1 : 0; return BUSCA_OK; }
    return fail(c, BUSCA_EINVAL, "busca_get_option: unknown option '%s'", name);
}

#include "capi_reid.hip.inc"
