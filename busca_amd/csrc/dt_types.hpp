// Host-visible types of the Decision-Transformer kernels: shared by every translation unit of libbusca_hip.so (busca_internal.hpp) and by the kernel sources.
#pragma once

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 rh8 __attribute__((ext_vector_type(8)));
typedef _Float16 rh4 __attribute__((ext_vector_type(4)));

// LDS tile of 128-byte rows, 16-byte slots XOR-swizzled by the row (ReID conv kernels, layer-wise Decision-Transformer GEMMs)
static __device__ __forceinline__ int swz8(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }


#define DT_MAX_LAYERS 8

struct DTLayerW {
    const u32x4* w_in;  const float* b_in;     // [3d,d] packed, [3d]
    const u32x4* w_out; const float* b_out;    // [d,d]
    const u32x4* w1;    const float* b1;       // [ff,d]
    const u32x4* w2;    const float* b2;       // [d,ff]
    const float *g1, *be1, *g2, *be2;          // LayerNorm affine
};

struct DTParams {
    const u32x4* w_embed; const float* b_embed;            // [d,E] packed
    const float *tok_sep, *tok_non, *tok_bad;
    DTLayerW layer[DT_MAX_LAYERS];
    const float *dec_g, *dec_b, *dec_w; float dec_bias;
    float dec_cb;             // sum_f dec_b[f] dec_w[f] + dec_bias (host, float64 accumulation): the row-independent part of the decoder's Linear(LayerNorm(x)) (layer-wise path)
    const _Float16 *lut_xy, *lut_sz, *lut_t; int lut_c;
    const float *mem_feat, *can_feat, *mem_ltrb, *can_ltrb;
    float* logits; float* probs; int* argmax; float* hidden; float* att;
    long long* prof;          // debug: per-wave phase timestamps of workgroup 0 (BUSCA_DT_PROF=1), else NULL
    int B, L, P, T, nlayers, act, fake_f64;
    // token-split tail (dt_fused_kernel<..., SPLIT = true>, launched after the one-track workgroups of tracks [0, nsingle)): the tracks from nsingle on are held
    // by one workgroup per 16-token tile; they exchange their K / V tiles per layer through xch (agent-scope stores) under the per-wave flags xflag
    int nsingle; unsigned xepoch; unsigned long long* xch; unsigned* xflag; float* xlg; int* xerr;
    int can_pos, nspec, sep_can;   // token layout (busca_dt_cfg::layout): position of the CAN token in its (SEP, CAN) pair, special candidates (NON [, BAD]), SEP encoded with the candidate's box
};
#define DT_PROF_SLOTS 64

#define DT_X3_XS 64.0f     // split-fp16 flavour (Prec<2>, dt_kernel.hip.inc): activation / weight pre-scales (powers of two: exact)
#define DT_X3_WS 256.0f
#define DT_XMAX_MT 4       // most token tiles (= workgroups) of a split track
#define DT_XFLAGS 5        // flags per (track, token tile): one per wave for the K / V tiles of a layer, one for the decoder logits
