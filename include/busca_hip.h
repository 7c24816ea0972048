/*
 * busca_hip.h - C-ABI of libbusca_hip.so, the MI355X (gfx950) implementation of BUSCA's per-frame
 * track-recovery hot path.  Plain pointers and sizes only; no torch / C++ types cross this boundary.
 *
 * The reference (lorenzovaquero/BUSCA) has no FFI: its boundary is the Python surface of
 * busca/network.py + busca/tracking.py.  Each entry point below names the reference code it replaces;
 * busca_amd/ (Python, same names and semantics as the reference's `busca` package) binds them with ctypes.
 *
 * Conventions
 *   - return 0 on success, negative BUSCA_E* on failure; busca_last_error(ctx) gives the message.
 *   - "dev" pointers are device (HBM) pointers owned by the caller (e.g. torch tensors' data_ptr()).
 *     "host" pointers are ordinary host memory.  `stream` is a hipStream_t (NULL = default stream).
 *   - one ctx per GPU/process; calls on one ctx are not re-entrant; no internal threads.
 *   - all launches are asynchronous on `stream`; nothing here synchronises the device except
 *     busca_ctx_destroy and the weight loaders.
 */
#ifndef BUSCA_HIP_H
#define BUSCA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct busca_ctx busca_ctx;

enum {
    BUSCA_OK = 0,
    BUSCA_EINVAL = -1,      /* bad argument / unsupported shape */
    BUSCA_ENOWEIGHTS = -2,  /* forward called before the matching load_weights */
    BUSCA_EHIP = -3,        /* a HIP runtime call failed */
    BUSCA_ENOMEM = -4
};

enum { BUSCA_ACT_RELU = 0, BUSCA_ACT_GELU = 1 };
/* arithmetic of the dense contractions; LayerNorm / softmax / residual stream are always f32.
 * BUSCA_PREC_F16X3 (ReID extractor, round 4; Decision Transformer, round 5): float32-EQUIVALENT products on the fp16 matrix cores -
 * every f32 operand is split into fp16 hi + lo and a product block is three fp16 MFMAs into one f32 accumulator (error-corrected
 * split GEMM; measured rms error 2.5e-8 of sum|a b| against float64, the f32 MFMA chain 2.8e-8); activations stay float32 in HBM,
 * statistics float64.  In the Decision Transformer the GEMMs (embed, Q / K / V, out-proj, FFN) take this form, attention, LayerNorm and
 * softmax are the f32 kernel's (logits 1e-5 from BUSCA_PREC_F32, 2.2x its speed); shapes beyond the one-kernel path run the exact f32
 * layer-wise kernels; weights beyond |w| = 255 are refused (load them with BUSCA_PREC_F32). */
enum { BUSCA_PREC_F32 = 0, BUSCA_PREC_F16 = 1, BUSCA_PREC_F16X3 = 2 };
enum { BUSCA_PAIR_CENTER = 0, BUSCA_PAIR_CENTER_WEIGHTED = 1, BUSCA_PAIR_IOU = 2, BUSCA_PAIR_IOU_COST = 3 };

/* ---- context ------------------------------------------------------------------------------ */
int busca_ctx_create(int device, busca_ctx** out);
void busca_ctx_destroy(busca_ctx* ctx);
const char* busca_last_error(const busca_ctx* ctx);
/* Library/ABI version: major*1000 + minor.  2000: busca_dt_cfg has the trailing `layout` field (36 bytes; a caller built against a
 * 1xxx header passes 32) and BUSCA_PREC_F16X3 exists.  busca_amd/_lib.py refuses a library whose major differs from the one it was written for. */
int busca_version(void);
/* The compiler flags this library was built with (busca_amd/build.py passes them in; bench.py records the string). */
const char* busca_build_info(void);
/* Options of one context.  Defaults are read from the environment ONCE, at busca_ctx_create (BUSCA_DT_NTRK, BUSCA_DT_TILED, BUSCA_DT_SPLIT, BUSCA_CROP_BAND);
 * no forward reads the environment.  Unknown name: BUSCA_EINVAL.
 *   "dt_ntrk"       0 auto / 1 / 2 tracks per workgroup of the f16 fused kernel (busca_amd.batcher pins it so that merged launches keep each step's flavour)
 *   "dt_tiled"      1 = force the layer-wise Decision-Transformer path (tests)
 *   "dt_split"      token-split tail of the fused kernel: -1 = the tracks of a launch's last, partial round run one 16-token tile per workgroup where that
 *                   pays (f32 / x3), 0 = never, 1 / 2 = every track that fits, one / two tracks per workgroup (tests)
 *   "dt_exact_f32"  1 = a context loaded with BUSCA_PREC_F16X3 runs its forwards in exact float32 on the f32 packing it keeps of the same weights (how the
 *                   host re-runs a step whose x3 forward reported a clipped operand)
 *   "dt_status"     get: 0 ok, 1 = a token-split launch lost a partner workgroup, 2 = a BUSCA_PREC_F16X3 forward had to clip an operand beyond |x| = 1023.5 -
 *                   valid once the forward's stream is synchronised; set 0: the caller has dealt with it (an uncleared status is returned by the next forward)
 *   "reid_status"   get: 0 ok, 2 = a BUSCA_PREC_F16X3 ReID forward since the last clear staged an activation beyond |x| = 1023.5 (its features are invalid:
 *                   non-finite BatchNorm statistics; nothing is clipped silently) - valid once the forwards' streams are synchronised; set 0 clears
 *   "crop_band"     1 = crops through the LDS-staged band kernel, 0 = one thread per output pixel (tests compare the two)
 *   get only: "last_dt_grid" / "last_dt_ntrk" / "last_dt_split" (workgroups, tracks per workgroup, token-split tracks of the last fused launch).
 * ReID schedule switches of a LOADED extractor that tests flip between two forwards (BUSCA_ENOWEIGHTS before weights are loaded): "reid_gram", "reid_halo",
 * "reid_fuse_c1", "reid_x3_fuse_c1", "reid_x3_gram_min", "reid_x3_merge_in_min", "reid_x3_row3", "reid_x3_ptail", "reid_x3_stem_halo", "reid_x3_stem_u8",
 * "reid_x3_stem_pool" (meanings: DESIGN.md section 5).  The remaining schedule thresholds are BUSCA_REID_* environment variables read by
 * busca_reid_load_weights (A/B runs). */
int busca_set_option(busca_ctx* ctx, const char* name, int32_t value);
int busca_get_option(busca_ctx* ctx, const char* name, int32_t* value);

/* ---- Decision Transformer (busca/network.py:176-244 without the ReID stage) ----------------- */
typedef struct {
    int32_t d;             /* trans_dim (network.py:17); 64, 256 or 512 */
    int32_t ff;            /* ff_size: a multiple of d, at most 8 d (2 d in every shipped config) */
    int32_t nhead;         /* d / nhead in {16, 32, 64, 128} (4 in every shipped config, config/ **.yml:3).  nhead = 4 with ff = 2 d runs as
                              ONE kernel when the tokens fit on chip; every other geometry runs layer-wise (same arithmetic type) */
    int32_t nlayers;       /* num_layer, <= 8 */
    int32_t E;             /* dim_embedding of the ReID feature (512) */
    int32_t activation;    /* BUSCA_ACT_*; the reference effectively runs RELU */
    int32_t fake_bbox_f64; /* 1: candidate-side bucket math in float64 (reference's pinned numpy 1.23.5,
                              busca/tracking.py:12 + busca/encodings.py:21,127); 0: float32 (numpy >= 2) */
    int32_t precision;     /* BUSCA_PREC_* */
    int32_t layout;        /* BUSCA_LAYOUT_* bits; 0 = MEM-SEP-CAN-BAD with separators encoded as the reference box (every shipped
                              config).  Selects among the token orders / box encodings of network.py:103-165, encodings.py:112-146 */
} busca_dt_cfg;
#define BUSCA_LAYOUT_CAN_FIRST 1   /* input_flavour MEM-CAN-SEP*: each candidate precedes its separator */
#define BUSCA_LAYOUT_NO_BAD 2      /* input_flavour without -BAD: no BAD token; logits / probs are [B, P+1] (the blob's bad_token is ignored) */
#define BUSCA_LAYOUT_SEP_AS_CAN 4  /* encode_separator_as_reference = false: a separator is encoded with its candidate's box */

/* Number of float32 values in the flat weight blob for `cfg` (layout below). */
size_t busca_dt_blob_floats(const busca_dt_cfg* cfg);

/*
 * Load Decision-Transformer weights (replaces nn.Module state held by busca/network.py:45-100 and
 * BUSCA.load_pretrained, network.py:432-467).  `blob` is HOST float32, reference state_dict tensors
 * concatenated in this order (row-major, torch layouts):
 *   encoder.weight[d,E] encoder.bias[d] sep_token[d] non_token[d] bad_token[d]
 *   for each layer: self_attn.in_proj_weight[3d,d] in_proj_bias[3d] out_proj.weight[d,d] out_proj.bias[d]
 *                   linear1.weight[ff,d] linear1.bias[ff] linear2.weight[d,ff] linear2.bias[d]
 *                   norm1.weight[d] norm1.bias[d] norm2.weight[d] norm2.bias[d]
 *   decoder.0.weight[d] decoder.0.bias[d] decoder.1.weight[d] decoder.1.bias[1]
 * `lut_xy`,`lut_sz` ([211, lut_c]) and `lut_t` ([61, lut_c]) are HOST IEEE fp16 bit patterns: the three
 * per-axis sin/cos tables the reference's 211x211x61xd table factors into (busca/encodings.py:23-32).
 * The library repacks the matrices into MFMA operand-fragment order and uploads everything.
 */
int busca_dt_load_weights(busca_ctx* ctx, const busca_dt_cfg* cfg, const float* blob, size_t blob_floats,
                          const uint16_t* lut_xy, const uint16_t* lut_sz, const uint16_t* lut_t, int32_t lut_c);

/*
 * One association step for B lost tracks x P proposals (replaces BUSCA.forward network.py:203-232,
 * PositionalEncoding.forward encodings.py:43-94 and the softmax/argmax of network.py:403,415-421).
 * All pointers dev.  T = L + 2*(P+2); with BUSCA_LAYOUT_NO_BAD read P+1 for every P+2 below (no BAD column).
 *   mem_feat [B,L,E] f32, can_feat [B,P,E] f32 : ReID features (what reid_encoder returns)
 *   mem_ltrb [B,L,4] f32, can_ltrb [B,P,4] f32 : boxes as passed to BUSCA.forward (ltrb)
 *   logits   [B,P+2] f32  pre-softmax, order [slot_0..slot_{P-1}, NON, BAD]   (required)
 *   probs    [B,P+2] f32  softmax(logits)                                      (may be NULL)
 *   argmax   [B]     i32  first index of the row maximum of probs              (may be NULL)
 *   hidden   [B,T,d] f32  transformer output (source of .logits/.mem_logits)   (may be NULL)
 *   att      [nlayers,B,nhead,T,T] f32 per-head attention weights              (may be NULL)
 * One Decision-Transformer forward per context at a time (its layer-wise workspace and the exchange buffers of the token-split tail belong
 * to the context; forwards on ONE stream are ordered by the stream - use one context per concurrently running stream).  A forward of the
 * BUSCA_PREC_F16X3 flavour that had to clip an operand, or a split launch that lost a partner workgroup, leaves a status word the caller reads once the
 * stream is synchronised (busca_get_option "dt_status"; busca_amd's wrappers do, and re-run a clipped step with "dt_exact_f32") - a status nobody
 * cleared is returned by the NEXT call (BUSCA_EINVAL / BUSCA_EHIP with the reason), whose own kernels are launched all the same.
 */
int busca_dt_forward(busca_ctx* ctx, const float* mem_feat, const float* can_feat, const float* mem_ltrb,
                     const float* can_ltrb, int32_t B, int32_t L, int32_t P, float* logits, float* probs,
                     int32_t* argmax, float* hidden, float* att, void* stream);

/* Shapes beyond the fused kernel's on-chip plan (T > 80, d = 512 with T > 64 in f16; T > 48 / 32 in f32) run layer by layer
 * with activations in a context-owned HBM workspace.  busca_dt_reserve sizes that workspace for (B, L, P) ahead of time, so
 * that no forward allocates; without it the first forward of a larger shape grows the workspace itself (one stream
 * synchronisation + hipMalloc).  Both precisions are served: the f32 flavour keeps the reference's arithmetic
 * (busca/custom_layers.py:30-41 computes every shape in float32). */
int busca_dt_reserve(busca_ctx* ctx, int32_t B, int32_t L, int32_t P, void* stream);

/* Bucket indices only (busca/encodings.py:150-235): ids [B,T,3] i32 = (xy, size, time) per token. */
int busca_dt_bucket_ids(busca_ctx* ctx, const float* mem_ltrb, const float* can_ltrb, int32_t B, int32_t L,
                        int32_t P, int32_t* ids, void* stream);

/* Average duration in ms of the dt_forward kernel launches issued through this ctx since the last
 * reset, measured with HIP events on the launch stream (used by bench.py's roofline leg).
 * Enabling timing adds two event records per launch. */
int busca_timing_enable(busca_ctx* ctx, int32_t on);
int busca_timing_read(busca_ctx* ctx, double* avg_ms, int64_t* launches, int32_t reset);

/* ---- proposal geometry (busca/tracking.py:23-60; adapters/.../matching.py:53-91,173-186) ------ */
/*
 * out[nA,nB] f64.  a,b: [n,4] f64 ltrb boxes (dev).  mode:
 *   CENTER           euclidean distance between box centres (center_distance, weight_size=False)
 *   CENTER_WEIGHTED  ... times max(sqrt(area_a)/sqrt(area_b), inverse)  (weight_size=True)
 *   IOU              cython_bbox.bbox_overlaps ("+1" pixel convention)
 *   IOU_COST         1 - IOU (iou_distance)
 * scores_b (dev f64 [nB]) may be NULL; with IOU_COST it applies fuse_score: 1 - iou*score_b.
 */
int busca_pairwise(busca_ctx* ctx, const double* a, int32_t nA, const double* b, int32_t nB, int32_t mode,
                   const double* scores_b, double* out, void* stream);

/* idx[B,P] i32 = per-row indices of the P smallest values of dist[B,N] f64, ascending, ties by lower
 * index; -1 pads rows when N < P (np.argsort(row)[:P] + None padding, network.py:333-338). */
int busca_topk_rows(busca_ctx* ctx, const double* dist, int32_t B, int32_t N, int32_t P, int32_t* idx, void* stream);

/* Detection coverage of the reliability gate (adapters/ByteTrack/yolox/tracker/byte_tracker.py:459-465,574-623):
 * *count (dev u64) = pixels of an H x W frame covered by the union of n filled rectangles.  rects: dev i32 [n,4]
 * x1,y1,x2,y2 with x1<=x2, y1<=y2, already int()-truncated and clipped to the frame (cv2.rectangle semantics, both
 * corner pixels included). */
int busca_coverage(busca_ctx* ctx, const int32_t* rects, int32_t n, int32_t H, int32_t W, uint64_t* count, void* stream);

/* Camera-motion compensation (adapters/ByteTrack/yolox/tracker/byte_tracker.py:626-650): cv2.cvtColor(BGR2GRAY) of both frames
 * + cv2.findTransformECC(templateImage = previous, inputImage = current, warpMatrix, motionType, (EPS | COUNT, max_iters, eps))
 * with the default 5x5 Gaussian pre-filter.  prev / cur: dev u8 [H,W,3] BGR (row strides in bytes).  motion 0 = MOTION_EUCLIDEAN,
 * 1 = MOTION_AFFINE.  warp: HOST float32 [6], row-major 2x3, in = initial guess (identity in the reference), out = estimate.
 * *cc (host) = the enhanced correlation coefficient the reference returns, *iters (host, may be NULL) = iterations run.
 * The per-iteration image work (warp, Jacobian, all reductions) runs on the GPU; the 3x3 / 6x6 solve between iterations on
 * the host, so this call SYNCHRONISES `stream` once per iteration.  Errors where OpenCV raises StsNoConv return BUSCA_EINVAL.
 * Third-party arithmetic (opencv_python 4.7.0.72) restated from the published algorithm: parity unpinned (oracle/ecc.py). */
int busca_ecc_align(busca_ctx* ctx, const uint8_t* prev, const uint8_t* cur, int32_t H, int32_t W, int32_t stride_prev, int32_t stride_cur,
                    int32_t motion, int32_t max_iters, double eps, float* warp, double* cc, int32_t* iters, void* stream);

/* ---- track state of the association rounds ------------------------------------------------ */
/* STrack.multi_predict (adapters/ByteTrack/yolox/tracker/byte_tracker.py:50-61): constant-velocity Kalman prediction
 * of n tracks in place.  mean dev f64 [n,8] (x,y,a,h,vx,vy,va,vh), cov dev f64 [n,8,8]; not_tracked dev u8 [n] or NULL:
 * 1 zeroes mean[7] first (state != Tracked, :55-56).  Process noise: std weights 1/20 (position) and 1/160 (velocity)
 * times the box height, 1e-2 / 1e-5 for the aspect ratio (KalmanFilter.multi_predict,
 * adapters/CenterTrack/src/lib/utils/mot_online/kalman_filter.py:154-190 - the copy byte_tracker.py:15 falls back to).
 * Bit-exact against the numpy evaluation. */
int busca_kalman_multi_predict(busca_ctx* ctx, double* mean, double* cov, const uint8_t* not_tracked, int32_t n, void* stream);
/* remove_duplicate_stracks (byte_tracker.py:685-698) on an IoU-cost matrix cost[nA,nB] (dev f64, busca_pairwise
 * IOU_COST): for every pair with cost < thresh (0.15) the track with the smaller age (frame_id - start_frame, dev i32)
 * is dropped; ties drop the A track.  keep_a [nA], keep_b [nB] dev u8: 1 = keep. */
int busca_duplicate_masks(busca_ctx* ctx, const double* cost, int32_t nA, int32_t nB, const int32_t* age_a, const int32_t* age_b,
                          double thresh, uint8_t* keep_a, uint8_t* keep_b, void* stream);

/* ---- crops (busca/tracking.py:62-113, busca/network.py:492-507) ------------------------------- */
/*
 * frame: dev u8 [H,W,3] (BGR, row stride `stride` bytes); boxes: dev f32 [n,4] x1y1x2y2.
 * out_u8 (may be NULL): dev u8 [n,384,128,3] BGR - exactly get_image_crops(normalize=False).
 * out_f16 (may be NULL): dev fp16 [n,384,128,4] RGB0 normalised ((x/255-mean)/std, ghost std): the layout the fp16 ReID stem
 *   stages internally (no entry point takes it; the ReID forwards read the u8 crops).
 */
int busca_crop_gather(busca_ctx* ctx, const uint8_t* frame, int32_t H, int32_t W, int32_t stride,
                      const float* boxes, int32_t n, uint8_t* out_u8, void* out_f16, void* stream);

/*
 * The same with the box extents already rounded by the caller and optional per-crop destinations:
 *   rects  dev i32 [n,4] = (floor(x1), floor(y1), ceil(x2), ceil(y2)) - busca/tracking.py:84-87 applies math.floor / math.ceil
 *          to the caller's float64 values (`tlbr * scale`); rounding them on the host in float64 keeps the cut-out extent
 *          identical where a float32 copy of the box would land on the other side of an integer.
 *   dst_u8 dev u64 [n] or NULL: when given, crop i (147 456 bytes, u8 BGR) is written to address dst_u8[i] (a slot of the
 *          device-resident crop pool, busca_amd/crop_pool.py) IN ADDITION to out_u8 + i*147456 when out_u8 is given; out_u8 may be NULL.
 * rects and dst_u8 may also point into PINNED host memory the GPU maps (hipHostMalloc): one thread per crop reads them once and the
 * kernels work on device copies - the caller then needs no host-to-device copy, but must keep the tables unchanged until the launch has run.
 */
int busca_crop_gather_ex(busca_ctx* ctx, const uint8_t* frame, int32_t H, int32_t W, int32_t stride, const int32_t* rects,
                         int32_t n, const uint64_t* dst_u8, uint8_t* out_u8, void* out_f16, void* stream);
/*
 * Crops of any output size (busca/tracking.py:62-71 `get_bbox_crop(..., output_size=(w, h))`, busca/network.py:492-507 `output_size`): the
 * same cut-out + pad + cv2.INTER_LINEAR restatement as busca_crop_gather_ex, u8 BGR [n, out_h, out_w, 3] only.  rects as in
 * busca_crop_gather_ex (device or pinned host).  The ReID path itself only ever uses 384 x 128 (busca_crop_gather[_ex]).
 */
int busca_crop_gather_sized(busca_ctx* ctx, const uint8_t* frame, int32_t H, int32_t W, int32_t stride, const int32_t* rects,
                            int32_t n, int32_t out_h, int32_t out_w, uint8_t* out_u8, void* stream);
/*
 * Index gather of track memories (busca/network.py:247-279 `_get_track_mem` + the np.array(...) stacking of :313,383):
 * out dev u8 [n,384,128,3]; crop i is copied from device address src[i] (dev u64 [n]); src[i] == 0 gives an all-zero crop
 * (incomplete memory :306, padded candidate :354).
 */
int busca_gather_crops(busca_ctx* ctx, const uint64_t* src, int32_t n, uint8_t* out, void* stream);

/* ---- ReID feature extractor (busca/network.py:510-575 + busca/reid/resnet.py:266-322) --------- */
/* Number of float32 values in the ReID weight blob (layout: see busca_amd/weights.py:reid_blob). */
size_t busca_reid_blob_floats(void);
int busca_reid_load_weights(busca_ctx* ctx, const float* blob, size_t blob_floats);      /* = _ex(..., BUSCA_PREC_F16) */
/* precision BUSCA_PREC_F16: fp16 activations/weights, f32 accumulation and statistics (fastest; features ~6e-3 from the reference).
 * precision BUSCA_PREC_F32: float32 activations/weights on exact-f32 MFMA - reference-exact (~1e-5), ~6x slower than F16.
 * precision BUSCA_PREC_F16X3: float32 activations, split-fp16 products (see the enum) - the same ~1e-5 against the reference at
 *           a third of the fp16 MFMA rate instead of a sixteenth. */
int busca_reid_load_weights_ex(busca_ctx* ctx, const float* blob, size_t blob_floats, int32_t precision);
/* crops: dev u8 [n,384,128,3] BGR (what the trackers keep in images_mem).  feats: dev f32 [n,512],
 * L2-normalised.  ONE CALL == ONE BatchNorm batch (train-mode statistics, network.py:553-556). */
int busca_reid_forward(busca_ctx* ctx, const uint8_t* crops, int32_t n, float* feats, void* stream);
/* zero_norm dev u8 [n] or NULL: 1 marks a crop whose pixels are 0.0 AFTER normalisation - the zero crops of
 * associate_embeddings(normalize_ims=False) (busca/network.py:285,306,354: float32 zeros that skip _normalize_embeddings_batch);
 * its bytes in `crops` are ignored.  With normalize_ims=True the zero crops are u8 zeros and need no flag. */
int busca_reid_forward_ex(busca_ctx* ctx, const uint8_t* crops, int32_t n, const uint8_t* zero_norm, float* feats, void* stream);
/* The same BatchNorm batch with repeated crops given ONCE: weights dev f32 [n] = how often crop i occurs in the batch the
 * reference builds (busca/network.py:340-358 picks each track's P nearest detections, so one detection's crop is repeated for
 * many tracks: 4 096 candidate slots over ~160 detections at 128 lost x 32 proposals), weight_sum = their sum (host value).
 * A crop's conv outputs do not depend on its copies; only the batch statistics do, and those are accumulated with the
 * multiplicities - the result equals the forward over the expanded batch up to floating-point summation order.
 * weights == NULL: every crop once (weight_sum ignored). */
int busca_reid_forward_w(busca_ctx* ctx, const uint8_t* crops, int32_t n, const uint8_t* zero_norm, const float* weights, double weight_sum,
                         float* feats, void* stream);
/* Bytes of device workspace busca_reid_forward needs for n crops (allocated lazily inside the ctx). */
size_t busca_reid_workspace_bytes(int32_t n);
/* Size the workspace that forwards on `stream` use for batches of up to n crops NOW (counterpart of busca_dt_reserve), so that no
 * later busca_reid_forward* call on that stream synchronises the device and allocates.  Needs loaded weights (the element size
 * follows the loaded precision).  One workspace per calling stream, at most 4 streams. */
int busca_reid_reserve(busca_ctx* ctx, int32_t n, void* stream);
/*
 * Building block of the large-batch ReID schedule, exposed for unit tests: train-mode BatchNorm (scale, shift) of a
 * 1x1 conv y = w x of stride `stride` WITHOUT running the conv, from the Gram matrix of its input
 * (nn.BatchNorm2d batch statistics, network.py:553-556, of resnet.py:108-128's conv3 / downsample).
 *   x      dev fp16 NHWC [n,H,W,Cin] (Cin 64, 128, 256 or 512);  in_ss dev f32 [Cin][2] or NULL: when given,
 *          the conv's input is relu(x*scale+shift) rounded to fp16 (the producer's BatchNorm, applied on the fly)
 *   w      dev fp16 [Cout][Cin];  gamma, beta dev f32 [Cout];  ss_out dev f32 [Cout][2] = (scale, shift), eps 1e-5
 * Synchronises `stream` (scratch is allocated and freed inside the call).
 */
int busca_bn_stats_1x1(busca_ctx* ctx, const void* x, const float* in_ss, int32_t n, int32_t H, int32_t W, int32_t Cin,
                       int32_t stride, const void* w, int32_t Cout, const float* gamma, const float* beta, float* ss_out,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BUSCA_HIP_H */
