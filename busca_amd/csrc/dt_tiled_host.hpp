// Host side of the layer-wise Decision-Transformer path (dt_tiled.hip.inc): one kernel sequence per forward.  Included by busca_dtl_f32 / _f16.hip after
// dt_kernel.hip.inc and dt_tiled.hip.inc.
#pragma once
#include "busca_internal.hpp"

// ---- tiled (layer-wise) path ---------------------------------------------------------------------------------------
template <int PREC, int D, int EPI, int RT>
static int dtl_gemm_rt(busca_ctx* c, hipStream_t s, const DTLArgs& a, int ncolblocks) {
    constexpr int BM = 32 * RT;
    const size_t lds = (size_t)(BM + D) * 128 + 2 * 4 * BM * sizeof(float);
    auto kern = dtl_gemm_kernel<PREC, D, EPI, RT>;
    { int rc = ensure_lds(c, (const void*)kern, lds); if (rc) return rc; }
    TimedLaunch tl(c, s);
    // (embed pass: GEMM workgroups for the compacted feature rows, then the workgroups that fill the special token rows)
    const int nbx = EPI == DTL_EPI_EMBED ? (a.Me + BM - 1) / BM + (a.M - a.Me + BM - 1) / BM : (a.M + BM - 1) / BM;
    hipLaunchKernelGGL(kern, dim3(nbx, ncolblocks), dim3(512), lds, s, a);
    return BUSCA_OK;
}

// 64-row tiles (two workgroups per CU: one's epilogue traffic overlaps the other's MFMA phase) for the epilogue-heavy GEMMs
// when the tile fits twice into the LDS (d <= 512); BUSCA_DTL_RT=4 / 2 forces either geometry.
template <int PREC, int D, int EPI>
static int dtl_gemm(busca_ctx* c, hipStream_t s, const DTLArgs& a, int ncolblocks) {
    const int rt_env = c->opt.dtl_rt, rt_mask = c->opt.dtl_rt_mask;      // bit EPI of the mask = 64-row tiles for that GEMM kind
    // default: 64-row tiles only when 128-row tiles would fill less than half the chip (fewer than 128 workgroups) - measured
    // 128 lost x 32 proposals x d512 (79 row blocks): 64-row tiles for the single-column-block GEMMs 0.93 -> 0.86 ms (f16), f32
    // 3.38 -> 2.9 ms; from 158 row blocks on (two such steps in flight) 64-row tiles LOSE 5-10 %, and at >= 256 workgroups
    // the two geometries are within +-8 % per GEMM kind with no consistent winner (512 x 64 x d512: 4.25 ms vs 4.6 ms)
    const bool underfilled = (long)((a.M + 127) / 128) * ncolblocks < 128;
    const bool small = rt_mask >= 0 ? ((rt_mask >> EPI) & 1) != 0 : (rt_env == 2 || (rt_env == 0 && underfilled));
    if (small && (size_t)(64 + D) * 128 + 2 * 4 * 64 * 4 <= 80 * 1024) return dtl_gemm_rt<PREC, D, EPI, 2>(c, s, a, ncolblocks);
    return dtl_gemm_rt<PREC, D, EPI, 4>(c, s, a, ncolblocks);
}

template <int PREC, int HD, int MT>
static int dtl_attention(busca_ctx* c, hipStream_t s, const void* qkv, void* O, int B, int T, int D, int NH, float* att) {
    constexpr int ES = Prec<PREC>::ES, TPK = Prec<PREC>::CHUNK * Prec<PREC>::nchunks(MT);
    const size_t lds = (size_t)16 * MT * (HD * ES + 16) + (size_t)HD * (TPK * ES + 16);
    if (lds > 160 * 1024) return fail(c, BUSCA_EINVAL, "tiled attention: %d tokens x head dim %d do not fit the LDS in this precision", T, HD);
    auto kern = dtl_attention_kernel<PREC, HD, MT>;
    { int rc = ensure_lds(c, (const void*)kern, lds); if (rc) return rc; }
    TimedLaunch tl(c, s);
    hipLaunchKernelGGL(kern, dim3(B, NH), dim3(256), lds, s, qkv, O, T, D, NH, att);
    return BUSCA_OK;
}

template <int PREC, int HD>
static int dtl_attention_mt(busca_ctx* c, hipStream_t s, int MT, const void* qkv, void* O, int B, int T, int D, int NH, float* att) {
    switch (MT) {
        case 1: return dtl_attention<PREC, HD, 1>(c, s, qkv, O, B, T, D, NH, att);
        case 2: return dtl_attention<PREC, HD, 2>(c, s, qkv, O, B, T, D, NH, att);
        case 3: return dtl_attention<PREC, HD, 3>(c, s, qkv, O, B, T, D, NH, att);
        case 4: return dtl_attention<PREC, HD, 4>(c, s, qkv, O, B, T, D, NH, att);
        case 5: return dtl_attention<PREC, HD, 5>(c, s, qkv, O, B, T, D, NH, att);
        case 6: return dtl_attention<PREC, HD, 6>(c, s, qkv, O, B, T, D, NH, att);
        case 7: return dtl_attention<PREC, HD, 7>(c, s, qkv, O, B, T, D, NH, att);
        case 8: return dtl_attention<PREC, HD, 8>(c, s, qkv, O, B, T, D, NH, att);
        case 9: return dtl_attention<PREC, HD, 9>(c, s, qkv, O, B, T, D, NH, att);
    }
    return fail(c, BUSCA_EINVAL, "tiled attention supports at most 144 tokens per track (got %d)", T);
}

// head width HD = d / nhead in {16, 32, 64, 128} (nhead 4 at d = 64 / 256 / 512 are 16 / 64 / 128)
template <int PREC>
static int dtl_attention_hd(busca_ctx* c, hipStream_t s, int MT, const void* qkv, void* O, int B, int T, int D, int NH, float* att) {
    switch (D / NH) {
        case 16: return dtl_attention_mt<PREC, 16>(c, s, MT, qkv, O, B, T, D, NH, att);
        case 32: return dtl_attention_mt<PREC, 32>(c, s, MT, qkv, O, B, T, D, NH, att);
        case 64: return dtl_attention_mt<PREC, 64>(c, s, MT, qkv, O, B, T, D, NH, att);
        case 128: return dtl_attention_mt<PREC, 128>(c, s, MT, qkv, O, B, T, D, NH, att);
    }
    return fail(c, BUSCA_EINVAL, "tiled attention: head width %d not built (16, 32, 64, 128)", D / NH);
}

// QKV projection + attention of a (track, head) in one kernel (dtl_qkv_attn_kernel): built for the shipped head geometry (four heads: d = 512 /
// 128-wide, d = 256 / 64-wide) and the token counts the one-kernel path cannot hold.  Returns false when this shape is not built.
template <int PREC, int D, int HD, int MT>
static int dtl_qkv_attn_launch(busca_ctx* c, hipStream_t s, const DTLQkvAttnArgs& a, int B) {
    typedef typename AttPrec<PREC>::type AP;         // (x3: the attention tiles are the f32 kernel's)
    constexpr int ES = Prec<PREC>::ES, AES = AP::ES, TP = 16 * MT, TPK = AP::CHUNK * AP::nchunks(MT);
    constexpr size_t ga = (size_t)TP * ((D / 2) * ES + 16), at = (size_t)2 * TP * (HD * AES + 16) + (size_t)HD * (TPK * AES + 16);
    constexpr size_t lds = ga > at ? ga : at;
    static_assert(lds <= 160 * 1024, "fused QKV + attention: LDS plan");
    auto kern = dtl_qkv_attn_kernel<PREC, D, HD, MT>;
    { int rc = ensure_lds(c, (const void*)kern, lds); if (rc) return rc; }
#ifdef BUSCA_CONV_PROBE
    if (getenv("BUSCA_QA_TS")) {        // phase stamps (probe build): wave 0 of every workgroup, 100 MHz s_memtime ticks summed over the launch
        static unsigned long long* ts = nullptr;
        if (!ts) hipMalloc((void**)&ts, 8 * 8);
        hipMemsetAsync(ts, 0, 64, s);
        DTLQkvAttnArgs b = a; b.ts = ts;
        hipLaunchKernelGGL(kern, dim3((unsigned)(a.NH * 8 * ((B + 7) / 8))), dim3(4 * HD), lds, s, b);
        hipStreamSynchronize(s);
        unsigned long long h[8]; hipMemcpy(h, ts, 64, hipMemcpyDeviceToHost);
        const double n = (double)B * a.NH; double tot = 0; for (int k = 0; k < 7; ++k) tot += (double)h[k];
        fprintf(stderr, "[qa_ts] %d workgroups, mean lifetime of wave 0: %.0f ticks | stage half 0 %.0f, GEMM half 0 %.0f, stage half 1 %.0f, GEMM half 1 %.0f, barrier %.0f, park Q K V^T %.0f, attention + O %.0f\n",
                B * a.NH, tot / n, h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5] / n, h[6] / n);
        return BUSCA_OK;
    }
#endif
    TimedLaunch tl(c, s);
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.NH * 8 * ((B + 7) / 8))), dim3(4 * HD), lds, s, a);      // 1-D: the kernel maps a workgroup to (track, head) XCD-aware
    return BUSCA_OK;
}
template <int PREC, int D>
static bool dtl_qkv_attn(busca_ctx* c, hipStream_t s, const DTLQkvAttnArgs& a, int B, int MT, int* rc) {
    constexpr int HD = D / 4;
    *rc = BUSCA_OK;
    if (D < 256 || a.NH != 4) return false;
#define QA(M_) case M_: *rc = dtl_qkv_attn_launch<PREC, (D >= 256 ? D : 256), (D >= 256 ? HD : 64), M_>(c, s, a, B); return true
    if constexpr (PREC == 1) { switch (MT) { QA(5); QA(6); QA(7); QA(8); QA(9); } }
    else { switch (MT) { QA(3); QA(4); QA(5); } }
#undef QA
    return false;
}

template <int PREC, int D>
static int dt_forward_tiled(busca_ctx* c, const DTParams& K, hipStream_t s) {
    DTState& S = c->dt;
    constexpr size_t ES = Prec<PREC>::ES;
    // GP: arithmetic of the generic GEMM / attention kernels (embed, geometries the fused layer kernels are not built for).  x3 has the two fused layer
    // kernels only (dtl_qkv_attn_kernel<2>, dtl_ffn_kernel<2>: split-fp16 products, float32-equivalent); everything else of an x3 forward runs the exact
    // f32 kernels on the row-major f32 matrices (S.tw) - never less than float32-equivalent.
    constexpr int GP = PREC == 2 ? 0 : PREC;
    const int T = K.T, B = K.B, L = K.L, P = K.P, FF = S.cfg.ff, NH = S.cfg.nhead, E = 512;
    const int MT = (T + 15) / 16;
    if (MT > 9) return fail(c, BUSCA_EINVAL, "tiled DT path supports at most 144 tokens per track (T=%d)", T);
    if (P + K.nspec > 128) return fail(c, BUSCA_EINVAL, "tiled DT path supports at most %d proposals (P=%d)", 128 - K.nspec, P);
    const size_t M = (size_t)B * T;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    { int rc = dtl_ws_ensure(c, dtl_ws_bytes(M, D, FF, ES), s); if (rc) return rc; }
    char* p = (char*)S.ws;
    float* X = (float*)p; p += al(M * D * 4);
    _Float16* Xh = (_Float16*)p; p += al(M * D * 2);
    char* QKV = p; p += al(M * 3 * D * ES);
    char* O = p; p += al(M * D * ES);
    char* H = p; p += al(M * FF * ES);
    int* ids = (int*)p; p += al(M * 3 * 4);
    float* rowlogit = (float*)p;                          // [M] decoder logit of every row (last layer kernel) -> dtl_decoder_rows_kernel
    bool dec_fused = false;
    const void* Xop = PREC != 1 ? (const void*)X : (const void*)Xh;      // GEMM operand copy of the residual stream
    dt_bucket_ids_launch(c, s, K.mem_ltrb, K.can_ltrb, B, L, P, K.fake_f64, K.can_pos, K.nspec, K.sep_can, ids);
    DTLArgs a{};
    a.M = (int)M; a.L = L; a.P = P; a.T = T; a.E = E; a.can_pos = K.can_pos; a.X = X; a.Xh = Xh; a.act = K.act;
    a.qscale = 1.0f / sqrtf((float)(D / NH));
    // embed + assembly + encoding
    a.W = S.tw.w_embed; a.K = E; a.bias = K.b_embed; a.mem_feat = K.mem_feat; a.can_feat = K.can_feat; a.ids = ids;
    a.lut_xy = K.lut_xy; a.lut_sz = K.lut_sz; a.lut_t = K.lut_t; a.lut_c = K.lut_c;
    a.tok_sep = K.tok_sep; a.tok_non = K.tok_non; a.tok_bad = K.tok_bad;
    a.Me = B * (L + P);
    a.skip_x32 = (PREC == 1 && D >= 256 && c->opt.dtl_ffn == 2) ? 1 : 0;       // every layer then runs dtl_ffn_kernel<.., OUTPROJ>, which carries the stream in Xh
    if constexpr (D >= 256) {
        // streamed-weight embed kernel in the layer kernels' arithmetic (x3: split-fp16 products; the generic GEMM would run exact f32)
        typedef DTLEmbedGeom<PREC, D> EG;
        DTLEmbedArgs e{};
        e.mem_feat = K.mem_feat; e.can_feat = K.can_feat; e.L = L; e.P = P; e.T = T; e.E = E; e.can_pos = K.can_pos; e.M = (int)M; e.Me = a.Me;
        e.w = K.w_embed; e.bias = K.b_embed; e.ids = ids; e.lut_xy = K.lut_xy; e.lut_sz = K.lut_sz; e.lut_t = K.lut_t; e.lut_c = K.lut_c;
        e.tok_sep = K.tok_sep; e.tok_non = K.tok_non; e.tok_bad = K.tok_bad; e.X = X; e.Xh = Xh; e.skip_x32 = a.skip_x32; e.xerr = K.xerr;
        auto kern = dtl_embed_kernel<PREC, D>;
        { int rc = ensure_lds(c, (const void*)kern, EG::LDS); if (rc) return rc; }
        const int nbx = (e.Me + EG::BM - 1) / EG::BM + (e.M - e.Me + EG::BM - 1) / EG::BM;
        TimedLaunch tl(c, s);
        hipLaunchKernelGGL(kern, dim3((unsigned)nbx), dim3(512), EG::LDS, s, e);
    } else
    { int rc = dtl_gemm<GP, D, DTL_EPI_EMBED>(c, s, a, 1); if (rc) return rc; }
    for (int l = 0; l < K.nlayers; ++l) {
        const DTLayerW& W = K.layer[l];
        float* att = K.att ? K.att + (size_t)l * B * NH * T * T : nullptr;
        bool fused_attn = false;
        if (c->opt.dtl_attn != 0) {
            DTLQkvAttnArgs q{};
            q.Xop = Xop; q.w_in = W.w_in; q.b_in = W.b_in; q.O = O; q.att = att; q.T = T; q.NH = NH; q.qscale = a.qscale; q.xerr = K.xerr; q.B = B;
            int rc = BUSCA_OK;
            fused_attn = dtl_qkv_attn<PREC, D>(c, s, q, B, MT, &rc);
            if (rc) return rc;
        }
        if (!fused_attn) {
            a.A = Xop; a.lda = D; a.W = S.tw.w_in[l]; a.K = D; a.bias = W.b_in; a.out16 = QKV; a.ldo = 3 * D;
            {
                int rc = dtl_gemm<GP, D, DTL_EPI_QKV>(c, s, a, 3);
                if (rc) return rc;
            }
            { int rc = dtl_attention_hd<GP>(c, s, MT, QKV, O, B, T, D, NH, att); if (rc) return rc; }
        }
        const int ffn_mode = D >= 256 ? c->opt.dtl_ffn : 0;   // 2: out-proj + norm1 + feed-forward + norm2 in one kernel; 1: feed-forward block only; 0: layer-wise GEMMs
        if (ffn_mode != 2) {
            a.A = O; a.lda = D; a.W = S.tw.w_out[l]; a.K = D; a.bias = W.b_out; a.gamma = W.g1; a.beta = W.be1;
            { int rc = dtl_gemm<GP, D, DTL_EPI_RESLN>(c, s, a, 1); if (rc) return rc; }
        }
        if (ffn_mode != 0) {
            // the row-local half of the layer in one kernel: x1 (mode 2) and H never reach HBM (dtl_ffn_kernel)
            DTLFfnArgs f{};
            f.Xop = Xop; f.Oop = O; f.X = X; f.Xh = Xh; f.w_out = W.w_out; f.w1 = W.w1; f.w2 = W.w2; f.b_out = W.b_out; f.g1 = W.g1; f.be1 = W.be1;
            f.b1 = W.b1; f.b2 = W.b2; f.gamma = W.g2; f.beta = W.be2; f.M = (int)M; f.FF = FF; f.act = K.act;
            f.xerr = K.xerr;
            const bool last = l == K.nlayers - 1;
            if (last && ffn_mode == 2) { f.dec_g = K.dec_g; f.dec_w = K.dec_w; f.dec_cb = K.dec_cb; f.dec_logit = rowlogit; dec_fused = true; }
            f.write_x32 = last && K.hidden != nullptr;   // (f16 flavour: the float32 copy of the residual stream only where `hidden` reads it - the decoder runs in this kernel)
            constexpr int DK = D >= 256 ? D : 256;
            constexpr int BMF = DTLFfnGeom<PREC, DK>::BM;
            constexpr size_t flds = DTLFfnGeom<PREC, DK>::LDS;
#ifdef BUSCA_CONV_PROBE
            if (ffn_mode == 2 && getenv("BUSCA_FFN_TS")) {      // phase stamps (probe build): per-wave cycle sums of the first 1 024 workgroups
                static unsigned long long* ts = nullptr;
                const int NWVp = DTLFfnGeom<PREC, DK>::NWV;
                if (!ts) hipMalloc((void**)&ts, (size_t)1024 * 8 * 8 * 8);
                hipMemsetAsync(ts, 0, (size_t)1024 * 8 * 8 * 8, s);
                f.ts = ts; f.dbg_h = nullptr;
                auto kern = dtl_ffn_kernel<PREC, DK, true>;
                { int rc = ensure_lds(c, (const void*)kern, flds); if (rc) return rc; }
                const unsigned nb = (unsigned)((M + BMF - 1) / BMF);
                hipLaunchKernelGGL(kern, dim3(nb), dim3(64 * NWVp), flds, s, f);
                hipStreamSynchronize(s);
                std::vector<unsigned long long> hts((size_t)1024 * 8 * 8);
                hipMemcpy(hts.data(), ts, hts.size() * 8, hipMemcpyDeviceToHost);
                double ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const int nbs = nb < 1024 ? (int)nb : 1024;
                for (int b = 0; b < nbs; ++b) for (int w = 0; w < NWVp; ++w) for (int k = 0; k < 8; ++k) ph[k] += (double)hts[((size_t)b * NWVp + w) * 8 + k];
                double tot = 0; for (int k = 0; k < 8; ++k) { ph[k] /= (double)nbs * NWVp; tot += ph[k]; }
                fprintf(stderr, "[ffn_ts] %u workgroups x %d waves, mean wave lifetime %.0f cycles | stage rows %.0f, out-proj GEMM %.0f, residual + LN1 + x1 %.0f, FFN1 GEMMs %.0f, activation + barrier %.0f, FFN2 GEMMs %.0f, block barrier %.0f, residual + LN2 + stores %.0f\n",
                        nb, NWVp, tot, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7]);
                continue;
            }
#endif
            TimedLaunch tl(c, s);
            if (ffn_mode == 2) {
                auto kern = dtl_ffn_kernel<PREC, DK, true>;
                { int rc = ensure_lds(c, (const void*)kern, flds); if (rc) return rc; }
                hipLaunchKernelGGL(kern, dim3((unsigned)((M + BMF - 1) / BMF)), dim3(64 * DTLFfnGeom<PREC, DK>::NWV), flds, s, f);
            } else {
                auto kern = dtl_ffn_kernel<PREC, DK, false>;
                { int rc = ensure_lds(c, (const void*)kern, flds); if (rc) return rc; }
                hipLaunchKernelGGL(kern, dim3((unsigned)((M + BMF - 1) / BMF)), dim3(64 * DTLFfnGeom<PREC, DK>::NWV), flds, s, f);
            }
            continue;
        }
        a.A = Xop; a.lda = D; a.W = S.tw.w1[l]; a.K = D; a.bias = W.b1; a.out16 = H; a.ldo = FF;
        {
            int rc = dtl_gemm<GP, D, DTL_EPI_FFN1>(c, s, a, FF / D);
            if (rc) return rc;
        }
        a.A = H; a.lda = FF; a.W = S.tw.w2[l]; a.K = FF; a.bias = W.b2; a.gamma = W.g2; a.beta = W.be2;
        { int rc = dtl_gemm<GP, D, DTL_EPI_RESLN>(c, s, a, 1); if (rc) return rc; }
    }
    if (K.hidden) HIP_TRY(c, hipMemcpyAsync(K.hidden, X, M * D * sizeof(float), hipMemcpyDeviceToDevice, s));
    { TimedLaunch tl(c, s);
      if (dec_fused) hipLaunchKernelGGL((dtl_decoder_rows_kernel<0>), dim3((B + 3) / 4), dim3(256), 0, s, (const float*)rowlogit, B, T, L, P, K.can_pos, K.nspec, K.logits, K.probs, K.argmax);
      else hipLaunchKernelGGL((dtl_decoder_kernel<D>), dim3(B), dim3(256), 0, s, (const float*)X, T, L, P, K.can_pos, K.nspec, K.dec_g, K.dec_b, K.dec_w, K.dec_bias,
                              K.logits, K.probs, K.argmax); }
    HIP_TRY(c, hipGetLastError());
    return BUSCA_OK;
}

