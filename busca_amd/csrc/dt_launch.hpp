// Host side of the fused Decision-Transformer kernel: launch geometry, the token-split tail, LDS limits.  Included by the units that instantiate the kernel
// (busca_dt_f32 / _f16 / _x3.hip), after dt_kernel.hip.inc.
#pragma once
#include "busca_internal.hpp"

// ---------------------------------------------------------------------------------------------------------
// How many of the last tracks of a launch run token-split (one workgroup per 16-token tile, `parts` per track), and how many tracks share a workgroup
// (*pair: 1 or 2).  The tracks of the last, partial round of one-track workgroups are spread over the CUs that round would leave idle: a single 32-track
// step of three tiles runs on 96 CUs instead of 32 (0.46 -> 0.22 ms); 640 such tracks on 256 CUs are two rounds + 192 workgroups holding one tile index
// of two tracks each (0.35 ms) instead of three rounds.  One track per workgroup is the faster flavour while its workgroups fit ONE pass over the CUs
// (a one-tile workgroup is bound by its weight stream through the CU's vector memory path: two to a CU take twice as long); beyond that two tracks
// share a workgroup, every streamed weight fragment feeding two tiles (f32 flavour, tracks of three tiles or more - with two tiles such a workgroup
// would do a whole track's work).  A partial round too large for either stays one workgroup per track.
static int dt_split_tracks(const busca_ctx* c, int B, int parts, int prec, bool can_pair, int* pair) {
    const DTState& S = c->dt;
    *pair = 1;
    if (S.xslots <= 0 || c->opt.dt_split == 0 || parts > DT_XMAX_MT) return 0;
    if (c->opt.dt_split > 0) {                // tests: 1 = one track per workgroup, 2 = two
        *pair = (c->opt.dt_split == 2 && can_pair) ? 2 : 1;
        return std::min(B, S.xslots / 2);
    }
    // (f16 is left alone: that kernel is bound by the weight stream, which every workgroup of a split track repeats - measured 0.083 vs 0.085 ms for a
    // 32-track step, slower from one pass of workgroups on)
    if (prec == BUSCA_PREC_F16) return 0;
    const int rem = B % S.num_cu;
    if (rem == 0 || rem + 1 > S.xslots) return 0;
    // x3 is bound by the weight stream (L1 / L2), not by the MFMA: a partial round already runs faster than a full one (640 tracks 0.607 ms against
    // 3 x 0.225) and split workgroups add weight traffic - they pay only while the launch is small (32-track step 0.183 -> 0.134 ms; 128 tracks of two
    // tiles at d = 512: 0.407 -> 0.430)
    if (prec == BUSCA_PREC_F16X3) return 2 * parts * rem <= S.num_cu ? rem : 0;
    if (parts * rem <= S.num_cu) return rem;
    if (can_pair && parts * ((rem + 1) / 2) <= S.num_cu) { *pair = 2; return rem; }
    return 0;
}

static int dt_prof_report(busca_ctx* c, long long* d, int nwg, hipStream_t s) {
    HIP_TRY(c, hipStreamSynchronize(s));
    long long h[4 * DT_PROF_SLOTS];
    HIP_TRY(c, hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipFree(d));
    fprintf(stderr, "DT_PROF grid=%d:", nwg);
    for (int w = 0; w < 4; ++w) {
        fprintf(stderr, "\n w%d", w);
        for (int i = 1; i < DT_PROF_SLOTS; ++i)
            if (h[w * DT_PROF_SLOTS + i]) fprintf(stderr, " %d:%lld", i, h[w * DT_PROF_SLOTS + i] - h[w * DT_PROF_SLOTS]);
    }
    fprintf(stderr, "\n");
    return BUSCA_OK;
}

template <int PREC, int MT, int D, int FF, int NCH, int NTRK>
static int dt_launch_split(busca_ctx* c, const DTParams& P0, int nsplit, hipStream_t s) {
    typedef DTLds<PREC, 1, D, FF, 512, NCH, NTRK> LD;
    DTState& S = c->dt;
    auto kern = dt_fused_kernel<PREC, MT, D, FF, 512, NCH, NTRK, true>;
    const int nwg = ((nsplit + NTRK - 1) / NTRK) * MT;
    { int rc = dt_split_ensure(c); if (rc) return rc; }
    DTParams P = P0;
    P.nsingle = P.B - nsplit;
    P.xepoch = ++S.xepoch; P.xch = (unsigned long long*)S.xch; P.xflag = S.xflag; P.xlg = S.xlg; P.xerr = S.xerr_dev;
    { int rc = ensure_lds(c, (const void*)kern, LD::TOTAL); if (rc) return rc; }
    if (c->opt.dt_prof == 2) {
        HIP_TRY(c, hipMalloc((void**)&P.prof, 4 * DT_PROF_SLOTS * sizeof(long long)));
        HIP_TRY(c, hipMemset(P.prof, 0, 4 * DT_PROF_SLOTS * sizeof(long long)));
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), LD::TOTAL, s, P);
        return dt_prof_report(c, P.prof, nwg, s);
    }
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), LD::TOTAL, s, P);
    return BUSCA_OK;
}

template <int PREC, int MT, int D, int FF, int NCH, int NTRK = 1>
static int dt_launch(busca_ctx* c, const DTParams& P, hipStream_t s) {
    typedef DTLds<PREC, MT, D, FF, 512, NCH, NTRK> LD;
    static_assert(LD::TOTAL <= 160 * 1024, "LDS plan exceeds the 160 KiB of a CU");
    auto kern = dt_fused_kernel<PREC, MT, D, FF, 512, NCH, NTRK>;
    const int nwg = (P.B + NTRK - 1) / NTRK;
    c->opt.last_dt_grid = nwg; c->opt.last_dt_ntrk = NTRK; c->opt.last_dt_split = 0;
    if constexpr (MT >= 2 && MT <= DT_XMAX_MT && NTRK == 1) {
        constexpr bool PAIR = PREC != 1 && MT >= 3;       // the two-tracks-per-workgroup split flavour: f32 / x3, three tiles or more (with two tiles it would do a whole track's work)
        // (every flavour is configured by the first forward of a shape, whichever it takes: a later launch of another track count must not pay for it)
        if (c->dt.xslots > 0) {
            { int rc = ensure_lds(c, (const void*)dt_fused_kernel<PREC, MT, D, FF, 512, NCH, 1, true>, DTLds<PREC, 1, D, FF, 512, NCH, 1>::TOTAL); if (rc) return rc; }
            if constexpr (PAIR) { int rc = ensure_lds(c, (const void*)dt_fused_kernel<PREC, MT, D, FF, 512, NCH, 2, true>, DTLds<PREC, 1, D, FF, 512, NCH, 2>::TOTAL); if (rc) return rc; }
        }
        { int rc = ensure_lds(c, (const void*)kern, LD::TOTAL); if (rc) return rc; }
        // whole rounds of one-track workgroups, then the tail's tracks one token tile per workgroup: ONE timed region (the step batch), two launches on the stream
        int pair = 1;
        const int nsplit = c->opt.dt_prof == 1 ? 0 : dt_split_tracks(c, P.B, MT, PREC, PAIR, &pair);
        if (nsplit > 0) {
            c->opt.last_dt_grid = P.B - nsplit + ((nsplit + pair - 1) / pair) * MT; c->opt.last_dt_split = nsplit; c->opt.last_dt_ntrk = pair;
            TimedLaunch tl(c, s);
            if (P.B > nsplit) hipLaunchKernelGGL(kern, dim3(P.B - nsplit), dim3(256), LD::TOTAL, s, P);
            int rc = BUSCA_OK;
            if constexpr (PAIR) { if (pair == 2) rc = dt_launch_split<PREC, MT, D, FF, NCH, 2>(c, P, nsplit, s); else rc = dt_launch_split<PREC, MT, D, FF, NCH, 1>(c, P, nsplit, s); }
            else rc = dt_launch_split<PREC, MT, D, FF, NCH, 1>(c, P, nsplit, s);
            if (rc) return rc;
            HIP_TRY(c, hipGetLastError());
            return BUSCA_OK;
        }
    }
    { int rc = ensure_lds(c, (const void*)kern, LD::TOTAL); if (rc) return rc; }
    const bool prof = c->opt.dt_prof == 1;   // debug: phase timestamps of workgroup 0
    if (prof) {
        DTParams Q = P;
        long long* d = nullptr;
        HIP_TRY(c, hipMalloc((void**)&d, 4 * DT_PROF_SLOTS * sizeof(long long)));
        HIP_TRY(c, hipMemset(d, 0, 4 * DT_PROF_SLOTS * sizeof(long long)));
        Q.prof = d;
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), LD::TOTAL, s, Q);
        return dt_prof_report(c, d, nwg, s);
    }
    {
        TimedLaunch tl(c, s);
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), LD::TOTAL, s, P);
    }
    HIP_TRY(c, hipGetLastError());
    return BUSCA_OK;
}

