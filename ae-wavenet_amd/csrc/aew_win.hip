// aew_win.hip — bf16 NT kernel with ONE LDS window for both taps of a dilated convolution (gfx950).
//
// wavenet.py:100-101 reads x[t] and x[t+d] of the SAME tensor; its input gradient reads dfg[t] and dfg[t-d].
// k_gemm_nt_bf16 treats the two taps as two unrelated K segments and stages 256 rows of each: 2 x 24 KiB of
// operand tiles per pair of MFMA K steps.  Here, when the first two segments of a descriptor are the same
// buffer at row offsets that differ by d <= 64, a K step of the pair is staged ONCE:
//
//     window   BM + 16*DWP rows x 32 channels of x           (DWP = 1: d <= 16, DWP = 4: d <= 64)
//     W tap 0  128 rows x 32        W tap 1  128 rows x 32
//
// and BOTH taps' MFMAs are issued from it - tap s reads fragment rows r + shift_s of the window, with that row's
// swizzle.  Per window step and block that is 33-36 KiB staged for 32 MFMAs per wave instead of 49 KiB, and one
// barrier instead of two.  Segments after the pair (the conditioning projection) are plain K steps of 32.
//
// Same tiles (256 x 128, or 192 x 128 where the launcher's cost model prefers them: MT = 4 | 3 MFMA row tiles per
// wave), wave layout (8 waves as 4 x 2), staging permutation, swizzle, XCD-aware tile order and epilogues as
// k_gemm_nt_bf16<EPI, false, MT>; the LDS ring has two stages of <= 36 KiB (two blocks per CU), the K loop waits for
// all of a step's LDS-DMA (no counted vmcnt: the prefetch distance of one window step equals the two plain steps of
// the 3-stage ring).  Accumulation order differs from the two-segment kernel (the taps interleave per K tile), so
// results agree with it to fp32 rounding, not bit for bit.
#pragma once

template <int MT, int DWP>
struct WinCfg {
    static constexpr int BM = 64 * MT;                       // 4 waves along m, MT 16-row MFMA tiles each
    static constexpr int NXP = BM / 16 + DWP;                // 16-row window pieces per step
    static constexpr int NFULL = NXP / 8;                    // pieces wave + 8j, j < NFULL: every wave has them
    static constexpr int REM = NXP % 8;                      // piece wave + 8*NFULL: waves < REM only (the "tail")
    static constexpr int PFULL = (BM / 16) / 8, PREM = (BM / 16) % 8;   // same for a plain step (no halo rows)
    static constexpr int W_OFF = NXP * 1024;                 // tap 0 W rows; tap 1 follows 8 KiB later
    static constexpr int STAGE_BYTES = W_OFF + 2 * NT_BN * NT_ROWB;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
    static_assert(NFULL >= 1 && NFULL <= 2 && (NFULL < 2 || REM <= 8) && NXP <= 24, "three pieces per wave at most");
};

// per-lane source pointers of the pieces a wave stages: window pieces wave, wave + 8, wave + 16 (those that exist),
// W piece `wave` of tap 0; tap 1's W piece is w + wdelta
struct WinPtrs {
    const char *x0, *x1, *x2, *w;
    int wdelta;                                              // wave-uniform
    int seg, left;                                           // walk over the segment table: `left` steps remain
};

// One step's operand tiles go out as the pieces every wave issues (main) plus the piece only the first REM waves
// have (tail: a wave-uniform branch, kept out of the K loop's main basic block so that the scheduler directives there
// can thread the other pieces between the MFMAs).
// (SC1X: the activation pieces are read device-scope - stage of a chained launch, see glds16_sc1)
template <int MT, int DWP, bool WINDOW, bool SC1X = false>
__device__ __forceinline__ void win_issue_tail(char* stage, int wave, WinPtrs& P) {
    typedef WinCfg<MT, DWP> Cfg;
    constexpr int nfull = WINDOW ? Cfg::NFULL : Cfg::PFULL, rem = WINDOW ? Cfg::REM : Cfg::PREM;
    if (rem > 0 && wave < rem) {
        if (nfull == 1) { glds16_x<SC1X>(P.x1, stage + (wave + 8) * 1024); P.x1 += NT_BK * 2; }
        else { glds16_x<SC1X>(P.x2, stage + (wave + 16) * 1024); P.x2 += NT_BK * 2; }
    }
}
template <int MT, int DWP, bool WINDOW, bool SC1X = false>
__device__ __forceinline__ void win_issue_main(char* stage, int wave, WinPtrs& P) {
    typedef WinCfg<MT, DWP> Cfg;
    constexpr int nfull = WINDOW ? Cfg::NFULL : Cfg::PFULL;
    glds16_x<SC1X>(P.x0, stage + wave * 1024);
    if (nfull == 2) glds16_x<SC1X>(P.x1, stage + (wave + 8) * 1024);
    glds16(P.w, stage + Cfg::W_OFF + wave * 1024);
    if (WINDOW) glds16(P.w + P.wdelta, stage + Cfg::W_OFF + NT_BN * NT_ROWB + wave * 1024);
    P.x0 += NT_BK * 2; P.w += NT_BK * 2;
    if (nfull == 2) P.x1 += NT_BK * 2;
}
constexpr int win_main_pieces(int MT, int DWP, bool window) {
    return 2 + (window ? 1 : 0) + (((window ? (4 * MT + DWP) / 8 : (4 * MT) / 8) == 2) ? 1 : 0);
}

// after a step has been issued: point P at the next one (wave-uniform, rare: scalar loads only here).  The step
// after the last window step is issued by the window loop with all its pieces: tap 1's W piece then repeats
// tap 0's (wdelta = 0) and pieces the plain step does not have come from the zero region; nobody reads them.
template <int MT>
__device__ __forceinline__ void win_advance(const aew_gemm_nt_t& g, int b, int m0, int wave, int lane, int klen, WinPtrs& P) {
    if (--P.left == 0) {
        const char* const zeros = reinterpret_cast<const char*>(aew_zero_region);
        if (P.seg == 1) P.w += klen * 2;                     // skip tap 1's columns
        ++P.seg;
        P.wdelta = 0;
        P.x2 = zeros;
        // (new values first, one set of assignments after the branch: stores to different members at the ends of the
        // two arms get merged into one store at a variable address, which sends the whole struct to scratch)
        const char *nx0 = zeros, *nx1 = zeros, *nw = zeros;
        int nleft = 1 << 30;                                 // K exhausted: keep the loops uniform, issue zeros
        if (P.seg < g.n_segs) {
            const aew_seg_t s = g.seg[P.seg];
            const int lr = lane >> 2, pc = lane & 3;
            bool ok0, ok1;
            const int r0 = wave * 16 + lr, r1 = r0 + 128;
            const char* s0 = seg_row_ptr_sel(s, b, m0 + r0, 2, ok0) + (nt_swz64(r0, pc) << 4);
            const char* s1 = seg_row_ptr_sel(s, b, m0 + r1, 2, ok1) + (nt_swz64(r1, pc) << 4);
            nx0 = ok0 ? s0 : zeros;
            nx1 = (ok1 && r1 < 64 * MT) ? s1 : zeros;
            nw = P.w;
            nleft = s.k_len / NT_BK;
        }
        P.x0 = nx0; P.x1 = nx1; P.w = nw; P.left = nleft;
    }
}

template <int MT, int DWP>
__device__ __forceinline__ const char* win_src(const aew_seg_t& sref, int b, int m0, int offmin, int piece, int lane) {
    const aew_seg_t s = sref;
    const int lr = lane >> 2, pc = lane & 3;
    const int r = piece * 16 + lr;
    const int64_t row = (int64_t)m0 + r + offmin;            // LDS row r of the window <- source row m0 + r + offmin
    const bool ok = row >= s.row_lo && row < s.row_hi && piece < WinCfg<MT, DWP>::NXP;
    const char* src = reinterpret_cast<const char*>(s.ptr) + ((int64_t)b * s.batch_stride + row * s.row_pitch) * 2 +
                      (nt_swz64(r, pc) << 4);
    return ok ? src : reinterpret_cast<const char*>(aew_zero_region);   // rows that do not exist stream zeros
}

// Measured and dropped (profiles/r03_notes.md): touching the window's L2 lines four K steps ahead (+10 us per launch:
// the loop is not waiting on latency), starting the blocks in odd thread-group slots of a CU a few us late so that the
// two resident blocks run out of phase (null to slightly worse).
// one output tile: block index L of the launch's tile order (WT: out0 stored write-through, see nt_epilogue)
template <int EPI, int MT, int DWP, bool WT = false>
__device__ __forceinline__ void win_tile(const aew_gemm_nt_t& g, char* smem, const int L) {
    typedef WinCfg<MT, DWP> Cfg;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;
    // tile order as in k_gemm_nt_bf16: the N tiles of one (batch, row tile) are consecutive on one XCD
    const int n_mt = (g.M + Cfg::BM - 1) / Cfg::BM, n_nt = g.N_pad / NT_BN;
    const int seq = L >> 3;
    const int rt = (seq / n_nt) * 8 + (L & 7);
    if (rt >= n_mt * g.batch) return;
    const int b = rt / n_mt;
    const int m0 = (rt - b * n_mt) * Cfg::BM, n0 = (seq % n_nt) * NT_BN;

    const int off0 = g.seg[0].row_off, off1 = g.seg[1].row_off;
    const int offmin = off0 < off1 ? off0 : off1;
    const int sh0 = off0 - offmin, sh1 = off1 - offmin;      // LDS row shift of each tap inside the window
    const int klen = g.seg[0].k_len;
    const int kw = klen / NT_BK;                             // window steps
    const int nsteps = kw + (g.K_total - 2 * klen) / NT_BK;

    f32x4_t acc[4][MT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    WinPtrs P;
    P.x0 = win_src<MT, DWP>(g.seg[0], b, m0, offmin, wave, lane);
    P.x1 = win_src<MT, DWP>(g.seg[0], b, m0, offmin, wave + 8, lane);
    P.x2 = win_src<MT, DWP>(g.seg[0], b, m0, offmin, wave + 16, lane);
    {   // W rows 16*wave .. +15 of the 128-row tile, staged in the epilogue's lane order (nt_wperm)
        const int lr = lane >> 2, pc = lane & 3;
        const int r = wave * 16 + lr;
        const int src_row = (r & ~63) + nt_wperm<EPI>(r & 63);
        P.w = reinterpret_cast<const char*>(g.W) + (int64_t)(n0 + src_row) * g.K_total * 2 + (nt_swz64(r, pc) << 4);
    }
    P.wdelta = klen * 2;                                     // tap 1's W columns start klen after tap 0's
    P.seg = 1; P.left = kw;

    const int fi = lane & 15, fg = lane >> 4;
    // fragment byte offsets inside a stage; MFMA tile i / j adds i * 1024 (the swizzle has period 16 rows)
    const int rw = wn * 64 + fi;
    const int woff = Cfg::W_OFF + rw * NT_ROWB + (nt_swz64(rw, fg) << 4);
    const int rx = wm * (16 * MT) + fi;
    const int xoff0 = (rx + sh0) * NT_ROWB + (nt_swz64(rx + sh0, fg) << 4);
    const int xoff1 = (rx + sh1) * NT_ROWB + (nt_swz64(rx + sh1, fg) << 4);
    const int xoffs = rx * NT_ROWB + (nt_swz64(rx, fg) << 4);

    win_issue_tail<MT, DWP, true, WT>(smem, wave, P);
    win_issue_main<MT, DWP, true, WT>(smem, wave, P);
    win_advance<MT>(g, b, m0, wave, lane, klen, P);
    int t = 0;
    for (; t < kw; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // step t has landed (this wave's pieces) ...
        __builtin_amdgcn_s_barrier();                        // ... everybody's; and everybody is past step t-1
        asm volatile("" ::: "memory");
        const char* st = smem + (t & 1) * Cfg::STAGE_BYTES;
        char* nst = smem + ((t + 1) & 1) * Cfg::STAGE_BYTES;
        win_issue_tail<MT, DWP, true, WT>(nst, wave, P);
        {
            bf16x8_t wf[4], xf[MT], wg[4], xg[MT];
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8_t*>(st + woff + i * 1024);
#pragma unroll
            for (int j = 0; j < MT; ++j) xf[j] = *reinterpret_cast<const bf16x8_t*>(st + xoff0 + j * 1024);
            win_issue_main<MT, DWP, true, WT>(nst, wave, P);
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) wg[i] = *reinterpret_cast<const bf16x8_t*>(st + woff + NT_BN * NT_ROWB + i * 1024);
#pragma unroll
            for (int j = 0; j < MT; ++j) xg[j] = *reinterpret_cast<const bf16x8_t*>(st + xoff1 + j * 1024);
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wg[i], xg[j], acc[i][j], 0, 0, 0);
            // shape of the step: tap 0's fragment reads, its MFMAs with one LDS-DMA piece of step t+1 after every
            // MT of them, tap 1's reads, its MFMAs
            constexpr int NP = win_main_pieces(MT, DWP, true);
            __builtin_amdgcn_sched_group_barrier(0x100, 4 + MT, 0);            // DS reads
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, MT, 0);            // MFMA
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);             // VMEM (LDS-DMA)
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT - NP * MT, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4 + MT, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT, 0);
        }
        win_advance<MT>(g, b, m0, wave, lane, klen, P);
    }
    for (; t < nsteps; ++t) {                                // plain K steps of the segments after the pair
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* st = smem + (t & 1) * Cfg::STAGE_BYTES;
        char* nst = smem + ((t + 1) & 1) * Cfg::STAGE_BYTES;
        win_issue_tail<MT, DWP, false, WT>(nst, wave, P);
        bf16x8_t wf[4], xf[MT];
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8_t*>(st + woff + i * 1024);
#pragma unroll
        for (int j = 0; j < MT; ++j) xf[j] = *reinterpret_cast<const bf16x8_t*>(st + xoffs + j * 1024);
        win_issue_main<MT, DWP, false, WT>(nst, wave, P);
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        constexpr int NP = win_main_pieces(MT, DWP, false);
        __builtin_amdgcn_sched_group_barrier(0x100, 4 + MT, 0);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, MT, 0);
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT - NP * MT, 0);
        win_advance<MT>(g, b, m0, wave, lane, klen, P);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the idle-tail LDS-DMA must land before the LDS is released
    nt_epilogue<EPI, false, MT, WT>(g, acc, b, m0, n0, wm, wn, lane);
}

template <int EPI, int MT, int DWP>
__global__ __launch_bounds__(512, 4) void k_gemm_nt_bf16_win(const aew_gemm_nt_t g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    win_tile<EPI, MT, DWP>(g, smem, blockIdx.x);
}

// ---- host side ----------------------------------------------------------------------------------------------

// window pieces beyond the tile's own that the descriptor needs (1 | 4), or 0 if it cannot use the window kernel
static int win_dwp(const aew_gemm_nt_t& g) {
    if (!AEW_T().nt_window || g.dtype != AEW_BF16 || g.impl != 0 || g.n_segs < 2 || g.W2) return 0;
    if (g.epi != AEW_EPI_GATED && g.epi != AEW_EPI_STORE) return 0;
    const aew_seg_t &a = g.seg[0], &c = g.seg[1];
    if (a.ptr != c.ptr || a.batch_stride != c.batch_stride || a.row_pitch != c.row_pitch || a.row_step != 1 ||
        c.row_step != 1 || a.row_lo != c.row_lo || a.row_hi != c.row_hi || a.k_len != c.k_len)
        return 0;
    const int d = a.row_off > c.row_off ? a.row_off - c.row_off : c.row_off - a.row_off;
    if (d < 1 || d > 64 || d > AEW_T().nt_window) return 0;
    for (int s = 0; s < g.n_segs; ++s)
        if (g.seg[s].k_len % NT_BK || g.seg[s].k_len * 2 > AEW_ZERO_SPAN) return 0;
    return d <= 16 ? 1 : 4;
}

template <int EPI, int MT, int DWP>
static int win_launch(const aew_gemm_nt_t& g, hipStream_t st) {
    static std::atomic<unsigned long long> attr_done{0};          // one bit per device (function attributes are per device)
    typedef WinCfg<MT, DWP> Cfg;
    constexpr int lds = Cfg::LDS_BYTES;
    static_assert(2 * lds <= 160 * 1024, "two blocks per CU");
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {      // (racing threads both set it: harmless)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_nt_bf16_win<EPI, MT, DWP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    const int row_tiles = ((g.M + Cfg::BM - 1) / Cfg::BM) * g.batch;
    dim3 grid(((row_tiles + 7) / 8) * 8 * (g.N_pad / NT_BN));
    hipLaunchKernelGGL((k_gemm_nt_bf16_win<EPI, MT, DWP>), grid, dim3(512), lds, st, g);
    return (int)hipGetLastError();
}

// t192: 192-row tiles (the launcher's per-CU cost model); they always take the 256-row window (DWP = 4: 16 pieces,
// two per wave)
static int launch_win(const aew_gemm_nt_t& g, int dwp, bool t192, hipStream_t st) {
    if (g.epi == AEW_EPI_GATED) {
        if (t192) return win_launch<AEW_EPI_GATED, 3, 4>(g, st);
        return dwp == 1 ? win_launch<AEW_EPI_GATED, 4, 1>(g, st) : win_launch<AEW_EPI_GATED, 4, 4>(g, st);
    }
    if (t192) return win_launch<AEW_EPI_STORE, 3, 4>(g, st);
    return dwp == 1 ? win_launch<AEW_EPI_STORE, 4, 1>(g, st) : win_launch<AEW_EPI_STORE, 4, 4>(g, st);
}
