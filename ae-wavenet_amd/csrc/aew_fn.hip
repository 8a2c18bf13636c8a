// aew_fn.hip — "full-N" NT GEMM kernels with producer / consumer wave specialisation, and the fused gated layer.
//
//   C[b][m][0..N)  = sum_s sum_k A_s[b][m*step_s + off_s][k] * W[n][K_s + k]                      (impl = 2)
//   fused layer    : z, dz/df, dz/dg = gate(C);   x_next[b][m][0..N2) = z[b][m][:] . W2[n2][:] + aux0[b][m][n2]
//                    (wavenet.py:100-109: dilated conv + cond projection -> tanh * sigmoid -> residual 1x1 + add)
//
// One workgroup = 12 waves on one CU:
//   * waves 0-7  CONSUMERS: wave wn owns the 16*NTW output channels [16*NTW*wn, ..) of ALL rows of the tile
//                (R = 16*mt rows, mt <= MTMAX); their only memory instructions are LDS fragment reads, the
//                epilogue's global stores and its aux loads.  They never wait on vmcnt for operand tiles, so the
//                epilogue's stores drain under the next K tiles / the next tile instead of in front of them
//                (vmcnt is one in-order counter for loads AND stores on gfx950).
//   * waves 8-11 LOADERS: stream the operand tiles global -> LDS with 16-byte LDS-DMA in K tiles of 64 (128-byte
//                rows: one full L2 line per row; measured 31 TB/s against 19 TB/s for 64-byte rows,
//                profiles/r02_notes.md), two stages, one K tile ahead of the consumers, ACROSS tile boundaries
//                (the first K tile of the next tile is in flight during the epilogue of the current one).
//   * hand-off: one raw s_barrier per K tile.  A loader waits vmcnt(0) before it arrives (tile t has landed);
//                a consumer arrives when its fragment reads of tile t-1 have been consumed (stage of t-1 is free).
// A block owns a contiguous chunk of rows of one batch element (ceil(rows / (256 / batch)) rounded to 16) and walks
// it in tiles of <= MTMAX*16 rows: every CU gets the same amount of work (no tile-wave tail), and the whole N of a
// row is produced by ONE block, which is what lets the gated pair fuse: the z tile goes to LDS in fragment layout
// and is the operand of the residual GEMM (its weights W2 are streamed by the loaders like W).
//
// Results are bit-identical to k_gemm_nt_bf16 (same MFMA, K ascending, same epilogue arithmetic).
#include "aew_common.h"

#ifndef AEW_FN_ABLATE
#define AEW_FN_ABLATE 0           /* 1: tools/fused_ablate.py, tools/gemm_sweep.py switches in aew_gemm_nt_t.reserved */
#endif
#define FN_ABL(g, bit) (AEW_FN_ABLATE && ((g).reserved & (bit)))
#define FN_NCONS 8
#define FN_NLOAD 4
#define FN_THREADS ((FN_NCONS + FN_NLOAD) * 64)

template <int NTW, int MTCAP = 0>
struct FnCfg {
    static constexpr int BN = 128 * NTW;                       // 8 consumer waves x NTW MFMA tiles x 16 channels
    static constexpr int MTFULL = NTW >= 4 ? 6 : (NTW == 3 ? 8 : (NTW == 2 ? 12 : 16));   // 96 accumulator VGPRs
    static constexpr int MTMAX = MTCAP > 0 ? MTCAP : MTFULL;   // (MTCAP: shorter tiles so that a three-stage ring fits)
    static_assert(MTMAX <= MTFULL, "accumulator budget");
    static constexpr int RMAX = 16 * MTMAX;
    static constexpr int XOFF = 0, WOFF = RMAX * 128;
    static constexpr int STAGE = (RMAX + BN) * 128;            // bytes: X rows, then W rows (128 B each)
    static constexpr int NWP = BN / 8;                         // 8-row W pieces per K tile
    static constexpr int WPL = NWP / FN_NLOAD;                 // ... per loader wave
    static constexpr int XPL = (RMAX / 8 + FN_NLOAD - 1) / FN_NLOAD;
};

// ---- work decomposition: identical arithmetic on host (grid size) and device
struct FnSched { int units_b, cu_units, n_cpb, n_chunks; };
__host__ __device__ inline FnSched fn_sched(int M, int batch) {
    FnSched s;
    s.units_b = (M + 15) / 16;                                 // 16-row units per batch element
    int cpb = 256 / batch;                                     // chunks per batch element: one chunk per CU
    if (cpb < 1) cpb = 1;
    s.cu_units = (s.units_b + cpb - 1) / cpb;
    s.n_cpb = (s.units_b + s.cu_units - 1) / s.cu_units;
    s.n_chunks = batch * s.n_cpb;
    return s;
}

struct FnTile { int b, m0, mt; };                              // rows [m0, m0 + 16*mt) of batch element b
struct FnWalk {                                                // the tiles of one block's chunk
    int b, unit, left, tiles_left;
    __device__ __forceinline__ void init(const FnSched& s, int chunk, int mtmax) {
        b = chunk / s.n_cpb;
        unit = (chunk - b * s.n_cpb) * s.cu_units;
        left = min(s.cu_units, s.units_b - unit);
        if (left < 0) left = 0;
        tiles_left = (left + mtmax - 1) / mtmax;
    }
    __device__ __forceinline__ bool next(FnTile& t) {          // even split of the remaining units
        if (tiles_left <= 0) return false;
        const int mt = (left + tiles_left - 1) / tiles_left;
        t.b = b; t.m0 = unit * 16; t.mt = mt;
        unit += mt; left -= mt; --tiles_left;
        return true;
    }
};

// W source row (inside a wave's slab of 16*NTW rows) of 8-row piece pw, without the per-lane part.
// Tile PAIRS (2u, 2u+1) are staged permuted so that a lane ends with 8 consecutive channels (nt_wperm in
// aew_gemm.hip); an odd last tile is staged in natural order.  lane part: perm -> (lr>>2)*8 + (lr&3), plain -> lr.
template <int EPI, int NTW>
__device__ __forceinline__ int fn_wpiece_row(int pw, bool& plain) {
    if (EPI == AEW_EPI_GATED && NTW == 4) { plain = false; return (pw & 1) * 32 + (pw >> 2) * 16 + ((pw >> 1) & 1) * 4; }
    if (EPI == AEW_EPI_GATED) { plain = true; return pw * 8; }   // NTW = 2: tile 0 = 16 filt, tile 1 = 16 gate columns
    const int u = pw >> 2;
    plain = (2 * u + 1 >= NTW);
    return plain ? u * 32 + (pw & 3) * 8 : u * 32 + (pw & 1) * 16 + ((pw >> 1) & 1) * 4;
}

// LDS-DMA, opaque to the compiler (the consumers' fragment reads must not inherit vmcnt waits):
// per-lane 64-bit source (X pieces: every lane has its own row pointer, masked rows point at zeros) ...
__device__ __forceinline__ void fn_dma_v(const void* gsrc, uint32_t lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(lds_off) : "memory", "m0");
}
// ... and scalar base + per-lane 32-bit offset (W pieces: no per-piece vector address arithmetic at all)
__device__ __forceinline__ void fn_dma_s(uint64_t sbase, uint32_t voff, uint32_t lds_off) {
    // s_nop 4: the scalar base may come straight from a v_readlane (SGPR spill reload = VALU write of an SGPR), which
    // needs 5 wait states before a VMEM instruction reads it as its address; hipcc pads nothing inside asm
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" :: "s"(sbase), "v"(voff), "s"(lds_off) : "memory", "m0");
}

__device__ __forceinline__ void fn_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------
// loader side
// ---------------------------------------------------------------------------------------------------------------
template <int NTW, int MTCAP = 0>
struct FnLoader {
    typedef FnCfg<NTW, MTCAP> Cfg;
    const char* xp[Cfg::XPL];      // per-lane source of this loader's X pieces (piece q = ld + 4*jj), current segment
    uint32_t voff_perm, voff_plain;
    int ld, lane;
    uint32_t lds0;
    // K position of the NEXT tile to issue
    int seg, kin, kt;              // segment, channel offset inside it, K tile index inside the GEMM
    FnTile tile;                   // tile the next K tile belongs to

    __device__ __forceinline__ void setup_x(const aew_gemm_nt_t& g) {
        const aew_seg_t s = g.seg[seg];
        const int lr = lane >> 3, pos = lane & 7;
#pragma unroll
        for (int jj = 0; jj < Cfg::XPL; ++jj) {
            const int q = ld + FN_NLOAD * jj;
            const int r = q * 8 + lr;
            bool ok;
            const char* src = seg_row_ptr_sel(s, tile.b, tile.m0 + r, 2, ok) + kin * 2 + ((pos ^ ((r >> 1) & 7)) << 4);
            ok = ok && q < 2 * tile.mt;
            xp[jj] = ok ? src : reinterpret_cast<const char*>(aew_zero_region) + (pos << 4);
        }
    }
    __device__ __forceinline__ void begin_tile(const aew_gemm_nt_t& g, const FnTile& t) {
        tile = t; seg = 0; kin = 0; kt = 0;
        setup_x(g);
    }
    // issue K tile `kt` of `tile` into the stage at byte offset `stage`.  ALLX: every X piece goes out (pieces beyond
    // the tile's rows stream zeros into rows nobody reads), so that a loader issues the same number of DMA
    // instructions per K tile whatever the tile - what a counted vmcnt wait needs.
    template <int EPI, bool ALLX = false>
    __device__ __forceinline__ void issue(const aew_gemm_nt_t& g, uint32_t stage) {
        if (FN_ABL(g, 2)) {                                    // (ablation: no operand traffic)
            ++kt; kin += 64;
            if (kin >= g.seg[seg].k_len && seg + 1 < g.n_segs) { ++seg; kin = 0; }
            return;
        }
#pragma unroll
        for (int jj = 0; jj < Cfg::XPL; ++jj) {
            const int q = ld + FN_NLOAD * jj;
            if (ALLX || q < 2 * tile.mt) fn_dma_v(xp[jj], lds0 + stage + Cfg::XOFF + q * 1024);
            xp[jj] += 128;
        }
        const uint64_t wk = reinterpret_cast<uint64_t>(g.W) + (uint64_t)kt * 128;
        const uint32_t pitch = (uint32_t)g.K_total * 2;
#pragma unroll
        for (int w = 0; w < Cfg::WPL; ++w) {
            const int p = ld * Cfg::WPL + w;                   // pieces of a loader are consecutive: p & 1 == w & 1
            const int slab = p / (2 * NTW), pw = p - slab * (2 * NTW);
            bool plain;
            const int row = slab * (16 * NTW) + fn_wpiece_row<EPI, NTW>(pw, plain);
            const uint32_t vo = (plain ? voff_plain : voff_perm) ^ ((uint32_t)(p & 1) << 6);
            fn_dma_s(wk + (uint64_t)row * pitch, vo, lds0 + stage + Cfg::WOFF + p * 1024);
        }
        // advance to the next K tile
        ++kt; kin += 64;
        if (kin >= g.seg[seg].k_len && seg + 1 < g.n_segs) { ++seg; kin = 0; setup_x(g); }
    }
    __device__ __forceinline__ void init(int ld_, int lane_, uint32_t lds0_, uint32_t pitch) {
        ld = ld_; lane = lane_; lds0 = lds0_;
        const int lr = lane >> 3, pos = lane & 7;
        const uint32_t ch = (uint32_t)((pos ^ (lr >> 1)) << 4);
        voff_perm = (uint32_t)((lr >> 2) * 8 + (lr & 3)) * pitch + ch;
        voff_plain = (uint32_t)lr * pitch + ch;
    }
};

// a plain [rows][K] weight matrix streamed in K tiles of 64 (the fused layer's W2): pieces p = ld*PL + w
template <int NTW2, int PL>
__device__ __forceinline__ void fn_issue_w2(const void* W2, int K2, int kt, int ld, uint32_t voff_perm, uint32_t voff_plain,
                                            uint32_t lds_dst) {
    const uint64_t wk = reinterpret_cast<uint64_t>(W2) + (uint64_t)kt * 128;
    const uint32_t pitch = (uint32_t)K2 * 2;
#pragma unroll
    for (int w = 0; w < PL; ++w) {
        const int p = ld * PL + w;
        const int slab = p / (2 * NTW2), pw = p - slab * (2 * NTW2);
        bool plain;
        const int row = slab * (16 * NTW2) + fn_wpiece_row<AEW_EPI_STORE, NTW2>(pw, plain);
        const uint32_t vo = (plain ? voff_plain : voff_perm) ^ ((uint32_t)(p & 1) << 6);
        fn_dma_s(wk + (uint64_t)row * pitch, vo, lds_dst + p * 1024);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// consumer side
// ---------------------------------------------------------------------------------------------------------------
// one K tile of 64: two MFMA K steps.  xs / ws: LDS byte addresses of the operand images (row 0); xo: this lane's
// fragment offset for K step 0 (K step 1 = xo ^ 64: chunk index + 4 under the row swizzle)
// MFMA group G of a K tile (NTW MFMAs), followed by the fragment read(s) that keep the reads ahead: K step 1's W
// fragments and first X fragment must be in before group MT, hence two reads per gap until then
template <int NTW, int MT, int G, int LEFT>
__device__ __forceinline__ void fn_sched_groups() {
    if constexpr (G < 2 * MT) {
        __builtin_amdgcn_sched_group_barrier(0x008, NTW, 0);
        constexpr int issued = 2 * (NTW + MT) - LEFT;
        constexpr int want = (issued < 2 * NTW + MT + 2) ? 2 : 1;
        constexpr int r = LEFT < want ? LEFT : want;
        if constexpr (r > 0) __builtin_amdgcn_sched_group_barrier(0x100, r, 0);
        fn_sched_groups<NTW, MT, G + 1, LEFT - r>();
    }
}

template <int NTW, int MT>
__device__ __forceinline__ void fn_compute(const char* xs, const char* ws, int xo, f32x4_t (&acc)[NTW][MT]) {
    if constexpr (NTW * MT >= 24) {
        // Full accumulator budget (24 tiles = 96 VGPRs of the 168): K step by K step in source order.  Any forced
        // read-ahead here (fragments of both K steps live, sched_group_barrier pipelining) makes the register
        // allocator spill INSIDE the loop (62 scratch accesses per K tile, 3x slower tiles); the compiler's own
        // order keeps one X fragment in flight and the partner wave on the SIMD covers the LDS latency.
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int o = xo ^ (ks << 6);
            bf16x8_t wf[NTW], xf[MT];
#pragma unroll
            for (int i = 0; i < NTW; ++i) wf[i] = *reinterpret_cast<const bf16x8_t*>(ws + o + i * 2048);
#pragma unroll
            for (int j = 0; j < MT; ++j) xf[j] = *reinterpret_cast<const bf16x8_t*>(xs + o + j * 2048);
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int i = 0; i < NTW; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
        return;
    }
    bf16x8_t wf[2][NTW], xf[2][MT];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int o = xo ^ (ks << 6);
#pragma unroll
        for (int i = 0; i < NTW; ++i) wf[ks][i] = *reinterpret_cast<const bf16x8_t*>(ws + o + i * 2048);
#pragma unroll
        for (int j = 0; j < MT; ++j) xf[ks][j] = *reinterpret_cast<const bf16x8_t*>(xs + o + j * 2048);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int i = 0; i < NTW; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][i], xf[ks][j], acc[i][j], 0, 0, 0);
    // Shape of the step (one basic block): the fragment reads run AHEAD of the MFMA groups that consume them - W
    // fragments and two X fragments first, then after every group of NTW MFMAs (one X fragment against all W
    // fragments) the next read(s), so that a wave never sits on an lgkmcnt(0) between groups.
    constexpr int total_reads = 2 * (NTW + MT);
    constexpr int pre = NTW + 2 < total_reads ? NTW + 2 : total_reads;
    __builtin_amdgcn_sched_group_barrier(0x100, pre, 0);
    fn_sched_groups<NTW, MT, 0, total_reads - pre>();
}

// 8 (pair) or 4 (single tile) consecutive channels of one row through the STORE-type flag epilogue / DFG.
// Same arithmetic, in the same order, as epi_store8_pf / epi_dfg8_pf of aew_gemm.hip.
template <int W>
__device__ __forceinline__ void fn_store_row(const EpiUni& U, unsigned fl, char* o0, char* o1, const char* a0, const char* a1,
                                             int n, float v[W], unsigned& zc) {
    if (fl & AEW_EF_RELU) {
#pragma unroll
        for (int r = 0; r < W; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
    }
    if (fl & AEW_EF_OUT1_PRE) row_store<W>(o1, U.dt_o1, n, v);
    if (fl & AEW_EF_ADD_AUX0) {
        float a[W];
        row_load<W>(a0, U.dt_a0, n, a);
#pragma unroll
        for (int r = 0; r < W; ++r) v[r] = v[r] + a[r];
    }
    if (fl & AEW_EF_RELU_POST) {
#pragma unroll
        for (int r = 0; r < W; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
    }
    if (fl & (AEW_EF_MUL_POS1 | AEW_EF_OUT1_POS1)) {
        float a[W], w[W];
        row_load<W>(a1, U.dt_a1, n, a);
#pragma unroll
        for (int r = 0; r < W; ++r) w[r] = a[r] > 0.f ? v[r] : 0.f;
        if (fl & AEW_EF_OUT1_POS1) row_store<W>(o1, U.dt_o1, n, w);
        if (fl & AEW_EF_MUL_POS1) {
#pragma unroll
            for (int r = 0; r < W; ++r) v[r] = w[r];
        }
    }
    if ((fl & AEW_EF_COUNT_ZERO) && o0) {
#pragma unroll
        for (int r = 0; r < W; ++r) zc += (n + r < U.N && v[r] == 0.f) ? 1u : 0u;
    }
    row_store<W>(o0, U.dt_o0, n, v);
}

template <int W>
__device__ __forceinline__ void fn_dfg_row(const EpiUni& U, char* o0, const char* a0, const char* a1, int n, const float dz[W]) {
    float pf[W], pg[W], df[W], dg[W];
    row_load<W>(a0, U.dt_a0, n, pf);
    row_load<W>(a1, U.dt_a1, n, pg);
#pragma unroll
    for (int r = 0; r < W; ++r) { df[r] = dz[r] * pf[r]; dg[r] = dz[r] * pg[r]; }
    const int np = (n >> 4) * 32 + (n & 15);
    row_store<W>(o0, U.dt_o0, np, df);
    row_store<W>(o0, U.dt_o0, np + 16, dg);
}

// STORE / DFG epilogue of a tile: views out0, out1, aux0, aux1 of `g` (or the fused layer's second GEMM when
// SECOND: out3 / aux0, flags = ADD_AUX0).
template <int NTW, int MT, int EPI, bool SECOND>
__device__ __forceinline__ void fn_epilogue(const aew_gemm_nt_t& g, f32x4_t (&acc)[NTW][MT], const FnTile& t, int wn, int lane) {
    const int fi = lane & 15, fg = lane >> 4;
    const int mbase = t.m0 + fi;
    const EpiViewCtx c0 = epi_view_ctx(SECOND ? g.out3 : g.out0, t.b, mbase);
    const EpiViewCtx c1 = SECOND ? c0 : epi_view_ctx(g.out1, t.b, mbase);
    const EpiViewCtx ca0 = epi_view_ctx(g.aux0, t.b, mbase);
    const EpiViewCtx ca1 = SECOND ? c0 : epi_view_ctx(g.aux1, t.b, mbase);
    EpiUni U = epi_uni(g);
    if (SECOND) { U.fl = AEW_EF_ADD_AUX0; U.N = g.N2; U.dt_o0 = g.out3.dtype; }
    const unsigned fl = U.fl;
    const int slab0 = wn * 16 * NTW;
    unsigned zc = 0;
    // bias folded into the accumulators (the same fp32 add, once per wave)
    if (EPI == AEW_EPI_STORE && !SECOND && (fl & AEW_EF_BIAS)) {
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const bool pair = (i | 1) < NTW;
            const int n = pair ? slab0 + (i >> 1) * 32 + 8 * fg + (i & 1) * 4 : slab0 + 16 * i + 4 * fg;
            if (n < U.N) {
                const float4 bb = *reinterpret_cast<const float4*>(g.bias + (int64_t)t.b * g.bias_bs + n);
#pragma unroll
                for (int j = 0; j < MT; ++j) { acc[i][j][0] += bb.x; acc[i][j][1] += bb.y; acc[i][j][2] += bb.z; acc[i][j][3] += bb.w; }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const bool row_ok = mbase + j * 16 < g.M;
        char* o0 = epi_view_row(c0, j);
        char* o1 = (EPI == AEW_EPI_STORE && !SECOND) ? epi_view_row(c1, j) : nullptr;
        const char* a0 = epi_view_row(ca0, j);
        const char* a1 = SECOND ? nullptr : epi_view_row(ca1, j);
        if (!row_ok || FN_ABL(g, 16)) continue;
#pragma unroll
        for (int u = 0; u < (NTW + 1) / 2; ++u) {
            if (2 * u + 1 < NTW) {                             // a tile pair: 8 consecutive channels
                const int n = slab0 + u * 32 + 8 * fg;
                float v[8] = {acc[2 * u][j][0], acc[2 * u][j][1], acc[2 * u][j][2], acc[2 * u][j][3],
                              acc[2 * u + 1][j][0], acc[2 * u + 1][j][1], acc[2 * u + 1][j][2], acc[2 * u + 1][j][3]};
                if (n < U.N) {
                    if (EPI == AEW_EPI_DFG) fn_dfg_row<8>(U, o0, a0, a1, n, v);
                    else fn_store_row<8>(U, fl, o0, o1, a0, a1, n, v, zc);
                }
            } else {                                           // odd last tile: 4 consecutive channels
                const int n = slab0 + 32 * u + 4 * fg;
                float v[4] = {acc[2 * u][j][0], acc[2 * u][j][1], acc[2 * u][j][2], acc[2 * u][j][3]};
                if (n < U.N) {
                    if (EPI == AEW_EPI_DFG) fn_dfg_row<4>(U, o0, a0, a1, n, v);
                    else fn_store_row<4>(U, fl, o0, o1, a0, a1, n, v, zc);
                }
            }
        }
    }
    if (EPI == AEW_EPI_STORE && !SECOND && (fl & AEW_EF_COUNT_ZERO)) {
        zc = (unsigned)wave_sum((float)zc);
        if (lane == 0 && zc) atomicAdd(g.counter, (unsigned long long)zc);
    }
}

// gated epilogue (wavenet.py:100-102).  NTW = 4: the wave's slab is 64 packed columns = 32 channels, tiles 0,1 filt,
// tiles 2,3 gate of channels 32*wn + 8*fg + {0..7}.  NTW = 2: 32 packed columns = 16 channels, tile 0 filt, tile 1
// gate of channels 16*wn + 4*fg + {0..3}.  With ZLDS the z tile is also written to LDS in the fragment layout of the
// residual GEMM's activation operand: [K tile = ch / 64][row][128 B], 16-byte chunk ^ ((row >> 1) & 7).
template <int NTW, int MT, bool ZLDS>
__device__ __forceinline__ void fn_epilogue_gated(const aew_gemm_nt_t& g, f32x4_t (&acc)[NTW][MT], const FnTile& t, int wn, int lane,
                                                  char* zl, int zl_ktile_bytes) {
    static_assert(NTW == 4 || NTW == 2, "gated slabs are 64 or 32 packed columns");
    constexpr int W = NTW == 4 ? 8 : 4;
    const int fi = lane & 15, fg = lane >> 4;
    const int mbase = t.m0 + fi;
    const EpiViewCtx c0 = epi_view_ctx(g.out0, t.b, mbase), c1 = epi_view_ctx(g.out1, t.b, mbase), c2 = epi_view_ctx(g.out2, t.b, mbase);
    const int ch = (8 * NTW) * wn + W * fg;
    const int np_f = (ch >> 4) * 32 + (ch & 15);
    float fb[W], gb[W];
    {
        const float* bp = g.bias + (int64_t)t.b * g.bias_bs + np_f;
#pragma unroll
        for (int q = 0; q < W / 4; ++q) {
            const float4 bf = *reinterpret_cast<const float4*>(bp + 4 * q);
            const float4 bg = *reinterpret_cast<const float4*>(bp + 16 + 4 * q);
            fb[4 * q] = bf.x; fb[4 * q + 1] = bf.y; fb[4 * q + 2] = bf.z; fb[4 * q + 3] = bf.w;
            gb[4 * q] = bg.x; gb[4 * q + 1] = bg.y; gb[4 * q + 2] = bg.z; gb[4 * q + 3] = bg.w;
        }
    }
    const bool ch_ok = ch < g.N;
    // z tile position of this lane's channels: K tile ch >> 6, 16-byte chunk (ch & 63) >> 3, byte (ch & 7) * 2 in it
    char* zrow = zl + (ch >> 6) * zl_ktile_bytes + fi * 128 + ((((ch & 63) >> 3) ^ ((fi >> 1) & 7)) << 4) + (ch & 7) * 2;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const bool row_ok = mbase + j * 16 < g.M;
        // two channels at a time, packed to bf16 at once: 3 * W / 2 live result registers instead of 3 * W floats
        // (at 168 VGPRs with 96 accumulators the wide form spilled into the K loop)
        uint32_t zp[W / 2], fp[W / 2], gp[W / 2];
#pragma unroll
        for (int h = 0; h < W / 2; ++h) {
            float zz[2], ff[2], gg[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int e = 2 * h + r;
                const float f = acc[e >> 2][j][e & 3], q = acc[NTW / 2 + (e >> 2)][j][e & 3];
                const float a = tanh_f(f + fb[e]);
                const float s = sigmoid_f(q + gb[e]);
                zz[r] = a * s;
                ff[r] = s * (1.0f - a * a);
                gg[r] = zz[r] * (1.0f - s);
            }
            zp[h] = pack2_bf16(zz[0], zz[1]); fp[h] = pack2_bf16(ff[0], ff[1]); gp[h] = pack2_bf16(gg[0], gg[1]);
        }
        if (ZLDS) {
            if constexpr (W == 8)
                *reinterpret_cast<uint4*>(zrow + j * 2048) = ch_ok ? make_uint4(zp[0], zp[1], zp[W / 2 - 2], zp[W / 2 - 1]) : make_uint4(0, 0, 0, 0);
            else
                *reinterpret_cast<uint2*>(zrow + j * 2048) = ch_ok ? make_uint2(zp[0], zp[1]) : make_uint2(0, 0);
        }
        if (row_ok && ch_ok && !FN_ABL(g, 16)) {               // (ablation: math but no global stores)
            char* o0 = epi_view_row(c0, j);
            char* o1 = epi_view_row(c1, j);
            char* o2 = epi_view_row(c2, j);
            if constexpr (W == 8) {
                if (o0) *reinterpret_cast<uint4*>(o0 + ch * 2) = make_uint4(zp[0], zp[1], zp[W / 2 - 2], zp[W / 2 - 1]);
                if (o1) *reinterpret_cast<uint4*>(o1 + ch * 2) = make_uint4(fp[0], fp[1], fp[W / 2 - 2], fp[W / 2 - 1]);
                if (o2) *reinterpret_cast<uint4*>(o2 + ch * 2) = make_uint4(gp[0], gp[1], gp[W / 2 - 2], gp[W / 2 - 1]);
            } else {
                if (o0) *reinterpret_cast<uint2*>(o0 + ch * 2) = make_uint2(zp[0], zp[1]);
                if (o1) *reinterpret_cast<uint2*>(o1 + ch * 2) = make_uint2(fp[0], fp[1]);
                if (o2) *reinterpret_cast<uint2*>(o2 + ch * 2) = make_uint2(gp[0], gp[1]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// the kernel.  NTW: MFMA tiles per consumer wave along N (N_pad = 128 * NTW).  NTW2 > 0: fused layer, the second
// GEMM has N2_pad = 128 * NTW2 and K2 = N_pad / 2 (the z channels).
// ---------------------------------------------------------------------------------------------------------------
template <int NTW, int NTW2>
struct FnFuse {                                                // LDS overlay of the second GEMM (see file header)
    typedef FnCfg<NTW> Cfg;
    static constexpr int W2S = 128 * NTW2 * 128;               // one K tile of W2
    static constexpr int K2T = NTW;                            // K tiles of the second GEMM: (128*NTW/2) / 64
    static constexpr int ZKT = Cfg::RMAX * 128;                // bytes per K tile of the z image
    static constexpr int WA = 0, WB = W2S;
    static constexpr int ZOFF = (2 * W2S > Cfg::STAGE) ? 2 * W2S : Cfg::STAGE;
    static constexpr int BYTES = ZOFF + K2T * ZKT;
    static constexpr int W2PL = (128 * NTW2 / 8) / FN_NLOAD;   // W2 pieces per loader and K tile
    static_assert(NTW2 == 0 || W2S <= Cfg::STAGE, "W2 stage A must fit inside operand stage 0");
};

template <int NTW, int EPI, int NTW2, int NST = 2, int MTCAP = 0>
constexpr int fn_lds_bytes() {
    return (NTW2 > 0 && FnFuse<NTW, NTW2 ? NTW2 : 1>::BYTES > 2 * FnCfg<NTW>::STAGE) ? FnFuse<NTW, NTW2 ? NTW2 : 1>::BYTES
                                                                                        : NST * FnCfg<NTW, MTCAP>::STAGE;
}

template <int NTW, int MT, int EPI, int NTW2, int NST = 2, int MTCAP = 0>
__device__ __forceinline__ void fn_consumer_tile(const aew_gemm_nt_t& g, char* smem, const FnTile& t, int wn, int lane,
                                                 int nkt, int& ktg) {
    typedef FnCfg<NTW, MTCAP> Cfg;
    const int fi = lane & 15, fg = lane >> 4;
    const int xo = fi * 128 + ((fg ^ ((fi >> 1) & 7)) << 4);
    f32x4_t acc[NTW][MT];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nkt; ++k) {
        fn_barrier();                                          // A_k: tile k landed; my reads of tile k-1 are consumed
        // (NST = 3: ktg is the ring slot of this tile's first K tile, kept in [0, 3))
        const char* st = smem + (NST == 2 ? ((ktg + k) & 1) : ((ktg + k) % NST)) * Cfg::STAGE;
        if (!FN_ABL(g, 1))                                     // (ablation: barriers only)
            fn_compute<NTW, MT>(st + Cfg::XOFF, st + Cfg::WOFF + wn * NTW * 2048, xo, acc);
    }
    ktg = NST == 2 ? ktg + nkt : (ktg + nkt) % NST;
    if constexpr (NTW2 == 0) {
        if (FN_ABL(g, 4)) return;                              // (ablation: no epilogue)
        if constexpr (EPI == AEW_EPI_GATED) fn_epilogue_gated<NTW, MT, false>(g, acc, t, wn, lane, nullptr, 0);
        else fn_epilogue<NTW, MT, EPI, false>(g, acc, t, wn, lane);
    } else {
        typedef FnFuse<NTW, NTW2> F;
        fn_barrier();                                          // G: every wave is done with the last operand stage
        if (!FN_ABL(g, 4)) fn_epilogue_gated<NTW, MT, true>(g, acc, t, wn, lane, smem + F::ZOFF, F::ZKT);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my z rows are in LDS
        fn_barrier();                                          // Z: z image complete, W2 K tiles 0 (and 1) landed
        f32x4_t acc2[NTW2][MT];
#pragma unroll
        for (int i = 0; i < NTW2; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc2[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int u = 0; u < F::K2T; ++u) {
            const char* ws = smem + ((u & 1) ? F::WB : F::WA) + wn * NTW2 * 2048;
            if (!FN_ABL(g, 8)) fn_compute<NTW2, MT>(smem + F::ZOFF + u * F::ZKT, ws, xo, acc2);
            if (u + 1 < F::K2T) fn_barrier();                  // H: W2 stage u & 1 is free, K tile u + 1 landed
        }
        fn_barrier();                                          // F: LDS is free for the next tile's operands
        if (!FN_ABL(g, 4)) fn_epilogue<NTW2, MT, AEW_EPI_STORE, true>(g, acc2, t, wn, lane);
    }
}

template <int N>
__device__ __forceinline__ void fn_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// NST = 3 (plain GEMMs only, NTW2 = 0): a three-stage operand ring, the loaders TWO K tiles ahead of the consumers with
// a counted vmcnt wait.  One block per CU fills its LDS at the rate of (bytes in flight) / (DMA latency): with one K tile
// of 50-56 KB in flight the long-K multi-segment GEMMs (skip sum K = 5120, cond gradient K = 10240: one tile per block,
// 80-160 K tiles) ran at 27 GB/s per CU.  MTCAP shortens the tiles so that three stages fit in 160 KB.
template <int NTW, int EPI, int NTW2, int NST = 2, int MTCAP = 0>
__global__ __launch_bounds__(FN_THREADS, 3) void k_fn(const aew_gemm_nt_t g) {
    typedef FnCfg<NTW, MTCAP> Cfg;
    static_assert(NST == 2 || (NST == 3 && NTW2 == 0 && (Cfg::RMAX / 8) % FN_NLOAD == 0), "ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const FnSched sch = fn_sched(g.M, g.batch);
    const int chunk = blockIdx.x;
    if (chunk >= sch.n_chunks) return;
    const int nkt = g.K_total / 64;
    FnWalk walk;
    walk.init(sch, chunk, Cfg::MTMAX);
    if (wave >= FN_NCONS) {
        // ------------------------------------------------------------------------------ loader
        __builtin_amdgcn_s_setprio(1);
        FnLoader<NTW, MTCAP> L;
        L.init(wave - FN_NCONS, lane, (uint32_t)(uintptr_t)AEW_LDS_PTR(smem), (uint32_t)g.K_total * 2);
        if constexpr (NST == 3) {
            // the K tiles of all of this block's tiles as ONE stream through a ring of three slots
            constexpr int PER = Cfg::XPL + Cfg::WPL;           // DMA instructions per K tile and loader
            FnTile cur, nxt;
            if (!walk.next(cur)) return;
            bool more = walk.next(nxt);
            L.begin_tile(g, cur);
            int in_tile = 0, head = 0, ahead = 0;
            auto issue_next = [&]() {
                if (in_tile == nkt) {
                    if (!more) return;
                    cur = nxt;
                    L.begin_tile(g, cur);
                    in_tile = 0;
                    more = walk.next(nxt);
                }
                L.template issue<EPI, true>(g, (uint32_t)(head * Cfg::STAGE));
                ++in_tile; ++ahead;
                head = head == 2 ? 0 : head + 1;
            };
            issue_next();
            issue_next();
            while (ahead > 0) {
                if (ahead >= 2) fn_wait_vm<PER>(); else wait_vm0();    // the OLDEST K tile in flight has landed
                fn_barrier();                                  // A_k (the consumers are done with K tile k - 1: its slot is `head`)
                --ahead;
                issue_next();
            }
            return;
        }
        uint32_t voff2_perm = 0, voff2_plain = 0;
        if constexpr (NTW2 > 0) {
            const int lr = lane >> 3, pos = lane & 7;
            const uint32_t pitch2 = (uint32_t)(64 * NTW) * 2, ch = (uint32_t)((pos ^ (lr >> 1)) << 4);
            voff2_perm = (uint32_t)((lr >> 2) * 8 + (lr & 3)) * pitch2 + ch;
            voff2_plain = (uint32_t)lr * pitch2 + ch;
        }
        FnTile cur, nxt;
        bool have = walk.next(cur);
        int ktg = 0;
        if (have) { L.begin_tile(g, cur); L.template issue<EPI>(g, 0); }
        while (have) {
            const bool more = walk.next(nxt);
            for (int k = 0; k < nkt; ++k) {
                wait_vm0();
                fn_barrier();                                  // A_k
                const uint32_t st = (uint32_t)(((ktg + k + 1) & 1) * Cfg::STAGE);
                if (k + 1 < nkt) L.template issue<EPI>(g, st);
                else if constexpr (NTW2 > 0) {
                    typedef FnFuse<NTW, NTW2> F;
                    fn_issue_w2<NTW2, F::W2PL>(g.W2, 64 * NTW, 0, L.ld, voff2_perm, voff2_plain, L.lds0 + F::WA);
                } else if (more) { L.begin_tile(g, nxt); L.template issue<EPI>(g, st); }
            }
            ktg += nkt;
            if constexpr (NTW2 > 0) {
                typedef FnFuse<NTW, NTW2> F;
                wait_vm0();
                fn_barrier();                                  // G
                if (F::K2T > 1) fn_issue_w2<NTW2, F::W2PL>(g.W2, 64 * NTW, 1, L.ld, voff2_perm, voff2_plain, L.lds0 + F::WB);
                wait_vm0();
                fn_barrier();                                  // Z
#pragma unroll 1
                for (int u = 0; u + 1 < F::K2T; ++u) {
                    wait_vm0();
                    fn_barrier();                              // H_{u+1}: stage u & 1 free
                    if (u + 2 < F::K2T)
                        fn_issue_w2<NTW2, F::W2PL>(g.W2, 64 * NTW, u + 2, L.ld, voff2_perm, voff2_plain,
                                                   L.lds0 + ((u & 1) ? F::WB : F::WA));
                }
                wait_vm0();
                fn_barrier();                                  // F
                if (more) { L.begin_tile(g, nxt); L.template issue<EPI>(g, (uint32_t)((ktg & 1) * Cfg::STAGE)); }
            }
            cur = nxt;
            have = more;
        }
        return;
    }
    // ---------------------------------------------------------------------------------- consumer
    const int wn = wave;
    FnTile t;
    int ktg = 0;
    while (walk.next(t)) {
        switch (t.mt) {
#define FN_CASE(MTV)                                                                                   \
            case MTV:                                                                                   \
                if constexpr (MTV <= Cfg::MTMAX) fn_consumer_tile<NTW, MTV, EPI, NTW2, NST, MTCAP>(g, smem, t, wn, lane, nkt, ktg); \
                break;
            FN_CASE(1) FN_CASE(2) FN_CASE(3) FN_CASE(4) FN_CASE(5) FN_CASE(6) FN_CASE(7) FN_CASE(8)
            FN_CASE(9) FN_CASE(10) FN_CASE(11) FN_CASE(12) FN_CASE(13) FN_CASE(14) FN_CASE(15) FN_CASE(16)
#undef FN_CASE
            default: break;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
extern "C" int aew_set_fn(int on) { g_tune.fn_enable = on ? 1 : 0; return 0; }
extern "C" int aew_set_fn_ring3(int min_k_tiles) { g_tune.fn_ring3 = min_k_tiles < 0 ? 0 : min_k_tiles; return 0; }
int g_fn_enable_flag() { return AEW_T().fn_enable; }

template <int NTW, int EPI, int NTW2, int NST = 2, int MTCAP = 0>
static int fn_launch(const aew_gemm_nt_t& g, hipStream_t st) {
    static std::atomic<unsigned long long> attr_done{0};          // one bit per device (function attributes are per device)
    constexpr int lds = fn_lds_bytes<NTW, EPI, NTW2, NST, MTCAP>();
    static_assert(lds <= 160 * 1024, "LDS budget");
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {      // (racing threads both set it: harmless)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fn<NTW, EPI, NTW2, NST, MTCAP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    const FnSched s = fn_sched(g.M, g.batch);
    hipLaunchKernelGGL((k_fn<NTW, EPI, NTW2, NST, MTCAP>), dim3(s.n_chunks), dim3(FN_THREADS), lds, st, g);
    return (int)hipGetLastError();
}

// can this descriptor run on the full-N kernels?  (bf16, whole N in one block, STORE / DFG / GATED epilogues with
// bf16 aux operands handled by row_load, segments that fit the zero span)
static bool fn_supported(const aew_gemm_nt_t& g) {
    if (g.dtype != AEW_BF16 || g.N_pad % 128 || g.N_pad > 512 || g.K_total % 64) return false;
    if (g.epi != AEW_EPI_STORE && g.epi != AEW_EPI_DFG && g.epi != AEW_EPI_GATED) return false;
    if (g.epi == AEW_EPI_GATED && g.N_pad != 512 && g.N_pad != 256) return false;
    if (g.epi == AEW_EPI_DFG && g.N_pad > 384) return false;
    for (int s = 0; s < g.n_segs; ++s)
        if (g.seg[s].k_len % 64 || g.seg[s].k_len * 2 > AEW_ZERO_SPAN) return false;
    if (g.W2) {
        if (g.epi != AEW_EPI_GATED || g.N2_pad % 128) return false;
        const int a = g.N_pad / 128, b2 = g.N2_pad / 128;
        if (!((a == 4 && b2 == 3) || (a == 2 && b2 == 1) || (a == 2 && b2 == 2))) return false;
        if ((g.K_total / 64) & 1) return false;                // the LDS overlay assumes the last K tile sits in stage 1
    }
    return true;
}

static int launch_fn(const aew_gemm_nt_t& g, hipStream_t st) {
    const int a = g.N_pad / 128;
    if (g.W2) {
        const int b2 = g.N2_pad / 128;
        if (a == 4 && b2 == 3) return fn_launch<4, AEW_EPI_GATED, 3>(g, st);
        if (a == 2 && b2 == 1) return fn_launch<2, AEW_EPI_GATED, 1>(g, st);
        if (a == 2 && b2 == 2) return fn_launch<2, AEW_EPI_GATED, 2>(g, st);
        return AEW_E_UNSUP;
    }
    switch (g.epi) {
        case AEW_EPI_GATED:
            return a == 4 ? fn_launch<4, AEW_EPI_GATED, 0>(g, st) : fn_launch<2, AEW_EPI_GATED, 0>(g, st);
        case AEW_EPI_DFG:
            if (a == 1) return fn_launch<1, AEW_EPI_DFG, 0>(g, st);
            if (a == 2) return fn_launch<2, AEW_EPI_DFG, 0>(g, st);
            return fn_launch<3, AEW_EPI_DFG, 0>(g, st);
        default:
            // long K (the multi-segment skip sum / cond gradient): three-stage ring, see k_fn
            if (AEW_T().fn_ring3 && g.K_total >= 64 * AEW_T().fn_ring3) {
                if (a == 1) return fn_launch<1, AEW_EPI_STORE, 0, 3, 0>(g, st);
                if (a == 2) return fn_launch<2, AEW_EPI_STORE, 0, 3, 10>(g, st);
            }
            if (a == 1) return fn_launch<1, AEW_EPI_STORE, 0>(g, st);
            if (a == 2) return fn_launch<2, AEW_EPI_STORE, 0>(g, st);
            if (a == 3) return fn_launch<3, AEW_EPI_STORE, 0>(g, st);
            return fn_launch<4, AEW_EPI_STORE, 0>(g, st);
    }
}

// Which kernel launch_gemm_nt runs for this descriptor under the current settings (bench.py / tools: per-kernel
// rooflines are grouped by the kernel that ran, like a rocprofv3 kernel trace groups them by name).
//   0 k_gemm_nt_bf16 (256- or 192-row tiles)   1 k_gemm_nt_bf16_p64 (64-row tiles: launches of few tiles)
//   2 k_fn (full-N loader / consumer)          3 k_gemm_nt_f32        4 k_gemm_nt_check      5 an A/B shape
//   6 k_gemm_nt_bf16_win (one LDS window for both taps of a dilated pair)
extern "C" int aew_nt_kernel(const aew_gemm_nt_t* gp) {
    if (!gp) return AEW_E_ARG;
    const aew_gemm_nt_t& g = *gp;
    if (g.impl == 1) return 4;
    if (g.dtype != AEW_BF16) return 3;
    if (g.impl == 2 && g_fn_enable_flag() && fn_supported(g)) return 2;
    if (AEW_T().nt_wave_rows != 64) return 5;
    bool zspan = true;
    for (int s = 0; s < g.n_segs; ++s) zspan = zspan && g.seg[s].k_len * 2 <= AEW_ZERO_SPAN;
    const int tiles256 = ((g.M + NT_BM - 1) / NT_BM) * g.batch * (g.N_pad / NT_BN);
    if (AEW_T().nt_small_tiles > 0 && tiles256 <= AEW_T().nt_small_tiles && zspan) return 1;
    return win_dwp(g) ? 6 : 0;
}

extern "C" int aew_set_nt_window(int max_dist) { g_tune.nt_window = max_dist < 0 ? 0 : (max_dist > 64 ? 64 : max_dist); return 0; }
