// aew_gemm.hip — multi-segment row-affine GEMMs on CDNA4 matrix cores (gfx950).
//
//   NT:  C[b][m][n]        = sum_s sum_k A_s[b][m*step_s+off_s][k] * W[n][K_s+k]    (fwd, dgrad)
//   TN:  dW[slab][n][K_s+k] = sum_m G[b][m*gs+go][n] * A_s[b][m*step_s+off_s][k]     (wgrad)
//
// Replaces every conv1d / conv_transpose1d / linear call on the reference's training path
// (wavenet.py:100-109,154,337,351,359-360; wave_encoder.py:39; vqema_bn.py:131) and their
// autograd backward.  Operand tiles are staged global->LDS with 16-byte LDS-DMA, swizzled on
// the source address; bf16 uses v_mfma_f32_16x16x32_bf16, fp32 uses v_mfma_f32_16x16x4_f32
// (an exact k-ordered fmaf chain, which is what makes the encoder->VQ path bit-exact against
// oracle/exact_chain.c).  MFMA operands are issued "swapped" (weights as the A operand) so a
// lane's 4 accumulator registers are 4 consecutive CHANNELS of one row: channels-last stores
// are then 8/16-byte vectors.
#include "aew_common.h"
#include <atomic>

// =============================================================================================
// epilogues: W consecutive channels n..n+W-1 of one output row (W = 8 in the bf16 MFMA kernel,
// 4 in the fp32 and check kernels).  Row pointers of the views are resolved once per row.
// =============================================================================================
struct EpiRow {                                      // per output row m: view row pointers
    char* o0; char* o1; char* o2;
    const char* a0; const char* a1;
};

__device__ __forceinline__ EpiRow epi_row(const aew_gemm_nt_t& g, int b, int m) {
    EpiRow R;
    R.o0 = view_rowptr(g.out0, b, m); R.o1 = view_rowptr(g.out1, b, m); R.o2 = view_rowptr(g.out2, b, m);
    R.a0 = view_rowptr(g.aux0, b, m); R.a1 = view_rowptr(g.aux1, b, m);
    return R;
}

// wave-uniform descriptor fields the epilogues need, read once (they live in kernarg memory)
struct EpiUni {
    unsigned fl;
    int N, n_split, dt_o0, dt_o1, dt_o2, dt_a0, dt_a1;
};
__device__ __forceinline__ EpiUni epi_uni(const aew_gemm_nt_t& g) {
    EpiUni u;
    u.fl = g.flags; u.N = g.N; u.n_split = g.n_split;
    u.dt_o0 = g.out0.dtype; u.dt_o1 = g.out1.dtype; u.dt_o2 = g.out2.dtype;
    u.dt_a0 = g.aux0.dtype; u.dt_a1 = g.aux1.dtype;
    return u;
}

template <int W>
__device__ __forceinline__ void epi_store(const aew_gemm_nt_t& g, const EpiUni& U, const EpiRow& R, int b, int n,
                                          float v[W], unsigned& zero_count, unsigned fl) {
    if (fl & AEW_EF_BIAS) {
        const float* bp = g.bias + (int64_t)b * g.bias_bs + n;
#pragma unroll
        for (int q = 0; q < W / 4; ++q) {
            const float4 bb = *reinterpret_cast<const float4*>(bp + 4 * q);
            v[4 * q] += bb.x; v[4 * q + 1] += bb.y; v[4 * q + 2] += bb.z; v[4 * q + 3] += bb.w;
        }
    }
    if (fl & AEW_EF_RELU) {
#pragma unroll
        for (int r = 0; r < W; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
    }
    if (fl & AEW_EF_OUT1_PRE) row_store<W>(R.o1, U.dt_o1, n, v);
    if (fl & AEW_EF_ADD_AUX0) {
        float a[W];
        row_load<W>(R.a0, U.dt_a0, n, a);
#pragma unroll
        for (int r = 0; r < W; ++r) v[r] = v[r] + a[r];
    }
    if (fl & AEW_EF_RELU_POST) {
#pragma unroll
        for (int r = 0; r < W; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
    }
    if (fl & (AEW_EF_MUL_POS1 | AEW_EF_OUT1_POS1)) {
        float a[W], w[W];
        row_load<W>(R.a1, U.dt_a1, n, a);
#pragma unroll
        for (int r = 0; r < W; ++r) w[r] = a[r] > 0.f ? v[r] : 0.f;
        if (fl & AEW_EF_OUT1_POS1) row_store<W>(R.o1, U.dt_o1, n, w);
        if (fl & AEW_EF_MUL_POS1) {
#pragma unroll
            for (int r = 0; r < W; ++r) v[r] = w[r];
        }
    }
    if ((fl & AEW_EF_COUNT_ZERO) && R.o0) {
#pragma unroll
        for (int r = 0; r < W; ++r) zero_count += (n + r < U.N && v[r] == 0.f) ? 1u : 0u;
    }
    row_store<W>(R.o0, U.dt_o0, n, v);
    if (fl & AEW_EF_OUT2_COPY) row_store<W>(R.o2, U.dt_o2, n, v);
}

// one gated unit from its pre-activations (bias included): z = tanh(f) sigmoid(g) and the two local derivatives
// dz/dfilt, dz/dgate, all from the fp32 factors (the saturated-tanh factor 1 - a^2 would suffer bf16 cancellation if
// formed in the backward)
__device__ __forceinline__ void gated_math(float f, float gt, float& z, float& pf, float& pg) {
    const float a = tanh_f(f);
    const float s = sigmoid_f(gt);
    z = a * s;
    pf = s * (1.0f - a * a);
    pg = z * (1.0f - s);
}

// filt / gate values of the same W channels ch..ch+W-1; np_f = packed column of filt channel ch
// (the W channels lie inside one 16-channel group, so their packed columns are contiguous)
// fbias / gbias: the W filt / gate biases of channels ch..ch+W-1 (loaded once per wave by the caller)
template <int W, bool ABL = false, bool WT = false>
__device__ __forceinline__ void epi_gated(const aew_gemm_nt_t& g, const EpiRow& R, int ch, const float f[W],
                                          const float gt[W], const float fbias[W], const float gbias[W]) {
    float z[W], pf[W], pg[W];
#pragma unroll
    for (int q = 0; q < W / 4; ++q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = 4 * q + r;
            if (ABL && (g.reserved & 512)) { z[e] = f[e] + fbias[e]; pf[e] = gt[e] + gbias[e]; pg[e] = f[e] - gt[e]; continue; }  // ablation
            gated_math(f[e] + fbias[e], gt[e] + gbias[e], z[e], pf[e], pg[e]);
        }
    }
    if (ABL && (g.reserved & 256)) {                   // ablation: math but no stores
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < W; ++e) t += z[e] + pf[e] + pg[e];
        asm volatile("" ::"v"(t));
        return;
    }
    row_store<W, WT>(R.o0, AEW_BF16, ch, z);             // WT: z is handed to a later stage of the same launch
    row_store<W>(R.o1, AEW_BF16, ch, pf);
    row_store<W>(R.o2, AEW_BF16, ch, pg);
}

template <int W>
__device__ __forceinline__ void epi_res_skip(const EpiUni& U, const EpiRow& R, int n, float v[W]) {
    if (n < U.n_split) {
        float a[W];
        row_load<W>(R.a0, U.dt_a0, n, a);
#pragma unroll
        for (int r = 0; r < W; ++r) v[r] += a[r];
        row_store<W>(R.o0, U.dt_o0, n, v);
    } else {
        const int c = n - U.n_split;
        if (!R.o1) return;
        if (U.fl & AEW_EF_ACCUM) {
            float a[W];
            row_load<W>(R.o1, U.dt_o1, c, a);
#pragma unroll
            for (int r = 0; r < W; ++r) v[r] += a[r];
        }
        row_store<W>(R.o1, U.dt_o1, c, v);
        if (U.fl & AEW_EF_OUT2_RELU) {
            float w[W];
#pragma unroll
            for (int r = 0; r < W; ++r) w[r] = v[r] > 0.f ? v[r] : 0.f;
            row_store<W>(R.o2, U.dt_o2, c, w);
        }
    }
}

template <int W>
__device__ __forceinline__ void epi_dfg(const EpiUni& U, const EpiRow& R, int n, const float dz[W]) {
    float pf[W], pg[W], df[W], dg[W];
    row_load<W>(R.a0, U.dt_a0, n, pf);
    row_load<W>(R.a1, U.dt_a1, n, pg);
#pragma unroll
    for (int r = 0; r < W; ++r) {
        df[r] = dz[r] * pf[r];
        dg[r] = dz[r] * pg[r];
    }
    const int np = (n >> 4) * 32 + (n & 15);           // W channels stay inside one 16-group
    row_store<W>(R.o0, U.dt_o0, np, df);
    row_store<W>(R.o0, U.dt_o0, np + 16, dg);
}

// =============================================================================================
// NT kernel, bf16: block tile 256 (rows m) x 128 (channels n), BK = 32.  Operand tiles go
// global -> LDS by 16-byte LDS-DMA into a 3-stage ring (3 x 24 KiB = 72 KiB, so TWO blocks are
// resident per CU and cover each other's barrier / epilogue stalls); tile t+2 is issued while
// tile t is computed, and the wait before the per-step barrier is a COUNTED vmcnt that leaves
// tile t+1's loads in flight (cdna_hip_programming.md T3+T4).  One raw s_barrier per K step.
//
// Two wave shapes (template MT = 16-row MFMA tiles per wave along m):
//   MT = 8  ("fat", default): 4 waves as 2(m) x 2(n), each 128 x 64 = 8x4 MFMA 16x16x32 tiles,
//           128 accumulator VGPRs, up to 256 VGPRs per wave (2 waves per SIMD over the two resident
//           blocks).  12 ds_read_b128 feed 32 MFMAs per K step: LDS fragment traffic per flop is
//           3/4 of the thin shape's, and the wide register budget lets all fragments of a step be
//           in flight before the first MFMA (the LDS array, not the MFMA pipe, was the co-critical
//           resource of the thin shape: 128 KiB of fragment reads + 48 KiB of DMA writes per K step
//           and CU against ~1100 MFMA cycles).
//   MT = 4  ("thin"): 8 waves as 4 x 2, each 64 x 64, 128 VGPRs per wave.
// =============================================================================================
#define NT_BM 256
#define NT_BN 128
#define NT_BK 32
#define NT_ROWB 64                                  // bytes per staged row (32 bf16)
#define NT_STAGES 3
// inline-asm forms used where instruction order / waits are placed by hand
#define AEW_DS_READ16(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr) : "memory")
#define AEW_MFMA_BF16(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#ifndef AEW_NT_COUNTED
#define AEW_NT_COUNTED 0     /* 1: thin shape with asm fragment reads, counted lgkmcnt waits and hand-placed
                                MFMA / LDS-DMA order.  Bit-identical, measured null (NT 4.84 vs 4.84 ms per
                                step), so the compiler-scheduled loop stays the default */
#endif
#ifndef AEW_TILE_ORDER_PROBE
#define AEW_TILE_ORDER_PROBE 0   /* 1 (tools library): aew_gemm_nt_t.reserved bits 16-17 pick another order of the row tiles */
#endif
#ifndef AEW_NT_SETPRIO
#define AEW_NT_SETPRIO 0     /* measured null on this structure (profiles/r01_notes.md) */
#endif

template <int MT, int NB = 1, int BMV = NT_BM>          // NB = block columns / 128, BMV = block rows (256 | 192)
struct NtCfg {
    static constexpr int BM = BMV;
    static constexpr int BN = 128 * NB;
    static constexpr int WAVES_M = BM / (16 * MT);      // 4 (thin; 192-row tiles: 48 x 64 per wave) or 2 (fat)
    static constexpr int WAVES_N = 2 * NB;
    static constexpr int NWAVES = WAVES_M * WAVES_N;
    static constexpr int THREADS = 64 * NWAVES;
    static constexpr int NXP = BM / 16, NWP = 8 * NB;   // 16-row pieces per K tile: X rows, W rows
    static constexpr int XP = (NXP + NWAVES - 1) / NWAVES;   // pieces a wave stages per K tile; piece indices
    static constexpr int WP = (NWP + NWAVES - 1) / NWAVES;   // beyond NXP / NWP are dummies (zero page -> pad)
    static constexpr int MINW = MT == 8 ? 2 : 4;   // waves per SIMD the register budget must allow
    static constexpr int PAD_OFF = (BM + BN) * NT_ROWB;      // 1 KiB landing pad of the dummy pieces
    static constexpr int STAGE_BYTES = PAD_OFF + ((NXP % NWAVES || NWP % NWAVES) ? 1024 : 0);
    static constexpr int LDS_BYTES = NT_STAGES * STAGE_BYTES;
};

// Per-lane source pointers of the X and W pieces a wave stages per K tile.  Computed once per
// segment (X) / once per kernel (W) and advanced by one K tile per issue, so the K loop carries no
// address arithmetic beyond 64-bit adds.
template <int MT, int NB = 1, int BMV = NT_BM>
struct NtPtrs {
    const char* x[NtCfg<MT, NB, BMV>::XP];
    const char* w[NtCfg<MT, NB, BMV>::WP];
    int xinc[NtCfg<MT, NB, BMV>::XP];
    int winc[NtCfg<MT, NB, BMV>::WP];
};

// after the last real K tile: the (uniform) loop keeps issuing, from the zero page, into stages
// nobody reads any more
template <int MT, int NB, int BMV>
__device__ __forceinline__ void nt_setup_idle(NtPtrs<MT, NB, BMV>& P) {
#pragma unroll
    for (int j = 0; j < NtCfg<MT, NB, BMV>::XP; ++j) { P.x[j] = reinterpret_cast<const char*>(aew_zero_page); P.xinc[j] = 0; }
#pragma unroll
    for (int j = 0; j < NtCfg<MT, NB, BMV>::WP; ++j) { P.w[j] = reinterpret_cast<const char*>(aew_zero_page); P.winc[j] = 0; }
}

// branch-free: the segment record is read with one batch of scalar loads, masked rows are selects
__device__ __forceinline__ const char* seg_row_ptr_sel(const aew_seg_t& s, int b, int m, int esize, bool& ok) {
    const int64_t row = (int64_t)m * s.row_step + s.row_off;
    ok = row >= s.row_lo && row < s.row_hi;
    return reinterpret_cast<const char*>(s.ptr) + ((int64_t)b * s.batch_stride + row * s.row_pitch) * esize;
}

template <int MT, int NB, int BMV>
__device__ __forceinline__ void nt_setup_x(const aew_gemm_nt_t& g, int seg, int b, int m0, int wave, int lane,
                                           NtPtrs<MT, NB, BMV>& P) {
    const aew_seg_t s = g.seg[seg];
    const int lr = lane >> 2, pc = lane & 3;
#pragma unroll
    for (int j = 0; j < NtCfg<MT, NB, BMV>::XP; ++j) {
        const int piece = wave * NtCfg<MT, NB, BMV>::XP + j;                  // BM / 16 pieces of 16 rows
        const int r = piece * 16 + lr;
        bool ok;
        const char* src = seg_row_ptr_sel(s, b, m0 + r, 2, ok) + (nt_swz64(r, pc) << 4);
        ok = ok && piece < NtCfg<MT, NB, BMV>::NXP;                           // dummy pieces read the zero page
        P.x[j] = ok ? src : reinterpret_cast<const char*>(aew_zero_page);
        P.xinc[j] = ok ? NT_BK * 2 : 0;
    }
}

// LDS row rho (0..63 inside a wave's 64-column slab) <- W row perm(rho).  With MFMA tile i = rho>>4
// and MFMA row q = rho&15 a lane (q>>2 = its 16-lane group) ends up holding, across the tile pair
// (2u, 2u+1), EIGHT consecutive channels: 16-byte epilogue accesses instead of 8-byte ones.
//   plain  : channel = u*32 + (q>>2)*8 + (i&1)*4 + (q&3)                       (u = i>>1)
//   gated  : tiles 0,1 = filt, tiles 2,3 = gate of channel c = (q>>2)*8 + (i&1)*4 + (q&3) in the
//            (16 filt | 16 gate)-interleaved packed order: (c>>4)*32 + (c&15) + 16*(i>>1)
template <int EPI>
__device__ __forceinline__ int nt_wperm(int rho) {
    const int i = rho >> 4, q = rho & 15;
    const int c = (q >> 2) * 8 + (i & 1) * 4 + (q & 3);
    if (EPI == AEW_EPI_GATED) return (c >> 4) * 32 + (c & 15) + 16 * (i >> 1);
    return (i >> 1) * 32 + c;
}

template <int EPI, int MT, int NB, int BMV>
__device__ __forceinline__ void nt_setup_w(const aew_gemm_nt_t& g, int n0, int wave, int lane, NtPtrs<MT, NB, BMV>& P) {
    const int lr = lane >> 2, pc = lane & 3;
#pragma unroll
    for (int j = 0; j < NtCfg<MT, NB, BMV>::WP; ++j) {
        const int piece = wave * NtCfg<MT, NB, BMV>::WP + j;                  // 8 * NB pieces of 16 rows
        const bool real = piece < NtCfg<MT, NB, BMV>::NWP;                    // (wave-uniform)
        const int r = piece * 16 + lr;
        const int src_row = (r & ~63) + nt_wperm<EPI>(r & 63);
        const char* src = reinterpret_cast<const char*>(g.W) + (int64_t)(n0 + src_row) * g.K_total * 2 + (nt_swz64(r, pc) << 4);
        P.w[j] = real ? src : reinterpret_cast<const char*>(aew_zero_page);
        P.winc[j] = real ? NT_BK * 2 : 0;
    }
}

template <int MT, int NB, int BMV, bool SC1X = false>       // SC1X: the activation pieces are read device-scope (chained stage)
__device__ __forceinline__ void nt_issue_bf16(char* stage, int wave, NtPtrs<MT, NB, BMV>& P) {
#pragma unroll
    for (int j = 0; j < NtCfg<MT, NB, BMV>::XP; ++j) {
        const int piece = wave * NtCfg<MT, NB, BMV>::XP + j;
        glds16_x<SC1X>(P.x[j], stage + (piece < NtCfg<MT, NB, BMV>::NXP ? piece * 1024 : NtCfg<MT, NB, BMV>::PAD_OFF));
        P.x[j] += P.xinc[j];
    }
#pragma unroll
    for (int j = 0; j < NtCfg<MT, NB, BMV>::WP; ++j) {
        typedef NtCfg<MT, NB, BMV> Cfg;
        const int piece = wave * Cfg::WP + j;
        glds16(P.w[j], stage + (piece < Cfg::NWP ? Cfg::BM * NT_ROWB + piece * 1024 : Cfg::PAD_OFF));
        P.w[j] += P.winc[j];
    }
}

struct NtIssue {                                    // walks K tiles across the segment table
    int seg, left, issued, slot;                    // left = K tiles still to issue from segment `seg`
};

// ---- split-phase forms for the MFMA kernels: the aux operands (bf16 in these kernels) arrive as raw
// 16-byte loads issued one row group AHEAD (nt_epilogue), so a row group's loads never sit behind the
// previous group's stores in the in-order vmcnt queue.
__device__ __forceinline__ void unpack8_bf16(const uint4& r, float o[8]) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        o[2 * q] = __uint_as_float(w[q] << 16);
        o[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u);
    }
}

template <bool WT = false>
__device__ __forceinline__ void epi_store8_pf(const EpiUni& U, const EpiRow& R, int n, float v[8], unsigned& zero_count,
                                              unsigned fl, const uint4& a0raw, const uint4& a1raw) {
    if (fl & AEW_EF_RELU) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
    }
    if (fl & AEW_EF_OUT1_PRE) row_store<8>(R.o1, U.dt_o1, n, v);
    if (fl & AEW_EF_ADD_AUX0) {
        float a[8];
        unpack8_bf16(a0raw, a);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = v[r] + a[r];
    }
    if (fl & AEW_EF_RELU_POST) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
    }
    if (fl & (AEW_EF_MUL_POS1 | AEW_EF_OUT1_POS1)) {
        float a[8], w[8];
        unpack8_bf16(a1raw, a);
#pragma unroll
        for (int r = 0; r < 8; ++r) w[r] = a[r] > 0.f ? v[r] : 0.f;
        if (fl & AEW_EF_OUT1_POS1) row_store<8>(R.o1, U.dt_o1, n, w);
        if (fl & AEW_EF_MUL_POS1) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = w[r];
        }
    }
    if ((fl & AEW_EF_COUNT_ZERO) && R.o0) {
#pragma unroll
        for (int r = 0; r < 8; ++r) zero_count += (n + r < U.N && v[r] == 0.f) ? 1u : 0u;
    }
    row_store<8, WT>(R.o0, U.dt_o0, n, v);
}

template <bool WT = false>
__device__ __forceinline__ void epi_dfg8_pf(const EpiUni& U, const EpiRow& R, int n, const float dz[8],
                                            const uint4& pfraw, const uint4& pgraw) {
    float pf[8], pg[8], df[8], dg[8];
    unpack8_bf16(pfraw, pf);
    unpack8_bf16(pgraw, pg);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        df[r] = dz[r] * pf[r];
        dg[r] = dz[r] * pg[r];
    }
    const int np = (n >> 4) * 32 + (n & 15);           // 8 channels stay inside one 16-group
    row_store<8, WT>(R.o0, U.dt_o0, np, df);
    row_store<8, WT>(R.o0, U.dt_o0, np + 16, dg);
}

// ---- epilogue of the bf16 NT kernels.  With the staging permutation nt_wperm, lane (fi, fg) holds
// for row m = m0 + wm*16*MT + j*16 + fi the 8 consecutive channels base + fg*8 + {0..7}:
// registers acc[2u][j][0..3] ++ acc[2u+1][j][0..3].
//
// Everything that does not depend on j is resolved ONCE per wave (EpiCtx): the descriptor lives in
// kernarg memory, and re-deriving the five view row pointers per 16-row group cost 54 serialized
// scalar-load round trips per wave = 25-32k cycles, 30 % of a block's lifetime (s_memtime phase
// clock, profiles/r02_notes.md).  Row j's pointer is then base + j * (16-row stride), and the
// biases of the lane's 8 (GATED: 8 + 8) channels are registers.
struct EpiViewCtx {
    const char* base;            // the view's buffer (wave-uniform)
    char* p;                     // lane's pointer for j = 0, nullptr if the view is absent
    int row;                     // lane's view row for j = 0
    int64_t inc;                 // bytes per 16 GEMM rows           (wave-uniform)
    int dstep, lo, hi;           // view rows per 16 GEMM rows, range (wave-uniform)
};

__device__ __forceinline__ EpiViewCtx epi_view_ctx(const aew_view_t& vref, int b, int m) {
    const aew_view_t v = vref;                                // one batch of scalar loads, no branches below
    EpiViewCtx c;
    const int es = v.dtype == AEW_BF16 ? 2 : 4;
    const int64_t row = (int64_t)m * v.row_step + v.row_off;
    c.row = (int)row;
    char* q = reinterpret_cast<char*>(v.ptr) + ((int64_t)b * v.batch_stride + row * v.row_pitch) * es;
    c.p = v.ptr ? q : nullptr;
    c.base = reinterpret_cast<const char*>(v.ptr);
    c.inc = (int64_t)16 * v.row_step * v.row_pitch * es;
    c.dstep = 16 * v.row_step;
    c.lo = (int)v.row_lo;
    c.hi = (int)v.row_hi;
    return c;
}

__device__ __forceinline__ char* epi_view_row(const EpiViewCtx& c, int j) {
    const int row = c.row + j * c.dstep;
    return (c.p && row >= c.lo && row < c.hi) ? c.p + j * c.inc : nullptr;
}

// (Round 4, measured and removed: the row loop below compiles to one `s_waitcnt vmcnt(0)` per 16-row group - its masks
// and flags are branches, and at a control-flow join the compiler's wait-count insertion gives up counting - so every
// group also waits for the stores of the one before.  A branch-free form of the gated / dz / STORE epilogues (masked
// stores into a sink, flags as selects: counted waits only, no store ever waited for; commit ae1bb12) changed nothing:
// 6.995 vs 6.968 ms per step.  The epilogue is bound by what it moves, not by how its instructions wait:
// tools/phase_clock.py, tools/overlap_probe.py, profiles/r04_notes.md.)
template <int EPI, bool ABL, int MT, bool WT = false>       // WT: out0 stored write-through (stage of a chained launch)
__device__ __forceinline__ void nt_epilogue(const aew_gemm_nt_t& g, f32x4_t (&acc)[4][MT], int b, int m0, int n0,
                                            int wm, int wn, int lane) {
    const int fi = lane & 15, fg = lane >> 4;
    const int mbase = m0 + wm * (16 * MT) + fi;
    // views each epilogue touches: GATED o0 o1 o2 | RES_SKIP o0 o1 o2 a0 | DFG o0 a0 a1 | STORE o0 o1 a0 a1
    const EpiViewCtx c0 = epi_view_ctx(g.out0, b, mbase);
    EpiViewCtx c1 = c0, c2 = c0, ca0 = c0, ca1 = c0;
    if (EPI != AEW_EPI_DFG) c1 = epi_view_ctx(g.out1, b, mbase);
    if (EPI == AEW_EPI_GATED || EPI == AEW_EPI_RES_SKIP) c2 = epi_view_ctx(g.out2, b, mbase);
    if (EPI != AEW_EPI_GATED) ca0 = epi_view_ctx(g.aux0, b, mbase);
    if (EPI == AEW_EPI_DFG || EPI == AEW_EPI_STORE) ca1 = epi_view_ctx(g.aux1, b, mbase);
    const EpiUni U = epi_uni(g);
    const unsigned fl = U.fl;
    unsigned zc = 0;
    // ---- biases of this lane's channels
    float bias_a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bias_b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int ch = ((n0 + wn * 64) >> 1) + 8 * fg;            // GATED: first of the lane's 8 channels
    if (EPI == AEW_EPI_GATED) {
        const int np_f = (ch >> 4) * 32 + (ch & 15);           // packed column of the filt half
        const float* bp = g.bias + (int64_t)b * g.bias_bs + np_f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 bf = *reinterpret_cast<const float4*>(bp + 4 * q);
            const float4 bg = *reinterpret_cast<const float4*>(bp + 16 + 4 * q);
            bias_a[4 * q] = bf.x; bias_a[4 * q + 1] = bf.y; bias_a[4 * q + 2] = bf.z; bias_a[4 * q + 3] = bf.w;
            bias_b[4 * q] = bg.x; bias_b[4 * q + 1] = bg.y; bias_b[4 * q + 2] = bg.z; bias_b[4 * q + 3] = bg.w;
        }
    } else if (EPI == AEW_EPI_STORE && (fl & AEW_EF_BIAS)) {
        // folded into the accumulators right away (the same fp32 add the row loop would do), so no
        // bias registers stay live next to the prefetched aux rows
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int n = n0 + wn * 64 + u * 32 + 8 * fg;
            if (n < U.N) {
                const float* bp = g.bias + (int64_t)b * g.bias_bs + n;
                const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    acc[2 * u][j][0] += b0.x; acc[2 * u][j][1] += b0.y; acc[2 * u][j][2] += b0.z; acc[2 * u][j][3] += b0.w;
                    acc[2 * u + 1][j][0] += b1.x; acc[2 * u + 1][j][1] += b1.y; acc[2 * u + 1][j][2] += b1.z; acc[2 * u + 1][j][3] += b1.w;
                }
            }
        }
    }
    // aux operands (STORE: aux0 = addend, aux1 = relu-mask source; DFG: aux0 = dz/df, aux1 = dz/dg) are
    // prefetched one (row group, channel octet) step ahead: step s = 2j + u
    constexpr bool PF = (EPI == AEW_EPI_STORE || EPI == AEW_EPI_DFG);
    const bool need0 = PF && (EPI == AEW_EPI_DFG || (fl & AEW_EF_ADD_AUX0));
    const bool need1 = PF && (EPI == AEW_EPI_DFG || (fl & (AEW_EF_MUL_POS1 | AEW_EF_OUT1_POS1)));
    // STORE keeps ONE prefetched operand in flight (register budget of the 128-VGPR shape): aux0 if
    // it is used, else aux1; with both in use aux1 is loaded in step.  DFG prefetches both.
    const bool pf1 = need1 && (EPI == AEW_EPI_DFG || !need0);
    uint4 raw0[2 * MT], raw1[2 * MT];
    // chained stage (WT): the aux operands may be rows an earlier stage of the SAME launch stored write-through (the residual
    // addend x_l of G2, dx of the layer above in dx): device-scope loads, masked rows through an out-of-range offset
    const __amdgpu_buffer_rsrc_t rs0 = buf_rsrc(WT ? ca0.base : nullptr), rs1 = buf_rsrc(WT ? ca1.base : nullptr);
    auto aux_get = [&](const EpiViewCtx& c, const __amdgpu_buffer_rsrc_t& rs, int j, int n) -> uint4 {
        const char* p = epi_view_row(c, j);
        if (WT) return ld16_sc1(rs, p ? (uint32_t)(p - c.base) + (uint32_t)(n * 2) : AEW_BUF_OOB);
        const char* z = reinterpret_cast<const char*>(aew_zero_region);
        return *reinterpret_cast<const uint4*>((p ? p : z) + n * 2);
    };
    auto aux_load = [&](int sidx) {
        const int j = sidx >> 1, u = sidx & 1;
        const int n = n0 + wn * 64 + u * 32 + 8 * fg;
        raw0[sidx] = make_uint4(0, 0, 0, 0);
        raw1[sidx] = make_uint4(0, 0, 0, 0);
        if (need0) raw0[sidx] = aux_get(ca0, rs0, j, n);
        if (EPI == AEW_EPI_DFG) raw1[sidx] = aux_get(ca1, rs1, j, n);
        else if (pf1) raw0[sidx] = aux_get(ca1, rs1, j, n);
    };
    if (PF) aux_load(0);
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const bool row_ok = mbase + j * 16 < g.M;
        EpiRow R;
        R.o0 = epi_view_row(c0, j);
        R.o1 = (EPI != AEW_EPI_DFG) ? epi_view_row(c1, j) : nullptr;
        R.o2 = (EPI == AEW_EPI_GATED || EPI == AEW_EPI_RES_SKIP) ? epi_view_row(c2, j) : nullptr;
        R.a0 = (EPI == AEW_EPI_RES_SKIP) ? epi_view_row(ca0, j) : nullptr;
        R.a1 = nullptr;
        if (EPI == AEW_EPI_GATED) {
            // wave slab = 64 packed columns = 32 channels; tiles 0,1 filt / 2,3 gate
            const float f[8] = {acc[0][j][0], acc[0][j][1], acc[0][j][2], acc[0][j][3],
                                acc[1][j][0], acc[1][j][1], acc[1][j][2], acc[1][j][3]};
            const float q[8] = {acc[2][j][0], acc[2][j][1], acc[2][j][2], acc[2][j][3],
                                acc[3][j][0], acc[3][j][1], acc[3][j][2], acc[3][j][3]};
            if (row_ok && ch < U.N) epi_gated<8, ABL, WT>(g, R, ch, f, q, bias_a, bias_b);
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (PF && 2 * j + u + 1 < 2 * MT) aux_load(2 * j + u + 1);
                const int n = n0 + wn * 64 + u * 32 + 8 * fg;
                float v[8] = {acc[2 * u][j][0], acc[2 * u][j][1], acc[2 * u][j][2], acc[2 * u][j][3],
                              acc[2 * u + 1][j][0], acc[2 * u + 1][j][1], acc[2 * u + 1][j][2], acc[2 * u + 1][j][3]};
                if (row_ok && n < U.N) {
                    if (EPI == AEW_EPI_STORE) {
                        uint4 a1 = raw0[2 * j + u];                       // aux1 rode in raw0 (pf1) ...
                        if (need1 && !pf1) a1 = aux_get(ca1, rs1, j, n);   // ... or is fetched now (both operands in use)
                        epi_store8_pf<WT>(U, R, n, v, zc, fl, raw0[2 * j + u], a1);
                    }
                    else if (EPI == AEW_EPI_RES_SKIP) epi_res_skip<8>(U, R, n, v);
                    else epi_dfg8_pf<WT>(U, R, n, v, raw0[2 * j + u], raw1[2 * j + u]);
                }
            }
        }
    }
    if (EPI == AEW_EPI_STORE && (fl & AEW_EF_COUNT_ZERO)) {
        zc = (unsigned)wave_sum((float)zc);
        if (lane == 0 && zc) atomicAdd(g.counter, (unsigned long long)zc);
    }
}

// ABL = true builds the ablation variant used by tools/ablate_gemm.py (switches in g.reserved).  It is instantiated
// in the tools library only (hipcc -DAEW_FN_ABLATE=1 -o lib/libaewavenet_hip_abl.so); the product library has the
// ABL = false instantiations, which contain none of that code and ignore g.reserved.
// ST = ring depth (round 4: deeper rings at one block per CU measured slower, aew_set_nt_deep; profiles/r04_notes.md §2).
template <int N>
__device__ __forceinline__ void nt_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one output tile: block index L of the launch's tile order
template <int EPI, bool ABL, int MT, int NB, int BMV, int ST, bool WT = false>
__device__ __forceinline__ void nt_tile(const aew_gemm_nt_t& g, char* smem, const int L_) {
    typedef NtCfg<MT, NB, BMV> Cfg;
    static_assert(ST >= 3 && (ST - 2) * (Cfg::XP + Cfg::WP) <= 63 && ST * Cfg::STAGE_BYTES <= 160 * 1024, "ring depth");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % Cfg::WAVES_N, wm = wave / Cfg::WAVES_N;
    // XCD-aware tile order.  Workgroup L runs on XCD L % 8 (observed dispatch rule; used for
    // speed only).  All N tiles of one (batch, row-tile) are consecutive on ONE XCD, so the
    // activation tile is fetched from HBM into that XCD's L2 once and re-hit by the others.
    const int n_mt = (g.M + Cfg::BM - 1) / Cfg::BM, n_nt = g.N_pad / Cfg::BN;
    const int L = L_, seq = L >> 3;
    int gi = seq / n_nt;                                       // group of 8 consecutive row tiles (one per XCD)
#if AEW_TILE_ORDER_PROBE
    {   // diagnostic (tools/b16_probe.py): other orders of the row-tile groups, selected by reserved bits 16-17
        const int ord = (g.reserved >> 16) & 3, ng = (n_mt * g.batch + 7) / 8;
        if (ord == 1) gi = (gi & 1) * ((ng + 1) / 2) + (gi >> 1);            // first and second half of the rows interleaved
        else if (ord == 2) gi = ng - 1 - gi;                                  // reversed
        else if (ord == 3) gi = (gi & 3) * ((ng + 3) / 4) + (gi >> 2);        // four quarters interleaved
        if (gi >= ng) return;
    }
#endif
    const int rt = gi * 8 + (L & 7);
    if (rt >= n_mt * g.batch) return;
    const int b = rt / n_mt;
    const int m0 = (rt - b * n_mt) * Cfg::BM, n0 = (seq % n_nt) * Cfg::BN;
    // RES_SKIP: skip-part tiles that lie entirely before the skip window do nothing
    if (EPI == AEW_EPI_RES_SKIP && n0 >= g.n_split) {
        const int64_t last = (int64_t)(min(m0 + Cfg::BM, g.M) - 1) * g.out1.row_step + g.out1.row_off;
        if (last < g.out1.row_lo) return;
    }
    const int nkt = g.K_total / NT_BK;
    const int abl = ABL ? g.reserved : 0;              // ablation switches (tools/ablate_gemm.py)
    if (abl & 32) return;                              // launch cost only
    // abl & 1024: s_memtime phase clock -> g.counter[0..5] = cycles in {prologue, vmcnt wait, barrier,
    // LDS fragment wait, MFMA + DMA issue, epilogue}, [6] = blocks (wave 0 of every block adds its sums)
    const bool clk = ABL && (abl & 1024);
    unsigned tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};          // SGPR-resident (wave-uniform) 32-bit cycle sums
    unsigned t_prev = clk ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
    auto lap = [&](int k) {
        if (!clk) return;
        const unsigned now = (unsigned)__builtin_amdgcn_s_memtime();
        tk[k] += now - t_prev;
        t_prev = now;
    };
    f32x4_t acc[4][MT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    NtPtrs<MT, NB, BMV> P;
    NtIssue is = {0, g.seg[0].k_len / NT_BK, 0, 0};
    nt_setup_w<EPI, MT, NB, BMV>(g, n0, wave, lane, P);
    nt_setup_x<MT, NB, BMV>(g, 0, b, m0, wave, lane, P);
    // One K tile is issued per loop step, unconditionally, so the loop body is a single basic block
    // that the scheduler directives below can shape; `advance` runs after a tile has been issued
    // and points P at the next one (next segment, or the zero page once K is exhausted).
    auto advance = [&]() {
        --is.left;
        ++is.issued;
        is.slot = (is.slot + 1 == ST) ? 0 : is.slot + 1;
        if (is.left == 0 && !(abl & 128)) {            // wave-uniform and rare: scalar loads only here
            if (is.issued >= nkt) {
                nt_setup_idle<MT, NB, BMV>(P);
                is.left = 1 << 30;
            } else {
                ++is.seg;
                is.left = g.seg[is.seg].k_len / NT_BK;
                nt_setup_x<MT, NB, BMV>(g, is.seg, b, m0, wave, lane, P);
            }
        }
    };
    if (abl & 16) return;                              // launch + pointer setup
#pragma unroll
    for (int q = 0; q < ST - 1; ++q) {                 // tiles 0 .. ST-2 in flight
        if (!(abl & 4)) nt_issue_bf16<MT, NB, BMV, WT>(smem + q * Cfg::STAGE_BYTES, wave, P);
        advance();
    }
    const int fi = lane & 15, fg = lane >> 4;
    // fragment byte offsets inside a stage (constant over the K loop)
    int woff[4], xoff[MT];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rw = wn * 64 + i * 16 + fi;
        woff[i] = Cfg::BM * NT_ROWB + rw * NT_ROWB + (nt_swz64(rw, fg) << 4);
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int rx = wm * (16 * MT) + j * 16 + fi;
        xoff[j] = rx * NT_ROWB + (nt_swz64(rx, fg) << 4);
    }
    int stage = 0;
    lap(0);
    for (int t = 0; t < nkt; ++t) {
        // tile t has landed once at most the loads of tiles t+1 .. t+ST-2 (XP + WP per wave each) are outstanding
        nt_wait_vm<(ST - 2) * (Cfg::XP + Cfg::WP)>();
        lap(1);
        if (!(abl & 64)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        lap(2);
        const char* st = smem + stage * Cfg::STAGE_BYTES;
        stage = (stage + 1 == ST) ? 0 : stage + 1;
        {
            bf16x8_t wf[4], xf[MT];
            constexpr bool COUNTED = AEW_NT_COUNTED && !ABL && MT == 4;
            if constexpr (COUNTED) {
                // fragment reads as inline asm in consumption order, so the waits below can be COUNTED: the
                // first MFMA group needs W0-3 + X0 (5 of 8 reads), each later group one more X fragment.  (The
                // compiler waits lgkmcnt(0) before the first MFMA whenever it sees the reads itself.)
                const uint32_t sb = (uint32_t)(uintptr_t)AEW_LDS_PTR(st);
                const uint32_t wa = sb + woff[0], xa = sb + xoff[0];       // tile i / j adds i*1024 bytes
                AEW_DS_READ16(wf[0], wa, 0); AEW_DS_READ16(wf[1], wa, 1024);
                AEW_DS_READ16(wf[2], wa, 2048); AEW_DS_READ16(wf[3], wa, 3072);
                AEW_DS_READ16(xf[0], xa, 0); AEW_DS_READ16(xf[1], xa, 1024);
                AEW_DS_READ16(xf[2], xa, 2048); AEW_DS_READ16(xf[3], xa, 3072);
            } else if (!(abl & 2)) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8_t*>(st + woff[i]);
#pragma unroll
                for (int j = 0; j < MT; ++j) xf[j] = *reinterpret_cast<const bf16x8_t*>(st + xoff[j]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = __builtin_bit_cast(bf16x8_t, (s16x8_t){1, 2, 3, 4, 5, 6, 7, (short)t});
#pragma unroll
                for (int j = 0; j < MT; ++j) xf[j] = wf[j & 3];
            }
            if (clk) {                                 // force the fragment reads to complete here
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[i]));
#pragma unroll
                for (int j = 0; j < MT; ++j) asm volatile("" : "+v"(xf[j]));
                lap(3);
            }
            // tile t+ST-1 goes into the stage that was computed at step t-1 (every wave is past it: barrier)
            if (!COUNTED && !(abl & 4)) nt_issue_bf16<MT, NB, BMV, WT>(smem + is.slot * Cfg::STAGE_BYTES, wave, P);
            if constexpr (COUNTED) {
                // hand-placed: wait for what the group needs, 4 MFMAs, one LDS-DMA piece of tile t+2
                // (asm MFMAs with a memory clobber so that neither they nor the DMA builtins move)
                char* dstage = smem + is.slot * Cfg::STAGE_BYTES;
                auto piece = [&](int q) {
                    if (q < Cfg::XP) {
                        glds16(P.x[q], dstage + (wave * Cfg::XP + q) * 1024);
                        P.x[q] += P.xinc[q];
                    } else if (q < Cfg::XP + Cfg::WP) {
                        const int w = q - Cfg::XP;
                        glds16(P.w[w], dstage + Cfg::BM * NT_ROWB + (wave * Cfg::WP + w) * 1024);
                        P.w[w] += P.winc[w];
                    }
                };
#define AEW_MFMA4(J)                                                                                          \
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %8, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %5, %8, %1\n\t" \
                             "v_mfma_f32_16x16x32_bf16 %2, %6, %8, %2\n\tv_mfma_f32_16x16x32_bf16 %3, %7, %8, %3"        \
                             : "+v"(acc[0][J]), "+v"(acc[1][J]), "+v"(acc[2][J]), "+v"(acc[3][J])                     \
                             : "v"(wf[0]), "v"(wf[1]), "v"(wf[2]), "v"(wf[3]), "v"(xf[J]) : "memory")
                asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(xf[0]));
                AEW_MFMA4(0);
                piece(0);
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xf[1]));
                AEW_MFMA4(1);
                piece(1);
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(xf[2]));
                AEW_MFMA4(2);
                piece(2);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[3]));
                AEW_MFMA4(3);
#undef AEW_MFMA4
                static_assert(!COUNTED || Cfg::XP + Cfg::WP == 3, "three pieces per wave and K tile in the thin shape");
            } else if (!(abl & 1)) {
                if (AEW_NT_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int j = 0; j < MT; ++j)           // j outer: X fragments are consumed in arrival order
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
                if (AEW_NT_SETPRIO) __builtin_amdgcn_s_setprio(0);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(wf[i]));
#pragma unroll
                for (int j = 0; j < MT; ++j) asm volatile("" ::"v"(xf[j]));
            }
            if (!ABL && !COUNTED) {
                // shape of the step: every fragment read in flight first, then the MFMAs with the
                // LDS-DMA pieces of tile t+2 threaded between them (one piece per 4 MFMAs)
                __builtin_amdgcn_sched_group_barrier(0x100, 4 + MT, 0);            // DS reads
#pragma unroll
                for (int q = 0; q < Cfg::XP + Cfg::WP; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);             // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);             // VMEM (LDS-DMA)
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT - 4 * (Cfg::XP + Cfg::WP), 0);
            }
        }
        advance();
        lap(4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the idle-tail LDS-DMA must land before the LDS is released
    if (AEW_NT_COUNTED && !ABL && MT == 4) {           // asm MFMAs are invisible to the compiler's hazard logic
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) asm volatile("" : "+v"(acc[i][j]));
    }
    if (clk) {
        lap(5);                                        // drain of the idle-tail DMA
        nt_epilogue<EPI, ABL, MT>(g, acc, b, m0, n0, wm, wn, lane);
        lap(6);                                        // epilogue instructions issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lap(7);                                        // stores retired
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) atomicAdd(g.counter + k, (unsigned long long)tk[k]);
            atomicAdd(g.counter + 8, 1ull);
        }
        return;
    }
    if (abl & 8) {                                     // keep the accumulators live, skip the epilogue
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    nt_epilogue<EPI, ABL, MT, WT>(g, acc, b, m0, n0, wm, wn, lane);
}

template <int EPI, bool ABL = false, int MT = 8, int NB = 1, int BMV = NT_BM, int ST = NT_STAGES>
__global__ __launch_bounds__((NtCfg<MT, NB, BMV>::THREADS), (NtCfg<MT, NB, BMV>::MINW)) void k_gemm_nt_bf16(const aew_gemm_nt_t g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    nt_tile<EPI, ABL, MT, NB, BMV, ST>(g, smem, blockIdx.x);
}

// (Round 4, measured and removed: the same tiles as PERSISTENT blocks - a grid of two blocks per CU, each walking the tile
// order with stride gridDim.x, no workgroup dispatched after the first wave - G2 33.1 vs 31.2 us, at B = 16 62.3 vs 53.4: the
// ~10 us a launch costs beyond its tiles is not workgroup dispatch.  profiles/r04_notes.md §13.)

// =============================================================================================
// Software-pipelined form of the fat-wave kernel ("pipe"): same tiles, waves, LDS ring and results
// as k_gemm_nt_bf16<.., 8, NB>, but the K loop is hand-scheduled:
//   * fragments are DOUBLE-BUFFERED IN REGISTERS (2 x 48 VGPRs): the 12 ds_read_b128 of tile t+1 are
//     issued right after the step barrier and land under the 32 MFMAs of tile t, so the post-barrier
//     burst in which every wave of the CU reads its fragments at once (96-128 KiB at 256 B/clk) no
//     longer idles the MFMA pipe;
//   * an LDS stage is free as soon as its tile is in registers, so the 3-stage ring carries a
//     prefetch distance of 3 tiles (tile t+3 is issued during step t);
//   * reads, LDS-DMA pieces and MFMAs are inline asm in exactly the order written (one DMA piece per
//     4 MFMAs); the only waits are one counted vmcnt + lgkmcnt(0) at the top of a step.
// K_total is a multiple of 64 (ABI), so the step count is even and the loop is unrolled by two
// (register sets A and B swap roles).
// =============================================================================================

template <int MT>
__device__ __forceinline__ void nt_read_frags(uint32_t wa, uint32_t xa, bf16x8_t (&wf)[4], bf16x8_t (&xf)[MT]) {
    static_assert(MT == 8, "fat waves only");
    AEW_DS_READ16(wf[0], wa, 0);    AEW_DS_READ16(xf[0], xa, 0);
    AEW_DS_READ16(wf[1], wa, 1024); AEW_DS_READ16(wf[2], wa, 2048); AEW_DS_READ16(wf[3], wa, 3072);
    AEW_DS_READ16(xf[1], xa, 1024); AEW_DS_READ16(xf[2], xa, 2048); AEW_DS_READ16(xf[3], xa, 3072);
    AEW_DS_READ16(xf[4], xa, 4096); AEW_DS_READ16(xf[5], xa, 5120); AEW_DS_READ16(xf[6], xa, 6144);
    AEW_DS_READ16(xf[7], xa, 7168);
}

// all 12 fragment registers of a set are valid after this (ties the asm reads to the MFMAs below)
__device__ __forceinline__ void nt_frags_ready(bf16x8_t (&wf)[4], bf16x8_t (&xf)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]),
                   "+v"(xf[3]), "+v"(xf[4]), "+v"(xf[5]), "+v"(xf[6]), "+v"(xf[7]));
}

template <int EPI, int NB>
__global__ __launch_bounds__((NtCfg<8, NB>::THREADS), 2) void k_gemm_nt_bf16_pipe(const aew_gemm_nt_t g) {
    constexpr int MT = 8;
    typedef NtCfg<MT, NB> Cfg;
    constexpr int PIECES = Cfg::XP + Cfg::WP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % Cfg::WAVES_N, wm = wave / Cfg::WAVES_N;
    const int n_mt = (g.M + NT_BM - 1) / NT_BM, n_nt = g.N_pad / Cfg::BN;
    const int L = blockIdx.x, seq = L >> 3;                  // XCD-aware order, see k_gemm_nt_bf16
    const int rt = (seq / n_nt) * 8 + (L & 7);
    if (rt >= n_mt * g.batch) return;
    const int b = rt / n_mt;
    const int m0 = (rt - b * n_mt) * NT_BM, n0 = (seq % n_nt) * Cfg::BN;
    if (EPI == AEW_EPI_RES_SKIP && n0 >= g.n_split) {
        const int64_t last = (int64_t)(min(m0 + NT_BM, g.M) - 1) * g.out1.row_step + g.out1.row_off;
        if (last < g.out1.row_lo) return;
    }
    const int nkt = g.K_total / NT_BK;                       // even
    f32x4_t acc[4][MT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    NtPtrs<MT, NB> P;
    NtIssue is = {0, g.seg[0].k_len / NT_BK, 0, 0};
    nt_setup_w<EPI, MT, NB, NT_BM>(g, n0, wave, lane, P);
    nt_setup_x<MT, NB, NT_BM>(g, 0, b, m0, wave, lane, P);
    auto advance = [&]() {
        --is.left;
        ++is.issued;
        is.slot = (is.slot + 1 == NT_STAGES) ? 0 : is.slot + 1;
        if (is.left == 0) {                                  // wave-uniform and rare
            if (is.issued >= nkt) {
                nt_setup_idle<MT, NB, NT_BM>(P);
                is.left = 1 << 30;
            } else {
                ++is.seg;
                is.left = g.seg[is.seg].k_len / NT_BK;
                nt_setup_x<MT, NB, NT_BM>(g, is.seg, b, m0, wave, lane, P);
            }
        }
    };
    // prologue: tiles 0, 1, 2 in flight; tile 0 into register set A
#pragma unroll
    for (int q = 0; q < NT_STAGES; ++q) {
        nt_issue_bf16<MT, NB, NT_BM>(smem + is.slot * Cfg::STAGE_BYTES, wave, P);
        advance();
    }
    const int fi = lane & 15, fg = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)AEW_LDS_PTR(smem);
    const int rw = wn * 64 + fi, rx = wm * (16 * MT) + fi;   // tile i / j adds i*16 rows = i*1024 bytes, same swizzle
    const uint32_t wlane = lds0 + NT_BM * NT_ROWB + rw * NT_ROWB + (nt_swz64(rw, fg) << 4);
    const uint32_t xlane = lds0 + rx * NT_ROWB + (nt_swz64(rx, fg) << 4);
    bf16x8_t wA[4], xA[MT], wB[4], xB[MT];
    if (PIECES == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    nt_read_frags<MT>(wlane, xlane, wA, xA);
    uint32_t rd = Cfg::STAGE_BYTES;                           // byte offset of the stage holding tile t+1

#define AEW_PIPE_STEP(WC, XC, WN_, XN_)                                                                    \
    do {                                                                                                   \
        /* tile t+1 has landed (mine): only tile t+2's pieces may still be in flight */                    \
        if (PIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                  \
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                              \
        nt_frags_ready(WC, XC);                               /* tile t is in registers */                 \
        __builtin_amdgcn_s_barrier();                         /* everyone's: t+1 visible, stage of t free */ \
        nt_read_frags<MT>(wlane + rd, xlane + rd, WN_, XN_);                                               \
        rd = (rd + Cfg::STAGE_BYTES == NT_STAGES * Cfg::STAGE_BYTES) ? 0u : rd + Cfg::STAGE_BYTES;        \
        char* dst = smem + is.slot * Cfg::STAGE_BYTES;        /* tile t+3 -> the stage tile t came from */ \
        _Pragma("unroll") for (int j = 0; j < MT; ++j) {                                                   \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) AEW_MFMA_BF16(acc[i][j], WC[i], XC[j]);          \
            if (j < Cfg::XP) {                                                                             \
                glds16(P.x[j], dst + (wave * Cfg::XP + j) * 1024);                                         \
                P.x[j] += P.xinc[j];                                                                       \
            } else if (j < PIECES) {                                                                       \
                glds16(P.w[j - Cfg::XP], dst + NT_BM * NT_ROWB + (wave * Cfg::WP + (j - Cfg::XP)) * 1024); \
                P.w[j - Cfg::XP] += P.winc[j - Cfg::XP];                                                                \
            }                                                                                              \
        }                                                                                                  \
        advance();                                                                                         \
    } while (0)

    for (int t = 0; t < nkt; t += 2) {
        AEW_PIPE_STEP(wA, xA, wB, xB);
        AEW_PIPE_STEP(wB, xB, wA, xA);
    }
#undef AEW_PIPE_STEP
    // drain: idle-tail DMA and the unused last fragment reads must land before LDS / registers are
    // reused; the asm MFMAs are invisible to the compiler's hazard logic, so pad before reading acc
    nt_frags_ready(wA, xA);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) asm volatile("" : "+v"(acc[i][j]));
    nt_epilogue<EPI, false, MT>(g, acc, b, m0, n0, wm, wn, lane);
}

// =============================================================================================
// "p64": the software-pipelined kernel with K tiles of 64 (LDS rows of 128 B = ONE FULL L2 LINE per
// operand row and tile).  With 64-byte rows every 128-byte line is pulled from L2 twice, by two
// consecutive K steps, because a step of both resident blocks stages 96 KiB of lines through a 32 KiB
// vector L1: the staging traffic at L2 was 2x the operand bytes and L2 bandwidth, not MFMA or LDS,
// bounded the K loop (DMA-only ablation 38-48 us vs 27 us of MFMA for a gated layer;
// profiles/r02_notes.md).
//   tile 256 x 256, 8 fat waves (2 x 4, each 128 x 64), one block per CU, 2 LDS stages of 64 KiB
//   a tile = two K sub-steps of 32 (register sets A / B); reads of the next sub-step are issued
//   before the MFMAs of the current one; ONE barrier per tile; the DMA of tile T+2 is issued under
//   the MFMAs of sub-step T.b into the stage tile T just left.
//   LDS row swizzle: 16-byte chunk c of row r sits at chunk c ^ ((r >> 1) & 7).
// =============================================================================================
template <int MT, int WM, int WN>
struct P64Cfg {
    static constexpr int BM = WM * 16 * MT, BN = WN * 64, NW = WM * WN, THREADS = 64 * NW;
    static constexpr int XP = BM / 8 / NW, WP = BN / 8 / NW, PIECES = XP + WP;   // 8-row pieces per wave and tile
    static constexpr int STAGE_BYTES = (BM + BN) * 128, LDS_BYTES = 2 * STAGE_BYTES;
};

template <int EPI>
__device__ __forceinline__ int p64_wpiece_row(int p) {       // W source row offset of 8-row piece p (0..7) in a 64-row slab
    if (EPI == AEW_EPI_GATED) return (p & 1) * 32 + (p >> 2) * 16 + ((p >> 1) & 1) * 4;
    return (p >> 2) * 32 + (p & 1) * 16 + ((p >> 1) & 1) * 4;
}

template <int XP>
struct P64Ptrs {
    const char* x[XP];                                       // per-lane source of the wave's X pieces
    const char* w;                                           // per-lane W base; piece j adds woff[j] (wave-uniform)
};

template <int XP>
__device__ __forceinline__ void p64_setup_x(const aew_gemm_nt_t& g, int seg, int b, int m0, int wave, int lane,
                                            P64Ptrs<XP>& P) {
    const aew_seg_t s = g.seg[seg];
    const int lr = lane >> 3, pos = lane & 7;
#pragma unroll
    for (int j = 0; j < XP; ++j) {
        const int r = (wave * XP + j) * 8 + lr;                      // pieces of 8 rows
        bool ok;
        const char* src = seg_row_ptr_sel(s, b, m0 + r, 2, ok) + ((pos ^ ((r >> 1) & 7)) << 4);
        P.x[j] = ok ? src : reinterpret_cast<const char*>(aew_zero_region);
    }
}

#define P64_READ4(WF, XF, WA_, XA_)                                                                         \
    do {                                                                                                    \
        AEW_DS_READ16(WF[0], WA_, 0);    AEW_DS_READ16(XF[0], XA_, 0);                                      \
        AEW_DS_READ16(WF[1], WA_, 2048); AEW_DS_READ16(WF[2], WA_, 4096); AEW_DS_READ16(WF[3], WA_, 6144);  \
        AEW_DS_READ16(XF[1], XA_, 2048); AEW_DS_READ16(XF[2], XA_, 4096); AEW_DS_READ16(XF[3], XA_, 6144);  \
    } while (0)
#define P64_READ1(WF, XF, WA_, XA_)                                                                         \
    do {                                                                                                    \
        AEW_DS_READ16(WF[0], WA_, 0);    AEW_DS_READ16(XF[0], XA_, 0);                                      \
        AEW_DS_READ16(WF[1], WA_, 2048); AEW_DS_READ16(WF[2], WA_, 4096); AEW_DS_READ16(WF[3], WA_, 6144);  \
    } while (0)
#define P64_READ8(WF, XF, WA_, XA_)                                                                         \
    do {                                                                                                    \
        P64_READ4(WF, XF, WA_, XA_);                                                                        \
        AEW_DS_READ16(XF[4], XA_, 8192); AEW_DS_READ16(XF[5], XA_, 10240); AEW_DS_READ16(XF[6], XA_, 12288); \
        AEW_DS_READ16(XF[7], XA_, 14336);                                                                   \
    } while (0)

template <int MT>
__device__ __forceinline__ void p64_frags_ready(bf16x8_t (&wf)[4], bf16x8_t (&xf)[MT]) {
    if constexpr (MT == 8) nt_frags_ready(wf, xf);
    else if constexpr (MT == 1)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(xf[0]));
    else asm volatile("s_waitcnt lgkmcnt(0)"
                      : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]));
}

// MT x WM x WN:  8 x 2 x 4 = 256 x 256 tile, 8 fat waves, one block per CU (long-K ops)
//                4 x 2 x 2 = 128 x 128 tile, 4 waves of 64 x 64, two blocks per CU
//                4 x 4 x 2 = 256 x 128 tile, 8 waves of 64 x 64, one block per CU (96 KiB)
//                4 x 1 x 2 =  64 x 128 tile, 2 waves, three blocks per CU: launches with few rows (the
//                            upsampler dgrads have 56 tiles of 256 rows for 256 CUs)
//                1 x 4 x 2 =  64 x 128 tile, 8 waves of 16 rows x 64 channels: the same tile for launches of <= 256
//                            blocks.  With 2 waves a K tile is 12 LDS-DMA instructions per wave at 60-185 cycles
//                            of issue each (MI355X_MICROARCH.md) against 512 cycles of MFMA: the two waves spent
//                            their time issuing loads (1870 cycles per tile measured).  8 waves issue 3 each.
//                1 x 4 x 1 =  64 x  64 tile, 4 waves: launches of so few blocks that even the 64 x 128 shape leaves most
//                            CUs idle (the encoder / upsampler dgrads: 4-54 blocks).  A lone block fills its LDS at
//                            ~30 GB/s whatever its ring, so its K loop lasts bytes-per-block / 30 GB/s: half the W rows
//                            per block = 2/3 of the bytes, twice the blocks (aew_set_nt_small_n64)
//   S = ring depth.  2: tile T+2 is issued one tile time ahead (the long-K shapes are MFMA-bound anyway).  The 64-row
//   shape runs launches of a few dozen blocks whose K loop is pure DMA latency at depth 2 (36 tiles x ~1 us for the
//   encoder dgrads against 0.2 us of MFMAs per tile); S = 5 keeps four tiles in flight (counted vmcnt).
template <int N>
__device__ __forceinline__ void p64_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int EPI, int MT, int WM, int WN, int S = 2>
__global__ __launch_bounds__((P64Cfg<MT, WM, WN>::THREADS), 2) void k_gemm_nt_bf16_p64(const aew_gemm_nt_t g) {
    typedef P64Cfg<MT, WM, WN> Cfg;
    static_assert(S >= 2 && (S - 1) * Cfg::PIECES <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WN, wm = wave / WN;
    const int n_mt = (g.M + Cfg::BM - 1) / Cfg::BM, n_nt = g.N_pad / Cfg::BN;
    // split-K of a small launch (aew_gemm_nt_t.k_split, the 64 x 64 shape only: see the seam below): the grid is ks copies of
    // the tile grid, copy sp contracts K tiles [kt0, kt0 + n_tiles)
    const int ks = (S == 5 && MT == 1 && WN == 1 && g.k_split > 1) ? g.k_split : 1;
    const int grid1 = (int)gridDim.x / ks;
    const int sp = (int)blockIdx.x / grid1;
    const int L = (int)blockIdx.x - sp * grid1, seq = L >> 3;                  // XCD-aware order, see k_gemm_nt_bf16
    const int rt = (seq / n_nt) * 8 + (L & 7);
    if (rt >= n_mt * g.batch) return;
    const int b = rt / n_mt;
    const int m0 = (rt - b * n_mt) * Cfg::BM, n0 = (seq % n_nt) * Cfg::BN;
    if (EPI == AEW_EPI_RES_SKIP && n0 >= g.n_split) {
        const int64_t last = (int64_t)(min(m0 + Cfg::BM, g.M) - 1) * g.out1.row_step + g.out1.row_off;
        if (last < g.out1.row_lo) return;
    }
    const int n_tiles = g.K_total / 64 / ks, kt0 = sp * n_tiles;
    f32x4_t acc[4][MT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // ---- staging pointers.  W pieces of wave w: p = w + NW*j (same row parity for all, so one
    // per-lane base serves them); piece p covers LDS rows 8p..8p+7 of the W region.
    P64Ptrs<Cfg::XP> P;
    int64_t woff[Cfg::WP];
    {
        const int lr = lane >> 3, pos = lane & 7;
        const int lanerow = (lr >> 2) * 8 + (lr & 3);                 // lane part of nt_wperm for 8-row pieces
        const int r0 = wave * 8 + lr;                                 // LDS row of piece j = 0 (swizzle is the same for j > 0)
        P.w = reinterpret_cast<const char*>(g.W) + ((int64_t)(n0 + lanerow) * g.K_total + (int64_t)kt0 * 64) * 2 + ((pos ^ ((r0 >> 1) & 7)) << 4);
#pragma unroll
        for (int j = 0; j < Cfg::WP; ++j) {
            const int p = wave + Cfg::NW * j;                         // slab = p >> 3, piece in slab = p & 7
            woff[j] = (int64_t)((p >> 3) * 64 + p64_wpiece_row<EPI>(p & 7)) * g.K_total * 2;
        }
    }
    static_assert(Cfg::NW % 2 == 0, "W pieces of a wave must share the row parity");
    int seg = 0, left = g.seg[0].k_len / 64, issued = 0;
    {   // the segment and the K tile inside it where this workgroup's range starts (kt0 = 0 without split-K)
        int skip = kt0;
        while (skip >= left) { skip -= left; ++seg; left = g.seg[seg].k_len / 64; }
        p64_setup_x<Cfg::XP>(g, seg, b, m0, wave, lane, P);
#pragma unroll
        for (int j = 0; j < Cfg::XP; ++j) P.x[j] += skip * 128;     // (masked rows stream the zero region: they advance too)
        left -= skip;
    }
    bool idle = false;
    auto issue_piece = [&](char* stage, int j) {                     // j < XP: X piece, else W piece
        if (j < Cfg::XP) {
            glds16(P.x[j], stage + (wave * Cfg::XP + j) * 1024);
            P.x[j] += 128;
        } else {
            const int q = j - Cfg::XP;
            glds16(idle ? P.w : P.w + woff[q], stage + Cfg::BM * 128 + (wave + Cfg::NW * q) * 1024);
        }
    };
    auto advance = [&]() {                                           // after a whole tile has been issued
        P.w += 128;
        --left;
        ++issued;
        if (left == 0 || issued == n_tiles) {                        // wave-uniform and rare (a split-K range may end inside a segment)
            if (issued >= n_tiles) {
                idle = true;
#pragma unroll
                for (int j = 0; j < Cfg::XP; ++j) P.x[j] = reinterpret_cast<const char*>(aew_zero_region);
                P.w = reinterpret_cast<const char*>(aew_zero_region);
                left = 1 << 30;
            } else {
                ++seg;
                left = g.seg[seg].k_len / 64;
                p64_setup_x<Cfg::XP>(g, seg, b, m0, wave, lane, P);
            }
        }
    };
    // prologue: tiles 0 .. S-1 in flight
#pragma unroll
    for (int q = 0; q < S; ++q) {
#pragma unroll
        for (int j = 0; j < Cfg::PIECES; ++j) issue_piece(smem + q * Cfg::STAGE_BYTES, j);
        advance();
    }
    const int fi = lane & 15, fg = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)AEW_LDS_PTR(smem);
    const int rw = wn * 64 + fi, rx = wm * (16 * MT) + fi;   // tile i / j adds 16 rows = 2048 bytes, same swizzle
    // sub-step 0 reads chunk fg, sub-step 1 chunk fg + 4: (fg + 4s) ^ z = (fg ^ z) ^ 4s  ->  address ^ 64
    const uint32_t wlane = lds0 + Cfg::BM * 128 + rw * 128 + ((fg ^ ((rw >> 1) & 7)) << 4);
    const uint32_t xlane = lds0 + rx * 128 + ((fg ^ ((rx >> 1) & 7)) << 4);
    bf16x8_t wA[4], xA[MT], wB[4], xB[MT];
#define P64_READ(WF, XF, WA_, XA_)                     \
    do {                                               \
        if constexpr (MT == 8) P64_READ8(WF, XF, WA_, XA_); \
        else if constexpr (MT == 1) P64_READ1(WF, XF, WA_, XA_); \
        else P64_READ4(WF, XF, WA_, XA_);              \
    } while (0)

    p64_wait_vm<(S - 1) * Cfg::PIECES>();                      // tile 0 (mine) landed
    __builtin_amdgcn_s_barrier();
    P64_READ(wA, xA, wlane, xlane);
    uint32_t cur = 0;                                          // byte offset of the stage holding tile T
    for (int T = 0; T < n_tiles; ++T) {
        // ---- sub-step a: MFMA(T.a) from A under the reads of T.b
        p64_frags_ready<MT>(wA, xA);
        {
            const uint32_t wa = (wlane + cur) ^ 64u, xa = (xlane + cur) ^ 64u;
            P64_READ(wB, xB, wa, xa);
        }
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) AEW_MFMA_BF16(acc[i][j], wA[i], xA[j]);
        // ---- sub-step b: tile T+1 has landed (T+2 .. T+S-1 may be in flight), tile T is entirely in registers ->
        // its stage is free
        p64_wait_vm<(S - 2) * Cfg::PIECES>();
        p64_frags_ready<MT>(wB, xB);
        __builtin_amdgcn_s_barrier();
        char* freed = smem + cur;
        cur = (cur + Cfg::STAGE_BYTES == (uint32_t)(S * Cfg::STAGE_BYTES)) ? 0u : cur + (uint32_t)Cfg::STAGE_BYTES;
        P64_READ(wA, xA, wlane + cur, xlane + cur);           // (T+1).a
#pragma unroll
        for (int j = 0; j < MT; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) AEW_MFMA_BF16(acc[i][j], wB[i], xB[j]);
#pragma unroll
            for (int q = j * Cfg::PIECES / MT; q < (j + 1) * Cfg::PIECES / MT; ++q)   // tile T+2, threaded
                issue_piece(freed, q);                                                 // between the MFMA groups
        }
        advance();
    }
#undef P64_READ
    p64_frags_ready<MT>(wA, xA);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) asm volatile("" : "+v"(acc[i][j]));
    if (ks > 1) {
        // Split-K seam of a small launch (the upsampler / encoder dgrads: 16-112 workgroups of a lone block's fill latency
        // each, on a chip of 256 CUs): per WAVE, as in k_gemm_nt_f32.  The wave's accumulators go to slab sp in a private
        // per-lane layout (the same lane of the same wave of the same tile holds the same outputs in every copy) with
        // write-through stores; after they have drained one lane takes a device-scope ticket; the wave that draws the last
        // ticket of its sub-tile reads the ks partials back device-scope, adds them in ascending order of sp - a fixed
        // order: deterministic results - and runs the epilogue.  No waiting.
        const int64_t per_wave = 4 * MT * 64;                                   // f32x4 per wave
        char* wsb = reinterpret_cast<char*>(g.ksplit_ws);
        const int64_t mine = (((int64_t)sp * grid1 + L) * Cfg::NW + wave) * per_wave + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j)
                store16_wt(wsb + (mine + (int64_t)(i * MT + j) * 64) * 16,
                           (u32x4_t){__float_as_uint(acc[i][j][0]), __float_as_uint(acc[i][j][1]), __float_as_uint(acc[i][j][2]),
                                     __float_as_uint(acc[i][j][3])});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned* tk = g.ksplit_tickets + (int64_t)L * Cfg::NW + wave;
        unsigned t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
        if (t != (unsigned)(ks - 1)) return;
        if (lane == 0) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        const __amdgpu_buffer_rsrc_t rs = buf_rsrc(g.ksplit_ws);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                f32x4_t sum = {0.f, 0.f, 0.f, 0.f};
                for (int q = 0; q < ks; ++q) {
                    const int64_t at = ((((int64_t)q * grid1 + L) * Cfg::NW + wave) * per_wave + (int64_t)(i * MT + j) * 64 + lane) * 16;
                    const uint4 v = ld16_sc1(rs, (uint32_t)at);
                    const f32x4_t pq = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
                    sum = q == 0 ? pq : sum + pq;
                }
                acc[i][j] = sum;
            }
    }
    nt_epilogue<EPI, false, MT>(g, acc, b, m0, n0, wm, wn, lane);
}

// =============================================================================================
// NT kernel, fp32 (exact fmaf chain): block tile 16*RT (rows) x 64 (channels), BK = 32 floats, 4 waves each
// 16 channels x 16*RT rows = RT accumulator tiles.  One accumulator per output, K strictly ascending: the
// dependent v_mfma_f32_16x16x4 chain over K is the floor of this path (M = B * N_e is tiny).
// Rows are the batch folded into one axis (row R = b * M + m, resolved per row when the source pointers and the
// epilogue rows are set up), so a tile may straddle windows and the 29..70-row windows of the encoder do not each
// pad to a multiple of the tile.  What bounds the kernel is the operand stream: every row tile streams its 64-channel
// slab of W (K x 256 B) through LDS, and at ~30 KB in flight per block the chip sustains ~7 TB/s of it whatever the
// source (L2 or MALL: an XCD-local tile order changed nothing).  RT = 2 halves the number of times W is streamed and
// gives each wave two independent chains; it is used when that still leaves enough blocks (launch_gemm_nt).
// =============================================================================================
#define NF_BN 64
#define NF_BK 32
#ifndef AEW_FN_ABLATE
#define AEW_FN_ABLATE 0           /* 1: tools library; switches in aew_gemm_nt_t.reserved (tools/f32_ablate.py) */
#endif
#define NF_ABL(g, bit) (AEW_FN_ABLATE && ((g).reserved & (bit)))   /* 1 no MFMA, 2 no operand DMA, 4 no fragment reads, 8 no epilogue, 16 no barrier */
// Ring of S stages, TWO K tiles per barrier: the fragment reads of both go out together and there is one barrier / one
// vmcnt wait per 64 channels of K; S - 2 tiles are in flight.  That depth is what matters: a K tile is 256 cycles of
// MFMA per chain but ~3500 cycles of LDS-DMA latency on a loaded chip, and the rows are so few that a launch is 100-400
// blocks - the bytes in flight per CU bound the operand stream (5 stages: 30 KB per block, ~7 TB/s chip-wide, 53 us for
// the 1.9 GFLOP of encoder layer 1 whose chains alone are 15 us).  Launches of <= 256 blocks therefore take the whole
// LDS of their CU (12-14 stages).  Past the last tile the ring is topped up from the zero page so that the counted
// waits stay uniform.  The order of every chain is unchanged.
template <int RT, int S>
struct NfCfg {
    static constexpr int BM = 16 * RT, STAGE_BYTES = (BM + NF_BN) * 128, LDS_BYTES = S * STAGE_BYTES;
    static constexpr int XPW = RT == 4 ? 2 : 1;              // X pieces (8 rows each) a wave stages per K tile
    static constexpr int PW = XPW + 2;                       // LDS-DMA instructions per wave and K tile
    static_assert(S >= 4 && (S - 2) * PW <= 63 && LDS_BYTES <= 160 * 1024, "ring depth");
    static_assert(RT == 1 || RT == 2 || RT == 4, "16-, 32- or 64-row tiles");
};

struct NfPtrs {
    const char* x;
    const char* x2;                                          // RT = 4: the wave's second X piece (piece wave + 4)
    const char* w[2];
    int xinc, x2inc, winc;
};

template <int RT>
__device__ __forceinline__ void nf_setup_x(const aew_gemm_nt_t& g, int seg, int R0, int wave, int lane, NfPtrs& P) {
    // staged rows = 2 RT + 8 pieces of 8 rows: the X pieces, then W (two per wave).  EVERY wave stages an X piece
    // (RT = 1: waves 2, 3 write pieces 0, 1 a second time - same bytes, same place): all waves then run the same
    // instruction stream with the same vmcnt, and a taken branch costs this loop ~40 cycles (tools/f32_ablate.py)
    const int lr = lane >> 3, pc = lane & 7;
    const aew_seg_t s = g.seg[seg];
    {
        const int r = (wave & (2 * RT - 1)) * 8 + lr;
        const int R = R0 + r, bb = min(R / g.M, g.batch - 1);
        bool ok;
        const char* src = seg_row_ptr_sel(s, bb, R - bb * g.M, 4, ok) + (nt_swz(r, pc) << 4);
        ok = ok && R < g.M * g.batch;
        P.x = ok ? src : reinterpret_cast<const char*>(aew_zero_page);
        P.xinc = ok ? NF_BK * 4 : 0;
    }
    if constexpr (RT == 4) {                                  // 8 pieces, 4 staging waves: pieces wave and wave + 4
        const int r = (wave + 4) * 8 + lr;
        const int R = R0 + r, bb = min(R / g.M, g.batch - 1);
        bool ok;
        const char* src = seg_row_ptr_sel(s, bb, R - bb * g.M, 4, ok) + (nt_swz(r, pc) << 4);
        ok = ok && R < g.M * g.batch;
        P.x2 = ok ? src : reinterpret_cast<const char*>(aew_zero_page);
        P.x2inc = ok ? NF_BK * 4 : 0;
    }
}

template <int RT>
__device__ __forceinline__ void nf_issue(char* stage, int wave, NfPtrs& P) {
    glds16(P.x, stage + (wave & (2 * RT - 1)) * 1024);
    P.x += P.xinc;
    if constexpr (RT == 4) {
        glds16(P.x2, stage + (wave + 4) * 1024);
        P.x2 += P.x2inc;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        glds16(P.w[j], stage + 16 * RT * 128 + (wave + 4 * j) * 1024);
        P.w[j] += P.winc;
    }
}

// Fragments.  v_mfma_f32_16x16x4_f32 takes k = 4 s + kq from lane (fi, kq = lane >> 4) at step s.  Reading that with
// ds_read_b32 costs 16 LDS instructions per K tile and wave for 8 MFMAs, and b32 reads of 16 rows x 16 bytes run at
// ~4.5 cycles each (2-way bank conflict at 16-byte granularity; tools/ubench/nf_prims): the LDS pipe was as busy as the
// MFMA pipe.  Instead lane (fi, kq) reads the 16-byte chunk 4 j + kq of its row (k = 16 j + 4 kq + e in register e:
// 2 ds_read_b128 per operand and K tile) and a 4 x 4 transpose between register index and 16-lane group puts
// k = 16 j + 4 s + kq into register s: two v_permlane32_swap and two v_permlane16_swap per chunk, on the otherwise idle
// VALU.  The MFMAs see exactly the operands they saw before - same chain, same order.
__device__ __forceinline__ void nf_tr(f32x4_t& v) {
    // (elements go through named floats: __builtin_bit_cast applied directly to an ext-vector element lvalue reads
    // element 0 for every index with this compiler - ROCm 7.2 clang)
    const float f0 = v[0], f1 = v[1], f2 = v[2], f3 = v[3];
    unsigned a = __float_as_uint(f0), b = __float_as_uint(f1), c = __float_as_uint(f2), d = __float_as_uint(f3);
    auto r = __builtin_amdgcn_permlane32_swap(a, c, false, false);   // lanes 32-63 of a <-> lanes 0-31 of c
    a = r[0]; c = r[1];
    r = __builtin_amdgcn_permlane32_swap(b, d, false, false);
    b = r[0]; d = r[1];
    r = __builtin_amdgcn_permlane16_swap(a, b, false, false);        // odd 16-lane rows of a <-> even rows of b
    a = r[0]; b = r[1];
    r = __builtin_amdgcn_permlane16_swap(c, d, false, false);
    c = r[0]; d = r[1];
    v = (f32x4_t){__uint_as_float(a), __uint_as_float(b), __uint_as_float(c), __uint_as_float(d)};
}

template <int RT>
struct NfFrag {
    f32x4_t w[2], x[RT][2];                                  // chunk group j = 0, 1 of the K tile
};

#define NF_DS_READ16(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory")

template <int RT>
__device__ __forceinline__ void nf_read(NfFrag<RT>& F, uint32_t wl, uint32_t xl) {
    NF_DS_READ16(F.w[0], wl);
    NF_DS_READ16(F.x[0][0], xl);
    if constexpr (RT >= 2) NF_DS_READ16(F.x[1][0], xl + 2048u);
    if constexpr (RT == 4) { NF_DS_READ16(F.x[2][0], xl + 4096u); NF_DS_READ16(F.x[3][0], xl + 6144u); }
    NF_DS_READ16(F.w[1], wl ^ 64u);
    NF_DS_READ16(F.x[0][1], xl ^ 64u);
    if constexpr (RT >= 2) NF_DS_READ16(F.x[1][1], (xl + 2048u) ^ 64u);
    if constexpr (RT == 4) { NF_DS_READ16(F.x[2][1], (xl + 4096u) ^ 64u); NF_DS_READ16(F.x[3][1], (xl + 6144u) ^ 64u); }
}

template <int RT>
__device__ __forceinline__ void nf_ready(NfFrag<RT>& F) {
    if constexpr (RT == 4)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(F.w[0]), "+v"(F.w[1]), "+v"(F.x[0][0]), "+v"(F.x[0][1]), "+v"(F.x[1][0]), "+v"(F.x[1][1]),
                     "+v"(F.x[2][0]), "+v"(F.x[2][1]), "+v"(F.x[3][0]), "+v"(F.x[3][1]));
    else if constexpr (RT == 2)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(F.w[0]), "+v"(F.w[1]), "+v"(F.x[0][0]), "+v"(F.x[0][1]), "+v"(F.x[1][0]), "+v"(F.x[1][1]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(F.w[0]), "+v"(F.w[1]), "+v"(F.x[0][0]), "+v"(F.x[0][1]));
}

template <int RT>
__device__ __forceinline__ void nf_transpose(NfFrag<RT>& F) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        nf_tr(F.w[j]);
#pragma unroll
        for (int r = 0; r < RT; ++r) nf_tr(F.x[r][j]);
    }
}

// The 8 (x RT) MFMAs of a K tile as ONE asm block: an issue slot between two MFMAs on the same accumulator costs ~43
// cycles (MI355X_MICROARCH.md), and left to itself the compiler threads the transposes of the next fragments between
// them.  k = 16 j + 4 e + kq: ascending.
__device__ __forceinline__ void nf_mfma8(f32x4_t& acc, const f32x4_t& w0, const f32x4_t& x0, const f32x4_t& w1, const f32x4_t& x1) {
    const float a0 = w0[0], a1 = w0[1], a2 = w0[2], a3 = w0[3], a4 = w1[0], a5 = w1[1], a6 = w1[2], a7 = w1[3];
    const float b0 = x0[0], b1 = x0[1], b2 = x0[2], b3 = x0[3], b4 = x1[0], b5 = x1[1], b6 = x1[2], b7 = x1[3];
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %9, %0\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %2, %10, %0\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %3, %11, %0\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %4, %12, %0\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %5, %13, %0\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %6, %14, %0\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %7, %15, %0\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %8, %16, %0"
                 : "+a"(acc)
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7),
                   "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7));
}

template <int RT>
__device__ __forceinline__ void nf_chain(NfFrag<RT>& F, f32x4_t (&acc)[RT], bool skip_mfma) {
    if (skip_mfma) {
#pragma unroll
        for (int r = 0; r < RT; ++r) asm volatile("" ::"v"(F.w[0]), "v"(F.w[1]), "v"(F.x[r][0]), "v"(F.x[r][1]));
        return;
    }
    if constexpr (RT == 4) {
        // four INDEPENDENT chains (one per 16-row sub-tile), interleaved step by step: the three other chains' MFMAs fill
        // the ~33 cycles a dependent MFMA waits for its accumulator, so the compiler may schedule freely here.  Every
        // accumulator still sees its own products in ascending k: the same chain as the single-tile form, bit for bit.
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int r = 0; r < RT; ++r)
                    acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(F.w[j][e], F.x[r][j][e], acc[r], 0, 0, 0);
        return;
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) nf_mfma8(acc[r], F.w[0], F.x[r][0], F.w[1], F.x[r][1]);
}

// LW = 1: four more waves per block that do nothing but stage operands (wave 4 + w issues what consumer wave w would):
// an LDS-DMA instruction costs its wave 60-185 issue cycles, and three of them per K tile sat in front of every chain
// step of a wave that has 264 cycles of MFMA per tile.  Loaders and consumers meet at the one barrier per tile.
template <int RT, int S, int LW>
__global__ __launch_bounds__(256 * (1 + LW)) void k_gemm_nt_f32(const aew_gemm_nt_t g) {
    typedef NfCfg<RT, S> Cfg;
    extern __shared__ __attribute__((aligned(128))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wv & 3;
    const bool loader = LW && wv >= 4, stages_own = !LW;       // stages_own: this wave issues its own LDS-DMA
    const int rows = g.M * g.batch, n_rt = (rows + Cfg::BM - 1) / Cfg::BM;
    // split-K (aew_gemm_nt_t.k_split): workgroup = (k range sp, tile); range sp is K tiles [kt0, kt0 + nkt)
    const int ks = g.k_split > 1 ? g.k_split : 1;
    const int tiles = n_rt * (g.N_pad / NF_BN);
    const int sp = (int)blockIdx.x / tiles, tile = (int)blockIdx.x - sp * tiles;
    const int nt_i = tile / n_rt;
    const int R0 = (tile - nt_i * n_rt) * Cfg::BM, n0 = nt_i * NF_BN;
    const int nkt = g.K_total / NF_BK / ks, kt0 = sp * nkt;
    f32x4_t acc[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    NfPtrs P;
    {
        const int lr = lane >> 3, pc = lane & 7;
        const char* wbase = reinterpret_cast<const char*>(g.W);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = (wave + 4 * j) * 8 + lr;
            P.w[j] = wbase + ((int64_t)(n0 + r) * g.K_total + (int64_t)kt0 * NF_BK) * 4 + (nt_swz(r, pc) << 4);
        }
        P.winc = NF_BK * 4;
    }
    const bool no_mfma = NF_ABL(g, 1), no_dma = NF_ABL(g, 2), no_reads = NF_ABL(g, 4), no_barrier = NF_ABL(g, 16),
               no_tr = NF_ABL(g, 32), no_issue = NF_ABL(g, 64), no_wait = NF_ABL(g, 128);
    int seg = 0, left = g.seg[0].k_len / NF_BK, issued = 0;      // left = K tiles still to issue from `seg`
    uint32_t slot = 0;                                           // byte offset of the stage the next tile goes to
    {   // the segment and the K tile inside it where this workgroup's range starts (wave-uniform; kt0 = 0 without split-K)
        int skip = kt0;
        while (skip >= left) { skip -= left; ++seg; left = g.seg[seg].k_len / NF_BK; }
        nf_setup_x<RT>(g, seg, R0, wave, lane, P);
        P.x += (int64_t)skip * P.xinc;
        if constexpr (RT == 4) P.x2 += (int64_t)skip * P.x2inc;
        left -= skip;
    }
    auto issue_next = [&]() {
        if (__builtin_expect(left == 0, 0)) {                  // the only scalar loads of the K loop
            if (issued >= nkt) {                               // K exhausted: keep the ring (and vmcnt) uniform
                P.x = P.x2 = P.w[0] = P.w[1] = reinterpret_cast<const char*>(aew_zero_page);
                P.xinc = P.x2inc = P.winc = 0;
                left = 1 << 30;
            } else {
                ++seg;
                left = g.seg[seg].k_len / NF_BK;
                nf_setup_x<RT>(g, seg, R0, wave, lane, P);
            }
        }
        if (!no_dma) nf_issue<RT>(smem + slot, wave, P);
        slot = (slot + Cfg::STAGE_BYTES == (uint32_t)Cfg::LDS_BYTES) ? 0u : slot + Cfg::STAGE_BYTES;
        --left;
        ++issued;
    };
    if (loader || stages_own)
        for (int i = 0; i < S - 1; ++i) issue_next();            // tiles 0 .. S-2
    if (loader) {                                                // the consumers' schedule, minus everything but the DMA
        p64_wait_vm<(S - 2) * Cfg::PW>();
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nkt; ++t) {
            p64_wait_vm<(S - 3) * Cfg::PW>();
            __builtin_amdgcn_s_barrier();
            issue_next();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the top-up loads still target this block's LDS
        return;
    }
    const int fi = lane & 15, kq = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)AEW_LDS_PTR(smem);
    // lane (fi, kq) reads chunk kq (j = 0) / 4 + kq (j = 1: address ^ 64) of its rows; stages are multiples of 128 bytes
    const uint32_t xlane = lds0 + fi * 128 + (nt_swz(fi, kq) << 4);
    const uint32_t wlane = lds0 + Cfg::BM * 128 + (wave * 16 + fi) * 128 + (nt_swz(fi, kq) << 4);
    NfFrag<RT> A, B;
    if (stages_own) p64_wait_vm<(S - 2) * Cfg::PW>();
    __builtin_amdgcn_s_barrier();
    nf_read<RT>(A, wlane, xlane);
    nf_ready<RT>(A);
    nf_transpose<RT>(A);
    uint32_t cur = 0;                                            // stage of tile T
    // Per K tile T: [tile T+1 landed] barrier -> stage of T-1 is free: top the ring up into it -> start the reads of
    // T+1 -> chain of T (its fragments were read and transposed one step earlier) -> wait for T+1, transpose it.
    // (the wait closes the step: registers an asm read is still filling must not be live across the loop's back
    // edge, where the compiler is free to copy them)
    auto step = [&](NfFrag<RT>& F, NfFrag<RT>& Nx) {
        if (stages_own && !no_wait) p64_wait_vm<(S - 3) * Cfg::PW>();
        if (!no_barrier) __builtin_amdgcn_s_barrier();
        if (stages_own && !no_issue) issue_next();
        cur = (cur + Cfg::STAGE_BYTES == (uint32_t)Cfg::LDS_BYTES) ? 0u : cur + Cfg::STAGE_BYTES;
        if (!no_reads) nf_read<RT>(Nx, wlane + cur, xlane + cur);
        nf_chain<RT>(F, acc, no_mfma);
        if (!no_reads) nf_ready<RT>(Nx);
        if (!no_tr) nf_transpose<RT>(Nx);
    };
    for (int t = 0; t + 1 < nkt; t += 2) {
        step(A, B);
        step(B, A);
    }
    if (nkt & 1) step(A, B);
    if (stages_own) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the top-up loads still target this block's LDS
    if (ks > 1) {
        // Split-K seam, per WAVE (a wave owns 16 channels of the tile's rows: no workgroup barrier, the loader waves
        // have left).  The partial sums go to slab sp with write-through stores; once they have drained one lane takes a
        // ticket (device-scope atomic).  The wave that draws the LAST ticket of its sub-tile reads all S partials back
        // device-scope and adds them in the fixed order (p0 + p1) + (p2 + p3) - arrival order does not enter the result -
        // then runs the epilogue; the others are done.  Nobody waits for anybody.
        const int rows_pad = (rows + 31) & ~31;
        const int64_t slab = (int64_t)rows_pad * g.N_pad;
        const int ncol = n0 + wave * 16 + 4 * kq;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int R = R0 + 16 * r + fi;
            if (R < rows_pad)
                store16_wt(g.ksplit_ws + sp * slab + (int64_t)R * g.N_pad + ncol,
                           (u32x4_t){__float_as_uint(acc[r][0]), __float_as_uint(acc[r][1]), __float_as_uint(acc[r][2]),
                                     __float_as_uint(acc[r][3])});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned* tk = g.ksplit_tickets + (int64_t)tile * 4 + wave;
        unsigned t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
        if (t != (unsigned)(ks - 1)) return;
        if (lane == 0) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        const __amdgpu_buffer_rsrc_t rs = buf_rsrc(g.ksplit_ws);
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int R = R0 + 16 * r + fi;
            f32x4_t p[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 v = (q < ks && R < rows_pad) ? ld16_sc1(rs, (uint32_t)(((int64_t)q * slab + (int64_t)R * g.N_pad + ncol) * 4))
                                                         : make_uint4(0, 0, 0, 0);
                p[q] = (f32x4_t){__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc[r][e] = ks == 2 ? p[0][e] + p[1][e] : (p[0][e] + p[1][e]) + (p[2][e] + p[3][e]);
        }
    }
    unsigned zc = 0;
    const EpiUni U = epi_uni(g);
    if (NF_ABL(g, 8)) {
#pragma unroll
        for (int r = 0; r < RT; ++r) asm volatile("" ::"v"(acc[r]));
        return;
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int R = R0 + 16 * r + fi;
        const int n = n0 + wave * 16 + 4 * kq;
        float v[4] = {acc[r][0], acc[r][1], acc[r][2], acc[r][3]};
        if (R < rows && n < g.N) {
            const int bb = R / g.M, m = R - bb * g.M;
            const EpiRow Rw = epi_row(g, bb, m);
            epi_store<4>(g, U, Rw, bb, n, v, zc, g.flags);
        }
    }
    if (g.flags & AEW_EF_COUNT_ZERO) {
        zc = (unsigned)wave_sum((float)zc);
        if (lane == 0 && zc) atomicAdd(g.counter, (unsigned long long)zc);
    }
}

// =============================================================================================
// NT check kernel (impl = 1): one thread per (m, channel quad); same operands, same epilogues,
// plain fp32 fmaf chain in ascending k.  Used by the tests to isolate MFMA-path bugs.
// =============================================================================================
template <typename T>
__device__ __forceinline__ float ld_elem(const T* p);
template <> __device__ __forceinline__ float ld_elem<uint16_t>(const uint16_t* p) { return bf2f(*p); }
template <> __device__ __forceinline__ float ld_elem<float>(const float* p) { return *p; }

template <typename T>
__global__ void k_gemm_nt_check(const aew_gemm_nt_t g) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;        // quad index along N_pad
    const int m = blockIdx.y, b = blockIdx.z;
    const int nq_total = (g.epi == AEW_EPI_GATED) ? g.N_pad / 8 : g.N_pad / 4;
    if (q >= nq_total) return;
    const T* W = reinterpret_cast<const T*>(g.W);
    float a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
    int n_f, n_g = 0, ch = 0;
    if (g.epi == AEW_EPI_GATED) {
        // quad q of channels -> packed filt columns and gate columns
        ch = q * 4;
        n_f = (ch >> 4) * 32 + (ch & 15);
        n_g = n_f + 16;
    } else {
        n_f = q * 4;
    }
    int kglob = 0;
    for (int s = 0; s < g.n_segs; ++s) {
        const aew_seg_t& sg = g.seg[s];
        const int64_t row = (int64_t)m * sg.row_step + sg.row_off;
        const bool ok = row >= sg.row_lo && row < sg.row_hi;
        const T* xr = reinterpret_cast<const T*>(sg.ptr) + (int64_t)b * sg.batch_stride + row * sg.row_pitch;
        for (int k = 0; k < sg.k_len; ++k) {
            const float xv = ok ? ld_elem<T>(xr + k) : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a0[r] = fmaf(ld_elem<T>(W + (int64_t)(n_f + r) * g.K_total + kglob + k), xv, a0[r]);
                if (g.epi == AEW_EPI_GATED)
                    a1[r] = fmaf(ld_elem<T>(W + (int64_t)(n_g + r) * g.K_total + kglob + k), xv, a1[r]);
            }
        }
        kglob += sg.k_len;
    }
    unsigned zc = 0;
    const EpiRow R = epi_row(g, b, m);
    if (g.epi == AEW_EPI_GATED) {
        if (ch < g.N) {
            const float* bp = g.bias + (int64_t)b * g.bias_bs + n_f;
            const float fb[4] = {bp[0], bp[1], bp[2], bp[3]}, gb[4] = {bp[16], bp[17], bp[18], bp[19]};
            epi_gated<4>(g, R, ch, a0, a1, fb, gb);
        }
    } else if (n_f < g.N) {
        if (g.epi == AEW_EPI_STORE) epi_store<4>(g, epi_uni(g), R, b, n_f, a0, zc, g.flags);
        else if (g.epi == AEW_EPI_RES_SKIP) epi_res_skip<4>(epi_uni(g), R, n_f, a0);
        else epi_dfg<4>(epi_uni(g), R, n_f, a0);
    }
    if ((g.flags & AEW_EF_COUNT_ZERO) && zc) atomicAdd(g.counter, (unsigned long long)zc);
}

// =============================================================================================
// TN kernel, bf16: out tile 128 (k cols of A-seg) x 128 (n cols of G); contraction staged 64
// rows at a time; fragments via ds_read_b64_tr_b16 (hardware 4x4 transpose).
// acc[ki][ni][r] = dW[n = n0+wn*64+ni*16+q][k = k0+wk*64+ki*16+4g+r]
// =============================================================================================
#define TN_BT 128
#define TN_RC 32                                    // contraction rows per stage (one MFMA K step)
#define TN_STAGE_BYTES (2 * TN_RC * 256)            // 16 KiB

struct TnTile { int seg, kin, koff; };              // which segment / column offset this block owns

__device__ __forceinline__ TnTile tn_locate(const aew_gemm_tn_t& g, int kt, int tile) {
    // kt-th tile of `tile` columns along the concatenated K axis
    TnTile t; t.seg = 0; t.koff = kt * tile; t.kin = t.koff;
    while (t.kin >= g.seg[t.seg].k_len) { t.kin -= g.seg[t.seg].k_len; ++t.seg; }
    return t;
}

// Per-lane staging state of the TN kernels: 4 (bf16) / 2 (f32) pieces of 4 rows per operand and
// wave.  Pointers advance by one stage (RC contraction rows) per step; validity is re-evaluated
// per stage from the running row indices (cheap integer compares).
template <int NP>
struct TnPtrs {
    const char* g[NP];
    const char* a[NP];
    int grow[NP], arow[NP], m[NP];
    int64_t ginc, ainc;
    int gstep, astep;
    int glo, ghi, alo, ahi;      // row ranges, copied out of the descriptor once (no scalar loads per stage)
    bool fast;                   // wave-uniform: every row this wave will stage (all stages) is valid
};

// swizzle of the 32 x 32 x 16 fragment pattern (tn32_tile): a 16-lane group of ds_read_b64_tr_b16 covers 4 rows x 16 columns
// and lanes 0-31 are served together - 4 rows x 4 chunks: row r & 3 selects one of four 64-byte quarters of the 256-byte row
__device__ __forceinline__ int tn_swz32(int row, int chunk) { return chunk ^ ((row & 3) << 2); }

template <int NP, int ESIZE, int RC, int SWZ = 0>
__device__ __forceinline__ void tn_setup(const aew_gemm_tn_t& g, const TnTile& tt, int b, int r_lo, int n0,
                                         int wave, int lane, TnPtrs<NP>& P, int nst = 0, int r_end = 0) {
    const aew_seg_t sa = g.seg[tt.seg], sg = g.g;             // one batch of scalar loads
    const int lr = lane >> 4, pc = lane & 15;
    constexpr int EPC = 16 / ESIZE;
    P.gstep = RC * sg.row_step; P.astep = RC * sa.row_step;
    P.ginc = (int64_t)P.gstep * sg.row_pitch * ESIZE;
    P.ainc = (int64_t)P.astep * sa.row_pitch * ESIZE;
    P.glo = (int)sg.row_lo; P.ghi = (int)sg.row_hi; P.alo = (int)sa.row_lo; P.ahi = (int)sa.row_hi;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int r = (wave * NP + j) * 4 + lr;
        const int c = ESIZE == 2 ? (SWZ ? tn_swz32(r, pc) : tn_swz_bf16(r, pc)) : tn_swz_f32(r, pc);
        const int m = r_lo + r;
        P.m[j] = m;
        P.grow[j] = m * sg.row_step + sg.row_off;
        P.arow[j] = m * sa.row_step + sa.row_off;
        P.g[j] = reinterpret_cast<const char*>(sg.ptr) +
                 ((int64_t)b * sg.batch_stride + (int64_t)P.grow[j] * sg.row_pitch + n0 + c * EPC) * ESIZE;
        P.a[j] = reinterpret_cast<const char*>(sa.ptr) +
                 ((int64_t)b * sa.batch_stride + (int64_t)P.arow[j] * sa.row_pitch + tt.kin + c * EPC) * ESIZE;
    }
    // interior blocks (the common case) skip the per-row range checks of every stage
    bool ok = nst > 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int last = nst - 1;
        const int g0 = P.grow[j], g1 = g0 + last * P.gstep, a0 = P.arow[j], a1 = a0 + last * P.astep;
        ok = ok && (P.m[j] + last * RC < r_end) && min(g0, g1) >= P.glo && max(g0, g1) < P.ghi &&
             min(a0, a1) >= P.alo && max(a0, a1) < P.ahi;
    }
    P.fast = __all(ok);
}

template <int NP, int RC, bool RAW = false>
__device__ __forceinline__ void tn_issue(const aew_gemm_tn_t& g, const TnTile& tt, char* stage, int r_end,
                                         int wave, TnPtrs<NP>& P) {
    if (RAW && P.fast) {
        const uint32_t l0 = (uint32_t)(uintptr_t)AEW_LDS_PTR(stage) + wave * NP * 1024;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            glds16_raw(P.g[j], l0 + j * 1024);
            glds16_raw(P.a[j], l0 + j * 1024 + RC * 256);
            P.g[j] += P.ginc; P.a[j] += P.ainc;
        }
        return;
    }
    const char* zp = reinterpret_cast<const char*>(aew_zero_page);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const bool in = P.m[j] < r_end;
        const bool gok = in && P.grow[j] >= P.glo && P.grow[j] < P.ghi;
        const bool aok = in && P.arow[j] >= P.alo && P.arow[j] < P.ahi;
        if (RAW) {
            const uint32_t l0 = (uint32_t)(uintptr_t)AEW_LDS_PTR(stage) + (wave * NP + j) * 1024;
            glds16_raw(gok ? P.g[j] : zp, l0);
            glds16_raw(aok ? P.a[j] : zp, l0 + RC * 256);
        } else {
            glds16(gok ? P.g[j] : zp, stage + (wave * NP + j) * 1024);
            glds16(aok ? P.a[j] : zp, stage + RC * 256 + (wave * NP + j) * 1024);
        }
        P.g[j] += P.ginc; P.a[j] += P.ainc;
        P.grow[j] += P.gstep; P.arow[j] += P.astep; P.m[j] += RC;
    }
}

template <int SAFE>
__device__ __forceinline__ bf16x8_t tn_frag_bf16(const char* tile, int r0, int c0, int lane) {
    // operand fragment for columns c0..c0+15, contraction rows r0..r0+31 of a [rows][128] bf16 tile
    const int q = lane & 15, gq = lane >> 4;
    s16x8_t out;
    if (!SAFE) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = r0 + 8 * gq + 4 * h + (q >> 2);
            const int col = c0 + 4 * (q & 3);
            const int chunk = tn_swz_bf16(row, col >> 3);
            const char* p = tile + row * 256 + (chunk << 4) + ((col & 7) << 1);
            const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4_t __attribute__((address_space(3)))*)AEW_LDS_PTR(p));
            out[4 * h + 0] = v[0]; out[4 * h + 1] = v[1]; out[4 * h + 2] = v[2]; out[4 * h + 3] = v[3];
        }
    } else {
        // scalar gather with the same result layout (debug aid if the transpose read misbehaves)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int row = r0 + 8 * gq + e;
            const int col = c0 + q;
            const int chunk = tn_swz_bf16(row, col >> 3);
            out[e] = *reinterpret_cast<const short*>(tile + row * 256 + (chunk << 4) + ((col & 7) << 1));
        }
    }
    return __builtin_bit_cast(bf16x8_t, out);
}

#define TN_STAGES 3
#define TN_LDS_RING_BYTES (TN_STAGES * TN_STAGE_BYTES)
#define TN_LDS_BYTES (TN_LDS_RING_BYTES + 256)       // 48 KiB ring (three blocks per CU) + the row cursor's mailbox
#define TN_THREADS 256

// 4 waves as 2 (k) x 2 (n): each wave owns 64 (k cols of the A segment) x 64 (n cols of G) of the
// 128 x 128 output tile (4x4 MFMA tiles -> one transpose-read per MFMA).  3-stage LDS ring with
// counted vmcnt like the NT kernel.
//
// tn_bf16_tile: one output tile (kt, nt) contracted over rows [r_lo, r_hi) of the batch elements [b_lo, b_hi), in
// that order, result to `out` ([N_pad][K_total] fp32).  SNAP: after each batch element the running sums of column
// g.snap_k go to g.snap_out (see aewavenet.h; the caller passes SNAP only when the block contracts over everything).
// Row cursor of a grouped launch (aew_gemm_tn_group_t.cursors).  The tiles of ONE matrix share their operands - the k tiles
// of an n column re-read the same 128 columns of G, the n tiles of a k column the same 128 columns of A - and sit on one
// XCD, whose L2 holds ~300 rows of the matrices marching through it: a tile that falls further behind its siblings than
// that re-fetches everything from the fabric (PMC, round 3: 7.8 GB against 3.3-4.7 GB least).  The cursor bounds the
// drift: the contraction is cut into epochs of `e` stages (32 rows each); tile i keeps prog[i] = the epoch it is about to
// issue (a plain store per boundary: a shared counter was tried first - 28 tiles adding to one word at the same moment
// serialise in the L2's atomic unit, 1.8 us per boundary), and starts issuing epoch x only when every tile has reached
// epoch x - d + 1.  The row is polled ONE step ahead by an LDS-DMA load into a mailbox (free: measured), read back at the
// boundary and min-reduced across the wave.  L2-scope loads (the tiles of a matrix are placed on one XCD; if they are not,
// or not all resident, the wait times out once and the tile runs free from there on - it keeps publishing).  Wave 0 does
// the protocol in front of the per-stage barrier, the other waves meet it there.  At most 64 tiles per matrix.
struct TnCursor { unsigned* prog; int e, d, n, me; };

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (unsigned)__shfl_xor((int)v, o, 64));
    return v;
}

template <int SAFE, bool SNAP, bool CUR = false>      // CUR: the row-cursor protocol is compiled in (its own kernel: the default path carries none of it)
__device__ __forceinline__ void tn_bf16_tile(const aew_gemm_tn_t& g, char* smem, int kt, int nt, int b_lo, int b_hi,
                                             int r_lo, int r_hi, float* out, const TnCursor C = {nullptr, 0, 0, 0, 0}) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave & 1, wn = wave >> 1;
    const int n0 = nt * TN_BT;
    const TnTile tt = tn_locate(g, kt, TN_BT);
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int nst = (r_hi - r_lo + TN_RC - 1) / TN_RC;
    const int total = nst * (b_hi - b_lo);
    TnPtrs<2> P;
    int st_in_b = 0, bcur = b_lo, issued = 0, slot = 0;
    auto issue_next = [&]() {
        if (issued == 0) {
            tn_setup<2, 2, TN_RC>(g, tt, bcur, r_lo, n0, wave, lane, P, nst, r_hi);
        } else if (++st_in_b == nst) {                 // wave-uniform: next batch element
            st_in_b = 0; ++bcur;
            tn_setup<2, 2, TN_RC>(g, tt, bcur, r_lo, n0, wave, lane, P, nst, r_hi);
        }
        tn_issue<2, TN_RC, true>(g, tt, smem + slot * TN_STAGE_BYTES, r_hi, wave, P);
        slot = (slot + 1 == TN_STAGES) ? 0 : slot + 1;
        ++issued;
    };
    if (total > 0) issue_next();
    if (total > 1) issue_next();
    int stage = 0;
    // SNAP: the lanes that hold column snap_k of this tile (if it lies in it): acc[si][j][sr] of lanes with
    // (lane >> 4) == sg in the waves with (wave & 1) == sw
    const int srel = SNAP ? g.snap_k - tt.koff : -1;
    const bool snap_here = SNAP && g.snap_out && srel >= 0 && srel < TN_BT;
    int c_in_b = 0, bdone = b_lo;
    // column sums of G (grouped form, first k tile, the two waves of k half 0): one more MFMA per n tile and stage with
    // an all-ones A operand - every row of the result is sum_m G[m][n]
    const bool snap_cs = SNAP && g.snap_out && g.snap_k < 0;   // snap_k = -1: the running column sums of G themselves
    const bool do_cs = SNAP && (g.colsum_out != nullptr || snap_cs) && kt == 0 && wk == 0;
    f32x4_t cs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cs[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, (s16x8_t){0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80});
    bool cur_wait = CUR && C.prog != nullptr;          // false once a wait has timed out
    int cur_next = (CUR && C.prog && C.e >= 2) ? C.e : 0x7fffffff;  // first stage of the next epoch to be issued (stages 0, 1 are out)
    // The poll of a boundary is an LDS-DMA load of the progress row into a mailbox behind the ring (no VGPR is written
    // asynchronously), issued ONE step ahead so that it is fresh, and counted: wave 0's queue then holds one (the poll) or
    // two (+ the progress store) more operations than the 4 loads per stage the other waves leave in flight.
    const uint32_t cur_box = (uint32_t)(uintptr_t)AEW_LDS_PTR(smem) + TN_LDS_RING_BYTES;
    const unsigned* cur_mine = C.prog ? C.prog + (lane < C.n ? lane : 0) : nullptr;     // the word this lane polls
    bool cur_polled = false;
    int cur_e1 = 0;                                    // wave 0's extra operations issued during the previous step
    for (int t = 0; t < total; ++t) {
        const bool w0 = CUR && C.prog != nullptr && wave == 0;
        const bool boundary = w0 && t + 2 == cur_next && t + 2 < total;       // this step issues the first stage of an epoch
        int e0 = 0;
        if (w0 && cur_wait && t + 3 == cur_next && t + 3 < total && cur_next / C.e >= C.d) {   // the next step is a boundary
            glds4_raw(cur_mine, cur_box);
            cur_polled = true;
            e0 = 1;
        }
        // 4 loads per stage per wave stay in flight (+ wave 0's extras; a boundary step lets its poll land)
        const int allowed = CUR ? 4 + e0 + (boundary ? 0 : cur_e1) : 4;
        if (t + 1 >= total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (allowed == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (allowed == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        cur_e1 = e0;
        if (boundary) {
            const int ep = cur_next / C.e;
            if (lane == 0) asm volatile("global_store_dword %0, %1, off" :: "v"(C.prog + C.me), "v"((unsigned)ep) : "memory");
            ++cur_e1;
            if (cur_wait && ep >= C.d) {
                const unsigned need = (unsigned)(ep - C.d + 1);
                // (ds_read by hand: through a generic pointer the compiler emits a FLAT load and waits vmcnt(0) for it -
                // the whole operand ring drained at every boundary, 2 us each)
                unsigned have = 0;
                if (cur_polled) asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(have) : "v"(cur_box + 4 * lane) : "memory");
                have = wave_min_u32(have);
                if (have < need) {
                    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                    while (wave_min_u32(__hip_atomic_load(cur_mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
                        __builtin_amdgcn_s_sleep(1);
                        if (__builtin_amdgcn_s_memtime() - t0 > 3000ull) { cur_wait = false; break; }   // 30 us of the 100 MHz clock
                    }
                }
            }
            cur_polled = false;
        }
        if (CUR && t + 2 == cur_next) cur_next += C.e;
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 2 < total) issue_next();
        const char* gs = smem + stage * TN_STAGE_BYTES;
        const char* as = gs + TN_RC * 256;
        stage = (stage + 1 == TN_STAGES) ? 0 : stage + 1;
        bf16x8_t af[4], gf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = tn_frag_bf16<SAFE>(as, 0, wk * 64 + i * 16, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) gf[j] = tn_frag_bf16<SAFE>(gs, 0, wn * 64 + j * 16, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], gf[j], acc[i][j], 0, 0, 0);
        if (do_cs) {                                   // wave-uniform
#pragma unroll
            for (int j = 0; j < 4; ++j) cs[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, gf[j], cs[j], 0, 0, 0);
        }
        if (SNAP && ++c_in_b == nst) {                 // wave-uniform, once per batch element
            c_in_b = 0;
            if (snap_here && wk == (srel >> 6) && (lane >> 4) == ((srel >> 2) & 3)) {
                const int si = (srel >> 4) & 3, sr = srel & 3;
                float* so = g.snap_out + (int64_t)bdone * g.snap_bs + n0 + wn * 64 + (lane & 15);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v = (i == si && r == sr) ? acc[i][j][r] : v;
                    so[j * 16] = v;
                }
            }
            if (snap_cs && do_cs && (lane >> 4) == 0) {          // every row of cs[j] is sum_m G[m][n] so far
                float* so = g.snap_out + (int64_t)bdone * g.snap_bs + n0 + wn * 64 + (lane & 15);
#pragma unroll
                for (int j = 0; j < 4; ++j) so[j * 16] = cs[j][0];
            }
            ++bdone;
        }
    }
    const int q = lane & 15, gq = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + q;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = tt.koff + wk * 64 + i * 16 + 4 * gq;
            *reinterpret_cast<float4*>(out + (int64_t)n * g.K_total + k) =
                make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
        if (do_cs && g.colsum_out && gq == 0 && n < g.N) g.colsum_out[n] = cs[j][0];
    }
}

template <int SAFE>
__global__ __launch_bounds__(TN_THREADS, 2) void k_gemm_tn_bf16(const aew_gemm_tn_t g, int splits,
                                                                int rows_per_split, int fold_batch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nkt = g.K_total / TN_BT;
    // XCD-aware order: the output tiles that contract over the same (batch, row chunk) are
    // consecutive on ONE XCD, so the G / A rows of that chunk are fetched into its L2 once.
    const int n_tiles = nkt * (g.N_pad / TN_BT);
    const int n_chunks = splits * (fold_batch ? 1 : g.batch);
    const int L = blockIdx.x, seq = L >> 3;
    const int chunk = (seq / n_tiles) * 8 + (L & 7);
    if (chunk >= n_chunks) return;
    const int tile = seq % n_tiles;
    const int sp = chunk % splits, bz = chunk / splits;
    const int r_lo = sp * rows_per_split, r_hi = min(g.Mc, r_lo + rows_per_split);
    const int slab = fold_batch ? sp : (bz * splits + sp);
    tn_bf16_tile<SAFE, false>(g, smem, tile % nkt, tile / nkt, fold_batch ? 0 : bz, fold_batch ? g.batch : bz + 1,
                              r_lo, r_hi, g.out + (int64_t)slab * g.out_batch_stride);
}

// ---------------------------------------------------------------------------------------------------------------
// The same 128 x 128 tile on v_mfma_f32_32x32x16_bf16 (round 6, aew_tuning_t.tn_mfma32): HALF the MFMA instructions per
// stage (8 instead of 16 per wave: 2 x 2 tiles of 32 x 32, two 16-row k steps) for the same fragment bytes - the K loops
// are issue-bound (DESIGN 5 "Round 6") - and the instruction the bf16 peak is quoted on.  A wave still owns 64 (k) x 64
// (n); fragments: lane l holds 8 consecutive contraction rows 16 s + 8 (l / 32) .. + 7 of column l % 32 - two
// ds_read_b64_tr_b16, whose 16-lane groups each cover 4 rows x 16 columns (rows Rb + (q >> 2), columns Cb + 4 (q & 3) in,
// rows Rb .. Rb + 3 of column Cb + q out); accumulator register r of lane l = dW[n = .. + l % 32][k = .. + 8 (r / 4) +
// 4 (l / 32) + r % 4].  Same products, summed in 16-row instead of 32-row groups: equal to fp32 rounding, not bit for bit,
// to the 16 x 16 x 32 form; one fixed order.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8_t tn_frag32(const char* tile, int s, int c0, int lane) {
    const int q = lane & 15, ch = (lane >> 4) & 1, kh = lane >> 5;
    s16x8_t out;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = 16 * s + 8 * kh + 4 * h + (q >> 2);
        const int col = c0 + 16 * ch + 4 * (q & 3);
        const int chunk = tn_swz32(row, col >> 3);
        const char* p = tile + row * 256 + (chunk << 4) + ((col & 7) << 1);
        const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)AEW_LDS_PTR(p));
        out[4 * h + 0] = v[0]; out[4 * h + 1] = v[1]; out[4 * h + 2] = v[2]; out[4 * h + 3] = v[3];
    }
    return __builtin_bit_cast(bf16x8_t, out);
}

template <bool SNAP>
__device__ __forceinline__ void tn32_tile(const aew_gemm_tn_t& g, char* smem, int kt, int nt, int b_lo, int b_hi,
                                          int r_lo, int r_hi, float* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave & 1, wn = wave >> 1;
    const int n0 = nt * TN_BT;
    const TnTile tt = tn_locate(g, kt, TN_BT);
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nst = (r_hi - r_lo + TN_RC - 1) / TN_RC;
    const int total = nst * (b_hi - b_lo);
    TnPtrs<2> P;
    int st_in_b = 0, bcur = b_lo, issued = 0, slot = 0;
    auto issue_next = [&]() {
        if (issued == 0) {
            tn_setup<2, 2, TN_RC, 1>(g, tt, bcur, r_lo, n0, wave, lane, P, nst, r_hi);
        } else if (++st_in_b == nst) {
            st_in_b = 0; ++bcur;
            tn_setup<2, 2, TN_RC, 1>(g, tt, bcur, r_lo, n0, wave, lane, P, nst, r_hi);
        }
        tn_issue<2, TN_RC, true>(g, tt, smem + slot * TN_STAGE_BYTES, r_hi, wave, P);
        slot = (slot + 1 == TN_STAGES) ? 0 : slot + 1;
        ++issued;
    };
    if (total > 0) issue_next();
    if (total > 1) issue_next();
    int stage = 0;
    const int srel = SNAP ? g.snap_k - tt.koff : -1;
    const bool snap_here = SNAP && g.snap_out && srel >= 0 && srel < TN_BT;
    int c_in_b = 0, bdone = b_lo;
    const bool snap_cs = SNAP && g.snap_out && g.snap_k < 0;
    const bool do_cs = SNAP && (g.colsum_out != nullptr || snap_cs) && kt == 0 && wk == 0;
    f32x16_t cs[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) cs[j][r] = 0.f;
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, (s16x8_t){0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80});
    for (int t = 0; t < total; ++t) {
        if (t + 1 >= total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 2 < total) issue_next();
        const char* gs = smem + stage * TN_STAGE_BYTES;
        const char* as = gs + TN_RC * 256;
        stage = (stage + 1 == TN_STAGES) ? 0 : stage + 1;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8_t af[2], gf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = tn_frag32(as, s2, wk * 64 + i * 32, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) gf[j] = tn_frag32(gs, s2, wn * 64 + j * 32, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], gf[j], acc[i][j], 0, 0, 0);
            if (do_cs) {                                   // wave-uniform
#pragma unroll
                for (int j = 0; j < 2; ++j) cs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, gf[j], cs[j], 0, 0, 0);
            }
        }
        if (SNAP && ++c_in_b == nst) {                 // wave-uniform, once per batch element
            c_in_b = 0;
            // column srel of the tile: wave half srel >> 6, 32-tile (srel >> 5) & 1, register 4 ((srel >> 3) & 3) + (srel & 3)
            // of the lanes with lane / 32 == (srel >> 2) & 1
            if (snap_here && wk == (srel >> 6) && (lane >> 5) == ((srel >> 2) & 1)) {
                const int si = (srel >> 5) & 1, sr = 4 * ((srel >> 3) & 3) + (srel & 3);
                float* so = g.snap_out + (int64_t)bdone * g.snap_bs + n0 + wn * 64 + (lane & 31);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v = 0.f;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) v = (i == si && r == sr) ? acc[i][j][r] : v;
                    so[j * 32] = v;
                }
            }
            if (snap_cs && do_cs && (lane >> 5) == 0) {          // every row of cs[j] is sum_m G[m][n] so far
                float* so = g.snap_out + (int64_t)bdone * g.snap_bs + n0 + wn * 64 + (lane & 31);
#pragma unroll
                for (int j = 0; j < 2; ++j) so[j * 32] = cs[j][0];
            }
            ++bdone;
        }
    }
    const int q = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + q;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int k = tt.koff + wk * 64 + i * 32 + 8 * rq + 4 * kh;
                *reinterpret_cast<float4*>(out + (int64_t)n * g.K_total + k) =
                    make_float4(acc[i][j][4 * rq], acc[i][j][4 * rq + 1], acc[i][j][4 * rq + 2], acc[i][j][4 * rq + 3]);
            }
        if (do_cs && g.colsum_out && kh == 0 && n < g.N) g.colsum_out[n] = cs[j][0];
    }
}

__global__ __launch_bounds__(TN_THREADS, 2) void k_gemm_tn_bf16_grp32(const aew_gemm_tn_t* __restrict__ descs,
                                                                      const int32_t* __restrict__ tile_map) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int rec = __builtin_amdgcn_readfirstlane(tile_map[blockIdx.x]);
    if (rec < 0) return;
    const aew_gemm_tn_t& g = descs[rec >> 22];
    const int tile = rec & 0xfff, chunk = (rec >> 12) & 0x3ff, nkt = g.K_total / TN_BT;
    if (g.grp_splits > 0) {
        const int bz = chunk / g.grp_splits, sp = chunk - bz * g.grp_splits;
        const int r_lo = sp * g.grp_rows;
        tn32_tile<false>(g, smem, tile % nkt, tile / nkt, bz, bz + 1, r_lo, min(g.Mc, r_lo + g.grp_rows),
                         g.out + (int64_t)chunk * g.out_batch_stride);
        return;
    }
    tn32_tile<true>(g, smem, tile % nkt, tile / nkt, 0, g.batch, 0, g.Mc, g.out);
}

// Grouped form (aew_gemm_tn_group_t): block p takes tile tile_map[p] of descriptor table `descs` (device memory,
// read with scalar loads: the index is wave-uniform and the table is never written while a plan runs) and contracts
// it over every row of every batch element - one result, no slabs.
template <bool CUR>
__device__ __forceinline__ void tn_grp_block(const aew_gemm_tn_t* __restrict__ descs, const int32_t* __restrict__ tile_map,
                                             char* smem, unsigned* cursors, int cur_stride, int cur_e, int cur_d) {
    const int rec = __builtin_amdgcn_readfirstlane(tile_map[blockIdx.x]);
    if (rec < 0) return;
    const aew_gemm_tn_t& g = descs[rec >> 22];
    const int tile = rec & 0xfff, chunk = (rec >> 12) & 0x3ff, nkt = g.K_total / TN_BT;
    if (g.grp_splits > 0) {                            // split like a stand-alone op: (batch element, row chunk) -> slab
        const int bz = chunk / g.grp_splits, sp = chunk - bz * g.grp_splits;
        const int r_lo = sp * g.grp_rows;
        tn_bf16_tile<0, false>(g, smem, tile % nkt, tile / nkt, bz, bz + 1, r_lo, min(g.Mc, r_lo + g.grp_rows),
                               g.out + (int64_t)chunk * g.out_batch_stride);
        return;
    }
    const int n_tiles = nkt * (g.N_pad / TN_BT);
    TnCursor C = {nullptr, 0, 0, 0, 0};
    if (CUR && cursors && n_tiles > 1 && n_tiles <= 64 && cur_stride >= 64)
        C = TnCursor{cursors + (int64_t)(rec >> 22) * cur_stride, cur_e, cur_d, n_tiles, tile};
    tn_bf16_tile<0, true, CUR>(g, smem, tile % nkt, tile / nkt, 0, g.batch, 0, g.Mc, g.out, C);
}

__global__ __launch_bounds__(TN_THREADS, 2) void k_gemm_tn_bf16_grp(const aew_gemm_tn_t* __restrict__ descs,
                                                                    const int32_t* __restrict__ tile_map) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tn_grp_block<false>(descs, tile_map, smem, nullptr, 0, 0, 0);
}

// the same launch with the row cursor (aew_gemm_tn_group_t.cursors + aew_set_tn_cursor)
__global__ __launch_bounds__(TN_THREADS, 2) void k_gemm_tn_bf16_grp_cur(const aew_gemm_tn_t* __restrict__ descs,
                                                                        const int32_t* __restrict__ tile_map,
                                                                        unsigned* cursors, int cur_stride, int cur_e, int cur_d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tn_grp_block<true>(descs, tile_map, smem, cursors, cur_stride, cur_e, cur_d);
}

// =============================================================================================
// TN kernel, bf16, big tiles ("tnb"): out tile 256 (k) x 256 (n) = 2 x 2 HALVES of 128 columns, 8 waves as
// 2 (k half) x 4 (64 n columns), each wave 128 (k) x 64 (n) = 8 x 4 MFMA tiles.  The small-tile kernel above stages
// (128 + 128) columns per 64 MFMAs = 256 B per MFMA and is bound by the LDS-DMA path (measured 18-30 B/clk/CU,
// profiles/r02_notes.md); this one stages (256 + 256) columns per 256 MFMAs = 128 B per MFMA.  A k half is a
// 128-column tile of ONE A segment (tn_locate), so tiles never straddle segments although segments are only
// 128-aligned; a half that does not exist (K_total or N_pad not a multiple of 256) is skipped by the waves that own it.
// LDS: 4-stage ring of 32-row stages, stage = [G half 0 | G half 1 | A half 0 | A half 1] x 32 rows x 256 B = 32 KiB;
// wave w stages region w >> 1 (pieces (w & 1) * 4 .. + 3): one source matrix / segment per wave.
// Slab layout and split-K are those of the small-tile kernel (the unpack table is unchanged).
// =============================================================================================
#define TNB_STAGES 4
#define TNB_STAGE_BYTES (4 * TN_RC * 256)            // 32 KiB
#define TNB_LDS_BYTES (TNB_STAGES * TNB_STAGE_BYTES)  // 128 KiB: one block per CU
#define TNB_THREADS 512

struct TnbSrc {                                      // what one wave stages: 4 pieces (4 rows each) of one region
    const char* p[4];
    int row[4], m[4];
    int64_t inc;
    int step, lo, hi;
    bool valid, fast;
};

__device__ __forceinline__ void tnb_setup(const aew_seg_t s, int col, int b, int r_lo, int wave, int lane, TnbSrc& S,
                                          int nst, int r_end) {
    const int lr = lane >> 4, pc = lane & 15;
    S.step = TN_RC * s.row_step;
    S.inc = (int64_t)S.step * s.row_pitch * 2;
    S.lo = (int)s.row_lo; S.hi = (int)s.row_hi;
    bool ok = nst > 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = ((wave & 1) * 4 + j) * 4 + lr;
        const int c = tn_swz_bf16(r, pc);
        const int m = r_lo + r;
        S.m[j] = m;
        S.row[j] = m * s.row_step + s.row_off;
        S.p[j] = reinterpret_cast<const char*>(s.ptr) +
                 ((int64_t)b * s.batch_stride + (int64_t)S.row[j] * s.row_pitch + col + c * 8) * 2;
        const int last = nst - 1, r0 = S.row[j], r1 = r0 + last * S.step;
        ok = ok && (m + last * TN_RC < r_end) && min(r0, r1) >= S.lo && max(r0, r1) < S.hi;
    }
    S.fast = __all(ok);
}

__device__ __forceinline__ void tnb_issue(uint32_t lds_region, int r_end, int wave, TnbSrc& S) {
    const char* zp = reinterpret_cast<const char*>(aew_zero_page);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t dst = lds_region + ((wave & 1) * 4 + j) * 1024;
        if (S.fast) {
            glds16_raw(S.p[j], dst);
        } else {
            const bool ok = S.m[j] < r_end && S.row[j] >= S.lo && S.row[j] < S.hi;
            glds16_raw(ok ? S.p[j] : zp, dst);
            S.row[j] += S.step; S.m[j] += TN_RC;
        }
        S.p[j] += S.inc;
    }
}

// tnb_tile: one 256 x 256 output tile (kt, nt) contracted over rows [r_lo, r_hi) of the batch elements [b_lo, b_hi), in
// that order, result to `out` ([N_pad][K_total] fp32).  SNAP as in tn_bf16_tile.
template <bool SNAP>
__device__ __forceinline__ void tnb_tile(const aew_gemm_tn_t& g, char* smem, int kt, int nt, int b_lo, int b_hi,
                                         int r_lo, int r_hi, float* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave & 1, wn = wave >> 1;                  // k half, 64-column n slab (n half = wn >> 1)
    const int nk128 = g.K_total / 128, nn128 = g.N_pad / 128;
    const int nst = (r_hi - r_lo + TN_RC - 1) / TN_RC;
    const int total = nst * (b_hi - b_lo);
    // ---- what this wave stages: region = wave >> 1: 0,1 = G halves, 2,3 = A halves
    const int region = wave >> 1, rhalf = region & 1;
    const bool stage_g = region < 2;
    const int my128 = stage_g ? 2 * nt + rhalf : 2 * kt + rhalf;          // 128-column tile index of the staged half
    const bool stage_valid = stage_g ? my128 < nn128 : my128 < nk128;
    TnTile st_tt = {0, 0, 0};
    if (!stage_g && stage_valid) st_tt = tn_locate(g, my128, 128);
    // ---- what this wave computes
    const int k128 = 2 * kt + wk, n128 = 2 * nt + (wn >> 1);
    const bool comp_valid = k128 < nk128 && n128 < nn128;
    TnTile ctt = {0, 0, 0};
    if (k128 < nk128) ctt = tn_locate(g, k128, 128);
    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    TnbSrc S;
    const uint32_t lds0 = (uint32_t)(uintptr_t)AEW_LDS_PTR(smem);
    int st_in_b = 0, bcur = b_lo, issued = 0, slot = 0;
    auto setup = [&]() {
        if (stage_g) tnb_setup(g.g, my128 * 128, bcur, r_lo, wave, lane, S, nst, r_hi);
        else tnb_setup(g.seg[st_tt.seg], st_tt.kin, bcur, r_lo, wave, lane, S, nst, r_hi);
    };
    auto issue_next = [&]() {
        if (stage_valid) {
            if (issued == 0) setup();
            else if (st_in_b == 0) setup();
            tnb_issue(lds0 + slot * TNB_STAGE_BYTES + region * (TN_RC * 256), r_hi, wave, S);
        }
        if (++st_in_b == nst) { st_in_b = 0; ++bcur; }
        slot = (slot + 1 == TNB_STAGES) ? 0 : slot + 1;
        ++issued;
    };
    // waves of an invalid region issue nothing: their vmcnt is 0, the counted waits below pass at once
#pragma unroll
    for (int q = 0; q < TNB_STAGES - 1; ++q)
        if (q < total) issue_next();
    int stage = 0;
    // SNAP: column snap_k lies in this wave's 128-column k half at offset srel (or not at all)
    const int srel = (SNAP && comp_valid) ? g.snap_k - ctt.koff : -1;
    const bool snap_here = SNAP && g.snap_out && srel >= 0 && srel < 128;
    int c_in_b = 0, bdone = b_lo;
    // column sums of G (see tn_bf16_tile): the waves of k half 0 of the first k tile
    const bool snap_cs = SNAP && g.snap_out && g.snap_k < 0;   // snap_k = -1: the running column sums of G themselves
    const bool do_cs = SNAP && (g.colsum_out != nullptr || snap_cs) && kt == 0 && wk == 0 && comp_valid;
    f32x4_t cs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cs[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, (s16x8_t){0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80});
    for (int t = 0; t < total; ++t) {
        // stage t has landed once at most the stages issued after it are outstanding (4 pieces per wave and stage)
        const int ahead = min(total - 1 - t, TNB_STAGES - 2);
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + TNB_STAGES - 1 < total) issue_next();          // into the stage computed at step t - 1
        const char* sb = smem + stage * TNB_STAGE_BYTES;
        stage = (stage + 1 == TNB_STAGES) ? 0 : stage + 1;
        if (comp_valid) {
            const char* gs = sb + (wn >> 1) * (TN_RC * 256);
            const char* as = sb + (2 + wk) * (TN_RC * 256);
            bf16x8_t af[8], gf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) gf[j] = tn_frag_bf16<0>(gs, 0, (wn & 1) * 64 + j * 16, lane);
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = tn_frag_bf16<0>(as, 0, i * 16, lane);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], gf[j], acc[i][j], 0, 0, 0);
            if (do_cs) {
#pragma unroll
                for (int j = 0; j < 4; ++j) cs[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, gf[j], cs[j], 0, 0, 0);
            }
        }
        if (SNAP && ++c_in_b == nst) {                         // wave-uniform, once per batch element
            c_in_b = 0;
            if (snap_here && (lane >> 4) == ((srel >> 2) & 3)) {
                const int si = srel >> 4, sr = srel & 3;
                float* so = g.snap_out + (int64_t)bdone * g.snap_bs + n128 * 128 + (wn & 1) * 64 + (lane & 15);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v = (i == si && r == sr) ? acc[i][j][r] : v;
                    so[j * 16] = v;
                }
            }
            if (snap_cs && do_cs && (lane >> 4) == 0) {
                float* so = g.snap_out + (int64_t)bdone * g.snap_bs + n128 * 128 + (wn & 1) * 64 + (lane & 15);
#pragma unroll
                for (int j = 0; j < 4; ++j) so[j * 16] = cs[j][0];
            }
            ++bdone;
        }
    }
    if (!comp_valid) return;
    const int q = lane & 15, gq = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n128 * 128 + (wn & 1) * 64 + j * 16 + q;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = ctt.koff + i * 16 + 4 * gq;
            *reinterpret_cast<float4*>(out + (int64_t)n * g.K_total + k) =
                make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
        if (do_cs && g.colsum_out && gq == 0 && n < g.N) g.colsum_out[n] = cs[j][0];
    }
}

__global__ __launch_bounds__(TNB_THREADS, 2) void k_gemm_tn_bf16_big(const aew_gemm_tn_t g, int splits, int rows_per_split,
                                                                     int fold_batch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nkt = (g.K_total / 128 + 1) / 2, nnt = (g.N_pad / 128 + 1) / 2;
    const int n_tiles = nkt * nnt;
    const int n_chunks = splits * (fold_batch ? 1 : g.batch);
    // XCD-aware order (see k_gemm_tn_bf16): the tiles that contract over one (batch, row chunk) sit on one XCD
    const int L = blockIdx.x, seq = L >> 3;
    const int chunk = (seq / n_tiles) * 8 + (L & 7);
    if (chunk >= n_chunks) return;
    const int tile = seq % n_tiles;
    const int sp = chunk % splits, bz = chunk / splits;
    const int r_lo = sp * rows_per_split, r_hi = min(g.Mc, r_lo + rows_per_split);
    const int slab = fold_batch ? sp : (bz * splits + sp);
    tnb_tile<false>(g, smem, tile % nkt, tile / nkt, fold_batch ? 0 : bz, fold_batch ? g.batch : bz + 1, r_lo, r_hi,
                    g.out + (int64_t)slab * g.out_batch_stride);
}

// Grouped form on the big tiles (aew_gemm_tn_group_t.tile = 256): tile = nt * ((K_total / 128 + 1) / 2) + kt.
__global__ __launch_bounds__(TNB_THREADS, 2) void k_gemm_tn_bf16_big_grp(const aew_gemm_tn_t* __restrict__ descs,
                                                                         const int32_t* __restrict__ tile_map) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int rec = __builtin_amdgcn_readfirstlane(tile_map[blockIdx.x]);
    if (rec < 0) return;
    const aew_gemm_tn_t& g = descs[rec >> 22];
    const int tile = rec & 0xfff, nkt = (g.K_total / 128 + 1) / 2;
    tnb_tile<true>(g, smem, tile % nkt, tile / nkt, 0, g.batch, 0, g.Mc, g.out);
}

// =============================================================================================
// TN kernel, bf16, EIGHT waves on a rectangular tile ("tn8", aew_gemm_tn_group_t.tile = 384; round 6).
// The 128 x 128 kernel above stages (128 + 128) columns per 64 MFMAs and runs three 4-wave blocks per CU; the NT bodies -
// 256 x 128 tiles, 8 waves of 64 x 64, two blocks per CU on a 3 x 24 KiB ring: (256 + 128) columns per 128 MFMAs, four
// waves per SIMD - move 1.6x the MFMAs per CU and K step in their K loop.  This is that shape for the weight gradients:
//   orientation 0  128 (k) x 256 (n): LDS blocks [G n-half 0 | G n-half 1 | A k tile], waves as 2 (k) x 4 (n)
//   orientation 1  256 (k) x 128 (n): LDS blocks [G | A k tile 2 kt | A k tile 2 kt + 1], waves as 4 (k) x 2 (n)
// (tn8_ori: the one that tiles the matrix without a half-empty tile where one does; a half that lies beyond N_pad /
// K_total is staged from the zero page and not stored).  A block is [32 rows][128 columns] bf16 = 8 KiB in the 128-tile
// kernel's layout (same swizzle, same ds_read_b64_tr_b16 fragments); every wave stages one 4-row piece of each block per
// step.  Every output element accumulates the same 32-row products in the same order (rows ascending, batch elements
// ascending) as in tn_bf16_tile: results are BIT-IDENTICAL to the 128-tile grouped launch, by-products included.
// =============================================================================================
#define TN8_THREADS 512
#define TN8_BLK_BYTES (TN_RC * 256)                  // 8 KiB
#define TN8_STAGE_BYTES (3 * TN8_BLK_BYTES)          // 24 KiB
#define TN8_STAGES 3
#define TN8_LDS_BYTES (TN8_STAGES * TN8_STAGE_BYTES) // 72 KiB: two blocks per CU

__host__ __device__ __forceinline__ int tn8_ori(int N_pad, int K_total) { return (K_total % 256 == 0 && N_pad % 256 != 0) ? 1 : 0; }

struct Tn8Src {                                       // per-lane source of one LDS block's piece
    const char* p;
    int64_t inc;
    int row, step, lo, hi;
    bool live;                                        // (wave-uniform) the block exists
};

__device__ __forceinline__ void tn8_src(Tn8Src& S, const aew_seg_t s, int col, int b, int m, int chunk, bool live) {
    S.step = TN_RC * s.row_step;
    S.inc = (int64_t)S.step * s.row_pitch * 2;
    S.lo = (int)s.row_lo; S.hi = (int)s.row_hi;
    S.row = m * s.row_step + s.row_off;
    S.p = reinterpret_cast<const char*>(s.ptr) + ((int64_t)b * s.batch_stride + (int64_t)S.row * s.row_pitch + col + chunk * 8) * 2;
    S.live = live;
}

template <bool SNAP>
__device__ __forceinline__ void tn8_tile(const aew_gemm_tn_t& g, char* smem, int kt, int nt, float* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ori = tn8_ori(g.N_pad, g.K_total);
    const int wk = ori ? (wave & 3) : (wave & 1), wn = ori ? (wave >> 2) : (wave >> 1);
    const int n0 = nt * (ori ? 128 : 256), kfirst = ori ? 2 * kt : kt;
    const int gblk = ori ? 0 : (wn >> 1), gcol = ori ? wn * 64 : (wn & 1) * 64;
    const int ablk = ori ? 1 + (wk >> 1) : 2, acol = ori ? (wk & 1) * 64 : wk * 64;
    const bool half1 = ori ? (kfirst + 1) * 128 < g.K_total : n0 + 128 < g.N_pad;     // the tile's second 128-column half exists
    const TnTile t0 = tn_locate(g, kfirst, TN_BT);
    const TnTile t1 = (ori && half1) ? tn_locate(g, kfirst + 1, TN_BT) : t0;
    const bool mine = ori ? (wk < 2 || half1) : (wn < 2 || half1);                     // this wave's 64 x 64 lies inside the matrix
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int nst = (g.Mc + TN_RC - 1) / TN_RC;
    const int total = nst * g.batch;
    const int lr = lane >> 4, pc = lane & 15;
    const int prow = wave * 4 + lr;                                                    // this lane's row of every 32-row block
    const int chunk = tn_swz_bf16(prow, pc);
    Tn8Src S[3];
    bool fast = false;
    int st_in_b = 0, bcur = 0, issued = 0, slot = 0, m_next = 0;
    auto setup = [&](int b) {
        const aew_seg_t sg = g.g;
        if (ori == 0) {
            tn8_src(S[0], sg, n0, b, prow, chunk, true);
            tn8_src(S[1], sg, n0 + 128, b, prow, chunk, half1);
            tn8_src(S[2], g.seg[t0.seg], t0.kin, b, prow, chunk, true);
        } else {
            tn8_src(S[0], sg, n0, b, prow, chunk, true);
            tn8_src(S[1], g.seg[t0.seg], t0.kin, b, prow, chunk, true);
            tn8_src(S[2], g.seg[t1.seg], t1.kin, b, prow, chunk, half1);
        }
        m_next = prow;
        // interior (the common case): every row this lane stages for this batch element is valid
        bool ok = true;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int r0 = S[q].row, r1 = r0 + (nst - 1) * S[q].step;
            ok = ok && (!S[q].live || (min(r0, r1) >= S[q].lo && max(r0, r1) < S[q].hi));
        }
        ok = ok && prow + (nst - 1) * TN_RC < g.Mc;
        fast = __all(ok);
    };
    auto issue_next = [&]() {
        if (issued == 0) setup(0);
        else if (++st_in_b == nst) { st_in_b = 0; ++bcur; setup(bcur); }
        const uint32_t l0 = (uint32_t)(uintptr_t)AEW_LDS_PTR(smem + slot * TN8_STAGE_BYTES) + wave * 1024;
        const char* zp = reinterpret_cast<const char*>(aew_zero_page);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const bool ok = S[q].live && (fast || (m_next < g.Mc && S[q].row >= S[q].lo && S[q].row < S[q].hi));
            glds16_raw(ok ? S[q].p : zp, l0 + q * TN8_BLK_BYTES);
            S[q].p += S[q].inc; S[q].row += S[q].step;
        }
        m_next += TN_RC;
        slot = (slot + 1 == TN8_STAGES) ? 0 : slot + 1;
        ++issued;
    };
    if (total > 0) issue_next();
    if (total > 1) issue_next();
    int stage = 0;
    const int koff0 = kfirst * TN_BT;                                                  // first column of the tile on the K axis
    const int ktile_w = ori ? 256 : 128;
    const int srel = SNAP ? g.snap_k - koff0 : -1;
    const bool snap_here = SNAP && g.snap_out && srel >= 0 && srel < ktile_w && (ori == 0 || srel < 128 || half1);
    int c_in_b = 0, bdone = 0;
    const bool snap_cs = SNAP && g.snap_out && g.snap_k < 0;
    const bool do_cs = SNAP && (g.colsum_out != nullptr || snap_cs) && kfirst == 0 && wk == 0 && mine;
    f32x4_t cs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cs[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, (s16x8_t){0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80});
    for (int t = 0; t < total; ++t) {
        if (t + 1 >= total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                          // the 3 pieces of stage t + 1 stay in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 2 < total) issue_next();
        const char* sb = smem + stage * TN8_STAGE_BYTES;
        const char* gs = sb + gblk * TN8_BLK_BYTES;
        const char* as = sb + ablk * TN8_BLK_BYTES;
        stage = (stage + 1 == TN8_STAGES) ? 0 : stage + 1;
        bf16x8_t af[4], gf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = tn_frag_bf16<0>(as, 0, acol + i * 16, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) gf[j] = tn_frag_bf16<0>(gs, 0, gcol + j * 16, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], gf[j], acc[i][j], 0, 0, 0);
        if (do_cs) {                                   // wave-uniform
#pragma unroll
            for (int j = 0; j < 4; ++j) cs[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, gf[j], cs[j], 0, 0, 0);
        }
        if (SNAP && ++c_in_b == nst) {                 // wave-uniform, once per batch element
            c_in_b = 0;
            if (snap_here && mine && wk == (srel >> 6) && (lane >> 4) == ((srel >> 2) & 3)) {
                const int si = (srel >> 4) & 3, sr = srel & 3;
                float* so = g.snap_out + (int64_t)bdone * g.snap_bs + n0 + wn * 64 + (lane & 15);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v = (i == si && r == sr) ? acc[i][j][r] : v;
                    so[j * 16] = v;
                }
            }
            if (snap_cs && do_cs && (lane >> 4) == 0) {
                float* so = g.snap_out + (int64_t)bdone * g.snap_bs + n0 + wn * 64 + (lane & 15);
#pragma unroll
                for (int j = 0; j < 4; ++j) so[j * 16] = cs[j][0];
            }
            ++bdone;
        }
    }
    if (!mine) return;
    const int q = lane & 15, gq = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + q;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = koff0 + wk * 64 + i * 16 + 4 * gq;
            *reinterpret_cast<float4*>(out + (int64_t)n * g.K_total + k) =
                make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
        if (do_cs && g.colsum_out && gq == 0 && n < g.N) g.colsum_out[n] = cs[j][0];
    }
}

// Grouped launch on the 8-wave tiles (aew_gemm_tn_group_t.tile = 384): tile = nt * nkt + kt in the descriptor's own
// orientation (tn8_ori): nkt = K_total / 128 (orientation 0) or ceil(K_total / 256) (orientation 1).
__global__ __launch_bounds__(TN8_THREADS, 4) void k_gemm_tn_bf16_grp8(const aew_gemm_tn_t* __restrict__ descs,
                                                                      const int32_t* __restrict__ tile_map) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int rec = __builtin_amdgcn_readfirstlane(tile_map[blockIdx.x]);
    if (rec < 0) return;
    const aew_gemm_tn_t& g = descs[rec >> 22];
    const int tile = rec & 0xfff;
    const int nkt = tn8_ori(g.N_pad, g.K_total) ? (g.K_total / 128 + 1) / 2 : g.K_total / 128;
    tn8_tile<true>(g, smem, tile % nkt, tile / nkt, g.out);
}

// =============================================================================================
// TN kernel, fp32: out tile 64 (k) x 64 (n), contraction staged 32 rows at a time, operands by
// ds_read_b32 (v_mfma_f32_16x16x4_f32 takes one row of the contraction per 16-lane group).
// =============================================================================================
#define TF_BT 64
#define TF_RC 32
#define TF_STAGE_BYTES (2 * TF_RC * 256)            // 16 KiB

__global__ __launch_bounds__(256) void k_gemm_tn_f32(const aew_gemm_tn_t g, int splits, int rows_per_split,
                                                     int fold_batch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave & 1, wn = wave >> 1;
    const int nkt = g.K_total / TF_BT;
    const int kt = blockIdx.x % nkt, nt = blockIdx.x / nkt;
    const int n0 = nt * TF_BT;
    const int sp = blockIdx.y;
    const TnTile tt = tn_locate(g, kt, TF_BT);
    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int b_lo = fold_batch ? 0 : blockIdx.z, b_hi = fold_batch ? g.batch : blockIdx.z + 1;
    const int r_lo = sp * rows_per_split, r_hi = min(g.Mc, r_lo + rows_per_split);
    const int nst = (r_hi - r_lo + TF_RC - 1) / TF_RC;
    const int total = nst * (b_hi - b_lo);
    TnPtrs<2> P;
    int st_in_b = 0, bcur = b_lo;
    if (total > 0) {
        tn_setup<2, 4, TF_RC>(g, tt, bcur, r_lo, n0, wave, lane, P);
        tn_issue<2, TF_RC>(g, tt, smem, r_hi, wave, P);
    }
    const int q = lane & 15, kq = lane >> 4;
    for (int t = 0; t < total; ++t) {
        wait_vm0();
        __syncthreads();
        if (t + 1 < total) {
            if (++st_in_b == nst) {
                st_in_b = 0; ++bcur;
                tn_setup<2, 4, TF_RC>(g, tt, bcur, r_lo, n0, wave, lane, P);
            }
            tn_issue<2, TF_RC>(g, tt, smem + ((t + 1) & 1) * TF_STAGE_BYTES, r_hi, wave, P);
        }
        const char* gs = smem + (t & 1) * TF_STAGE_BYTES;
        const char* as = gs + TF_RC * 256;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int row = ks * 4 + kq;
            float af[2], gf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int col = wk * 32 + i * 16 + q;
                af[i] = *reinterpret_cast<const float*>(as + row * 256 + (tn_swz_f32(row, col >> 2) << 4) + ((col & 3) << 2));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = wn * 32 + j * 16 + q;
                gf[j] = *reinterpret_cast<const float*>(gs + row * 256 + (tn_swz_f32(row, col >> 2) << 4) + ((col & 3) << 2));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], gf[j], acc[i][j], 0, 0, 0);
        }
    }
    const int slab = fold_batch ? sp : (blockIdx.z * splits + sp);
    float* out = g.out + (int64_t)slab * g.out_batch_stride;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 32 + j * 16 + q;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = tt.koff + wk * 32 + i * 16 + 4 * kq;
            *reinterpret_cast<float4*>(out + (int64_t)n * g.K_total + k) =
                make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    }
}

// TN check kernel: one thread per output element, ascending (b, m) order.
template <typename T>
__global__ void k_gemm_tn_check(const aew_gemm_tn_t g, int splits, int rows_per_split, int fold_batch) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    const int nslab_b = fold_batch ? 1 : g.batch;
    const int slab = blockIdx.z;                      // (b, sp) or sp
    const int sp = fold_batch ? slab : slab % splits;
    const int bz = fold_batch ? 0 : slab / splits;
    (void)nslab_b;
    if (k >= g.K_total) return;
    int s = 0, kin = k;
    while (kin >= g.seg[s].k_len) { kin -= g.seg[s].k_len; ++s; }
    const aew_seg_t& sa = g.seg[s];
    const int r_lo = sp * rows_per_split, r_hi = min(g.Mc, r_lo + rows_per_split);
    float acc = 0.f;
    const int b_lo = fold_batch ? 0 : bz, b_hi = fold_batch ? g.batch : bz + 1;
    for (int b = b_lo; b < b_hi; ++b)
        for (int m = r_lo; m < r_hi; ++m) {
            const int64_t rg = (int64_t)m * g.g.row_step + g.g.row_off;
            const int64_t ra = (int64_t)m * sa.row_step + sa.row_off;
            if (rg < g.g.row_lo || rg >= g.g.row_hi || ra < sa.row_lo || ra >= sa.row_hi) continue;
            const float gv = ld_elem<T>(reinterpret_cast<const T*>(g.g.ptr) + (int64_t)b * g.g.batch_stride + rg * g.g.row_pitch + n);
            const float av = ld_elem<T>(reinterpret_cast<const T*>(sa.ptr) + (int64_t)b * sa.batch_stride + ra * sa.row_pitch + kin);
            acc = fmaf(gv, av, acc);
        }
    g.out[(int64_t)slab * g.out_batch_stride + (int64_t)n * g.K_total + k] = acc;
}

#include "aew_win.hip"                                // k_gemm_nt_bf16_win: one LDS window for both dilation taps
#include "aew_chain.hip"                              // k_nt_chain: a run of dependent NT GEMMs as one launch

#ifndef AEW_DEV_KERNELS_ONLY   /* a development TU (ISA inspection of single kernels) stops here */
// =============================================================================================
// host-side launchers
// =============================================================================================

// kernels using more than 64 KiB of dynamic LDS must opt in once per process
static int ensure_big_lds() {
    static std::atomic<unsigned long long> done{0};               // one bit per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long dev_bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & dev_bit) return 0;
    hipError_t e;
#define AEW_SET_LDS(fn, bytes)                                                                       \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); \
    if (e != hipSuccess) return (int)e;
#define NT_LDS_BYTES (NtCfg<4, 1>::LDS_BYTES)
#define AEW_SET_NT(EPI)                                                   \
    AEW_SET_LDS((k_gemm_nt_bf16_p64<EPI, 8, 2, 4>), (P64Cfg<8, 2, 4>::LDS_BYTES)) \
    AEW_SET_LDS((k_gemm_nt_bf16_p64<EPI, 4, 2, 2>), (P64Cfg<4, 2, 2>::LDS_BYTES)) \
    AEW_SET_LDS((k_gemm_nt_bf16_p64<EPI, 4, 4, 2>), (P64Cfg<4, 4, 2>::LDS_BYTES)) \
    AEW_SET_LDS((k_gemm_nt_bf16_p64<EPI, 4, 1, 2>), (P64Cfg<4, 1, 2>::LDS_BYTES)) \
    AEW_SET_LDS((k_gemm_nt_bf16_p64<EPI, 4, 1, 2, 5>), (5 * P64Cfg<4, 1, 2>::STAGE_BYTES)) \
    AEW_SET_LDS((k_gemm_nt_bf16_p64<EPI, 1, 4, 2, 5>), (5 * P64Cfg<1, 4, 2>::STAGE_BYTES)) \
    AEW_SET_LDS((k_gemm_nt_bf16_p64<EPI, 1, 4, 1, 5>), (5 * P64Cfg<1, 4, 1>::STAGE_BYTES)) \
    AEW_SET_LDS((k_gemm_nt_bf16_pipe<EPI, 1>), (NtCfg<8, 1>::LDS_BYTES))      \
    AEW_SET_LDS((k_gemm_nt_bf16_pipe<EPI, 2>), (NtCfg<8, 2>::LDS_BYTES))      \
    AEW_SET_LDS((k_gemm_nt_bf16<EPI, false, 8>), NT_LDS_BYTES)           \
    AEW_SET_LDS((k_gemm_nt_bf16<EPI, false, 8, 2>), (NtCfg<8, 2>::LDS_BYTES)) \
    AEW_SET_LDS((k_gemm_nt_bf16<EPI, false, 4, 2>), (NtCfg<4, 2>::LDS_BYTES)) \
    AEW_SET_LDS((k_gemm_nt_bf16<EPI, false, 4>), NT_LDS_BYTES)          \
    AEW_SET_LDS((k_gemm_nt_bf16<EPI, false, 3, 1, 192>), (NtCfg<3, 1, 192>::LDS_BYTES)) \
    AEW_SET_LDS((k_gemm_nt_bf16<EPI, false, 4, 1, 128>), (NtCfg<4, 1, 128>::LDS_BYTES)) \
    AEW_SET_LDS((k_gemm_nt_bf16<EPI, false, 4, 1, 256, 6>), (6 * NtCfg<4, 1, 256>::STAGE_BYTES)) \
    AEW_SET_LDS((k_gemm_nt_bf16<EPI, false, 8, 2, 256, 5>), (5 * NtCfg<8, 2, 256>::STAGE_BYTES))
    AEW_SET_NT(AEW_EPI_STORE)
    AEW_SET_NT(AEW_EPI_GATED)
    AEW_SET_NT(AEW_EPI_RES_SKIP)
    AEW_SET_NT(AEW_EPI_DFG)
#if AEW_FN_ABLATE       /* tools library only: the ablation variants (ABL = true) do not exist in the product build */
    AEW_SET_LDS((k_gemm_nt_bf16<AEW_EPI_GATED, true, 8>), NT_LDS_BYTES)
    AEW_SET_LDS((k_gemm_nt_bf16<AEW_EPI_GATED, true, 8, 2>), (NtCfg<8, 2>::LDS_BYTES))
    AEW_SET_LDS((k_gemm_nt_bf16<AEW_EPI_GATED, true, 4>), NT_LDS_BYTES)
    AEW_SET_LDS((k_gemm_nt_bf16<AEW_EPI_STORE, true, 4>), NT_LDS_BYTES)
    AEW_SET_LDS((k_gemm_nt_bf16<AEW_EPI_DFG, true, 4>), NT_LDS_BYTES)
#endif
#undef AEW_SET_NT
    AEW_SET_LDS((k_gemm_nt_f32<1, 7, 0>), (NfCfg<1, 7>::LDS_BYTES))
    AEW_SET_LDS((k_gemm_nt_f32<1, 14, 0>), (NfCfg<1, 14>::LDS_BYTES))
    AEW_SET_LDS((k_gemm_nt_f32<2, 12, 0>), (NfCfg<2, 12>::LDS_BYTES))
    AEW_SET_LDS((k_gemm_nt_f32<1, 7, 1>), (NfCfg<1, 7>::LDS_BYTES))
    AEW_SET_LDS((k_gemm_nt_f32<1, 14, 1>), (NfCfg<1, 14>::LDS_BYTES))
    AEW_SET_LDS((k_gemm_nt_f32<2, 12, 1>), (NfCfg<2, 12>::LDS_BYTES))
    AEW_SET_LDS((k_gemm_nt_f32<4, 4, 0>), (NfCfg<4, 4>::LDS_BYTES))
    AEW_SET_LDS((k_gemm_nt_f32<4, 4, 1>), (NfCfg<4, 4>::LDS_BYTES))
    AEW_SET_LDS(k_gemm_tn_bf16_big, TNB_LDS_BYTES)
    AEW_SET_LDS(k_gemm_tn_bf16<0>, TN_LDS_BYTES)
    AEW_SET_LDS(k_gemm_tn_bf16<1>, TN_LDS_BYTES)
    AEW_SET_LDS(k_gemm_tn_bf16_grp, TN_LDS_BYTES)
    AEW_SET_LDS(k_gemm_tn_bf16_grp_cur, TN_LDS_BYTES)
    AEW_SET_LDS(k_gemm_tn_bf16_big_grp, TNB_LDS_BYTES)
    AEW_SET_LDS(k_gemm_tn_bf16_grp8, TN8_LDS_BYTES)
    AEW_SET_LDS(k_gemm_tn_bf16_grp32, TN_LDS_BYTES)
    AEW_SET_LDS((k_nt_chain<0>), CHAIN_LDS_BYTES)
    AEW_SET_LDS((k_nt_chain<1>), CHAIN_LDS_BYTES)
#undef AEW_SET_LDS
    done.fetch_or(dev_bit, std::memory_order_release);
    return 0;
}

static int check_seg(const aew_seg_t& s, int esize, int ktile) {
    if (!s.ptr) return AEW_E_ARG;
    if (s.k_len <= 0 || s.k_len % ktile) return AEW_E_ARG;
    if (((uintptr_t)s.ptr & 15) || ((int64_t)s.row_pitch * esize) % 16 || ((int64_t)s.batch_stride * esize) % 16)
        return AEW_E_ALIGN;
    return 0;
}

// full-N kernels (aew_fn.hip, same translation unit)
static bool fn_supported(const aew_gemm_nt_t& g);
static int launch_fn(const aew_gemm_nt_t& g, hipStream_t st);
extern int g_fn_enable_flag();

// the 64 x 64 shape of launches of very few blocks (see the p64 kernel's table): does the launcher take it for g, under
// the calling thread's tuning record?  *blocks = its grid (one K range)
static bool nt_small64(const aew_gemm_nt_t& g, int* blocks) {
    if (g.dtype != AEW_BF16 || g.impl != 0 || g.N_pad % 128) return false;
    for (int s = 0; s < g.n_segs; ++s)
        if (g.seg[s].k_len * 2 > AEW_ZERO_SPAN) return false;
    const int tiles256 = ((g.M + NT_BM - 1) / NT_BM) * g.batch * (g.N_pad / NT_BN);
    const bool p64r = AEW_T().nt_wave_rows == 64 && AEW_T().nt_small_tiles > 0 && tiles256 <= AEW_T().nt_small_tiles;
    if (!(p64r && AEW_T().nt_small_w8 && AEW_T().nt_small_n64 > 0 &&
          ((g.M + 63) / 64) * g.batch * (g.N_pad / 128) <= AEW_T().nt_small_n64))
        return false;
    const int row_tiles = ((g.M + 63) / 64) * g.batch;
    *blocks = ((row_tiles + 7) / 8) * 8 * (g.N_pad / 64);
    return true;
}

extern "C" int aew_gemm_nt_small_split(const aew_gemm_nt_t* g, int target_blocks, int* k_split, int64_t* ws_bytes, int* n_tickets) {
    if (!g || !k_split || !ws_bytes || !n_tickets) return AEW_E_ARG;
    *k_split = 1;
    *ws_bytes = 0;
    *n_tickets = 0;
    int blocks = 0;
    if (!nt_small64(*g, &blocks)) return 0;
    const int kt = g->K_total / 64;
    int best = 1;
    for (int S = 2; S <= 8; ++S)                             // at least four K tiles per range: the ring is five deep
        if (kt % S == 0 && kt / S >= 4 && blocks * S <= target_blocks) best = S;
    if (best == 1) return 0;
    *k_split = best;
    *ws_bytes = (int64_t)best * blocks * (P64Cfg<1, 4, 1>::NW * 4 * 64 * 16);
    *n_tickets = blocks * P64Cfg<1, 4, 1>::NW;
    return 0;
}

static int launch_gemm_nt(const aew_gemm_nt_t& g, hipStream_t st) {
    if (g.n_segs < 1 || g.n_segs > AEW_MAX_SEGS || g.M <= 0 || g.batch <= 0 || !g.W) return AEW_E_ARG;
    if (g.W2) {
        // fused gated layer: z tile -> residual 1x1 (see aewavenet.h)
        if (g.epi != AEW_EPI_GATED || g.dtype != AEW_BF16 || g.N2 <= 0 || g.N2 > g.N2_pad || (g.N2 & 7) || g.N2_pad % 128 ||
            !g.out3.ptr || !g.out0.ptr || g.out0.dtype != AEW_BF16 || ((uintptr_t)g.W2 & 15))
            return AEW_E_ARG;
        {   // the residual GEMM consumes z for every row m in [0, M): out0 must hold them all
            const int64_t r0 = g.out0.row_off, r1 = (int64_t)(g.M - 1) * g.out0.row_step + g.out0.row_off;
            if (r0 < g.out0.row_lo || r0 >= g.out0.row_hi || r1 < g.out0.row_lo || r1 >= g.out0.row_hi) return AEW_E_ARG;
        }
        if (!(g.impl == 2 && g_fn_enable_flag() && fn_supported(g))) {
            // unfused execution with the same result: GATED, then a STORE | ADD_AUX0 GEMM over the z it wrote
            aew_gemm_nt_t a = g;
            a.W2 = nullptr;
            int rc = launch_gemm_nt(a, st);
            if (rc) return rc;
            aew_gemm_nt_t b2 = {};
            b2.dtype = AEW_BF16; b2.impl = g.impl == 2 ? 0 : g.impl;
            b2.M = g.M; b2.N = g.N2; b2.N_pad = g.N2_pad; b2.batch = g.batch;
            b2.n_segs = 1; b2.K_total = g.N_pad / 2;
            b2.seg[0].ptr = g.out0.ptr; b2.seg[0].batch_stride = g.out0.batch_stride; b2.seg[0].row_pitch = g.out0.row_pitch;
            b2.seg[0].row_step = g.out0.row_step; b2.seg[0].row_off = g.out0.row_off;
            b2.seg[0].row_lo = g.out0.row_lo; b2.seg[0].row_hi = g.out0.row_hi; b2.seg[0].k_len = g.N_pad / 2;
            b2.W = g.W2; b2.epi = AEW_EPI_STORE; b2.flags = g.aux0.ptr ? AEW_EF_ADD_AUX0 : 0;
            b2.out0 = g.out3; b2.aux0 = g.aux0;
            return launch_gemm_nt(b2, st);
        }
    }
    if (g.impl == 2 && g.dtype == AEW_BF16) {
        if (g_fn_enable_flag() && fn_supported(g)) {
            const int es2 = 2;
            int ks = 0;
            for (int s = 0; s < g.n_segs; ++s) {
                const int rc = check_seg(g.seg[s], es2, 64);
                if (rc) return rc;
                ks += g.seg[s].k_len;
            }
            if (ks != g.K_total || g.N > g.N_pad || (g.N & 7)) return AEW_E_ARG;
            return launch_fn(g, st);
        }
        aew_gemm_nt_t a = g;                                  // not covered: the tiled kernel (same results)
        a.impl = 0;
        return launch_gemm_nt(a, st);
    }
    const int es = g.dtype == AEW_BF16 ? 2 : 4;
    const int kt = g.dtype == AEW_BF16 ? 64 : NF_BK;        // ABI contract: bf16 segments are 64-aligned
    const int ntile = g.dtype == AEW_BF16 ? NT_BN : NF_BN;
    int ksum = 0;
    for (int s = 0; s < g.n_segs; ++s) {
        const int rc = check_seg(g.seg[s], es, kt);
        if (rc) return rc;
        ksum += g.seg[s].k_len;
    }
    if (ksum != g.K_total || g.N_pad % ntile || g.N > g.N_pad || (g.N & (g.dtype == AEW_BF16 ? 7 : 3))) return AEW_E_ARG;
    if (g.epi == AEW_EPI_RES_SKIP && (g.n_split % ntile)) return AEW_E_ARG;
    if (g.epi != AEW_EPI_STORE && g.dtype != AEW_BF16) return AEW_E_UNSUP;
    if (g.dtype == AEW_BF16 && g.impl != 1 && (g.epi == AEW_EPI_STORE || g.epi == AEW_EPI_DFG) &&
        ((g.aux0.ptr && g.aux0.dtype != AEW_BF16) || (g.aux1.ptr && g.aux1.dtype != AEW_BF16)))
        return AEW_E_UNSUP;                                  // the MFMA epilogues prefetch aux rows as bf16
    if (g.dtype == AEW_BF16 && g.impl != 1 && g.epi == AEW_EPI_STORE && (g.flags & AEW_EF_OUT2_COPY))
        return AEW_E_UNSUP;                                  // the copy output exists in the fp32 / check epilogues only
    if (g.impl == 1) {
        const int nq = (g.epi == AEW_EPI_GATED) ? g.N_pad / 8 : g.N_pad / 4;
        dim3 grid((nq + 63) / 64, g.M, g.batch);
        if (g.dtype == AEW_BF16) hipLaunchKernelGGL(k_gemm_nt_check<uint16_t>, grid, dim3(64), 0, st, g);
        else hipLaunchKernelGGL(k_gemm_nt_check<float>, grid, dim3(64), 0, st, g);
    } else if (g.dtype == AEW_BF16) {
        const int rc = ensure_big_lds();
        if (rc) return rc;
        bool zspan = true;                                   // masked rows stream from aew_zero_region
        for (int s = 0; s < g.n_segs; ++s) zspan = zspan && g.seg[s].k_len * 2 <= AEW_ZERO_SPAN;
        const bool wide = AEW_T().nt_wave_rows == 256 && g.N_pad % 256 == 0 && !(g.epi == AEW_EPI_RES_SKIP && g.n_split % 256);
        const bool p64 = wide && AEW_T().nt_pipe == 2 && zspan;    // 256 x 256 tiles, K tiles of 64
        // 256 x 256 tiles as SIXTEEN waves of 64 x 64 (A/B, round 6: the A operand staged once per two N tiles, with the wave
        // count of two 8-wave blocks - what the 8-fat-wave form of this tile lacks); one block per CU, 3 x 32 KiB ring
        const bool wide16 = AEW_T().nt_wave_rows == 512 && g.N_pad % 256 == 0 && g.epi != AEW_EPI_RES_SKIP;
        // launches that would be a small fraction of one tile wave use 64-row tiles (default shape only)
        const int tiles256 = ((g.M + NT_BM - 1) / NT_BM) * g.batch * (g.N_pad / NT_BN);
        const bool p64r = AEW_T().nt_wave_rows == 64 && AEW_T().nt_small_tiles > 0 && tiles256 <= AEW_T().nt_small_tiles && zspan;
        // memory-bound launches (the K loop of G2 is 8 steps, dz carries 100 MB of epilogue operands): 128-row tiles, so
        // that the launch is several tile waves and one block's stores run under another's K loop
        const bool memb = AEW_T().nt_mem128 && AEW_T().nt_wave_rows == 64 && !p64r && !win_dwp(g) && zspan &&
                          (g.epi == AEW_EPI_DFG || (g.epi == AEW_EPI_STORE && g.K_total <= 256));
        const bool t128 = memb && AEW_T().nt_mem128 == 1;
        const bool p128 = !p64r && (AEW_T().nt_wave_rows == 0 || (memb && AEW_T().nt_mem128 == 2)) && zspan;      // 128 x 128 tiles, K tiles of 64
        const bool p256 = AEW_T().nt_wave_rows == 1 && zspan;      // 256 x 128 tiles, K tiles of 64, one block per CU
        // 192-row tiles (8 waves of 48 x 64) where they shorten the launch.  Blocks spread over the 256 CUs before
        // they double up, and a CU is MFMA-bound with one block already, so a launch costs about
        // ceil(tiles / 256) * rows-per-tile; the 192-row shape is ~5 % less efficient per row (12 MFMAs per wave
        // and K step instead of 16), hence the margin.
        bool t192 = false;
        if (AEW_T().nt_wave_rows == 64 && !p64r && AEW_T().nt_rows192 && !memb) {
            const int tiles192 = ((g.M + 191) / 192) * g.batch * (g.N_pad / NT_BN);
            const int c256 = ((tiles256 + 255) / 256) * 256, c192 = ((tiles192 + 255) / 256) * 192;
            t192 = AEW_T().nt_rows192 == 2 || c192 * 10 < c256 * 9;
        }
        const bool deep2 = (AEW_T().nt_deep == 2 || AEW_T().nt_deep == 3) && AEW_T().nt_wave_rows == 64 && !p64r && !memb && g.N_pad % 256 == 0 &&
                           !(g.epi == AEW_EPI_RES_SKIP && g.n_split % 256);
        const bool deep1 = (AEW_T().nt_deep == 1 || AEW_T().nt_deep == 2) && !deep2 && AEW_T().nt_wave_rows == 64 && !p64r && !memb;
        if (deep1 || deep2) t192 = false;
        if (!p64r && AEW_T().nt_wave_rows == 64 && !deep1 && !deep2) {   // both taps of a dilated pair from one LDS window
            const int dwp = win_dwp(g);
            if (dwp) return launch_win(g, dwp, t192, st);
        }
        // 64 x 64 tiles for launches of very few 64 x 128 blocks (see the p64 kernel's table)
        int blocks64 = 0;
        const bool p64n = nt_small64(g, &blocks64);
        const int bm = p64r ? 64 : ((p128 || t128) ? 128 : (t192 ? 192 : NT_BM)), bn = p64n ? 64 : ((p128 || p64r) ? 128 : ((wide || deep2 || wide16) ? 256 : NT_BN));
        const int row_tiles = ((g.M + bm - 1) / bm) * g.batch;
        dim3 grid(((row_tiles + 7) / 8) * 8 * (g.N_pad / bn));
        // split-K of a small launch: honoured by the 64 x 64 shape (every other shape contracts the whole K axis; the
        // plan's slabs then stay unused).  bf16 has no canonical order to keep: the partial sums are added in ascending order
        if (g.k_split > 1 && p64n && !(AEW_FN_ABLATE && g.reserved)) {
            const int64_t need = (int64_t)g.k_split * grid.x * (P64Cfg<1, 4, 1>::NW * 4 * 64 * 16);
            if (g.k_split > 8 || g.K_total % (64 * g.k_split) || !g.ksplit_ws || !g.ksplit_tickets || ((uintptr_t)g.ksplit_ws & 15) ||
                need > 0x7fffffff)
                return AEW_E_ARG;
            grid.x *= g.k_split;
        }
#define AEW_NT_GO(EPI, ABL)                                                                                   \
    do {                                                                                                      \
        if (!ABL && deep2)                                                                                     \
            hipLaunchKernelGGL((k_gemm_nt_bf16<EPI, false, 8, 2, 256, 5>), grid, dim3((NtCfg<8, 2>::THREADS)), (5 * NtCfg<8, 2, 256>::STAGE_BYTES), st, g); \
        else if (!ABL && deep1)                                                                                \
            hipLaunchKernelGGL((k_gemm_nt_bf16<EPI, false, 4, 1, 256, 6>), grid, dim3((NtCfg<4>::THREADS)), (6 * NtCfg<4, 1, 256>::STAGE_BYTES), st, g); \
        else if (!ABL && t192)                                                                                 \
            hipLaunchKernelGGL((k_gemm_nt_bf16<EPI, false, 3, 1, 192>), grid, dim3((NtCfg<3, 1, 192>::THREADS)), (NtCfg<3, 1, 192>::LDS_BYTES), st, g); \
        else if (!ABL && t128)                                                                                 \
            hipLaunchKernelGGL((k_gemm_nt_bf16<EPI, false, 4, 1, 128>), grid, dim3((NtCfg<4, 1, 128>::THREADS)), (NtCfg<4, 1, 128>::LDS_BYTES), st, g); \
        else if (!ABL && p64n)                                                                                 \
            hipLaunchKernelGGL((k_gemm_nt_bf16_p64<EPI, 1, 4, 1, 5>), grid, dim3(256), (5 * P64Cfg<1, 4, 1>::STAGE_BYTES), st, g); \
        else if (!ABL && p64r && (int)grid.x <= AEW_T().nt_small_deep && AEW_T().nt_small_w8)                              \
            hipLaunchKernelGGL((k_gemm_nt_bf16_p64<EPI, 1, 4, 2, 5>), grid, dim3(512), (5 * P64Cfg<1, 4, 2>::STAGE_BYTES), st, g); \
        else if (!ABL && p64r && (int)grid.x <= AEW_T().nt_small_deep)                                               \
            hipLaunchKernelGGL((k_gemm_nt_bf16_p64<EPI, 4, 1, 2, 5>), grid, dim3(128), (5 * P64Cfg<4, 1, 2>::STAGE_BYTES), st, g); \
        else if (!ABL && p64r)                                                                                 \
            hipLaunchKernelGGL((k_gemm_nt_bf16_p64<EPI, 4, 1, 2>), grid, dim3(128), (P64Cfg<4, 1, 2>::LDS_BYTES), st, g); \
        else if (!ABL && p256)                                                                                 \
            hipLaunchKernelGGL((k_gemm_nt_bf16_p64<EPI, 4, 4, 2>), grid, dim3(512), (P64Cfg<4, 4, 2>::LDS_BYTES), st, g); \
        else if (!ABL && p128)                                                                                 \
            hipLaunchKernelGGL((k_gemm_nt_bf16_p64<EPI, 4, 2, 2>), grid, dim3(256), (P64Cfg<4, 2, 2>::LDS_BYTES), st, g); \
        else if (!ABL && p64)                                                                                  \
            hipLaunchKernelGGL((k_gemm_nt_bf16_p64<EPI, 8, 2, 4>), grid, dim3(512), (P64Cfg<8, 2, 4>::LDS_BYTES), st, g); \
        else if (!ABL && wide16)                                                                               \
            hipLaunchKernelGGL((k_gemm_nt_bf16<EPI, false, 4, 2>), grid, dim3((NtCfg<4, 2>::THREADS)), (NtCfg<4, 2>::LDS_BYTES), st, g); \
        else if (!ABL && AEW_T().nt_pipe && wide)                                                                   \
            hipLaunchKernelGGL((k_gemm_nt_bf16_pipe<EPI, 2>), grid, dim3((NtCfg<8, 2>::THREADS)), (NtCfg<8, 2>::LDS_BYTES), st, g); \
        else if (!ABL && AEW_T().nt_pipe && AEW_T().nt_wave_rows == 128)                                                  \
            hipLaunchKernelGGL((k_gemm_nt_bf16_pipe<EPI, 1>), grid, dim3((NtCfg<8, 1>::THREADS)), (NtCfg<8, 1>::LDS_BYTES), st, g); \
        else if (wide)                                                                                        \
            hipLaunchKernelGGL((k_gemm_nt_bf16<EPI, ABL, 8, 2>), grid, dim3((NtCfg<8, 2>::THREADS)), (NtCfg<8, 2>::LDS_BYTES), st, g); \
        else if (AEW_T().nt_wave_rows == 128)                                                                       \
            hipLaunchKernelGGL((k_gemm_nt_bf16<EPI, ABL, 8>), grid, dim3(NtCfg<8>::THREADS), NT_LDS_BYTES, st, g); \
        else                                                                                                  \
            hipLaunchKernelGGL((k_gemm_nt_bf16<EPI, ABL, 4>), grid, dim3(NtCfg<4>::THREADS), NT_LDS_BYTES, st, g); \
    } while (0)
        switch (g.epi) {
            case AEW_EPI_STORE:
#if AEW_FN_ABLATE
                if (g.reserved && AEW_T().nt_wave_rows == 64) { AEW_NT_GO(AEW_EPI_STORE, true); break; }
#endif
                AEW_NT_GO(AEW_EPI_STORE, false);
                break;
            case AEW_EPI_GATED:
#if AEW_FN_ABLATE
                if (g.reserved) { AEW_NT_GO(AEW_EPI_GATED, true); break; }
#endif
                AEW_NT_GO(AEW_EPI_GATED, false);
                break;
            case AEW_EPI_RES_SKIP: AEW_NT_GO(AEW_EPI_RES_SKIP, false); break;
            case AEW_EPI_DFG:
#if AEW_FN_ABLATE
                if (g.reserved && AEW_T().nt_wave_rows == 64) { AEW_NT_GO(AEW_EPI_DFG, true); break; }
#endif
                AEW_NT_GO(AEW_EPI_DFG, false);
                break;
            default: return AEW_E_UNSUP;
        }
#undef AEW_NT_GO
    } else {
        const int rc = ensure_big_lds();
        if (rc) return rc;
        // shape by block count (see the kernel's header): one block per CU with the whole LDS as ring where the launch
        // is that small - 16-row tiles first (more, shorter chains), else 32-row tiles - and the 3-blocks-per-CU shape
        // for anything larger
        const int rows = g.M * g.batch, n_nt = g.N_pad / NF_BN;
        const int ksp = g.k_split > 1 ? g.k_split : 1;
        if (ksp > 1) {
            if ((ksp != 2 && ksp != 4) || g.impl != 0 || g.epi != AEW_EPI_STORE || !g.ksplit_ws || !g.ksplit_tickets ||
                g.K_total % (NF_BK * ksp) || ((uintptr_t)g.ksplit_ws & 15) || (int64_t)ksp * ((rows + 31) & ~31) * g.N_pad * 4 > 0x7fffffff)
                return AEW_E_ARG;
        }
        // (workgroups: tiles x k ranges; the shape thresholds see the whole grid)
        const int tiles1 = ((rows + 15) / 16) * n_nt * ksp, tiles2 = ((rows + 31) / 32) * n_nt * ksp;
#define AEW_NF_GO(RT, S, GRID)                                                                                       \
    do {                                                                                                             \
        if (AEW_T().nf_loaders) hipLaunchKernelGGL((k_gemm_nt_f32<RT, S, 1>), dim3(GRID), dim3(512), (NfCfg<RT, S>::LDS_BYTES), st, g); \
        else hipLaunchKernelGGL((k_gemm_nt_f32<RT, S, 0>), dim3(GRID), dim3(256), (NfCfg<RT, S>::LDS_BYTES), st, g);  \
    } while (0)
        // split-K launches: 64-row tiles (four chains per wave share one W fragment: four times the MFMAs per weight byte
        // staged; the k ranges restore the workgroup count), two workgroups per CU on a 4-stage ring
        if (ksp > 1 && rows >= 64) AEW_NF_GO(4, 4, ((rows + 63) / 64) * n_nt * ksp);
        else if (AEW_T().nf_deep && tiles1 <= AEW_T().nf_deep) AEW_NF_GO(1, 14, tiles1);
        else if (AEW_T().nf_deep && tiles2 <= AEW_T().nf_deep) AEW_NF_GO(2, 12, tiles2);
        else AEW_NF_GO(1, 7, tiles1);
#undef AEW_NF_GO
    }
    return (int)hipGetLastError();
}


// does this op run on the big-tile kernel?  (bf16, MFMA path, at least one full 256 x 256 tile's worth of output)
static bool tn_use_big(const aew_gemm_tn_t& g) {
    return AEW_T().tn_big && !AEW_T().tn_safe && g.dtype == AEW_BF16 && g.impl != 1 && g.N_pad >= 256 && g.K_total >= 256 &&
           (int64_t)g.Mc * g.batch > AEW_T().tn_fold_rows;
}

// split heuristic: aim for >= ~2 blocks per CU
static void tn_plan(const aew_gemm_tn_t& g, int tile, int rc, int* splits, int* rps, int* fold) {
    if (tn_use_big(g)) {
        const int tiles = ((g.N_pad / 128 + 1) / 2) * ((g.K_total / 128 + 1) / 2);
        int want = (AEW_T().tn_big_target + tiles * g.batch - 1) / (tiles * g.batch);
        int max_sp = (g.Mc + 8 * rc - 1) / (8 * rc);       // keep >= 8 stages per block
        if (max_sp < 1) max_sp = 1;
        int sp = want < 1 ? 1 : (want > max_sp ? max_sp : want);
        int r = (g.Mc + sp - 1) / sp;
        r = ((r + rc - 1) / rc) * rc;
        sp = (g.Mc + r - 1) / r;
        *splits = sp; *rps = r; *fold = 0;
        return;
    }
    const int tiles = (g.N_pad / tile) * (g.K_total / tile);
    int f = ((int64_t)g.Mc * g.batch <= AEW_T().tn_fold_rows) ? 1 : 0;   // short contractions: fold the batch loop
    int slabs_b = f ? 1 : g.batch;
    // small outputs (a handful of tiles) would need ~100 row splits to fill the chip on their own, and
    // every split is a slab that is written here and read again by the unpack; they run on the side lane
    // next to chip-filling kernels, so they are split much less
    const int target = (tiles <= AEW_T().tn_small_tiles) ? AEW_T().tn_small_target : AEW_T().tn_target_blocks;
    int want = (target + tiles * slabs_b - 1) / (tiles * slabs_b);
    int max_sp = (g.Mc + 4 * rc - 1) / (4 * rc);        // keep >= 4 stages per block
    if (max_sp < 1) max_sp = 1;
    int sp = want < 1 ? 1 : (want > max_sp ? max_sp : want);
    if (f) sp = 1;
    int r = (g.Mc + sp - 1) / sp;
    r = ((r + rc - 1) / rc) * rc;
    sp = (g.Mc + r - 1) / r;
    *splits = sp; *rps = r; *fold = f;
}

extern "C" int aew_tn_slabs(const aew_gemm_tn_t* g) {
    // number of fp32 partial slabs the TN op writes (the unpack step sums them)
    int sp, rps, fold;
    const int tile = g->dtype == AEW_BF16 ? TN_BT : TF_BT;
    const int rc = g->dtype == AEW_BF16 ? TN_RC : TF_RC;
    tn_plan(*g, tile, rc, &sp, &rps, &fold);
    return fold ? sp : g->batch * sp;
}

extern "C" int aew_tn_fold(const aew_gemm_tn_t* g) {
    int sp, rps, fold;
    const int tile = g->dtype == AEW_BF16 ? TN_BT : TF_BT;
    const int rc = g->dtype == AEW_BF16 ? TN_RC : TF_RC;
    tn_plan(*g, tile, rc, &sp, &rps, &fold);
    return fold;
}

extern "C" int aew_set_tn_fold_rows(int rows) { g_tune.tn_fold_rows = rows; return 0; }

// host copy of a group's descriptors is not available (they live in device memory): the builder validated them with
// aew_tn_group_check before uploading
extern "C" int aew_tn_group_check(const aew_gemm_tn_t* g) {
    if (!g) return AEW_E_ARG;
    if (g->dtype != AEW_BF16 || g->n_segs < 1 || g->n_segs > AEW_MAX_SEGS || g->Mc <= 0 || g->batch <= 0 || !g->out)
        return AEW_E_ARG;
    int ksum = 0;
    for (int s = 0; s < g->n_segs; ++s) {
        const int rc = check_seg(g->seg[s], 2, TN_BT);
        if (rc) return rc;
        ksum += g->seg[s].k_len;
    }
    aew_seg_t gg = g->g; gg.k_len = TN_BT;
    const int rc = check_seg(gg, 2, TN_BT);
    if (rc) return rc;
    if (ksum != g->K_total || g->N_pad % TN_BT || (g->N_pad / TN_BT) * (g->K_total / TN_BT) > 0xfff) return AEW_E_ARG;
    if (g->snap_out && (g->snap_k < -1 || g->snap_k >= g->K_total)) return AEW_E_ARG;
    if (g->grp_splits < 0 || g->grp_splits * g->batch > 0x3ff) return AEW_E_ARG;
    if (g->grp_splits > 0 && (g->grp_rows <= 0 || g->grp_rows % TN_RC || (int64_t)g->grp_splits * g->grp_rows < g->Mc ||
                              g->snap_out || g->colsum_out))
        return AEW_E_ARG;
    return 0;
}

static int launch_gemm_tn_group(const aew_gemm_tn_group_t& p, hipStream_t st) {
    if (!p.descs || !p.tile_map || p.n_descs <= 0 || p.n_blocks <= 0) return AEW_E_ARG;
    const int rc = ensure_big_lds();
    if (rc) return rc;
    if (p.tile == 256)
        hipLaunchKernelGGL(k_gemm_tn_bf16_big_grp, dim3(p.n_blocks), dim3(TNB_THREADS), TNB_LDS_BYTES, st, p.descs, p.tile_map);
    else if (p.tile == 384)
        hipLaunchKernelGGL(k_gemm_tn_bf16_grp8, dim3(p.n_blocks), dim3(TN8_THREADS), TN8_LDS_BYTES, st, p.descs, p.tile_map);
    else if (p.tile == 128) {
        // row cursor: the caller provides one zeroed progress word per tile (64 per descriptor) when it wants the launch
        // paced; the tuning record picks epoch / slack (0: the defaults 4 / 2) or vetoes it (-1).  Its own kernel: the
        // unpaced launch carries none of it
        const int te = AEW_T().tn_cursor_epoch;
        const int e = te > 0 ? te : 4, d = te > 0 ? AEW_T().tn_cursor_slack : 2;
        const bool on = p.cursors && p.cursor_stride >= 64 && te >= 0 && e >= 3 && d >= 1;
        if (on)
            hipLaunchKernelGGL(k_gemm_tn_bf16_grp_cur, dim3(p.n_blocks), dim3(TN_THREADS), TN_LDS_BYTES, st, p.descs, p.tile_map,
                               p.cursors, p.cursor_stride, e, d);
        else if (AEW_T().tn_mfma32)
            hipLaunchKernelGGL(k_gemm_tn_bf16_grp32, dim3(p.n_blocks), dim3(TN_THREADS), TN_LDS_BYTES, st, p.descs, p.tile_map);
        else
            hipLaunchKernelGGL(k_gemm_tn_bf16_grp, dim3(p.n_blocks), dim3(TN_THREADS), TN_LDS_BYTES, st, p.descs, p.tile_map);
    }
    else
        return AEW_E_ARG;
    return (int)hipGetLastError();
}

static int launch_gemm_tn(const aew_gemm_tn_t& g, hipStream_t st) {
    if (g.n_segs < 1 || g.n_segs > AEW_MAX_SEGS || g.Mc <= 0 || g.batch <= 0 || !g.out) return AEW_E_ARG;
    const int es = g.dtype == AEW_BF16 ? 2 : 4;
    const int tile = g.dtype == AEW_BF16 ? TN_BT : TF_BT;
    const int rc = g.dtype == AEW_BF16 ? TN_RC : TF_RC;
    int ksum = 0;
    for (int s = 0; s < g.n_segs; ++s) {
        const int rcode = check_seg(g.seg[s], es, tile);
        if (rcode) return rcode;
        ksum += g.seg[s].k_len;
    }
    aew_seg_t gg = g.g; gg.k_len = tile;
    const int rcode = check_seg(gg, es, tile);
    if (rcode) return rcode;
    if (ksum != g.K_total || g.N_pad % tile) return AEW_E_ARG;
    int sp, rps, fold;
    tn_plan(g, tile, rc, &sp, &rps, &fold);
    if (g.impl == 1) {
        dim3 grid((g.K_total + 63) / 64, g.N_pad, fold ? sp : g.batch * sp);
        if (g.dtype == AEW_BF16) hipLaunchKernelGGL(k_gemm_tn_check<uint16_t>, grid, dim3(64), 0, st, g, sp, rps, fold);
        else hipLaunchKernelGGL(k_gemm_tn_check<float>, grid, dim3(64), 0, st, g, sp, rps, fold);
    } else if (g.dtype == AEW_BF16) {
        const int n_chunks = sp * (fold ? 1 : g.batch);
        const int rc = ensure_big_lds();
        if (rc) return rc;
        if (tn_use_big(g)) {
            const int tiles = ((g.N_pad / 128 + 1) / 2) * ((g.K_total / 128 + 1) / 2);
            dim3 gridb(((n_chunks + 7) / 8) * 8 * tiles);
            hipLaunchKernelGGL(k_gemm_tn_bf16_big, gridb, dim3(TNB_THREADS), TNB_LDS_BYTES, st, g, sp, rps, fold);
            return (int)hipGetLastError();
        }
        dim3 grid(((n_chunks + 7) / 8) * 8 * (g.N_pad / TN_BT) * (g.K_total / TN_BT));
        if (AEW_T().tn_safe) hipLaunchKernelGGL(k_gemm_tn_bf16<1>, grid, dim3(TN_THREADS), TN_LDS_BYTES, st, g, sp, rps, fold);
        else hipLaunchKernelGGL(k_gemm_tn_bf16<0>, grid, dim3(TN_THREADS), TN_LDS_BYTES, st, g, sp, rps, fold);
    } else {
        dim3 grid((g.N_pad / TF_BT) * (g.K_total / TF_BT), sp, fold ? 1 : g.batch);
        hipLaunchKernelGGL(k_gemm_tn_f32, grid, dim3(256), 2 * TF_STAGE_BYTES, st, g, sp, rps, fold);
    }
    return (int)hipGetLastError();
}
#endif  /* AEW_DEV_KERNELS_ONLY */
