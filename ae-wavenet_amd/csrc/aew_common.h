// aew_common.h — device helpers shared by the gfx950 kernels (CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/aewavenet.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define AEW_WAVE 64
#define AEW_LDS_PTR(p) ((void __attribute__((address_space(3)))*)(p))
#define AEW_GLB_PTR(p) ((const void __attribute__((address_space(1)))*)(p))

// 16 zero bytes: source of every masked (out-of-range) 16-byte LDS-DMA piece.
__device__ __attribute__((aligned(16))) unsigned int aew_zero_page[4] = {0u, 0u, 0u, 0u};
// 32 KiB of zeros: masked rows of the K-streaming kernels point here and ADVANCE like real rows
// (so they need no per-row increment register); a segment is at most AEW_ZERO_SPAN bytes long.
#define AEW_ZERO_SPAN 16384
__device__ __attribute__((aligned(128))) unsigned int aew_zero_region[2 * AEW_ZERO_SPAN / 4];

// ---- tuning context (aew_tuning_t, aewavenet.h): the process-wide record the aew_set_* switches edit, and the record of
// the call in progress when a caller passed its own (aew_run_plan_tuned): launchers read AEW_T().field
// (field order of aew_tuning_t; the ONE place the library's defaults are written down: aew_tuning_default returns the same)
#define AEW_TUNING_DEFAULTS {64, 1, 1, 128, 256, 1, 256, 64, 0, 0, 1, 256, 1, 16, 0, 0, 256, 4096, 512, 8, 128, 0, 0, 0, 1, 1, 0, {0, 0, 0, 0, 0}}
static aew_tuning_t g_tune = AEW_TUNING_DEFAULTS;
static thread_local const aew_tuning_t* t_tune = nullptr;
static inline const aew_tuning_t& AEW_T() { return t_tune ? *t_tune : g_tune; }

// ---- bf16 <-> f32, round-to-nearest-even (matches torch .to(bfloat16)) --------------------
__device__ __forceinline__ uint16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

__device__ __forceinline__ uint2 pack4_bf16(const float v[4]) {
    uint2 r;
    r.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    r.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    return r;
}
// ---- view access: W (4 or 8) consecutive channels of one row ---------------------------------
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    // v_cvt_pk_bf16_f32 (round-to-nearest-even, same as torch .to(bfloat16))
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t));
}

// byte pointer to channel 0 of the view row that GEMM row m maps to; nullptr if the row does not
// exist (loads then read zero, stores are dropped)
// (branch-free on purpose: the view record sits in kernarg memory; a conditional early-out makes the
// compiler fetch its fields in several dependent scalar-load round trips instead of one batch)
__device__ __forceinline__ char* view_rowptr(const aew_view_t& vref, int b, int m) {
    const aew_view_t v = vref;
    const int64_t row = (int64_t)m * v.row_step + v.row_off;
    char* q = reinterpret_cast<char*>(v.ptr) +
              ((int64_t)b * v.batch_stride + row * v.row_pitch) * (v.dtype == AEW_BF16 ? 2 : 4);
    return (v.ptr && row >= v.row_lo && row < v.row_hi) ? q : nullptr;
}

template <int W>
__device__ __forceinline__ void row_load(const char* rp, int dtype, int n, float out[W]) {
    if (!rp) {
#pragma unroll
        for (int r = 0; r < W; ++r) out[r] = 0.f;
        return;
    }
    if (dtype == AEW_BF16) {
        uint32_t w[W / 2];
        if (W == 8) { const uint4 t = *reinterpret_cast<const uint4*>(rp + n * 2); w[0] = t.x; w[1] = t.y; w[W / 2 - 2] = t.z; w[W / 2 - 1] = t.w; }
        else { const uint2 t = *reinterpret_cast<const uint2*>(rp + n * 2); w[0] = t.x; w[1] = t.y; }
#pragma unroll
        for (int r = 0; r < W / 2; ++r) {
            out[2 * r] = __uint_as_float(w[r] << 16);
            out[2 * r + 1] = __uint_as_float(w[r] & 0xffff0000u);
        }
    } else {
#pragma unroll
        for (int q = 0; q < W / 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(rp + (n + 4 * q) * 4);
            out[4 * q] = t.x; out[4 * q + 1] = t.y; out[4 * q + 2] = t.z; out[4 * q + 3] = t.w;
        }
    }
}

// 16-byte WRITE-THROUGH store (sc1): the bytes leave the XCD's L2 for memory, where a workgroup on any XCD reads them
// once this wave's `s_waitcnt vmcnt(0)` has passed - the payload store of an in-launch hand-off (chained NT launches;
// MI355X_MICROARCH.md, inter-workgroup visibility).  Inline asm: there is no builtin for a scoped 16-byte global store;
// the s_nop covers the ">8-byte store data" hazard the compiler pads only for its own stores.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
__device__ __forceinline__ void store16_wt(void* p, u32x4_t v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}

// WT: write-through stores (bf16 rows of 8 channels and fp32 rows; the other forms are not produced by a chained stage)
template <int W, bool WT = false>
__device__ __forceinline__ void row_store(char* rp, int dtype, int n, const float v[W]) {
    if (!rp) return;
    if (dtype == AEW_BF16) {
        if (W == 8 && WT) {
            store16_wt(rp + n * 2, (u32x4_t){pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]),
                                             pack2_bf16(v[W - 4], v[W - 3]), pack2_bf16(v[W - 2], v[W - 1])});
        } else if (W == 8) {
            *reinterpret_cast<uint4*>(rp + n * 2) = make_uint4(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]),
                                                               pack2_bf16(v[W - 4], v[W - 3]), pack2_bf16(v[W - 2], v[W - 1]));
        } else {
            *reinterpret_cast<uint2*>(rp + n * 2) = make_uint2(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]));
        }
    } else {
#pragma unroll
        for (int q = 0; q < W / 4; ++q) {
            if (WT) store16_wt(rp + (n + 4 * q) * 4, (u32x4_t){__float_as_uint(v[4 * q]), __float_as_uint(v[4 * q + 1]),
                                                               __float_as_uint(v[4 * q + 2]), __float_as_uint(v[4 * q + 3])});
            else *reinterpret_cast<float4*>(rp + (n + 4 * q) * 4) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
    }
}

// ---- activation functions (fp32; outputs are rounded to bf16 by the caller) ----------------
// v_exp_f32 + v_rcp_f32 (1 ulp): results are rounded to bf16 (8-bit mantissa) by the caller, so
// the IEEE division sequence (~10 instructions) would buy nothing.
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) {
    // tanh(x) = 1 - 2/(exp(2x)+1); saturates cleanly for |x| large
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// ---- LDS tile geometry ----------------------------------------------------------------------
// All GEMM tiles are staged as rows of ROWB bytes made of 16-byte chunks.  A wave-level
// LDS-DMA (`global_load_lds_dwordx4`) writes 64 lanes x 16 B = 1 KiB *linearly* at a
// wave-uniform LDS address, so bank-conflict swizzles are applied on the per-lane SOURCE
// address and mirrored on the fragment read (cdna_hip_programming.md §5.4 rule 21).

// NT tiles, fp32 kernel: 128-byte rows (8 chunks).  chunk' = chunk ^ (row & 7).
__device__ __forceinline__ int nt_swz(int row, int chunk) { return chunk ^ (row & 7); }
// NT tiles, bf16 kernel: 64-byte rows (4 chunks = 32 bf16 = one MFMA K step).  A 256-byte bank
// row holds 4 LDS rows.  A ds_read_b128 is served in lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...
// (MI355X_MICROARCH.md, LDS table): a group mixes fragment rows fi = 0-3,12-15 at chunk c with rows 4-11 at
// chunk c^1, so with g(a) the XOR of row block a = row>>2, {g(a0), g(a0+3), 1^g(a0+1), 1^g(a0+2)} must be distinct.
// g = (0,2,0,2) = 2 * (a & 1) satisfies that for EVERY first row of the 16-row fragment that the kernels use: aligned
// (a multiple of 16: k_gemm_nt_bf16) and shifted by a dilation 1, 2, 4, 8 (the second tap of k_gemm_nt_bf16_win, which
// reads rows r + d of one staged window; found by enumeration over g = f(row>>2) ^ h(row&3)).  History: g = a was
// 2-way conflicted on every read (SQ_LDS_BANK_CONFLICT = SQ_LDS_IDX_ACTIVE / 2, profiles/r02_notes.md); g = -a & 3 was
// conflict-free aligned but 2-way conflicted for shifts 1, 2, 4 (11-13 % of the window kernel's LDS cycles).
__device__ __forceinline__ int nt_swz64(int row, int chunk) { return chunk ^ ((row >> 1) & 2); }

// TN tiles: 256-byte rows (16 chunks); fragments are read with ds_read_b64_tr_b16 (bf16) whose
// 16-lane group touches 4 rows x 32 B.  XOR the 32-byte group index with
// f(row) = (row&3) | ((row>>1)&4) so the 8 (row) pieces of a 32-lane half hit 8 distinct slots.
__device__ __forceinline__ int tn_swz_bf16(int row, int chunk) {
    return chunk ^ ((((row & 3) | ((row >> 1) & 4))) << 1);
}
// f32 TN fragments are plain ds_read_b32 of 16 consecutive floats (4 chunks) on rows r..r+3.
__device__ __forceinline__ int tn_swz_f32(int row, int chunk) { return chunk ^ ((row & 3) << 2); }

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    // 16 B per lane; LDS destination = lds_wave_base + lane*16 (wave-uniform base)
    __builtin_amdgcn_global_load_lds(AEW_GLB_PTR(gsrc), AEW_LDS_PTR(lds_wave_base), 16, 0, 0);
}

// glds16 with device-scope (sc1) reads: never served from this CU's L1 - the operand load of a chained launch's consumer
// stage, whose producer stored the rows write-through (store16_wt) and raised its counter after they had drained
// (cdna_hip_programming.md Guideline 16: sc1 loads may stand in for the acquire when the producer stored sc1).
__device__ __forceinline__ void glds16_sc1(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(AEW_GLB_PTR(gsrc), AEW_LDS_PTR(lds_wave_base), 16, 0, 16);
}
template <bool SC1>
__device__ __forceinline__ void glds16_x(const void* gsrc, void* lds_wave_base) {
    if (SC1) glds16_sc1(gsrc, lds_wave_base);
    else glds16(gsrc, lds_wave_base);
}
// 16-byte sc1 load through a raw buffer resource (compiler-tracked, unlike inline asm): base + off; an offset beyond the
// resource's range (AEW_BUF_OOB) reads zeros - the masked rows of an epilogue operand need no zero page
#define AEW_BUF_OOB 0xfffffff0u
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ uint4 ld16_sc1(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 16);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Same LDS-DMA as glds16 but opaque to the compiler.  Needed where the LDS tile is read back with
// ds_read_b64_tr_b16: that builtin carries no alias information, so after a builtin LDS-DMA the
// compiler inserts s_waitcnt vmcnt(0) in front of it, i.e. it drains the prefetch it was just given
// (seen in the ISA of k_gemm_tn_bf16; profiles/r02_notes.md).  Callers order it by hand with counted
// vmcnt waits + barriers.  lds_off = byte offset in LDS (wave-uniform).
__device__ __forceinline__ void glds16_raw(const void* gsrc, uint32_t lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(gsrc), "s"(lds_off) : "memory", "m0");
}

// one dword per lane (all lanes may pass the same address): LDS[lds_off + 4 * lane]
__device__ __forceinline__ void glds4_raw(const void* gsrc, uint32_t lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off sc1"
                 :: "v"(gsrc), "s"(lds_off) : "memory", "m0");
}

__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- wave reductions -------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
