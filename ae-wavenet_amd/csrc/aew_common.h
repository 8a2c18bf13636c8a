// aew_common.h — device helpers shared by the gfx950 kernels (CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/aewavenet.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define AEW_WAVE 64
#define AEW_LDS_PTR(p) ((void __attribute__((address_space(3)))*)(p))
#define AEW_GLB_PTR(p) ((const void __attribute__((address_space(1)))*)(p))

// 16 zero bytes: source of every masked (out-of-range) 16-byte LDS-DMA piece.
__device__ __attribute__((aligned(16))) unsigned int aew_zero_page[4] = {0u, 0u, 0u, 0u};

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
__device__ __forceinline__ void unpack4_bf16(uint2 r, float v[4]) {
    v[0] = __uint_as_float(r.x << 16);
    v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16);
    v[3] = __uint_as_float(r.y & 0xffff0000u);
}

// ---- view access on quads of 4 consecutive channels ---------------------------------------
__device__ __forceinline__ bool view_row(const aew_view_t& v, int m, int64_t& row) {
    row = (int64_t)m * v.row_step + v.row_off;
    return row >= v.row_lo && row < v.row_hi;
}
__device__ __forceinline__ void view_load4(const aew_view_t& v, int b, int m, int n, float out[4]) {
    int64_t row;
    if (!view_row(v, m, row)) { out[0] = out[1] = out[2] = out[3] = 0.f; return; }
    const int64_t idx = (int64_t)b * v.batch_stride + row * v.row_pitch + n;
    if (v.dtype == AEW_BF16) {
        unpack4_bf16(*reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(v.ptr) + idx), out);
    } else {
        const float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(v.ptr) + idx);
        out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w;
    }
}
__device__ __forceinline__ void view_store4(const aew_view_t& v, int b, int m, int n, const float val[4]) {
    int64_t row;
    if (!view_row(v, m, row)) return;
    const int64_t idx = (int64_t)b * v.batch_stride + row * v.row_pitch + n;
    if (v.dtype == AEW_BF16) {
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(v.ptr) + idx) = pack4_bf16(val);
    } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(v.ptr) + idx) = make_float4(val[0], val[1], val[2], val[3]);
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

// NT tiles: 128-byte rows (8 chunks).  chunk' = chunk ^ (row & 7): the 16 rows x 1 chunk of a
// ds_read_b128 MFMA fragment then land on 16 distinct 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int nt_swz(int row, int chunk) { return chunk ^ (row & 7); }

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
