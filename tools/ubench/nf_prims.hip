// Micro-benchmark of the primitives that bound the fp32 exact-chain kernel (k_gemm_nt_f32) on gfx950:
//   * a dependent chain of v_mfma_f32_16x16x4_f32 (one accumulator): cycles per MFMA with 1..3 waves per SIMD
//   * the same with RT independent chains per wave
//   * s_barrier in a loop: 4 waves per block, 1..3 blocks per CU
//   * ds_read_b32 x 16 + s_waitcnt per iteration (the fragment reads of one K tile)
// hipcc --offload-arch=gfx950 -O3 nf_prims.hip -o nf_prims
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int RT>
__global__ __launch_bounds__(256) void k_chain(float* out, int iters) {
    f32x4_t acc[RT];
    for (int r = 0; r < RT; ++r) acc[r] = (f32x4_t){0, 0, 0, 0};
    float a = (float)(threadIdx.x & 3), b = 1.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int r = 0; r < RT; ++r)
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[r]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = 0;
    for (int r = 0; r < RT; ++r) s += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_barrier(float* out, int iters) {
    extern __shared__ char smem[];
    float s = 0;
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + smem[0];
}

template <int WITH_BARRIER>
__global__ __launch_bounds__(256) void k_reads(float* out, int iters) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, fi = lane & 15, kq = lane >> 4, wave = threadIdx.x >> 6;
    const unsigned base = (unsigned)(uintptr_t)smem;   // LDS offset
    const unsigned xo = base + fi * 128 + ((fi & 7) << 4) + kq * 4, wo = base + 2048 + (wave * 16 + fi) * 128 + ((fi & 7) << 4) + kq * 4;
    float s = 0;
    for (int it = 0; it < iters; ++it) {
        float v[16];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            asm volatile("ds_read_b32 %0, %1" : "=v"(v[2 * ks]) : "v"(xo ^ (ks << 4)));
            asm volatile("ds_read_b32 %0, %1" : "=v"(v[2 * ks + 1]) : "v"(wo ^ (ks << 4)));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) asm volatile("" ::"v"(v[ks]));
        s += v[0];
        if (WITH_BARRIER) __builtin_amdgcn_s_barrier();
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static float timed(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    float* out;
    hipMalloc(&out, 4096 * 256 * 4);
    const double ghz = 2.4;
    const int iters = 2000;
    for (int bpc = 1; bpc <= 3; ++bpc) {
        const int blocks = 256 * bpc;
        float ms = timed([&] { hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(256), 0, 0, out, iters); });
        printf("dependent f32 16x16x4 chain, %d wave(s)/SIMD, 1 chain/wave : %7.1f cycles per MFMA of a chain (%.3f ms)\n", bpc,
               ms * 1e6 * ghz / (iters * 8.0), ms);
        ms = timed([&] { hipLaunchKernelGGL(k_chain<2>, dim3(blocks), dim3(256), 0, 0, out, iters); });
        printf("                             %d wave(s)/SIMD, 2 chains/wave: %7.1f cycles per MFMA of a chain\n", bpc,
               ms * 1e6 * ghz / (iters * 8.0));
    }
    for (int bpc = 1; bpc <= 3; ++bpc) {
        const int blocks = 256 * bpc;
        float ms = timed([&] { hipLaunchKernelGGL(k_barrier, dim3(blocks), dim3(256), 50 * 1024, 0, out, iters * 4); });
        printf("s_barrier loop, 4 waves/block, %d block(s)/CU: %7.1f cycles per barrier\n", bpc, ms * 1e6 * ghz / (iters * 4.0));
        ms = timed([&] { hipLaunchKernelGGL(k_reads<0>, dim3(blocks), dim3(256), 50 * 1024, 0, out, iters); });
        printf("16 x ds_read_b32 + wait, %d block(s)/CU        : %7.1f cycles per iteration\n", bpc, ms * 1e6 * ghz / iters);
        ms = timed([&] { hipLaunchKernelGGL(k_reads<1>, dim3(blocks), dim3(256), 50 * 1024, 0, out, iters); });
        printf("16 x ds_read_b32 + wait + barrier, %d blk/CU   : %7.1f cycles per iteration\n", bpc, ms * 1e6 * ghz / iters);
    }
    // launch cost of an empty-ish kernel of 480 blocks
    float ms = timed([&] { hipLaunchKernelGGL(k_barrier, dim3(480), dim3(256), 50 * 1024, 0, out, 1); });
    printf("480-block launch, 1 barrier: %.2f us per launch (back to back)\n", ms * 1e3);
    return 0;
}
