// Micro-benchmark: sustained v_mfma_f32_16x16x32_bf16 rate on gfx950 with the wave shapes the NT
// kernel uses (no memory traffic).  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int NACC, int THREADS>
__global__ __launch_bounds__(THREADS) void k_mfma(float* out, int iters) {
    f32x4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4_t){0, 0, 0, 0};
    bf16x8_t a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int threads>
static void run(const char* name, int blocks, int iters) {
    float* out;
    hipMalloc(&out, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mfma<NACC, threads>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_mfma<NACC, threads>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double n_mfma = (double)blocks * (threads / 64) * iters * NACC;
    const double tf = n_mfma * 16 * 16 * 32 * 2 / (ms * 1e-3) / 1e12;
    // MFMAs per SIMD = n_mfma / (256 CUs * 4)
    printf("%-34s %8.3f ms  %8.1f TFLOP/s   %6.2f ns per MFMA per SIMD\n", name, ms, tf, ms * 1e6 / (n_mfma / 1024.0));
    hipFree(out);
}

int main() {
    run<32, 256>("1 wave/SIMD  (256 blk x 256 thr) 32acc", 256, 4000);
    run<32, 512>("2 waves/SIMD (256 x 512) 32acc", 256, 4000);
    run<16, 512>("2 waves/SIMD (256 x 512) 16acc", 256, 4000);
    run<16, 512>("4 waves/SIMD (512 x 512) 16acc", 512, 4000);
    run<32, 256>("2 waves/SIMD, 2 blk/CU (512x256) 32acc", 512, 4000);
    run<32, 512>("short: 2 waves/SIMD 28x32 mfma", 256, 28);
    run<32, 512>("short: 2 waves/SIMD 56x32 mfma", 256, 56);
    run<16, 512>("short: 4 waves/SIMD 28x16 mfma (512 blk)", 512, 28);
    run<16, 512>("short: 4 waves/SIMD 28x16 mfma (896 blk)", 896, 28);
    run<32, 256>("short: 2w/SIMD 2blk/CU 28x32 (896 blk)", 896, 28);
    return 0;
}
