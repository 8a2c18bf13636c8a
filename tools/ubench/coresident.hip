// Micro-benchmark: does a small-footprint kernel on a second stream get CU slots while a kernel that holds
// 3 x 48 KiB of LDS on every CU (the grouped weight-gradient launch: 768 blocks of 256 threads) is running?
// A spins for ~`ms` milliseconds per block; B (blocks of `bthreads` threads, `blds` bytes of LDS, `bvgpr`-ish
// registers) is launched right after it on another stream.  Reported: when B finished relative to A's launch.
//   hipcc --offload-arch=gfx950 -O3 coresident.hip -o coresident
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_hog(long long cycles, float* sink) {
    extern __shared__ char smem[];
    float acc[96];                                             // ~100 VGPRs like the TN kernel
#pragma unroll
    for (int i = 0; i < 96; ++i) acc[i] = threadIdx.x * 0.001f + i;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {
#pragma unroll
        for (int i = 0; i < 96; ++i) acc[i] = acc[i] * 1.0001f + 0.5f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 96; ++i) s += acc[i];
    if (s == 1234.5f) { smem[threadIdx.x] = 1; sink[0] = s + smem[0]; }
}

template <int NACC>
__global__ void k_small(long long cycles, float* sink) {
    extern __shared__ char smem[];
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 0.001f + i;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = acc[i] * 1.0001f + 0.5f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    if (s == 1234.5f) { smem[threadIdx.x] = 1; sink[0] = s + smem[0]; }
}

template <int NACC>
static void run(const char* name, int hog_lds, int bblocks, int bthreads, int blds) {
    float* sink; hipMalloc(&sink, 64);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_hog), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_small<NACC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, ea, eb; hipEventCreate(&e0); hipEventCreate(&ea); hipEventCreate(&eb);
    const long long ms_cycles = 100000;                        // wall_clock64 runs at 100 MHz: 1 ms
    for (int rep = 0; rep < 2; ++rep) {
        hipDeviceSynchronize();
        hipEventRecord(e0, s1);
        hipLaunchKernelGGL(k_hog, dim3(768), dim3(256), hog_lds, s1, ms_cycles, sink);
        hipEventRecord(ea, s1);
        hipStreamWaitEvent(s2, e0, 0);
        hipLaunchKernelGGL(k_small<NACC>, dim3(bblocks), dim3(bthreads), blds, s2, ms_cycles / 50, sink);   // 20 us of work
        hipEventRecord(eb, s2);
        hipDeviceSynchronize();
    }
    float ta, tb; hipEventElapsedTime(&ta, e0, ea); hipEventElapsedTime(&tb, e0, eb);
    printf("%-72s hog done %.3f ms, small done %.3f ms  %s\n", name, ta, tb, tb < 0.5f * ta ? "CO-RESIDENT" : "waited");
    hipFree(sink);
}

// a CHAIN of n small kernels on the second stream (each ~10 us of work, stream-ordered) next to the hog: when does each
// one start / end relative to the hog's start (device clock, 100 MHz)?
__global__ void k_stamp(long long cycles, long long* stamps, int idx) {
    extern __shared__ char smem[];
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * idx] = t0;
    while (wall_clock64() - t0 < cycles) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * idx + 1] = wall_clock64();
    if (cycles < 0) smem[threadIdx.x] = 1;
}
__global__ __launch_bounds__(256) void k_hog_stamp(long long cycles, long long* stamps) {
    extern __shared__ char smem[];
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = t0;
    while (wall_clock64() - t0 < cycles) {}
    if (cycles < 0) smem[threadIdx.x] = 1;
}
static void chain(const char* name, int hog_lds, int n, int bblocks, int bthreads, int blds, bool with_hog) {
    long long* st; hipMalloc(&st, 8 * 2 * (n + 1)); hipMemset(st, 0, 8 * 2 * (n + 1));
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_hog_stamp), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0; hipEventCreate(&e0);
    for (int rep = 0; rep < 2; ++rep) {
        hipDeviceSynchronize();
        hipEventRecord(e0, s1);
        if (with_hog) hipLaunchKernelGGL(k_hog_stamp, dim3(768), dim3(256), hog_lds, s1, 100000LL, st);
        hipStreamWaitEvent(s2, e0, 0);
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_stamp, dim3(bblocks), dim3(bthreads), blds, s2, 1000LL, st, i + 1);
        hipDeviceSynchronize();
    }
    long long h[2 * 65];
    hipMemcpy(h, st, 8 * 2 * (n + 1), hipMemcpyDeviceToHost);
    const long long t0 = with_hog ? h[0] : h[2];
    printf("%-60s starts (us after %s):", name, with_hog ? "hog start" : "first");
    for (int i = 1; i <= n; ++i) printf(" %.0f", (h[2 * i] - t0) / 100.0);
    printf("   last end %.0f\n", (h[2 * n + 1] - t0) / 100.0);
    hipFree(st);
}

int main() {
    chain("chain of 12 x (48 blk x 256 thr, 8 KiB), no hog", 0, 12, 48, 256, 8 * 1024, false);
    chain("chain of 12 x (48 blk x 256 thr, 8 KiB) next to the hog", 48 * 1024, 12, 48, 256, 8 * 1024, true);
    chain("chain of 12 x (48 blk x 256 thr, 0 LDS) next to the hog", 48 * 1024, 12, 48, 256, 0, true);
    chain("chain of 12 x (48 blk x 256 thr, 24 KiB) next to a 45 KiB hog", 45 * 1024, 12, 48, 256, 24 * 1024, true);
    run<32>("hog 3 x 48 KiB | small 48 x 64 thr, 16 KiB LDS, ~40 VGPR", 48 * 1024, 48, 64, 16 * 1024);
    run<32>("hog 3 x 48 KiB | small 48 x 256 thr, 16 KiB LDS, ~40 VGPR", 48 * 1024, 48, 256, 16 * 1024);
    run<32>("hog 3 x 48 KiB | small 48 x 256 thr, 8 KiB LDS", 48 * 1024, 48, 256, 8 * 1024);
    run<32>("hog 3 x 48 KiB | small 48 x 256 thr, 0 LDS", 48 * 1024, 48, 256, 0);
    run<32>("hog 3 x 48 KiB | small 48 x 256 thr, 24 KiB LDS (does not fit)", 48 * 1024, 48, 256, 24 * 1024);
    run<96>("hog 3 x 48 KiB | small 48 x 256 thr, 16 KiB LDS, ~100 VGPR", 48 * 1024, 48, 256, 16 * 1024);
    run<96>("hog 3 x 48 KiB | small 256 x 512 thr, 16 KiB LDS, ~100 VGPR", 48 * 1024, 256, 512, 16 * 1024);
    run<32>("hog 3 x 45 KiB | small 48 x 256 thr, 24 KiB LDS", 45 * 1024, 48, 256, 24 * 1024);
    return 0;
}
