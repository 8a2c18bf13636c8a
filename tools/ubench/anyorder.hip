// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950?  (hip_ext.h says "not supported on GFX9xx".)
// Two launches of a kernel that occupies 64 CUs for ~50 us: in order 2 x t, overlapped ~t.
//   hipcc --offload-arch=gfx950 -O3 anyorder.hip -o anyorder && ./anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long cycles, int* sink) {
    const long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (cycles < 0) sink[0] = 1;
}
int main() {
    int* sink; (void)hipMalloc(&sink, 4);
    hipStream_t st; (void)hipStreamCreate(&st);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int flags = 0; flags < 2; ++flags) {
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            (void)hipEventRecord(e0, st);
            hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, nullptr, nullptr, 0u, 100000L, sink);
            hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, nullptr, nullptr, (unsigned)flags, 100000L, sink);
            (void)hipEventRecord(e1, st);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("second launch flags = %d: two launches take %.1f us\n", flags, best * 1e3);
    }
    return 0;
}
