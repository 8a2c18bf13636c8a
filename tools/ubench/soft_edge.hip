// What does a cross-stream dependency cost, as a HIP event (hipEventRecord + hipStreamWaitEvent) and as a flag in device
// memory (a one-thread "signal" kernel on the producer's stream, a one-wave "wait" kernel that spins on the flag on the
// consumer's stream)?  Three 256-block kernels of ~40 us each, K2 on a second stream between K1 and K3 of the first:
//   serial       A: K1 K2 K3
//   events       A: K1 rec(e1) | B: wait(e1) K2 rec(e2) | A: wait(e2) K3
//   flags        A: K1 sig(f1) waitk(f2) K3 | B: waitk(f1) K2 sig(f2)
// The difference to `serial` is what one fork + one join costs.
//   hipcc --offload-arch=gfx950 -O3 soft_edge.hip -o soft_edge && ./soft_edge
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long cycles, int* sink) {
    const long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (cycles < 0) sink[0] = 1;
}
// flags are monotonic counters: signal adds one, the waiter counts its own visits and waits until the flag has caught up
__global__ void k_signal(unsigned* flag) { __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_wait(unsigned* flag, unsigned* visits, unsigned* timeouts) {
    const unsigned want = ++visits[0];
    long spins = 0;
    while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > 20000000L) { timeouts[0] += 1; break; }
    }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    int* sink; CK(hipMalloc(&sink, 4));
    unsigned* f; CK(hipMalloc(&f, 64)); CK(hipMemset(f, 0, 64));       // f[0], f[1] flags; f[2], f[3] visit counters; f[4] timeouts
    hipStream_t A, B; CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    hipEvent_t t0, t1, e1, e2; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    const long CYC = 100000;
    auto K = [&](hipStream_t s) { hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, CYC, sink); };
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f, sum = 0;
        const int REP = 20;
        for (int r = 0; r < REP + 3; ++r) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, A));
            if (mode == 0) { K(A); K(A); K(A); }
            else if (mode == 1) {
                K(A); CK(hipEventRecord(e1, A)); CK(hipStreamWaitEvent(B, e1, 0)); K(B); CK(hipEventRecord(e2, B));
                CK(hipStreamWaitEvent(A, e2, 0)); K(A);
            } else {
                K(A); hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, A, f);
                hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, B, f, f + 2, f + 4); K(B);
                hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, B, f + 1);
                hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, A, f + 1, f + 3, f + 4); K(A);
            }
            CK(hipEventRecord(t1, A));
            CK(hipEventSynchronize(t1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, t0, t1));
            if (r >= 3) { best = ms < best ? ms : best; sum += ms; }
        }
        unsigned h[5]; CK(hipMemcpy(h, f, 20, hipMemcpyDeviceToHost));
        printf("%-8s three kernels take %.1f us (mean %.1f)   timeouts %u\n", mode == 0 ? "serial" : mode == 1 ? "events" : "flags",
               best * 1e3, sum / REP * 1e3, h[4]);
    }
    return 0;
}
