// Micro-benchmark: cost of workgroup-to-workgroup signalling on gfx950 (agent-scope atomics through L2 / fabric),
// the number that sizes the layer-pipelined sampler (DESIGN.md "sampler").  One workgroup per CU (96 KiB LDS each).
//   ping-pong   : A releases a flag (+ optional 1 KiB payload), B acquires, answers; time per round trip
//   group sync  : G workgroups add to a counter and spin until everyone arrived; time per barrier
// Workgroup L runs on XCD L % 8 (dispatch rule the GEMM kernels also rely on), so (0, 8) share an L2, (0, 1) do not.
// Every spin is bounded: a lost partner sets `abort` instead of hanging the GPU.
//   hipcc --offload-arch=gfx950 -O3 flag_latency.hip -o flag_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define SPIN_MAX (1 << 22)

__device__ __forceinline__ unsigned ld_acq(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_rel(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool wait_ge(const unsigned* p, unsigned v, unsigned* abort_flag) {
    for (int s = 0; s < SPIN_MAX; ++s) {
        if (ld_acq(p) >= v) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

// flags[0] = a->b, flags[32] = b->a (separate cache lines), flags[64] = abort; data = 2 x 1 KiB payload
__global__ __launch_bounds__(64) void k_pingpong(unsigned* flags, uint4* data, int a, int b, int iters, int payload,
                                                  unsigned* sink) {
    extern __shared__ char lds[];
    const int L = blockIdx.x, lane = threadIdx.x;
    if (L != a && L != b) return;
    unsigned acc = 0;
    for (int i = 1; i <= iters; ++i) {
        if (L == a) {
            if (payload) data[lane] = make_uint4(i, lane, i, lane);
            if (lane == 0) st_rel(flags, (unsigned)i);          // wave-level: the release orders the wave's stores
            bool ok = true;
            if (lane == 0) ok = wait_ge(flags + 32, (unsigned)i, flags + 64);
            ok = __shfl(ok, 0);
            if (!ok) break;
            if (payload) acc += data[64 + lane].x;
        } else {
            bool ok = true;
            if (lane == 0) ok = wait_ge(flags, (unsigned)i, flags + 64);
            ok = __shfl(ok, 0);
            if (!ok) break;
            if (payload) {
                acc += data[lane].x;
                data[64 + lane] = make_uint4(i, lane, 0, 0);
            }
            if (lane == 0) st_rel(flags + 32, (unsigned)i);
        }
    }
    if (payload) sink[L * 64 + lane] = acc;
}

// members: L % 8 == xcd (or every block when xcd < 0) and L / 8 < per_xcd
__global__ __launch_bounds__(64) void k_groupsync(unsigned* flags, int xcd, int per_xcd, int n_members, int iters) {
    extern __shared__ char lds[];
    const int L = blockIdx.x, lane = threadIdx.x;
    const bool member = (xcd < 0 || (L & 7) == xcd) && (L >> 3) < per_xcd;
    if (!member) return;
    for (int i = 1; i <= iters; ++i) {
        bool ok = true;
        if (lane == 0) {
            __hip_atomic_fetch_add(flags, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            ok = wait_ge(flags, (unsigned)(i * n_members), flags + 64);
        }
        ok = __shfl(ok, 0);
        if (!ok) break;
    }
}

int main() {
    const int LDS = 96 * 1024, GRID = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_pingpong), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_groupsync), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    unsigned *flags, *sink;
    uint4* data;
    hipMalloc(&flags, 4096); hipMalloc(&sink, GRID * 64 * 4); hipMalloc(&data, 128 * 16);
    const int iters = 20000;
    struct { int a, b; const char* what; } pairs[] = {{0, 8, "same XCD"}, {0, 1, "other XCD"}, {0, 4, "other XCD (4)"}};
    for (auto& p : pairs)
        for (int payload = 0; payload < 2; ++payload) {
            hipMemset(flags, 0, 4096);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_pingpong, dim3(GRID), dim3(64), LDS, 0, flags, data, p.a, p.b, iters, payload, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned h[65]; hipMemcpy(h, flags, sizeof(h), hipMemcpyDeviceToHost);
            printf("ping-pong %-14s payload %4d B : %7.3f us per round trip (one way %.3f)%s\n", p.what,
                   payload ? 1024 : 0, ms * 1e3 / iters, ms * 1e3 / iters / 2, h[64] ? "  ABORTED" : "");
        }
    struct { int xcd, per, n; const char* what; } groups[] = {{0, 4, 4, "4 WGs, one XCD"}, {0, 10, 10, "10 WGs, one XCD"},
                                                              {0, 32, 32, "32 WGs, one XCD"}, {-1, 2, 16, "16 WGs, 8 XCDs"},
                                                              {-1, 32, 256, "256 WGs (grid)"}};
    for (auto& g : groups) {
        hipMemset(flags, 0, 4096);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_groupsync, dim3(GRID), dim3(64), LDS, 0, flags, g.xcd, g.per, g.n, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned h[65]; hipMemcpy(h, flags, sizeof(h), hipMemcpyDeviceToHost);
        printf("group sync %-18s : %7.3f us per barrier%s\n", g.what, ms * 1e3 / iters, h[64] ? "  ABORTED" : "");
    }
    return 0;
}
