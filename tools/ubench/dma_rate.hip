// Micro-benchmark: how fast can a CU pull GEMM-shaped operand tiles through the vector memory path?
// Every block streams an "X-like" operand (rows private to a group of `xshare` blocks, like the activation
// tile that the N-tile blocks of one row-tile share) and a "W-like" operand (the same rows for every block,
// L2 resident) in K steps, as LDS-DMA (global_load_lds_dwordx4, mode 0) or as plain global_load_dwordx4 into
// VGPRs (mode 1).  No MFMA, no LDS reads: the number is the ceiling of the staging path alone.
//   hipcc --offload-arch=gfx950 -O3 dma_rate.hip -o dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define LDS_PTR(p) ((void __attribute__((address_space(3)))*)(p))
#define GLB_PTR(p) ((const void __attribute__((address_space(1)))*)(p))

struct Args {
    const char* x; const char* w;
    long x_pitch, w_pitch;        // bytes per row
    int rowb;                     // contiguous bytes per row and K step (64 | 128 | 256 | 512 | 1024)
    int xrows, wrows;             // tile rows per block (X part / W part)
    int iters;                    // K steps
    int xshare;                   // blocks sharing one X tile
    int depth;                    // K steps in flight
    int barrier;                  // 1: s_barrier per K step
};

template <int WAVES, int MODE, int PPW>
__global__ __launch_bounds__(WAVES * 64) void k_dma(Args a, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lpr = a.rowb / 16;                        // lanes per row
    const int rpp = 64 / lpr;                           // rows per 1 KiB piece
    const int xp = a.xrows / rpp, wp = a.wrows / rpp;   // pieces per K step
    const int np = xp + wp;
    constexpr int ppw = PPW;                            // pieces per wave per K step
    const long xg = (long)(blockIdx.x / a.xshare) * a.xrows;
    const char* src[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int p = wave * ppw + j;
        const int r = (p < xp ? p : p - xp) * rpp + lane / lpr;
        if (p < xp) src[j] = a.x + (xg + r) * a.x_pitch + (lane % lpr) * 16;
        else if (p < np) src[j] = a.w + (long)r * a.w_pitch + (lane % lpr) * 16;
        else src[j] = a.w + (lane % lpr) * 16;
    }
    const int stage_bytes = np * 1024;
    float acc = 0.f;
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t v[PPW];
    int slot = 0;
    for (int t = 0; t < a.iters; ++t) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            if (MODE == 0) {
                __builtin_amdgcn_global_load_lds(GLB_PTR(src[j]), LDS_PTR(smem + slot * stage_bytes + (wave * ppw + j) * 1024), 16, 0, 0);
            } else {
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[j]) : "v"(src[j]) : "memory");
            }
            src[j] += a.rowb;
        }
        slot = (slot + 1 == a.depth) ? 0 : slot + 1;
        // leave depth-1 steps in flight
        if (a.depth == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (a.depth == 2) { if (ppw == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else if (ppw == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else if (ppw == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else if (ppw == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else if (ppw == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else if (a.depth == 3) { if (ppw == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else if (ppw == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else if (ppw == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else if (ppw == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else if (ppw == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
        else { if (ppw == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else if (ppw == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else if (ppw == 3) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else if (ppw == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else if (ppw == 6) asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); }
        if (a.barrier) __builtin_amdgcn_s_barrier();
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < PPW; ++j) asm volatile("" :: "v"(v[j]));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) acc += __uint_as_float(v[j][0]);
    }
    if (acc == 12345.f) sink[0] = acc;
}

template <int WAVES, int MODE, int PPW>
static void run_p(const char* name, int blocks, Args a) {
    const int lpr = a.rowb / 16, rpp = 64 / lpr;
    const int np = a.xrows / rpp + a.wrows / rpp;
    const int ppw = (np + WAVES - 1) / WAVES;
    const size_t lds = (size_t)a.depth * np * 1024;
    float* sink;
    hipMalloc(&sink, 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_dma<WAVES, MODE, PPW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_dma<WAVES, MODE, PPW>), dim3(blocks), dim3(WAVES * 64), lds, 0, a, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_dma<WAVES, MODE, PPW>), dim3(blocks), dim3(WAVES * 64), lds, 0, a, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double bytes = (double)blocks * WAVES * ppw * a.iters * 1024.0;
    const double tbs = bytes / (ms * 1e-3) / 1e12;
    printf("%-58s %7.3f ms  %6.2f TB/s  %6.1f B/clk/CU@2.4GHz  lds=%zuK ppw=%d %s\n", name, ms, tbs,
           bytes / (ms * 1e-3) / 2.4e9 / 256.0, lds / 1024, ppw, hipGetErrorString(hipGetLastError()));
    hipFree(sink);
}

template <int WAVES, int MODE>
static void run(const char* name, int blocks, Args a) {
    const int lpr = a.rowb / 16, rpp = 64 / lpr;
    const int np = a.xrows / rpp + a.wrows / rpp;
    const int ppw = (np + WAVES - 1) / WAVES;
    switch (ppw) {
        case 1: run_p<WAVES, MODE, 1>(name, blocks, a); break;
        case 2: run_p<WAVES, MODE, 2>(name, blocks, a); break;
        case 3: run_p<WAVES, MODE, 3>(name, blocks, a); break;
        case 4: run_p<WAVES, MODE, 4>(name, blocks, a); break;
        case 6: run_p<WAVES, MODE, 6>(name, blocks, a); break;
        case 8: run_p<WAVES, MODE, 8>(name, blocks, a); break;
        default: printf("%s: unsupported ppw %d\n", name, ppw);
    }
}

int main(int argc, char** argv) {
    const size_t XB = (size_t)2 << 30, WB = (size_t)64 << 20;
    char *x, *w;
    hipMalloc(&x, XB); hipMalloc(&w, WB);
    hipMemset(x, 1, XB); hipMemset(w, 1, WB);
    hipDeviceSynchronize();
    auto mk = [&](int rowb, int xrows, int wrows, int iters, int xshare, int depth, int barrier) {
        Args a; a.x = x; a.w = w; a.rowb = rowb; a.xrows = xrows; a.wrows = wrows; a.iters = iters;
        a.xshare = xshare; a.depth = depth; a.barrier = barrier;
        a.x_pitch = (long)iters * rowb; a.w_pitch = (long)iters * rowb;
        return a;
    };
    printf("== the current NT shape: 256x128 tile, 64-byte rows (K step 32), 8 waves, 2 blocks/CU, X shared by 4 blocks\n");
    run<8, 0>("dma 256+128 rows x64B, depth 3, barrier, 512 blk", 512, mk(64, 256, 128, 128, 4, 3, 1));
    run<8, 0>("dma 256+128 rows x64B, depth 3, no barrier, 512 blk", 512, mk(64, 256, 128, 128, 4, 3, 0));
    run<8, 0>("dma 256+128 rows x64B, depth 4, no barrier, 512 blk", 512, mk(64, 256, 128, 128, 4, 4, 0));
    run<8, 1>("vgpr 256+128 rows x64B, depth 3, no barrier, 512 blk", 512, mk(64, 256, 128, 128, 4, 3, 0));
    printf("== row bytes (K step) sweep, 256+128 rows... same bytes per step = rows scale down\n");
    run<8, 0>("dma 128+64 rows x128B, depth 3, 512 blk", 512, mk(128, 128, 64, 128, 4, 3, 0));
    run<8, 0>("dma 64+32 rows x256B, depth 3, 512 blk", 512, mk(256, 64, 32, 128, 4, 3, 0));
    run<8, 0>("dma 16+8 rows x1024B, depth 3, 512 blk", 512, mk(1024, 16, 8, 128, 4, 3, 0));
    run<8, 0>("dma 256+128 rows x128B (K step 64), depth 2, 256 blk", 256, mk(128, 256, 128, 64, 4, 2, 0));
    run<8, 0>("dma 256+256 rows x128B (K step 64), depth 2, 256 blk", 256, mk(128, 256, 256, 64, 2, 2, 0));
    run<8, 0>("dma 256+256 rows x64B (K step 32), depth 4, 256 blk", 256, mk(64, 256, 256, 128, 2, 4, 0));
    run<8, 0>("dma 256+256 rows x64B (K step 32), depth 4, 256 blk, barrier", 256, mk(64, 256, 256, 128, 2, 4, 1));
    run<16, 0>("dma 256+256 rows x64B, 16 waves, depth 4, 256 blk", 256, mk(64, 256, 256, 128, 2, 4, 0));
    run<4, 0>("dma 256+256 rows x64B, 4 waves, depth 4, 256 blk", 256, mk(64, 256, 256, 128, 2, 4, 0));
    printf("== all-W (fully L2 resident, shared by everybody) vs all-X private (HBM stream)\n");
    run<8, 0>("dma 0+384 rows W only x64B, depth 3, 512 blk", 512, mk(64, 0, 384, 128, 1, 3, 0));
    run<8, 0>("dma 384+0 rows X private x64B, depth 3, 512 blk", 512, mk(64, 384, 0, 128, 1, 3, 0));
    run<8, 0>("dma 384+0 rows X shared by 4 x64B, depth 3, 512 blk", 512, mk(64, 384, 0, 128, 4, 3, 0));
    run<8, 0>("dma 0+384 rows W only x128B, depth 3, 512 blk", 512, mk(128, 0, 384, 64, 1, 3, 0));
    run<8, 1>("vgpr 0+384 rows W only x64B, depth 3, 512 blk", 512, mk(64, 0, 384, 128, 1, 3, 0));
    run<8, 0>("dma 0+96 rows W only x256B, depth 3, 512 blk", 512, mk(256, 0, 96, 128, 1, 3, 0));
    run<8, 0>("dma 0+24 rows W only x1024B, depth 3, 512 blk", 512, mk(1024, 0, 24, 128, 1, 3, 0));
    printf("== occupancy: W only x64B, depth 3\n");
    run<8, 0>("  256 blk x 8 waves", 256, mk(64, 0, 384, 128, 1, 3, 0));
    run<8, 0>("  768 blk x 8 waves (3/CU)", 768, mk(64, 0, 384, 128, 1, 3, 0));
    run<16, 0>("  256 blk x 16 waves", 256, mk(64, 0, 384, 128, 1, 3, 0));
    run<16, 0>("  512 blk x 16 waves", 512, mk(64, 0, 384, 128, 1, 3, 0));
    return 0;
}
