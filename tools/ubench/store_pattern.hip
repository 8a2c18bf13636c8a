// Micro-benchmark (round 4): what does the access pattern of an epilogue's 16-byte global stores / loads cost?
// Every wave owns 64 x 64 bf16 sub-tiles (8 KiB) of a [rows][pitch] matrix, as in the NT kernels' epilogue, and writes
// (mode 0) or reads (mode 1) them with 8 wave instructions of 16 B per lane:
//   pat 0  MFMA layout: instruction (j, u) covers rows j*16 + (lane & 15), bytes u*64 + (lane >> 4)*16 .. +16
//          = 16 row segments of 64 B per instruction (what nt_epilogue does today)
//   pat 1  row-contiguous: instruction k covers rows k*8 + (lane >> 3), bytes (lane & 7)*16 .. +16
//          = 8 row segments of 128 B (one full L2 line each)
//   pat 2  as 1 but a tile is 32 rows x 256 B (a wave owns 128 channels): 4 row segments of 256 B per instruction
//   pat 3  fully linear: 1 KiB contiguous per instruction (pitch = tile width; the ceiling)
//   hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

template <int PAT, int MODE>
__global__ __launch_bounds__(512) void k(char* base, long pitch, int row_tiles, int col_tiles, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ntile = row_tiles * col_tiles;
    uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int t = blockIdx.x * 8 + wave; t < ntile; t += gridDim.x * 8) {
        const int rt = t / col_tiles, ct = t - rt * col_tiles;
        char* p;
        if (PAT == 2) p = base + (long)rt * 32 * pitch + (long)ct * 256;
        else if (PAT == 3) p = base + (long)t * 8192;
        else p = base + (long)rt * 64 * pitch + (long)ct * 128;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            char* q;
            if (PAT == 0) q = p + (long)((i >> 1) * 16 + (lane & 15)) * pitch + (i & 1) * 64 + (lane >> 4) * 16;
            else if (PAT == 1) q = p + (long)(i * 8 + (lane >> 3)) * pitch + (lane & 7) * 16;
            else if (PAT == 2) q = p + (long)(i * 4 + (lane >> 4)) * pitch + (lane & 15) * 16;
            else q = p + i * 1024 + lane * 16;
            if (MODE == 0) *reinterpret_cast<uint4*>(q) = v;
            else { const uint4 r = *reinterpret_cast<const uint4*>(q); acc.x ^= r.x; acc.y ^= r.y; acc.z ^= r.z; acc.w ^= r.w; }
        }
        v.x += 1;
    }
    if (MODE == 1 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1.f;
}

template <int PAT, int MODE>
static void run(const char* name, char* buf, long pitch, int rows, float* sink, int blocks) {
    const int row_tiles = rows / 64 * (PAT == 2 ? 2 : 1), col_tiles = (int)(pitch / (PAT == 2 ? 256 : 128));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<PAT, MODE>), dim3(blocks), dim3(512), 0, 0, buf, pitch, row_tiles, col_tiles, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best) best = ms;
    }
    const double bytes = (double)rows * pitch;
    printf("%-52s %8.1f us  %6.2f TB/s  %5.1f B/clk/CU\n", name, best * 1e3, bytes / (best * 1e-3) / 1e12,
           bytes / (best * 1e-3) / 256 / 2.4e9);
}

int main(int argc, char** argv) {
    const long pitch = argc > 1 ? atol(argv[1]) : 768;     // bytes per row (384 bf16 channels)
    const int rows = argc > 2 ? atoi(argv[2]) : 8 * 7040;
    const int blocks = argc > 3 ? atoi(argv[3]) : 512;
    char* buf; float* sink;
    hipMalloc(&buf, (size_t)rows * pitch + (1 << 20));
    hipMalloc(&sink, 64);
    hipMemset(buf, 1, (size_t)rows * pitch);
    printf("matrix %d rows x %ld B = %.1f MB, %d blocks x 8 waves\n", rows, pitch, rows * (double)pitch / 1e6, blocks);
    run<0, 0>("store, MFMA layout (16 x 64 B per instruction)", buf, pitch, rows, sink, blocks);
    run<1, 0>("store, row-contiguous (8 x 128 B)", buf, pitch, rows, sink, blocks);
    run<2, 0>("store, row-contiguous (4 x 256 B)", buf, pitch, rows, sink, blocks);
    run<3, 0>("store, linear 1 KiB", buf, pitch, rows, sink, blocks);
    run<0, 1>("load,  MFMA layout (16 x 64 B per instruction)", buf, pitch, rows, sink, blocks);
    run<1, 1>("load,  row-contiguous (8 x 128 B)", buf, pitch, rows, sink, blocks);
    run<2, 1>("load,  row-contiguous (4 x 256 B)", buf, pitch, rows, sink, blocks);
    run<3, 1>("load,  linear 1 KiB", buf, pitch, rows, sink, blocks);
    return 0;
}
