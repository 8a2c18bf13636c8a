#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned l = threadIdx.x;
    unsigned a = l * 10 + 0, b = l * 10 + 1, c = l * 10 + 2, d = l * 10 + 3;
    auto r = __builtin_amdgcn_permlane32_swap(a, c, false, false); a = r[0]; c = r[1];
    r = __builtin_amdgcn_permlane32_swap(b, d, false, false); b = r[0]; d = r[1];
    r = __builtin_amdgcn_permlane16_swap(a, b, false, false); a = r[0]; b = r[1];
    r = __builtin_amdgcn_permlane16_swap(c, d, false, false); c = r[0]; d = r[1];
    out[l * 4 + 0] = a; out[l * 4 + 1] = b; out[l * 4 + 2] = c; out[l * 4 + 3] = d;
}
int main() {
    unsigned* o; hipMalloc(&o, 256 * 4); unsigned h[256];
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o); hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int s = 0; s < 4; ++s) {
        int q = l >> 4, fi = l & 15; unsigned want = (unsigned)((s * 16 + fi) * 10 + q);
        if (h[l * 4 + s] != want) { if (bad < 8) printf("lane %d reg %d: got %u want %u\n", l, s, h[l * 4 + s], want); ++bad; }
    }
    printf("bad %d\n", bad);
    for (int l = 0; l < 64; l += 16) printf("lane %d: %u %u %u %u\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
}
