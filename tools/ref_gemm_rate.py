"""Reference point, not part of the product: what rate does the vendor GEMM (hipBLASLt through torch.matmul)
reach on the layer shapes of this model?  (plain bf16 GEMM, no segments, no epilogue)"""
import torch, time
dev = "cuda:0"
def rate(M, N, K, reps=20):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(3): (a @ b.t())
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): c = a @ b.t()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    return dt * 1e6, 2.0 * M * N * K / dt / 1e12
for name, M, N, K in [("G1  (gated)", 48000, 512, 896), ("dx", 48000, 384, 1024), ("dz", 48000, 256, 640),
                      ("G2", 48000, 384, 256), ("wgrad fg (TN)", 512, 896, 48000), ("big square", 8192, 8192, 8192)]:
    if name.startswith("wgrad"):
        a = torch.randn(48000, 512, device=dev, dtype=torch.bfloat16); b = torch.randn(48000, 896, device=dev, dtype=torch.bfloat16)
        for _ in range(3): (a.t() @ b)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): c = a.t() @ b
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
        us, tf = dt * 1e6, 2.0 * 512 * 896 * 48000 / dt / 1e12
    else:
        us, tf = rate(M, N, K)
    print(f"{name:16s} M={M:6d} N={N:5d} K={K:6d}: {us:8.1f} us  {tf:7.1f} TFLOP/s")
