"""How fast does the vendor GEMM (torch.matmul -> hipBLASLt) run the 1x1 convolutions' shapes?  fp16, batch 240.
   python tools/gemm_ref.py          (a reference point for conv1x1_g, not a product path)"""
import time
import torch
dev = torch.device("cuda:0")
shapes = [("qkv_16_512", 240 * 256, 512, 1536), ("proj16_1024_512", 240 * 256, 1024, 512), ("out16_512_512", 240 * 256, 512, 512),
          ("proj32_768_256", 240 * 1024, 768, 256), ("proj64_384_128", 240 * 4096, 384, 128), ("proj8_1024_512", 240 * 64, 1024, 512)]
for name, M, K, N in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16)
    for _ in range(5):
        y = a @ w.t()
    torch.cuda.synchronize()
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        for _ in range(20):
            y = a @ w.t()
        torch.cuda.synchronize()
        n += 20
    dt = (time.perf_counter() - t0) / n
    print("%-18s M=%-7d K=%-5d N=%-5d %8.1f us  %7.1f TFLOP/s" % (name, M, K, N, dt * 1e6, 2.0 * M * K * N / dt / 1e12))
