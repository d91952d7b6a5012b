"""Reference point only (not used by the product): what rocBLAS/hipBLASLt (through torch.matmul) reach on the
encoder's GEMM shapes, f16 in / f16 out, to size the head-room of csrc/k_gemm.hip."""
import torch, time
dev = torch.device("cuda", 0)
for (M, N, K) in [(16000, 2048, 512), (16000, 1536, 512), (16000, 512, 2048), (16000, 512, 512), (16384, 2048, 512), (8192, 8192, 8192)]:
    A = torch.randn(M, K, device=dev, dtype=torch.float16)
    W = torch.randn(N, K, device=dev, dtype=torch.float16)
    b = torch.randn(N, device=dev, dtype=torch.float16)
    for name, fn in (("matmul", lambda: A @ W.t()), ("linear+bias", lambda: torch.nn.functional.linear(A, W, b)),
                     ("linear+bias+relu", lambda: torch.relu_(torch.nn.functional.linear(A, W, b)))):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print("M=%d N=%d K=%d %-18s %.1f us  %.0f TF" % (M, N, K, name, us, 2.0 * M * N * K / us / 1e6), flush=True)
