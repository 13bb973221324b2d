"""The tuned-library GEMM rate on the shape of the dominant launch (M = 50400 rows, N = 512, K = 9 * 512): what hipBLASLt reaches
with plain f16 / bf16 operands -- the anchor the emulated-float32 conv kernels are compared with (2 or 3 such products per product)."""
import torch
for dt in (torch.float16, torch.bfloat16):
    for (M, N, K) in ((50400, 512, 4608), (50400, 256, 2304), (8192, 8192, 8192)):
        a = torch.randn(M, K, device="cuda", dtype=dt)
        b = torch.randn(K, N, device="cuda", dtype=dt)
        for _ in range(5):
            c = a @ b
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            c = a @ b
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 20
        print("%s  M %d N %d K %d: %.1f us  %.1f TFLOP/s" % (str(dt).split(".")[1], M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)
