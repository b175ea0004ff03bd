"""What does the vendor GEMM reach on this board?  fp16 torch.mm (hipBLASLt / rocBLAS) on N(0,1) data vs all-zero data:
the ratio shows how much of the 2.5 PFLOP/s peak the 1,400 W cap leaves for full-entropy operands.
    python tools/gemm_ceiling.py"""
import time
import torch

def run(n, fill, iters=30):
    a = torch.empty((n, n), dtype=torch.float16, device="cuda")
    b = torch.empty((n, n), dtype=torch.float16, device="cuda")
    if fill == "randn":
        a.normal_(); b.normal_()
    else:
        a.zero_(); b.zero_()
    for _ in range(5):
        c = a @ b
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(iters):
        c = a @ b
    torch.cuda.synchronize()
    dt = (time.time() - t) / iters
    return 2.0 * n ** 3 / dt / 1e12

for n in (8192, 16384):
    for fill in ("zeros", "randn", "randn"):
        print("fp16 GEMM %5d^3 %-6s: %7.1f TFLOP/s" % (n, fill, run(n, fill)), flush=True)
