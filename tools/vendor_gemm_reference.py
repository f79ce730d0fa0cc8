"""Vendor GEMM library (torch.matmul -> hipBLASLt / rocBLAS) on the layer-GEMM problem sizes, for scale (run on the GPU box):
    python tools/vendor_gemm_reference.py
Compare with `TUNE_RING=1 TUNE_K=<K> tools/bin/tune_gemm` (plain FT store).  Not used by the library."""
import os, torch, time
dev = "cuda:0"
for K in ([int(os.environ["VENDOR_K"])] if os.environ.get("VENDOR_K") else (1024, 1536, 4096)):
    x = (torch.rand(65536, K, device=dev) * 0.2 - 0.1).to(torch.bfloat16)
    w = (torch.rand(1024, K, device=dev) * 0.2 - 0.1).to(torch.bfloat16)
    for _ in range(5): y = x @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(5):
        e0.record()
        for _ in range(20): y = x @ w.t()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    fl = 2.0 * 65536 * 1024 * K
    print(f"torch.matmul (hipBLASLt/rocBLAS) bf16 65536x1024x{K}: {best*1e3:.1f} us  {fl/best*1e-9:.0f} TF")
