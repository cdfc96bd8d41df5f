"""Context for the f64 MFMA ceiling: what the vendor library (rocBLAS / hipBLASLt through torch.matmul) reaches on the
same shapes as our hand-written gemm_mfma.  Not part of the product (no library GEMM is linked)."""
import json, time, torch
dev = torch.device("cuda", 0)
for (m, n, k, dt) in [(4096, 4096, 4096, torch.float64), (8192, 8192, 8192, torch.float64), (4096, 4096, 4096, torch.complex128),
                      (32768, 133, 32768, torch.float64)]:
    a = torch.randn(m, k, dtype=dt, device=dev); b = torch.randn(k, n, dtype=dt, device=dev)
    for _ in range(3): c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps): c = a @ b
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * m * n * k * (4 if dt == torch.complex128 else 1)
    print(json.dumps({"library": "torch.matmul (rocBLAS/hipBLASLt)", "m": m, "n": n, "k": k, "dtype": str(dt).split(".")[-1],
                      "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 2)}), flush=True)
