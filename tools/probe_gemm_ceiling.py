"""What the vendor GEMM reaches on this box (torch.matmul = hipBLASLt/rocBLAS), as a yardstick for the all-pairs
forward: square bf16 GEMMs (the practical matrix-pipe ceiling under the power cap) and the training
step's own shape, 8704 x 768 by 768 x N in doc chunks (the similarity matrix written out, which the fused kernel never
does).  Diagnostic only; nothing in the product calls torch.matmul."""
import json
import sys
import time

import torch


def rate(m, n, k, dt, iters=20):
    a = torch.randn(m, k, device="cuda", dtype=dt)
    b = torch.randn(n, k, device="cuda", dtype=dt)
    out = torch.empty(m, n, device="cuda", dtype=dt)
    for _ in range(3):
        torch.matmul(a, b.t(), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.matmul(a, b.t(), out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return {"m": m, "n": n, "k": k, "ms": round(ms, 4), "tflops": round(2.0 * m * n * k / ms / 1e9, 1)}


def main():
    res = []
    for dt, name in ((torch.bfloat16, "bf16"), (torch.float16, "f16")):
        for (m, n, k) in ((8192, 8192, 8192), (16384, 16384, 4096), (8704, 8704, 768), (8704, 26112, 768),
                          (8704, 52224, 768), (8704, 208896, 768)):
            r = rate(m, n, k, dt)
            r["dtype"] = name
            res.append(r)
            print(json.dumps(r), flush=True)
    json.dump(res, open(sys.argv[1], "w"), indent=1) if len(sys.argv) > 1 else None


if __name__ == "__main__":
    main()
