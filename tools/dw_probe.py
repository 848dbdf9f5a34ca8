"""ms per weight-gradient GEMM dW[64, D] = dY^T[64, B] X[B, D] of the wide first layers (B = 131072 rows): the BLAS call
torch autograd makes, against the same contraction cut into S chunks and run as one batched GEMM + a sum."""
import json, os, sys
import torch
dev = "cuda:0"
res = {}
def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / 10, 3)
for D in (689, 3208):
    B = 131072
    x = torch.randn(B, D, device=dev); gy = torch.randn(B, 64, device=dev); w = torch.randn(64, D, device=dev); b = torch.randn(64, device=dev)
    res[f"D{D}:blas_dW"] = timed(lambda: gy.t() @ x)
    res[f"D{D}:fwd_addmm"] = timed(lambda: torch.addmm(b, x, w.t()))
    for S in (32, 64, 128, 256, 512):
        res[f"D{D}:bmm_S{S}"] = timed(lambda: torch.bmm(gy.view(S, B // S, 64).transpose(1, 2), x.view(S, B // S, D)).sum(0))
    ref = gy.t() @ x
    alt = torch.bmm(gy.view(128, B // 128, 64).transpose(1, 2), x.view(128, B // 128, D)).sum(0)
    res[f"D{D}:max_rel_diff"] = float((ref - alt).abs().max() / ref.abs().max())
print(json.dumps(res))
