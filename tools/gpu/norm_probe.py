"""Which fp32 expression is torch.norm(x[:, :2], dim=-1) on this build?  (round 6: the densification statistics inside k_project_bwd
must reproduce add_densification_stats bit for bit)"""
import numpy as np
import torch
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = (torch.randn(1 << 20, 3, generator=g) * torch.exp(torch.randn(1 << 20, 1, generator=g) * 4)).to(dev)
mask = torch.rand(1 << 20, generator=g).to(dev) > 0.3
ref = torch.norm(x[mask, :2], dim=-1, keepdim=True).cpu().numpy().reshape(-1)
a = x[mask][:, 0].cpu().numpy().astype(np.float32)
b = x[mask][:, 1].cpu().numpy().astype(np.float32)
f32 = np.float32
def fma(p, q, r):
    return (p.astype(np.float64) * q.astype(np.float64) + r.astype(np.float64)).astype(np.float32)   # (double product exact; one rounding)
cands = {
    "sqrt(a*a + b*b) unfused": np.sqrt((a * a + b * b).astype(f32)),
    "sqrt(fma(b,b,a*a))": np.sqrt(fma(b, b, (a * a).astype(f32))),
    "sqrt(fma(a,a,b*b))": np.sqrt(fma(a, a, (b * b).astype(f32))),
    "sqrt in double of exact sum, rounded": np.sqrt(a.astype(np.float64) ** 2 + b.astype(np.float64) ** 2).astype(f32),
    "hypot": np.hypot(a, b).astype(f32),
}
for k, v in cands.items():
    d = (v.view(np.uint32).astype(np.int64) - ref.view(np.uint32).astype(np.int64))
    print("NORM %-40s mismatches %8d of %d, max ulp %d" % (k, int((d != 0).sum()), len(d), int(np.abs(d).max())))
