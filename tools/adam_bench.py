"""k_adam alone: python tools/adam_bench.py [P] [iters]  (GHR_LIB_PATH selects a variant build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianhaircut_amd.optim import FusedAdam

P = int(sys.argv[1]) if len(sys.argv) > 1 else 479488
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
shapes = [(P, 3), (P, 1, 3), (P, 15, 3), (P, 1), (P, 3), (P, 4), (P, 1), (P, 1)]
ps = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes]
opt = FusedAdam([{"params": [p], "lr": 1e-3, "name": str(i)} for i, p in enumerate(ps)], eps=1e-15)
opt.flat_grad.normal_()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
for i in range(5):
    opt.step(zero_grad=True, nan_scan=False)
for a, b in ev:
    a.record(); opt.step(zero_grad=True, nan_scan=False); b.record()
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) for a, b in ev)
n = opt.flat_param.numel()
print("ADAM lib=%s n=%d med %.4f min %.4f ms  (%.2f TB/s at 32 B per element)" %
      (os.path.basename(os.environ.get("GHR_LIB_PATH", "") or "product"), n, t[len(t) // 2], t[0], 32.0 * n / t[len(t) // 2] / 1e9))
