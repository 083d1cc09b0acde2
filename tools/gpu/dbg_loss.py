"""where the marching loss kernels differ from the tile kernels (debug)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gaussianhaircut_amd.fused_loss import gt_ssim_stats, stage1_loss
dev = torch.device("cuda:0")
for H, W, mc, cached in [(48, 64, True, False), (52, 100, False, True), (64, 64, True, True), (48, 64, True, True), (270, 480, True, True)]:
    g = torch.Generator().manual_seed(11 * H + W)
    gt = torch.rand(3, H, W, generator=g); r = torch.rand(10, H, W, generator=g)
    r[5:8] = torch.randn(3, H, W, generator=g) * 0.3; r[8] = r[8] * 2 + 0.05
    gt_mask = (torch.rand(2, H, W, generator=g) > 0.35).float()
    ga, go = torch.rand(1, H, W, generator=g), torch.rand(1, H, W, generator=g)
    c = [t.to(dev) for t in (gt, gt_mask, ga, go)]
    outs = []
    for scalar in (True, False):
        if scalar: os.environ["GHR_LOSS_SCALAR"] = "1"
        else: os.environ.pop("GHR_LOSS_SCALAR", None)
        st = gt_ssim_stats(c[0], c[1], mc) if cached else None
        a = r.to(dev).requires_grad_(True)
        loss = stage1_loss(a, *c, 0.8, 0.2, 0.2, 0.1, mask_colours=mc, gt_stats=st)
        loss.backward(); torch.cuda.synchronize()
        outs.append((float(loss), a.grad.clone(), st))
    d = (outs[0][1] - outs[1][1]).abs()
    print("case", H, W, mc, cached, "loss", outs[0][0], outs[1][0])
    for ch in range(10):
        nz = (d[ch] > 0).nonzero()
        if len(nz):
            print("  plane", ch, "differs at", len(nz), "px; rows", int(nz[:, 0].min()), int(nz[:, 0].max()), "cols", int(nz[:, 1].min()), int(nz[:, 1].max()),
                  "max", float(d[ch].max()), "ref max", float(outs[0][1][ch].abs().max()))
    if cached:
        ds = (outs[0][2] - outs[1][2]).abs()
        print("  stats diff", float(ds.max()))
