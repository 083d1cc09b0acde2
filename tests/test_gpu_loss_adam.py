"""`-m gpu`: fused loss and fused Adam against the PyTorch implementations they replace."""
import numpy as np
import pytest
import torch

from gaussianhaircut_amd.utils import loss_utils as lu

pytestmark = pytest.mark.gpu


def _torch_loss(image, mask, gt_image, gt_mask, w):
    m = gt_mask[1:]
    return (w[0] * lu.l1_loss(image, gt_image, mask=m) + w[1] * (1.0 - lu.ssim(image * m, gt_image * m)) +
            w[2] * lu.l1_loss(mask, gt_mask))


@pytest.mark.parametrize("H,W", [(48, 64), (37, 70), (135, 240), (1080, 1920)])
def test_fused_loss_matches_torch(H, W):
    from gaussianhaircut_amd.fused_loss import photometric_loss
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(H * 1000 + W)
    # smooth-ish images so SSIM is away from 0, plus noise
    base = torch.rand(3, H // 4 + 2, W // 4 + 2, generator=g)
    gt = torch.nn.functional.interpolate(base[None], size=(H, W), mode="bilinear")[0]
    image = (gt + 0.15 * torch.randn(3, H, W, generator=g)).clamp(-0.2, 1.3)
    gt_mask = (torch.rand(2, H, W, generator=g) > 0.35).float()
    mask = torch.rand(2, H, W, generator=g)
    w = (0.8, 0.2, 0.2)
    a = [t.to(dev).requires_grad_(r) for t, r in ((image, True), (mask, True), (gt, False), (gt_mask, False))]
    b = [t.detach().clone().requires_grad_(t.requires_grad) for t in a]
    lf = photometric_loss(a[0], a[1], a[2], a[3], *w)
    lt = _torch_loss(b[0], b[1], b[2], b[3], w)
    assert abs(float(lf) - float(lt)) < 2e-6 * max(1.0, abs(float(lt)))
    (lf * 0.37).backward()
    (lt * 0.37).backward()
    for x, y, name in ((a[0].grad, b[0].grad, "image"), (a[1].grad, b[1].grad, "mask")):
        x, y = x.cpu().numpy(), y.cpu().numpy()
        scale = np.abs(y).max()
        assert np.abs(x - y).max() <= 2e-4 * scale, (name, np.abs(x - y).max(), scale)


def test_fused_adam_matches_torch_adam_and_nan_guard():
    from gaussianhaircut_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    shapes = [(1000, 3), (1000, 1, 3), (1000, 15, 3), (1000, 1), (1000, 4)]
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 0.05, 1e-3]
    init = [torch.randn(*s, generator=g) for s in shapes]
    pa = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    pb = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    fa = FusedAdam([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(pa, lrs))], eps=1e-15)
    tb = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(pb, lrs)], lr=0.0, eps=1e-15)
    for it in range(5):
        grads = [torch.randn(*s, generator=g).to(dev) * (10.0 ** (it - 2)) for s in shapes]
        for p, q, gr in zip(pa, pb, grads):
            p.grad.copy_(gr)
            q.grad = gr.clone()
        if it == 2:
            fa.param_groups[0]["lr"] = tb.param_groups[0]["lr"] = 7e-5  # lr schedule edits must be honoured
        fa.step()
        tb.step()
        assert float(fa.flat_grad.abs().sum()) == 0.0  # folded zero_grad
        for p, q in zip(pa, pb):
            np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)
    # NaN guard: nothing moves, the step counter does not advance, the gradient is cleared
    before = [p.detach().clone() for p in pa]
    step_before = int(fa.state_dev[0])
    pa[1].grad[5, 0, 1] = float("nan")
    pa[0].grad.fill_(1.0)
    fa.step()
    assert int(fa.state_dev[0]) == step_before and int(fa.state_dev[1]) == 0
    for p, q in zip(pa, before):
        assert torch.equal(p.detach(), q)
    assert float(fa.flat_grad.abs().sum()) == 0.0
