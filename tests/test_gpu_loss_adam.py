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


def test_vector_adam_kernel_equals_the_scalar_one_bit_for_bit(monkeypatch):
    """k_adam_v4 (four elements per thread, 16-B accesses) against k_adam (GHR_ADAM_SCALAR=1, read per call): group
    boundaries that fall inside a float4 (P = 1001), a tail that is not a multiple of four, a skipped group after surgery,
    torch.optim.Adam as the third party."""
    from gaussianhaircut_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    P = 1001
    shapes = [(P, 3), (P, 1, 3), (P, 15, 3), (P, 1), (P, 4), (P, 1)]
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 0.05, 1e-3, 5e-3]
    g = torch.Generator().manual_seed(3)
    init = [torch.randn(*s, generator=g) for s in shapes]
    runs = []
    for scalar in (True, False):
        if scalar:
            monkeypatch.setenv("GHR_ADAM_SCALAR", "1")
        else:
            monkeypatch.delenv("GHR_ADAM_SCALAR", raising=False)
        ps = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
        fa = FusedAdam([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(ps, lrs))], eps=1e-15)
        gg = torch.Generator().manual_seed(4)
        for it in range(4):
            for p, s_ in zip(ps, shapes):
                p.grad.copy_(torch.randn(*s_, generator=gg).to(dev))
            if it == 2:
                fa._skip_next = 0b000100  # group 2 sits this step out (as after optimizer surgery)
            fa.step()
        torch.cuda.synchronize()
        runs.append((fa.flat_param.clone(), fa.exp_avg.clone(), fa.exp_avg_sq.clone(), fa.state_dev.clone(), fa.flat_grad.clone()))
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    assert float(runs[1][4].abs().sum()) == 0.0 and runs[1][0].numel() % 4 != 0


@pytest.mark.parametrize("chunks", [1, 3, 4, 7])
def test_chunked_adam_equals_whole_buffer_adam_bit_for_bit(chunks):
    """FusedAdam.step_chunked (the update applied range by range, as the chunks of the gradient all-reduce arrive) must
    be the same update as step(): parameters, both moments, step counter, cleared gradients -- and the same skip when the
    producer-side NaN flag is up."""
    from gaussianhaircut_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    shapes = [(3001, 3), (3001, 1, 3), (3001, 15, 3), (3001, 1), (3001, 4)]   # group ends not multiples of the chunking
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 0.05, 1e-3]
    init = [torch.randn(*s, generator=g) for s in shapes]
    opts = []
    for _ in range(2):
        ps = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
        opts.append(FusedAdam([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(ps, lrs))],
                              eps=1e-15))
    whole, chunked = opts
    assert len(chunked._chunk_ranges(chunks)) >= 1 and chunked._chunk_ranges(chunks)[-1][1] == chunked.flat_param.numel()
    for it in range(4):
        grad = torch.randn(whole.flat_grad.numel(), generator=g).to(dev) * (10.0 ** (it - 2))
        nan_step = it == 2
        for o in opts:
            o.flat_grad.copy_(grad)
            o._direct_backwards = 1          # "every gradient came through the fused backward"
            if nan_step:
                o.state_dev[1] = 1           # ... which raised its flag
        whole.step(zero_grad=True, nan_scan=False)
        chunked.step_chunked(chunks=chunks, zero_grad=True)
        torch.cuda.synchronize()
        for name in ("flat_param", "exp_avg", "exp_avg_sq", "flat_grad", "state_dev"):
            assert torch.equal(getattr(whole, name), getattr(chunked, name)), (it, name)
    assert int(whole.state_dev[0]) == 3 and int(whole.state_dev[1]) == 0   # three applied steps, one skipped


@pytest.mark.parametrize("begin,count", [(0, 9003), (9003, 9003), (9004, 3000), (63021, 12004), (9003 + 1, 45015 + 3)])
def test_out_of_place_adam_range_equals_copy_plus_in_place_range_bit_for_bit(begin, count, monkeypatch):
    """ghr_adam_step_range_to (ABI 20: p, m, v read from one buffer set, written to another; the late groups of a fused strand
    step) against what it replaces -- three copies in -> out and ghr_adam_step_range on the out buffers: the same bits in the
    range, nothing touched outside it, gradients zeroed alike; with the step's own flag word up (and with a group in
    skip_mask) the range is copied, not updated.  Vector and scalar kernels, aligned and unaligned ranges."""
    import ctypes
    from gaussianhaircut_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    L = _lib.lib()
    sizes = [9003, 9003, 45015, 3001, 12004]
    n = sum(sizes)
    ends = (ctypes.c_int64 * len(sizes))(*[sum(sizes[:i + 1]) for i in range(len(sizes))])
    lrs = (ctypes.c_float * len(sizes))(1.6e-4, 2.5e-3, 1.25e-4, 0.05, 1e-3)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    for scalar in (False, True):
        if scalar:
            monkeypatch.setenv("GHR_ADAM_SCALAR", "1")
        for flag_up, skip_mask in ((0, 0), (1, 0), (0, 0b00101)):
            p_in = torch.randn(n, generator=g).to(dev)
            m_in, v_in = (torch.randn(n, generator=g) * 0.01).to(dev), (torch.rand(n, generator=g) * 1e-3).to(dev)
            grad = torch.randn(n, generator=g).to(dev)
            state = torch.zeros(_lib.ADAM_STATE, dtype=torch.int32, device=dev)
            state[0] = 6
            flag = torch.tensor([flag_up], dtype=torch.int32, device=dev)
            junk = [torch.full((n,), 7.0, device=dev) for _ in range(6)]
            # reference: copy + in-place range (the skip word moved through state[1])
            po, mo, vo = junk[0], junk[1], junk[2]
            po[begin:begin + count].copy_(p_in[begin:begin + count])
            mo[begin:begin + count].copy_(m_in[begin:begin + count])
            vo[begin:begin + count].copy_(v_in[begin:begin + count])
            g_ref, st_ref = grad.clone(), state.clone()
            st_ref[1] = flag_up
            _lib.check(L.ghr_adam_step_range(stream, n, begin, count, ptr(po), ptr(g_ref), ptr(mo), ptr(vo), ptr(st_ref), len(sizes),
                                             ends, lrs, 0.9, 0.999, 1e-15, 2, 1, 0, skip_mask))
            # out of place
            pt, mt, vt = junk[3], junk[4], junk[5]
            g_new, st_new = grad.clone(), state.clone()
            _lib.check(L.ghr_adam_step_range_to(stream, n, begin, count, ptr(p_in), ptr(m_in), ptr(v_in), ptr(pt), ptr(g_new),
                                                ptr(mt), ptr(vt), ptr(st_new), ptr(flag), 1, len(sizes), ends, lrs, 0.9, 0.999,
                                                1e-15, 1, skip_mask))
            torch.cuda.synchronize()
            for a, b, src in ((po, pt, p_in), (mo, mt, m_in), (vo, vt, v_in)):
                assert torch.equal(a, b), (scalar, flag_up, skip_mask)
                if flag_up:
                    assert torch.equal(b[begin:begin + count], src[begin:begin + count])
                elif skip_mask == 0:
                    assert not torch.equal(b[begin:begin + count], src[begin:begin + count])
                assert float((b[:begin] - 7.0).abs().sum()) == 0.0 and float((b[begin + count:] - 7.0).abs().sum()) == 0.0
            assert torch.equal(g_ref, g_new) and float(g_new[begin:begin + count].abs().sum()) == 0.0
            assert torch.equal(st_new, state)   # neither the counter nor state[1] is touched
            assert int(flag) == flag_up         # finite gradients: nan_mark leaves the word alone
            # nan_mark: a NaN among the range's gradients raises the word (the scan ghr_adam_nan_scan did in front of the update);
            # one outside the range does not
            for where, want in ((begin + count // 2, 1), ((begin + count) % n if count < n else None, 0)):
                if where is None or (want == 0 and begin <= where < begin + count):
                    continue
                g_nan = grad.clone()
                g_nan[where] = float("nan")
                fl = torch.zeros(1, dtype=torch.int32, device=dev)
                _lib.check(L.ghr_adam_step_range_to(stream, n, begin, count, ptr(p_in), ptr(m_in), ptr(v_in), ptr(pt), ptr(g_nan),
                                                    ptr(mt), ptr(vt), ptr(st_new), ptr(fl), 1, len(sizes), ends, lrs, 0.9, 0.999,
                                                    1e-15, 0, 0))
                torch.cuda.synchronize()
                assert int(fl) == want, (scalar, where, want)
    # in == out is refused (that is ghr_adam_step_range)
    t = torch.zeros(8, device=dev)
    st = torch.zeros(18, dtype=torch.int32, device=dev)
    rc = L.ghr_adam_step_range_to(stream, 8, 0, 8, ptr(t), ptr(t), ptr(t), ptr(t), ptr(t), ptr(t), ptr(t), ptr(st), None, 0, 1,
                                  (ctypes.c_int64 * 1)(8), (ctypes.c_float * 1)(0.1), 0.9, 0.999, 1e-15, 1, 0)
    assert rc != 0 and b"must differ" in L.ghr_last_error()


def _torch_stage1(renders, gt_image, gt_mask, gt_angle, gt_oconf, w):
    """The reference's loss code path on the packed output (gaussian_renderer/__init__.py:100-105 +
    train_gaussians.py:126-140), PyTorch autograd."""
    from gaussianhaircut_amd.gaussian_renderer import orient_angle_from
    image, mask, cov2d, oconf, _ = renders.split([3, 2, 3, 1, 1], dim=0)
    angle = orient_angle_from(cov2d)
    base = _torch_loss(image, mask, gt_image, gt_mask, w[:3])
    weight = torch.ones_like(gt_mask[:1]) * gt_oconf
    lo = lu.or_loss(angle, gt_angle, oconf, weight=weight, mask=gt_mask[:1])
    lo = torch.where(torch.isnan(lo), torch.zeros_like(lo), lo)
    return base + w[3] * lo


@pytest.mark.parametrize("H,W,w_orient", [(48, 64, 0.1), (37, 70, 0.1), (270, 480, 0.1), (48, 64, 0.0)])
def test_fused_stage1_loss_with_orientation_matches_torch(H, W, w_orient):
    from gaussianhaircut_amd.fused_loss import stage1_loss
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7 * H + W)
    base = torch.rand(3, H // 4 + 2, W // 4 + 2, generator=g)
    gt = torch.nn.functional.interpolate(base[None], size=(H, W), mode="bilinear")[0]
    r = torch.zeros(10, H, W)
    r[0:3] = (gt + 0.15 * torch.randn(3, H, W, generator=g)).clamp(-0.2, 1.3)
    r[3:5] = torch.rand(2, H, W, generator=g)
    r[5:8] = torch.randn(3, H, W, generator=g) * 0.3          # 2D direction (+ unused z)
    r[8] = torch.rand(H, W, generator=g) * 2 + 0.05           # orientation confidence > 0
    r[9] = torch.rand(H, W, generator=g) * 5
    r[5:7, : H // 6] = 0.0                                     # empty pixels: zero direction (normalize's eps branch)
    r[8, : H // 6] = 0.0
    gt_mask = (torch.rand(2, H, W, generator=g) > 0.35).float()
    gt_angle = torch.rand(1, H, W, generator=g)
    gt_oconf = torch.rand(1, H, W, generator=g)
    w = (0.8, 0.2, 0.2, w_orient)
    a = r.to(dev).requires_grad_(True)
    b = r.to(dev).requires_grad_(True)
    consts = [t.to(dev) for t in (gt, gt_mask, gt_angle, gt_oconf)]
    lf = stage1_loss(a, *consts, *w)
    lt = _torch_stage1(b, *consts, w)
    assert abs(float(lf.detach()) - float(lt.detach())) < 5e-6 * max(1.0, abs(float(lt.detach())))
    (lf * 0.37).backward()
    (lt * 0.37).backward()
    x, y = a.grad.cpu().numpy(), b.grad.cpu().numpy()
    assert np.isfinite(x).all()
    for lo_, hi_, name in ((0, 3, "image"), (3, 5, "mask"), (5, 7, "dir2d"), (7, 8, "dir z"), (8, 9, "conf"), (9, 10, "depth")):
        scale = max(np.abs(y[lo_:hi_]).max(), 1e-30)
        # orientation gradients are discontinuous where the wrapped-difference branch or the mirror flips: allow a
        # handful of pixels sitting exactly on a branch boundary to pick the other side
        bad = np.abs(x[lo_:hi_] - y[lo_:hi_]) > 2e-4 * scale
        assert bad.sum() <= (3 if name in ("dir2d", "conf") else 0), (name, int(bad.sum()), scale)


def test_fused_stage1_loss_nan_orientation_is_dropped():
    from gaussianhaircut_amd.fused_loss import stage1_loss
    dev = torch.device("cuda:0")
    H, W = 32, 48
    g = torch.Generator().manual_seed(5)
    r = torch.rand(10, H, W, generator=g)
    gt, gt_mask = torch.rand(3, H, W, generator=g), (torch.rand(2, H, W, generator=g) > 0.5).float()
    gt_angle, gt_oconf = torch.rand(1, H, W, generator=g), torch.zeros(1, H, W)   # weight.sum() == 0 -> 0/0 = NaN
    a = r.to(dev).requires_grad_(True)
    b = r.to(dev).requires_grad_(True)
    consts = [t.to(dev) for t in (gt, gt_mask, gt_angle, gt_oconf)]
    lf = stage1_loss(a, *consts, 0.8, 0.2, 0.2, 0.1)
    lt = _torch_stage1(b, *consts, (0.8, 0.2, 0.2, 0.1))
    assert torch.isfinite(lf) and abs(float(lf.detach()) - float(lt.detach())) < 5e-6
    lf.backward()
    assert torch.isfinite(a.grad).all() and float(a.grad[5:].abs().max()) == 0.0


def test_fused_strand_stage_loss_unmasked_colours_matches_torch():
    """mask_colours=False: L1 / SSIM over the whole image (src/train_strands.py:128-129) + mask L1 + orientation."""
    from gaussianhaircut_amd.fused_loss import stage1_loss
    from gaussianhaircut_amd.gaussian_renderer import orient_angle_from
    dev = torch.device("cuda:0")
    H, W = 70, 96
    g = torch.Generator().manual_seed(77)
    base = torch.rand(3, H // 4 + 2, W // 4 + 2, generator=g)
    gt = torch.nn.functional.interpolate(base[None], size=(H, W), mode="bilinear")[0]
    r = torch.rand(10, H, W, generator=g)
    r[0:3] = (gt + 0.1 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
    r[5:8] = torch.randn(3, H, W, generator=g) * 0.3
    r[8] = torch.rand(H, W, generator=g) + 0.1
    gt_mask = (torch.rand(2, H, W, generator=g) > 0.4).float()
    gt_angle, gt_oconf = torch.rand(1, H, W, generator=g), torch.rand(1, H, W, generator=g)
    w = (0.8, 0.2, 0.1, 0.1)
    a = r.to(dev).requires_grad_(True)
    b = r.to(dev).requires_grad_(True)
    c = [t.to(dev) for t in (gt, gt_mask, gt_angle, gt_oconf)]
    lf = stage1_loss(a, *c, *w, mask_colours=False)
    image, mask, cov2d, oconf, _ = b.split([3, 2, 3, 1, 1], dim=0)
    lo = lu.or_loss(orient_angle_from(cov2d), c[2], oconf, weight=torch.ones_like(c[1][:1]) * c[3], mask=c[1][:1])
    lt = w[0] * lu.l1_loss(image, c[0]) + w[1] * (1.0 - lu.ssim(image, c[0])) + w[2] * lu.l1_loss(mask, c[1]) + w[3] * lo
    assert abs(float(lf.detach()) - float(lt.detach())) < 5e-6 * max(1.0, abs(float(lt.detach())))
    lf.backward()
    lt.backward()
    x, y = a.grad.cpu().numpy(), b.grad.cpu().numpy()
    for lo_, hi_ in ((0, 3), (3, 5)):
        assert np.abs(x[lo_:hi_] - y[lo_:hi_]).max() <= 2e-4 * np.abs(y[lo_:hi_]).max()


@pytest.mark.parametrize("H,W,mask_colours", [(48, 64, True), (37, 70, False), (270, 480, True), (1080, 1920, True)])
def test_cached_ground_truth_window_moments_give_bit_identical_loss(H, W, mask_colours):
    """gt_ssim_stats(): the SSIM window moments of the (masked) ground truth are constants of a training view; with
    them the forward convolves three moments instead of five.  Same arithmetic per moment => same bits: loss value and
    every gradient element."""
    from gaussianhaircut_amd.fused_loss import gt_ssim_stats, stage1_loss
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(H + 3 * W)
    gt = torch.rand(3, H, W, generator=g)
    r = torch.rand(10, H, W, generator=g)
    r[5:8] = torch.randn(3, H, W, generator=g) * 0.3
    r[8] = r[8] * 2 + 0.05
    gt_mask = (torch.rand(2, H, W, generator=g) > 0.35).float()
    gt_angle, gt_oconf = torch.rand(1, H, W, generator=g), torch.rand(1, H, W, generator=g)
    consts = [t.to(dev) for t in (gt, gt_mask, gt_angle, gt_oconf)]
    stats = gt_ssim_stats(consts[0], consts[1], mask_colours)
    assert stats.shape == (2, 3, H, W) and torch.isfinite(stats).all()
    # spot check against torch: mu2 = conv(y), E[y^2] = conv(y^2) with the reference's 11x11 window (zero padding)
    from gaussianhaircut_amd.utils.loss_utils import _window
    y = consts[0] * (consts[1][1:] if mask_colours else 1.0)
    win = _window(11, 3, y)
    mu = torch.nn.functional.conv2d(y[None], win, padding=5, groups=3)[0]
    e2 = torch.nn.functional.conv2d((y * y)[None], win, padding=5, groups=3)[0]
    assert torch.allclose(stats[0], mu, atol=2e-6) and torch.allclose(stats[1], e2, atol=2e-6)
    outs = []
    for st in (None, stats):
        a = r.to(dev).requires_grad_(True)
        loss = stage1_loss(a, *consts, 0.8, 0.2, 0.2, 0.1, mask_colours=mask_colours, gt_stats=st)
        loss.backward()
        outs.append((loss.detach().clone(), a.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("H,W,mask_colours,cached", [(48, 64, True, False), (52, 100, False, True), (270, 480, True, True),
                                                      (1080, 1920, True, True), (1080, 1920, True, False)])
def test_vector_loss_kernels_equal_the_scalar_ones_bit_for_bit(H, W, mask_colours, cached, monkeypatch):
    """Images with 16-B aligned rows (W % 4 == 0) take the vector form of the loss kernels (csrc/ghr_loss.h: aligned float4
    window loads, a block marching down a strip of 32 columns, DPP wave sums); GHR_LOSS_SCALAR=1 (read per call) forces the
    tile kernels.  Same arithmetic in the same order per output: the image / mask gradients and the cached window moments
    are the same bits; the loss value and the orientation term's normaliser differ only by the order their partial sums are
    added in."""
    from gaussianhaircut_amd.fused_loss import gt_ssim_stats, stage1_loss
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11 * H + W)
    gt = torch.rand(3, H, W, generator=g)
    r = torch.rand(10, H, W, generator=g)
    r[5:8] = torch.randn(3, H, W, generator=g) * 0.3
    r[8] = r[8] * 2 + 0.05
    gt_mask = (torch.rand(2, H, W, generator=g) > 0.35).float()
    gt_angle, gt_oconf = torch.rand(1, H, W, generator=g), torch.rand(1, H, W, generator=g)
    consts = [t.to(dev) for t in (gt, gt_mask, gt_angle, gt_oconf)]
    outs = []
    for scalar in (True, False):
        if scalar:
            monkeypatch.setenv("GHR_LOSS_SCALAR", "1")
        else:
            monkeypatch.delenv("GHR_LOSS_SCALAR", raising=False)
        st = gt_ssim_stats(consts[0], consts[1], mask_colours) if cached else None
        a = r.to(dev).requires_grad_(True)
        loss = stage1_loss(a, *consts, 0.8, 0.2, 0.2, 0.1, mask_colours=mask_colours, gt_stats=st)
        (loss * 0.37).backward()
        torch.cuda.synchronize()
        outs.append((float(loss.detach()), a.grad.clone(), st))
    assert abs(outs[0][0] - outs[1][0]) <= 2e-6 * abs(outs[0][0])
    ga, gb = outs[0][1], outs[1][1]
    assert torch.equal(ga[:5], gb[:5]) and torch.equal(ga[7], gb[7]) and torch.equal(ga[9], gb[9])
    # the orientation gradients carry 1 / sum(weights): a sum of per-block partial sums, i.e. order-dependent in its last bit
    for c in (5, 6, 8):
        assert float((ga[c] - gb[c]).abs().max()) <= 3e-7 * float(ga[c].abs().max())
    if cached:
        assert torch.equal(outs[0][2], outs[1][2])


def test_step_after_optimizer_surgery_passes_the_replaced_groups_by_like_torch_adam():
    """surgery (densify / prune / opacity reset) sits between backward() and optimizer.step() in the
    reference's loop (train_gaussians.py:158-181).  The nn.Parameters it creates have grad None, so torch.optim.Adam's
    step() passes them by on that iteration: no moment decay, no update, and their per-parameter step counter does not
    advance.  FusedAdam must do the same (ghr_adam_step's skip_mask + per-group step lag), including the bias correction
    of the following steps."""
    from gaussianhaircut_amd.scene.gaussian_model import OptimizationParams
    from gaussianhaircut_amd.utils import synthetic as syn
    from tests.test_reference_golden import _densify_sequence
    dev = torch.device("cuda:0")
    spec = syn.CONFIGS["tiny"]
    opt = OptimizationParams()
    models = []
    for fused in (True, False):
        m = syn.make_model(spec, dev)
        m.training_setup(opt, fused=fused)
        _densify_sequence(m, opt, dev)  # two ordinary steps with seeded gradients
        models.append(m)
    mf, mt = models

    def set_grads(seed, skip=()):
        gd = torch.Generator().manual_seed(seed)
        for gf, gt in zip(mf.optimizer.param_groups, mt.optimizer.param_groups):
            assert gf["name"] == gt["name"]
            pf, pt = gf["params"][0], gt["params"][0]
            g = (torch.randn(pt.shape, generator=gd) * 1e-3).to(dev)
            pf.grad.copy_(g)
            pt.grad = None if gt["name"] in skip else g.clone()

    def compare(what):
        off = 0
        for gf, gt in zip(mf.optimizer.param_groups, mt.optimizer.param_groups):
            pf, pt = gf["params"][0], gt["params"][0]
            k = pf.numel()
            st = mt.optimizer.state[pt]
            np.testing.assert_allclose(pf.detach().cpu().numpy(), pt.detach().cpu().numpy(), rtol=3e-6, atol=1e-7,
                                       err_msg="%s param %s" % (what, gf["name"]))
            np.testing.assert_allclose(mf.optimizer.exp_avg[off:off + k].cpu().numpy().reshape(pt.shape),
                                       st["exp_avg"].cpu().numpy(), rtol=3e-6, atol=1e-9,
                                       err_msg="%s exp_avg %s" % (what, gf["name"]))
            off += k

    # iteration with an opacity reset only: every group has this iteration's gradient; the opacity parameter is replaced
    set_grads(1)
    with torch.no_grad():
        mf.reset_opacity()
        mt.reset_opacity()
    assert mt._opacity.grad is None                      # the reference semantics this test is about
    op_f, op_t = mf._opacity.detach().clone(), mt._opacity.detach().clone()
    mf.optimizer.step(zero_grad=True)
    mt.optimizer.step()
    mt.optimizer.zero_grad(set_to_none=True)
    assert torch.equal(mf._opacity.detach(), op_f) and torch.equal(mt._opacity.detach(), op_t)  # passed by
    compare("after reset_opacity")
    # two ordinary iterations: the opacity group's own step count now lags the others by one
    for seed in (2, 3):
        set_grads(seed)
        mf.optimizer.step(zero_grad=True)
        mt.optimizer.step()
        compare("ordinary step %d" % seed)
    # iteration with densify + prune: every parameter is new, the step is a no-op on both
    set_grads(4)
    for m in (mf, mt):
        m.xyz_gradient_accum = (torch.rand(m.get_xyz.shape[0], 1, generator=torch.Generator().manual_seed(7)) * 1.2e-3).to(dev)
        m.denom = torch.ones(m.get_xyz.shape[0], 1, device=dev)
        m.max_radii2D = torch.zeros(m.get_xyz.shape[0], device=dev)
        torch.manual_seed(991)
        m.densify_and_prune(opt.densify_grad_threshold, 0.005, 2.5, None)
    assert mf.get_xyz.shape[0] == mt.get_xyz.shape[0] != spec.P
    before = mf.optimizer.flat_param.detach().clone()
    step_before = int(mf.optimizer.state_dev[0])
    mf.optimizer.step(zero_grad=True)
    mt.optimizer.step()
    assert torch.equal(mf.optimizer.flat_param, before)
    assert int(mf.optimizer.state_dev[0]) == step_before + 1 and int(mf.optimizer.state_dev[2]) == 1
    compare("after densify_and_prune")
    set_grads(5)
    mf.optimizer.step(zero_grad=True)
    mt.optimizer.step()
    compare("first step on the densified model")


def test_optimizer_checkpoints_interchange_with_torch_adam():
    """The reference checkpoints ``optimizer.state_dict()`` inside GaussianModel.capture() and restores it with
    load_state_dict (src/scene/gaussian_model.py:84-127): a torch.optim.Adam checkpoint must restore into FusedAdam and
    a FusedAdam checkpoint into torch.optim.Adam, and training must continue identically either way."""
    from gaussianhaircut_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    shapes = [(500, 3), (500, 15, 3), (500, 1)]
    lrs = [1.6e-4, 1.25e-4, 0.05]
    init = [torch.randn(*s, generator=g) for s in shapes]

    def pair(tensors):
        pa = [torch.nn.Parameter(t.clone().to(dev)) for t in tensors]
        pb = [torch.nn.Parameter(t.clone().to(dev)) for t in tensors]
        fa = FusedAdam([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(pa, lrs))], eps=1e-15)
        tb = torch.optim.Adam([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(pb, lrs))],
                              lr=0.0, eps=1e-15)
        return pa, pb, fa, tb

    def run(pa, pb, fa, tb, steps):
        for _ in range(steps):
            for p, q, s in zip(pa, pb, shapes):
                gr = torch.randn(*s, generator=g).to(dev)
                p.grad.copy_(gr)
                q.grad = gr.clone()
            fa.step()
            tb.step()

    pa, pb, fa, tb = pair(init)
    run(pa, pb, fa, tb, 3)
    sd_f, sd_t = fa.state_dict(), tb.state_dict()
    assert set(sd_f) == set(sd_t) == {"state", "param_groups"}
    for i in sd_t["state"]:
        assert float(sd_f["state"][i]["step"]) == float(sd_t["state"][i]["step"]) == 3.0
        np.testing.assert_allclose(sd_f["state"][i]["exp_avg"].cpu().numpy(), sd_t["state"][i]["exp_avg"].cpu().numpy(),
                                   rtol=2e-6, atol=1e-7)  # lerp of O(1) gradients: 1 ulp of the summands
    # cross-load into a fresh pair that starts from the trained parameters, then keep training
    now = [p.detach().cpu() for p in pb]
    pa2, pb2, fa2, tb2 = pair(now)
    fa2.load_state_dict(sd_t)  # torch checkpoint -> FusedAdam
    tb2.load_state_dict(sd_f)  # FusedAdam checkpoint -> torch Adam
    assert int(fa2.state_dev[0]) == 3
    run(pa2, pb2, fa2, tb2, 3)
    run(pa, pb, fa, tb, 0)
    for p, q in zip(pa2, pb2):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)
    # and its own checkpoint round-trips bit for bit
    fa3 = pair(now)[2]
    fa3.load_state_dict(fa2.state_dict())
    assert torch.equal(fa3.exp_avg, fa2.exp_avg) and torch.equal(fa3.exp_avg_sq, fa2.exp_avg_sq)
    assert torch.equal(fa3.state_dev, fa2.state_dev)
