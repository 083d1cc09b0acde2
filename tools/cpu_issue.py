"""Host-side issue time of the views of a step (GPU box tool): how long Python needs to queue render + loss + backward."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gaussianhaircut_amd.trainer as tr
from gaussianhaircut_amd.scene.gaussian_model import OptimizationParams
from gaussianhaircut_amd.scene.cameras import ring_cameras
from gaussianhaircut_amd.utils import synthetic as syn
dev = torch.device("cuda", 0)
spec = syn.CONFIGS["cfg3"]
opt = OptimizationParams(); opt.lambda_dorient = 0.1
model = syn.make_model(spec, dev)
cams = ring_cameras(4, spec.W, spec.H, device=dev)
bg = syn.background(dev)
with torch.no_grad():
    gt = syn.make_model(spec, dev); tr.make_ground_truth(gt, cams, bg); del gt
model.training_setup(opt)
orig = tr._views_forward_backward
acc = []
def timed(*a, **k):
    t0 = time.perf_counter(); r = orig(*a, **k); acc.append(time.perf_counter() - t0); return r
tr._views_forward_backward = timed
for i in range(5): tr.training_step(model, cams, bg, opt, i + 1)
torch.cuda.synchronize(); acc.clear()
t0 = time.perf_counter()
for i in range(20): tr.training_step(model, cams, bg, opt, 6 + i)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / 20
print("STEP %.3f ms; host issue of 4 views %.3f ms (%.0f us/view)" % (1e3 * tot, 1e3 * sum(acc) / len(acc), 1e6 * sum(acc) / len(acc) / 4))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(10): tr.training_step(model, cams, bg, opt, 30 + i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
