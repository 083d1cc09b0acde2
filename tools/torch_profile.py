"""torch.profiler breakdown of one measured step (GPU box): which kernels the 'step' time goes to."""
import sys
sys.path.insert(0, ".")
import torch
from torch.profiler import ProfilerActivity, profile
from gaussianhaircut_amd.parallel import FlatGradBucket
from gaussianhaircut_amd.scene.cameras import ring_cameras
from gaussianhaircut_amd.scene.gaussian_model import OptimizationParams
from gaussianhaircut_amd.trainer import make_ground_truth, training_step
from gaussianhaircut_amd.utils import synthetic as syn

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
dev = torch.device("cuda:0")
spec = syn.CONFIGS[cfg]
model = syn.make_model(spec, dev)
cams = ring_cameras(1, spec.W, spec.H, device=dev)
bg = syn.background(dev)
with torch.no_grad():
    make_ground_truth(syn.make_model(spec, dev), cams, bg)
opt = OptimizationParams()
model.training_setup(opt)
bucket = None
for i in range(3):
    training_step(model, cams, bg, opt, i + 1, bucket=bucket, global_views=1)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(3):
        training_step(model, cams, bg, opt, i + 4, bucket=bucket, global_views=1)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=70))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=70))
