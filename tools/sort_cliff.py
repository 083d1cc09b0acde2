"""How expensive is one tile whose list outgrows the LDS sort?  N small Gaussians in the middle of a
64x64 image (4 tiles share them), N swept across GHR_SORT_CAP.  Run under ``rocprofv3 --kernel-trace``; with
``--parse <dir>`` reads the trace back and prints the k_tile_sort / k_render_fwd / k_render_bwd durations per N.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/sc -- python tools/sort_cliff.py
    python tools/sort_cliff.py --parse /tmp/sc
"""
import csv
import glob
import os
import sys

NS = [1024, 2048, 2560, 4096, 8192, 16384, 32768, 65536]
REPS = 3


def parse(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = {}
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].split("::")[-1]
        per.setdefault(name, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("%8s %16s %14s %14s %14s   (us, min of %d)" % ("N", "k_tile_sort_big", "k_tile_sort", "k_render_fwd",
                                                          "k_render_bwd", REPS))
    for i, n in enumerate(NS):
        vals = []
        for k in ("k_tile_sort_big", "k_tile_sort", "k_render_fwd", "k_render_bwd"):
            key = [x for x in per if x == k or (k.startswith("k_render") and x.startswith(k))]
            v = sum((per[x] for x in key), [])
            vals.append(min(v[i * REPS:(i + 1) * REPS]) if len(v) >= (i + 1) * REPS else float("nan"))
        print("%8d %16.1f %14.1f %14.1f %14.1f" % (n, *vals))


def main():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tests.gpu_helpers import GpuRun, to_dev
    from tests.test_gpu_parity import _manual_inputs
    dev = torch.device("cuda:0")
    for n in NS:
        g = torch.Generator().manual_seed(n)
        xyz = torch.zeros(n, 3)
        xyz[:, :2] = (torch.rand(n, 2, generator=g) - 0.5) * 0.05
        xyz[:, 2] = torch.rand(n, generator=g) * 0.2
        ri = to_dev(_manual_inputs(dev, xyz, torch.full((n, 3), 0.004), torch.full((n,), 0.02), W=64, H=64), dev)
        for _ in range(REPS):
            run = GpuRun(ri, "B_sr", debug=False)
            run.backward(torch.ones(10, 64, 64))
        torch.cuda.synchronize()
        print("N %d R %d" % (n, run.R), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--parse":
        parse(sys.argv[2])
    else:
        main()
