"""sha256 over the sources of K8 (k_render_bwd_cells): recorded next to the PMC traffic figure when it is taken
(profiles/pmc_k_render_bwd.json, tools/gpu/profile_r05.sh) and compared by bench.py, which prints ``"traffic_stale": true``
when the kernel has changed since (the figure is read from a committed file, not measured in the run)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ("ghr_render_bwd3.h", "ghr_render_bwd2.h")  # the kernel and its wave primitives (ghr_device.h is shared by every kernel)


def k8_source_hash() -> str:
    h = hashlib.sha256()
    for f in FILES:
        with open(os.path.join(ROOT, "gaussianhaircut_amd", "csrc", f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


if __name__ == "__main__":
    print(k8_source_hash())
