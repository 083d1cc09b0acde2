"""Build-time guard for the hand-counted `s_waitcnt vmcnt(N)` of k_render_bwd_cells (csrc/ghr_render_bwd3.h).

The kernel issues its record gathers as LDS-DMA two chunks ahead and waits for them with immediates that count the vector
memory operations issued since -- which is only right if the compiler adds none of its own.  A register spill is a scratch
access, scratch counts in vmcnt, and a spill inside the chunk loop lets LDS reads overtake their gathers: wrong gradients
and a "faster" kernel (seen in a round-3 experiment build).  So: the kernel must compile without scratch and without
spills at the occupancy it is tuned for.  Cross-compiles the device code to assembly (no GPU needed) and reads the
kernel descriptors."""
import os
import re
import subprocess
import tempfile

import pytest

from gaussianhaircut_amd import _lib


import functools


@functools.lru_cache(maxsize=1)
def _descriptors():
    hipcc = _lib._hipcc()
    if hipcc is None:
        pytest.skip("hipcc not found")
    flags = [f for f in _lib.HIPCC_FLAGS if f not in ("-fPIC", "-shared")]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run([hipcc] + flags + ["-I" + os.path.join(os.path.dirname(_lib.CSRC), "..", "include"), "-S",
                                          "--cuda-device-only", "-o", out, os.path.join(_lib.CSRC, "ghr_capi.hip")],
                       check=True, capture_output=True)
        txt = open(out).read()
    meta = {}
    for blk in txt.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if name:
            meta[name.group(1)] = {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)\s*\n", blk)}
    return meta


def test_gradient_walk_compiles_without_scratch_and_within_its_register_budget():
    meta = _descriptors()
    k8 = [v for k, v in meta.items() if "k_render_bwd_cells" in k]
    assert len(k8) == 1, list(meta)
    k8 = k8[0]
    assert k8["private_segment_fixed_size"] == 0 and k8["vgpr_spill_count"] == 0 and k8["sgpr_spill_count"] == 0, k8
    assert k8["vgpr_count"] <= 96, k8          # five waves per SIMD (512 / 5, granule 8)
    assert k8["group_segment_fixed_size"] <= 32768, k8   # five workgroups per CU of 160 KiB LDS
    k7 = [v for k, v in meta.items() if "k_render_fwd" in k][0]
    assert k7["private_segment_fixed_size"] == 0 and k7["vgpr_count"] <= 80, k7   # six waves per SIMD


def test_marching_loss_and_adam_kernels_stay_within_their_occupancy_steps():
    """The marching loss kernels (csrc/ghr_loss.h) are one wave per workgroup with 12-14 KB of LDS and live on the number
    of waves in flight; their loops must not spill (a spill is a scratch access inside a loop whose prefetched batch the
    compiler's waits count).  With the SLP vectoriser's packed FMAs they once spilled 50-80 SCALAR registers into VGPR
    lanes: the window FMAs are single v_fmac_f32 with the weight in an SGPR since -- a handful of scalar spills in the
    prologue are tolerated, no vector ones."""
    meta = _descriptors()

    def one(sub):
        ks = [v for k, v in meta.items() if sub in k]
        assert len(ks) == 1, (sub, [k for k in meta if sub in k])
        return ks[0]

    fwd, bwd, adam = one("k_loss_fwd_cached_v"), one("k_loss_bwd_v"), one("k_adam_v4")
    for k in (fwd, bwd, adam):
        assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0, k
    assert fwd["sgpr_spill_count"] <= 32 and bwd["sgpr_spill_count"] <= 32, (fwd, bwd)
    assert fwd["vgpr_count"] <= 128, fwd                     # four waves per SIMD
    assert bwd["vgpr_count"] <= 168, bwd                     # three
    assert fwd["group_segment_fixed_size"] <= 12288, fwd     # 13 waves per CU
    assert bwd["group_segment_fixed_size"] <= 14336, bwd     # 11
    assert adam["vgpr_count"] <= 64, adam


def test_round6_kernels_compile_without_scratch():
    """k_strand_build / k_strand_build_bwd (dynamic LDS only), k_sh_grad_from_views (48 accumulators + 16 basis values per thread,
    11.5 KB of LDS) and the four k_project_bwd variants: no scratch, no vector spills."""
    meta = _descriptors()
    for sub, vgprs in (("k_strand_build", 128), ("k_strand_build_bwd", 128), ("k_sh_grad_from_views", 168)):
        ks = [v for k, v in meta.items() if sub in k and (sub != "k_strand_build" or "bwd" not in k)]
        assert len(ks) == 1, (sub, [k for k in meta if sub in k])
        k = ks[0]
        assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0 and k["vgpr_count"] <= vgprs, (sub, k)
    pb = [v for k, v in meta.items() if "k_project_bwd" in k]
    assert len(pb) == 4, [k for k in meta if "k_project_bwd" in k]
    for k in pb:
        assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0 and k["vgpr_count"] <= 168, k   # three waves per SIMD


def test_no_kernel_of_the_library_uses_scratch():
    """Every kernel of libghr_hip.so keeps its arrays in registers / LDS: a private segment means an array the compiler could
    not promote (round 6 found one in both camera-gradient instantiations of k_project_bwd: two stores sunk into one with a
    selected address) or a vector spill -- on this part a scratch access also counts in vmcnt, which breaks hand-counted waits."""
    meta = _descriptors()
    assert len(meta) >= 30
    bad = {k: v["private_segment_fixed_size"] for k, v in meta.items()
           if v.get("private_segment_fixed_size", 0) or v.get("vgpr_spill_count", 0)}
    assert not bad, bad
