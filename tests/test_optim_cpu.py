"""Host logic of gaussianhaircut_amd.optim that needs no GPU: the ZeRO-1 range plan and the argument checks."""
import pytest

from gaussianhaircut_amd.optim import FusedAdam, _zero_grad_mode


def test_shard_plan_covers_every_range_once_and_keeps_slices_aligned():
    plan = [(0, 3_000_000, "sum"), (3_000_000, 25_500_000, ("rest", 500_000, 15, 3)), (25_500_000, 30_500_123, "sum"),
            (30_500_123, 30_500_200, "none")]
    for G in (1, 2, 3, 8):
        out = FusedAdam._shard_plan(plan, G)
        # same cover, same order, nothing overlapping
        assert out[0][0] == 0 and out[-1][1] == plan[-1][1]
        assert all(a[1] == b[0] for a, b in zip(out, out[1:]))
        for a, b, how in out:
            if how == "shard":
                assert (b - a) % (G * 256) == 0 and b > a      # every rank's slice is a whole number of 1-KiB pieces
            elif how == "sum":
                assert b - a < G * 256 or G == 0                # only the tail that does not divide stays replicated
        kept = [(a, b, h) for a, b, h in out if h not in ("shard", "sum")]
        assert kept == [p for p in plan if p[2] not in ("sum",)]
        assert sum(b - a for a, b, h in out if h in ("shard", "sum")) == sum(b - a for a, b, h in plan if h == "sum")


def test_zero_grad_argument_is_validated():
    assert _zero_grad_mode(True) == (True, False) and _zero_grad_mode(False) == (False, False)
    assert _zero_grad_mode("defer") == (False, True)
    for bad in ("true", "Defer", "", "zero"):
        with pytest.raises(ValueError):
            _zero_grad_mode(bad)
