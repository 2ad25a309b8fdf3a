"""The one-pass multi-value tree of the fused batch moments (gym_amd/csrc/mxv_kernels.hip: wave_sums), restated in NumPy: at lane bits 0, 1
(and 3 for eight values) a lane and its partner SPLIT the value set — each keeps the half whose index bit equals its own lane bit and adds
the partner's copy — then the remaining lane bits are plain butterfly stages.  Checks what the kernel relies on: every value's total ends
up in the lanes the store mask names, it is the sum over all 64 lanes, and the order of additions is a fixed binary tree over the lane
index (so the totals of two tiles can be compared bit for bit however the envs got there).  No device needed."""
import numpy as np
import pytest


def wave_sums(v):
    """v: [64 lanes][V values] -> per-lane result of the device function (float64 arithmetic in the same order)."""
    lanes, V = v.shape
    assert lanes == 64 and V in (4, 8)
    lane = np.arange(64)
    cur = [v[:, i].copy() for i in range(V)]

    def halve(vals, bit):
        b = (lane >> bit) & 1
        partner = lane ^ (1 << bit)
        out = []
        for p in range(len(vals) // 2):
            lo, hi = vals[2 * p], vals[2 * p + 1]
            keep = np.where(b == 1, hi, lo)
            send = np.where(b == 1, lo, hi)
            out.append(keep + send[partner])
        return out

    cur = halve(cur, 0)
    cur = halve(cur, 1)
    if V == 8:
        cur = halve(cur, 3)
        x = cur[0]
        rest = (2, 4, 5)
    else:
        x = cur[0] + cur[0][lane ^ 8]
        rest = (2, 4, 5)
    for bit in rest:
        x = x + x[lane ^ (1 << bit)]
    return x


@pytest.mark.parametrize("V", [4, 8])
def test_totals_land_in_the_lanes_the_store_mask_names(V):
    rng = np.random.default_rng(V)
    v = rng.standard_normal((64, V)) * np.exp(rng.uniform(-20, 20, (64, V)))     # wide dynamic range: order of additions matters
    x = wave_sums(v)
    lane = np.arange(64)
    idx = ((lane & 3) | ((lane >> 1) & 4)) if V == 8 else (lane & 3)
    holder = ((lane & 0x34) == 0) if V == 8 else ((lane & 0x3C) == 0)
    assert holder.sum() == V and sorted(idx[holder]) == list(range(V))           # one storing lane per value
    for i in range(V):
        same = x[idx == i]
        assert np.all(same == same[0])                                           # every lane of a value's class holds the same total
        np.testing.assert_allclose(same[0], np.sum(v[:, i].astype(np.longdouble)).astype(np.float64), rtol=1e-12)
    # a fixed tree: permuting WHICH env sits in which lane changes the rounding, running the same lanes again does not
    assert np.array_equal(wave_sums(v), x)
    # linearity in exact arithmetic: integers stay exact whatever the order
    k = rng.integers(-1000, 1000, (64, V)).astype(np.float64)
    xk = wave_sums(k)
    for i in range(V):
        assert xk[idx == i][0] == k[:, i].sum()
