"""Trajectory tensors placed by HBM class: the C ABI's mxv_placed_alloc (HIP virtual-memory API) and the product
default, DeviceRollout.trajectory_buffers(layout="sorted") (ordinary allocations classified with mxv_hbm_pair_probe).  What is pinned
here: the memory is real and private (torch sees it, values survive, freeing returns it), results do not depend on where the tensors
live (bit for bit), the reports say what happened.  That the placement makes the rollout FAST is a measurement, not a test:
profiles/r3/r3b_*, r3d_* and the bench line's roofline.write_probe."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gym_amd import _native  # noqa: E402
from gym_amd.rollout import DeviceRollout  # noqa: E402

K, N = 64, 1 << 20
SPECS = [("obs", (K, N, 4), "<f4", 0), ("reward", (K, N), "<f8", 1), ("actions", (K, N), "<i8", 1), ("terminated", (K, N), "|u1", -1),
         ("truncated", (K, N), "|u1", -1)]   # 1 + 0.5 + 0.5 + 2 x 0.0625 GiB = 2.125 GiB: above MXV_PLACED_MIN_BYTES


def test_placed_memory_is_real_private_and_returned():
    import gc

    gc.collect()
    torch.cuda.empty_cache()            # (whatever earlier tests left in the caching allocator must not be released between the two readings)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(0)
    mem = _native.PlacedMemory(0, SPECS)
    info = mem.info
    assert info["placed"] is True and info["chunks_kept"] == 4 + 2 + 2 + 1 + 1 and info["held_GiB"] == 2.5
    assert info["peak_GiB"] <= 2 * 2.5 + info["jumped_GiB"] + 0.01 and info["stop_reason"] in ("balanced", "chunk cap", "jump budget")
    assert info["balanced"] == (info["stop_reason"] == "balanced") and 3.0 < info["same_class_us"] < 6.0
    t = mem.tensors()
    assert {k: tuple(v.shape) for k, v in t.items()} == {name: shape for name, shape, _, _ in SPECS}
    assert t["obs"].dtype == torch.float32 and t["reward"].dtype == torch.float64 and t["actions"].dtype == torch.int64
    assert all(v.is_cuda and v.is_contiguous() for v in t.values())
    ptrs = sorted((v.data_ptr(), v.numel() * v.element_size()) for v in t.values())
    assert all(a + n <= b for (a, n), (b, _) in zip(ptrs, ptrs[1:]))          # disjoint virtual ranges
    for i, v in enumerate(t.values()):                                         # every page is backed and private to its tensor
        v.fill_(i + 1)
    torch.cuda.synchronize()
    for i, (k, v) in enumerate(t.items()):
        flat = v.view(-1)
        assert float(flat[0]) == i + 1 and float(flat[-1]) == i + 1 and float(flat[flat.numel() // 2]) == i + 1, k
        assert int((flat[:: 4099] != i + 1).sum()) == 0, k
    free1, _ = torch.cuda.mem_get_info(0)
    assert free0 - free1 >= int(2.4 * 2**30)
    del t, v, flat
    mem.close()
    mem.close()                                                                # idempotent
    free2, _ = torch.cuda.mem_get_info(0)
    assert free2 - free1 >= int(2.4 * 2**30)                                   # physical memory is back (the virtual ranges are not reused)


def test_small_or_one_sided_sets_take_ordinary_allocations():
    small = _native.PlacedMemory(0, [("obs", (4, 1024, 4), "<f4", 0), ("reward", (4, 1024), "<f8", 1)])
    assert small.info["placed"] is False and small.info["requested_GiB"] < 0.01
    t = small.tensors()
    t["obs"].fill_(2.0)
    assert float(t["obs"].sum()) == 2.0 * 4 * 1024 * 4
    one_sided = _native.PlacedMemory(0, [("obs", (K, N, 4), "<f4", 0), ("more", (K, N, 4), "<f4", 0), ("flags", (K, N), "|u1", -1)])
    assert one_sided.info["placed"] is False       # nothing to keep apart
    with pytest.raises(_native.MxvError):
        _native.PlacedMemory(0, [("obs", (4,), "<f4", 7)])                      # group must be -1, 0 or 1


def test_pair_probe_is_a_sane_measurement():
    a = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    b = torch.empty(128 << 20, dtype=torch.uint8, device="cuda")
    us = [_native.hbm_pair_probe(0, a.data_ptr(), b.data_ptr()) for _ in range(3)]
    same = _native.hbm_pair_probe(0, a.data_ptr(), a.data_ptr() + (256 << 20))
    assert all(2.5 < x < 8.0 for x in us + [same])      # 24 B x 2^20 lanes per step: 3.9-4.4 us on every box seen
    assert max(us) / min(us) < 1.15


@pytest.mark.parametrize("layout", ["placed", "sorted"])
def test_rollout_results_do_not_depend_on_where_the_trajectory_tensors_live(layout):
    n, k = 1 << 18, 256                                # 2^18 envs x 256 steps x 34 B = 2.1 GiB: both layouts really place
    a = DeviceRollout("CartPole-v1", n, seed=3, action_seed=4)
    b = DeviceRollout("CartPole-v1", n, seed=3, action_seed=4)
    a.reset(seed=3)
    b.reset(seed=3)
    ta = a.trajectory_buffers(k, layout=layout)
    rep = a.last_placement
    tb = b.trajectory_buffers(k, layout="separate")
    if layout == "placed":
        assert rep["placed"] is True and rep["chunks_kept"] >= 8
    else:
        assert rep["kind"] == "sorted" and "balanced" in rep and rep["parked_GiB"] <= 112
    for _ in range(2):
        a.rollout_per_step(k, out=ta)
        b.rollout_per_step(k, out=tb)
    a.synchronize()
    b.synchronize()
    for key in tb:
        assert torch.equal(ta[key], tb[key]), key
    assert np.array_equal(a.handle.get_state()[0], b.handle.get_state()[0])
    a.close()
    b.close()


def test_default_trajectory_buffers_sort_large_sets_and_leave_small_ones_alone():
    r = DeviceRollout("CartPole-v1", 1 << 18, seed=0, action_seed=1)
    r.reset(seed=0)
    small = r.trajectory_buffers(16)
    assert not hasattr(r, "last_placement")
    big = r.rollout_per_step(256)                      # out=None: what a user gets
    assert r.last_placement["kind"] == "sorted" and big["obs"].shape == (256, 1 << 18, 4)
    del small, big
    r.close()


def test_tabular_rollout_sorts_its_four_streams_two_and_two():
    """TabularRollout.trajectory_buffers: obs + reward on one HBM class, actions + prob on another for sets of 2 GiB and more; same bits."""
    from gym_amd.toy_text import TabularRollout

    n, k = 1 << 19, 128                                 # 34 B x 2^19 x 128 = 2.1 GiB
    a = TabularRollout("FrozenLake-v1", n, seed=2, action_seed=3)
    b = TabularRollout("FrozenLake-v1", n, seed=2, action_seed=3)
    a.reset(seed=2)
    b.reset(seed=2)
    ta = a.trajectory_buffers(k)
    assert a.last_placement["kind"] == "sorted" and "balanced" in a.last_placement
    tb = b.trajectory_buffers(k, layout="separate")
    a.rollout_per_step(k, out=ta)
    b.rollout_per_step(k, out=tb)
    a.synchronize()
    b.synchronize()
    for key in tb:
        assert torch.equal(ta[key], tb[key]), key
    with pytest.raises(ValueError):
        a.trajectory_buffers(k, layout="nowhere")
    a.close()
    b.close()


def test_sorted_tensors_degrades_to_ordinary_allocations_when_there_is_nothing_to_sort():
    from gym_amd.placement import sorted_tensors

    dev = torch.device("cuda", 0)
    out, rep = sorted_tensors([("obs", (8, 1024, 4), torch.float32, False), ("reward", (8, 1024), torch.float64, True)], {"obs": 0, "reward": 1}, dev)
    assert rep["balanced"] is False and "too small" in rep["note"] and float(out["reward"].sum()) == 0.0
    out, rep = sorted_tensors([("obs", (8, 1024, 4), torch.float32, False)], {"obs": 0}, dev)
    assert "nothing to keep apart" in rep["note"]


def test_a_learners_loop_pays_for_the_placement_of_its_first_two_sets_only(monkeypatch):
    """`out = r.rollout_per_step(K)` without out=: the set of call i is released when call i + 1 has returned, so torch's caching allocator
    alternates between two sets of blocks; gym_amd.placement remembers what it measured about them (block address + size, valid until a
    segment goes back to the driver) and from the third call on launches no probe.  Also: the 2^17-env shard of an 8-GPU strong-scaling
    job (1.06 GiB of trajectory tensors) is sorted."""
    import time

    monkeypatch.setenv("MXV_PLACEMENT", "search")      # (the memo is what is tested: let the first set walk as far as it needs to be balanced)
    torch.cuda.empty_cache()       # whatever earlier tests left in the allocator's cache would be split for these requests
    r = DeviceRollout("CartPole-v1", 1 << 17, seed=0, action_seed=1)
    r.reset(seed=0)
    out, reports, walls = None, [], []
    for _ in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = r.rollout_per_step(256)
        r.synchronize()
        walls.append(time.perf_counter() - t0)
        reports.append(dict(r.last_placement))
    assert all(rep["kind"] == "sorted" and rep["balanced"] for rep in reports), reports
    assert reports[0]["remembered"] == 0
    assert all(rep["remembered"] == 3 and rep["parked_GiB"] == 0 for rep in reports[-3:]), reports     # anchor + reward + actions
    assert max(walls[-3:]) < 0.25 * walls[0], walls
    # a flush by anyone drops the memo: the next set is measured again (and is still balanced)
    del out
    torch.cuda.empty_cache()
    out = r.rollout_per_step(256)
    assert r.last_placement["remembered"] == 0 and r.last_placement["balanced"]
    del out
    r.close()
