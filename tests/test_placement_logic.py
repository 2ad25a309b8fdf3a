"""gym_amd.placement.sorted_tensors — the search logic on a SIMULATED device (no GPU): allocations walk through an address space made of
class regions, the pair probe answers "slow" for two addresses of one class and "fast" otherwise, exactly what mxv_hbm_pair_probe
measures on the MI355X (profiles/r3/r3a_vmm_probe7_chunk_matrix_and_prediction.jsonl).  Every branch is driven: nothing to do, a fresh
device (everything in one class for 90 GiB), a boundary inside the anchor, a boundary inside group 0 (restart), the budget running out
(best effort, balanced = False), scrambled free lists."""
import bisect

import pytest
import torch

from gym_amd.placement import NARROW, WIDE, sorted_tensors

GiB = 1 << 30


class SimDevice:
    """Addresses are handed out in increasing order; `regions` = [(end_address, class)] in increasing order of end_address."""

    def __init__(self, regions, free=288 * GiB, noise=0.0):
        self.ends = [e for e, _ in regions]
        self.classes = [c for _, c in regions]
        self.cursor, self.free, self.live, self.peak, self.noise = 0, free, 0, 0, noise
        self.probes = 0

    def klass(self, addr):
        return self.classes[min(bisect.bisect_right(self.ends, addr), len(self.classes) - 1)]

    class T:
        def __init__(self, dev, addr, n):
            self.dev, self.addr, self.n = dev, addr, n
            dev.live += n
            dev.peak = max(dev.peak, dev.live)

        def __del__(self):
            self.dev.live -= self.n

    def alloc(self, shape, dtype, zero):
        n = 1
        for s in shape:
            n *= s
        n *= torch.empty((), dtype=dtype).element_size()
        if self.live + n > self.free:
            raise MemoryError("simulated device out of memory")
        t = SimDevice.T(self, self.cursor, n)
        self.cursor += n
        return t

    def ptr(self, t):
        return t.addr

    def sync(self):
        pass

    def free_bytes(self):
        return self.free - self.live

    def probe(self, wide, narrow):
        self.probes += 1
        # a window that straddles a boundary behaves like the class of its majority
        cw, cn = self.klass(wide + WIDE // 2), self.klass(narrow + NARROW // 2)
        return (4.25 if cw == cn else 3.88) * (1.0 + self.noise * ((self.probes * 2654435761) % 1000 / 1000.0 - 0.5))

    def release(self):
        pass


K, N = 256, 1 << 20
CARTPOLE = [("obs", (K, N, 4), torch.float32, False), ("reward", (K, N), torch.float64, False), ("terminated", (K, N), torch.uint8, False),
            ("truncated", (K, N), torch.uint8, False), ("actions", (K, N), torch.int64, False)]
CP_GROUPS = {"obs": 0, "reward": 1, "actions": 1}
TAB = [(n, (128, N), torch.int64, False) for n in ("obs", "reward", "actions", "prob")] + [("terminated", (128, N), torch.uint8, True)]
TAB_GROUPS = {"obs": 0, "reward": 0, "actions": 1, "prob": 1}


def _classes(dev, out, specs):
    res = {}
    for name, shape, dt, _ in specs:
        n = out[name].n
        res[name] = {dev.klass(out[name].addr + 1), dev.klass(out[name].addr + n - 1)}
    return res


@pytest.fixture(autouse=True)
def _search_mode(monkeypatch):
    """These tests drive the walk itself: MXV_PLACEMENT=search (up to 112 GiB parked).  The default mode — at most 8 GiB — has its own
    tests at the end of the file."""
    monkeypatch.setenv("MXV_PLACEMENT", "search")
    monkeypatch.delenv("MXV_PLACEMENT_MAX_PARK_GIB", raising=False)


def test_fresh_device_walks_to_the_next_class_and_releases_what_it_parked():
    dev = SimDevice([(91 * GiB, "A"), (187 * GiB, "B"), (288 * GiB, "C")])
    out, rep = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=dev)
    c = _classes(dev, out, CARTPOLE)
    assert rep["balanced"] is True and c["obs"] == {"A"} and c["reward"] == {"B"} and c["actions"] == {"B"}
    assert 80 <= rep["parked_GiB"] <= 92 and rep["candidates"] >= 40
    assert dev.live == sum(t.n for t in out.values())            # parked tensors are gone
    assert dev.peak <= 112 * GiB + 9 * GiB
    assert list(out) == [n for n, *_ in CARTPOLE]


def test_scrambled_free_lists_need_little():
    regions, addr = [], 0
    for i in range(200):                                         # runs of 6 GiB, alternating classes
        addr += 6 * GiB
        regions.append((addr, "AB"[i % 2]))
    dev = SimDevice(regions)
    out, rep = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=dev)
    c = _classes(dev, out, CARTPOLE)
    assert rep["balanced"] is True and len(c["obs"]) == 1 and c["reward"].isdisjoint(c["obs"]) and c["actions"].isdisjoint(c["obs"])
    assert rep["parked_GiB"] <= 12


def test_an_anchor_that_straddles_a_boundary_is_replaced():
    dev = SimDevice([(2 * GiB, "A"), (60 * GiB, "B"), (288 * GiB, "A")])     # the first 4-GiB observation tensor is half A, half B
    out, rep = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=dev)
    c = _classes(dev, out, CARTPOLE)
    assert rep["balanced"] is True and c["obs"] == {"B"} and c["reward"] == {"A"} and c["actions"] == {"A"}
    assert out["obs"].addr == 4 * GiB


def test_a_boundary_inside_group_zero_restarts_the_group():
    # tabular set: obs + reward (1 GiB each) must share a class; here the boundary falls between them
    dev = SimDevice([(1 * GiB, "A"), (40 * GiB, "B"), (288 * GiB, "A")])
    out, rep = sorted_tensors(TAB, TAB_GROUPS, None, _backend=dev)
    c = _classes(dev, out, TAB)
    assert rep["balanced"] is True and c["obs"] == c["reward"] == {"B"} and c["actions"] == c["prob"] == {"A"}
    assert rep["parked_GiB"] >= 1.0


def test_the_budget_bounds_what_is_parked_and_the_report_says_so():
    dev = SimDevice([(288 * GiB, "A")])                          # one class only, as far as the budget reaches
    out, rep = sorted_tensors(CARTPOLE, CP_GROUPS, None, budget_bytes=20 * GiB, _backend=dev)
    assert rep["balanced"] is False and rep["parked_GiB"] <= 20.0
    assert dev.peak <= 20 * GiB + 11 * GiB and set(out) == {n for n, *_ in CARTPOLE}
    assert dev.live == sum(t.n for t in out.values())


def test_little_free_memory_means_a_small_budget():
    dev = SimDevice([(288 * GiB, "A")], free=30 * GiB)
    out, rep = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=dev)
    assert rep["balanced"] is False and dev.peak <= 30 * GiB and rep["parked_GiB"] <= 15.0


def test_nothing_to_sort():
    dev = SimDevice([(288 * GiB, "A")])
    out, rep = sorted_tensors([("obs", (8, 64, 4), torch.float32, False), ("reward", (8, 64), torch.float64, False)], {"obs": 0, "reward": 1}, None,
                              _backend=dev)
    assert "too small" in rep["note"] and dev.probes == 0
    out, rep = sorted_tensors(CARTPOLE, {"obs": 0}, None, _backend=dev)
    assert "nothing to keep apart" in rep["note"] and dev.probes == 0


def test_measurement_noise_of_two_percent_does_not_flip_a_classification():
    dev = SimDevice([(91 * GiB, "A"), (288 * GiB, "B")], noise=0.04)           # +-2 % on every probe: the gap between the modes is 9 %
    out, rep = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=dev)
    c = _classes(dev, out, CARTPOLE)
    assert rep["balanced"] is True and c["obs"] == {"A"} and c["reward"] == {"B"} and c["actions"] == {"B"}


class CachingSim(SimDevice):
    """SimDevice + the behaviour of torch's caching allocator that the memo relies on: a freed block is handed out again for a request
    of exactly its size; release() (= empty_cache) returns every unused block to the driver and moves the segment-free count."""

    _instances = 0

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.cache, self.frees = [], 0
        CachingSim._instances += 1
        self._key = ("simulated device", CachingSim._instances)      # never reused (id() of a dead simulator may be)

    class T(SimDevice.T):
        def __del__(self):
            self.dev.live -= self.n
            self.dev.cache.append((self.addr, self.n))

    def alloc(self, shape, dtype, zero):
        n = 1
        for s in shape:
            n *= s
        n *= torch.empty((), dtype=dtype).element_size()
        for i, (addr, m) in enumerate(self.cache):
            if m == n:
                del self.cache[i]
                return CachingSim.T(self, addr, n)
        t = CachingSim.T(self, self.cursor, n)
        self.cursor += n
        return t

    def release(self):
        self.frees += len(self.cache)
        self.cache.clear()

    def segment_frees(self):
        return self.frees

    def key(self):
        return self._key


def test_blocks_the_allocator_hands_out_again_are_not_measured_again():
    """`out = r.rollout_per_step(K)` in a loop: the learner alternates between two sets of blocks; from the third call on nothing is probed."""
    dev = CachingSim([(6 * GiB, "A"), (40 * GiB, "B"), (80 * GiB, "A"), (288 * GiB, "B")])
    first, rep1 = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=dev)
    assert rep1["balanced"] and rep1["remembered"] == 0 and dev.probes > 0
    second, rep2 = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=dev)      # the first set is still alive: new blocks, new measurements
    assert rep2["balanced"] and {t.addr for t in first.values()}.isdisjoint({t.addr for t in second.values()})
    calls = []
    for _ in range(4):                                                          # the loop proper: the older set is dropped, then a new one asked for
        first = None
        before = dev.probes
        first, rep = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=dev)
        first, second = second, first
        calls.append((dev.probes - before, rep["remembered"], rep["balanced"], rep["parked_GiB"]))
    assert all(b for _, _, b, _ in calls)
    assert calls[-1][0] == 0 and calls[-2][0] == 0 and calls[-1][1] == 3, calls   # anchor + reward + actions remembered, no probe launch
    c = _classes(dev, second, CARTPOLE)
    assert len(c["obs"]) == 1 and c["reward"].isdisjoint(c["obs"]) and c["actions"].isdisjoint(c["obs"])


def test_a_cache_flush_by_anyone_drops_what_was_remembered():
    dev = CachingSim([(6 * GiB, "A"), (40 * GiB, "B"), (80 * GiB, "A"), (288 * GiB, "B")])
    a, _ = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=dev)
    b, _ = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=dev)
    a = None
    a, rep = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=dev)
    b = None
    dev.release()                      # torch.cuda.empty_cache() by the user: the blocks of `b` go back to the driver
    before = dev.probes
    b, rep = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=dev)
    assert dev.probes > before and rep["balanced"]


class RemappingSim(CachingSim):
    """CachingSim + what makes a stale memo dangerous on the real device: once a block has gone back to the driver (release()), a later
    allocation may get the SAME virtual address over DIFFERENT physical memory.  Classes belong to physical addresses, which only ever
    grow here; virtual addresses of released blocks are handed out again for requests of the same size."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.pcur, self.va_free, self.blocks = 0, [], {}          # blocks: va -> (bytes, physical start)

    def klass(self, addr):
        for va, (n, phys) in self.blocks.items():
            if va <= addr < va + n:
                return SimDevice.klass(self, phys + (addr - va))
        raise AssertionError(f"address {addr} is not mapped")

    def alloc(self, shape, dtype, zero):
        n = 1
        for s in shape:
            n *= s
        n *= torch.empty((), dtype=dtype).element_size()
        for i, (addr, m) in enumerate(self.cache):                # the allocator's own cache first: same block, same memory
            if m == n:
                del self.cache[i]
                return CachingSim.T(self, addr, n)
        va = next((v for v, m in self.va_free if m == n), None)   # the driver re-uses a virtual range it got back ...
        if va is None:
            va = self.cursor
            self.cursor += n
        else:
            self.va_free.remove((va, n))
        self.blocks[va] = (n, self.pcur)                          # ... over new physical memory
        self.pcur += n
        return CachingSim.T(self, va, n)

    def release(self):
        self.frees += len(self.cache)
        for addr, m in self.cache:
            del self.blocks[addr]
            self.va_free.append((addr, m))
        self.cache.clear()


def test_memo_never_claims_a_balance_the_memory_does_not_have():
    """Random learner behaviour on the caching simulated allocator — sets kept, dropped, the cache flushed at random moments, class regions
    of random sizes — and one invariant: whenever sorted_tensors reports balanced = True, the tensors it returned really lie the way it
    says (checked against the simulator's ground truth), whether their classification was measured in this call or remembered."""
    import random

    rnd = random.Random(20260923)
    for scenario in range(60):
        regions, addr = [], 0
        while addr < 288 * GiB:
            addr += rnd.choice([3, 6, 11, 24, 48, 96]) * GiB
            regions.append((addr, "ABC"[len(regions) % 3] if rnd.random() < 0.5 else rnd.choice("AB")))
        dev = RemappingSim(regions)
        held = []
        for call in range(12):
            action = rnd.random()
            if held and action < 0.45:
                held.pop(rnd.randrange(len(held)))               # the learner drops a set: its blocks go to the allocator's cache
            if action > 0.9:
                dev.release()                                    # somebody calls empty_cache()
            try:
                out, rep = sorted_tensors(CARTPOLE, CP_GROUPS, None, budget_bytes=60 * GiB, _backend=dev)
            except MemoryError:
                held.clear()
                continue
            c = _classes(dev, out, CARTPOLE)
            if rep["balanced"]:
                assert len(c["obs"]) == 1, (scenario, call, c, rep)
                assert c["reward"].isdisjoint(c["obs"]) and c["actions"].isdisjoint(c["obs"]), (scenario, call, c, rep)
            held.append(out)
            if len(held) > 3:
                held.pop(0)


# ---- round 4: the search is safe beside a co-resident learner (VERDICT r3 engineering #10) ---------------------------------
def test_out_of_memory_inside_the_search_falls_back_to_ordinary_allocations():
    """Another process holds most of the device (or took it while the search was parking): the search must end with ordinary
    allocations and balanced = False, never with an exception — only a set that does not fit at all may raise."""
    set_bytes = sum(torch.empty((), dtype=dt).element_size() * K * N * (4 if name == "obs" else 1) for name, _, dt, _ in CARTPOLE)

    class Shrinking(SimDevice):        # a neighbour grabs memory after the third allocation: free space collapses to the set + 1 GiB
        def alloc(self, shape, dtype, zero):
            if self.cursor > 12 * GiB and not getattr(self, "robbed", False):
                self.robbed = True
                self.free = self.live + 1 * GiB
            return super().alloc(shape, dtype, zero)

        def release(self):
            self.released = True

    dev = Shrinking([(91 * GiB, "A"), (187 * GiB, "B"), (288 * GiB, "C")])
    dev.free = 288 * GiB
    out, rep = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=dev)
    assert rep["balanced"] is False and "out of device memory" in rep["note"] and getattr(dev, "released", False)
    assert dev.live == set_bytes == sum(t.n for t in out.values()) and list(out) == [n for n, *_ in CARTPOLE]
    # a device that cannot even hold the set: that error is the caller's, as it would be without placement
    tiny = SimDevice([(288 * GiB, "A")], free=4 * GiB)
    with pytest.raises(MemoryError):
        sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=tiny)


def test_the_budget_counts_the_set_itself_and_obeys_the_cap(monkeypatch):
    """200 GiB of the device belong to someone else: the search may park half of what is left AFTER the 9-GiB set, not half of what is free
    now; MXV_PLACEMENT_MAX_PARK_GIB lowers the cap; peak memory stays inside both."""
    dev = SimDevice([(291 * GiB, "A"), (387 * GiB, "B")], free=88 * GiB)      # 88 GiB free, one class for all of it
    out, rep = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=dev)
    assert rep["balanced"] is False and rep["budget_GiB"] == round((88 - 8.5) / 2, 2)
    assert dev.peak <= 88 * GiB and dev.live == sum(t.n for t in out.values())
    monkeypatch.setenv("MXV_PLACEMENT_MAX_PARK_GIB", "4")
    dev = SimDevice([(91 * GiB, "A"), (187 * GiB, "B"), (288 * GiB, "C")])
    out, rep = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=dev)
    assert rep["budget_GiB"] == 4.0 and rep["parked_GiB"] <= 4.0 and rep["balanced"] is False
    assert dev.peak <= sum(t.n for t in out.values()) + 4 * GiB + 3 * GiB     # the set + the cap + the candidate in hand


def test_a_straddling_anchor_without_budget_is_not_reported_as_balanced():
    dev = SimDevice([(2 * GiB, "A"), (60 * GiB, "B"), (288 * GiB, "A")])     # the first observation tensor is half A, half B
    out, rep = sorted_tensors(CARTPOLE, CP_GROUPS, None, budget_bytes=0, _backend=dev)
    assert rep["balanced"] is False and rep["parked_GiB"] == 0.0


def test_a_set_of_remembered_blocks_costs_no_probe_and_no_synchronisation():
    """The learner's loop: `out = r.rollout_per_step(K)` gets the same allocator blocks again; the second call must neither launch a probe
    nor synchronise the device (three device-wide synchronisations per call even when the memo hits every block)."""
    class Recycling(SimDevice):
        syncs = 0

        def alloc(self, shape, dtype, zero):      # a caching allocator: the same request gets the same block once it is free again
            n = torch.empty((), dtype=dtype).element_size()
            for s in shape:
                n *= s
            pool = self.__dict__.setdefault("pool", {})
            if pool.get(n):
                return SimDevice.T(self, pool[n].pop(0), n)
            return super().alloc(shape, dtype, zero)

        def give_back(self, tensors):
            for t in tensors:
                self.pool.setdefault(t.n, []).append(t.addr)

        def sync(self):
            Recycling.syncs += 1

        def segment_frees(self):
            return 0

        def key(self):
            return "recycling-sim"

    regions, addr = [], 0
    for i in range(100):
        addr += 6 * GiB
        regions.append((addr, "AB"[i % 2]))
    dev = Recycling(regions)
    out, rep = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=dev)
    assert rep["balanced"] is True and dev.probes > 0 and Recycling.syncs > 0
    first = {k: t.addr for k, t in out.items()}
    dev.give_back(out.values())
    del out
    probes, syncs = dev.probes, Recycling.syncs
    out2, rep2 = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=dev)
    assert {k: t.addr for k, t in out2.items()} == first
    assert rep2["balanced"] is True and rep2["remembered"] >= 3
    assert dev.probes == probes and Recycling.syncs == syncs


def test_the_default_mode_parks_at_most_eight_gib(monkeypatch):
    """the product default must not take a quarter of the device.  On a fresh device (one class for the first 91 GiB)
    the default walk gives up after 8 GiB and says so (balanced False, mode "cheap"); where the allocator's blocks are spread over the
    classes — a learner's process — the same 8 GiB are plenty; MXV_PLACEMENT=search restores the long walk."""
    from gym_amd import placement

    monkeypatch.setenv("MXV_PLACEMENT", "cheap")
    assert placement.mode() == "cheap" and placement.max_park_bytes() == 8 * GiB
    fresh = SimDevice([(91 * GiB, "A"), (187 * GiB, "B"), (288 * GiB, "C")])
    out, rep = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=fresh)
    assert rep["mode"] == "cheap" and rep["balanced"] is False and rep["parked_GiB"] <= 8.0 and rep["budget_GiB"] == 8.0
    assert fresh.peak <= 8.5 * GiB + 8 * GiB + 3 * GiB and fresh.live == sum(t.n for t in out.values())
    regions, addr = [], 0
    for i in range(200):                                         # runs of 6 GiB, alternating classes
        addr += 6 * GiB
        regions.append((addr, "AB"[i % 2]))
    out, rep = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=SimDevice(regions))
    assert rep["balanced"] is True and rep["parked_GiB"] <= 8.0
    monkeypatch.setenv("MXV_PLACEMENT", "search")
    assert placement.mode() == "search" and placement.max_park_bytes() == 112 * GiB
    out, rep = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=SimDevice([(91 * GiB, "A"), (187 * GiB, "B"), (288 * GiB, "C")]))
    assert rep["mode"] == "search" and rep["balanced"] is True and rep["parked_GiB"] >= 80


def test_auto_walks_far_only_on_an_otherwise_empty_device(monkeypatch):
    """The default (MXV_PLACEMENT unset = auto): the long walk when at least 90 % of the device is free — a dedicated rollout / benchmark
    process: the parked memory disturbs nobody and is released before the call returns —, the 8-GiB cap as soon as anybody (this process's
    learner, another process, another rank) holds memory."""
    from gym_amd import placement

    monkeypatch.delenv("MXV_PLACEMENT", raising=False)
    assert placement.mode() == "auto"
    regions = [(91 * GiB, "A"), (187 * GiB, "B"), (288 * GiB, "C")]

    class WithTotal(SimDevice):
        def total_bytes(self):
            return 288 * GiB

    empty = WithTotal(regions, free=286 * GiB)
    out, rep = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=empty)
    assert rep["mode"] == "auto->search" and rep["balanced"] is True and rep["parked_GiB"] >= 80 and empty.live == sum(t.n for t in out.values())
    shared = WithTotal(regions, free=200 * GiB)                    # somebody holds 88 GiB
    out, rep = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=shared)
    assert rep["mode"] == "auto->cheap" and rep["balanced"] is False and rep["parked_GiB"] <= 8.0 and rep["budget_GiB"] == 8.0
    out, rep = sorted_tensors(CARTPOLE, CP_GROUPS, None, _backend=SimDevice(regions))      # a backend that cannot say: the cap
    assert rep["mode"] == "auto->cheap" and rep["parked_GiB"] <= 8.0
    assert placement.resolve_mode(280 * GiB, 288 * GiB) == "search" and placement.resolve_mode(250 * GiB, 288 * GiB) == "cheap"
    monkeypatch.setenv("MXV_PLACEMENT", "cheap")
    assert placement.resolve_mode(288 * GiB, 288 * GiB) == "cheap"


def test_environment_switch(monkeypatch):
    from gym_amd import placement

    monkeypatch.delenv("MXV_PLACEMENT", raising=False)
    assert placement.enabled() and placement.mode() == "auto"
    for v in ("off", "0", "OFF", "no", "false"):
        monkeypatch.setenv("MXV_PLACEMENT", v)
        assert not placement.enabled()
    monkeypatch.setenv("MXV_PLACEMENT", "on")
    assert placement.enabled()
    monkeypatch.setenv("MXV_PLACEMENT_MAX_PARK_GIB", "1.5")
    assert placement.max_park_bytes() == 3 << 29


# ---- round 6: the brakes ----------------------------------------------------------------------------------
def test_a_rank_of_a_multi_process_job_never_takes_the_long_walk_on_its_own(monkeypatch):
    """WORLD_SIZE > 1: "auto" resolves to "cheap" even on an empty device (eight ranks starting together would each hold ~90 GiB for
    seconds before their first barrier; ranks sharing a device would each see it as empty), and the walk is bounded at 0.5 s."""
    from gym_amd import placement

    monkeypatch.delenv("MXV_PLACEMENT", raising=False)
    monkeypatch.delenv("MXV_PLACEMENT_MAX_SECONDS", raising=False)
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert placement.resolve_mode(280 * GiB, 288 * GiB) == "search" and placement.max_seconds() is None
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert placement.resolve_mode(280 * GiB, 288 * GiB) == "cheap" and placement.max_seconds() == 0.5
    monkeypatch.setenv("MXV_PLACEMENT", "search")                      # an explicit mode still wins
    assert placement.resolve_mode(280 * GiB, 288 * GiB) == "search"
    monkeypatch.setenv("MXV_PLACEMENT_MAX_SECONDS", "2.5")
    assert placement.max_seconds() == 2.5


def test_the_walk_stops_when_its_time_is_up(monkeypatch):
    """A fresh device (one class for the first 96 GiB) and a generous park budget: unbounded, the walk parks ~90 GiB and comes back
    balanced; with a wall-time bound that the simulated clock exceeds after a few probes it stops early, unbalanced, and says why."""
    monkeypatch.setenv("MXV_PLACEMENT", "search")
    regions = [(96 * GiB, 0), (192 * GiB, 1), (288 * GiB, 2)]
    dev = SimDevice(regions)
    out, rep = sorted_tensors(CARTPOLE, CP_GROUPS, "sim", _backend=dev)
    assert rep["balanced"] and rep["parked_GiB"] > 50 and "stopped_by" not in rep

    class Clocked(SimDevice):
        t = 0.0

        def now(self):
            return self.t

        def probe(self, wide, narrow):
            self.t += 0.02                                  # every probe window costs 20 ms of simulated wall time
            return super().probe(wide, narrow)

    monkeypatch.setenv("MXV_PLACEMENT_MAX_SECONDS", "0.5")
    dev = Clocked(regions)
    out, rep = sorted_tensors(CARTPOLE, CP_GROUPS, "sim", _backend=dev)
    assert not rep["balanced"] and rep["stopped_by"] == "time" and rep["seconds"] < 1.5 and rep["parked_GiB"] < 40
    assert set(out) == {n for n, *_ in CARTPOLE}


def test_the_walk_backs_off_when_somebody_else_starts_allocating(monkeypatch):
    """the emptiness test of "auto" is a point-in-time check.  Here a neighbour takes 60 GiB while the walk is under way: the
    device's free memory shrinks by more than the walk itself holds, and it stops parking at once instead of racing the neighbour to the
    last GiB."""
    monkeypatch.setenv("MXV_PLACEMENT", "search")
    monkeypatch.delenv("MXV_PLACEMENT_MAX_SECONDS", raising=False)

    class Neighbour(SimDevice):
        def probe(self, wide, narrow):
            if self.probes == 12:
                self.live += 60 * GiB                       # somebody else's allocation, not ours
            return super().probe(wide, narrow)

    dev = Neighbour([(96 * GiB, 0), (192 * GiB, 1), (288 * GiB, 2)])
    out, rep = sorted_tensors(CARTPOLE, CP_GROUPS, "sim", _backend=dev)
    assert rep["stopped_by"] == "crowded" and not rep["balanced"] and rep["parked_GiB"] < 30
