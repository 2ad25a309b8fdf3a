"""Trajectory tensors sorted by HBM class.

The MI355X's HBM address space consists of three contiguous classes of 96 GB (what the three ranks of a 12-high HBM3E stack would give).
Long store streams written concurrently interfere when the physical memory behind them shares a class: the fused CartPole rollout
(observations 16 B per env-step | rewards 8 + actions 8) runs 5.4 / 5.7 / 6.4 us per 2^20-env step with none / one / both of {rewards,
actions} in the observations' class, the tabular rollout (four 8-B streams) 5.7 / 6.1 / 7.1 us split 2 + 2 / 1 + 3 / 4 + 0
(profiles/r3/r3a_*, r3g_tab_class_ab.jsonl).  The rule: split the streams into two groups of about equal bytes per env-step and keep the
groups on different classes.  A fresh process is handed ONE class for its first ~90 GiB, so back-to-back allocations do the opposite.

`sorted_tensors` gets there with ordinary (torch / hipMalloc) allocations: the first group-0 tensor is the anchor; every other grouped
tensor is allocated, classified against the anchor with mxv_hbm_pair_probe (a 16-B/lane stream into the anchor next to an 8-B/lane
stream into the candidate, at both ends of both tensors; a pair known to share a class — two parts of the anchor — is timed right
next to it, so no absolute threshold is involved) and, if it lies on the wrong side, parked and replaced by the next allocation,
which lies further along in physical memory.  Parked tensors are released at the end.  Typical: a few GiB parked for 0.1 s; a fresh
device: up to ~90 GiB for ~3 s.  Nothing here touches the results: only WHERE the tensors live.

Modes (MXV_PLACEMENT):
  auto / on / unset (the default)  "cheap" whenever anybody holds device memory — this process's learner, another process, another rank:
        less than 90 % of the device free beyond the set —, "search" only on an otherwise EMPTY device, where parked memory disturbs
        nobody and is released before the call returns (a dedicated rollout or benchmark process).  The report says which it was;
  cheap   the walk may park at most 8 GiB beside the set (and never more than half of the memory that is free beyond the set): in a
        learner's process, whose allocator already holds blocks all over the device, that is enough to find the two classes; in a fresh
        process whose first ~90 GiB all lie in one class it is not — the set then comes back with balanced = False and runs 5-17 % slower
        than a sorted one (bench.py: variants.placement_cheap against the headline).  A product default must not take a quarter of the
        device beside a learner;
  search  up to 112 GiB parked transiently (~1-3 s on a fresh device) to leave the first class, whoever else is there;
  off     never sort — ordinary allocations, no probe launches, no synchronisation, nothing parked.
MXV_PLACEMENT_MAX_PARK_GIB overrides the cap of either mode.  An out-of-memory error inside the walk ends it with ordinary allocations
instead of reaching the caller.  Three more brakes (round 6): a process that is one of several ranks (WORLD_SIZE > 1) never takes the long
walk on its own authority ("auto" resolves to "cheap": eight ranks starting together must not each hold ~90 GiB for seconds before their
first barrier, and ranks that share a device would each see it as empty) and every walk there ends after MXV_PLACEMENT_MAX_SECONDS
(default 0.5 s with WORLD_SIZE > 1, unbounded otherwise) with what it has; and every walk re-reads the device's free memory as it goes —
if it shrank by more than the walk itself took (a learner or another process started allocating in the meantime) it stops parking at once.
The report carries `seconds` and, when a brake ended the walk, `stopped_by`.

What was measured is remembered per device (`_ClassMemo`): torch's caching allocator hands a learner's loop the same blocks again and
again (`out = r.rollout_per_step(K)` alternates between two sets), and a block keeps its physical memory for as long as its segment is
not returned to the driver.  The memo is keyed by block address and size and is dropped whenever the allocator's count of segments
returned to the driver has moved since it was written (`torch.cuda.empty_cache()` by anyone, an out-of-memory retry) — a set built from
remembered blocks costs no probe launch at all.
"""
from __future__ import annotations

import math
import os
import threading
import time
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _native

MiB = 1 << 20
WIDE, NARROW = 256 * MiB, 128 * MiB     # what one probe window writes: 16 steps x 2^20 lanes x 16 B / 8 B
SAME_RATIO = 0.955                      # a different-class pair runs at 0.89-0.91 of the same-class time
MIN_SET_BYTES = _native.SORTED_MIN_BYTES
CHEAP_MAX_PARK_BYTES = 8 << 30          # the default: never more than this parked beside the set itself
SEARCH_MAX_PARK_BYTES = 112 << 30       # MXV_PLACEMENT=search (a fresh device needs ~90 GiB to leave its first class)
DEFAULT_MAX_PARK_BYTES = CHEAP_MAX_PARK_BYTES


EMPTY_DEVICE_FRACTION = 0.90            # auto: the long walk only when at least this much of the device is free (nobody to disturb)


def mode() -> str:
    """"off" | "auto" | "cheap" | "search" from MXV_PLACEMENT (unset / on / 1 / auto -> "auto")."""
    v = os.environ.get("MXV_PLACEMENT", "auto").strip().lower()
    if v in ("off", "0", "no", "false"):
        return "off"
    return v if v in ("search", "cheap") else "auto"


def world_size() -> int:
    try:
        return max(1, int(os.environ.get("WORLD_SIZE", "1")))
    except ValueError:
        return 1


def resolve_mode(free_bytes: Optional[int] = None, total_bytes: Optional[int] = None) -> str:
    """"cheap" or "search" for this call: an explicit MXV_PLACEMENT wins; "auto" walks far only on an otherwise empty device, and only in
    a process that is not one of several ranks."""
    m = mode()
    if m != "auto":
        return m
    if free_bytes is None or not total_bytes or world_size() > 1:
        return "cheap"
    return "search" if free_bytes >= EMPTY_DEVICE_FRACTION * total_bytes else "cheap"


def max_seconds() -> Optional[float]:
    """Wall-time bound of one walk: MXV_PLACEMENT_MAX_SECONDS if set (<= 0: none), else 0.5 s for a rank of a multi-process job, none
    for a single process (whose walk is bounded by what it may park)."""
    try:
        v = float(os.environ["MXV_PLACEMENT_MAX_SECONDS"])
        return v if v > 0 else None
    except (KeyError, ValueError):
        return 0.5 if world_size() > 1 else None


def enabled() -> bool:
    """MXV_PLACEMENT=off (or 0 / no / false): trajectory_buffers(layout="auto") never sorts — ordinary allocations, no probe launches, no
    device synchronisation, no memory parked.  For a process that shares its GPU with a learner that cannot spare even 8 GiB transiently."""
    return mode() != "off"


def max_park_bytes(resolved: Optional[str] = None) -> int:
    """Upper bound of the memory the walk may hold transiently: MXV_PLACEMENT_MAX_PARK_GIB if set, else 112 GiB for a "search" walk and
    8 GiB otherwise (`resolved`: what resolve_mode() returned for this call; default: the environment's explicit mode)."""
    try:
        return max(0, int(float(os.environ["MXV_PLACEMENT_MAX_PARK_GIB"]) * (1 << 30)))
    except (KeyError, ValueError):
        return SEARCH_MAX_PARK_BYTES if (resolved or mode()) == "search" else CHEAP_MAX_PARK_BYTES


def _is_oom(e: BaseException) -> bool:
    oom = getattr(torch, "OutOfMemoryError", None)
    return (oom is not None and isinstance(e, oom)) or isinstance(e, MemoryError) or "out of memory" in str(e).lower()


def _nbytes(shape, dtype) -> int:
    return math.prod(shape) * torch.empty((), dtype=dtype).element_size()


class _Device:
    """What sorted_tensors needs from the device: allocation, the pair probe, free memory.  tests/test_placement_logic.py substitutes a
    simulated address space with class regions to drive every branch of the search on the CPU."""

    def __init__(self, device, stream):
        self.dev = torch.device(device)
        self.stream = stream

    def alloc(self, shape, dtype, zero):
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else torch.cuda.device(self.dev)
        with ctx:
            return (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.dev)

    def ptr(self, t) -> int:
        return t.data_ptr()

    def sync(self):
        torch.cuda.synchronize(self.dev)

    def free_bytes(self) -> int:
        return torch.cuda.mem_get_info(self.dev)[0]

    def total_bytes(self) -> int:
        return torch.cuda.mem_get_info(self.dev)[1]

    def probe(self, wide_ptr: int, narrow_ptr: int) -> float:
        return _native.hbm_pair_probe(self.dev.index, wide_ptr, narrow_ptr, 4)

    def release(self):
        torch.cuda.empty_cache()

    def segment_frees(self):
        """How many device segments the caching allocator has returned to the driver so far (None: unknown — nothing is remembered).
        With expandable segments the allocator maps and unmaps physical memory under stable addresses, which this count does not
        describe: nothing is remembered then."""
        conf = " ".join(os.environ.get(k, "") for k in ("PYTORCH_CUDA_ALLOC_CONF", "PYTORCH_HIP_ALLOC_CONF", "PYTORCH_ALLOC_CONF"))
        if "expandable_segments:true" in conf.replace(" ", "").lower():
            return None
        return torch.cuda.memory_stats(self.dev).get("num_device_free")

    def key(self):
        return self.dev.index


class _ClassMemo:
    """Per device: what the probes found out about blocks of the caching allocator.  `single[(ptr, nbytes)]`: the block lies in one HBM
    class from end to end (anchor-grade); `rel[(anchor ptr, anchor nbytes, ptr, nbytes)]`: the block's relation to that anchor (+1 / -1 / 0;
    it is probed at BOTH ends of the anchor, so the anchor's size is part of the key)."""

    def __init__(self):
        self.stamp, self.single, self.rel = None, set(), {}

    def validate(self, frees):
        if frees is None or frees != self.stamp:
            self.single.clear()
            self.rel.clear()

    def seal(self, frees):
        self.stamp = frees
        if frees is None:
            self.single.clear()
            self.rel.clear()


_MEMO: Dict[object, _ClassMemo] = {}
_LOCK = threading.Lock()                # one search at a time per process: the memo, the probes' timing and the parked memory are shared


def sorted_tensors(specs: Sequence[Tuple[str, tuple, torch.dtype, bool]], groups: Dict[str, int], device: torch.device,
                   stream: Optional[torch.cuda.Stream] = None, budget_bytes: Optional[int] = None, _backend=None):
    """specs: [(name, shape, dtype, zero_fill)]; groups: {name: 0 | 1} for the tensors that carry long store streams (the others are
    allocated last, wherever).  Returns ({name: tensor}, report).  report["balanced"]: every group-0 tensor shares the anchor's class
    from end to end and every group-1 tensor lies outside it.

    Never fails because of the SEARCH: the memory it may park is bounded by budget_bytes (default: half of what is free beyond the set
    itself, at most MXV_PLACEMENT_MAX_PARK_GIB), and an out-of-memory error anywhere inside it — another process took the memory in
    the meantime, several ranks share the device — drops everything parked and returns ordinary allocations with balanced = False
    (only if the SET itself does not fit does the error reach the caller, as it would without placement)."""
    be = _backend if _backend is not None else _Device(device, stream)
    with _LOCK:
        try:
            return _sorted_locked(specs, groups, be, budget_bytes)
        except Exception as e:  # noqa: BLE001
            if not _is_oom(e):
                raise
            note = f"placement search ran out of device memory ({type(e).__name__}): ordinary allocations"
        # the frames of the failed search (and with them every tensor it held) are gone here
        getattr(be, "release", lambda: None)()
        memo = _MEMO.get(be.key() if hasattr(be, "key") else id(be))
        if memo is not None:
            memo.seal(None)
        report = {"kind": "sorted", "balanced": False, "parked_GiB": 0.0, "candidates": 0, "remembered": 0, "note": note,
                  "requested_GiB": round(sum(_nbytes(shape, dt) for _, shape, dt, _ in specs) / 2**30, 3)}
        return {n: be.alloc(shape, dt, zero) for n, shape, dt, zero in specs}, report


def _sorted_locked(specs, groups, be, budget_bytes):
    now = getattr(be, "now", time.perf_counter)
    t_begin = now()
    limit = max_seconds()
    memo = _MEMO.setdefault(be.key() if hasattr(be, "key") else id(be), _ClassMemo())
    frees_of = getattr(be, "segment_frees", lambda: None)
    memo.validate(frees_of())
    names = [n for n, *_ in specs]
    spec = {n: (shape, dt, zero) for n, shape, dt, zero in specs}
    nbytes = {n: _nbytes(spec[n][0], spec[n][1]) for n in names}
    g0 = [n for n in names if groups.get(n) == 0]
    g1 = [n for n in names if groups.get(n) == 1]
    free_now = be.free_bytes()
    resolved = resolve_mode(free_now, getattr(be, "total_bytes", lambda: None)())
    report = {"kind": "sorted", "mode": resolved if mode() != "auto" else f"auto->{resolved}", "balanced": False, "parked_GiB": 0.0,
              "candidates": 0, "remembered": 0, "requested_GiB": round(sum(nbytes.values()) / 2**30, 3)}

    dirty = [True]     # allocations (and their zero fills) issued since the last device synchronisation
    held = [0]         # bytes this call has allocated and not let go (the set + what is parked)

    def alloc(name):
        dirty[0] = True
        held[0] += nbytes[name]
        return be.alloc(*spec[name])

    def plain(note):
        report["note"] = note
        return {n: alloc(n) for n in names}, report

    if not g0 or not g1:
        return plain("nothing to keep apart: ordinary allocations")
    anchor_name = max(g0, key=lambda n: nbytes[n])
    if nbytes[anchor_name] < WIDE + NARROW or any(nbytes[n] < NARROW for n in g0 + g1):
        return plain("tensors too small to classify: ordinary allocations")
    # what may be parked beside the set: half of what is free once the set itself is counted, never more than the cap
    budget = (min(max(free_now - sum(nbytes.values()), 0) // 2, max_park_bytes(resolved)) if budget_bytes is None else int(budget_bytes))
    report["budget_GiB"] = round(budget / 2**30, 2)
    parked, parked_bytes = [], 0
    warmed = [False]

    def brake():
        """Why the walk must stop parking NOW, or None: out of time, or the device's free memory shrank by more than this call took
        (1 GiB of slack for allocator rounding) — somebody else started allocating."""
        if limit is not None and now() - t_begin > limit:
            return "time"
        if free_now - be.free_bytes() > held[0] + (1 << 30):
            return "crowded"
        return None

    def probe(wide_ptr, narrow_ptr, _warm_at=None):
        if dirty[0]:                                       # a probe times the device: it must be idle.  A set made of remembered blocks
            be.sync()                                      # never gets here — no synchronisation at all on a learner's steady loop
            dirty[0] = False
        if not warmed[0]:                                  # clock ramp + first touch, once per call and only if anything is measured at all
            warmed[0] = True
            w = wide_ptr if _warm_at is None else _warm_at
            for _ in range(40):
                be.probe(w, w + WIDE)
        return be.probe(wide_ptr, narrow_ptr)

    def park(*named):
        nonlocal parked_bytes
        for n, t in named:
            parked.append(t)
            parked_bytes += nbytes[n]

    out, ok = {}, True
    for attempt in range(4):
        # the anchor: one class from end to end (an allocation that straddles a class boundary is parked and replaced)
        anchor = alloc(anchor_name)
        report["candidates"] += 1
        a0 = be.ptr(anchor)
        a1 = a0 + nbytes[anchor_name]
        if (a0, nbytes[anchor_name]) in memo.single:
            report["remembered"] += 1
        else:
            same = probe(a0, a0 + WIDE, a0)                # both streams inside the first 384 MiB of one allocation: a same-class pair
            straddles = probe(a0, a1 - NARROW) <= SAME_RATIO * same
            if straddles and parked_bytes + nbytes[anchor_name] <= budget and not brake():
                park((anchor_name, anchor))
                del anchor
                continue
            if straddles:
                ok = False                                 # no budget left to replace it: the set is built on an anchor that spans two classes
            else:
                memo.single.add((a0, nbytes[anchor_name]))
            report["same_class_us"] = round(same, 3)

        def relation(t, n):
            """+1: same class as the anchor at both ends, -1: another class at both ends, 0: mixed."""
            c0 = be.ptr(t)
            c1 = c0 + nbytes[n]
            key = (a0, a1 - a0, c0, nbytes[n])             # the relation is probed at both ends of the anchor: its size is part of the key
            known = memo.rel.get(key)
            if known is not None:
                report["remembered"] += 1
                return known
            cal = probe(a0, a0 + WIDE, a0)
            p0, p1 = probe(a0, c0), probe(a1 - WIDE, c1 - NARROW)
            s0, s1 = p0 > SAME_RATIO * cal, p1 > SAME_RATIO * cal
            if not s0 and not s1:
                report.setdefault("different_class_us", round(min(p0, p1), 3))
            r = 1 if (s0 and s1) else -1 if (not s0 and not s1) else 0
            memo.rel[key] = r
            return r

        out = {anchor_name: anchor}
        restart = False
        for n in g0:                                       # the rest of the anchor's group: allocated right behind it
            if n == anchor_name:
                continue
            t = alloc(n)
            report["candidates"] += 1
            if relation(t, n) == 1:
                out[n] = t
            elif parked_bytes + sum(nbytes[m] for m in out) + nbytes[n] <= budget and attempt < 3 and not brake():
                park((n, t), *out.items())                 # a class boundary inside the group: start it again from here
                restart = True
                break
            else:
                out[n], ok = t, False
        if restart:
            continue
        for n in g1:                                       # the other group: walk on until the class changes
            while True:
                t = alloc(n)
                report["candidates"] += 1
                r = relation(t, n)
                why = None if r == -1 else brake()
                if r == -1 or parked_bytes + nbytes[n] > budget or why:
                    ok = ok and r == -1
                    out[n] = t
                    if why:
                        report["stopped_by"] = why
                    break
                park((n, t))
        break
    else:   # four starts in a row ran into a class boundary: give up sorting
        ok, out = False, {}
        anchor = t = None                                  # (the last parked tensors are referenced from here too: let them go with the rest)
        for n in g0 + g1:
            out[n] = alloc(n)
    for n in names:
        if n not in out:
            out[n] = alloc(n)
    had_parked = bool(parked)
    del parked
    anchor = t = None
    if had_parked:
        # the parked tensors go back to the driver, not into the allocator's cache — and so does every other unused block the allocator
        # held: only what this call returns is known to keep its memory
        be.release()
        keep = {be.ptr(t) for t in out.values()}
        memo.single = {k for k in memo.single if k[0] in keep}
        memo.rel = {k: v for k, v in memo.rel.items() if k[0] in keep and k[2] in keep}
    memo.seal(frees_of())  # the release moved the allocator's count: what survives stays valid from the new count on
    report.update({"balanced": ok, "parked_GiB": round(parked_bytes / 2**30, 2), "seconds": round(now() - t_begin, 3)})
    return {n: out[n] for n in names}, report
