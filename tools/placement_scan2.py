#!/usr/bin/env python3
"""Placement scan 2: candidates for a LAYOUT RULE (follow-up of tools/placement_scan.py, profiles/r02q_placement_scan_one_allocation.jsonl).

Scan 1 showed, inside ONE allocation on a fast box: five tensors packed back to back from offset 0 (bases at multiples of 256 MiB) run
in the slow mode (6.8 us per 2^20-env CartPole step), the same packed layout started 3 GiB into the block runs fast (5.8), and eight
of eight layouts with shuffled order and random GiB-scale gaps run fast (5.76-5.95): the mode is a function of the ADDRESSES.
Here: how small can irregular gaps be and still be fast, does a fixed irregular pattern work at other shard sizes / dtypes, and is a
front pad alone enough?  Box check first (slow-kind boxes exit)."""
import argparse
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

GiB = 1 << 30
ORDER = ["obs", "reward", "actions", "terminated", "truncated"]
# irregular fractions (of the tensors' total size) for front pad and the four gaps: fixed "random-looking" constants
IRREG = [0.0219, 0.0437, 0.0718, 0.0271, 0.0483]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    args = ap.parse_args()
    import torch
    from gym_amd.rollout import DeviceRollout

    K = 256

    def run_size(n, compact, env="CartPole-v1", check_box=False):
        r = DeviceRollout(env, n, seed=0, action_seed=1, reward_f32=compact, action_i32=compact)
        r.reset(seed=0)
        O = r.O
        rb, ab = (4, 4) if compact else (8, 8 if r.NA > 0 else 4)
        sizes = {"obs": K * n * O * 4, "reward": K * n * rb, "actions": K * n * ab, "terminated": K * n, "truncated": K * n}
        total = sum(sizes.values())
        with torch.cuda.stream(r.stream):
            block = torch.empty(int(total * 1.7) + 4 * GiB, dtype=torch.uint8, device=r.device)
            block.zero_()
        r.synchronize()
        base0 = (-block.data_ptr()) % (2 << 20)

        def views(offs):
            o = {}
            for name in ORDER:
                a = base0 + offs[name]
                b = block[a:a + sizes[name]]
                if name == "obs":
                    o[name] = b.view(torch.float32).view(K, n, O)
                elif name == "reward":
                    o[name] = b.view(r.reward_dtype).view(K, n)
                elif name == "actions":
                    o[name] = b.view(r.action_dtype).view(K, n)
                else:
                    o[name] = b.view(K, n)
            return o

        def timed(traj, launches=6):
            for _ in range(2):
                r.rollout_per_step(K, out=traj)
            r.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(r.stream)
            for _ in range(launches):
                r.rollout_per_step(K, out=traj)
            e1.record(r.stream)
            r.synchronize()
            return round(e0.elapsed_time(e1) / launches / K * 1e3, 3)

        def lay(front, gaps, perm=ORDER, align=4096):
            offs, off = {}, int(front)
            for name, g in zip(perm, list(gaps) + [0]):
                off = (off + align - 1) // align * align
                offs[name] = off
                off += sizes[name] + int(g)
            return offs

        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:
            r.rollout_per_step(K, out=views(lay(0, [0] * 4)))
            r.synchronize()
        if check_box:
            probe = [timed(views(lay(s * GiB, [0] * 4))) for s in (0, 3, 0, 3)]
            kind = "fast" if min(probe) < 6.2 else "slow"
            print(json.dumps({"box": kind, "probe_us_per_step": probe}), flush=True)
            if kind == "slow" and not args.force:
                r.close()
                return False
        layouts = {"packed from 0": lay(0, [0] * 4)}
        for f in (0.25, 1, 2, 3):
            layouts[f"packed, front pad {f} GiB"] = lay(f * GiB, [0] * 4)
        for scale in (1.0, 0.25, 0.0625, 0.015625):
            fr = [x * scale * total for x in IRREG]
            layouts[f"irregular gaps x{scale} ({sum(fr) / GiB:.2f} GiB)"] = lay(fr[0], fr[1:])
        layouts["irregular x1.0, no front pad"] = lay(0, [x * total for x in IRREG[1:]])
        layouts["front pad 0.0219*total only"] = lay(IRREG[0] * total, [0] * 4)
        for pw in (21, 24, 27, 28):   # regular power-of-two gaps for contrast
            layouts[f"regular gaps 2^{pw}"] = lay(0, [1 << pw] * 4)
        rng = random.Random(7)
        for i in range(6):
            fr = [rng.uniform(0.005, 0.08) * total for _ in range(5)]
            perm = ORDER[:]
            rng.shuffle(perm)
            layouts[f"random small gaps {i} ({sum(fr) / GiB:.2f} GiB, {'/'.join(p[:3] for p in perm)})"] = lay(fr[0], fr[1:], perm)
        names = list(layouts)
        res = {k: [] for k in names}
        for rep in range(2):
            rng.shuffle(names)
            for k in names:
                res[k].append(timed(views(layouts[k])))
        for k in layouts:
            print(json.dumps({"env": env, "n": n, "compact": compact, "layout": k, "us_per_step": res[k],
                              "offsets_MiB": {a: round(b / (1 << 20), 2) for a, b in layouts[k].items()}}), flush=True)
        r.close()
        del block
        torch.cuda.empty_cache()
        return True

    if not run_size(1 << 20, False, check_box=True):
        return
    run_size(1 << 20, True)
    run_size(1 << 19, False)
    run_size(1 << 17, False)
    run_size(1 << 20, False, env="Pendulum-v1")


if __name__ == "__main__":
    main()
