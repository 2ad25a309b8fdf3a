"""The headline measurement of bench.py (BASELINE.json configs[1]: CartPole-v1, 2^20 envs, fused trajectory launches) as a function of the
parsed arguments: process group, sharded engine, placement, warm-up, the timed region bracketed by barrier + synchronize, work_check, the
gather alone and the per-launch cadence A/B at N > 1, write probe, max over ranks, the long-form record and the one compact stdout line.
The CPU baseline is passed in by bench.py (the only file that may touch oracle/)."""
import json
import os
import socket
import subprocess
import sys
import time

from .common import (CHECK_ENVS, ENV_ID, ENVS_TOTAL, HBM_PEAK_GBS, ROOT, algorithmic_bytes_per_env_step, read_traffic, spinup_steps,
                     timed_repeats, warm_until_stable, work_checksum)

LINE_LIMIT = 4096         # bytes of the final stdout line (the driver's record keeps the last 8 KB of stdout)
XGMI_BUS_GBS = (80.0, 150.0)   # what RCCL's ring reaches per direction for MB-sized messages on 7 point-to-point links


def self_launch(args, bench_file) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in the environment as torch.distributed.run would set them), wait, and fail fast and readably if
    any rank dies or the job exceeds --launch-timeout.  Rank 0 prints the JSON line on the inherited stdout."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MXV_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(bench_file)] + sys.argv[1:], env=env))
    deadline = time.time() + args.launch_timeout
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            rc = next((c for c in codes if c), 0)
            if rc:
                print(f"bench.py: ranks exited with codes {codes}", file=sys.stderr, flush=True)
            break
        bad = [(i, c) for i, c in enumerate(codes) if c not in (None, 0)]
        if bad or time.time() > deadline:
            why = f"rank {bad[0][0]} exited with code {bad[0][1]}" if bad else f"no result after {args.launch_timeout:.0f} s"
            print(f"bench.py: {why}; stopping the other ranks", file=sys.stderr, flush=True)
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(10)
                except subprocess.TimeoutExpired:
                    p.kill()
            rc = bad[0][1] if bad else 124
            break
        time.sleep(0.05)
    return rc



def _sig(x, digits=8):
    """Floats of the line at `digits` significant digits (the line has a size limit; nothing measured is more precise than that);
    whole numbers stay exact."""
    if isinstance(x, float):
        if x != x or abs(x) == float("inf"):
            return None
        return x if abs(x) < 2.0**53 and x == int(x) else float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def compact_line(full, limit=LINE_LIMIT):
    """The one stdout line from the long-form record: prose dropped, floats at 8 significant digits, per-rank reports as rows.  If it
    still exceeded `limit` (it does not for 1..8 ranks: tests/test_bench_helpers.py) the optional parts go, least important first."""
    cfg, roof = full["config"], full["roofline"]
    pl = cfg.get("placement") or {}
    wc = cfg.get("work_check")
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data")}
    line["config"] = {
        "workload": cfg["workload"], "num_envs_per_gpu": cfg["num_envs_per_gpu"], "chunk": cfg["chunk"], "repeats": cfg["repeats"],
        "timed_steps": cfg["timed_steps"], "timed_region_ms": cfg["timed_region_ms"], "launch": cfg["launch"], "outputs": cfg["outputs"],
        "placement": {k: pl[k] for k in ("kind", "mode", "balanced", "parked_GiB", "seconds", "error") if k in pl},
        "parallelism": cfg["parallelism"], "ranks_seen": cfg["ranks_seen"], "gathers_in_timed_region": cfg["gathers_in_timed_region"],
        "gather_every": cfg["gather_every"], "gather_transport": cfg["gather_transport"], "gather_us": cfg.get("gather_us"),
        "comm": cfg["comm"],
        "per_rank_fields": ["rank", "device", "kernel_us_per_step", "write_probe_us_per_step", "placement", "placement_seconds"],
        "per_rank": [[r["rank"], r["device"], r["kernel_us_per_step"], r["write_probe_us_per_step"],
                      (r.get("placement") or {}).get("kind", "").split(" ")[0], (r.get("placement") or {}).get("seconds")] for r in cfg["per_rank"]],
        "work_check": None if wc is None else {k: v for k, v in wc.items() if k != "what"},
        "launch_info": cfg["launch_info"],
    }
    if cfg.get("cadence_ab"):
        line["config"]["cadence_ab"] = cfg["cadence_ab"]
    line["roofline"] = {k: roof[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_env_step",
                                             "algorithmic_bytes_per_launch", "env_steps_per_launch", "steps_per_launch", "avg_launch_us") if k in roof}
    wp = roof.get("write_probe")
    if wp:
        line["roofline"].update(write_probe_us_per_step=wp["us_per_step"], kernel_us_per_step=wp["kernel_us_per_step"],
                                kernel_over_probe=wp["kernel_over_probe"])
    cb = full.get("cpu_baseline")
    if cb:
        rp = cb.get("reference_python") or {}
        line["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "single_core_value": cb["single_core_value"], "sample": cb["sample_short"],
                                "reference_python": {"value": rp.get("value"), "unit": rp.get("unit"), "num_envs": rp.get("num_envs"),
                                                     "source": "profiles/reference_cpu_baseline.json", "live": (rp.get("live") or {}).get("value")}}
    if "variants" in full:
        from benchmarks.variants import summary
        line["variants_fields"] = ["us_per_step", "roofline_frac"]
        line["variants"] = summary(full["variants"])
    if full.get("details"):
        line["details"] = full["details"]
    line = _sig(line)
    line["value"] = full["value"]
    for drop in ("details", "variants_fields", ("config", "cadence_ab"), ("config", "outputs"), ("config", "launch"), "variants"):
        if len(json.dumps(line)) < limit:
            break
        if isinstance(drop, tuple):
            line[drop[0]].pop(drop[1], None)
        else:
            line.pop(drop, None)
    return line



def run(args, bench_file, cpu_baseline=None):
    """Everything `python bench.py` does after parsing its arguments.  bench_file: the script the self-launcher starts the ranks with;
    cpu_baseline(sample_steps) -> dict: called on rank 0 at N = 1 unless --no-cpu-baseline."""
    if args.placement == "off":
        os.environ["MXV_PLACEMENT"] = "off"      # inherited by self-launched ranks
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, bench_file))

    # stdout carries ONE JSON line and nothing else: RCCL ("Hostname : ... / Librccl path : ...") and gloo ("[Gloo] Rank 0 is connected
    # ...") print from C++ straight to file descriptor 1 when a communicator comes up.  The descriptor is pointed at stderr for the whole
    # run (in every rank), and the line goes out through a private duplicate of the original one.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; gym_amd has no CPU fallback")
    ndev = torch.cuda.device_count()
    if args.backend == "gloo":
        local_rank %= ndev   # debug path: more ranks than GPUs
    elif local_rank >= ndev:
        raise SystemExit(f"bench.py: rank {rank} needs device {local_rank} but only {ndev} HIP device(s) are visible "
                         f"(--gpus {args.gpus} with --backend nccl is one process per GPU; --backend gloo shares devices)")
    torch.cuda.set_device(local_rank)
    comm_info = {"backend": None}
    solo_group = world == 1 and args.force_gather and args.comm == "torch"    # a REAL one-rank process group: torch's RCCL path, minus the links
    if solo_group and "MASTER_PORT" not in os.environ:
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or solo_group:
        import datetime

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL's kernels run on high-priority streams: a chunk's all-gather gets CUs as soon as rollout waves retire instead of
        # queueing behind the next chunk's (long-running, chip-filling) rollout launch
        from gym_amd.distributed import prefer_high_priority_collectives
        prefer_high_priority_collectives()        # TORCH_NCCL_HIGH_PRIORITY=1 unless the caller set it: explicit, this process only
        to = datetime.timedelta(seconds=args.init_timeout)
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=to)
            else:
                dist.init_process_group("gloo", timeout=to)
            # first collective: how many ranks does the communicator really span?
            one = torch.ones(1, dtype=torch.int64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(one)
            ranks_seen = int(one.item())
        except Exception as e:  # noqa: BLE001
            raise SystemExit(f"bench.py: rank {rank}/{world}: process group ({args.backend}, {os.environ.get('MASTER_ADDR')}:"
                             f"{os.environ.get('MASTER_PORT')}) failed within {args.init_timeout:.0f} s: {e}")
        comm_info = {"backend": args.backend, "ranks_seen": ranks_seen, "transport": args.comm,
                     "launcher": "bench.py" if os.environ.get("MXV_BENCH_SELF_LAUNCHED") else "external"}
        if args.backend == "nccl":
            try:
                comm_info["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception:  # noqa: BLE001
                comm_info["rccl_version"] = None
        if ranks_seen != world:
            raise SystemExit(f"bench.py: the communicator spans {ranks_seen} ranks, expected {world}")

    t_start = time.perf_counter()

    def trace(what):
        """Phase markers on stderr at N > 1 (never stdout: that carries the one JSON line): if a multi-GPU run stalls or a rank is slow,
        the log says where.  MXV_BENCH_TRACE=0 silences them, =1 forces them at N = 1."""
        flag = os.environ.get("MXV_BENCH_TRACE")
        if flag == "0" or (world == 1 and flag != "1"):
            return
        print(f"[bench rank {rank}/{world} +{time.perf_counter() - t_start:7.2f}s] {what}", file=sys.stderr, flush=True)

    trace(f"process group up ({comm_info.get('backend')}, ranks_seen={comm_info.get('ranks_seen', 1)}), device {local_rank}")
    from gym_amd.distributed import ShardedRollout

    total_envs = ENVS_TOTAL * (world if args.scaling == "weak" else 1)
    if total_envs % (4 * world):
        raise SystemExit(f"{total_envs} envs do not split into {world} shards of a multiple of 4 envs")
    local_envs = total_envs // world
    sr = ShardedRollout(ENV_ID, total_envs, rank=rank, world_size=world, device=local_rank, seed=0, action_seed=1,
                        reward_f32=args.compact_outputs, action_i32=args.compact_outputs, comm=args.comm)
    if solo_group:
        sr._force_collective = True     # dist.all_gather_into_tensor for real (ShardedRollout short-cuts a one-rank gather to a local copy)
    eng = sr.engine
    mode = "eager" if args.no_graph else args.mode
    sr.reset(seed=0)
    # [chunk][N] obs / reward / flags / actions, reused every chunk
    placement = None
    if args.placement in ("first", "off") or mode != "fused":   # (a one-launch-per-step mode: nothing to place)
        traj, placement = eng.trajectory_buffers(args.chunk, layout="separate"), {"kind": "first ordinary allocation"}
    else:
        from gym_amd import _native
        big = local_envs * args.chunk * 34 >= _native.SORTED_MIN_BYTES      # 2^17-env shards (8 GPUs) and larger are sorted by HBM class
        try:
            traj = eng.trajectory_buffers(args.chunk, layout=args.placement if big else "separate")
            placement = dict(getattr(eng, "last_placement", None) or {}) if big else {"kind": "ordinary allocations (set below 1 GiB)"}
            if big:
                placement["kind"] = {"sorted": "sorted (ordinary allocations classified with mxv_hbm_pair_probe)",
                                     "placed": "placed (mxv_placed_alloc)"}[args.placement]
        except (RuntimeError, MemoryError) as e:   # e.g. a device someone else is using: measure on ordinary allocations instead of dying
            torch.cuda.empty_cache()
            traj = eng.trajectory_buffers(args.chunk, layout="separate")
            placement = {"kind": "ordinary allocations", "error": f"{args.placement} placement failed: {e}"[:300]}
    trace(f"engine + trajectory tensors ready: {local_envs} envs, placement {placement.get('kind') if placement else None}"
          f" balanced={placement.get('balanced') if placement else None} parked_GiB={placement.get('parked_GiB') if placement else None}")
    launches = [0]
    since_gather = [0]
    issued = [0]
    gathers = [0]
    gathering = world > 1 or args.force_gather

    def run(steps, gather=True, every=None):
        """`steps` vector steps as chunk-step launches; at N > 1 the final tensors are all-gathered (asynchronously, overlapping
        the next launch) every time `every` (default --gather-every) steps have accumulated — the cadence does not depend on how
        `steps` was cut."""
        every = every or args.gather_every
        done = 0
        while done < steps:
            k = min(args.chunk, steps - done)
            sr.rollout_per_step(k, mode=mode, out=traj, record_actions=True)
            launches[0] += 1 if mode == "fused" else k
            done += k
            issued[0] += k
            since_gather[0] += k
            if gathering and gather and since_gather[0] >= every:
                sr.gather_async()
                since_gather[0] = 0
                gathers[0] += 1

    def fence():
        sr.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # Clock state first: the same workload until its rate has settled (time-based, so rank-dependent: no collective inside), THEN
    # reset(seed=0) — which restarts the step index and every env's reset stream — so that everything from here on is a pure
    # function of the arguments and config.work_check is reproducible.
    warm_s, warm_calls = (0.0, 0)
    if args.warm_max_s > 0:
        warm_s, warm_calls = warm_until_stable(lambda: sr.rollout_per_step(args.chunk, mode=mode, out=traj, record_actions=True),
                                               sr.synchronize, max_s=args.warm_max_s)
    trace(f"rate settled after {warm_s:.2f} s ({warm_calls} launches)")
    sr.reset(seed=0)
    # device spin-up: a fixed number of untimed steps; then W warmup steps, which also instantiate the hipGraph(s) and RCCL communicators
    # used in the timed region
    spin = spinup_steps(args.spinup_ms, args.chunk, local_envs)
    run(spin, gather=False)
    fence()   # ranks leave placement and spin-up at different times
    trace(f"spin-up done ({spin} steps), all ranks at the fence")
    since_gather[0] = 0
    run(max(args.warmup, 1))
    if gathering:
        sr.gather()
    fence()
    trace("warm-up done (first gather through the transport included)")

    repeats = args.repeats if args.repeats > 0 else timed_repeats(args.steps, args.chunk, local_envs, args.min_timed_ms, mode)
    timed_steps = args.steps * repeats

    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    fence()
    launches[0] = 0
    since_gather[0] = 0
    gathers[0] = 0
    t0 = time.perf_counter()
    ev0.record(eng.stream)
    run(timed_steps)
    ev1.record(eng.stream)
    if gathering:
        sr.wait_gather()
    fence()
    elapsed_local = time.perf_counter() - t0
    trace(f"timed region done: {timed_steps} steps in {elapsed_local * 1e3:.1f} ms")
    timed_launches, timed_gathers = launches[0], gathers[0]

    launch_ms = ev0.elapsed_time(ev1) / timed_launches  # avg step-kernel launch duration on the engine's stream
    steps_per_launch = timed_steps / timed_launches
    # what the timed region computed, in a form the oracle can reproduce (tests/test_gpu_bench_line.py): the last launch's flags and
    # actions of the first CHECK_ENVS envs, and how many env-steps of that launch ended an episode
    work_check = None
    if rank == 0 and mode == "fused":
        k_last = args.chunk if timed_steps % args.chunk == 0 else timed_steps % args.chunk
        with torch.cuda.stream(eng.stream):
            term, trunc, act = traj["terminated"][:k_last], traj["truncated"][:k_last], traj["actions"][:k_last]
            ended = int(((term | trunc) != 0).sum().item())
            c = min(CHECK_ENVS, local_envs)
            o64 = traj["obs"][:k_last, :c].to(torch.float64)
            work_check = {"what": "last launch of the timed region; checksum = bench.work_checksum(terminated, truncated, actions) over "
                                  f"its {k_last} steps x the first {c} envs; obs_abs_sum / obs_sq_sum / reward_sum over the same block "
                                  "(float64 sums of the float32 observations: the oracle reproduces them to 1e-6 relative, "
                                  "tests/test_gpu_bench_line.py; bit-level observation parity of this very instantiation: tests/test_gpu_soak.py)",
                          "obs_abs_sum": float(o64.abs().sum().item()), "obs_sq_sum": float((o64 * o64).sum().item()),
                          "reward_sum": float(traj["reward"][:k_last, :c].to(torch.float64).sum().item()),
                          "first_step_index": issued[0] - k_last, "steps": k_last, "envs": c,
                          "checksum": work_checksum(term[:, :c], trunc[:, :c], act[:, :c]) if eng.NA > 0 else None,
                          "autoresets_per_env_step": ended / float(k_last * local_envs),
                          "seed": 0, "action_seed": 1}

    # The gather by itself, blocking, after the timed region (rank 0's wall time of `gather_reps` back-to-back gathers between fences)
    # beside what the link model predicts for it: the first real 8-GPU run can be read in one glance.  And the older cadence — one
    # gather per launch, rounds 1-3 — as a second, shorter region, so that scaling figures stay comparable across rounds.
    gather_us = cadence_ab = None
    if gathering:
        sr._snapshots()
        shard_bytes = getattr(sr, "_shard_bytes", None) or sum(t.numel() * t.element_size() for t in eng.final_tensors())
        reps_g = 1 if args.backend == "gloo" else 4      # (a gloo gather of ranks that share one GPU is a multi-second CPU affair: once is enough)
        fence()
        tg = time.perf_counter()
        for _ in range(reps_g):
            sr.gather()
        fence()
        measured = (time.perf_counter() - tg) / reps_g * 1e6
        recv = (world - 1) * shard_bytes
        gather_us = {"measured_blocking": measured, "predicted": [recv / (XGMI_BUS_GBS[1] * 1e3) + 20.0, recv / (XGMI_BUS_GBS[0] * 1e3) + 20.0],
                     "bytes_received_per_rank": recv, "steps_of_this_shard_it_equals": measured / (launch_ms * 1e3 / steps_per_launch)}
        # (more than two ranks over gloo = ranks sharing one GPU in a rehearsal: every gather is a 4-s CPU affair, and the two-rank rehearsal
        # covers this region)
        if world > 1 and mode == "fused" and args.gather_every != args.chunk and not (args.backend == "gloo" and world > 2):
            ab_steps = max(args.chunk, (timed_steps // 4) // args.chunk * args.chunk)
            fence()
            since_gather[0] = 0
            ta = time.perf_counter()
            run(ab_steps, every=args.chunk)
            sr.wait_gather()
            fence()
            ab_local = torch.tensor([time.perf_counter() - ta], dtype=torch.float64)
            if args.backend == "nccl":
                ab_local = ab_local.cuda()
            dist.all_reduce(ab_local, op=dist.ReduceOp.MAX)
            cadence_ab = {"gather_every": args.chunk, "steps": ab_steps, "ms_per_step": float(ab_local.item()) / ab_steps * 1e3}
        trace(f"gather alone: {measured:.0f} us blocking; cadence A/B {cadence_ab}")

    # (the probe overwrites the trajectory tensors: it runs after work_check has read them)
    # what THIS rank's placement sustains for the kernel's store pattern with the physics removed (mxv_write_probe, include/mxv.h), into
    # the very tensors the timed region wrote: a rank whose tensors ended up in one HBM class shows here, not only in the job's maximum
    probe_us = None
    if mode == "fused" and not args.compact_outputs and local_envs % 1024 == 0:
        from gym_amd import _native
        torch.cuda.synchronize()
        probe_us = _native.write_probe(local_rank, local_envs, args.chunk, 20, traj["obs"], traj["reward"], traj["actions"],
                                       traj["terminated"], traj["truncated"])
    t = torch.tensor([elapsed_local], dtype=torch.float64, device="cuda")
    per_rank = [{"rank": rank, "device": local_rank, "kernel_us_per_step": launch_ms * 1e3 / steps_per_launch,
                 "timed_region_ms": elapsed_local * 1e3, "write_probe_us_per_step": probe_us,
                 "kernel_over_probe": (launch_ms * 1e3 / steps_per_launch / probe_us) if probe_us else None,
                 "placement": {k: placement.get(k) for k in ("kind", "balanced", "candidates", "parked_GiB", "chunks_created", "class_chunks",
                                                             "seconds", "stopped_by", "mode", "peak_GiB", "jumped_GiB", "chosen_us_per_step", "error") if k in placement}}]
    if world > 1:
        if args.backend == "gloo":
            tc = t.cpu()
            dist.all_reduce(tc, op=dist.ReduceOp.MAX)
            t = tc
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
    elapsed = float(t.item())

    out = None
    if rank == 0:
        value = total_envs * timed_steps / elapsed
        b_env_step = algorithmic_bytes_per_env_step(mode, steps_per_launch)
        algo_bytes = b_env_step * local_envs * steps_per_launch
        achieved = algo_bytes / (launch_ms * 1e-3) / 1e9
        traffic, traffic_source = read_traffic(mode, steps_per_launch, local_envs, args.compact_outputs)
        out = {
            "metric": "env-steps/sec at num_envs=2^20, CartPole-v1" if args.scaling == "strong"
                      else "env-steps/sec at num_envs=2^20 per GPU, CartPole-v1",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / timed_steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{ENV_ID}, num_envs={total_envs} ({local_envs} per GPU), on-device autoreset + "
                            "Philox4x32-10 sampled actions, fp64 state (BASELINE.json configs[1])",
                "num_envs_per_gpu": local_envs,
                "repeats": repeats,
                "timed_steps": timed_steps,
                "timed_region_ms": elapsed * 1e3,
                "launch": {"fused": f"fused: 1 launch per {args.chunk}-step chunk, state in registers",
                           "graph": "1 launch per step, hipGraph replay", "eager": "1 launch per step, eager"}[mode],
                "outputs": "per-step obs/reward/terminated/truncated/actions to [chunk][N] tensors"
                           + (" (f32 rewards, i32 actions)" if args.compact_outputs else " (f64 rewards, i64 actions: the reference's dtypes)"),
                "chunk": args.chunk,
                "placement": placement,
                "spinup": f"{warm_s:.2f} s of the workload until its rate settled ({warm_calls} launches; before reset(seed=0)), then {spin} "
                          f"untimed steps (nominally {args.spinup_ms:.0f} ms) before the {args.warmup} warmup steps",
                "parallelism": f"env-shard x{world}" + (f", async all-gather of the final tensors every {args.gather_every} steps" if world > 1 else ""),
                "ranks_seen": comm_info.get("ranks_seen", 1),
                "gathers_in_timed_region": timed_gathers,
                "gather_every": args.gather_every if gathering else None,
                "gather_transport": (args.comm if gathering else None),
                "gather_us": gather_us,
                "cadence_ab": cadence_ab,
                "comm": comm_info,
                "per_rank": per_rank,
                "work_check": work_check,
                "launch_info": eng.handle.last_launch(),     # mxv_last_launch: the kernel instantiation the timed region ran
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "rollout_kernel_v3<CartPole>" if mode == "fused" else "step_kernel<CartPole>",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "algorithmic_bytes_per_env_step": b_env_step,
                "algorithmic_bytes_per_launch": algo_bytes,
                "env_steps_per_launch": local_envs * steps_per_launch,
                "steps_per_launch": steps_per_launch,
                "avg_launch_us": launch_ms * 1e3,
            },
        }
        if probe_us:
            real_b = 34.0 * local_envs
            out["roofline"]["write_probe"] = {
                "what": "same store pattern, no physics (mxv_write_probe), same tensors",
                "us_per_step": probe_us, "real_GBs": real_b / probe_us / 1e3,
                "kernel_us_per_step": launch_ms * 1e3 / steps_per_launch,
                "kernel_over_probe": launch_ms * 1e3 / steps_per_launch / probe_us}

    sr.close()
    del traj
    torch.cuda.empty_cache()
    if rank == 0:
        for r in per_rank:       # one line per rank on stderr: what every rank measured on ITS tensors
            print("[bench per-rank] " + json.dumps(_sig({"rank": r["rank"], "device": r["device"], "kernel_us_per_step": r["kernel_us_per_step"],
                                                         "write_probe_us_per_step": r["write_probe_us_per_step"],
                                                         "placement.kind": (r.get("placement") or {}).get("kind"),
                                                         "placement.seconds": (r.get("placement") or {}).get("seconds"),
                                                         "placement.stopped_by": (r.get("placement") or {}).get("stopped_by")})), file=sys.stderr, flush=True)
        if world == 1 and not args.no_cpu_baseline and cpu_baseline is not None:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_steps)
        side = {}
        if world == 1 and mode == "fused" and not args.compact_outputs and not args.no_variants:
            from benchmarks.variants import run_all
            # reported beside the headline, never instead of it; one JSON object per group on stderr as it finishes
            out["variants"] = run_all(torch, args.chunk, emit=lambda name, res: print(
                "[bench variant] " + json.dumps({name: res}), file=sys.stderr, flush=True))
            side["variants"] = _write_json(args.variants_file, out["variants"])
        side["headline"] = _write_json(args.headline_file, out)
        out["details"] = {k: v for k, v in side.items() if v}
        print(json.dumps(compact_line(out)), file=json_out, flush=True)

    if world > 1 or solo_group:
        dist.barrier()
        dist.destroy_process_group()



def _write_json(path, obj):
    """Write a side file; returns the path as written relative to the repo, or None (a read-only tree costs the side file, not the line)."""
    if not path:
        return None
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
        with open(path, "w") as f:
            json.dump(obj, f, indent=1)
        return os.path.relpath(os.path.abspath(path), ROOT)
    except OSError as e:
        print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
        return None


