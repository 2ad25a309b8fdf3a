"""N > 1 path on CPU: world_size-2 `gloo` process group, the product's ShardedRollout / MixedRollout driving an
oracle-backed engine (tests/oracle_engine.py).  Checks the partition rule, that the Philox streams
are indexed by GLOBAL env index (a 2-way sharding reproduces the 1-way run bit for bit) and the all-gather of the
final tensors of a chunk."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, total, steps, q):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle_engine import OracleRollout

        if case == "mixed":
            from gym_amd.mixed import MixedRollout

            mr = MixedRollout(total, engine_factory=OracleRollout, seed=5, action_seed=6)
            assert mr.world_size == world and mr.rank == rank
            mr.reset(seed=5)
            mr.rollout(steps)
            got = mr.gather()
            if rank == 0:
                q.put({k: [t.numpy().copy() for t in v] for k, v in got.items()})
            mr.close()
        else:
            from gym_amd.distributed import ShardedRollout

            sr = ShardedRollout(case, total, engine_factory=OracleRollout, seed=11, action_seed=12)
            assert (sr.env_offset, sr.local_envs) == (rank * total // world, total // world)
            sr.reset(seed=11)
            sr.rollout(steps // 2)
            sr.gather_async()          # overlapped gather of chunk 1 ...
            sr.rollout(steps - steps // 2)  # ... while chunk 2 steps
            first = [t.numpy().copy() for t in sr.wait_gather()]
            last = [t.numpy().copy() for t in sr.gather()]
            if rank == 0:
                q.put((first, last))
            sr.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(case, total, steps, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, total, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return out


def _single(env_id, total, steps_list, seed, action_seed):
    sys.path.insert(0, HERE)
    from oracle_engine import OracleRollout

    e = OracleRollout(env_id, total, seed=seed, action_seed=action_seed)
    e.reset(seed=seed)
    outs = []
    for k in steps_list:
        e.rollout(k)
        outs.append([t.numpy().copy() for t in e.final_tensors()])
    return outs


@pytest.mark.parametrize("env_id", ["CartPole-v1", "Pendulum-v1"])
def test_two_rank_sharding_equals_single_rank(env_id):
    total, steps = 256, 30
    first, last = _run(env_id, total, steps)
    ref_first, ref_last = _single(env_id, total, [steps // 2, steps - steps // 2], 11, 12)
    for got, ref in ((first, ref_first), (last, ref_last)):
        assert got[0].shape == (total, ref[0].shape[1])
        for g, r in zip(got, ref):
            assert np.array_equal(g, r)  # global-index Philox streams: sharding is invisible in the results
    assert last[2].sum() + last[3].sum() >= 0 and not np.array_equal(first[0], last[0])


def test_mixed_batch_two_ranks():
    from gym_amd.mixed import DEFAULT_MIX

    total, steps = 128, 12
    got = _run("mixed", total, steps)
    assert list(got) == list(DEFAULT_MIX)
    for s, env_id in enumerate(DEFAULT_MIX):
        (ref,) = _single(env_id, total // len(DEFAULT_MIX), [steps], 5 + 1000003 * s, 6 + 1000003 * s)
        for g, r in zip(got[env_id], ref):
            assert np.array_equal(g, r), env_id


def test_partition_rule():
    from gym_amd.distributed import partition

    assert partition(1 << 22, 8, 3) == (3 << 19, 1 << 19)
    assert [partition(64, 4, r) for r in range(4)] == [(0, 16), (16, 16), (32, 16), (48, 16)]
    with pytest.raises(ValueError):
        partition(100, 8, 0)   # not divisible
    with pytest.raises(ValueError):
        partition(24, 4, 0)    # shard of 6 envs is not a multiple of the Philox group (4)
