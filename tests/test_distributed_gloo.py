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

        if case.startswith("normalize:"):
            from gym_amd.distributed import ShardedRollout
            from gym_amd.normalize import RunningNormalizer
            from oracle_engine import OracleNormBackend

            env_id = case.split(":", 1)[1]
            sr = ShardedRollout(env_id, total, engine_factory=OracleRollout, seed=11, action_seed=12)
            obs0 = sr.reset(seed=11)
            tr = sr.rollout_per_step(steps)
            O = obs0.shape[1]
            rn = sr.make_normalizer(O, gamma=0.9, backend=OracleNormBackend(sr.local_envs, O))
            assert rn.world_size == world and rn.total_envs == total
            y0 = rn.normalize_obs(obs0.contiguous())                 # the reset batch, then the K step batches
            y = rn.normalize_obs(tr["obs"].contiguous())
            o = rn.normalize_rewards(tr["reward"].contiguous(), tr["terminated"].contiguous(), tr["truncated"].contiguous())
            parts = [None] * world
            dist.all_gather_object(parts, (y0.numpy(), y.numpy(), o.numpy(), rn.obs_rms.mean, rn.obs_rms.var,
                                           rn.return_rms.var, rn.obs_rms.count))
            if rank == 0:
                q.put(parts)
            sr.close()
        elif case == "mixed":
            from gym_amd.mixed import MixedRollout

            mr = MixedRollout(total, engine_factory=OracleRollout, seed=5, action_seed=6)
            assert mr.world_size == world and mr.rank == rank
            mr.reset(seed=5)
            mr.rollout(steps)
            got = mr.gather()
            if rank == 0:
                q.put({k: [t.numpy().copy() for t in v] for k, v in got.items()})
            mr.close()
        elif case.startswith("inkernel:"):
            from gym_amd.distributed import ShardedRollout
            from oracle_engine import OracleRolloutInKernelSnapshot

            sr = ShardedRollout(case.split(":", 1)[1], total, engine_factory=OracleRolloutInKernelSnapshot, seed=11, action_seed=12)
            assert sr._in_kernel
            sr.reset(seed=11)
            firsts = []
            for chunk in range(4):                      # gather_async -> rollout -> wait_gather, several times: both sets, both orders
                sr.rollout(steps)
                sr.gather_async()
                pend = sr._pending
                sr.rollout(steps)                       # flips _cur to the other snapshot set
                assert sr._cur != pend[1]
                work = sr._works[pend[1]]
                got = sr.wait_gather()
                assert work is None or work.is_completed()   # the gather that was pending has been waited for, not the other set's
                assert sr._works[pend[1]] is None
                firsts.append([t.numpy().copy() for t in got])
            if rank == 0:
                q.put(firsts)
            sr.close()
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
    import time

    deadline = time.time() + 300
    while q.empty():                      # a worker that died never fills the queue: fail instead of blocking for ever
        if any(p.exitcode not in (None, 0) for p in procs) or time.time() > deadline:
            for p in procs:
                p.kill()
            raise AssertionError(f"worker exit codes {[p.exitcode for p in procs]} before any result")
        time.sleep(0.05)
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


def test_gather_async_then_rollout_then_wait_gather_waits_for_the_pending_gather():
    """The documented overlap pattern on the in-kernel-snapshot path (what the HIP engine runs at world size > 1): the rollout
    issued between gather_async() and wait_gather() flips the snapshot set; wait_gather must still wait for (and return) the
    gather that was pending — the results are the odd chunks' finals of the unsharded run."""
    total, steps = 128, 7
    firsts = _run("inkernel:CartPole-v1", total, steps)
    refs = _single("CartPole-v1", total, [steps] * 8, 11, 12)
    for i, got in enumerate(firsts):
        for g, r in zip(got, refs[2 * i]):
            assert np.array_equal(g, r)


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


@pytest.mark.parametrize("env_id", ["CartPole-v1", "Pendulum-v1"])
def test_two_rank_running_normalizer_equals_single_rank(env_id):
    """SURVEY.md §8(f)-2 sharded: every rank reduces its shard to per-step column sums, the sums are all-gathered, every
    rank runs the same running update -> identical statistics on all ranks, equal to the unsharded run."""
    sys.path.insert(0, HERE)
    from oracle_engine import OracleRollout
    from oracle.oracle import RunningNorm

    total, steps = 128, 20
    parts = _run("normalize:" + env_id, total, steps)
    e = OracleRollout(env_id, total, seed=11, action_seed=12)
    obs0 = e.reset(seed=11).numpy()
    tr = {k: v.numpy() for k, v in e.rollout_per_step(steps).items()}
    rn = RunningNorm(total, obs0.shape[1], gamma=0.9, mode=1)
    y0 = rn.normalize_obs(obs0)
    y = rn.normalize_obs(tr["obs"])
    o = rn.normalize_rewards(tr["reward"], tr["terminated"], tr["truncated"])
    got_y0 = np.concatenate([p[0] for p in parts], axis=0)
    got_y = np.concatenate([p[1] for p in parts], axis=1)
    got_o = np.concatenate([p[2] for p in parts], axis=1)
    np.testing.assert_allclose(got_y0, y0, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(got_y, y, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(got_o, o, rtol=1e-12, atol=0)
    for a, b in zip(parts[0][3:], parts[1][3:]):
        assert np.array_equal(a, b)                       # bit-identical statistics on every rank
    np.testing.assert_allclose(parts[0][3], rn.obs_mean, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(parts[0][4], rn.obs_var, rtol=1e-12)
    assert parts[0][6] == rn.obs_count[0]
