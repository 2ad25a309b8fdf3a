"""The C ABI's own collective (mxv_comm_init / mxv_allgather_outputs: RCCL opened by libmxv.so itself, include/mxv.h) on the
one GPU gpurun exposes: world size 1 exercises the bootstrap (unique id, communicator, side stream, events, grouped
ncclAllGather launches) and the ordering against the engine's stream; the world_size-2 control flow of the gather is covered by
the gloo tests (tests/test_distributed_gloo.py) through the torch transport, which shares everything but the transport call."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_native_allgather_world_1_matches_the_outputs_and_overlaps_the_next_chunk():
    import torch

    from gym_amd.distributed import ShardedRollout

    n, K = 1 << 16, 32
    a = ShardedRollout("CartPole-v1", n, rank=0, world_size=1, device=0, seed=3, action_seed=4, comm="mxv")
    b = ShardedRollout("CartPole-v1", n, rank=0, world_size=1, device=0, seed=3, action_seed=4, comm="torch")
    assert a.engine.handle.comm_stream != 0 and a.engine.handle.comm_stream != a.engine.handle.stream
    for sr in (a, b):
        sr.reset(seed=3)
    for chunk in range(3):
        outs = []
        for sr in (a, b):
            sr.rollout_per_step(K)
            sr.gather_async()
            sr.rollout_per_step(K)                      # the next chunk is launched while the gather is in flight
            g = sr.wait_gather()
            sr.synchronize()
            outs.append([t.cpu().numpy().copy() for t in g])
        for x, y in zip(*outs):
            assert x.shape[0] == n and np.array_equal(x, y)
    obs, rew, term, trunc = a.gather()
    a.synchronize()
    fin = a.engine.final_tensors()
    assert torch.equal(obs, fin[0]) and torch.equal(rew, fin[1]) and torch.equal(term, fin[2]) and torch.equal(trunc, fin[3])
    a.close(), b.close()


def test_native_comm_through_the_plain_c_abi():
    """No torch.distributed anywhere: unique id -> comm_init -> grouped gather of arbitrary device buffers -> host wait."""
    import torch

    from gym_amd import _native

    n = 4096
    h = _native.Handle(_native.ACROBOT, n, 500, seed=1, action_seed=2)
    uid = _native.comm_unique_id()
    assert len(uid) == _native.COMM_ID_BYTES and any(uid)
    h.comm_init(0, 1, uid)
    dev = torch.device("cuda", 0)
    obs = torch.zeros((n, 6), dtype=torch.float32, device=dev)
    rew = torch.zeros(n, dtype=torch.float64, device=dev)
    term = torch.zeros(n, dtype=torch.uint8, device=dev)
    trunc = torch.zeros(n, dtype=torch.uint8, device=dev)
    all_obs, all_rew, all_term = torch.full_like(obs, -1), torch.full_like(rew, -1), torch.full_like(term, 9)
    torch.cuda.synchronize()
    h.reset(obs)
    h.step_sampled(obs, rew, term, trunc)
    h.allgather_outputs(obs, rew, term, None, all_obs, all_rew, all_term, None)     # truncated skipped (NULL pair)
    h.allgather_wait(host_sync=True)
    h.allgather_wait(host_sync=True, age=1)          # no such gather yet: a no-op
    h.sync()
    assert torch.equal(all_obs, obs) and torch.equal(all_rew, rew) and torch.equal(all_term, term)
    assert float(all_rew.min()) == -1.0 and float(all_rew.max()) <= 0.0
    h.comm_destroy()
    h.comm_init(0, 1, _native.comm_unique_id())        # a communicator can be rebuilt
    with pytest.raises(_native.MxvError):
        h.comm_init(1, 1, uid)                         # rank outside the world
    h.close()


@pytest.mark.parametrize("mode", ["fused", "eager", "graph"])
def test_final_snapshot_is_the_last_step_of_every_launch_mode(mode):
    """mxv_set_final_snapshot: the last of the K steps also lands in the caller's snapshot buffers — written by the fused kernel
    itself, copied on the handle's stream for the per-step launch modes — for [K][N] trajectories and final-tensor mode alike."""
    import torch

    from gym_amd.rollout import DeviceRollout

    n, K = 5000, 19
    r = DeviceRollout("Acrobot-v1", n, seed=2, action_seed=3, max_episode_steps=11)
    r.reset(seed=2)
    dev = r.device
    snap = [torch.full((n, 6), -7.0, dtype=torch.float32, device=dev), torch.full((n,), -7.0, dtype=torch.float64, device=dev),
            torch.full((n,), 9, dtype=torch.uint8, device=dev), torch.full((n,), 9, dtype=torch.uint8, device=dev)]
    torch.cuda.synchronize()
    r.handle.set_final_snapshot(*snap)
    out = r.rollout_per_step(K, mode=mode)
    r.synchronize()
    for s, key in zip(snap, ("obs", "reward", "terminated", "truncated")):
        assert torch.equal(s, out[key][K - 1]), (mode, key)
    fin = r.rollout(K, mode=mode)
    r.synchronize()
    for s, t in zip(snap, fin):
        assert torch.equal(s, t), mode
    r.handle.set_final_snapshot()                         # detach: later rollouts leave the buffers alone
    keep = [s.clone() for s in snap]
    r.rollout(K, mode=mode)
    r.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(keep, snap))
    r.close()


def test_gathers_alternate_snapshot_sets_and_survive_back_to_back_chunks():
    """ShardedRollout deposits the final tensors of chunk i into snapshot set i % 2 from inside the rollout kernel and gathers
    it while chunk i+1 (armed with the other set) runs: many chunks issued without any host synchronisation must gather, every
    time, exactly the final tensors of their own chunk — for both transports, and when several rollouts precede a gather."""
    import torch

    from gym_amd.distributed import ShardedRollout

    n, K, chunks = 1 << 15, 16, 12
    for comm in ("torch", "mxv"):
        sr = ShardedRollout("CartPole-v1", n, rank=0, world_size=1, device=0, seed=5, action_seed=6, comm=comm)
        ref = ShardedRollout("CartPole-v1", n, rank=0, world_size=1, device=0, seed=5, action_seed=6, comm="torch")
        sr.reset(seed=5), ref.reset(seed=5)
        got = []
        for c in range(chunks):
            sr.rollout_per_step(K)
            if c % 3 == 2:
                sr.rollout(K)                               # two rollouts before this gather: the same set is written twice
            sr.gather_async()
            g = sr.wait_gather()                            # GPU-side ordering only; the host runs ahead
            with torch.cuda.stream(sr.engine.stream):
                got.append([t.clone() for t in g])
        sr.synchronize()
        for c in range(chunks):
            ref.engine.rollout_per_step(K)
            if c % 3 == 2:
                ref.engine.rollout(K)
            ref.engine.synchronize()
            for x, y in zip(got[c], ref.engine.final_tensors()):
                assert torch.equal(x, y), (comm, c)
        sr.close(), ref.close()
