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
    h.sync()
    assert torch.equal(all_obs, obs) and torch.equal(all_rew, rew) and torch.equal(all_term, term)
    assert float(all_rew.min()) == -1.0 and float(all_rew.max()) <= 0.0
    h.comm_destroy()
    h.comm_init(0, 1, _native.comm_unique_id())        # a communicator can be rebuilt
    with pytest.raises(_native.MxvError):
        h.comm_init(1, 1, uid)                         # rank outside the world
    h.close()
