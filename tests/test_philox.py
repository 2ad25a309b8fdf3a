"""Philox4x32-10: the oracle's CPU twin against the Random123 known-answer vectors, and (GPU) the device
implementation against the twin through the C ABI's RNG contract (include/mxv.h)."""
import numpy as np
import pytest

from helpers import ENV_IDS, ENV_NAMES, DISCRETE, LIMITS
from oracle import oracle

# Random123 kat_vectors, philox4x32 10 rounds (also SURVEY.md App. D)
KATS = [
    ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
     (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


@pytest.mark.parametrize("ctr,key,want", KATS)
def test_oracle_philox_known_answers(ctr, key, want):
    assert tuple(int(v) for v in oracle.philox4x32_10(ctr, key)) == want


def test_oracle_action_stream_contract():
    """g = env>>2, word = env&3: shards draw the numbers of the unsharded env; Discrete(n) = (w*n)>>32."""
    whole = oracle.OracleVecEnv(ENV_IDS["Acrobot"], 64, 500, action_seed=7)
    a = whole.sample_actions(t=3)
    lo = oracle.OracleVecEnv(ENV_IDS["Acrobot"], 32, 500, action_seed=7, env_offset=0).sample_actions(t=3)
    hi = oracle.OracleVecEnv(ENV_IDS["Acrobot"], 32, 500, action_seed=7, env_offset=32).sample_actions(t=3)
    assert np.array_equal(a, np.concatenate([lo, hi]))
    assert a.min() >= 0 and a.max() <= 2
    ctr = [5, 0, 3, 1 << 28]
    w = oracle.philox4x32_10(ctr, [7, 0])
    assert [int((int(x) * 3) >> 32) for x in w] == list(a[20:24])


def test_oracle_discrete2_actions_are_bit_slices_of_one_call_per_32_steps():
    """Discrete(2): ctr = (g, b = t >> 5, stream 6), action = bit (t & 31) of the env's word — also across block boundaries
    and for t beyond 2^32 * 32."""
    o = oracle.OracleVecEnv(ENV_IDS["CartPole"], 64, 500, action_seed=(9 << 32) | 7, env_offset=4096)
    for t in (0, 1, 31, 32, 33, 63, 64, 1000, (5 << 37) + 17):
        a = o.sample_actions(t=t)
        b = t >> 5
        for g_local in (0, 5, 15):
            g = (4096 >> 2) + g_local
            w = oracle.philox4x32_10([g, 0, b & 0xFFFFFFFF, ((b >> 32) & 0x0FFFFFFF) | (6 << 28)], [7, 9])
            assert [(int(x) >> (t & 31)) & 1 for x in w] == list(a[4 * g_local:4 * g_local + 4]), (t, g_local)
    # one bit per step is exactly uniform: over 32 steps every env plays each bit of its word once
    acts = np.stack([o.sample_actions(t=t) for t in range(64, 96)])           # block b = 2
    w = np.array([oracle.philox4x32_10([(4096 >> 2) + g, 0, 2, 6 << 28], [7, 9]) for g in range(16)], dtype=np.uint64).reshape(-1)
    assert np.array_equal((acts * (1 << np.arange(32, dtype=np.uint64))[:, None]).sum(axis=0), w)


def test_oracle_reset_stream_is_indexed_by_each_envs_own_reset_ordinal():
    """ctr = (k, 0, 0, 2 << 28), k = resets this env has had since seeding — whatever the vector step at which they happen."""
    n, seed, off = 16, 1000, 8
    o = oracle.OracleVecEnv(ENV_IDS["CartPole"], n, 3, seed=seed, env_offset=off)   # TimeLimit 3: autoresets every third step
    o.reset(seed=seed)

    def want(i, k):
        w = oracle.philox4x32_10([k, 0, 0, 2 << 28], [(seed + off + i) & 0xFFFFFFFF, (seed + off + i) >> 32])
        return np.array([-0.05 + (0.05 - (-0.05)) * ((float(x) + 0.5) * 2.0 ** -32) for x in w])

    assert all(np.array_equal(o.state[:, i], want(i, 0)) for i in range(n)) and np.all(o.episodes == 1)
    mask = np.zeros(n, np.uint8)
    mask[[2, 5]] = 1
    o.reset(mask=mask)                                     # a masked explicit reset advances only those envs
    assert np.array_equal(o.state[:, 2], want(2, 1)) and np.array_equal(o.state[:, 3], want(3, 0))
    for _ in range(3):
        o.step(o.sample_actions())
    assert np.all(o.elapsed == 0)                          # truncated + autoreset at the third step
    assert np.array_equal(o.state[:, 2], want(2, 2)) and np.array_equal(o.state[:, 3], want(3, 1))
    assert list(o.episodes[:6]) == [2, 2, 3, 2, 2, 3]
    o.reset(seed=seed)                                     # reseeding restarts every stream
    assert np.all(o.episodes == 1) and np.array_equal(o.state[:, 5], want(5, 0))


def test_oracle_action_distribution():
    o = oracle.OracleVecEnv(ENV_IDS["CartPole"], 1 << 16, 500, action_seed=123)
    a = o.sample_actions(t=0)
    assert abs(a.mean() - 0.5) < 0.01
    p = oracle.OracleVecEnv(ENV_IDS["Pendulum"], 1 << 16, 200, action_seed=5).sample_actions(t=9)
    assert p.dtype == np.float32 and p.min() >= -2 and p.max() <= 2 and abs(p.mean()) < 0.03
    assert abs(p.std() - 4 / np.sqrt(12)) < 0.02


def test_oracle_reset_distribution_and_ranges():
    o = oracle.OracleVecEnv(ENV_IDS["CartPole"], 1 << 15, 500, seed=11)
    obs = o.reset(seed=11)
    assert obs.shape == (1 << 15, 4) and np.all(np.abs(o.state) < 0.05)
    assert abs(o.state.mean()) < 1e-3 and abs(o.state.std() - 0.1 / np.sqrt(12)) < 5e-4
    a = oracle.OracleVecEnv(ENV_IDS["Acrobot"], 1024, 500, seed=3)
    a.reset(seed=3)
    assert np.array_equal(a.state, a.state.astype(np.float32).astype(np.float64))  # float32-rounded (acrobot.py:188-190)
    m = oracle.OracleVecEnv(ENV_IDS["MountainCar"], 1024, 200, seed=3)
    m.reset(seed=3)
    assert np.all((m.state[0] >= -0.6) & (m.state[0] < -0.4)) and np.all(m.state[1] == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ENV_NAMES)
def test_device_action_stream_matches_twin(name):
    import torch
    from gym_amd import _native

    n = 5000  # not a multiple of the tile: exercises the tail
    h = _native.Handle(ENV_IDS[name], n, LIMITS[name], action_seed=0xDEADBEEFCAFE, env_offset=1024)
    h.set_counters(12345678901, 0)
    dt = torch.int64 if DISCRETE[name] else torch.float32
    out = torch.zeros(n, dtype=dt, device="cuda")
    torch.cuda.synchronize()  # the handle launches on its own non-blocking stream
    h.sample_actions(out)
    h.sync()
    ref = oracle.OracleVecEnv(ENV_IDS[name], n, LIMITS[name], action_seed=0xDEADBEEFCAFE, env_offset=1024)
    want = ref.sample_actions(t=12345678901)
    assert np.array_equal(out.cpu().numpy(), want)  # bit-exact, floats included


@pytest.mark.gpu
@pytest.mark.parametrize("name", ENV_NAMES)
def test_device_reset_stream_matches_twin(name):
    from gym_amd import _native

    n = 3001
    h = _native.Handle(ENV_IDS[name], n, LIMITS[name], seed=2**40 + 17, env_offset=2048)
    obs = h.reset_host()
    st, el = h.get_state()
    ref = oracle.OracleVecEnv(ENV_IDS[name], n, LIMITS[name], seed=2**40 + 17, env_offset=2048)
    robs = ref.reset(seed=2**40 + 17)
    assert np.array_equal(st, ref.state)  # reset states are pure Philox + exact fp64 affine map: bit-exact
    assert np.all(el == 0)
    from helpers import ulps32, MAX_OBS_ULPS
    assert ulps32(obs, robs).max() <= MAX_OBS_ULPS
    # explicit per-env seeds and a masked second reset
    seeds = np.arange(n, dtype=np.uint64)[::-1].copy() * 977
    h.seed(0, seeds)
    ref.reset(seed=seeds)
    mask = (np.arange(n) % 3 == 0).astype(np.uint8)
    h.reset_host(mask=None)
    obs2 = h.reset_host(mask=mask)
    ref.reset(mask=mask)
    st2, _ = h.get_state()
    assert np.array_equal(st2, ref.state)
