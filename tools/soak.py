#!/usr/bin/env python3
"""Parity soak on the MI355X: the device-vs-oracle-twin rollout comparison of tests/test_gpu_parity.py (Philox actions,
TimeLimit, autoreset; masks / actions / reset states bit-exact, observations <= 2 float32 ulps) over many seeds, sizes and
time limits, plus the tabular and Blackjack twins.  Prints one line per case; exits non-zero on the first mismatch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    import test_gpu_parity as tp
    from helpers import ENV_NAMES

    t0 = time.time()
    total = 0
    extra = int(os.environ.get("SOAK_EXTRA_SEEDS", "0"))      # more device-vs-oracle-twin seeds (each: 5 env kinds x 2 shapes)
    for seed in (11, 222, 3333, 44444) + tuple(100003 + 7919 * i for i in range(extra)):
        for name in ENV_NAMES:
            for n, steps, limit in ((4097, 300, None), (1000, 120, 17)):
                nd = tp._rollout_compare(name, n=n, steps=steps, seed=seed, limit=limit, env_offset=(seed % 7) * 4096)
                total += n * steps
                print(f"ok {name:22s} seed={seed:<6d} n={n:<5d} steps={steps:<4d} limit={limit} dones={nd}", flush=True)
    # tabular + blackjack twins
    from gym_amd import _native
    from gym_amd.toy_text import TOY_TEXT_REGISTRY
    from oracle.oracle import OracleBlackjack, OracleTabEnv

    for seed in (5, 66, 777):
        for gid, spec in TOY_TEXT_REGISTRY.items():
            mdp = spec.build()
            n, steps = 5003, 150
            h = _native.Tab(mdp.num_states, mdp.num_actions, mdp.cum_prob, mdp.prob, mdp.next_state, mdp.reward, mdp.terminated,
                            mdp.initial_cum, n, 23, seed=seed, action_seed=seed + 1)
            o = OracleTabEnv(mdp.cum_prob, mdp.prob, mdp.next_state, mdp.reward, mdp.terminated, mdp.initial_cum, n, 23,
                             seed=seed, action_seed=seed + 1)
            assert np.array_equal(h.reset_host(), o.reset())
            for t in range(steps):
                r = o.step()
                obs, rew, term, trunc, prob, fin, fprob = h.step_host(r["actions"])
                assert np.array_equal(obs, r["obs"]) and np.array_equal(rew, r["reward"]) and np.array_equal(term, r["terminated"])
                assert np.array_equal(trunc, r["truncated"]) and np.array_equal(prob, r["prob"]), (gid, seed, t)
            h.close()
            total += n * steps
            print(f"ok {gid:22s} seed={seed}", flush=True)
        hb = _native.Blackjack(4099, sab=True, seed=seed, action_seed=seed + 1)
        ob = OracleBlackjack(4099, sab=True, seed=seed, action_seed=seed + 1)
        assert np.array_equal(hb.reset_host(), ob.reset())
        for t in range(200):
            r = ob.step()
            obs, rew, term, trunc, fin = hb.step_host(r["actions"])
            assert np.array_equal(obs, r["obs"]) and np.array_equal(rew, r["reward"]) and np.array_equal(term, r["terminated"])
        hb.close()
        total += 4099 * 200
        print(f"ok Blackjack-v1 seed={seed}", flush=True)
    # Blackjack, round-5 kernel and draw contract: fused sampled rollouts (both dtype sets, every rule set, TimeLimits, ragged sizes, shard
    # offsets, random launch lengths) against the oracle twin stepping with its own sampled actions — every output of every step
    from gym_amd.toy_text import BlackjackRollout

    rng_b = np.random.default_rng(int(os.environ.get("SOAK_RANDOM_SEED", "20260924")))
    for case in range(int(os.environ.get("SOAK_BLACKJACK_CASES", "24"))):
        n = int(rng_b.choice([1, 63, 64, 65, 255, 256, 257, 4099, 70001]))
        rules = [dict(sab=True), dict(natural=True, sab=False), dict(natural=False, sab=False)][case % 3]
        limit = None if rng_b.random() < 0.5 else int(rng_b.integers(1, 6))
        seed, aseed, off = int(rng_b.integers(1 << 40)), int(rng_b.integers(1 << 40)), int(rng_b.integers(0, 1 << 30)) * 4
        compact = bool(case & 1)
        r = BlackjackRollout(n, seed=seed, action_seed=aseed, env_offset=off, compact=compact, max_episode_steps=limit, **rules)
        o = OracleBlackjack(n, seed=seed, action_seed=aseed, env_offset=off, max_episode_steps=limit, natural=rules.get("natural", False),
                            sab=rules.get("sab", False))
        assert np.array_equal(r.reset(seed=seed).cpu().numpy(), o.reset(seed=seed))
        games = 0
        for launch in range(3):
            K = int(rng_b.integers(1, 70))
            out = r.rollout_per_step(K, out=r.trajectory_buffers(K, layout="separate", want_final=True))
            r.synchronize()
            d = {k: v.cpu().numpy() for k, v in out.items()}
            for k in range(K):
                w = o.step()
                assert np.array_equal(d["actions"][k], w["actions"]) and np.array_equal(d["obs"][k], w["obs"]), (case, launch, k)
                assert np.array_equal(d["reward"][k].astype(np.float64), w["reward"]), (case, launch, k)
                assert np.array_equal(d["terminated"][k].astype(bool), w["terminated"].astype(bool))
                assert np.array_equal(d["truncated"][k].astype(bool), w["truncated"].astype(bool))
                m = w["final_mask"].astype(bool)
                assert np.array_equal(d["final_obs"][k][:, m], w["final_obs"][:, m])
                games += int(m.sum())
            total += n * K
        r.close()
        print(f"ok Blackjack fused vs twin case={case:<3d} n={n:<6d} {'compact' if compact else 'ref    '} rules={rules} limit={limit} games={games}", flush=True)
    # long horizons: the fused rollout (rollout_kernel_v3: K steps per launch, ready-made resets in LDS, look-ahead refills) against
    # one launch per step (step_kernel), bit for bit over thousands of steps — every output of every step, then state, elapsed
    # steps, reset ordinals and counters
    import torch
    from gym_amd.rollout import DeviceRollout
    from helpers import GYM_IDS

    for name in ENV_NAMES:
        for n, chunks, K, limit in ((6000, 8, 256, None), (70001, 3, 192, 40)):
            kw = {} if limit is None else dict(max_episode_steps=limit)
            a = DeviceRollout(GYM_IDS[name], n, seed=91, action_seed=92, env_offset=3 << 20, **kw)
            b = DeviceRollout(GYM_IDS[name], n, seed=91, action_seed=92, env_offset=3 << 20, **kw)
            a.reset(seed=91), b.reset(seed=91)
            dones = 0
            for c in range(chunks):
                fa = a.rollout_per_step(K, mode="fused")
                fb = b.rollout_per_step(K, mode="eager")
                a.synchronize(), b.synchronize()
                for key in ("obs", "reward", "terminated", "truncated", "actions"):
                    assert torch.equal(fa[key], fb[key]), (name, n, c, key)
                dones += int((fa["terminated"] | fa["truncated"]).sum())
            for x, y in zip(a.handle.get_state(), b.handle.get_state()):
                assert np.array_equal(x, y), (name, n)
            assert np.array_equal(a.handle.get_episodes(), b.handle.get_episodes())
            assert a.handle.get_counters() == b.handle.get_counters()
            a.close(), b.close()
            total += n * chunks * K
            print(f"ok fused==per-step {name:22s} n={n:<6d} steps={chunks * K:<5d} limit={limit} dones={dones}", flush=True)
    # randomized shapes: sampled rollout (fused) == one launch per step (eager) == tape-driven rollout fed the recorded actions,
    # every output of every step and the final state, over random env kinds, sizes, chunk lengths, time limits, shard offsets
    # and dtype sets
    rng = np.random.default_rng(int(os.environ.get("SOAK_RANDOM_SEED", "20260923")))
    for case in range(int(os.environ.get("SOAK_RANDOM_CASES", "80"))):
        name = ENV_NAMES[int(rng.integers(len(ENV_NAMES)))]
        n = int(rng.choice([1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 1000, 4097])) if rng.random() < 0.5 else int(rng.integers(1, 6000))
        K = int(rng.integers(1, 90))
        limit = None if rng.random() < 0.3 else int(rng.integers(1, 30))
        off = int(rng.integers(0, 1 << 34)) * 4
        compact = bool(rng.random() < 0.3)
        kw = dict(seed=int(rng.integers(1 << 30)), action_seed=int(rng.integers(1 << 30)), env_offset=off, reward_f32=compact,
                  action_i32=compact)
        if limit is not None:
            kw["max_episode_steps"] = limit
        envs = [DeviceRollout(GYM_IDS[name], n, **kw) for _ in range(3)]
        for e in envs:
            e.reset(seed=kw["seed"])
        for rep in range(2):
            want_final = bool(rng.random() < 0.5)
            fa = envs[0].rollout_per_step(K, mode="fused", out=envs[0].trajectory_buffers(K, want_final=want_final))
            fb = envs[1].rollout_per_step(K, mode="eager", out=envs[1].trajectory_buffers(K, want_final=want_final))
            envs[0].synchronize(), envs[1].synchronize()
            keys = ["obs", "reward", "terminated", "truncated", "actions"]
            for key in keys:
                assert torch.equal(fa[key], fb[key]), (case, name, n, K, limit, key)
            done = (fa["terminated"] | fa["truncated"]).bool()
            if want_final:
                assert torch.equal(fa["final_obs"][done], fb["final_obs"][done]), (case, name, "final_obs")
            if K > 1:
                fc = envs[2].rollout_tape(fa["actions"].clone(), out=envs[2].trajectory_buffers(K, want_final=want_final))
            else:
                o, r, te, tr = envs[2].step(fa["actions"][0].clone(), want_final=want_final)
                fc = {"obs": o[None], "reward": r[None], "terminated": te[None], "truncated": tr[None]}
            envs[2].synchronize()
            for key in keys[:4]:
                assert torch.equal(fa[key], fc[key]), (case, name, n, K, limit, "tape", key)
        for other in envs[1:]:
            for x, y in zip(envs[0].handle.get_state(), other.handle.get_state()):
                assert np.array_equal(x, y), (case, name)
            assert np.array_equal(envs[0].handle.get_episodes(), other.handle.get_episodes())
        for e in envs:
            e.close()
        total += 2 * 3 * n * K
    print(f"ok randomized fused == per-step == tape: {case + 1} cases", flush=True)
    # round 6: step(actions) in its four forms — ordinary, compact outputs, the observation carries the state (mxv_adopt_obs), both — fed the
    # SAME actions, below and above the size where the launch switches to one env per lane (2^19): every output of every step and the
    # final fp64 state bit for bit (compact rewards: the float32 cast of the ordinary ones), through autoresets and partial resets
    for gid in ("CartPole-v1", "MountainCar-v0", "MountainCarContinuous-v0"):
        for n in (70001, (1 << 19) + 77):
            forms = [DeviceRollout(gid, n, seed=7, action_seed=8, max_episode_steps=25, reward_f32=c, action_i32=c, obs_carries_state=h)
                     for c, h in ((False, False), (True, False), (False, True), (True, True))]
            for e in forms:
                e.reset(seed=7)
            steps = int(os.environ.get("SOAK_STEP_FORMS_STEPS", "150"))
            for t in range(steps):
                forms[0].sample_actions()
                forms[0].synchronize()
                act = forms[0].actions.clone()
                torch.cuda.synchronize()
                outs = [e.step(act.to(e.actions.dtype)) for e in forms]
                for e in forms:
                    e.synchronize()
                for c, o in enumerate(outs[1:], 1):
                    assert torch.equal(o[0], outs[0][0]) and torch.equal(o[2], outs[0][2]) and torch.equal(o[3], outs[0][3]), (gid, n, t, c)
                    assert torch.equal(o[1].double(), outs[0][1].to(o[1].dtype).double()), (gid, n, t, c, "reward")
                    assert torch.equal(forms[c].final_obs, forms[0].final_obs), (gid, n, t, c, "final_obs")
                if t == steps // 2:
                    mask = (torch.arange(n, device="cuda") % 5 == 0).to(torch.uint8)
                    torch.cuda.synchronize()
                    first = forms[0].reset(mask=mask).clone()
                    for e in forms[1:]:
                        assert torch.equal(e.reset(mask=mask), first)
            ref = forms[0].handle.get_state()
            for e in forms[1:]:
                st = e.handle.get_state()
                assert np.array_equal(st[0], ref[0]) and np.array_equal(st[1], ref[1]) and np.array_equal(e.handle.get_episodes(), forms[0].handle.get_episodes())
            for e in forms:
                e.close()
            total += 4 * n * steps
            print(f"ok step(actions) four forms {gid:26s} n={n:<7d} steps={steps}", flush=True)
    print(f"soak passed: {total:.3e} env-steps compared in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
