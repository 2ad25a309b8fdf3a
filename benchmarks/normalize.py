"""NormalizeObservation / NormalizeReward over the trajectory tensors, stand-alone and with the moments fused into the rollout."""
from .common import *  # noqa: F401,F403
from .common import _event_us, _hbm, _spin


def measure_normalize(torch, envs, chunk, reps=6):
    """NormalizeObservation / NormalizeReward (gym/wrappers/normalize.py:57-144) on the [K][N] trajectory tensors of a
    fused CartPole rollout: per chunk, the batch moments of every step (one streaming read), then the affine map with the statistics as
    they stood after that step's update (read + write).  Algorithmic bytes per env-step: observations 4 O (sums) + 4 O + 4 O (apply,
    float32 out) = 48; rewards 8 + 2 (sums: reward + both flags) + 8 + 8 (apply) = 26."""
    from gym_amd import _native
    from gym_amd.rollout import DeviceRollout

    dr = DeviceRollout(ENV_ID, envs, seed=0, action_seed=1)
    dr.reset(seed=0)
    tr = dr.rollout_per_step(chunk, out=dr.trajectory_buffers(chunk, layout="separate"))
    dr.synchronize()
    s, O = dr.stream, dr.O
    no, nr = _native.Norm(O, envs, stream=s.cuda_stream), _native.Norm(1, envs, stream=s.cuda_stream)
    with torch.cuda.stream(s):
        y32 = torch.empty((chunk, envs, O), dtype=torch.float32, device=dr.device)
        o64 = torch.empty((chunk, envs), dtype=torch.float64, device=dr.device)
    out = {"workload": f"{ENV_ID}, num_envs={envs}, the [K={chunk}][N] trajectory tensors of one fused launch normalised in place of the "
                       "reference's per-step wrappers (running mean / var updated once per step, exactly their order)",
           "normalize_obs": _hbm(_event_us(torch, s, lambda: no.observations(chunk, tr["obs"], y32, True, 1e-8), reps, chunk), envs, 12 * O,
                                 kernels="mxv_norm.hip: obs sums (read 4 O) + scan + apply (read 4 O, write 4 O float32)"),
           "normalize_reward": _hbm(_event_us(torch, s, lambda: nr.rewards(chunk, tr["reward"], False, tr["terminated"], tr["truncated"], o64, 0.99, 1e-8),
                                              reps, chunk), envs, 26, kernels="mxv_norm.hip: discounted-return sums (read 8 + 2) + scan + apply (read 8, write 8)")}
    # the batch moments formed by the rollout itself (mxv_set_obs_partials): what NormalizeObservation then costs ON TOP of the rollout
    try:
        tr = None                                                  # (the set normalised above: its numbers are taken, its 9 GiB are needed)
        torch.cuda.empty_cache()
        trp = dr.trajectory_buffers(chunk, obs_partials=True)     # sorted by HBM class, as a caller gets them by default
        plain = {k: t for k, t in trp.items() if k != "obs_partials"}
        nf = _native.Norm(O, envs, stream=s.cuda_stream)
        r0 = _event_us(torch, s, lambda: dr.rollout_per_step(chunk, out=plain), reps, chunk)
        r1 = _event_us(torch, s, lambda: dr.rollout_per_step(chunk, out=trp), reps, chunk)
        sums = torch.empty((chunk, 2 * O), dtype=torch.float64, device=dr.device)

        def fused():
            nf.obs_sums_partials(chunk, trp["obs_partials"], trp["obs_partials"].shape[1], sums)
            nf.obs_apply(chunk, trp["obs"], y32, True, 1e-8, sums.unsqueeze(0), 1, envs)

        nfu = _event_us(torch, s, fused, reps, chunk)
        inc = nfu + (r1 - r0)
        b = 8 * O + 2 * 16 * O * trp["obs_partials"].shape[1] / envs   # apply: read 4 O + write 4 O; partials: 2 O doubles per tile, written + read
        out["normalize_obs_fused_moments"] = dict(_hbm(inc, envs, b, kernels="rollout_kernel_v3<..., STATS> writes per-tile column sums; mxv_norm.hip: tree over the "
                                                                              "partials + scan + apply (read 4 O, write 4 O float32); no pass reads the observations back"),
                                                  rollout_us_per_step=r0, rollout_with_partials_us_per_step=r1, normalize_from_partials_us_per_step=nfu,
                                                  separate_us_per_step=out["normalize_obs"]["us_per_step"],
                                                  note="us_per_step = what normalisation adds to the rollout: (rollout with partials - rollout) + tree + scan + apply")
        # ... and NormalizeReward's discounted returns (mxv_set_return_partials), alone and together with the observation moments
        nrf = _native.Norm(1, envs, stream=s.cuda_stream)

        class _Returns:      # what DeviceRollout.fuse_reward_normalizer needs of a normaliser: its returns array and its discount
            gamma = 0.99
            backend = nrf

        dr.fuse_reward_normalizer(_Returns)
        leaves = trp["obs_partials"].shape[1]
        with torch.cuda.stream(s):
            rp = torch.empty((chunk, leaves, 2), dtype=torch.float64, device=dr.device)
        rets, both = dict(plain, ret_partials=rp), dict(trp, ret_partials=rp)
        r2 = _event_us(torch, s, lambda: dr.rollout_per_step(chunk, out=rets), reps, chunk)
        r3 = _event_us(torch, s, lambda: dr.rollout_per_step(chunk, out=both), reps, chunk)
        rsums = torch.empty((chunk, 2), dtype=torch.float64, device=dr.device)

        def fused_reward():
            nrf.reward_sums_partials(chunk, rp, leaves, rsums)
            nrf.reward_apply(chunk, trp["reward"], False, o64, 1e-8, rsums.unsqueeze(0), 1, envs)

        nru = _event_us(torch, s, fused_reward, reps, chunk)
        out["normalize_reward_fused_moments"] = dict(_hbm(nru + (r2 - r0), envs, 16 + 2 * 16 * leaves / envs,
                                                          kernels="rollout_kernel_v3<..., STATS = 2> advances the discounted returns; tree + scan + apply (read 8, write 8)"),
                                                     rollout_with_partials_us_per_step=r2, normalize_from_partials_us_per_step=nru,
                                                     separate_us_per_step=out["normalize_reward"]["us_per_step"])
        out["rollout_and_both_normalisations"] = {"separate_us_per_step": r0 + out["normalize_obs"]["us_per_step"] + out["normalize_reward"]["us_per_step"],
                                                  "fused_us_per_step": r3 + nfu + nru, "rollout_with_both_partials_us_per_step": r3}
        dr.handle.set_obs_partials(None)
        dr.handle.set_return_partials(None, 0.0, None)            # nothing may point into nrf's returns any more
        nf.close(), nrf.close()
        del trp, plain, sums, rp, rets, both, rsums
    except Exception as e:  # noqa: BLE001
        out.setdefault("normalize_obs_fused_moments", {"error": f"{type(e).__name__}: {e}"[:300]})
        out.setdefault("normalize_reward_fused_moments", {"error": f"{type(e).__name__}: {e}"[:300]})
    no.close(), nr.close(), dr.close()
    del tr, y32, o64
    torch.cuda.empty_cache()
    return out
