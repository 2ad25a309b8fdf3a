"""The secondary measurements of a one-GPU run (the other BASELINE.json configs, the contract-dtype twins, SURVEY.md §8(f)'s engines, the
learner-in-the-loop paths).  They are reported BESIDE the headline and never inside its line: every group is its own JSON object
(stderr + gpurun_out/bench_variants.json); the headline line carries one [us_per_step, roofline_frac] pair per group."""
from .common import ENV_ID, ENVS_TOTAL
from .fused import measure_fused, measure_mixed
from .loops import measure_numpy_loop, measure_policy_loop, measure_step_kernel, measure_step_loop
from .normalize import measure_normalize, measure_subenv_normalize
from .toy_text import measure_blackjack, measure_tabular, measure_toytext_episode_stats


def groups(torch, chunk):
    """[(name, thunk)] in the order they run (the hipGraph recording last)."""
    f = lambda env_id, n, **kw: (lambda: measure_fused(torch, env_id, n, chunk, **kw))   # noqa: E731
    half = 1 << 19
    return [
        ("configs2_pendulum", f("Pendulum-v1", half, valu=True)),
        ("configs2_mountaincar_continuous", f("MountainCarContinuous-v0", half, valu=True)),
        ("mountaincar", f("MountainCar-v0", half, valu=True)),
        ("configs3_acrobot_shard", f("Acrobot-v1", half, valu=True)),
        # the contract-dtype twins (SURVEY.md §8d prices 4-byte rewards and actions; the NumPy adapter widens at the API,
        # gym/vector/sync_vector_env.py:66-71): float32 rewards + int32 actions on the device tensors, every env kind
        ("compact_cartpole", f(ENV_ID, ENVS_TOTAL, compact=True, valu=True)),
        ("compact_pendulum", f("Pendulum-v1", half, compact=True)),
        ("compact_mountaincar_continuous", f("MountainCarContinuous-v0", half, compact=True)),
        ("compact_mountaincar", f("MountainCar-v0", half, compact=True)),
        ("compact_acrobot", f("Acrobot-v1", half, compact=True)),
        # the wrappers and toy_text engines behind the same library
        ("normalize", lambda: measure_normalize(torch, ENVS_TOTAL, 128)),
        ("subenv_normalize", lambda: measure_subenv_normalize(torch, ENVS_TOTAL)),
        ("frozenlake8x8", lambda: measure_tabular(torch, "FrozenLake8x8-v1", ENVS_TOTAL, 128)),
        ("taxi", lambda: measure_tabular(torch, "Taxi-v3", ENVS_TOTAL, 128)),
        ("compact_frozenlake8x8", lambda: measure_tabular(torch, "FrozenLake8x8-v1", ENVS_TOTAL, 128, compact=True)),
        ("compact_taxi", lambda: measure_tabular(torch, "Taxi-v3", ENVS_TOTAL, 128, compact=True)),
        ("blackjack", lambda: measure_blackjack(torch, ENVS_TOTAL, 128)),
        ("compact_blackjack", lambda: measure_blackjack(torch, ENVS_TOTAL, 128, compact=True)),
        ("toytext_episode_stats", lambda: measure_toytext_episode_stats(torch, ENVS_TOTAL, 64)),
        # the headline configuration placed under the 8-GiB cap (MXV_PLACEMENT=cheap: what the default does as soon as anybody else holds
        # device memory; on the otherwise empty device of a benchmark the default walks as far as it must): what the cap costs on this box
        ("placement_cheap", f(ENV_ID, ENVS_TOTAL, placement_mode="cheap")),
        ("configs4_mixed_share", lambda: measure_mixed(torch, 1 << 15, chunk)),
        ("strong_scaling_share_of_8", f(ENV_ID, ENVS_TOTAL // 8)),
        ("step_loop", lambda: {
            "what": "DeviceRollout.step(actions): one launch per vector step with caller-provided actions, 2^20 envs "
                    "(learner-in-the-loop; 66 algorithmic B per env-step)",
            "one_engine": measure_step_loop(torch, ENVS_TOTAL),
            "one_engine_compact": measure_step_loop(torch, ENVS_TOTAL, compact=True),
            # round 6: the observation buffer doubles as the float32 half of the state (DeviceRollout(obs_carries_state=True), mxv_adopt_obs)
            "one_engine_obs_carries_state": measure_step_loop(torch, ENVS_TOTAL, obs_carries_state=True),
            "one_engine_obs_carries_state_compact": measure_step_loop(torch, ENVS_TOTAL, compact=True, obs_carries_state=True),
            # the same launch on 4x the envs: its fixed part (4.9 us) amortised, the working set still inside the Infinity Cache (profiles/r6/r6i_*)
            "one_engine_obs_carries_state_compact_4x_envs": measure_step_loop(torch, 4 * ENVS_TOTAL, steps=300, compact=True, obs_carries_state=True),
            "two_half_engines": measure_step_loop(torch, ENVS_TOTAL, halves=2),
            "kernel": measure_step_kernel(torch, ENVS_TOTAL)}),
        ("numpy_loop", lambda: {"num_envs_2^20": measure_numpy_loop(ENVS_TOTAL, 60),
                                "configs0_num_envs_8": measure_numpy_loop(8, 1000)}),      # BASELINE.json configs[0]: the plumbing case
        ("policy_loop_4096_envs", lambda: measure_policy_loop(torch, 4096)),
    ]


def run_all(torch, chunk, emit=None):
    """Every group, each behind its own try: a failing secondary measurement costs neither the headline nor the other groups.
    emit(name, result) is called as each group finishes."""
    out = {}
    for name, fn in groups(torch, chunk):
        try:
            out[name] = fn()
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"[:400]}
            torch.cuda.empty_cache()
        if emit:
            emit(name, out[name])
    return out


_HEADLINE_OF = {"normalize": "normalize_obs", "subenv_normalize": "normalize_obs", "step_loop": "one_engine", "numpy_loop": "num_envs_2^20", "policy_loop_4096_envs": "recorded_in_a_hipgraph"}


def summary(v):
    """{group: [us_per_step, roofline frac or None]} — what the headline line carries of the groups (4 significant digits), or
    {group: "error"}."""
    s = {}
    for name, r in v.items():
        if not isinstance(r, dict) or "error" in r:
            s[name] = "error"
            continue
        r = r.get(_HEADLINE_OF.get(name), r) if name in _HEADLINE_OF else r
        us, frac = r.get("us_per_step"), (r.get("roofline") or {}).get("frac")
        s[name] = [float(f"{us:.4g}") if us else None, float(f"{frac:.4g}") if frac else None]
    return s
