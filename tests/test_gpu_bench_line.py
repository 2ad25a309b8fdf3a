"""The bench line is self-evidencing: config.work_check (what the LAST launch of the timed region computed, for the first 4096
envs) is reproduced here from the oracle twin with the same seeds — the timed region did the work it claims; and the line carries
the measurements the round-2 review asked for (reference baseline from a committed run, per-config variants with rooflines)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def run(tmp_path_factory):
    """`python bench.py --gpus 1 --steps 20 --warmup 5`, the driver's command (plus a short CPU sample and side files in a temporary
    directory): (the stdout line parsed, the raw stdout, the full variant records from the side file, stderr)."""
    d = tmp_path_factory.mktemp("bench")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-sample-steps", "4",
                        "--variants-file", str(d / "variants.json"), "--headline-file", str(d / "headline.json")], cwd=ROOT,
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads(p.stdout), p.stdout, json.load(open(d / "variants.json")), json.load(open(d / "headline.json")), p.stderr


@pytest.fixture(scope="module")
def line(run):
    return run[0]


def test_stdout_is_one_line_the_driver_can_keep(run):
    """VERDICT r4, Next #1: the driver's record keeps the last 8 KB of stdout and round 4's 24.9-KB line fell out of it.  stdout is ONE
    JSON line under 4 KB that parses alone and carries the contract fields, roofline and cpu_baseline; everything else is beside it."""
    import bench

    line, raw, variants, headline, err = run
    assert raw.endswith("\n") and len(raw.strip().splitlines()) == 1 and len(raw.encode()) < bench.LINE_LIMIT, len(raw)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "variants"):
        assert k in line, k
    assert (line["steps"], line["warmup"], line["n_gpus"], line["unit"], line["dtype"]) == (20, 5, 1, "env-steps/s", "f64")
    assert line["value"] == pytest.approx((1 << 20) / (line["ms_per_step"] * 1e-3), rel=1e-6)
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_env_step", "avg_launch_us",
              "kernel_over_probe"):
        assert k in line["roofline"], k
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] >= cb["single_core_value"] > 1e6 and cb["reference_python"]["value"] > 1e4
    # one [us_per_step, roofline_frac] pair per secondary measurement; the full records are in the side file and on stderr
    assert set(line["variants"]) == set(variants) and len(variants) >= 20
    assert line["variants"]["compact_cartpole"][0] == pytest.approx(variants["compact_cartpole"]["us_per_step"], rel=1e-3)
    assert line["variants"]["blackjack"][1] == pytest.approx(variants["blackjack"]["roofline"]["frac"], rel=1e-3)
    assert err.count("[bench variant] ") == len(variants) and "[bench per-rank] " in err
    for l in err.splitlines():
        if l.startswith("[bench variant] "):
            assert len(l) < bench.LINE_LIMIT + 16, (len(l), l[:80])
            json.loads(l[len("[bench variant] "):])
    # the long-form copy holds what the line dropped (prose, placement reports) and the same numbers
    assert headline["value"] == line["value"] and "what" in headline["config"]["work_check"] and headline["variants"] == variants
    assert headline["roofline"]["frac"] == pytest.approx(line["roofline"]["frac"], rel=1e-6)


def test_work_check_is_reproduced_by_the_oracle(line):
    import bench
    from oracle.oracle import OracleVecEnv

    wc = line["config"]["work_check"]
    n, K, t0 = wc["envs"], wc["steps"], wc["first_step_index"]
    assert n == 4096 and K == 256 and t0 == bench.spinup_steps(150.0, 256, 1 << 20) + 5 + line["config"]["timed_steps"] - 256
    o = OracleVecEnv(0, n, 500, seed=wc["seed"], action_seed=wc["action_seed"])
    o.reset(seed=wc["seed"])
    o.rollout(t0)
    term, trunc, act = (np.zeros((K, n), dtype=np.uint8) for _ in range(3))
    obs_abs = obs_sq = rew = 0.0
    for k in range(K):
        a = o.sample_actions()
        ob, rw, te, tr, _, _ = o.step(a)
        term[k], trunc[k], act[k] = te, tr, a
        ob = ob.astype(np.float64)
        obs_abs, obs_sq, rew = obs_abs + np.abs(ob).sum(), obs_sq + (ob * ob).sum(), rew + rw.sum()
    assert bench.work_checksum(term, trunc, act) == wc["checksum"]
    # observations and rewards of the same block (the oracle ran 10^4 free steps from the same seeds: last-bit libm differences have
    # had 22-step episodes to grow in, hence a relative bar instead of bits; tests/test_gpu_soak.py holds the bits of this instantiation)
    assert wc["reward_sum"] == rew == float(K * n)
    assert abs(wc["obs_abs_sum"] - obs_abs) <= 1e-6 * obs_abs and abs(wc["obs_sq_sum"] - obs_sq) <= 1e-6 * obs_sq
    li = line["config"]["launch_info"]
    assert (li["kernel"], li["envs_per_lane"], li["safe"], li["out_mode"], li["tape"], li["steps"]) == (1, 2, 0, 1, 0, 256), li
    assert abs(wc["autoresets_per_env_step"] - (term | trunc).mean()) < 0.004       # 2^20 envs vs the first 4096 of them
    assert 0.035 < wc["autoresets_per_env_step"] < 0.055                           # random-policy CartPole: ~ 1 / 22 steps


def test_line_carries_what_the_review_asked_for(run):
    line, _, v, headline, _ = run
    assert line["n_gpus"] == 1 and line["config"]["ranks_seen"] == 1 and line["dtype"] == "f64"
    roof, cfg = line["roofline"], line["config"]
    assert roof["frac"] == pytest.approx(roof["achieved"] / 8000.0) and 0.3 < roof["frac"] < 0.9
    # the product default (MXV_PLACEMENT unset = auto): this benchmark process has the device to itself, so the placement walks as far as it
    # must and releases what it parked; beside anybody else's memory it would stop at 8 GiB — what that cap costs on
    # this box is variants.placement_cheap against the headline
    assert cfg["placement"]["kind"].startswith("sorted") and cfg["placement"]["mode"] == "auto->search"
    assert cfg["placement"]["balanced"] is True and cfg["placement"]["parked_GiB"] <= 112
    pc = v["placement_cheap"]
    assert pc["placement"]["mode"] == "cheap" and pc["placement"]["parked_GiB"] <= 8.0
    assert 0.93 * line["ms_per_step"] * 1e3 <= pc["us_per_step"] <= 1.35 * line["ms_per_step"] * 1e3      # (balanced by luck: equal; all in one class: +17 %)
    ref = headline["cpu_baseline"]["reference_python"]
    assert ref["source"].startswith("profiles/reference_cpu_baseline.json") and ref["value"] > 1e4
    assert not [k for k, x in v.items() if isinstance(x, dict) and "error" in x], v
    for key in ("configs2_pendulum", "configs2_mountaincar_continuous", "mountaincar", "configs3_acrobot_shard", "compact_cartpole",
                "compact_pendulum", "compact_mountaincar_continuous", "compact_mountaincar", "compact_acrobot"):
        assert v[key]["roofline"]["frac"] > 0.1 and v[key]["write_probe"]["kernel_over_probe"] > 0.9, key
        li = v[key]["launch_info"]
        assert li["kernel"] == 1 and li["safe"] == 0 and li["out_mode"] == (2 if key.startswith("compact") else 1), (key, li)
    # the contract dtypes store what §8(d) prices: 4 O + 4 + 4 + 2 bytes per env-step
    twins = {"compact_pendulum": ("configs2_pendulum", 3), "compact_mountaincar_continuous": ("configs2_mountaincar_continuous", 2),
             "compact_mountaincar": ("mountaincar", 2), "compact_acrobot": ("configs3_acrobot_shard", 6)}
    for key, (ref_key, o) in twins.items():
        assert v[key]["roofline"]["stored_bytes_per_env_step"] == 4 * o + 10, key
        both_balanced = all((v[k].get("placement") or {}).get("balanced") for k in (key, ref_key))
        # fewer bytes per env-step cannot be slower — where both sets ended up sorted by HBM class (one that did not runs 10-20 % slower)
        assert v[key]["value"] >= (0.95 if both_balanced else 0.75) * v[ref_key]["value"], (key, v[key].get("placement"), v[ref_key].get("placement"))
    assert v["compact_cartpole"]["roofline"]["stored_bytes_per_env_step"] == 26 and v["compact_cartpole"]["value"] >= line["value"]
    rv = v["configs3_acrobot_shard"]["roofline_valu"]
    assert rv["source"].startswith("profiles/valu_") and rv["frac"] > 0.5 and 500 < rv["valu_instructions_per_env_step"] < 900
    for key in ("compact_frozenlake8x8", "compact_taxi"):                # the table engine with the contract dtypes: 18 B stored
        assert v[key]["value"] >= 0.95 * v[key.replace("compact_", "")]["value"] and v[key]["roofline"]["frac"] > 0.4, key
    for key in ("frozenlake8x8", "taxi", "compact_frozenlake8x8", "compact_taxi"):
        assert v[key]["kernel"].startswith("tab_traj_kernel"), key          # the specialised trajectory kernel is the one measured
    for key in ("frozenlake8x8", "taxi", "blackjack"):                    # SURVEY.md §8(f)-4 engines, driver-run
        assert v[key]["value"] > 2e10 and 0.05 < v[key]["roofline"]["frac"] < 1.0, key
    for key in ("normalize_obs", "normalize_reward"):                     # §8(f)-2
        assert v["normalize"][key]["roofline"]["frac"] > 0.15, key
    fm = v["normalize"]["normalize_obs_fused_moments"]                    # the batch moments formed by the rollout: cheaper than the second pass
    assert "error" not in fm and fm["us_per_step"] < fm["separate_us_per_step"] and fm["rollout_with_partials_us_per_step"] < 1.35 * fm["rollout_us_per_step"]
    fr, fb = v["normalize"]["normalize_reward_fused_moments"], v["normalize"]["rollout_and_both_normalisations"]
    assert "error" not in fr and fr["us_per_step"] < fr["separate_us_per_step"] and fb["fused_us_per_step"] < fb["separate_us_per_step"]
    nl = v["numpy_loop"]                                                  # SURVEY §8(d)'s third number: PCIe- and Python-inclusive
    assert nl["num_envs_2^20"]["value"] > 2e8 and nl["configs0_num_envs_8"]["value"] > 5e4 and nl["num_envs_2^20"]["episodes_ended"] > 0
    assert v["configs4_mixed_share"]["value"] > 1e10
    share = v["strong_scaling_share_of_8"]                    # 2^17 envs: what each GPU of an 8-GPU strong-scaling job steps
    assert share["placement"]["balanced"] is True and share["us_per_step"] * 8 < 1.35 * line["ms_per_step"] * 1e3
    sl = v["step_loop"]
    assert sl["one_engine"]["roofline"]["algorithmic_bytes_per_env_step"] == 66
    assert sl["one_engine"]["roofline"]["frac"] > 0.38        # round 2's loop: 0.34 (a cross-stream wait per step), kernel 0.44
    assert sl["one_engine_compact"]["value"] >= 0.97 * sl["one_engine"]["value"]
    pl = v["policy_loop_4096_envs"]                          # the launch-bound regime: the loop recorded once in a caller's hipGraph
    assert pl["recorded_in_a_hipgraph"]["us_per_step"] < pl["one_call_per_step"]["us_per_step"] and pl["speedup"] > 1.3
    assert min(pl["episodes_ended"]) > 0
