"""The REFERENCE's own test files, run where they lie (/root/reference/tests, a sub-process with that tree as its root), against the engine:
tests/gym_amd_refplugin.py makes gym.make / gym.vector.make of a classic-control id return `hip/<id>` (oracle-backed handle on this CPU
box).  Everything the reference tests about the VectorEnv / Env surface and about ITS wrappers on top of an env — infos layout, final
observations, RecordEpisodeStatistics, VectorListInfo, TimeLimit, ClipAction, RescaleAction, Transform*, TimeAwareObservation, AutoReset,
step-API compatibility, VectorEnvWrapper forwarding, the env checker and the determinism rollout of tests/envs/test_envs.py — must pass unchanged.  Left out, by name: tests that reach into Python sub-envs
(`env.envs[i]`, `env_fns`) or assert WHICH wrapper classes gym.make stacked (`has_wrapper(env, OrderEnforcing)`: the engine enforces the
order itself, the hip/ ids are registered without the flag), and tests of the reference's own env classes (pygame rendering)."""
import os
import re
import subprocess
import sys

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "tests")), reason="the reference tree is only in the build container")

FILES = ["tests/vector/test_vector_env_info.py::test_vector_env_info", "tests/vector/test_vector_make.py::test_vector_make_num_envs", "tests/vector/test_vector_env_wrapper.py", "tests/wrappers/test_vector_list_info.py",
         "tests/wrappers/test_record_episode_statistics.py", "tests/wrappers/test_time_limit.py", "tests/wrappers/test_clip_action.py",
         "tests/wrappers/test_rescale_action.py", "tests/wrappers/test_transform_observation.py", "tests/wrappers/test_transform_reward.py",
         "tests/wrappers/test_time_aware_observation.py", "tests/wrappers/test_autoreset.py", "tests/wrappers/test_step_compatibility.py",
         "tests/wrappers/test_flatten_observation.py", "tests/envs/test_action_dim_check.py"]
# tests/envs/test_envs.py parametrises over the REGISTRY, where the plugin has put the hip/ ids: the reference's env checker and its
# determinism rollout (two envs, same seed: equal observations, rewards, flags, infos over 100 steps) run on them as on any other env
FILES += [f"tests/envs/test_envs.py::{t}[hip/{i}]" for t in ("test_envs_pass_env_checker", "test_env_determinism_rollout")
          for i in ("CartPole-v0", "CartPole-v1", "MountainCar-v0", "MountainCarContinuous-v0", "Pendulum-v1", "Acrobot-v1",
                    "FrozenLake-v1", "FrozenLake8x8-v1", "Taxi-v3", "CliffWalking-v0", "Blackjack-v1")]
DESELECT = ["tests/wrappers/test_record_episode_statistics.py::test_record_episode_statistics_with_vectorenv"]      # envs.env.envs[0].spec / env_fns


def test_the_references_own_tests_pass_on_the_engine():
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([REF, ROOT, os.path.join(ROOT, "tests")]))
    cmd = [sys.executable, "-m", "pytest", "-q", "-p", "gym_amd_refplugin", "-p", "no:cacheprovider", "-o", "addopts="] + FILES
    for d in DESELECT:
        cmd += ["--deselect", d]
    p = subprocess.run(cmd, cwd=REF, env=env, capture_output=True, text=True, timeout=600)
    tail = p.stdout[-3000:]
    assert p.returncode == 0, tail + p.stderr[-2000:]
    m = re.search(r"(\d+) passed", tail)
    c = re.search(r"gym.make -> engine (\d+) times, gym.vector.make -> engine (\d+) times", tail)
    assert m and int(m.group(1)) >= 96 and "failed" not in tail, tail
    assert c and int(c.group(1)) >= 40 and int(c.group(2)) >= 12, tail      # ... and they really met the engine
