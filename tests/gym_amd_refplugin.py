"""pytest plugin for tests/test_reference_own_tests.py: the REFERENCE's own test files (under /root/reference/tests, run where they lie, in a
sub-process) meet the engine instead of the Python envs — gym.make / gym.vector.make of a classic-control id return `hip/<id>` (the
oracle-backed handle stands in for the device on the CPU box; the contract logic under test is the host side's)."""
import numpy as np

for _name, _val in (("bool8", np.bool_), ("float_", np.float64), ("alltrue", np.all)):      # NumPy-2 aliases the reference still uses
    if not hasattr(np, _name):
        setattr(np, _name, _val)

TABULAR = ("FrozenLake-v1", "FrozenLake8x8-v1", "Taxi-v3", "CliffWalking-v0")
if not hasattr(np, "cast"):      # np.cast[dtype](x), removed in NumPy 2 (tests/envs/test_action_dim_check.py:117)
    np.cast = type("_Cast", (), {"__getitem__": lambda self, dtype: (lambda x: np.asarray(x, dtype=dtype))})()

CLASSIC = ("CartPole-v0", "CartPole-v1", "Pendulum-v1", "Acrobot-v1", "MountainCar-v0", "MountainCarContinuous-v0")


def pytest_configure(config):
    import gym

    from gym_amd import _native, plugin
    from oracle_engine import FakeBlackjack, FakeHandle, FakeTab

    _native.Handle, _native.Tab, _native.Blackjack = FakeHandle, FakeTab, FakeBlackjack
    plugin.register_envs(gym)
    make, vector_make, counts = gym.make, gym.vector.make, {"make": 0, "vector_make": 0}

    def engine_make(id, **kwargs):
        name = id if isinstance(id, str) else getattr(id, "id", None)      # (EnvSpec.make() passes the spec itself)
        if name in CLASSIC + TABULAR + ("Blackjack-v1",):
            counts["make"] += 1
            return make("hip/" + name, **kwargs)
        return make(id, **kwargs)

    def engine_vector_make(id, num_envs=1, asynchronous=True, wrappers=None, disable_env_checker=None, **kwargs):
        if id not in CLASSIC + TABULAR:
            return vector_make(id, num_envs=num_envs, asynchronous=asynchronous, wrappers=wrappers, disable_env_checker=disable_env_checker, **kwargs)
        counts["vector_make"] += 1
        if wrappers is not None:
            kwargs["wrappers"] = wrappers
        return make("hip/" + id, num_envs=num_envs, disable_env_checker=True, **kwargs)

    gym.make, gym.vector.make = engine_make, engine_vector_make
    gym.envs.registration.make = engine_make          # what EnvSpec.make() calls (registration.py:163-165)
    config._gym_amd_counts = counts


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    c = config._gym_amd_counts
    terminalreporter.write_line(f"gym_amd_refplugin: gym.make -> engine {c['make']} times, gym.vector.make -> engine {c['vector_make']} times")
