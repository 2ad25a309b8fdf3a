"""Env ids, time limits, single-env spaces and attribute tables of the classic-control path.

Mirrors what gym.make / gym.vector.make resolve for these ids: the registry entries of
gym/envs/__init__.py:11-50 (id -> entry point, max_episode_steps, reward_threshold) and the
spaces each env's __init__ builds.  Only the ids on the hot path exist here; everything else
of the reference's registry (gym/envs/registration.py) is out of scope.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np

from . import error
from .spaces import Box, Discrete

CARTPOLE, PENDULUM, ACROBOT, MOUNTAINCAR, MOUNTAINCAR_CONT = range(5)

_F32MAX = np.finfo(np.float32).max


@dataclass
class EnvSpec:
    id: str
    kind: int
    max_episode_steps: Optional[int]
    reward_threshold: Optional[float] = None
    kwargs: dict = field(default_factory=dict)


# gym/envs/__init__.py:11-50
registry: Dict[str, EnvSpec] = {
    "CartPole-v0": EnvSpec("CartPole-v0", CARTPOLE, 200, 195.0),
    "CartPole-v1": EnvSpec("CartPole-v1", CARTPOLE, 500, 475.0),
    "MountainCar-v0": EnvSpec("MountainCar-v0", MOUNTAINCAR, 200, -110.0),
    "MountainCarContinuous-v0": EnvSpec("MountainCarContinuous-v0", MOUNTAINCAR_CONT, 999, 90.0),
    "Pendulum-v1": EnvSpec("Pendulum-v1", PENDULUM, 200, None),
    "Acrobot-v1": EnvSpec("Acrobot-v1", ACROBOT, 500, -100.0),
}


def spec(env_id: str) -> EnvSpec:
    try:
        return registry[env_id]
    except KeyError:
        raise error.UnregisteredEnv(
            f"No HIP classic-control engine for id {env_id!r}; supported: {sorted(registry)}") from None


# Attribute name -> index into the engine's parameter vector (include/mxv.h), per kind.  These are the
# attributes the reference's env objects hold (cartpole.py:90-102, pendulum.py:95-101, acrobot.py:143-165,
# mountain_car.py:103-111, continuous_mountain_car.py:108-118) and that VectorEnv.get_attr/set_attr address.
PARAM_NAMES = {
    CARTPOLE: ["gravity", "masscart", "masspole", "total_mass", "length", "polemass_length", "force_mag", "tau",
               "theta_threshold_radians", "x_threshold", "kinematics_integrator"],
    PENDULUM: ["max_speed", "max_torque", "dt", "g", "m", "l"],
    ACROBOT: ["dt", "LINK_LENGTH_1", "LINK_LENGTH_2", "LINK_MASS_1", "LINK_MASS_2", "LINK_COM_POS_1", "LINK_COM_POS_2",
              "LINK_MOI", "MAX_VEL_1", "MAX_VEL_2", "torque_noise_max", "book_or_nips"],
    MOUNTAINCAR: ["min_position", "max_position", "max_speed", "goal_position", "goal_velocity", "force", "gravity"],
    MOUNTAINCAR_CONT: ["min_action", "max_action", "min_position", "max_position", "max_speed", "goal_position",
                       "goal_velocity", "power"],
}

# Attributes whose reference value is a string; encoded as 0.0 / 1.0 in the parameter vector.
ENUM_PARAMS = {
    (CARTPOLE, "kinematics_integrator"): ["euler", "semi-implicit"],
    (ACROBOT, "book_or_nips"): ["book", "nips"],
}

# Constructor kwargs the reference envs accept (gym.make(id, **kwargs)) -> parameter name.
CTOR_KWARGS = {
    PENDULUM: {"g": "g"},                            # pendulum.py:91
    MOUNTAINCAR: {"goal_velocity": "goal_velocity"},  # mountain_car.py:103
    MOUNTAINCAR_CONT: {"goal_velocity": "goal_velocity"},
}


def single_spaces(kind: int):
    """(observation_space, action_space) exactly as the reference env's __init__ builds them."""
    if kind == CARTPOLE:  # cartpole.py:106-117
        high = np.array([2.4 * 2, _F32MAX, (12 * 2 * np.pi / 360) * 2, _F32MAX], dtype=np.float32)
        return Box(-high, high, dtype=np.float32), Discrete(2)
    if kind == PENDULUM:  # pendulum.py:107-116
        high = np.array([1.0, 1.0, 8.0], dtype=np.float32)
        return Box(low=-high, high=high, dtype=np.float32), Box(low=-2.0, high=2.0, shape=(1,), dtype=np.float32)
    if kind == ACROBOT:  # acrobot.py:172-178
        high = np.array([1.0, 1.0, 1.0, 1.0, 4 * np.pi, 9 * np.pi], dtype=np.float32)
        return Box(low=-high, high=high, dtype=np.float32), Discrete(3)
    low = np.array([-1.2, -0.07], dtype=np.float32)  # mountain_car.py:113-124, continuous_mountain_car.py:119-137
    high = np.array([0.6, 0.07], dtype=np.float32)
    if kind == MOUNTAINCAR:
        return Box(low, high, dtype=np.float32), Discrete(3)
    if kind == MOUNTAINCAR_CONT:
        return Box(low=low, high=high, dtype=np.float32), Box(low=-1.0, high=1.0, shape=(1,), dtype=np.float32)
    raise ValueError(f"unknown env kind {kind}")
