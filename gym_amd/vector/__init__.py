"""`gym.vector` namespace of the engine, so that code written as `gym.vector.make(...)` runs with `import gym_amd as gym`
(gym/vector/__init__.py:1-73): `make`, `VectorEnv`, `VectorEnvWrapper`, and `utils` (the batching helpers of gym/vector/utils).  `SyncVectorEnv` / `AsyncVectorEnv` take lists of
Python env constructors and have no counterpart here — all sub-envs of one id live in one kernel launch."""
from ..vector_env import HipVectorEnv, VectorEnv, VectorEnvWrapper, make
from . import utils

__all__ = ["make", "VectorEnv", "VectorEnvWrapper", "HipVectorEnv", "utils"]
