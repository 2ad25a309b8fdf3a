"""gym_amd — MI355X-native vectorised classic-control environment engine.

A drop-in for the `gym.vector.SyncVectorEnv` hot path of openai/gym 0.26.2 on CartPole-v0/v1,
Pendulum-v1, Acrobot-v1, MountainCar-v0 and MountainCarContinuous-v0: hand-written gfx950 HIP
kernels behind the C ABI of include/mxv.h, bound with ctypes (gym_amd._native) and wrapped by
`HipVectorEnv` (NumPy contract of the reference) and `DeviceRollout` (device-resident tensors).
There is no CPU fallback: without the built extension / without a HIP device the engine raises.
"""
__version__ = "0.6.0"

_LAZY = {
    "HipVectorEnv": ("gym_amd.vector_env", "HipVectorEnv"),
    "VectorEnv": ("gym_amd.vector_env", "VectorEnv"),
    "VectorEnvWrapper": ("gym_amd.vector_env", "VectorEnvWrapper"),
    "make": ("gym_amd.vector_env", "make"),
    "HipEnv": ("gym_amd.single_env", "HipEnv"),
    "DeviceRollout": ("gym_amd.rollout", "DeviceRollout"),
    "ShardedRollout": ("gym_amd.distributed", "ShardedRollout"),
    "MixedRollout": ("gym_amd.mixed", "MixedRollout"),
    "RecordEpisodeStatistics": ("gym_amd.wrappers", "RecordEpisodeStatistics"),
    "VectorListInfo": ("gym_amd.wrappers", "VectorListInfo"),
    "NormalizeObservation": ("gym_amd.wrappers", "NormalizeObservation"),
    "NormalizeReward": ("gym_amd.wrappers", "NormalizeReward"),
    "RunningNormalizer": ("gym_amd.normalize", "RunningNormalizer"),
    "HipTabularVectorEnv": ("gym_amd.toy_text", "HipTabularVectorEnv"),
    "TabularRollout": ("gym_amd.toy_text", "TabularRollout"),
    "HipBlackjackVectorEnv": ("gym_amd.toy_text", "HipBlackjackVectorEnv"),
    "BlackjackRollout": ("gym_amd.toy_text", "BlackjackRollout"),
}


def __getattr__(name):
    if name in ("vector", "wrappers", "spaces", "error", "plugin", "toy_text"):  # sub-modules, gym-shaped
        import importlib

        return importlib.import_module(f"gym_amd.{name}")
    if name in _LAZY:
        import importlib

        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(mod), attr)
    raise AttributeError(f"module 'gym_amd' has no attribute {name!r}")
