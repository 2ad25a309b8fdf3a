"""ctypes binding of the C ABI in include/mxv.h (gym_amd/_lib/libmxv.so).

This is the only door into the engine: there is no CPU fallback.  If the shared library has
not been built (`python __graft_entry__.py` or gym_amd/csrc/build.sh) importing this module
raises; if no HIP device is visible `Handle(...)` raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MXV_LIB_PATH: tuning hook (tools/kbench.py loads envs-per-lane build variants through it)
LIB_PATH = os.environ.get("MXV_LIB_PATH") or os.path.join(_HERE, "_lib", "libmxv.so")

# env kinds / flags / status codes: keep in sync with include/mxv.h
CARTPOLE, PENDULUM, ACROBOT, MOUNTAINCAR, MOUNTAINCAR_CONT = range(5)
FLAG_ACTION_I32 = 1
FLAG_REWARD_F32 = 2
FLAG_NO_AUTORESET = 4
OK = 0
ERR_INVALID_ARG = -1
ERR_HIP = -2
ERR_INVALID_ACTION = -3
ERR_RESET_NEEDED = -4
ERR_UNSUPPORTED = -5
MAX_PARAMS = 12
ENV_ALIGN = 4
ROLLOUT_EAGER, ROLLOUT_GRAPH, ROLLOUT_FUSED = 0, 1, 2
TAB_FLAG_COMPACT = 1
TAB_FLAG_GENERAL_KERNEL = 2
TAB_KERNEL_NONE, TAB_KERNEL_GENERAL, TAB_KERNEL_TRAJECTORY = 0, 1, 2
COMM_ID_BYTES = 128

EXPORTS = (
    "mxv_env_dims", "mxv_default_params", "mxv_default_reset_bounds", "mxv_version", "mxv_create", "mxv_destroy",
    "mxv_last_error", "mxv_seed", "mxv_seed_actions", "mxv_reset", "mxv_step", "mxv_step_sampled", "mxv_rollout",
    "mxv_rollout_tape", "mxv_sample_actions", "mxv_last_launch", "mxv_reset_host", "mxv_step_host", "mxv_get_state", "mxv_set_state", "mxv_get_counters",
    "mxv_set_counters", "mxv_set_device_clock", "mxv_set_obs_partials", "mxv_set_return_partials", "mxv_obs_partials_layout", "mxv_get_episodes", "mxv_set_episodes", "mxv_get_params", "mxv_set_params", "mxv_set_params_per_env", "mxv_get_params_per_env", "mxv_episode_stats", "mxv_set_episode_outputs", "mxv_episode_stats_host", "mxv_set_running_returns", "mxv_sync", "mxv_get_stream", "mxv_set_stream",
    "mxv_rollout_mixed", "mxv_set_final_snapshot", "mxv_comm_unique_id", "mxv_comm_init", "mxv_comm_destroy", "mxv_allgather_outputs", "mxv_allgather_wait", "mxv_comm_stream",
    "mxv_host_io", "mxv_step_mapped", "mxv_reset_mapped", "mxv_final_packed", "mxv_final_packed_view", "mxv_final_packed_stats_view", "mxv_write_probe", "mxv_write_probe_env", "mxv_host_alloc", "mxv_host_free",
                "mxv_host_block_layout", "mxv_step_host_block", "mxv_wait_stream", "mxv_staging_view", "mxv_adopt_obs",
    "mxv_norm_create", "mxv_norm_destroy", "mxv_norm_last_error", "mxv_norm_set_stream", "mxv_norm_get_state",
    "mxv_norm_set_state", "mxv_norm_observations", "mxv_norm_rewards", "mxv_norm_obs_sums", "mxv_norm_obs_sums_partials", "mxv_norm_reward_sums_partials", "mxv_norm_returns_ptr", "mxv_norm_obs_apply",
    "mxv_norm_reward_sums", "mxv_norm_reward_apply",
    "mxv_subnorm_create", "mxv_subnorm_destroy", "mxv_subnorm_last_error", "mxv_subnorm_set_stream", "mxv_subnorm_get_state",
    "mxv_subnorm_set_state", "mxv_subnorm_observations", "mxv_subnorm_rewards",
    "mxv_tab_create", "mxv_tab_destroy", "mxv_tab_last_error", "mxv_tab_seed", "mxv_tab_seed_actions", "mxv_tab_reset",
    "mxv_tab_step", "mxv_tab_rollout", "mxv_tab_rollout_tape", "mxv_tab_reset_host", "mxv_tab_step_host", "mxv_tab_get_state",
    "mxv_tab_set_state", "mxv_tab_episode_stats", "mxv_tab_set_episode_outputs", "mxv_tab_episode_stats_host",
    "mxv_bj_episode_stats", "mxv_bj_set_episode_outputs", "mxv_bj_episode_stats_host", "mxv_tab_set_running_returns", "mxv_bj_set_running_returns", "mxv_tab_get_counters", "mxv_tab_set_counters", "mxv_tab_sync", "mxv_tab_last_kernel", "mxv_tab_word_threshold", "mxv_tab_set_device_clock", "mxv_bj_set_device_clock", "mxv_tab_set_stream",
    "mxv_get_beyond", "mxv_set_beyond",
    "mxv_bj_create", "mxv_bj_destroy", "mxv_bj_last_error", "mxv_bj_seed", "mxv_bj_reset", "mxv_bj_step", "mxv_bj_rollout", "mxv_bj_rollout_compact",
    "mxv_bj_reset_host", "mxv_bj_step_host", "mxv_bj_get_state", "mxv_bj_set_state", "mxv_bj_get_counters", "mxv_bj_sync", "mxv_bj_set_stream",
    "mxv_placed_alloc", "mxv_placed_free", "mxv_placed_info_get", "mxv_placed_last_error", "mxv_hbm_pair_probe",
)


class MxvLaunchInfo(C.Structure):
    """mxv_launch_info (include/mxv.h): the kernel instantiation of a handle's last step / rollout launch."""
    _fields_ = [(k, C.c_int32) for k in ("kernel", "env_id", "param_mode", "envs_per_lane", "safe", "out_mode", "tape", "steps")] + \
               [("grid", C.c_uint32), ("block", C.c_uint32)]


class MxvConfig(C.Structure):
    _fields_ = [
        ("env_id", C.c_int32),
        ("device", C.c_int32),
        ("num_envs", C.c_int64),
        ("env_offset", C.c_int64),
        ("max_episode_steps", C.c_int32),
        ("flags", C.c_int32),
        ("seed", C.c_uint64),
        ("action_seed", C.c_uint64),
    ]


class MxvTabConfig(C.Structure):
    _fields_ = [
        ("device", C.c_int32),
        ("num_states", C.c_int32),
        ("num_actions", C.c_int32),
        ("max_transitions", C.c_int32),
        ("num_envs", C.c_int64),
        ("env_offset", C.c_int64),
        ("max_episode_steps", C.c_int32),
        ("flags", C.c_int32),
        ("seed", C.c_uint64),
        ("action_seed", C.c_uint64),
    ]


class MxvBjConfig(C.Structure):
    _fields_ = [
        ("device", C.c_int32),
        ("natural", C.c_int32),
        ("sab", C.c_int32),
        ("max_episode_steps", C.c_int32),
        ("num_envs", C.c_int64),
        ("env_offset", C.c_int64),
        ("seed", C.c_uint64),
        ("action_seed", C.c_uint64),
    ]


BJ_MAX_DRAWS = 24
SNAPSHOT_FORMAT = 2   # Handle.snapshot(): 2 = per-env reset ordinals (`episodes`) are part of the RNG position
PLACED_CHUNK_BYTES = 256 << 20
PLACED_MIN_BYTES = 2 << 30      # mxv_placed_alloc (HIP virtual-memory mappings): sets below it get ordinary allocations
SORTED_MIN_BYTES = 1 << 30      # placement.sorted_tensors (ordinary allocations sorted by class): "auto" sorts sets of 1 GiB and more — the 2^17-env
                                # shard of an 8-GPU strong-scaling job runs 0.79-0.82 us per step sorted, 0.87-0.93 unsorted (profiles/r3/r3k_*)
PLACED_PLAIN, PLACED_NO_JUMP = 1, 2


class MxvPlacedInfo(C.Structure):
    _fields_ = [
        ("placed", C.c_int32),
        ("balanced", C.c_int32),
        ("chunks_created", C.c_int32),
        ("chunks_kept", C.c_int32),
        ("classes_seen", C.c_int32),
        ("class_chunks", C.c_int32 * 4),
        ("solo_group", C.c_int32),
        ("solo_class", C.c_int32),
        ("stop_reason", C.c_int32),
        ("same_class_us", C.c_double),
        ("different_class_us", C.c_double),
        ("seconds", C.c_double),
        ("requested_bytes", C.c_size_t),
        ("held_bytes", C.c_size_t),
        ("peak_bytes", C.c_size_t),
        ("jumped_bytes", C.c_size_t),
    ]


class MxvError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"mxv error {code}: {message}")
        self.code = code
        self.message = message


def _share_torch_hip_runtime():
    """One HIP/HSA runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 under torch/lib and
    name them WITHOUT version suffix in their NEEDED entries, so they do not match a system runtime that is already loaded:
    `import gym_amd; env = gym_amd.make(...); import torch` would put two runtimes into the process and
    torch.cuda.is_available() turns False (seen on the MI355X box: NormalizeObservation created after a NumPy-only loop).
    The other order works — libmxv's NEEDED libamdhip64.so.7 matches the soname of torch's copy once that is loaded — so,
    when torch is installed (not imported: only located), its bundled runtime is loaded first and libmxv binds to it.
    MXV_SYSTEM_HIP=1 keeps the system runtime (then do not use torch.cuda in the same process)."""
    if os.environ.get("MXV_SYSTEM_HIP") == "1" or "torch" in sys.modules:
        return
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        libdir = os.path.join(os.path.dirname(spec.origin), "lib")
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            path = os.path.join(libdir, name)
            if not os.path.exists(path):
                return
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        pass   # a torch without a usable bundled runtime: fall through to the system one


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or gym_amd/csrc/build.sh). gym_amd has no CPU fallback."
        )
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_uint32
    sig = {
        "mxv_env_dims": ([i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)], C.c_int),
        "mxv_default_params": ([i32, vp], C.c_int),
        "mxv_default_reset_bounds": ([i32, vp], C.c_int),
        "mxv_version": ([], C.c_char_p),
        "mxv_create": ([C.POINTER(MxvConfig), C.POINTER(vp)], C.c_int),
        "mxv_destroy": ([vp], C.c_int),
        "mxv_last_error": ([vp], C.c_char_p),
        "mxv_seed": ([vp, u64, vp], C.c_int),
        "mxv_seed_actions": ([vp, u64], C.c_int),
        "mxv_reset": ([vp, vp, vp, vp], C.c_int),
        "mxv_step": ([vp, vp, vp, vp, vp, vp, vp], C.c_int),
        "mxv_step_sampled": ([vp, vp, vp, vp, vp, vp, vp], C.c_int),
        "mxv_rollout": ([vp, i32, i32, i32, vp, vp, vp, vp, vp, vp], C.c_int),
        "mxv_rollout_tape": ([vp, i32, i32, vp, vp, vp, vp, vp, vp], C.c_int),
        "mxv_sample_actions": ([vp, vp], C.c_int),
        "mxv_last_launch": ([vp, C.POINTER(MxvLaunchInfo)], C.c_int),
        "mxv_reset_host": ([vp, vp, vp, vp], C.c_int),
        "mxv_step_host": ([vp, vp, vp, vp, vp, vp, vp], C.c_int),
        "mxv_get_state": ([vp, vp, vp], C.c_int),
        "mxv_adopt_obs": ([vp, vp], C.c_int),
        "mxv_set_state": ([vp, vp, vp], C.c_int),
        "mxv_get_counters": ([vp, C.POINTER(u64), C.POINTER(u32)], C.c_int),
        "mxv_set_counters": ([vp, u64, u32], C.c_int),
        "mxv_set_device_clock": ([vp, i32], C.c_int),
        "mxv_set_obs_partials": ([vp, vp], C.c_int),
        "mxv_set_return_partials": ([vp, vp, C.c_double, vp], C.c_int),
        "mxv_obs_partials_layout": ([vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i32)], C.c_int),
        "mxv_rollout_mixed": ([vp, C.c_int32, C.c_int32, C.c_int32, vp], C.c_int),
        "mxv_set_final_snapshot": ([vp, vp, vp, vp, vp], C.c_int),
        "mxv_write_probe": ([C.c_int32, i64, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, C.POINTER(C.c_double)], C.c_int),
        "mxv_write_probe_env": ([C.c_int32, C.c_int32, C.c_int32, i64, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, C.POINTER(C.c_double)], C.c_int),
        "mxv_wait_stream": ([vp, vp], C.c_int),
        "mxv_staging_view": ([vp] + [C.POINTER(vp)] * 4, C.c_int),
        "mxv_host_alloc": ([C.c_size_t, C.POINTER(vp)], C.c_int),
        "mxv_host_free": ([vp], C.c_int),
        "mxv_host_block_layout": ([vp] + [C.POINTER(C.c_size_t)] * 6, C.c_int),
        "mxv_step_host_block": ([vp, vp, vp, C.c_int32], C.c_int),
        "mxv_final_packed": ([vp, C.c_int32, C.POINTER(C.c_int32)], C.c_int),
        "mxv_final_packed_stats_view": ([vp, C.POINTER(vp), C.POINTER(vp)], C.c_int),
        "mxv_final_packed_view": ([vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)], C.c_int),
        "mxv_comm_unique_id": ([vp], C.c_int),
        "mxv_comm_init": ([vp, C.c_int32, C.c_int32, vp], C.c_int),
        "mxv_comm_destroy": ([vp], C.c_int),
        "mxv_allgather_outputs": ([vp] * 9, C.c_int),
        "mxv_allgather_wait": ([vp, C.c_int32, C.c_int32], C.c_int),
        "mxv_comm_stream": ([vp, C.POINTER(vp)], C.c_int),
        "mxv_get_episodes": ([vp, vp], C.c_int),
        "mxv_set_episodes": ([vp, vp], C.c_int),
        "mxv_get_beyond": ([vp, vp], C.c_int),
        "mxv_set_beyond": ([vp, vp], C.c_int),
        "mxv_get_params": ([vp, vp], C.c_int),
        "mxv_set_params": ([vp, vp], C.c_int),
        "mxv_set_params_per_env": ([vp, vp], C.c_int),
        "mxv_get_params_per_env": ([vp, vp], C.c_int),
        "mxv_episode_stats": ([vp, i32], C.c_int),
        "mxv_set_episode_outputs": ([vp, vp, vp], C.c_int),
        "mxv_episode_stats_host": ([vp, vp, vp, vp], C.c_int),
        "mxv_set_running_returns": ([vp, vp], C.c_int),
        "mxv_sync": ([vp], C.c_int),
        "mxv_get_stream": ([vp, C.POINTER(vp)], C.c_int),
        "mxv_set_stream": ([vp, vp], C.c_int),
        "mxv_host_io": ([vp] + [C.POINTER(vp)] * 6, C.c_int),
        "mxv_step_mapped": ([vp], C.c_int),
        "mxv_reset_mapped": ([vp, vp], C.c_int),
        "mxv_norm_create": ([i32, i32, i64, vp, C.POINTER(vp)], C.c_int),
        "mxv_norm_destroy": ([vp], C.c_int),
        "mxv_norm_last_error": ([vp], C.c_char_p),
        "mxv_norm_set_stream": ([vp, vp], C.c_int),
        "mxv_norm_get_state": ([vp, vp, vp, vp, vp], C.c_int),
        "mxv_norm_set_state": ([vp, vp, vp, C.c_double, vp], C.c_int),
        "mxv_norm_observations": ([vp, i32, vp, vp, i32, C.c_double], C.c_int),
        "mxv_norm_rewards": ([vp, i32, vp, i32, vp, vp, vp, C.c_double, C.c_double], C.c_int),
        "mxv_norm_obs_sums": ([vp, i32, vp, vp], C.c_int),
        "mxv_norm_obs_sums_partials": ([vp, i32, vp, i64, vp], C.c_int),
        "mxv_norm_reward_sums_partials": ([vp, i32, vp, i64, vp], C.c_int),
        "mxv_norm_returns_ptr": ([vp, C.POINTER(vp)], C.c_int),
        "mxv_norm_obs_apply": ([vp, i32, vp, vp, i32, C.c_double, vp, i32, i64], C.c_int),
        "mxv_norm_reward_sums": ([vp, i32, vp, i32, vp, vp, C.c_double, vp], C.c_int),
        "mxv_norm_reward_apply": ([vp, i32, vp, i32, vp, C.c_double, vp, i32, i64], C.c_int),
        "mxv_subnorm_create": ([i32, i32, i64, vp, C.POINTER(vp)], C.c_int),
        "mxv_subnorm_destroy": ([vp], C.c_int),
        "mxv_subnorm_last_error": ([vp], C.c_char_p),
        "mxv_subnorm_set_stream": ([vp, vp], C.c_int),
        "mxv_subnorm_get_state": ([vp, vp, vp, vp, vp], C.c_int),
        "mxv_subnorm_set_state": ([vp, vp, vp, vp, vp], C.c_int),
        "mxv_subnorm_observations": ([vp, i32, vp, vp, vp, vp, vp, i32, vp, C.c_double], C.c_int),
        "mxv_subnorm_rewards": ([vp, i32, vp, i32, vp, vp, vp, C.c_double, C.c_double], C.c_int),
        "mxv_tab_create": ([C.POINTER(MxvTabConfig), vp, vp, vp, vp, vp, vp, C.POINTER(vp)], C.c_int),
        "mxv_tab_destroy": ([vp], C.c_int),
        "mxv_tab_last_error": ([vp], C.c_char_p),
        "mxv_tab_seed": ([vp, u64, vp], C.c_int),
        "mxv_tab_seed_actions": ([vp, u64], C.c_int),
        "mxv_tab_reset": ([vp, vp, vp], C.c_int),
        "mxv_tab_step": ([vp] * 10, C.c_int),
        "mxv_tab_rollout": ([vp, i32, i32] + [vp] * 8, C.c_int),
        "mxv_tab_rollout_tape": ([vp, i32, i32] + [vp] * 8, C.c_int),
        "mxv_tab_reset_host": ([vp, vp, vp], C.c_int),
        "mxv_tab_step_host": ([vp] * 10, C.c_int),
        "mxv_tab_get_state": ([vp, vp, vp], C.c_int),
        "mxv_tab_episode_stats": ([vp, i32], C.c_int),
        "mxv_tab_set_episode_outputs": ([vp, vp, vp], C.c_int),
        "mxv_tab_episode_stats_host": ([vp, vp, vp, vp], C.c_int),
        "mxv_bj_episode_stats": ([vp, i32], C.c_int),
        "mxv_bj_set_episode_outputs": ([vp, vp, vp], C.c_int),
        "mxv_bj_episode_stats_host": ([vp, vp, vp, vp], C.c_int),
        "mxv_tab_set_running_returns": ([vp, vp], C.c_int),
        "mxv_bj_set_running_returns": ([vp, vp], C.c_int),
        "mxv_tab_set_state": ([vp, vp, vp], C.c_int),
        "mxv_tab_get_counters": ([vp, C.POINTER(u64), C.POINTER(u32)], C.c_int),
        "mxv_tab_set_counters": ([vp, u64, u32], C.c_int),
        "mxv_tab_sync": ([vp], C.c_int),
        "mxv_tab_last_kernel": ([vp], C.c_int),
        "mxv_tab_set_device_clock": ([vp, i32], C.c_int),
        "mxv_bj_set_device_clock": ([vp, i32], C.c_int),
        "mxv_tab_word_threshold": ([C.c_double], u64),
        "mxv_tab_set_stream": ([vp, vp], C.c_int),
        "mxv_bj_create": ([C.POINTER(MxvBjConfig), C.POINTER(vp)], C.c_int),
        "mxv_bj_destroy": ([vp], C.c_int),
        "mxv_bj_last_error": ([vp], C.c_char_p),
        "mxv_bj_seed": ([vp, u64, vp, u64], C.c_int),
        "mxv_bj_reset": ([vp, vp, vp, vp], C.c_int),
        "mxv_bj_step": ([vp] * 8, C.c_int),
        "mxv_bj_rollout": ([vp, i32, i32] + [vp] * 7, C.c_int),
        "mxv_bj_rollout_compact": ([vp, i32, i32] + [vp] * 7, C.c_int),
        "mxv_bj_reset_host": ([vp, vp, vp], C.c_int),
        "mxv_bj_step_host": ([vp] * 8, C.c_int),
        "mxv_bj_get_state": ([vp, vp, vp], C.c_int),
        "mxv_bj_set_state": ([vp, vp, vp, u64, u32], C.c_int),
        "mxv_bj_get_counters": ([vp, vp, vp], C.c_int),
        "mxv_bj_sync": ([vp], C.c_int),
        "mxv_bj_set_stream": ([vp, vp], C.c_int),
        "mxv_placed_alloc": ([i32, i32, vp, vp, i32, vp, C.POINTER(vp)], C.c_int),
        "mxv_placed_free": ([vp], C.c_int),
        "mxv_hbm_pair_probe": ([i32, vp, vp, i32, C.POINTER(C.c_double)], C.c_int),
        "mxv_placed_info_get": ([vp, C.POINTER(MxvPlacedInfo)], C.c_int),
        "mxv_placed_last_error": ([vp], C.c_char_p),
    }
    for name, (argtypes, restype) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = the .so does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = restype
    return lib


lib = _load()


def env_dims(env_id: int):
    s, o, na = C.c_int32(), C.c_int32(), C.c_int32()
    rc = lib.mxv_env_dims(env_id, C.byref(s), C.byref(o), C.byref(na))
    if rc != OK:
        raise MxvError(rc, f"unknown env_id {env_id}")
    return s.value, o.value, na.value


def default_params(env_id: int) -> np.ndarray:
    p = np.zeros(MAX_PARAMS, dtype=np.float64)
    rc = lib.mxv_default_params(env_id, p.ctypes.data)
    if rc != OK:
        raise MxvError(rc, f"unknown env_id {env_id}")
    return p


def default_reset_bounds(env_id: int) -> np.ndarray:
    b = np.zeros(2, dtype=np.float64)
    rc = lib.mxv_default_reset_bounds(env_id, b.ctypes.data)
    if rc != OK:
        raise MxvError(rc, f"unknown env_id {env_id}")
    return b


def _ptr(x):
    """Device pointer (int / torch tensor) or host ndarray -> c_void_p value."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    return x.data_ptr()  # torch tensor


class _Lease:
    """One hand-out of a pooled buffer.  The array the caller receives is np.asarray(lease): NumPy makes the lease the `.base` of that
    array AND of every view later derived from it (a view's base is "the first object that is not an array", so slices and reshapes of
    the array reference the lease, not the array).  The lease therefore lives exactly as long as anything that can still see the
    memory, and its finalizer — not a reference COUNT read at some moment — is what returns the buffer to the pool: ownership is
    explicit, and correct under tracers, debuggers holding frames, other interpreters' refcount conventions."""
    __slots__ = ("__array_interface__", "_keep", "__weakref__")

    def __init__(self, ptr: int, shape, typestr: str, keep):
        self.__array_interface__ = {"data": (ptr, False), "shape": tuple(shape), "typestr": typestr, "version": 3}
        self._keep = keep            # whatever owns the memory (a NumPy array, a pinned block): alive as long as any lease of it


class _ArrayPool:
    """Recycles the host arrays the NumPy adapter returns.  SyncVectorEnv(copy=True) hands the caller a fresh array per call
    (`deepcopy(self.observations)`, gym/vector/sync_vector_env.py:163); allocating one with np.empty means a fresh anonymous
    mapping per step for anything above glibc's mmap threshold — 42 MB of page faults and kernel page zeroing per step at 2^20
    CartPole envs, which is what made that loop 2-3 ms per step (the DMA itself is 0.9 ms).  The pool keeps a few buffers per
    (shape, dtype) and hands one out again ONLY once the previous hand-out is unreachable — the array it returned and every view
    derived from it have been dropped (see _Lease) — so the contract the caller sees is unchanged: an array it got is never
    overwritten while it can still see it, and the pages stay mapped.  If the caller keeps everything, `limit` buffers per key are out
    and further arrays are plain np.empty."""

    def __init__(self, limit: int = 6):
        self._free = {}      # key -> buffers nobody sees
        self._made = {}      # key -> buffers this pool owns (free or leased): never more than `limit`
        self._limit = limit

    def _give_back(self, key, store):
        self._free[key].append(store)      # (list.append is atomic: finalizers may run on any thread)

    def take(self, shape, dtype) -> np.ndarray:
        shape = tuple(np.atleast_1d(shape).tolist()) if not isinstance(shape, tuple) else shape
        dt = np.dtype(dtype)
        key = (shape, dt.str)
        free = self._free.setdefault(key, [])
        try:
            store = free.pop()
        except IndexError:
            if self._made.get(key, 0) >= self._limit:
                return np.empty(shape, dtype=dt)      # the caller holds `limit` arrays of this kind: no pooling beyond that
            store = np.empty(shape, dtype=dt)
            self._made[key] = self._made.get(key, 0) + 1
        lease = _Lease(store.ctypes.data, shape, dt.str, store)
        weakref.finalize(lease, self._give_back, key, store)
        return np.asarray(lease)


class _PinnedBlock:
    """One mxv_host_alloc block; freed when the pool and every lease of it are gone."""

    def __init__(self, nbytes: int):
        p = C.c_void_p()
        rc = lib.mxv_host_alloc(nbytes, C.byref(p))
        if rc != OK:
            raise MxvError(rc, (lib.mxv_last_error(None) or b"").decode())
        self.ptr, self.nbytes = p.value, nbytes

    def __del__(self):
        try:
            lib.mxv_host_free(self.ptr)
        except Exception:
            pass


class _BlockPool:
    """Pinned output blocks of the NumPy adapter (mxv_step_host_block): the caller gets VIEWS of one block per step; a block is
    handed out again only once every array derived from the previous hand-out is gone (see _Lease) — the same "never overwritten
    while the caller can see it" contract as _ArrayPool.  A caller that keeps everything has `limit` blocks out; take() then returns
    None and the step falls back to separate, unpinned arrays."""

    def __init__(self, nbytes: int, limit: int = 6):
        self._nbytes, self._limit, self._free, self._made = nbytes, limit, [], 0

    def _give_back(self, blk):
        self._free.append(blk)

    def take(self):
        try:
            blk = self._free.pop()
        except IndexError:
            if self._made >= self._limit:
                return None
            blk = _PinnedBlock(self._nbytes)
            self._made += 1
        lease = _Lease(blk.ptr, (self._nbytes,), "|u1", blk)
        weakref.finalize(lease, self._give_back, blk)
        return np.asarray(lease)


def pinned_pool(nbytes: int, limit: int = 6) -> "_BlockPool":
    """A pool of pinned host blocks of `nbytes` (see _BlockPool): take() returns a uint8 array (or None when the caller holds
    every block), recycled once no view of it is alive."""
    return _BlockPool(nbytes, limit)


class _DestroyLater:
    """Owns the mxv_destroy of a handle whose pinned I/O block is referenced by NumPy views."""

    def __init__(self, h):
        self._h = h

    def __del__(self):
        try:
            lib.mxv_destroy(self._h)
        except Exception:
            pass


class Handle:
    """One engine handle = one device + one stream + N device-resident envs (see include/mxv.h)."""

    def __init__(self, env_id: int, num_envs: int, max_episode_steps: int, *, device: int = 0, env_offset: int = 0,
                 seed: int = 0, action_seed: int = 0, flags: int = 0):
        self.env_id = int(env_id)
        self.num_envs = int(num_envs)
        self.S, self.O, self.NA = env_dims(self.env_id)
        self.flags = int(flags)
        self.device = int(device)
        self.env_offset = int(env_offset)
        cfg = MxvConfig(self.env_id, self.device, self.num_envs, self.env_offset, int(max_episode_steps), self.flags,
                        int(seed) & (2**64 - 1), int(action_seed) & (2**64 - 1))
        h = C.c_void_p()
        rc = lib.mxv_create(C.byref(cfg), C.byref(h))
        if rc != OK:
            raise MxvError(rc, (lib.mxv_last_error(None) or b"").decode())
        self._h = h
        self.max_episode_steps = int(max_episode_steps)
        self._base_seed, self._per_env_seeds = int(seed) & (2**64 - 1), None
        self._action_seed = int(action_seed) & (2**64 - 1)
        self._stats_on = False
        self._per_env_params = False

    # -- plumbing -------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != OK:
            raise MxvError(rc, (lib.mxv_last_error(self._h) or b"").decode())

    @property
    def action_dtype(self):
        if self.NA > 0:
            return np.int32 if self.flags & FLAG_ACTION_I32 else np.int64
        return np.float32

    @property
    def reward_dtype(self):
        return np.float32 if self.flags & FLAG_REWARD_F32 else np.float64

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            keep = getattr(self, "_io_keep", None)
            if keep is not None:
                self._io_keep = None   # mapped-I/O views may outlive close(): the last one to die destroys the handle
            else:
                lib.mxv_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- ABI calls ------------------------------------------------------------------------
    def seed(self, base_seed: int, per_env_seeds=None):
        p = None
        if per_env_seeds is not None:
            per_env_seeds = np.ascontiguousarray(per_env_seeds, dtype=np.uint64)
            assert per_env_seeds.shape == (self.num_envs,)
            p = per_env_seeds.ctypes.data
        self._check(lib.mxv_seed(self._h, int(base_seed) & (2**64 - 1), p))
        self._base_seed = int(base_seed) & (2**64 - 1)
        self._per_env_seeds = None if per_env_seeds is None else per_env_seeds.copy()

    def seed_actions(self, action_seed: int):
        self._check(lib.mxv_seed_actions(self._h, int(action_seed) & (2**64 - 1)))
        self._action_seed = int(action_seed) & (2**64 - 1)

    @staticmethod
    def _bounds(bounds):
        if bounds is None:
            return None, None
        b = np.ascontiguousarray(bounds, dtype=np.float64)
        return b, b.ctypes.data

    def reset(self, obs_dev=None, mask_dev=None, bounds=None):
        b, bp = self._bounds(bounds)
        self._check(lib.mxv_reset(self._h, _ptr(mask_dev), bp, _ptr(obs_dev)))

    def step(self, actions_dev, obs_dev, reward_dev=None, terminated_dev=None, truncated_dev=None, final_obs_dev=None):
        self._check(lib.mxv_step(self._h, _ptr(actions_dev), _ptr(obs_dev), _ptr(reward_dev), _ptr(terminated_dev),
                                 _ptr(truncated_dev), _ptr(final_obs_dev)))

    def step_sampled(self, obs_dev, reward_dev=None, terminated_dev=None, truncated_dev=None, final_obs_dev=None,
                     actions_out_dev=None):
        self._check(lib.mxv_step_sampled(self._h, _ptr(actions_out_dev), _ptr(obs_dev), _ptr(reward_dev),
                                         _ptr(terminated_dev), _ptr(truncated_dev), _ptr(final_obs_dev)))

    def rollout(self, K, obs_dev, reward_dev=None, terminated_dev=None, truncated_dev=None, final_obs_dev=None,
                actions_out_dev=None, per_step=False, mode=ROLLOUT_FUSED):
        self._check(lib.mxv_rollout(self._h, int(K), int(per_step), int(mode), _ptr(actions_out_dev),
                                    _ptr(obs_dev), _ptr(reward_dev), _ptr(terminated_dev), _ptr(truncated_dev),
                                    _ptr(final_obs_dev)))

    def rollout_tape(self, K, actions_tape_dev, obs_dev, reward_dev=None, terminated_dev=None, truncated_dev=None,
                     final_obs_dev=None, per_step=False):
        self._check(lib.mxv_rollout_tape(self._h, int(K), int(per_step), _ptr(actions_tape_dev), _ptr(obs_dev),
                                         _ptr(reward_dev), _ptr(terminated_dev), _ptr(truncated_dev),
                                         _ptr(final_obs_dev)))

    def sample_actions(self, actions_out_dev):
        self._check(lib.mxv_sample_actions(self._h, _ptr(actions_out_dev)))

    def reset_host(self, mask=None, bounds=None) -> np.ndarray:
        obs = np.empty((self.num_envs, self.O), dtype=np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        b, bp = self._bounds(bounds)
        self._check(lib.mxv_reset_host(self._h, _ptr(m), bp, obs.ctypes.data))
        return obs

    def step_host(self, actions, want_final=True, pooled=False):
        """One vector step through host arrays.  pooled=True (the NumPy adapter): outputs come from the handle's array pool
        (see _ArrayPool) and `final_obs` rows of envs that did not finish hold stale data instead of zeros."""
        n, O = self.num_envs, self.O
        a = np.ascontiguousarray(actions, dtype=self.action_dtype).reshape(n)
        if pooled:
            pool = self.__dict__.setdefault("_pool", _ArrayPool())
            obs, rew = pool.take((n, O), np.float32), pool.take((n,), self.reward_dtype)
            term, trunc = pool.take((n,), np.uint8), pool.take((n,), np.uint8)
            fin = pool.take((n, O), np.float32) if want_final and not getattr(self, "_packed_final", False) else None
        else:
            obs = np.empty((n, O), dtype=np.float32)
            rew = np.empty(n, dtype=self.reward_dtype)
            term = np.empty(n, dtype=np.uint8)
            trunc = np.empty(n, dtype=np.uint8)
            fin = np.zeros((n, O), dtype=np.float32) if want_final else None
        self._check(lib.mxv_step_host(self._h, a.ctypes.data, obs.ctypes.data, rew.ctypes.data, term.ctypes.data,
                                      trunc.ctypes.data, _ptr(fin)))
        return obs, rew, term.view(np.bool_), trunc.view(np.bool_), fin

    def step_host_block(self, actions, want_final=True):
        """One vector step whose outputs are views of ONE pooled, pinned host block filled by a single DMA
        (mxv_step_host_block); -> obs, reward, terminated, truncated, final_obs (None when the rows are packed or not wanted).
        Falls back to step_host(pooled=True) when the caller holds on to every block of the pool."""
        n, O = self.num_envs, self.O
        lay = getattr(self, "_block_layout", None)
        if lay is None:
            v = [C.c_size_t() for _ in range(6)]
            self._check(lib.mxv_host_block_layout(self._h, *[C.byref(x) for x in v]))
            lay = self._block_layout = tuple(x.value for x in v)
            self._block_pool = _BlockPool(lay[0])
        raw = self._block_pool.take()
        if raw is None:
            return self.step_host(actions, want_final=want_final, pooled=True)
        a = np.ascontiguousarray(actions, dtype=self.action_dtype).reshape(n)
        self._check(lib.mxv_step_host_block(self._h, a.ctypes.data, raw.ctypes.data, int(bool(want_final))))
        _, o_fin, o_obs, o_rew, o_te, o_tr = lay
        rb = np.dtype(self.reward_dtype).itemsize
        obs = raw[o_obs:o_obs + 4 * n * O].view(np.float32).reshape(n, O)
        rew = raw[o_rew:o_rew + rb * n].view(self.reward_dtype)
        term = raw[o_te:o_te + n].view(np.bool_)
        trunc = raw[o_tr:o_tr + n].view(np.bool_)
        fin = None
        if want_final and not getattr(self, "_packed_final", False):
            fin = raw[o_fin:o_fin + 4 * n * O].view(np.float32).reshape(n, O)
        return obs, rew, term, trunc, fin

    def host_io(self):
        """NumPy views of the pinned, device-mapped I/O block (zero-copy stepping): dict(actions, obs, reward, terminated,
        truncated, final_obs).  Valid until close(); overwritten by every step_mapped() / reset_mapped()."""
        ptrs = [C.c_void_p() for _ in range(6)]
        self._check(lib.mxv_host_io(self._h, *[C.byref(p) for p in ptrs]))
        n, O = self.num_envs, self.O
        if getattr(self, "_io_keep", None) is None:
            self._io_keep = _DestroyLater(self._h)
        keep = self._io_keep

        def view(p, dtype, shape):
            count = int(np.prod(shape))
            buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(p.value)
            buf._keepalive = keep   # every array derived from this buffer keeps the pinned block (the handle) alive
            return np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)

        return dict(actions=view(ptrs[0], self.action_dtype, (n,)), obs=view(ptrs[1], np.float32, (n, O)),
                    reward=view(ptrs[2], self.reward_dtype, (n,)), terminated=view(ptrs[3], np.bool_, (n,)),
                    truncated=view(ptrs[4], np.bool_, (n,)), final_obs=view(ptrs[5], np.float32, (n, O)))

    def step_mapped(self):
        self._check(lib.mxv_step_mapped(self._h))

    def final_packed(self, enable: bool = True) -> bool:
        """Ask the host step calls to leave info["final_observation"] rows packed (see include/mxv.h); False for small envs."""
        ok = C.c_int32()
        self._check(lib.mxv_final_packed(self._h, int(bool(enable)), C.byref(ok)))
        self._packed_final = bool(ok.value) and bool(enable)
        if self._packed_final and getattr(self, "_packed_views", None) is None:
            pc, pi, pr = C.c_void_p(), C.c_void_p(), C.c_void_p()
            self._check(lib.mxv_final_packed_view(self._h, C.byref(pc), C.byref(pi), C.byref(pr)))
            n, O = self.num_envs, self.O
            self._packed_views = (np.frombuffer((C.c_char * 4).from_address(pc.value), dtype=np.int32, count=1),
                                  np.frombuffer((C.c_char * (4 * n)).from_address(pi.value), dtype=np.int32, count=n),
                                  np.frombuffer((C.c_char * (4 * n * O)).from_address(pr.value), dtype=np.float32, count=n * O).reshape(n, O))
        return self._packed_final

    def final_packed_rows(self):
        """(indices, rows) of the envs that finished the LAST host step — copies (the library's buffer is overwritten by the next)."""
        cnt, idx, rows = self._packed_views
        c = int(cnt[0])
        return idx[:c].copy(), rows[:c].copy()

    def staging_view(self):
        """mxv_staging_view: (obs, reward, terminated, truncated) addresses of the last host step's outputs as the GPU sees them
        (ints, valid until the next host call)."""
        p = [C.c_void_p() for _ in range(4)]
        self._check(lib.mxv_staging_view(self._h, *[C.byref(x) for x in p]))
        return tuple(x.value for x in p)

    def staging_final(self) -> int:
        """Device address of the dense terminal-observation rows [N][O] float32 of the last host step (valid where that step's
        terminated | truncated; the host steps always leave them there, packed transfer or not): the block's layout applied to
        staging_view()'s observation address."""
        lay = getattr(self, "_block_layout", None)
        if lay is None:
            v = [C.c_size_t() for _ in range(6)]
            self._check(lib.mxv_host_block_layout(self._h, *[C.byref(x) for x in v]))
            lay = self._block_layout = tuple(x.value for x in v)
            self._block_pool = _BlockPool(lay[0])
        return self.staging_view()[0] - lay[2] + lay[1]

    def final_packed_stats(self):
        """(indices, episode returns float32, episode lengths int32) of the envs that finished the LAST host step — copies.
        Needs episode statistics on and packed final rows (large envs)."""
        if getattr(self, "_packed_stats_views", None) is None:
            pr, pl = C.c_void_p(), C.c_void_p()
            self._check(lib.mxv_final_packed_stats_view(self._h, C.byref(pr), C.byref(pl)))
            n = self.num_envs
            self._packed_stats_views = (np.frombuffer((C.c_char * (4 * n)).from_address(pr.value), dtype=np.float32, count=n),
                                        np.frombuffer((C.c_char * (4 * n)).from_address(pl.value), dtype=np.int32, count=n))
        cnt, idx, _ = self._packed_views
        c = int(cnt[0])
        r, l = self._packed_stats_views
        return idx[:c].copy(), r[:c].copy(), l[:c].copy()

    def reset_mapped(self, bounds=None):
        b, bp = self._bounds(bounds)
        self._check(lib.mxv_reset_mapped(self._h, bp))

    def get_state(self):
        st = np.empty((self.S, self.num_envs), dtype=np.float64)
        el = np.empty(self.num_envs, dtype=np.int32)
        self._check(lib.mxv_get_state(self._h, st.ctypes.data, el.ctypes.data))
        return st, el

    def set_state(self, state=None, elapsed=None):
        st = None if state is None else np.ascontiguousarray(state, dtype=np.float64)
        el = None if elapsed is None else np.ascontiguousarray(elapsed, dtype=np.int32)
        if st is not None:
            assert st.shape == (self.S, self.num_envs), st.shape
        if el is not None:
            assert el.shape == (self.num_envs,)
        self._check(lib.mxv_set_state(self._h, _ptr(st), _ptr(el)))

    def last_launch(self) -> dict:
        """Which kernel instantiation the last step / rollout launch took (mxv_last_launch)."""
        info = MxvLaunchInfo()
        self._check(lib.mxv_last_launch(self._h, C.byref(info)))
        return {k: int(getattr(info, k)) for k, _ in MxvLaunchInfo._fields_}

    def get_counters(self):
        t, r = C.c_uint64(), C.c_uint32()
        self._check(lib.mxv_get_counters(self._h, C.byref(t), C.byref(r)))
        return t.value, r.value

    def set_counters(self, t: int, r: int):
        self._check(lib.mxv_set_counters(self._h, int(t), int(r)))

    def set_obs_partials(self, partials_dev=None):
        """Attach (None: detach) the [K][leaves][2 O] float64 buffer in which sampled trajectory launches leave the column sums and
        sums of squares of the observations they write (mxv_set_obs_partials)."""
        self._check(lib.mxv_set_obs_partials(self._h, _ptr(partials_dev)))

    def set_return_partials(self, returns_state_ptr=None, gamma: float = 0.99, partials_dev=None):
        """NormalizeReward's running returns advanced by the rollout, their per-tile sums left in partials_dev [K][leaves][2]
        (mxv_set_return_partials); partials_dev None detaches."""
        self._check(lib.mxv_set_return_partials(self._h, _ptr(returns_state_ptr), float(gamma), _ptr(partials_dev)))

    def obs_partials_layout(self):
        """(leaves, envs per leaf, values per leaf = 2 O) of that buffer."""
        lv, per, vals = C.c_int64(), C.c_int64(), C.c_int32()
        self._check(lib.mxv_obs_partials_layout(self._h, C.byref(lv), C.byref(per), C.byref(vals)))
        return lv.value, per.value, vals.value

    def set_device_clock(self, on: bool = True):
        """The step index lives in device memory and advances on the stream (mxv_set_device_clock): calls of this handle can then be
        recorded into a caller's hipGraph and replayed."""
        self._check(lib.mxv_set_device_clock(self._h, 1 if on else 0))

    def get_episodes(self) -> np.ndarray:
        """Per-env reset ordinals (uint32 [N]): how many resets each env has had since seeding = the position of its reset
        stream (RNG contract, include/mxv.h)."""
        ep = np.empty(self.num_envs, dtype=np.uint32)
        self._check(lib.mxv_get_episodes(self._h, ep.ctypes.data))
        return ep

    def set_episodes(self, episodes):
        ep = np.ascontiguousarray(episodes, dtype=np.uint32)
        assert ep.shape == (self.num_envs,)
        self._check(lib.mxv_set_episodes(self._h, ep.ctypes.data))

    def get_params(self) -> np.ndarray:
        p = np.zeros(MAX_PARAMS, dtype=np.float64)
        self._check(lib.mxv_get_params(self._h, p.ctypes.data))
        return p

    def set_params(self, params):
        p = np.ascontiguousarray(params, dtype=np.float64)
        assert p.shape == (MAX_PARAMS,)
        self._check(lib.mxv_set_params(self._h, p.ctypes.data))
        self._per_env_params = False

    def get_params_per_env(self) -> np.ndarray:
        """[MAX_PARAMS, N] attribute-major table (broadcast values are expanded)."""
        p = np.zeros((MAX_PARAMS, self.num_envs), dtype=np.float64)
        self._check(lib.mxv_get_params_per_env(self._h, p.ctypes.data))
        return p

    def set_params_per_env(self, table):
        p = np.ascontiguousarray(table, dtype=np.float64)
        assert p.shape == (MAX_PARAMS, self.num_envs), p.shape
        self._check(lib.mxv_set_params_per_env(self._h, p.ctypes.data))
        self._per_env_params = True

    def adopt_obs(self, obs_dev=None):
        """mxv_adopt_obs: `obs_dev` (float32 [N][O] device tensor / address; None releases) doubles as the float32 half of the state between
        single steps that pass it as their obs — it must not be written in between and must outlive the adoption."""
        self._check(lib.mxv_adopt_obs(self._h, _ptr(obs_dev)))

    def episode_stats(self, enable: bool = True):
        self._check(lib.mxv_episode_stats(self._h, 1 if enable else 0))
        self._stats_on = bool(enable)

    def set_running_returns(self, running):
        r = np.ascontiguousarray(running, dtype=np.float32)
        assert r.shape == (self.num_envs,)
        self._check(lib.mxv_set_running_returns(self._h, r.ctypes.data))

    # -- checkpoint -----------------------------------------------------------------------
    def snapshot(self) -> dict:
        """Everything a fresh handle needs to continue this one bit-identically (plain NumPy / ints: picklable): env state,
        TimeLimit counters, RNG seeds and counters, physics parameters, running episode returns."""
        state, elapsed = self.get_state()
        t, r = self.get_counters()
        snap = dict(format=SNAPSHOT_FORMAT, env_id=self.env_id, num_envs=self.num_envs, max_episode_steps=self.max_episode_steps,
                    env_offset=self.env_offset, flags=self.flags, base_seed=self._base_seed,
                    per_env_seeds=None if self._per_env_seeds is None else self._per_env_seeds.copy(),
                    action_seed=self._action_seed, state=state, elapsed=elapsed, t=t, r=r, episodes=self.get_episodes(),
                    params=self.get_params_per_env() if self._per_env_params else self.get_params(),
                    per_env_params=self._per_env_params, stats_on=self._stats_on, running_returns=None, beyond=self.get_beyond())
        if self._stats_on:
            snap["running_returns"] = self.episode_stats_host(want_running=True)[2]
        return snap

    def restore(self, snap: dict):
        # format 2 (round 2): reset draws are indexed by per-env reset ordinals (`episodes`).  Older snapshots carry no ordinals and
        # were written under the step-indexed reset stream: they cannot continue bit-identically, and restoring them silently would
        # replay reset states — refuse with a message instead of a KeyError
        # (snapshots written between the introduction of the ordinals and of the `format` key carry `episodes` and no key: they ARE
        # format 2 and restore as such)
        fmt = snap.get("format", SNAPSHOT_FORMAT if "episodes" in snap else 1)
        if fmt != SNAPSHOT_FORMAT or "episodes" not in snap:
            raise ValueError(f"snapshot format {fmt} (this build reads {SNAPSHOT_FORMAT}): written before the reset "
                             "stream was indexed by per-env reset ordinals; it cannot be continued bit-identically — re-create the env "
                             "and set_state() from snap['state'] / snap['elapsed'] if an approximate resume is enough")
        for k in ("env_id", "num_envs", "env_offset", "flags", "max_episode_steps"):
            if snap[k] != getattr(self, k):
                raise ValueError(f"snapshot {k}={snap[k]!r} does not fit this handle ({getattr(self, k)!r})")
        self.seed(snap["base_seed"], snap["per_env_seeds"])   # resets the counters; set below
        self.seed_actions(snap["action_seed"])
        if snap["per_env_params"]:
            self.set_params_per_env(snap["params"])
        else:
            self.set_params(snap["params"])
        self.episode_stats(bool(snap["stats_on"]))
        if snap["stats_on"]:
            self.set_running_returns(snap["running_returns"])
        self.set_state(snap["state"], snap["elapsed"])
        self.set_counters(snap["t"], snap["r"])
        self.set_episodes(snap["episodes"])
        if snap.get("beyond") is not None:             # after set_state (which clears the marks: an injected state is a fresh one)
            self.set_beyond(snap["beyond"])

    def get_beyond(self) -> np.ndarray:
        """CartPole's steps_beyond_terminated marks (uint8 [N]; all zero for handles that keep none: mxv_get_beyond)."""
        b = np.zeros(self.num_envs, dtype=np.uint8)
        self._check(lib.mxv_get_beyond(self._h, b.ctypes.data))
        return b

    def set_beyond(self, beyond):
        b = np.ascontiguousarray(beyond, dtype=np.uint8).reshape(self.num_envs)
        self._check(lib.mxv_set_beyond(self._h, b.ctypes.data))

    def set_episode_outputs(self, ep_return_dev=None, ep_length_dev=None):
        self._check(lib.mxv_set_episode_outputs(self._h, _ptr(ep_return_dev), _ptr(ep_length_dev)))

    def episode_stats_host(self, want_running: bool = False):
        """(returns, lengths[, running_returns]) of the last step_host call; valid where terminated | truncated."""
        r = np.zeros(self.num_envs, dtype=np.float32)
        l = np.zeros(self.num_envs, dtype=np.int32)
        run = np.zeros(self.num_envs, dtype=np.float32) if want_running else None
        self._check(lib.mxv_episode_stats_host(self._h, r.ctypes.data, l.ctypes.data, _ptr(run)))
        return (r, l, run) if want_running else (r, l)

    def sync(self):
        self._check(lib.mxv_sync(self._h))

    def wait_stream(self, stream_ptr: int):
        """mxv_wait_stream: the handle's stream waits (on the GPU) for everything queued on hipStream_t `stream_ptr` so far."""
        self._check(lib.mxv_wait_stream(self._h, C.c_void_p(stream_ptr)))

    @property
    def stream(self) -> int:
        s = C.c_void_p()
        self._check(lib.mxv_get_stream(self._h, C.byref(s)))
        return s.value or 0

    def set_stream(self, stream_ptr: int):
        self._check(lib.mxv_set_stream(self._h, C.c_void_p(stream_ptr)))

    def set_final_snapshot(self, obs=None, reward=None, terminated=None, truncated=None):
        """Every following K-step rollout also writes its last step's outputs into these device buffers (None: detach)."""
        self._check(lib.mxv_set_final_snapshot(self._h, _ptr(obs), _ptr(reward), _ptr(terminated), _ptr(truncated)))

    # -- collectives of a sharded vector env (RCCL behind the C ABI) ------------------------------------------------
    def comm_init(self, rank: int, world: int, unique_id: bytes):
        assert len(unique_id) == COMM_ID_BYTES
        buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        self._check(lib.mxv_comm_init(self._h, int(rank), int(world), C.cast(buf, C.c_void_p)))
        self.comm_world, self.comm_rank = int(world), int(rank)

    def comm_destroy(self):
        self._check(lib.mxv_comm_destroy(self._h))

    def allgather_outputs(self, obs=None, reward=None, terminated=None, truncated=None, all_obs=None, all_reward=None,
                          all_terminated=None, all_truncated=None):
        """Asynchronous grouped all-gather of this shard's outputs into [world][...] device buffers (see include/mxv.h)."""
        self._check(lib.mxv_allgather_outputs(self._h, _ptr(obs), _ptr(reward), _ptr(terminated), _ptr(truncated),
                                              _ptr(all_obs), _ptr(all_reward), _ptr(all_terminated), _ptr(all_truncated)))

    def allgather_wait(self, host_sync: bool = False, age: int = 0):
        """Wait for the last gather (age 0) or the one before (age 1); on the handle's stream, or on the host with host_sync."""
        self._check(lib.mxv_allgather_wait(self._h, int(age), int(bool(host_sync))))

    @property
    def comm_stream(self) -> int:
        s = C.c_void_p()
        self._check(lib.mxv_comm_stream(self._h, C.byref(s)))
        return s.value or 0


class StepOutputs(C.Structure):
    """mxv_step_outputs (include/mxv.h): the output pointers of one segment of a mixed-batch launch."""
    _fields_ = [("actions_out", C.c_void_p), ("obs", C.c_void_p), ("reward", C.c_void_p), ("terminated", C.c_void_p),
                ("truncated", C.c_void_p), ("final_obs", C.c_void_p)]


MAX_MIXED = 8


def rollout_mixed(handles, K: int, outs, per_step: bool = True):
    """mxv_rollout_mixed: K sampled steps of several homogeneous handles (one device) in ONE kernel launch.  outs[i] = dict
    with obs (required) / reward / terminated / truncated / actions / final_obs device tensors of handles[i]."""
    n = len(handles)
    assert 1 <= n <= MAX_MIXED and len(outs) == n
    hs = (C.c_void_p * n)(*[h._h for h in handles])
    arr = (StepOutputs * n)()
    for i, o in enumerate(outs):
        arr[i].actions_out = _ptr(o.get("actions"))
        arr[i].obs = _ptr(o["obs"])
        arr[i].reward = _ptr(o.get("reward"))
        arr[i].terminated = _ptr(o.get("terminated"))
        arr[i].truncated = _ptr(o.get("truncated"))
        arr[i].final_obs = _ptr(o.get("final_obs"))
    rc = lib.mxv_rollout_mixed(C.cast(hs, C.c_void_p), n, int(K), int(bool(per_step)), C.cast(arr, C.c_void_p))
    if rc != OK:
        raise MxvError(rc, (lib.mxv_last_error(handles[0]._h) or b"").decode())


def write_probe(device, num_envs, K, launches, obs, reward, actions, terminated, truncated) -> float:
    """mxv_write_probe: us per vector step of the rollout's store pattern alone into the given [K][N] device tensors."""
    us = C.c_double()
    rc = lib.mxv_write_probe(int(device), int(num_envs), int(K), int(launches), _ptr(obs), _ptr(reward), _ptr(actions),
                             _ptr(terminated), _ptr(truncated), C.byref(us))
    if rc != OK:
        raise MxvError(rc, (lib.mxv_last_error(None) or b"").decode())
    return us.value


def hbm_pair_probe(device: int, wide_ptr: int, narrow_ptr: int, launches: int = 4) -> float:
    """mxv_hbm_pair_probe: us per step of a 16-B/lane store stream over 256 MiB at wide_ptr running next to an 8-B/lane stream over
    128 MiB at narrow_ptr — ~10 % faster when the two lie in different HBM classes (include/mxv.h).  Contents are destroyed."""
    us = C.c_double()
    rc = lib.mxv_hbm_pair_probe(int(device), C.c_void_p(int(wide_ptr)), C.c_void_p(int(narrow_ptr)), int(launches), C.byref(us))
    if rc != OK:
        raise MxvError(rc, (lib.mxv_placed_last_error(None) or b"").decode())
    return us.value


class _PlacedArray:
    """One tensor of a PlacedMemory as a __cuda_array_interface__ object (what torch.as_tensor turns into a zero-copy tensor;
    the tensor keeps this object — and through it the whole PlacedMemory — alive)."""

    def __init__(self, owner, ptr, shape, typestr):
        self._owner = owner
        self.__cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class PlacedMemory:
    """mxv_placed_alloc: device memory for a set of tensors whose long store streams must not share a physical memory class
    (include/mxv.h "placed device memory").  specs: [(name, shape, numpy dtype, group)], group 0 / 1 = the two sides that are
    kept apart (observations | rewards + actions), -1 = does not matter.  `pointers[name]` are device addresses;
    `arrays()` gives __cuda_array_interface__ objects, `tensors(device)` torch tensors sharing the memory.  The memory lives
    until close() or until the object and every tensor made from it are gone."""

    def __init__(self, device: int, specs, *, flags: int = 0):
        self.device = int(device)
        self.specs = [(name, tuple(int(x) for x in shape), np.dtype(dt), int(group)) for name, shape, dt, group in specs]
        n = len(self.specs)
        nbytes = (C.c_size_t * n)(*[max(1, int(np.prod(shape)) * dt.itemsize) for _, shape, dt, _ in self.specs])
        groups = (C.c_int32 * n)(*[g for *_, g in self.specs])
        ptrs = (C.c_void_p * n)()
        h = C.c_void_p()
        rc = lib.mxv_placed_alloc(self.device, n, nbytes, groups, int(flags), ptrs, C.byref(h))
        if rc != OK:
            raise MxvError(rc, (lib.mxv_placed_last_error(None) or b"").decode())
        self._h = h
        self.pointers = {name: int(ptrs[i]) for i, (name, *_rest) in enumerate(self.specs)}
        info = MxvPlacedInfo()
        lib.mxv_placed_info_get(self._h, C.byref(info))
        self.info = {"placed": bool(info.placed), "balanced": bool(info.balanced), "chunks_created": info.chunks_created,
                     "chunks_kept": info.chunks_kept, "classes_seen": info.classes_seen, "class_chunks": list(info.class_chunks),
                     "solo_group": info.solo_group, "solo_class": info.solo_class,
                     "stop_reason": ["balanced", "chunk cap", "jump budget", "spacer allocation failed", "chunk allocation failed"][info.stop_reason],
                     "note": (lib.mxv_placed_last_error(self._h) or b"").decode(), "same_class_us": round(info.same_class_us, 3),
                     "different_class_us": round(info.different_class_us, 3), "seconds": round(info.seconds, 3),
                     "requested_GiB": round(info.requested_bytes / 2**30, 3), "held_GiB": round(info.held_bytes / 2**30, 3),
                     "peak_GiB": round(info.peak_bytes / 2**30, 3), "jumped_GiB": round(info.jumped_bytes / 2**30, 3)}

    def arrays(self) -> dict:
        return {name: _PlacedArray(self, self.pointers[name], shape, dt.str) for name, shape, dt, _ in self.specs}

    def tensors(self) -> dict:
        import torch

        dev = torch.device("cuda", self.device)
        return {name: torch.as_tensor(a, device=dev) for name, a in self.arrays().items()}

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.mxv_placed_free(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_probe_env(device, env_kind, flags, num_envs, K, launches, obs, reward, actions, terminated, truncated) -> float:
    """mxv_write_probe_env: the store pattern of env kind `env_kind`'s fused rollout (its observation width, envs per lane, dtypes)."""
    us = C.c_double()
    rc = lib.mxv_write_probe_env(int(device), int(env_kind), int(flags), int(num_envs), int(K), int(launches), _ptr(obs), _ptr(reward),
                                 _ptr(actions), _ptr(terminated), _ptr(truncated), C.byref(us))
    if rc != OK:
        raise MxvError(rc, (lib.mxv_last_error(None) or b"").decode())
    return us.value


def comm_unique_id() -> bytes:
    """RCCL unique id (created on one rank, shipped to the others, passed to Handle.comm_init on every rank)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib.mxv_comm_unique_id(C.cast(buf, C.c_void_p))
    if rc != OK:
        raise MxvError(rc, (lib.mxv_last_error(None) or b"").decode())
    return buf.raw


class Norm:
    """One mxv_norm = one device-resident RunningMeanStd (+ NormalizeReward's return accumulators); see include/mxv.h."""

    def __init__(self, dim: int, num_envs: int, *, device: int = 0, stream: int = 0):
        self.dim, self.num_envs, self.device = int(dim), int(num_envs), int(device)
        h = C.c_void_p()
        rc = lib.mxv_norm_create(self.device, self.dim, self.num_envs, C.c_void_p(stream or None), C.byref(h))
        if rc != OK:
            raise MxvError(rc, (lib.mxv_norm_last_error(None) or b"").decode())
        self._h = h

    def _check(self, rc: int):
        if rc != OK:
            raise MxvError(rc, (lib.mxv_norm_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib.mxv_norm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr: int):
        self._check(lib.mxv_norm_set_stream(self._h, C.c_void_p(stream_ptr or None)))

    def get_state(self, want_returns: bool = False):
        """(mean[dim], var[dim], count[, returns[num_envs]]) as float64 host values."""
        mean = np.zeros(self.dim, np.float64)
        var = np.zeros(self.dim, np.float64)
        count = C.c_double()
        ret = np.zeros(self.num_envs, np.float64) if want_returns else None
        self._check(lib.mxv_norm_get_state(self._h, mean.ctypes.data, var.ctypes.data, C.addressof(count), _ptr(ret)))
        return (mean, var, count.value, ret) if want_returns else (mean, var, count.value)

    def set_state(self, mean, var, count: float, returns=None):
        m = np.ascontiguousarray(mean, dtype=np.float64).reshape(self.dim)
        v = np.ascontiguousarray(var, dtype=np.float64).reshape(self.dim)
        r = None if returns is None else np.ascontiguousarray(returns, dtype=np.float64).reshape(self.num_envs)
        self._check(lib.mxv_norm_set_state(self._h, m.ctypes.data, v.ctypes.data, float(count), _ptr(r)))

    def observations(self, K, x_dev, y_dev, out_f32: bool, epsilon: float):
        self._check(lib.mxv_norm_observations(self._h, int(K), _ptr(x_dev), _ptr(y_dev), int(out_f32), float(epsilon)))

    def rewards(self, K, reward_dev, reward_f32: bool, terminated_dev, truncated_dev, out_dev, gamma: float, epsilon: float):
        self._check(lib.mxv_norm_rewards(self._h, int(K), _ptr(reward_dev), int(reward_f32), _ptr(terminated_dev),
                                         _ptr(truncated_dev), _ptr(out_dev), float(gamma), float(epsilon)))

    def obs_sums(self, K, x_dev, sums_dev):
        self._check(lib.mxv_norm_obs_sums(self._h, int(K), _ptr(x_dev), _ptr(sums_dev)))

    def reward_sums_partials(self, K, partials_dev, leaves: int, sums_dev):
        self._check(lib.mxv_norm_reward_sums_partials(self._h, int(K), _ptr(partials_dev), int(leaves), _ptr(sums_dev)))

    def returns_ptr(self) -> int:
        """Device address of the running discounted returns [N] float64 (mxv_norm_returns_ptr)."""
        p = C.c_void_p()
        self._check(lib.mxv_norm_returns_ptr(self._h, C.byref(p)))
        return int(p.value)

    def obs_sums_partials(self, K, partials_dev, leaves: int, sums_dev):
        """[K][leaves][2 dim] partial sums a rollout left behind (Handle.set_obs_partials) -> sums_dev [K][2 dim]."""
        self._check(lib.mxv_norm_obs_sums_partials(self._h, int(K), _ptr(partials_dev), int(leaves), _ptr(sums_dev)))

    def obs_apply(self, K, x_dev, y_dev, out_f32: bool, epsilon: float, all_sums_dev, world: int, total_rows: int):
        self._check(lib.mxv_norm_obs_apply(self._h, int(K), _ptr(x_dev), _ptr(y_dev), int(out_f32), float(epsilon),
                                           _ptr(all_sums_dev), int(world), int(total_rows)))

    def reward_sums(self, K, reward_dev, reward_f32: bool, terminated_dev, truncated_dev, gamma: float, sums_dev):
        self._check(lib.mxv_norm_reward_sums(self._h, int(K), _ptr(reward_dev), int(reward_f32), _ptr(terminated_dev),
                                             _ptr(truncated_dev), float(gamma), _ptr(sums_dev)))

    def reward_apply(self, K, reward_dev, reward_f32: bool, out_dev, epsilon: float, all_sums_dev, world: int, total_rows: int):
        self._check(lib.mxv_norm_reward_apply(self._h, int(K), _ptr(reward_dev), int(reward_f32), _ptr(out_dev),
                                              float(epsilon), _ptr(all_sums_dev), int(world), int(total_rows)))


class SubNorm:
    """One mxv_subnorm = num_envs device-resident RunningMeanStd objects of shape (dim,), each fed with batches of one row (+ every
    sub-env's discounted return): the per-sub-env NormalizeObservation / NormalizeReward of `make(wrappers=[...])`; see
    include/mxv_norm.h."""

    def __init__(self, dim: int, num_envs: int, *, device: int = 0, stream: int = 0):
        self.dim, self.num_envs, self.device = int(dim), int(num_envs), int(device)
        h = C.c_void_p()
        rc = lib.mxv_subnorm_create(self.device, self.dim, self.num_envs, C.c_void_p(stream or None), C.byref(h))
        if rc != OK:
            raise MxvError(rc, (lib.mxv_subnorm_last_error(None) or b"").decode())
        self._h = h

    def _check(self, rc: int):
        if rc != OK:
            raise MxvError(rc, (lib.mxv_subnorm_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib.mxv_subnorm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr: int):
        self._check(lib.mxv_subnorm_set_stream(self._h, C.c_void_p(stream_ptr or None)))

    def get_state(self):
        """(mean[num_envs][dim], var[num_envs][dim], count[num_envs], returns[num_envs]) as float64 host arrays."""
        mean = np.zeros((self.num_envs, self.dim), np.float64)
        var = np.zeros((self.num_envs, self.dim), np.float64)
        count = np.zeros(self.num_envs, np.float64)
        ret = np.zeros(self.num_envs, np.float64)
        self._check(lib.mxv_subnorm_get_state(self._h, mean.ctypes.data, var.ctypes.data, count.ctypes.data, ret.ctypes.data))
        return mean, var, count, ret

    def set_state(self, mean, var, count, returns=None):
        m = np.ascontiguousarray(mean, dtype=np.float64).reshape(self.num_envs, self.dim)
        v = np.ascontiguousarray(var, dtype=np.float64).reshape(self.num_envs, self.dim)
        c = np.ascontiguousarray(np.broadcast_to(np.asarray(count, dtype=np.float64), (self.num_envs,)))
        r = None if returns is None else np.ascontiguousarray(returns, dtype=np.float64).reshape(self.num_envs)
        self._check(lib.mxv_subnorm_set_state(self._h, m.ctypes.data, v.ctypes.data, c.ctypes.data, _ptr(r)))

    def observations(self, K, x_dev, final_dev, terminated_dev, truncated_dev, y_dev, out_f32: bool, final_y_dev, epsilon: float):
        self._check(lib.mxv_subnorm_observations(self._h, int(K), _ptr(x_dev), _ptr(final_dev), _ptr(terminated_dev), _ptr(truncated_dev),
                                                 _ptr(y_dev), int(out_f32), _ptr(final_y_dev), float(epsilon)))
        return y_dev

    def rewards(self, K, reward_dev, reward_f32: bool, terminated_dev, truncated_dev, out_dev, gamma: float, epsilon: float):
        self._check(lib.mxv_subnorm_rewards(self._h, int(K), _ptr(reward_dev), int(reward_f32), _ptr(terminated_dev), _ptr(truncated_dev),
                                            _ptr(out_dev), float(gamma), float(epsilon)))
        return out_dev


class Tab:
    """One mxv_tab handle = one device + one stream + N device-resident copies of a tabular MDP (see include/mxv.h)."""

    def __init__(self, num_states, num_actions, cum_prob, prob, next_state, reward, terminated, initial_cum, num_envs,
                 max_episode_steps, *, device=0, env_offset=0, seed=0, action_seed=0, compact=False, general_kernel=False):
        """compact=True (MXV_TAB_FLAG_COMPACT): rollout() / rollout_tape() take and produce int32 observations / actions and float32
        rewards / probs; everything else keeps the reference's int64 / float64.  general_kernel=True (MXV_TAB_FLAG_GENERAL_KERNEL):
        trajectory rollouts stay on the general kernel even where the specialised one applies (the tests' A/B switch)."""
        S, A = int(num_states), int(num_actions)
        self.compact = bool(compact)
        cum = np.ascontiguousarray(cum_prob, dtype=np.float64)
        assert cum.ndim == 3 and cum.shape[:2] == (S, A), cum.shape
        M = cum.shape[2]
        pr = np.ascontiguousarray(prob, dtype=np.float64).reshape(S, A, M)
        nx = np.ascontiguousarray(next_state, dtype=np.int32).reshape(S, A, M)
        rw = np.ascontiguousarray(reward, dtype=np.float64).reshape(S, A, M)
        te = np.ascontiguousarray(terminated, dtype=np.uint8).reshape(S, A, M)
        ic = np.ascontiguousarray(initial_cum, dtype=np.float64).reshape(S)
        self.S, self.A, self.M, self.num_envs, self.device = S, A, M, int(num_envs), int(device)
        cfg = MxvTabConfig(self.device, S, A, M, self.num_envs, int(env_offset), int(max_episode_steps),
                           (TAB_FLAG_COMPACT if compact else 0) | (TAB_FLAG_GENERAL_KERNEL if general_kernel else 0),
                           int(seed) & (2**64 - 1), int(action_seed) & (2**64 - 1))
        h = C.c_void_p()
        rc = lib.mxv_tab_create(C.byref(cfg), cum.ctypes.data, pr.ctypes.data, nx.ctypes.data, rw.ctypes.data,
                                te.ctypes.data, ic.ctypes.data, C.byref(h))
        if rc != OK:
            raise MxvError(rc, (lib.mxv_tab_last_error(None) or b"").decode())
        self._h = h
        self._base_seed, self._per_env_seeds = int(seed) & (2**64 - 1), None
        self._action_seed = int(action_seed) & (2**64 - 1)

    def _check(self, rc: int):
        if rc != OK:
            raise MxvError(rc, (lib.mxv_tab_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib.mxv_tab_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def seed(self, base_seed: int, per_env_seeds=None):
        p = None
        if per_env_seeds is not None:
            per_env_seeds = np.ascontiguousarray(per_env_seeds, dtype=np.uint64)
            assert per_env_seeds.shape == (self.num_envs,)
            p = per_env_seeds.ctypes.data
        self._check(lib.mxv_tab_seed(self._h, int(base_seed) & (2**64 - 1), p))
        self._base_seed = int(base_seed) & (2**64 - 1)
        self._per_env_seeds = None if per_env_seeds is None else per_env_seeds.copy()

    def seed_actions(self, action_seed: int):
        self._check(lib.mxv_tab_seed_actions(self._h, int(action_seed) & (2**64 - 1)))
        self._action_seed = int(action_seed) & (2**64 - 1)

    def snapshot(self) -> dict:
        """State + TimeLimit counters + RNG seeds and counters (the table itself belongs to the caller's MDP)."""
        st, el = self.get_state()
        t, r = self.get_counters()
        on = getattr(self, "_stats_on", False)
        return dict(num_envs=self.num_envs, state=st, elapsed=el, t=t, r=r, base_seed=self._base_seed,
                    per_env_seeds=None if self._per_env_seeds is None else self._per_env_seeds.copy(),
                    action_seed=self._action_seed, stats_on=on,
                    running_returns=self.episode_stats_host(want_running=True)[2] if on else None)

    def restore(self, snap: dict):
        if snap["num_envs"] != self.num_envs:
            raise ValueError(f"snapshot of {snap['num_envs']} envs does not fit this handle ({self.num_envs})")
        self.seed(snap["base_seed"], snap["per_env_seeds"])
        self.seed_actions(snap["action_seed"])
        self.set_state(snap["state"], snap["elapsed"])
        self.set_counters(snap["t"], snap["r"])
        if snap.get("stats_on"):
            self.episode_stats(True)
            self.set_running_returns(snap["running_returns"])

    def set_device_clock(self, on: bool = True):
        """The step index in device memory, advanced on the stream: step / rollout calls become recordable in a caller's hipGraph."""
        self._check(lib.mxv_tab_set_device_clock(self._h, 1 if on else 0))

    def last_kernel(self) -> int:
        """TAB_KERNEL_GENERAL / TAB_KERNEL_TRAJECTORY: which kernel the last step / rollout call launched (mxv_tab_last_kernel)."""
        return int(lib.mxv_tab_last_kernel(self._h))

    def reset(self, obs_dev=None, mask_dev=None):
        self._check(lib.mxv_tab_reset(self._h, _ptr(mask_dev), _ptr(obs_dev)))

    def step(self, actions_dev, obs_dev, reward_dev=None, terminated_dev=None, truncated_dev=None, prob_dev=None,
             final_obs_dev=None, final_prob_dev=None, uniforms_dev=None):
        self._check(lib.mxv_tab_step(self._h, _ptr(actions_dev), _ptr(uniforms_dev), _ptr(obs_dev), _ptr(reward_dev),
                                     _ptr(terminated_dev), _ptr(truncated_dev), _ptr(prob_dev), _ptr(final_obs_dev),
                                     _ptr(final_prob_dev)))

    def rollout(self, K, obs_dev, reward_dev=None, terminated_dev=None, truncated_dev=None, prob_dev=None,
                final_obs_dev=None, final_prob_dev=None, actions_out_dev=None, per_step=False):
        self._check(lib.mxv_tab_rollout(self._h, int(K), int(per_step), _ptr(actions_out_dev), _ptr(obs_dev),
                                        _ptr(reward_dev), _ptr(terminated_dev), _ptr(truncated_dev), _ptr(prob_dev),
                                        _ptr(final_obs_dev), _ptr(final_prob_dev)))

    def rollout_tape(self, K, actions_tape_dev, obs_dev, reward_dev=None, terminated_dev=None, truncated_dev=None,
                     prob_dev=None, final_obs_dev=None, final_prob_dev=None, per_step=False):
        self._check(lib.mxv_tab_rollout_tape(self._h, int(K), int(per_step), _ptr(actions_tape_dev), _ptr(obs_dev),
                                             _ptr(reward_dev), _ptr(terminated_dev), _ptr(truncated_dev), _ptr(prob_dev),
                                             _ptr(final_obs_dev), _ptr(final_prob_dev)))

    def reset_host(self, mask=None) -> np.ndarray:
        obs = np.empty(self.num_envs, dtype=np.int64)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self._check(lib.mxv_tab_reset_host(self._h, _ptr(m), obs.ctypes.data))
        return obs

    def step_host(self, actions, uniforms=None, pooled=False):
        """-> obs i64[N], reward f64[N], terminated bool[N], truncated bool[N], prob f64[N], final_obs i64[N], final_prob f64[N]
        (final_* valid where terminated | truncated).  uniforms: None or float64 [2][N] (transition, autoreset).
        pooled=True (the NumPy adapter): the arrays come from a recycling pool (_ArrayPool: no fresh mappings per step) and
        final_* hold stale values, not zeros, where no episode ended."""
        n = self.num_envs
        a = np.ascontiguousarray(actions, dtype=np.int64).reshape(n)
        u = None if uniforms is None else np.ascontiguousarray(uniforms, dtype=np.float64).reshape(2, n)
        if pooled:
            pool = self.__dict__.setdefault("_pool", _ArrayPool(limit=12))
            obs, rew, prob = pool.take((n,), np.int64), pool.take((n,), np.float64), pool.take((n,), np.float64)
            term, trunc = pool.take((n,), np.uint8), pool.take((n,), np.uint8)
            fin, fprob = pool.take((n,), np.int64), pool.take((n,), np.float64)
        else:
            obs = np.empty(n, np.int64)
            rew = np.empty(n, np.float64)
            term = np.empty(n, np.uint8)
            trunc = np.empty(n, np.uint8)
            prob = np.empty(n, np.float64)
            fin = np.zeros(n, np.int64)
            fprob = np.zeros(n, np.float64)
        self._check(lib.mxv_tab_step_host(self._h, a.ctypes.data, _ptr(u), obs.ctypes.data, rew.ctypes.data, term.ctypes.data,
                                          trunc.ctypes.data, prob.ctypes.data, fin.ctypes.data, fprob.ctypes.data))
        return obs, rew, term.view(np.bool_), trunc.view(np.bool_), prob, fin, fprob

    # -- episode statistics (gym.wrappers.RecordEpisodeStatistics fused into the kernels; include/mxv_toytext.h) --------------------------
    def episode_stats(self, enable: bool = True):
        self._check(lib.mxv_tab_episode_stats(self._h, 1 if enable else 0))
        self._stats_on = bool(enable)

    def set_episode_outputs(self, ep_return_dev=None, ep_length_dev=None):
        """[N] / [K][N] device arrays (float32 returns, int32 lengths) the device-pointer calls fill where an episode ended; None detaches."""
        self._check(lib.mxv_tab_set_episode_outputs(self._h, _ptr(ep_return_dev), _ptr(ep_length_dev)))

    def episode_stats_host(self, want_running: bool = False):
        """(returns, lengths[, running_returns]) of the last step_host call; valid where terminated | truncated."""
        r = np.zeros(self.num_envs, dtype=np.float32)
        l = np.zeros(self.num_envs, dtype=np.int32)
        run = np.zeros(self.num_envs, dtype=np.float32) if want_running else None
        self._check(lib.mxv_tab_episode_stats_host(self._h, r.ctypes.data, l.ctypes.data, _ptr(run)))
        return (r, l, run) if want_running else (r, l)

    def set_running_returns(self, running):
        r = np.ascontiguousarray(running, dtype=np.float32).reshape(self.num_envs)
        self._check(lib.mxv_tab_set_running_returns(self._h, r.ctypes.data))

    def get_state(self):
        st = np.empty(self.num_envs, dtype=np.int32)
        el = np.empty(self.num_envs, dtype=np.int32)
        self._check(lib.mxv_tab_get_state(self._h, st.ctypes.data, el.ctypes.data))
        return st, el

    def set_state(self, state=None, elapsed=None):
        st = None if state is None else np.ascontiguousarray(state, dtype=np.int32).reshape(self.num_envs)
        el = None if elapsed is None else np.ascontiguousarray(elapsed, dtype=np.int32).reshape(self.num_envs)
        self._check(lib.mxv_tab_set_state(self._h, _ptr(st), _ptr(el)))

    def get_counters(self):
        t, r = C.c_uint64(), C.c_uint32()
        self._check(lib.mxv_tab_get_counters(self._h, C.byref(t), C.byref(r)))
        return t.value, r.value

    def set_counters(self, t: int, r: int):
        self._check(lib.mxv_tab_set_counters(self._h, int(t), int(r)))

    def sync(self):
        self._check(lib.mxv_tab_sync(self._h))

    def set_stream(self, stream_ptr: int):
        self._check(lib.mxv_tab_set_stream(self._h, C.c_void_p(stream_ptr)))


# Draw contract of the Blackjack engine's Philox streams (include/mxv.h RNG contract): 1 = rounds 2-4 (one card per word, consumed in the
# reference's order; word-per-step actions), 2 = round 5 on (one call per step = eight cards with fixed roles; bit-stream actions).
BJ_DRAW_CONTRACT = 2


class Blackjack:
    """One mxv_bj handle: N device-resident Blackjack-v1 tables (see include/mxv.h)."""

    def __init__(self, num_envs, *, natural=False, sab=False, max_episode_steps=-1, device=0, env_offset=0, seed=0, action_seed=0):
        self.num_envs, self.device = int(num_envs), int(device)
        cfg = MxvBjConfig(self.device, int(bool(natural)), int(bool(sab)), int(max_episode_steps), self.num_envs,
                          int(env_offset), int(seed) & (2**64 - 1), int(action_seed) & (2**64 - 1))
        h = C.c_void_p()
        rc = lib.mxv_bj_create(C.byref(cfg), C.byref(h))
        if rc != OK:
            raise MxvError(rc, (lib.mxv_bj_last_error(None) or b"").decode())
        self._h = h
        self._base_seed, self._per_env_seeds = int(seed) & (2**64 - 1), None
        self._action_seed = int(action_seed) & (2**64 - 1)

    def _check(self, rc: int):
        if rc != OK:
            raise MxvError(rc, (lib.mxv_bj_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib.mxv_bj_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def seed(self, base_seed: int, per_env_seeds=None, action_seed: int = 0):
        p = None
        if per_env_seeds is not None:
            per_env_seeds = np.ascontiguousarray(per_env_seeds, dtype=np.uint64)
            assert per_env_seeds.shape == (self.num_envs,)
            p = per_env_seeds.ctypes.data
        self._check(lib.mxv_bj_seed(self._h, int(base_seed) & (2**64 - 1), p, int(action_seed) & (2**64 - 1)))
        self._base_seed = int(base_seed) & (2**64 - 1)
        self._per_env_seeds = None if per_env_seeds is None else per_env_seeds.copy()
        self._action_seed = int(action_seed) & (2**64 - 1)

    def set_device_clock(self, on: bool = True):
        """The step index in device memory, advanced on the stream: step / rollout calls become recordable in a caller's hipGraph."""
        self._check(lib.mxv_bj_set_device_clock(self._h, 1 if on else 0))

    def get_counters(self):
        t, r = C.c_uint64(), C.c_uint32()
        self._check(lib.mxv_bj_get_counters(self._h, C.byref(t), C.byref(r)))
        return t.value, r.value

    def snapshot(self) -> dict:
        st, el = self.get_state()
        t, r = self.get_counters()
        on = getattr(self, "_stats_on", False)
        return dict(num_envs=self.num_envs, state=st, elapsed=el, t=t, r=r, base_seed=self._base_seed,
                    per_env_seeds=None if self._per_env_seeds is None else self._per_env_seeds.copy(),
                    action_seed=self._action_seed, draw_contract=BJ_DRAW_CONTRACT, stats_on=on,
                    running_returns=self.episode_stats_host(want_running=True)[2] if on else None)

    def restore(self, snap: dict):
        if snap["num_envs"] != self.num_envs:
            raise ValueError(f"snapshot of {snap['num_envs']} envs does not fit this handle ({self.num_envs})")
        if snap.get("draw_contract", 1) != BJ_DRAW_CONTRACT:
            # API level 5 changed which cards and actions a (seed, step) pair yields: the hands in the snapshot are valid, the streams that
            # continue them are not the ones the snapshot's run would have drawn (include/mxv.h, "API levels")
            raise ValueError(f"Blackjack snapshot was taken under draw contract {snap.get('draw_contract', 1)} (gym_amd < 0.5); this library "
                             f"draws under contract {BJ_DRAW_CONTRACT}: the run cannot be continued bit-identically — reset and reseed, "
                             "or restore with the library version that took the snapshot")
        self.seed(snap["base_seed"], snap["per_env_seeds"], snap["action_seed"])
        self.set_state(snap["state"], snap["elapsed"], snap["t"], snap["r"])
        if snap.get("stats_on"):
            self.episode_stats(True)
            self.set_running_returns(snap["running_returns"])

    def reset(self, obs_dev=None, mask_dev=None, cards_dev=None):
        self._check(lib.mxv_bj_reset(self._h, _ptr(mask_dev), _ptr(cards_dev), _ptr(obs_dev)))

    def step(self, actions_dev, obs_dev, reward_dev=None, terminated_dev=None, truncated_dev=None, final_obs_dev=None, cards_dev=None):
        self._check(lib.mxv_bj_step(self._h, _ptr(actions_dev), _ptr(cards_dev), _ptr(obs_dev), _ptr(reward_dev),
                                    _ptr(terminated_dev), _ptr(truncated_dev), _ptr(final_obs_dev)))

    def rollout(self, K, obs_dev, reward_dev=None, terminated_dev=None, truncated_dev=None, final_obs_dev=None,
                actions_out_dev=None, actions_tape_dev=None, per_step=False, compact=False):
        """compact: int32 observations / actions_out / final_obs and float32 rewards (mxv_bj_rollout_compact) instead of the
        reference's int64 / float64."""
        fn = lib.mxv_bj_rollout_compact if compact else lib.mxv_bj_rollout
        self._check(fn(self._h, int(K), int(per_step), _ptr(actions_tape_dev), _ptr(actions_out_dev), _ptr(obs_dev), _ptr(reward_dev),
                       _ptr(terminated_dev), _ptr(truncated_dev), _ptr(final_obs_dev)))

    def reset_host(self, cards=None) -> np.ndarray:
        """-> obs int64 [3][N].  cards: None or int8 [N][4] (dealer's two cards, then the player's)."""
        obs = np.empty((3, self.num_envs), np.int64)
        c = None if cards is None else np.ascontiguousarray(cards, dtype=np.int8).reshape(self.num_envs, 4)
        self._check(lib.mxv_bj_reset_host(self._h, _ptr(c), obs.ctypes.data))
        return obs

    def step_host(self, actions, cards=None, pooled=False):
        """-> obs i64[3][N], reward f64[N], terminated bool[N], truncated bool[N], final_obs i64[3][N] (pooled: see Tab.step_host)."""
        n = self.num_envs
        a = np.ascontiguousarray(actions, dtype=np.int64).reshape(n)
        c = None if cards is None else np.ascontiguousarray(cards, dtype=np.int8).reshape(n, BJ_MAX_DRAWS)
        if pooled:
            pool = self.__dict__.setdefault("_pool", _ArrayPool(limit=12))
            obs, fin = pool.take((3, n), np.int64), pool.take((3, n), np.int64)
            rew = pool.take((n,), np.float64)
            term, trunc = pool.take((n,), np.uint8), pool.take((n,), np.uint8)
        else:
            obs = np.empty((3, n), np.int64)
            rew = np.empty(n, np.float64)
            term = np.empty(n, np.uint8)
            trunc = np.empty(n, np.uint8)
            fin = np.zeros((3, n), np.int64)
        self._check(lib.mxv_bj_step_host(self._h, a.ctypes.data, _ptr(c), obs.ctypes.data, rew.ctypes.data, term.ctypes.data,
                                         trunc.ctypes.data, fin.ctypes.data))
        return obs, rew, term.view(np.bool_), trunc.view(np.bool_), fin

    # -- episode statistics (gym.wrappers.RecordEpisodeStatistics fused into the kernels; include/mxv_toytext.h) --------------------------
    def episode_stats(self, enable: bool = True):
        self._check(lib.mxv_bj_episode_stats(self._h, 1 if enable else 0))
        self._stats_on = bool(enable)

    def set_episode_outputs(self, ep_return_dev=None, ep_length_dev=None):
        """[N] / [K][N] device arrays (float32 returns, int32 lengths) the device-pointer calls fill where an episode ended; None detaches."""
        self._check(lib.mxv_bj_set_episode_outputs(self._h, _ptr(ep_return_dev), _ptr(ep_length_dev)))

    def episode_stats_host(self, want_running: bool = False):
        """(returns, lengths[, running_returns]) of the last step_host call; valid where terminated | truncated."""
        r = np.zeros(self.num_envs, dtype=np.float32)
        l = np.zeros(self.num_envs, dtype=np.int32)
        run = np.zeros(self.num_envs, dtype=np.float32) if want_running else None
        self._check(lib.mxv_bj_episode_stats_host(self._h, r.ctypes.data, l.ctypes.data, _ptr(run)))
        return (r, l, run) if want_running else (r, l)

    def set_running_returns(self, running):
        r = np.ascontiguousarray(running, dtype=np.float32).reshape(self.num_envs)
        self._check(lib.mxv_bj_set_running_returns(self._h, r.ctypes.data))

    def get_state(self):
        st = np.empty(self.num_envs, np.int32)
        el = np.empty(self.num_envs, np.int32)
        self._check(lib.mxv_bj_get_state(self._h, st.ctypes.data, el.ctypes.data))
        return st, el

    def set_state(self, state=None, elapsed=None, t: int = 0, r: int = 1):
        st = None if state is None else np.ascontiguousarray(state, dtype=np.int32).reshape(self.num_envs)
        el = None if elapsed is None else np.ascontiguousarray(elapsed, dtype=np.int32).reshape(self.num_envs)
        self._check(lib.mxv_bj_set_state(self._h, _ptr(st), _ptr(el), int(t), int(r)))

    def sync(self):
        self._check(lib.mxv_bj_sync(self._h))

    def set_stream(self, stream_ptr: int):
        self._check(lib.mxv_bj_set_stream(self._h, C.c_void_p(stream_ptr)))
