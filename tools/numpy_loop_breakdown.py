#!/usr/bin/env python3
"""Where the time of HipVectorEnv.step goes at 2^20 envs: the C-ABI calls alone, the adapter's NumPy work, the DMA ceiling."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gym_amd import _native


def t(fn, reps=20):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return round((time.perf_counter() - t0) / reps * 1e6, 1)


out = {}
for n in (1 << 16, 1 << 20):
    h = _native.Handle(_native.CARTPOLE, n, 500, seed=0, action_seed=1)
    h.reset_host()
    a = np.random.default_rng(0).integers(0, 2, n)
    out[f"step_host_fresh_arrays_{n}"] = t(lambda: h.step_host(a, want_final=True))
    out[f"step_host_pooled_{n}"] = t(lambda: h.step_host(a, want_final=True, pooled=True))
    out[f"step_host_pooled_nofinal_{n}"] = t(lambda: h.step_host(a, want_final=False, pooled=True))
    h.final_packed(True)
    out[f"step_host_pooled_packed_final_{n}"] = t(lambda: (h.step_host(a, want_final=True, pooled=True), h.final_packed_rows()))
    h.final_packed(False)
    io = h.host_io()
    h.reset_mapped()
    io["actions"][:] = a
    out[f"step_mapped_{n}"] = t(lambda: h.step_mapped())
    out[f"copy_actions_into_pinned_{n}"] = t(lambda: np.copyto(io["actions"], a))
    out[f"read_pinned_obs_copy_{n}"] = t(lambda: io["obs"].copy())
    out[f"or_flags_pinned_{n}"] = t(lambda: io["terminated"] | io["truncated"])
    o = np.empty((n, 4), np.float32)
    out[f"pageable_copy_16B_per_env_{n}"] = t(lambda: np.copyto(o, o))
    d = (np.arange(n) % 22 == 0)
    out[f"flatnonzero_{n}"] = t(lambda: np.flatnonzero(d))
    h.close()
print(json.dumps(out, indent=1))
