#!/usr/bin/env python3
"""Where do Acrobot's near-threshold mask differences come from?  (What's weak #1)

tests/golden/Acrobot_p1_threshold.npz holds 4096 reference steps whose post-step height -cos(t1) - cos(t2 + t1) lies within 0..15 000
ulps of the termination threshold 1.0 (acrobot.py:235).  This tool counts, per arithmetic variant, the masks that differ from the
reference's (glibc) and from the reference on a correctly rounded libm (Acrobot_p1_threshold_cr.npz):

    python tools/acrobot_threshold_ab.py --cpu                      # host emulation of the hot path (tools/acrobot_threshold_ab.c) + the exact path
    python tools/acrobot_threshold_ab.py --gpu name=lib.so [...]    # the device itself, one libmxv build variant per entry

One JSON line per variant.  Committed result: profiles/r4/r4a_acrobot_threshold_flip_split.jsonl."""
import ctypes
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "Acrobot_p1_threshold.npz"))
CR = np.load(os.path.join(ROOT, "tests", "golden", "Acrobot_p1_threshold_cr.npz"))
N = len(G["action"])
ULP = np.abs(G["margin"]) / 2.0 ** -52
P = ctypes.c_void_p


def report(name, term, state):
    want, want_cr = G["terminated"].astype(bool), CR["terminated"].astype(bool)
    bad, bad_cr = term != want, term != want_cr
    ang = (state[:, :2] == G["state1"][:, :2]).all(axis=1)
    print(json.dumps({"variant": name, "states": N, "within_8_ulps": int((ULP <= 8).sum()), "within_1_ulp": int((ULP <= 1).sum()),
                      "masks_differing_from_reference_glibc": int(bad.sum()), "of_those_with_bit_equal_post_step_angles": int((bad & ang).sum()),
                      "furthest_differing_mask_ulps": float(ULP[bad].max()) if bad.any() else 0.0,
                      "masks_differing_from_reference_on_correctly_rounded_libm": int(bad_cr.sum()),
                      "post_step_angles_bit_equal_to_reference": float(ang.mean()),
                      "post_step_states_bit_equal_to_reference": float((state == G["state1"]).all(axis=1).mean())}), flush=True)


def cpu():
    d = tempfile.mkdtemp(prefix="acro_ab_")
    # hot.so: the arithmetic of rounds 1-3 (compensated cosine sum; the rows below switch the rest); hot4.so: round 4's hot path
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-DFDLIBM_COS", "-shared", "-fPIC", "-o", f"{d}/hot.so", os.path.join(ROOT, "tools", "acrobot_threshold_ab.c"), "-lm"])
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC", "-o", f"{d}/hot4.so", os.path.join(ROOT, "tools", "acrobot_threshold_ab.c"), "-lm"])
    shim = f"{d}/exact.cpp"
    open(shim, "w").write('#include "%s"\nextern "C" void x_acro(int n, double *st, const long *a, unsigned char *t, double *sc) {\n'
                          '  const double pi = 3.141592653589793, P[12] = {0.2, 1, 1, 1, 1, 0.5, 0.5, 1, 4 * pi, 9 * pi, 0, 0};\n'
                          '  for (int i = 0; i < n; ++i) t[i] = mxv::exact::acrobot_step_exact(P, st + 4 * i, (double)(a[i] - 1), sc + 4 * i);\n}\n'
                          % os.path.join(ROOT, "gym_amd", "csrc", "mxv_exact.hpp"))
    subprocess.check_call(["g++", "-O2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC", "-o", f"{d}/exact.so", shim])
    hot, ex = ctypes.CDLL(f"{d}/hot.so"), ctypes.CDLL(f"{d}/exact.so")
    modes = {"round-3 hot path (own sincos, angle addition with the argument roundings carried) [emulated]": (0, 1, 0),
             "own sincos, angle addition in the stages, terminal cosines direct (own sincos)": (0, 1, 1),
             "own sincos, angle addition in the stages, terminal cosines direct (glibc)": (0, 1, 2),
             "own sincos, every cosine direct": (0, 0, 0),
             "own sincos direct in the stages, terminal cosines glibc": (0, 0, 2),
             "glibc sincos, angle addition everywhere": (1, 1, 0),
             "glibc sincos, angle addition in the stages, terminal cosines direct": (1, 1, 2),
             "glibc, every cosine direct (= oracle/classic_control.c)": (1, 0, 0)}
    for name, (a, b, c) in modes.items():
        hot.set_mode(a, b, c)
        st = np.ascontiguousarray(G["state0"].copy())
        t, h = np.zeros(N, np.uint8), np.zeros(N)
        hot.acro_batch(N, st.ctypes.data_as(P), G["action"].ctypes.data_as(P), t.ctypes.data_as(P), h.ctypes.data_as(P))
        report(name, t.astype(bool), st)
    hot4 = ctypes.CDLL(f"{d}/hot4.so")
    hot4.set_mode(0, 1, 0)
    hot4.set_carry(0)
    st = np.ascontiguousarray(G["state0"].copy())
    t, h = np.zeros(N, np.uint8), np.zeros(N)
    hot4.acro_batch(N, st.ctypes.data_as(P), G["action"].ctypes.data_as(P), t.ctypes.data_as(P), h.ctypes.data_as(P))
    report("round-4 hot path alone (plain angle addition, two-FMA cosine tail; what runs outside the exact band) [emulated]", t.astype(bool), st)
    st = np.ascontiguousarray(G["state0"].copy())
    t, sc = np.zeros(N, np.uint8), np.zeros((N, 4))
    ex.x_acro(N, st.ctypes.data_as(P), G["action"].ctypes.data_as(P), t.ctypes.data_as(P), sc.ctypes.data_as(P))
    report("exact path (mxv_exact.hpp on the host: correctly rounded sincos, every cosine direct)", t.astype(bool), st)


def gpu(name, lib):
    code = ("import os,sys,json,numpy as np; sys.path.insert(0,%r); sys.path.insert(0,%r)\n"
            "from helpers import HipEngine, load_golden\n"
            "g=load_golden('Acrobot','p1_threshold'); n=len(g['action'])\n"
            "e=HipEngine('Acrobot',n,0,autoreset=False); e.set_state(g['state0'].T,np.full(n,5,np.int32))\n"
            "o,r,t,tr,f=e.step(g['action']); np.savez(sys.argv[1],term=t,state=e.get_state()[0].T)\n") % (ROOT, os.path.join(ROOT, "tests"))
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call([sys.executable, "-c", code, f"{d}/o.npz"], env=dict(os.environ, MXV_LIB_PATH=os.path.abspath(lib)))
        o = np.load(f"{d}/o.npz")
        report(f"device: {name}", o["term"].astype(bool), o["state"])


if __name__ == "__main__":
    if "--cpu" in sys.argv:
        cpu()
    if "--gpu" in sys.argv:
        for spec in sys.argv[sys.argv.index("--gpu") + 1:]:
            gpu(*spec.split("=", 1))
