"""gym_amd/csrc/mxv_exact.hpp on the CPU: the header is plain C++, so the very code the device runs in Acrobot's near-threshold band is
compiled here by g++ and held against (a) 400-bit mpmath — cr_sincos must return THE correctly rounded sine and cosine, not "within an
ulp" — and (b) the reference itself run on a correctly rounded libm (tests/golden/Acrobot_p1_threshold_cr.npz, made by
tests/golden/make_golden_acrobot_cr.py): masks and observations of all 4096 threshold states bit for bit."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

HDR = os.path.join(ROOT, "gym_amd", "csrc", "mxv_exact.hpp")
SHIM = r'''
#include "%s"
extern "C" {
void x_sincos(int n, const double *x, double *s, double *c) { for (int i = 0; i < n; ++i) mxv::exact::cr_sincos(x[i], s + i, c + i); }
void x_acro(int n, double *st, const long *a, unsigned char *t, double *sc) {
    const double pi = 3.141592653589793, P[12] = {0.2, 1, 1, 1, 1, 0.5, 0.5, 1, 4 * pi, 9 * pi, 0, 0};
    for (int i = 0; i < n; ++i) t[i] = mxv::exact::acrobot_step_exact(P, st + 4 * i, (double)(a[i] - 1), sc + 4 * i);
}
}
'''
P = ctypes.c_void_p


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("exact")
    src = d / "shim.cpp"
    src.write_text(SHIM % HDR)
    so = d / "libexact.so"
    # -ffp-contract=off as on the device: the double-double primitives must not be re-associated or fused behind their backs
    subprocess.check_call(["g++", "-O2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(so), str(src)])
    return ctypes.CDLL(str(so))


def _sincos(lib, x):
    s, c = np.empty_like(x), np.empty_like(x)
    lib.x_sincos(len(x), x.ctypes.data_as(P), s.ctypes.data_as(P), c.ctypes.data_as(P))
    return s, c


def test_cr_sincos_is_correctly_rounded(lib):
    mp = pytest.importorskip("mpmath")
    mp.mp.prec = 300
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(-16, 16, 12000), rng.uniform(-np.pi, np.pi, 12000),
                        rng.integers(-10, 11, 6000) * np.pi / 2 + rng.uniform(-1e-6, 1e-6, 6000),      # cancellation in the reduction
                        rng.integers(-10, 11, 2000) * np.pi / 2 + rng.uniform(-1e-13, 1e-13, 2000),
                        rng.uniform(-1e-3, 1e-3, 2000), rng.uniform(-3000, 3000, 2000),
                        np.array([0.0, np.pi / 4, -np.pi / 4, np.pi / 2, np.pi, -np.pi, 2 * np.pi, 1e-300, 5e-324])])
    s, c = _sincos(lib, x)
    ws = np.array([float(mp.sin(mp.mpf(v))) for v in x])
    wc = np.array([float(mp.cos(mp.mpf(v))) for v in x])
    assert np.array_equal(s, ws), f"{(s != ws).sum()} sines are not the correctly rounded value"
    assert np.array_equal(c, wc), f"{(c != wc).sum()} cosines are not the correctly rounded value"
    # what the band buys: the libm this process links (the reference's, through NumPy) misses the correctly rounded value now and then
    miss = (np.sin(x) != ws).sum() + (np.cos(x) != wc).sum()
    print(f"glibc / NumPy: {miss} of {2 * len(x)} values are not correctly rounded")


def test_cr_sincos_agrees_with_libm_to_an_ulp(lib):
    """No mpmath needed: within one ulp of the platform libm everywhere, identical for the overwhelming majority."""
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-40, 40, 400000), rng.integers(-20, 21, 50000) * np.pi / 2 + rng.uniform(-1e-9, 1e-9, 50000)])
    s, c = _sincos(lib, x)
    for got, want in ((s, np.sin(x)), (c, np.cos(x))):
        assert np.all(np.abs(got - want) <= np.spacing(np.abs(want)))
        assert (got != want).mean() < 5e-3


def test_exact_acrobot_step_equals_the_reference_on_a_correctly_rounded_libm(lib):
    g = np.load(os.path.join(ROOT, "tests", "golden", "Acrobot_p1_threshold.npz"))
    cr = np.load(os.path.join(ROOT, "tests", "golden", "Acrobot_p1_threshold_cr.npz"))
    n = len(g["action"])
    st = np.ascontiguousarray(g["state0"].copy())
    t, sc = np.zeros(n, np.uint8), np.zeros((n, 4))
    lib.x_acro(n, st.ctypes.data_as(P), g["action"].ctypes.data_as(P), t.ctypes.data_as(P), sc.ctypes.data_as(P))
    assert np.array_equal(t, cr["terminated"]), "masks must be the reference's on a correctly rounded libm, for every threshold state"
    obs = np.stack([sc[:, 1], sc[:, 0], sc[:, 3], sc[:, 2], st[:, 2], st[:, 3]], axis=1).astype(np.float32)
    assert np.array_equal(obs, cr["obs"])
    assert np.array_equal(st[:, :2], cr["state1"][:, :2]), "post-step angles bit for bit"
    # velocities: the reference's `dtheta ** 2` is libm pow (acrobot.py:261, 273), which misses the correctly rounded square in ~1e-4 of
    # the calls; the engine squares by multiplication (exact rounding) — one ulp in a handful of rows
    np.testing.assert_allclose(st, cr["state1"], rtol=3e-16, atol=0)
    assert (st == cr["state1"]).all(axis=1).mean() > 0.995
    # against the glibc run of the reference: the only masks that differ are the ones glibc's own rounding decides
    differ = t != g["terminated"]
    assert np.array_equal(differ, cr["terminated"] != g["terminated"]) and differ.sum() <= 4
    assert np.all(g["margin"][differ] == 0.0), "only heights that ROUND to exactly 1.0 in the glibc run may differ"


def test_glibc_powf_square_restatement_equals_this_images_libm():
    """Pendulum's `u ** 2` is libm powf(u, 2.0f) (pendulum.py:129), which glibc does not round correctly.  The restatement of glibc 2.35's
    algorithm behind the MXV_PENDULUM_GLIBC_POWF build hook (gym_amd/csrc/mxv_device.hpp: glibc_powf_square) has a NumPy twin in
    tools/powf_variants.py: equal to libm's powf on every input, while libm itself differs from the correctly rounded product on ~0.07 %
    of them (profiles/r6/r6g_pendulum_powf.md: why the hook is off by default)."""
    import ctypes
    import ctypes.util
    import importlib.util
    import os

    import numpy as np

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("powf_variants", os.path.join(root, "tools", "powf_variants.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    if b"2.35" not in (os.confstr("CS_GNU_LIBC_VERSION") or "").encode():
        import pytest

        pytest.skip("the restatement is of glibc 2.35's powf (the image's)")
    u = np.random.default_rng(3).uniform(-2, 2, 150_000).astype(np.float32)
    ref = mod.powf2(u)
    assert np.array_equal(mod.glibc_powf_square(u), ref)
    assert 20 < int((ref != (u.astype(np.float64) ** 2).astype(np.float32)).sum()) < 400
