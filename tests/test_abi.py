"""The C-ABI library: loads, exports every symbol include/mxv.h declares, static tables agree with the oracle.
No compute calls here (no GPU needed); the compute checks are the -m gpu parity tests."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import HAS_GPU, ROOT
from helpers import ENV_IDS, ENV_NAMES
from oracle import oracle


HEADERS = ("mxv.h", "mxv_norm.h", "mxv_toytext.h", "mxv_comm.h", "mxv_diag.h")     # mxv.h includes the next three; mxv_diag.h is optional


def _declared_symbols(headers=HEADERS):
    out = set()
    for name in headers:
        text = open(os.path.join(ROOT, "include", name)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out.update(re.findall(r"\b(mxv_[a-z_0-9]+)\s*\(", text))
    return sorted(out)


def test_library_loads_and_exports_every_declared_symbol():
    from gym_amd import _native

    lib = ctypes.CDLL(_native.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for sym in declared:
        assert hasattr(lib, sym), f"libmxv.so does not export {sym}"
    assert sorted(_native.EXPORTS) == declared, "gym_amd/_native.py binds a different symbol set than include/mxv.h declares"
    assert b"gfx950" in _native.lib.mxv_version()


def test_headers_are_self_contained_and_the_core_is_small():
    """116 entry points in one 55-KB header.  The surface is split by subsystem — mxv.h (the engine a drop-in
    VectorEnv needs) includes mxv_norm.h / mxv_toytext.h / mxv_comm.h; mxv_diag.h (probes, launch introspection, placed memory) is
    optional — every header compiles on its own as C and as C++, and the diagnostics are declared nowhere else."""
    import shutil
    import subprocess
    import tempfile

    core, diag = set(_declared_symbols(("mxv.h",))), set(_declared_symbols(("mxv_diag.h",)))
    assert len(core) <= 64 and not core & diag and {"mxv_last_launch", "mxv_write_probe", "mxv_hbm_pair_probe", "mxv_placed_alloc"} <= diag
    assert all(s.startswith(("mxv_norm_", "mxv_subnorm_")) for s in _declared_symbols(("mxv_norm.h",)))
    assert all(s.startswith(("mxv_tab_", "mxv_bj_")) for s in _declared_symbols(("mxv_toytext.h",)))
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    d = tempfile.mkdtemp(prefix="mxv_hdr_")
    try:
        for name in HEADERS:
            for comp, ext, std in (("gcc", "c", "-std=c99"), ("g++", "cpp", "-std=c++11")):
                src = os.path.join(d, f"t.{ext}")
                open(src, "w").write(f'#include "{name}"\nint main(void) {{ return 0; }}\n')
                p = subprocess.run([comp, std, "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), src],
                                   capture_output=True, text=True)
                assert p.returncode == 0, (name, comp, p.stderr[-1500:])
    finally:
        shutil.rmtree(d, ignore_errors=True)
    text = open(os.path.join(ROOT, "include", "mxv.h")).read()
    assert "#define MXV_API_LEVEL 6" in text and '#include "mxv_diag.h"' not in text


def test_every_exported_symbol_is_mapped_to_a_reference_interface_in_the_integration_notes():
    """INTEGRATION.md is the drop-in contract: each entry point of include/mxv.h appears there next to the reference
    interface (file:line) it replaces, or is marked as having no reference analogue."""
    notes = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [sym for sym in _declared_symbols() if sym not in notes]
    assert not missing, missing


def test_static_tables_match_oracle():
    from gym_amd import _native

    for name in ENV_NAMES:
        k = ENV_IDS[name]
        S, O, NA = _native.env_dims(k)
        assert S == oracle.lib().orc_state_dim(k) and O == oracle.lib().orc_obs_dim(k)
        assert NA == oracle.DISCRETE.get(k, 0)
        assert np.array_equal(_native.default_params(k), oracle.default_params(k))
        assert np.array_equal(_native.default_reset_bounds(k), oracle.default_reset_bounds(k))
    with pytest.raises(_native.MxvError):
        _native.env_dims(9)


def test_config_struct_layout_matches_header():
    from gym_amd import _native

    assert ctypes.sizeof(_native.MxvConfig) == 4 + 4 + 8 + 8 + 4 + 4 + 8 + 8
    assert _native.MxvConfig.num_envs.offset == 8 and _native.MxvConfig.seed.offset == 32


@pytest.mark.skipif(HAS_GPU, reason="only meaningful where no HIP device exists")
def test_no_device_fails_loudly_no_cpu_fallback():
    from gym_amd import _native

    with pytest.raises(_native.MxvError) as ei:
        _native.Handle(0, 8, 500)
    assert ei.value.code in (_native.ERR_HIP, _native.ERR_INVALID_ARG)
    import gym_amd
    with pytest.raises(Exception):
        gym_amd.make("CartPole-v1", 8)


def test_bad_config_rejected_before_touching_a_device():
    from gym_amd import _native

    for kw in (dict(env_id=7, num_envs=8), dict(env_id=0, num_envs=0), dict(env_id=0, num_envs=8, env_offset=3),
               dict(env_id=0, num_envs=(1 << 28) + 4)):      # MXV_MAX_NUM_ENVS: the fused kernels' 32-bit slice offsets
        with pytest.raises(_native.MxvError) as ei:
            _native.Handle(kw["env_id"], kw["num_envs"], 500, env_offset=kw.get("env_offset", 0))
        assert ei.value.code == _native.ERR_INVALID_ARG


def test_product_code_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under gym_amd/ may import, link or execute it."""
    pkg = os.path.join(ROOT, "gym_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", ".sh")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in src.lower().replace("# oracle-free", ""), f"{f} mentions the oracle"
                assert "liborc" not in src
    # the bench: only bench.py's cpu_baseline leg may load the oracle (as the host baseline); benchmarks/*.py and examples/ never import it
    import ast

    for d in ("benchmarks", "examples"):
        for f in sorted(os.listdir(os.path.join(ROOT, d))):
            if not f.endswith(".py"):
                continue
            tree = ast.parse(open(os.path.join(ROOT, d, f)).read())
            mods = [n.module or "" for n in ast.walk(tree) if isinstance(n, ast.ImportFrom)] + \
                   [a.name for n in ast.walk(tree) if isinstance(n, ast.Import) for a in n.names]
            assert not [m for m in mods if m.split(".")[0] == "oracle"], f"{d}/{f} imports the oracle"
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    users = [fn.name for fn in ast.walk(tree) if isinstance(fn, ast.FunctionDef)
             for n in ast.walk(fn) if isinstance(n, ast.ImportFrom) and (n.module or "").split(".")[0] == "oracle"]
    assert users == ["cpu_baseline"], users


def _header_prototypes():
    """{symbol: (return type, [argument type strings])} parsed from include/mxv.h."""
    src = "\n".join(open(os.path.join(ROOT, "include", name)).read() for name in HEADERS)
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    out = {}
    for m in re.finditer(r"(?:^|\n)\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)(mxv_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        out[name] = (ret, [] if args in ("", "void") else [a.strip() for a in args.split(",")])
    return out


def _kind(c_type: str) -> str:
    """Coarse class of a C parameter type: what must agree between the header and a ctypes argtypes entry."""
    t = c_type.replace("const", " ").strip()
    if "*" in t:
        return "pointer"
    base = t.split()[0] if len(t.split()) == 1 else " ".join(t.split()[:-1])      # drop the parameter name
    return {"int": "i32", "int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "uint32_t": "u32", "size_t": "size", "double": "f64",
            "float": "f32"}[base]


def _ctypes_kind(t) -> str:
    if t in (ctypes.c_int, ctypes.c_int32):
        return "i32"
    if t is ctypes.c_int64:
        return "i64"
    if t is ctypes.c_uint64:
        return "u64"
    if t is ctypes.c_uint32:
        return "u32"
    if t is ctypes.c_size_t:
        return "size" if ctypes.sizeof(ctypes.c_size_t) != 8 or True else "u64"
    if t is ctypes.c_double:
        return "f64"
    if t is ctypes.c_float:
        return "f32"
    return "pointer"      # c_void_p, c_char_p, POINTER(...)


def test_ctypes_signatures_agree_with_the_header_prototypes():
    """Every prototype of include/mxv.h against the argtypes / restype gym_amd/_native.py declares for it: the same number of parameters,
    each of the same class (pointer, 32/64-bit signed/unsigned integer, size_t, float, double), the same kind of return value.  A drift
    between the hand-written ctypes mirror and the header would otherwise surface as a crash on the GPU box."""
    from gym_amd import _native

    protos = _header_prototypes()
    assert set(protos) == set(_declared_symbols()), set(protos) ^ set(_declared_symbols())
    undeclared = []
    for name, (ret, args) in sorted(protos.items()):
        f = getattr(_native.lib, name)
        if f.argtypes is None:
            undeclared.append(name)
            continue
        assert len(f.argtypes) == len(args), (name, args, f.argtypes)
        for i, (a, t) in enumerate(zip(args, f.argtypes)):
            got, want = _ctypes_kind(t), _kind(a)
            if want == "size":
                assert t in (ctypes.c_size_t, ctypes.c_uint64), (name, i, a)
            else:
                assert got == want, (name, i, a, t)
        if "char" in ret:
            assert f.restype is ctypes.c_char_p, name
        elif "uint64_t" in ret:
            assert f.restype is ctypes.c_uint64, name
        else:
            assert f.restype in (ctypes.c_int, ctypes.c_int32), (name, ret)
    assert not undeclared, f"exported by the header, but gym_amd/_native.py declares no signature: {undeclared}"
