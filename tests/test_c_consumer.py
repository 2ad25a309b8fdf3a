"""include/mxv.h consumed from C (tests/c_consumer/abi_consumer.c), with no ctypes mirror of the prototypes or struct layouts in between.

CPU: the header is valid strict C99 and C++11; the consumer compiles against it with -Wall -Wextra -Werror, loads libmxv.so, resolves every
symbol it uses and calls the device-free entry points; sizeof(mxv_config) as C sees it equals the ctypes mirror's.
GPU: the consumer steps all five env kinds through mxv_reset_host / mxv_step_host and holds them against the oracle
(oracle/_build/liborc.so, opened by the C program itself) step by step — the C-language twin of __graft_entry__.smoke()."""
import ctypes
import os
import shutil
import subprocess
import tempfile

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "c_consumer", "abi_consumer.c")
HDR = os.path.join(ROOT, "include", "mxv.h")
LIB = os.path.join(ROOT, "gym_amd", "_lib", "libmxv.so")
ORC = os.path.join(ROOT, "oracle", "_build", "liborc.so")


@pytest.fixture(scope="module")
def consumer():
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    d = tempfile.mkdtemp(prefix="mxv_cc_")
    exe = os.path.join(d, "abi_consumer")
    p = subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", exe, SRC, "-ldl", "-lm"],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    yield exe
    shutil.rmtree(d, ignore_errors=True)


def test_header_is_strict_c99_and_cxx11():
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", HDR],
                ["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++", HDR],
                ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", HDR.replace("mxv.h", "mxv_diag.h")]):
        if shutil.which(cmd[0]) is None:
            pytest.skip(f"no {cmd[0]}")
        p = subprocess.run(cmd, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr


def test_c_consumer_links_the_symbols_it_uses_and_agrees_on_the_config_layout(consumer):
    p = subprocess.run([consumer, "--symbols-only", LIB], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "symbols ok" in p.stdout, p.stdout + p.stderr
    from gym_amd import _native

    size_in_c = int(p.stdout.split("sizeof(mxv_config) = ")[1].split()[0])
    assert size_in_c == ctypes.sizeof(_native.MxvConfig)
    # every struct that crosses the boundary: size and the offset of every field, as the C compiler lays the header's struct out, against
    # the ctypes mirror the Python host uses
    mirrors = {"mxv_config": _native.MxvConfig, "mxv_tab_config": _native.MxvTabConfig, "mxv_bj_config": _native.MxvBjConfig,
               "mxv_placed_info": _native.MxvPlacedInfo, "mxv_step_outputs": _native.StepOutputs, "mxv_launch_info": _native.MxvLaunchInfo}
    seen = set()
    for line in p.stdout.splitlines():
        if not line.startswith("layout "):
            continue
        _, name, size, *fields = line.split()
        mirror = mirrors[name]
        seen.add(name)
        assert int(size) == ctypes.sizeof(mirror), name
        c_fields = dict(f.split("=") for f in fields)
        if c_fields:
            assert list(c_fields) == [f[0] for f in mirror._fields_], name          # same fields, same order
        for fname, off in c_fields.items():
            assert getattr(mirror, fname).offset == int(off), (name, fname)
    assert seen == set(mirrors)


@pytest.mark.gpu
def test_c_consumer_steps_all_five_env_kinds_against_the_oracle(consumer):
    p = subprocess.run([consumer, LIB, ORC, "1000", "260"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = [l for l in p.stdout.splitlines() if " ok " in l]
    assert len(lines) == 6 and "all five env kinds and Blackjack agree" in p.stdout, p.stdout
    for l in lines:                                   # TimeLimit 37: every env ends ~7 episodes in 260 steps
        assert int(l.split("episodes_ended=")[1].split()[0]) >= 5000, l
