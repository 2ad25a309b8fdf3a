"""The C ABI under hostile arguments and under sanitizers (SURVEY.md §5 "race detection / sanitizers").

tests/c_consumer/gen_abi_fuzz.py turns every prototype of include/*.h into calls of a plain-C program (no ctypes, no Python between the
caller and the library): phase 1 — zeros and NULLs everywhere, no device needed; phases 2-4 — live objects with NULL / hostile /
misaligned everything else.  Each call must return a status code (with a message on failure), never crash.

CPU:  the program linked against gym_amd/_lib/asan/libmxv_asan.so (gym_amd/csrc/build_asan.sh: the library's host side under
      AddressSanitizer + UBSan, leak detection on) runs phase 1 over all declared entry points — the count is asserted equal to the headers'.
GPU:  phases 1-4 against the shipped libmxv.so, and again against the sanitized library."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

import pytest

from conftest import ROOT

CLANG = "/opt/rocm/lib/llvm/bin/clang"
GEN = os.path.join(ROOT, "tests", "c_consumer", "gen_abi_fuzz.py")
ASAN_LIB = os.path.join(ROOT, "gym_amd", "_lib", "asan", "libmxv_asan.so")
SAN = ["-fsanitize=address,undefined", "-fno-sanitize=alignment", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def _declared():
    sys.path.insert(0, os.path.dirname(GEN))
    try:
        import gen_abi_fuzz
        return gen_abi_fuzz.prototypes()
    finally:
        sys.path.pop(0)


def _asan_library():
    """The sanitized library, rebuilt when a source is newer than it (90 s of hipcc; __graft_entry__.build() makes it too)."""
    csrc = os.path.join(ROOT, "gym_amd", "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    if not os.path.exists(ASAN_LIB) or os.path.getmtime(ASAN_LIB) < max(os.path.getmtime(f) for f in srcs):
        if not os.path.exists("/opt/rocm/bin/hipcc"):
            pytest.skip("no hipcc to build the sanitized library")
        subprocess.check_call(["bash", os.path.join(csrc, "build_asan.sh")], stdout=subprocess.DEVNULL)
    return ASAN_LIB


def _build(tmp, lib_path, sanitize):
    src = os.path.join(tmp, "abi_fuzz.c")
    with open(src, "w") as f:
        subprocess.check_call([sys.executable, GEN], stdout=f)
    exe = os.path.join(tmp, "abi_fuzz_asan" if sanitize else "abi_fuzz")
    libdir, libname = os.path.dirname(lib_path), os.path.basename(lib_path)[3:-3]
    cc = [CLANG] + SAN if sanitize else ["gcc"]
    p = subprocess.run(cc + ["-std=gnu99", "-Wall", "-Wextra", "-Wno-comment", "-I", os.path.join(ROOT, "include"), "-o", exe, src, f"-L{libdir}",
                             f"-l{libname}", f"-Wl,-rpath,{libdir}"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    return exe


def _run(exe, *args, timeout=60):
    import signal

    p = subprocess.Popen([exe, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=ENV)
    try:
        so, se = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:         # a hang: SIGABRT makes the program name the call it sits in before it dies
        p.send_signal(signal.SIGABRT)
        try:
            so, se = p.communicate(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
            so, se = p.communicate()
        raise AssertionError(f"abi_fuzz hung (> {timeout} s): " + (so + se)[-2000:])
    out = so + se
    m = re.search(r"abi_fuzz: functions_declared=(\d+) calls=(\d+) bad=(\d+) phases=(\S+)", out)
    assert m and "CRASH" not in out, out[-3000:]
    assert "AddressSanitizer" not in out and "runtime error:" not in out and "LeakSanitizer" not in out, out[-4000:]
    assert p.returncode == 0 and int(m.group(3)) == 0, out[-3000:]
    return int(m.group(1)), int(m.group(2)), m.group(4)


@pytest.fixture()
def tmp():
    d = tempfile.mkdtemp(prefix="mxv_fuzz_")
    yield d
    shutil.rmtree(d, ignore_errors=True)


def test_every_entry_point_refuses_null_objects_under_asan_and_ubsan(tmp):
    """Phase 1 on the CPU box: all declared entry points called with zeros / NULLs through the sanitized library.  Every function that takes
    an object handle refuses NULL with a negative status (destroy / free / last_error accept it by definition); nothing crashes, leaks
    or trips UBSan; the number of functions covered equals the number the headers declare."""
    if not os.path.exists(CLANG):
        pytest.skip("no clang with sanitizer runtimes")
    protos = _declared()
    from gym_amd import _native

    assert len(protos) == len(_native.EXPORTS) >= 135 and sorted(fn for _, fn, _ in protos) == sorted(_native.EXPORTS)
    exe = _build(tmp, _asan_library(), sanitize=True)
    declared, calls, phases = _run(exe)
    assert declared == len(protos) == calls and phases == "1"


def test_the_generated_program_is_the_same_every_run(tmp):
    """The generator's hostile values are deterministic (crc32, not hash()): the same program every run."""
    a = subprocess.check_output([sys.executable, GEN])
    b = subprocess.check_output([sys.executable, GEN], env=dict(os.environ, PYTHONHASHSEED="12345"))
    assert a == b and a.count(b"cur_phase = ") == 4


@pytest.mark.gpu
def test_live_objects_survive_null_hostile_and_misaligned_arguments(tmp):
    """Phases 1-4 on the device against the shipped library: live CartPole / tabular / Blackjack / normaliser objects, every other
    argument NULL, then hostile integers with misaligned pointers, then valid integers with misaligned pointers.  Status codes only,
    a message with every failure, and the objects still work afterwards (the program destroys them cleanly)."""
    from gym_amd import _native

    exe = _build(tmp, _native.LIB_PATH, sanitize=False)
    declared, calls, phases = _run(exe, "gpu")
    assert phases == "1-4" and calls > 3 * declared


@pytest.mark.gpu
def test_live_objects_under_the_sanitized_library(tmp):
    """The same four phases with the library's host side under AddressSanitizer + UBSan (prebuilt by __graft_entry__.build(): the GPU box
    only runs it).  Leak detection is off here: the HIP runtime keeps process-lifetime allocations the leak checker cannot tell from ours."""
    if not os.path.exists(ASAN_LIB) or not os.path.exists(CLANG):
        pytest.skip("sanitized library or clang not present on this box")
    exe = _build(tmp, ASAN_LIB, sanitize=True)
    global ENV
    env_before = ENV
    ENV = dict(ENV, ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0")
    try:
        declared, calls, phases = _run(exe, "gpu")
    finally:
        ENV = env_before
    assert phases == "1-4" and calls > 3 * declared
