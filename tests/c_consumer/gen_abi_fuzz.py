#!/usr/bin/env python3
"""Generates tests/c_consumer/abi_fuzz.c: a plain-C program that calls EVERY entry point include/*.h declares with hostile arguments and
checks that each call comes back with a status code (and, on failure, a message) instead of crashing.

    python tests/c_consumer/gen_abi_fuzz.py > <dir>/abi_fuzz.c
    clang -std=gnu99 -fsanitize=address,undefined -I include -o abi_fuzz abi_fuzz.c -L gym_amd/_lib/asan -lmxv_asan -Wl,-rpath,...
    ./abi_fuzz            phase 1 only (no device needed)       ./abi_fuzz gpu      all four phases

The calls are generated from the prototypes themselves (the header is the single source: a new entry point is fuzzed as soon as it is
declared; tests/test_abi_fuzz.py asserts that the number of functions covered equals the number declared):

  phase 1  every argument zero / NULL (handles included).  Needs no device.  Every function whose first parameter is an object handle
           must refuse (negative status) — except *_destroy(NULL) and *_last_error(NULL), which are defined to accept it.
  phase 2  a LIVE object of the right kind, every other argument zero / NULL.
  phase 3  a live object, integers hostile (K: 0 and -1; sizes and counts: -1 and INT32_MAX / 2^31; device: 99), pointers MISALIGNED
           (a device-visible pinned allocation + 1 byte).
  phase 4  a live object, integers small and valid (1), pointers misaligned: the deep paths with the alignment checks.

Status codes must lie in [MXV_ERR_UNSUPPORTED, MXV_OK]; a negative status of a call on a live object must leave a non-empty message in
that object's *_last_error.  Nothing may crash, hang (every K passed is <= 1) or trip AddressSanitizer / UBSan in the library's host code.
"""
import os
import re
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HEADERS = ("mxv.h", "mxv_norm.h", "mxv_toytext.h", "mxv_comm.h", "mxv_diag.h")
OBJECTS = {"mxv_handle": ("h", "mxv_last_error"), "mxv_tab": ("tab", "mxv_tab_last_error"), "mxv_bj": ("bj", "mxv_bj_last_error"),
           "mxv_norm": ("nm", "mxv_norm_last_error"), "mxv_subnorm": ("sn", "mxv_subnorm_last_error"), "mxv_placed": (None, "mxv_placed_last_error")}
INTS = ("int32_t", "int64_t", "uint64_t", "uint32_t", "size_t", "int", "unsigned", "uint8_t", "int8_t")


def prototypes():
    out = []
    for name in HEADERS:
        text = open(os.path.join(ROOT, "include", name)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"^([A-Za-z_][\w \*]*?)\b(mxv_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.M | re.S):
            ret, fn, args = m.group(1).strip(), m.group(2), re.sub(r"\s+", " ", m.group(3).strip())
            params = []
            if args != "void":
                for a in args.split(","):
                    a = a.strip()
                    pm = re.match(r"(.*?)(\w+)$", a)
                    params.append((pm.group(1).strip(), pm.group(2)))
            out.append((ret, fn, params))
    return out


def obj_of(ctype):
    base = re.sub(r"\bconst\b", "", ctype).replace("*", "").strip()
    return base if base in OBJECTS and ctype.count("*") == 1 else None


def arg(ctype, name, phase, first, fn):
    """C expression for one argument."""
    o = obj_of(ctype)
    if "*" in ctype:
        if o and first:
            var = OBJECTS[o][0]
            return "NULL" if phase == 1 or var is None else var
        if phase in (1, 2) or (name.endswith("stream") and ctype.count("*") == 1):       # (a hipStream_t is an opaque handle of ANOTHER library: garbage there is HIP's crash, not ours)
            return "NULL"
        if ctype.replace(" ", "") == "mxv_handle*const*":
            return "hs"
        if re.search(r"mxv_\w*config\b", ctype):
            return "NULL" if phase == 3 else f"({ctype})mis"
        if ctype.count("*") == 2:                       # output object pointers: a valid slot (a misaligned one is the caller's own bug)
            return f"({ctype})slot"
        return f"({ctype})mis"
    if ctype in ("double", "float"):
        return "0.0" if phase != 3 else "(0.0 / zero)"        # NaN
    if ctype.split()[-1] in INTS or ctype in INTS:
        if phase in (1, 2):
            return "0"
        if phase == 4:
            return "0" if name in ("device", "env_id", "flags", "enable", "on", "stream", "mode", "per_step") else "1"
        # phase 3: hostile
        if name in ("K", "k", "steps"):
            return "-1" if (zlib.crc32(fn.encode()) & 1) else "0"
        if name == "device":
            return "99"
        if ctype in ("uint64_t", "size_t"):
            return "(~(uint64_t)0)" if ctype == "uint64_t" else "(size_t)-1"
        if ctype == "int64_t":
            return "((int64_t)1 << 31)" if (zlib.crc32((fn + name).encode()) & 1) else "-1"
        return "2147483647" if (zlib.crc32((fn + name).encode()) & 1) else "-1"
    raise SystemExit(f"gen_abi_fuzz: unhandled parameter type {ctype!r} ({name}) of {fn}")


def main():
    protos = prototypes()
    w = sys.stdout.write
    w("/* GENERATED by tests/c_consumer/gen_abi_fuzz.py from the headers under include/ -- do not edit; see that file for what the phases are. */\n")
    w("#include <stdint.h>\n#include <stdio.h>\n#include <stdlib.h>\n#include <string.h>\n\n#include \"mxv.h\"\n#include \"mxv_diag.h\"\n\n")
    w("#include <signal.h>\n#include <unistd.h>\n")
    w("static int n_calls, n_bad, cur_phase;\nstatic volatile double zero = 0.0;\nstatic const char *volatile cur_fn = \"(setup)\";\n")
    w("static void crashed(int sig) {   /* name the call that took the process down, then die with the signal's default action */\n"
      "    char buf[160]; const int n = snprintf(buf, sizeof buf, \"CRASH signal %d in phase %d: %s\\n\", sig, cur_phase, cur_fn);\n"
      "    if (n > 0) { ssize_t r = write(2, buf, (size_t)n); (void)r; }\n    signal(sig, SIG_DFL); raise(sig);\n}\n")
    w("static void status(const char *fn, int rc, const char *msg, int must_fail, int live) {\n"
      "    ++n_calls;\n"
      "    if (rc > MXV_OK || rc < MXV_ERR_UNSUPPORTED) { ++n_bad; printf(\"BAD phase %d %s: status %d is no MXV_* code\\n\", cur_phase, fn, rc); }\n"
      "    if (must_fail && rc >= 0) { ++n_bad; printf(\"BAD phase %d %s: accepted a NULL object (status %d)\\n\", cur_phase, fn, rc); }\n"
      "    if (live && rc < 0 && (!msg || !msg[0])) { ++n_bad; printf(\"BAD phase %d %s: status %d without a message\\n\", cur_phase, fn, rc); }\n"
      "}\n\n")
    w("static int trace;\nstatic mxv_handle *h; static mxv_tab *tab; static mxv_bj *bj; static mxv_norm *nm; static mxv_subnorm *sn;\n"
      "static void settle(void) {   /* ABI_FUZZ_TRACE=1: name every call and drain every stream behind it, so that an asynchronous device fault names its call */\n"
      "    if (!trace) return;\n    fprintf(stderr, \"ok   phase %d %s\\n\", cur_phase, cur_fn);\n"
      "    if (h) { (void)mxv_sync(h); (void)mxv_tab_sync(tab); (void)mxv_bj_sync(bj); (void)mxv_norm_get_state(nm, 0, 0, 0, 0); (void)mxv_subnorm_get_state(sn, 0, 0, 0, 0); }\n"
      "    fprintf(stderr, \"done phase %d %s\\n\", cur_phase, cur_fn);\n}\n\n")
    w("int main(int argc, char **argv) {\n    const int gpu = argc > 1 && !strcmp(argv[1], \"gpu\");\n    trace = getenv(\"ABI_FUZZ_TRACE\") != NULL;\n"
      "    signal(SIGSEGV, crashed); signal(SIGBUS, crashed); signal(SIGFPE, crashed); signal(SIGABRT, crashed); signal(SIGILL, crashed);\n")
    w("    void *pin = NULL; char *mis = NULL; void *slot_store[4] = {0}; void *slot = slot_store; mxv_handle *hs_store[1]; mxv_handle *const *hs = hs_store;\n")
    skip_live = {"mxv_destroy", "mxv_tab_destroy", "mxv_bj_destroy", "mxv_norm_destroy", "mxv_subnorm_destroy", "mxv_host_free", "mxv_placed_free",
                 "mxv_comm_init"}     # (the objects are destroyed at the end; mxv_comm_init with a garbage id block would hand it to RCCL)
    for phase in (1, 2, 3, 4):
        if phase == 2:
            w("    if (!gpu) goto done;\n"
              "    {   /* live objects: 64 CartPole envs, a 2-state MDP, 64 Blackjack tables, two normalisers, 1 MiB of pinned device-visible memory */\n"
              "        mxv_config c; memset(&c, 0, sizeof c); c.env_id = MXV_CARTPOLE; c.num_envs = 64; c.max_episode_steps = 500;\n"
              "        if (mxv_create(&c, &h) != MXV_OK) { printf(\"setup: mxv_create: %s\\n\", mxv_last_error(NULL)); return 3; }\n"
              "        mxv_tab_config t; memset(&t, 0, sizeof t); t.num_states = 2; t.num_actions = 2; t.max_transitions = 1; t.num_envs = 64; t.max_episode_steps = 5;\n"
              "        const double cum[4] = {1, 1, 1, 1}, prob[4] = {1, 1, 1, 1}, rew[4] = {0, 1, 0, 1}, init[2] = {1, 1};\n"
              "        const int32_t nxt[4] = {0, 1, 0, 1}; const uint8_t term[4] = {0, 1, 0, 1};\n"
              "        if (mxv_tab_create(&t, cum, prob, nxt, rew, term, init, &tab) != MXV_OK) { printf(\"setup: mxv_tab_create: %s\\n\", mxv_tab_last_error(NULL)); return 3; }\n"
              "        mxv_bj_config b; memset(&b, 0, sizeof b); b.sab = 1; b.num_envs = 64;\n"
              "        if (mxv_bj_create(&b, &bj) != MXV_OK) { printf(\"setup: mxv_bj_create: %s\\n\", mxv_bj_last_error(NULL)); return 3; }\n"
              "        if (mxv_norm_create(0, 4, 64, NULL, &nm) != MXV_OK || mxv_subnorm_create(0, 4, 64, NULL, &sn) != MXV_OK) { printf(\"setup: normalisers\\n\"); return 3; }\n"
              "        if (mxv_host_alloc((size_t)1 << 20, &pin) != MXV_OK) { printf(\"setup: mxv_host_alloc\\n\"); return 3; }\n"
              "        memset(pin, 0, (size_t)1 << 20); mis = (char *)pin + 1; hs_store[0] = h;\n"
              "        mxv_reset_host(h, NULL, NULL, NULL); mxv_tab_reset_host(tab, NULL, NULL); mxv_bj_reset_host(bj, NULL, NULL);\n"
              "    }\n")
        w(f"    cur_phase = {phase};\n")
        for ret, fn, params in protos:
            first_obj = obj_of(params[0][0]) if params else None
            if phase > 1 and fn in skip_live:
                continue
            if phase > 1 and (first_obj is None or OBJECTS[first_obj][0] is None) and phase == 2:
                continue                                  # phase 2 == phase 1 for functions without a live object
            args = ", ".join(arg(t, n, phase, i == 0, fn) for i, (t, n) in enumerate(params))
            call = f"{fn}({args})"
            if ret == "int" and fn != "mxv_tab_last_kernel":        # (that one returns MXV_TAB_KERNEL_*, not a status)
                live = phase > 1 and first_obj is not None and OBJECTS[first_obj][0] is not None
                msg = f"{OBJECTS[first_obj][1]}({OBJECTS[first_obj][0]})" if live else "NULL"
                must_fail = int(phase == 1 and first_obj is not None and not fn.endswith(("_destroy", "_free")))
                w(f"    cur_fn = \"{fn}\";\n")
                w(f"    {{ const int rc = {call}; status(\"{fn}\", rc, {msg}, {must_fail}, {int(live)}); settle(); }}\n")
            else:
                w(f"    cur_fn = \"{fn}\"; (void){call}; ++n_calls; settle();\n")
        if phase > 1:   # surface asynchronous device faults of this phase here, not in the next one
            w("    (void)mxv_sync(h); (void)mxv_tab_sync(tab); (void)mxv_bj_sync(bj);\n")
    w("done:\n    if (gpu) { mxv_destroy(h); mxv_tab_destroy(tab); mxv_bj_destroy(bj); mxv_norm_destroy(nm); mxv_subnorm_destroy(sn); mxv_host_free(pin); }\n")
    w(f"    printf(\"abi_fuzz: functions_declared={len(protos)} calls=%d bad=%d phases=%s\\n\", n_calls, n_bad, gpu ? \"1-4\" : \"1\");\n")
    w("    return n_bad ? 1 : 0;\n}\n")


if __name__ == "__main__":
    main()
