"""Resource budget of the hot kernels, guarded on the CPU (hipcc cross-compiles gfx950 without a GPU; ~1 minute).

The rollout kernels run at 4 waves per SIMD (128-VGPR budget).  Round 2's CartPole kernel sat AT the cap with 24 registers parked in
scratch around the K-step loop; one more live value would have put scratch traffic — or an unconditional `s_waitcnt vmcnt(0)`, which on
gfx9 also waits for every store in flight — INTO the loop, and the only symptom would have been a slower bench line.  Round 3's
scalar-base stores freed ~10 VGPRs per kernel (no spills anywhere); this test keeps it that way.  This test compiles gym_amd/csrc/mxv_kernels.hip with -Rpass-analysis=kernel-resource-usage
and -save-temps and asserts, for the five trajectory-recording instantiations (one per env kind): the occupancy the launch code
assumes, VGPR spills no larger than today's, and, from the ISA, no scratch access and no unconditional vmcnt wait inside the
innermost (Depth=1) loops that contain the per-step stores."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"
SRC = os.path.join(ROOT, "gym_amd", "csrc", "mxv_kernels.hip")
# env kind -> (envs per lane of the full-size launch, occupancy the launch code pins, VGPR-spill bound = today's figure)
HOT = {0: (2, 4, 0), 1: (1, 4, 0), 2: (1, 4, 0), 3: (2, 4, 0), 4: (2, 4, 0)}


@pytest.fixture(scope="module")
def build():
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    d = tempfile.mkdtemp(prefix="mxv_kres_")
    try:
        p = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-c", SRC,
                            "-o", os.path.join(d, "k.o"), "-Rpass-analysis=kernel-resource-usage", "-save-temps"], cwd=d, capture_output=True,
                           text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")]
        assert asm
        yield p.stderr, open(os.path.join(d, asm[0])).read()
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _resources(remarks):
    out, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"remark: [^ ]+ +(Function Name|VGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill|"
                      r"LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "Function Name":
            cur = out.setdefault(v, {})
        elif cur is not None:
            cur[k.split(" [")[0]] = int(v)
    return out


def _symbol(env, e):
    return f"_ZN3mxv12_GLOBAL__N_117rollout_kernel_v3ILi{env}ELb1ELi{e}ELb0ELi1ELb0ELi0EEEvNS_8StepArgsE"


def _function_body(asm, sym):
    start = asm.index(f"\n{sym}:")
    end = asm.index(".end_amdhsa_kernel", start)
    body = asm[start:end]
    return body[: body.rindex("s_endpgm")]


def _inner_loops(body):
    """[(header label, text)] of the Depth=1 loops of a function.  The assembly printer annotates every basic block (a `.LBBn_m:` label or a
    `; %bb.m:` comment, the annotation on that line or the next one) with `in Loop: Header=BBn_m Depth=1` / `Parent Loop BBn_m Depth=1`;
    a loop's text = its header block plus every block carrying its tag."""
    blocks, cur = [], []
    for line in body.splitlines():
        if re.match(r"(\.LBB\d+_\d+:|; %bb\.\d+:)", line) and cur:
            blocks.append(cur)
            cur = []
        cur.append(line)
    blocks.append(cur)
    loops = []
    for i, blk in enumerate(blocks):
        head = "\n".join(blk[:3])
        m = re.match(r"\.L(BB\d+_\d+):", blk[0])
        if not m or not re.search(r"=>This (?:Inner )?Loop Header: Depth=1", head):
            continue
        hdr = m.group(1)
        member = [blk] + [b for b in blocks[i + 1:] if re.search(rf"(Header={hdr} Depth=1|Parent Loop {hdr} Depth=1)", "\n".join(b[:3]))]
        loops.append((hdr, "\n".join("\n".join(b) for b in member)))
    return loops


def test_hot_rollout_kernels_keep_their_resource_budget(build):
    remarks, asm = build
    res = _resources(remarks)
    assert len(res) >= 100
    for env, (e, occ, spill_bound) in HOT.items():
        r = res[_symbol(env, e)]
        assert r["Occupancy"] == occ, (env, r)
        assert r["VGPRs"] <= 128, (env, r)
        assert r["VGPRs Spill"] <= spill_bound, (env, r)          # today's figures; a larger spill is one step from scratch traffic in the loop


def test_the_instantiations_with_fused_observation_moments_keep_four_waves(build):
    """rollout_kernel_v3<..., STATS = true> (mxv_set_obs_partials): the column sums of a step's observations and the wave-level tree on
    top of the trajectory kernel — still inside the 128-VGPR budget, nothing spilled, for every env kind and both dtype sets."""
    remarks, asm = build
    res = _resources(remarks)
    stats_e = {0: 2, 1: 2, 2: 1, 3: 2, 4: 2}        # stats_envs_per_lane: Pendulum takes two envs per lane here (the tree's cost per env halves)
    spill = {(0, 3): 8, (1, 2): 8, (1, 3): 10, (2, 1): 2, (2, 2): 2, (2, 3): 4}      # parked around the K-step loop, not inside it (checked below for CartPole; Acrobot's two
                                                               # arrived with the 16-bit elapsed[] branch at entry / exit, round 6: same 21 scratch ops in its loop as before)
    for env, (_, occ, _) in HOT.items():
        e = stats_e[env]
        for out, st in ((1, 1), (2, 1), (1, 2), (2, 2), (1, 3), (2, 3)):      # st: 1 observation moments, 2 discounted returns, 3 both
            r = res[f"_ZN3mxv12_GLOBAL__N_117rollout_kernel_v3ILi{env}ELb1ELi{e}ELb0ELi{out}ELb0ELi{st}EEEvNS_8StepArgsE"]
            assert r["Occupancy"] == occ and r["VGPRs"] <= 128 and r["VGPRs Spill"] <= spill.get((env, st), 0), (env, out, st, r)
    body = _function_body(asm, "_ZN3mxv12_GLOBAL__N_117rollout_kernel_v3ILi0ELb1ELi2ELb0ELi1ELb0ELi1EEEvNS_8StepArgsE")
    loops = [(h, t) for h, t in _inner_loops(body) if "global_store" in t]
    text = max(loops, key=lambda ht: ht[1].count("global_store"))[1]
    assert text.count("v_permlane16_swap") == 2 and text.count("v_permlane32_swap") == 2 and text.count("ds_swizzle") == 2   # ONE value left at these stages
    assert 10 <= sum(1 for l in text.splitlines() if "_dpp" in l) <= 20                                                    # 8 -> 4 -> 2 -> 1 values: 14 moves


# Round 3: with scalar-base stores (pin32, mxv_kernels.hip) no kernel spills a vector register any more; round 2's CartPole kernel parked
# 24 VGPRs in scratch around the loop and MountainCarContinuous reloaded a spilled store address inside it.
SCRATCH_IN_LOOP = {0: 0, 1: 0, 2: 0, 3: 0, 4: 0}
STORE_BLOCKS_WAITING = {0: 0, 1: 0, 2: 0, 3: 0, 4: 0}
SADDR_STORES_AT_LEAST = {0: 10, 1: 5, 2: 5, 3: 6, 4: 6}     # per-step stores that take their base from SGPRs (`, s[a:b]` / vcc)


def test_no_scratch_and_no_unconditional_vmcnt_wait_inside_the_step_loops(build):
    _, asm = build
    for env, (e, _, _) in HOT.items():
        body = _function_body(asm, _symbol(env, e))
        loops = [(h, t) for h, t in _inner_loops(body) if "global_store" in t]
        assert loops, f"env {env}: no store loop found"
        hdr, text = max(loops, key=lambda ht: ht[1].count("global_store"))      # the K-step loop: the one that stores every output
        assert text.count("global_store") >= 4 * e, (env, hdr)
        blocks = re.split(r"\n(?=\.LBB\d+_\d+:|; %bb\.\d+:)", text)
        # Acrobot's loop contains the cold exact path (mxv_exact.hpp, entered when the height is within 2^-40 of the threshold: ~1 env-step
        # in 10^12): the blocks that call cr_sincos (s_swappc) save and reload registers through scratch.  Every other block is the step.
        cold = [b for b in blocks if "s_swappc_b64" in b]
        assert (len(cold) >= 1) == (env == 2), f"env {env}: {len(cold)} blocks with calls in the K-step loop"
        assert all("global_store" not in b for b in cold)
        text = "\n".join(b for b in blocks if "s_swappc_b64" not in b)
        assert text.count("scratch_") <= SCRATCH_IN_LOOP[env], f"env {env}: scratch access inside the K-step loop ({hdr})"
        saddr = [l for l in text.splitlines() if "global_store" in l and re.search(r", (s\[\d+:\d+\]|vcc)\s*(offset:\S+)?\s*$", l.split(";")[0].rstrip())]
        assert len(saddr) >= SADDR_STORES_AT_LEAST[env], f"env {env}: only {len(saddr)} stores with a scalar base: the 32-bit lane offsets lost their pin"
        # a vmcnt(0) wait may sit in a conditional block that issued a load itself (per-env seeds of explicit seed lists); the blocks
        # every step runs through — the ones with the stores — must not wait for the stores in flight
        blocks = [b for b in blocks if "s_swappc_b64" not in b]
        waiting = [b.splitlines()[0] for b in blocks if "global_store" in b and "global_load" not in b and re.search(r"s_waitcnt[^\n]*vmcnt\(0\)", b)]
        assert len(waiting) <= STORE_BLOCKS_WAITING[env], f"env {env}: store blocks of the K-step loop wait for vmcnt(0): {waiting}"


def test_the_instruction_forms_this_round_bought_are_still_in_the_loops(build):
    """Round 3 removed instructions the compiler emits on its own: the `v_mov_b64 + v_fmac_f64` pair per polynomial term
    that loop-invariant code motion leaves behind (three-address `v_fma_f64` instead: Pendulum, MountainCarContinuous), compare + select
    clamps (`v_max` / `v_min` instead), the branchy NumPy `%` of Pendulum's bounded path.  A later edit or compiler that loses them would
    only show as a slower variants line; here it fails.  Static counts of the K-step loop, with generous margins."""
    _, asm = build

    def loop_text(env):
        e = HOT[env][0]
        loops = [(h, t) for h, t in _inner_loops(_function_body(asm, _symbol(env, e))) if "global_store" in t]
        return max(loops, key=lambda ht: ht[1].count("global_store"))[1]

    def count(text, op):
        return sum(1 for l in text.splitlines() if l.strip().startswith(op))

    pend, mcc, mc = loop_text(1), loop_text(4), loop_text(3)
    # three-address polynomial steps: 10 per sincos evaluated in the loop body (hot path + reset path), hardly any 64-bit copies left
    assert count(pend, "v_fma_f64") >= 20 and count(pend, "v_mov_b64") <= 10, (count(pend, "v_fma_f64"), count(pend, "v_mov_b64"))
    assert count(mcc, "v_fma_f64") >= 20 and count(mcc, "v_mov_b64") <= 10, (count(mcc, "v_fma_f64"), count(mcc, "v_mov_b64"))
    # clamps: v_max / v_min pairs, and since round 4 a NaN passes through them as it does through the reference's (np.clip, `if x > hi`):
    # float64 = one v_cmp_u_f64 + one v_cndmask on the high dword per pair, float32 = the NaN-propagating v_maximum3 / v_minimum3
    assert count(pend, "v_max_f64") + count(pend, "v_min_f64") >= 2 and count(pend, "v_cmp_u_f64") >= 1
    assert count(pend, "v_maximum3_f32") >= 1 and count(pend, "v_minimum3_f32") >= 1
    assert count(mc, "v_max_f64") + count(mc, "v_min_f64") >= 4 * HOT[3][0] - 2 and count(mc, "v_cmp_u_f64") >= 2 * HOT[3][0]
    assert count(mcc, "v_maximum3_f32") >= 2 and count(mcc, "v_minimum3_f32") >= 2
    acro = loop_text(2)
    assert count(acro, "v_bitop3_b32") >= 12          # quadrant signs of the eight hot sincos (round 4), one instruction each
    # Pendulum's K-step loop: 614 instructions / 27 branches before, 501 / 20 after
    n_instr = sum(1 for l in pend.splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";")))
    assert n_instr <= 540 and sum(1 for l in pend.splitlines() if "s_cbranch" in l) <= 22, n_instr


# ---- the other engines' trajectory kernels (seconds to compile) -------------------------------------------------------------------------
def _compile(src_name):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    d = tempfile.mkdtemp(prefix="mxv_kres2_")
    try:
        p = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-c",
                            os.path.join(ROOT, "gym_amd", "csrc", src_name), "-o", os.path.join(d, "k.o"),
                            "-Rpass-analysis=kernel-resource-usage", "-save-temps"], cwd=d, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")]
        return p.stderr, open(os.path.join(d, asm[0])).read()
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_tabular_trajectory_kernel_stays_lean():
    """tab_traj_kernel (gym_amd/csrc/mxv_tab.hip): every instantiation at 8 waves per SIMD without scratch; the step loop of the two
    shapes the bench measures holds exactly the six per-step stores of four unrolled steps, one LDS read per table level, no fp64
    conversion of the Philox words (the integer thresholds), and the quad transpose in DPP moves instead of ds_bpermute."""
    remarks, asm = _compile("mxv_tab.hip")
    res = _resources(remarks)
    traj = {k: v for k, v in res.items() if "tab_traj_kernel" in k}
    assert len(traj) == 16
    for k, r in traj.items():
        assert r["Occupancy"] == 8 and r["VGPRs"] <= 64 and r["ScratchSize"] == 0, (k, r)
    for sym, lds_reads in (("_ZN12_GLOBAL__N_115tab_traj_kernelILi1ELb1ELb0ELb1EEEvNS_11TabTrajArgsE", 4),      # Taxi: M = 1, search in a cold block
                           ("_ZN12_GLOBAL__N_115tab_traj_kernelILi3ELb1ELb1ELb1EEEvNS_11TabTrajArgsE", 8)):     # FrozenLake8x8: M = 3, point-mass start
        body = _function_body(asm, sym)
        assert body.count("ds_bpermute") == 0 and body.count("v_cvt_f64_u32") == 0, sym
        assert sum(1 for l in body.splitlines() if "_dpp" in l and "quad_perm" in l) >= 4, sym
        loops = [t for _, t in _inner_loops(body) if "global_store" in t]
        main = max(loops, key=lambda t: t.count("global_store"))
        # 6 outputs x 4 unrolled steps, + the two episode-statistics stores per step behind their wave-uniform branch (round 6: skipped when
        # the accumulators are off, taken only by waves in which an episode ended when they are on)
        assert main.count("global_store") == 32, (sym, main.count("global_store"))
        hot_reads = sum(1 for l in main.splitlines() if re.match(r"\s+ds_read", l))
        assert hot_reads >= lds_reads, (sym, hot_reads)


def test_normalisation_kernels_keep_their_pipelines():
    """mxv_norm.hip: the vector path of returns_sums_kernel waits for its ring with vmcnt(>= 9) — a derived value in the ring, or an exit
    between the unrolled steps, shows up here as vmcnt(0..4) (round 4) — and reduces through DPP / permlane, not ds_bpermute."""
    remarks, asm = _compile("mxv_norm.hip")
    for sym, floor in (("_ZN12_GLOBAL__N_119returns_sums_kernelIdEEvPKT_PKhS5_PdlidlS6_i", 12), ("_ZN12_GLOBAL__N_119returns_sums_kernelIfEEvPKT_PKhS5_PdlidlS6_i", 9)):
        body = _function_body(asm, sym)
        assert body.count("ds_bpermute") == 0 and body.count("v_permlane32_swap") >= 4, sym
        vec = [t for _, t in _inner_loops(body) if "global_load_dwordx4" in t and "global_load_ubyte" not in t]
        assert len(vec) == 1, (sym, len(vec))
        waits = [int(m) for m in re.findall(r"s_waitcnt vmcnt\((\d+)\)", vec[0])]
        assert waits and min(waits) >= floor, (sym, waits)


def test_blackjack_kernel_is_one_philox_call_of_straight_line_code_per_step():
    """bj_kernel (gym_amd/csrc/mxv_bj.hip, round 5): rounds 2-4 issued ~300 VALU instructions per table-step — two to three draw-stream
    Philox calls and a divergent dealer loop — and ran at 9-10 us per 2^20-table step, VALU-issue bound.  The
    round-5 draw contract (one call per step, eight cards with fixed roles) makes a step straight-line code: the K-step loop of the sampled
    instantiations holds exactly ONE draw call beside the action-bit refill (2 x 20 multiplies + the eight card digits), at most 330 VALU
    instructions statically — the rare late-draw loop and the once-per-32-steps refill included; ~175 on the path a step takes — no
    scratch, 8 waves per SIMD (two full rounds of the 16 waves per SIMD of a 2^20-table launch), and every per-step store in the
    scalar-base + 32-bit-lane-offset form (no 64-bit address arithmetic in vector registers)."""
    remarks, asm = _compile("mxv_bj.hip")
    res = {k: v for k, v in _resources(remarks).items() if "bj_kernel" in k}
    assert len(res) == 10, list(res)
    for k, r in res.items():
        if k.endswith("ELb1EEEvNS_6BjArgsE"):     # STATS = true (round 6: RecordEpisodeStatistics fused): an instantiation of its own, 7 waves allowed
            assert r["ScratchSize"] == 0 and r["Occupancy"] >= 7 and r["VGPRs"] <= 72, (k, r)
        else:                                     # the launches without statistics keep round 5's budget exactly
            assert r["ScratchSize"] == 0 and r["Occupancy"] == 8 and r["VGPRs"] <= 64, (k, r)
    for sym in ("_ZN12_GLOBAL__N_19bj_kernelILb0ELb1ELi1ELb0EEEvNS_6BjArgsE", "_ZN12_GLOBAL__N_19bj_kernelILb0ELb1ELi2ELb0EEEvNS_6BjArgsE"):
        body = _function_body(asm, sym)
        loops = [t for _, t in _inner_loops(body) if "global_store" in t]
        main = max(loops, key=lambda t: t.count("global_store"))
        valu = sum(1 for l in main.splitlines() if re.match(r"\s+v_", l))
        mads = sum(1 for l in main.splitlines() if re.match(r"\s+v_mad_u64_u32", l))
        assert valu <= 330, (sym, valu)
        stores = [l for l in main.splitlines() if re.match(r"\s+global_store_", l)]
        assert len(stores) >= 7 and all(re.search(r", s\[\d+:\d+\]", l) for l in stores), (sym, stores)
        assert 40 <= mads <= 50, (sym, mads)          # action refill (20, once per 32 steps) + the step's draw call (20) + card digits
        assert main.count("scratch_") == 0 and main.count("ds_bpermute") == 0, sym


def test_acrobot_building_blocks_keep_their_instruction_budget(tmp_path):
    """Acrobot's 583 VALU instructions per env-step are 8 sincos_medium + 4 dsdt + glue (profiles/r6/r6h_acrobot_valu.md).  The two
    functions compiled in isolation with the library's flags: dsdt holds exactly TWO v_rcp_f64 (the three quotients by d1 share one
    refined reciprocal, the fourth divisor has its own — 8 quarter-rate instructions per env-step, as the counters say) and no
    v_div_scale / v_div_fmas (the compiler's own `/`), sincos_medium at most 54 VALU instructions, no scratch."""
    src = tmp_path / "blocks.hip"
    src.write_text('#include "mxv_device.hpp"\nusing namespace mxv;\n'
                   '__global__ void k_sincos(const double *x, double *o) { double s, c; mx_sincos<false, fma3_for<MXV_ACROBOT>()>(x[threadIdx.x], &s, &c); '
                   'o[threadIdx.x] = s; o[threadIdx.x + 64] = c; }\n'
                   '__global__ void k_dsdt(const double *x, double *o) { EnvParams PP{}; const Par<PM_DEFAULT> P(PP); double sa[4] = {x[0], x[1], x[2], x[3]}, '
                   'sc[4] = {x[4], x[5], x[6], x[7]}, out[4]; Env<MXV_ACROBOT>::dsdt<PM_DEFAULT>(P, sa, sc, x[8], out); o[0] = out[2]; o[1] = out[3]; }\n')
    out = tmp_path / "blocks.s"
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-I",
                        os.path.join(ROOT, "gym_amd", "csrc"), "-S", "--cuda-device-only", str(src), "-o", str(out)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    asm = out.read_text()

    def valu(sym):
        i = asm.index(f"\n{sym}:")
        return [l.split()[0] for l in asm[i:asm.index("s_endpgm", i)].splitlines() if re.match(r"\s+v_", l)]

    d, s = valu("_Z6k_dsdtPKdPd"), valu("_Z8k_sincosPKdPd")
    assert d.count("v_rcp_f64_e32") == 2 and not any(i.startswith(("v_div_scale", "v_div_fmas")) for i in d) and len(d) <= 64, (len(d), d.count("v_rcp_f64_e32"))
    assert len(s) <= 54 and "scratch_" not in asm, len(s)
