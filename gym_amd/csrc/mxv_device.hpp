// mxv_device.hpp — device-side building blocks of the classic-control step kernels (gfx950).
//
// One lane owns E environments; everything here is per-environment scalar fp64 code that
// the kernels in mxv_kernels.hip instantiate E times per lane (independent chains = ILP).
// Arithmetic follows the reference operation by operation (file:line cited per function;
// paths relative to the reference root, openai/gym 0.26.2) and the translation unit is built
// with -ffp-contract=off so no a*b+c is fused: the reference rounds every operation.
// Python's x**2 is written x*x (correctly rounded; libm pow(x,2) is within 1 ulp of it).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mxv.h"
#include "../../include/mxv_diag.h"

#define MXV_XFN __device__ inline
#define MXV_XCONST __device__ const
#define MXV_XCOLD __device__ __noinline__ inline
#include "mxv_exact.hpp"

#ifndef MXV_CARTPOLE_RCP
#define MXV_CARTPOLE_RCP 1  // tuning hook: 0 = the compiler's `/` for CartPole's one runtime division
#endif

namespace mxv {

struct EnvParams {
    double p[MXV_MAX_PARAMS];
};

// Parameter access policy (template argument DEF of everything below):
//   PM_DEFAULT   (1, `true`)  folds the reference's default attribute values into the instruction stream;
//   PM_BROADCAST (0, `false`) reads the values set through mxv_set_params() (VectorEnv.set_attr with one value for
//                             all sub-envs) from the kernel argument segment (SGPRs);
//   PM_PER_ENV   (2)          reads this env's own values from a [MXV_MAX_PARAMS][N] device array
//                             (mxv_set_params_per_env: set_attr with a list of differing values,
//                             gym/vector/sync_vector_env.py:192-214).
enum { PM_BROADCAST = 0, PM_DEFAULT = 1, PM_PER_ENV = 2 };
template <int DEF>
struct Par {
    const EnvParams &P;
    const double *pe;  // per-env table or nullptr
    int64_t n, e;
    __device__ __forceinline__ explicit Par(const EnvParams &p, const double *per_env = nullptr, int64_t n_ = 0, int64_t e_ = 0)
        : P(p), pe(per_env), n(n_), e(e_) {}
    __device__ __forceinline__ Par at(int64_t env) const { return Par(P, pe, n, env); }  // bind to one env (PM_PER_ENV)
    __device__ __forceinline__ double get(int i, double dflt) const {
        if constexpr (DEF == PM_DEFAULT) return dflt;
        else if constexpr (DEF == PM_BROADCAST) return P.p[i];
        else return pe[(int64_t)i * n + e];
    }
};

constexpr double kPi = 3.141592653589793;

// s_waitcnt vmcnt(0) (expcnt / lgkmcnt left open), placed between a kernel's entry loads and its K-step loop.  On gfx9-class
// ISAs vmcnt counts stores as well as loads: without this the compiler parks the waits for the ENTRY loads at their first
// use inside the loop, where from the second iteration on they wait for the acknowledgement of the stores of the previous
// (and the current) step — the whole HBM write latency, once per wave and step.  With the loads settled at entry the loop
// body carries no vmcnt wait at all and a wave keeps several steps of output stores in flight.
#ifndef MXV_SETTLE_LOADS
#define MXV_SETTLE_LOADS 1  // tuning hook: 0 = leave the waits where the compiler puts them
#endif
__device__ __forceinline__ void settle_entry_loads() {
#if MXV_SETTLE_LOADS
    __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
}
// The same for a load INSIDE the loop on a conditional path (per-env seeds of a reset draw): using the value in the branch
// that loaded it keeps the compiler's vmcnt wait inside that branch instead of at the join every wave passes through.
template <typename T>
__device__ __forceinline__ T landed(T v) {
#if MXV_SETTLE_LOADS
    asm volatile("" : "+v"(v));
#endif
    return v;
}

// ------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  Counter-based: no RNG state is kept in memory.
// ------------------------------------------------------------------------------------------
struct U4 {
    uint32_t x, y, z, w;
};

constexpr int kPhiloxRounds = 10;  // Philox4x32-10
__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < kPhiloxRounds; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;  // one v_mad_u64_u32 yields hi and lo
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        U4 n;
        n.x = hi1 ^ c.y ^ k0;
        n.y = lo1;
        n.z = hi0 ^ c.w ^ k1;
        n.w = lo0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// Same function for a per-lane (VGPR) key: the two three-input XORs of a round are one v_bitop3_b32 each (truth table 0x96 =
// a ^ b ^ c; gfx950's compiler emits two v_xor_b32 for them), 20 VALU instructions fewer per call.  Bit-identical by
// construction.  Only for callers whose key lives in VGPRs (the inline-asm operands are VGPR-constrained; a uniform key would
// first be copied into one).
__device__ __forceinline__ uint32_t xor3_vgpr(uint32_t x, uint32_t y, uint32_t z) {
    uint32_t d;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(d) : "v"(x), "v"(y), "v"(z));
    return d;
}
// x ^ (y & mask) in one instruction (mask: a wave-uniform constant, taken from an SGPR)
#ifndef MXV_BITOP3_SIGNS
#define MXV_BITOP3_SIGNS 1   // A/B hook (which env kinds use the form: MXV_BITOP3_ENVS below)
#endif
__device__ __forceinline__ uint32_t xor_masked(uint32_t x, uint32_t y, uint32_t mask) {
#if MXV_BITOP3_SIGNS
    uint32_t d;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x78" : "=v"(d) : "v"(x), "v"(y), "s"(mask));
    return d;
#else
    return x ^ (y & mask);
#endif
}
__device__ __forceinline__ U4 philox4x32_10_vkey(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < kPhiloxRounds; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        U4 n;
        n.x = xor3_vgpr((uint32_t)(p1 >> 32), c.y, k0);
        n.y = (uint32_t)p1;
        n.z = xor3_vgpr((uint32_t)(p0 >> 32), c.w, k1);
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

constexpr uint32_t kStreamAction = 1u;
constexpr uint32_t kStreamReset = 2u;
constexpr uint32_t kStreamStepNoise = 4u;  // (3 = the tabular engine's transition stream, 5 = Blackjack's draw stream)
constexpr uint32_t kStreamActionBits = 6u; // Discrete(2) action stream of the classic engine: one bit per step

// u in (0,1): (w + 0.5) * 2^-32, exact in fp64.
__device__ __forceinline__ double u01(uint32_t w) { return ((double)w + 0.5) * (1.0 / 4294967296.0); }

// Word-per-step action stream: words of the 4 envs of group g (global env indices 4g..4g+3) at vector step t.  Used as is by
// the tabular and Blackjack engines (mxv_tab.hip, mxv_bj.hip); the classic-control kernels go through env_action_words<ENV>
// below, which is this stream for Discrete(3) / Box and the bit-sliced stream for Discrete(2).
__device__ __forceinline__ U4 action_words(uint64_t action_seed, uint64_t t, uint64_t g) {
    U4 c;
    c.x = (uint32_t)g;
    c.y = (uint32_t)(g >> 32);
    c.z = (uint32_t)t;
    c.w = ((uint32_t)(t >> 32) & 0x0fffffffu) | (kStreamAction << 28);
    return philox4x32_10(c, (uint32_t)action_seed, (uint32_t)(action_seed >> 32));
}

// The four action words of a quad change lanes (tabular and Blackjack engines: lane q of a quad evaluates step q of an aligned block of
// four steps for the quad's four envs; afterwards lane q holds its own env's words for steps 0..3): a 4 x 4 transpose in two DPP
// butterfly stages.
template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}

// b[j] at lane L of a quad = a[L] at lane j of the quad
__device__ __forceinline__ void quad_transpose(uint32_t (&a)[4], uint32_t q) {
    const bool odd = (q & 1u) != 0, hi = (q & 2u) != 0;
#pragma unroll
    for (int p = 0; p < 4; p += 2) {  // lanes L and L ^ 1 trade a[p + 1] of the even lane for a[p] of the odd one
        const uint32_t recv = quad_perm<0xB1>(odd ? a[p] : a[p + 1]);
        a[p] = odd ? recv : a[p];
        a[p + 1] = odd ? a[p + 1] : recv;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {     // lanes L and L ^ 2: a[p + 2] of the low pair for a[p] of the high pair
        const uint32_t recv = quad_perm<0x4E>(hi ? a[p] : a[p + 2]);
        a[p] = hi ? recv : a[p];
        a[p + 2] = hi ? a[p + 2] : recv;
    }
}

// Step-indexed reset words (explicit resets of the tabular and Blackjack engines: step t, ordinal r of the reset call).
__device__ __forceinline__ U4 reset_words(uint64_t seed, uint64_t t, uint32_t r) {
    U4 c;
    c.x = (uint32_t)t;
    c.y = (uint32_t)(t >> 32);
    c.z = r;
    c.w = (kStreamReset << 28);
    return philox4x32_10_vkey(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// Reset stream of the classic-control engine: the k-th reset of an env since its seeding (explicit reset() or the autoreset
// inside a vector step; k = 0, 1, 2, ...) draws from ctr = (k, 0, 0, 2 << 28) under the env's own 64-bit seed — the order in
// which ONE env's generator is consumed in the reference (each sub-env owns a generator that advances once per reset of that
// env, gym/envs/classic_control/cartpole.py:202), not a function of the global step index.  Because the draw of an env's NEXT
// reset is known as soon as the current episode starts, the fused rollout computes it off the critical path, for many envs
// per Philox call (rollout_kernel_v3).
__device__ __forceinline__ U4 episode_reset_words(uint64_t seed, uint32_t k) {
    U4 c;
    c.x = k;
    c.y = 0u;
    c.z = 0u;
    c.w = (kStreamReset << 28);
    return philox4x32_10_vkey(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// Step-noise stream (Acrobot's torque noise, acrobot.py:202-205): key = the env's seed, ctr = (t_lo, t_hi, 0, 4 << 28), word x.
__device__ __forceinline__ uint32_t step_noise_word(uint64_t seed, uint64_t t) {
    U4 c;
    c.x = (uint32_t)t;
    c.y = (uint32_t)(t >> 32);
    c.z = 0;
    c.w = (kStreamStepNoise << 28);
    return philox4x32_10_vkey(c, (uint32_t)seed, (uint32_t)(seed >> 32)).x;
}

// ------------------------------------------------------------------------------------------
// Arithmetic helpers that keep IEEE results while shedding VALU work.
// ------------------------------------------------------------------------------------------
// x / c for a compile-time constant c, correctly rounded (== the IEEE quotient the reference computes) in three
// fp64 instructions instead of the ~13-instruction v_div_scale/v_rcp/v_div_fmas/v_div_fixup sequence:
// with rc = RN(1/c), q0 = RN(x*rc), r = x - c*q0 (exact in an FMA), q1 = RN(q0 + r*rc) is the correctly rounded
// quotient whenever c's significand is not all ones (Markstein 1990; Cornea, Harrison, Tang 2002) and no
// intermediate is subnormal (|x/c| > 2^-969: always true for these dynamics).  Checked against `/` on 3e8 random
// operands for the constants used here (0 mismatches).  Only used on the default-parameter path.
__device__ __forceinline__ double div_by_const(double x, double c, double rc) {
    const double q0 = x * rc;
    const double r = __fma_rn(-c, q0, x);
    return __fma_rn(r, rc, q0);
}
// Several quotients by the SAME runtime divisor d (Acrobot divides three times by d1 per RK4 stage): the compiler's IEEE
// fp64 division is v_div_scale x2, v_rcp_f64, two Newton steps on the reciprocal, q0 = x*r, rem = fma(-d, q0, x),
// v_div_fmas (= fma(rem, r, q0)), v_div_fixup.  For operands that need no scaling or fix-up (normal, exponents far from
// the limits, x != -0: true for the default-parameter dynamics) the scale factors are 1 and the fix-up is the identity,
// so running the reciprocal part once and the 3-instruction tail per dividend reproduces `/` bit for bit
// (tools/divcheck.hip: 0 mismatches against `/` on 4e9 random operand pairs on the MI355X).
__device__ __forceinline__ double refined_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __fma_rn(-d, r, 1.0);
    r = __fma_rn(r, e, r);
    e = __fma_rn(-d, r, 1.0);
    return __fma_rn(r, e, r);
}
__device__ __forceinline__ double div_with_rcp(double x, double d, double r) {
    const double q0 = x * r;
    return __fma_rn(__fma_rn(-d, q0, x), r, q0);
}
// FINITE = the caller vouches for a finite dividend (the fused rollout on states the dynamics produced).  The Markstein sequence turns
// an infinite dividend into a NaN (Inf * rc - c * Inf) where IEEE division keeps it infinite, and an injected state may hold one
// (tests/golden/*_p1_nonfinite.npz: the reference's own outputs on such states): the guarded instantiations (step_kernel, rollouts after a
// state injection) append the hardware's own special-case pass, v_div_fixup_f64 — what the compiler's `/` ends with: the quotient goes
// through unchanged for ordinary operands, NaN / Inf / 0 operands get IEEE's answers — one instruction instead of the ten of a full division
// (round 4 first used `/` here: step(actions) 18.5 -> 20.1 us per 2^20-env step).
#ifndef MXV_GUARDED_DIV_FIXUP
#define MXV_GUARDED_DIV_FIXUP 1   // A/B hook
#endif
template <int DEF, bool FINITE = true>
__device__ __forceinline__ double div_par(double x, double c) {
    if constexpr (DEF == PM_DEFAULT && FINITE)
        return div_by_const(x, c, 1.0 / c);  // c is a literal on this path: 1.0 / c folds at compile time
    else if constexpr (DEF == PM_DEFAULT)
#if MXV_GUARDED_DIV_FIXUP
        return __builtin_amdgcn_div_fixup(div_by_const(x, c, 1.0 / c), c, x);
#else
        return div_by_const(x, c, 1.0 / c);
#endif
    else
        return x / c;
}

// sin and cos of |x| <= pi/4 without argument reduction: the fdlibm/msun kernel polynomials (Sun Microsystems
// 1993; error < 1 ulp, the same bound glibc documents for its own sin/cos), evaluated in Horner form with explicit
// FMAs.  The no-contraction rule of this file protects the REFERENCE's arithmetic (every Python operator rounds
// once); what happens inside a libm call is the library's business, and FMA Horner steps are both more accurate and
// half the instructions of separate multiply/add.  CartPole's pole angle never leaves (-0.42, 0.42) on an
// autoresetting trajectory, so its sin/cos need no reduction at all; callers fall back to the general sincos()
// outside the interval.
// a * b + k for a coefficient k that lives on (a Horner step inside a loop): the compiler forms `v_fmac_f64 acc, a, b` with the accumulator
// preloaded — fine while k is a literal it has to materialise anyway — and loop-invariant code motion then hoists the materialisation,
// leaving `v_mov_b64 acc, k_regs ; v_fmac_f64 acc, a, b`: one full-rate copy per polynomial term, 11 per sincos (88 of Acrobot's 730 VALU
// instructions per env-step).  The three-address form reads the coefficient where it is.  Same operation, same bits.
// It pays where the coefficients stay in registers across the loop (Pendulum, MountainCarContinuous: -7 % / -9 % instructions in the
// K-step loop); the kernels at their 128-VGPR budget (CartPole and MountainCar at two envs per lane, Acrobot) re-materialise or spill
// instead, so the form is chosen per env kind: bit `env id` of MXV_FMA3_ENVS (A/B hook; profiles/r3/r3p_fma3_ab.jsonl).
#ifndef MXV_FMA3_ENVS
#define MXV_FMA3_ENVS ((1 << MXV_PENDULUM) | (1 << MXV_MOUNTAINCAR_CONT))
#endif
// (The same with the coefficients in SGPR pairs — 20 VGPRs freed for 20 SGPRs — was tried for the register-bound kernels: Acrobot -1 % of its
// loop's VALU instructions, every other kind more: dropped.)
// CartPole with ONE env per lane (shards of up to 2^17 envs: the per-GPU share of an 8-GPU strong-scaling job, latency-bound, 87 VGPRs)
// gains 5 % (0.818 -> 0.778 us per step); at two envs per lane (113 VGPRs) it is the compiler's form that wins.
#ifndef MXV_FMA3_CARTPOLE_E1
#define MXV_FMA3_CARTPOLE_E1 1
#endif
// ... and bit 1 of the same per-kind flag word: the quadrant signs of sincos_medium as v_bitop3_b32 (xor_masked) — Acrobot and Pendulum
// lose 2 instructions per sincos, MountainCar (cosine only, two envs per lane) gains 15 in its loop: chosen per env kind as well.
#ifndef MXV_BITOP3_ENVS
#define MXV_BITOP3_ENVS ((1 << MXV_ACROBOT) | (1 << MXV_PENDULUM))
#endif
// ... bit 2 (round 6): the twelve polynomial coefficients of sincos_kernel read from a table in constant memory by SCALAR loads inside every
// call and used as the FMAs' scalar operand (v_fma_f64 v, v, v, s[..]).  A kernel at its VGPR cap re-materialises the literals next to
// every Horner step — 20 of sincos_medium's 52 VALU instructions are v_mov_b32 of coefficient halves, 8 sincos per Acrobot step —
// and keeping them in SGPRs ACROSS the loop spills (round 3's attempt: -1 %).  Loaded where they are used (s_load_dwordx8 x 3 through a
// pointer the optimiser cannot see through, so the loads are not hoisted) they cost the scalar unit three instructions and the
// vector unit none: sincos_medium alone goes from 52 to 36 VALU instructions.  MEASURED in the kernel that matters (round 6, Acrobot,
// 2^19 envs): 583.4 -> 579.2 VALU instructions per env-step and the same 10.5 us — inside the K-step loop the compiler already keeps
// the coefficients in ~24 VGPRs across the steps, so the 20 moves were never executed per call; the form only frees those registers
// (128 -> 126 VGPRs, 4 -> 0 spilled).  An A/B hook, off (profiles/r6/r6h_acrobot_valu.md).
#ifndef MXV_SCOEF_ENVS
#define MXV_SCOEF_ENVS 0
#endif
template <int ENV>
constexpr int fma3_for() {
    return (((MXV_FMA3_ENVS >> ENV) & 1) ? 1 : 0) | (((MXV_BITOP3_ENVS >> ENV) & 1) ? 2 : 0) | (((MXV_SCOEF_ENVS >> ENV) & 1) ? 4 : 0);
}
__constant__ const double kSinCosCoef[12] = {
    1.58969099521155010221e-10, -2.50507602534068634195e-08, 2.75573137070700676789e-06, -1.98412698298579493134e-04,
    8.33333333332248946124e-03, -1.66666666666666324348e-01,                                                        // S6 .. S1
    -1.13596475577881948265e-11, 2.08757232129817482790e-09, -2.75573143513906633035e-07, 2.48015872894767294178e-05,
    -1.38888888888741095749e-03, 4.16666666666666019037e-02};                                                       // C6 .. C1
typedef const double __attribute__((address_space(4))) cdouble_t;
__device__ __forceinline__ double fma_scoef(double a, double b, double k_sgpr) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k_sgpr));
    return r;
}

template <int F3>
__device__ __forceinline__ double fma_coef(double a, double b, double k) {
    if constexpr ((F3 & 1) != 0) {
        double r;
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(k));
        return r;
    } else {
        return __fma_rn(a, b, k);
    }
}

#ifndef MXV_FDLIBM_COS
#define MXV_FDLIBM_COS 0
#endif
template <int F3 = 0>
__device__ __forceinline__ void sincos_kernel(double x, double *sn, double *cs) {
    const double z = x * x;
    if constexpr ((F3 & 4) != 0) {      // the same FMAs in the same order with ten of the twelve coefficients as scalar operands (MXV_SCOEF_ENVS;
        cdouble_t *t = (cdouble_t *)kSinCosCoef;   // an instruction reads ONE scalar operand: the leading coefficient of each polynomial stays a literal)
        asm volatile("" : "+s"(t));
        double r = fma_scoef(z, 1.58969099521155010221e-10, t[1]);
        r = fma_scoef(z, r, t[2]);
        r = fma_scoef(z, r, t[3]);
        r = fma_scoef(z, r, t[4]);
        r = fma_scoef(z, r, t[5]);
        *sn = __fma_rn(x * z, r, x);
        double c = fma_scoef(z, -1.13596475577881948265e-11, t[7]);
        c = fma_scoef(z, c, t[8]);
        c = fma_scoef(z, c, t[9]);
        c = fma_scoef(z, c, t[10]);
        c = fma_scoef(z, c, t[11]);
        c = __fma_rn(z, c, -0.5);
        *cs = __fma_rn(z, c, 1.0);
        return;
    }
    // sin: x + x*z*(S1 + z*(S2 + z*(S3 + z*(S4 + z*(S5 + z*S6)))))
    double r = fma_coef<F3>(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    r = fma_coef<F3>(z, r, 2.75573137070700676789e-06);
    r = fma_coef<F3>(z, r, -1.98412698298579493134e-04);
    r = fma_coef<F3>(z, r, 8.33333333332248946124e-03);
    r = fma_coef<F3>(z, r, -1.66666666666666324348e-01);
    *sn = __fma_rn(x * z, r, x);
    // cos: 1 + z*(-1/2 + z*(C1 + z*(C2 + ... z*C6))) in plain Horner form.  fdlibm sums 1 - z/2 + z*z*(...) so that the rounding error of
    // 1 - z/2 is recovered (7 instructions for the last two terms): < 0.751 ulp on |x| <= pi/4 with FMA steps, against < 0.884 ulp for the
    // two FMAs here (3e7 arguments against 80-bit cosl; 3.6 % vs 4.2 % of the values are not the correctly rounded one).  Five
    // instructions per evaluation (40 of Acrobot's 622 per env-step) for 0.13 ulp of a function whose medium-range version is bounded by
    // its argument reduction anyway: sincos_medium's maxima, 1.466 / 1.498 ulp, are the same with either form (tools/fast_sincos_check.c
    // -DPLAIN).  MXV_FDLIBM_COS = 1 restores the compensated sum (A/B hook).
    double c = fma_coef<F3>(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    c = fma_coef<F3>(z, c, -2.75573143513906633035e-07);
    c = fma_coef<F3>(z, c, 2.48015872894767294178e-05);
    c = fma_coef<F3>(z, c, -1.38888888888741095749e-03);
    c = fma_coef<F3>(z, c, 4.16666666666666019037e-02);
#if MXV_FDLIBM_COS
    const double hz = 0.5 * z;
    const double t = 1.0 - hz;
    *cs = t + __fma_rn(z, z * c, (1.0 - t) - hz);
#else
    c = __fma_rn(z, c, -0.5);
    *cs = __fma_rn(z, c, 1.0);
#endif
}

// General-range sin/cos.  ocml's sincos costs ~80 VALU instructions per call on gfx950 (a 3-term Cody-Waite reduction kept in
// double-double, ~20 v_mov to materialise polynomial coefficients next to v_fmac, a Payne-Hanek branch for |x| >= 2^30) and
// Acrobot calls it eight times per step — two thirds of the most VALU-bound kernel of the engine (917 VALU instructions per
// env-step at 82 % of the VALU issue rate, profiles/r2/r02e_rooflines.jsonl).  mx_sincos is the medium-range version the dynamics
// need: k = rint(x * 2/pi); r = x - k*pi/2 with pi/2 split 33 + 33 + 53 bits (k * P1 and, once r is small, k * P2 are exact, so
// cancellation near multiples of pi/2 costs no accuracy: three FMAs); the fdlibm kernel polynomials of sincos_kernel on
// |r| <= pi/4; quadrant swap and signs from k.  Error <= 1.5 ulp over |x| <= 40 including 3e6 arguments within 5e-7 of a
// multiple of pi/2 (ocml documents 2 ulp; checked on the CPU against 80-bit sinl/cosl with the same FMA arithmetic,
// tools/fast_sincos_check.c), ~35 instructions.  |x| >= 2^19 (k * P1 no longer exact; never produced by these dynamics) goes to
// ocml.  As everywhere in this file, what happens inside a libm-like helper is the library's business: the reference's own
// sin/cos are glibc's resp. NumPy's SIMD loops, which differ from each other in the last bit too.
#ifndef MXV_FAST_TRIG
#define MXV_FAST_TRIG 1  // A/B hook: 0 = ocml's sincos / cos everywhere
#endif
template <int F3 = 0>
__device__ __forceinline__ void sincos_medium(double x, double *sn, double *cs) {
    const double k = rint(x * 6.36619772367581382433e-01);        // 2/pi
    double r = __fma_rn(-k, 1.5707963267341256, x);               // pi/2, first 33 bits: exact product for |k| < 2^20
    r = __fma_rn(-k, 6.077100506303966e-11, r);                   // next 33 bits
    r = __fma_rn(-k, 2.0222662487959506e-21, r);                  // the following 53
    double s, c;
    sincos_kernel<F3>(r, &s, &c);
    const uint32_t q = (uint32_t)(int)k;
    const bool swap = (q & 1u) != 0u;
    const double ss = swap ? c : s, cc = swap ? s : c;
    // sin changes sign in quadrants 2, 3; cos in quadrants 1, 2: bit 1 of q resp. q + 1, moved to bit 31 and XORed into the high dword —
    // `hi ^ (shifted & 0x80000000)` is ONE v_bitop3_b32 (truth table 0x78 = a ^ (b & c)) where the compiler emits v_and + v_xor
    if constexpr ((F3 & 2) != 0) {
        *sn = __hiloint2double((int)xor_masked(__double2hiint(ss), q << 30, 0x80000000u), __double2loint(ss));
        *cs = __hiloint2double((int)xor_masked(__double2hiint(cc), (q + 1u) << 30, 0x80000000u), __double2loint(cc));
    } else {
        *sn = __hiloint2double(__double2hiint(ss) ^ (int)((q & 2u) << 30), __double2loint(ss));
        *cs = __hiloint2double(__double2hiint(cc) ^ (int)(((q + 1u) & 2u) << 30), __double2loint(cc));
    }
}
// GUARD = false: the caller knows |x| < 2^19 (the fused rollout on states the dynamics themselves produced: Acrobot wraps its
// angles to [-pi, pi] and bounds the velocities, MountainCar's argument is 3 * position, a time-limited Pendulum turns at most
// 0.4 rad per step) — no range check and none of ocml's code or registers in the kernel.  mxv_set_state and unusual reset
// bounds break that knowledge for one launch, which then takes the guarded instantiation (see launch_step_env, SAFE).
template <bool GUARD = true, int F3 = 0>
__device__ __forceinline__ void mx_sincos(double x, double *sn, double *cs) {
#if MXV_FAST_TRIG
    if (!GUARD || fabs(x) < 524288.0)
        sincos_medium<F3>(x, sn, cs);
    else
#endif
        sincos(x, sn, cs);
}
template <bool GUARD = true, int F3 = 0>
__device__ __forceinline__ double mx_cos(double x) {
#if MXV_FAST_TRIG
    double s, c;
    mx_sincos<GUARD, F3>(x, &s, &c);
    return c;
#else
    return cos(x);
#endif
}
template <bool GUARD = true, int F3 = 0>
__device__ __forceinline__ double mx_sin(double x) {
    double s, c;
    mx_sincos<GUARD, F3>(x, &s, &c);  // the sine of sincos, so that a cached sine (Pendulum aux) and a fresh one are the same bits
    return s;
}

__device__ __forceinline__ void sincos_small_or_general(double x, double *sn, double *cs) {
    if (fabs(x) <= 0.78539816339744830962) {
        sincos_kernel(x, sn, cs);
    } else {
        mx_sincos(x, sn, cs);
    }
}

// The reference's clamps — np.clip(x, lo, hi), `if x > hi: x = hi`, min(max(x, lo), hi) — as v_max / v_min instead of a compare, a wait
// state for vcc and one select per dword each.  For every x that is not a NaN, `x < lo ? lo : x` IS max(x, lo) bit for bit (lo and hi are
// never zeros of opposite sign, the one other case where the forms differ); a NaN — which only a caller can bring in: a Box action of a
// diverged policy, an injected state — passes through every one of the reference's forms unchanged (np.clip propagates it; Python's
// `NaN > hi`, max(NaN, lo), min(NaN, hi) keep the first operand), whereas v_max / v_min return the OTHER operand.  So the pair is followed
// by a NaN pass-through: float64 = one v_cmp_u_f64 + one v_cndmask_b32 per dword (both: a NaN whose payload sits in the low dword only
// — 0x7FF00000:xxxxxxxx, which a caller can craft — would turn into an infinity on a bound's zero low dword otherwise);
// float32 = nothing extra, gfx950 has the IEEE-754-2019 NaN-propagating v_maximum3_f32 / v_minimum3_f32.
#ifndef MXV_MINMAX_CLAMPS
#define MXV_MINMAX_CLAMPS 1   // A/B hook: 0 = compare + select
#endif
// np.clip(x, lo, hi) = minimum(maximum(x, lo), hi): the lower bound first (the order only shows when a caller sets lo > hi); NaN passes through
__device__ __forceinline__ double nan_through(double x, double r) {
    const bool nan = x != x;
    return __hiloint2double(nan ? __double2hiint(x) : __double2hiint(r), nan ? __double2loint(x) : __double2loint(r));
}
__device__ __forceinline__ double clamp_range(double x, double lo, double hi) {
#if MXV_MINMAX_CLAMPS
    return nan_through(x, __builtin_fmin(__builtin_fmax(x, lo), hi));
#else
    x = (x < lo) ? lo : x;
    return (x > hi) ? hi : x;
#endif
}
__device__ __forceinline__ float clamp_range(float x, float lo, float hi) {
#if MXV_MINMAX_CLAMPS
    float r;
    asm("v_maximum3_f32 %0, %1, %2, %2" : "=v"(r) : "v"(x), "v"(lo));
    asm("v_minimum3_f32 %0, %1, %2, %2" : "=v"(r) : "v"(r), "v"(hi));
    return r;
#else
    x = (x < lo) ? lo : x;
    return (x > hi) ? hi : x;
#endif
}
// `if x > hi: x = hi` then `if x < lo: x = lo` (continuous_mountain_car.py:149-157): the upper bound first
__device__ __forceinline__ double clamp_range_hi_first(double x, double lo, double hi) {
#if MXV_MINMAX_CLAMPS
    return nan_through(x, __builtin_fmax(__builtin_fmin(x, hi), lo));
#else
    x = (x > hi) ? hi : x;
    return (x < lo) ? lo : x;
#endif
}
__device__ __forceinline__ float clamp_range_hi_first(float x, float lo, float hi) {
#if MXV_MINMAX_CLAMPS
    float r;
    asm("v_minimum3_f32 %0, %1, %2, %2" : "=v"(r) : "v"(x), "v"(hi));
    asm("v_maximum3_f32 %0, %1, %2, %2" : "=v"(r) : "v"(r), "v"(lo));
    return r;
#else
    x = (x > hi) ? hi : x;
    return (x < lo) ? lo : x;
#endif
}

// C fmod(a, b) for a compile-time b > 0 and |a| < 2^20 * b in ~8 instructions.  fmod is exact by definition, so any
// exact algorithm returns identical bits: q = trunc(|a| * (1/b)) is the true quotient or off by one, |a| - q*b is
// then exactly representable (a multiple of ulp(b) below 2b), so the FMA computes it without rounding and one
// conditional +-b repairs the off-by-one.  Larger |a| (never produced by these dynamics) goes to ocml's fmod.
__device__ __forceinline__ double fmod_const(double a, double b, double inv_b) {
    const double aa = fabs(a);
    if (!(aa < 1048576.0 * b)) return fmod(a, b);
    const double q = trunc(aa * inv_b);
    double r = __fma_rn(-q, b, aa);
    if (r < 0.0) r += b;
    else if (r >= b) r -= b;
    return copysign(r, a);
}

// ------------------------------------------------------------------------------------------
// Per-env traits: S state scalars, O observation scalars, NA discrete actions (0 = Box).
// step(): dynamics only.  s[] in/out (fp64), `fresh` = elapsed == 0 (first step after reset),
// ai / af = discrete / continuous action.  Returns terminated; writes reward and obs.
// ------------------------------------------------------------------------------------------
template <int ENV>
struct Env;

// ---- CartPole: gym/envs/classic_control/cartpole.py:130-188 -------------------------------
template <>
struct Env<MXV_CARTPOLE> {
    static constexpr int S = 4, O = 4, NA = 2;
    static constexpr int AUX = 0;  // fp64 values derived from the state that a fused rollout carries across steps
    template <bool GUARD = true>
    __device__ __forceinline__ static void prime(const double *s, double *aux) {}
    // SAFE = false (rollout fast path, default parameters only): the caller guarantees |theta| <= pi/4 on entry, which
    // holds inductively after reset() under autoreset (an env leaves (-0.2095, 0.2095) only in the step that ends it);
    // mxv_set_state() breaks the induction, so the launch after it uses the SAFE instantiation (see mxv_api.cpp).
    template <int DEF, bool SAFE = true, int EPL = 0>   // EPL: envs per lane of the calling kernel (0 = not said)
    __device__ __forceinline__ static bool step(const Par<DEF> &P, double *s, double *, bool, int ai, float, double &reward,
                                                float *obs) {
        const double gravity = P.get(0, 9.8), masspole = P.get(2, 0.1), total_mass = P.get(3, 0.1 + 1.0);
        const double length = P.get(4, 0.5), polemass_length = P.get(5, 0.1 * 0.5), force_mag = P.get(6, 10.0);
        const double tau = P.get(7, 0.02), theta_thr = P.get(8, 12 * 2 * kPi / 360), x_thr = P.get(9, 2.4);
        const bool semi_implicit = P.get(10, 0.0) != 0.0;
        double x = s[0], x_dot = s[1], theta = s[2], theta_dot = s[3];
        const double force = (ai == 1) ? force_mag : -force_mag;  // :135
        double sintheta, costheta;
        if constexpr (DEF == PM_DEFAULT && !SAFE)
            sincos_kernel<((fma3_for<MXV_CARTPOLE>() & 1) || (EPL == 1 && MXV_FMA3_CARTPOLE_E1)) ? 1 : 0>(theta, &sintheta, &costheta);
        else
            sincos_small_or_general(theta, &sintheta, &costheta);  // :136-137
        const double temp = div_par<DEF, !SAFE>(force + polemass_length * (theta_dot * theta_dot) * sintheta, total_mass);  // :141-143
        const double ta_num = gravity * sintheta - costheta * temp;
        const double ta_den = length * (4.0 / 3.0 - div_par<DEF, !SAFE>(masspole * (costheta * costheta), total_mass));  // :144-146
        double thetaacc;
        if constexpr (DEF == PM_DEFAULT && !SAFE && MXV_CARTPOLE_RCP)  // ta_den in [0.62, 0.67], ta_num normal and never -0: the scale / fix-up
            thetaacc = div_with_rcp(ta_num, ta_den, refined_rcp(ta_den));  // steps of `/` are identities (same bits, 3 slots fewer)
        else
            thetaacc = ta_num / ta_den;
        const double xacc = temp - div_par<DEF, !SAFE>(polemass_length * thetaacc * costheta, total_mass);      // :147
        if (!semi_implicit) {  // "euler" :149-153
            x = x + tau * x_dot;
            x_dot = x_dot + tau * xacc;
            theta = theta + tau * theta_dot;
            theta_dot = theta_dot + tau * thetaacc;
        } else {  // :154-158
            x_dot = x_dot + tau * xacc;
            x = x + tau * x_dot;
            theta_dot = theta_dot + tau * thetaacc;
            theta = theta + tau * theta_dot;
        }
        s[0] = x; s[1] = x_dot; s[2] = theta; s[3] = theta_dot;
        reward = 1.0;  // :169-184 (autoreset: steps_beyond_terminated is always None)
        obs[0] = (float)x; obs[1] = (float)x_dot; obs[2] = (float)theta; obs[3] = (float)theta_dot;  // :188
        return (x < -x_thr) || (x > x_thr) || (theta < -theta_thr) || (theta > theta_thr);          // :162-167
    }
    template <bool GUARD = true>
    __device__ __forceinline__ static void observe(const double *s, float *obs, double * = nullptr) {  // :207
        obs[0] = (float)s[0]; obs[1] = (float)s[1]; obs[2] = (float)s[2]; obs[3] = (float)s[3];
    }
    // np_random.uniform(low, high, size=(4,)) :202
    __device__ __forceinline__ static void reset(U4 w, double b0, double b1, double *s) {
        s[0] = b0 + (b1 - b0) * u01(w.x);
        s[1] = b0 + (b1 - b0) * u01(w.y);
        s[2] = b0 + (b1 - b0) * u01(w.z);
        s[3] = b0 + (b1 - b0) * u01(w.w);
    }
};

// ---- Pendulum's `u ** 2` (pendulum.py:129) ------------------------------------------------------------------------------------------
// `u` is a np.float32 scalar, so NumPy evaluates `u ** 2` as libm powf(u, 2.0f) — and glibc's powf (>= 2.28: Szabolcs Nagy's
// optimized-routines powf, sysdeps/ieee754/flt-32/e_powf.c with powf_log2_data.c and exp2f_data.c) is NOT correctly rounded: it
// computes exp2(2 * log2|u|) in double (log2 by a 16-entry table and a degree-5 polynomial, exp2 by a 32-entry table and a degree-3
// one; relative error up to 1.27 * 2^-26 by its own header) and rounds that to float, which differs from the correctly rounded
// product u * u for 0.07 % of float32 u in [-2, 2] (tools/powf_variants.py: 1470 of 2 000 000) by one float32 ulp.  Rounds 1-5 multiplied
// (u * u) and carried an `atol 1e-9` on Pendulum's rewards for it.  This is glibc 2.35's algorithm for y = 2 restated — the tables and
// coefficients below are its published constants (the build without rounding intrinsics: POWF_SCALE = 1, SHIFT = 0x1.8p52 / 32), the
// operations in its order — checked bit for bit against the libm of this image on 2.3 * 10^6 inputs (tools/powf_variants.py --emulation)
// and, on the device, against the reference's outputs (profiles/r6/r6g_*: the largest reward deviation drops from 2.3e-10 to 3.6e-15).
// It is a BUILD HOOK, off by default, because the measurement says so: ~35 fp64 instructions and two table reads on a kernel of 115
// cost Pendulum's fused rollout 25-30 % (3.55 vs 2.76 us per 2^19-env step, same tensors), and bit-equality is still out of reach — 9 of
// 9 296 golden rewards stay one fp64 ulp off because `angle_normalize(th) ** 2` and `thdot ** 2` are libm pow(x, 2.0) on np.float64
// scalars, glibc's double-precision pow (two 128-entry tables, ~100 instructions, < 1 ULP but not correctly rounded either).  The
// default keeps u * u and x * x — correctly rounded, within 1 ulp of whatever libm the reference runs on, inside north_star's bar by
// five orders of magnitude — and README's Parity table carries the footnote with these numbers.
#ifndef MXV_PENDULUM_GLIBC_POWF
#define MXV_PENDULUM_GLIBC_POWF 0   // 1: restate glibc 2.35's powf(u, 2.0f) (tools/build_variants.sh "...:-DMXV_PENDULUM_GLIBC_POWF=1")
#endif
static __device__ const uint64_t kExp2fTab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};
static __device__ const double kPowfLog2Tab[16][2] = {   // {1 / c, log2(c)} for the 16 sub-intervals of [0x1.66p-1, 0x1.66p0)
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2}, {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2},
};
__device__ __forceinline__ float glibc_powf_square(float x) {
    uint32_t ix = __float_as_uint(x) & 0x7fffffffu;               // y = 2 is an even integer: the sign goes, sign_bias = 0
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {          // zero, subnormal, Inf, NaN (e_powf.c: the "special cases" block)
        if (ix == 0u || ix >= 0x7f800000u) return x * x;          //   0 -> 0, Inf -> Inf, NaN -> NaN
        ix = __float_as_uint(x * 0x1p23f) & 0x7fffffffu;          //   subnormal: normalise
        ix -= 23u << 23;
    }
    // log2_inline
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const uint32_t top = tmp & 0xff800000u;
    const double k = (double)((int32_t)top >> 23);
    const double z = (double)__uint_as_float(ix - top);
    const double invc = kPowfLog2Tab[i][0], logc = kPowfLog2Tab[i][1];
    const double r = z * invc - 1.0;
    const double y0 = logc + k;
    const double r2 = r * r;
    double y = 0x1.27616c9496e0bp-2 * r + -0x1.71969a075c67ap-2;
    const double p = 0x1.ec70a6ca7baddp-2 * r + -0x1.7154748bef6c8p-1;
    const double r4 = r2 * r2;
    double q = 0x1.71547652ab82bp+0 * r + y0;
    q = p * r2 + q;
    y = y * r4 + q;
    const double ylogx = 2.0 * y;
    if (ylogx <= -150.0) return 0.0f;                             // __math_may_uflowf (|x| < 2^-75)
    // exp2_inline (no overflow: |x| <= FLT_MAX would need ylogx > 127.99..., i.e. |x| > 1.8e19 — kept for completeness)
    if (ylogx > 0x1.fffffffd1d571p+6) return __uint_as_float(0x7f800000u);
    double kd = ylogx + 0x1.8p+47;
    const uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd -= 0x1.8p+47;
    const double rr = ylogx - kd;
    uint64_t t = kExp2fTab[ki & 31u];
    t += ki << 47;
    const double sc = __longlong_as_double((long long)t);
    const double zz = 0x1.c6af84b912394p-5 * rr + 0x1.ebfce50fac4f3p-3;
    const double rr2 = rr * rr;
    double yy = 0x1.62e42ff0c52d6p-1 * rr + 1.0;
    yy = zz * rr2 + yy;
    yy = yy * sc;
    return (float)yy;
}

// ---- Pendulum: gym/envs/classic_control/pendulum.py:119-139,161-163,270-271 ---------------
// state fp64, action fp32; python-float (op) np.float32 stays float32 under NumPy-2 promotion.
__device__ __forceinline__ double np_remainder(double a, double b) {  // numpy float64 `%` (npy_divmod); b is a literal
    double mod = fmod_const(a, b, 1.0 / b);
    if (b == 0.0) return mod;
    if (mod != 0.0) {
        if ((b < 0) != (mod < 0)) mod += b;
    } else {
        mod = copysign(0.0, b);
    }
    return mod;
}
// The same value for a literal b > 0 when the caller knows |a| < 2^20 * b (the fused rollout on a time-limited Pendulum: |theta| < 90),
// as straight-line selects: no range guard with ocml's fmod behind it, no exec-mask blocks around single additions.  Every step is exact
// (see fmod_const), so the bits are those of np_remainder by construction.
#ifndef MXV_REMAINDER_SELECTS
#define MXV_REMAINDER_SELECTS 1   // A/B hook
#endif
__device__ __forceinline__ double np_remainder_bounded(double a, double b) {
#if MXV_REMAINDER_SELECTS
    const double aa = fabs(a);
    const double q = trunc(aa * (1.0 / b));
    double r = __fma_rn(-q, b, aa);              // exact: the true remainder or off by one period
    const double up = r + b, down = r - b;
    r = (r < 0.0) ? up : ((r >= b) ? down : r);
    double mod = copysign(r, a);                 // C fmod: the sign of the dividend
    const double lifted = mod + b;               // Python / NumPy `%`: the sign of the divisor
    mod = (mod < 0.0) ? lifted : mod;
    return (mod == 0.0) ? 0.0 : mod;             // copysign(0.0, b) for a zero remainder
#else
    return np_remainder(a, b);
#endif
}

template <>
struct Env<MXV_PENDULUM> {
    static constexpr int S = 2, O = 3, NA = 0;
    static constexpr int AUX = 1;  // fp64 values derived from the state that a fused rollout carries across steps
    template <bool GUARD = true>
    __device__ __forceinline__ static void prime(const double *s, double *aux) { aux[0] = mx_sin<GUARD, fma3_for<MXV_PENDULUM>()>(s[0]); }
    // aux[0] = sin(theta): `sin(th)` of step t+1 (:131) is the sine _get_obs took at the end of step t (:162)
    template <bool GUARD = true, int F3 = fma3_for<MXV_PENDULUM>()>
    __device__ __forceinline__ static void observe(const double *s, float *obs, double *aux) {  // :161-163
        double sn, cs;
        mx_sincos<GUARD, F3>(s[0], &sn, &cs);
        obs[0] = (float)cs; obs[1] = (float)sn; obs[2] = (float)s[1];
        aux[0] = sn;
    }
    template <int DEF, bool SAFE = true, int EPL = 0>   // EPL: envs per lane of the calling kernel (0 = not said)
    __device__ __forceinline__ static bool step(const Par<DEF> &P, double *s, double *aux, bool, int, float a0,
                                                double &reward, float *obs) {
        const double max_speed = P.get(0, 8.0), max_torque = P.get(1, 2.0), dt = P.get(2, 0.05);
        const double g = P.get(3, 10.0), m = P.get(4, 1.0), l = P.get(5, 1.0);
        const double th = s[0], thdot = s[1];
        const float lo = (float)(-max_torque), hi = (float)max_torque;  // np.clip(u, -max_torque, max_torque)[0] :127
        float u = a0;
        u = clamp_range(u, lo, hi);
#if MXV_PENDULUM_GLIBC_POWF
        const float uterm = (float)0.001 * glibc_powf_square(u);          // 0.001 * (u**2) in float32 :129 — u**2 = libm powf(u, 2.0f)
#else
        const float uterm = (float)0.001 * (u * u);
#endif
        const double an = (SAFE ? np_remainder(th + kPi, 2 * kPi) : np_remainder_bounded(th + kPi, 2 * kPi)) - kPi;  // angle_normalize :270-271
        const double costs = an * an + 0.1 * (thdot * thdot) + (double)uterm;
        const double A = 3 * g / (2 * l);                                  // python floats :131
        const float B = (float)(3.0 / (m * (l * l)));
        const float Bu = B * u;                                            // python float * np.float32 -> f32
        double newthdot = thdot + (A * aux[0] + (double)Bu) * dt;         // aux[0] = sin(th)
        newthdot = clamp_range(newthdot, -max_speed, max_speed);    // np.clip :132
        const double newth = th + newthdot * dt;                           // :133
        s[0] = newth; s[1] = newthdot;
        reward = -costs;                                                   // :139
        // two envs per lane (the launches with fused batch moments only) run at the 128-VGPR cap: the polynomial coefficients the
        // three-address FMA form keeps in ~40 registers across the loop came back from scratch every step there; the compiler's own
        // form (literals materialised where used) costs instructions instead.  Same operations, same bits.
        observe<SAFE, (EPL == 2 ? (fma3_for<MXV_PENDULUM>() & ~1) : fma3_for<MXV_PENDULUM>())>(s, obs, aux);
        return false;
    }
    // high = (x_init, y_init), low = -high; np_random.uniform(low, high) :141-154
    __device__ __forceinline__ static void reset(U4 w, double b0, double b1, double *s) {
        s[0] = -b0 + (b0 - (-b0)) * u01(w.x);
        s[1] = -b1 + (b1 - (-b1)) * u01(w.y);
    }
};

// ---- Acrobot: gym/envs/classic_control/acrobot.py:196-277 (step, _dsdt), 378-465 (wrap, bound, rk4) ----
// Trig budget.  The reference evaluates, per RK4 stage, sin/cos(theta2), cos(theta1 + theta2 - pi/2) and cos(theta1 - pi/2)
// (:252-265), and after the step cos/sin of both angles (:225-230) plus cos(theta1), cos(theta2 + theta1) (:235): 6 sincos +
// 9 cos per step, each ~70 fp64 instructions, i.e. ~3/4 of this VALU-bound kernel.  Every one of those values is a function
// of sin/cos of the two stage angles, so each stage takes ONE sincos per angle and rebuilds the shifted / summed cosines by
// angle addition:
//     cos(t1 - pi/2) = sin t1,   cos(t1 + t2 - pi/2) = sin(t1 + t2) = s1 c2 + c1 s2,   cos(t2 + t1) = c1 c2 - s1 s2
// and the sin/cos of the post-step angles (the observation) ARE the stage-1 values of the next step, so a fused rollout carries
// them in registers: 8 sincos per step instead of 6 sincos + 9 cos.
// What the identities leave out is the ROUNDING OF THE REFERENCE'S ARGUMENTS: it takes the cosine of fl(fl(t1 + t2) - fl(pi/2)), which
// is off the exact t1 + t2 - pi/2 by up to ~5e-16, i.e. a few ulps of the cosine.  Rounds 2-3 carried those roundings explicitly
// (three TwoSum residual chains per stage: cos(fl(x - p)) = sin x + (delta - e) cos x, ...; MXV_ACROBOT_CARRY_ARG_ROUNDING = 1): 90 of the
// kernel's 732 VALU instructions per env-step for agreement to ~2 ulps instead of ~5 per stage cosine — against bars of 1e-12 on the
// fp64 state and 2 float32 ulps on the observations (golden P1 states: 99 % of post-step state components within 59 fp64 ulps of the
// reference instead of 16, the worst the same 2.6e-12 relative; tools/acrobot_threshold_ab.c).  Round 4: the mask no longer depends
// on those ulps (the exact band below decides near the threshold), so the hot path takes the identities as they are.
#ifndef MXV_ACROBOT_CARRY_ARG_ROUNDING
#define MXV_ACROBOT_CARRY_ARG_ROUNDING 0   // A/B hook
#endif
__device__ __forceinline__ double two_sum_residual(double a, double b, double sum) {  // a + b == sum + residual exactly
    const double bb = sum - a;
    return (a - (sum - bb)) + (b - bb);
}
constexpr double kHalfPiTail = 6.123233995736766036e-17;  // pi/2 - fl(pi/2)
#ifndef MXV_ACROBOT_EXACT_BAND
#define MXV_ACROBOT_EXACT_BAND 1   // A/B hook: 0 = the hot path's mask everywhere (rounds 1-3)
#endif
#ifndef MXV_ACROBOT_DIRECT_COS
#define MXV_ACROBOT_DIRECT_COS 0   // measurement only (tools/acrobot_threshold_ab.py): every cosine evaluated directly, no angle addition
#endif

template <>
struct Env<MXV_ACROBOT> {
    static constexpr int S = 4, O = 6, NA = 3;
    static constexpr int AUX = 4;  // sin(theta1), cos(theta1), sin(theta2), cos(theta2) of the current state
    template <bool GUARD = true>
    __device__ __forceinline__ static void prime(const double *s, double *aux) {
        mx_sincos<GUARD, fma3_for<MXV_ACROBOT>()>(s[0], &aux[0], &aux[1]);
        mx_sincos<GUARD, fma3_for<MXV_ACROBOT>()>(s[1], &aux[2], &aux[3]);
    }
    // sc = sin/cos of sa[0], sa[1]
    template <int DEF>
    __device__ __forceinline__ static void dsdt(const Par<DEF> &P, const double *sa, const double *sc, double a, double *out) {
        const double m1 = P.get(3, 1.0), m2 = P.get(4, 1.0), l1 = P.get(1, 1.0);
        const double lc1 = P.get(5, 0.5), lc2 = P.get(6, 0.5), I1 = P.get(7, 1.0), I2 = P.get(7, 1.0);
        const bool nips = P.get(11, 0.0) != 0.0;
        const double g = 9.8;  // :245
        [[maybe_unused]] const double theta1 = sa[0], theta2 = sa[1];   // (the angles themselves: only the A/B forms below read them)
        const double dtheta1 = sa[2], dtheta2 = sa[3];
        const double s1 = sc[0], c1 = sc[1], s2 = sc[2], c2 = sc[3];
        [[maybe_unused]] const double halfpi = kPi / 2.0;
        // cos(theta1 + theta2 - pi / 2.0) :259 and cos(theta1 - pi / 2) :264 from the stage's four values (see the trig budget above)
        const double S12 = __fma_rn(s1, c2, c1 * s2);
#if MXV_ACROBOT_DIRECT_COS
        const double cos_t12_shift = mx_cos(theta1 + theta2 - halfpi), cos_t1_shift = mx_cos(theta1 - halfpi);
#elif MXV_ACROBOT_CARRY_ARG_ROUNDING
        const double t12 = theta1 + theta2;
        const double a12 = t12 - halfpi;
        const double eps12 = kHalfPiTail - two_sum_residual(theta1, theta2, t12) - two_sum_residual(t12, -halfpi, a12);
        const double C12 = __fma_rn(c1, c2, -(s1 * s2));
        const double cos_t12_shift = __fma_rn(eps12, C12, S12);
        const double a1 = theta1 - halfpi;
        const double cos_t1_shift = __fma_rn(kHalfPiTail - two_sum_residual(theta1, -halfpi, a1), c1, s1);
#else
        const double cos_t12_shift = S12, cos_t1_shift = s1;   // sin(t1 + t2), sin(t1)
#endif
        const double d1 = m1 * (lc1 * lc1) + m2 * ((l1 * l1) + (lc2 * lc2) + 2 * l1 * lc2 * c2) + I1 + I2;  // :252-257
        const double d2 = m2 * ((lc2 * lc2) + l1 * lc2 * c2) + I2;                                          // :258
        const double phi2 = m2 * lc2 * g * cos_t12_shift;                                                   // :259
        const double phi1 = -m2 * l1 * lc2 * (dtheta2 * dtheta2) * s2 - 2 * m2 * l1 * lc2 * dtheta2 * dtheta1 * s2 +
                            (m1 * lc1 + m2 * l1) * g * cos_t1_shift + phi2;                                 // :260-265
        double ddtheta2, ddtheta1;
        if constexpr (DEF == PM_DEFAULT) {  // "book" dynamics, three quotients by d1 share one reciprocal (same bits as `/`)
            const double r1 = refined_rcp(d1);
            // the quotient by (m2 lc2^2 + I2 - d2^2/d1) in [0.57, 1.25]: same reciprocal-based sequence, no scaling / fix-up needed
            const double den2 = m2 * (lc2 * lc2) + I2 - div_with_rcp(d2 * d2, d1, r1);
            ddtheta2 = div_with_rcp(a + div_with_rcp(d2, d1, r1) * phi1 - m2 * l1 * lc2 * (dtheta1 * dtheta1) * s2 - phi2, den2,
                                    refined_rcp(den2));  // :270-275
            ddtheta1 = div_with_rcp(-(d2 * ddtheta2 + phi1), d1, r1);              // :276
        } else {
            if (nips) {  // :266-269
                ddtheta2 = (a + d2 / d1 * phi1 - phi2) / (m2 * (lc2 * lc2) + I2 - (d2 * d2) / d1);
            } else {  // "book" :270-275
                ddtheta2 = (a + d2 / d1 * phi1 - m2 * l1 * lc2 * (dtheta1 * dtheta1) * s2 - phi2) /
                           (m2 * (lc2 * lc2) + I2 - (d2 * d2) / d1);
            }
            ddtheta1 = -(d2 * ddtheta2 + phi1) / d1;  // :276
        }
        out[0] = dtheta1; out[1] = dtheta2; out[2] = ddtheta1; out[3] = ddtheta2;  // :277 (5th component is 0.0)
    }
    __device__ __forceinline__ static double wrap(double x, double m, double M) {  // :378-396
        const double diff = M - m;
        while (x > M) x = x - diff;
        while (x < m) x = x + diff;
        return x;
    }
    __device__ __forceinline__ static double bound(double x, double m, double M) {  // :399-415 min(max(x, m), M)
        return clamp_range(x, m, M);
    }
    template <bool GUARD = true>
    __device__ __forceinline__ static void observe(const double *s, float *obs, double *aux = nullptr) {  // :225-230
        double s0, c0, s1, c1;
        mx_sincos<GUARD, fma3_for<MXV_ACROBOT>()>(s[0], &s0, &c0);
        mx_sincos<GUARD, fma3_for<MXV_ACROBOT>()>(s[1], &s1, &c1);
        obs[0] = (float)c0; obs[1] = (float)s0; obs[2] = (float)c1; obs[3] = (float)s1;
        obs[4] = (float)s[2]; obs[5] = (float)s[3];
        if (aux) { aux[0] = s0; aux[1] = c0; aux[2] = s1; aux[3] = c1; }
    }
    // the cold exact path (mxv_exact.hpp) with this launch's parameter values.  The step's arithmetic is inlined into the caller's cold
    // block; cr_sincos, which it calls 21 times, is the one real function call (40 VGPRs, inside the caller-saved range, so the kernel's
    // register budget stays its own: a fully non-inlined chain pushed the kernel to 198-248 VGPRs through the callee-saved ranges)
    template <int DEF>
    __device__ __forceinline__ static bool acrobot_exact(const Par<DEF> &P, double *s, double torque, double *sc) {
        const double Pv[12] = {P.get(0, 0.2), P.get(1, 1.0), P.get(2, 1.0), P.get(3, 1.0),  P.get(4, 1.0),  P.get(5, 0.5),
                               P.get(6, 0.5), P.get(7, 1.0), P.get(8, 4 * kPi), P.get(9, 9 * kPi), P.get(10, 0.0), P.get(11, 0.0)};
        return exact::acrobot_step_exact(Pv, s, torque, sc);
    }
    template <int DEF, bool SAFE = true, int EPL = 0>   // EPL: envs per lane of the calling kernel (0 = not said)
    // `noise_word`: for this env the Box-action slot of the shared step() signature carries the raw Philox word of the
    // step-noise stream (bit pattern in a float), consumed only when torque_noise_max > 0 (never on the default path).
    __device__ __forceinline__ static bool step(const Par<DEF> &P, double *s, double *aux, bool, int ai, float noise_word,
                                                double &reward, float *obs) {
        double torque = (double)(ai - 1);  // AVAIL_TORQUE[a] = [-1.0, 0.0, +1] :157,199
        if constexpr (DEF != PM_DEFAULT) {
            const double nm = P.get(10, 0.0);  // torque += np_random.uniform(-torque_noise_max, torque_noise_max) :202-205
            if (nm > 0.0) torque += -nm + (nm - (-nm)) * u01(__float_as_uint(noise_word));
        }
        const double dt = P.get(0, 0.2) - 0;    // t[i+1] - this, t = [0, self.dt] :210,449
        const double dt2 = dt / 2.0;            // :450
        const double y0[4] = {s[0], s[1], s[2], s[3]};
        double k1[4], k2[4], k3[4], k4[4], y[4], sc[4];
        dsdt(P, y0, aux, torque, k1);  // :453  (the augmented torque component has derivative 0.0: it stays `torque`)
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = y0[k] + dt2 * k1[k];
        mx_sincos<SAFE, fma3_for<MXV_ACROBOT>()>(y[0], &sc[0], &sc[1]);
        mx_sincos<SAFE, fma3_for<MXV_ACROBOT>()>(y[1], &sc[2], &sc[3]);
        dsdt(P, y, sc, torque, k2);   // :454
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = y0[k] + dt2 * k2[k];
        mx_sincos<SAFE, fma3_for<MXV_ACROBOT>()>(y[0], &sc[0], &sc[1]);
        mx_sincos<SAFE, fma3_for<MXV_ACROBOT>()>(y[1], &sc[2], &sc[3]);
        dsdt(P, y, sc, torque, k3);   // :455
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = y0[k] + dt * k3[k];
        mx_sincos<SAFE, fma3_for<MXV_ACROBOT>()>(y[0], &sc[0], &sc[1]);
        mx_sincos<SAFE, fma3_for<MXV_ACROBOT>()>(y[1], &sc[2], &sc[3]);
        dsdt(P, y, sc, torque, k4);   // :456
        const double dt6 = dt / 6.0;
        double ns[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ns[k] = y0[k] + dt6 * (k1[k] + 2 * k2[k] + 2 * k3[k] + k4[k]);  // :463
        s[0] = wrap(ns[0], -kPi, kPi);                       // :213
        s[1] = wrap(ns[1], -kPi, kPi);                       // :214
        s[2] = bound(ns[2], -P.get(8, 4 * kPi), P.get(8, 4 * kPi));  // :215
        s[3] = bound(ns[3], -P.get(9, 9 * kPi), P.get(9, 9 * kPi));  // :216
        double s0, c0, s1, c1;
        mx_sincos<SAFE, fma3_for<MXV_ACROBOT>()>(s[0], &s0, &c0);
        mx_sincos<SAFE, fma3_for<MXV_ACROBOT>()>(s[1], &s1, &c1);
        // cos(s[1] + s[0]) :235 from the same four values
        [[maybe_unused]] const double t21 = s[1] + s[0];
#if MXV_ACROBOT_DIRECT_COS
        const double cos21 = mx_cos(t21);
#elif MXV_ACROBOT_CARRY_ARG_ROUNDING
        const double cos21 = __fma_rn(two_sum_residual(s[1], s[0], t21), __fma_rn(s0, c1, c0 * s1), __fma_rn(c0, c1, -(s0 * s1)));
#else
        const double cos21 = __fma_rn(c0, c1, -(s0 * s1));
#endif
        const double height = -c0 - cos21;
        bool term = height > 1.0;                            // :235
#if MXV_ACROBOT_EXACT_BAND
        // Within 2^-40 of the threshold (the hot path's own error is a few 2^-52; ~1 env-step in 10^12 comes here) the mask is not
        // decided by 1.5-ulp trigonometry: the whole step is taken again as the reference writes it, on correctly rounded sin / cos
        // (mxv_exact.hpp), and THAT state, observation and mask stand.
        if (__builtin_expect(fabs(height - 1.0) < 0x1p-40, 0)) {
            double sx[4] = {y0[0], y0[1], y0[2], y0[3]}, scx[4];
            term = acrobot_exact(P, sx, torque, scx);
            // the results come back through scratch memory: they are consumed (`landed`) and every access of this block is waited for
            // HERE, so that the block every step runs through keeps no vmcnt wait of its own (it would wait for the stores in flight)
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] = landed(sx[k]);
            s0 = landed(scx[0]); c0 = landed(scx[1]); s1 = landed(scx[2]); c1 = landed(scx[3]);
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
#endif
        reward = (!term) ? -1.0 : 0.0;                     // :219
        obs[0] = (float)c0; obs[1] = (float)s0; obs[2] = (float)c1; obs[3] = (float)s1;
        obs[4] = (float)s[2]; obs[5] = (float)s[3];
        aux[0] = s0; aux[1] = c0; aux[2] = s1; aux[3] = c1;  // stage-1 values of the next step
        return term;
    }
    // np_random.uniform(low, high, size=(4,)).astype(np.float32) :188-190
    __device__ __forceinline__ static void reset(U4 w, double b0, double b1, double *s) {
        s[0] = (double)(float)(b0 + (b1 - b0) * u01(w.x));
        s[1] = (double)(float)(b0 + (b1 - b0) * u01(w.y));
        s[2] = (double)(float)(b0 + (b1 - b0) * u01(w.z));
        s[3] = (double)(float)(b0 + (b1 - b0) * u01(w.w));
    }
};

// ---- MountainCar: gym/envs/classic_control/mountain_car.py:127-148 ------------------------
template <>
struct Env<MXV_MOUNTAINCAR> {
    static constexpr int S = 2, O = 2, NA = 3;
    static constexpr int AUX = 0;  // fp64 values derived from the state that a fused rollout carries across steps
    template <bool GUARD = true>
    __device__ __forceinline__ static void prime(const double *s, double *aux) {}
    template <bool GUARD = true>
    __device__ __forceinline__ static void observe(const double *s, float *obs, double * = nullptr) {
        obs[0] = (float)s[0]; obs[1] = (float)s[1];
    }
    template <int DEF, bool SAFE = true, int EPL = 0>   // EPL: envs per lane of the calling kernel (0 = not said)
    __device__ __forceinline__ static bool step(const Par<DEF> &P, double *s, double *, bool, int ai, float, double &reward,
                                                float *obs) {
        const double min_position = P.get(0, -1.2), max_position = P.get(1, 0.6), max_speed = P.get(2, 0.07);
        const double goal_position = P.get(3, 0.5), goal_velocity = P.get(4, 0.0);
        const double force = P.get(5, 0.001), gravity = P.get(6, 0.0025);
        double position = s[0], velocity = s[1];
        velocity = velocity + ((double)(ai - 1) * force + mx_cos<SAFE, fma3_for<MXV_MOUNTAINCAR>()>(3 * position) * (-gravity));  // :133
        velocity = clamp_range(velocity, -max_speed, max_speed);                           // np.clip :134
        position = position + velocity;                                                    // :135
        position = clamp_range(position, min_position, max_position);                      // np.clip :136
        if (position == min_position && velocity < 0) velocity = 0;                        // :137-138
        s[0] = position; s[1] = velocity;
        reward = -1.0;                                                                     // :143
        observe(s, obs);
        return position >= goal_position && velocity >= goal_velocity;                     // :140-142
    }
    // np.array([np_random.uniform(low, high), 0]) :160
    __device__ __forceinline__ static void reset(U4 w, double b0, double b1, double *s) {
        s[0] = b0 + (b1 - b0) * u01(w.x);
        s[1] = 0.0;
    }
};

// ---- MountainCarContinuous: gym/envs/classic_control/continuous_mountain_car.py:142-175 ----
// NumPy-2 semantics: float32 state/update after the first step, float64 on the first step after
// reset (`fresh`), force term always float32 arithmetic on the float32 action unless clipped.
template <>
struct Env<MXV_MOUNTAINCAR_CONT> {
    static constexpr int S = 2, O = 2, NA = 0;
    static constexpr int AUX = 0;  // fp64 values derived from the state that a fused rollout carries across steps
    template <bool GUARD = true>
    __device__ __forceinline__ static void prime(const double *s, double *aux) {}
    template <bool GUARD = true>
    __device__ __forceinline__ static void observe(const double *s, float *obs, double * = nullptr) {
        obs[0] = (float)s[0]; obs[1] = (float)s[1];
    }
    template <int DEF, bool SAFE = true, int EPL = 0>   // EPL: envs per lane of the calling kernel (0 = not said)
    __device__ __forceinline__ static bool step(const Par<DEF> &P, double *s, double *, bool fresh, int, float a0,
                                                double &reward, float *obs) {
        const double min_action = P.get(0, -1.0), max_action = P.get(1, 1.0);
        const double min_position = P.get(2, -1.2), max_position = P.get(3, 0.6), max_speed = P.get(4, 0.07);
        const double goal_position = P.get(5, 0.45), goal_velocity = P.get(6, 0.0), power = P.get(7, 0.0015);
        // force = min(max(action[0], min_action), max_action) :146 (python max/min)
        const bool clipped_lo = min_action > (double)a0;
        const double tmp = clipped_lo ? min_action : (double)a0;
        const bool clipped_hi = max_action < tmp;
        const bool clipped = clipped_lo || clipped_hi;
        const double force_py = clipped_hi ? max_action : min_action;
        const float fp = a0 * (float)power;  // np.float32 * python float -> float32
        bool term;
        if (fresh) {
            double position = s[0], velocity = s[1];
            const double g = 0.0025 * mx_cos<SAFE, fma3_for<MXV_MOUNTAINCAR_CONT>()>(3 * position);  // :148
            const double inc = clipped ? (force_py * power - g) : (double)(fp - (float)g);
            velocity = velocity + inc;
            velocity = clamp_range_hi_first(velocity, -max_speed, max_speed);  // :149-152
            position = position + velocity;                    // :153
            position = clamp_range_hi_first(position, min_position, max_position);  // :154-157
            if (position == min_position && velocity < 0) velocity = 0;  // :158-159
            term = position >= goal_position && velocity >= goal_velocity;  // :162-164
            s[0] = (double)(float)position;  // :171 dtype=np.float32
            s[1] = (double)(float)velocity;
        } else {
            float position = (float)s[0], velocity = (float)s[1];
            const float three_p = 3.0f * position;            // int * np.float32 -> float32
            const double g = 0.0025 * mx_cos<SAFE, fma3_for<MXV_MOUNTAINCAR_CONT>()>((double)three_p);
            const float inc = clipped ? (float)(force_py * power - g) : (fp - (float)g);
            velocity = velocity + inc;
            velocity = clamp_range_hi_first(velocity, (float)(-max_speed), (float)max_speed);
            position = position + velocity;
            position = clamp_range_hi_first(position, (float)min_position, (float)max_position);
            if (position == (float)min_position && velocity < 0) velocity = 0;
            term = position >= (float)goal_position && velocity >= (float)goal_velocity;
            s[0] = (double)position;
            s[1] = (double)velocity;
        }
        double rew = term ? 100.0 : 0.0;                // :166-168
        rew = rew - ((double)a0 * (double)a0) * 0.1;    // math.pow(action[0], 2) * 0.1 :169
        reward = rew;
        observe(s, obs);                                // :175 returns self.state
        return term;
    }
    __device__ __forceinline__ static void reset(U4 w, double b0, double b1, double *s) {  // :182
        s[0] = b0 + (b1 - b0) * u01(w.x);
        s[1] = 0.0;
    }
};

// Action stream of the classic-control engine (see include/mxv.h, RNG contract).  One Philox call serves the group of 4
// consecutive global envs g = env >> 2; which STEPS it serves depends on the action space:
//   Discrete(2)  (CartPole): a uniform action is one random bit, so a call is indexed by the 32-step block b = t >> 5
//                (stream kStreamActionBits) and step t takes bit t & 31 of the env's word — exactly uniform (every bit of a
//                Philox word is), and 1/32 of the Philox work of a word per step;
//   Discrete(3), Box: one word per env and step (stream kStreamAction): (word * n) >> 32 resp. float32(lo + (hi - lo) * u01).
template <int ENV>
constexpr bool action_bits() { return Env<ENV>::NA == 2; }
template <int ENV>
constexpr int action_unit_shift() { return action_bits<ENV>() ? 5 : 0; }   // steps per call = 1 << shift

template <int ENV>
__device__ __forceinline__ U4 action_unit_counter(uint64_t unit, uint64_t g) {
    U4 c;
    c.x = (uint32_t)g;
    c.y = (uint32_t)(g >> 32);
    c.z = (uint32_t)unit;
    c.w = ((uint32_t)(unit >> 32) & 0x0fffffffu) | ((action_bits<ENV>() ? kStreamActionBits : kStreamAction) << 28);
    return c;
}
// words of group g for the call that covers step t
template <int ENV>
__device__ __forceinline__ U4 env_action_words(uint64_t action_seed, uint64_t t, uint64_t g) {
    return philox4x32_10(action_unit_counter<ENV>(t >> action_unit_shift<ENV>(), g), (uint32_t)action_seed,
                         (uint32_t)(action_seed >> 32));
}

template <int ENV, int DEF>
__device__ __forceinline__ void action_from_word(const Par<DEF> &P, uint32_t w, uint64_t t, int &ai, float &af) {
    if constexpr (action_bits<ENV>()) {
        ai = (int)((w >> ((uint32_t)t & 31u)) & 1u);
        af = 0.0f;
    } else if constexpr (Env<ENV>::NA > 0) {
        ai = (int)(((uint64_t)w * (uint32_t)Env<ENV>::NA) >> 32);
        af = 0.0f;
    } else if constexpr (ENV == MXV_PENDULUM) {
        const double lo = -P.get(1, 2.0), hi = P.get(1, 2.0);  // Box(-max_torque, max_torque) pendulum.py:113-115
        ai = 0;
        af = (float)(lo + (hi - lo) * u01(w));
    } else {
        const double lo = P.get(0, -1.0), hi = P.get(1, 1.0);  // Box(min_action, max_action) continuous_mountain_car.py:132-134
        ai = 0;
        af = (float)(lo + (hi - lo) * u01(w));
    }
}

}  // namespace mxv
