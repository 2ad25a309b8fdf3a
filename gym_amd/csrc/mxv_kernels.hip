// mxv_kernels.hip — hand-written gfx950 kernels of the vectorised classic-control engine.
//
// step_kernel: ONE launch = K x SyncVectorEnv.step_wait (gym/vector/sync_vector_env.py:135-169) for
// every env of the shard: dynamics + TimeLimit (gym/wrappers/time_limit.py:50-54) + termination
// + autoreset, optionally with the actions drawn on device (Philox action stream); K > 1 fuses a
// rollout chunk into one launch with the env state resident in registers between steps.
//
// Mapping.  A workgroup of 256 lanes (4 wave64) owns a tile of E*256 consecutive envs; lane `tid`
// owns envs tile0 + j*256 + tid, j < E.  For every j a wave touches 64 consecutive elements of
// each struct-of-arrays state component (512 B of fp64), of elapsed[], reward[], the flag arrays
// and 64 consecutive observation rows — every global access is a fully coalesced, line-aligned
// burst, and the E independent env chains per lane give the VALU instruction-level parallelism
// while E*(S+1) loads per lane are in flight.  No MFMA: this is element-wise fp64 physics.
// LDS is used only to transpose the Philox action words: one Philox call yields the words of 4
// consecutive envs (group g = env>>2), which belong to 4 different lanes under the mapping above.
#include <cstdlib>
#include <type_traits>

#include "mxv_kernels.hpp"

namespace mxv {

namespace {

// The error word only ever receives single-bit codes, so a plain store does what an atomic OR would — and it also works when the
// word lives in pinned host memory (host steps of small envs, mxv_api.cpp: ErrInBlock), where a PCIe atomic might not.
__device__ __forceinline__ void raise_error(int32_t *err, int32_t bit) { *reinterpret_cast<volatile int32_t *>(err) = bit; }

// The single-step launch's stores carry the nontemporal hint: nothing it writes is read again before the launch ends, and lines that do
// not linger in the L2s shorten the write-back at the end of the kernel — step(actions) at 2^20 envs 18.8 -> 18.4 / 18.7 us in two boxes, i.e.
// 1-2 % (on the loads the same hint COSTS 4 us: profiles/r6/r6i_step_launch_shape.md).  -DMXV_STEP_NT_STORES=0 is the A/B hook.
#ifndef MXV_STEP_NT_STORES
#define MXV_STEP_NT_STORES 1
#endif
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x2_t __attribute__((ext_vector_type(2)));
template <bool NT, typename T>
__device__ __forceinline__ void st(T *p, T v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
// elapsed[] is stored in 16 bits whenever the handle's TimeLimit fits (mxv_api.cpp: elapsed16): step(actions) reads and writes it every
// launch, 2 + 2 instead of 4 + 4 bytes per env-step.  The counter saturates at 65535 (only reachable without autoreset, far beyond the limit:
// `truncated` stays set, `elapsed == 0` stays false).  A wave-uniform branch on a kernel argument at entry and exit, nothing in the loops.
__device__ __forceinline__ int32_t load_elapsed(const void *p, int32_t el16, int64_t e) {
    return el16 ? (int32_t) static_cast<const uint16_t *>(p)[e] : static_cast<const int32_t *>(p)[e];
}
template <bool NT = false>
__device__ __forceinline__ void store_elapsed(void *p, int32_t el16, int64_t e, int32_t v) {
    if (el16)
        st<NT>(static_cast<uint16_t *>(p) + e, (uint16_t)(v > 65535 ? 65535 : v));
    else
        st<NT>(static_cast<int32_t *>(p) + e, v);
}

template <int O, bool NT = false>
__device__ __forceinline__ void store_obs(float *base, int64_t e, const float *o) {
    if constexpr (O == 4) {
        st<NT>(reinterpret_cast<f32x4_t *>(base) + e, f32x4_t{o[0], o[1], o[2], o[3]});
    } else if constexpr (O == 2) {
        st<NT>(reinterpret_cast<f32x2_t *>(base) + e, f32x2_t{o[0], o[1]});
    } else if constexpr (O == 6) {
        float2 *p = reinterpret_cast<float2 *>(base) + e * 3;
        p[0] = make_float2(o[0], o[1]);
        p[1] = make_float2(o[2], o[3]);
        p[2] = make_float2(o[4], o[5]);
    } else {
        float *p = base + e * O;
#pragma unroll
        for (int k = 0; k < O; ++k) p[k] = o[k];
    }
}

// A 32-bit lane offset that the instruction selector still sees as zext(i32) INSIDE the loop body.  Loop-invariant code motion hoists the
// 64-bit extension of a lane offset out of the K-step loop; instruction selection works block by block, so in the loop the offset then
// arrives as an opaque 64-bit pair and every store pays a v_lshl_add_u64 (offset pair + scalar base) and two VGPRs — ten per CartPole
// wave-step.  An empty asm on the 32-bit value pins the extension behind it: the store takes its scalar base in the instruction's saddr
// field, `global_store_dwordx4 v_off, v[data], s[base:base+1]`, with one VGPR per offset and no address arithmetic.
__device__ __forceinline__ uint32_t pin32(uint32_t v) {
#if MXV_SADDR_STORES
    asm volatile("" : "+v"(v));
#endif
    return v;
}

// store_obs at a byte offset (a wave-uniform base + a small 32-bit lane offset: see pin32)
template <int O>
__device__ __forceinline__ void store_obs_at(char *base, uint32_t byte_off, const float *o) {
    char *q = base + byte_off;
    if constexpr (O == 4) {
        *reinterpret_cast<float4 *>(q) = make_float4(o[0], o[1], o[2], o[3]);
    } else if constexpr (O == 2) {
        *reinterpret_cast<float2 *>(q) = make_float2(o[0], o[1]);
    } else if constexpr (O == 6) {
        float2 *p = reinterpret_cast<float2 *>(q);
        p[0] = make_float2(o[0], o[1]);
        p[1] = make_float2(o[2], o[3]);
        p[2] = make_float2(o[4], o[5]);
    } else {
        float *p = reinterpret_cast<float *>(q);
#pragma unroll
        for (int k = 0; k < O; ++k) p[k] = o[k];
    }
}

// XCD-aware workgroup -> tile map.  The hardware deals workgroup ids round-robin over the 8 XCDs (id % 8), each with
// its own L2.  Handing out tiles in id order makes every XCD write 4-8 KiB crumbs interleaved with the other seven
// all over each output row; giving XCD x the x-th contiguous eighth of the tiles instead lets each L2 stream long
// contiguous runs to its memory channels.  Measured on the rollout's store pattern with the physics removed
// (profiles/r1/r01_wbench.txt; the probe lives on as mxv_write_probe): 4.7 -> 5.7 TB/s.  Works for any tile count (remainder tiles go to the
// low XCDs, matching how many ids of each residue exist).
constexpr unsigned kXcds = 8;
__device__ __forceinline__ unsigned xcd_contiguous_tile(unsigned bid, unsigned ntiles) {
#if MXV_XCD_MAP
    const unsigned x = bid % kXcds, idx = bid / kXcds;
#if MXV_XCD_BLOCK > 0
    // XCDs take turns in blocks of MXV_XCD_BLOCK tiles (tuning variant; tiles past the end are skipped by the callers' bounds)
    if (ntiles % (kXcds * MXV_XCD_BLOCK) == 0) return ((idx / MXV_XCD_BLOCK) * kXcds + x) * MXV_XCD_BLOCK + idx % MXV_XCD_BLOCK;
#endif
    const unsigned base = ntiles / kXcds, rem = ntiles % kXcds;
    return x * base + (x < rem ? x : rem) + idx;
#else
    return bid;
#endif
}

// Word `idx` (0..3, runtime) of a Philox result.
[[maybe_unused]] __device__ __forceinline__ uint32_t pick_word(const U4 &w, uint32_t idx) {
    return idx == 0 ? w.x : (idx == 1 ? w.y : (idx == 2 ? w.z : w.w));
}

// ---- the fp64 state as (float32, int32) pairs: "the observation carries the state" (mxv_adopt_obs, round 6) -------------------------------
// step(actions) moves the whole fp64 state both ways every launch — 16 S of its ~108 bytes per env-step — because fp32 state fails the
// parity bar.  For CartPole and both MountainCars the observation the step writes anyway IS float32(state), component by
// component.  With hi = float32(x) (round to nearest) the remainder r = x - hi is exact in fp64 (Sterbenz), a multiple of x's ulp and at
// most half a float32 ulp, so r / 2^(exponent(hi) - 53) is an INTEGER of magnitude <= 2^29: the pair (hi, that int32) holds x exactly, and
// the state costs 4 + 4 + 4 bytes per component per step (read hi, read lo, write lo; hi is written as the observation in any case)
// instead of 16.  What the pair cannot hold — |x| below float32's normal range (but not 0), +-Inf, NaN — escapes: lo = INT32_MIN and
// the fp64 value itself goes to the handle's ordinary state array, which stays allocated (one divergent load for values no dynamics
// produce).  The contract this buys is the caller's: the observation buffer of the previous call must come back untouched.
constexpr int32_t kHiloEscape = INT32_MIN;
// Both directions are straight-line code on the path every value takes (two conversions, one v_ldexp_f64, one add / subtract, selects);
// the escape is tested per WAVE (`__any`) and handled out of line — as branches per component the pair cost more than it saved
// (profiles/r6/r6i_*: 21.4 instead of 19.6 us per 2^20-env step with 16 bytes per env-step fewer).
__device__ __forceinline__ double hilo_decode_fast(float hi, int32_t lo) {
    const int e = (int)((__float_as_uint(hi) >> 23) & 0xffu);
    const double sum = (double)hi + ldexp((double)lo, e - 180);      // exact: the sum IS the double that was encoded
    return lo == 0 ? (double)hi : sum;                               // (keeps the sign of a zero: -0.0 + 0.0 would be +0.0)
}
__device__ __forceinline__ double hilo_decode(float hi, int32_t lo, const double *side) {
    return lo == kHiloEscape ? *side : hilo_decode_fast(hi, lo);
}
// -> the int32 half, or kHiloEscape when the pair cannot hold x (the caller then stores x itself in the fp64 array)
__device__ __forceinline__ int32_t hilo_encode_fast(double x) {
    const float hi = (float)x;
    const uint32_t e = (__float_as_uint(hi) >> 23) & 0xffu;
    const double r = x - (double)hi;                                 // exact (Sterbenz); NaN for an infinite or NaN x
    const int32_t lo = (int32_t)ldexp(r, 180 - (int)e);              // an integer of magnitude <= 2^29 whenever hi is a normal float32
    const bool holds = (e - 1u < 254u) || r == 0.0;                  // normal hi, or x itself a float32 value (zeros, float32 denormals)
    return holds ? lo : kHiloEscape;
}
__device__ __forceinline__ int32_t hilo_encode(double x, double *side) {
    const int32_t lo = hilo_encode_fast(x);
    if (lo == kHiloEscape) *side = x;
    return lo;
}
template <int O>
__device__ __forceinline__ void load_obs(const float *base, int64_t e, float *o) {
    if constexpr (O == 4) {
        const f32x4_t v = reinterpret_cast<const f32x4_t *>(base)[e];
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    } else if constexpr (O == 2) {
        const f32x2_t v = reinterpret_cast<const f32x2_t *>(base)[e];
        o[0] = v.x; o[1] = v.y;
    } else {
#pragma unroll
        for (int k = 0; k < O; ++k) o[k] = base[e * O + k];
    }
}
// the int32 halves live row-major like the observations ([N][S]: one 16- / 8-byte access per env, as wide as the observation row's)
template <int S>
__device__ __forceinline__ void load_lo(const int32_t *base, int64_t e, int32_t *v) {
    if constexpr (S == 4) {
        const i32x4_t q = reinterpret_cast<const i32x4_t *>(base)[e];
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
        const i32x2_t q = reinterpret_cast<const i32x2_t *>(base)[e];
        v[0] = q.x; v[1] = q.y;
    }
}
template <int S, bool NT = false>
__device__ __forceinline__ void store_lo(int32_t *base, int64_t e, const int32_t *v) {
    if constexpr (S == 4)
        st<NT>(reinterpret_cast<i32x4_t *>(base) + e, i32x4_t{v[0], v[1], v[2], v[3]});
    else
        st<NT>(reinterpret_cast<i32x2_t *>(base) + e, i32x2_t{v[0], v[1]});
}
template <int ENV>
__global__ void __launch_bounds__(kBlock) hilo_split_kernel(const double *state, float *hi_obs, int32_t *lo, double *side, int64_t n) {
    constexpr int S = Env<ENV>::S;
    static_assert(Env<ENV>::O == S, "the observation must be float32(state), component by component");
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= n) return;
    static_assert(S == 2 || S == 4, "row accesses of the int32 halves");
    float o[S];
    int32_t l[S];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const double x = state[(int64_t)k * n + e];
        o[k] = (float)x;
        l[k] = hilo_encode(x, side + (int64_t)k * n + e);
    }
    store_obs<S>(hi_obs, e, o);
    store_lo<S>(lo, e, l);
}
template <int ENV>
__global__ void __launch_bounds__(kBlock) hilo_join_kernel(const float *hi_obs, const int32_t *lo, double *state, int64_t n) {
    constexpr int S = Env<ENV>::S;
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= n) return;
    float o[S];
    int32_t l[S];
    load_obs<S>(hi_obs, e, o);
    load_lo<S>(lo, e, l);
#pragma unroll
    for (int k = 0; k < S; ++k) state[(int64_t)k * n + e] = hilo_decode(o[k], l[k], state + (int64_t)k * n + e);
}

// step_kernel<ENV, DEF, E, CONSEC>: a.K vector steps in ONE launch, env state held in registers between
// steps (K = 1 is the plain step() call).  Per step the only HBM traffic is the step's outputs; state
// and elapsed[] are read once at entry and written once at exit, i.e. 16*S/K + 8/K bytes per env-step.
//   CONSEC = false: lane owns envs tile0 + j*256 + tid (every access of a wave is a dense burst);
//                   Philox action words are transposed through LDS (one call = 4 consecutive envs).
//   CONSEC = true : lane owns the E consecutive envs tile0 + tid*E + j: a Philox action group is
//                   lane-private (no LDS, no barrier) and the flag bytes of a lane are contiguous.
//   CLOCK = true  : the launch advances the device clock itself (single steps with default parameters of a handle in device-clock mode,
//                   small grids).  An instantiation of its own: as a run-time branch at the exit of the one kernel it cost every
//                   launch 1.1 us per 2^20-env step (18.6 -> 19.8, profiles/r4/r4q_step_clock_tail_ab.txt).
//   HILO = true   : the state arrives as (previous observation, int32 residual) pairs and leaves the same way (see hilo_encode above):
//                   single steps with default parameters of a handle that adopted the caller's observation buffer.
template <int ENV, int DEF, int E, bool CONSEC, bool MULTI, bool CLOCK = false, bool HILO = false>
__global__ void __launch_bounds__(kBlock, MXV_MIN_WAVES) step_kernel(const StepArgs a) {
    using EV = Env<ENV>;
    constexpr int S = EV::S, O = EV::O, NA = EV::NA;
    static_assert(!HILO || (!MULTI && S == O && EV::AUX == 0), "HILO: one step, observation = float32(state)");
    constexpr int TILE = E * kBlock;
    constexpr bool NT = !MULTI && MXV_STEP_NT_STORES != 0;
    static_assert(!CONSEC || E == 1 || E == 2 || E % 4 == 0, "CONSEC needs E in {1, 2, 4k}");
    const int tid = threadIdx.x;
    const int64_t tile0 = (int64_t)xcd_contiguous_tile(blockIdx.x, gridDim.x) * TILE;
    const int64_t n = a.n;
    const Par<DEF> P(a.P, a.params_pe, a.n);
    const uint64_t t0 = a.t + (a.t_dev ? *a.t_dev : 0);
    const bool autoreset = !(a.flags & MXV_FLAG_NO_AUTORESET);
    const bool sampled = a.actions == nullptr;
    auto env_of = [&](int j) -> int64_t { return CONSEC ? tile0 + (int64_t)tid * E + j : tile0 + (int64_t)j * kBlock + tid; };

    // ---- entry: state + elapsed of the lane's E envs ----
    constexpr int AUXN = EV::AUX > 0 ? EV::AUX : 1;
    double s[E][S], aux[E][AUXN];
    int32_t el[E];
    bool valid[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int64_t e = env_of(j);
        valid[j] = e < n;
        const int64_t ec = valid[j] ? e : 0;
        if constexpr (HILO) {
            float hi[O];
            int32_t l[S];
            load_obs<O>(a.hi_in, ec, hi);
            load_lo<S>(a.lo, ec, l);
            bool esc = false;
#pragma unroll
            for (int k = 0; k < S; ++k) {
                s[j][k] = hilo_decode_fast(hi[k], l[k]);
                esc = esc || l[k] == kHiloEscape;
            }
            if (__builtin_expect(__any(esc), 0)) {   // values outside float32's normal range: the fp64 array holds them (never on a trajectory the dynamics produce)
#pragma unroll
                for (int k = 0; k < S; ++k)
                    if (l[k] == kHiloEscape) s[j][k] = a.state[(int64_t)k * n + ec];
            }
        } else {
#pragma unroll
            for (int k = 0; k < S; ++k) s[j][k] = a.state[(int64_t)k * n + ec];
        }
        el[j] = load_elapsed(a.elapsed, a.elapsed16, ec);
        EV::prime(s[j], aux[j]);
    }
    float er[E];  // running episode return (RecordEpisodeStatistics.episode_returns)
#pragma unroll
    for (int j = 0; j < E; ++j) er[j] = a.ep_acc ? a.ep_acc[valid[j] ? env_of(j) : 0] : 0.0f;
    // reset ordinals (index of each env's next draw from the reset stream); only the autoreset path needs them.  A one-step launch
    // (MULTI = false: mxv_step, the learner-in-the-loop call) reads and advances the ordinal inside the reset branch instead, i.e. only in
    // the few lanes whose env finishes: the dense 4-byte read per env-step of round 2 shrinks to the lines those lanes touch.
    constexpr bool LAZY_EP = !MULTI;
    uint32_t ep[E], ep_in[E];
#pragma unroll
    for (int j = 0; j < E; ++j) ep[j] = ep_in[j] = (autoreset && !LAZY_EP) ? a.episodes[valid[j] ? env_of(j) : 0] : 0u;
    __shared__ uint32_t sw[CONSEC ? 4 : TILE];

    const int nsteps = MULTI ? a.K : 1;  // MULTI = false: the plain step() launch, no loop-carried bookkeeping
    if (MULTI) settle_entry_loads();
    for (int step = 0; step < nsteps; ++step) {
        const uint64_t t = t0 + (uint64_t)step;
        const int64_t so = MULTI ? (int64_t)step * a.slice : 0;      // output slice offset (envs): [K][N] trajectories or 0
        const int64_t sa = MULTI ? (int64_t)step * a.act_slice : 0;  // action tape offset

        // ---- actions ----
        int ai[E];
        float af[E];
        if (sampled) {
            if constexpr (CONSEC) {
#pragma unroll
                for (int j0 = 0; j0 < E; j0 += 4) {
                    const uint64_t ge = a.env0 + (uint64_t)env_of(j0);
                    const U4 w = env_action_words<ENV>(a.action_seed, t, ge >> 2);
#pragma unroll
                    for (int q = 0; q < 4 && j0 + q < E; ++q)
                        action_from_word<ENV, DEF>(P.at(valid[j0 + q] ? env_of(j0 + q) : 0), pick_word(w, (uint32_t)((ge + q) & 3)), t, ai[j0 + q], af[j0 + q]);
                }
            } else {
                // thread c computes the 4 words of group (env0 + tile0)/4 + c; LDS hands them to the owning lanes
                if (MULTI && step > 0) __syncthreads();
                for (int c = tid; c < TILE / 4; c += kBlock) {
                    const uint64_t g = ((a.env0 + (uint64_t)tile0) >> 2) + (uint64_t)c;
                    const U4 w = env_action_words<ENV>(a.action_seed, t, g);
                    reinterpret_cast<uint4 *>(sw)[c] = make_uint4(w.x, w.y, w.z, w.w);
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < E; ++j) action_from_word<ENV, DEF>(P.at(valid[j] ? env_of(j) : 0), sw[j * kBlock + tid], t, ai[j], af[j]);
            }
            if (a.actions_out != nullptr) {
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    if (!valid[j]) continue;
                    const int64_t e = so + env_of(j);
                    if constexpr (NA > 0) {
                        if (a.flags & MXV_FLAG_ACTION_I32)
                            static_cast<int32_t *>(a.actions_out)[e] = ai[j];
                        else
                            static_cast<int64_t *>(a.actions_out)[e] = ai[j];
                    } else {
                        static_cast<float *>(a.actions_out)[e] = af[j];
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const int64_t ec = sa + (valid[j] ? env_of(j) : 0);
                if constexpr (NA > 0) {
                    const int64_t v = (a.flags & MXV_FLAG_ACTION_I32)
                                          ? (int64_t) static_cast<const int32_t *>(a.actions)[ec]
                                          : static_cast<const int64_t *>(a.actions)[ec];
                    // Discrete.contains (cartpole.py:131-132): out of range -> latch, leave the env untouched
                    if (valid[j] && (v < 0 || v >= NA)) {
                        raise_error(a.err, 1);
                        valid[j] = false;
                    }
                    ai[j] = (int)v;
                    af[j] = 0.0f;
                } else {
                    ai[j] = 0;
                    af[j] = static_cast<const float *>(a.actions)[ec];
                }
            }
        }

        if constexpr (ENV == MXV_ACROBOT && DEF != PM_DEFAULT) {  // torque noise: one draw per env-step when the attribute is set
            if (a.step_noise) {
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    const int64_t e = valid[j] ? env_of(j) : 0;
                    const uint64_t seed = a.seeds ? landed(a.seeds[e]) : a.base_seed + a.env0 + (uint64_t)e;
                    af[j] = __uint_as_float(step_noise_word(seed, t));
                }
            }
        }

        // ---- dynamics + TimeLimit, E independent chains ----
        float obs[E][O];
        double rew[E];
        bool term[E], trunc[E], pend[E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            term[j] = EV::template step<DEF>(P.at(valid[j] ? env_of(j) : 0), s[j], aux[j], el[j] == 0, ai[j], af[j], rew[j], obs[j]);
            el[j] += 1;                                              // time_limit.py:51
            trunc[j] = (a.max_steps > 0) && (el[j] >= a.max_steps);  // time_limit.py:53-54
            pend[j] = autoreset && (term[j] || trunc[j]);
        }
        if (a.ep_acc != nullptr) {  // record_episode_statistics.py:119-143
#pragma unroll
            for (int j = 0; j < E; ++j) {
                er[j] = (float)((double)er[j] + rew[j]);  // float32 array += float64 rewards
                if (valid[j] && (term[j] || trunc[j])) {
                    if (a.ep_return_out) a.ep_return_out[so + env_of(j)] = er[j];
                    if (a.ep_length_out) a.ep_length_out[so + env_of(j)] = el[j];
                    er[j] = 0.0f;
                }
            }
        }

        // ---- autoreset (sync_vector_env.py:152-156), compacted: every pass each lane resets its first
        // pending env, so a wave spends max-over-lanes(#finished) Philox calls, not E ----
        while (true) {
            int jsel = -1;
#pragma unroll
            for (int j = E - 1; j >= 0; --j)
                if (pend[j]) jsel = j;
            if (!__any(jsel >= 0)) break;
            if (jsel >= 0) {
                const int64_t e = CONSEC ? tile0 + (int64_t)tid * E + jsel : tile0 + (int64_t)jsel * kBlock + tid;
                bool v = false;
                uint32_t k_reset = 0;
                float cur[O];
#pragma unroll
                for (int j = 0; j < E; ++j)
                    if (j == jsel) {
                        v = valid[j];
                        k_reset = ep[j];
#pragma unroll
                        for (int k = 0; k < O; ++k) cur[k] = obs[j][k];
                    }
                if (a.final_obs != nullptr && v) store_obs<O>(a.final_obs, so + e, cur);  // info["final_observation"]
                if constexpr (LAZY_EP) {
                    k_reset = landed(a.episodes[v ? e : 0]);
                    if (v) a.episodes[e] = k_reset + 1;
                }
                const uint64_t seed = a.seeds ? landed(a.seeds[v ? e : 0]) : a.base_seed + a.env0 + (uint64_t)e;
                const U4 w = episode_reset_words(seed, k_reset);
                double ns[S], naux[AUXN];
                float nobs[O];
                EV::reset(w, a.b0, a.b1, ns);
                EV::observe(ns, nobs, naux);
#pragma unroll
                for (int j = 0; j < E; ++j)
                    if (j == jsel) {
#pragma unroll
                        for (int k = 0; k < EV::AUX; ++k) aux[j][k] = naux[k];
#pragma unroll
                        for (int k = 0; k < S; ++k) s[j][k] = ns[k];
#pragma unroll
                        for (int k = 0; k < O; ++k) obs[j][k] = nobs[k];
                        el[j] = 0;  // time_limit.py:67
                        if constexpr (!LAZY_EP) ep[j] += 1;
                        pend[j] = false;
                    }
            }
        }

        // ---- this step's outputs ----
#pragma unroll
        for (int j = 0; j < E; ++j) {
            if (!valid[j]) continue;
            const int64_t e = so + env_of(j);
            store_obs<O, NT>(a.obs, e, obs[j]);
            if (a.reward != nullptr) {
                if (a.flags & MXV_FLAG_REWARD_F32)
                    st<NT>(static_cast<float *>(a.reward) + e, (float)rew[j]);
                else
                    st<NT>(static_cast<double *>(a.reward) + e, rew[j]);
            }
            if (a.terminated != nullptr) st<NT>(a.terminated + e, (uint8_t)(term[j] ? 1 : 0));
            if (a.truncated != nullptr) st<NT>(a.truncated + e, (uint8_t)(trunc[j] ? 1 : 0));
        }
        if constexpr (ENV == MXV_CARTPOLE) {
            // steps_beyond_terminated (cartpole.py:169-184): an env that is stepped on after it terminated — only possible without
            // autoreset, which is when the marks exist — pays 1.0 in the step the pole falls and 0.0 in every later step that is (still)
            // terminated.  Done BEHIND the step's stores, as a correction of the reward just written (same lane, same address: the later
            // store stands): a wave-uniform test of the pointer keeps every other launch off this code, and no block boundary cuts
            // through the step's arithmetic.  (In the middle of the dynamics loop the same lines cost step(actions) 22.1 -> 23.3 us
            // per 2^20-env step although they never ran: profiles/r4/r4d_step_path_ab.json.)
            if (MXV_CARTPOLE_BEYOND && a.beyond != nullptr) {
#pragma unroll
                for (int j = 0; j < E; ++j)
                    if (valid[j] && term[j]) {
                        uint8_t *mark = a.beyond + env_of(j);
                        if (landed((uint32_t)*mark) == 0u) {
                            *mark = 1;
                        } else if (a.reward != nullptr) {
                            if (a.flags & MXV_FLAG_REWARD_F32)
                                static_cast<float *>(a.reward)[so + env_of(j)] = 0.0f;
                            else
                                static_cast<double *>(a.reward)[so + env_of(j)] = 0.0;
                        }
                    }
            }
        }
    }

    // ---- exit: state + elapsed back to HBM ----
#pragma unroll
    for (int j = 0; j < E; ++j) {
        if (!valid[j]) continue;
        const int64_t e = env_of(j);
        if constexpr (HILO) {   // (the float32 half is the observation this step stored above)
            int32_t l[S];
            bool esc = false;
#pragma unroll
            for (int k = 0; k < S; ++k) {
                l[k] = hilo_encode_fast(s[j][k]);
                esc = esc || l[k] == kHiloEscape;
            }
            if (__builtin_expect(__any(esc), 0)) {
#pragma unroll
                for (int k = 0; k < S; ++k)
                    if (l[k] == kHiloEscape) a.state[(int64_t)k * n + e] = s[j][k];
            }
            store_lo<S, NT>(a.lo, e, l);
        } else {
#pragma unroll
            for (int k = 0; k < S; ++k) st<NT>(a.state + (int64_t)k * n + e, s[j][k]);
        }
        store_elapsed<NT>(a.elapsed, a.elapsed16, e, el[j]);
        if (ep[j] != ep_in[j]) a.episodes[e] = ep[j];
        if (a.ep_acc) a.ep_acc[e] = er[j];
    }
    // ---- device clock (mxv_set_device_clock, small grids): the LAST workgroup to get here advances the step index, so that a launch
    // recorded in a caller's hipGraph needs no second kernel behind it.  Every workgroup read *t_dev before it took its ticket, and
    // the next launch of the stream starts after this one has finished.
    if constexpr (CLOCK) {
        if (tid == 0) {
            const uint32_t ticket = atomicAdd(a.clock_ticket, 1u);
            if (ticket == gridDim.x - 1) {
                *a.clock_ticket = 0u;
                *a.clock_out = t0 - a.t + (uint64_t)a.K;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// rollout_kernel_v3<ENV, DEF, E, SAFE, OUT>: the sampled-action + autoreset fast path (mxv_rollout FUSED), a.K vector steps
// per launch with the env state in registers.
//
// Workgroup = ONE wave64 owning a tile of E*64 consecutive envs (lane L owns envs tile0 + j*64 + L): no s_barrier anywhere;
// every global access of a wave is a dense, line-aligned burst.  Output addressing: the per-step base of every array is a
// scalar; lanes add a 32-bit byte offset fixed for the whole launch, so the loop holds no 64-bit vector address arithmetic.
//
// What the RNG contract (include/mxv.h) buys here: nothing random sits on the per-step critical path.
//   * Resets.  The draw of an env's NEXT reset is a function of (its seed, its reset ordinal), known as soon as the current
//     episode starts.  Every lane keeps, per env slot, a ready-made reset entry — new fp64 state, carried aux values and the
//     float32 observation — in a lane-private LDS slot (LDS as a register spill area: no other lane reads it).  An env that
//     finishes step t copies its entry (three ds_read_b128 for CartPole) and marks the slot empty; empty slots are refilled by
//     look-ahead passes, one Philox call of the lanes that need one, every MXV_ROLLOUT_PASS_PERIOD steps per slot.  CartPole
//     (~6 of a wave's 128 envs finish per step): 0.25 masked Philox calls + reset arithmetic per wave-step instead of 1.0;
//     envs that only truncate (Pendulum at step 200, ...): one call per episode.  An env that finishes again before its
//     slot was refilled (TimeLimit of a few steps) forces the pass early — any schedule gives the same bits, the words
//     are pure functions of (seed, ordinal).
//   * Actions.  Discrete(2): one random BIT per step; a full-wave Philox call (64 lanes x 4 words) holds the bits of
//     64/(16E) blocks of 32 steps for the tile: one call per 64 (E = 2) steps.  Discrete(3) / Box: one word per step, a
//     full-wave call serves 64/(16E) steps; the words wait in an LDS ring and the next step's word is read a step ahead.
// ------------------------------------------------------------------------------------------------------------
constexpr int kWave = 64;
constexpr int kSimds = 1024;  // 256 CUs x 4 SIMDs

template <int S, int O, int AUX>
struct alignas(16) ResetEntry {
    double s[S];
    double x[AUX];  // Env::AUX values carried across steps
    float o[O];
};
template <int S, int O>
struct alignas(16) ResetEntry<S, O, 0> {  // no carried values: CartPole's entry is 48 B, 6 KiB of LDS per workgroup
    double s[S];
    float o[O];
};

// Waves per SIMD the register allocator must leave room for.  2^20 envs at two envs per lane are 8192 single-wave workgroups
// = exactly two rounds of 4 waves per SIMD: 128 VGPRs is the budget of the default-parameter kernels (the few values the
// allocator then parks in scratch are stored before and reloaded after the K-step loop, never inside it; checked in the ISA).
// The rarely launched instantiations (runtime parameters; CartPole right after a state injection) keep the allocator's choice.
template <int ENV, bool DEF, bool SAFE>
constexpr int rollout_min_waves() {
    if (MXV_ROLLOUT_MIN_WAVES > 1) return MXV_ROLLOUT_MIN_WAVES;
    return (!DEF || (ENV == MXV_CARTPOLE && SAFE)) ? 1 : 4;
}

// ---- wave-level sums of V values per lane (the fused batch moments of NormalizeObservation, STATS launches) ------------------------
// V per-lane doubles -> V wave totals in one pass of a fixed binary tree over the lane index.  Bits 0, 1 (and 3 for V = 8) HALVE the
// value set: a lane and its partner split the values between them — each keeps the half whose index bit equals its lane bit and adds
// the partner's copy of that half — so after log2(V) stages every lane holds ONE value, the sum over its 2^log2(V) partners; the
// remaining lane bits (2, 4, 5; and 3 for V = 4) are plain butterfly stages.  V + V/2 + ... additions instead of 6 V, and the tree is
// the same for every launch shape: totals are bit-reproducible and do not depend on how a vector env is sharded.
template <int CTRL>
__device__ __forceinline__ double stat_dpp(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double stat_xor4(double v) {  // lane ^ 4 through the swizzle crossbar (bit-mask mode: and 0x1f, or 0, xor 4)
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x101F), hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x101F);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int N>
__device__ __forceinline__ void stat_halve(double (&v)[N], bool bit) {  // N values -> N / 2 in v[0 .. N/2)
#pragma unroll
    for (int p = 0; p < N / 2; ++p) {
        const double lo = v[2 * p], hi = v[2 * p + 1];
        v[p] = (bit ? hi : lo) + stat_dpp<CTRL>(bit ? lo : hi);
    }
}
// returns the lane's total; the total of value i sits in the lanes with (lane & 3) | (lane >> 3 & 1) << 2 == i (V = 8) resp. (lane & 3) == i
template <int V>
__device__ __forceinline__ double wave_sums(double (&v)[V], uint32_t lane) {
    static_assert(V == 4 || V == 8, "batches of 4 or 8 values");
    stat_halve<0xB1, V>(v, (lane & 1u) != 0);             // quad_perm [1, 0, 3, 2]
    if constexpr (V == 8) {
        double w[4] = {v[0], v[1], v[2], v[3]};
        stat_halve<0x4E, 4>(w, (lane & 2u) != 0);         // quad_perm [2, 3, 0, 1]
        double u[2] = {w[0], w[1]};
        stat_halve<0x128, 2>(u, (lane & 8u) != 0);        // row_ror:8 = lane ^ 8
        v[0] = u[0];
    } else {
        double w[2] = {v[0], v[1]};
        stat_halve<0x4E, 2>(w, (lane & 2u) != 0);
        v[0] = w[0] + stat_dpp<0x128>(w[0]);              // lane ^ 8
    }
    double x = v[0];
    x += stat_xor4(x);
    {
        const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        x = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
    {
        const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        x = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
    return x;
}
// One batch: `count` (<= V) values summed over the wave, value i stored to dst[i] by the lane that holds it.
template <int V>
__device__ __forceinline__ void wave_sums_store(double (&v)[V], uint32_t lane, double *dst, int count) {
    const double x = wave_sums<V>(v, lane);
    const uint32_t idx = V == 8 ? ((lane & 3u) | ((lane >> 1) & 4u)) : (lane & 3u);
    const bool holder = V == 8 ? (lane & 0x34u) == 0 : (lane & 0x3Cu) == 0;
    if (holder && (int)idx < count) dst[idx] = x;
}

// LDS of one rollout workgroup (= one wave): the ring of action words and the lane-private reset entries.
template <int ENV, int E>
struct RolloutLds {
    static constexpr int TILE = E * kWave;
    static constexpr int H = kWave / (TILE / 4);
    using Entry = ResetEntry<Env<ENV>::S, Env<ENV>::O, Env<ENV>::AUX>;
    uint32_t act[H * TILE];  // slot (q % H) holds the action words of unit u0 + q
    Entry res[TILE];         // lane-private: the ready-made next reset of env slot j * 64 + lane
};

// The kernel body as a device function of (arguments, workgroup index, workgroups of this segment, LDS): rollout_kernel_v3 runs
// it for one homogeneous vector env, mixed_rollout_kernel for the segment a workgroup belongs to.
template <int ENV, bool DEF, int E, bool SAFE, int OUT, bool TAPE = false, int STATS = 0>
__device__ __forceinline__ void rollout_body_v3(const StepArgs &a, const unsigned bid, const unsigned nblk, RolloutLds<ENV, E> &lds) {
    using EV = Env<ENV>;
    constexpr int S = EV::S, O = EV::O, NA = EV::NA;
    constexpr int TILE = E * kWave;
    constexpr int NACT = TILE / 4;        // lanes that draw the action words of ONE unit (step, or 32-step block) for the tile
    constexpr int H = kWave / NACT;       // units one full-wave call produces = slots of the LDS ring
    constexpr int SH = action_unit_shift<ENV>();
    constexpr int PERIOD = MXV_ROLLOUT_PASS_PERIOD;
    static_assert(NACT <= kWave && (PERIOD & (PERIOD - 1)) == 0 && PERIOD >= E, "E <= 4; pass period a power of two >= E");
    constexpr int AUXN = EV::AUX > 0 ? EV::AUX : 1;
    using Entry = ResetEntry<S, O, EV::AUX>;
    uint32_t *const lds_act = lds.act;
    Entry *const lds_res = lds.res;

    const int lane = threadIdx.x;
    const uint32_t tile = xcd_contiguous_tile(bid, nblk);
    const int64_t tile0 = (int64_t)tile * TILE;
    const int64_t n = a.n;
    const Par<DEF> P(a.P);
    const uint64_t t0 = a.t + (a.t_dev ? *a.t_dev : 0);
    // OUT != 0: every per-step output array is present, no final_obs, no episode statistics, dtypes fixed (1: float64 rewards
    // + int64 actions, 2: float32 + int32) — the trajectory-recording launch.  The ~25 wave-uniform branches and ~60 scalar
    // instructions per step that the optional outputs cost fold away at compile time.
    constexpr bool FULL = OUT != 0;
    const bool act_i32 = FULL ? (OUT == 2) : ((a.flags & MXV_FLAG_ACTION_I32) != 0);
    const bool rew_f32 = FULL ? (OUT == 2) : ((a.flags & MXV_FLAG_REWARD_F32) != 0);
    const bool ep_on = !FULL && a.ep_acc != nullptr;
    const uint64_t group0 = (a.env0 + (uint64_t)tile0) >> 2;

    double s[E][S], aux[E][AUXN];
    int32_t el[E];
    uint32_t ep[E];  // reset ordinal = index of the env's next draw from the reset stream
    bool valid[E];
    uint32_t le[E];  // env index inside the shard (fits 32 bits: mxv_create caps num_envs)
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int64_t e = tile0 + j * kWave + lane;
        valid[j] = e < n;
        le[j] = (uint32_t)(valid[j] ? e : 0);
#pragma unroll
        for (int k = 0; k < S; ++k) s[j][k] = a.state[(int64_t)k * n + le[j]];
        el[j] = load_elapsed(a.elapsed, a.elapsed16, le[j]);
        ep[j] = a.episodes[le[j]];
        EV::template prime<SAFE>(s[j], aux[j]);
    }
    float er[E];  // running episode return (RecordEpisodeStatistics.episode_returns)
#pragma unroll
    for (int j = 0; j < E; ++j) er[j] = ep_on ? a.ep_acc[le[j]] : 0.0f;
    float *p_epr = FULL ? nullptr : a.ep_return_out;
    int32_t *p_epl = FULL ? nullptr : a.ep_length_out;

    // ---- action words: units u0 .. u0 + H - 1 now, the following H units whenever the ring runs dry ----
    const uint64_t u0 = t0 >> SH;
    auto draw_units = [&](uint64_t q) {  // all 64 lanes: lane L draws group L % NACT of unit u0 + q + L / NACT
        const U4 w = philox4x32_10(action_unit_counter<ENV>(u0 + q + (uint64_t)(lane / NACT), group0 + (uint64_t)(lane % NACT)),
                                   (uint32_t)a.action_seed, (uint32_t)(a.action_seed >> 32));
        reinterpret_cast<uint4 *>(lds_act)[lane] = make_uint4(w.x, w.y, w.z, w.w);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };
    uint32_t word[E];
    if constexpr (!TAPE) {
        draw_units(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < E; ++j) word[j] = lds_act[j * kWave + lane];
    }
    // TAPE: the actions come from the caller's [K][act_slice] tape (mxv_rollout_tape) instead of the Philox stream.  A load inside
    // the loop shares the in-order vmcnt counter with the stores: consumed in the step that issues it (step_kernel's way), it
    // drains every store of the wave once per step — 12.5 instead of 5.7 us per 2^20-env CartPole step.  So the tape is read ONE
    // STEPS AHEAD into registers (two register sets, the loop unrolled by two), and the loop is arranged so that the compiler can
    // wait for that load with vmcnt(N > 0) — "everything older than the last two steps' stores" — instead of vmcnt(0): the load
    // is unconditional (row clamped to the last one; a conditional load goes through temporaries and a copy behind s_waitcnt
    // vmcnt(0) at the join) and no store of a step sits behind an exec-mask branch (ALLV below).  One step ahead is not enough
    // for the light kernels: a read issued into write-saturated HBM takes longer than their 1.5-us wave-step.
    const uint32_t tape_b = NA > 0 ? (act_i32 ? 4u : 8u) : 4u;
    const char *const p_tape = reinterpret_cast<const char *>(a.actions);
    struct TapeRegs {
        int64_t i[E];
        float f[E];
    } tape_even, tape_odd;  // the tape rows of the next even / odd step
    auto load_tape = [&](int step, TapeRegs &r) {
        const char *base = p_tape + (int64_t)min(step, a.K - 1) * a.act_slice * (int64_t)tape_b;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const char *q = base + le[j] * tape_b;
            if constexpr (NA > 0)
                r.i[j] = act_i32 ? (int64_t) * reinterpret_cast<const int32_t *>(q) : *reinterpret_cast<const int64_t *>(q);
            else
                r.f[j] = *reinterpret_cast<const float *>(q);
        }
    };
    bool bad[E];  // TAPE: the tape held an action outside [0, NA) for this env: error latched, state never written back
#pragma unroll
    for (int j = 0; j < E; ++j) bad[j] = false;
    if constexpr (TAPE) {
        load_tape(0, tape_even);
        load_tape(1, tape_odd);
    }

    // ---- reset entries: need[j] = this lane's slot j holds no entry (an i1 per lane: lives in an SGPR pair) ----
    bool need[E];
    auto fill_entries = [&](int j) {  // one Philox call of the lanes whose slot j is empty
        if (need[j]) {
            const uint64_t seed = a.seeds ? landed(a.seeds[le[j]]) : a.base_seed + a.env0 + (uint64_t)le[j];
            const U4 w = episode_reset_words(seed, ep[j]);
            Entry r;
            EV::reset(w, a.b0, a.b1, r.s);  // autoresets draw inside the env's default bounds: arguments far inside the unguarded range
            if constexpr (EV::AUX > 0) {
                EV::template observe<false>(r.s, r.o, r.x);
            } else {
                double none[1];
                EV::template observe<false>(r.s, r.o, none);
            }
            lds_res[j * kWave + lane] = r;
        }
        need[j] = false;
    };
#pragma unroll
    for (int j = 0; j < E; ++j) {
        need[j] = valid[j];
        fill_entries(j);
    }

    // per-step output bases (scalars)
    char *p_obs = reinterpret_cast<char *>(a.obs);
    char *p_rew = reinterpret_cast<char *>(a.reward);
    char *p_act = reinterpret_cast<char *>(a.actions_out);
    char *p_term = reinterpret_cast<char *>(a.terminated);
    char *p_trunc = reinterpret_cast<char *>(a.truncated);
    char *p_fin = FULL ? nullptr : reinterpret_cast<char *>(a.final_obs);
    // STATS bit 0: per-tile column sums of the observations ([K][tiles][2 O]); bit 1: the discounted returns of NormalizeReward
    // (normalize.py:132-136) advanced in registers, their per-tile sum and sum of squares per step ([K][tiles][2]); tile = leaf index
    [[maybe_unused]] double *p_part = (STATS & 1) ? a.obs_part + (int64_t)tile * (2 * O) : nullptr;
    [[maybe_unused]] double *p_rpart = (STATS & 2) ? a.ret_part + (int64_t)tile * 2 : nullptr;
    [[maybe_unused]] double ret[E], ret_s = 0.0, ret_q = 0.0;
#pragma unroll
    for (int j = 0; j < E; ++j) ret[j] = ((STATS & 2) && valid[j]) ? a.ret_state[le[j]] : 0.0;
    const uint32_t rew_b = rew_f32 ? 4u : 8u;
    const uint32_t act_b = (NA > 0 && !act_i32) ? 8u : 4u;
    uint32_t lo[E];  // index of the lane's env slot inside one step's slice of every output array
#if MXV_SADDR_STORES
    // every output base is moved to this wave's tile (wave-uniform: scalar arithmetic), lanes keep an offset below E * 64 elements: the
    // byte offsets fit 32 bits whatever the shard size, which is what lets the stores use scalar-base addressing (pin32)
    const int64_t slice = a.slice;
#pragma unroll
    for (int j = 0; j < E; ++j) lo[j] = (uint32_t)(j * kWave + lane);
    p_obs += tile0 * (int64_t)(O * sizeof(float));
    if (p_rew) p_rew += tile0 * (int64_t)rew_b;
    if (p_act) p_act += tile0 * (int64_t)act_b;
    if (p_term) p_term += tile0;
    if (p_trunc) p_trunc += tile0;
    if (p_fin) p_fin += tile0 * (int64_t)(O * sizeof(float));
    if (p_epr) p_epr += tile0;
    if (p_epl) p_epl += tile0;
#else
    const int64_t slice = a.slice;
#pragma unroll
    for (int j = 0; j < E; ++j) lo[j] = le[j];
#endif


    settle_entry_loads();
    // The loop exists twice in a tape-driven kernel: ALLV = every env slot of the wave is a real env (all tiles but possibly the
    // last), so no store sits behind an exec-mask branch — the compiler can then prove how many stores follow a tape load on
    // every path and waits for the load with s_waitcnt vmcnt(N > 0) instead of draining the wave's stores.
    auto one_step = [&](const int step, auto allv_tag, TapeRegs &tape) __attribute__((always_inline)) {
        constexpr bool ALLV = decltype(allv_tag)::value;
        const uint64_t t = t0 + (uint64_t)step;

        // ---- this step's actions ----
        int ai[E];
        float af[E];
        if constexpr (TAPE) {
#pragma unroll
            for (int j = 0; j < E; ++j) {
                if constexpr (NA > 0) {
                    const int64_t v = tape.i[j];
                    const bool oob = v < 0 || v >= NA;
                    if ((ALLV || valid[j]) && oob && !bad[j]) {  // Discrete.contains (cartpole.py:131-132): latch the error; the env's
                        raise_error(a.err, 1);                      // state is never written back (its outputs are garbage from here on)
                        bad[j] = true;
                    }
                    ai[j] = oob ? 0 : (int)v;
                    af[j] = 0.0f;
                } else {
                    ai[j] = 0;
                    af[j] = tape.f[j];
                }
            }
            load_tape(step + 2, tape);
        } else {
#pragma unroll
            for (int j = 0; j < E; ++j) action_from_word<ENV, DEF>(P, word[j], t, ai[j], af[j]);
        }
        if (TAPE ? (!FULL && p_act != nullptr) : (FULL || p_act != nullptr)) {  // a tape-driven trajectory launch records no actions
#pragma unroll
            for (int j = 0; j < E; ++j) {
                if (!ALLV && !valid[j]) continue;
                char *q = p_act + pin32(lo[j] * act_b);
                if constexpr (NA > 0) {
                    if (act_i32)
                        *reinterpret_cast<int32_t *>(q) = ai[j];
                    else
                        *reinterpret_cast<int64_t *>(q) = (int64_t)ai[j];
                } else {
                    *reinterpret_cast<float *>(q) = af[j];
                }
            }
        }

        // ---- dynamics + TimeLimit, E independent chains ----
        float obs[E][O];
        double rew[E];
        bool term[E], trunc[E], pend[E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            term[j] = EV::template step<DEF, SAFE, E>(P, s[j], aux[j], el[j] == 0, ai[j], af[j], rew[j], obs[j]);
            el[j] += 1;                                              // time_limit.py:51
            trunc[j] = (a.max_steps > 0) && (el[j] >= a.max_steps);  // time_limit.py:53-54
            pend[j] = (ALLV || valid[j]) && (term[j] || trunc[j]);
        }
        if constexpr ((STATS & 2) != 0) {  // returns = returns * gamma + rews; (sums for return_rms.update); returns[dones] = 0
            ret_s = ret_q = 0.0;
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const double r = rew_f32 ? (double)(float)rew[j] : rew[j];   // what NormalizeReward reads from the reward tensor
                ret[j] = ret[j] * a.ret_gamma + r;
                if (ALLV || valid[j]) {
                    ret_s += ret[j];
                    ret_q = __fma_rn(ret[j], ret[j], ret_q);
                }
                if (term[j] || trunc[j]) ret[j] = 0.0;
            }
        }
        if (ep_on) {  // record_episode_statistics.py:119-143
#pragma unroll
            for (int j = 0; j < E; ++j) {
                er[j] = (float)((double)er[j] + rew[j]);  // float32 array += float64 rewards
                if (pend[j]) {
                    if (p_epr) *reinterpret_cast<float *>(reinterpret_cast<char *>(p_epr) + pin32(lo[j] * 4u)) = er[j];
                    if (p_epl) *reinterpret_cast<int32_t *>(reinterpret_cast<char *>(p_epl) + pin32(lo[j] * 4u)) = el[j];
                    er[j] = 0.0f;
                }
            }
        }
        // outputs that do not depend on the reset go out first: reward, flags, info["final_observation"]
#pragma unroll
        for (int j = 0; j < E; ++j) {
            if (!ALLV && !valid[j]) continue;
            if (FULL || p_rew != nullptr) {
                char *q = p_rew + pin32(lo[j] * rew_b);
                if (rew_f32)
                    *reinterpret_cast<float *>(q) = (float)rew[j];
                else
                    *reinterpret_cast<double *>(q) = rew[j];
            }
            if (FULL || p_term != nullptr) *reinterpret_cast<uint8_t *>(p_term + pin32(lo[j])) = term[j] ? 1 : 0;
            if (FULL || p_trunc != nullptr) *reinterpret_cast<uint8_t *>(p_trunc + pin32(lo[j])) = trunc[j] ? 1 : 0;
            if (!FULL && pend[j] && p_fin != nullptr) store_obs_at<O>(p_fin, pin32(lo[j] * (uint32_t)(O * sizeof(float))), obs[j]);
        }

        // ---- autoreset (sync_vector_env.py:152-156): finished envs take their ready-made entry ----
#pragma unroll
        for (int j = 0; j < E; ++j) {
            if (__any(pend[j])) {
                if (__any(pend[j] && need[j])) fill_entries(j);  // rare: finished again before the look-ahead pass came round
                if (pend[j]) {
                    const Entry r = lds_res[j * kWave + lane];
#pragma unroll
                    for (int k = 0; k < S; ++k) s[j][k] = r.s[k];
                    if constexpr (EV::AUX > 0) {
#pragma unroll
                        for (int k = 0; k < EV::AUX; ++k) aux[j][k] = r.x[k];
                    }
#pragma unroll
                    for (int k = 0; k < O; ++k) obs[j][k] = r.o[k];
                    el[j] = 0;  // time_limit.py:67
                    ep[j] += 1;
                    need[j] = true;
                }
            }
        }

        // ---- next step's action words (LDS latency hides behind the observation stores) ----
        if (!TAPE && step + 1 < a.K) {
            const uint64_t q0 = (t >> SH) - u0, q1 = ((t + 1) >> SH) - u0;
            if (SH == 0 || q1 != q0) {
                if (q1 % H == 0) draw_units(q1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int j = 0; j < E; ++j) word[j] = lds_act[(uint32_t)(q1 % H) * TILE + j * kWave + lane];
            }
        }
#pragma unroll
        for (int j = 0; j < E; ++j)
            if (ALLV || valid[j]) store_obs_at<O>(p_obs, pin32(lo[j] * (uint32_t)(O * sizeof(float))), obs[j]);
        // ---- STATS: this tile's column sums and sums of squares of the observations just stored (NormalizeObservation's batch
        //      moments, gym/wrappers/normalize.py:17-29, as partials [K][tiles][2 O] for mxv_norm's tree) — the pass that would read
        //      them back from HBM is not launched
        if constexpr ((STATS & 1) != 0) {
            double sm[O], sq[O];
#pragma unroll
            for (int k = 0; k < O; ++k) sm[k] = sq[k] = 0.0;
#pragma unroll
            for (int j = 0; j < E; ++j) {
#pragma unroll
                for (int k = 0; k < O; ++k) {
                    const double x = (ALLV || valid[j]) ? (double)obs[j][k] : 0.0;
                    sm[k] += x;
                    sq[k] = __fma_rn(x, x, sq[k]);  // x * x is exact in fp64
                }
            }
            if constexpr (2 * O <= 8) {
                constexpr int V = 2 * O <= 4 ? 4 : 8;
                double v[V];
#pragma unroll
                for (int k = 0; k < V; ++k) v[k] = 0.0;
#pragma unroll
                for (int k = 0; k < O; ++k) {
                    v[k] = sm[k];
                    v[O + k] = sq[k];
                }
                wave_sums_store<V>(v, (uint32_t)lane, p_part, 2 * O);
            } else {
                static_assert(O <= 8, "two batches of at most 8 values");
                double v[8], w[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = w[k] = 0.0;
#pragma unroll
                for (int k = 0; k < O; ++k) {
                    v[k] = sm[k];
                    w[k] = sq[k];
                }
                wave_sums_store<8>(v, (uint32_t)lane, p_part, O);
                wave_sums_store<8>(w, (uint32_t)lane, p_part + O, O);
            }
            p_part += (int64_t)nblk * (2 * O);
        }
        if constexpr ((STATS & 2) != 0) {
            double v[4] = {ret_s, ret_q, 0.0, 0.0};
            wave_sums_store<4>(v, (uint32_t)lane, p_rpart, 2);
            p_rpart += (int64_t)nblk * 2;
        }
        // ---- the chunk's FINAL tensors once more, into the caller's snapshot (what a sharded vector env all-gathers while the
        //      next chunk runs: written here, no copy kernels between rollout and gather) ----
        if (step + 1 == a.K && a.snap_obs != nullptr) {
#pragma unroll
            for (int j = 0; j < E; ++j) {
                if (!ALLV && !valid[j]) continue;
                store_obs<O>(a.snap_obs, le[j], obs[j]);
                if (a.snap_reward != nullptr) {
                    if (rew_f32)
                        static_cast<float *>(a.snap_reward)[le[j]] = (float)rew[j];
                    else
                        static_cast<double *>(a.snap_reward)[le[j]] = rew[j];
                }
                if (a.snap_terminated != nullptr) a.snap_terminated[le[j]] = term[j] ? 1 : 0;
                if (a.snap_truncated != nullptr) a.snap_truncated[le[j]] = trunc[j] ? 1 : 0;
            }
        }

        // ---- look-ahead pass: refill the empty reset slots j of this wave, every PERIOD steps per slot ----
#pragma unroll
        for (int j = 0; j < E; ++j)
            if ((step & (PERIOD - 1)) == j * (PERIOD / E) && __any(need[j])) fill_entries(j);

        // ---- advance the scalar output bases to the next trajectory slice ----
        p_obs += slice * (int64_t)(O * sizeof(float));
        if (FULL || p_rew != nullptr) p_rew += slice * (int64_t)rew_b;
        if (TAPE ? (!FULL && p_act != nullptr) : (FULL || p_act != nullptr)) p_act += slice * (int64_t)act_b;
        if (FULL || p_term != nullptr) p_term += slice;
        if (FULL || p_trunc != nullptr) p_trunc += slice;
        if (!FULL && p_fin != nullptr) p_fin += slice * (int64_t)(O * sizeof(float));
        if (!FULL && p_epr != nullptr) p_epr += slice;
        if (!FULL && p_epl != nullptr) p_epl += slice;
    };
    if constexpr (TAPE) {
        // two steps per iteration, each with its own tape registers: a row requested at the top of step s is consumed at the top
        // of step s + 2, with two steps' stores provably issued in between
        auto run_pairs = [&](auto allv_tag) __attribute__((always_inline)) {
            int step = 0;
            for (; step + 1 < a.K; step += 2) {
                one_step(step, allv_tag, tape_even);
                one_step(step + 1, allv_tag, tape_odd);
            }
            if (step < a.K) one_step(step, allv_tag, tape_even);
        };
        bool allv = true;
#pragma unroll
        for (int j = 0; j < E; ++j) allv = allv && __all(valid[j]);
        if (allv)
            run_pairs(std::true_type{});
        else
            run_pairs(std::false_type{});
    } else {
        for (int step = 0; step < a.K; ++step) one_step(step, std::false_type{}, tape_even);
    }

#pragma unroll
    for (int j = 0; j < E; ++j) {
        if (!valid[j] || bad[j]) continue;
#pragma unroll
        for (int k = 0; k < S; ++k) a.state[(int64_t)k * n + le[j]] = s[j][k];
        store_elapsed(a.elapsed, a.elapsed16, le[j], el[j]);
        a.episodes[le[j]] = ep[j];
        if (ep_on) a.ep_acc[le[j]] = er[j];
        if constexpr ((STATS & 2) != 0) a.ret_state[le[j]] = ret[j];
    }
}


// ... and the waves per SIMD it may NOT exceed.  The default-parameter kernels are pinned to exactly 4: a shard of 2^19 / 2^20
// envs is a whole number of 4-wave rounds (8192 or 16384 single-wave workgroups on 1024 SIMDs), and a kernel that fits 5 waves
// (Pendulum: 87 VGPRs) runs 3.2 rounds' worth of work in 4 rounds, the last one a fifth full: measured 6.4 instead of 5.9 us per
// 2^20-env Pendulum step when the medium-range sincos shrank the kernel to 5 waves (profiles/r2/r02f_fast_trig_ab.jsonl).
template <int ENV, bool DEF, bool SAFE>
constexpr int rollout_max_waves() {
    return rollout_min_waves<ENV, DEF, SAFE>() == 1 ? 8 : rollout_min_waves<ENV, DEF, SAFE>();
}

template <int ENV, bool DEF, int E, bool SAFE, int OUT = 0, bool TAPE = false, int STATS = 0>
__global__ void __launch_bounds__(kWave)
    __attribute__((amdgpu_waves_per_eu(rollout_min_waves<ENV, DEF, SAFE>(), rollout_max_waves<ENV, DEF, SAFE>())))
    rollout_kernel_v3(const StepArgs a) {
    __shared__ RolloutLds<ENV, E> lds;
    rollout_body_v3<ENV, DEF, E, SAFE, OUT, TAPE, STATS>(a, blockIdx.x, gridDim.x, lds);
}

// ------------------------------------------------------------------------------------------------------------
// mixed_rollout_kernel: heterogeneous dispatch in ONE launch (BASELINE.json configs[4]; SURVEY.md §2.3 "block -> segment
// table, homogeneous workgroups").  A mixed batch is a concatenation of homogeneous segments {CartPole, Pendulum, Acrobot,
// MountainCar, ...} (the reference has no other semantics: gym/vector/vector_env.py:20-23, sync_vector_env.py:220-234).  Every
// workgroup (= one wave) looks up the segment its index falls into and runs THAT env kind's rollout body on the segment's
// own arguments: waves stay homogeneous (no per-lane switch that would serialise four code paths), and all segments share
// one grid, so the chip is filled by one launch instead of four small grids on four streams.  Each body is the code of
// rollout_kernel_v3 with one env per lane (segments of a mixed batch are small): results are bit-identical to launching the
// segments separately.
// ------------------------------------------------------------------------------------------------------------
union MixedLds {
    RolloutLds<MXV_CARTPOLE, 1> cartpole;
    RolloutLds<MXV_PENDULUM, 1> pendulum;
    RolloutLds<MXV_ACROBOT, 1> acrobot;
    RolloutLds<MXV_MOUNTAINCAR, 1> mountaincar;
    RolloutLds<MXV_MOUNTAINCAR_CONT, 1> mountaincar_cont;
    __device__ MixedLds() {}
};

// Block -> segment: segment i owns the contiguous block range [first_block[i], first_block[i+1]).  (An interleaved table — chunks of
// 8 blocks dealt round-robin to the segments, order rotated per round — was measured too: 4.52 us per mixed step against 3.27 us
// for the contiguous ranges, profiles/r2/r02e_mixed_dispatch_interleaved.jsonl.  The hardware deals consecutive workgroups over the
// SIMDs, so what matters is WHICH two waves end up sharing a SIMD: the interleaving paired Acrobot waves with each other.)
__global__ void __launch_bounds__(kWave) mixed_rollout_kernel(const MixedArgs m) {
    __shared__ MixedLds lds;
    unsigned sidx = 0;
#pragma unroll
    for (int i = 1; i < MXV_MAX_MIXED; ++i)
        if (i < m.count && blockIdx.x >= m.first_block[i]) sidx = (unsigned)i;
    const StepArgs &a = m.seg[sidx];
    const unsigned bid = blockIdx.x - m.first_block[sidx], nblk = m.first_block[sidx + 1] - m.first_block[sidx];
    switch (m.kind[sidx]) {   // wave-uniform: one body per workgroup
        case MXV_CARTPOLE:  // SAFE: a segment may have had its state injected
            rollout_body_v3<MXV_CARTPOLE, true, 1, true, 0>(a, bid, nblk, lds.cartpole);
            break;
        case MXV_PENDULUM: rollout_body_v3<MXV_PENDULUM, true, 1, true, 0>(a, bid, nblk, lds.pendulum); break;
        case MXV_ACROBOT: rollout_body_v3<MXV_ACROBOT, true, 1, true, 0>(a, bid, nblk, lds.acrobot); break;
        case MXV_MOUNTAINCAR: rollout_body_v3<MXV_MOUNTAINCAR, true, 1, true, 0>(a, bid, nblk, lds.mountaincar); break;
        default: rollout_body_v3<MXV_MOUNTAINCAR_CONT, true, 1, true, 0>(a, bid, nblk, lds.mountaincar_cont); break;
    }
}

// Explicit reset (SyncVectorEnv.reset_wait, sync_vector_env.py:90-129): one env per lane.
template <int ENV>
__global__ void __launch_bounds__(kBlock) reset_kernel(const ResetArgs a) {
    using EV = Env<ENV>;
    constexpr int S = EV::S, O = EV::O;
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= a.n) return;
    if (a.mask != nullptr && a.mask[e] == 0) return;
    const uint64_t seed = a.seeds ? landed(a.seeds[e]) : a.base_seed + a.env0 + (uint64_t)e;
    const uint32_t k = a.episodes[e];  // this env's reset ordinal since seeding
    a.episodes[e] = k + 1u;
    const U4 w = episode_reset_words(seed, k);
    double s[S];
    EV::reset(w, a.b0, a.b1, s);
#pragma unroll
    for (int k = 0; k < S; ++k) a.state[(int64_t)k * a.n + e] = s[k];
    store_elapsed(a.elapsed, a.elapsed16, e, 0);  // time_limit.py:67
    if (a.ep_acc != nullptr) a.ep_acc[e] = 0.0f;  // record_episode_statistics.py:91-94
    if (a.beyond != nullptr) a.beyond[e] = 0;     // steps_beyond_terminated = None (cartpole.py:205)
    if (a.obs != nullptr) {
        float o[O];
        double aux_unused[EV::AUX > 0 ? EV::AUX : 1];
        EV::observe(s, o, aux_unused);
        store_obs<O>(a.obs, e, o);
    }
}

// action_space.sample() without stepping: one Philox call (4 envs) per lane.
template <int ENV, int DEF>
__global__ void __launch_bounds__(kBlock) sample_kernel(const SampleArgs a) {
    constexpr int NA = Env<ENV>::NA;
    const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;  // local group index
    if (c * 4 >= a.n) return;
    const Par<DEF> P(a.P, a.params_pe, a.n);
    const uint64_t t = a.t + (a.t_dev ? *a.t_dev : 0);
    const U4 w = env_action_words<ENV>(a.action_seed, t, (a.env0 >> 2) + (uint64_t)c);
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t e = c * 4 + q;
        if (e >= a.n) break;
        int ai;
        float af;
        action_from_word<ENV, DEF>(P.at(e), ws[q], t, ai, af);
        if constexpr (NA > 0) {
            if (a.flags & MXV_FLAG_ACTION_I32)
                static_cast<int32_t *>(a.actions_out)[e] = ai;
            else
                static_cast<int64_t *>(a.actions_out)[e] = ai;
        } else {
            static_cast<float *>(a.actions_out)[e] = af;
        }
    }
}

__global__ void set_word_kernel(uint64_t *dst, uint64_t value) { *dst = value; }
__global__ void add_word_kernel(uint64_t *dst, uint64_t delta) { *dst += delta; }

// info["final_observation"] for a host caller: of the dense [N][O] final_obs rows only those of the envs that finished this step
// (terminated | truncated: ~5 % of a random-policy CartPole batch) are meaningful.  Instead of sending the whole array over PCIe
// every step, pack (env index, row) pairs of the finished envs IN ASCENDING ENV ORDER (= np.flatnonzero(terminated | truncated)):
// two small kernels over chunks of kCompactChunk envs — count per chunk, then every chunk's workgroup sums the counts of the
// chunks before it and writes its pairs at the right offset.  No atomics: a first version took one returning atomicAdd per wave on
// a single counter, 16 384 same-address atomics at 2^20 envs = 163 us for a 5-us job (profiles/r2/r03d_numpy_loop_trace.md).
constexpr int kCompactChunk = 4096, kCompactIters = kCompactChunk / kBlock, kCompactWaves = kBlock / kWave;

__device__ __forceinline__ bool compact_done(const CompactArgs &a, int64_t e) {
    return e < a.n && (a.terminated[e] | a.truncated[e]) != 0;
}

__global__ void __launch_bounds__(kBlock) final_count_kernel(const CompactArgs a) {
    __shared__ int32_t part[kCompactWaves];
    const int64_t e0 = (int64_t)blockIdx.x * kCompactChunk + threadIdx.x;
    int32_t c = 0;
#pragma unroll 4
    for (int i = 0; i < kCompactIters; ++i) c += (int32_t)__popcll(__ballot(compact_done(a, e0 + (int64_t)i * kBlock)));
    if (threadIdx.x % kWave == 0) part[threadIdx.x / kWave] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t t = 0;
        for (int w = 0; w < kCompactWaves; ++w) t += part[w];
        a.chunk_counts[blockIdx.x] = t;
    }
}

template <int O>
__global__ void __launch_bounds__(kBlock) final_pack_kernel(const CompactArgs a) {
    static_assert(kCompactIters * kCompactWaves == kWave, "one lane per (iteration, wave) cell in the prefix below");
    __shared__ int32_t cell[kWave];      // finished envs per (iteration, wave) of this chunk, in env order
    __shared__ int32_t red[kCompactWaves];
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    const int64_t e0 = (int64_t)blockIdx.x * kCompactChunk + threadIdx.x;
    // offset of this chunk = finished envs in all chunks before it
    int32_t before = 0;
    for (unsigned c = threadIdx.x; c < blockIdx.x; c += kBlock) before += a.chunk_counts[c];
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) before += __shfl_xor(before, d);
    if (lane == 0) red[wave] = before;
    uint64_t masks[kCompactIters];
#pragma unroll
    for (int i = 0; i < kCompactIters; ++i) {
        masks[i] = __ballot(compact_done(a, e0 + (int64_t)i * kBlock));
        if (lane == 0) cell[i * kCompactWaves + wave] = (int32_t)__popcll(masks[i]);
    }
    __syncthreads();
    int32_t base = 0;
#pragma unroll
    for (int w = 0; w < kCompactWaves; ++w) base += red[w];
    // exclusive prefix over the 64 cells (lane l owns cell l)
    const int32_t mine = cell[lane];
    int32_t incl = mine;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int32_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    const int32_t excl = incl - mine;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kWave - 1) *a.count = base + incl;   // lane 63 of wave 0: the chunk's total
#pragma unroll
    for (int i = 0; i < kCompactIters; ++i) {
        const uint64_t m = masks[i];
        const int32_t off = __shfl(excl, i * kCompactWaves + wave);
        if ((m >> lane) & 1) {
            const int64_t e = e0 + (int64_t)i * kBlock;
            const uint32_t slot = (uint32_t)(base + off) + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            a.idx[slot] = (int32_t)e;
#pragma unroll
            for (int k = 0; k < O; ++k) a.rows[(size_t)slot * O + k] = a.final_obs[(size_t)e * O + k];
            if (a.ep_return_in != nullptr) {   // RecordEpisodeStatistics: return and length of the episode that just ended
                a.ep_return[slot] = a.ep_return_in[e];
                a.ep_length[slot] = a.ep_length_in[e];
            }
        }
    }
}

template <int ENV, bool DEF, int ER, bool SAFE, int OUT, bool TAPE = false, int STATS = 0>
void launch_rollout_out(unsigned grid, hipStream_t stream, const StepArgs &a, LaunchInfo *info) {
    if (info) *info = LaunchInfo{1, ENV, DEF ? PM_DEFAULT : PM_BROADCAST, ER, SAFE ? 1 : 0, OUT, TAPE ? 1 : 0, a.K, grid, (uint32_t)kWave};
    // amdgpu_waves_per_eu only budgets registers; what physically keeps a 5-wave-sized kernel at 4 waves per SIMD (see
    // rollout_max_waves) is its LDS footprint: padded with unused dynamic LDS to 9.5 KiB per single-wave workgroup, 16 of
    // them fill the CU's 160 KiB and a 17th does not fit.
    constexpr size_t kLdsPerWorkgroup = 9728, kStatic = sizeof(RolloutLds<ENV, ER>);
    const size_t pad = (rollout_min_waves<ENV, DEF, SAFE>() == 4 && kStatic < kLdsPerWorkgroup) ? kLdsPerWorkgroup - kStatic : 0;
    hipLaunchKernelGGL((rollout_kernel_v3<ENV, DEF, ER, SAFE, OUT, TAPE, STATS>), dim3(grid), dim3(kWave), pad, stream, a);
}
template <int ENV, bool DEF, int ER, bool SAFE>
void launch_rollout(unsigned grid, hipStream_t stream, const StepArgs &a, LaunchInfo *info) {
    // the trajectory-recording shape (all outputs, no final_obs / statistics) has its own straight-line instantiations; a
    // tape-driven launch of that shape records everything but the actions (the caller holds them)
    const bool tape = a.actions != nullptr;
    int out = 0;
    if (DEF && a.reward && (tape ? !a.actions_out : a.actions_out != nullptr) && a.terminated && a.truncated && !a.final_obs && !a.ep_acc) {
        const int f = a.flags & (MXV_FLAG_REWARD_F32 | MXV_FLAG_ACTION_I32);
        out = f == 0 ? 1 : (f == (MXV_FLAG_REWARD_F32 | MXV_FLAG_ACTION_I32) ? 2 : 0);
    }
    if constexpr (DEF) {
        if (tape) {  // (tapes exist for default parameters only, launch_step_is_rollout; the compact dtypes take the generic body)
            if (out == 1) return launch_rollout_out<ENV, DEF, ER, SAFE, 1, true>(grid, stream, a, info);
            return launch_rollout_out<ENV, DEF, ER, SAFE, 0, true>(grid, stream, a, info);
        }
        if (out == 1) return launch_rollout_out<ENV, DEF, ER, SAFE, 1>(grid, stream, a, info);
        if (out == 2) return launch_rollout_out<ENV, DEF, ER, SAFE, 2>(grid, stream, a, info);
    }
    launch_rollout_out<ENV, DEF, ER, SAFE, 0>(grid, stream, a, info);
}

// STATS launches (fused batch moments: StepArgs::obs_part / ret_part; the host has checked launch_step_supports_stats): default
// parameters, unguarded trigonometry, the trajectory-recording shape — and ALWAYS stats_envs_per_lane(ENV) envs per lane, whatever the
// shard size: the leaves of the sum tree (one per tile) must not depend on it, and two envs per lane halve the wave-level tree's cost per
// env (Pendulum, one env per lane otherwise, takes two here).
constexpr int stats_envs_per_lane(int env_id) { return env_id == MXV_ACROBOT ? 1 : 2; }
template <int ENV>
hipError_t launch_rollout_stats(const StepArgs &a, hipStream_t stream, LaunchInfo *info) {
    constexpr int SE = stats_envs_per_lane(ENV);
    const int64_t rtile = (int64_t)SE * kWave;
    const unsigned grid = (unsigned)((a.n + rtile - 1) / rtile);
    const int st = (a.obs_part != nullptr ? 1 : 0) | (a.ret_part != nullptr ? 2 : 0);
    const bool compact = (a.flags & (MXV_FLAG_REWARD_F32 | MXV_FLAG_ACTION_I32)) != 0;
    if (st == 1 && !compact) launch_rollout_out<ENV, true, SE, false, 1, false, 1>(grid, stream, a, info);
    else if (st == 1) launch_rollout_out<ENV, true, SE, false, 2, false, 1>(grid, stream, a, info);
    else if (st == 2 && !compact) launch_rollout_out<ENV, true, SE, false, 1, false, 2>(grid, stream, a, info);
    else if (st == 2) launch_rollout_out<ENV, true, SE, false, 2, false, 2>(grid, stream, a, info);
    else if (!compact) launch_rollout_out<ENV, true, SE, false, 1, false, 3>(grid, stream, a, info);
    else launch_rollout_out<ENV, true, SE, false, 2, false, 3>(grid, stream, a, info);
    return hipGetLastError();
}

#ifndef MXV_STEP_E1_FROM   // (A/B hook: tools/ab_step_shape.sh builds the library with this out of reach)
#define MXV_STEP_E1_FROM ((int64_t)1 << 19)
#endif
// Single CartPole steps of at least this many envs run ONE env per lane: 2 x 8 waves per SIMD instead of one round of 8, so one
// round's stores overlap the other's loads (profiles/r6/r6i_step_launch_shape.md: 19.4 -> 18.7 us at 2^20; below 2^19 the launch
// is latency-bound and two envs per lane win).
constexpr int64_t kStepOneEnvPerLaneFrom = MXV_STEP_E1_FROM;

template <int ENV>
hipError_t launch_step_env(int pm, const StepArgs &a, hipStream_t stream, LaunchInfo *info) {
    const bool def = pm == PM_DEFAULT;
    // Sampled actions + autoreset, several steps per launch: the fused fast path.  (Single-step launches stay on
    // step_kernel: they are latency-bound and its 58 VGPRs give twice the occupancy.)
    if ((a.obs_part != nullptr || a.ret_part != nullptr) && launch_step_supports_stats(ENV, pm, a)) return launch_rollout_stats<ENV>(a, stream, info);
    if (launch_step_is_rollout(pm, a)) {
        // SAFE = false: the state obeys the invariants the dynamics maintain (CartPole: |theta| <= pi/4; the others: trig arguments
        // below 2^19), so sin/cos need no range check.  A state injection (mxv_set_state) or unusual explicit-reset bounds break
        // that for one launch; an unlimited Pendulum can turn without bound.
        const bool bounded = ENV != MXV_PENDULUM || (a.max_steps > 0 && a.max_steps < 1000000);
        const bool fast = def && !a.state_injected && bounded && MXV_FAST_TRIG;
        auto go = [&](auto er_tag) {
            constexpr int ER = decltype(er_tag)::value;
            const int64_t rtile = (int64_t)ER * kWave;
            const unsigned rgrid = (unsigned)((a.n + rtile - 1) / rtile);
            if (!def) {
                launch_rollout<ENV, false, ER, true>(rgrid, stream, a, info);
            } else if (fast) {
                launch_rollout<ENV, true, ER, false>(rgrid, stream, a, info);
            } else {
                launch_rollout<ENV, true, ER, true>(rgrid, stream, a, info);
            }
        };
        // Two envs per lane (the tuned choice of the light envs: two independent chains of ILP) only pay when the shard fills
        // the chip: below one E = 2 wave per SIMD (1024 SIMDs x 128 envs) the work is latency-bound and one env per lane
        // puts twice as many waves on it (profiles/r2/r02a_shard_sweep.jsonl: 0.76 vs 1.05 us per step at 2^16 CartPole envs;
        // at 2^17 itself, one E = 2 wave per SIMD: 0.92 vs 1.01, profiles/r3/r3k_small_shard_e1_ab.jsonl) — the shard sizes of an 8-GPU
        // strong-scaling or mixed-batch job.
        constexpr int ER = rollout_envs_per_lane(ENV);
        if (ER > 1 && a.n < (int64_t)kSimds * ER * kWave * MXV_ROLLOUT_E1_FACTOR + MXV_ROLLOUT_E1_INCLUSIVE && MXV_ROLLOUT_SMALL_E1)
            go(std::integral_constant<int, 1>{});
        else
            go(std::integral_constant<int, ER>{});
        return hipGetLastError();
    }
    constexpr int E = envs_per_lane(ENV);
    constexpr bool C = MXV_CONSEC != 0;
    const int64_t tile = (int64_t)E * kBlock;
    const unsigned grid = (unsigned)((a.n + tile - 1) / tile);
    if (info) *info = LaunchInfo{0, ENV, pm, E, 1, 0, a.actions != nullptr ? 1 : 0, a.K, grid, (uint32_t)kBlock};
    // A single step of a BIG CartPole batch with two envs per lane is 8192 waves = exactly ONE round at 8 waves per SIMD: every wave
    // loads, then computes, then stores, all in step — the read phase and the write phase of the launch do not overlap (the launch's
    // own access pattern without physics takes 17 of its 19.5 us; moving 16 bytes per env-step fewer did not shorten it: profiles/r6/
    // r6i_step_launch_shape.md).  One env per lane makes it two rounds, the second one's loads under the first one's stores:
    // 19.4 -> 18.7 us per 2^20-env step; occupancy caps (more, thinner rounds) and four envs per lane both lose.
    if constexpr (ENV == MXV_CARTPOLE && E > 1) {
        if (a.K == 1 && pm == PM_DEFAULT && a.clock_ticket == nullptr && a.n >= kStepOneEnvPerLaneFrom) {
            const unsigned grid1 = (unsigned)((a.n + kBlock - 1) / kBlock);
            if (info) *info = LaunchInfo{0, ENV, pm, 1, 1, 0, a.actions != nullptr ? 1 : 0, a.K, grid1, (uint32_t)kBlock};
            hipLaunchKernelGGL((step_kernel<ENV, PM_DEFAULT, 1, C, false>), dim3(grid1), dim3(kBlock), 0, stream, a);
            return hipGetLastError();
        }
    }
    if (a.K > 1) {
        if (pm == PM_DEFAULT)
            hipLaunchKernelGGL((step_kernel<ENV, PM_DEFAULT, E, C, true>), dim3(grid), dim3(kBlock), 0, stream, a);
        else if (pm == PM_BROADCAST)
            hipLaunchKernelGGL((step_kernel<ENV, PM_BROADCAST, E, C, true>), dim3(grid), dim3(kBlock), 0, stream, a);
        else
            hipLaunchKernelGGL((step_kernel<ENV, PM_PER_ENV, E, C, true>), dim3(grid), dim3(kBlock), 0, stream, a);
    } else {
        if (pm == PM_DEFAULT && a.clock_ticket != nullptr)
            hipLaunchKernelGGL((step_kernel<ENV, PM_DEFAULT, E, C, false, true>), dim3(grid), dim3(kBlock), 0, stream, a);
        else if (pm == PM_DEFAULT)
            hipLaunchKernelGGL((step_kernel<ENV, PM_DEFAULT, E, C, false>), dim3(grid), dim3(kBlock), 0, stream, a);
        else if (pm == PM_BROADCAST)
            hipLaunchKernelGGL((step_kernel<ENV, PM_BROADCAST, E, C, false>), dim3(grid), dim3(kBlock), 0, stream, a);
        else
            hipLaunchKernelGGL((step_kernel<ENV, PM_PER_ENV, E, C, false>), dim3(grid), dim3(kBlock), 0, stream, a);
    }
    return hipGetLastError();
}

template <int ENV>
hipError_t launch_sample_env(int pm, const SampleArgs &a, hipStream_t stream) {
    const int64_t groups = (a.n + 3) / 4;
    const unsigned grid = (unsigned)((groups + kBlock - 1) / kBlock);
    if (pm == PM_DEFAULT)
        hipLaunchKernelGGL((sample_kernel<ENV, PM_DEFAULT>), dim3(grid), dim3(kBlock), 0, stream, a);
    else if (pm == PM_BROADCAST)
        hipLaunchKernelGGL((sample_kernel<ENV, PM_BROADCAST>), dim3(grid), dim3(kBlock), 0, stream, a);
    else
        hipLaunchKernelGGL((sample_kernel<ENV, PM_PER_ENV>), dim3(grid), dim3(kBlock), 0, stream, a);
    return hipGetLastError();
}

}  // namespace

bool hilo_supported(int env_id) { return env_id == MXV_CARTPOLE || env_id == MXV_MOUNTAINCAR || env_id == MXV_MOUNTAINCAR_CONT; }

template <int ENV>
static hipError_t launch_hilo_step_env(const StepArgs &a, hipStream_t stream, LaunchInfo *info) {
    constexpr int E = envs_per_lane(ENV);
    constexpr bool C = MXV_CONSEC != 0;
    if constexpr (ENV == MXV_CARTPOLE && E > 1) {      // (two rounds instead of one: see launch_step_env)
        if (a.n >= kStepOneEnvPerLaneFrom) {
            const unsigned grid1 = (unsigned)((a.n + kBlock - 1) / kBlock);
            if (info) *info = LaunchInfo{0, ENV, PM_DEFAULT, 1, 1, 3, a.actions != nullptr ? 1 : 0, 1, grid1, (uint32_t)kBlock};
            hipLaunchKernelGGL((step_kernel<ENV, PM_DEFAULT, 1, C, false, false, true>), dim3(grid1), dim3(kBlock), 0, stream, a);
            return hipGetLastError();
        }
    }
    const int64_t tile = (int64_t)E * kBlock;
    const unsigned grid = (unsigned)((a.n + tile - 1) / tile);
    if (info) *info = LaunchInfo{0, ENV, PM_DEFAULT, E, 1, 3 /* out_mode 3: the observation carries the state */, a.actions != nullptr ? 1 : 0, 1, grid, (uint32_t)kBlock};
    hipLaunchKernelGGL((step_kernel<ENV, PM_DEFAULT, E, C, false, false, true>), dim3(grid), dim3(kBlock), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_hilo_step(int env_id, const StepArgs &a, hipStream_t stream, LaunchInfo *info) {
    switch (env_id) {
        case MXV_CARTPOLE: return launch_hilo_step_env<MXV_CARTPOLE>(a, stream, info);
        case MXV_MOUNTAINCAR: return launch_hilo_step_env<MXV_MOUNTAINCAR>(a, stream, info);
        case MXV_MOUNTAINCAR_CONT: return launch_hilo_step_env<MXV_MOUNTAINCAR_CONT>(a, stream, info);
        default: return hipErrorInvalidValue;
    }
}
hipError_t launch_hilo_split(int env_id, const double *state, float *hi_obs, int32_t *lo, double *side, int64_t n, hipStream_t stream) {
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
    switch (env_id) {
        case MXV_CARTPOLE: hipLaunchKernelGGL(hilo_split_kernel<MXV_CARTPOLE>, grid, block, 0, stream, state, hi_obs, lo, side, n); break;
        case MXV_MOUNTAINCAR: hipLaunchKernelGGL(hilo_split_kernel<MXV_MOUNTAINCAR>, grid, block, 0, stream, state, hi_obs, lo, side, n); break;
        case MXV_MOUNTAINCAR_CONT: hipLaunchKernelGGL(hilo_split_kernel<MXV_MOUNTAINCAR_CONT>, grid, block, 0, stream, state, hi_obs, lo, side, n); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t launch_hilo_join(int env_id, const float *hi_obs, const int32_t *lo, double *state, int64_t n, hipStream_t stream) {
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
    switch (env_id) {
        case MXV_CARTPOLE: hipLaunchKernelGGL(hilo_join_kernel<MXV_CARTPOLE>, grid, block, 0, stream, hi_obs, lo, state, n); break;
        case MXV_MOUNTAINCAR: hipLaunchKernelGGL(hilo_join_kernel<MXV_MOUNTAINCAR>, grid, block, 0, stream, hi_obs, lo, state, n); break;
        case MXV_MOUNTAINCAR_CONT: hipLaunchKernelGGL(hilo_join_kernel<MXV_MOUNTAINCAR_CONT>, grid, block, 0, stream, hi_obs, lo, state, n); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// Sampled actions, or an action tape with the default physics parameters, + autoreset, several steps per launch: the fused fast
// path (rollout_kernel_v3).  Everything else — single steps with caller actions, tapes with changed parameters, per-env
// parameters, torque noise, no autoreset — runs step_kernel.
bool launch_step_is_rollout(int pm, const StepArgs &a) {
    const bool sampled_or_tape = a.actions == nullptr || (a.act_slice != 0 && pm == PM_DEFAULT);
    return sampled_or_tape && !(a.flags & MXV_FLAG_NO_AUTORESET) && a.K > 1 && pm != PM_PER_ENV && !a.step_noise;
}

// STATS (StepArgs::obs_part) exists for the sampled trajectory-recording launch with default parameters, a state inside the invariants,
// every per-step output present and one dtype set — and always with the env kind's own envs-per-lane, so that the leaves of the sum
// tree (one per tile) are the same however small the shard is.
bool launch_step_supports_stats(int env_id, int pm, const StepArgs &a) {
    if (!launch_step_is_rollout(pm, a) || pm != PM_DEFAULT || a.actions != nullptr || a.state_injected || !MXV_FAST_TRIG) return false;
    if (env_id == MXV_PENDULUM && !(a.max_steps > 0 && a.max_steps < 1000000)) return false;
    if (!(a.reward && a.actions_out && a.terminated && a.truncated && !a.final_obs && !a.ep_acc) || a.slice == 0) return false;
    const int f = a.flags & (MXV_FLAG_REWARD_F32 | MXV_FLAG_ACTION_I32);
    return f == 0 || f == (MXV_FLAG_REWARD_F32 | MXV_FLAG_ACTION_I32);
}
int64_t stats_leaf_envs(int env_id) {
    switch (env_id) {
        case MXV_CARTPOLE: return (int64_t)stats_envs_per_lane(MXV_CARTPOLE) * kWave;
        case MXV_PENDULUM: return (int64_t)stats_envs_per_lane(MXV_PENDULUM) * kWave;
        case MXV_ACROBOT: return (int64_t)stats_envs_per_lane(MXV_ACROBOT) * kWave;
        case MXV_MOUNTAINCAR: return (int64_t)stats_envs_per_lane(MXV_MOUNTAINCAR) * kWave;
        default: return (int64_t)stats_envs_per_lane(MXV_MOUNTAINCAR_CONT) * kWave;
    }
}

hipError_t launch_step(int env_id, int default_params, const StepArgs &a, hipStream_t stream, LaunchInfo *info) {
    switch (env_id) {
        case MXV_CARTPOLE: return launch_step_env<MXV_CARTPOLE>(default_params, a, stream, info);
        case MXV_PENDULUM: return launch_step_env<MXV_PENDULUM>(default_params, a, stream, info);
        case MXV_ACROBOT: return launch_step_env<MXV_ACROBOT>(default_params, a, stream, info);
        case MXV_MOUNTAINCAR: return launch_step_env<MXV_MOUNTAINCAR>(default_params, a, stream, info);
        case MXV_MOUNTAINCAR_CONT: return launch_step_env<MXV_MOUNTAINCAR_CONT>(default_params, a, stream, info);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_reset(int env_id, const ResetArgs &a, hipStream_t stream) {
    const unsigned grid = (unsigned)((a.n + kBlock - 1) / kBlock);
    switch (env_id) {
        case MXV_CARTPOLE: hipLaunchKernelGGL(reset_kernel<MXV_CARTPOLE>, dim3(grid), dim3(kBlock), 0, stream, a); break;
        case MXV_PENDULUM: hipLaunchKernelGGL(reset_kernel<MXV_PENDULUM>, dim3(grid), dim3(kBlock), 0, stream, a); break;
        case MXV_ACROBOT: hipLaunchKernelGGL(reset_kernel<MXV_ACROBOT>, dim3(grid), dim3(kBlock), 0, stream, a); break;
        case MXV_MOUNTAINCAR: hipLaunchKernelGGL(reset_kernel<MXV_MOUNTAINCAR>, dim3(grid), dim3(kBlock), 0, stream, a); break;
        case MXV_MOUNTAINCAR_CONT:
            hipLaunchKernelGGL(reset_kernel<MXV_MOUNTAINCAR_CONT>, dim3(grid), dim3(kBlock), 0, stream, a);
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_sample(int env_id, int default_params, const SampleArgs &a, hipStream_t stream) {
    switch (env_id) {
        case MXV_CARTPOLE: return launch_sample_env<MXV_CARTPOLE>(default_params, a, stream);
        case MXV_PENDULUM: return launch_sample_env<MXV_PENDULUM>(default_params, a, stream);
        case MXV_ACROBOT: return launch_sample_env<MXV_ACROBOT>(default_params, a, stream);
        case MXV_MOUNTAINCAR: return launch_sample_env<MXV_MOUNTAINCAR>(default_params, a, stream);
        case MXV_MOUNTAINCAR_CONT: return launch_sample_env<MXV_MOUNTAINCAR_CONT>(default_params, a, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_mixed_rollout(const MixedArgs &m, hipStream_t stream) {
    hipLaunchKernelGGL(mixed_rollout_kernel, dim3(m.first_block[m.count]), dim3(kWave), 0, stream, m);
    return hipGetLastError();
}

int64_t compact_chunks(int64_t n) { return (n + kCompactChunk - 1) / kCompactChunk; }

hipError_t launch_compact_final(int obs_dim, const CompactArgs &a, hipStream_t stream) {
    const unsigned grid = (unsigned)compact_chunks(a.n);
    hipLaunchKernelGGL(final_count_kernel, dim3(grid), dim3(kBlock), 0, stream, a);
    switch (obs_dim) {
        case 2: hipLaunchKernelGGL(final_pack_kernel<2>, dim3(grid), dim3(kBlock), 0, stream, a); break;
        case 3: hipLaunchKernelGGL(final_pack_kernel<3>, dim3(grid), dim3(kBlock), 0, stream, a); break;
        case 4: hipLaunchKernelGGL(final_pack_kernel<4>, dim3(grid), dim3(kBlock), 0, stream, a); break;
        case 6: hipLaunchKernelGGL(final_pack_kernel<6>, dim3(grid), dim3(kBlock), 0, stream, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// Diagnostic: the store pattern of the fused CartPole rollout with the physics removed (tools/wbench5.hip, M = 0) — one wave per
// workgroup, two envs per lane, XCD-contiguous tiles, the five output streams with the reference's dtypes.  What THIS box sustains
// for the pattern; bench.py prints it next to the kernel's own time (boxes differ by 20 %).
__global__ void __launch_bounds__(kWave, 4) write_probe_kernel(float4 *obs, double *rew, int64_t *act, uint8_t *term, uint8_t *trunc,
                                                               int64_t n, int K) {
    const unsigned tile = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x;
    const int64_t e0 = (int64_t)tile * 128 + lane, e1 = e0 + 64;
    float x = (float)lane;
    for (int k = 0; k < K; ++k) {
        const int64_t so = (int64_t)k * n;
        x = x * 1.0001f + 0.5f;
        act[so + e0] = k & 1;
        act[so + e1] = (k >> 1) & 1;
        rew[so + e0] = 1.0;
        rew[so + e1] = 1.0;
        term[so + e0] = x > 1e30f;
        term[so + e1] = 0;
        trunc[so + e0] = 0;
        trunc[so + e1] = 0;
        obs[so + e0] = make_float4(x, x + 1, 0.f, 1.f);
        obs[so + e1] = make_float4(x + 2, x, 1.f, 0.f);
    }
}

hipError_t launch_write_probe(float *obs, double *rew, int64_t *act, uint8_t *term, uint8_t *trunc, int64_t n, int K, hipStream_t stream) {
    hipLaunchKernelGGL(write_probe_kernel, dim3((unsigned)(n / 128)), dim3(kWave), 0, stream, reinterpret_cast<float4 *>(obs), rew, act, term,
                       trunc, n, K);
    return hipGetLastError();
}

// The same for any env kind: observation rows of O floats, E envs per lane as the fused rollout of that kind runs them, rewards of 8 or
// 4 bytes, actions of 8 / 4 bytes (int64 / int32 / float32) — the store pattern of rollout_kernel_v3<ENV, ..., OUT != 0>.
template <int O, int E>
__global__ void __launch_bounds__(kWave, 4) write_probe_env_kernel(float *obs, void *rew, void *act, uint8_t *term, uint8_t *trunc, int64_t n, int K,
                                                                   int rew_f32, int act_bytes) {
    const unsigned tile = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x;
    float x = (float)lane;
    for (int k = 0; k < K; ++k) {
        const int64_t so = (int64_t)k * n;
        x = x * 1.0001f + 0.5f;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int64_t e = so + (int64_t)tile * (E * kWave) + j * kWave + lane;
            if (e - so >= n) continue;
            if (act_bytes == 8)
                static_cast<int64_t *>(act)[e] = (k + j) & 1;
            else
                static_cast<int32_t *>(act)[e] = (k + j) & 1;
            if (rew_f32)
                static_cast<float *>(rew)[e] = 1.0f;
            else
                static_cast<double *>(rew)[e] = 1.0;
            term[e] = x > 1e30f;
            trunc[e] = 0;
            float o[O];
#pragma unroll
            for (int c = 0; c < O; ++c) o[c] = x + (float)(c + j);
            store_obs<O>(obs, e, o);
        }
    }
}

hipError_t launch_write_probe_env(int env_id, int flags, float *obs, void *rew, void *act, uint8_t *term, uint8_t *trunc, int64_t n, int K,
                                  hipStream_t stream) {
    const int rew_f32 = (flags & MXV_FLAG_REWARD_F32) ? 1 : 0;
    auto go = [&](auto o_tag, auto e_tag, int act_bytes) {
        constexpr int O = decltype(o_tag)::value, E = decltype(e_tag)::value;
        const unsigned grid = (unsigned)((n + E * kWave - 1) / (E * kWave));
        hipLaunchKernelGGL((write_probe_env_kernel<O, E>), dim3(grid), dim3(kWave), 0, stream, obs, rew, act, term, trunc, n, K, rew_f32, act_bytes);
    };
    const int disc = (flags & MXV_FLAG_ACTION_I32) ? 4 : 8;
    using std::integral_constant;
    switch (env_id) {
        case MXV_CARTPOLE: go(integral_constant<int, 4>{}, integral_constant<int, rollout_envs_per_lane(MXV_CARTPOLE)>{}, disc); break;
        case MXV_PENDULUM: go(integral_constant<int, 3>{}, integral_constant<int, rollout_envs_per_lane(MXV_PENDULUM)>{}, 4); break;
        case MXV_ACROBOT: go(integral_constant<int, 6>{}, integral_constant<int, rollout_envs_per_lane(MXV_ACROBOT)>{}, disc); break;
        case MXV_MOUNTAINCAR: go(integral_constant<int, 2>{}, integral_constant<int, rollout_envs_per_lane(MXV_MOUNTAINCAR)>{}, disc); break;
        case MXV_MOUNTAINCAR_CONT: go(integral_constant<int, 2>{}, integral_constant<int, rollout_envs_per_lane(MXV_MOUNTAINCAR_CONT)>{}, 4); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_set_word(uint64_t *dst, uint64_t value, hipStream_t stream) {
    hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, stream, dst, value);
    return hipGetLastError();
}

hipError_t launch_add_word(uint64_t *dst, uint64_t delta, hipStream_t stream) {
    hipLaunchKernelGGL(add_word_kernel, dim3(1), dim3(1), 0, stream, dst, delta);
    return hipGetLastError();
}

}  // namespace mxv
