// mxv_subnorm.hip — SURVEY.md §8(f)-2, the PER-SUB-ENV form: what `gym.vector.make(id, n, wrappers=[NormalizeObservation,
// NormalizeReward])` runs in the reference (gym/vector/__init__.py:56-65 puts the wrappers around EVERY sub-env), as gfx950
// kernels behind the mxv_subnorm_* C ABI (include/mxv_norm.h).
//
// Every sub-env owns a RunningMeanStd (gym/wrappers/normalize.py:8-29) that is updated with batches of ONE row:
// update_mean_var_count_from_moments (:32-47) with batch_mean = the row, batch_var = 0, batch_count = 1.  There is no cross-env
// reduction at all — one lane owns one env, keeps its statistics (mean[dim], var[dim], count: 2 dim + 1 fp64) in registers across
// the K steps of a trajectory tensor and touches them in HBM once per launch, so a K-step launch moves the observations in and out
// (8 dim + 2 bytes per env-step, float32 both ways) plus 16 (2 dim + 1) / K bytes of statistics; K = 1 is the step() form.
//
// Order of events per env and step, as the reference's wrappers see it (sync_vector_env.py:142-156): the sub-env's step() returns
// the TERMINAL observation of an episode that ends (normalised: one update; it travels as info["final_observation"], float64),
// then the autoreset calls the sub-env's reset(), whose observation is normalised too (a second update) and is the row of the
// batched observations.  NormalizeReward: returns = returns * gamma + reward; update(returns); reward / sqrt(var + epsilon);
// returns = 0 where the episode ended (:132-145).
//
// Arithmetic: the reference's, operation for operation and in its order (float32 row - float64 mean -> float64; IEEE fp64 `/` and
// sqrt; the translation unit is built with -ffp-contract=off), so the statistics and the float64 results are bit-identical to the
// NumPy wrappers' on the same inputs; the float32 batched observations are the float64 results rounded once, as np.stack into
// the float32 observation buffer rounds them (numpy_utils.py:49-50).  tests/test_gpu_subnorm.py holds this against the CPU
// restatement that the test suite pins to the reference's own run bit for bit.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <new>
#include <string>
#include <vector>

#include "../../include/mxv.h"

namespace {

constexpr int kThreads = 256;

template <int D>
__device__ __forceinline__ void load_row(const float *base, int64_t row, float (&v)[D]) {
    if constexpr (D == 4) {
        const float4 q = reinterpret_cast<const float4 *>(base)[row];
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else if constexpr (D == 2) {
        const float2 q = reinterpret_cast<const float2 *>(base)[row];
        v[0] = q.x; v[1] = q.y;
    } else if constexpr (D == 6) {
        const float2 *p = reinterpret_cast<const float2 *>(base) + row * 3;
        const float2 a = p[0], b = p[1], c = p[2];
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y;
    } else {
#pragma unroll
        for (int j = 0; j < D; ++j) v[j] = base[row * D + j];
    }
}

template <int D>
__device__ __forceinline__ void store_row(float *base, int64_t row, const double (&y)[D]) {
    if constexpr (D == 4) {
        reinterpret_cast<float4 *>(base)[row] = make_float4((float)y[0], (float)y[1], (float)y[2], (float)y[3]);
    } else if constexpr (D == 2) {
        reinterpret_cast<float2 *>(base)[row] = make_float2((float)y[0], (float)y[1]);
    } else if constexpr (D == 6) {
        float2 *p = reinterpret_cast<float2 *>(base) + row * 3;
        p[0] = make_float2((float)y[0], (float)y[1]);
        p[1] = make_float2((float)y[2], (float)y[3]);
        p[2] = make_float2((float)y[4], (float)y[5]);
    } else {
#pragma unroll
        for (int j = 0; j < D; ++j) base[row * D + j] = (float)y[j];
    }
}

template <int D>
__device__ __forceinline__ void store_row(double *base, int64_t row, const double (&y)[D]) {
    if constexpr (D % 2 == 0) {
        double2 *p = reinterpret_cast<double2 *>(base) + row * (D / 2);
#pragma unroll
        for (int j = 0; j < D / 2; ++j) p[j] = make_double2(y[2 * j], y[2 * j + 1]);
    } else {
#pragma unroll
        for (int j = 0; j < D; ++j) base[row * D + j] = y[j];
    }
}

// Three quotients by the SAME divisor per update (delta / tot, square(delta) * count / tot, M2 / tot).  The compiler's IEEE fp64 `/` is
// v_div_scale x2, v_rcp_f64, two Newton steps, q0 = x * r, rem = fma(-d, q0, x), v_div_fmas, v_div_fixup; for operands that need no
// scaling (exponents far from the limits) the scale factors are 1, so the reciprocal part can run once per update and each dividend pays
// the three-instruction tail plus v_div_fixup (IEEE's answers for 0 / Inf / NaN operands) — the same bits as `/` (the form mxv_device.hpp
// uses for Acrobot's shared divisor; tools/divcheck.hip: 0 mismatches on 4e9 operand pairs on the MI355X).  Operands outside that range
// (a dividend outside 2^-723 .. 2^677, a divisor outside 2^-64 .. 2^64 — v_div_scale rescales when the dividend's exponent is tiny, the
// exponents differ by 768 or more, or the quotient would be subnormal — i.e. statistics somebody injected, never the dynamics) send the
// WAVE down the plain `/` path, so the result is IEEE division bit for bit everywhere; tests/test_gpu_subnorm.py walks both paths.
#ifndef MXV_SUBNORM_SHARED_RCP
#define MXV_SUBNORM_SHARED_RCP 1   // A/B hook: 0 = the compiler's `/` everywhere
#endif
// (v_frexp_exp_i32_f64 answers 0 for zero, Inf and NaN — v_div_fixup's business, accepted — and the true exponent for subnormals: one
// instruction, an add and an unsigned compare per test.)
__device__ __forceinline__ bool plain_operand(double x) {   // finite non-zero needs 2^-723 <= |x| < 2^678
    return (uint32_t)(__builtin_amdgcn_frexp_exp(x) + 722) < 1401u;
}
// delta = row - mean: with |delta| in 2^-300 .. 2^300 (or 0 / Inf / NaN) and count in 2^-64 .. 2^64 (or 0), square(delta) * count lies in
// 2^-664 .. 2^664 (or is 0 / Inf / NaN): plain by construction, no test of its own.
__device__ __forceinline__ bool plain_delta(double x) { return (uint32_t)(__builtin_amdgcn_frexp_exp(x) + 299) < 601u; }
__device__ __forceinline__ bool plain_divisor(double d) {   // count + 1: 1.0001 .. 2^53 in any real run
    const uint32_t e = ((uint32_t)__double2hiint(d) >> 20) & 0x7ffu;
    return e - 959u < 129u;
}
__device__ __forceinline__ double refined_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __fma_rn(-d, r, 1.0);
    r = __fma_rn(r, e, r);
    e = __fma_rn(-d, r, 1.0);
    return __fma_rn(r, e, r);
}
__device__ __forceinline__ double div_shared(double x, double d, double r) {
    const double q0 = x * r;
    return __builtin_amdgcn_div_fixup(__fma_rn(__fma_rn(-d, q0, x), r, q0), d, x);
}

// RunningMeanStd.update with a batch of one row (normalize.py:17-22 -> :32-47): batch_mean = x, batch_var = 0, batch_count = 1.
template <int D>
__device__ __forceinline__ void update_one(const float (&x)[D], double (&mean)[D], double (&var)[D], double &count) {
    const double tot = count + 1.0;  // :37
    double delta[D], sq[D];
    bool ok = plain_divisor(tot) && (plain_divisor(count) || count == 0.0);
#pragma unroll
    for (int j = 0; j < D; ++j) {
        delta[j] = (double)x[j] - mean[j];                           // :36   float32 - float64 -> float64
        sq[j] = (delta[j] * delta[j]) * count;                       // :42   square(delta) * count (* 1)
        ok = ok && plain_delta(delta[j]);
    }
    if (MXV_SUBNORM_SHARED_RCP && __all(ok)) {
        const double r = refined_rcp(tot);
        double m2[D];
        bool ok2 = true;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            m2[j] = var[j] * count + div_shared(sq[j], tot, r);      // :40-42  m_a + m_b(= 0) + ... / tot
            mean[j] = mean[j] + div_shared(delta[j], tot, r);        // :39   delta * 1 / tot
            ok2 = ok2 && plain_operand(m2[j]);
        }
        if (__all(ok2)) {
#pragma unroll
            for (int j = 0; j < D; ++j) var[j] = div_shared(m2[j], tot, r);   // :43
        } else {
#pragma unroll
            for (int j = 0; j < D; ++j) var[j] = m2[j] / tot;
        }
    } else {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const double m2 = var[j] * count + sq[j] / tot;
            mean[j] = mean[j] + delta[j] / tot;
            var[j] = m2 / tot;
        }
    }
    count = tot;  // :44
}

// NormalizeObservation.normalize after the update (:92-93)
template <int D>
__device__ __forceinline__ void normalise(const float (&x)[D], const double (&mean)[D], const double (&var)[D], double eps, double (&y)[D]) {
#pragma unroll
    for (int j = 0; j < D; ++j) y[j] = ((double)x[j] - mean[j]) / sqrt(var[j] + eps);
}

struct SubObsArgs {
    const float *x;        // [K][N][D] the batched observations (post-autoreset rows where an episode ended)
    const float *fin;      // [K][N][D] terminal observations, valid where terminated | truncated; or nullptr
    const uint8_t *te, *tr;  // [K][N] or nullptr (reset(): nobody has finished)
    void *y;               // [K][N][D] float32 or float64; may alias x when float32
    double *yfin;          // [K][N][D] float64 normalised terminal observations (rows of finished envs only) or nullptr
    double *stat;          // [2 D + 1][N]: mean, var, count
    int64_t n;
    int32_t K;
    double eps;
};

// One lane owns one env and walks that env's EVENTS in order: per step the terminal observation where the episode ended (if the caller
// passed them), then the row of the batch.  Every loop iteration handles ONE event of every lane, so the fp64 update + normalisation runs
// once per iteration for the whole wave, and the lanes drift apart by the number of episodes their env has finished so far (kl = the lane's
// own step index).  Walking the steps in lockstep instead makes a wave pay the terminal branch whenever ANY of its 64 envs finished — 95 %
// of CartPole's wave-steps under random actions (22.0 -> 20.1 us per 2^20-env step at K = 64; 18.1 with the cheaper range tests:
// profiles/r6/r6k_subnorm_event_walk.md — 300 VALU per env-step, 0.73 of the issue rate at 5 waves per SIMD; 6 waves change nothing, 8 spill).
// The loop ends when the slowest lane is through: K + max-over-lanes(episodes ended) iterations.
// Lanes that drifted read and write their 16-byte rows in different [N] slices; the rows of a line meet again in the L2.
template <int D, typename OUT>
__global__ void __launch_bounds__(kThreads) subnorm_obs_kernel(const SubObsArgs a) {
    const int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const bool mine = e < a.n;
    const int64_t ec = mine ? e : 0;
    double mean[D], var[D], count;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        mean[j] = a.stat[(int64_t)j * a.n + ec];
        var[j] = a.stat[(int64_t)(D + j) * a.n + ec];
    }
    count = a.stat[(int64_t)(2 * D) * a.n + ec];
    int kl = mine ? 0 : a.K;   // the lane's step
    bool fin_done = false;     // the terminal observation of step kl has been handled
    const bool has_fin = a.te != nullptr && a.fin != nullptr;
    while (__any(kl < a.K)) {
        if (kl < a.K) {   // (lanes that are through sit out: the wave-level votes inside update_one count the lanes at work only)
            const int64_t r = (int64_t)kl * a.n + e;
            const bool use_fin = has_fin && !fin_done && ((a.te[r] | a.tr[r]) != 0);
            float x[D];
            load_row<D>(use_fin ? a.fin : a.x, r, x);
            double y[D];
            update_one<D>(x, mean, var, count);
            normalise<D>(x, mean, var, a.eps, y);
            if (use_fin) {   // the sub-env's step() returned the terminal observation first ...
                if (a.yfin != nullptr) store_row<D>(a.yfin, r, y);
                fin_done = true;
            } else {         // ... then (or only) the row of the batch
                store_row<D>(static_cast<OUT *>(a.y), r, y);
                fin_done = false;
                kl += 1;
            }
        }
    }
    if (!mine) return;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        a.stat[(int64_t)j * a.n + e] = mean[j];
        a.stat[(int64_t)(D + j) * a.n + e] = var[j];
    }
    a.stat[(int64_t)(2 * D) * a.n + e] = count;
}

struct SubRewArgs {
    const void *rew;         // [K][N] float64 / float32
    const uint8_t *te, *tr;  // [K][N]
    void *out;               // [K][N], may alias rew
    double *stat;            // [3][N]: mean, var, count of return_rms
    double *returns;         // [N]
    int64_t n;
    int32_t K;
    double gamma, eps;
};

template <typename RT>
__global__ void __launch_bounds__(kThreads) subnorm_rew_kernel(const SubRewArgs a) {
    const int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (e >= a.n) return;
    double mean = a.stat[e], var = a.stat[a.n + e], count = a.stat[2 * a.n + e], ret = a.returns[e];
    const RT *rew = static_cast<const RT *>(a.rew);
    RT *out = static_cast<RT *>(a.out);
    for (int k = 0; k < a.K; ++k) {
        const int64_t r = (int64_t)k * a.n + e;
        const double rw = (double)rew[r];
        const bool done = (a.te[r] | a.tr[r]) != 0;
        ret = ret * a.gamma + rw;                                   // :132
        {                                                           // :144 return_rms.update(self.returns): a batch of one (update_one)
            const double tot = count + 1.0, delta = ret - mean, sq = (delta * delta) * count;
            const bool ok = plain_divisor(tot) && (plain_divisor(count) || count == 0.0) && plain_delta(delta);
            if (MXV_SUBNORM_SHARED_RCP && __all(ok)) {
                const double rc = refined_rcp(tot);
                const double m2 = var * count + div_shared(sq, tot, rc);
                mean = mean + div_shared(delta, tot, rc);
                var = __all(plain_operand(m2)) ? div_shared(m2, tot, rc) : m2 / tot;
            } else {
                const double m2 = var * count + sq / tot;
                mean = mean + delta / tot;
                var = m2 / tot;
            }
            count = tot;
        }
        out[r] = (RT)(rw / sqrt(var + a.eps));                       // :145
        if (done) ret = 0.0;                                         // :134-135
    }
    a.stat[e] = mean;
    a.stat[a.n + e] = var;
    a.stat[2 * a.n + e] = count;
    a.returns[e] = ret;
}

}  // namespace

struct mxv_subnorm {
    int device = 0, dim = 0;
    int64_t n = 0;
    hipStream_t stream = nullptr;
    double *stat = nullptr;     // [2 dim + 1][n]
    double *returns = nullptr;  // [n]
    std::string error;
};

namespace {

thread_local std::string g_subnorm_create_error;

int sfail(mxv_subnorm *nm, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (nm)
        nm->error = buf;
    else
        g_subnorm_create_error = buf;
    return code;
}

#define SUB_HIP(nm, expr)                                                                               \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return sfail((nm), MXV_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

bool sub_dim_supported(int d) { return d == 1 || d == 2 || d == 3 || d == 4 || d == 6; }

int sub_checks(mxv_subnorm *nm, int32_t K) {
    if (!nm) return sfail(nullptr, MXV_ERR_INVALID_ARG, "NULL mxv_subnorm");
    if (K <= 0) return sfail(nm, MXV_ERR_INVALID_ARG, "K must be positive (got %d)", K);
    if ((int64_t)K * nm->n > ((int64_t)1 << 40)) return sfail(nm, MXV_ERR_INVALID_ARG, "K * num_envs too large");
    SUB_HIP(nm, hipSetDevice(nm->device));
    return MXV_OK;
}

int init_stats(mxv_subnorm *nm) {  // RunningMeanStd.__init__: mean 0, var 1, count = epsilon = 1e-4 (normalize.py:12-15), every sub-env
    const size_t n = (size_t)nm->n;
    std::vector<double> init((size_t)(2 * nm->dim + 1) * n, 0.0);
    for (size_t i = (size_t)nm->dim * n; i < (size_t)(2 * nm->dim) * n; ++i) init[i] = 1.0;
    for (size_t i = (size_t)(2 * nm->dim) * n; i < init.size(); ++i) init[i] = 1e-4;
    SUB_HIP(nm, hipMemcpy(nm->stat, init.data(), init.size() * sizeof(double), hipMemcpyHostToDevice));
    SUB_HIP(nm, hipMemset(nm->returns, 0, n * sizeof(double)));  // np.zeros(num_envs), :123
    return MXV_OK;
}

template <typename OUT>
int launch_obs(mxv_subnorm *nm, const SubObsArgs &a) {
    const dim3 grid((unsigned)((a.n + kThreads - 1) / kThreads)), block(kThreads);
    switch (nm->dim) {
        case 1: hipLaunchKernelGGL((subnorm_obs_kernel<1, OUT>), grid, block, 0, nm->stream, a); break;
        case 2: hipLaunchKernelGGL((subnorm_obs_kernel<2, OUT>), grid, block, 0, nm->stream, a); break;
        case 3: hipLaunchKernelGGL((subnorm_obs_kernel<3, OUT>), grid, block, 0, nm->stream, a); break;
        case 4: hipLaunchKernelGGL((subnorm_obs_kernel<4, OUT>), grid, block, 0, nm->stream, a); break;
        default: hipLaunchKernelGGL((subnorm_obs_kernel<6, OUT>), grid, block, 0, nm->stream, a); break;
    }
    SUB_HIP(nm, hipGetLastError());
    return MXV_OK;
}

}  // namespace

extern "C" {

int mxv_subnorm_create(int32_t device, int32_t dim, int64_t num_envs, void *stream, mxv_subnorm **out) {
    if (!out) return sfail(nullptr, MXV_ERR_INVALID_ARG, "NULL output pointer");
    *out = nullptr;
    if (!sub_dim_supported(dim)) return sfail(nullptr, MXV_ERR_UNSUPPORTED, "dim must be one of 1, 2, 3, 4, 6 (got %d)", dim);
    if (num_envs <= 0 || num_envs > ((int64_t)1 << 31)) return sfail(nullptr, MXV_ERR_INVALID_ARG, "num_envs must be in 1 .. 2^31");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return sfail(nullptr, MXV_ERR_HIP, "no HIP device available (%s): the engine has no CPU fallback",
                     e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= ndev) return sfail(nullptr, MXV_ERR_INVALID_ARG, "device %d out of range", device);
    mxv_subnorm *nm = new (std::nothrow) mxv_subnorm();
    if (!nm) return sfail(nullptr, MXV_ERR_INVALID_ARG, "out of host memory");
    nm->device = device;
    nm->dim = dim;
    nm->n = num_envs;
    nm->stream = (hipStream_t)stream;
    hipError_t err = hipSetDevice(device);
    if (err == hipSuccess) err = hipMalloc((void **)&nm->stat, (size_t)(2 * dim + 1) * (size_t)num_envs * sizeof(double));
    if (err == hipSuccess) err = hipMalloc((void **)&nm->returns, (size_t)num_envs * sizeof(double));
    if (err != hipSuccess || init_stats(nm) != MXV_OK) {
        sfail(nullptr, MXV_ERR_HIP, "mxv_subnorm_create: %s", err != hipSuccess ? hipGetErrorString(err) : nm->error.c_str());
        mxv_subnorm_destroy(nm);
        return MXV_ERR_HIP;
    }
    *out = nm;
    return MXV_OK;
}

int mxv_subnorm_destroy(mxv_subnorm *nm) {
    if (!nm) return MXV_OK;
    (void)hipSetDevice(nm->device);
    (void)hipStreamSynchronize(nm->stream);
    if (nm->stat) (void)hipFree(nm->stat);
    if (nm->returns) (void)hipFree(nm->returns);
    delete nm;
    return MXV_OK;
}

const char *mxv_subnorm_last_error(const mxv_subnorm *nm) { return nm ? nm->error.c_str() : g_subnorm_create_error.c_str(); }

int mxv_subnorm_set_stream(mxv_subnorm *nm, void *stream) {
    if (!nm) return sfail(nullptr, MXV_ERR_INVALID_ARG, "NULL mxv_subnorm");
    SUB_HIP(nm, hipSetDevice(nm->device));
    SUB_HIP(nm, hipStreamSynchronize(nm->stream));
    nm->stream = (hipStream_t)stream;
    return MXV_OK;
}

int mxv_subnorm_get_state(mxv_subnorm *nm, double *mean_host, double *var_host, double *count_host, double *returns_host) {
    if (!nm) return sfail(nullptr, MXV_ERR_INVALID_ARG, "NULL mxv_subnorm");
    SUB_HIP(nm, hipSetDevice(nm->device));
    const size_t n = (size_t)nm->n, D = (size_t)nm->dim;
    std::vector<double> st((2 * D + 1) * n);
    SUB_HIP(nm, hipMemcpyAsync(st.data(), nm->stat, st.size() * sizeof(double), hipMemcpyDeviceToHost, nm->stream));
    if (returns_host) SUB_HIP(nm, hipMemcpyAsync(returns_host, nm->returns, n * sizeof(double), hipMemcpyDeviceToHost, nm->stream));
    SUB_HIP(nm, hipStreamSynchronize(nm->stream));
    for (size_t i = 0; i < n; ++i) {  // device layout [2 D + 1][n] -> the wrappers' per-env arrays [n][D]
        for (size_t j = 0; j < D; ++j) {
            if (mean_host) mean_host[i * D + j] = st[j * n + i];
            if (var_host) var_host[i * D + j] = st[(D + j) * n + i];
        }
        if (count_host) count_host[i] = st[2 * D * n + i];
    }
    return MXV_OK;
}

int mxv_subnorm_set_state(mxv_subnorm *nm, const double *mean_host, const double *var_host, const double *count_host,
                          const double *returns_host) {
    if (!nm) return sfail(nullptr, MXV_ERR_INVALID_ARG, "NULL mxv_subnorm");
    if (!mean_host || !var_host || !count_host) return sfail(nm, MXV_ERR_INVALID_ARG, "mean / var / count pointer is NULL");
    SUB_HIP(nm, hipSetDevice(nm->device));
    const size_t n = (size_t)nm->n, D = (size_t)nm->dim;
    std::vector<double> st((2 * D + 1) * n);
    for (size_t i = 0; i < n; ++i) {
        for (size_t j = 0; j < D; ++j) {
            st[j * n + i] = mean_host[i * D + j];
            st[(D + j) * n + i] = var_host[i * D + j];
        }
        st[2 * D * n + i] = count_host[i];
    }
    SUB_HIP(nm, hipMemcpyAsync(nm->stat, st.data(), st.size() * sizeof(double), hipMemcpyHostToDevice, nm->stream));
    if (returns_host) SUB_HIP(nm, hipMemcpyAsync(nm->returns, returns_host, n * sizeof(double), hipMemcpyHostToDevice, nm->stream));
    SUB_HIP(nm, hipStreamSynchronize(nm->stream));
    return MXV_OK;
}

int mxv_subnorm_observations(mxv_subnorm *nm, int32_t K, const float *x_dev, const float *final_dev, const uint8_t *terminated_dev,
                             const uint8_t *truncated_dev, void *y_dev, int32_t out_f32, double *final_y_dev, double epsilon) {
    if (int rc = sub_checks(nm, K)) return rc;
    if (!x_dev || !y_dev) return sfail(nm, MXV_ERR_INVALID_ARG, "x / y pointer is NULL");
    if ((terminated_dev == nullptr) != (truncated_dev == nullptr))
        return sfail(nm, MXV_ERR_INVALID_ARG, "terminated and truncated go together (both NULL: nobody has finished, the reset() form)");
    if (final_dev && !terminated_dev) return sfail(nm, MXV_ERR_INVALID_ARG, "terminal observations without the flags that say whose they are");
    if (final_y_dev && !final_dev) return sfail(nm, MXV_ERR_INVALID_ARG, "final_y without the terminal observations");
    if (!out_f32 && (const void *)x_dev == (const void *)y_dev)
        return sfail(nm, MXV_ERR_INVALID_ARG, "float64 results cannot alias the float32 observations");
    {   // rows on the boundary of the vector width the kernel moves them with
        const int D = nm->dim;
        const size_t a32 = D % 4 == 0 ? 16 : (D % 2 == 0 ? 8 : 4), a64 = D % 2 == 0 ? 16 : 8;
        const struct { const void *p; size_t b; const char *what; } t[] = {{x_dev, a32, "x"}, {final_dev, a32, "final"}, {y_dev, out_f32 ? a32 : a64, "y"},
                                                                           {final_y_dev, a64, "final_y"}};
        for (const auto &e : t)
            if (e.p && ((uintptr_t)e.p & (e.b - 1)) != 0) return sfail(nm, MXV_ERR_INVALID_ARG, "%s pointer %p is not %zu-byte aligned", e.what, e.p, e.b);
    }
    SubObsArgs a{x_dev, final_dev, terminated_dev, truncated_dev, y_dev, final_y_dev, nm->stat, nm->n, K, epsilon};
    return out_f32 ? launch_obs<float>(nm, a) : launch_obs<double>(nm, a);
}

int mxv_subnorm_rewards(mxv_subnorm *nm, int32_t K, const void *reward_dev, int32_t reward_f32, const uint8_t *terminated_dev,
                        const uint8_t *truncated_dev, void *out_dev, double gamma, double epsilon) {
    if (int rc = sub_checks(nm, K)) return rc;
    if (nm->dim != 1) return sfail(nm, MXV_ERR_INVALID_ARG, "reward statistics need dim == 1 (got %d)", nm->dim);
    if (!reward_dev || !terminated_dev || !truncated_dev || !out_dev) return sfail(nm, MXV_ERR_INVALID_ARG, "NULL device pointer");
    if ((((uintptr_t)reward_dev | (uintptr_t)out_dev) & (reward_f32 ? 3u : 7u)) != 0)
        return sfail(nm, MXV_ERR_INVALID_ARG, "reward / out pointer is not %d-byte aligned", reward_f32 ? 4 : 8);
    SubRewArgs a{reward_dev, terminated_dev, truncated_dev, out_dev, nm->stat, nm->returns, nm->n, K, gamma, epsilon};
    const dim3 grid((unsigned)((a.n + kThreads - 1) / kThreads)), block(kThreads);
    if (reward_f32)
        hipLaunchKernelGGL(subnorm_rew_kernel<float>, grid, block, 0, nm->stream, a);
    else
        hipLaunchKernelGGL(subnorm_rew_kernel<double>, grid, block, 0, nm->stream, a);
    SUB_HIP(nm, hipGetLastError());
    return MXV_OK;
}

}  // extern "C"
