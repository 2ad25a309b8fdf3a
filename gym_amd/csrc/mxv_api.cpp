// mxv_api.cpp — host side of the C ABI declared in include/mxv.h.
// Owns the device-resident env state, the HIP stream, the step/reset bookkeeping (step index t,
// explicit-reset ordinal r), the staging buffers of the *_host convenience calls and the
// hipGraph cache of mxv_rollout.  No torch types, no C++ types cross the boundary.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is opened with dlopen() by mxv_comm_init (no link-time dependency)

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <tuple>
#include <vector>

#include "mxv_kernels.hpp"

using namespace mxv;

namespace {

constexpr int kStateDim[MXV_NUM_ENV_KINDS] = {4, 2, 4, 2, 2};
constexpr int kObsDim[MXV_NUM_ENV_KINDS] = {4, 3, 6, 2, 2};
constexpr int kNumActions[MXV_NUM_ENV_KINDS] = {2, 0, 3, 3, 0};

thread_local std::string g_create_error;

void default_params(int env_id, double *P) {
    std::memset(P, 0, sizeof(double) * MXV_MAX_PARAMS);
    switch (env_id) {
        case MXV_CARTPOLE:  // cartpole.py:90-102
            P[0] = 9.8; P[1] = 1.0; P[2] = 0.1; P[3] = P[2] + P[1]; P[4] = 0.5; P[5] = P[2] * P[4];
            P[6] = 10.0; P[7] = 0.02; P[8] = 12 * 2 * kPi / 360; P[9] = 2.4; P[10] = 0.0;
            break;
        case MXV_PENDULUM:  // pendulum.py:95-101
            P[0] = 8.0; P[1] = 2.0; P[2] = 0.05; P[3] = 10.0; P[4] = 1.0; P[5] = 1.0;
            break;
        case MXV_ACROBOT:  // acrobot.py:143-165
            P[0] = 0.2; P[1] = 1.0; P[2] = 1.0; P[3] = 1.0; P[4] = 1.0; P[5] = 0.5; P[6] = 0.5; P[7] = 1.0;
            P[8] = 4 * kPi; P[9] = 9 * kPi; P[10] = 0.0; P[11] = 0.0;
            break;
        case MXV_MOUNTAINCAR:  // mountain_car.py:103-111
            P[0] = -1.2; P[1] = 0.6; P[2] = 0.07; P[3] = 0.5; P[4] = 0.0; P[5] = 0.001; P[6] = 0.0025;
            break;
        case MXV_MOUNTAINCAR_CONT:  // continuous_mountain_car.py:108-118
            P[0] = -1.0; P[1] = 1.0; P[2] = -1.2; P[3] = 0.6; P[4] = 0.07; P[5] = 0.45; P[6] = 0.0; P[7] = 0.0015;
            break;
        default: break;
    }
}

void default_bounds(int env_id, double *b) {
    switch (env_id) {
        case MXV_CARTPOLE: b[0] = -0.05; b[1] = 0.05; break;  // cartpole.py:199-201
        case MXV_PENDULUM: b[0] = kPi; b[1] = 1.0; break;     // pendulum.py:14-15 (x_init, y_init)
        case MXV_ACROBOT: b[0] = -0.1; b[1] = 0.1; break;     // acrobot.py:185-187
        default: b[0] = -0.6; b[1] = -0.4; break;             // mountain_car.py:159, continuous_mountain_car.py:181
    }
}

}  // namespace

struct mxv_handle {
    mxv_config cfg{};
    int S = 0, O = 0, NA = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    double *state = nullptr;
    // mxv_adopt_obs: the caller's observation buffer doubles as the float32 half of the state (hilo_encode, mxv_kernels.hip)
    float *adopted_obs = nullptr;   // the buffer (caller-owned), or nullptr
    int32_t *lo = nullptr;          // [N][S] the int32 halves (library-owned, allocated on adoption)
    bool hilo = false;              // the state currently lives as (adopted_obs, lo) pairs; `state` holds only escaped components
    void *elapsed = nullptr;    // uint16 [N] when elapsed16, else int32 [N] (always n * 4 bytes allocated)
    bool elapsed16 = false;     // 0 < max_episode_steps <= 65535: the step kernels move 2 instead of 4 bytes each way
    uint32_t *episodes = nullptr;  // [N] resets of each env since seeding = index of its next draw from the reset stream
    uint64_t *seeds = nullptr;  // optional per-env seeds
    uint64_t *t_dev = nullptr;  // device-resident step index for graph replay
    double *ret_state = nullptr, *ret_part = nullptr;   // mxv_set_return_partials: NormalizeReward's running returns and their per-tile sums
    double ret_gamma = 0.0;
    double *obs_part = nullptr;        // mxv_set_obs_partials: where trajectory launches leave the observations' column sums (caller's memory)
    uint32_t *clock_ticket = nullptr;  // finished-workgroup counter of launches that advance the device clock themselves
    bool dev_clock = false;     // mxv_set_device_clock: every launch reads the step index from t_dev and a one-thread kernel advances it
    int32_t *err = nullptr;     // latched kernel error word
    uint64_t base_seed = 0, action_seed = 0;
    uint64_t t = 0;
    uint32_t r = 0;
    bool was_reset = false;
    bool state_injected = false;  // set by mxv_set_state / unusual reset bounds, consumed by the next step launch
    bool state_out_of_range = false;  // Pendulum only: an injected angle beyond the unguarded range stays until a full reset
    LaunchInfo last_launch{-1, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // mxv_last_launch
    bool step_noise = false;      // Acrobot torque_noise_max > 0 (acrobot.py:202-205): the step draws from the step-noise stream
    EnvParams P{};
    bool default_params = true;
    float *ep_acc = nullptr;        // running episode returns when episode statistics are enabled
    uint8_t *beyond = nullptr;      // CartPole + MXV_FLAG_NO_AUTORESET: per-env "terminated before" marks (cartpole.py:169-184)
    float *ep_return_out = nullptr; // caller-attached outputs of the statistics (device)
    int32_t *ep_length_out = nullptr;
    float *st_ep_r = nullptr;       // staging for mxv_step_host / mxv_episode_stats_host
    int32_t *st_ep_l = nullptr;
    double *params_pe = nullptr;  // [MXV_MAX_PARAMS][N] when per-env physics parameters are active
    int param_mode() const { return params_pe ? PM_PER_ENV : (default_params ? PM_DEFAULT : PM_BROADCAST); }
    double bounds[2] = {0, 0};
    // staging for *_host calls
    void *st_actions = nullptr;
    float *st_obs = nullptr, *st_final = nullptr;
    void *st_reward = nullptr;
    uint8_t *st_term = nullptr, *st_trunc = nullptr, *st_mask = nullptr;
    // Small vector envs (all step I/O <= kHostMapLimit bytes) stage through ONE block of pinned, device-mapped host memory:
    // the kernel reads the actions from it and writes its outputs into it over PCIe, so a *_host call is one launch and one
    // stream synchronisation instead of one H2D and five D2H copies — the latency-bound regime of BASELINE configs[0].
    // Larger vector envs keep device staging (a kernel writing 1-byte flags over PCIe is slow) and move it with two DMA
    // copies between the device block and an identically laid out pinned block (the zero-copy calls) or per-array
    // hipMemcpy to the caller's pageable buffers (the copying calls).
    void *hm_block = nullptr;   // pinned host block (exists when hostmap or after mxv_host_io)
    void *dv_block = nullptr;   // device staging block (exists when !hostmap)
    size_t io_total = 0, io_out_off = 0, io_out_bytes = 0, io_act_bytes = 0;
    int32_t *hm_err = nullptr;
    bool hostmap = false;       // the kernels address the pinned block directly
    // large envs: info["final_observation"] travels as packed (index, row) pairs of the finished envs only (final_count_kernel + final_pack_kernel)
    char *fin_dev = nullptr;    // device: count (256 B) | idx int32[N] | rows float[N][O]
    char *fin_host = nullptr;   // pinned mirror
    bool err_in_block = false;  // host steps of a small env: the kernel raises the error word in the pinned I/O block itself
    size_t fin_last = 0;        // finished envs of the previous host step (sizes the speculative DMA)
    bool fin_packed = false;    // mxv_final_packed: host step calls leave the rows packed (no scatter into a dense array)
    // hipGraph cache of mxv_rollout: key = (K, per_step, output pointers)
    using GraphKey = std::tuple<int, int, void *, void *, void *, void *, void *, void *>;
    std::map<GraphKey, hipGraphExec_t> graphs;
    // RCCL communicator of a sharded vector env (mxv_comm_init): the gather runs on a side stream so that the next rollout
    // launch on `stream` overlaps it
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_outputs = nullptr, ev_gathered[2] = {nullptr, nullptr};  // completion of the last two gathers (ring)
    uint64_t gathers = 0;                                                    // gathers issued so far
    // optional second destination of a fused rollout's final tensors (mxv_set_final_snapshot)
    float *snap_obs = nullptr;
    void *snap_reward = nullptr;
    uint8_t *snap_term = nullptr, *snap_trunc = nullptr;
    hipEvent_t ev_wait = nullptr;   // mxv_wait_stream
    hipEvent_t ev_mixed = nullptr;  // orders this handle's stream against a mixed-batch launch issued on another handle's stream
    std::string error;

    size_t action_bytes() const {
        if (NA > 0) return (cfg.flags & MXV_FLAG_ACTION_I32) ? 4 : 8;
        return 4;
    }
    size_t reward_bytes() const { return (cfg.flags & MXV_FLAG_REWARD_F32) ? 4 : 8; }
};

namespace {

int fail(mxv_handle *h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h)
        h->error = buf;
    else
        g_create_error = buf;
    return code;
}

#define MXV_HIP(h, expr)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return fail((h), MXV_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define MXV_CHECK_HANDLE(h) \
    if (!(h)) return fail(nullptr, MXV_ERR_INVALID_ARG, "NULL handle")

int use_device(mxv_handle *h) {
    MXV_HIP(h, hipSetDevice(h->cfg.device));
    return MXV_OK;
}

int check_latched(mxv_handle *h) {
    int32_t e = 0;
    MXV_HIP(h, hipMemcpyAsync(&e, h->err, sizeof e, hipMemcpyDeviceToHost, h->stream));
    MXV_HIP(h, hipStreamSynchronize(h->stream));
    if (e != 0) {
        MXV_HIP(h, hipMemsetAsync(h->err, 0, sizeof(int32_t), h->stream));
        if (e & 1)
            return fail(h, MXV_ERR_INVALID_ACTION, "discrete action outside [0, %d) (Discrete.contains assert)", h->NA);
        return fail(h, MXV_ERR_INVALID_ARG, "kernel error word 0x%x", e);
    }
    return MXV_OK;
}

// Host steps of a small env (the kernel reads and writes the pinned I/O block over PCIe): the error word lives in that block too,
// so the call is one launch and one synchronisation — no 4-byte copy to fetch a device-side latch (a quarter of the 17 us).
struct ErrInBlock {
    mxv_handle *h;
    explicit ErrInBlock(mxv_handle *hh) : h(hh) { h->err_in_block = h->hostmap; }
    ~ErrInBlock() { h->err_in_block = false; }
};

int take_block_error(mxv_handle *h) {  // after the stream drained
    const int32_t e = *h->hm_err;
    if (e == 0) return MXV_OK;
    *h->hm_err = 0;
    if (e & 1) return fail(h, MXV_ERR_INVALID_ACTION, "discrete action outside [0, %d) (Discrete.contains assert)", h->NA);
    return fail(h, MXV_ERR_INVALID_ARG, "kernel error word 0x%x", e);
}

// The handle has just launched `delta` vector steps (negative: a step that is being taken back).  Device-clock mode: the device word
// follows through a one-thread kernel on the same stream — captured with the launch when a caller's hipGraph is being recorded.
constexpr int64_t kClockInKernelEnvs = 65536;

int clock_add(mxv_handle *h, int64_t delta) {
    h->t += (uint64_t)delta;
    if (h->dev_clock) MXV_HIP(h, launch_add_word(h->t_dev, (uint64_t)delta, h->stream));
    return MXV_OK;
}

void fill_step_args(mxv_handle *h, StepArgs &a) {
    a.state = h->state;
    a.elapsed = h->elapsed;
    a.elapsed16 = h->elapsed16 ? 1 : 0;
    a.episodes = h->episodes;
    a.seeds = h->seeds;
    a.t_dev = nullptr;
    a.err = h->err_in_block ? h->hm_err : h->err;
    a.n = h->cfg.num_envs;
    a.env0 = (uint64_t)h->cfg.env_offset;
    a.base_seed = h->base_seed;
    a.action_seed = h->action_seed;
    a.t = h->t;
    if (h->dev_clock) {  // the step index lives on the device (a caller's hipGraph replays this launch at other step indices)
        a.t_dev = h->t_dev;
        a.t = 0;
    }
    a.b0 = h->bounds[0];
    a.b1 = h->bounds[1];
    a.max_steps = h->cfg.max_episode_steps;
    a.flags = h->cfg.flags;
    a.K = 1;
    a.state_injected = (h->state_injected || h->state_out_of_range) ? 1 : 0;
    a.step_noise = h->step_noise ? 1 : 0;
    a.slice = 0;
    a.act_slice = 0;
    a.params_pe = h->params_pe;
    a.beyond = h->beyond;
    a.ep_acc = h->ep_acc;
    a.ep_return_out = h->ep_acc ? h->ep_return_out : nullptr;
    a.ep_length_out = h->ep_acc ? h->ep_length_out : nullptr;
    a.snap_obs = nullptr;   // only the K-step calls deposit a snapshot (fused_launch, mxv_rollout_mixed)
    a.snap_reward = nullptr;
    a.snap_terminated = a.snap_truncated = nullptr;
    a.P = h->P;
}

// launches that never form the fused batch moments: with a buffer attached they fail instead of leaving it stale
int no_partials_here(mxv_handle *h, const char *what) {
    if (h->obs_part || h->ret_part)
        return fail(h, MXV_ERR_UNSUPPORTED, "partial sums are attached (mxv_set_obs_partials / mxv_set_return_partials), but %s does not produce "
                                            "them (only sampled [K][N] trajectory launches of mxv_rollout, MXV_ROLLOUT_FUSED, do): detach first", what);
    return MXV_OK;
}

// A launch recorded into a caller's hipGraph with the step index as a by-value kernel argument would repeat that index — the same action
// and noise draws — on every replay, silently.  With the device clock (mxv_set_device_clock) the index is read from device memory and
// replays continue the streams; without it, recording is refused.  
int no_capture_without_clock(mxv_handle *h) {
    if (h->dev_clock) return MXV_OK;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(h->stream, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone)
        return fail(h, MXV_ERR_UNSUPPORTED,
                    "the handle's stream is being captured into a hipGraph but the step index travels by value: replays would repeat the same "
                    "draws — call mxv_set_device_clock(h, 1) before recording");
    return MXV_OK;
}

// The state back in its fp64 array (a no-op unless the last steps ran in the observation-carries-state form): called by everything that
// reads or writes h->state except the single step that can stay in that form.
int ensure_f64(mxv_handle *h) {
    if (!h->hilo) return MXV_OK;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(h->stream, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone)
        return fail(h, MXV_ERR_UNSUPPORTED, "the state lives in the adopted observation buffer (mxv_adopt_obs) and this call needs it back in "
                                            "fp64: that conversion cannot be part of a recorded hipGraph — release the buffer before recording");
    MXV_HIP(h, launch_hilo_join(h->cfg.env_id, h->adopted_obs, h->lo, h->state, h->cfg.num_envs, h->stream));
    h->hilo = false;
    return MXV_OK;
}

// Every caller-owned tensor a launch touches must sit on its element's natural boundary (observations: the row's vector width — the kernels
// store them as float4 / float2).  An odd address would not fault on this device (unaligned global access is enabled) but tears every
// coalesced burst, and it is a caller bug either way: refused up front with the name of the tensor (tests/c_consumer/abi_fuzz.c).
int check_aligned(mxv_handle *h, const void *p, size_t bytes, const char *what) {
    if (p && ((uintptr_t)p & (bytes - 1)) != 0)
        return fail(h, MXV_ERR_INVALID_ARG, "%s pointer %p is not %zu-byte aligned", what, p, bytes);
    return MXV_OK;
}
int check_step_buffers(mxv_handle *h, const StepArgs &a) {
    const size_t obs_al = h->O == 4 ? 16 : ((h->O % 2 == 0) ? 8 : 4);
    const size_t act_al = h->NA > 0 ? ((a.flags & MXV_FLAG_ACTION_I32) ? 4 : 8) : 4, rew_al = (a.flags & MXV_FLAG_REWARD_F32) ? 4 : 8;
    if (int rc = check_aligned(h, a.obs, obs_al, "obs")) return rc;
    if (int rc = check_aligned(h, a.final_obs, obs_al, "final_obs")) return rc;
    if (int rc = check_aligned(h, a.snap_obs, obs_al, "snapshot obs")) return rc;
    if (int rc = check_aligned(h, a.reward, rew_al, "reward")) return rc;
    if (int rc = check_aligned(h, a.snap_reward, rew_al, "snapshot reward")) return rc;
    if (int rc = check_aligned(h, a.actions, act_al, "actions")) return rc;
    if (int rc = check_aligned(h, a.actions_out, act_al, "actions_out")) return rc;
    if (int rc = check_aligned(h, a.ep_return_out, 4, "episode return")) return rc;
    if (int rc = check_aligned(h, a.ep_length_out, 4, "episode length")) return rc;
    if (int rc = check_aligned(h, a.obs_part, 8, "obs partials")) return rc;
    if (int rc = check_aligned(h, a.ret_part, 8, "return partials")) return rc;
    return check_aligned(h, a.ret_state, 8, "returns state");
}

int do_step(mxv_handle *h, const void *actions, void *actions_out, float *obs, void *reward, uint8_t *term,
            uint8_t *trunc, float *final_obs) {
    if (!h->was_reset)
        return fail(h, MXV_ERR_RESET_NEEDED, "Cannot call step before calling reset (gym.error.ResetNeeded)");
    if (int rc = no_partials_here(h, "a single step")) return rc;
    if (!obs) return fail(h, MXV_ERR_INVALID_ARG, "obs pointer is NULL");
    if (int rc = use_device(h)) return rc;
    if (int rc = no_capture_without_clock(h)) return rc;
    // the observation-carries-state form (mxv_adopt_obs): this step reads the previous observations from `obs` and leaves int32 residuals
    const bool hilo_step = h->adopted_obs != nullptr && obs == h->adopted_obs && h->param_mode() == PM_DEFAULT && !h->step_noise;
    if (!hilo_step) {
        if (int rc = ensure_f64(h)) return rc;
    } else if (!h->hilo) {   // first step in that form: split the fp64 state (the buffer receives float32(state) = the current observations)
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(h->stream, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone)
            return fail(h, MXV_ERR_UNSUPPORTED, "take one step outside the capture first: the first step after mxv_adopt_obs / reset converts the "
                                                "state, which must not be replayed");
        MXV_HIP(h, launch_hilo_split(h->cfg.env_id, h->state, h->adopted_obs, h->lo, h->state, h->cfg.num_envs, h->stream));
        h->hilo = true;
    }
    StepArgs a{};
    fill_step_args(h, a);
    a.actions = actions;
    a.actions_out = actions_out;
    a.obs = obs;
    a.reward = reward;
    a.terminated = term;
    a.truncated = trunc;
    a.final_obs = final_obs;
    // device clock, batches of at most kClockInKernelEnvs envs (a few hundred workgroups: one returning atomic each is nothing; at
    // 2^20 envs thousands of same-address atomics would cost more than the one-thread kernel they replace): the launch advances it
    const bool self_clock = h->dev_clock && h->cfg.num_envs <= kClockInKernelEnvs && h->param_mode() == PM_DEFAULT && !hilo_step;   // (step_kernel<..., CLOCK = true> exists for these)
    if (self_clock) {
        a.clock_out = h->t_dev;
        a.clock_ticket = h->clock_ticket;
    }
    if (int rc = check_step_buffers(h, a)) return rc;
    if (hilo_step) {
        a.hi_in = h->adopted_obs;
        a.lo = h->lo;
        MXV_HIP(h, launch_hilo_step(h->cfg.env_id, a, h->stream, &h->last_launch));
    } else {
        MXV_HIP(h, launch_step(h->cfg.env_id, h->param_mode(), a, h->stream, &h->last_launch));
    }
    h->state_injected = false;
    if (self_clock) {
        h->t += 1;
        return MXV_OK;
    }
    return clock_add(h, 1);
}

int parse_bounds(mxv_handle *h, const double *b, double *out) {
    if (!b) {
        default_bounds(h->cfg.env_id, out);
        return MXV_OK;
    }
    if (std::isnan(b[0]) || std::isnan(b[1])) return fail(h, MXV_ERR_INVALID_ARG, "reset bounds are NaN");
    if (h->cfg.env_id != MXV_PENDULUM && b[0] > b[1])  // classic_control/utils.py:41-44
        return fail(h, MXV_ERR_INVALID_ARG, "Lower bound (%g) must be lower than higher bound (%g).", b[0], b[1]);
    out[0] = b[0];
    out[1] = b[1];
    return MXV_OK;
}

int do_reset(mxv_handle *h, const uint8_t *mask_dev, const double *bounds, float *obs_dev) {
    if (int rc = use_device(h)) return rc;
    if (int rc = ensure_f64(h)) return rc;
    double b[2];
    if (int rc = parse_bounds(h, bounds, b)) return rc;
    h->r += 1;
    ResetArgs a{};
    a.state = h->state;
    a.elapsed = h->elapsed;
    a.elapsed16 = h->elapsed16 ? 1 : 0;
    a.episodes = h->episodes;
    a.obs = obs_dev;
    a.mask = mask_dev;
    a.seeds = h->seeds;
    a.ep_acc = h->ep_acc;
    a.beyond = h->beyond;
    a.n = h->cfg.num_envs;
    a.env0 = (uint64_t)h->cfg.env_offset;
    a.base_seed = h->base_seed;
    a.b0 = b[0];
    a.b1 = b[1];
    if (int rc = check_aligned(h, a.obs, h->O == 4 ? 16 : ((h->O % 2 == 0) ? 8 : 4), "obs")) return rc;
    MXV_HIP(h, launch_reset(h->cfg.env_id, a, h->stream));
    h->was_reset = true;
    if (mask_dev == nullptr) h->state_out_of_range = false;  // every env re-drawn
    // CartPole's range-reduction-free sin/cos (rollout fast path, SAFE = false) assumes |theta| <= pi/4 on entry; reset bounds
    // from reset(options={"low","high"}) beyond that break the induction exactly like an injected state does: the next
    // fused launch takes the SAFE instantiation (every such env terminates in its first step and autoresets with the
    // default bounds, so one launch is enough).
    const double widest = std::fmax(std::fabs(b[0]), std::fabs(b[1]));
    if ((h->cfg.env_id == MXV_CARTPOLE && widest > 0.78539816339744830962) || widest > 65536.0)
        h->state_injected = true;  // (other kinds: the unguarded medium-range sin/cos of the fused rollout, mx_sincos<false>)
    if (h->cfg.env_id == MXV_PENDULUM && widest > 65536.0) h->state_out_of_range = true;  // the angle is never wrapped: see mxv_set_state
    return MXV_OK;
}

// Packed record of the envs that finished a host step (device buffer and its pinned mirror share the offsets):
//   count (256 B) | idx int32[N] | rows float[N][O] | ep_return float[N] | ep_length int32[N]      (+ device only: chunk counts)
struct FinLayout {
    size_t idx, rows, ep_r, ep_l, host_bytes, chunks, dev_bytes;
    FinLayout(size_t n, size_t O) {
        idx = 256;
        rows = idx + n * sizeof(int32_t);
        ep_r = rows + n * O * sizeof(float);
        ep_l = ep_r + n * sizeof(float);
        host_bytes = ep_l + n * sizeof(int32_t);
        chunks = host_bytes;
        dev_bytes = chunks + (size_t)compact_chunks((int64_t)n) * sizeof(int32_t);
    }
};

// Up to this size the step kernels address the pinned block directly; above it they use device staging.
constexpr size_t kHostMapLimit = 2 * 1024 * 1024;

// Layout of one I/O block (pinned and device copies share it): actions | final_obs | obs | reward | term | trunc | mask | err
// (final_obs sits outside the contiguous per-step output region obs .. trunc: for large envs it is not copied densely)
int ensure_staging(mxv_handle *h, bool want_pinned = false) {
    const size_t n = (size_t)h->cfg.num_envs;
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t b_act = up(n * 8), b_obs = up(n * h->O * sizeof(float)), b_rew = up(n * 8), b_flag = up(n);
    if (!h->st_obs) {
        h->io_total = b_act + 2 * b_obs + b_rew + 3 * b_flag + 256;
        h->io_act_bytes = n * h->action_bytes();
        h->io_out_off = b_act + b_obs;
        h->io_out_bytes = b_obs + b_rew + 2 * b_flag;
        h->hostmap = h->io_total <= kHostMapLimit;
        if (h->hostmap) {
            MXV_HIP(h, hipHostMalloc(&h->hm_block, h->io_total, hipHostMallocDefault));
        } else {
            MXV_HIP(h, hipMalloc(&h->dv_block, h->io_total));
        }
        char *p = (char *)(h->hostmap ? h->hm_block : h->dv_block);
        h->st_actions = p; p += b_act;
        h->st_final = (float *)p; p += b_obs;
        h->st_obs = (float *)p; p += b_obs;
        h->st_reward = p; p += b_rew;
        h->st_term = (uint8_t *)p; p += b_flag;
        h->st_trunc = (uint8_t *)p; p += b_flag;
        h->st_mask = (uint8_t *)p; p += b_flag;
        if (h->hostmap) {
            h->hm_err = (int32_t *)p;
            *h->hm_err = 0;
        } else {
            const FinLayout L(n, (size_t)h->O);
            MXV_HIP(h, hipMalloc((void **)&h->fin_dev, L.dev_bytes));
            MXV_HIP(h, hipHostMalloc((void **)&h->fin_host, L.host_bytes, hipHostMallocDefault));
        }
    }
    if (want_pinned && !h->hm_block) {  // large env: pinned mirror of the device block for the zero-copy calls
        MXV_HIP(h, hipHostMalloc(&h->hm_block, h->io_total, hipHostMallocDefault));
        h->hm_err = (int32_t *)((char *)h->hm_block + h->io_total - 256);
        *h->hm_err = 0;
    }
    return MXV_OK;
}

// Actions of a host step, caller's (pageable) array -> device staging.  One plain hipMemcpyAsync: the runtime stages pageable
// memory through its own pinned buffers, and does it better than a hand-rolled version — copying 1-MiB slices into a pinned
// buffer of the library's and queueing one DMA per slice (so that slice k's DMA overlaps slice k+1's memcpy) made the 2^20-env
// CartPole step 130 us SLOWER (1054 vs 917 us, profiles/r2/r03g_upload_ab.txt); narrowing int64 discrete actions to one byte each on
// the host (AVX-512 vpmovqb loop with the range check folded in) so that 1 MB instead of 8 crosses the link lost as well, 1006 vs
// 940 us: one core reading the caller's 8 MB takes longer than the DMA time it saves.
int upload_actions(mxv_handle *h, const void *actions_host) {
    const size_t bytes = (size_t)h->cfg.num_envs * h->action_bytes();
    if (h->hostmap)
        std::memcpy(h->st_actions, actions_host, bytes);
    else
        MXV_HIP(h, hipMemcpyAsync(h->st_actions, actions_host, bytes, hipMemcpyHostToDevice, h->stream));
    return MXV_OK;
}

// Large envs: pack the final_obs rows (and, with episode statistics on, the return and length) of the envs that finished this
// step on the device — ascending env order — and bring count | indices, rows and statistics over in small DMAs sized
// speculatively from the previous step's count; the rest follows only if more envs finished.
int queue_final_rows(mxv_handle *h, size_t *first_out) {
    const size_t n = (size_t)h->cfg.num_envs, O = (size_t)h->O;
    const FinLayout L(n, O);
    const bool stats = h->ep_acc != nullptr;
    CompactArgs c{};
    c.terminated = h->st_term;
    c.truncated = h->st_trunc;
    c.final_obs = h->st_final;
    c.count = (int32_t *)h->fin_dev;
    c.chunk_counts = (int32_t *)(h->fin_dev + L.chunks);
    c.idx = (int32_t *)(h->fin_dev + L.idx);
    c.rows = (float *)(h->fin_dev + L.rows);
    c.ep_return_in = stats ? h->st_ep_r : nullptr;
    c.ep_length_in = stats ? h->st_ep_l : nullptr;
    c.ep_return = (float *)(h->fin_dev + L.ep_r);
    c.ep_length = (int32_t *)(h->fin_dev + L.ep_l);
    c.n = (int64_t)n;
    MXV_HIP(h, launch_compact_final(h->O, c, h->stream));
    // speculative size of the first DMA: what finished last step plus a margin (episode ends arrive at a smooth rate)
    const size_t first = std::min(n, std::max<size_t>(1024, h->fin_last + h->fin_last / 4 + 1024));
    MXV_HIP(h, hipMemcpyAsync(h->fin_host, h->fin_dev, 256 + first * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    MXV_HIP(h, hipMemcpyAsync(h->fin_host + L.rows, h->fin_dev + L.rows, first * O * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (stats) {
        MXV_HIP(h, hipMemcpyAsync(h->fin_host + L.ep_r, h->fin_dev + L.ep_r, first * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        MXV_HIP(h, hipMemcpyAsync(h->fin_host + L.ep_l, h->fin_dev + L.ep_l, first * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    }
    *first_out = first;
    return MXV_OK;
}

// After the stream drained: fetch what the speculative DMA missed, scatter if asked.
int finish_final_rows(mxv_handle *h, size_t first, float *final_host) {
    const size_t n = (size_t)h->cfg.num_envs, O = (size_t)h->O;
    const FinLayout L(n, O);
    const int32_t *p_idx = (const int32_t *)(h->fin_host + L.idx);
    const float *p_rows = (const float *)(h->fin_host + L.rows);
    const size_t count = (size_t) * (const int32_t *)h->fin_host;
    h->fin_last = count;
    if (count > first) {
        const size_t more = count - first;
        auto tail = [&](size_t off, size_t elem) {
            return hipMemcpyAsync(h->fin_host + off + first * elem, h->fin_dev + off + first * elem, more * elem, hipMemcpyDeviceToHost,
                                  h->stream);
        };
        MXV_HIP(h, tail(L.idx, sizeof(int32_t)));
        MXV_HIP(h, tail(L.rows, O * sizeof(float)));
        if (h->ep_acc) {
            MXV_HIP(h, tail(L.ep_r, sizeof(float)));
            MXV_HIP(h, tail(L.ep_l, sizeof(int32_t)));
        }
        MXV_HIP(h, hipStreamSynchronize(h->stream));
    }
    // scattering ~5 % of 2^20 rows into a dense array is 0.5 ms of cache misses: callers that can consume the packed pairs
    // (mxv_final_packed) skip it
    if (final_host)
        for (size_t i = 0; i < count; ++i) std::memcpy(final_host + (size_t)p_idx[i] * O, p_rows + i * O, O * sizeof(float));
    return MXV_OK;
}

int fetch_final_rows(mxv_handle *h, float *final_host) {
    size_t first = 0;
    if (int rc = queue_final_rows(h, &first)) return rc;
    MXV_HIP(h, hipStreamSynchronize(h->stream));
    return finish_final_rows(h, first, final_host);
}

void free_graphs(mxv_handle *h) {
    for (auto &kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
    h->graphs.clear();
}

}  // namespace

extern "C" {

const char *mxv_version(void) { return "mxv 0.6.0 (gfx950)"; }

int mxv_env_dims(int32_t env_id, int32_t *state_dim, int32_t *obs_dim, int32_t *num_actions) {
    if (env_id < 0 || env_id >= MXV_NUM_ENV_KINDS) return MXV_ERR_INVALID_ARG;
    if (state_dim) *state_dim = kStateDim[env_id];
    if (obs_dim) *obs_dim = kObsDim[env_id];
    if (num_actions) *num_actions = kNumActions[env_id];
    return MXV_OK;
}

int mxv_default_params(int32_t env_id, double *params_host) {
    if (env_id < 0 || env_id >= MXV_NUM_ENV_KINDS || !params_host) return MXV_ERR_INVALID_ARG;
    default_params(env_id, params_host);
    return MXV_OK;
}

int mxv_default_reset_bounds(int32_t env_id, double *bounds2_host) {
    if (env_id < 0 || env_id >= MXV_NUM_ENV_KINDS || !bounds2_host) return MXV_ERR_INVALID_ARG;
    default_bounds(env_id, bounds2_host);
    return MXV_OK;
}

const char *mxv_last_error(const mxv_handle *h) { return h ? h->error.c_str() : g_create_error.c_str(); }

int mxv_create(const mxv_config *cfg, mxv_handle **out) {
    if (!cfg || !out) return fail(nullptr, MXV_ERR_INVALID_ARG, "NULL config or output pointer");
    *out = nullptr;
    if (cfg->env_id < 0 || cfg->env_id >= MXV_NUM_ENV_KINDS)
        return fail(nullptr, MXV_ERR_INVALID_ARG, "unknown env_id %d", cfg->env_id);
    if (cfg->num_envs <= 0) return fail(nullptr, MXV_ERR_INVALID_ARG, "num_envs must be positive (got %lld)", (long long)cfg->num_envs);
    // the fused kernels address one step's slice of every output array with 32-bit byte offsets (16 B per env at most)
    if (cfg->num_envs > MXV_MAX_NUM_ENVS)
        return fail(nullptr, MXV_ERR_INVALID_ARG, "num_envs %lld exceeds the per-handle maximum of %lld: shard the vector env (env_offset)",
                    (long long)cfg->num_envs, (long long)MXV_MAX_NUM_ENVS);
    if (cfg->env_offset < 0 || cfg->env_offset % MXV_ENV_ALIGN != 0)
        return fail(nullptr, MXV_ERR_INVALID_ARG, "env_offset must be a non-negative multiple of %d", MXV_ENV_ALIGN);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, MXV_ERR_HIP, "no HIP device available (%s): the engine has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, MXV_ERR_INVALID_ARG, "device %d out of range (%d devices)", cfg->device, ndev);
    mxv_handle *h = new (std::nothrow) mxv_handle();
    if (!h) return fail(nullptr, MXV_ERR_INVALID_ARG, "out of host memory");
    h->cfg = *cfg;
    h->S = kStateDim[cfg->env_id];
    h->O = kObsDim[cfg->env_id];
    h->NA = kNumActions[cfg->env_id];
    h->base_seed = cfg->seed;
    h->action_seed = cfg->action_seed;
    default_params(cfg->env_id, h->P.p);
    default_bounds(cfg->env_id, h->bounds);
    const size_t n = (size_t)cfg->num_envs;
#define MXV_CREATE_HIP(expr)                                                                \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            fail(nullptr, MXV_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));             \
            mxv_destroy(h);                                                                 \
            return MXV_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)
    MXV_CREATE_HIP(hipSetDevice(cfg->device));
    MXV_CREATE_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
    MXV_CREATE_HIP(hipMalloc((void **)&h->state, n * h->S * sizeof(double)));
    h->elapsed16 = cfg->max_episode_steps > 0 && cfg->max_episode_steps <= 65535 && !getenv("MXV_ELAPSED32");   // (A/B hook)
    MXV_CREATE_HIP(hipMalloc((void **)&h->elapsed, n * sizeof(int32_t)));
    MXV_CREATE_HIP(hipMalloc((void **)&h->episodes, n * sizeof(uint32_t)));
    MXV_CREATE_HIP(hipMalloc((void **)&h->t_dev, sizeof(uint64_t)));
    MXV_CREATE_HIP(hipMalloc((void **)&h->clock_ticket, sizeof(uint32_t)));
    MXV_CREATE_HIP(hipMalloc((void **)&h->err, sizeof(int32_t)));
    MXV_CREATE_HIP(hipMemsetAsync(h->state, 0, n * h->S * sizeof(double), h->stream));
    MXV_CREATE_HIP(hipMemsetAsync(h->elapsed, 0, n * sizeof(int32_t), h->stream));
    MXV_CREATE_HIP(hipMemsetAsync(h->episodes, 0, n * sizeof(uint32_t), h->stream));
    MXV_CREATE_HIP(hipMemsetAsync(h->t_dev, 0, sizeof(uint64_t), h->stream));
    MXV_CREATE_HIP(hipMemsetAsync(h->clock_ticket, 0, sizeof(uint32_t), h->stream));
    MXV_CREATE_HIP(hipMemsetAsync(h->err, 0, sizeof(int32_t), h->stream));
    if (cfg->env_id == MXV_CARTPOLE && (cfg->flags & MXV_FLAG_NO_AUTORESET)) {   // steps_beyond_terminated marks (cartpole.py:169-184)
        MXV_CREATE_HIP(hipMalloc((void **)&h->beyond, n));
        MXV_CREATE_HIP(hipMemsetAsync(h->beyond, 0, n, h->stream));
    }
    MXV_CREATE_HIP(hipStreamSynchronize(h->stream));
#undef MXV_CREATE_HIP
    *out = h;
    return MXV_OK;
}

int mxv_destroy(mxv_handle *h) {
    if (!h) return MXV_OK;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    (void)mxv_comm_destroy(h);
    if (h->ev_mixed) (void)hipEventDestroy(h->ev_mixed);
    if (h->ev_wait) (void)hipEventDestroy(h->ev_wait);
    free_graphs(h);
    if (h->hm_block) (void)hipHostFree(h->hm_block);
    if (h->fin_host) (void)hipHostFree(h->fin_host);
    if (h->fin_dev) (void)hipFree(h->fin_dev);
    void *bufs[] = {h->lo, h->state, h->elapsed, h->episodes, h->seeds, h->t_dev, h->clock_ticket, h->err, h->params_pe, h->ep_acc, h->st_ep_r, h->st_ep_l, h->dv_block,
                    h->beyond};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return MXV_OK;
}

int mxv_seed(mxv_handle *h, uint64_t base_seed, const uint64_t *per_env_seeds_host) {
    MXV_CHECK_HANDLE(h);
    if (int rc = use_device(h)) return rc;
    MXV_HIP(h, hipStreamSynchronize(h->stream));
    free_graphs(h);  // captured kernel arguments hold the old seeds
    h->base_seed = base_seed;
    h->t = 0;
    h->r = 0;
    if (h->dev_clock) MXV_HIP(h, launch_set_word(h->t_dev, 0, h->stream));
    // a (re)seeded env starts its reset stream from the beginning: reset ordinals back to 0
    MXV_HIP(h, hipMemsetAsync(h->episodes, 0, (size_t)h->cfg.num_envs * sizeof(uint32_t), h->stream));
    if (per_env_seeds_host) {
        const size_t bytes = (size_t)h->cfg.num_envs * sizeof(uint64_t);
        if (!h->seeds) MXV_HIP(h, hipMalloc((void **)&h->seeds, bytes));
        MXV_HIP(h, hipMemcpyAsync(h->seeds, per_env_seeds_host, bytes, hipMemcpyHostToDevice, h->stream));
        MXV_HIP(h, hipStreamSynchronize(h->stream));
    } else if (h->seeds) {
        MXV_HIP(h, hipStreamSynchronize(h->stream));
        MXV_HIP(h, hipFree(h->seeds));
        h->seeds = nullptr;
    }
    return MXV_OK;
}

int mxv_seed_actions(mxv_handle *h, uint64_t action_seed) {
    MXV_CHECK_HANDLE(h);
    if (int rc = use_device(h)) return rc;
    MXV_HIP(h, hipStreamSynchronize(h->stream));
    free_graphs(h);
    h->action_seed = action_seed;
    return MXV_OK;
}

int mxv_reset(mxv_handle *h, const uint8_t *mask_dev, const double *bounds2_host, float *obs_dev) {
    MXV_CHECK_HANDLE(h);
    return do_reset(h, mask_dev, bounds2_host, obs_dev);
}

int mxv_step(mxv_handle *h, const void *actions_dev, float *obs_dev, void *reward_dev, uint8_t *terminated_dev,
             uint8_t *truncated_dev, float *final_obs_dev) {
    MXV_CHECK_HANDLE(h);
    if (!actions_dev) return fail(h, MXV_ERR_INVALID_ARG, "actions pointer is NULL (use mxv_step_sampled)");
    return do_step(h, actions_dev, nullptr, obs_dev, reward_dev, terminated_dev, truncated_dev, final_obs_dev);
}

int mxv_step_sampled(mxv_handle *h, void *actions_out_dev, float *obs_dev, void *reward_dev, uint8_t *terminated_dev,
                     uint8_t *truncated_dev, float *final_obs_dev) {
    MXV_CHECK_HANDLE(h);
    return do_step(h, nullptr, actions_out_dev, obs_dev, reward_dev, terminated_dev, truncated_dev, final_obs_dev);
}

}  // extern "C" (re-opened below)

namespace {

int rollout_checks(mxv_handle *h, int32_t K, const float *obs_dev) {
    if (K <= 0) return fail(h, MXV_ERR_INVALID_ARG, "K must be positive");
    if (!h->was_reset)
        return fail(h, MXV_ERR_RESET_NEEDED, "Cannot call step before calling reset (gym.error.ResetNeeded)");
    if (!obs_dev) return fail(h, MXV_ERR_INVALID_ARG, "obs pointer is NULL");
    if (int rc = use_device(h)) return rc;
    if (int rc = ensure_f64(h)) return rc;     // K-step launches keep the state in registers: they start from (and leave) the fp64 array
    return no_capture_without_clock(h);
}

// Launch paths whose kernel does not write the snapshot itself: copy the last step's outputs (device to device, same stream).
int copy_final_snapshot(mxv_handle *h, int32_t K, int32_t per_step, const float *obs, const void *reward, const uint8_t *term,
                        const uint8_t *trunc) {
    const size_t n = (size_t)h->cfg.num_envs, last = per_step ? (size_t)(K - 1) * n : 0;
    MXV_HIP(h, hipMemcpyAsync(h->snap_obs, obs + last * h->O, n * h->O * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    if (h->snap_reward && reward)
        MXV_HIP(h, hipMemcpyAsync(h->snap_reward, (const char *)reward + last * h->reward_bytes(), n * h->reward_bytes(),
                                  hipMemcpyDeviceToDevice, h->stream));
    if (h->snap_term && term) MXV_HIP(h, hipMemcpyAsync(h->snap_term, term + last, n, hipMemcpyDeviceToDevice, h->stream));
    if (h->snap_trunc && trunc) MXV_HIP(h, hipMemcpyAsync(h->snap_trunc, trunc + last, n, hipMemcpyDeviceToDevice, h->stream));
    return MXV_OK;
}

// One launch, K steps, env state in registers between steps.
int fused_launch(mxv_handle *h, int32_t K, int32_t per_step, const void *actions_tape, void *actions_out, float *obs,
                 void *reward, uint8_t *term, uint8_t *trunc, float *final_obs) {
    StepArgs a{};
    fill_step_args(h, a);
    a.K = K;
    a.slice = per_step ? h->cfg.num_envs : 0;
    a.act_slice = actions_tape ? h->cfg.num_envs : 0;
    a.actions = actions_tape;
    a.actions_out = actions_out;
    a.obs = obs;
    a.reward = reward;
    a.terminated = term;
    a.truncated = trunc;
    a.final_obs = final_obs;
    if (h->obs_part || h->ret_part) {  // fused batch moments (mxv_set_obs_partials / mxv_set_return_partials): produced, or the call fails
        a.obs_part = h->obs_part;
        if (h->ret_part) {
            a.ret_part = h->ret_part;
            a.ret_state = h->ret_state;
            a.ret_gamma = h->ret_gamma;
        }
        if (!launch_step_supports_stats(h->cfg.env_id, h->param_mode(), a))
            return fail(h, MXV_ERR_UNSUPPORTED, "partial sums are attached (mxv_set_obs_partials / mxv_set_return_partials), but this launch cannot produce them: they "
                                                "exist for sampled [K][N] trajectory launches (K >= 2) with every per-step output, default physics "
                                                "parameters, autoreset and a state the engine produced itself; detach (NULL) and use mxv_norm_obs_sums / mxv_norm_reward_sums");
    }
    const bool in_kernel = h->snap_obs && launch_step_is_rollout(h->param_mode(), a);
    if (in_kernel) {
        a.snap_obs = h->snap_obs;
        a.snap_reward = h->snap_reward;
        a.snap_terminated = h->snap_term;
        a.snap_truncated = h->snap_trunc;
    }
    if (int rc = check_step_buffers(h, a)) return rc;
    MXV_HIP(h, launch_step(h->cfg.env_id, h->param_mode(), a, h->stream, &h->last_launch));
    h->state_injected = false;
    if (int rc = clock_add(h, K)) return rc;
    if (h->snap_obs && !in_kernel) return copy_final_snapshot(h, K, per_step, obs, reward, term, trunc);
    return MXV_OK;
}

}  // namespace

extern "C" {

int mxv_rollout(mxv_handle *h, int32_t K, int32_t per_step, int32_t mode, void *actions_out_dev, float *obs_dev,
                void *reward_dev, uint8_t *terminated_dev, uint8_t *truncated_dev, float *final_obs_dev) {
    MXV_CHECK_HANDLE(h);
    if (int rc = rollout_checks(h, K, obs_dev)) return rc;
    if (mode == MXV_ROLLOUT_FUSED)
        return fused_launch(h, K, per_step, nullptr, actions_out_dev, obs_dev, reward_dev, terminated_dev,
                            truncated_dev, final_obs_dev);
    if (mode != MXV_ROLLOUT_EAGER && mode != MXV_ROLLOUT_GRAPH) return fail(h, MXV_ERR_INVALID_ARG, "unknown rollout mode %d", mode);
    if (int rc = no_partials_here(h, "a rollout of single-step launches (MXV_ROLLOUT_EAGER / _GRAPH)")) return rc;
    const size_t n = (size_t)h->cfg.num_envs;
    auto slice = [&](void *p, size_t elem_bytes, int k) -> void * {
        if (!p) return nullptr;
        return per_step ? (void *)((char *)p + (size_t)k * n * elem_bytes) : p;
    };
    {
        StepArgs a0{};
        fill_step_args(h, a0);
        a0.actions_out = actions_out_dev; a0.obs = obs_dev; a0.reward = reward_dev; a0.final_obs = final_obs_dev;
        if (int rc = check_step_buffers(h, a0)) return rc;
    }
    auto launch_k = [&](int k, const uint64_t *t_dev, uint64_t t) -> hipError_t {
        StepArgs a{};
        fill_step_args(h, a);
        a.t_dev = t_dev;
        a.t = t;
        a.actions = nullptr;
        a.actions_out = slice(actions_out_dev, h->action_bytes(), k);
        a.obs = (float *)slice(obs_dev, h->O * sizeof(float), k);
        a.reward = slice(reward_dev, h->reward_bytes(), k);
        a.terminated = (uint8_t *)slice(terminated_dev, 1, k);
        a.truncated = (uint8_t *)slice(truncated_dev, 1, k);
        a.final_obs = (float *)slice(final_obs_dev, h->O * sizeof(float), k);
        a.ep_return_out = (float *)slice(a.ep_return_out, sizeof(float), k);
        a.ep_length_out = (int32_t *)slice(a.ep_length_out, sizeof(int32_t), k);
        return launch_step(h->cfg.env_id, h->param_mode(), a, h->stream, &h->last_launch);
    };
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(h->stream, &capturing);   // inside a caller's capture: plain launches (they are being recorded already)
    if (mode == MXV_ROLLOUT_EAGER || capturing != hipStreamCaptureStatusNone) {
        for (int k = 0; k < K; ++k)
            MXV_HIP(h, launch_k(k, h->dev_clock ? h->t_dev : nullptr, h->dev_clock ? (uint64_t)k : h->t + (uint64_t)k));
    } else {
        // The captured launches read the base step index from device memory (t_dev) and add their
        // own offset k, so one instantiated graph replays for any t.  Seeds, bounds and params are
        // baked into the captured kernel arguments: the cache is dropped whenever they change.
        mxv_handle::GraphKey key{K, per_step, actions_out_dev, obs_dev, reward_dev, terminated_dev, truncated_dev,
                                 final_obs_dev};
        auto it = h->graphs.find(key);
        if (it == h->graphs.end()) {
            hipGraph_t graph = nullptr;
            MXV_HIP(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
            hipError_t le = hipSuccess;
            for (int k = 0; k < K && le == hipSuccess; ++k) le = launch_k(k, h->t_dev, (uint64_t)k);
            hipError_t ce = hipStreamEndCapture(h->stream, &graph);
            if (le != hipSuccess || ce != hipSuccess) {
                if (graph) (void)hipGraphDestroy(graph);
                return fail(h, MXV_ERR_HIP, "graph capture: %s", hipGetErrorString(le != hipSuccess ? le : ce));
            }
            hipGraphExec_t exec = nullptr;
            const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            MXV_HIP(h, ie);
            it = h->graphs.emplace(key, exec).first;
        }
        if (!h->dev_clock) MXV_HIP(h, launch_set_word(h->t_dev, h->t, h->stream));
        MXV_HIP(h, hipGraphLaunch(it->second, h->stream));
    }
    h->state_injected = false;
    if (int rc = clock_add(h, K)) return rc;
    if (h->snap_obs) return copy_final_snapshot(h, K, per_step, obs_dev, reward_dev, terminated_dev, truncated_dev);
    return MXV_OK;
}

int mxv_set_final_snapshot(mxv_handle *h, float *obs_dev, void *reward_dev, uint8_t *terminated_dev, uint8_t *truncated_dev) {
    MXV_CHECK_HANDLE(h);
    if (!obs_dev && (reward_dev || terminated_dev || truncated_dev))
        return fail(h, MXV_ERR_INVALID_ARG, "mxv_set_final_snapshot: the observation buffer is required (NULL everything to detach)");
    h->snap_obs = obs_dev;
    h->snap_reward = reward_dev;
    h->snap_term = terminated_dev;
    h->snap_trunc = truncated_dev;
    return MXV_OK;
}

int mxv_rollout_tape(mxv_handle *h, int32_t K, int32_t per_step, const void *actions_tape_dev, float *obs_dev,
                     void *reward_dev, uint8_t *terminated_dev, uint8_t *truncated_dev, float *final_obs_dev) {
    MXV_CHECK_HANDLE(h);
    if (!actions_tape_dev) return fail(h, MXV_ERR_INVALID_ARG, "actions tape pointer is NULL");
    if (int rc = rollout_checks(h, K, obs_dev)) return rc;
    return fused_launch(h, K, per_step, actions_tape_dev, nullptr, obs_dev, reward_dev, terminated_dev, truncated_dev,
                        final_obs_dev);
}

int mxv_sample_actions(mxv_handle *h, void *actions_out_dev) {
    MXV_CHECK_HANDLE(h);
    if (!actions_out_dev) return fail(h, MXV_ERR_INVALID_ARG, "actions_out pointer is NULL");
    if (int rc = use_device(h)) return rc;
    SampleArgs a{};
    a.actions_out = actions_out_dev;
    a.t_dev = nullptr;
    a.n = h->cfg.num_envs;
    a.env0 = (uint64_t)h->cfg.env_offset;
    a.action_seed = h->action_seed;
    a.t = h->t;
    if (h->dev_clock) {
        a.t_dev = h->t_dev;
        a.t = 0;
    }
    a.flags = h->cfg.flags;
    a.params_pe = h->params_pe;
    a.P = h->P;
    if (int rc = check_aligned(h, actions_out_dev, h->NA > 0 && !(h->cfg.flags & MXV_FLAG_ACTION_I32) ? 8 : 4, "actions_out")) return rc;
    MXV_HIP(h, launch_sample(h->cfg.env_id, h->param_mode(), a, h->stream));
    return MXV_OK;
}

int mxv_reset_host(mxv_handle *h, const uint8_t *mask_host, const double *bounds2_host, float *obs_host) {
    MXV_CHECK_HANDLE(h);
    if (int rc = use_device(h)) return rc;
    if (int rc = ensure_staging(h)) return rc;
    const size_t n = (size_t)h->cfg.num_envs;
    if (h->hostmap) {
        if (mask_host) std::memcpy(h->st_mask, mask_host, n);
        if (int rc = do_reset(h, mask_host ? h->st_mask : nullptr, bounds2_host, obs_host ? h->st_obs : nullptr)) return rc;
        MXV_HIP(h, hipStreamSynchronize(h->stream));
        if (obs_host) std::memcpy(obs_host, h->st_obs, n * h->O * sizeof(float));
        return MXV_OK;
    }
    if (mask_host) MXV_HIP(h, hipMemcpyAsync(h->st_mask, mask_host, n, hipMemcpyHostToDevice, h->stream));
    if (int rc = do_reset(h, mask_host ? h->st_mask : nullptr, bounds2_host, obs_host ? h->st_obs : nullptr)) return rc;
    if (obs_host)
        MXV_HIP(h, hipMemcpyAsync(obs_host, h->st_obs, n * h->O * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    MXV_HIP(h, hipStreamSynchronize(h->stream));
    return MXV_OK;
}

int mxv_step_host(mxv_handle *h, const void *actions_host, float *obs_host, void *reward_host, uint8_t *terminated_host,
                  uint8_t *truncated_host, float *final_obs_host) {
    MXV_CHECK_HANDLE(h);
    if (!actions_host || !obs_host) return fail(h, MXV_ERR_INVALID_ARG, "actions/obs pointer is NULL");
    if (int rc = use_device(h)) return rc;
    if (int rc = ensure_staging(h)) return rc;
    const size_t n = (size_t)h->cfg.num_envs;
    if (int rc = upload_actions(h, actions_host)) return rc;
    float *keep_r = h->ep_return_out;
    int32_t *keep_l = h->ep_length_out;
    if (h->ep_acc) {  // host callers read the statistics of this step with mxv_episode_stats_host()
        h->ep_return_out = h->st_ep_r;
        h->ep_length_out = h->st_ep_l;
    }
    struct Restore {
        mxv_handle *h; float *r; int32_t *l;
        ~Restore() { h->ep_return_out = r; h->ep_length_out = l; }
    } restore{h, keep_r, keep_l};
    {
        ErrInBlock guard(h);
        if (int rc = do_step(h, h->st_actions, nullptr, h->st_obs, reward_host ? h->st_reward : nullptr,
                             terminated_host ? h->st_term : nullptr, truncated_host ? h->st_trunc : nullptr,
                             (final_obs_host || h->fin_packed) ? h->st_final : nullptr))
            return rc;
    }
    if (h->hostmap) {  // outputs and the error word are already in host memory once the stream drains
        MXV_HIP(h, hipStreamSynchronize(h->stream));
        std::memcpy(obs_host, h->st_obs, n * h->O * sizeof(float));
        if (reward_host) std::memcpy(reward_host, h->st_reward, n * h->reward_bytes());
        if (terminated_host) std::memcpy(terminated_host, h->st_term, n);
        if (truncated_host) std::memcpy(truncated_host, h->st_trunc, n);
        if (final_obs_host) std::memcpy(final_obs_host, h->st_final, n * h->O * sizeof(float));
        const int rc = take_block_error(h);
        if (rc != MXV_OK) (void)clock_add(h, -1);  // the reference raises before stepping anything further
        return rc;
    }
    // the packed final rows go first: their DMAs target pinned memory and are truly asynchronous, the copies into the caller's
    // (pageable) arrays below are not
    const bool packed = (h->fin_packed || final_obs_host) && terminated_host && truncated_host;
    size_t first = 0;
    if (packed)
        if (int rc = queue_final_rows(h, &first)) return rc;
    MXV_HIP(h, hipMemcpyAsync(obs_host, h->st_obs, n * h->O * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (reward_host)
        MXV_HIP(h, hipMemcpyAsync(reward_host, h->st_reward, n * h->reward_bytes(), hipMemcpyDeviceToHost, h->stream));
    if (terminated_host)
        MXV_HIP(h, hipMemcpyAsync(terminated_host, h->st_term, n, hipMemcpyDeviceToHost, h->stream));
    if (truncated_host)
        MXV_HIP(h, hipMemcpyAsync(truncated_host, h->st_trunc, n, hipMemcpyDeviceToHost, h->stream));
    if (!packed && final_obs_host)
        MXV_HIP(h, hipMemcpyAsync(final_obs_host, h->st_final, n * h->O * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    int rc = check_latched(h);  // synchronises
    if (rc == MXV_ERR_INVALID_ACTION) (void)clock_add(h, -1);  // the reference raises before stepping anything further
    if (rc == MXV_OK && packed)                   // packed rows of the finished envs: left packed (mxv_final_packed_view) or scattered
        rc = finish_final_rows(h, first, h->fin_packed ? nullptr : final_obs_host);
    return rc;
}

int mxv_staging_view(mxv_handle *h, const float **obs, const void **reward, const uint8_t **terminated, const uint8_t **truncated) {
    MXV_CHECK_HANDLE(h);
    if (!h->st_obs) return fail(h, MXV_ERR_INVALID_ARG, "mxv_staging_view: no host step or reset has run on this handle yet");
    if (obs) *obs = h->st_obs;
    if (reward) *reward = h->st_reward;
    if (terminated) *terminated = h->st_term;
    if (truncated) *truncated = h->st_trunc;
    return MXV_OK;
}

int mxv_host_alloc(size_t bytes, void **ptr) {
    if (!ptr || bytes == 0) return fail(nullptr, MXV_ERR_INVALID_ARG, "mxv_host_alloc: NULL pointer or zero size");
    MXV_HIP(nullptr, hipHostMalloc(ptr, bytes, hipHostMallocPortable));
    return MXV_OK;
}

int mxv_host_free(void *ptr) {
    if (ptr) MXV_HIP(nullptr, hipHostFree(ptr));
    return MXV_OK;
}

int mxv_host_block_layout(mxv_handle *h, size_t *bytes, size_t *final_obs_off, size_t *obs_off, size_t *reward_off,
                          size_t *terminated_off, size_t *truncated_off) {
    MXV_CHECK_HANDLE(h);
    if (int rc = use_device(h)) return rc;
    if (int rc = ensure_staging(h)) return rc;
    const char *base = (const char *)h->st_final;
    if (bytes) *bytes = (size_t)((const char *)h->st_mask - base);
    if (final_obs_off) *final_obs_off = 0;
    if (obs_off) *obs_off = (size_t)((const char *)h->st_obs - base);
    if (reward_off) *reward_off = (size_t)((const char *)h->st_reward - base);
    if (terminated_off) *terminated_off = (size_t)((const char *)h->st_term - base);
    if (truncated_off) *truncated_off = (size_t)((const char *)h->st_trunc - base);
    return MXV_OK;
}

int mxv_step_host_block(mxv_handle *h, const void *actions_host, void *block_host, int32_t want_final) {
    MXV_CHECK_HANDLE(h);
    if (!actions_host || !block_host) return fail(h, MXV_ERR_INVALID_ARG, "actions/block pointer is NULL");
    if (int rc = use_device(h)) return rc;
    if (int rc = ensure_staging(h)) return rc;
    if (int rc = upload_actions(h, actions_host)) return rc;
    float *keep_r = h->ep_return_out;
    int32_t *keep_l = h->ep_length_out;
    if (h->ep_acc) {
        h->ep_return_out = h->st_ep_r;
        h->ep_length_out = h->st_ep_l;
    }
    struct Restore {
        mxv_handle *h; float *r; int32_t *l;
        ~Restore() { h->ep_return_out = r; h->ep_length_out = l; }
    } restore{h, keep_r, keep_l};
    {
        ErrInBlock guard(h);
        if (int rc = do_step(h, h->st_actions, nullptr, h->st_obs, h->st_reward, h->st_term, h->st_trunc, want_final ? h->st_final : nullptr))
            return rc;
    }
    const bool packed = want_final && h->fin_packed && !h->hostmap;
    const bool dense_final = want_final && !packed;
    const char *src = (const char *)(dense_final ? (void *)h->st_final : (void *)h->st_obs);
    const size_t off = (size_t)(src - (const char *)h->st_final), len = (size_t)((const char *)h->st_mask - src);
    size_t first = 0;
    int rc;
    if (h->hostmap) {
        MXV_HIP(h, hipStreamSynchronize(h->stream));
        std::memcpy((char *)block_host + off, src, len);
        rc = take_block_error(h);
    } else {
        if (packed)
            if (int rc2 = queue_final_rows(h, &first)) return rc2;
        MXV_HIP(h, hipMemcpyAsync((char *)block_host + off, src, len, hipMemcpyDeviceToHost, h->stream));   // ONE DMA: obs .. truncated
        rc = check_latched(h);  // synchronises
        if (rc == MXV_OK && packed) rc = finish_final_rows(h, first, nullptr);
    }
    if (rc == MXV_ERR_INVALID_ACTION) (void)clock_add(h, -1);  // the reference raises before stepping anything further
    return rc;
}

int mxv_host_io(mxv_handle *h, void **actions, float **obs, void **reward, uint8_t **terminated, uint8_t **truncated,
                float **final_obs) {
    MXV_CHECK_HANDLE(h);
    if (int rc = use_device(h)) return rc;
    if (int rc = ensure_staging(h, /*want_pinned=*/true)) return rc;
    const ptrdiff_t shift = (char *)h->hm_block - (char *)(h->hostmap ? h->hm_block : h->dv_block);  // same layout
    auto host = [&](void *kernel_side) { return (void *)((char *)kernel_side + shift); };
    if (actions) *actions = host(h->st_actions);
    if (obs) *obs = (float *)host(h->st_obs);
    if (reward) *reward = host(h->st_reward);
    if (terminated) *terminated = (uint8_t *)host(h->st_term);
    if (truncated) *truncated = (uint8_t *)host(h->st_trunc);
    if (final_obs) *final_obs = (float *)host(h->st_final);
    return MXV_OK;
}

}  // extern "C" (re-opened below)

namespace {

int mapped_finish(mxv_handle *h, bool stepped) {
    if (!h->hostmap) {  // one DMA copy brings obs | reward | terminated | truncated into the pinned mirror ...
        MXV_HIP(h, hipMemcpyAsync((char *)h->hm_block + h->io_out_off, (char *)h->dv_block + h->io_out_off, h->io_out_bytes,
                                  hipMemcpyDeviceToHost, h->stream));
        if (stepped) {  // ... and the final_obs rows of the finished envs follow packed
            float *fin_pinned = (float *)((char *)h->hm_block + ((char *)h->st_final - (char *)h->dv_block));
            if (int rc = fetch_final_rows(h, h->fin_packed ? nullptr : fin_pinned)) return rc;
        }
    }
    if (!h->hostmap) {  // (a small env's kernel raised the word in the pinned block itself: see ErrInBlock)
        MXV_HIP(h, hipMemcpyAsync(h->hm_err, h->err, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        MXV_HIP(h, hipStreamSynchronize(h->stream));
        if (*h->hm_err != 0) MXV_HIP(h, hipMemsetAsync(h->err, 0, sizeof(int32_t), h->stream));
    } else {
        MXV_HIP(h, hipStreamSynchronize(h->stream));
    }
    const int rc = take_block_error(h);
    if (rc != MXV_OK && stepped) (void)clock_add(h, -1);  // the reference raises before stepping anything further
    return rc;
}

}  // namespace

extern "C" {

int mxv_step_mapped(mxv_handle *h) {
    MXV_CHECK_HANDLE(h);
    if (!h->hm_block) return fail(h, MXV_ERR_INVALID_ARG, "call mxv_host_io() first: the mapped I/O block does not exist yet");
    if (int rc = use_device(h)) return rc;
    if (!h->hostmap)
        MXV_HIP(h, hipMemcpyAsync(h->dv_block, h->hm_block, h->io_act_bytes, hipMemcpyHostToDevice, h->stream));
    float *keep_r = h->ep_return_out;
    int32_t *keep_l = h->ep_length_out;
    if (h->ep_acc) {
        h->ep_return_out = h->st_ep_r;
        h->ep_length_out = h->st_ep_l;
    }
    struct Restore {
        mxv_handle *h; float *r; int32_t *l;
        ~Restore() { h->ep_return_out = r; h->ep_length_out = l; }
    } restore{h, keep_r, keep_l};
    {
        ErrInBlock guard(h);
        if (int rc = do_step(h, h->st_actions, nullptr, h->st_obs, h->st_reward, h->st_term, h->st_trunc, h->st_final)) return rc;
    }
    return mapped_finish(h, true);
}

int mxv_reset_mapped(mxv_handle *h, const double *bounds2_host) {
    MXV_CHECK_HANDLE(h);
    if (!h->hm_block) return fail(h, MXV_ERR_INVALID_ARG, "call mxv_host_io() first: the mapped I/O block does not exist yet");
    if (int rc = use_device(h)) return rc;
    if (int rc = do_reset(h, nullptr, bounds2_host, h->st_obs)) return rc;
    return mapped_finish(h, false);
}

/* see include/mxv.h */
int mxv_adopt_obs(mxv_handle *h, float *obs_dev) {
    MXV_CHECK_HANDLE(h);
    if (int rc = use_device(h)) return rc;
    if (int rc = ensure_f64(h)) return rc;            // whatever was adopted before hands the state back first
    if (!obs_dev) {
        h->adopted_obs = nullptr;
        return MXV_OK;
    }
    if (!hilo_supported(h->cfg.env_id))
        return fail(h, MXV_ERR_UNSUPPORTED, "the observation carries the state only where it is float32(state) component by component: "
                                            "CartPole, MountainCar, MountainCarContinuous");
    if (int rc = check_aligned(h, obs_dev, h->O == 4 ? 16 : 8, "obs")) return rc;
    if (!h->lo) MXV_HIP(h, hipMalloc((void **)&h->lo, (size_t)h->cfg.num_envs * h->S * sizeof(int32_t)));
    h->adopted_obs = obs_dev;
    return MXV_OK;
}

int mxv_get_state(mxv_handle *h, double *state_soa_host, int32_t *elapsed_host) {
    MXV_CHECK_HANDLE(h);
    if (int rc = use_device(h)) return rc;
    if (int rc = ensure_f64(h)) return rc;
    const size_t n = (size_t)h->cfg.num_envs;
    if (state_soa_host)
        MXV_HIP(h, hipMemcpyAsync(state_soa_host, h->state, n * h->S * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    std::vector<uint16_t> narrow;
    if (elapsed_host) {
        if (h->elapsed16) {
            narrow.resize(n);
            MXV_HIP(h, hipMemcpyAsync(narrow.data(), h->elapsed, n * sizeof(uint16_t), hipMemcpyDeviceToHost, h->stream));
        } else {
            MXV_HIP(h, hipMemcpyAsync(elapsed_host, h->elapsed, n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        }
    }
    MXV_HIP(h, hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < narrow.size(); ++i) elapsed_host[i] = narrow[i];
    return MXV_OK;
}

int mxv_set_state(mxv_handle *h, const double *state_soa_host, const int32_t *elapsed_host) {
    MXV_CHECK_HANDLE(h);
    if (int rc = use_device(h)) return rc;
    if (int rc = ensure_f64(h)) return rc;
    const size_t n = (size_t)h->cfg.num_envs;
    if (state_soa_host)
        MXV_HIP(h, hipMemcpyAsync(h->state, state_soa_host, n * h->S * sizeof(double), hipMemcpyHostToDevice, h->stream));
    std::vector<uint16_t> narrow;
    if (elapsed_host) {
        if (h->elapsed16) {
            narrow.resize(n);
            for (size_t i = 0; i < n; ++i) {
                if (elapsed_host[i] < 0) return fail(h, MXV_ERR_INVALID_ARG, "elapsed[%zu] = %d is negative", i, elapsed_host[i]);
                narrow[i] = (uint16_t)(elapsed_host[i] > 65535 ? 65535 : elapsed_host[i]);
            }
            MXV_HIP(h, hipMemcpyAsync(h->elapsed, narrow.data(), n * sizeof(uint16_t), hipMemcpyHostToDevice, h->stream));
        } else {
            MXV_HIP(h, hipMemcpyAsync(h->elapsed, elapsed_host, n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        }
    }
    if (state_soa_host && h->beyond) MXV_HIP(h, hipMemsetAsync(h->beyond, 0, n, h->stream));  // a fresh state: steps_beyond_terminated = None
    MXV_HIP(h, hipStreamSynchronize(h->stream));
    h->was_reset = true;  // an injected state stands in for reset() (parity harness, checkpoint restore)
    if (state_soa_host) {
        // The fused rollout's unguarded trigonometry (SAFE = false) rests on invariants the dynamics maintain: CartPole |theta| <= pi/4,
        // every other trig argument below 2^19.  A state that obeys them — a restored checkpoint — keeps the fast kernels; one that
        // does not takes the guarded instantiation for the next launch, after which termination + autoreset (CartPole), the wrap
        // loops (Acrobot) and the position clamps (MountainCar*) have restored them.  Only Pendulum keeps an out-of-range angle
        // (it is never wrapped, pendulum.py:133) until its episode is truncated: guarded launches until the next full reset.
        // Non-finite values count as outside: NaN goes through both code paths alike, an infinite CartPole velocity does not (above).
        bool in_range = true;
        const int S = h->S;
        for (int k = 0; k < S && in_range; ++k) {
            const double *row = state_soa_host + (size_t)k * n;
            // (CartPole's other components must keep every intermediate finite — the unguarded path's quotients turn an infinite dividend
            // into a NaN where IEEE division and the reference give +-Inf — and polemass_length * theta_dot^2 * sintheta overflows from
            // |theta_dot| ~ 1.3e154 on: beyond 1e150 a state takes the guarded launch.  ADVICE r4.)
            const double lim = (h->cfg.env_id == MXV_CARTPOLE) ? (k == 2 ? 0.78539816339744830962 : 1e150) : 65536.0;
            for (size_t i = 0; i < n; ++i)
                if (!(std::fabs(row[i]) <= lim)) {
                    in_range = false;
                    break;
                }
        }
        h->state_injected = !in_range;
        h->state_out_of_range = !in_range && h->cfg.env_id == MXV_PENDULUM;
    }
    return MXV_OK;
}

int mxv_last_launch(mxv_handle *h, mxv_launch_info *out) {
    MXV_CHECK_HANDLE(h);
    if (!out) return fail(h, MXV_ERR_INVALID_ARG, "mxv_last_launch: out is NULL");
    static_assert(sizeof(mxv_launch_info) == sizeof(LaunchInfo), "mxv_launch_info mirrors LaunchInfo member for member");
    std::memcpy(out, &h->last_launch, sizeof(*out));
    return MXV_OK;
}

int mxv_get_counters(mxv_handle *h, uint64_t *t, uint32_t *r) {
    MXV_CHECK_HANDLE(h);
    if (h->dev_clock) {  // graphs the caller replays advance the device word only: read it (synchronises the handle's stream)
        if (int rc = use_device(h)) return rc;
        MXV_HIP(h, hipMemcpyAsync(&h->t, h->t_dev, sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
        MXV_HIP(h, hipStreamSynchronize(h->stream));
    }
    if (t) *t = h->t;
    if (r) *r = h->r;
    return MXV_OK;
}

int mxv_set_counters(mxv_handle *h, uint64_t t, uint32_t r) {
    MXV_CHECK_HANDLE(h);
    h->t = t;
    h->r = r;
    if (h->dev_clock) {
        if (int rc = use_device(h)) return rc;
        MXV_HIP(h, launch_set_word(h->t_dev, h->t, h->stream));
    }
    return MXV_OK;
}

int mxv_set_obs_partials(mxv_handle *h, double *partials_dev) {
    MXV_CHECK_HANDLE(h);
    h->obs_part = partials_dev;
    return MXV_OK;
}

int mxv_set_return_partials(mxv_handle *h, double *returns_state_dev, double gamma, double *partials_dev) {
    MXV_CHECK_HANDLE(h);
    if (partials_dev && !returns_state_dev) return fail(h, MXV_ERR_INVALID_ARG, "mxv_set_return_partials: the running returns [N] are required");
    h->ret_part = partials_dev;
    h->ret_state = partials_dev ? returns_state_dev : nullptr;
    h->ret_gamma = gamma;
    return MXV_OK;
}

int mxv_obs_partials_layout(mxv_handle *h, int64_t *leaves, int64_t *envs_per_leaf, int32_t *values) {
    MXV_CHECK_HANDLE(h);
    const int64_t per = stats_leaf_envs(h->cfg.env_id);
    if (leaves) *leaves = (h->cfg.num_envs + per - 1) / per;
    if (envs_per_leaf) *envs_per_leaf = per;
    if (values) *values = 2 * h->O;
    return MXV_OK;
}

int mxv_set_device_clock(mxv_handle *h, int32_t on) {
    MXV_CHECK_HANDLE(h);
    if (int rc = use_device(h)) return rc;
    if (on && !h->dev_clock) {
        MXV_HIP(h, launch_set_word(h->t_dev, h->t, h->stream));
        h->dev_clock = true;
    } else if (!on && h->dev_clock) {
        MXV_HIP(h, hipMemcpyAsync(&h->t, h->t_dev, sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
        MXV_HIP(h, hipStreamSynchronize(h->stream));
        h->dev_clock = false;
    }
    return MXV_OK;
}

int mxv_write_probe(int32_t device, int64_t num_envs, int32_t K, int32_t launches, float *obs_dev, double *reward_dev, int64_t *actions_dev,
                    uint8_t *terminated_dev, uint8_t *truncated_dev, double *us_per_step) {
    if (!obs_dev || !reward_dev || !actions_dev || !terminated_dev || !truncated_dev || !us_per_step || K < 1 || launches < 1 ||
        num_envs < 1024 || num_envs % 1024 != 0)
        return fail(nullptr, MXV_ERR_INVALID_ARG, "mxv_write_probe: [K][num_envs] buffers of all five outputs, num_envs a multiple of 1024");
    MXV_HIP(nullptr, hipSetDevice(device));
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t err = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (err == hipSuccess) err = hipEventCreate(&e0);
    if (err == hipSuccess) err = hipEventCreate(&e1);
    for (int i = 0; i < 2 && err == hipSuccess; ++i)
        err = launch_write_probe(obs_dev, reward_dev, actions_dev, terminated_dev, truncated_dev, num_envs, K, s);
    if (err == hipSuccess) err = hipEventRecord(e0, s);
    for (int i = 0; i < launches && err == hipSuccess; ++i)
        err = launch_write_probe(obs_dev, reward_dev, actions_dev, terminated_dev, truncated_dev, num_envs, K, s);
    if (err == hipSuccess) err = hipEventRecord(e1, s);
    if (err == hipSuccess) err = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (s) (void)hipStreamDestroy(s);
    if (err != hipSuccess) return fail(nullptr, MXV_ERR_HIP, "mxv_write_probe: %s", hipGetErrorString(err));
    *us_per_step = (double)ms * 1e3 / ((double)launches * K);
    return MXV_OK;
}

int mxv_write_probe_env(int32_t device, int32_t env_id, int32_t flags, int64_t num_envs, int32_t K, int32_t launches, float *obs_dev,
                        void *reward_dev, void *actions_dev, uint8_t *terminated_dev, uint8_t *truncated_dev, double *us_per_step) {
    if (!obs_dev || !reward_dev || !actions_dev || !terminated_dev || !truncated_dev || !us_per_step || K < 1 || launches < 1 || num_envs < 1 ||
        env_id < 0 || env_id >= MXV_NUM_ENV_KINDS)
        return fail(nullptr, MXV_ERR_INVALID_ARG, "mxv_write_probe_env: [K][num_envs] buffers of all five outputs and a valid env kind");
    MXV_HIP(nullptr, hipSetDevice(device));
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t err = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (err == hipSuccess) err = hipEventCreate(&e0);
    if (err == hipSuccess) err = hipEventCreate(&e1);
    for (int i = 0; i < 2 && err == hipSuccess; ++i)
        err = launch_write_probe_env(env_id, flags, obs_dev, reward_dev, actions_dev, terminated_dev, truncated_dev, num_envs, K, s);
    if (err == hipSuccess) err = hipEventRecord(e0, s);
    for (int i = 0; i < launches && err == hipSuccess; ++i)
        err = launch_write_probe_env(env_id, flags, obs_dev, reward_dev, actions_dev, terminated_dev, truncated_dev, num_envs, K, s);
    if (err == hipSuccess) err = hipEventRecord(e1, s);
    if (err == hipSuccess) err = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (s) (void)hipStreamDestroy(s);
    if (err != hipSuccess) return fail(nullptr, MXV_ERR_HIP, "mxv_write_probe_env: %s", hipGetErrorString(err));
    *us_per_step = (double)ms * 1e3 / ((double)launches * K);
    return MXV_OK;
}

int mxv_final_packed(mxv_handle *h, int32_t enable, int32_t *supported) {
    MXV_CHECK_HANDLE(h);
    if (int rc = use_device(h)) return rc;
    if (int rc = ensure_staging(h)) return rc;
    const bool ok = !h->hostmap;  // small envs: the kernel writes final_obs straight into the pinned block, nothing to pack
    h->fin_packed = ok && enable != 0;
    if (supported) *supported = ok ? 1 : 0;
    return MXV_OK;
}

int mxv_final_packed_view(mxv_handle *h, const int32_t **count, const int32_t **idx, const float **rows) {
    MXV_CHECK_HANDLE(h);
    if (!h->fin_host) return fail(h, MXV_ERR_INVALID_ARG, "mxv_final_packed_view: packed final rows exist for large envs only (see mxv_final_packed)");
    const FinLayout L((size_t)h->cfg.num_envs, (size_t)h->O);
    if (count) *count = (const int32_t *)h->fin_host;
    if (idx) *idx = (const int32_t *)(h->fin_host + L.idx);
    if (rows) *rows = (const float *)(h->fin_host + L.rows);
    return MXV_OK;
}

int mxv_final_packed_stats_view(mxv_handle *h, const float **ep_return, const int32_t **ep_length) {
    MXV_CHECK_HANDLE(h);
    if (!h->fin_host) return fail(h, MXV_ERR_INVALID_ARG, "mxv_final_packed_stats_view: packed records exist for large envs only (see mxv_final_packed)");
    const FinLayout L((size_t)h->cfg.num_envs, (size_t)h->O);
    if (ep_return) *ep_return = (const float *)(h->fin_host + L.ep_r);
    if (ep_length) *ep_length = (const int32_t *)(h->fin_host + L.ep_l);
    return MXV_OK;
}

int mxv_get_episodes(mxv_handle *h, uint32_t *episodes_host) {
    MXV_CHECK_HANDLE(h);
    if (!episodes_host) return fail(h, MXV_ERR_INVALID_ARG, "episodes pointer is NULL");
    if (int rc = use_device(h)) return rc;
    MXV_HIP(h, hipMemcpyAsync(episodes_host, h->episodes, (size_t)h->cfg.num_envs * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    MXV_HIP(h, hipStreamSynchronize(h->stream));
    return MXV_OK;
}

int mxv_set_episodes(mxv_handle *h, const uint32_t *episodes_host) {
    MXV_CHECK_HANDLE(h);
    if (!episodes_host) return fail(h, MXV_ERR_INVALID_ARG, "episodes pointer is NULL");
    if (int rc = use_device(h)) return rc;
    MXV_HIP(h, hipMemcpyAsync(h->episodes, episodes_host, (size_t)h->cfg.num_envs * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    MXV_HIP(h, hipStreamSynchronize(h->stream));
    return MXV_OK;
}

// steps_beyond_terminated marks (CartPole without autoreset): part of a checkpoint — a restored env that had terminated before must
// keep paying 0.0, not 1.0 once more.  Handles that carry no marks (any other configuration): get fills zeros, set accepts
// all-zero marks only.
int mxv_get_beyond(mxv_handle *h, uint8_t *beyond_host) {
    MXV_CHECK_HANDLE(h);
    if (!beyond_host) return fail(h, MXV_ERR_INVALID_ARG, "beyond pointer is NULL");
    const size_t n = (size_t)h->cfg.num_envs;
    if (!h->beyond) {
        std::memset(beyond_host, 0, n);
        return MXV_OK;
    }
    if (int rc = use_device(h)) return rc;
    MXV_HIP(h, hipMemcpyAsync(beyond_host, h->beyond, n, hipMemcpyDeviceToHost, h->stream));
    MXV_HIP(h, hipStreamSynchronize(h->stream));
    return MXV_OK;
}

int mxv_set_beyond(mxv_handle *h, const uint8_t *beyond_host) {
    MXV_CHECK_HANDLE(h);
    if (!beyond_host) return fail(h, MXV_ERR_INVALID_ARG, "beyond pointer is NULL");
    const size_t n = (size_t)h->cfg.num_envs;
    if (!h->beyond) {
        for (size_t i = 0; i < n; ++i)
            if (beyond_host[i]) return fail(h, MXV_ERR_UNSUPPORTED, "this handle keeps no steps_beyond_terminated marks (CartPole with MXV_FLAG_NO_AUTORESET does)");
        return MXV_OK;
    }
    if (int rc = use_device(h)) return rc;
    MXV_HIP(h, hipMemcpyAsync(h->beyond, beyond_host, n, hipMemcpyHostToDevice, h->stream));
    MXV_HIP(h, hipStreamSynchronize(h->stream));
    return MXV_OK;
}

int mxv_get_params(mxv_handle *h, double *params_host) {
    MXV_CHECK_HANDLE(h);
    if (!params_host) return fail(h, MXV_ERR_INVALID_ARG, "params pointer is NULL");
    std::memcpy(params_host, h->P.p, sizeof h->P.p);
    return MXV_OK;
}

int mxv_set_params(mxv_handle *h, const double *params_host) {
    MXV_CHECK_HANDLE(h);
    if (!params_host) return fail(h, MXV_ERR_INVALID_ARG, "params pointer is NULL");
    if (h->cfg.env_id == MXV_ACROBOT && params_host[10] < 0.0)
        return fail(h, MXV_ERR_INVALID_ARG, "Acrobot torque_noise_max must be >= 0");
    std::memcpy(h->P.p, params_host, sizeof h->P.p);
    h->step_noise = h->cfg.env_id == MXV_ACROBOT && params_host[10] > 0.0;
    double d[MXV_MAX_PARAMS];
    default_params(h->cfg.env_id, d);
    h->default_params = std::memcmp(d, h->P.p, sizeof d) == 0;
    if (int rc = use_device(h)) return rc;
    MXV_HIP(h, hipStreamSynchronize(h->stream));
    free_graphs(h);  // captured kernel arguments hold the old parameters
    if (h->params_pe) {  // one value for all sub-envs again
        MXV_HIP(h, hipFree(h->params_pe));
        h->params_pe = nullptr;
    }
    return MXV_OK;
}

int mxv_set_params_per_env(mxv_handle *h, const double *params_host) {
    MXV_CHECK_HANDLE(h);
    if (!params_host) return fail(h, MXV_ERR_INVALID_ARG, "params pointer is NULL");
    const size_t n = (size_t)h->cfg.num_envs;
    bool noise = false;
    if (h->cfg.env_id == MXV_ACROBOT)
        for (size_t i = 0; i < n; ++i) {
            if (params_host[10 * n + i] < 0.0) return fail(h, MXV_ERR_INVALID_ARG, "Acrobot torque_noise_max must be >= 0");
            noise = noise || params_host[10 * n + i] > 0.0;
        }
    h->step_noise = noise;
    if (int rc = use_device(h)) return rc;
    MXV_HIP(h, hipStreamSynchronize(h->stream));
    free_graphs(h);
    const size_t bytes = n * MXV_MAX_PARAMS * sizeof(double);
    if (!h->params_pe) MXV_HIP(h, hipMalloc((void **)&h->params_pe, bytes));
    MXV_HIP(h, hipMemcpyAsync(h->params_pe, params_host, bytes, hipMemcpyHostToDevice, h->stream));
    MXV_HIP(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < MXV_MAX_PARAMS; ++k) h->P.p[k] = params_host[(size_t)k * n];  // env 0's values for mxv_get_params
    return MXV_OK;
}

int mxv_get_params_per_env(mxv_handle *h, double *params_host) {
    MXV_CHECK_HANDLE(h);
    if (!params_host) return fail(h, MXV_ERR_INVALID_ARG, "params pointer is NULL");
    const size_t n = (size_t)h->cfg.num_envs;
    if (!h->params_pe) {
        for (int k = 0; k < MXV_MAX_PARAMS; ++k)
            for (size_t i = 0; i < n; ++i) params_host[(size_t)k * n + i] = h->P.p[k];
        return MXV_OK;
    }
    if (int rc = use_device(h)) return rc;
    MXV_HIP(h, hipMemcpyAsync(params_host, h->params_pe, n * MXV_MAX_PARAMS * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    MXV_HIP(h, hipStreamSynchronize(h->stream));
    return MXV_OK;
}

int mxv_episode_stats(mxv_handle *h, int32_t enable) {
    MXV_CHECK_HANDLE(h);
    if (int rc = use_device(h)) return rc;
    MXV_HIP(h, hipStreamSynchronize(h->stream));
    free_graphs(h);
    const size_t n = (size_t)h->cfg.num_envs;
    if (enable && !h->ep_acc) {
        if (h->cfg.flags & MXV_FLAG_NO_AUTORESET)
            return fail(h, MXV_ERR_UNSUPPORTED, "episode statistics need autoreset (episode length is the TimeLimit counter)");
        MXV_HIP(h, hipMalloc((void **)&h->ep_acc, n * sizeof(float)));
        MXV_HIP(h, hipMalloc((void **)&h->st_ep_r, n * sizeof(float)));
        MXV_HIP(h, hipMalloc((void **)&h->st_ep_l, n * sizeof(int32_t)));
        MXV_HIP(h, hipMemsetAsync(h->ep_acc, 0, n * sizeof(float), h->stream));
        MXV_HIP(h, hipMemsetAsync(h->st_ep_r, 0, n * sizeof(float), h->stream));
        MXV_HIP(h, hipMemsetAsync(h->st_ep_l, 0, n * sizeof(int32_t), h->stream));
        MXV_HIP(h, hipStreamSynchronize(h->stream));
    } else if (!enable && h->ep_acc) {
        MXV_HIP(h, hipFree(h->ep_acc));
        MXV_HIP(h, hipFree(h->st_ep_r));
        MXV_HIP(h, hipFree(h->st_ep_l));
        h->ep_acc = nullptr;
        h->st_ep_r = nullptr;
        h->st_ep_l = nullptr;
    }
    return MXV_OK;
}

int mxv_set_episode_outputs(mxv_handle *h, float *ep_return_dev, int32_t *ep_length_dev) {
    MXV_CHECK_HANDLE(h);
    if (int rc = use_device(h)) return rc;
    MXV_HIP(h, hipStreamSynchronize(h->stream));
    free_graphs(h);  // captured kernel arguments hold the old pointers
    h->ep_return_out = ep_return_dev;
    h->ep_length_out = ep_length_dev;
    return MXV_OK;
}

int mxv_episode_stats_host(mxv_handle *h, float *ep_return_host, int32_t *ep_length_host, float *running_return_host) {
    MXV_CHECK_HANDLE(h);
    if (!h->ep_acc) return fail(h, MXV_ERR_INVALID_ARG, "episode statistics are not enabled (mxv_episode_stats)");
    if (int rc = use_device(h)) return rc;
    const size_t n = (size_t)h->cfg.num_envs;
    if (ep_return_host)
        MXV_HIP(h, hipMemcpyAsync(ep_return_host, h->st_ep_r, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (ep_length_host)
        MXV_HIP(h, hipMemcpyAsync(ep_length_host, h->st_ep_l, n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    if (running_return_host)
        MXV_HIP(h, hipMemcpyAsync(running_return_host, h->ep_acc, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    MXV_HIP(h, hipStreamSynchronize(h->stream));
    return MXV_OK;
}

int mxv_set_running_returns(mxv_handle *h, const float *running_return_host) {
    MXV_CHECK_HANDLE(h);
    if (!h->ep_acc) return fail(h, MXV_ERR_INVALID_ARG, "episode statistics are not enabled (mxv_episode_stats)");
    if (!running_return_host) return fail(h, MXV_ERR_INVALID_ARG, "running_return pointer is NULL");
    if (int rc = use_device(h)) return rc;
    MXV_HIP(h, hipMemcpyAsync(h->ep_acc, running_return_host, (size_t)h->cfg.num_envs * sizeof(float), hipMemcpyHostToDevice,
                              h->stream));
    MXV_HIP(h, hipStreamSynchronize(h->stream));
    return MXV_OK;
}

int mxv_sync(mxv_handle *h) {
    MXV_CHECK_HANDLE(h);
    if (int rc = use_device(h)) return rc;
    return check_latched(h);
}

int mxv_get_stream(mxv_handle *h, void **stream) {
    MXV_CHECK_HANDLE(h);
    if (!stream) return fail(h, MXV_ERR_INVALID_ARG, "stream pointer is NULL");
    *stream = (void *)h->stream;
    return MXV_OK;
}

int mxv_wait_stream(mxv_handle *h, void *other_stream) {
    MXV_CHECK_HANDLE(h);
    if ((hipStream_t)other_stream == h->stream) return MXV_OK;
    if (int rc = use_device(h)) return rc;
    if (!h->ev_wait) MXV_HIP(h, hipEventCreateWithFlags(&h->ev_wait, hipEventDisableTiming));
    MXV_HIP(h, hipEventRecord(h->ev_wait, (hipStream_t)other_stream));
    MXV_HIP(h, hipStreamWaitEvent(h->stream, h->ev_wait, 0));
    return MXV_OK;
}

int mxv_set_stream(mxv_handle *h, void *stream) {
    MXV_CHECK_HANDLE(h);
    if (int rc = use_device(h)) return rc;
    if (h->stream) MXV_HIP(h, hipStreamSynchronize(h->stream));
    free_graphs(h);
    if (h->own_stream && h->stream) MXV_HIP(h, hipStreamDestroy(h->stream));
    h->stream = (hipStream_t)stream;
    h->own_stream = false;
    return MXV_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// Heterogeneous dispatch: K sampled steps of several homogeneous segments in ONE launch (BASELINE.json configs[4]).
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int mxv_rollout_mixed(mxv_handle *const *handles, int32_t count, int32_t K, int32_t per_step, const mxv_step_outputs *outs) {
    if (!handles || !outs || count < 1 || count > MXV_MAX_MIXED)
        return fail(nullptr, MXV_ERR_INVALID_ARG, "mxv_rollout_mixed: 1..%d handles with their outputs", MXV_MAX_MIXED);
    mxv_handle *h0 = handles[0];
    MXV_CHECK_HANDLE(h0);
    if (K < 2) return fail(h0, MXV_ERR_UNSUPPORTED, "mxv_rollout_mixed fuses K >= 2 steps (use mxv_step_sampled per handle for single steps)");
    MixedArgs m{};
    m.count = count;
    uint32_t blocks = 0;
    for (int i = 0; i < count; ++i) {
        mxv_handle *h = handles[i];
        MXV_CHECK_HANDLE(h);
        if (h->cfg.device != h0->cfg.device) return fail(h0, MXV_ERR_INVALID_ARG, "segment %d lives on another device", i);
        for (int j = 0; j < i; ++j)
            if (handles[j] == h) return fail(h0, MXV_ERR_INVALID_ARG, "handle listed twice (segments %d and %d)", j, i);
        if (int rc = rollout_checks(h, K, outs[i].obs)) {
            if (h != h0) h0->error = h->error;
            return rc;
        }
        if (int rc = no_partials_here(h, "mxv_rollout_mixed")) {
            if (h != h0) h0->error = h->error;
            return rc;
        }
        if (h->param_mode() != PM_DEFAULT || h->step_noise || (h->cfg.flags & MXV_FLAG_NO_AUTORESET))
            return fail(h0, MXV_ERR_UNSUPPORTED, "segment %d needs the per-handle kernels (non-default physics attributes or no autoreset): "
                                                 "launch it with mxv_rollout", i);
        StepArgs &a = m.seg[i];
        fill_step_args(h, a);
        a.K = K;
        a.slice = per_step ? h->cfg.num_envs : 0;
        a.actions = nullptr;
        a.actions_out = outs[i].actions_out;
        a.obs = outs[i].obs;
        a.reward = outs[i].reward;
        a.terminated = outs[i].terminated;
        a.truncated = outs[i].truncated;
        a.final_obs = outs[i].final_obs;
        a.snap_obs = h->snap_obs;
        a.snap_reward = h->snap_reward;
        a.snap_terminated = h->snap_term;
        a.snap_truncated = h->snap_trunc;
        if (int rc = check_step_buffers(h, a)) {
            if (h != h0) h0->error = h->error;
            return rc;
        }
        m.kind[i] = h->cfg.env_id;
        m.first_block[i] = blocks;
        blocks += (uint32_t)((h->cfg.num_envs + 63) / 64);  // one env per lane, one wave per workgroup
    }
    m.first_block[count] = blocks;
    // the launch goes to the first handle's stream; the other handles' streams are ordered before and after it on the GPU
    for (int i = 1; i < count; ++i) {
        mxv_handle *h = handles[i];
        if (h->stream == h0->stream) continue;
        if (!h->ev_mixed) MXV_HIP(h0, hipEventCreateWithFlags(&h->ev_mixed, hipEventDisableTiming));
        MXV_HIP(h0, hipEventRecord(h->ev_mixed, h->stream));
        MXV_HIP(h0, hipStreamWaitEvent(h0->stream, h->ev_mixed, 0));
    }
    MXV_HIP(h0, launch_mixed_rollout(m, h0->stream));
    for (int i = 0; i < count; ++i) {
        mxv_handle *h = handles[i];
        h->state_injected = false;
        if (i > 0 && h->stream != h0->stream) {
            MXV_HIP(h0, hipEventRecord(h->ev_mixed, h0->stream));
            MXV_HIP(h0, hipStreamWaitEvent(h->stream, h->ev_mixed, 0));
        }
        if (int rc = clock_add(h, K)) return rc;   // (device-clock mode: on the segment's own stream, behind the launch)
    }
    return MXV_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Collectives behind the C ABI: the one exchange of a sharded vector env is what np.stack does in the
// reference (gym/vector/sync_vector_env.py:159-169, gym/vector/utils/numpy_utils.py:49-50; AsyncVectorEnv gathers its workers'
// results the same way, async_vector_env.py:319-346) — concatenating the shards' outputs in rank = global-env order.  RCCL is
// used directly (ncclAllGather over xGMI), loaded with dlopen so that libmxv.so itself has no link-time dependency on it.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

RcclApi &rccl() {
    static RcclApi api = [] {
        RcclApi a;
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.lib) break;
        }
        if (!a.lib) {
            a.error = std::string("cannot load librccl.so: ") + (dlerror() ? dlerror() : "not found");
            return a;
        }
        auto sym = [&](const char *n) -> void * {
            void *p = dlsym(a.lib, n);
            if (!p && a.error.empty()) a.error = std::string("librccl.so lacks ") + n;
            return p;
        };
        a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
        a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
        a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
        a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
        return a;
    }();
    return api;
}

#define MXV_NCCL(h, expr)                                                                                              \
    do {                                                                                                               \
        ncclResult_t r_ = (expr);                                                                                      \
        if (r_ != ncclSuccess) return fail((h), MXV_ERR_HIP, "%s: %s", #expr, rccl().GetErrorString(r_));             \
    } while (0)

}  // namespace

extern "C" {

int mxv_comm_unique_id(void *id_out) {
    static_assert(sizeof(ncclUniqueId) == MXV_COMM_ID_BYTES, "MXV_COMM_ID_BYTES must be RCCL's unique-id size");
    if (!id_out) return fail(nullptr, MXV_ERR_INVALID_ARG, "unique-id pointer is NULL");
    RcclApi &api = rccl();
    if (!api.error.empty()) return fail(nullptr, MXV_ERR_UNSUPPORTED, "%s", api.error.c_str());
    ncclUniqueId id;
    MXV_NCCL(nullptr, api.GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof id);
    return MXV_OK;
}

int mxv_comm_init(mxv_handle *h, int32_t rank, int32_t world, const void *unique_id) {
    MXV_CHECK_HANDLE(h);
    if (!unique_id || world < 1 || rank < 0 || rank >= world)
        return fail(h, MXV_ERR_INVALID_ARG, "mxv_comm_init: rank %d of %d with id %p", rank, world, unique_id);
    RcclApi &api = rccl();
    if (!api.error.empty()) return fail(h, MXV_ERR_UNSUPPORTED, "%s", api.error.c_str());
    if (int rc = use_device(h)) return rc;
    if (int rc = mxv_comm_destroy(h)) return rc;
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    MXV_NCCL(h, api.CommInitRank(&h->comm, world, id, rank));
    h->comm_rank = rank;
    h->comm_world = world;
    int lo = 0, hi = 0;  // the gather's kernels get CUs as soon as rollout waves retire instead of queueing behind the next launch
    MXV_HIP(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
    MXV_HIP(h, hipStreamCreateWithPriority(&h->comm_stream, hipStreamNonBlocking, hi));
    MXV_HIP(h, hipEventCreateWithFlags(&h->ev_outputs, hipEventDisableTiming));
    MXV_HIP(h, hipEventCreateWithFlags(&h->ev_gathered[0], hipEventDisableTiming));
    MXV_HIP(h, hipEventCreateWithFlags(&h->ev_gathered[1], hipEventDisableTiming));
    h->gathers = 0;
    return MXV_OK;
}

int mxv_comm_destroy(mxv_handle *h) {
    MXV_CHECK_HANDLE(h);
    if (!h->comm) return MXV_OK;
    (void)hipSetDevice(h->cfg.device);
    if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
    (void)rccl().CommDestroy(h->comm);
    h->comm = nullptr;
    if (h->ev_outputs) (void)hipEventDestroy(h->ev_outputs);
    for (hipEvent_t &e : h->ev_gathered) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
    h->ev_outputs = nullptr;
    h->comm_stream = nullptr;
    h->comm_world = 0;
    h->gathers = 0;
    return MXV_OK;
}

int mxv_allgather_outputs(mxv_handle *h, const float *obs_dev, const void *reward_dev, const uint8_t *terminated_dev,
                          const uint8_t *truncated_dev, float *all_obs_dev, void *all_reward_dev, uint8_t *all_terminated_dev,
                          uint8_t *all_truncated_dev) {
    MXV_CHECK_HANDLE(h);
    if (!h->comm) return fail(h, MXV_ERR_INVALID_ARG, "mxv_allgather_outputs: call mxv_comm_init first");
    if (int rc = use_device(h)) return rc;
    RcclApi &api = rccl();
    const size_t n = (size_t)h->cfg.num_envs;
    // the gather reads what the handle's stream has produced so far; it runs on the communicator's own stream
    MXV_HIP(h, hipEventRecord(h->ev_outputs, h->stream));
    MXV_HIP(h, hipStreamWaitEvent(h->comm_stream, h->ev_outputs, 0));
    MXV_NCCL(h, api.GroupStart());  // the four gathers travel as one fused RCCL launch
    ncclResult_t r = ncclSuccess;
    if (obs_dev && all_obs_dev && r == ncclSuccess)
        r = api.AllGather(obs_dev, all_obs_dev, n * h->O * sizeof(float), ncclUint8, h->comm, h->comm_stream);
    if (reward_dev && all_reward_dev && r == ncclSuccess)
        r = api.AllGather(reward_dev, all_reward_dev, n * h->reward_bytes(), ncclUint8, h->comm, h->comm_stream);
    if (terminated_dev && all_terminated_dev && r == ncclSuccess)
        r = api.AllGather(terminated_dev, all_terminated_dev, n, ncclUint8, h->comm, h->comm_stream);
    if (truncated_dev && all_truncated_dev && r == ncclSuccess)
        r = api.AllGather(truncated_dev, all_truncated_dev, n, ncclUint8, h->comm, h->comm_stream);
    const ncclResult_t e = api.GroupEnd();
    if (r != ncclSuccess || e != ncclSuccess)
        return fail(h, MXV_ERR_HIP, "ncclAllGather: %s", api.GetErrorString(r != ncclSuccess ? r : e));
    MXV_HIP(h, hipEventRecord(h->ev_gathered[h->gathers & 1], h->comm_stream));
    h->gathers += 1;
    return MXV_OK;
}

int mxv_allgather_wait(mxv_handle *h, int32_t age, int32_t host_sync) {
    MXV_CHECK_HANDLE(h);
    if (age < 0 || age > 1) return fail(h, MXV_ERR_INVALID_ARG, "mxv_allgather_wait: age must be 0 (last gather) or 1 (the one before)");
    if (h->gathers <= (uint64_t)age) return MXV_OK;  // no such gather yet
    if (int rc = use_device(h)) return rc;
    hipEvent_t ev = h->ev_gathered[(h->gathers - 1 - (uint64_t)age) & 1];
    if (host_sync)
        MXV_HIP(h, hipEventSynchronize(ev));
    else
        MXV_HIP(h, hipStreamWaitEvent(h->stream, ev, 0));
    return MXV_OK;
}

int mxv_comm_stream(mxv_handle *h, void **stream) {
    MXV_CHECK_HANDLE(h);
    if (!stream) return fail(h, MXV_ERR_INVALID_ARG, "stream pointer is NULL");
    *stream = (void *)h->comm_stream;
    return MXV_OK;
}

}  // extern "C"
