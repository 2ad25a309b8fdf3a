// mxv_bj.hip — Blackjack-v1 (gym/envs/toy_text/blackjack.py), the one toy_text env that is not a P-table,
// behind the mxv_bj_* C ABI (include/mxv.h).
//
// Reference: deck = [1..10, 10, 10, 10], draw_card = int(np_random.choice(deck)) (:14-19); a hand's total counts one ace as
// 11 when that does not bust (usable_ace / sum_hand, :26-33); step (:121-148): hit -> the player draws, bust ends the episode
// with -1; stick -> the dealer draws until its total reaches 17, reward = cmp(score(player), score(dealer)), with the
// Sutton-Barto rule (`sab`: a natural beats a non-natural dealer) or the casino rule (`natural`: a winning natural pays 1.5);
// observation = (sum_hand(player), dealer[0], usable_ace(player)) (:150-151); reset (:153-160): dealer = 2 cards, then player
// = 2 cards.  What the dynamics need of a hand is its raw sum (aces as 1), whether it holds an ace, and whether it is still
// the two initial cards (is_natural: sorted(hand) == [1, 10], :44-45) — one packed int32 per env:
//   bits 0-5 player sum | 6 player ace | 7 player has two cards | 8-11 dealer's first card | 12-17 dealer sum | 18 dealer ace |
//   19 dealer has two cards.
// Cards: card = deck[(word * 13) >> 32] from the engine's Philox streams — per step the draw stream (key = env seed,
// ctr = (t_lo, t_hi, call, 5 << 28), four cards per call, consumed in the reference's order: the hit card or the dealer's
// cards, then on termination the new dealer hand and player hand) — or, for bit-exact replays of the reference, injected
// (`cards_dev`: int8 [N][MXV_BJ_MAX_DRAWS] in consumption order).  Explicit reset: the reset stream (words x,y = dealer,
// z,w = player).  One env per lane; per env-step 3 int64 observations + reward + 2 flags + action = 42 B: HBM-bound.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "mxv_device.hpp"

using namespace mxv;

namespace {

#ifndef MXV_BJ_QUAD_ACTIONS
#define MXV_BJ_QUAD_ACTIONS 1   // 1: the lanes of a quad share one action-word Philox call per four steps; 0: one call per lane per step (A/B hook)
#endif
#ifndef MXV_BJ_DPP_TRANSPOSE
#define MXV_BJ_DPP_TRANSPOSE 1   // 1: the quad's action words change lanes by quad_transpose (DPP); 0: three shuffles and select chains (A/B hook)
#endif
#ifndef MXV_BJ_PACKED_DRAWS
#define MXV_BJ_PACKED_DRAWS 1   // 1: the step's first eight cards evaluated up front (CardSource); 0: a Philox call at every draw (A/B hook)
#endif
#ifndef MXV_BJ_NEXT4
#define MXV_BJ_NEXT4 1          // 1: a reset's four cards in one go, the third call decided once per wave (A/B hook)
#endif
constexpr int kBjBlock = 256;
constexpr uint32_t kStreamDraw = 5u;

struct Hand {
    int sum, ace, two;
    __device__ __forceinline__ bool usable() const { return ace && sum + 10 <= 21; }   // :26-27
    __device__ __forceinline__ int total() const { return usable() ? sum + 10 : sum; }  // :30-33
    __device__ __forceinline__ int score() const { return total() > 21 ? 0 : total(); } // :36-41
    __device__ __forceinline__ bool natural() const { return two && ace && sum == 11; } // :44-45
    __device__ __forceinline__ void add(int c) { sum += c; ace |= (c == 1); }
};

__device__ __forceinline__ int card_of(uint32_t w) {
    const int i = (int)(((uint64_t)w * 13u) >> 32);  // index into deck (:15)
    return i < 9 ? i + 1 : 10;
}

struct BjArgs {
    int32_t *state, *elapsed;
    const uint64_t *seeds;
    const int64_t *actions;   // [N] / tape [K][N] or nullptr -> sampled
    int64_t *actions_out;
    const int8_t *cards;      // injected draws [N][MXV_BJ_MAX_DRAWS] or nullptr -> Philox
    int64_t *obs;             // [3][N] (or [K][3][N]): player total, dealer's first card, usable ace
    double *reward;
    uint8_t *terminated, *truncated;
    int64_t *final_obs;       // [3][N] / [K][3][N], columns of finished envs only
    int32_t *err;
    int64_t n;
    uint64_t env0, base_seed, action_seed, t;
    const uint64_t *t_dev;    // device clock (mxv_bj_set_device_clock): the step index = t + *t_dev; nullptr: t
    int32_t max_steps, K, natural, sab;
    int64_t slice, act_slice;
};

// The step's cards.  Philox draws: a step consumes 1 card (a hit that does not bust) to 4 + the dealer's draws + 4 (a stick, then the next
// episode's hands), lane by lane, and a lane-by-lane "refill when the cursor crosses a call boundary" makes the WAVE run a Philox call
// at nearly every draw (some lane always crosses).  Instead the first two calls of the step's draw stream are evaluated once, up front,
// and their eight cards packed four bits each into one register: a draw is a shift and a mask.  Draws past the eighth (a dealer hand
// of five and more cards) evaluate their call on the spot — except the four cards of the next episode's hands, which are taken in one go
// (next4): whether ANY lane of the wave reaches into cards 8..11 is decided once (in a wave of 64 tables some dealer has drawn five
// cards in about every second step) and the third call is then evaluated once, not at each of the four draws.
struct CardSource {
    const int8_t *inj;
    uint64_t seed, t;
    int cursor;
    uint32_t pk;   // cards 0..7 of the step's draw stream, 4 bits each
    uint32_t pk2;  // cards 8..11 (valid inside next4 only)
    __device__ __forceinline__ U4 call(uint32_t i) const {
        U4 ctr;
        ctr.x = (uint32_t)t; ctr.y = (uint32_t)(t >> 32); ctr.z = i; ctr.w = (kStreamDraw << 28);
        return philox4x32_10_vkey(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
    }
    __device__ __forceinline__ void begin() {
        if (inj) return;
#if MXV_BJ_PACKED_DRAWS
        const U4 w0 = call(0), w1 = call(1);
        pk = (uint32_t)card_of(w0.x) | ((uint32_t)card_of(w0.y) << 4) | ((uint32_t)card_of(w0.z) << 8) | ((uint32_t)card_of(w0.w) << 12) |
             ((uint32_t)card_of(w1.x) << 16) | ((uint32_t)card_of(w1.y) << 20) | ((uint32_t)card_of(w1.z) << 24) | ((uint32_t)card_of(w1.w) << 28);
#endif
    }
    static __device__ __forceinline__ uint32_t pack4(const U4 &w) {
        return (uint32_t)card_of(w.x) | ((uint32_t)card_of(w.y) << 4) | ((uint32_t)card_of(w.z) << 8) | ((uint32_t)card_of(w.w) << 12);
    }
    // the next four cards (a reset: dealer's two, player's two)
    __device__ __forceinline__ void next4(int c[4]) {
#if MXV_BJ_PACKED_DRAWS && MXV_BJ_NEXT4
        if (!inj) {
            const bool fast = cursor <= 8;                 // all four inside cards 0..11
            if (__any(fast && cursor > 4)) pk2 = pack4(call(2));   // somebody's four reach past card 7: one call for the wave
            if (fast) {
                const uint64_t both = (uint64_t)pk | ((uint64_t)pk2 << 32);
                const uint32_t four = (uint32_t)(both >> (4 * cursor)) & 0xffffu;
                c[0] = (int)(four & 15u); c[1] = (int)((four >> 4) & 15u); c[2] = (int)((four >> 8) & 15u); c[3] = (int)(four >> 12);
                cursor += 4;
                return;
            }
        }
#endif
#pragma unroll 1
        for (int i = 0; i < 4; ++i) c[i] = next();
    }
    __device__ __forceinline__ int next() {
        int c;
        if (inj) {
            c = inj[cursor < MXV_BJ_MAX_DRAWS ? cursor : MXV_BJ_MAX_DRAWS - 1];
        } else if (MXV_BJ_PACKED_DRAWS && cursor < 8) {
            c = (int)((pk >> (4 * cursor)) & 15u);
        } else {
            const U4 w = call((uint32_t)(cursor >> 2));
            const int q = cursor & 3;
            c = card_of(q == 0 ? w.x : (q == 1 ? w.y : (q == 2 ? w.z : w.w)));
        }
        ++cursor;
        return c;
    }
};

__device__ __forceinline__ void unpack(int32_t s, Hand &p, Hand &d, int &dfirst) {
    p.sum = s & 63; p.ace = (s >> 6) & 1; p.two = (s >> 7) & 1;
    dfirst = (s >> 8) & 15;
    d.sum = (s >> 12) & 63; d.ace = (s >> 18) & 1; d.two = (s >> 19) & 1;
}
__device__ __forceinline__ int32_t pack(const Hand &p, const Hand &d, int dfirst) {
    return p.sum | (p.ace << 6) | (p.two << 7) | (dfirst << 8) | (d.sum << 12) | (d.ace << 18) | (d.two << 19);
}
__device__ __forceinline__ void deal(int c1, int c2, Hand &h) {
    h.sum = c1 + c2; h.ace = (c1 == 1) | (c2 == 1); h.two = 1;
}

#ifndef MXV_BJ_WAVES
#define MXV_BJ_WAVES 0   // 0: the register allocator's choice (89 VGPRs: 5 waves per SIMD); 6 / 8: a budget of 80 / 64 VGPRs (A/B hook)
#endif
__global__ void __launch_bounds__(kBjBlock)
#if MXV_BJ_WAVES > 0
    __attribute__((amdgpu_waves_per_eu(MXV_BJ_WAVES, MXV_BJ_WAVES)))
#endif
    bj_step_kernel(BjArgs a) {
    const int64_t e = (int64_t)blockIdx.x * kBjBlock + threadIdx.x;
    const bool valid = e < a.n;  // sampled actions: lanes past the end stay in the loop, their quad partners need their action words
    if (!valid && a.actions) return;
    const uint64_t ge = a.env0 + (uint64_t)e;
    const uint64_t seed = (a.seeds && valid) ? a.seeds[e] : a.base_seed + ge;
    Hand p, d;
    int dfirst;
    unpack(valid ? a.state[e] : 0, p, d, dfirst);
    int32_t el = valid ? a.elapsed[e] : 0;
    // Action words: one Philox call yields the words of the 4 envs of group g = env >> 2 at ONE step.  The four lanes of a quad (= one
    // group) each evaluate a different step of the aligned block 4 * (t >> 2) .. + 3 and trade words through quad_transpose (two DPP butterfly stages): one call per
    // lane per four steps instead of one per step (the same stream, the same words: mxv_tab.hip's scheme).
    const uint32_t q = (uint32_t)(ge & 3);
    uint64_t act_block = ~0ull;
    uint32_t act_word[4] = {0, 0, 0, 0};
    const uint64_t t_base = a.t + (a.t_dev ? *a.t_dev : 0);
    mxv::settle_entry_loads();
    for (int k = 0; k < a.K; ++k) {
        const uint64_t t = t_base + (uint64_t)k;
        const int64_t o = (int64_t)k * a.slice * 3 + e;   // observation columns: o, o + slice', ...
        const int64_t o1 = (int64_t)k * a.slice + e;
        const int64_t col = a.slice ? a.slice : a.n;      // distance between the three observation columns
        int64_t act;
        if (a.actions) {
            act = a.actions[(int64_t)k * a.act_slice + e];
            if (act < 0 || act > 1) {  // `assert self.action_space.contains(action)` (:122)
                *reinterpret_cast<volatile int32_t *>(a.err) = 1;  // single-bit code: a plain store (the word may live in pinned host memory)
                continue;
            }
        } else {
#if !MXV_BJ_QUAD_ACTIONS
            {   // A/B hook: round 2's one call per lane per step
                const U4 w1 = action_words(a.action_seed, t, ge >> 2);
                act_word[t & 3] = q == 0 ? w1.x : (q == 1 ? w1.y : (q == 2 ? w1.z : w1.w));
            }
#endif
            if (MXV_BJ_QUAD_ACTIONS && (t >> 2) != act_block) {  // uniform across the launch: every lane refills its cache at the same step
                act_block = t >> 2;
                const U4 w = action_words(a.action_seed, (act_block << 2) + q, ge >> 2);
#if MXV_BJ_DPP_TRANSPOSE
                act_word[0] = w.x; act_word[1] = w.y; act_word[2] = w.z; act_word[3] = w.w;
                quad_transpose(act_word, q);   // lane q evaluated step q of the block for the quad's four envs -> its own env's words for steps 0..3
#else
                const uint32_t wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (uint32_t r = 0; r < 4; ++r) {
                    const uint32_t i = q ^ r;
                    const uint32_t send = i == 0 ? wv[0] : (i == 1 ? wv[1] : (i == 2 ? wv[2] : wv[3]));
                    const uint32_t recv = r == 0 ? send : (uint32_t)__shfl_xor((int)send, (int)r, 64);
                    act_word[0] = i == 0 ? recv : act_word[0];
                    act_word[1] = i == 1 ? recv : act_word[1];
                    act_word[2] = i == 2 ? recv : act_word[2];
                    act_word[3] = i == 3 ? recv : act_word[3];
                }
#endif
            }
            const uint32_t j = (uint32_t)(t & 3);
            const uint32_t word = j == 0 ? act_word[0] : (j == 1 ? act_word[1] : (j == 2 ? act_word[2] : act_word[3]));
            act = (int64_t)(((uint64_t)word * 2u) >> 32);
            if (!valid) continue;
            if (a.actions_out) a.actions_out[o1] = act;
        }
        CardSource src{a.cards ? a.cards + e * MXV_BJ_MAX_DRAWS : nullptr, seed, t, 0, 0u, 0u};
        src.begin();
        bool term;
        double rew;
        if (act) {                                             // hit (:123-130)
            p.add(src.next());
            p.two = 0;
            term = p.total() > 21;
            rew = term ? -1.0 : 0.0;
        } else {                                               // stick (:131-146)
            term = true;
            while (d.total() < 17) {
                d.add(src.next());
                d.two = 0;
            }
            const int ps = p.score(), ds = d.score();
            rew = (double)(ps > ds) - (double)(ps < ds);       // cmp (:10-11)
            if (a.sab && p.natural() && !d.natural()) rew = 1.0;
            else if (!a.sab && a.natural && p.natural() && rew == 1.0) rew = 1.5;
        }
        el += 1;
        const bool trunc = a.max_steps > 0 && el >= a.max_steps;
        if (term || trunc) {                                   // sync_vector_env.py:152-156
            if (a.final_obs) {
                a.final_obs[o] = p.total();
                a.final_obs[o + col] = dfirst;
                a.final_obs[o + 2 * col] = p.usable() ? 1 : 0;
            }
            int c4[4];
            src.next4(c4);                                     // reset (:157-158): the dealer's hand first
            deal(c4[0], c4[1], d);
            dfirst = c4[0];
            deal(c4[2], c4[3], p);
            el = 0;
        }
        a.obs[o] = p.total();
        a.obs[o + col] = dfirst;
        a.obs[o + 2 * col] = p.usable() ? 1 : 0;
        if (a.reward) a.reward[o1] = rew;
        if (a.terminated) a.terminated[o1] = term ? 1 : 0;
        if (a.truncated) a.truncated[o1] = trunc ? 1 : 0;
    }
    if (!valid) return;
    a.state[e] = pack(p, d, dfirst);
    a.elapsed[e] = el;
}

struct BjResetArgs {
    int32_t *state, *elapsed;
    const uint64_t *seeds;
    const uint8_t *mask;
    const int8_t *cards;  // injected [N][4] (dealer 2, player 2) or nullptr
    int64_t *obs;
    int64_t n;
    uint64_t env0, base_seed, t;
    const uint64_t *t_dev;
    uint32_t r;
};

__global__ void bj_set_word_kernel(uint64_t *dst, uint64_t v) { *dst = v; }
__global__ void bj_add_word_kernel(uint64_t *dst, uint64_t d) { *dst += d; }

__global__ void __launch_bounds__(kBjBlock) bj_reset_kernel(BjResetArgs a) {
    const int64_t e = (int64_t)blockIdx.x * kBjBlock + threadIdx.x;
    if (e >= a.n) return;
    Hand p, d;
    int dfirst;
    if (a.mask && !a.mask[e]) {
        unpack(a.state[e], p, d, dfirst);
    } else {
        int c[4];
        if (a.cards) {
            for (int i = 0; i < 4; ++i) c[i] = a.cards[e * 4 + i];
        } else {
            const uint64_t seed = a.seeds ? a.seeds[e] : a.base_seed + a.env0 + (uint64_t)e;
            const U4 w = reset_words(seed, a.t + (a.t_dev ? *a.t_dev : 0), a.r);
            c[0] = card_of(w.x); c[1] = card_of(w.y); c[2] = card_of(w.z); c[3] = card_of(w.w);
        }
        deal(c[0], c[1], d);
        dfirst = c[0];
        deal(c[2], c[3], p);
        a.state[e] = pack(p, d, dfirst);
        a.elapsed[e] = 0;
    }
    if (a.obs) {
        a.obs[e] = p.total();
        a.obs[a.n + e] = dfirst;
        a.obs[2 * a.n + e] = p.usable() ? 1 : 0;
    }
}

}  // namespace

struct mxv_bj {
    mxv_bj_config cfg{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int32_t *state = nullptr, *elapsed = nullptr, *err = nullptr;
    uint64_t *seeds = nullptr;
    uint64_t base_seed = 0, action_seed = 0, t = 0;
    uint64_t *t_dev = nullptr;   // device clock (mxv_bj_set_device_clock)
    bool dev_clock = false;
    uint32_t r = 0;
    bool was_reset = false;
    // staging of the *_host calls
    int64_t *st_actions = nullptr, *st_obs = nullptr, *st_final = nullptr;
    double *st_reward = nullptr;
    uint8_t *st_term = nullptr, *st_trunc = nullptr;
    int8_t *st_cards = nullptr;
    // small vector envs: the staging arrays are slices of ONE pinned, device-mapped host block (see mxv_tab.hip)
    char *hm_block = nullptr;
    int32_t *hm_err = nullptr;
    bool hostmap = false, err_in_block = false;
    std::string error;
};

namespace {

thread_local std::string g_bj_create_error;

int bfail(mxv_bj *h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h)
        h->error = buf;
    else
        g_bj_create_error = buf;
    return code;
}

#define BJ_HIP(h, expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return bfail((h), MXV_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define BJ_CHECK(h) \
    if (!(h)) return bfail(nullptr, MXV_ERR_INVALID_ARG, "NULL mxv_bj")

int bj_latched(mxv_bj *h) {
    int32_t e = 0;
    BJ_HIP(h, hipMemcpyAsync(&e, h->err, sizeof e, hipMemcpyDeviceToHost, h->stream));
    BJ_HIP(h, hipStreamSynchronize(h->stream));
    if (e != 0) {
        BJ_HIP(h, hipMemsetAsync(h->err, 0, sizeof(int32_t), h->stream));
        return bfail(h, MXV_ERR_INVALID_ACTION, "action outside {0, 1} (Discrete(2).contains assert, blackjack.py:122)");
    }
    return MXV_OK;
}

// see mxv_set_device_clock (include/mxv.h): the step index on the device, advanced on the stream
int bj_clock_add(mxv_bj *h, int64_t delta) {
    h->t += (uint64_t)delta;
    if (h->dev_clock) {
        hipLaunchKernelGGL(bj_add_word_kernel, dim3(1), dim3(1), 0, h->stream, h->t_dev, (uint64_t)delta);
        BJ_HIP(h, hipGetLastError());
    }
    return MXV_OK;
}
int bj_clock_set(mxv_bj *h) {
    if (h->dev_clock) {
        hipLaunchKernelGGL(bj_set_word_kernel, dim3(1), dim3(1), 0, h->stream, h->t_dev, h->t);
        BJ_HIP(h, hipGetLastError());
    }
    return MXV_OK;
}

int bj_launch(mxv_bj *h, int K, int64_t slice, const int64_t *actions, int64_t act_slice, int64_t *actions_out,
              const int8_t *cards, int64_t *obs, double *reward, uint8_t *term, uint8_t *trunc, int64_t *final_obs) {
    if (!h->was_reset) return bfail(h, MXV_ERR_RESET_NEEDED, "Cannot call step before calling reset (gym.error.ResetNeeded)");
    if (!obs) return bfail(h, MXV_ERR_INVALID_ARG, "obs pointer is NULL");
    if (K <= 0) return bfail(h, MXV_ERR_INVALID_ARG, "K must be positive");
    if (cards && K != 1) return bfail(h, MXV_ERR_INVALID_ARG, "injected cards are per step: K must be 1");
    BJ_HIP(h, hipSetDevice(h->cfg.device));
    BjArgs a{};
    a.state = h->state; a.elapsed = h->elapsed; a.seeds = h->seeds; a.actions = actions; a.actions_out = actions_out;
    a.cards = cards; a.obs = obs; a.reward = reward; a.terminated = term; a.truncated = trunc; a.final_obs = final_obs;
    a.err = h->err_in_block ? h->hm_err : h->err; a.n = h->cfg.num_envs; a.env0 = (uint64_t)h->cfg.env_offset; a.base_seed = h->base_seed;
    a.action_seed = h->action_seed; a.t = h->dev_clock ? 0 : h->t; a.t_dev = h->dev_clock ? h->t_dev : nullptr;
    a.max_steps = h->cfg.max_episode_steps; a.K = K;
    a.natural = h->cfg.natural; a.sab = h->cfg.sab; a.slice = slice; a.act_slice = act_slice;
    const unsigned blocks = (unsigned)((h->cfg.num_envs + kBjBlock - 1) / kBjBlock);
    hipLaunchKernelGGL(bj_step_kernel, dim3(blocks), dim3(kBjBlock), 0, h->stream, a);
    BJ_HIP(h, hipGetLastError());
    return bj_clock_add(h, K);
}

int bj_do_reset(mxv_bj *h, const uint8_t *mask_dev, const int8_t *cards_dev, int64_t *obs_dev) {
    BJ_HIP(h, hipSetDevice(h->cfg.device));
    h->r += 1;
    BjResetArgs a{};
    a.state = h->state; a.elapsed = h->elapsed; a.seeds = h->seeds; a.mask = mask_dev; a.cards = cards_dev; a.obs = obs_dev;
    a.n = h->cfg.num_envs; a.env0 = (uint64_t)h->cfg.env_offset; a.base_seed = h->base_seed; a.t = h->dev_clock ? 0 : h->t; a.t_dev = h->dev_clock ? h->t_dev : nullptr; a.r = h->r;
    const unsigned blocks = (unsigned)((h->cfg.num_envs + kBjBlock - 1) / kBjBlock);
    hipLaunchKernelGGL(bj_reset_kernel, dim3(blocks), dim3(kBjBlock), 0, h->stream, a);
    BJ_HIP(h, hipGetLastError());
    h->was_reset = true;
    return MXV_OK;
}

int bj_staging(mxv_bj *h) {
    if (h->st_obs) return MXV_OK;
    const size_t n = (size_t)h->cfg.num_envs;
    {
        auto up = [](size_t b) { return (b + 255) / 256 * 256; };
        const size_t b8 = up(n * 8), b24 = up(3 * n * 8), b1 = up(n), bc = up(n * MXV_BJ_MAX_DRAWS), total = 2 * b8 + 2 * b24 + 2 * b1 + bc + 256;
        if (total <= (size_t)2 << 20) {
            BJ_HIP(h, hipHostMalloc((void **)&h->hm_block, total, hipHostMallocDefault));
            char *p = h->hm_block;
            h->st_actions = (int64_t *)p; p += b8;
            h->st_obs = (int64_t *)p; p += b24;
            h->st_final = (int64_t *)p; p += b24;
            h->st_reward = (double *)p; p += b8;
            h->st_term = (uint8_t *)p; p += b1;
            h->st_trunc = (uint8_t *)p; p += b1;
            h->st_cards = (int8_t *)p; p += bc;
            h->hm_err = (int32_t *)p;
            *h->hm_err = 0;
            h->hostmap = true;
            return MXV_OK;
        }
    }
    BJ_HIP(h, hipMalloc((void **)&h->st_actions, n * 8));
    BJ_HIP(h, hipMalloc((void **)&h->st_obs, 3 * n * 8));
    BJ_HIP(h, hipMalloc((void **)&h->st_final, 3 * n * 8));
    BJ_HIP(h, hipMalloc((void **)&h->st_reward, n * 8));
    BJ_HIP(h, hipMalloc((void **)&h->st_term, n));
    BJ_HIP(h, hipMalloc((void **)&h->st_trunc, n));
    BJ_HIP(h, hipMalloc((void **)&h->st_cards, n * MXV_BJ_MAX_DRAWS));
    return MXV_OK;
}

}  // namespace

extern "C" {

int mxv_bj_create(const mxv_bj_config *cfg, mxv_bj **out) {
    if (!cfg || !out) return bfail(nullptr, MXV_ERR_INVALID_ARG, "NULL config or output pointer");
    *out = nullptr;
    if (cfg->num_envs <= 0) return bfail(nullptr, MXV_ERR_INVALID_ARG, "num_envs must be positive");
    if (cfg->env_offset < 0 || cfg->env_offset % MXV_ENV_ALIGN != 0)
        return bfail(nullptr, MXV_ERR_INVALID_ARG, "env_offset must be a non-negative multiple of %d", MXV_ENV_ALIGN);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return bfail(nullptr, MXV_ERR_HIP, "no HIP device available (%s): the engine has no CPU fallback",
                     e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (cfg->device < 0 || cfg->device >= ndev) return bfail(nullptr, MXV_ERR_INVALID_ARG, "device %d out of range", cfg->device);
    mxv_bj *h = new (std::nothrow) mxv_bj();
    if (!h) return bfail(nullptr, MXV_ERR_INVALID_ARG, "out of host memory");
    h->cfg = *cfg;
    h->base_seed = cfg->seed;
    h->action_seed = cfg->action_seed;
    const size_t n = (size_t)cfg->num_envs;
    hipError_t err = hipSetDevice(cfg->device);
    if (err == hipSuccess) err = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    h->own_stream = err == hipSuccess;
    if (err == hipSuccess) err = hipMalloc((void **)&h->state, n * 4);
    if (err == hipSuccess) err = hipMalloc((void **)&h->elapsed, n * 4);
    if (err == hipSuccess) err = hipMalloc((void **)&h->err, 4);
    if (err == hipSuccess) err = hipMalloc((void **)&h->t_dev, 8);
    if (err == hipSuccess) err = hipMemsetAsync(h->state, 0, n * 4, h->stream);
    if (err == hipSuccess) err = hipMemsetAsync(h->elapsed, 0, n * 4, h->stream);
    if (err == hipSuccess) err = hipMemsetAsync(h->err, 0, 4, h->stream);
    if (err == hipSuccess) err = hipMemsetAsync(h->t_dev, 0, 8, h->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(h->stream);
    if (err != hipSuccess) {
        bfail(nullptr, MXV_ERR_HIP, "mxv_bj_create: %s", hipGetErrorString(err));
        mxv_bj_destroy(h);
        return MXV_ERR_HIP;
    }
    *out = h;
    return MXV_OK;
}

int mxv_bj_destroy(mxv_bj *h) {
    if (!h) return MXV_OK;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void *bufs[] = {h->state, h->elapsed, h->err, h->seeds, h->t_dev};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    if (h->hostmap) {
        (void)hipHostFree(h->hm_block);
    } else {
        void *stage[] = {h->st_actions, h->st_obs, h->st_final, h->st_reward, h->st_term, h->st_trunc, h->st_cards};
        for (void *p : stage)
            if (p) (void)hipFree(p);
    }
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return MXV_OK;
}

const char *mxv_bj_last_error(const mxv_bj *h) { return h ? h->error.c_str() : g_bj_create_error.c_str(); }

int mxv_bj_seed(mxv_bj *h, uint64_t base_seed, const uint64_t *per_env_seeds_host, uint64_t action_seed) {
    BJ_CHECK(h);
    BJ_HIP(h, hipSetDevice(h->cfg.device));
    BJ_HIP(h, hipStreamSynchronize(h->stream));
    h->base_seed = base_seed;
    h->action_seed = action_seed;
    h->t = 0;
    h->r = 0;
    if (int rc = bj_clock_set(h)) return rc;
    if (per_env_seeds_host) {
        const size_t bytes = (size_t)h->cfg.num_envs * sizeof(uint64_t);
        if (!h->seeds) BJ_HIP(h, hipMalloc((void **)&h->seeds, bytes));
        BJ_HIP(h, hipMemcpy(h->seeds, per_env_seeds_host, bytes, hipMemcpyHostToDevice));
    } else if (h->seeds) {
        BJ_HIP(h, hipFree(h->seeds));
        h->seeds = nullptr;
    }
    return MXV_OK;
}

int mxv_bj_reset(mxv_bj *h, const uint8_t *mask_dev, const int8_t *cards_dev, int64_t *obs_dev) {
    BJ_CHECK(h);
    return bj_do_reset(h, mask_dev, cards_dev, obs_dev);
}

int mxv_bj_step(mxv_bj *h, const int64_t *actions_dev, const int8_t *cards_dev, int64_t *obs_dev, double *reward_dev,
                uint8_t *terminated_dev, uint8_t *truncated_dev, int64_t *final_obs_dev) {
    BJ_CHECK(h);
    if (!actions_dev) return bfail(h, MXV_ERR_INVALID_ARG, "actions pointer is NULL (use mxv_bj_rollout for sampled actions)");
    return bj_launch(h, 1, 0, actions_dev, 0, nullptr, cards_dev, obs_dev, reward_dev, terminated_dev, truncated_dev, final_obs_dev);
}

int mxv_bj_rollout(mxv_bj *h, int32_t K, int32_t per_step, const int64_t *actions_tape_dev, int64_t *actions_out_dev,
                   int64_t *obs_dev, double *reward_dev, uint8_t *terminated_dev, uint8_t *truncated_dev, int64_t *final_obs_dev) {
    BJ_CHECK(h);
    return bj_launch(h, K, per_step ? h->cfg.num_envs : 0, actions_tape_dev, actions_tape_dev ? h->cfg.num_envs : 0,
                     actions_tape_dev ? nullptr : actions_out_dev, nullptr, obs_dev, reward_dev, terminated_dev, truncated_dev,
                     final_obs_dev);
}

int mxv_bj_reset_host(mxv_bj *h, const int8_t *cards_host, int64_t *obs_host) {
    BJ_CHECK(h);
    BJ_HIP(h, hipSetDevice(h->cfg.device));
    if (int rc = bj_staging(h)) return rc;
    const size_t n = (size_t)h->cfg.num_envs;
    if (h->hostmap) {
        if (cards_host) std::memcpy(h->st_cards, cards_host, n * 4);
        if (int rc = bj_do_reset(h, nullptr, cards_host ? h->st_cards : nullptr, obs_host ? h->st_obs : nullptr)) return rc;
        BJ_HIP(h, hipStreamSynchronize(h->stream));
        if (obs_host) std::memcpy(obs_host, h->st_obs, 3 * n * 8);
        return MXV_OK;
    }
    if (cards_host) BJ_HIP(h, hipMemcpyAsync(h->st_cards, cards_host, n * 4, hipMemcpyHostToDevice, h->stream));
    if (int rc = bj_do_reset(h, nullptr, cards_host ? h->st_cards : nullptr, obs_host ? h->st_obs : nullptr)) return rc;
    if (obs_host) BJ_HIP(h, hipMemcpyAsync(obs_host, h->st_obs, 3 * n * 8, hipMemcpyDeviceToHost, h->stream));
    BJ_HIP(h, hipStreamSynchronize(h->stream));
    return MXV_OK;
}

int mxv_bj_step_host(mxv_bj *h, const int64_t *actions_host, const int8_t *cards_host, int64_t *obs_host, double *reward_host,
                     uint8_t *terminated_host, uint8_t *truncated_host, int64_t *final_obs_host) {
    BJ_CHECK(h);
    if (!actions_host || !obs_host) return bfail(h, MXV_ERR_INVALID_ARG, "actions/obs pointer is NULL");
    BJ_HIP(h, hipSetDevice(h->cfg.device));
    if (int rc = bj_staging(h)) return rc;
    const size_t n = (size_t)h->cfg.num_envs;
    if (h->hostmap) {   // one launch + one synchronisation: the kernel reads and writes the pinned block itself
        std::memcpy(h->st_actions, actions_host, n * 8);
        if (cards_host) std::memcpy(h->st_cards, cards_host, n * MXV_BJ_MAX_DRAWS);
        h->err_in_block = true;
        const int lrc = bj_launch(h, 1, 0, h->st_actions, 0, nullptr, cards_host ? h->st_cards : nullptr, h->st_obs, h->st_reward,
                                  h->st_term, h->st_trunc, final_obs_host ? h->st_final : nullptr);
        h->err_in_block = false;
        if (lrc) return lrc;
        BJ_HIP(h, hipStreamSynchronize(h->stream));
        std::memcpy(obs_host, h->st_obs, 3 * n * 8);
        if (reward_host) std::memcpy(reward_host, h->st_reward, n * 8);
        if (terminated_host) std::memcpy(terminated_host, h->st_term, n);
        if (truncated_host) std::memcpy(truncated_host, h->st_trunc, n);
        if (final_obs_host) std::memcpy(final_obs_host, h->st_final, 3 * n * 8);
        if (*h->hm_err != 0) {
            *h->hm_err = 0;
            (void)bj_clock_add(h, -1);
            return bfail(h, MXV_ERR_INVALID_ACTION, "action outside {0, 1} (Discrete(2).contains assert, blackjack.py:122)");
        }
        return MXV_OK;
    }
    BJ_HIP(h, hipMemcpyAsync(h->st_actions, actions_host, n * 8, hipMemcpyHostToDevice, h->stream));
    if (cards_host) BJ_HIP(h, hipMemcpyAsync(h->st_cards, cards_host, n * MXV_BJ_MAX_DRAWS, hipMemcpyHostToDevice, h->stream));
    if (int rc = bj_launch(h, 1, 0, h->st_actions, 0, nullptr, cards_host ? h->st_cards : nullptr, h->st_obs, h->st_reward,
                           h->st_term, h->st_trunc, final_obs_host ? h->st_final : nullptr))
        return rc;
    BJ_HIP(h, hipMemcpyAsync(obs_host, h->st_obs, 3 * n * 8, hipMemcpyDeviceToHost, h->stream));
    if (reward_host) BJ_HIP(h, hipMemcpyAsync(reward_host, h->st_reward, n * 8, hipMemcpyDeviceToHost, h->stream));
    if (terminated_host) BJ_HIP(h, hipMemcpyAsync(terminated_host, h->st_term, n, hipMemcpyDeviceToHost, h->stream));
    if (truncated_host) BJ_HIP(h, hipMemcpyAsync(truncated_host, h->st_trunc, n, hipMemcpyDeviceToHost, h->stream));
    if (final_obs_host) BJ_HIP(h, hipMemcpyAsync(final_obs_host, h->st_final, 3 * n * 8, hipMemcpyDeviceToHost, h->stream));
    int rc = bj_latched(h);
    if (rc == MXV_ERR_INVALID_ACTION) (void)bj_clock_add(h, -1);
    return rc;
}

int mxv_bj_get_state(mxv_bj *h, int32_t *state_host, int32_t *elapsed_host) {
    BJ_CHECK(h);
    BJ_HIP(h, hipSetDevice(h->cfg.device));
    const size_t n = (size_t)h->cfg.num_envs;
    if (state_host) BJ_HIP(h, hipMemcpyAsync(state_host, h->state, n * 4, hipMemcpyDeviceToHost, h->stream));
    if (elapsed_host) BJ_HIP(h, hipMemcpyAsync(elapsed_host, h->elapsed, n * 4, hipMemcpyDeviceToHost, h->stream));
    BJ_HIP(h, hipStreamSynchronize(h->stream));
    return MXV_OK;
}

int mxv_bj_get_counters(mxv_bj *h, uint64_t *t, uint32_t *r) {
    BJ_CHECK(h);
    if (h->dev_clock) {
        BJ_HIP(h, hipSetDevice(h->cfg.device));
        BJ_HIP(h, hipMemcpyAsync(&h->t, h->t_dev, sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
        BJ_HIP(h, hipStreamSynchronize(h->stream));
    }
    if (t) *t = h->t;
    if (r) *r = h->r;
    return MXV_OK;
}

int mxv_bj_set_state(mxv_bj *h, const int32_t *state_host, const int32_t *elapsed_host, uint64_t t, uint32_t r) {
    BJ_CHECK(h);
    BJ_HIP(h, hipSetDevice(h->cfg.device));
    const size_t n = (size_t)h->cfg.num_envs;
    if (state_host) BJ_HIP(h, hipMemcpyAsync(h->state, state_host, n * 4, hipMemcpyHostToDevice, h->stream));
    if (elapsed_host) BJ_HIP(h, hipMemcpyAsync(h->elapsed, elapsed_host, n * 4, hipMemcpyHostToDevice, h->stream));
    BJ_HIP(h, hipStreamSynchronize(h->stream));
    h->t = t;
    h->r = r;
    h->was_reset = true;
    return bj_clock_set(h);
}

int mxv_bj_set_device_clock(mxv_bj *h, int32_t on) {
    BJ_CHECK(h);
    BJ_HIP(h, hipSetDevice(h->cfg.device));
    if (on && !h->dev_clock) {
        h->dev_clock = true;
        return bj_clock_set(h);
    }
    if (!on && h->dev_clock) {
        BJ_HIP(h, hipMemcpyAsync(&h->t, h->t_dev, sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
        BJ_HIP(h, hipStreamSynchronize(h->stream));
        h->dev_clock = false;
    }
    return MXV_OK;
}

int mxv_bj_sync(mxv_bj *h) {
    BJ_CHECK(h);
    BJ_HIP(h, hipSetDevice(h->cfg.device));
    return bj_latched(h);
}

int mxv_bj_set_stream(mxv_bj *h, void *stream) {
    BJ_CHECK(h);
    BJ_HIP(h, hipSetDevice(h->cfg.device));
    if (h->stream) BJ_HIP(h, hipStreamSynchronize(h->stream));
    if (h->own_stream && h->stream) BJ_HIP(h, hipStreamDestroy(h->stream));
    h->stream = (hipStream_t)stream;
    h->own_stream = false;
    return MXV_OK;
}

}  // extern "C"
