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
// Cards: from the engine's Philox draw stream (draw_call below: one call per step, eight cards with fixed roles) — or, for bit-exact
// replays of the reference, injected (`cards_dev`: int8 [N][MXV_BJ_MAX_DRAWS] in the reference's consumption order: the hit card or
// the dealer's cards, then on termination the new dealer hand and player hand).  Explicit reset: the reset stream (words x,y =
// dealer, z,w = player).  One env per lane; per env-step 3 int64 observations + reward + 2 flags + action = 42 B stored in the
// reference's dtypes (22 B with the contract's 4-byte scalars: mxv_bj_rollout_compact).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "mxv_device.hpp"

using namespace mxv;

namespace {

constexpr int kBjBlock = 256;
constexpr uint32_t kStreamDraw = 5u;
constexpr unsigned kBjXcds = 8;   // MI355X: consecutive workgroup ids go round the 8 XCDs
#ifndef MXV_BJ_TILEMAP
#define MXV_BJ_TILEMAP 1          // 1: XCD x steps (and stores) the x-th contiguous eighth of the tables; 0: tile = workgroup id (A/B hook)
#endif
__device__ __forceinline__ unsigned bj_tile(unsigned bid, unsigned ntiles) {
#if MXV_BJ_TILEMAP
    const unsigned x = bid % kBjXcds, idx = bid / kBjXcds;
    const unsigned base = ntiles / kBjXcds, rem = ntiles % kBjXcds;
    return x * base + (x < rem ? x : rem) + idx;
#else
    return bid;
#endif
}

struct Hand {
    int sum, ace, two;
    __device__ __forceinline__ bool usable() const { return ace && sum + 10 <= 21; }   // :26-27
    __device__ __forceinline__ int total() const { return usable() ? sum + 10 : sum; }  // :30-33
    __device__ __forceinline__ int score() const { return total() > 21 ? 0 : total(); } // :36-41
    __device__ __forceinline__ bool natural() const { return two && ace && sum == 11; } // :44-45
    __device__ __forceinline__ void add(int c) { sum += c; ace |= (c == 1); }
    // add(c) where `on`, nothing otherwise — without a branch
    __device__ __forceinline__ void add_if(bool on, int c) { sum += on ? c : 0; ace |= (int)(on & (c == 1)); two &= (int)!on; }
};

__device__ __forceinline__ int total_of(int sum, int ace) { return sum + ((ace && sum <= 11) ? 10 : 0); }   // sum_hand (:30-33)
__device__ __forceinline__ int card_of_index(uint32_t i) { return (int)(i < 9u ? i + 1u : 10u); }   // deck[i], deck = [1..10, 10, 10, 10] (:15)
__device__ __forceinline__ int card_of(uint32_t w) { return card_of_index((uint32_t)(((uint64_t)w * 13u) >> 32)); }   // one card per word (explicit resets)
// Two cards per word: the word as a base-13 fraction, its first two digits (d0 = floor(13 x), d1 = floor(13 frac(13 x))).  Jointly
// uniform over the 169 pairs up to 169 / 2^32 = 4e-8 relative (the one-card map is uniform up to 13 / 2^32), rejection-free.
__device__ __forceinline__ void cards_of(uint32_t w, int &c0, int &c1) {
    const uint64_t p = (uint64_t)w * 13u;
    c0 = card_of_index((uint32_t)(p >> 32));
    c1 = card_of_index((uint32_t)(((uint64_t)(uint32_t)p * 13u) >> 32));
}

struct BjArgs {
    int32_t *state, *elapsed;
    const uint64_t *seeds;
    const int64_t *actions;   // [N] / tape [K][N] or nullptr -> sampled
    void *actions_out;        // int64 (OUT = 1) / int32 (OUT = 2)
    const int8_t *cards;      // injected draws [N][MXV_BJ_MAX_DRAWS] or nullptr -> Philox
    void *obs;                // [3][N] (or [K][3][N]): player total, dealer's first card, usable ace; int64 / int32
    void *reward;             // float64 / float32
    uint8_t *terminated, *truncated;
    void *final_obs;          // [3][N] / [K][3][N], columns of finished envs only
    int32_t *err;
    int64_t n;
    uint64_t env0, base_seed, action_seed, t;
    const uint64_t *t_dev;    // device clock (mxv_bj_set_device_clock): the step index = t + *t_dev; nullptr: t
    int32_t max_steps, K, natural, sab;
    int64_t slice, act_slice;
    // episode statistics (gym/wrappers/record_episode_statistics.py:96-151), all nullptr when disabled (mxv_bj_episode_stats)
    float *ep_acc;           // [N] running episode return (float32, the reference's accumulator dtype)
    float *ep_return_out;    // [N] / [K][N], written only where terminated | truncated
    int32_t *ep_length_out;  // [N] / [K][N]
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

// The draw stream of one env at vector step t (round 5; rounds 2-4 drew one card per word, consumed in the reference's order, which cost
// two to three Philox calls in every step of every wave).  ONE call per step, ctr = (t_lo, t_hi, 0, 5 << 28) under the env's seed, yields
// eight cards, two per word (cards_of), with FIXED roles:
//     words x, y -> cards 0..3 : the hit card (card 0) or the dealer's first four draws of a stick;
//     word  z    -> cards 4, 5 : the next episode's dealer hand;      word w -> cards 6, 7 : the next episode's player hand.
// A dealer that draws a fifth card and more (a hand of seven cards: ~1 stick in 400) continues with call index 1, 2, ...: draw j >= 4 is
// card (j + 4) & 7 of call (j + 4) >> 3.  Every card is used at most once and all are independent uniform deck draws, so an episode is
// distributed exactly as the reference's (up to the 4e-8 of cards_of); a step is straight-line code: no cursor, no packing, no second call.
__device__ __forceinline__ U4 draw_call(uint64_t seed, uint64_t t, uint32_t i) {
    U4 ctr;
    ctr.x = (uint32_t)t; ctr.y = (uint32_t)(t >> 32); ctr.z = i; ctr.w = (kStreamDraw << 28);
    return philox4x32_10_vkey(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
}
__device__ __noinline__ int late_draw(uint64_t seed, uint64_t t, int j) {   // the dealer's draw j >= 4 (rare; kept out of line)
    const int g = j + 4;
    const U4 w = draw_call(seed, t, (uint32_t)(g >> 3));
    const int q = (g >> 1) & 3;
    int c0, c1;
    cards_of(q == 0 ? w.x : (q == 1 ? w.y : (q == 2 ? w.z : w.w)), c0, c1);
    return (g & 1) ? c1 : c0;
}

__device__ __forceinline__ uint32_t bj_pin32(uint32_t v) {  // see pin32 in mxv_kernels.hip: keeps a store's saddr + 32-bit voffset form
    asm volatile("" : "+v"(v));
    return v;
}

template <int OUT> struct BjOut;
template <> struct BjOut<1> { using I = int64_t; using R = double; };   // the reference's dtypes
template <> struct BjOut<2> { using I = int32_t; using R = float; };    // the contract's 4-byte scalars

// One vector step of Blackjack-v1 for K steps, one table per lane, the hands in registers.
//   INJ     : cards come from `a.cards` in the reference's consumption order (bit-exact replays; K = 1) instead of the draw stream.
//   SAMPLED : actions from the engine's Discrete(2) stream (one random bit per step: include/mxv.h, RNG contract) instead of a.actions.
//   OUT     : output dtypes.
// Both arms of `if action:` (:123-146) are evaluated for every lane and selected — under random or learned policies every wave holds
// hitters and stickers, so a branch would run both anyway, plus its bookkeeping.
#ifndef MXV_BJ_WAVES
#define MXV_BJ_WAVES 8   // 8: 64 VGPRs and <= 96 SGPRs = two full rounds of the 16 waves per SIMD a 2^20-table launch needs (no spills); 0: the allocator's choice (63 VGPRs but 7 waves: SGPRs) (A/B hook)
#endif
// STATS (round 6): the episode-statistics accumulators (mxv_bj_episode_stats) are an instantiation of their own — the kernel sits exactly
// at the 64-VGPR budget of 8 waves per SIMD, and as a run-time branch the three extra live values put 32 bytes of every launch's lanes
// into scratch, statistics or not.  The STATS = true twins may take 7 waves.
template <bool INJ, bool SAMPLED, int OUT, bool STATS = false>
__global__ void __launch_bounds__(kBjBlock)
#if MXV_BJ_WAVES > 0
    __attribute__((amdgpu_waves_per_eu(STATS ? 4 : MXV_BJ_WAVES, MXV_BJ_WAVES)))
#endif
    bj_kernel(BjArgs a) {
    using I = typename BjOut<OUT>::I;
    using R = typename BjOut<OUT>::R;
    constexpr uint32_t IB = sizeof(I), RB = sizeof(R);
    const uint32_t tid = threadIdx.x;
    const int64_t tile0 = (int64_t)bj_tile(blockIdx.x, gridDim.x) * kBjBlock;
    const int64_t e = tile0 + tid;
    if (e >= a.n) return;
    const uint64_t ge = a.env0 + (uint64_t)e;
    const uint64_t seed = a.seeds ? a.seeds[e] : a.base_seed + ge;
    Hand p, d;
    int dfirst;
    unpack(a.state[e], p, d, dfirst);
    int32_t el = a.elapsed[e];
    constexpr bool stats = STATS;
    float er = stats ? a.ep_acc[e] : 0.0f;
    const int8_t *inj = INJ ? a.cards + (size_t)e * MXV_BJ_MAX_DRAWS : nullptr;
    uint64_t act_block = ~0ull;
    uint32_t act_bits = 0;
    const uint64_t t_base = a.t + (a.t_dev ? *a.t_dev : 0);
    // Rows of the current step as wave-uniform byte pointers (scalar registers) that advance by one slice per step; the lane adds a small
    // 32-bit offset (pin32: the stores keep the scalar-base + 32-bit-voffset form, no 64-bit address arithmetic in vector registers).
    const int64_t col = (a.slice ? a.slice : a.n) * (int64_t)IB;      // distance between the three observation columns, bytes
    char *p_o0 = static_cast<char *>(a.obs) + tile0 * IB, *p_o1 = p_o0 + col, *p_o2 = p_o1 + col;
    char *p_f0 = a.final_obs ? static_cast<char *>(a.final_obs) + tile0 * IB : nullptr, *p_f1 = p_f0 + col, *p_f2 = p_f1 + col;
    char *p_act = a.actions_out ? static_cast<char *>(a.actions_out) + tile0 * IB : nullptr;
    char *p_rew = a.reward ? static_cast<char *>(a.reward) + tile0 * RB : nullptr;
    uint8_t *p_term = a.terminated ? a.terminated + tile0 : nullptr, *p_trunc = a.truncated ? a.truncated + tile0 : nullptr;
    const int64_t* p_in = SAMPLED ? nullptr : a.actions + tile0;
    const int64_t step_o = a.slice * 3 * (int64_t)IB, step_i = a.slice * (int64_t)IB, step_r = a.slice * (int64_t)RB, step_b = a.slice;
    const uint32_t off_i = tid * IB, off_r = tid * RB, off_b = tid;
    mxv::settle_entry_loads();
    for (int k = 0; k < a.K; ++k, p_o0 += step_o, p_o1 += step_o, p_o2 += step_o, p_f0 += step_o, p_f1 += step_o, p_f2 += step_o,
                                  p_act += step_i, p_rew += step_r, p_term += step_b, p_trunc += step_b, p_in += a.act_slice) {
        const uint64_t t = t_base + (uint64_t)k;
        int act;
        if constexpr (SAMPLED) {
            if ((t >> 5) != act_block) {    // uniform across the launch: one Philox call per 32 steps
                act_block = t >> 5;
                const U4 w = env_action_words<MXV_CARTPOLE>(a.action_seed, t, ge >> 2);   // the Discrete(2) bit stream (kStreamActionBits)
                const uint32_t q = (uint32_t)(ge & 3);
                act_bits = q == 0 ? w.x : (q == 1 ? w.y : (q == 2 ? w.z : w.w));
            }
            act = (int)((act_bits >> ((uint32_t)t & 31u)) & 1u);
            if (a.actions_out) *reinterpret_cast<I *>(p_act + bj_pin32(off_i)) = (I)act;
        } else {
            const int64_t av = p_in[tid];
            if (av < 0 || av > 1) {  // `assert self.action_space.contains(action)` (:122)
                *reinterpret_cast<volatile int32_t *>(a.err) = 1;  // single-bit code: a plain store (the word may live in pinned host memory)
                continue;
            }
            act = (int)av;
        }
        int c[8];
        if constexpr (INJ) {
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = inj[j];
        } else {
            const U4 w = draw_call(seed, t, 0);
            cards_of(w.x, c[0], c[1]);
            cards_of(w.y, c[2], c[3]);
            cards_of(w.z, c[4], c[5]);
            cards_of(w.w, c[6], c[7]);
        }
        // (plain ints and selects from here on: struct-valued ?: made the compiler keep the hands in scratch memory)
        // hit (:123-130): the player draws card 0
        const int hsum = p.sum + c[0], hace = p.ace | (int)(c[0] == 1);
        const bool bust = total_of(hsum, hace) > 21;
        // stick (:131-146): the dealer draws until its total reaches 17
        int ssum = d.sum, sace = d.ace, stwo = d.two, n = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool need = total_of(ssum, sace) < 17;
            ssum += need ? c[j] : 0;
            sace |= (int)(need & (c[j] == 1));
            stwo = need ? 0 : stwo;
            n += need ? 1 : 0;
        }
        if (__builtin_expect(__any(!act && total_of(ssum, sace) < 17), 0)) {
            // (injected cards come from the caller: a zero among them must not keep the dealer drawing forever — the loop is bounded by the deck's
            // length; the engine's own draws are 1..10 and end within 17)
            while (!act && total_of(ssum, sace) < 17 && (!INJ || n < MXV_BJ_MAX_DRAWS)) {
                const int cj = INJ ? (int)inj[n < MXV_BJ_MAX_DRAWS ? n : MXV_BJ_MAX_DRAWS - 1] : late_draw(seed, t, n);
                ssum += cj; sace |= (int)(cj == 1); stwo = 0;
                ++n;
            }
        }
        const int pt = p.total(), st = total_of(ssum, sace);
        const int ps = pt > 21 ? 0 : pt, ds = st > 21 ? 0 : st;    // score (:36-41)
        const bool pnat = p.natural(), snat = stwo && sace && ssum == 11;
        int r2 = 2 * ((int)(ps > ds) - (int)(ps < ds));             // twice the reward: cmp (:10-11)
        if (a.sab) r2 = (pnat && !snat) ? 2 : r2;                    // :143-145
        else if (a.natural) r2 = (pnat && r2 == 2) ? 3 : r2;         // a winning natural pays 1.5 (:146-148)
        r2 = act ? (bust ? -2 : 0) : r2;
        const bool term = act ? bust : true;
        p.sum = act ? hsum : p.sum; p.ace = act ? hace : p.ace; p.two = act ? 0 : p.two;
        d.sum = act ? d.sum : ssum; d.ace = act ? d.ace : sace; d.two = act ? d.two : stwo;
        el += 1;
        const bool trunc = a.max_steps > 0 && el >= a.max_steps;
        const bool done = term || trunc;
        if constexpr (stats) {   // record_episode_statistics.py:119-143: float32 array += float64 reward (-1, 0, 1, 1.5: every partial sum is exact)
            er = (float)((double)er + 0.5 * (double)r2);
            if (__any(done)) {   // whole lines (zeros where no episode ended): 73 % of the tables finish per step, see mxv_tab.hip
                const int64_t so = (int64_t)k * a.slice + e;     // (addresses formed here, from the argument segment: nothing extra lives across the loop)
                if (a.ep_return_out) a.ep_return_out[so] = done ? er : 0.0f;
                if (a.ep_length_out) a.ep_length_out[so] = done ? el : 0;
            }
            er = done ? 0.0f : er;
        }
        if (a.final_obs && done) {                             // sync_vector_env.py:152-156
            *reinterpret_cast<I *>(p_f0 + bj_pin32(off_i)) = (I)p.total();
            *reinterpret_cast<I *>(p_f1 + bj_pin32(off_i)) = (I)dfirst;
            *reinterpret_cast<I *>(p_f2 + bj_pin32(off_i)) = (I)(p.usable() ? 1 : 0);
        }
        if constexpr (INJ) {                                   // reset (:157-158): the four cards after the step's draws, the dealer's hand first
            const int cur = act ? 1 : n;
#pragma unroll
            for (int j = 0; j < 4; ++j) c[4 + j] = inj[cur + j < MXV_BJ_MAX_DRAWS ? cur + j : MXV_BJ_MAX_DRAWS - 1];
        }
        d.sum = done ? c[4] + c[5] : d.sum; d.ace = done ? (int)((c[4] == 1) | (c[5] == 1)) : d.ace; d.two = done ? 1 : d.two;
        dfirst = done ? c[4] : dfirst;
        p.sum = done ? c[6] + c[7] : p.sum; p.ace = done ? (int)((c[6] == 1) | (c[7] == 1)) : p.ace; p.two = done ? 1 : p.two;
        el = done ? 0 : el;
        *reinterpret_cast<I *>(p_o0 + bj_pin32(off_i)) = (I)p.total();
        *reinterpret_cast<I *>(p_o1 + bj_pin32(off_i)) = (I)dfirst;
        *reinterpret_cast<I *>(p_o2 + bj_pin32(off_i)) = (I)(p.usable() ? 1 : 0);
        if (a.reward) *reinterpret_cast<R *>(p_rew + bj_pin32(off_r)) = (R)(0.5f * (float)r2);    // -1, 0, 1, 1.5: exact
        if (a.terminated) p_term[bj_pin32(off_b)] = term ? 1 : 0;
        if (a.truncated) p_trunc[bj_pin32(off_b)] = trunc ? 1 : 0;
    }
    a.state[e] = pack(p, d, dfirst);
    a.elapsed[e] = el;
    if constexpr (stats) a.ep_acc[e] = er;
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
    float *ep_acc;        // may be nullptr: zeroed for the envs being reset (record_episode_statistics.py:91-94)
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
        if (a.ep_acc) a.ep_acc[e] = 0.0f;
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
    // episode statistics (mxv_bj_episode_stats): running returns, the caller's trajectory outputs, dense staging of host steps
    float *ep_acc = nullptr, *ep_return_out = nullptr, *st_ep_r = nullptr;
    int32_t *ep_length_out = nullptr, *st_ep_l = nullptr;
    bool ep_host_step = false;
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

int bj_aligned(mxv_bj *h, const void *p, size_t bytes, const char *what) {   // see check_aligned in mxv_api.cpp
    if (p && ((uintptr_t)p & (bytes - 1)) != 0) return bfail(h, MXV_ERR_INVALID_ARG, "%s pointer %p is not %zu-byte aligned", what, p, bytes);
    return MXV_OK;
}

int bj_launch(mxv_bj *h, int K, int64_t slice, const int64_t *actions, int64_t act_slice, void *actions_out,
              const int8_t *cards, void *obs, void *reward, uint8_t *term, uint8_t *trunc, void *final_obs, int out_mode = 1) {
    if (!h->was_reset) return bfail(h, MXV_ERR_RESET_NEEDED, "Cannot call step before calling reset (gym.error.ResetNeeded)");
    if (!obs) return bfail(h, MXV_ERR_INVALID_ARG, "obs pointer is NULL");
    if (K <= 0) return bfail(h, MXV_ERR_INVALID_ARG, "K must be positive");
    if (cards && K != 1) return bfail(h, MXV_ERR_INVALID_ARG, "injected cards are per step: K must be 1");
    if (cards && (!actions || out_mode != 1)) return bfail(h, MXV_ERR_INVALID_ARG, "injected cards need given actions and the reference's dtypes");
    {
        const size_t w = out_mode == 1 ? 8 : 4;
        const struct { const void *p; size_t b; const char *what; } t[] = {
            {actions, 8, "actions"}, {actions_out, w, "actions_out"}, {obs, w, "obs"}, {reward, w, "reward"}, {final_obs, w, "final_obs"},
            {h->ep_return_out, 4, "episode return"}, {h->ep_length_out, 4, "episode length"}};
        for (const auto &e : t)
            if (int rc = bj_aligned(h, e.p, e.b, e.what)) return rc;
    }
    BJ_HIP(h, hipSetDevice(h->cfg.device));
    BjArgs a{};
    a.state = h->state; a.elapsed = h->elapsed; a.seeds = h->seeds; a.actions = actions; a.actions_out = actions_out;
    a.cards = cards; a.obs = obs; a.reward = reward; a.terminated = term; a.truncated = trunc; a.final_obs = final_obs;
    a.err = h->err_in_block ? h->hm_err : h->err; a.n = h->cfg.num_envs; a.env0 = (uint64_t)h->cfg.env_offset; a.base_seed = h->base_seed;
    a.action_seed = h->action_seed; a.t = h->dev_clock ? 0 : h->t; a.t_dev = h->dev_clock ? h->t_dev : nullptr;
    a.max_steps = h->cfg.max_episode_steps; a.K = K;
    a.natural = h->cfg.natural; a.sab = h->cfg.sab; a.slice = slice; a.act_slice = act_slice;
    a.ep_acc = h->ep_acc;
    a.ep_return_out = h->ep_acc ? (h->ep_host_step ? h->st_ep_r : h->ep_return_out) : nullptr;
    a.ep_length_out = h->ep_acc ? (h->ep_host_step ? h->st_ep_l : h->ep_length_out) : nullptr;
    const dim3 grid((unsigned)((h->cfg.num_envs + kBjBlock - 1) / kBjBlock)), block(kBjBlock);
    if (a.ep_acc) {
        if (cards) hipLaunchKernelGGL((bj_kernel<true, false, 1, true>), grid, block, 0, h->stream, a);
        else if (out_mode == 1 && actions) hipLaunchKernelGGL((bj_kernel<false, false, 1, true>), grid, block, 0, h->stream, a);
        else if (out_mode == 1) hipLaunchKernelGGL((bj_kernel<false, true, 1, true>), grid, block, 0, h->stream, a);
        else if (actions) hipLaunchKernelGGL((bj_kernel<false, false, 2, true>), grid, block, 0, h->stream, a);
        else hipLaunchKernelGGL((bj_kernel<false, true, 2, true>), grid, block, 0, h->stream, a);
    } else if (cards) hipLaunchKernelGGL((bj_kernel<true, false, 1>), grid, block, 0, h->stream, a);
    else if (out_mode == 1 && actions) hipLaunchKernelGGL((bj_kernel<false, false, 1>), grid, block, 0, h->stream, a);
    else if (out_mode == 1) hipLaunchKernelGGL((bj_kernel<false, true, 1>), grid, block, 0, h->stream, a);
    else if (actions) hipLaunchKernelGGL((bj_kernel<false, false, 2>), grid, block, 0, h->stream, a);
    else hipLaunchKernelGGL((bj_kernel<false, true, 2>), grid, block, 0, h->stream, a);
    BJ_HIP(h, hipGetLastError());
    return bj_clock_add(h, K);
}

int bj_do_reset(mxv_bj *h, const uint8_t *mask_dev, const int8_t *cards_dev, int64_t *obs_dev) {
    if (int rc = bj_aligned(h, obs_dev, 8, "obs")) return rc;
    BJ_HIP(h, hipSetDevice(h->cfg.device));
    h->r += 1;
    BjResetArgs a{};
    a.state = h->state; a.elapsed = h->elapsed; a.seeds = h->seeds; a.mask = mask_dev; a.cards = cards_dev; a.obs = obs_dev;
    a.n = h->cfg.num_envs; a.env0 = (uint64_t)h->cfg.env_offset; a.base_seed = h->base_seed; a.t = h->dev_clock ? 0 : h->t; a.t_dev = h->dev_clock ? h->t_dev : nullptr; a.r = h->r;
    a.ep_acc = h->ep_acc;
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
    if (cfg->num_envs <= 0 || cfg->num_envs > ((int64_t)1 << 28))
        return bfail(nullptr, MXV_ERR_INVALID_ARG, "num_envs must be in [1, 2^28] (one handle; shard larger batches over handles: env_offset)");
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
    void *bufs[] = {h->state, h->elapsed, h->err, h->seeds, h->t_dev, h->ep_acc, h->st_ep_r, h->st_ep_l};
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

int mxv_bj_rollout_compact(mxv_bj *h, int32_t K, int32_t per_step, const int64_t *actions_tape_dev, int32_t *actions_out_dev,
                           int32_t *obs_dev, float *reward_dev, uint8_t *terminated_dev, uint8_t *truncated_dev, int32_t *final_obs_dev) {
    BJ_CHECK(h);
    return bj_launch(h, K, per_step ? h->cfg.num_envs : 0, actions_tape_dev, actions_tape_dev ? h->cfg.num_envs : 0,
                     actions_tape_dev ? nullptr : actions_out_dev, nullptr, obs_dev, reward_dev, terminated_dev, truncated_dev,
                     final_obs_dev, 2);
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
    if (cards_host)   // an injected deck: card values (1 = ace ... 10, blackjack.py:14), 0 = padding behind the draws a step consumes
        for (size_t i = 0; i < (size_t)h->cfg.num_envs * MXV_BJ_MAX_DRAWS; ++i)
            if (cards_host[i] < 0 || cards_host[i] > 10)
                return bfail(h, MXV_ERR_INVALID_ARG, "injected card %d at [%zu][%zu] is no card value (1..10; 0 pads)", (int)cards_host[i],
                             i / MXV_BJ_MAX_DRAWS, i % MXV_BJ_MAX_DRAWS);
    BJ_HIP(h, hipSetDevice(h->cfg.device));
    if (int rc = bj_staging(h)) return rc;
    const size_t n = (size_t)h->cfg.num_envs;
    if (h->hostmap) {   // one launch + one synchronisation: the kernel reads and writes the pinned block itself
        std::memcpy(h->st_actions, actions_host, n * 8);
        if (cards_host) std::memcpy(h->st_cards, cards_host, n * MXV_BJ_MAX_DRAWS);
        h->err_in_block = true;
        h->ep_host_step = true;
        const int lrc = bj_launch(h, 1, 0, h->st_actions, 0, nullptr, cards_host ? h->st_cards : nullptr, h->st_obs, h->st_reward,
                                  h->st_term, h->st_trunc, final_obs_host ? h->st_final : nullptr);
        h->err_in_block = false;
        h->ep_host_step = false;
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
    h->ep_host_step = true;
    const int lrc = bj_launch(h, 1, 0, h->st_actions, 0, nullptr, cards_host ? h->st_cards : nullptr, h->st_obs, h->st_reward,
                              h->st_term, h->st_trunc, final_obs_host ? h->st_final : nullptr);
    h->ep_host_step = false;
    if (lrc) return lrc;
    BJ_HIP(h, hipMemcpyAsync(obs_host, h->st_obs, 3 * n * 8, hipMemcpyDeviceToHost, h->stream));
    if (reward_host) BJ_HIP(h, hipMemcpyAsync(reward_host, h->st_reward, n * 8, hipMemcpyDeviceToHost, h->stream));
    if (terminated_host) BJ_HIP(h, hipMemcpyAsync(terminated_host, h->st_term, n, hipMemcpyDeviceToHost, h->stream));
    if (truncated_host) BJ_HIP(h, hipMemcpyAsync(truncated_host, h->st_trunc, n, hipMemcpyDeviceToHost, h->stream));
    if (final_obs_host) BJ_HIP(h, hipMemcpyAsync(final_obs_host, h->st_final, 3 * n * 8, hipMemcpyDeviceToHost, h->stream));
    int rc = bj_latched(h);
    if (rc == MXV_ERR_INVALID_ACTION) (void)bj_clock_add(h, -1);
    return rc;
}

/* gym.wrappers.RecordEpisodeStatistics fused into the step (record_episode_statistics.py:96-151): see mxv_toytext.h */
int mxv_bj_episode_stats(mxv_bj *h, int32_t enable) {
    BJ_CHECK(h);
    BJ_HIP(h, hipSetDevice(h->cfg.device));
    BJ_HIP(h, hipStreamSynchronize(h->stream));
    const size_t n = (size_t)h->cfg.num_envs;
    if (enable && !h->ep_acc) {
        BJ_HIP(h, hipMalloc((void **)&h->ep_acc, n * sizeof(float)));
        BJ_HIP(h, hipMalloc((void **)&h->st_ep_r, n * sizeof(float)));
        BJ_HIP(h, hipMalloc((void **)&h->st_ep_l, n * sizeof(int32_t)));
        BJ_HIP(h, hipMemsetAsync(h->ep_acc, 0, n * sizeof(float), h->stream));
        BJ_HIP(h, hipMemsetAsync(h->st_ep_r, 0, n * sizeof(float), h->stream));
        BJ_HIP(h, hipMemsetAsync(h->st_ep_l, 0, n * sizeof(int32_t), h->stream));
        BJ_HIP(h, hipStreamSynchronize(h->stream));
    } else if (!enable && h->ep_acc) {
        BJ_HIP(h, hipFree(h->ep_acc));
        BJ_HIP(h, hipFree(h->st_ep_r));
        BJ_HIP(h, hipFree(h->st_ep_l));
        h->ep_acc = h->st_ep_r = nullptr;
        h->st_ep_l = nullptr;
    }
    return MXV_OK;
}

int mxv_bj_set_episode_outputs(mxv_bj *h, float *ep_return_dev, int32_t *ep_length_dev) {
    BJ_CHECK(h);
    BJ_HIP(h, hipSetDevice(h->cfg.device));
    BJ_HIP(h, hipStreamSynchronize(h->stream));
    h->ep_return_out = ep_return_dev;
    h->ep_length_out = ep_length_dev;
    return MXV_OK;
}

int mxv_bj_episode_stats_host(mxv_bj *h, float *ep_return_host, int32_t *ep_length_host, float *running_return_host) {
    BJ_CHECK(h);
    if (!h->ep_acc) return bfail(h, MXV_ERR_INVALID_ARG, "episode statistics are not enabled (mxv_bj_episode_stats)");
    BJ_HIP(h, hipSetDevice(h->cfg.device));
    const size_t n = (size_t)h->cfg.num_envs;
    if (ep_return_host) BJ_HIP(h, hipMemcpyAsync(ep_return_host, h->st_ep_r, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (ep_length_host) BJ_HIP(h, hipMemcpyAsync(ep_length_host, h->st_ep_l, n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    if (running_return_host) BJ_HIP(h, hipMemcpyAsync(running_return_host, h->ep_acc, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    BJ_HIP(h, hipStreamSynchronize(h->stream));
    return MXV_OK;
}

int mxv_bj_set_running_returns(mxv_bj *h, const float *running_return_host) {
    BJ_CHECK(h);
    if (!h->ep_acc) return bfail(h, MXV_ERR_INVALID_ARG, "episode statistics are not enabled (mxv_bj_episode_stats)");
    if (!running_return_host) return bfail(h, MXV_ERR_INVALID_ARG, "NULL pointer");
    BJ_HIP(h, hipSetDevice(h->cfg.device));
    BJ_HIP(h, hipMemcpyAsync(h->ep_acc, running_return_host, (size_t)h->cfg.num_envs * sizeof(float), hipMemcpyHostToDevice, h->stream));
    BJ_HIP(h, hipStreamSynchronize(h->stream));
    return MXV_OK;
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
