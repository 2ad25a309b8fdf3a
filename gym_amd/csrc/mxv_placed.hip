// mxv_placed.hip — device memory for trajectory tensors, placed by HBM class (include/mxv.h: mxv_placed_*).
//
// Why.  The fused rollout writes every step's outputs as a few long, parallel store streams (CartPole: observations 16 B per env and
// step, rewards 8 B, actions 8 B, two flag bytes).  On the MI355X a 16-B/lane stream and an 8-B/lane stream written concurrently run
// 10-12 % slower when the PHYSICAL memory behind them belongs to the same CLASS of HBM regions, and the whole trajectory launch runs
// 5.4 / 5.7 / 6.4 us per 2^20-env step when none / one / both of {rewards, actions} share the observations' class
// (profiles/r3/r3a_*; the probes were removed in round 4, tools/README.md).  The classes are three contiguous thirds of the physical address space —
// 3 x 96 GB, what the three ranks of a 12-high HBM3E stack would give — (tools/vmm_classmap.hip, profiles/r3/r3c_hbm_class_map_whole_device.jsonl):
// a fresh process is handed the first third for its first ~90 GiB, so ordinary allocations all share a class ("slow box") unless
// earlier activity has scrambled the driver's free lists ("fast placement").  That is the placement lottery of DESIGN.md §3.  Nothing in
// software sees the class of a page — but HIP's virtual-memory API decides which physical memory backs which virtual range, and the class
// of a chunk can be MEASURED: two concurrent streams, one into the chunk, one into a reference chunk of a known class.
//
// How.  Physical memory is created in 256-MiB chunks (hipMemCreate); every chunk is mapped at a scratch address of its own and
// classified against one reference chunk per class seen so far; chunks are created until one class can serve the group-0 tensors
// and the other classes the group-1 tensors.  While only one class has been seen and the caller allows it, the search JUMPS: it
// parks unmapped spacer allocations (4 GiB each) so that the next chunk comes from further along in physical memory, until a chunk of
// another class appears.  Spacers and surplus chunks are released before the call returns; every tensor then gets a fresh virtual
// range with chunks of its group's class(es) mapped into it.
//
// Runtime facts this code is written around (ROCm 7.0 / this driver; profiles/r3/r3a_* hold the test runs):
//   * a virtual address that has been mapped once keeps translating to the FIRST physical memory it saw, even after hipMemUnmap and
//     hipMemMap of another handle -> every mapping here uses a fresh address, and no reservation is ever given back
//     (hipMemAddressFree) so the runtime cannot hand a used address out again; freeing leaks virtual address space only;
//   * hipMemSetAccess accepts the START of a reservation only; hipMemUnmap must be called exactly as hipMemMap was.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <initializer_list>
#include <new>
#include <string>
#include <vector>

#include "../../include/mxv.h"
#include "../../include/mxv_diag.h"

namespace {

thread_local std::string g_placed_error;
constexpr size_t kChunk = MXV_PLACED_CHUNK_BYTES;
constexpr int64_t kProbeLanes = 1 << 20;   // the classification probe is the 2^20-env CartPole window: 16 steps x 16 B = one chunk
constexpr int kProbeSteps = 16;

// Two of the rollout's store streams, physics removed: `wide` takes 16 B per lane and step (observations), `narrow` 8 B (rewards).
// One wave per workgroup, two slots per lane, XCD-contiguous tiles — the launch shape of rollout_kernel_v3<CartPole>.
__global__ void __launch_bounds__(64, 4) placed_probe_kernel(float4 *wide, double *narrow, int64_t row, int K) {
    const unsigned bid = blockIdx.x, ntiles = gridDim.x;
    const unsigned tile = (bid % 8) * (ntiles / 8) + bid / 8;
    const int lane = threadIdx.x;
    const int64_t e0 = (int64_t)tile * 128 + lane, e1 = e0 + 64;
    float x = (float)lane;
    for (int k = 0; k < K; ++k) {
        const int64_t so = (int64_t)k * row;
        x = x * 1.0001f + 0.5f;
        narrow[so + e0] = 1.0;
        narrow[so + e1] = 1.0;
        wide[so + e0] = make_float4(x, x + 1, 0.f, 1.f);
        wide[so + e1] = make_float4(x + 2, x, 1.f, 0.f);
    }
}

struct Chunk {
    hipMemGenericAllocationHandle_t handle{};
    char *scratch = nullptr;   // where it is mapped for classification (unmapped before the final mapping)
    bool scratch_mapped = false;
    int klass = -1;
};

struct Tensor {
    size_t bytes = 0, reserved = 0;
    int group = -1;
    char *va = nullptr;
    std::vector<int> chunks;   // indices into mxv_placed::chunks, in mapping order
    void *plain = nullptr;     // hipMalloc path
};

int pfail(mxv_placed *p, int code, const char *fmt, ...);

}  // namespace

struct mxv_placed {
    int device = 0;
    std::vector<hipMemGenericAllocationHandle_t> spacers;   // unmapped allocations parked during the search (released before alloc returns)
    std::vector<Chunk> chunks;
    std::vector<Tensor> tensors;
    mxv_placed_info info{};
    std::string error;
};

namespace {

int pfail(mxv_placed *p, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (p)
        p->error = buf;
    g_placed_error = buf;
    return code;
}

#define PL_HIP(p, expr)                                                                                 \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return pfail((p), MXV_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

struct Prober {
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipMemAllocationProp prop{};
    hipMemAccessDesc acc{};

    hipError_t init(int device) {
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        hipError_t e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        return e;
    }
    ~Prober() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (stream) (void)hipStreamDestroy(stream);
    }
    // us per step of the two-stream window: wide stream over all of `w`, narrow stream over the first half of `n`; best of `reps`
    hipError_t time_pair(const Chunk &w, const Chunk &n, int launches, int reps, float *us) {
        float best = 1e30f;
        for (int rep = 0; rep < reps; ++rep) {
            hipError_t e = hipEventRecord(e0, stream);
            if (e != hipSuccess) return e;
            for (int j = 0; j < launches; ++j)
                hipLaunchKernelGGL(placed_probe_kernel, dim3((unsigned)(kProbeLanes / 128)), dim3(64), 0, stream, reinterpret_cast<float4 *>(w.scratch),
                                   reinterpret_cast<double *>(n.scratch), kProbeLanes, kProbeSteps);
            if ((e = hipGetLastError()) != hipSuccess) return e;
            if ((e = hipEventRecord(e1, stream)) != hipSuccess) return e;
            if ((e = hipEventSynchronize(e1)) != hipSuccess) return e;
            float ms = 0.f;
            if ((e = hipEventElapsedTime(&ms, e0, e1)) != hipSuccess) return e;
            best = std::min(best, ms * 1e3f / (launches * kProbeSteps));
        }
        *us = best;
        return hipSuccess;
    }
};

int new_chunk(mxv_placed *p, Prober &pr) {
    Chunk c;
    PL_HIP(p, hipMemCreate(&c.handle, kChunk, &pr.prop, 0));
    hipError_t e = hipMemAddressReserve(reinterpret_cast<void **>(&c.scratch), kChunk, 0, nullptr, 0);
    if (e == hipSuccess) e = hipMemMap(c.scratch, kChunk, 0, c.handle, 0);
    if (e == hipSuccess) {
        c.scratch_mapped = true;
        e = hipMemSetAccess(c.scratch, kChunk, &pr.acc, 1);
    }
    p->chunks.push_back(c);
    p->info.chunks_created++;
    if (e != hipSuccess) return pfail(p, MXV_ERR_HIP, "mapping a %zu-MiB chunk for classification: %s", kChunk >> 20, hipGetErrorString(e));
    return MXV_OK;
}

void drop_chunk(Chunk &c) {
    if (c.scratch_mapped) {
        (void)hipMemUnmap(c.scratch, kChunk);
        c.scratch_mapped = false;
    }
    if (c.klass != -2) (void)hipMemRelease(c.handle);
    c.klass = -2;   // released
}

}  // namespace

extern "C" {

const char *mxv_placed_last_error(const mxv_placed *p) { return p ? p->error.c_str() : g_placed_error.c_str(); }

int mxv_placed_free(mxv_placed *p) {
    if (!p) return MXV_OK;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    for (Tensor &t : p->tensors) {
        if (t.plain) (void)hipFree(t.plain);
        if (!t.va) continue;   // never mapped (an allocation that failed half-way)
        for (size_t j = 0; j < t.chunks.size(); ++j) (void)hipMemUnmap(t.va + j * kChunk, kChunk);   // exactly as mapped; the range itself is kept (see top)
    }
    for (Chunk &c : p->chunks) drop_chunk(c);
    for (auto h : p->spacers) (void)hipMemRelease(h);   // only non-empty when an allocation failed half-way
    delete p;
    return MXV_OK;
}

int mxv_hbm_pair_probe(int32_t device, void *wide_dev, void *narrow_dev, int32_t launches, double *us_per_step) {
    if (!wide_dev || !narrow_dev || !us_per_step || launches < 1) return pfail(nullptr, MXV_ERR_INVALID_ARG, "mxv_hbm_pair_probe: NULL argument");
    if (hipSetDevice(device) != hipSuccess) return pfail(nullptr, MXV_ERR_HIP, "hipSetDevice(%d) failed", device);
    // the window writes 256 MiB behind wide_dev and 128 MiB behind narrow_dev: the sizes are part of the contract, so they are checked
    // against the allocations the pointers belong to (a short buffer would be a device fault, i.e. the caller's process)
    const struct { void *p; size_t need; const char *what; } spans[] = {{wide_dev, (size_t)256 << 20, "wide_dev"}, {narrow_dev, (size_t)128 << 20, "narrow_dev"}};
    for (const auto &sp : spans) {
        hipDeviceptr_t base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange(&base, &size, sp.p) != hipSuccess || !base) {
            (void)hipGetLastError();
            return pfail(nullptr, MXV_ERR_INVALID_ARG, "mxv_hbm_pair_probe: %s is not inside a device allocation", sp.what);
        }
        const size_t off = (size_t)((char *)sp.p - (char *)base);
        if (((uintptr_t)sp.p & 15) != 0 || off > size || size - off < sp.need)
            return pfail(nullptr, MXV_ERR_INVALID_ARG, "mxv_hbm_pair_probe: %s needs %zu MiB of 16-byte aligned device memory behind it (%zu bytes there)",
                         sp.what, sp.need >> 20, off <= size ? size - off : (size_t)0);
    }
    Prober pr;
    if (hipError_t e = pr.init(device); e != hipSuccess) return pfail(nullptr, MXV_ERR_HIP, "stream / events: %s", hipGetErrorString(e));
    Chunk w, n;
    w.scratch = static_cast<char *>(wide_dev);
    n.scratch = static_cast<char *>(narrow_dev);
    float us = 0.f;
    hipError_t e = pr.time_pair(w, n, 2, 1, &us);   // first touch
    if (e == hipSuccess) e = pr.time_pair(w, n, launches, 3, &us);
    if (e != hipSuccess) return pfail(nullptr, MXV_ERR_HIP, "mxv_hbm_pair_probe: %s", hipGetErrorString(e));
    *us_per_step = us;
    return MXV_OK;
}

int mxv_placed_info_get(const mxv_placed *p, mxv_placed_info *out) {
    if (!p || !out) return pfail(nullptr, MXV_ERR_INVALID_ARG, "mxv_placed_info_get: NULL argument");
    *out = p->info;
    return MXV_OK;
}

int mxv_placed_alloc(int32_t device, int32_t count, const size_t *bytes, const int32_t *group, int32_t flags, void **ptrs_out, mxv_placed **out) {
    if (count < 1 || !bytes || !group || !ptrs_out || !out) return pfail(nullptr, MXV_ERR_INVALID_ARG, "mxv_placed_alloc: NULL argument or count < 1");
    for (int i = 0; i < count; ++i)
        if (bytes[i] == 0 || group[i] < -1 || group[i] > 1) return pfail(nullptr, MXV_ERR_INVALID_ARG, "mxv_placed_alloc: tensor %d: bytes > 0 and group in {-1, 0, 1}", i);
    const auto t_begin = std::chrono::steady_clock::now();
    mxv_placed *p = new (std::nothrow) mxv_placed;
    if (!p) return pfail(nullptr, MXV_ERR_HIP, "out of host memory");
    p->device = device;
    p->tensors.resize(count);
    size_t total = 0;
    int need[3] = {0, 0, 0};   // chunks for group 0, group 1, group -1
    for (int i = 0; i < count; ++i) {
        Tensor &t = p->tensors[i];
        t.bytes = bytes[i];
        t.group = group[i];
        t.reserved = (bytes[i] + kChunk - 1) / kChunk * kChunk;
        total += bytes[i];
        need[group[i] < 0 ? 2 : group[i]] += (int)(t.reserved / kChunk);
    }
    auto bail = [&](int rc) {
        std::string msg = p->error;
        mxv_placed_free(p);
        g_placed_error = msg;
        return rc;
    };
    if (hipSetDevice(device) != hipSuccess) return bail(pfail(p, MXV_ERR_HIP, "hipSetDevice(%d) failed", device));
    p->info.requested_bytes = total;
    const bool want_placed = !(flags & MXV_PLACED_PLAIN) && need[0] > 0 && need[1] > 0 && total >= MXV_PLACED_MIN_BYTES;
    if (!want_placed) {   // small sets (the modes were only ever seen from about a GiB) and one-group sets: ordinary allocations
        for (int i = 0; i < count; ++i) {
            hipError_t e = hipMalloc(&p->tensors[i].plain, bytes[i]);
            if (e != hipSuccess) return bail(pfail(p, MXV_ERR_HIP, "hipMalloc(%zu): %s", bytes[i], hipGetErrorString(e)));
            ptrs_out[i] = p->tensors[i].plain;
        }
        p->info.held_bytes = total;
        *out = p;
        return MXV_OK;
    }

    Prober pr;
    if (hipError_t e = pr.init(device); e != hipSuccess) return bail(pfail(p, MXV_ERR_HIP, "stream / events: %s", hipGetErrorString(e)));
    const int needed = need[0] + need[1] + need[2];
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    // Physical memory the search may hold at any moment.  Chunks: twice the request.  Spacers (the jump, MXV_PLACED_NO_JUMP forbids it):
    // a class is a third of the device (96 GB), so crossing into the next one can take that much — allowed up to half of what is
    // free right now, never into the last 16 GiB.
    const int cap = 2 * needed;
    const size_t kSpacer = (size_t)4 << 30;
    size_t jump_budget = (flags & MXV_PLACED_NO_JUMP) ? 0 : std::min<size_t>(free_b / 2, (size_t)112 << 30);
    if (free_b < ((size_t)16 << 30) + (size_t)cap * kChunk) jump_budget = 0;
#define PL_TIME(w, n, launches, reps, out)                                                                                      \
    do {                                                                                                                        \
        if (hipError_t e_ = pr.time_pair(p->chunks[w], p->chunks[n], launches, reps, out); e_ != hipSuccess)                     \
            return bail(pfail(p, MXV_ERR_HIP, "classification probe: %s", hipGetErrorString(e_)));                               \
    } while (0)
    // Bootstrap on the first three chunks, timed pairwise.  Created back to back they come from one neighbourhood: if all three
    // pairings agree they share a class and (0, 1) is the calibration pair; if one pairing is clearly slower than the fastest, that
    // pair shares a class and the third chunk opens a second one.  The device is spun up first and the timings are repeated until two
    // rounds agree (a cold clock stretches the early ones).
    for (int i = 0; i < 3; ++i)
        if (int rc = new_chunk(p, pr)) return bail(rc);
    const float kDiffRatio = 0.955f;   // measured: a different-class pair runs at 0.89-0.91 of the same-class time
    float t01 = 0, t02 = 0, t12 = 0;
    {
        const auto t0 = std::chrono::steady_clock::now();
        float prev = 0.f, us = 0.f;
        for (int round = 0; round < 40; ++round) {
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.15) PL_TIME(0, 1, 8, 1, &us);
            PL_TIME(0, 1, 4, 2, &t01);
            PL_TIME(0, 2, 4, 2, &t02);
            PL_TIME(1, 2, 4, 2, &t12);
            const float sum = t01 + t02 + t12;
            if (round > 0 && std::fabs(sum - prev) < 0.012f * sum) break;
            prev = sum;
        }
    }
    constexpr int kMaxClasses = 4;
    std::vector<int> refs;       // one reference chunk per class, class k = refs[k]
    int cal_a = 0, cal_b = 1;    // two chunks known to share a class: their pairing, timed next to a candidate's, is "what same-class costs now"
    {
        const float hi = std::max(t01, std::max(t02, t12)), lo = std::min(t01, std::min(t02, t12));
        if (lo > kDiffRatio * hi) {   // all three alike
            refs = {0};
            p->chunks[0].klass = p->chunks[1].klass = p->chunks[2].klass = 0;
        } else {                      // the slowest pair shares a class, the third chunk has another
            const int odd = hi == t01 ? 2 : hi == t02 ? 1 : 0;
            cal_a = odd == 0 ? 1 : 0;
            cal_b = odd == 2 ? 1 : 2;
            refs = {cal_a, odd};
            for (int c = 0; c < 3; ++c) p->chunks[c].klass = c == odd ? 1 : 0;
        }
    }
    float t_same = std::max(t01, std::max(t02, t12)), t_diff = refs.size() > 1 ? std::min(t01, std::min(t02, t12)) : 0.f;
    auto classify = [&](int c) -> int {   // sets chunks[c].klass (>= 0, or -1 = could not tell); MXV_OK or an error
        Chunk &ch = p->chunks[c];
        for (int attempt = 0; attempt < 3; ++attempt) {
            float same = 0.f, worst = 0.f, second = 0.f;
            int worst_k = -1;
            PL_TIME(cal_a, cal_b, 4, 2, &same);
            for (int k = 0; k < (int)refs.size(); ++k) {
                float a = 0.f;
                PL_TIME(c, refs[k], 4, 2, &a);
                if (a > worst) { second = worst; worst = a; worst_k = k; } else if (a > second) second = a;
            }
            const bool slow = worst > kDiffRatio * same;                              // some pairing costs what a same-class pair costs
            const bool clear = refs.size() < 2 || second < kDiffRatio * worst + 0.02f * same;   // ... and only one does
            if (slow && clear) { ch.klass = worst_k; t_same = 0.8f * t_same + 0.2f * worst; return MXV_OK; }
            if (!slow && worst < 0.94f * same && (int)refs.size() < kMaxClasses) {       // fast against every class seen so far: a new class, confirmed once
                float again = 0.f, same2 = 0.f;
                PL_TIME(cal_a, cal_b, 4, 3, &same2);
                PL_TIME(c, refs[worst_k], 4, 3, &again);
                if (again < kDiffRatio * same2) { ch.klass = (int)refs.size(); refs.push_back(c); t_diff = again; return MXV_OK; }
            }
        }
        ch.klass = -1;   // e.g. a chunk that straddles a class boundary: only fit for tensors that do not care
        return MXV_OK;
    };
    int have[kMaxClasses + 1] = {0, 0, 0, 0, 0};   // per class; [kMaxClasses] = undecided
    auto tally = [&](int c) { have[p->chunks[c].klass < 0 ? kMaxClasses : p->chunks[c].klass]++; };
    for (int c = 0; c < 3; ++c) tally(c);
    auto live = [&]() { int n = 0; for (int k = 0; k <= kMaxClasses; ++k) n += have[k]; return n; };
    // one class serves group 0 alone, everything of the other classes serves group 1 — or the other way round
    auto enough = [&](int *solo_class, int *solo_group) {
        int decided = 0;
        for (int k = 0; k < kMaxClasses; ++k) decided += have[k];
        for (int g = 0; g < 2; ++g)
            for (int k = 0; k < kMaxClasses; ++k)
                if (have[k] >= need[g] && decided - have[k] >= need[1 - g] && live() >= needed) { *solo_class = k; *solo_group = g; return true; }
        return false;
    };
    int solo_class = 0, solo_group = 0;
    int peak_live = 3, stop_reason = 0;   // 0 balanced, 1 chunk cap, 2 jump budget, 3 spacer allocation failed, 4 chunk allocation failed
    size_t spacer_bytes = 0, peak_bytes = 3 * kChunk;
    std::vector<hipMemGenericAllocationHandle_t> &spacers = p->spacers;
    auto release_spacers = [&]() {
        for (auto h : spacers) (void)hipMemRelease(h);
        spacers.clear();
        spacer_bytes = 0;
    };
    while (!enough(&solo_class, &solo_group)) {
        if (p->info.chunks_created >= 8 * needed + 64) { stop_reason = 1; break; }   // released blocks come straight back: never loop on them
        if (live() >= cap) {
            // Twice the request and still short of a second class (or of enough of it).  Drop what is surplus of the largest class —
            // the jump below needs the room, and a class can never need more than the larger group — then jump or give up.
            int major = 0;
            for (int k = 1; k < kMaxClasses; ++k) if (have[k] > have[major]) major = k;
            const int keep = std::max(need[0], need[1]) + need[2];
            for (int c = (int)p->chunks.size() - 1; c >= 0 && have[major] > keep; --c)
                if (p->chunks[c].klass == major && c != cal_a && c != cal_b && c != refs[major]) { drop_chunk(p->chunks[c]); have[major]--; }
            if (live() >= cap) { stop_reason = 1; break; }   // nothing to drop: best effort
        }
        const bool one_class_only = live() - have[kMaxClasses] == have[p->chunks[cal_a].klass];
        if (one_class_only && live() >= std::max(need[0], need[1]) + need[2] + 2) {
            // Enough of this class for whichever group ends up on it; more of the same is useless.  Jump: park a spacer so that the next
            // chunk comes from further along in physical memory.
            if (spacer_bytes + kSpacer > jump_budget) { stop_reason = 2; break; }
            // (a spacer is 16 allocations of the chunk size, not one of 4 GiB: the driver serves large requests from elsewhere — 112 GiB of
            //  4-GiB spacers did not move the chunks out of their class, 256-MiB allocations walk through memory in order)
            bool failed = false;
            for (size_t b = 0; b < kSpacer && !failed; b += kChunk) {
                hipMemGenericAllocationHandle_t h;
                if (hipError_t e = hipMemCreate(&h, kChunk, &pr.prop, 0); e != hipSuccess) {
                    pfail(p, MXV_ERR_HIP, "spacer hipMemCreate(%zu): %s", kChunk, hipGetErrorString(e));
                    failed = true;
                } else {
                    spacers.push_back(h);
                    spacer_bytes += kChunk;
                }
            }
            if (failed) { stop_reason = 3; break; }
        }
        const size_t before = p->chunks.size();
        if (new_chunk(p, pr) != MXV_OK) {   // out of device memory: best effort with what there is
            if (p->chunks.size() > before) drop_chunk(p->chunks.back());
            stop_reason = 4;
            break;
        }
        const int c = (int)p->chunks.size() - 1;
        if (int rc = classify(c)) return bail(rc);
        tally(c);
        peak_live = std::max(peak_live, live());
        peak_bytes = std::max(peak_bytes, (size_t)live() * kChunk + spacer_bytes);
        if (one_class_only && p->chunks[c].klass == p->chunks[cal_a].klass && !spacers.empty()) {
            drop_chunk(p->chunks[c]);   // a probe chunk between two spacers that is still the old class: not needed, its room is
            have[p->chunks[cal_a].klass]--;
        }
    }
    const size_t jumped = spacer_bytes;
    release_spacers();
    const bool balanced = enough(&solo_class, &solo_group);
    while (live() < needed) {   // best effort still serves the whole request
        if (int rc = new_chunk(p, pr)) return bail(rc);
        const int c = (int)p->chunks.size() - 1;
        if (int rc = classify(c)) return bail(rc);
        tally(c);
        peak_bytes = std::max(peak_bytes, (size_t)live() * kChunk);
    }
    if (!balanced) {   // best effort: the larger group alone on the largest class
        solo_group = need[0] >= need[1] ? 0 : 1;
        solo_class = 0;
        for (int k = 1; k < kMaxClasses; ++k) if (have[k] > have[solo_class]) solo_class = k;
    }
    if (live() < needed) return bail(pfail(p, MXV_ERR_HIP, "mxv_placed_alloc: %d chunks of %zu MiB needed, %d available (out of device memory?)", needed, kChunk >> 20, live()));

    // hand out: the solo group from its class; the other group from the other classes (then from anything); group -1 from what nobody wants
    std::vector<int> pool_solo, pool_other, pool_any;
    for (int c = 0; c < (int)p->chunks.size(); ++c) {
        const int k = p->chunks[c].klass;
        if (k == -2) continue;
        (k < 0 ? pool_any : k == solo_class ? pool_solo : pool_other).push_back(c);
    }
    auto take = [&](std::initializer_list<std::vector<int> *> order) {
        for (std::vector<int> *v : order)
            if (!v->empty()) { const int c = v->front(); v->erase(v->begin()); return c; }
        return -1;
    };
    int mismatched = 0;
    for (int pass = 0; pass < 3; ++pass)
        for (Tensor &t : p->tensors) {
            const int g = pass == 0 ? solo_group : pass == 1 ? 1 - solo_group : -1;
            if (t.group != g) continue;
            for (size_t j = 0; j < t.reserved / kChunk; ++j) {
                int c;
                if (pass == 0) { c = take({&pool_solo, &pool_any, &pool_other}); if (c >= 0 && p->chunks[c].klass != solo_class) mismatched++; }
                else if (pass == 1) { c = take({&pool_other, &pool_any, &pool_solo}); if (c >= 0 && (p->chunks[c].klass < 0 || p->chunks[c].klass == solo_class)) mismatched++; }
                else c = take({&pool_any, pool_solo.size() >= pool_other.size() ? &pool_solo : &pool_other, pool_solo.size() >= pool_other.size() ? &pool_other : &pool_solo});
                if (c < 0) return bail(pfail(p, MXV_ERR_HIP, "mxv_placed_alloc: ran out of chunks"));
                t.chunks.push_back(c);
            }
        }
    for (std::vector<int> *v : {&pool_solo, &pool_other, &pool_any})
        for (int c : *v) drop_chunk(p->chunks[c]);   // surplus
    // final mappings, every one at a fresh address
    for (int i = 0; i < count; ++i) {
        Tensor &t = p->tensors[i];
        hipError_t e = hipMemAddressReserve(reinterpret_cast<void **>(&t.va), t.reserved, 0, nullptr, 0);
        if (e != hipSuccess) return bail(pfail(p, MXV_ERR_HIP, "hipMemAddressReserve(%zu): %s", t.reserved, hipGetErrorString(e)));
        for (size_t j = 0; j < t.chunks.size(); ++j) {
            Chunk &c = p->chunks[t.chunks[j]];
            if (c.scratch_mapped) {
                (void)hipMemUnmap(c.scratch, kChunk);
                c.scratch_mapped = false;
            }
            if ((e = hipMemMap(t.va + j * kChunk, kChunk, 0, c.handle, 0)) != hipSuccess) {
                t.chunks.resize(j);
                return bail(pfail(p, MXV_ERR_HIP, "hipMemMap: %s", hipGetErrorString(e)));
            }
        }
        if ((e = hipMemSetAccess(t.va, t.reserved, &pr.acc, 1)) != hipSuccess) return bail(pfail(p, MXV_ERR_HIP, "hipMemSetAccess: %s", hipGetErrorString(e)));
        ptrs_out[i] = t.va;
    }
    if (hipError_t e = hipStreamSynchronize(pr.stream); e != hipSuccess) return bail(pfail(p, MXV_ERR_HIP, "%s", hipGetErrorString(e)));
    p->info.placed = 1;
    p->info.balanced = balanced && mismatched == 0;
    p->info.chunks_kept = needed;
    p->info.classes_seen = (int32_t)refs.size();
    for (int k = 0; k < 4; ++k) p->info.class_chunks[k] = have[k];
    p->info.stop_reason = stop_reason;
    p->info.solo_group = solo_group;
    p->info.solo_class = solo_class;
    p->info.same_class_us = t_same;
    p->info.different_class_us = t_diff;
    p->info.held_bytes = (size_t)needed * kChunk;
    p->info.peak_bytes = peak_bytes;
    p->info.jumped_bytes = jumped;
    p->info.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    *out = p;
    return MXV_OK;
}

}  // extern "C"
