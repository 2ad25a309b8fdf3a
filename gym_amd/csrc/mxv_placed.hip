// mxv_placed.hip — device memory for trajectory tensors, placed by physical memory class (include/mxv.h: mxv_placed_*).
//
// Why.  The fused rollout writes every step's outputs as a few long, parallel store streams (CartPole: observations 16 B per env and
// step, rewards 8 B, actions 8 B, two flag bytes).  On the MI355X a 16-B/lane stream and an 8-B/lane stream written concurrently run
// 10-12 % slower when the PHYSICAL memory behind them belongs to the same one of two classes of HBM regions — GiB-scale runs of
// irregular length in physical allocation order — and the whole trajectory launch runs 5.4 / 5.7 / 6.4 us per 2^20-env step when
// none / one / both of {rewards, actions} share the observations' class (tools/vmm_probe5.hip, vmm_probe7.hip; profiles/r3a_*).
// That is the "placement lottery" of DESIGN.md §6: hipMalloc'ed tensors land wherever the allocator is, all in one run more often
// than not.  Nothing in software sees the class of a page — but HIP's virtual-memory API decides which physical memory backs which
// virtual range, and the class of a chunk can be MEASURED: two concurrent streams, one into the chunk, one into a reference chunk.
//
// How.  Physical memory is created in 256-MiB chunks (hipMemCreate); every chunk is mapped at a scratch address of its own and
// classified against a reference chunk of each class; chunks are created until both classes can serve their groups of tensors
// (surplus chunks are released: transient memory stays below 2x the request); then every tensor gets a fresh virtual range with
// chunks of its group's class mapped into it.  Group 0 and group 1 tensors never share a class; group -1 tensors take what is left.
//
// Runtime facts this code is written around (ROCm 7.0 / this driver; tools/_bin/vmmdbg2.hip is the test):
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
#include <new>
#include <string>
#include <vector>

#include "../../include/mxv.h"

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
        for (size_t j = 0; j < t.chunks.size(); ++j) (void)hipMemUnmap(t.va + j * kChunk, kChunk);   // exactly as mapped; the range itself is kept (see top)
    }
    for (Chunk &c : p->chunks) drop_chunk(c);
    delete p;
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
    // Transient physical memory: 2x the request — more only while that is still a small part of what the device has free (a run of one
    // class can be 15 GiB long and cannot be crossed without holding it: the driver hands released blocks straight out again), never
    // beyond 6x; MXV_PLACED_WIDE_SEARCH allows 6x outright.
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    int cap = 2 * needed;
    const int roomy = (int)std::min<size_t>((size_t)6 * needed, free_b / 10 / kChunk);
    cap = std::max(cap, (flags & MXV_PLACED_WIDE_SEARCH) ? 6 * needed : roomy);
#define PL_TIME(w, n, launches, reps, out)                                                                                      \
    do {                                                                                                                        \
        if (hipError_t e_ = pr.time_pair(p->chunks[w], p->chunks[n], launches, reps, out); e_ != hipSuccess)                     \
            return bail(pfail(p, MXV_ERR_HIP, "classification probe: %s", hipGetErrorString(e_)));                               \
    } while (0)
    // Bootstrap on the first three chunks, timed pairwise.  With two classes at least two of any three chunks share one, so the slowest
    // pair is a same-class pair; if the fastest pair is clearly faster, its odd member belongs to the other class.  The device is
    // spun up first and the three timings are repeated until two rounds agree (a cold clock would stretch the early ones).
    for (int i = 0; i < 3; ++i)
        if (int rc = new_chunk(p, pr)) return bail(rc);
    const float kDiffRatio = 0.955f;   // measured: a different-class pair runs at 0.89-0.91 of the same-class time
    const float kMargin = 1.03f;
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
    int ref0 = 0, twin0 = 1, ref1 = -1;   // ref0, twin0: two chunks of class 0 (by definition); ref1: a chunk of class 1 once one is known
    {
        const float hi = std::max(t01, std::max(t02, t12)), lo = std::min(t01, std::min(t02, t12));
        if (lo > kDiffRatio * hi) {   // all three alike
            p->chunks[0].klass = p->chunks[1].klass = p->chunks[2].klass = 0;
        } else {                      // the slowest pair shares a class, the third chunk has the other
            const int odd = hi == t01 ? 2 : hi == t02 ? 1 : 0;
            ref0 = odd == 0 ? 1 : 0;
            twin0 = odd == 2 ? 1 : 2;
            ref1 = odd;
            for (int c = 0; c < 3; ++c) p->chunks[c].klass = c == odd ? 1 : 0;
        }
    }
    float t_same = std::max(t01, std::max(t02, t12)), t_diff = ref1 >= 0 ? std::min(t01, std::min(t02, t12)) : 0.f;
    auto classify = [&](int c) -> int {   // sets chunks[c].klass (0, 1, or -1 = could not tell); MXV_OK or an error
        Chunk &ch = p->chunks[c];
        for (int attempt = 0; attempt < 3; ++attempt) {
            float a = 0.f, b = 0.f;
            if (ref1 >= 0) {   // a reference of each class, timed back to back: the slower pairing names the class — no absolute threshold
                PL_TIME(c, ref0, 4, 2, &a);
                PL_TIME(c, ref1, 4, 2, &b);
                if (a > kMargin * b) { ch.klass = 0; t_same = 0.9f * t_same + 0.1f * a; t_diff = 0.9f * t_diff + 0.1f * b; return MXV_OK; }
                if (b > kMargin * a) { ch.klass = 1; t_same = 0.9f * t_same + 0.1f * b; t_diff = 0.9f * t_diff + 0.1f * a; return MXV_OK; }
            } else {           // only one class seen so far: against a known same-class pair timed right next to it
                PL_TIME(c, ref0, 4, 2, &a);
                PL_TIME(twin0, ref0, 4, 2, &b);
                if (a > kDiffRatio * b) { ch.klass = 0; t_same = b; return MXV_OK; }
                float a2 = 0.f, b2 = 0.f;   // looks like the other class: confirm before it becomes the reference
                PL_TIME(c, ref0, 4, 3, &a2);
                PL_TIME(twin0, ref0, 4, 3, &b2);
                if (a2 < kDiffRatio * b2) { ch.klass = 1; ref1 = c; t_same = b2; t_diff = a2; return MXV_OK; }
            }
        }
        ch.klass = -1;   // e.g. a chunk that straddles a class boundary: only fit for tensors that do not care
        return MXV_OK;
    };
    // create until one class covers group 0 and the other group 1 (either way round) and the rest covers group -1
    int have[3] = {0, 0, 0};   // class 0, class 1, undecided
    for (int c = 0; c < 3; ++c) have[p->chunks[c].klass]++;
    auto enough = [&](int *cls_of_group0) {
        for (int a = 0; a < 2; ++a)
            if (have[a] >= need[0] && have[1 - a] >= need[1] && have[0] + have[1] + have[2] >= needed) { *cls_of_group0 = a; return true; }
        return false;
    };
    int cls0 = 0;
    int peak_live = 3;
    while (!enough(&cls0)) {
        if (have[0] + have[1] + have[2] >= cap) break;   // best effort below; the report says so (balanced = 0)
        const size_t before = p->chunks.size();
        if (new_chunk(p, pr) != MXV_OK) {   // out of device memory: best effort with what there is
            if (p->chunks.size() > before) drop_chunk(p->chunks.back());
            break;
        }
        const int c = (int)p->chunks.size() - 1;
        if (int rc = classify(c)) return bail(rc);
        have[p->chunks[c].klass < 0 ? 2 : p->chunks[c].klass]++;
        peak_live = std::max(peak_live, have[0] + have[1] + have[2]);
    }
    const bool balanced = enough(&cls0);
    if (!balanced) cls0 = have[0] >= have[1] ? (need[0] >= need[1] ? 0 : 1) : (need[0] >= need[1] ? 1 : 0);   // the larger group gets the larger class
    if (have[0] + have[1] + have[2] < needed) return bail(pfail(p, MXV_ERR_HIP, "mxv_placed_alloc: %d chunks of %zu MiB needed, %d available (out of device memory?)", needed, kChunk >> 20, have[0] + have[1] + have[2]));

    // hand out: group 0 from class cls0, group 1 from the other, group -1 from whatever is left (then shortfalls from anything)
    std::vector<int> by_class[3];   // class 0, class 1, undecided
    for (int c = 0; c < (int)p->chunks.size(); ++c)
        if (p->chunks[c].klass >= -1) by_class[p->chunks[c].klass < 0 ? 2 : p->chunks[c].klass].push_back(c);
    auto take = [&](int want) {   // want: 0 / 1, or 2 = anything, least useful first
        const int order[3][3] = {{0, 2, 1}, {1, 2, 0}, {2, by_class[0].size() >= by_class[1].size() ? 0 : 1, by_class[0].size() >= by_class[1].size() ? 1 : 0}};
        for (int k : order[want])
            if (!by_class[k].empty()) {
                const int c = by_class[k].front();
                by_class[k].erase(by_class[k].begin());
                return c;
            }
        return -1;
    };
    int mismatched = 0;
    for (int g : {0, 1, -1})
        for (Tensor &t : p->tensors) {
            if (t.group != g) continue;
            for (size_t j = 0; j < t.reserved / kChunk; ++j) {
                const int want = g == 0 ? cls0 : g == 1 ? 1 - cls0 : 2;
                const int c = take(want);
                if (c < 0) return bail(pfail(p, MXV_ERR_HIP, "mxv_placed_alloc: ran out of chunks"));
                if (g >= 0 && p->chunks[c].klass != want) mismatched++;
                t.chunks.push_back(c);
            }
        }
    for (int k = 0; k < 3; ++k)
        for (int c : by_class[k]) drop_chunk(p->chunks[c]);   // surplus
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
    p->info.class_chunks[0] = have[0];
    p->info.class_chunks[1] = have[1];
    p->info.group0_class = cls0;
    p->info.same_class_us = t_same;
    p->info.different_class_us = t_diff;
    p->info.held_bytes = (size_t)needed * kChunk;
    p->info.peak_bytes = (size_t)peak_live * kChunk;
    p->info.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    *out = p;
    return MXV_OK;
}

}  // extern "C"
