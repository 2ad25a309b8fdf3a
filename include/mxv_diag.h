/* mxv_diag.h — OPTIONAL diagnostics: which kernel a launch took, what a box sustains for the kernels' store patterns, HBM-class probes and placed
 * memory.  Nothing a drop-in needs; benchmarks, tests and gym_amd/placement.py use them (API level 4).
 * Part of the C ABI of libmxv.so (see mxv.h for the engine's handle, status codes, RNG and numerical contracts).  Including mxv.h
 * includes this file — EXCEPT that this one is optional and must be included by itself. */
#ifndef MXV_DIAG_H
#define MXV_DIAG_H

#include "mxv.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Which kernel instantiation the handle's LAST step / rollout launch took — the library picks among ~150 (one launch per step or K
 * fused steps, envs per lane by shard size, guarded or unguarded trigonometry, folded or runtime physics parameters, the output-dtype
 * specialisations of the trajectory launch).  Lets a test or a benchmark state which code it measured (tests/test_gpu_soak.py,
 * bench.py config.launch_info).  kernel = -1 before the first launch; mxv_rollout_mixed does not update it. */
typedef struct mxv_launch_info {
    int32_t kernel;        /* 0 = step_kernel (one launch per step; also EAGER / GRAPH rollouts), 1 = rollout_kernel_v3 (FUSED) */
    int32_t env_id;
    int32_t param_mode;    /* 1 = default attributes folded into the code, 0 = common runtime values, 2 = per-env values */
    int32_t envs_per_lane;
    int32_t safe;          /* 1 = guarded sin / cos (state injected, unusual reset bounds, non-default attributes) */
    int32_t out_mode;      /* rollout_kernel_v3: 1 = trajectory outputs float64 rewards + int64 actions, 2 = float32 + int32, 0 = generic */
    int32_t tape;          /* actions supplied by the caller */
    int32_t steps;         /* K of the launch */
    uint32_t grid, block;
} mxv_launch_info;
int mxv_last_launch(mxv_handle *h, mxv_launch_info *out);


/* -- diagnostics ------------------------------------------------------------------------------------------------------------------
 * What this GPU sustains for the store pattern of the fused CartPole rollout with the physics removed (one wave per workgroup, two
 * envs per lane, XCD-aware tiles; obs float32 [K][N][4], reward float64 [K][N], actions int64 [K][N], two flag bytes [K][N]: 34 B
 * per env-step): microseconds per vector step, averaged over `launches` K-step launches into the caller's [K][num_envs] buffers
 * (contents destroyed).  MI355X boxes differ by 20 % on this pattern; bench.py prints the figure next to the
 * kernel's own time so that a number can be read against the box it was taken on.  Synchronises. */
int mxv_write_probe(int32_t device, int64_t num_envs, int32_t K, int32_t launches, float *obs_dev, double *reward_dev, int64_t *actions_dev,
                    uint8_t *terminated_dev, uint8_t *truncated_dev, double *us_per_step);

/* The same for any env kind and output dtypes (flags: MXV_FLAG_REWARD_F32 | MXV_FLAG_ACTION_I32): observation rows of that kind's width,
 * envs per lane as its fused rollout runs them, Box actions float32.  mxv_write_probe_env(MXV_CARTPOLE, 0, ...) is mxv_write_probe. */
int mxv_write_probe_env(int32_t device, int32_t env_id, int32_t flags, int64_t num_envs, int32_t K, int32_t launches, float *obs_dev,
                        void *reward_dev, void *actions_dev, uint8_t *terminated_dev, uint8_t *truncated_dev, double *us_per_step);

/* -- placed device memory for trajectory tensors ------------------------------------------------------------------------------------
 *    The fused rollout's outputs are a few long, parallel store streams.  On the MI355X a 16-byte-per-lane stream (observations) and
 *    an 8-byte-per-lane stream (rewards, actions) written concurrently run 10-12 % slower when the PHYSICAL memory behind them lies in
 *    the same CLASS of HBM regions — the classes are three contiguous thirds of the physical address space (3 x 96 GB: what the three
 *    ranks of a 12-high HBM3E stack would give; profiles/r3/r3c_hbm_class_map_whole_device.jsonl) —; a whole CartPole trajectory launch
 *    runs 5.4 / 5.7 / 6.4 us per 2^20-env step with none / one / both of {rewards, actions} in the observations' class (DESIGN.md §3,
 *    profiles/r3/r3a_*).  A fresh process is handed the first third for its first ~90 GiB, so hipMalloc'ed tensors all share a class unless
 *    earlier activity scrambled the driver's free lists — the "placement lottery" of rounds 1-2.  This call builds the tensors from
 *    256-MiB physical chunks (hipMemCreate) whose class it MEASURES (two concurrent streams against a reference chunk of each class
 *    seen so far) and maps chunks of one class under the tensors of one group and chunks of the other classes under the other group
 *    (group -1: whatever is left), each tensor contiguous in a fresh virtual range.
 *    Transient physical memory: chunks up to 2x the request; in addition, while only ONE class has been seen, unmapped spacer
 *    allocations (4 GiB each) that make the next chunk come from further along in physical memory — up to half of the device's free
 *    memory (at most 112 GiB; never into the last 16 GiB), released before the call returns; MXV_PLACED_NO_JUMP forbids the spacers
 *    (then a process that sits deep inside one class gets best effort: info.balanced = 0).  0.2-1.5 s.
 *    Sets below MXV_PLACED_MIN_BYTES (2 GiB: the real kernel runs 4-10 % slower on memory mapped through this API than on hipMalloc'ed
 *    memory, which the 8 % a 2^17-env shard of 1 GiB gains from separated classes does not win back — such sets are better served by
 *    ordinary allocations SORTED by class with mxv_hbm_pair_probe, what gym_amd/placement.py does from 1 GiB on), sets with an empty
 *    group and MXV_PLACED_PLAIN take ordinary hipMalloc allocations (info.placed = 0).
 *    bytes[i] > 0, group[i] in {-1, 0, 1}; ptrs_out[i] receives tensor i's device address (contents uninitialised).  mxv_placed_free
 *    releases the physical memory; the virtual ranges are NOT returned to the runtime (this runtime keeps stale translations for an
 *    address that is mapped a second time), i.e. every call consumes a little virtual address space for the life of the process. */
typedef struct mxv_placed mxv_placed;
#define MXV_PLACED_CHUNK_BYTES ((size_t)256 << 20)
#define MXV_PLACED_MIN_BYTES ((size_t)2 << 30)
enum { MXV_PLACED_PLAIN = 1, MXV_PLACED_NO_JUMP = 2 };
typedef struct mxv_placed_info {
    int32_t placed;            /* 1: chunks placed by class; 0: ordinary allocations */
    int32_t balanced;          /* 1: the two groups share no class */
    int32_t chunks_created;    /* physical chunks created (and classified) in total */
    int32_t chunks_kept;
    int32_t classes_seen;
    int32_t class_chunks[4];   /* chunks held of each class when the search ended (class 0 = the class of the first chunk) */
    int32_t solo_group;        /* the group that sits alone on class solo_class; the other group takes the other classes */
    int32_t solo_class;
    int32_t stop_reason;       /* why the search ended: 0 balanced, 1 chunk cap, 2 jump budget, 3 spacer allocation failed, 4 chunk allocation failed */
    double same_class_us;      /* the two-stream probe window, us per 2^20-lane step, both streams in one class ... */
    double different_class_us; /* ... and in different classes (0 if never seen) */
    double seconds;            /* wall time of the call */
    size_t requested_bytes, held_bytes, peak_bytes, jumped_bytes; /* peak: chunks + spacers at the worst moment; jumped: spacers */
} mxv_placed_info;
/* The measurement underneath: one 16-step window of two concurrent store streams of the rollout's launch shape (2^20 lanes), a 16-B/lane
 * stream over the 256 MiB at wide_dev and an 8-B/lane stream over the 128 MiB at narrow_dev (contents destroyed), us per step, best of
 * three timings of `launches` launches.  The same-class time of a box is ~4.2-4.4 us, a different-class pair runs at 0.89-0.91 of it:
 * compare against a pair known to share a class (two halves of one allocation), timed next to it. */
int mxv_hbm_pair_probe(int32_t device, void *wide_dev, void *narrow_dev, int32_t launches, double *us_per_step);
int mxv_placed_alloc(int32_t device, int32_t count, const size_t *bytes, const int32_t *group, int32_t flags, void **ptrs_out,
                     mxv_placed **out);
int mxv_placed_free(mxv_placed *p);
int mxv_placed_info_get(const mxv_placed *p, mxv_placed_info *out);
const char *mxv_placed_last_error(const mxv_placed *p); /* p may be NULL: last failed mxv_placed_alloc on this thread */

#ifdef __cplusplus
}
#endif
#endif /* MXV_DIAG_H */
