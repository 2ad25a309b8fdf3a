/* mxv_comm.h — collectives of a sharded vector env: RCCL all-gather of the final tensors (API level 3).
 * Part of the C ABI of libmxv.so (see mxv.h for the engine's handle, status codes, RNG and numerical contracts).  Including mxv.h
 * includes this file. */
#ifndef MXV_COMM_H
#define MXV_COMM_H

#include "mxv.h"

#ifdef __cplusplus
extern "C" {
#endif

/* -- collectives of a sharded vector env ------------------------------------------------------------------
 *    One logical vector env of N_total envs = `world` handles, one per GPU / process, rank r holding the contiguous global
 *    index range [r * N, (r+1) * N) (mxv_config.env_offset = r * N).  Env instances never interact (gym/vector/vector_env.py:
 *    13-16), so stepping needs no collective; the one exchange is the concatenation np.stack performs in the reference
 *    (gym/vector/sync_vector_env.py:159-169; AsyncVectorEnv gathers its workers' results the same way, async_vector_env.py:
 *    319-346): an all-gather of the shards' step outputs in rank order, here over RCCL / xGMI (librccl.so is opened with
 *    dlopen at mxv_comm_init: libmxv.so has no link-time dependency on it).
 *    Bootstrap like NCCL: one rank calls mxv_comm_unique_id and ships the MXV_COMM_ID_BYTES bytes to the others by any means
 *    (MPI, a file, torch.distributed's store ...); then every rank calls mxv_comm_init on its handle. ----------------------- */
#define MXV_COMM_ID_BYTES 128
int mxv_comm_unique_id(void *id_out);
int mxv_comm_init(mxv_handle *h, int32_t rank, int32_t world, const void *unique_id);
int mxv_comm_destroy(mxv_handle *h);
/* Asynchronous all-gather of this shard's outputs (obs float32 [N][O], reward in the handle's reward dtype [N], terminated /
 * truncated uint8 [N]; device pointers, e.g. the last slices of a rollout chunk's trajectory tensors) into [world][...]
 * device buffers = the full (N_total, ...) tensors in global env order.  The four gathers are issued as ONE grouped RCCL
 * launch on the communicator's own high-priority stream, ordered after everything launched so far on the handle's stream;
 * the call returns immediately and later launches on the handle's stream (the next rollout chunk) overlap it.  Any
 * send/receive pair may be NULL (skipped).  The send buffers must stay untouched until the gather has completed. */
int mxv_allgather_outputs(mxv_handle *h, const float *obs_dev, const void *reward_dev, const uint8_t *terminated_dev,
                          const uint8_t *truncated_dev, float *all_obs_dev, void *all_reward_dev, uint8_t *all_terminated_dev,
                          uint8_t *all_truncated_dev);
/* Wait for the last gather (age 0) or the one before it (age 1: what a caller that alternates between two snapshot buffers
 * needs before it lets a rollout overwrite the older one — the younger gather keeps overlapping).  host_sync == 0: the handle's
 * stream waits on the GPU, the host does not block; host_sync != 0: block the host until those gathered tensors are complete. */
int mxv_allgather_wait(mxv_handle *h, int32_t age, int32_t host_sync);
/* the hipStream_t the gathers run on (NULL before mxv_comm_init) */
int mxv_comm_stream(mxv_handle *h, void **stream);

#ifdef __cplusplus
}
#endif
#endif /* MXV_COMM_H */
