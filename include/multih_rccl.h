// multih_rccl.h — C ABI of libmultih_rccl.so (multi-h_amd/host/rccl_transport.cpp): the multi-GPU transport of the propose stage, native: RCCL's ncclAllGather enqueued on the engine's
// HIP stream ("one RCCL ncclAllGather of int32[M/G] inlier counts per rank over xGMI").  No Python, no
// host synchronisation: `mhr_allgather` has the signature mh_set_transport / MultiH::SetShardingStream expect for a
// stream-ordered transport, with the communicator as its context.  The reference has no counterpart (single process,
// SURVEY 2.2).  Built as libmultih_rccl.so (links librccl); the engine and the host class do not depend on it.
//
// Bootstrap (one process per GPU): rank 0 makes the 128-byte unique id (mhr_unique_id) and hands it to the other ranks by
// whatever channel the launcher has — multih_harness --ranks N passes it through pipes made before it forks,
// bench.py broadcasts it over its torch.distributed process group — then every rank calls mhr_init.
#pragma once
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mhr_comm mhr_comm;
#define MHR_ID_BYTES 128

__attribute__((visibility("default"))) int mhr_unique_id(unsigned char id[MHR_ID_BYTES]);
// Collective over the `world` ranks: every rank calls it with the same id.  `device` = the HIP device of this rank.
__attribute__((visibility("default"))) int mhr_init(mhr_comm** out, int rank, int world, const unsigned char id[MHR_ID_BYTES], int device);
// The same with the id travelling through `path`: rank 0 creates the file (atomically: an exclusively created temporary
// renamed over it), the others wait up to timeout_s for it.  The caller owns the path: it must not exist beforehand (a stale
// file of an earlier run would be read as this run's id) and is the caller's to remove.  multih_harness --ranks N hands the id
// to its children through pipes instead.
__attribute__((visibility("default"))) int mhr_init_from_file(mhr_comm** out, int rank, int world, const char* path, int device, int timeout_s);
// mh_allgather_stream_fn: all-gather `bytes_per_rank` bytes per rank, device buffers, rank order, on `hip_stream`.
__attribute__((visibility("default"))) int mhr_allgather(void* comm, const void* send_dev, void* recv_dev, unsigned long long bytes_per_rank, void* hip_stream);
__attribute__((visibility("default"))) void mhr_destroy(mhr_comm* c);
__attribute__((visibility("default"))) const char* mhr_last_error(void);
__attribute__((visibility("default"))) long long mhr_calls(const mhr_comm* c);      // collectives enqueued so far (tests, logs)
// r06: what RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank) — read back into every bench line
// and harness log, so that a multi-GPU number states how many ranks the collective really spanned — and the library's
// version code (ncclGetVersion: major * 10000 + minor * 100 + patch).  -1 on failure (mhr_last_error).
__attribute__((visibility("default"))) int mhr_count(const mhr_comm* c);
__attribute__((visibility("default"))) int mhr_rank(const mhr_comm* c);
__attribute__((visibility("default"))) int mhr_version(void);

#ifdef __cplusplus
}
#endif
