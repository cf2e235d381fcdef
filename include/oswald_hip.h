/*
 * include/oswald_hip.h -- the drop-in boundary of the MI355X search path.
 *
 * C ABI of liboswald_hip.so: plain pointers and sizes, no C++ or torch types.
 * It stands where OSWALD's host code talks to its accelerator: the Altera
 * OpenCL bring-up in host/src/utils.c:99-191 and the enqueue path of
 * fpga_search() in host/src/FPGAsearch.c:82-238 (the same sequence appears in
 * the "FPGA thread" of host/src/HybridSearch.c:133-228, :640-754).  The
 * reference has no formal plugin API; every entry point below names the
 * reference call sequence it replaces.  INTEGRATION.md shows the binding a
 * maintainer of the reference would write.
 *
 * Conventions
 *   - every function returns 0 on success and a negative OSWALD_HIP_E* code on
 *     failure; oswald_hip_last_error() returns the message of the calling
 *     thread's last failure.  (The reference prints and exit()s on every
 *     error, utils.c:256-262; a caller that wants that behaviour does it on a
 *     non-zero return.)
 *   - host buffers stay owned by the caller; the layer copies what it needs
 *     before the call returns unless the name says _async, in which case the
 *     buffers must stay valid until oswald_hip_wait().
 *   - one caller thread at a time per context DEVICE; any thread may be that caller.  The per-device calls (chunk
 *     upload / set_index / search / release / topr, wait on one device, stats) touch only their device's state and may
 *     be made concurrently for different devices of a context -- the CLI drives N GPUs from N threads, so that no
 *     device waits for another one's search to be planned.  Context-level calls (init, finalize, set_scoring,
 *     set_queries, topr_begin, topr, comm_*, wait on all devices) need every other caller out of the context.
 *   - database residues use the preprocessed alphabet (0..22, dummy 23,
 *     reference host/src/sequences.c:165-175, sequences.h:16-17) and the
 *     reference's interleaved group layout b[disp[g] + j*W + lane]
 *     (host/src/sequences.c:479-498); scores come back as int32
 *     [nq][ngroups*W], the layout of the reference's score table
 *     (host/src/FPGAsearch.c:236-237).
 */
#ifndef OSWALD_HIP_H
#define OSWALD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSWALD_HIP_OK 0
#define OSWALD_HIP_EINVAL (-1)   /* bad argument */
#define OSWALD_HIP_ENODEV (-2)   /* no usable GPU / device index out of range */
#define OSWALD_HIP_ERUNTIME (-3) /* HIP runtime call failed */
#define OSWALD_HIP_ENOMEM (-4)   /* host or device allocation failed */
#define OSWALD_HIP_ESTATE (-5)   /* call sequence violated (e.g. search before set_queries) */
#define OSWALD_HIP_ECOMM (-6)    /* an RCCL call failed (multi-GPU top-r gather) */

#define OSWALD_HIP_ABI_VERSION 5

typedef struct oswald_hip_ctx oswald_hip_ctx;

/* ABI version of the loaded library (compare with OSWALD_HIP_ABI_VERSION). */
int oswald_hip_abi_version(void);

/* Message of the calling thread's last failed call ("" if none). */
const char *oswald_hip_last_error(void);

/* Number of visible GPUs.  Replaces getDevices() in utils.c:115-118. */
int oswald_hip_device_count(int *count);

/* Page-locked host memory for the buffers the caller hands to oswald_hip_chunk_upload* and the score tables it has
 * filled: the DMA engines read / write it directly and asynchronously (pageable memory works too; the runtime then
 * stages it).  Stands where the reference allocates its host buffers 64-byte aligned "for DMA":
 * posix_memalign(AOCL_ALIGNMENT, ...), host/src/sequences.h:15, host/src/FPGAsearch.c:69-74.  Needs a GPU. */
int oswald_hip_host_alloc(size_t bytes, void **ptr);
int oswald_hip_host_free(void *ptr);

/* The same for memory the caller already has: page-locks [ptr, ptr + bytes) in place (hipHostRegister), so that uploads
 * from it are plain asynchronous DMA at the link's rate instead of copies the runtime stages through its own buffers on the
 * caller's thread (measured on MI355X / PCIe Gen5: ~18 GB/s staged, ~50 GB/s direct).  Meant for a database the caller has
 * read or mapped as a whole -- the CLI registers its group cache chunk by chunk on a helper thread while the devices are
 * brought up -- where the reference allocates every chunk buffer 64-byte aligned "for DMA" before its timed region starts
 * (posix_memalign(AOCL_ALIGNMENT, ...), host/src/sequences.c:470-476, :540-560).  A range is registered once and
 * unregistered (by its start address) before it is freed or unmapped; both calls need a GPU and may be made from any
 * thread, also while searches are running.  Read-only mappings cannot be registered on every kernel: on failure the
 * memory simply stays pageable (uploads from it still work) and the call says why. */
int oswald_hip_host_register(void *ptr, size_t bytes);
int oswald_hip_host_unregister(void *ptr);

/* Bring-up: one stream and one buffer set per device; device_ids == NULL means
 * devices 0..ndev-1.  Replaces init(), utils.c:99-173 (platform, context,
 * queues, program, kernels).  Nothing is retained after oswald_hip_finalize(). */
int oswald_hip_init(int ndev, const int *device_ids, oswald_hip_ctx **ctx);

/* Releases every device and host resource.  Replaces cleanup(), utils.c:176-191
 * and the clReleaseMemObject block of FPGAsearch.c:361-368. */
int oswald_hip_finalize(oswald_hip_ctx *ctx);

/* Human-readable device description for `-O info` (utils.c:216-253), written
 * NUL-terminated into buf (truncated to buflen). */
int oswald_hip_info(oswald_hip_ctx *ctx, int dev, char *buf, size_t buflen);

/* Scoring system.  submat is the reference's 24 x 32 int8 table
 * (host/src/submat.c); open_gap / extend_gap as on the command line (a gap of
 * length L costs open + L*extend).  Replaces the static kernel arguments 3 and
 * 4 of FPGAsearch.c:101-109 and the score-profile build of :143-177 (done on
 * the device here).  cell_bits selects the cell arithmetic of the first pass:
 * 16 (packed int16, the default; also selected by 0: exact below 22256), 32
 * (plain int32) or 8 (four 7-bit cells per register, the reference's int8
 * first pass sw.cl:60-78; slower than 16 on this GPU, which has no packed
 * 8-bit maximum).  Results are exact in every mode: what leaves the range of
 * the 8-bit cells is re-run in int16, what reaches the ceiling of the int16
 * cells in int32, on the device (the reference escalates int8 -> int16 ->
 * int32 on the host, HybridSearch.c:1670-1680,:1774-1784).  With cell_bits 8
 * a matrix whose entries span more than 127, or a gap penalty above 127,
 * makes the search run on the int16 cells alone.
 * Column 23 of the table is the dummy residue's and columns 24..31 are
 * padding; in the reference's matrices all nine hold zeros, and its
 * preprocessing emits the residue codes 0..23 only (sequences.c:60-116).  The
 * library treats a database residue code of 24..31 as code 23, and refuses
 * (OSWALD_HIP_EINVAL) a table whose columns 24..31 do not repeat column 23 --
 * the one case in which that would change a score. */
int oswald_hip_set_scoring(oswald_hip_ctx *ctx, const int8_t *submat, int open_gap, int extend_gap, int cell_bits);

/* Query set: residues of all queries back to back (codes 0..23), lengths m[],
 * offsets a_disp[] (nq entries used).  Replaces cl_a / kernel arguments 0-2,
 * FPGAsearch.c:85, :206-210.  Queries may be given in any order. */
int oswald_hip_set_queries(oswald_hip_ctx *ctx, const uint8_t *a, uint64_t Q, const uint16_t *m,
                           const uint32_t *a_disp, uint32_t nq);

/* One database chunk made resident on device `dev`: uploads the interleaved
 * groups and re-tiles them for the kernels.  b: vD bytes; n[g]: (padded) group
 * lengths; disp[g]: byte offset of group g in b; lane_width W: 16 (the
 * reference's device layout), 32 (its AVX2 layout), 64 or 128.  Replaces the four clEnqueueWriteBuffer + clFinish of
 * FPGAsearch.c:180-198, minus the 23x score profile.  *chunk receives a handle
 * valid until oswald_hip_chunk_release() or finalize. */
int oswald_hip_chunk_upload(oswald_hip_ctx *ctx, int dev, const uint8_t *b, uint64_t vD, const uint16_t *n,
                            const uint32_t *disp, uint32_t ngroups, uint32_t lane_width, int *chunk);

/* The same, without waiting for the device: returns once the copies and the re-tile kernel are queued on the
 * device's UPLOAD stream -- uploads have a stream of their own, so chunk k+1 comes in while chunk k is being
 * searched: queue search k, then upload k+1 (the reference uploads and searches strictly in turn,
 * FPGAsearch.c:180-223).  b must stay valid until the chunk has been released (n and disp are copied before the call returns)
 * (oswald_hip_chunk_release waits for the upload; oswald_hip_chunk_search does NOT -- it queues the search behind the
 * upload on the device) or oswald_hip_wait() has returned.  With several
 * devices this is also what lets their uploads overlap: queue all of them, then search each (the reference's four clEnqueueWriteBuffer per device
 * are asynchronous too and share one clFinish per device, FPGAsearch.c:180-198).
 * A large chunk that arrives while its device has nothing to do -- the first chunk of a search -- is cut by the library into
 * a HEAD of whole 128-sequence blocks and the REST (two copies, two re-tiles, and later two searches behind one another):
 * the device starts on the head while the rest is still on the link, instead of waiting for all of it (2.3 ms per 128 MiB).
 * Nothing of this shows at the boundary: one handle, one score table, one index. */
int oswald_hip_chunk_upload_async(oswald_hip_ctx *ctx, int dev, const uint8_t *b, uint64_t vD, const uint16_t *n,
                                  const uint32_t *disp, uint32_t ngroups, uint32_t lane_width, int *chunk);

/* The device's own working memory (the per-wave spill scratch of the long-query rounds) is made by oswald_hip_init since ABI
 * version 5 -- it belongs to the kernels, not to a search: see there.  This call remains for callers of earlier versions: it makes
 * sure the scratch holds blocks of max_sequence_length residues (it does) and returns. */
int oswald_hip_reserve(oswald_hip_ctx *ctx, int dev, uint32_t max_sequence_length);

/* Optional: the buffers of `slots` chunk slots for chunks of up to chunk_bytes bytes in ngroups groups of lane_width sequences, on
 * device dev (dev < 0: all), made in ONE place instead of by the first uploads and searches as they come.  Two kinds, as in the
 * reference: the slots' page-locked HOST staging (work queues, live extents, block tables, copies of n[] / disp[]) -- the
 * reference allocates its host buffers, aligned for DMA, BEFORE its clock starts (posix_memalign, FPGAsearch.c:69-74; tick at
 * :80) -- and their DEVICE buffers (staging copy, re-tiled residues, block tables, and -- with nq > 0 -- score table and re-run
 * queue for a set of nq queries) -- the reference creates its six device buffers, sized for the largest chunk, once per search
 * INSIDE its timed region and re-uses them for every chunk (clCreateBuffer, FPGAsearch.c:85-96).
 * oswald_hip_reserve_host makes the former only; oswald_hip_reserve_chunks makes whatever of both is not there yet.  A caller that
 * mirrors the reference's clock (the command-line tool's -m 0, bench.py's `inclusive` leg) calls _reserve_host before its clock
 * and _reserve_chunks inside.  Hints only: a chunk that needs more grows its slot as before. */
int oswald_hip_reserve_host(oswald_hip_ctx *ctx, int dev, uint32_t ngroups, uint32_t lane_width, uint32_t nq, uint32_t slots);
int oswald_hip_reserve_chunks(oswald_hip_ctx *ctx, int dev, uint64_t chunk_bytes, uint32_t ngroups, uint32_t lane_width,
                              uint32_t nq, uint32_t slots);

/* ... and the reverse of the device part: the DEVICE buffers of every chunk slot of device dev (dev < 0: all) that holds no chunk
 * are given back (the reference releases its cl_mem objects at the end of a search, FPGAsearch.c:361-368; its host buffers live
 * on).  Waits for the device. */
int oswald_hip_release_chunks(oswald_hip_ctx *ctx, int dev);

/* The largest chunk -- in bytes of b, i.e. padded residues, the unit of the command line's -k -- device dev can hold for
 * a set of nq queries and sequences of up to max_sequence_length residues: 0.8 of its free memory (less the work
 * space still to be allocated) over what a byte of chunk costs at worst (three resident chunks per device -- one
 * searched while the next two come in --, each with its staging copy, re-tiled residues, scores and re-run queues).  Replaces the clamp of
 * max_chunk_size to the device's global memory in init(), utils.c:162-168 (0.8 x memory / 23 score-profile rows). */
int oswald_hip_max_chunk_size(oswald_hip_ctx *ctx, int dev, uint32_t nq, uint32_t max_sequence_length, uint64_t *bytes);

/* All queries against a chunk, asynchronously on the device's stream.  The chunk's upload need not have landed: the
 * search is queued behind it on the device and the call returns at once (its work queues are then planned on the group
 * lengths n[] instead of the live extents the upload brings back; the scores do not depend on the plan).
 * If scores_out != NULL the int32 scores [nq][ngroups*W] are copied there
 * (valid after oswald_hip_wait).  Replaces the per-query clSetKernelArg +
 * clEnqueueNDRangeKernel loop, clWaitForEvents, clEnqueueReadBuffer and the
 * host-side overflow re-computation, FPGAsearch.c:204-274. */
int oswald_hip_chunk_search(oswald_hip_ctx *ctx, int dev, int chunk, int32_t *scores_out);

/* All queries against SEVERAL chunks that are resident on device dev, as ONE launch (ABI 5, round 6).  For a database that STAYS on
 * the device -- a service that answers query sets against it; bench.py's resident steps -- : every launch boundary costs a ramp, a ragged
 * end and, for short launches, clock (one launch instead of three: 7 % for one query against 1 M sequences, 1.2 % for twenty).  The
 * chunks' uploads must have landed (the call waits for them); their order is the order of the score columns and of the folds into the
 * top lists (oswald_hip_topr_begin / _chunk_set_index as for oswald_hip_chunk_search).  scores_out: null, or int32 [nq][sum of the
 * chunks' ngroups * lane_width], valid after oswald_hip_wait.  A caller that streams chunks in (the reference's loop, FPGAsearch.c:132-238)
 * searches them one by one with oswald_hip_chunk_search, each behind its upload; the reference has no resident mode. */
int oswald_hip_search_resident(oswald_hip_ctx *ctx, int dev, const int *chunks, uint32_t nchunks, int32_t *scores_out);

/* Gives the chunk's slot back.  Returns once the chunk's upload has landed (the caller's b / n / disp are free); a
 * search of the chunk may still be running -- the next upload into the slot waits for it on the device. */
int oswald_hip_chunk_release(oswald_hip_ctx *ctx, int dev, int chunk);

/* Convenience: upload + search + release in one ASYNCHRONOUS call, the exact per-chunk step of FPGAsearch.c:132-238:
 * the copies go out on the device's copy stream, the search is queued behind them on the device, and the call returns
 * without waiting for either (the reference's four clEnqueueWriteBuffer per device are non-blocking too and share one
 * clFinish, FPGAsearch.c:180-198) -- with several devices, call it for each of them in turn and their uploads overlap.
 * b / n / disp and scores_out must stay valid until oswald_hip_wait() has returned for the device; the chunk's slot is
 * given back then. */
int oswald_hip_search_chunk_async(oswald_hip_ctx *ctx, int dev, const uint8_t *b, uint64_t vD, const uint16_t *n,
                                  const uint32_t *disp, uint32_t ngroups, uint32_t lane_width, int32_t *scores_out);

/* Blocks until the LAST SEARCH queued for this chunk is through (its upload, its kernels, its top list; the download of its score
 * table, if one was asked for) -- whatever else has been queued on the device behind it meanwhile.  Replaces the
 * clWaitForEvents on one device's kernel events, FPGAsearch.c:223, for a caller that keeps the device's queue filled ahead. */
int oswald_hip_chunk_wait(oswald_hip_ctx *ctx, int dev, int chunk);

/* Blocks until everything queued on `dev` has finished (dev < 0: all devices).
 * Replaces clFinish / clWaitForEvents, FPGAsearch.c:197, :223, :279. */
int oswald_hip_wait(oswald_hip_ctx *ctx, int dev);

/* Top-r selection on the device over the scores of the chunk's last search:
 * for every query the r best (score, index-in-chunk) pairs, descending score,
 * ties by DESCENDING index -- the order sort_scores() produces
 * (host/src/utils.c:3-86).  nvalid = number of real sequences in the chunk
 * (padding lanes beyond it are ignored).  scores/index: [nq][r], host memory.
 * Replaces sort_scores() + the top-r print loop, FPGAsearch.c:312-321, for the
 * per-GPU part of the multi-GPU merge. */
int oswald_hip_chunk_topr(oswald_hip_ctx *ctx, int dev, int chunk, uint32_t nvalid, uint32_t r,
                          int32_t *scores, uint32_t *index);

/* Context-level top-r: the r best (score, database index) pairs per query over EVERY chunk searched on EVERY
 * device of the context -- and, with a process-level communicator (below), of every process of the job -- in the
 * order sort_scores() produces (descending score, equal scores by DESCENDING database index,
 * host/src/utils.c:3-86).  Replaces the merge of the devices' score tables into the global one
 * (host/src/FPGAsearch.c:236-237), sort_scores() (utils.c:71-86) and the top-r loop (FPGAsearch.c:312-321).  This is
 * the multi-GPU gather of the path, and it runs on the GPUs: every device keeps a running list of r tagged keys per
 * query, a chunk's r best are selected and folded into it on the device right behind the search, the GPUs of the
 * context all-gather their lists over RCCL/xGMI (ncclCommInitAll over the context's distinct GPUs at bring-up; a
 * context on one GPU needs no communicator), GPU 0 folds them, and nq x r pairs come to the host in one copy.  Use:
 *   oswald_hip_topr_begin(ctx, r)       start collecting (drops what was collected before); r <= 1024; after
 *                                       oswald_hip_set_queries (a new query set needs a new _begin);
 *   oswald_hip_chunk_set_index(...)     after an upload: the chunk's place in the database -- sequence k of the chunk
 *                                       is database sequence first_index + k, or index_map[k] if a map is given (it is
 *                                       copied, in any order; a chunk need not be one contiguous run); nvalid real sequences;
 *   oswald_hip_chunk_search(...)        of a chunk that has its index now also selects the chunk's r best on its
 *                                       device, queued on the device's stream behind the search: nothing waits, and the
 *                                       chunk may be released (its slot re-used) right after the call.  (A chunk
 *                                       searched twice counts once: lists hold distinct database keys.)
 *   oswald_hip_topr(ctx, r, ...)        gathers, waits for all devices and returns everything collected since _begin
 *                                       (r <= the r given to _begin).  scores / db_index: [nq][r], host memory; slots
 *                                       beyond the number of database sequences seen: score -1, index 0xffffffff.
 * oswald_hip_merge_candidates is the same order as host logic (ncand candidates per query, [nq][ncand]; score < 0 =
 * empty slot); it works without a GPU.  Nothing in the library falls back to it: it is there for callers that hold
 * lists of their own, and as the checker of the tests. */
int oswald_hip_topr_begin(oswald_hip_ctx *ctx, uint32_t r);
int oswald_hip_chunk_set_index(oswald_hip_ctx *ctx, int dev, int chunk, uint32_t first_index, uint32_t nvalid,
                               const uint32_t *index_map);
int oswald_hip_topr(oswald_hip_ctx *ctx, uint32_t r, int32_t *scores, uint32_t *db_index);
int oswald_hip_merge_candidates(uint32_t nq, uint64_t ncand, const int32_t *cand_scores, const uint32_t *cand_index,
                                uint32_t r, int32_t *scores, uint32_t *db_index);

/* Process-level communicator: one rank per process (one process per GPU, or per group of GPUs).  With it
 * oswald_hip_topr all-gathers the contexts' lists between the processes over RCCL (from context device 0) and every
 * rank receives the list of the whole job; the call is then COLLECTIVE -- every rank must make it, with the same r and
 * the same number of queries.  Rank 0 calls oswald_hip_comm_unique_id and hands the id (OSWALD_HIP_COMM_ID_BYTES
 * bytes) to the other ranks by any means it has (MPI, a file, torch.distributed ...); then every rank calls
 * oswald_hip_comm_init_rank (collective; wraps ncclCommInitRank on context device 0).  The reference has one process
 * and one host merge (FPGAsearch.c:236-237); this is what stands there for a job of several processes.
 * oswald_hip_comm_info: out[0] = ranks of the in-context communicator (distinct GPUs of the context, as RCCL
 * reports them; 1 = none needed), out[1] = ranks of the process-level communicator as RCCL reports them (0 = none),
 * out[2] = this process's rank in it (-1 = none), out[3] = RCCL version code. */
#define OSWALD_HIP_COMM_ID_BYTES 128
int oswald_hip_comm_unique_id(void *id, size_t id_bytes);
int oswald_hip_comm_init_rank(oswald_hip_ctx *ctx, const void *id, size_t id_bytes, int nranks, int rank);
int oswald_hip_comm_info(oswald_hip_ctx *ctx, int *out4);
/* Gives the process-level communicator up again (ncclCommAbort: no rank waits for another): oswald_hip_topr is a local
 * call afterwards.  For a job whose ranks could not ALL join -- the ranks that did must not stay in a communicator the
 * others are not in; every rank calls it (no-op without a communicator) before the job takes another way. */
int oswald_hip_comm_destroy(oswald_hip_ctx *ctx);

/* Device time spent in the DP kernels since the last reset, measured with HIP
 * events on the device's stream (enabled by oswald_hip_set_profiling). */
int oswald_hip_set_profiling(oswald_hip_ctx *ctx, int enable);
int oswald_hip_kernel_stats(oswald_hip_ctx *ctx, int dev, double *dp_kernel_ms, uint64_t *dp_launches,
                            uint64_t *rerun_items, int reset);

/* Device time of the escalation tiers since the last reset of oswald_hip_kernel_stats (profiling enabled): out[0] = the
 * int16 re-run launches behind the 8-bit pass, out[1] = the int32 re-run launches (both are part of dp_kernel_ms).
 * What the reference spends in its two overflow branches, HybridSearch.c:1683-1771, :1787-1874 / sw_host
 * FPGAsearch.c:377-507. */
int oswald_hip_rerun_stats(oswald_hip_ctx *ctx, int dev, double *ms2);

/* Escalations of the most recent search on `dev`: out[0] = work items the 8-bit
 * pass queued for the int16 re-run, out[1] = sequences the int16 cells queued
 * for the int32 re-run (the reference's two overflow tests,
 * HybridSearch.c:1670-1680, :1774-1784). */
int oswald_hip_rerun_counts(oswald_hip_ctx *ctx, int dev, uint64_t *out2);

/* Geometry of a resident chunk as the kernels see it (for roofline accounting):
 * out[0] = wave blocks, out[1] = stored 4-column groups, out[2] = 4-column
 * groups after trimming all-dummy tail columns, out[3] = bytes of re-tiled
 * residues the DP kernel reads per query, out[4] = work items of the last
 * search's queue, out[5] = log2 of the widest wave geometry in it, out[6] = bytes the
 * strip-boundary spill of one search writes and reads back according to its plan
 * (long queries do not fit the registers + LDS of a wave: SURVEY 8d "long-query
 * spill"), out[7] = 0 (reserved). */
int oswald_hip_chunk_geometry(oswald_hip_ctx *ctx, int dev, int chunk, uint64_t *out8);

#ifdef __cplusplus
}
#endif
#endif
