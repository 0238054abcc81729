/*
 * fft_wgpu_amd.h -- C ABI of the MI355X-native batched 1-D complex fp32 FFT.
 *
 * This is the drop-in boundary for the compute path of the Rust crate
 * TYPEmber/fft_wgpu.  The crate reaches its GPU through inline `wgpu` calls
 * (its FFI seam `src/wgpu_helper.rs` is an empty file declared at
 * `src/lib.rs:8`); every entry point below names the reference call it
 * replaces.  Plain pointers and sizes only: no C++ types, no exceptions, no
 * torch types cross this boundary.  INTEGRATION.md shows the Rust `extern "C"`
 * block and the `Forward<'a>`-style wrappers a maintainer would add.
 *
 * Data layout: a buffer is `batch` transforms back to back, transform b at
 * element offset b*fft_len, each element {f32 re, f32 im} interleaved, little
 * endian (reference `src/lib.rs:10-15`, `src/kernel/fft4.wgsl:21-22`).
 * Offsets and sizes are 64-bit (the reference truncates to u32:
 * `src/processor.rs:30-31`).
 *
 * Threading: ctx / plan creation and destruction are not thread-safe.
 * One in-flight fwa_plan_exec per plan (plans own scratch).  One ctx per
 * device ordinal; multi-GPU = one process (or one ctx) per device: every
 * entry point makes its context's device current before it touches HIP.
 * Internal streams: the pipelined plans (2^20 two-pass, tiled) run their
 * groups of transforms on up to "streams" chain streams that belong to the
 * CONTEXT and are shared by all its plans: two plans executed on two caller
 * streams of one context therefore serialise on the chains (each exec still
 * forks from and joins to its own caller stream; results are unaffected).
 * Give concurrent pipelines their own context.  Streams this library creates
 * (the chains, fwa_stream_create) are checked to really overlap their peers
 * with ~40-us spin kernels at creation -- only when there is a peer to overlap
 * with, never on the null stream, refused (FWA_ERR_UNSUPPORTED) while a stream
 * of the context captures a graph; fwa_ctx_set_i64(ctx, "chain_check", 0)
 * turns the check off.
 *
 * Lifetimes: plans and communicators go before their context, buffers before the plans that use them.  Buffer, stream
 * and event HANDLES may be destroyed after their context (hosts with garbage collection free in any order): the context
 * keeps a list of its live handles and detaches them in fwa_ctx_destroy, so USING one afterwards -- upload, download,
 * copy, plan creation, event record ... -- returns FWA_ERR_INVALID_ARG; it never touches freed memory.
 *
 * Errors: every function returns an fwa_status (0 = ok).  Nothing aborts or
 * throws across the ABI.  fwa_last_error_string() gives detail.
 */
#ifndef FFT_WGPU_AMD_H
#define FFT_WGPU_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FWA_ABI_VERSION 4

typedef enum fwa_status {
    FWA_OK = 0,
    FWA_ERR_INVALID_ARG = 1,  /* non power-of-two fft_len, size not a multiple of 8*fft_len, NULL handle ... */
    FWA_ERR_OUT_OF_MEMORY = 2,
    FWA_ERR_HIP = 3,          /* HIP runtime call failed */
    FWA_ERR_LAUNCH = 4,       /* kernel launch failed */
    FWA_ERR_NO_DEVICE = 5,    /* no usable gfx950 device (reference: prepare_gpu -> None, src/lib.rs:43,59) */
    FWA_ERR_UNSUPPORTED = 6
} fwa_status;

/* The four plan structs of reference src/processor.rs. */
typedef enum fwa_plan_kind {
    FWA_FORWARD = 0,           /* processor.rs:7-159   + kernel/fft4.wgsl    : unnormalised forward          */
    FWA_INVERSE_SCALED = 1,    /* processor.rs:231-341 + kernel/ifft.wgsl    : inverse, 1/n fused            */
    FWA_INVERSE_UNSCALED = 2,  /* processor.rs:566-670 + kernel/onlyifft.wgsl: inverse, no scale             */
    FWA_NORMALIZE = 3          /* processor.rs:409-505 + kernel/normalize.wgsl: b[i] = a[i] / f32(fft_len)   */
} fwa_plan_kind;

typedef struct fwa_ctx fwa_ctx;       /* wgpu::Instance + Adapter + Device + Queue  (lib.rs:29-62)        */
typedef struct fwa_stream fwa_stream; /* wgpu::CommandEncoder + Queue::submit + Device::poll              */
typedef struct fwa_buf fwa_buf;       /* wgpu::Buffer                                                      */
typedef struct fwa_plan fwa_plan;     /* Forward / Inverse / Onlyinverse / Normalize                       */
typedef struct fwa_event fwa_event;   /* (no reference analogue: the reference has timestamp_writes: None) */

/* ---- library ---------------------------------------------------------- */
int32_t fwa_abi_version(void);
/* Detail of the most recent failure on `ctx` (or of the last ctx-less call when ctx == NULL).
 * The pointer stays valid until the next failing call on the same ctx. */
const char *fwa_last_error_string(const fwa_ctx *ctx);
const char *fwa_status_string(int32_t status);

/* ---- device / queue : replaces lib.rs:29-62, examples/basic.rs:6-30 ---- */
int32_t fwa_device_count(int32_t *count);
/* instance.enumerate_adapters(..) (lib.rs:33-35): describes ordinal 0 <= device_ordinal < fwa_device_count without
 * creating a context.  name: gcnArchName, truncated to name_cap; *usable = 1 when fwa_ctx_create would accept the
 * device (gfx950).  Any output pointer may be NULL. */
int32_t fwa_device_info(int32_t device_ordinal, char *name, size_t name_cap, int32_t *compute_units,
                        uint64_t *hbm_bytes, int32_t *usable);
int32_t fwa_ctx_create(int32_t device_ordinal, fwa_ctx **out);
int32_t fwa_ctx_destroy(fwa_ctx *ctx);
/* device.poll(Maintain::wait()) (examples/basic.rs:106): returns once ALL work submitted to this device, on
 * any stream, has completed. */
int32_t fwa_ctx_synchronize(fwa_ctx *ctx);
/* Context counters (no reference analogue).  Keys: "device"; plan cache (tables of a transform length are
 * built once per context and shared by every later plan of that length; ring allocations of destroyed plans
 * are reused): "table_builds", "table_cache_hits", "ring_allocs", "ring_reuses", "pooled_ring_bytes" (<= 1 GiB);
 * "last_plan_create_us" (wall time of the most recent fwa_plan_create); "mem_free_bytes" / "mem_total_bytes"
 * (hipMemGetInfo of the context's device). */
int32_t fwa_ctx_get_i64(const fwa_ctx *ctx, const char *key, int64_t *value);
/* Settable: "chain_check" (1 default; 0 = streams this library creates are not tested for overlap: see Threading). */
int32_t fwa_ctx_set_i64(fwa_ctx *ctx, const char *key, int64_t value);
/* name: NUL-terminated gcnArchName ("gfx950:..."), truncated to name_cap. */
int32_t fwa_ctx_device_info(const fwa_ctx *ctx, char *name, size_t name_cap,
                            int32_t *compute_units, uint64_t *hbm_bytes);

/* ---- stream : replaces create_command_encoder / submit / poll(wait)
 *               (examples/basic.rs:76,92,105-106) ------------------------- */
int32_t fwa_stream_create(fwa_ctx *ctx, fwa_stream **out);
/* Wrap an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = the null stream.
 * The wrapped stream is not destroyed by fwa_stream_destroy and must OUTLIVE its wrapper: the library queries wrapped
 * handles (capture status before it creates streams of its own) and HIP does not validate stream handles. */
int32_t fwa_stream_wrap(fwa_ctx *ctx, void *hip_stream, fwa_stream **out);
int32_t fwa_stream_synchronize(fwa_stream *stream);
int32_t fwa_stream_destroy(fwa_stream *stream);

/* ---- buffers : replaces create_buffer / write_buffer / copy_buffer_to_buffer /
 *                map_async+get_mapped_range (examples/basic.rs:50-64,73,84-90,105-122) */
int32_t fwa_buf_alloc(fwa_ctx *ctx, uint64_t bytes, fwa_buf **out);
/* Zero-copy interop: wrap device memory owned by someone else (never freed by fwa_buf_free). */
int32_t fwa_buf_wrap(fwa_ctx *ctx, void *device_ptr, uint64_t bytes, fwa_buf **out);
int32_t fwa_buf_free(fwa_buf *buf);
int32_t fwa_buf_upload(fwa_buf *dst, uint64_t dst_offset, const void *host, uint64_t bytes,
                       fwa_stream *stream);
int32_t fwa_buf_download(void *host, const fwa_buf *src, uint64_t src_offset, uint64_t bytes,
                         fwa_stream *stream);
/* Device-to-device copy, stream-ordered on `stream` (a stream of dst's or src's context; NULL = the null stream of dst's
 * device).  The two buffers may belong to contexts on DIFFERENT devices (one process driving several GPUs, SURVEY.md
 * 8(e)): the copy is then an explicit peer copy over xGMI / PCIe, peer access being enabled on first use; devices that
 * cannot reach each other give FWA_ERR_UNSUPPORTED (stage through the host, or use fwa_comm_*), never a wild pointer. */
int32_t fwa_buf_copy(fwa_buf *dst, uint64_t dst_offset, const fwa_buf *src, uint64_t src_offset,
                     uint64_t bytes, fwa_stream *stream);
/* *kind: 0 = the devices of the two contexts cannot reach each other, 1 = same device, 2 = peer access (enabled here). */
int32_t fwa_ctx_peer_access(fwa_ctx *ctx, fwa_ctx *peer, int32_t *kind);
/* Pinned (page-locked) host staging memory: with it fwa_buf_upload is truly asynchronous and
 * fwa_buf_download_async can overlap with transforms on another stream -- the reference's benchmark loop
 * (examples/basic.rs:72-127: write_buffer -> proc -> copy -> map) pipelined over three streams. */
int32_t fwa_host_alloc(fwa_ctx *ctx, uint64_t bytes, void **out);
int32_t fwa_host_free(fwa_ctx *ctx, void *ptr);
/* Stream-ordered device->host copy that does NOT wait (fwa_buf_download does, like map_async + poll). */
int32_t fwa_buf_download_async(void *host, const fwa_buf *src, uint64_t src_offset, uint64_t bytes,
                               fwa_stream *stream);
/* Make `stream` wait (on the device) for everything enqueued on `other` so far. */
int32_t fwa_stream_wait_stream(fwa_stream *stream, fwa_stream *other);
void *fwa_buf_device_ptr(const fwa_buf *buf);
uint64_t fwa_buf_size(const fwa_buf *buf);

/* ---- plans : replaces X::new / X::proc of processor.rs ------------------
 * fwa_plan_create mirrors
 *   Forward::new(&device,&queue,&src,fft_len)              processor.rs:22-27   (src2 == NULL)
 *   Inverse::new(&device,&queue,&src,fft_len)              processor.rs:245-250 (src2 == NULL)
 *   Onlyinverse::new(&device,&queue,&src,&src2,fft_len)    processor.rs:580-586 (src2 required)
 *   Normalize::new(&device,&queue,&buffer1,&buffer2,fft_len) processor.rs:422-428 (src2 required)
 * batch is implicit: fwa_buf_size(src) / 8 / fft_len.  Buffers stay caller-owned and must
 * outlive the plan.  Rejected with FWA_ERR_INVALID_ARG: fft_len not a power of two (or 0),
 * size(src) % (8*fft_len) != 0, size(src2) != size(src), missing/extra src2.
 *
 * fwa_plan_exec mirrors X::proc(&self, &mut encoder) -> &wgpu::Buffer
 * (processor.rs:110,293,467,622): asynchronous and stream-ordered, allocates nothing, and
 * returns in *result the buffer that will hold the natural-order output, following the
 * reference's rule (processor.rs:153-157,335-339,664-668): `src` when log2(fft_len) is even,
 * otherwise the second buffer (plan-owned for FORWARD / INVERSE_SCALED, caller's src2 for
 * INVERSE_UNSCALED).  NORMALIZE reads (buffer1 if log2 even else buffer2) and writes and returns
 * the other (processor.rs:433-439,504).  The contents of the non-result buffer are unspecified
 * afterwards (the reference clobbers its input).  *result may be NULL-checked by callers that
 * do not need it: pass result == NULL. */
int32_t fwa_plan_create(fwa_ctx *ctx, int32_t kind, uint32_t fft_len, fwa_buf *src,
                        fwa_buf *src2_or_null, fwa_plan **out);
int32_t fwa_plan_exec(fwa_plan *plan, fwa_stream *stream, fwa_buf **result);
int32_t fwa_plan_destroy(fwa_plan *plan);

/* Pure host logic, no device needed: which path and factorisation a plan of length fft_len uses (for a batch
 * of at least 4 transforms; at n = 2^20 smaller batches take the tiled path, see "factors").
 * *path as in fwa_plan_get_i64("path"); log2_factors[0..2] = log2 of the per-pass FFT lengths
 * (0 = unused), e.g. 2^20 -> {10,10,0}, 2^24 -> {9,7,8}, 2^18 -> {8,10,0}, 512 -> {9,0,0}.  The answer is the
 * many-transform (throughput) regime; a plan over few transforms may pick smaller tiles: fwa_plan_get_i64("factors") on
 * the plan is authoritative. */
int32_t fwa_describe_path(uint32_t fft_len, int32_t *path, uint32_t log2_factors[3]);

/* Introspection / tuning (no reference analogue).  No key changes what a plan computes: every path and geometry
 * of a size produces the same transform (the tests compare them bit for bit where they share arithmetic).
 * Keys for fwa_plan_get_i64:
 *   "batch", "fft_len",
 *   "path": 0 one-launch kernels (n <= 32768), 1 two-pass 2^20 pipeline (one launch per pass and group of
 *           transforms), 2 literal radix-2 recurrence (one launch per stage, kernel/fft.wgsl:27-62; forced only),
 *           3 normalize, 4 identity (n = 1),
 *           7 tiled pipeline: two passes at 2^16..2^19 and 2^21..2^23, three at 2^24..2^30 and at 2^20 with fewer than 4 transforms,
 *   "factors": log2(N1) | log2(N2) << 8 | log2(N3) << 16 of a multi-pass plan,
 *   "launches_per_exec", "scratch_bytes", "tables_shared" (other holders of this plan's twiddle tables),
 *   "group" (transforms per launch), "streams" (chain streams the groups alternate over, <= 16)   [paths 1, 7]
 *   "xcd_swizzle" (XCD-aware block -> tile mapping: bit 0 = every XCD takes a contiguous run of tiles; path 1 also bit 2 =
 *                  the two resident workgroups of a CU take adjacent tiles; path 1 defaults to 5, path 7 per size: 5 at 2^16 and
 *                  2^21, 1 at 2^17 .. 2^19, 0 elsewhere)                                                              [paths 1, 7]
 *   "colsw" (tiled plans whose first factor is 256 / 512: 1 = k_colsw, 256 x 64 / 512 x 32 column tiles, as pass A -- the
 *   default in the many-transform regime at 2^16..2^19 and 2^24..2^28 -- 0 = the generic tile kernel),
 *   "tile_ring" (k_colsw followed by the row kernel: 1 = tile-contiguous ring slab (default), 0 = matrix layout),
 *   "p1_gen" (tiled plans whose first factor is 1024: 1 = the 2^20 pipeline's column kernel at run-time
 *   pitch as pass A (default), 0 = the generic tile kernel),
 *   "rows32" (two-pass tiled plans whose second factor is 512 .. 4096: 1 = 32-point-per-thread row kernel with the
 *   transposed store as last pass (default), 0 = the generic tile kernel; 2048 and 4096 exist only in the former).
 * Settable with fwa_plan_set_i64 before the first exec: "group", "streams", "xcd_swizzle", "factors", "colsw",
 * "tile_ring", "p1_gen", "rows32", "path" (2 anywhere).
 * fwa_ctx_get_i64: "device", "table_builds", "table_cache_hits", "ring_allocs", "ring_reuses", "pooled_ring_bytes",
 * "last_plan_create_us", "mem_free_bytes", "mem_total_bytes", "chain_streams" (chain streams created so far),
 * "chain_checks" / "chain_rejects" (candidates tested / discarded because they did not overlap the other chains),
 * "chain_single_us" / "chain_pair_us" (the check's spin kernel alone / on all chains at once), "chain_check",
 * "live_streams" / "live_buffers" (alive fwa_stream / fwa_buf handles of the context).
 *
 * Laboratory build only (fft_wgpu_amd/libfft_wgpu_amd_lab.so, `make -C fft_wgpu_amd/csrc lab`; the product library
 * answers FWA_ERR_UNSUPPORTED): kernel families that measured slower than the shipped ones, kept for A/B timing and
 * bit-identity tests --
 *   "path" 5 (the 2^20 pipeline as ONE persistent launch with a small ring; 1 <-> 5 at 2^20) with "depth", "ring_slots",
 *   "wgs"; "device_error" (path 5; synchronises the device; non-zero = a bounded in-kernel spin timed out);
 *   "small_reg" (n <= 32768: 1 = the shipped kernels; 3 = the direct-addressing 16-point kernels at 16 .. 4096; 2 = the
 *   same up to 256 with the wavefront-shuffle (`__shfl_xor`) exchange at 32 / 64 / 128); "ring_rotate" (2^20: the groups
 *   walk through a ring that many times larger); "inject_launch_failure" (pipelined paths: the launch of that group
 *   fails once -- the error path of fwa_plan_exec under test).
 * "tile_w" reads 16 (the only tile width of the 2^20 pipeline; it can be "set" to 16 only). */
int32_t fwa_plan_get_i64(const fwa_plan *plan, const char *key, int64_t *value);
int32_t fwa_plan_set_i64(fwa_plan *plan, const char *key, int64_t value);

/* ---- multi-GPU: batch sharding (no reference analogue: the reference drives one device, one queue) ------------------
 * Transforms are independent (kernel/fft4.wgsl:21-23: one `offset` per workgroup), so a batch shards as contiguous slabs
 * of whole transforms, one slab per GPU, and the transform itself never communicates.
 *
 * fwa_slab: the slab rule -- rank r of `world` owns transforms [*first, *first + *count), counts differ by at most one
 * (pure host logic, no device needed; fft_wgpu_amd/sharding.py::slab and fft_wgpu::slab are this function).
 *
 * Moving slabs when the batch starts on (or must return to) one GPU:
 *  - ONE process driving several contexts holds every pointer: fwa_buf_copy between buffers of two contexts (peer copy);
 *    fft_wgpu::ShardedBatch (include/fft_wgpu.hpp) and fft_wgpu_amd.ShardedBatch are built on it.
 *  - one process PER GPU (torchrun-style deployment): fwa_comm_* below, grouped ncclSend / ncclRecv over RCCL (xGMI).
 *    librccl is dlopen'ed on the first fwa_comm_* call; FWA_ERR_UNSUPPORTED if it cannot be loaded.
 * fwa_comm_unique_id: rank 0 makes the id and hands the 128 bytes to the other ranks by any means (file, socket,
 * torch.distributed store).  fwa_comm_create is collective over the `world` processes holding the same id, one rank per
 * device.  ENVIRONMENT: RCCL maps its peers' buffers through HIP IPC.  On hosts whose kernel driver supports only dmabuf
 * IPC (every box of the pool this library was developed on) the process must have HSA_ENABLE_IPC_MODE_LEGACY=0 in its
 * environment BEFORE the HIP runtime is loaded (i.e. before the first call into this library or into any other HIP user
 * of the process; setenv() from main() is early enough, after fwa_ctx_create it is not): without it communicator creation
 * or the first exchange fails inside RCCL ("hipIpcGetMemHandle: invalid argument" / ncclUnhandledCudaError).  When
 * fwa_comm_create fails and the variable is unset or not "0", the error text says so.  Scatter / gather are collective too, stream-ordered on `stream` (a stream of the communicator's context):
 *   scatter: root's `full` (batch transforms) -> every rank's `slab` (>= its count transforms); `full` is ignored elsewhere
 *   gather : every rank's `slab` -> root's `full`
 * fwa_comm_sendrecv: the primitive under both -- one send and/or one receive in one group (rank < 0 = none); sending to the
 * own rank needs the matching receive in the same call (a ring shift of slabs is `sendrecv(to = r+1, from = r-1)`).
 * Errors in a collective are fatal for the COMMUNICATOR: a rank whose arguments are rejected (FWA_ERR_INVALID_ARG: slab
 * buffer too small, root without a full buffer, foreign stream) has posted nothing, while its peers have already enqueued
 * the matching sends / receives and will wait on their streams for ever -- there is no timeout.  Validate sizes before
 * the call on every rank; after a failure destroy the communicator on every rank (and the processes with it, if peers
 * are already blocked).  These calls allocate a few small host vectors on the root; fwa_plan_exec never allocates.
 * fwa_comm_pieces: pure host logic, no device and no RCCL needed -- the point-to-point pieces rank `rank` posts in a
 * scatter or gather of `batch` transforms rooted at `root` (the table both collectives are built on; arrays of `world`
 * entries): a non-root rank posts one piece (offset 0 into its slab buffer, its whole slab, peer = root), the root posts
 * `world` pieces (byte offset of rank p's slab in the full batch, its bytes, peer = p, or -1 for the root's own slab,
 * which moves by a device copy on the same stream). */
typedef struct fwa_comm fwa_comm;
#define FWA_COMM_ID_BYTES 128
int32_t fwa_slab(uint64_t batch, int32_t rank, int32_t world, uint64_t *first, uint64_t *count);
int32_t fwa_comm_pieces(uint64_t batch, uint32_t fft_len, int32_t root, int32_t rank, int32_t world, uint64_t *offset,
                        uint64_t *bytes, int32_t *peer, int32_t *n_pieces);
int32_t fwa_comm_unique_id(uint8_t id[FWA_COMM_ID_BYTES]);
int32_t fwa_comm_create(fwa_ctx *ctx, const uint8_t id[FWA_COMM_ID_BYTES], int32_t world, int32_t rank, fwa_comm **out);
int32_t fwa_comm_destroy(fwa_comm *comm);
/* Keys: "rank", "world", "device". */
int32_t fwa_comm_get_i64(const fwa_comm *comm, const char *key, int64_t *value);
int32_t fwa_comm_sendrecv(fwa_comm *comm, const fwa_buf *send, uint64_t send_offset, uint64_t send_bytes, int32_t send_to,
                          fwa_buf *recv, uint64_t recv_offset, uint64_t recv_bytes, int32_t recv_from,
                          fwa_stream *stream);
int32_t fwa_comm_scatter(fwa_comm *comm, int32_t root, const fwa_buf *full_or_null, fwa_buf *slab, uint32_t fft_len,
                         uint64_t batch, fwa_stream *stream);
int32_t fwa_comm_gather(fwa_comm *comm, int32_t root, const fwa_buf *slab, fwa_buf *full_or_null, uint32_t fft_len,
                        uint64_t batch, fwa_stream *stream);

/* ---- measurement helpers (HIP events on the launch stream) -------------- */
int32_t fwa_event_create(fwa_ctx *ctx, fwa_event **out);
int32_t fwa_event_record(fwa_event *ev, fwa_stream *stream);
/* Host waits for the work recorded before `ev`. */
int32_t fwa_event_synchronize(fwa_event *ev);
/* Device-side dependency on ONE recorded point of another stream (fwa_stream_wait_stream waits for everything
 * enqueued so far): what the pipelined form of the reference's loop (examples/basic.rs:72-127) needs per slot. */
int32_t fwa_stream_wait_event(fwa_stream *stream, fwa_event *ev);
/* Blocks until `end` has completed. */
int32_t fwa_event_elapsed_ms(fwa_event *start, fwa_event *end, float *ms);
int32_t fwa_event_destroy(fwa_event *ev);

/* Deterministic synthetic input generated on the device: sample i of
 * transform t is a pure function of (seed, (first_transform+t)*fft_len + i); re, im uniform
 * in [-1,1) times `scale`.  Bit-identical to oracle/ref_fft.c:fwo_gen_input. */
int32_t fwa_fill_synthetic(fwa_buf *dst, uint64_t seed, uint64_t first_transform,
                           uint32_t fft_len, float scale, fwa_stream *stream);
/* Streaming copy of `bytes` from src to dst (the measured-ceiling calibration kernel: one workgroup per 64-KiB chunk, nt).
 * With dst and src the SAME memory it is an in-place streaming pass instead -- every line read, then written back, in the
 * launch shape of the one-launch FFT kernels -- the ceiling a single-pass transform is compared with. */
int32_t fwa_calib_copy(fwa_buf *dst, const fwa_buf *src, uint64_t bytes, fwa_stream *stream);

#ifdef __cplusplus
}
#endif
#endif /* FFT_WGPU_AMD_H */
