// comm.cpp -- fwa_comm_*: moving slabs of whole transforms between the GPUs of one node over RCCL (xGMI).
//
// The transform itself never communicates (reference src/kernel/fft4.wgsl:21-23: one `offset` per workgroup, no
// cross-transform access).  This file only serves callers whose batch starts -- or must end up -- on
// one GPU while the other GPUs belong to OTHER processes (one process per GPU, the deployment north_star names): root
// scatters slabs with grouped ncclSend / ncclRecv, the mirror gathers.  One process that drives several devices
// itself needs none of it: it holds every pointer and moves slabs with fwa_buf_copy (peer copies, buffers.cpp).
//
// librccl is loaded on the first fwa_comm_* call (dlopen), not linked: a caller that never shards pays neither its
// load time nor its dependency.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "internal.h"

struct fwa_comm {
    fwa_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int32_t world = 0, rank = 0;
};

namespace {

struct Rccl {
    void *so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;  // why loading failed
};

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.so) break;
        }
        if (!r.so) { r.why = std::string("dlopen(librccl.so.1): ") + dlerror(); return; }
        auto sym = [&](const char *n) {
            void *p = dlsym(r.so, n);
            if (!p && r.why.empty()) r.why = std::string("librccl has no symbol ") + n;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return &r;
}

int32_t need_rccl(const fwa_ctx *ctx, Rccl **out)
{
    Rccl *r = rccl();
    if (!r->so || !r->why.empty()) return fwa_int::fail(ctx, FWA_ERR_UNSUPPORTED, "RCCL is not available: " + r->why);
    *out = r;
    return FWA_OK;
}

int32_t fail_nccl(const fwa_ctx *ctx, Rccl *r, ncclResult_t e, const char *what)
{
    return fwa_int::fail(ctx, FWA_ERR_HIP,
                         std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(e) : "RCCL error"));
}

// "" when HSA_ENABLE_IPC_MODE_LEGACY=0 is in the environment, otherwise the sentence appended to an RCCL failure
std::string ipc_mode_hint()
{
    const char *v = std::getenv("HSA_ENABLE_IPC_MODE_LEGACY");
    if (v && std::strcmp(v, "0") == 0) return "";
    return std::string("; HSA_ENABLE_IPC_MODE_LEGACY is ") + (v ? "\"" + std::string(v) + "\"" : "unset") +
           ": on hosts whose driver supports only dmabuf IPC RCCL's peer mappings fail unless HSA_ENABLE_IPC_MODE_LEGACY=0 "
           "is in the environment before the HIP runtime loads (include/fft_wgpu_amd.h, fwa_comm)";
}

static_assert(sizeof(ncclUniqueId) == FWA_COMM_ID_BYTES, "FWA_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

struct Piece {  // one side of a point-to-point transfer: bytes at ptr, to / from `peer` (-1: none)
    char *ptr;
    uint64_t bytes;
    int32_t peer;
};

// ranges of fwa_buf handles are checked by the callers through these two accessors of the public ABI
bool in_range(const fwa_buf *b, uint64_t off, uint64_t bytes) { return off <= fwa_buf_size(b)
                                                               && bytes <= fwa_buf_size(b) - off; }
char *at(const fwa_buf *b, uint64_t off) { return static_cast<char *>(fwa_buf_device_ptr(b)) + off; }

// Grouped sends and receives of this rank on `stream`; a piece addressed to the rank itself must have its partner in
// the same group (RCCL pairs them up as a local copy).
int32_t exchange(fwa_comm *c, const Piece *sends, size_t ns, const Piece *recvs, size_t nr, fwa_stream *stream)
{
    Rccl *r = nullptr;
    int32_t st = need_rccl(c->ctx, &r);
    if (st) return st;
    if ((st = fwa_int::use_device(c->ctx))) return st;
    hipStream_t hs = fwa_int::stream_raw(stream);
    ncclResult_t e = r->GroupStart();
    if (e != ncclSuccess) return fail_nccl(c->ctx, r, e, "ncclGroupStart");
    for (size_t i = 0; i < ns && e == ncclSuccess; ++i)
        if (sends[i].peer >= 0 && sends[i].bytes) e = r->Send(sends[i].ptr, sends[i].bytes, ncclInt8, sends[i].peer,
                                                              c->comm, hs);
    for (size_t i = 0; i < nr && e == ncclSuccess; ++i)
        if (recvs[i].peer >= 0 && recvs[i].bytes) e = r->Recv(recvs[i].ptr, recvs[i].bytes, ncclInt8, recvs[i].peer,
                                                              c->comm, hs);
    const ncclResult_t ge = r->GroupEnd();
    if (e != ncclSuccess || ge != ncclSuccess) {
        const int32_t rc = e != ncclSuccess ? fail_nccl(c->ctx, r, e, "ncclSend/ncclRecv")
                                            : fail_nccl(c->ctx, r, ge, "ncclGroupEnd");
        const std::string hint = ipc_mode_hint();   // peers are mapped on the first exchange
        if (!hint.empty()) (void)fwa_int::fail(c->ctx, rc, std::string(fwa_last_error_string(c->ctx)) + hint);
        return rc;
    }
    return FWA_OK;
}

}  // namespace

extern "C" {

int32_t fwa_comm_unique_id(uint8_t id[FWA_COMM_ID_BYTES])
{
    if (!id) return fwa_int::fail(nullptr, FWA_ERR_INVALID_ARG, "id is NULL");
    Rccl *r = nullptr;
    int32_t st = need_rccl(nullptr, &r);
    if (st) return st;
    ncclUniqueId u;
    const ncclResult_t e = r->GetUniqueId(&u);
    if (e != ncclSuccess) return fail_nccl(nullptr, r, e, "ncclGetUniqueId");
    std::memcpy(id, &u, FWA_COMM_ID_BYTES);
    return FWA_OK;
}

int32_t fwa_comm_create(fwa_ctx *ctx, const uint8_t id[FWA_COMM_ID_BYTES], int32_t world, int32_t rank, fwa_comm **out)
{
    if (!out) return fwa_int::fail(ctx, FWA_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!ctx || !id) return fwa_int::fail(ctx, FWA_ERR_INVALID_ARG, "ctx/id is NULL");
    if (world < 1 || rank < 0 || rank >= world) return fwa_int::fail(ctx, FWA_ERR_INVALID_ARG, "bad rank / world size");
    Rccl *r = nullptr;
    int32_t st = need_rccl(ctx, &r);
    if (st) return st;
    if ((st = fwa_int::use_device(ctx))) return st;
    fwa_comm *c = new (std::nothrow) fwa_comm;
    if (!c) return fwa_int::fail(ctx, FWA_ERR_OUT_OF_MEMORY, "host allocation failed");
    ncclUniqueId u;
    std::memcpy(&u, id, FWA_COMM_ID_BYTES);
    // collective over the `world` callers that hold the same id: one rank per device (RCCL refuses two on one GPU)
    const ncclResult_t e = r->CommInitRank(&c->comm, world, u, rank);
    if (e != ncclSuccess) {
        delete c;
        const int32_t rc = fail_nccl(ctx, r, e, "ncclCommInitRank");
        // the one misconfiguration this library has met in the field: hosts whose driver only supports dmabuf IPC
        // (include/fft_wgpu_amd.h, fwa_comm block)
        const std::string hint = ipc_mode_hint();
        if (!hint.empty()) (void)fwa_int::fail(ctx, rc, std::string(fwa_last_error_string(ctx)) + hint);
        return rc;
    }
    c->ctx = ctx; c->world = world; c->rank = rank;
    *out = c;
    return FWA_OK;
}

int32_t fwa_comm_destroy(fwa_comm *comm)
{
    if (!comm) return FWA_OK;
    Rccl *r = rccl();
    if (comm->comm && r->CommDestroy) {
        (void)fwa_int::use_device(comm->ctx);
        (void)r->CommDestroy(comm->comm);
    }
    delete comm;
    return FWA_OK;
}

int32_t fwa_comm_get_i64(const fwa_comm *comm, const char *key, int64_t *value)
{
    if (!comm || !key || !value) return fwa_int::fail(comm ? comm->ctx : nullptr, FWA_ERR_INVALID_ARG, "NULL argument");
    const std::string k(key);
    if (k == "rank") *value = comm->rank;
    else if (k == "world") *value = comm->world;
    else if (k == "device") *value = fwa_int::ctx_device(comm->ctx);
    else return fwa_int::fail(comm->ctx, FWA_ERR_INVALID_ARG, "unknown key: " + k);
    return FWA_OK;
}

int32_t fwa_comm_sendrecv(fwa_comm *comm, const fwa_buf *send, uint64_t send_offset, uint64_t send_bytes,
                          int32_t send_to,
                          fwa_buf *recv, uint64_t recv_offset, uint64_t recv_bytes, int32_t recv_from,
                              fwa_stream *stream)
{
    if (!comm) return fwa_int::fail(nullptr, FWA_ERR_INVALID_ARG, "comm is NULL");
    fwa_ctx *ctx = comm->ctx;
    if (send_to >= comm->world || recv_from >= comm->world)
        return fwa_int::fail(ctx, FWA_ERR_INVALID_ARG, "peer rank out of range");
    if (send_to >= 0 && (!send || !in_range(send, send_offset, send_bytes)))
        return fwa_int::fail(ctx, FWA_ERR_INVALID_ARG, "send range exceeds buffer");
    if (recv_from >= 0 && (!recv || !in_range(recv, recv_offset, recv_bytes)))
        return fwa_int::fail(ctx, FWA_ERR_INVALID_ARG, "receive range exceeds buffer");
    if ((send_to == comm->rank) != (recv_from == comm->rank) || (send_to == comm->rank && send_bytes != recv_bytes))
        return fwa_int::fail(ctx, FWA_ERR_INVALID_ARG,
                             "a send to this rank itself needs the matching receive in the same call");
    if (stream && fwa_int::stream_ctx(stream) != ctx)
        return fwa_int::fail(ctx, FWA_ERR_INVALID_ARG, "the stream belongs to another context");
    const Piece s{send_to >= 0 ? at(send, send_offset) : nullptr, send_bytes, send_to};
    const Piece r{recv_from >= 0 ? at(recv, recv_offset) : nullptr, recv_bytes, recv_from};
    return exchange(comm, &s, 1, &r, 1, stream);
}

// Slabs follow fwa_slab(batch, rank, world): rank r owns transforms [first_r, first_r + count_r).
//
// The point-to-point pieces one rank posts in a scatter or a gather -- pure host logic (no device, no RCCL), so that
// the table both collectives are built on can be checked for every world size without a second GPU.  A rank other
// than the root posts ONE piece (its whole slab, peer = root, offset 0 into its slab buffer); the root posts `world`
// pieces, piece p = rank p's slab at byte offset first_p * 8 * fft_len of the full batch, peer = p, and peer = -1 for
// its own slab (moved by a device copy on the same stream instead).
int32_t fwa_comm_pieces(uint64_t batch, uint32_t fft_len, int32_t root, int32_t rank, int32_t world, uint64_t *offset,
                        uint64_t *bytes, int32_t *peer, int32_t *n_pieces)
{
    if (!offset || !bytes || !peer || !n_pieces) return fwa_int::fail(nullptr, FWA_ERR_INVALID_ARG, "NULL argument");
    *n_pieces = 0;
    if (!fft_len) return fwa_int::fail(nullptr, FWA_ERR_INVALID_ARG, "fft_len is 0");
    if (world < 1 || rank < 0 || rank >= world || root < 0 || root >= world)
        return fwa_int::fail(nullptr, FWA_ERR_INVALID_ARG, "bad rank / root / world size");
    const uint64_t tb = 8ull * fft_len;
    uint64_t f = 0, n = 0;
    if (rank != root) {
        (void)fwa_slab(batch, rank, world, &f, &n);
        offset[0] = 0; bytes[0] = n * tb; peer[0] = root;
        *n_pieces = 1;
        return FWA_OK;
    }
    for (int32_t p = 0; p < world; ++p) {
        (void)fwa_slab(batch, p, world, &f, &n);
        offset[p] = f * tb; bytes[p] = n * tb; peer[p] = p == root ? -1 : p;
    }
    *n_pieces = world;
    return FWA_OK;
}

namespace {

// scatter (gather = false): root's `full` -> every rank's `slab`; gather: the mirror image.  Host memory: two small
// vectors per call on the root (these calls, unlike fwa_plan_exec, are not allocation-free).
int32_t move_slabs(fwa_comm *comm, bool gather, int32_t root, const fwa_buf *slab, const fwa_buf *full,
                   uint32_t fft_len,
                   uint64_t batch, fwa_stream *stream)
{
    if (!comm || !slab || !fft_len)
        return fwa_int::fail(comm ? comm->ctx : nullptr, FWA_ERR_INVALID_ARG, "comm/slab is NULL or fft_len is 0");
    fwa_ctx *ctx = comm->ctx;
    if (root < 0 || root >= comm->world) return fwa_int::fail(ctx, FWA_ERR_INVALID_ARG, "root out of range");
    if (stream && fwa_int::stream_ctx(stream) != ctx)
        return fwa_int::fail(ctx, FWA_ERR_INVALID_ARG, "the stream belongs to another context");
    const bool is_root = comm->rank == root;
    std::vector<uint64_t> off((size_t)(is_root ? comm->world : 1)), len(off.size());
    std::vector<int32_t> peer(off.size());
    int32_t np = 0;
    int32_t st = fwa_comm_pieces(batch, fft_len, root, comm->rank, comm->world, off.data(), len.data(), peer.data(),
                                 &np);
    if (st) return st;
    const uint64_t tb = 8ull * fft_len;
    uint64_t first = 0, count = 0;
    (void)fwa_slab(batch, comm->rank, comm->world, &first, &count);
    if (!in_range(slab, 0, count * tb))
        return fwa_int::fail(ctx, FWA_ERR_INVALID_ARG, "slab buffer is smaller than this rank's slab");
    if (!is_root) {
        const Piece one{at(slab, 0), len[0], peer[0]};
        return gather ? exchange(comm, &one, 1, nullptr, 0, stream) : exchange(comm, nullptr, 0, &one, 1, stream);
    }
    if (!full || !in_range(full, 0, batch * tb))
        return fwa_int::fail(ctx, FWA_ERR_INVALID_ARG, "root needs the full batch buffer");
    std::vector<Piece> pieces((size_t)np);
    for (int32_t p = 0; p < np; ++p) pieces[(size_t)p] = Piece{at(full, off[(size_t)p]), len[(size_t)p],
                                                               peer[(size_t)p]};
    st = gather ? exchange(comm, nullptr, 0, pieces.data(), pieces.size(), stream)
                : exchange(comm, pieces.data(), pieces.size(), nullptr, 0, stream);
    if (st) return st;
    // the root's own slab: a device copy on the same stream (skipped when the caller's slab IS that part of the batch)
    if (count && at(slab, 0) != at(full, first * tb))
        st = gather ? fwa_buf_copy(const_cast<fwa_buf *>(full), first * tb, slab, 0, count * tb, stream)
                    : fwa_buf_copy(const_cast<fwa_buf *>(slab), 0, full, first * tb, count * tb, stream);
    return st;
}

}  // namespace

// move_slabs allocates small host vectors: nothing may throw across the C ABI
static int32_t move_slabs_nothrow(fwa_comm *comm, bool gather, int32_t root, const fwa_buf *slab, const fwa_buf *full,
                           uint32_t fft_len, uint64_t batch, fwa_stream *stream)
{
    try {
        return move_slabs(comm, gather, root, slab, full, fft_len, batch, stream);
    } catch (const std::bad_alloc &) {
        return fwa_int::fail(comm ? comm->ctx : nullptr, FWA_ERR_OUT_OF_MEMORY, "host allocation failed");
    } catch (...) {
        return fwa_int::fail(comm ? comm->ctx : nullptr, FWA_ERR_HIP, "unexpected exception in fwa_comm_scatter / gather");
    }
}

int32_t fwa_comm_scatter(fwa_comm *comm, int32_t root, const fwa_buf *full_or_null, fwa_buf *slab, uint32_t fft_len,
                         uint64_t batch, fwa_stream *stream)
{
    return move_slabs_nothrow(comm, false, root, slab, full_or_null, fft_len, batch, stream);
}

int32_t fwa_comm_gather(fwa_comm *comm, int32_t root, const fwa_buf *slab, fwa_buf *full_or_null, uint32_t fft_len,
                        uint64_t batch, fwa_stream *stream)
{
    return move_slabs_nothrow(comm, true, root, slab, full_or_null, fft_len, batch, stream);
}

}  // extern "C"
