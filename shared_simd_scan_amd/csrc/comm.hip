// comm.hip -- the ONE exchange step of the multi-GPU path: per-shard bitmaps -> root, hit counts summed,
// as direct RCCL calls behind the C ABI (one process per GPU, one communicator rank per process).
//
// Rows are independent, so a sharded scan is each rank's local scan with no data-path collective, followed by
//   * a gather of the ranks' bitmap slices into the root's final bitmap: grouped ncclSend / ncclRecv, every remote
//     slice received straight at its byte offset (no padding to the largest shard, no concatenation pass).  On
//     MI355X the root's 7 peers sit on 7 distinct xGMI links, so the 7 receives run in parallel (a ring all-gather
//     would push all 7 slices through every single link);
//   * an ncclAllReduce(sum) of the uint64 hit counts.
// RCCL is bound at run time (dlopen): a host that never shards needs no librccl, and inside a PyTorch process the
// already-loaded librccl is the one that is used.
#include "ctx.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <vector>

using namespace mi355;

namespace {

// types and prototypes come from rccl.h; the functions themselves are looked up with dlsym (nothing links librccl)
static_assert(sizeof(ncclUniqueId) == MI355_COMM_ID_BYTES, "unique id size");

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;
};

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // MI355_RCCL_LIB names the one library to use (a site's own build; also how the tests force the "no librccl"
        // case); otherwise the usual names, the process's already-loaded copy first
        const char *forced = getenv("MI355_RCCL_LIB");
        const char *defaults[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        std::string last;
        auto try_open = [&](const char *nm) {
            r.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (!r.handle) {
                const char *e = dlerror(); // ONE call: dlerror() clears the message it returns
                last = e ? e : "?";
            }
            return r.handle != nullptr;
        };
        if (forced && *forced)
            try_open(forced);
        else
            for (const char *nm : defaults)
                if (try_open(nm)) break;
        if (!r.handle) {
            r.why = std::string("librccl not found: ") + last;
            return;
        }
        bool ok = true;
        auto sym = [&](const char *name) {
            void *p = dlsym(r.handle, name);
            if (!p) {
                ok = false;
                r.why = std::string("librccl lacks ") + name;
            }
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        if (!ok) {
            dlclose(r.handle);
            r.handle = nullptr;
        }
    });
    return &r;
}

#define RCCL_TRY(R, expr)                                                                                     \
    do {                                                                                                      \
        ncclResult_t e_ = (expr);                                                                             \
        if (e_ != ncclSuccess) return fail(MI355_E_COMM, "%s: %s", #expr, (R)->GetErrorString ? (R)->GetErrorString(e_) : "?"); \
    } while (0)

} // namespace

struct mi355_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
};

extern "C" {

int mi355_comm_get_unique_id(void *id_out)
{
    if (!id_out) return fail(MI355_E_INVALID, "id_out is null");
    Rccl *R = rccl();
    if (!R->handle) return fail(MI355_E_COMM, "%s", R->why.c_str());
    ncclUniqueId id;
    RCCL_TRY(R, R->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return MI355_OK;
}

int mi355_comm_create(mi355_ctx *ctx, int world, int rank, const void *id, mi355_comm **out)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    if (!out || !id) return fail(MI355_E_INVALID, "null pointer");
    if (world < 1 || rank < 0 || rank >= world) return fail(MI355_E_INVALID, "rank %d outside a world of %d", rank, world);
    Rccl *R = rccl();
    if (!R->handle) return fail(MI355_E_COMM, "%s", R->why.c_str());
    CtxLock lk(ctx->mu);
    if ((rc = bind(ctx))) return rc; // ncclCommInitRank binds the communicator to the current device
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    mi355_comm *c = new mi355_comm;
    c->world = world;
    c->rank = rank;
    c->device = ctx->device;
    ncclResult_t e = R->CommInitRank(&c->comm, world, uid, rank);
    if (e != ncclSuccess) {
        delete c;
        return fail(MI355_E_COMM, "ncclCommInitRank(world %d, rank %d, device %d): %s", world, rank, ctx->device,
                    R->GetErrorString(e));
    }
    *out = c;
    return MI355_OK;
}

int mi355_comm_destroy(mi355_comm *comm)
{
    if (!comm) return MI355_OK;
    Rccl *R = rccl();
    if (R->handle && comm->comm) (void)R->CommDestroy(comm->comm);
    delete comm;
    return MI355_OK;
}

int mi355_comm_info(const mi355_comm *comm, int *world, int *rank)
{
    if (!comm) return fail(MI355_E_INVALID, "comm is null");
    if (world) *world = comm->world;
    if (rank) *rank = comm->rank;
    return MI355_OK;
}

int mi355_gather_bitmaps_at_dev(mi355_ctx *ctx, mi355_comm *comm, const void *local_dev, const uint64_t *bytes_per_rank,
                                const uint64_t *offset_per_rank, int root, void *out_dev)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    if (!comm || !bytes_per_rank) return fail(MI355_E_INVALID, "null pointer");
    if (root < 0 || root >= comm->world) return fail(MI355_E_INVALID, "root %d outside a world of %d", root, comm->world);
    if (ctx->device != comm->device) return fail(MI355_E_INVALID, "context and communicator are bound to different devices");
    const uint64_t mine = bytes_per_rank[comm->rank];
    if (mine && !local_dev) return fail(MI355_E_INVALID, "local_dev is null");
    uint64_t total = 0;
    for (int r = 0; r < comm->world; r++) total += bytes_per_rank[r];
    if (comm->rank == root && total && !out_dev) return fail(MI355_E_INVALID, "out_dev is null on the root");
    if (offset_per_rank) {
        // the slices land at caller-chosen offsets of ONE buffer: two that overlap would race on the root.  Checked on
        // EVERY rank (all ranks pass the same arrays), so a bad layout fails everywhere instead of leaving peers in a send
        for (int a = 0; a < comm->world; a++)
            for (int b = a + 1; b < comm->world; b++) {
                const uint64_t a0 = offset_per_rank[a], a1 = a0 + bytes_per_rank[a];
                const uint64_t b0 = offset_per_rank[b], b1 = b0 + bytes_per_rank[b];
                if (a1 < a0 || b1 < b0) return fail(MI355_E_INVALID, "offset + bytes overflows");
                if (bytes_per_rank[a] && bytes_per_rank[b] && a0 < b1 && b0 < a1)
                    return fail(MI355_E_INVALID, "slices of ranks %d and %d overlap: [%llu, %llu) and [%llu, %llu)", a, b,
                                (unsigned long long)a0, (unsigned long long)a1, (unsigned long long)b0, (unsigned long long)b1);
            }
    }
    Rccl *R = rccl();
    if (!R->handle) return fail(MI355_E_COMM, "%s", R->why.c_str());
    CtxLock lk(ctx->mu);
    if ((rc = bind(ctx))) return rc;
    if (comm->rank != root) {
        if (mine) RCCL_TRY(R, R->Send(local_dev, (size_t)mine, ncclUint8, root, comm->comm, ctx->stream));
        return MI355_OK;
    }
    // root: its own slice is a device-to-device copy, every remote slice is received at its final offset; the
    // receives are one group, so they progress concurrently (7 peers = 7 xGMI links on an 8-GPU node)
    uint64_t off = 0, own_off = 0;
    int nrecv = 0;
    for (int r = 0; r < comm->world; r++)
        if (r != root && bytes_per_rank[r]) nrecv++;
    ncclResult_t first_err = ncclSuccess, ge = ncclSuccess;
    if (nrecv) RCCL_TRY(R, R->GroupStart());
    for (int r = 0; r < comm->world; r++) {
        const uint64_t at = offset_per_rank ? offset_per_rank[r] : off;
        if (r == root)
            own_off = at;
        else if (bytes_per_rank[r]) {
            ncclResult_t e = R->Recv((uint8_t *)out_dev + at, (size_t)bytes_per_rank[r], ncclUint8, r, comm->comm, ctx->stream);
            if (e != ncclSuccess && first_err == ncclSuccess) first_err = e;
        }
        off += bytes_per_rank[r];
    }
    if (nrecv) ge = R->GroupEnd();
    if (first_err != ncclSuccess) return fail(MI355_E_COMM, "ncclRecv: %s", R->GetErrorString(first_err));
    if (ge != ncclSuccess) return fail(MI355_E_COMM, "ncclGroupEnd: %s", R->GetErrorString(ge));
    if (mine && (const uint8_t *)local_dev != (const uint8_t *)out_dev + own_off)
        HIP_TRY(hipMemcpyAsync((uint8_t *)out_dev + own_off, local_dev, (size_t)mine, hipMemcpyDeviceToDevice, ctx->stream));
    return MI355_OK;
}

int mi355_gather_bitmaps_dev(mi355_ctx *ctx, mi355_comm *comm, const void *local_dev, const uint64_t *bytes_per_rank,
                             int root, void *out_dev)
{
    return mi355_gather_bitmaps_at_dev(ctx, comm, local_dev, bytes_per_rank, nullptr, root, out_dev);
}

int mi355_allreduce_hits_dev(mi355_ctx *ctx, mi355_comm *comm, uint64_t *hits_dev, unsigned count)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    if (!comm || !hits_dev) return fail(MI355_E_INVALID, "null pointer");
    if (ctx->device != comm->device) return fail(MI355_E_INVALID, "context and communicator are bound to different devices");
    if (count == 0) return MI355_OK;
    Rccl *R = rccl();
    CtxLock lk(ctx->mu);
    if ((rc = bind(ctx))) return rc;
    RCCL_TRY(R, R->AllReduce(hits_dev, hits_dev, count, ncclUint64, ncclSum, comm->comm, ctx->stream));
    return MI355_OK;
}

static int sharded_finish(mi355_ctx *ctx, mi355_comm *comm, const void *local_bitmap_dev, const uint64_t *rows_per_rank,
                          int root, void *full_bitmap_dev, uint64_t *hits_dev)
{
    if (!rows_per_rank) return fail(MI355_E_INVALID, "rows_per_rank is null");
    std::vector<uint64_t> bytes(comm->world);
    for (int r = 0; r < comm->world; r++) {
        // every shard but the last must end on a whole bitmap byte, or the slices could not be laid end to end
        if (r + 1 < comm->world && rows_per_rank[r] % 8 != 0)
            return fail(MI355_E_INVALID, "shard %d has %llu rows: shards other than the last must be multiples of 8 rows "
                                         "(mi355_shard_rows yields multiples of 8192)", r, (unsigned long long)rows_per_rank[r]);
        bytes[r] = (rows_per_rank[r] + 7) / 8;
    }
    int rc = mi355_gather_bitmaps_dev(ctx, comm, local_bitmap_dev, bytes.data(), root, full_bitmap_dev);
    if (rc) return rc;
    if (hits_dev) rc = mi355_allreduce_hits_dev(ctx, comm, hits_dev, 1);
    return rc;
}

int mi355_sharded_scan_eq_dev(mi355_ctx *ctx, mi355_comm *comm, const void *packed_dev, unsigned c, int32_t key,
                              void *local_bitmap_dev, const uint64_t *rows_per_rank, int root, void *full_bitmap_dev,
                              uint64_t *hits_dev)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    if (!comm || !rows_per_rank) return fail(MI355_E_INVALID, "null pointer");
    CtxLock lk(ctx->mu);
    if ((rc = mi355_scan_eq_dev(ctx, packed_dev, rows_per_rank[comm->rank], c, key, local_bitmap_dev, hits_dev))) return rc;
    return sharded_finish(ctx, comm, local_bitmap_dev, rows_per_rank, root, full_bitmap_dev, hits_dev);
}

int mi355_sharded_scan_range_dev(mi355_ctx *ctx, mi355_comm *comm, const void *packed_dev, unsigned c, uint32_t lo,
                                 uint32_t hi, void *local_bitmap_dev, const uint64_t *rows_per_rank, int root,
                                 void *full_bitmap_dev, uint64_t *hits_dev)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    if (!comm || !rows_per_rank) return fail(MI355_E_INVALID, "null pointer");
    CtxLock lk(ctx->mu);
    if ((rc = mi355_scan_range_dev(ctx, packed_dev, rows_per_rank[comm->rank], c, lo, hi, local_bitmap_dev, hits_dev))) return rc;
    return sharded_finish(ctx, comm, local_bitmap_dev, rows_per_rank, root, full_bitmap_dev, hits_dev);
}

} // extern "C"
