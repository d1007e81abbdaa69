// capi.hip -- implementation of include/mi355_scan.h (the C ABI of libmi355scan.so).
//
// Host-side plumbing only: argument checks, device buffers, stream ordering, kernel dispatch by
// width.  All arithmetic of the path happens in the HIP kernels of kernels.hpp; there is no CPU
// implementation of any operation in this library.
#include "ctx.hpp"

#include <cstdlib>
#include <cstring>
#include <vector>

#include "dispatch.hpp"
#include "extras/gather.hpp"
#include "kernels.hpp"
#include "extras/aggregate.hpp"
#include "extras/histogram.hpp"

using namespace mi355;

namespace mi355 {

namespace {
thread_local std::string g_err;

// The default context is per THREAD: the reference's functions are stateless and re-entrant (its own
// shared_scan_128_threaded calls scan_128 from an OpenMP loop, src/simd_scan_shared.cpp:25-32), so the drop-in path
// (ctx == NULL everywhere in include/simd_scan.hpp) must be callable from several host threads at once.  Each thread
// gets its own context -- own hit-count scratch, kernel scratch, key ring, device-buffer pool -- on device 0 and the
// null stream; it is destroyed when the thread exits.
struct ThreadDefault {
    mi355_ctx *ctx = nullptr;
    ~ThreadDefault()
    {
        if (ctx) {
            mi355_ctx *c = ctx;
            ctx = nullptr;
            (void)mi355_ctx_destroy(c);
        }
    }
};
thread_local ThreadDefault t_default;
} // namespace

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

const char *last_error() { return g_err.c_str(); }

int resolve(mi355_ctx *&ctx)
{
    if (ctx) return MI355_OK;
    if (!t_default.ctx) {
        int rc = mi355_ctx_create(0, nullptr, &t_default.ctx);
        if (rc != MI355_OK) return rc;
        t_default.ctx->is_thread_default = true;
    }
    ctx = t_default.ctx;
    return MI355_OK;
}

int bind(mi355_ctx *ctx)
{
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != ctx->device) {
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess) return fail(MI355_E_HIP, "hipSetDevice(%d): %s", ctx->device, hipGetErrorString(e));
    }
    return MI355_OK;
}

int pool_get(mi355_ctx *ctx, int slot, size_t bytes, void **out)
{
    if (ctx->pool_bytes[slot] < bytes) {
        // whatever still uses the old buffer is ordered on the context's stream
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->pool[slot]) HIP_TRY(hipFree(ctx->pool[slot]));
        ctx->pool[slot] = nullptr;
        ctx->pool_bytes[slot] = 0;
        const size_t want = (bytes + (bytes >> 3) + 4095) / 4096 * 4096; // 12 % slack: sizes that creep up do not reallocate each call
        HIP_TRY(hipMalloc(&ctx->pool[slot], want));
        ctx->pool_bytes[slot] = want;
    }
    *out = ctx->pool[slot];
    return MI355_OK;
}

} // namespace mi355

namespace {

int check_width(unsigned c)
{
    if (c < 1 || c > 32) return fail(MI355_E_INVALID, "bit width c=%u outside 1..32", c);
    return MI355_OK;
}

typedef hipError_t (*group_fn)(const LaunchReq &);
const group_fn kGroups[kNumGroups] = {launch_group_0, launch_group_1, launch_group_2, launch_group_3,
                                      launch_group_4, launch_group_5, launch_group_6, launch_group_7};

// rows from which a launch uses what mi355_tune_dev measured (smaller columns take microseconds whatever the grid)
constexpr uint64_t kTuneMinRows = 50000000ull;

// one entry per kernel shape: op, width, and for the scans whether a bitmap is written and whether a mask is read
inline uint32_t tune_key(int op, unsigned c, bool writes_bitmap, bool reads_mask)
{
    return (uint32_t)op * 64u + c + (writes_bitmap ? 0u : 1u << 12) + (reads_mask ? 1u << 13 : 0u);
}

int launch(mi355_ctx *ctx, LaunchReq &r)
{
    if (int rc = bind(ctx)) return rc;
    r.stream = ctx->stream;
    r.device = ctx->device;
    r.num_cus = ctx->num_cus;
    r.max_blocks_per_cu = ctx->max_blocks_per_cu;
    r.dma_aux = ctx->dma_aux;
    r.scan_nt_stores = ctx->scan_nt_stores;
    r.shared_vpl = ctx->shared_vpl;
    r.scan_burst = ctx->scan_burst;
    // the A/B switches of the shared scans (bits 0-8) never reach select_kernel, whose switches live in bits 9-12 of the
    // option and arrive as its bits 1-4: two TIMING ablations (2 no expansion, 4 no look-back: wrong ids by construction) and
    // two A/B switches of the chunk hand-out (8 chunks dealt out by block index as in round 2, 16 no barrier per generation)
    // (option bits 13.. = further switches of the shared scans; they arrive as their bits 9..)
    r.scan.flags = r.op == kOpSelect ? ((ctx->kernel_flags >> 8) & 0x7eu) : ((ctx->kernel_flags & 0x1ffu) | ((ctx->kernel_flags >> 4) & 0xdfe00u));
    r.scan.scratch = ctx->kernel_scratch;
    if (r.max_blocks_per_cu == 0 && !ctx->tuned_bpc.empty()) {
        const bool scan = r.op == kOpScanEq || r.op == kOpScanRange;
        if ((scan && r.scan.n >= kTuneMinRows) || (r.op == kOpDecompress && r.decomp.n >= kTuneMinRows)) {
            const auto it = ctx->tuned_bpc.find(scan ? tune_key(r.op, r.c, r.scan.out != nullptr, r.scan.and_mask != nullptr)
                                                     : tune_key(r.op, r.c, true, false));
            if (it != ctx->tuned_bpc.end()) r.max_blocks_per_cu = it->second;
        }
    }
    hipError_t e = kGroups[(r.c - 1) / 4](r);
    if (e != hipSuccess) return fail(MI355_E_HIP, "kernel launch (op %d, c=%u): %s", r.op, r.c, hipGetErrorString(e));
    return MI355_OK;
}

constexpr int kKeySlots = 8;
constexpr size_t kKeySlotInts = kMaxKeys + 8;

// P > 8 keys -> device memory (padded to a multiple of 8 with copies of the last key), asynchronously on the stream
int upload_keys(mi355_ctx *ctx, const int32_t *keys_host, unsigned P, const int32_t **keys_dev)
{
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return fail(MI355_E_INVALID, "key lists longer than 8 are uploaded per call and cannot be captured into a graph");
    const int slot = ctx->key_next;
    ctx->key_next = (slot + 1) % kKeySlots;
    if (ctx->key_used[slot]) HIP_TRY(hipEventSynchronize(ctx->key_events[slot])); // the copy out of this slot is done
    int32_t *h = ctx->keys_pinned + (size_t)slot * kKeySlotInts;
    int32_t *d = ctx->keys_scratch + (size_t)slot * kKeySlotInts;
    const unsigned npad = (P + 7) / 8 * 8;
    memcpy(h, keys_host, P * sizeof(int32_t));
    for (unsigned k = P; k < npad; k++) h[k] = keys_host[P - 1];
    HIP_TRY(hipMemcpyAsync(d, h, npad * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->key_events[slot], ctx->stream));
    ctx->key_used[slot] = true;
    *keys_dev = d;
    return MI355_OK;
}

size_t bitmap_bytes(uint64_t n) { return (size_t)((n + 7) / 8); }

} // namespace

extern "C" {

const char *mi355_last_error(void) { return mi355::last_error(); }
const char *mi355_version(void) { return "mi355scan 0.2 (gfx950)"; }

int mi355_device_count(int *count)
{
    if (!count) return fail(MI355_E_INVALID, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(MI355_E_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return MI355_OK;
}

int mi355_ctx_create(int device, void *hip_stream, mi355_ctx **out)
{
    if (!out) return fail(MI355_E_INVALID, "out is null");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(MI355_E_NODEVICE, "no HIP device visible: this engine has no CPU fallback");
    if (device < 0 || device >= n) return fail(MI355_E_INVALID, "device %d out of range (0..%d)", device, n - 1);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MI355_E_NODEVICE, "device %d is %s; libmi355scan is built for gfx950 (MI355X) only", device,
                    prop.gcnArchName);
    HIP_TRY(hipSetDevice(device));
    mi355_ctx *c = new mi355_ctx;
    c->device = device;
    c->stream = (hipStream_t)hip_stream;
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char *s = getenv("MI355_MAX_BLOCKS_PER_CU")) c->max_blocks_per_cu = atoi(s);
    if (const char *s = getenv("MI355_DMA_AUX")) c->dma_aux = atoi(s);
    if (const char *s = getenv("MI355_SCAN_BURST")) c->scan_burst = atoi(s);
    if (const char *s = getenv("MI355_SHARED_VPL")) c->shared_vpl = atoi(s);
    if (const char *s = getenv("MI355_KERNEL_FLAGS")) c->kernel_flags = (unsigned)atoi(s);
    // host-pointer flavours: the kernels write the hit counts here, straight into pinned (device-visible) host memory
    hipError_t e = hipHostMalloc((void **)&c->hits_scratch, kMaxKeys * sizeof(unsigned long long), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void **)&c->kernel_scratch, kScratchWords * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(c->kernel_scratch, 0, kScratchWords * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void **)&c->keys_scratch, 8 * (kMaxKeys + 8) * sizeof(int32_t));
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->keys_pinned, 8 * (kMaxKeys + 8) * sizeof(int32_t), hipHostMallocDefault);
    for (int i = 0; i < 8 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&c->key_events[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->order_event, hipEventDisableTiming);
    if (e != hipSuccess) {
        if (c->hits_scratch) (void)hipHostFree(c->hits_scratch);
        (void)hipFree(c->kernel_scratch);
        (void)hipFree(c->keys_scratch);
        if (c->keys_pinned) (void)hipHostFree(c->keys_pinned);
        for (int i = 0; i < 8; i++)
            if (c->key_events[i]) (void)hipEventDestroy(c->key_events[i]);
        if (c->order_event) (void)hipEventDestroy(c->order_event);
        delete c;
        return fail(MI355_E_HIP, "hipMalloc(scratch): %s", hipGetErrorString(e));
    }
    *out = c;
    return MI355_OK;
}

int mi355_ctx_destroy(mi355_ctx *ctx)
{
    if (!ctx) return MI355_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->hits_scratch) (void)hipHostFree(ctx->hits_scratch);
    (void)hipFree(ctx->kernel_scratch);
    (void)hipFree(ctx->keys_scratch);
    if (ctx->keys_pinned) (void)hipHostFree(ctx->keys_pinned);
    for (int i = 0; i < 8; i++)
        if (ctx->key_events[i]) (void)hipEventDestroy(ctx->key_events[i]);
    if (ctx->order_event) (void)hipEventDestroy(ctx->order_event);
    (void)hipFree(ctx->rowid_ws);
    for (int i = 0; i < mi355_ctx::kPoolSlots; i++) (void)hipFree(ctx->pool[i]);
    delete ctx;
    return MI355_OK;
}

int mi355_ctx_synchronize(mi355_ctx *ctx)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MI355_OK;
}

int mi355_ctx_set_stream(mi355_ctx *ctx, void *hip_stream)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    hipStream_t next = (hipStream_t)hip_stream;
    if (next == ctx->stream) return MI355_OK;
    if ((rc = bind(ctx))) return rc;
    // The scratch, the key slots and the buffer pool belong to the context, not to a stream: work already enqueued on
    // the old stream must be ordered before work on the new one.  A stream that is being captured cannot take part in
    // that (and a captured graph is ordered by whoever launches it): then the caller orders the two streams.
    hipStreamCaptureStatus c0 = hipStreamCaptureStatusNone, c1 = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(ctx->stream, &c0);
    (void)hipStreamIsCapturing(next, &c1);
    if (c0 == hipStreamCaptureStatusNone && c1 == hipStreamCaptureStatusNone && hipStreamQuery(ctx->stream) != hipSuccess) {
        HIP_TRY(hipEventRecord(ctx->order_event, ctx->stream));
        HIP_TRY(hipStreamWaitEvent(next, ctx->order_event, 0));
    }
    (void)hipGetLastError(); // hipStreamQuery's hipErrorNotReady is not a failure
    ctx->stream = next;
    return MI355_OK;
}

int mi355_shard_rows(uint64_t n, unsigned world, unsigned rank, uint64_t *first, uint64_t *count)
{
    if (!first || !count || world < 1 || rank >= world) return fail(MI355_E_INVALID, "bad shard arguments");
    const uint64_t align = 8192;
    uint64_t per = (n + world - 1) / world;
    per = (per + align - 1) / align * align;
    const uint64_t a = (uint64_t)rank * per < n ? (uint64_t)rank * per : n;
    const uint64_t b = (uint64_t)(rank + 1) * per < n ? (uint64_t)(rank + 1) * per : n;
    *first = a;
    *count = b - a;
    return MI355_OK;
}

int mi355_ctx_set_option(mi355_ctx *ctx, const char *name, int value)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    if (!name) return fail(MI355_E_INVALID, "name is null");
    if (!strcmp(name, "max_blocks_per_cu"))
        ctx->max_blocks_per_cu = value;
    else if (!strcmp(name, "dma_aux"))
        ctx->dma_aux = value;
    else if (!strcmp(name, "scan_nt_stores"))
        ctx->scan_nt_stores = value;
    else if (!strcmp(name, "shared_vpl"))
        ctx->shared_vpl = value;
    else if (!strcmp(name, "select_kernel"))
        ctx->select_kernel = value;
    else if (!strcmp(name, "scan_burst"))
        ctx->scan_burst = value;
    else if (!strcmp(name, "kernel_flags"))
        ctx->kernel_flags = (unsigned)value;
    else
        return fail(MI355_E_INVALID, "unknown option %s", name);
    return MI355_OK;
}

/* ---- sizing: src/simd_scan.hpp:20-40 ---- */
size_t mi355_compressed_buffer_size(unsigned c, size_t n)
{
    size_t bits = (size_t)c * n;
    return bits / 8 + (bits % 8 != 0) + 256;
}
size_t mi355_decompression_output_buffer_size(size_t n) { return n * 4 + 32; }
size_t mi355_scan_output_buffer_size(size_t n) { return n / 8 + (n % 8 != 0) + 32; }
size_t mi355_bitmap_stride(size_t n) { return (n / 8 + (n % 8 != 0) + 255) / 256 * 256; }

/* ---- device memory ---- */
int mi355_dev_alloc(mi355_ctx *ctx, size_t bytes, void **dptr)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    if (!dptr) return fail(MI355_E_INVALID, "dptr is null");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
    return MI355_OK;
}
int mi355_dev_free(mi355_ctx *ctx, void *dptr)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    HIP_TRY(hipFree(dptr));
    return MI355_OK;
}
int mi355_dev_upload(mi355_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MI355_OK;
}
int mi355_dev_download(mi355_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MI355_OK;
}
int mi355_dev_memset(mi355_ctx *ctx, void *dst_dev, int value, size_t bytes)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    HIP_TRY(hipMemsetAsync(dst_dev, value, bytes, ctx->stream));
    return MI355_OK;
}

/* ---- pack / generate ---- */
static int pack_launch(mi355_ctx *ctx, int src, const void *values_dev, uint64_t n, uint64_t first_row, uint64_t param,
                       unsigned c, void *packed_dev)
{
    int rc = check_width(c);
    if (rc) return rc;
    if (!packed_dev) return fail(MI355_E_INVALID, "packed_dev is null");
    if (((uintptr_t)packed_dev & 3) != 0) return fail(MI355_E_INVALID, "packed_dev must be 4-byte aligned");
    if ((src == kSrcU16 || src == kSrcU32) && !values_dev && n) return fail(MI355_E_INVALID, "values_dev is null");
    if (src == kSrcMod && param == 0) return fail(MI355_E_INVALID, "modulus 0");
    if (int brc = bind(ctx)) return brc;
    PackArgs a;
    a.values = values_dev;
    a.n = n;
    a.first_row = first_row;
    a.param = param;
    a.out = (uint32_t *)packed_dev;
    a.out_dwords = mi355_compressed_buffer_size(c, n) / 4; // payload + pad, whole dwords
    a.c = c;
    uint64_t blocks = (a.out_dwords + 255) / 256;
    uint64_t cap = (uint64_t)ctx->num_cus * 8;
    unsigned grid = (unsigned)(blocks < cap ? (blocks ? blocks : 1) : cap);
    switch (src) {
#define PACK_BY_WIDTH(SRC)                                                                                          \
    do { /* values per output dword: at most floor(31/c) + 2; one block per 8192-value tile, 4 resident per CU */  \
        uint64_t tiles = (n + kPackTile - 1) / kPackTile;                                                           \
        uint64_t tcap = (uint64_t)ctx->num_cus * 4;                                                                 \
        unsigned tgrid = (unsigned)(tiles < tcap ? (tiles ? tiles : 1) : tcap);                                     \
        if (c >= 16) hipLaunchKernelGGL((pack_tiled_kernel<SRC, 3>), dim3(tgrid), dim3(256), 0, ctx->stream, a);    \
        else if (c >= 8) hipLaunchKernelGGL((pack_tiled_kernel<SRC, 5>), dim3(tgrid), dim3(256), 0, ctx->stream, a); \
        else if (c >= 4) hipLaunchKernelGGL((pack_tiled_kernel<SRC, 9>), dim3(tgrid), dim3(256), 0, ctx->stream, a); \
        else if (c >= 2) hipLaunchKernelGGL((pack_tiled_kernel<SRC, 17>), dim3(tgrid), dim3(256), 0, ctx->stream, a); \
        else hipLaunchKernelGGL((pack_tiled_kernel<SRC, 32>), dim3(tgrid), dim3(256), 0, ctx->stream, a);           \
    } while (0)
    case kSrcU16: PACK_BY_WIDTH(kSrcU16); break;
    case kSrcU32: PACK_BY_WIDTH(kSrcU32); break;
#undef PACK_BY_WIDTH
    case kSrcMod: hipLaunchKernelGGL(pack_kernel<kSrcMod>, dim3(grid), dim3(256), 0, ctx->stream, a); break;
    case kSrcSplitmix: hipLaunchKernelGGL(pack_kernel<kSrcSplitmix>, dim3(grid), dim3(256), 0, ctx->stream, a); break;
    case kSrcIndex: hipLaunchKernelGGL(pack_kernel<kSrcIndex>, dim3(grid), dim3(256), 0, ctx->stream, a); break;
    default: return fail(MI355_E_INVALID, "unknown pack source %d", src);
    }
    HIP_TRY(hipGetLastError());
    // the trailing (compressed_buffer_size % 4) pad bytes, if any
    size_t total = mi355_compressed_buffer_size(c, n);
    if (total % 4) HIP_TRY(hipMemsetAsync((uint8_t *)packed_dev + total / 4 * 4, 0, total % 4, ctx->stream));
    return MI355_OK;
}

int mi355_pack_u16_dev(mi355_ctx *ctx, const uint16_t *values_dev, uint64_t n, unsigned c, void *packed_dev)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    return pack_launch(ctx, kSrcU16, values_dev, n, 0, 0, c, packed_dev);
}
int mi355_pack_u32_dev(mi355_ctx *ctx, const uint32_t *values_dev, uint64_t n, unsigned c, void *packed_dev)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    return pack_launch(ctx, kSrcU32, values_dev, n, 0, 0, c, packed_dev);
}
int mi355_generate_dev(mi355_ctx *ctx, int kind, uint64_t first_row, uint64_t n, unsigned c, uint64_t param,
                       void *packed_dev)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    int src = kind == MI355_GEN_MOD ? kSrcMod : kind == MI355_GEN_SPLITMIX ? kSrcSplitmix : kind == MI355_GEN_INDEX ? kSrcIndex : -1;
    if (src < 0) return fail(MI355_E_INVALID, "unknown generator kind %d", kind);
    return pack_launch(ctx, src, nullptr, n, first_row, param, c, packed_dev);
}

static int pack_host(mi355_ctx *ctx, int src, const void *values, size_t elem, uint64_t n, unsigned c, void *packed_host)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    if ((rc = check_width(c))) return rc;
    if (!packed_host || (!values && n)) return fail(MI355_E_INVALID, "null pointer");
    if ((rc = bind(ctx))) return rc;
    void *dv = nullptr, *dp = nullptr;
    size_t pbytes = mi355_compressed_buffer_size(c, n);
    if ((rc = pool_get(ctx, mi355_ctx::kPoolIn, n * elem + 16, &dv))) return rc;
    if ((rc = pool_get(ctx, mi355_ctx::kPoolOut, pbytes + 16, &dp))) return rc;
    HIP_TRY(hipMemcpyAsync(dv, values, n * elem, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = pack_launch(ctx, src, dv, n, 0, 0, c, dp))) return rc;
    HIP_TRY(hipMemcpyAsync(packed_host, dp, pbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MI355_OK;
}
int mi355_pack_u16(mi355_ctx *ctx, const uint16_t *values, uint64_t n, unsigned c, void *packed_host)
{
    return pack_host(ctx, kSrcU16, values, 2, n, c, packed_host);
}
int mi355_pack_u32(mi355_ctx *ctx, const uint32_t *values, uint64_t n, unsigned c, void *packed_host)
{
    return pack_host(ctx, kSrcU32, values, 4, n, c, packed_host);
}

/* ---- decompress ---- */
int mi355_decompress_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, int32_t *out_dev)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    if ((rc = check_width(c))) return rc;
    if (n == 0) return MI355_OK;
    if (!packed_dev || !out_dev) return fail(MI355_E_INVALID, "null device pointer");
    if (((uintptr_t)packed_dev & 15) != 0) return fail(MI355_E_INVALID, "packed_dev must be 16-byte aligned");
    if (((uintptr_t)out_dev & 15) != 0) return fail(MI355_E_INVALID, "out_dev must be 16-byte aligned");
    LaunchReq r{};
    r.op = kOpDecompress;
    r.c = c;
    r.decomp.packed = (const uint8_t *)packed_dev;
    r.decomp.n = n;
    r.decomp.out = out_dev;
    return launch(ctx, r);
}

/* ---- scans (device pointers) ---- */
static int scan_common_dev(mi355_ctx *ctx, int op, const void *packed_dev, uint64_t n, unsigned c, uint32_t k0,
                           uint32_t k1, void *bitmap_dev, uint64_t *hits_dev)
{
    int rc = check_width(c);
    if (rc) return rc;
    if (n == 0) {
        if (hits_dev) HIP_TRY(hipMemsetAsync(hits_dev, 0, sizeof(uint64_t), ctx->stream));
        return MI355_OK;
    }
    if (!packed_dev || !bitmap_dev) return fail(MI355_E_INVALID, "null device pointer");
    if (((uintptr_t)packed_dev & 15) != 0) return fail(MI355_E_INVALID, "packed_dev must be 16-byte aligned");
    if (((uintptr_t)bitmap_dev & 15) != 0) return fail(MI355_E_INVALID, "bitmap_dev must be 16-byte aligned");
    LaunchReq r{};
    r.op = op;
    r.c = c;
    r.scan.packed = (const uint8_t *)packed_dev;
    r.scan.n = n;
    r.scan.out = (uint8_t *)bitmap_dev;
    r.scan.hits = (unsigned long long *)hits_dev;
    r.scan.key[0] = k0;
    r.scan.key[1] = k1;
    r.scan.nkeys = 1;
    return launch(ctx, r);
}

int mi355_scan_eq_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, int32_t key, void *bitmap_dev,
                      uint64_t *hits_dev)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    return scan_common_dev(ctx, kOpScanEq, packed_dev, n, c, (uint32_t)key, 0, bitmap_dev, hits_dev);
}

int mi355_scan_range_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, uint32_t lo, uint32_t hi,
                         void *bitmap_dev, uint64_t *hits_dev)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    if (lo > hi) {
        // empty range: all-zero bitmap, zero hits
        if ((rc = check_width(c))) return rc;
        if (hits_dev) HIP_TRY(hipMemsetAsync(hits_dev, 0, sizeof(uint64_t), ctx->stream));
        if (n && !bitmap_dev) return fail(MI355_E_INVALID, "null device pointer");
        if (n) HIP_TRY(hipMemsetAsync(bitmap_dev, 0, bitmap_bytes(n), ctx->stream));
        return MI355_OK;
    }
    return scan_common_dev(ctx, kOpScanRange, packed_dev, n, c, lo, hi - lo, bitmap_dev, hits_dev);
}

int mi355_shared_scan_eq_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, const int32_t *keys_host,
                             unsigned P, int layout, void *out_dev, uint64_t stride_bytes, uint64_t *hits_dev)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    if ((rc = check_width(c))) return rc;
    if (P < 1 || P > (unsigned)kMaxKeys) return fail(MI355_E_INVALID, "P=%u outside 1..%u", P, kMaxKeys);
    if (!keys_host) return fail(MI355_E_INVALID, "keys is null");
    if (layout != MI355_LAYOUT_PER_PREDICATE && layout != MI355_LAYOUT_LINEAR)
        return fail(MI355_E_INVALID, "unknown layout %d", layout);
    if (n == 0) {
        if (hits_dev) HIP_TRY(hipMemsetAsync(hits_dev, 0, P * sizeof(uint64_t), ctx->stream));
        return MI355_OK;
    }
    if (!packed_dev || !out_dev) return fail(MI355_E_INVALID, "null device pointer");
    if (((uintptr_t)packed_dev & 15) != 0) return fail(MI355_E_INVALID, "packed_dev must be 16-byte aligned");
    if (layout == MI355_LAYOUT_PER_PREDICATE) {
        if (((uintptr_t)out_dev & 15) != 0 || (stride_bytes & 15) != 0)
            return fail(MI355_E_INVALID, "out_dev and stride_bytes must be multiples of 16");
        if (stride_bytes < bitmap_bytes(n)) return fail(MI355_E_INVALID, "stride_bytes smaller than ceil(n/8)");
    } else if (((uintptr_t)out_dev & 15) != 0) {
        return fail(MI355_E_INVALID, "out_dev must be 16-byte aligned");
    }
    // one predicate: both layouts are the plain bitmap of that key -- the equality scan kernel does it at twice the speed
    if (P == 1) return scan_common_dev(ctx, kOpScanEq, packed_dev, n, c, (uint32_t)keys_host[0], 0, out_dev, hits_dev);
    LaunchReq r{};
    r.op = kOpSharedScan;
    r.c = c;
    r.scan.packed = (const uint8_t *)packed_dev;
    r.scan.n = n;
    r.scan.out = (uint8_t *)out_dev;
    r.scan.out_stride = stride_bytes;
    r.scan.hits = (unsigned long long *)hits_dev;
    r.scan.nkeys = P;
    r.scan.layout = (uint32_t)layout;
    if (P <= (unsigned)kMaxKeysPerPass) {
        for (unsigned q = 0; q < (unsigned)kMaxKeysPerPass; q++) r.scan.key[q] = (uint32_t)keys_host[q < P ? q : P - 1];
    } else {
        if ((rc = bind(ctx))) return rc;
        if ((rc = upload_keys(ctx, keys_host, P, &r.scan.keys_dev))) return rc;
    }
    return launch(ctx, r);
}

/* ---- predicates and bitmap consumers beyond the reference ---- */
// every comparison is an inclusive range [lo, hi] over the column's domain [0, 2^c), possibly negated: fills
// key[0] = lo, key[1] = hi - lo and the negation word of a scan request
static void fill_predicate(ScanArgs &sa, unsigned c, int op, int64_t a, int64_t b)
{
    const int64_t vmax = c == 32 ? 0xffffffffll : ((1ll << c) - 1);
    int64_t lo = 0, hi = vmax;
    bool invert = false, empty = false;
    switch (op) {
    case MI355_CMP_EQ: lo = hi = a; break;
    case MI355_CMP_NE: lo = hi = a; invert = true; break;
    case MI355_CMP_LT: hi = a - 1; break;
    case MI355_CMP_LE: hi = a; break;
    case MI355_CMP_GT: lo = a + 1; break;
    case MI355_CMP_GE: lo = a; break;
    case MI355_CMP_BETWEEN: lo = a; hi = b; break;
    case MI355_CMP_NOT_BETWEEN: lo = a; hi = b; invert = true; break;
    }
    if (lo < 0) lo = 0;
    if (hi > vmax) hi = vmax;
    if (lo > hi) empty = true; // matches nothing (or, negated, everything)
    sa.invert = invert ? 0xffffffffu : 0u;
    if (empty) { // lo above every value: t = x - lo is never <= span 0 unless x == 0xffffffff, which needs c == 32 ...
        sa.key[0] = 0xffffffffu;
        sa.key[1] = 0;
        if (c == 32) { // ... so fold the empty case into the negation flag on the full range
            sa.key[0] = 0;
            sa.key[1] = 0xffffffffu;
            sa.invert = invert ? 0u : 0xffffffffu;
        }
    } else {
        sa.key[0] = (uint32_t)lo;
        sa.key[1] = (uint32_t)(hi - lo);
    }
}

int mi355_scan_combine_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, int op, int64_t a, int64_t b,
                           int mask_op, const void *mask_dev, void *bitmap_dev, uint64_t *hits_dev)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    if ((rc = check_width(c))) return rc;
    if (op < MI355_CMP_EQ || op > MI355_CMP_NOT_BETWEEN) return fail(MI355_E_INVALID, "unknown comparison %d", op);
    if (mask_op < MI355_BITMAP_AND || mask_op > MI355_BITMAP_ANDNOT) return fail(MI355_E_INVALID, "unknown mask op %d", mask_op);
    if (!bitmap_dev && !hits_dev) return fail(MI355_E_INVALID, "bitmap_dev and hits_dev are both null: nothing to compute");
    if (n == 0) {
        if (hits_dev) HIP_TRY(hipMemsetAsync(hits_dev, 0, sizeof(uint64_t), ctx->stream));
        return MI355_OK;
    }
    if (!packed_dev) return fail(MI355_E_INVALID, "null device pointer");
    if (((uintptr_t)packed_dev & 15) || ((uintptr_t)bitmap_dev & 15) || ((uintptr_t)mask_dev & 15))
        return fail(MI355_E_INVALID, "packed_dev, bitmap_dev and mask_dev must be 16-byte aligned");
    LaunchReq r{};
    r.op = kOpScanRange;
    r.c = c;
    r.scan.packed = (const uint8_t *)packed_dev;
    r.scan.n = n;
    r.scan.out = (uint8_t *)bitmap_dev;
    r.scan.hits = (unsigned long long *)hits_dev;
    r.scan.nkeys = 1;
    r.scan.and_mask = (const uint8_t *)mask_dev;
    r.scan.mask_op = (uint32_t)mask_op;
    fill_predicate(r.scan, c, op, a, b);
    return launch(ctx, r);
}

int mi355_scan_select_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, int op, int64_t a, int64_t b, int mask_op,
                          const void *mask_dev, uint64_t first_row, uint64_t *rowids_dev, uint64_t capacity, uint64_t *count_dev)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    if ((rc = check_width(c))) return rc;
    if (op < MI355_CMP_EQ || op > MI355_CMP_NOT_BETWEEN) return fail(MI355_E_INVALID, "unknown comparison %d", op);
    if (mask_op < MI355_BITMAP_AND || mask_op > MI355_BITMAP_ANDNOT) return fail(MI355_E_INVALID, "unknown mask op %d", mask_op);
    if (!count_dev) return fail(MI355_E_INVALID, "count_dev is null");
    if ((rc = bind(ctx))) return rc;
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(count_dev, 0, sizeof(uint64_t), ctx->stream));
        return MI355_OK;
    }
    if (!packed_dev || (!rowids_dev && capacity)) return fail(MI355_E_INVALID, "null device pointer");
    if (((uintptr_t)packed_dev & 15) || ((uintptr_t)mask_dev & 15))
        return fail(MI355_E_INVALID, "packed_dev and mask_dev must be 16-byte aligned");
    // one state word per chunk of tiles (decoupled look-back), zeroed in front of the launch
    const uint64_t tile_values = 64 * (uint64_t)scan_vpl((int)c, kModeRange);
    const uint64_t ntiles = (n + tile_values - 1) / tile_values;
    const uint64_t nchunks = (ntiles + select_tiles((int)c) - 1) / select_tiles((int)c);
    // + the chunk-ticket counter on its own line behind them (select_state_words); the same memset zeroes both
    const uint64_t nwords = select_state_words(nchunks);
    if (ctx->rowid_ws_entries < nwords) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->rowid_ws) HIP_TRY(hipFree(ctx->rowid_ws));
        ctx->rowid_ws = nullptr;
        ctx->rowid_ws_entries = 0;
        HIP_TRY(hipMalloc((void **)&ctx->rowid_ws, nwords * sizeof(unsigned long long)));
        ctx->rowid_ws_entries = nwords;
    }
    HIP_TRY(hipMemsetAsync(ctx->rowid_ws, 0, nwords * sizeof(unsigned long long), ctx->stream));
    // the count is written by atomic max (the last chunk's total, or ~0 from a wave that gave up): start it at 0
    HIP_TRY(hipMemsetAsync(count_dev, 0, sizeof(uint64_t), ctx->stream));
    LaunchReq r{};
    r.op = kOpSelect;
    r.c = c;
    r.scan.packed = (const uint8_t *)packed_dev;
    r.scan.n = n;
    r.scan.hits = (unsigned long long *)count_dev;
    r.scan.nkeys = 1;
    r.scan.and_mask = (const uint8_t *)mask_dev;
    r.scan.mask_op = (uint32_t)mask_op;
    r.scan.tile_state = ctx->rowid_ws;
    r.scan.rowids = rowids_dev;
    r.scan.capacity = capacity;
    r.scan.first_row = first_row;
    fill_predicate(r.scan, c, op, a, b);
    // Which kernel: select2_kernel (decoder + expander waves, one look-back per block and generation) is ahead of the
    // single-role select_kernel at every width and selectivity measured (profiles/r03_select_widths.txt: 1.01 - 1.2 x when
    // almost nothing qualifies, 1.2 - 3.3 x from 1/64 up), so it is what runs; option "select_kernel" = 1 keeps the older kernel
    // reachable for A/B runs (0 / 2: select2_kernel).
    r.select_single = ctx->select_kernel == 1;
    return launch(ctx, r);
}

int mi355_scan2_dev(mi355_ctx *ctx, const void *packed1_dev, unsigned c1, int op1, int64_t a1, int64_t b1, const void *packed2_dev,
                    unsigned c2, int op2, int64_t a2, int64_t b2, uint64_t n, int combine_op, void *bitmap_dev, uint64_t *hits_dev)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    if ((rc = check_width(c1)) || (rc = check_width(c2))) return rc;
    if (op1 < MI355_CMP_EQ || op1 > MI355_CMP_NOT_BETWEEN || op2 < MI355_CMP_EQ || op2 > MI355_CMP_NOT_BETWEEN)
        return fail(MI355_E_INVALID, "unknown comparison");
    if (combine_op < MI355_BITMAP_AND || combine_op > MI355_BITMAP_ANDNOT) return fail(MI355_E_INVALID, "unknown combine op %d", combine_op);
    if (!bitmap_dev && !hits_dev) return fail(MI355_E_INVALID, "bitmap_dev and hits_dev are both null: nothing to compute");
    if (n == 0) {
        if (hits_dev) HIP_TRY(hipMemsetAsync(hits_dev, 0, sizeof(uint64_t), ctx->stream));
        return MI355_OK;
    }
    if (!packed1_dev || !packed2_dev) return fail(MI355_E_INVALID, "null device pointer");
    if (((uintptr_t)packed1_dev & 15) || ((uintptr_t)packed2_dev & 15) || ((uintptr_t)bitmap_dev & 15))
        return fail(MI355_E_INVALID, "packed columns and bitmap_dev must be 16-byte aligned");
    if (c1 != c2) {
        // columns of different widths have different tile geometries: two launches, the first predicate's bitmap
        // combined inside the second scan, in place (every wave reads the mask bytes of a tile before it stores them)
        void *tmp = bitmap_dev;
        if (!tmp && (rc = pool_get(ctx, mi355_ctx::kPoolAux, bitmap_bytes(n) + 16, &tmp))) return rc;
        if ((rc = mi355_scan_combine_dev(ctx, packed1_dev, n, c1, op1, a1, b1, MI355_BITMAP_AND, nullptr, tmp, nullptr))) return rc;
        return mi355_scan_combine_dev(ctx, packed2_dev, n, c2, op2, a2, b2, combine_op, tmp, bitmap_dev, hits_dev);
    }
    LaunchReq r{};
    r.op = kOpScan2;
    r.c = c1;
    r.scan.packed = (const uint8_t *)packed1_dev;
    r.scan.packed2 = (const uint8_t *)packed2_dev;
    r.scan.n = n;
    r.scan.out = (uint8_t *)bitmap_dev;
    r.scan.hits = (unsigned long long *)hits_dev;
    r.scan.nkeys = 1;
    r.scan.mask_op = (uint32_t)combine_op;
    ScanArgs second{};
    fill_predicate(second, c2, op2, a2, b2);
    fill_predicate(r.scan, c1, op1, a1, b1);
    r.scan.key2[0] = second.key[0];
    r.scan.key2[1] = second.key[1];
    r.scan.invert2 = second.invert;
    return launch(ctx, r);
}

int mi355_scan_where_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, int op, int64_t a, int64_t b,
                         const void *and_mask_dev, void *bitmap_dev, uint64_t *hits_dev)
{
    return mi355_scan_combine_dev(ctx, packed_dev, n, c, op, a, b, MI355_BITMAP_AND, and_mask_dev, bitmap_dev, hits_dev);
}

int mi355_scan_in_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, const int32_t *keys_host, unsigned P,
                      int negate, const void *and_mask_dev, void *bitmap_dev, uint64_t *hits_dev)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    if ((rc = check_width(c))) return rc;
    if (P < 1 || P > (unsigned)kMaxKeys) return fail(MI355_E_INVALID, "P=%u outside 1..%u", P, kMaxKeys);
    if (!keys_host) return fail(MI355_E_INVALID, "keys is null");
    if (n == 0) {
        if (hits_dev) HIP_TRY(hipMemsetAsync(hits_dev, 0, sizeof(uint64_t), ctx->stream));
        return MI355_OK;
    }
    if (!packed_dev || !bitmap_dev) return fail(MI355_E_INVALID, "null device pointer");
    if (((uintptr_t)packed_dev & 15) || ((uintptr_t)bitmap_dev & 15) || ((uintptr_t)and_mask_dev & 15))
        return fail(MI355_E_INVALID, "packed_dev, bitmap_dev and and_mask_dev must be 16-byte aligned");
    if ((rc = bind(ctx))) return rc;
    const int32_t *keys_dev = nullptr;
    if ((rc = upload_keys(ctx, keys_host, P, &keys_dev))) return rc;
    LaunchReq r{};
    r.op = kOpScanIn;
    r.c = c;
    r.scan.packed = (const uint8_t *)packed_dev;
    r.scan.n = n;
    r.scan.out = (uint8_t *)bitmap_dev;
    r.scan.hits = (unsigned long long *)hits_dev;
    r.scan.keys_dev = keys_dev;
    r.scan.nkeys = P;
    r.scan.and_mask = (const uint8_t *)and_mask_dev;
    r.scan.invert = negate ? 0xffffffffu : 0u;
    return launch(ctx, r);
}

static int bitmap_launch(mi355_ctx *ctx, int op, const void *a, const void *b, void *out, uint64_t n, uint64_t *count_dev)
{
    if (int brc = bind(ctx)) return brc;
    if (n == 0) {
        if (count_dev) HIP_TRY(hipMemsetAsync(count_dev, 0, sizeof(uint64_t), ctx->stream));
        return MI355_OK;
    }
    if (!a || (op != kBitCount && (!b || !out))) return fail(MI355_E_INVALID, "null device pointer");
    if (((uintptr_t)a & 15) || ((uintptr_t)b & 15) || ((uintptr_t)out & 15))
        return fail(MI355_E_INVALID, "bitmaps must be 16-byte aligned");
    BitmapArgs g;
    g.a = (const uint8_t *)a;
    g.b = (const uint8_t *)b;
    g.out = (uint8_t *)out;
    g.nbytes = bitmap_bytes(n);
    // partial counts go to the (all-zero) hit-count replicas of the context scratch, then to count_dev
    g.count = count_dev ? ctx->kernel_scratch : nullptr;
    uint64_t blocks = (g.nbytes / 16 + 255) / 256;
    unsigned grid = (unsigned)(blocks < (uint64_t)ctx->num_cus * 4 ? (blocks ? blocks : 1) : (uint64_t)ctx->num_cus * 4);
    switch (op) {
    case kBitAnd: hipLaunchKernelGGL(bitmap_kernel<kBitAnd>, dim3(grid), dim3(256), 0, ctx->stream, g); break;
    case kBitOr: hipLaunchKernelGGL(bitmap_kernel<kBitOr>, dim3(grid), dim3(256), 0, ctx->stream, g); break;
    case kBitXor: hipLaunchKernelGGL(bitmap_kernel<kBitXor>, dim3(grid), dim3(256), 0, ctx->stream, g); break;
    case kBitAndNot: hipLaunchKernelGGL(bitmap_kernel<kBitAndNot>, dim3(grid), dim3(256), 0, ctx->stream, g); break;
    case kBitCount: hipLaunchKernelGGL(bitmap_kernel<kBitCount>, dim3(grid), dim3(256), 0, ctx->stream, g); break;
    default: return fail(MI355_E_INVALID, "unknown bitmap op %d", op);
    }
    if (count_dev)
        hipLaunchKernelGGL(sum_slots_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->kernel_scratch, (unsigned long long *)count_dev);
    HIP_TRY(hipGetLastError());
    return MI355_OK;
}

int mi355_bitmap_combine_dev(mi355_ctx *ctx, int op, const void *a_dev, const void *b_dev, void *out_dev, uint64_t n,
                             uint64_t *count_dev)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    if (op < MI355_BITMAP_AND || op > MI355_BITMAP_ANDNOT) return fail(MI355_E_INVALID, "unknown bitmap op %d", op);
    return bitmap_launch(ctx, op, a_dev, b_dev, out_dev, n, count_dev);
}

int mi355_bitmap_count_dev(mi355_ctx *ctx, const void *bitmap_dev, uint64_t n, uint64_t *count_dev)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    if (!count_dev) return fail(MI355_E_INVALID, "count_dev is null");
    return bitmap_launch(ctx, kBitCount, bitmap_dev, nullptr, nullptr, n, count_dev);
}

int mi355_bitmap_to_rowids_dev(mi355_ctx *ctx, const void *bitmap_dev, uint64_t n, uint64_t first_row, uint64_t *rowids_dev,
                               uint64_t capacity, uint64_t *count_dev)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    if ((rc = bind(ctx))) return rc;
    if (!count_dev) return fail(MI355_E_INVALID, "count_dev is null");
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(count_dev, 0, sizeof(uint64_t), ctx->stream));
        return MI355_OK;
    }
    if (!bitmap_dev || (!rowids_dev && capacity)) return fail(MI355_E_INVALID, "null device pointer");
    if ((uintptr_t)bitmap_dev & 3) return fail(MI355_E_INVALID, "bitmap_dev must be 4-byte aligned");
    RowidArgs g;
    g.bitmap = (const uint8_t *)bitmap_dev;
    g.nbytes = bitmap_bytes(n);
    g.first_row = first_row;
    g.nchunks = (g.nbytes + kRowidChunk - 1) / kRowidChunk;
    const uint64_t ngroups = (g.nchunks + kRowidScanGroup - 1) / kRowidScanGroup;
    const uint64_t ws_entries = g.nchunks + 1 + ngroups; // chunk counts, the total, one total per scan group
    if (ctx->rowid_ws_entries < ws_entries) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->rowid_ws) HIP_TRY(hipFree(ctx->rowid_ws));
        ctx->rowid_ws = nullptr;
        ctx->rowid_ws_entries = 0;
        HIP_TRY(hipMalloc((void **)&ctx->rowid_ws, ws_entries * sizeof(unsigned long long)));
        ctx->rowid_ws_entries = ws_entries;
    }
    g.chunk_counts = ctx->rowid_ws;
    g.rowids = rowids_dev;
    g.capacity = capacity;
    uint64_t blocks = (g.nchunks + 3) / 4;
    unsigned grid = (unsigned)(blocks < (uint64_t)ctx->num_cus * 8 ? blocks : (uint64_t)ctx->num_cus * 8);
    hipLaunchKernelGGL(rowid_count_kernel, dim3(grid), dim3(256), 0, ctx->stream, g);
    hipLaunchKernelGGL(rowid_scan_kernel, dim3((unsigned)ngroups), dim3(256), 0, ctx->stream, g);
    hipLaunchKernelGGL(rowid_write_kernel, dim3(grid), dim3(256), 0, ctx->stream, g);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(count_dev, g.chunk_counts + g.nchunks, sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
    return MI355_OK;
}

int mi355_gather_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, uint64_t first_row, const uint64_t *rowids_dev,
                     const uint64_t *count_dev, uint64_t capacity, int32_t *out_dev)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    if ((rc = check_width(c))) return rc;
    if ((rc = bind(ctx))) return rc;
    if (!count_dev) return fail(MI355_E_INVALID, "count_dev is null (the number of ids is read on the device)");
    if (capacity == 0) return MI355_OK;
    if (!packed_dev || !rowids_dev || !out_dev) return fail(MI355_E_INVALID, "null device pointer");
    if ((uintptr_t)packed_dev & 3) return fail(MI355_E_INVALID, "packed_dev must be 4-byte aligned");
    GatherArgs g;
    g.packed = (const uint8_t *)packed_dev;
    g.n = n;
    g.c = c;
    g.first_row = first_row;
    g.rowids = rowids_dev;
    g.count_dev = count_dev;
    g.capacity = capacity;
    g.out = out_dev;
    // the grid is sized for `capacity` (the count is only known on the device); idle blocks leave at once
    const uint64_t blocks = (capacity + 255) / 256;
    const unsigned grid = (unsigned)(blocks < (uint64_t)ctx->num_cus * 16 ? blocks : (uint64_t)ctx->num_cus * 16);
    hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(256), 0, ctx->stream, g);
    HIP_TRY(hipGetLastError());
    return MI355_OK;
}

int mi355_aggregate_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, const void *mask_dev, uint64_t *out_dev)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    if ((rc = check_width(c))) return rc;
    if ((rc = bind(ctx))) return rc;
    if (!out_dev) return fail(MI355_E_INVALID, "out_dev is null");
    if (n && !packed_dev) return fail(MI355_E_INVALID, "null device pointer");
    if (((uintptr_t)packed_dev & 15) || ((uintptr_t)mask_dev & 3)) return fail(MI355_E_INVALID, "packed_dev must be 16-byte, mask_dev 4-byte aligned");
    AggArgs a;
    a.packed = (const uint8_t *)packed_dev;
    a.n = n;
    a.mask = (const uint8_t *)mask_dev;
    a.out = (unsigned long long *)out_dev;
    if (n == 0) {
        hipLaunchKernelGGL(aggregate_init_kernel, dim3(1), dim3(1), 0, ctx->stream, a.out);
    } else if (!launch_aggregate_width(c, a, ctx->num_cus, ctx->stream)) {
        return fail(MI355_E_INVALID, "width %u", c);
    }
    HIP_TRY(hipGetLastError());
    return MI355_OK;
}

int mi355_histogram_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, const void *mask_dev, uint64_t *counts_dev)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    if ((rc = check_width(c))) return rc;
    if (c > (unsigned)kHistogramMaxBits) return fail(MI355_E_INVALID, "histogram: widths up to %d bits (2^c counters in LDS), got %u", kHistogramMaxBits, c);
    if ((rc = bind(ctx))) return rc;
    if (!counts_dev) return fail(MI355_E_INVALID, "counts_dev is null");
    if (n && !packed_dev) return fail(MI355_E_INVALID, "null device pointer");
    if (((uintptr_t)packed_dev & 15) || ((uintptr_t)mask_dev & 3)) return fail(MI355_E_INVALID, "packed_dev must be 16-byte, mask_dev 4-byte aligned");
    HIP_TRY(hipMemsetAsync(counts_dev, 0, sizeof(uint64_t) << c, ctx->stream));
    if (n == 0) return MI355_OK;
    HistArgs a;
    a.packed = (const uint8_t *)packed_dev;
    a.n = n;
    a.mask = (const uint8_t *)mask_dev;
    a.out = (unsigned long long *)counts_dev;
    if (!launch_histogram_width(c, a, ctx->num_cus, ctx->stream)) return fail(MI355_E_INVALID, "width %u", c);
    HIP_TRY(hipGetLastError());
    return MI355_OK;
}

/* ---- host-pointer (copying, synchronous) flavours: the drop-in path ----
 * Device buffers come from the context's grow-only pool (no hipMalloc / hipFree per call).  The reference's callers
 * hand over compressed_buffer_size(c, n) bytes, pad included (src/simd_scan.hpp:20-26); only the payload travels, the
 * pad the kernels may touch (<= 15 bytes past the payload) is zeroed on the device. */
static int upload_packed(mi355_ctx *ctx, const void *packed_host, uint64_t n, unsigned c, void **dp)
{
    const size_t payload = (size_t)((n * c + 7) / 8);
    int rc = pool_get(ctx, mi355_ctx::kPoolIn, payload + 256, dp);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(*dp, packed_host, payload, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync((uint8_t *)*dp + payload, 0, 16, ctx->stream));
    return MI355_OK;
}

// ---- load-time tuning ------------------------------------------------------------------------------------------
// The resident blocks per CU at which the streaming kernels run fastest differ between MI355X boxes for the SAME
// binary (equality scan, c = 9: two blocks 3-5 % ahead of one on two boxes, 5 % behind on a third; decompress:
// profiles/r02_decompress_bpc_sweep.txt), so a static default leaves a few per cent behind somewhere.  This call
// measures 1 / 2 / 4 blocks per CU on the caller's own column, back to back as a query stream would issue them
// (isolated launches between event pairs rank the candidates differently), and keeps the winners in the context.
int mi355_tune_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, unsigned what)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    if ((rc = check_width(c))) return rc;
    if (!packed_dev || ((uintptr_t)packed_dev & 15)) return fail(MI355_E_INVALID, "packed_dev must be a 16-byte aligned device pointer");
    if (what == 0 || (what & ~(unsigned)MI355_TUNE_ALL)) return fail(MI355_E_INVALID, "what=%u: a mask of MI355_TUNE_* bits", what);
    if (n < kTuneMinRows) return MI355_OK; // nothing to learn: such launches never consult the table
    if ((rc = bind(ctx))) return rc;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)
        return fail(MI355_E_INVALID, "mi355_tune_dev synchronises: not while the stream is being captured");
    const size_t stride = mi355_bitmap_stride(n);
    void *bitmaps = nullptr, *values = nullptr;
    if ((rc = pool_get(ctx, mi355_ctx::kPoolOut, 2 * stride, &bitmaps))) return rc;
    uint8_t *bitmap = (uint8_t *)bitmaps, *mask = bitmap + stride;
    const uint64_t decomp_rows = n < (1ull << 28) ? n : (1ull << 28); // 1 GiB of output is plenty to rank the grids
    if ((what & MI355_TUNE_DECOMPRESS) && (rc = pool_get(ctx, mi355_ctx::kPoolAux, decomp_rows * 4 + 64, &values))) return rc;
    uint64_t *hits = (uint64_t *)ctx->hits_scratch;
    const uint32_t top = c >= 32 ? 0xffffffffu : (1u << c) - 1;
    const uint32_t lo = top / 4, hi = top / 2;
    HIP_TRY(hipMemsetAsync(mask, 0x5a, stride, ctx->stream));
    struct Shape {
        unsigned bit;
        int op;
        bool bitmap, mask;
    };
    // (mi355_scan_combine_dev and everything built on it -- count-only, fused masks, comparisons -- run the range kernel)
    const Shape shapes[] = {{MI355_TUNE_SCAN, kOpScanEq, true, false},
                            {MI355_TUNE_SCAN, kOpScanRange, true, false},
                            {MI355_TUNE_COUNT, kOpScanRange, false, false},
                            {MI355_TUNE_MASK, kOpScanRange, true, true},
                            {MI355_TUNE_DECOMPRESS, kOpDecompress, true, false}};
    const int cands[3] = {1, 2, 4};
    constexpr int kRounds = 3, kBurst = 6;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) {
        (void)hipEventDestroy(e0);
        return fail(MI355_E_HIP, "hipEventCreate failed");
    }
    const int saved = ctx->max_blocks_per_cu;
    for (const Shape &sh : shapes) {
        if (!(what & sh.bit)) continue;
        auto once = [&]() -> int {
            if (sh.op == kOpDecompress) return mi355_decompress_dev(ctx, packed_dev, decomp_rows, c, (int32_t *)values);
            if (sh.op == kOpScanEq) return mi355_scan_eq_dev(ctx, packed_dev, n, c, (int32_t)lo, bitmap, hits);
            return mi355_scan_combine_dev(ctx, packed_dev, n, c, MI355_CMP_BETWEEN, lo, hi, MI355_BITMAP_AND, sh.mask ? mask : nullptr,
                                          sh.bitmap ? bitmap : nullptr, hits);
        };
        float best[3] = {0, 0, 0};
        for (int round = 0; round < kRounds && rc == MI355_OK; round++)
            for (int k = 0; k < 3 && rc == MI355_OK; k++) {
                ctx->max_blocks_per_cu = cands[k];
                rc = once(); // the first launch after a change of grid is not timed
                if (rc == MI355_OK && hipEventRecord(e0, ctx->stream) != hipSuccess) rc = fail(MI355_E_HIP, "hipEventRecord failed");
                for (int i = 0; i < kBurst && rc == MI355_OK; i++) rc = once();
                if (rc == MI355_OK && (hipEventRecord(e1, ctx->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess))
                    rc = fail(MI355_E_HIP, "hipEventSynchronize failed");
                float ms = 0;
                if (rc == MI355_OK && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && (round == 0 || ms < best[k])) best[k] = ms;
            }
        ctx->max_blocks_per_cu = saved;
        if (rc != MI355_OK) break;
        int win = 0;
        for (int k = 1; k < 3; k++)
            if (best[k] < best[win] * 0.995f) win = k; // a later candidate has to win by more than the timer's noise
        ctx->tuned_bpc[tune_key(sh.op, c, sh.bitmap, sh.mask)] = cands[win];
    }
    ctx->max_blocks_per_cu = saved;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int mi355_tuned_blocks_per_cu(mi355_ctx *ctx, unsigned c, unsigned what, int range)
{
    if (resolve(ctx)) return 0;
    CtxLock lk(ctx->mu);
    uint32_t key;
    if (what == MI355_TUNE_SCAN) key = tune_key(range ? kOpScanRange : kOpScanEq, c, true, false);
    else if (what == MI355_TUNE_COUNT) key = tune_key(kOpScanRange, c, false, false);
    else if (what == MI355_TUNE_MASK) key = tune_key(kOpScanRange, c, true, true);
    else if (what == MI355_TUNE_DECOMPRESS) key = tune_key(kOpDecompress, c, true, false);
    else return 0;
    const auto it = ctx->tuned_bpc.find(key);
    return it == ctx->tuned_bpc.end() ? 0 : it->second;
}

int mi355_decompress(mi355_ctx *ctx, const void *packed_host, uint64_t n, unsigned c, int32_t *out_host)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    if ((rc = check_width(c))) return rc;
    if (n == 0) return MI355_OK;
    if (!packed_host || !out_host) return fail(MI355_E_INVALID, "null pointer");
    if ((rc = bind(ctx))) return rc;
    void *dp = nullptr, *dout = nullptr;
    if ((rc = upload_packed(ctx, packed_host, n, c, &dp))) return rc;
    if ((rc = pool_get(ctx, mi355_ctx::kPoolOut, n * 4, &dout))) return rc;
    if ((rc = mi355_decompress_dev(ctx, dp, n, c, (int32_t *)dout))) return rc;
    HIP_TRY(hipMemcpyAsync(out_host, dout, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MI355_OK;
}

static int scan_host(mi355_ctx *ctx, int op, const void *packed_host, uint64_t n, unsigned c, uint32_t k0, uint32_t k1,
                     uint8_t *bitmap_host, uint64_t *hits)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    if ((rc = check_width(c))) return rc;
    if (hits) *hits = 0;
    if (n == 0) return MI355_OK;
    if (!packed_host || !bitmap_host) return fail(MI355_E_INVALID, "null pointer");
    if ((rc = bind(ctx))) return rc;
    void *dp = nullptr, *db = nullptr;
    if ((rc = upload_packed(ctx, packed_host, n, c, &dp))) return rc;
    if ((rc = pool_get(ctx, mi355_ctx::kPoolOut, bitmap_bytes(n) + 16, &db))) return rc;
    // the kernel delivers the hit count straight into pinned host memory: no second download
    if (op == kOpScanRange)
        rc = mi355_scan_range_dev(ctx, dp, n, c, k0, k1, db, (uint64_t *)ctx->hits_scratch);
    else
        rc = mi355_scan_eq_dev(ctx, dp, n, c, (int32_t)k0, db, (uint64_t *)ctx->hits_scratch);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(bitmap_host, db, bitmap_bytes(n), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (hits) *hits = (uint64_t)ctx->hits_scratch[0];
    return MI355_OK;
}

int mi355_scan_eq(mi355_ctx *ctx, const void *packed_host, uint64_t n, unsigned c, int32_t key, uint8_t *bitmap_host,
                  uint64_t *hits)
{
    return scan_host(ctx, kOpScanEq, packed_host, n, c, (uint32_t)key, 0, bitmap_host, hits);
}
int mi355_scan_range(mi355_ctx *ctx, const void *packed_host, uint64_t n, unsigned c, uint32_t lo, uint32_t hi,
                     uint8_t *bitmap_host, uint64_t *hits)
{
    return scan_host(ctx, kOpScanRange, packed_host, n, c, lo, hi, bitmap_host, hits);
}

static int shared_host(mi355_ctx *ctx, const void *packed_host, uint64_t n, unsigned c, const int32_t *keys, unsigned P,
                       int layout, uint8_t *const *outputs, uint8_t *linear_out, uint64_t *hits)
{
    int rc = resolve(ctx);
    if (rc) return rc;
    CtxLock lk(ctx->mu);
    if ((rc = check_width(c))) return rc;
    if (P < 1 || P > (unsigned)kMaxKeys) return fail(MI355_E_INVALID, "P=%u outside 1..%u", P, kMaxKeys);
    if (!keys) return fail(MI355_E_INVALID, "keys is null");
    if (hits) memset(hits, 0, P * sizeof(uint64_t));
    if (n == 0) return MI355_OK;
    if (!packed_host || (layout == MI355_LAYOUT_PER_PREDICATE ? !outputs : !linear_out))
        return fail(MI355_E_INVALID, "null pointer");
    if ((rc = bind(ctx))) return rc;
    const size_t nb = bitmap_bytes(n);
    const size_t stride = mi355_bitmap_stride(n);
    void *dp = nullptr, *dout = nullptr;
    if ((rc = upload_packed(ctx, packed_host, n, c, &dp))) return rc;
    if ((rc = pool_get(ctx, mi355_ctx::kPoolOut, (layout == MI355_LAYOUT_PER_PREDICATE ? stride : nb) * P + 16, &dout))) return rc;
    // the reference's shared scans return no counts: only count when the caller asked
    if ((rc = mi355_shared_scan_eq_dev(ctx, dp, n, c, keys, P, layout, dout, stride, hits ? (uint64_t *)ctx->hits_scratch : nullptr)))
        return rc;
    if (layout == MI355_LAYOUT_PER_PREDICATE) {
        for (unsigned k = 0; k < P; k++) {
            if (!outputs[k]) return fail(MI355_E_INVALID, "outputs[%u] is null", k);
            HIP_TRY(hipMemcpyAsync(outputs[k], (uint8_t *)dout + k * stride, nb, hipMemcpyDeviceToHost, ctx->stream));
        }
    } else {
        HIP_TRY(hipMemcpyAsync(linear_out, dout, nb * P, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (hits) memcpy(hits, ctx->hits_scratch, P * sizeof(uint64_t));
    return MI355_OK;
}

int mi355_shared_scan_eq(mi355_ctx *ctx, const void *packed_host, uint64_t n, unsigned c, const int32_t *keys,
                         unsigned P, uint8_t *const *outputs, uint64_t *hits)
{
    return shared_host(ctx, packed_host, n, c, keys, P, MI355_LAYOUT_PER_PREDICATE, outputs, nullptr, hits);
}
int mi355_shared_scan_eq_linear(mi355_ctx *ctx, const void *packed_host, uint64_t n, unsigned c, const int32_t *keys,
                                unsigned P, uint8_t *output, uint64_t *hits)
{
    return shared_host(ctx, packed_host, n, c, keys, P, MI355_LAYOUT_LINEAR, nullptr, output, hits);
}

/* ---- introspection ---- */
const char *mi355_kernel_name(const char *op, unsigned c)
{
    static thread_local char buf[96];
    if (!op || c < 1 || c > 32) return nullptr;
    if (!strcmp(op, "scan_eq"))
        snprintf(buf, sizeof buf, "mi355::scan_burst_kernel<%u, 0, ", c);
    else if (!strcmp(op, "scan_range"))
        snprintf(buf, sizeof buf, "mi355::scan_burst_kernel<%u, 1, ", c);
    else if (!strcmp(op, "shared_scan"))
        snprintf(buf, sizeof buf, "mi355::shared_lut_kernel<%u, ", c);
    else if (!strcmp(op, "decompress"))
        snprintf(buf, sizeof buf, "mi355::decompress_kernel<%u, ", c);
    else if (!strcmp(op, "pack"))
        snprintf(buf, sizeof buf, "mi355::pack_kernel<");
    else
        return nullptr;
    return buf;
}

const char *mi355_shared_scan_kernel(mi355_ctx *ctx, unsigned c, unsigned P, int layout, int with_hits)
{
    if (resolve(ctx) != MI355_OK) return nullptr;
    if (c < 1 || c > 32 || P < 1 || P > (unsigned)kMaxKeys) return nullptr;
    if (P == 1) return "scan_burst_kernel";
    CtxLock lk(ctx->mu);
    LaunchReq r{};
    int choice = -1;
    unsigned long long dummy = 0;
    r.op = kOpSharedScan;
    r.c = c;
    r.choice_out = &choice;
    r.scan.n = 1;
    r.scan.nkeys = P;
    r.scan.layout = (uint32_t)layout;
    r.scan.hits = with_hits ? &dummy : nullptr;
    if (launch(ctx, r) != MI355_OK) return nullptr;
    static const char *const names[] = {"shared_lut_kernel", "shared_lut_kernel(multi-pass)", "shared_wide_kernel", "shared_general_kernel",
                                        "shared_linear_kernel", "shared_pair_kernel"};
    return choice >= 0 && choice < 6 ? names[choice] : nullptr;
}

uint64_t mi355_tile_values(unsigned c)
{
    if (c < 1 || c > 32) return 0;
    return 64 * (uint64_t)scan_vpl((int)c, kModeEq);
}

} // extern "C"
