/*
 * mi355_scan.h -- C ABI of libmi355scan.so: MI355X (gfx950) engine for bit-packed column
 * decompress / predicate scan / shared multi-predicate scan.
 *
 * This is the drop-in boundary for the hot path of RRr89/Shared_SIMD_Scan.  The reference has no
 * FFI layer: its boundary is the set of C++ free functions declared in src/simd_scan.hpp:20-123.
 * Each entry point below names the reference interface it replaces ("replaces:" lines, paths
 * relative to the reference repository).  include/simd_scan.hpp maps the reference's C++ names
 * onto this ABI; INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every function returns an int status (MI355_OK == 0);
 *     mi355_last_error() gives the message of the calling thread's last failure.
 *   - `c` is the bit width (the reference's compile-time BITS_NEEDED, src/simd_scan.hpp:12),
 *     1..32, passed at run time.  `n` is always a number of VALUES, never bytes (as in the
 *     reference).  Counts are 64-bit (the reference returns `int hits`, src/simd_scan.hpp:89).
 *   - packed format: value i occupies bits [c*i, c*i+c) of a little-endian byte stream, LSB
 *     first (src/simd_scan_compression.cpp:53-104).  A packed buffer must be at least
 *     mi355_compressed_buffer_size(c, n) bytes (payload + 256 B pad, src/simd_scan.hpp:20-26);
 *     device packed buffers must be 16-byte aligned.
 *   - bitmap format: bit i = byte i/8, bit i%8 (src/util.cpp:51-58).  The engine writes exactly
 *     ceil(n/8) bytes; bits >= n of the last byte are 0; hits = popcount over [0, n).  (The
 *     reference's variants disagree with each other past n; see DESIGN.md "tail rule".)
 *   - keys are signed 32-bit and compared unmasked: a key outside [0, 2^c) matches nothing
 *     (reference behaviour, SURVEY 8c hazard 5).
 *   - *_dev functions take DEVICE pointers, enqueue on the context's stream and return without
 *     synchronising; the others take HOST pointers, copy in/out and return when done.
 *   - a context binds one device and one stream and owns the scratch the kernels use.  Threading contract (the
 *     reference's functions are stateless and re-entrant; its own shared_scan_128_threaded calls scan_128 from an
 *     OpenMP loop, src/simd_scan_shared.cpp:25-32): every entry point may be called from any number of host threads.
 *     ctx == NULL is the CALLING THREAD's default context (own scratch, own buffer pool; device 0, the null stream),
 *     so concurrent drop-in calls never share state.  An explicit context may be shared by threads too: each call
 *     holds the context's lock while it touches it (host-pointer calls for their whole duration), i.e. calls on one
 *     context serialise; use one context per thread / stream for concurrency.
 *   - there is no CPU fallback: without a usable gfx950 device every compute entry point fails
 *     with MI355_E_NODEVICE / MI355_E_HIP.
 */
#ifndef MI355_SCAN_H
#define MI355_SCAN_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define MI355_API __attribute__((visibility("default")))
#else
#define MI355_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_OK 0
#define MI355_E_INVALID (-1)  /* bad argument (width, null pointer, alignment, P) */
#define MI355_E_HIP (-2)      /* a HIP runtime call failed */
#define MI355_E_NODEVICE (-3) /* no gfx950 device visible */
#define MI355_E_COMM (-4)     /* RCCL unavailable or an RCCL call failed (multi-GPU exchange step only) */

#define MI355_LAYOUT_PER_PREDICATE 0 /* src/simd_scan_shared.cpp:82  outputs[key][group]      */
#define MI355_LAYOUT_LINEAR 1        /* src/simd_scan_shared_linear.cpp:57  outputs[group*P+key] */

/* synthetic column kinds for mi355_generate_dev */
#define MI355_GEN_MOD 0      /* v = (first_row + i) % param           (src/benchmark.cpp:173, :277) */
#define MI355_GEN_SPLITMIX 1 /* v = splitmix64(seed=param, first_row + i) & (2^c - 1)             */
#define MI355_GEN_INDEX 2    /* v = (first_row + i) & (2^c - 1)       (src/benchmark.cpp:81)       */

typedef struct mi355_ctx mi355_ctx;

MI355_API const char *mi355_last_error(void);
MI355_API const char *mi355_version(void);

/* ---- context: device + stream.  ctx == NULL in any call below means the calling thread's default
 * context (device 0, the null stream), created on the thread's first use and destroyed when the thread
 * exits; mi355_ctx_set_option(NULL, ...) therefore configures the calling thread only. ------------- */
MI355_API int mi355_ctx_create(int device, void *hip_stream /* hipStream_t or NULL */, mi355_ctx **out);
MI355_API int mi355_ctx_destroy(mi355_ctx *ctx);
MI355_API int mi355_ctx_synchronize(mi355_ctx *ctx);
MI355_API int mi355_device_count(int *count);
/* re-point the context at another HIP stream (e.g. a capture stream: the *_dev scan / decompress / bitmap entry
 * points enqueue work only -- no allocation, no synchronisation -- so they can be captured into a hipGraph;
 * exceptions: mi355_shared_scan_eq_dev / mi355_scan_in_dev with P > 8 upload the key list per call (asynchronously,
 * through a ring of pinned slots: no stream synchronisation, but refused while the stream is capturing), and
 * mi355_bitmap_to_rowids_dev may grow its workspace) */
/* Scratch, key slots and the buffer pool belong to the context, not to a stream: when neither stream is being captured,
 * work already enqueued on the old stream is ordered before anything enqueued on the new one (event wait, no host
 * synchronisation).  While either stream is capturing the caller orders the two. */
MI355_API int mi355_ctx_set_stream(mi355_ctx *ctx, void *hip_stream);
/* tuning knobs: "max_blocks_per_cu" (0 = the engine's per-kernel default), "dma_aux" (bits 0-3: cache policy of
 * the HBM->LDS loads, 0 default / 2 non-temporal; bit 4: non-temporal output stores in decompress; default 18),
 * "scan_nt_stores" (result stores of the scans: -1 chosen by output size (default: write-through while the bitmap fits
 * the Infinity Cache, non-temporal beyond), 0 plain, 1 non-temporal, 2 write-through), "select_kernel" (mi355_scan_select_dev:
 * 0 / 2 = decoder / expander roles (default), 1 = the older single-role kernel, kept for A/B runs), "kernel_flags" (A/B switches of the kernels; results never depend on them except the two timing
 * ablations of the selection documented in DESIGN.md) */
MI355_API int mi355_ctx_set_option(mi355_ctx *ctx, const char *name, int value);

/* ---- load-time tuning (optional) ----------------------------------------------------------------------------------
 * The resident blocks per CU at which the streaming kernels run fastest differ between MI355X boxes for the same
 * binary by 3-10 %.  mi355_tune_dev measures 1 / 2 / 4 blocks per CU on the caller's own resident column (`what` = a
 * mask of MI355_TUNE_* bits: equality + range scans, their count-only forms, the fused-mask scan, decompress), with
 * launches back to back as a query stream issues them, and keeps the winners in the context: later launches of that
 * kind and width over >= 5e7 rows use them ("max_blocks_per_cu", when set, still wins).  It BLOCKS (synchronises the
 * context's stream; about 40 launches per kind), uses the context's pooled scratch for its outputs, and never changes
 * any result.  Columns below 5e7 rows: returns MI355_OK without measuring.  Call it once per width after loading. */
enum {
    MI355_TUNE_SCAN = 1,
    MI355_TUNE_COUNT = 2,
    MI355_TUNE_MASK = 4,
    MI355_TUNE_DECOMPRESS = 8,
    MI355_TUNE_ALL = 15
};
MI355_API int mi355_tune_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, unsigned what);
/* what mi355_tune_dev kept for one kind (ONE MI355_TUNE_* bit; range != 0: the range-scan kernel), 0 = not tuned */
MI355_API int mi355_tuned_blocks_per_cu(mi355_ctx *ctx, unsigned c, unsigned what, int range);

/* ---- buffer sizing, in bytes.  replaces: compressed_buffer_size / decompression_output_buffer_size
 * / scan_output_buffer_size (src/simd_scan.hpp:20-40), same formulas. -------------------------- */
MI355_API size_t mi355_compressed_buffer_size(unsigned c, size_t n);
MI355_API size_t mi355_decompression_output_buffer_size(size_t n);
MI355_API size_t mi355_scan_output_buffer_size(size_t n);
/* recommended distance in bytes between the per-predicate bitmaps of a shared scan over n rows: ceil(n/8) rounded up to
 * 256, so that every bitmap starts on a whole 128-byte line (see mi355_shared_scan_eq_dev) */
MI355_API size_t mi355_bitmap_stride(size_t n);

/* ---- device memory helpers (hipMalloc / hipMemcpy / hipMemset on the context's device) ------- */
MI355_API int mi355_dev_alloc(mi355_ctx *ctx, size_t bytes, void **dptr);
MI355_API int mi355_dev_free(mi355_ctx *ctx, void *dptr);
MI355_API int mi355_dev_upload(mi355_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
MI355_API int mi355_dev_download(mi355_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
MI355_API int mi355_dev_memset(mi355_ctx *ctx, void *dst_dev, int value, size_t bytes);

/* ---- compression.  replaces: compress_9bit_input (src/simd_scan_compression.cpp:53-104) at any
 * width.  Values must be < 2^c (the reference does not mask either; the engine masks to c bits
 * instead of corrupting neighbours).  Writes mi355_compressed_buffer_size(c,n) bytes (pad zeroed). */
MI355_API int mi355_pack_u16(mi355_ctx *ctx, const uint16_t *values, uint64_t n, unsigned c, void *packed_host);
MI355_API int mi355_pack_u32(mi355_ctx *ctx, const uint32_t *values, uint64_t n, unsigned c, void *packed_host);
MI355_API int mi355_pack_u16_dev(mi355_ctx *ctx, const uint16_t *values_dev, uint64_t n, unsigned c, void *packed_dev);
MI355_API int mi355_pack_u32_dev(mi355_ctx *ctx, const uint32_t *values_dev, uint64_t n, unsigned c, void *packed_dev);
/* synthesise a packed column on the device from the global row index (no upload): the shapes of
 * bench_scan / bench_shared_scan / bench_decompression (src/benchmark.cpp:81,:173,:277) */
MI355_API int mi355_generate_dev(mi355_ctx *ctx, int kind, uint64_t first_row, uint64_t n, unsigned c, uint64_t param,
                       void *packed_dev);

/* ---- decompression to int32.  replaces: decompress_unvectorized / decompress_128* / decompress_256
 * / decompress_256_avx2 (src/simd_scan_decompression.cpp, src/simd_scan.hpp:51-73): one kernel, same
 * result on [0,n).  Writes exactly n ints. ----------------------------------------------------- */
MI355_API int mi355_decompress(mi355_ctx *ctx, const void *packed_host, uint64_t n, unsigned c, int32_t *out_host);
MI355_API int mi355_decompress_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, int32_t *out_dev);

/* ---- equality scan.  replaces: scan_unvectorized / scan_128 / scan_128_unrolled / scan_256 /
 * scan_256_unrolled (src/simd_scan.cpp, src/simd_scan.hpp:89-96).  hits may be NULL.
 * _dev: bitmap_dev 16-byte aligned, >= ceil(n/8) bytes; hits_dev is a device uint64, overwritten. */
MI355_API int mi355_scan_eq(mi355_ctx *ctx, const void *packed_host, uint64_t n, unsigned c, int32_t key, uint8_t *bitmap_host,
                  uint64_t *hits);
MI355_API int mi355_scan_eq_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, int32_t key, void *bitmap_dev,
                      uint64_t *hits_dev);

/* ---- inclusive range scan lo <= v <= hi.  replaces: the declared-but-unimplemented
 * `int scan(int predicate_low, int predicate_high, __m128i*, int)` (src/simd_scan.hpp:76-84). */
MI355_API int mi355_scan_range(mi355_ctx *ctx, const void *packed_host, uint64_t n, unsigned c, uint32_t lo, uint32_t hi,
                     uint8_t *bitmap_host, uint64_t *hits);
MI355_API int mi355_scan_range_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, uint32_t lo, uint32_t hi,
                         void *bitmap_dev, uint64_t *hits_dev);

/* ---- shared (multi-predicate) equality scan: P keys, one pass over the column.
 * replaces (per-predicate outputs): shared_scan_128_{sequential,sequential_unrolled,threaded,standard,
 *   standard_unrolled,parallel}, shared_scan_256_{sequential,standard,parallel}
 *   (src/simd_scan_shared.cpp, src/simd_scan.hpp:102-113);
 * replaces (linear output): shared_scan_128_linear_{standard,simple} and
 *   shared_scan_128_linear_static<NUM> (src/simd_scan_shared_linear.cpp, src/simd_scan.hpp:119-236).
 * 1 <= P <= 1024.  hits (P entries) may be NULL.
 *   host, per-predicate: outputs[k] points at >= ceil(n/8) bytes for key k;
 *   host, linear:        output holds ceil(n/8)*P bytes, byte of 8-value group g and key k at g*P+k;
 *   _dev: out_dev is one 16-byte-aligned device buffer; per-predicate bitmaps start at out_dev + k*stride_bytes
 *         (stride_bytes a multiple of 16, >= ceil(n/8); use mi355_bitmap_stride(n): bitmaps that start in the middle of
 *         a 128-byte line make every wave store touch partial lines -- measured up to 2x slower at P >= 64, where the
 *         result streams bound the kernel); linear ignores stride_bytes.
 * The reference's shared scans return no counts: pass hits = NULL for exactly its work (with counts the engine
 * may pick a different kernel; the bitmaps are the same).  "scan_nt_stores" also governs these result stores. */
MI355_API int mi355_shared_scan_eq(mi355_ctx *ctx, const void *packed_host, uint64_t n, unsigned c, const int32_t *keys,
                         unsigned P, uint8_t *const *outputs, uint64_t *hits);
MI355_API int mi355_shared_scan_eq_linear(mi355_ctx *ctx, const void *packed_host, uint64_t n, unsigned c, const int32_t *keys,
                                unsigned P, uint8_t *output, uint64_t *hits);
MI355_API int mi355_shared_scan_eq_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, const int32_t *keys_host,
                             unsigned P, int layout, void *out_dev, uint64_t stride_bytes, uint64_t *hits_dev);

/* ---- beyond the reference: the predicates and bitmap consumers its header only hints at
 * (src/simd_scan.hpp:76-84 "predicate_low<=key<=predicate_high").  Device pointers only. ------- */
#define MI355_CMP_EQ 0
#define MI355_CMP_NE 1
#define MI355_CMP_LT 2
#define MI355_CMP_LE 3
#define MI355_CMP_GT 4
#define MI355_CMP_GE 5
#define MI355_CMP_BETWEEN 6     /* a <= v <= b */
#define MI355_CMP_NOT_BETWEEN 7 /* v < a or v > b */
/* bitmap[i] = (v_i OP a [, b]) AND (and_mask_dev ? and_mask[i] : 1).  Unsigned comparison on the decoded value;
 * a, b are clamped to the column's domain [0, 2^c).  and_mask_dev (nullable): a canonical bitmap of
 * >= ceil(n/8) bytes, 16-byte aligned -- chains conjunctions over several columns without a separate AND pass. */
MI355_API int mi355_scan_where_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, int op, int64_t a, int64_t b,
                                   const void *and_mask_dev, void *bitmap_dev, uint64_t *hits_dev);
/* The general form, fused consumers included (the intent of src/simd_scan.hpp:76-84):
 *   p[i]      = (v_i OP a [, b])
 *   bitmap[i] = mask_dev ? COMBINE(p[i], mask[i]) : p[i]     mask_op: MI355_BITMAP_AND  p & mask   (conjunction)
 *                                                                      MI355_BITMAP_OR   p | mask   (disjunction)
 *                                                                      MI355_BITMAP_XOR  p ^ mask
 *                                                                      MI355_BITMAP_ANDNOT  mask & ~p  (rows of mask that fail p)
 * so a chain of predicates over several columns never needs a separate bitmap pass.
 * bitmap_dev == NULL: COUNT-ONLY scan -- only hits_dev is produced, nothing is stored (the kernel then runs at the
 * speed of the read stream alone; `SELECT count(*) WHERE ...`). */
MI355_API int mi355_scan_combine_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, int op, int64_t a, int64_t b,
                                     int mask_op, const void *mask_dev, void *bitmap_dev, uint64_t *hits_dev);

/* bitmap[i] = (v_i IN {keys[0..P-1]}) AND (and_mask ? and_mask[i] : 1), negated when `negate` != 0 (NOT IN).
 * 1 <= P <= 1024; keys outside [0, 2^c) match nothing.  c <= 16: bitset lookup, cost independent of P;
 * c > 16: compare chain, O(P) per value. */
MI355_API int mi355_scan_in_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, const int32_t *keys_host, unsigned P,
                                int negate, const void *and_mask_dev, void *bitmap_dev, uint64_t *hits_dev);

/* Predicates over TWO columns of n rows each in one call:
 *   bitmap[i] = COMBINE(p2(v2_i), p1(v1_i))    combine_op: MI355_BITMAP_AND p1 & p2, _OR p1 | p2, _XOR, _ANDNOT p1 & ~p2
 * Columns of the same width run as ONE launch (both tiles in LDS, both decoded in registers: the first predicate's
 * bitmap never exists in memory); different widths run as two launches with the combination fused into the second.
 * bitmap_dev == NULL: count only. */
MI355_API int mi355_scan2_dev(mi355_ctx *ctx, const void *packed1_dev, unsigned c1, int op1, int64_t a1, int64_t b1,
                              const void *packed2_dev, unsigned c2, int op2, int64_t a2, int64_t b2, uint64_t n, int combine_op,
                              void *bitmap_dev, uint64_t *hits_dev);

/* Fused scan + selection vector: rowids_dev receives first_row + i, ascending, for every row i with
 *   mask_dev ? COMBINE(p[i], mask[i]) : p[i]        (p, mask_op as in mi355_scan_combine_dev)
 * (at most `capacity` ids are written), count_dev the number of such rows -- ONE launch, and no bitmap ever goes to HBM
 * (the chain scan -> mi355_bitmap_to_rowids_dev writes the bitmap once and reads it twice).  If the in-launch look-back
 * ever gave up waiting (it cannot while the device makes progress) count_dev reads UINT64_MAX.  Not capturable into a
 * graph when the context's workspace has to grow (first call for a larger column). */
MI355_API int mi355_scan_select_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, int op, int64_t a, int64_t b,
                                    int mask_op, const void *mask_dev, uint64_t first_row, uint64_t *rowids_dev, uint64_t capacity,
                                    uint64_t *count_dev);

#define MI355_BITMAP_AND 0
#define MI355_BITMAP_OR 1
#define MI355_BITMAP_XOR 2
#define MI355_BITMAP_ANDNOT 3 /* a & ~b */
/* out = a OP b over n bits (canonical bitmaps, 16-byte aligned; out may alias a or b); count_dev (nullable) gets
 * popcount(out) */
MI355_API int mi355_bitmap_combine_dev(mi355_ctx *ctx, int op, const void *a_dev, const void *b_dev, void *out_dev, uint64_t n,
                                       uint64_t *count_dev);
MI355_API int mi355_bitmap_count_dev(mi355_ctx *ctx, const void *bitmap_dev, uint64_t n, uint64_t *count_dev);
/* selection vector: writes first_row + i for every set bit i < n, ascending, into rowids_dev (at most `capacity`
 * entries are written) and the total number of set bits into count_dev.  workspace: ctx-owned, grows on demand. */
MI355_API int mi355_bitmap_to_rowids_dev(mi355_ctx *ctx, const void *bitmap_dev, uint64_t n, uint64_t first_row,
                                         uint64_t *rowids_dev, uint64_t capacity, uint64_t *count_dev);

/* "take": out_dev[i] = value of row rowids_dev[i] of the packed column (row ids as the selection calls above produce them:
 * first_row + index; any order, duplicates allowed), for i < min(*count_dev, capacity) -- the count is read ON THE DEVICE,
 * so the call composes with mi355_scan_select_dev / mi355_bitmap_to_rowids_dev without a host round trip.  An id outside
 * [first_row, first_row + n) yields -1 (at c = 32 indistinguishable from the value 0xffffffff).  packed_dev as everywhere:
 * readable 8 bytes past the last value (mi355_compressed_buffer_size covers it). */
MI355_API int mi355_gather_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, uint64_t first_row,
                               const uint64_t *rowids_dev, const uint64_t *count_dev, uint64_t capacity, int32_t *out_dev);

/* Aggregates of a packed column, optionally only over the rows of a bitmap (SELECT sum(b), count(*), min(b), max(b) WHERE
 * <mask>): ONE pass at the speed of the column stream, nothing decoded to memory.  out_dev[0] = sum, [1] = count of rows that
 * counted, [2] = min, [3] = max (values as unsigned c-bit integers; count = 0: sum 0, min UINT64_MAX, max 0).
 * mask_dev == NULL: every row counts.  Asynchronous on the context's stream. */
MI355_API int mi355_aggregate_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, const void *mask_dev,
                                  uint64_t *out_dev);

/* Histogram: counts_dev[v] = number of rows (of the bitmap mask_dev, or all rows when NULL) whose value is v, for every
 * v in [0, 2^c) -- `SELECT v, count(*) ... GROUP BY v` over a dictionary-coded column in one pass.  c <= 14 (the counters
 * live in LDS); counts_dev: 2^c uint64, overwritten.  Asynchronous on the context's stream. */
MI355_API int mi355_histogram_dev(mi355_ctx *ctx, const void *packed_dev, uint64_t n, unsigned c, const void *mask_dev,
                                  uint64_t *counts_dev);

/* ---- row-range sharding helper (one process per GPU): rank's rows [first, first+count) of an n-row
 * column split over `world` ranks at multiples of 8192 rows, so every shard's packed slice starts 16-byte aligned on
 * a whole value and its bitmap slice on a whole byte.  Pure arithmetic, no device needed. */
MI355_API int mi355_shard_rows(uint64_t n, unsigned world, unsigned rank, uint64_t *first, uint64_t *count);

/* ---- multi-GPU exchange step: one process per GPU, one communicator rank per process, RCCL over xGMI.
 * The scan itself needs no communication (rows are independent: rank r scans its own row range); the only exchange is
 * the final gather of the per-shard bitmaps to one rank plus the sum of the hit counts.  RCCL is bound at run time
 * (dlopen of librccl.so.1): hosts that never shard do not need it.  All calls enqueue on the context's stream and
 * return without synchronising.
 *   bootstrap: rank 0 calls mi355_comm_get_unique_id and hands the MI355_COMM_ID_BYTES bytes to every rank over any
 *   host channel (file, socket, MPI, a torch.distributed store); every rank then calls mi355_comm_create (collective:
 *   it returns once all `world` ranks have joined).  The communicator is bound to the context's device. */
#define MI355_COMM_ID_BYTES 128
typedef struct mi355_comm mi355_comm;
MI355_API int mi355_comm_get_unique_id(void *id_out /* MI355_COMM_ID_BYTES */);
MI355_API int mi355_comm_create(mi355_ctx *ctx, int world, int rank, const void *id, mi355_comm **out);
MI355_API int mi355_comm_destroy(mi355_comm *comm);
MI355_API int mi355_comm_info(const mi355_comm *comm, int *world, int *rank);
/* gather: rank r contributes bytes_per_rank[r] bytes at local_dev (all ranks pass the same array); on `root` they land
 * in out_dev at offset sum(bytes_per_rank[0..r)) -- grouped ncclSend / ncclRecv straight into the final buffer, the
 * root's own slice a device-to-device copy (skipped when local_dev already is that slice of out_dev).  out_dev is
 * ignored on the other ranks. */
MI355_API int mi355_gather_bitmaps_dev(mi355_ctx *ctx, mi355_comm *comm, const void *local_dev, const uint64_t *bytes_per_rank,
                                       int root, void *out_dev);
/* the same with explicit destinations: rank r's bytes land at out_dev + offset_per_rank[r] (NULL: packed as above).
 * Used to gather a shard piece by piece while later pieces are still being scanned (each piece of every rank goes
 * straight to its place in the final bitmap). */
MI355_API int mi355_gather_bitmaps_at_dev(mi355_ctx *ctx, mi355_comm *comm, const void *local_dev, const uint64_t *bytes_per_rank,
                                          const uint64_t *offset_per_rank, int root, void *out_dev);
/* in-place sum over the ranks of `count` device uint64 counters (ncclAllReduce): every rank gets the column-wide counts */
MI355_API int mi355_allreduce_hits_dev(mi355_ctx *ctx, mi355_comm *comm, uint64_t *hits_dev, unsigned count);
/* sharded scans = local scan of this rank's rows_per_rank[rank] rows (packed_dev = the rank's slice, starting on a
 * whole value) + gather of the bitmaps to `root` + all-reduce of the hit count.  Every shard but the last must hold a
 * multiple of 8 rows (mi355_shard_rows gives multiples of 8192).  local_bitmap_dev: >= ceil(rows/8) bytes, 16-byte
 * aligned; full_bitmap_dev (root only): >= sum of the shards' bitmap bytes; hits_dev (nullable): the COLUMN-wide count
 * on every rank. */
MI355_API int mi355_sharded_scan_eq_dev(mi355_ctx *ctx, mi355_comm *comm, const void *packed_dev, unsigned c, int32_t key,
                                        void *local_bitmap_dev, const uint64_t *rows_per_rank, int root, void *full_bitmap_dev,
                                        uint64_t *hits_dev);
MI355_API int mi355_sharded_scan_range_dev(mi355_ctx *ctx, mi355_comm *comm, const void *packed_dev, unsigned c, uint32_t lo,
                                           uint32_t hi, void *local_bitmap_dev, const uint64_t *rows_per_rank, int root,
                                           void *full_bitmap_dev, uint64_t *hits_dev);

/* ---- introspection used by bench.py / tests --------------------------------------------------- */
/* name of the HIP kernel a given op dispatches to at width c ("scan_eq", "scan_range", "shared_scan",
 * "decompress", "pack"); returns NULL for unknown ops */
MI355_API const char *mi355_kernel_name(const char *op, unsigned c);
/* kernel family a shared scan of P keys at width c dispatches to (nothing is launched): "scan_burst_kernel" (P = 1),
 * "shared_pair_kernel" (P = 2), "shared_lut_kernel" (P <= 8), "shared_lut_kernel(multi-pass)", "shared_wide_kernel" (stands
 * for the 32-keys-per-lookup kernels), "shared_linear_kernel" (linear rows, lanes in memory order), "shared_general_kernel"
 * (tables do not fit in LDS) */
MI355_API const char *mi355_shared_scan_kernel(mi355_ctx *ctx, unsigned c, unsigned P, int layout, int with_hits);
/* rows per wave tile of the scan kernels at width c (shard boundaries should be multiples of it) */
MI355_API uint64_t mi355_tile_values(unsigned c);

#ifdef __cplusplus
}
#endif
#endif /* MI355_SCAN_H */
