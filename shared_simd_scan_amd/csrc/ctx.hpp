// ctx.hpp -- the context object behind include/mi355_scan.h and the helpers every host-side translation unit of
// libmi355scan.so shares (capi.hip, comm.hip): error reporting, default-context resolution, locking.
#pragma once

#include "../../include/mi355_scan.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>

struct mi355_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cus = 256;
    int max_blocks_per_cu = 0;
    int scan_nt_stores = -1; // -1: by bitmap size (see width_group.hip), 0 plain, 1 non-temporal
    int shared_vpl = 0;  // 0: engine's choice
    int select_kernel = 0; // mi355_scan_select_dev: 0 = by the predicate's expected selectivity, 1 = single-role kernel, 2 = decoder / expander roles
    unsigned kernel_flags = 0; // experiment switches handed to the kernels (ScanArgs::flags)
    // mi355_tune_dev: blocks per CU measured on THIS device for the large streaming launches; key = tune_key() in capi.hip
    std::map<uint32_t, int> tuned_bpc;
    int scan_burst = 0;  // 0: tiles per store burst by width; 1: one tile per burst
    int dma_aux = 18; // bits 0-3: policy of the HBM->LDS loads (2 = non-temporal: the column is streamed once);
                      // bit 4: non-temporal stores in decompress
    // Every entry point that touches the state below holds `mu` while it does (the host-pointer flavours from their
    // first copy to their final synchronisation), so one context may be shared by several host threads; contexts are
    // independent of each other.  Recursive: the host-pointer flavours call the *_dev ones.
    std::recursive_mutex mu;
    unsigned long long *hits_scratch = nullptr; // host-pointer API: where the kernels deliver hit counts
    unsigned long long *kernel_scratch = nullptr; // kScratchWords words, all zero between launches (kernels.hpp hits_finalize)
    // key lists longer than 8 travel through device memory: a ring of kKeySlots pinned host slots and device slots of
    // 1024 + 8 keys each, so uploading a list never waits for the stream (only for the copy that used the slot
    // kKeySlots calls ago)
    int32_t *keys_scratch = nullptr;            // device: kKeySlots x (1024 + 8) keys
    int32_t *keys_pinned = nullptr;             // host (pinned): the same
    hipEvent_t key_events[8] = {};
    bool key_used[8] = {};
    int key_next = 0;
    hipEvent_t order_event = nullptr;           // mi355_ctx_set_stream: new stream waits for the old one
    unsigned long long *rowid_ws = nullptr;     // chunk counts of mi355_bitmap_to_rowids_dev / mi355_scan_select_dev
    size_t rowid_ws_entries = 0;
    // host-pointer (drop-in) flavours: grow-only device buffers kept between calls -- no hipMalloc / hipFree per call
    enum { kPoolIn = 0, kPoolOut = 1, kPoolAux = 2, kPoolSlots = 3 };
    void *pool[kPoolSlots] = {};
    size_t pool_bytes[kPoolSlots] = {};
    bool is_thread_default = false;
};

namespace mi355 {

int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
const char *last_error();

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return ::mi355::fail(MI355_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ctx == NULL -> the calling thread's default context (device 0, the null stream), created on first use and
// destroyed when the thread exits
int resolve(mi355_ctx *&ctx);
// a context is bound to one device: make it current for this thread before touching it
int bind(mi355_ctx *ctx);
// grow-only device buffer of the context (host-pointer flavours); the caller holds ctx->mu
int pool_get(mi355_ctx *ctx, int slot, size_t bytes, void **out);

typedef std::lock_guard<std::recursive_mutex> CtxLock;

} // namespace mi355
