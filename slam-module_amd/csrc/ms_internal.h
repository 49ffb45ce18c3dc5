// ms_internal.h -- shared internals of libmi355slam (product code; never includes oracle/).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/mi355slam.h"

struct ms_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t slots[1024] = {nullptr};     // ms_event_mark: created on first use
    int n_cu = 0;
    void *scratch = nullptr;      // growable device scratch for small per-call argument tables
    size_t scratch_bytes = 0;
    int hamming_path = 0;         // 0 = automatic (matrix-core kernel for unmasked searches), 1 = popcount kernel for everything (ms_hamming_set_path)
    bool greedy_attr_done[2] = {false, false};   // dynamic-LDS attribute of k_greedy_big_nodes<false/true> set on this context's device
    int greedy_path = 0;          // 0 = node-parallel greedy matchers (sequential redo of pairs that need it), 1 = one wave per pair, sequential (ms_match_set_path)
    // device blocks of destroyed bundle-adjustment handles, kept for the next ms_ba_create on this context: a window per keyframe then allocates nothing after
    // warm-up (hipFree synchronises the whole device -- with one sequence per context that stalled every other sequence's stream once per keyframe)
    struct BaBlock { void *p = nullptr; size_t bytes = 0; } ba_cache[4];
    // asynchronous downloads (ms_dev_download_async): a stream of their own, ordered after the work enqueued before them; work that overwrites what
    // they read is ordered after them on the device (ms_ctx_order_after_downloads, called by ms_orb_extract)
    hipStream_t d2h_stream = nullptr;
    hipEvent_t ev_d2h_gate = nullptr, ev_d2h_done = nullptr;
    bool d2h_pending = false;
    void *pinned = nullptr;       // page-locked host staging of small result blocks (ms_pinned), grow-only
    size_t pinned_bytes = 0;
    // page-locked staging of ms_ba_create's small uploads (the inputs of all its problems and their descriptors, copied asynchronously: the solver launch follows
    // in stream order, nobody waits); ba_stage_ev = the end of the last upload from it, waited for before the block is written again
    void *ba_stage = nullptr;
    size_t ba_stage_bytes = 0;
    hipEvent_t ba_stage_ev = nullptr;
    bool ba_stage_busy = false;
    void *ba_handle_pool[4] = {nullptr, nullptr, nullptr, nullptr};      // destroyed bundle-adjustment handle OBJECTS (their vectors keep their capacity, their event stays): ms_ba_create takes one back
    char err[512] = {0};
};

// every allocation the library makes on the host or the device after a context exists (handle objects, host scratch growth, device blocks, pinned staging, events):
// a per-keyframe path must leave it unchanged after warm-up (ms_debug_host_allocs)
#include <atomic>
extern std::atomic<long long> g_ms_host_allocs;
void ms_ba_release_pool(ms_ctx *ctx);      // ba.hip: deletes the pooled handle objects (ms_ctx_destroy)

// page-locked host staging of at least `bytes` in ctx->pinned (contents are valid until the next call that uses it)
int ms_pinned(ms_ctx *ctx, size_t bytes);

// makes the context stream wait (on the device) for the asynchronous downloads issued so far
int ms_ctx_order_after_downloads(ms_ctx *ctx);

// device scratch of at least `bytes`, reused across calls on the context stream
int ms_scratch(ms_ctx *ctx, size_t bytes, void **out);

inline int ms_fail(ms_ctx *ctx, int code, const char *fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define MS_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return ms_fail((ctx), MS_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                           __FILE__, __LINE__);                                                    \
    } while (0)

#define MS_KERNEL_CHECK(ctx, name)                                                                 \
    do {                                                                                           \
        hipError_t e__ = hipGetLastError();                                                        \
        if (e__ != hipSuccess)                                                                     \
            return ms_fail((ctx), MS_ERR_HIP, "launch of %s failed: %s", name, hipGetErrorString(e__)); \
    } while (0)

// host geometry (geometry.cpp) -- restates static_settings.cpp:9-60 and image_pyramid.cpp:76-78
namespace msgeo {
void scale_factors(int levels, float f, float *out);
void level_sigma_sq(int levels, float f, float *out);
void level_quotas(int levels, float f, int max_kpts, int32_t *out);
void level_sizes(int levels, float f, int w0, int h0, int32_t *w, int32_t *h);
void umax(int32_t *u16);
// cv::resize INTER_LINEAR 8U coefficient tables (11-bit fixed point)
void resize_tables(int src_n, int dst_n, bool is_x, std::vector<int16_t> &ofs, std::vector<int16_t> &coef);
}  // namespace msgeo

// Stage ranges for a profiler's marker trace (ms_set_trace_ranges): roctxRangePushA / roctxRangePop, resolved at run time from
// librocprofiler-sdk-roctx / libroctx64 -- no link dependency, nothing happens (one relaxed load) while they are off.
void ms_range_push(const char *name);
void ms_range_pop();
struct MsRange {
    explicit MsRange(const char *name) { ms_range_push(name); }
    ~MsRange() { end(); }
    void end() { if (open_) { ms_range_pop(); open_ = false; } }      // close before the scope does (an early return still closes it)
    bool open_ = true;
    MsRange(const MsRange &) = delete;
    MsRange &operator=(const MsRange &) = delete;
};

inline int ms_div_up(int a, int b) { return (a + b - 1) / b; }
inline size_t ms_align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }
