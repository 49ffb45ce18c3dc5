// ctx.cpp -- device context, stream, event timing and plain memory helpers of the C ABI.
#include "ms_internal.h"
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <unistd.h>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <vector>
#include <algorithm>

namespace {
std::atomic<int> g_ranges_on{0};
std::once_flag g_ranges_once;
int (*g_push)(const char *) = nullptr;
int (*g_pop)() = nullptr;
void ranges_resolve() {
    for (const char *lib : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
        void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        g_push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
        g_pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (g_push && g_pop) return;
        g_push = nullptr; g_pop = nullptr;
    }
}
}  // namespace

void ms_range_push(const char *name) { if (g_ranges_on.load(std::memory_order_relaxed) && g_push) (void)g_push(name); }
void ms_range_pop() { if (g_ranges_on.load(std::memory_order_relaxed) && g_pop) (void)g_pop(); }

int ms_ctx_order_after_downloads(ms_ctx *c) {
    if (c->d2h_pending) MS_HIP(c, hipStreamWaitEvent(c->stream, c->ev_d2h_done, 0));
    return MS_OK;
}

int ms_pinned(ms_ctx *c, size_t bytes) {
    if (bytes > c->pinned_bytes) {
        if (c->pinned) MS_HIP(c, hipHostFree(c->pinned));
        c->pinned = nullptr; c->pinned_bytes = 0;
        const size_t want = ms_align_up(bytes * 2, 4096);
        MS_HIP(c, hipHostMalloc(&c->pinned, want, hipHostMallocDefault));
        c->pinned_bytes = want;
        ++g_ms_host_allocs;
    }
    return MS_OK;
}

int ms_scratch(ms_ctx *c, size_t bytes, void **out) {
    if (bytes > c->scratch_bytes) {
        MS_HIP(c, hipStreamSynchronize(c->stream));
        if (c->scratch) MS_HIP(c, hipFree(c->scratch));
        c->scratch = nullptr; c->scratch_bytes = 0;
        const size_t want = ms_align_up(bytes * 2, 4096);
        MS_HIP(c, hipMalloc(&c->scratch, want));
        c->scratch_bytes = want;
        ++g_ms_host_allocs;
    }
    *out = c->scratch;
    return MS_OK;
}

std::atomic<long long> g_ms_host_allocs{0};

extern "C" {

long long ms_debug_host_allocs(void) { return g_ms_host_allocs.load(); }

const char *ms_version(void) { return "mi355slam 0.1 (gfx950)"; }

int ms_set_trace_ranges(int on) {
    if (on) {
        std::call_once(g_ranges_once, ranges_resolve);
        if (!g_push || !g_pop) return MS_ERR_INVALID;               // no roctx library on this machine
    }
    g_ranges_on.store(on ? 1 : 0, std::memory_order_relaxed);
    return MS_OK;
}

void ms_trace_range_push(const char *name) { ms_range_push(name ? name : ""); }
void ms_trace_range_pop(void) { ms_range_pop(); }

// Is the ROCm runtime of this process up?  Its first act is to open the kernel driver's device node, so a descriptor on /dev/kfd says so -- whoever initialised
// it (this library, torch, a profiler's preloaded tool), and without making a HIP call that would initialise it as a side effect.
static bool gpu_runtime_is_up() {
    DIR *d = opendir("/proc/self/fd");
    if (!d) return false;
    bool up = false;
    char link[64], target[64];
    while (const dirent *e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        std::snprintf(link, sizeof(link), "/proc/self/fd/%s", e->d_name);
        const ssize_t n = readlink(link, target, sizeof(target) - 1);
        if (n == 8 && std::memcmp(target, "/dev/kfd", 8) == 0) { up = true; break; }
    }
    closedir(d);
    return up;
}

int ms_prepare_process(int concurrent_contexts) {
    if (concurrent_contexts < 1) return MS_ERR_INVALID;
    if (std::getenv("GPU_MAX_HW_QUEUES")) return MS_OK;              // the caller's own setting wins (and was there when the runtime came up, if it has)
    if (gpu_runtime_is_up()) return MS_ERR_TOO_LATE;                 // the variable has been read: setting it now would only pretend
    // TWO queues per context: the runtime deals its queues to streams in turn and the process holds more streams than its sequences' (a context's download stream, the
    // application's own), and streams that share a hardware queue wait for each other's kernels -- with one queue per context eight sequences on one GPU made 5.6 k frames/s, with 12 ... 32 queues 8.4-8.7 k (tools/hw_queue_sweep.sh, round 4)
    const int q = std::max(4, std::min(2 * concurrent_contexts, 32));
    return setenv("GPU_MAX_HW_QUEUES", std::to_string(q).c_str(), 0) == 0 ? MS_OK : MS_ERR_INVALID;
}

int ms_ctx_create(int device, ms_ctx **out) {
    if (!out) return MS_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MS_ERR_NO_DEVICE;   // no CPU fallback
    if (device < 0 || device >= n) return MS_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return MS_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return MS_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return MS_ERR_NO_DEVICE;   // kernels are gfx950-only
    if (const char *e = std::getenv("MS_TRACE_RANGES")) if (e[0] == '1') (void)ms_set_trace_ranges(1);
    ms_ctx *c = new ms_ctx();
    c->device = device;
    c->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        ms_ctx_destroy(c);
        return MS_ERR_HIP;
    }
    *out = c;
    return MS_OK;
}

void ms_ctx_destroy(ms_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    ms_ba_release_pool(c);
    if (c->d2h_stream) { (void)hipStreamSynchronize(c->d2h_stream); (void)hipStreamDestroy(c->d2h_stream); }
    if (c->ev_d2h_gate) (void)hipEventDestroy(c->ev_d2h_gate);
    if (c->ev_d2h_done) (void)hipEventDestroy(c->ev_d2h_done);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->pinned) (void)hipHostFree(c->pinned);
    for (auto &b : c->ba_cache) if (b.p) (void)hipFree(b.p);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (auto &e : c->slots) if (e) (void)hipEventDestroy(e);
    delete c;
}

int ms_ctx_sync(ms_ctx *c) {
    if (!c) return MS_ERR_INVALID;
    MS_HIP(c, hipStreamSynchronize(c->stream));
    if (c->d2h_pending) { MS_HIP(c, hipStreamSynchronize(c->d2h_stream)); c->d2h_pending = false; }
    return MS_OK;
}

void *ms_ctx_stream(ms_ctx *c) { return c ? static_cast<void *>(c->stream) : nullptr; }

const char *ms_last_error(const ms_ctx *c) { return c ? c->err : "null context"; }

int ms_timer_start(ms_ctx *c) {
    if (!c) return MS_ERR_INVALID;
    MS_HIP(c, hipEventRecord(c->ev0, c->stream));
    return MS_OK;
}

int ms_timer_stop_ms(ms_ctx *c, float *ms) {
    if (!c || !ms) return MS_ERR_INVALID;
    MS_HIP(c, hipEventRecord(c->ev1, c->stream));
    MS_HIP(c, hipEventSynchronize(c->ev1));
    MS_HIP(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return MS_OK;
}

int ms_event_mark(ms_ctx *c, int slot) {
    if (!c || slot < 0 || slot >= 1024) return MS_ERR_INVALID;
    if (!c->slots[slot]) MS_HIP(c, hipEventCreate(&c->slots[slot]));
    MS_HIP(c, hipEventRecord(c->slots[slot], c->stream));
    return MS_OK;
}

int ms_event_elapsed_ms(ms_ctx *c, int a, int b, float *ms) {
    if (!c || !ms || a < 0 || a >= 1024 || b < 0 || b >= 1024 || !c->slots[a] || !c->slots[b]) return MS_ERR_INVALID;
    MS_HIP(c, hipEventSynchronize(c->slots[b]));
    MS_HIP(c, hipEventElapsedTime(ms, c->slots[a], c->slots[b]));
    return MS_OK;
}

int ms_dev_alloc(ms_ctx *c, size_t bytes, void **out) {
    if (!c || !out) return MS_ERR_INVALID;
    MS_HIP(c, hipSetDevice(c->device));
    MS_HIP(c, hipMalloc(out, bytes ? bytes : 1));
    return MS_OK;
}

int ms_dev_free(ms_ctx *c, void *p) {
    if (!c) return MS_ERR_INVALID;
    MS_HIP(c, hipStreamSynchronize(c->stream));
    MS_HIP(c, hipFree(p));
    return MS_OK;
}

int ms_host_alloc(ms_ctx *c, size_t bytes, void **out) {
    if (!c || !out) return MS_ERR_INVALID;
    MS_HIP(c, hipSetDevice(c->device));
    MS_HIP(c, hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return MS_OK;
}

int ms_host_free(ms_ctx *c, void *p) {
    if (!c) return MS_ERR_INVALID;
    if (p) MS_HIP(c, hipHostFree(p));
    return MS_OK;
}

int ms_dev_upload(ms_ctx *c, void *dst, const void *src, size_t bytes) {
    if (!c || (!dst && bytes) || (!src && bytes)) return MS_ERR_INVALID;
    MS_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    MS_HIP(c, hipStreamSynchronize(c->stream));
    return MS_OK;
}

int ms_dev_download_async(ms_ctx *c, void *dst, const void *src, size_t bytes) {
    if (!c || (!dst && bytes) || (!src && bytes)) return MS_ERR_INVALID;
    if (!bytes) return MS_OK;
    MS_HIP(c, hipSetDevice(c->device));
    if (!c->d2h_stream) {
        MS_HIP(c, hipStreamCreateWithFlags(&c->d2h_stream, hipStreamNonBlocking));
        MS_HIP(c, hipEventCreateWithFlags(&c->ev_d2h_gate, hipEventDisableTiming));
        MS_HIP(c, hipEventCreateWithFlags(&c->ev_d2h_done, hipEventDisableTiming));
    }
    MS_HIP(c, hipEventRecord(c->ev_d2h_gate, c->stream));               // after everything enqueued on the context stream so far
    MS_HIP(c, hipStreamWaitEvent(c->d2h_stream, c->ev_d2h_gate, 0));
    MS_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->d2h_stream));
    MS_HIP(c, hipEventRecord(c->ev_d2h_done, c->d2h_stream));
    c->d2h_pending = true;
    return MS_OK;
}

int ms_dev_download_wait(ms_ctx *c) {
    if (!c) return MS_ERR_INVALID;
    if (c->d2h_pending) { MS_HIP(c, hipStreamSynchronize(c->d2h_stream)); c->d2h_pending = false; }
    return MS_OK;
}

int ms_dev_download(ms_ctx *c, void *dst, const void *src, size_t bytes) {
    if (!c || (!dst && bytes) || (!src && bytes)) return MS_ERR_INVALID;
    MS_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    MS_HIP(c, hipStreamSynchronize(c->stream));
    return MS_OK;
}

}  // extern "C"
