// Native driver of the bench's C5 leg: S independent sequences, one host thread + one context (stream) + its own handles per sequence,
// every call through the C ABI of libmi355slam.so -- what a C++ integrator's backend thread does (mapper_helpers.cpp:1011-1131 order):
// per frame extract -> match against the previous frame -> ratio test; on every k-th frame one local bundle adjustment of a NEW window
// (create + solve + download + destroy).  `bench.py --c5-native` calls c5_prepare() / c5_go() through ctypes instead of starting Python threads; it exists to
// show that the C5 numbers are the library's and not the interpreter's: 8 sequences on one MI355X give 3.4 k frames/s + 690 BA/s with either driver (the limit
// is how many small kernels of 8 streams the GPU runs side by side).  Measurement infrastructure only: nothing in the product depends on this file.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "mi355slam.h"

namespace {
struct Shared {
    std::atomic<int> ready{0};
    std::atomic<int> go{0};
};
struct Seq {
    int frames_done = 0, ba_done = 0, matches = 0, status = 0;
    double seconds = 0, pose_seconds = 0;
    char err[256] = {0};
};

#define C5_CHECK(call)                                                                                      \
    do {                                                                                                    \
        const int rc_ = (call);                                                                             \
        if (rc_ != MS_OK) { out.status = rc_; std::snprintf(out.err, sizeof(out.err), "%s: %s", #call, ctx ? ms_last_error(ctx) : "no context"); goto done; } \
    } while (0)

// The deployment shape (mapper.cpp:356-393 beside :229-279): a sequence's BACK END -- a thread and a context of its own that takes the keyframes its front end
// hands over and runs the two-stage localBundleAdjust of a NEW window for each (create, solve, create, copy_state, solve, download, destroy).
struct BackEnd {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::pair<int, std::chrono::steady_clock::time_point>> pending;
    bool stopping = false;
    std::vector<double> lat_ms, wait_ms;
    int status = 0;
    char err[256] = {0};
    void keyframe(int k) { { std::lock_guard<std::mutex> lk(mu); pending.emplace_back(k, std::chrono::steady_clock::now()); } cv.notify_one(); }
    void stop() { { std::lock_guard<std::mutex> lk(mu); stopping = true; } cv.notify_one(); }
};
void run_back(int device, const ms_ba_problem *stage1, const ms_ba_problem *stage2, int n_windows, int extra_pose, Shared &sh, BackEnd &B) {
    ms_ctx *ctx = nullptr;
    bool released = false;
    std::vector<double> pose_out, point_out;
    auto window = [&](int k) -> int {
        const ms_ba_problem &P1 = stage1[k % n_windows], &P2 = stage2[k % n_windows];
        ms_ba *h1 = nullptr, *h2 = nullptr;
        int rc = ms_ba_create(ctx, &P1, 1, &h1);
        if (rc == MS_OK) rc = ms_ba_solve(h1);                                  // stage 1 runs while the host builds stage 2's index structures
        if (rc == MS_OK) rc = ms_ba_create(ctx, &P2, 1, &h2);
        const int32_t extra = extra_pose;
        if (rc == MS_OK) rc = ms_ba_copy_state(h2, h1, &extra);
        if (rc == MS_OK) rc = ms_ba_solve(h2);
        ms_ba_result res;
        pose_out.resize(7 * (size_t)P2.n_pose); point_out.resize(3 * (size_t)P2.n_point);
        if (rc == MS_OK) rc = ms_ba_download(h2, 0, pose_out.data(), point_out.data(), nullptr, &res);
        if (h1) ms_ba_destroy(h1);
        if (h2) ms_ba_destroy(h2);
        return rc;
    };
    int rc = ms_ctx_create(device, &ctx);
    if (rc == MS_OK) rc = window(0);
    if (rc == MS_OK) rc = ms_ctx_sync(ctx);
    sh.ready.fetch_add(1); released = true;
    if (rc == MS_OK) {
        while (sh.go.load(std::memory_order_acquire) == 0) std::this_thread::yield();
        for (;;) {
            std::pair<int, std::chrono::steady_clock::time_point> item;
            {
                std::unique_lock<std::mutex> lk(B.mu);
                B.cv.wait(lk, [&] { return !B.pending.empty() || B.stopping; });
                if (B.pending.empty()) break;
                item = B.pending.front(); B.pending.pop_front();
            }
            const auto t0 = std::chrono::steady_clock::now();
            rc = window(item.first);
            if (rc != MS_OK) break;
            B.lat_ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            B.wait_ms.push_back(std::chrono::duration<double, std::milli>(t0 - item.second).count());
        }
    }
    if (rc != MS_OK) { B.status = rc; std::snprintf(B.err, sizeof(B.err), "back end: %s", ctx ? ms_last_error(ctx) : "no context"); }
    (void)released;
    if (ctx) ms_ctx_destroy(ctx);
}

void run_sequence(int device, const uint8_t *frames_host, int n_frames, int n_distinct, int W, int H, const ms_ba_problem *windows, int n_windows, int kf_every,
                  const ms_orb_config &cfg, float ratio, Shared &sh, int n_seq, Seq &out, const ms_ba_problem *pose_probs = nullptr, int n_pose_probs = 0, BackEnd *back = nullptr) {
    ms_ctx *ctx = nullptr;
    ms_orb *ex[2] = {nullptr, nullptr};
    ms_keypoints view[2];
    bool have_view[2] = {false, false};
    void *d_frames = nullptr, *bi = nullptr, *bd = nullptr, *sd = nullptr, *match = nullptr;
    std::vector<double> pose_out, point_out;
    std::vector<int32_t> match_host;
    int cap = 0;
    const size_t fbytes = (size_t)W * H;
    bool released = false;
    // the sequence walks its n_distinct images forwards and backwards (0, 1, .., n-1, n-2, .., 1, 0, 1, ..): consecutive frames always overlap
    auto image_of = [&](int i) { if (n_distinct < 2) return 0; const int j = i % (2 * n_distinct - 2); return j < n_distinct ? j : 2 * n_distinct - 2 - j; };
    auto frame = [&](int i, bool count) -> int {
        ms_orb *e = ex[i & 1];
        int rc = ms_orb_extract(e, static_cast<const uint8_t *>(d_frames) + (size_t)image_of(i) * fbytes, 1, 1, fbytes, (size_t)W, nullptr, nullptr, nullptr);
        if (rc != MS_OK) return rc;
        if (!have_view[i & 1]) { rc = ms_orb_device_view(e, &view[i & 1]); if (rc != MS_OK) return rc; have_view[i & 1] = true; }
        if (i) {
            const ms_keypoints &q = view[i & 1], &t = view[(i - 1) & 1];
            rc = ms_hamming_best2_sets(ctx, q.desc, cap, q.count, t.desc, cap, t.count, nullptr, nullptr, 1, static_cast<int32_t *>(bi), static_cast<uint16_t *>(bd), static_cast<uint16_t *>(sd));
            if (rc != MS_OK) return rc;
            rc = ms_ratio_test(ctx, static_cast<int32_t *>(bi), static_cast<uint16_t *>(bd), static_cast<uint16_t *>(sd), cap, ratio, 50, static_cast<int32_t *>(match));
            if (rc != MS_OK) return rc;
        }
        if (n_pose_probs > 0) {                        // poseBundleAdjust of this frame as a NEW problem (mapper_helpers.cpp:1043-1050)
            const ms_ba_problem &P = pose_probs[i % n_pose_probs];
            const auto tp = std::chrono::steady_clock::now();
            ms_ba_result res;
            pose_out.resize(7 * (size_t)P.n_pose);
            rc = ms_ba_solve_host(ctx, &P, pose_out.data(), nullptr, nullptr, &res);
            if (rc != MS_OK) return rc;
            if (count) out.pose_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp).count();
        }
        if (back) { if (count && i % kf_every == 0) back->keyframe(i / kf_every); }
        else if (n_windows > 0 && i % kf_every == 0) {
            const ms_ba_problem &P = windows[(i / kf_every) % n_windows];
            ms_ba *b = nullptr;
            rc = ms_ba_create(ctx, &P, 1, &b);
            if (rc != MS_OK) return rc;
            rc = ms_ba_solve(b);
            ms_ba_result res;
            pose_out.resize(7 * (size_t)P.n_pose); point_out.resize(3 * (size_t)P.n_point);
            if (rc == MS_OK) rc = ms_ba_download(b, 0, pose_out.data(), point_out.data(), nullptr, &res);
            ms_ba_destroy(b);
            if (rc != MS_OK) return rc;
            if (count) ++out.ba_done;
        }
        return MS_OK;
    };
    C5_CHECK(ms_ctx_create(device, &ctx));
    C5_CHECK(ms_dev_alloc(ctx, fbytes * n_distinct, &d_frames));
    C5_CHECK(ms_dev_upload(ctx, d_frames, frames_host, fbytes * n_distinct));
    for (int k = 0; k < 2; ++k) C5_CHECK(ms_orb_create(ctx, &cfg, &ex[k]));
    cap = ms_orb_capacity(ex[0]);
    C5_CHECK(ms_dev_alloc(ctx, 4 * (size_t)cap + 16, &bi));
    C5_CHECK(ms_dev_alloc(ctx, 2 * (size_t)cap + 16, &bd));
    C5_CHECK(ms_dev_alloc(ctx, 2 * (size_t)cap + 16, &sd));
    C5_CHECK(ms_dev_alloc(ctx, 4 * (size_t)cap + 16, &match));
    for (int i = 0; i < (n_frames < 2 ? n_frames : 2); ++i) C5_CHECK(frame(i, false));     // warm-up: lazily built state, first launches
    C5_CHECK(ms_ctx_sync(ctx));
    sh.ready.fetch_add(1);
    released = true;
    while (sh.go.load(std::memory_order_acquire) == 0) std::this_thread::yield();
    {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n_frames; ++i) { C5_CHECK(frame(i, true)); ++out.frames_done; }
        C5_CHECK(ms_ctx_sync(ctx));
        out.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    match_host.resize(cap);
    C5_CHECK(ms_dev_download(ctx, match_host.data(), match, 4 * (size_t)cap));
    for (int v : match_host) out.matches += v >= 0;
done:
    if (!released) sh.ready.fetch_add(1);               // a sequence that failed during set-up must not leave the others waiting
    (void)n_seq;
    for (auto &e : ex) if (e) ms_orb_destroy(e);
    if (ctx) {
        for (void *p : {d_frames, bi, bd, sd, match}) if (p) (void)ms_dev_free(ctx, p);
        ms_ctx_destroy(ctx);
    }
}
}  // namespace

// c5_prepare starts the sequences' threads: each builds its context and handles, uploads its frames, runs two warm-up frames and waits.  frames: n_seq
// pointers to n_distinct * W * H bytes each (a sequence of n_frames frames walks them back and forth) (they and `windows` must stay valid until c5_go returns).  c5_go releases them together, joins them and fills
// the per-sequence outputs (arrays of n_seq); *seconds_out = wall time from the release to the last sequence's end.  Returns MS_OK, or the first failing
// sequence's status with its message in err (err_len bytes).  The job is freed by c5_go.
struct C5Job {
    Shared sh;
    std::vector<Seq> seqs;
    std::vector<std::thread> th;
    ms_orb_config cfg;
};

extern "C" void *c5_prepare(int device, int n_seq, int n_frames, int n_distinct, int W, int H, const uint8_t *const *frames, const ms_ba_problem *windows, int n_windows, int kf_every,
                            int levels, float scale_factor, int max_kpts, int fast_threshold, float lowe_ratio) {
    if (n_seq < 1 || n_frames < 1 || n_distinct < 1 || !frames || kf_every < 1) return nullptr;
    C5Job *J = new C5Job();
    std::memset(&J->cfg, 0, sizeof(J->cfg));
    J->cfg.width = W; J->cfg.height = H; J->cfg.levels = levels; J->cfg.scale_factor = scale_factor; J->cfg.max_kpts = max_kpts; J->cfg.lk_track_level = 0;
    J->cfg.fast_threshold = fast_threshold; J->cfg.max_tracks = 0; J->cfg.max_batch = 1; J->cfg.min_distance = 0;
    J->seqs.resize(n_seq);
    for (int s = 0; s < n_seq; ++s)
        J->th.emplace_back(run_sequence, device, frames[s], n_frames, n_distinct, W, H, windows, n_windows, kf_every, std::cref(J->cfg), lowe_ratio, std::ref(J->sh), n_seq, std::ref(J->seqs[s]),
                           static_cast<const ms_ba_problem *>(nullptr), 0, static_cast<BackEnd *>(nullptr));
    while (J->sh.ready.load() < n_seq) std::this_thread::yield();
    return J;
}

extern "C" int c5_go(void *job, double *seconds_out, double *seq_seconds, int32_t *frames_done, int32_t *ba_done, int32_t *last_matches, char *err, int err_len) {
    C5Job *J = static_cast<C5Job *>(job);
    if (!J) return MS_ERR_INVALID;
    const int n_seq = (int)J->seqs.size();
    const auto t0 = std::chrono::steady_clock::now();
    J->sh.go.store(1, std::memory_order_release);
    for (auto &t : J->th) t.join();
    if (seconds_out) *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int rc = MS_OK;
    for (int s = 0; s < n_seq; ++s) {
        const Seq &q = J->seqs[s];
        if (seq_seconds) seq_seconds[s] = q.seconds;
        if (frames_done) frames_done[s] = q.frames_done;
        if (ba_done) ba_done[s] = q.ba_done;
        if (last_matches) last_matches[s] = q.matches;
        if (q.status != MS_OK && rc == MS_OK) { rc = q.status; if (err && err_len > 0) std::snprintf(err, err_len, "sequence %d: %s", s, q.err); }
    }
    delete J;
    return rc;
}

// The same sequences in the deployment shape: per sequence a front-end thread (extract, match, ratio test, poseBundleAdjust as a new problem per frame) and a back-end
// thread (run_back) fed one keyframe per kf_every frames.  c5p_prepare / c5p_go mirror c5_prepare / c5_go; the outputs add the keyframes handled and the windows'
// median latency and queueing time.
struct C5PJob {
    Shared sh;
    std::vector<Seq> seqs;
    std::vector<BackEnd> backs;
    std::vector<std::thread> fronts, back_threads;
    ms_orb_config cfg;
};
extern "C" void *c5p_prepare(int device, int n_seq, int n_frames, int n_distinct, int W, int H, const uint8_t *const *frames, const ms_ba_problem *pose_probs, int n_pose_probs,
                             const ms_ba_problem *stage1, const ms_ba_problem *stage2, int n_windows, int extra_pose, int kf_every,
                             int levels, float scale_factor, int max_kpts, int fast_threshold, float lowe_ratio) {
    if (n_seq < 1 || n_frames < 1 || n_distinct < 1 || !frames || kf_every < 1 || n_windows < 1 || !stage1 || !stage2) return nullptr;
    C5PJob *J = new C5PJob();
    std::memset(&J->cfg, 0, sizeof(J->cfg));
    J->cfg.width = W; J->cfg.height = H; J->cfg.levels = levels; J->cfg.scale_factor = scale_factor; J->cfg.max_kpts = max_kpts; J->cfg.lk_track_level = 0;
    J->cfg.fast_threshold = fast_threshold; J->cfg.max_tracks = 0; J->cfg.max_batch = 1; J->cfg.min_distance = 0;
    J->seqs.resize(n_seq);
    J->backs = std::vector<BackEnd>(n_seq);
    for (int s = 0; s < n_seq; ++s) {
        J->back_threads.emplace_back(run_back, device, stage1, stage2, n_windows, extra_pose, std::ref(J->sh), std::ref(J->backs[s]));
        J->fronts.emplace_back(run_sequence, device, frames[s], n_frames, n_distinct, W, H, static_cast<const ms_ba_problem *>(nullptr), 0, kf_every, std::cref(J->cfg), lowe_ratio, std::ref(J->sh), n_seq,
                               std::ref(J->seqs[s]), pose_probs, n_pose_probs, &J->backs[s]);
    }
    while (J->sh.ready.load() < 2 * n_seq) std::this_thread::yield();
    return J;
}
extern "C" int c5p_go(void *job, double *seconds_out, double *seq_seconds, double *pose_ms_per_frame, int32_t *keyframes_handled, double *window_ms_median, double *wait_ms_median, char *err, int err_len) {
    C5PJob *J = static_cast<C5PJob *>(job);
    if (!J) return MS_ERR_INVALID;
    const int n_seq = (int)J->seqs.size();
    const auto t0 = std::chrono::steady_clock::now();
    J->sh.go.store(1, std::memory_order_release);
    for (auto &t : J->fronts) t.join();
    if (seconds_out) *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (auto &b : J->backs) b.stop();
    for (auto &t : J->back_threads) t.join();
    int rc = MS_OK;
    std::vector<double> lat, wait;
    int handled = 0;
    double pose_s = 0; long long frames = 0;
    for (int s = 0; s < n_seq; ++s) {
        const Seq &q = J->seqs[s];
        if (seq_seconds) seq_seconds[s] = q.seconds;
        pose_s += q.pose_seconds; frames += q.frames_done;
        handled += (int)J->backs[s].lat_ms.size();
        lat.insert(lat.end(), J->backs[s].lat_ms.begin(), J->backs[s].lat_ms.end());
        wait.insert(wait.end(), J->backs[s].wait_ms.begin(), J->backs[s].wait_ms.end());
        if (q.status != MS_OK && rc == MS_OK) { rc = q.status; if (err && err_len > 0) std::snprintf(err, err_len, "sequence %d: %s", s, q.err); }
        if (J->backs[s].status != MS_OK && rc == MS_OK) { rc = J->backs[s].status; if (err && err_len > 0) std::snprintf(err, err_len, "sequence %d: %s", s, J->backs[s].err); }
    }
    auto median = [](std::vector<double> &v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    if (pose_ms_per_frame) *pose_ms_per_frame = frames ? pose_s / frames * 1e3 : 0.0;
    if (keyframes_handled) *keyframes_handled = handled;
    if (window_ms_median) *window_ms_median = median(lat);
    if (wait_ms_median) *wait_ms_median = median(wait);
    delete J;
    return rc;
}
