// Native driver of the bench's C5 leg: S independent sequences, one host thread + one context (stream) + its own handles per sequence,
// every call through the C ABI of libmi355slam.so -- what a C++ integrator's backend thread does (mapper_helpers.cpp:1011-1131 order):
// per frame extract -> match against the previous frame -> ratio test; on every k-th frame one local bundle adjustment of a NEW window
// (create + solve + download + destroy).  `bench.py --c5-native` calls c5_prepare() / c5_go() through ctypes instead of starting Python threads; it exists to
// show that the C5 numbers are the library's and not the interpreter's: 8 sequences on one MI355X give 3.4 k frames/s + 690 BA/s with either driver (the limit
// is how many small kernels of 8 streams the GPU runs side by side).  Measurement infrastructure only: nothing in the product depends on this file.
#include <atomic>
#include <chrono>
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
    double seconds = 0;
    char err[256] = {0};
};

#define C5_CHECK(call)                                                                                      \
    do {                                                                                                    \
        const int rc_ = (call);                                                                             \
        if (rc_ != MS_OK) { out.status = rc_; std::snprintf(out.err, sizeof(out.err), "%s: %s", #call, ctx ? ms_last_error(ctx) : "no context"); goto done; } \
    } while (0)

void run_sequence(int device, const uint8_t *frames_host, int n_frames, int n_distinct, int W, int H, const ms_ba_problem *windows, int n_windows, int kf_every,
                  const ms_orb_config &cfg, float ratio, Shared &sh, int n_seq, Seq &out) {
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
        if (n_windows > 0 && i % kf_every == 0) {
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
        J->th.emplace_back(run_sequence, device, frames[s], n_frames, n_distinct, W, H, windows, n_windows, kf_every, std::cref(J->cfg), lowe_ratio, std::ref(J->sh), n_seq, std::ref(J->seqs[s]));
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
