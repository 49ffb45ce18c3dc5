// bundle_adjuster.hpp -- host mirror of localBundleAdjust / poseBundleAdjust / globalBundleAdjust (bundle_adjuster.hpp:30-45).
//
// The reference walks MapDB to pick the window (bundle_adjuster.cpp:156-240); that graph walk stays with the caller,
// which hands over the window as flat arrays (BaWindow).  This file reproduces what happens between
// "optimizer.initializeOptimization()" and the write-back: the two-stage schedule (:245-373), the soft orientation prior
// (:341-370), the chi2 > 5.991 outlier rule (:376-388) and the result copy (:114-137), with the optimisation on the GPU.
#pragma once
#include <cmath>
#include <cstdio>
#include <functional>
#include <array>
#include <set>
#include <stdexcept>
#include "common.hpp"

namespace mi355slam {

constexpr float CHI2_THRESHOLD = 5.991f;                                     // bundle_adjuster.cpp:28

// BaStats (ba_stats.hpp:9-84): which kind of bundle adjustment ran in a frame, per frame and in total.  update() is called where
// the reference calls it -- NEIGHBOR when localBundleAdjust stops after stage 1 (bundle_adjuster.cpp:330), LOCAL after stage 2
// (:392), POSE / GLOBAL by the caller of poseBundleAdjust / globalBundleAdjust (mapper_helpers.cpp:1047, :1111) -- and finishFrame()
// by the mapper (mapper.cpp:392, :431).  The reference prints through the parent project's log_info; here the table goes to `sink`.
class BaStats {
public:
    enum class Ba { NONE, POSE, NEIGHBOR, LOCAL, GLOBAL, LAST };             // the interface the reference's call sites use (ba_stats.hpp:11-19)
    explicit BaStats(bool enabled, std::function<void(const char *)> sink = {}) : on_(enabled), sink_(std::move(sink)) {}
    void update(Ba kind) { if (on_) ++frame_[slot(kind)]; }
    // end of a frame: a frame in which nothing ran counts as "none"; prints the table the reference prints (kind, this frame, since the start) and clears the frame's counters
    void finishFrame() {
        if (!on_) return;
        bool ran = false;
        for (int n : frame_) ran = ran || n != 0;
        if (!ran) frame_[slot(Ba::NONE)] = 1;
        Row total_row{};
        print("", nullptr);
        print("TYPE   \tNUM\tTOTAL", nullptr);
        for (size_t k = 0; k < kKinds; ++k) {
            total_[k] += frame_[k];
            const Row r{frame_[k], total_[k]};
            total_row.frame += r.frame; total_row.total += r.total;
            print(kLabel[k], &r);
        }
        print(kLabel[kKinds], &total_row);
        frame_.fill(0);
    }
    int frameCount(Ba kind) const { return on_ ? frame_[slot(kind)] : 0; }      // counters since the last finishFrame (not in the reference: for tests)
    int totalCount(Ba kind) const { return on_ ? total_[slot(kind)] : 0; }
private:
    struct Row { int frame, total; };
    static constexpr size_t kKinds = static_cast<size_t>(Ba::LAST);
    static constexpr const char *kLabel[kKinds + 1] = {"none     ", "pose     ", "neighbor ", "local    ", "global   ", "TOTAL    "};
    static size_t slot(Ba kind) { return static_cast<size_t>(kind) < kKinds ? static_cast<size_t>(kind) : throw std::out_of_range("BaStats: not a kind of bundle adjustment"); }
    void print(const char *label, const Row *r) const {
        char line[96];
        if (r) std::snprintf(line, sizeof(line), "%s\t%d\t%d", label, r->frame, r->total); else std::snprintf(line, sizeof(line), "%s", label);
        if (sink_) sink_(line); else std::printf("%s\n", line);
    }
    bool on_;
    std::function<void(const char *)> sink_;
    std::array<int, kKinds> frame_{}, total_{};
};

// WorkspaceBA (bundle_adjuster.hpp:16-25): the id sets localBundleAdjust refills on every call (bundle_adjuster.cpp:158-223; MpId / KfId
// are plain ints here, id.hpp:9-24) and the statistics.  The window selection that fills the sets walks MapDB and stays with the caller.
struct WorkspaceBA {
    std::set<int> localMpIds;
    std::set<int, std::greater<int>> localKfIds;
    BaStats baStats;
    explicit WorkspaceBA(bool enableBaStats) : baStats(enableBaStats) {}
};

struct BaWindow {
    // vertices
    std::vector<std::array<double, 7>> poses;      // world->camera, qx,qy,qz,qw,tx,ty,tz  (matrixToPose(kf.poseCW), :39-41,:250)
    std::vector<std::array<double, 3>> points;     // mapPoint.position (:266)
    int currentKeyframe = 0;                       // index into poses of `keyframe` (the only free pose in stage 1, :252)
    // EdgeSE3ProjectXYZ per observation (:272-290, setMapPointMeasurement :43-63)
    std::vector<std::int32_t> obsPose, obsPoint;
    std::vector<std::array<double, 2>> obsUv;      // bearing.xy / bearing.z
    std::vector<double> obsInfo;                   // focal^2 / levelSigmaSq[octave]
    // EdgeSE3Expmap: odometry chain (:296-311) and loop closures (:314-319)
    std::vector<std::int32_t> edgeI, edgeJ;
    std::vector<std::array<double, 7>> edgeMeas;
    std::vector<std::array<double, 36>> edgeInfo;
};

struct BaOutcome {
    bool ran = false;
    std::vector<std::uint8_t> outlier;             // per observation: chi2 > 5.991 after stage 2 (:376-388)
    ms_ba_result stage1{}, stage2{};
};

namespace detail {
struct TraceRange {                                       // a roctx range named like the reference's timer (no-op unless ms_set_trace_ranges(1))
    explicit TraceRange(const char *name) { ms_trace_range_push(name); }
    ~TraceRange() { ms_trace_range_pop(); }
};
inline ms_ba_problem as_problem(const BaWindow &w, const std::vector<std::uint8_t> &poseFixed, const std::vector<std::uint8_t> *pointFixed, int iters) {
    ms_ba_problem p{};
    p.n_pose = (std::int32_t)w.poses.size(); p.n_point = (std::int32_t)w.points.size(); p.n_obs = (std::int32_t)w.obsPose.size();
    p.n_pose_edge = (std::int32_t)w.edgeI.size();
    p.pose = w.poses.empty() ? nullptr : w.poses[0].data(); p.pose_fixed = poseFixed.data();
    p.point = w.points.empty() ? nullptr : w.points[0].data(); p.point_fixed = pointFixed ? pointFixed->data() : nullptr;
    p.obs_pose = w.obsPose.data(); p.obs_point = w.obsPoint.data(); p.obs_uv = w.obsUv.empty() ? nullptr : w.obsUv[0].data(); p.obs_info = w.obsInfo.data();
    p.huber_delta = std::sqrt(CHI2_THRESHOLD);                              // rk->setDelta(std::sqrt(CHI2_THRESHOLD)) (:56)
    p.edge_i = w.edgeI.data(); p.edge_j = w.edgeJ.data();
    p.edge_meas = w.edgeMeas.empty() ? nullptr : w.edgeMeas[0].data(); p.edge_info = w.edgeInfo.empty() ? nullptr : w.edgeInfo[0].data();
    p.max_iters = iters;
    return p;
}
}  // namespace detail

// localBundleAdjust (bundle_adjuster.cpp:141-394) on a prepared window; updates w.poses / w.points in place.
inline BaOutcome localBundleAdjust(Context &ctx, BaWindow &w, int problemMaxSize, const Parameters &parameters, bool neighbourhoodStage = true,
                                   WorkspaceBA *workspace = nullptr) {
    detail::TraceRange range("localBundleAdjust");                            // timer(slam::TIME_STATS, "localBundleAdjust"), mapper_helpers.cpp:1080
    BaOutcome out;
    const int iterations = static_cast<int>(1 + std::sqrt(static_cast<double>(problemMaxSize)));      // :156
    std::vector<double> chi2(w.obsPose.size());
    // stage 1: fix all but the current keyframe (:251-252), all points free (:268)
    std::vector<std::uint8_t> fixed(w.poses.size(), 1);
    fixed[w.currentKeyframe] = 0;
    ms_ba_problem p1 = detail::as_problem(w, fixed, nullptr, iterations);
    if (!neighbourhoodStage) {                                                 // "Skip neighbordhood BA" (:326-332)
        ctx.check(ms_ba_solve_host(ctx.get(), &p1, w.poses[0].data(), w.points.empty() ? nullptr : w.points[0].data(), chi2.data(), &out.stage1), "ms_ba_solve_host");
        out.ran = true;
        if (workspace) workspace->baStats.update(BaStats::Ba::NEIGHBOR);
        return out;
    }
    // stage 2: unfix every keyframe (:335-337); soft orientation prior against the stage-1 pose (:341-370).  Its graph differs from stage 1's only in
    // the fixed flags, one more (fixed) vertex and one more edge, none of which depends on stage 1's result: both handles are built up front and the
    // stage-1 state moves to stage 2 on the device (ms_ba_copy_state) -- one download per window instead of two downloads and an upload
    BaWindow w2 = w;
    w2.poses.push_back(w.poses[w.currentKeyframe]);                            // conv.custom(0): fixed copy of the just-optimised pose (value set on the device)
    std::vector<std::uint8_t> fixed2(w2.poses.size(), 0);
    fixed2.back() = 1;
    const double r = 100 * parameters.odometryPriorStrengthRotation;          // :364
    std::array<double, 36> info{};
    for (int i = 0; i < 3; ++i) info[6 * i + i] = r * r;                       // rotation block r^2 I, translation block 0 (:366-367)
    w2.edgeI.push_back((std::int32_t)w2.poses.size() - 1);                     // vertex 0 = the fixed copy (:355)
    w2.edgeJ.push_back(w.currentKeyframe);                                     // vertex 1 = the keyframe (:356)
    w2.edgeMeas.push_back({0, 0, 0, 1, 0, 0, 0});                              // identity measurement (:357)
    w2.edgeInfo.push_back(info);
    ms_ba_problem p2 = detail::as_problem(w2, fixed2, nullptr, iterations);
    struct Handle { ms_ba *h = nullptr; ~Handle() { ms_ba_destroy(h); } } b1, b2;
    ctx.check(ms_ba_create(ctx.get(), &p1, 1, &b1.h), "ms_ba_create");
    ctx.check(ms_ba_solve(b1.h), "ms_ba_solve");                               // stage 1 runs on the device (k_ba_one_pose, ~0.3 ms) ...
    ctx.check(ms_ba_create(ctx.get(), &p2, 1, &b2.h), "ms_ba_create");         // ... while the host builds stage 2's index structures (~0.3 ms)
    const std::int32_t cur = w.currentKeyframe;
    ctx.check(ms_ba_copy_state(b2.h, b1.h, &cur), "ms_ba_copy_state");
    ctx.check(ms_ba_solve(b2.h), "ms_ba_solve");
    ctx.check(ms_ba_download(b1.h, 0, nullptr, nullptr, nullptr, &out.stage1), "ms_ba_download");      // a failed stage 1 stops here, the window untouched
    if (ms_ba_team_fallbacks(b1.h) > 0) {                                      // stage 1 was solved again (a team barrier had given up): stage 2 started from a state
        ctx.check(ms_ba_copy_state(b2.h, b1.h, &cur), "ms_ba_copy_state");     // that is gone -- chain it again from the repeated solve
        ctx.check(ms_ba_solve(b2.h), "ms_ba_solve");
    }
    out.ran = true;
    ctx.check(ms_ba_download(b2.h, 0, w2.poses[0].data(), w2.points.empty() ? nullptr : w2.points[0].data(), chi2.data(), &out.stage2), "ms_ba_download");
    for (std::size_t i = 0; i < w.poses.size(); ++i) w.poses[i] = w2.poses[i];    // applyBundleAdjustResults (:114-137)
    w.points = w2.points;
    out.outlier.resize(chi2.size());
    for (std::size_t i = 0; i < chi2.size(); ++i) out.outlier[i] = chi2[i] > CHI2_THRESHOLD;   // :378
    if (workspace) workspace->baStats.update(BaStats::Ba::LOCAL);             // :392
    return out;
}

// poseBundleAdjust (bundle_adjuster.cpp:396-491): one free pose, every point fixed (:465), parameters.poseBAIterations.
inline bool poseBundleAdjust(Context &ctx, BaWindow &w, int poseBAIterations, ms_ba_result *res = nullptr) {
    detail::TraceRange range("poseBundleAdjust");                             // mapper_helpers.cpp:1044
    if (w.obsPose.empty()) return false;                                       // :410-412
    std::vector<std::uint8_t> fixed(w.poses.size(), 1), pfixed(w.points.size(), 1);
    fixed[w.currentKeyframe] = 0;
    ms_ba_problem p = detail::as_problem(w, fixed, &pfixed, poseBAIterations);
    ms_ba_result r{};
    ctx.check(ms_ba_solve_host(ctx.get(), &p, w.poses[0].data(), nullptr, nullptr, &r), "ms_ba_solve_host");
    if (res) *res = r;
    return true;
}

// globalBundleAdjust (bundle_adjuster.cpp:493-604): every keyframe of the map with only the current one fixed (:515), every
// observed map point free (:531), the odometry chain (:554-571) and the loop-closure edges (:574-578), ONE optimisation of
// parameters.globalBAIterations, then the chi2 > 5.991 outlier rule (:584-597) and the write-back of everything (:599-604).
// `w` holds the whole map as flat arrays (w.currentKeyframe = the fixed one); up to 2048 free keyframes (the device spreads the
// factorisation of systems beyond 176 poses over a team of workgroups).
inline BaOutcome globalBundleAdjust(Context &ctx, BaWindow &w, int globalBAIterations) {
    detail::TraceRange range("globalBundleAdjust");
    BaOutcome out;
    std::vector<std::uint8_t> fixed(w.poses.size(), 0);
    fixed[w.currentKeyframe] = 1;
    std::vector<double> chi2(w.obsPose.size());
    ms_ba_problem p = detail::as_problem(w, fixed, nullptr, globalBAIterations);
    ctx.check(ms_ba_solve_host(ctx.get(), &p, w.poses[0].data(), w.points.empty() ? nullptr : w.points[0].data(), chi2.data(), &out.stage1), "ms_ba_solve_host");
    out.ran = true;
    out.outlier.resize(chi2.size());
    for (std::size_t i = 0; i < chi2.size(); ++i) out.outlier[i] = chi2[i] > CHI2_THRESHOLD;   // :586
    return out;
}

}  // namespace mi355slam
