// bundle_adjuster.hpp -- host mirror of localBundleAdjust / poseBundleAdjust / globalBundleAdjust (bundle_adjuster.hpp:30-45).
//
// The reference walks MapDB to pick the window (bundle_adjuster.cpp:156-240); that graph walk stays with the caller,
// which hands over the window as flat arrays (BaWindow).  This file reproduces what happens between
// "optimizer.initializeOptimization()" and the write-back: the two-stage schedule (:245-373), the soft orientation prior
// (:341-370), the chi2 > 5.991 outlier rule (:376-388) and the result copy (:114-137), with the optimisation on the GPU.
#pragma once
#include <cmath>
#include "common.hpp"

namespace mi355slam {

constexpr float CHI2_THRESHOLD = 5.991f;                                     // bundle_adjuster.cpp:28

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
inline BaOutcome localBundleAdjust(Context &ctx, BaWindow &w, int problemMaxSize, const Parameters &parameters, bool neighbourhoodStage = true) {
    BaOutcome out;
    const int iterations = static_cast<int>(1 + std::sqrt(static_cast<double>(problemMaxSize)));      // :156
    std::vector<double> chi2(w.obsPose.size());
    // stage 1: fix all but the current keyframe (:251-252), all points free (:268)
    std::vector<std::uint8_t> fixed(w.poses.size(), 1);
    fixed[w.currentKeyframe] = 0;
    ms_ba_problem p1 = detail::as_problem(w, fixed, nullptr, iterations);
    ctx.check(ms_ba_solve_host(ctx.get(), &p1, w.poses[0].data(), w.points.empty() ? nullptr : w.points[0].data(), chi2.data(), &out.stage1), "ms_ba_solve_host");
    out.ran = true;
    if (!neighbourhoodStage) return out;                                       // "Skip neighbordhood BA" (:326-332)
    // stage 2: unfix every keyframe (:335-337); soft orientation prior against the stage-1 pose (:341-370)
    BaWindow w2 = w;
    w2.poses.push_back(w.poses[w.currentKeyframe]);                            // conv.custom(0): fixed copy of the just-optimised pose
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
    ctx.check(ms_ba_solve_host(ctx.get(), &p2, w2.poses[0].data(), w2.points.empty() ? nullptr : w2.points[0].data(), chi2.data(), &out.stage2), "ms_ba_solve_host");
    for (std::size_t i = 0; i < w.poses.size(); ++i) w.poses[i] = w2.poses[i];    // applyBundleAdjustResults (:114-137)
    w.points = w2.points;
    out.outlier.resize(chi2.size());
    for (std::size_t i = 0; i < chi2.size(); ++i) out.outlier[i] = chi2[i] > CHI2_THRESHOLD;   // :378
    return out;
}

// poseBundleAdjust (bundle_adjuster.cpp:396-491): one free pose, every point fixed (:465), parameters.poseBAIterations.
inline bool poseBundleAdjust(Context &ctx, BaWindow &w, int poseBAIterations, ms_ba_result *res = nullptr) {
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
