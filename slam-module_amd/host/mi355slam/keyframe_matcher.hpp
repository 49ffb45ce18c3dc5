// keyframe_matcher.hpp -- host mirror of the free functions of keyframe_matcher.hpp:33-91.
//
// The reference's Keyframe / MapDB carry the graph; the device only needs flat per-keyframe arrays, so a
// KeyframeFeatures view is built once per keyframe (INTEGRATION.md shows the 15 lines that fill it from
// kf.shared->keyPoints, kf.mapPoints and kf.shared->bowFeatureVec) and uploaded by DeviceKeyframe.
#pragma once
#include <map>
#include <utility>
#include "common.hpp"

namespace mi355slam {

constexpr unsigned HAMMING_DIST_THR_LOW = MS_HAMMING_THR_LOW, HAMMING_DIST_THR_HIGH = MS_HAMMING_THR_HIGH, MAX_HAMMING_DIST = MS_HAMMING_MAX;

struct KeyframeFeatures {
    const KeyPointVector *keyPoints = nullptr;                              // kf.shared->keyPoints
    std::vector<std::uint8_t> usable;                                        // per keypoint: the matcher-specific map-point gate
    std::map<unsigned, std::vector<unsigned>> bowFeatureVec;                  // DBoW2::FeatureVector (ordered node -> keypoint indices)
};

// Device-resident copy of one keyframe's matcher inputs.
class DeviceKeyframe {
public:
    DeviceKeyframe(Context &ctx, const KeyframeFeatures &kf) : ctx_(ctx) {
        const auto &kps = *kf.keyPoints;
        const std::size_t n = kps.size();
        std::vector<std::uint32_t> desc(8 * n); std::vector<float> ang(n); std::vector<std::int32_t> oct(n); std::vector<double> bear(3 * n);
        for (std::size_t i = 0; i < n; ++i) {
            for (int k = 0; k < 8; ++k) desc[8 * i + k] = kps[i].descriptor[k];
            ang[i] = kps[i].angle; oct[i] = kps[i].octave;
            for (int k = 0; k < 3; ++k) bear[3 * i + k] = kps[i].bearing[k];
        }
        std::vector<std::int32_t> node_id, node_start{0}, kp_idx;
        for (const auto &kv : kf.bowFeatureVec) {                             // std::map iterates node ids ascending
            node_id.push_back((std::int32_t)kv.first);
            for (unsigned i : kv.second) kp_idx.push_back((std::int32_t)i);
            node_start.push_back((std::int32_t)kp_idx.size());
        }
        f_.n = (std::int32_t)n;
        f_.desc = up(desc); f_.angle = up(ang); f_.octave = up(oct); f_.bearing = up(bear); f_.usable = up(kf.usable);
        f_.bow.n_nodes = (std::int32_t)node_id.size(); f_.bow.node_id = up(node_id); f_.bow.node_start = up(node_start); f_.bow.kp_idx = up(kp_idx);
    }
    ~DeviceKeyframe() { for (void *p : owned_) ms_dev_free(ctx_.get(), p); }
    DeviceKeyframe(const DeviceKeyframe &) = delete;
    const ms_match_frame &frame() const { return f_; }
private:
    template <typename T> const T *up(const std::vector<T> &v) {
        void *d = nullptr;
        ctx_.check(ms_dev_alloc(ctx_.get(), v.size() * sizeof(T) + 16, &d), "ms_dev_alloc");
        owned_.push_back(d);
        if (!v.empty()) ctx_.check(ms_dev_upload(ctx_.get(), d, v.data(), v.size() * sizeof(T)), "ms_dev_upload");
        return static_cast<const T *>(d);
    }
    Context &ctx_;
    ms_match_frame f_{};
    std::vector<void *> owned_;
};

namespace detail {
inline unsigned run_greedy(Context &ctx, bool triangulation, const DeviceKeyframe &kf1, const DeviceKeyframe &kf2, std::vector<int> &out,
                           float ratio, const double *E12, const std::vector<float> *scaleFactors, float thrDeg) {
    const std::size_t n1 = (std::size_t)kf1.frame().n;
    void *d_m = nullptr, *d_n = nullptr, *d_E = nullptr, *d_sf = nullptr;
    ctx.check(ms_dev_alloc(ctx.get(), 4 * n1 + 16, &d_m), "ms_dev_alloc");
    ctx.check(ms_dev_alloc(ctx.get(), 16, &d_n), "ms_dev_alloc");
    std::int32_t *mptr = static_cast<std::int32_t *>(d_m);
    ms_match_frame f1 = kf1.frame(), f2 = kf2.frame();
    int rc;
    if (triangulation) {
        ctx.check(ms_dev_alloc(ctx.get(), 72, &d_E), "ms_dev_alloc"); ctx.check(ms_dev_upload(ctx.get(), d_E, E12, 72), "upload");
        ctx.check(ms_dev_alloc(ctx.get(), scaleFactors->size() * 4 + 16, &d_sf), "ms_dev_alloc");
        ctx.check(ms_dev_upload(ctx.get(), d_sf, scaleFactors->data(), scaleFactors->size() * 4), "upload");
        rc = ms_match_triangulation(ctx.get(), &f1, &f2, 1, static_cast<const double *>(d_E), static_cast<const float *>(d_sf), thrDeg, 1, &mptr,
                                    static_cast<std::int32_t *>(d_n));
    } else {
        rc = ms_match_loop_closure(ctx.get(), &f1, &f2, 1, ratio, 1, &mptr, static_cast<std::int32_t *>(d_n));
    }
    ctx.check(rc, "greedy matcher");
    std::int32_t num = 0;
    out.assign(n1, -1);
    ctx.check(ms_dev_download(ctx.get(), &num, d_n, 4), "download");
    if (n1) ctx.check(ms_dev_download(ctx.get(), out.data(), d_m, 4 * n1), "download");
    for (void *p : {d_m, d_n, d_E, d_sf}) if (p) ms_dev_free(ctx.get(), p);
    return (unsigned)num;
}
}  // namespace detail

// matchForLoopClosures (keyframe_matcher.hpp:33-40, keyframe_matcher.cpp:50-158).
// usable1 = keypoint has a map point (and it is TRIANGULATED when requireTringulationForLoopClosures, :79-84);
// usable2 = keypoint has a TRIANGULATED map point (:94-96).
inline unsigned matchForLoopClosures(Context &ctx, const DeviceKeyframe &kf1, const DeviceKeyframe &kf2,
                                     std::vector<int> &matchedMapPoints, const Parameters &parameters) {
    return detail::run_greedy(ctx, false, kf1, kf2, matchedMapPoints, parameters.loopClosureFeatureMatchLoweRatio, nullptr, nullptr, 0.f);
}

// matchForTriangulationDBoW (keyframe_matcher.hpp:53, keyframe_matcher.cpp:160-293).  usable = keypoint has NO map point.
// E12 = create_E_21(kf2.R, kf2.t, kf1.R, kf1.t) (keyframe_matcher.cpp:171-175), row-major.
inline std::vector<std::pair<int, int>> matchForTriangulationDBoW(Context &ctx, const DeviceKeyframe &kf1, const DeviceKeyframe &kf2,
                                                                  const double E12[9], const StaticSettings &settings) {
    std::vector<int> m;
    detail::run_greedy(ctx, true, kf1, kf2, m, 0.f, E12, &settings.scaleFactors, settings.parameters.epipolarCheckThresholdDegrees);
    std::vector<std::pair<int, int>> matches;                               // ascending idx_1 (:279-292)
    for (std::size_t i = 0; i < m.size(); ++i) if (m[i] >= 0) matches.emplace_back((int)i, m[i]);
    return matches;
}

// create_E_21 (openvslam/essential_solver.cc:157-162), row-major 3x3
inline void create_E_21(const double R1w[9], const double t1w[3], const double R2w[9], const double t2w[3], double E[9]) {
    double R21[9], t21[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += R2w[3 * i + k] * R1w[3 * j + k]; R21[3 * i + j] = s; }
    for (int i = 0; i < 3; ++i) { double s = 0; for (int k = 0; k < 3; ++k) s += -R21[3 * i + k] * t1w[k]; t21[i] = s + t2w[i]; }
    const double S[9] = {0, -t21[2], t21[1], t21[2], 0, -t21[0], -t21[1], t21[0], 0};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += S[3 * i + k] * R21[3 * k + j]; E[3 * i + j] = s; }
}

}  // namespace mi355slam
