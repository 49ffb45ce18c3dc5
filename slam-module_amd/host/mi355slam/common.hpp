// common.hpp -- plain C++ value types shared by the host-side mirrors of the reference surfaces.
//
// The reference's own types (tracker::Image, tracker::Camera, slam::Keyframe, slam::MapDB ...) live in the
// parent project that is not part of the reference tree, so these mirrors use the minimal plain-data
// equivalents below; INTEGRATION.md shows the glue that maps the reference's types onto them.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/mi355slam.h"

namespace mi355slam {

// odometry::ParametersSlam fields read on this path (SURVEY section 5 "Config / flags")
struct Parameters {
    unsigned orbScaleLevels = 8;
    float orbScaleFactor = 1.2f;
    unsigned maxKeypoints = 2000;
    unsigned orbLkTrackLevel = 0;
    int fastThreshold = 20;                       // this build's detector (feature_detector.cpp:89-98 is external)
    unsigned maxTracks = 512;
    float gfttMinDistance = 0.f;                  // tracker.gfttMinDistance (feature_detector.cpp:79-82); 0 = off
    float loopClosureFeatureMatchLoweRatio = 0.75f;
    bool requireTringulationForLoopClosures = true;
    float epipolarCheckThresholdDegrees = 2.0f;
    double odometryPriorStrengthRotation = 100.0, odometryPriorStrengthPosition = 50.0;
    unsigned minVisibleMapPointsInNeighborhoodBA = 0;
};

// slam::StaticSettings (static_settings.hpp:9-21)
struct StaticSettings {
    Parameters parameters;
    std::vector<float> scaleFactors, levelSigmaSq;
    static constexpr unsigned ORB_PATCH_RADIUS = MS_ORB_PATCH_RADIUS;
    explicit StaticSettings(const Parameters &p) : parameters(p), scaleFactors(p.orbScaleLevels), levelSigmaSq(p.orbScaleLevels) {
        ms_scale_factors((int)p.orbScaleLevels, p.orbScaleFactor, scaleFactors.data());
        ms_level_sigma_sq((int)p.orbScaleLevels, p.orbScaleFactor, levelSigmaSq.data());
    }
    std::vector<std::size_t> maxNumberOfKeypointsPerLevel() const {       // static_settings.cpp:39-60
        std::vector<int32_t> q(parameters.orbScaleLevels);
        ms_level_quotas((int)parameters.orbScaleLevels, parameters.orbScaleFactor, (int)parameters.maxKeypoints, q.data());
        return std::vector<std::size_t>(q.begin(), q.end());
    }
};

// slam::KeyPoint (key_point.hpp:11-28) without the Eigen dependency
struct KeyPoint {
    struct Point { float x, y; } pt;
    float angle;
    int octave;
    std::array<double, 3> bearing;                 // filled by the caller (keyframe.cpp:55-68), as in the reference
    using Descriptor = std::array<std::uint32_t, 8>;
    Descriptor descriptor;
};
using KeyPointVector = std::vector<KeyPoint>;

struct TrackPoint { float x, y; int id; };        // tracker::Feature: points[0] and id (orb_extractor.cpp:89-90,:122)

struct ImageView {                                  // tracker::Image stand-in: 8-bit grey image, host or device memory
    const std::uint8_t *data;
    int width, height;
    std::size_t stride;
    bool onDevice;
};

// One device context shared by the objects of a pipeline (the reference runs this path on one backend thread).
class Context {
public:
    explicit Context(int device = 0) {
        if (ms_ctx_create(device, &ctx_) != MS_OK) throw std::runtime_error("mi355slam: no usable gfx950 device (no CPU fallback)");
    }
    ~Context() { if (ws_) ms_dev_free(ctx_, ws_); ms_ctx_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    ms_ctx *get() const { return ctx_; }
    // Errors of the C ABI become exceptions HERE, in the shims, and only for conditions the reference treats as fatal too (it asserts; it has no error
    // codes and no exceptions): no device, a failed HIP call, a capacity given at create time exceeded.  Soft failures keep the reference's
    // conventions (false / 0 / empty set).  Nothing throws across the C ABI itself.
    void check(int rc, const char *what) const {
        if (rc != MS_OK) throw std::runtime_error(std::string(what) + ": " + ms_last_error(ctx_));
    }
    // Device workspace of the shims' per-call tables (queries in, scores out): grows, never shrinks, one allocation instead of a dozen per call.
    // Like every handle it belongs to the one thread that drives this context.
    unsigned char *workspace(std::size_t bytes) {
        if (bytes > wsBytes_) {
            if (ws_) ms_dev_free(ctx_, ws_);
            ws_ = nullptr; wsBytes_ = 0;
            check(ms_dev_alloc(ctx_, 2 * bytes + 256, &ws_), "ms_dev_alloc");
            wsBytes_ = 2 * bytes + 256;
        }
        return static_cast<unsigned char *>(ws_);
    }
    std::vector<unsigned char> &staging() { return stage_; }          // host side of the same tables, reused across calls
private:
    ms_ctx *ctx_ = nullptr;
    void *ws_ = nullptr;
    std::size_t wsBytes_ = 0;
    std::vector<unsigned char> stage_;
};

}  // namespace mi355slam
