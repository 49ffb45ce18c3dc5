// orb_extractor.hpp -- host mirror of slam::OrbExtractor / ImagePyramid / FeatureDetector.
//
//   struct OrbExtractor { static unique_ptr<OrbExtractor> build(const StaticSettings&);
//                         virtual void detectAndExtract(tracker::Image&, const tracker::Camera&,
//                                                       const vector<tracker::Feature>&, KeyPointVector&, vector<int>&); }
//   (orb_extractor.hpp:11-30; implementation orb_extractor.cpp:73-164)
//
// Same call shape, same output order (tracker points first, then level-major), same early return on zero keypoints;
// pyramid + detector are built lazily on the first frame like orb_extractor.cpp:80-81.
#pragma once
#include <cstring>
#include <functional>
#include "common.hpp"

namespace mi355slam {

struct OrbExtractor {
    virtual ~OrbExtractor() = default;
    // cameraValidMask: width x height bytes at level-0 resolution (0 = invalid pixel) or nullptr.  The mask is uploaded when it CHANGES: a new
    // pointer, or new contents behind the same pointer -- the contents are hashed on every call (64 bits over all bytes, ~50 us for 720p), so a
    // caller that edits its mask in place is seen.  A caller that knows when its mask changes passes a version number instead (overload below)
    // and pays nothing per frame.
    virtual void detectAndExtract(const ImageView &img, const std::uint8_t *cameraValidMask,
                                  const std::vector<TrackPoint> &tracks, KeyPointVector &keyPoints,
                                  std::vector<int> &keyPointTrackIds) = 0;
    // the same with the caller's own change counter: the mask is uploaded when (pointer, version) differs from the last call's
    virtual void detectAndExtract(const ImageView &img, const std::uint8_t *cameraValidMask, std::uint64_t maskVersion,
                                  const std::vector<TrackPoint> &tracks, KeyPointVector &keyPoints,
                                  std::vector<int> &keyPointTrackIds) = 0;
    // The same call with the camera model itself: isValidPixel(x, y) is evaluated on the host at exactly the sub-pixel positions the reference
    // passes to tracker::Camera::isValidPixel -- pt.x, pt.y of a tracker point (orb_extractor.cpp:101) and kp.pt.x * scale, kp.pt.y * scale of a
    // detected corner (:231), the same float32 products the output carries (:153-162) -- so the surviving set equals the reference's at the rim of
    // the valid region too.  Dropping after description instead of before changes nothing: no later step depends on the dropped points.
    void detectAndExtract(const ImageView &img, const std::function<bool(float, float)> &isValidPixel,
                          const std::vector<TrackPoint> &tracks, KeyPointVector &keyPoints, std::vector<int> &keyPointTrackIds) {
        detectAndExtract(img, static_cast<const std::uint8_t *>(nullptr), tracks, keyPoints, keyPointTrackIds);
        if (!isValidPixel) return;
        std::size_t n = 0;
        for (std::size_t i = 0; i < keyPoints.size(); ++i)
            if (isValidPixel(keyPoints[i].pt.x, keyPoints[i].pt.y)) { keyPoints[n] = keyPoints[i]; keyPointTrackIds[n] = keyPointTrackIds[i]; ++n; }
        keyPoints.resize(n); keyPointTrackIds.resize(n);
    }
    void detectAndExtract(const ImageView &img, std::nullptr_t, const std::vector<TrackPoint> &tracks, KeyPointVector &keyPoints, std::vector<int> &keyPointTrackIds) {
        detectAndExtract(img, static_cast<const std::uint8_t *>(nullptr), tracks, keyPoints, keyPointTrackIds);     // every pixel valid
    }
    // ImagePyramid::getLevel / getBlurredLevel (image_pyramid.hpp:24-25): CPU-readable copy of one level
    virtual std::vector<std::uint8_t> getLevel(std::size_t level, bool blurred, int &w, int &h) = 0;
    // FeatureDetector::detect (feature_detector.hpp:20-22): per-level corners of the last frame
    virtual std::size_t detections(std::vector<KeyPointVector> &keypointsPerLevel) = 0;
    static std::unique_ptr<OrbExtractor> build(Context &ctx, const StaticSettings &settings);
};

namespace detail {
class OrbExtractorImplementation final : public OrbExtractor {
public:
    OrbExtractorImplementation(Context &ctx, const StaticSettings &s) : ctx_(ctx), settings_(s) {}
    ~OrbExtractorImplementation() override { ms_orb_destroy(orb_); }

    void detectAndExtract(const ImageView &img, const std::uint8_t *mask, const std::vector<TrackPoint> &tracks,
                          KeyPointVector &keypts, std::vector<int> &keyptTrackIds) override {
        run(img, mask, mask ? contentHash(mask, (std::size_t)img.width * img.height) : 0, tracks, keypts, keyptTrackIds);
    }
    void detectAndExtract(const ImageView &img, const std::uint8_t *mask, std::uint64_t maskVersion, const std::vector<TrackPoint> &tracks,
                          KeyPointVector &keypts, std::vector<int> &keyptTrackIds) override {
        run(img, mask, maskVersion, tracks, keypts, keyptTrackIds);
    }

private:
    static std::uint64_t contentHash(const std::uint8_t *p, std::size_t n) {           // multiply-xorshift over 8-byte words, tail bytes one by one
        std::uint64_t h = 0x9E3779B97F4A7C15ull ^ n;
        std::size_t i = 0;
        for (; i + 8 <= n; i += 8) { std::uint64_t w; std::memcpy(&w, p + i, 8); h = (h ^ w) * 0x9E3779B97F4A7C15ull; h ^= h >> 32; }
        for (; i < n; ++i) { h = (h ^ p[i]) * 0x100000001B3ull; }
        return h;
    }
    void run(const ImageView &img, const std::uint8_t *mask, std::uint64_t maskTag, const std::vector<TrackPoint> &tracks,
             KeyPointVector &keypts, std::vector<int> &keyptTrackIds) {
        const auto &p = settings_.parameters;
        if (!orb_) {                                                     // lazily built on the first frame (:80-81)
            ms_orb_config c{img.width, img.height, (int)p.orbScaleLevels, p.orbScaleFactor, (int)p.maxKeypoints,
                            (int)p.orbLkTrackLevel, p.fastThreshold, (int)p.maxTracks, 1, p.gfttMinDistance};
            ctx_.check(ms_orb_create(ctx_.get(), &c, &orb_), "ms_orb_create");
            cap_ = ms_orb_capacity(orb_);
            x_.resize(cap_); y_.resize(cap_); a_.resize(cap_); o_.resize(cap_); t_.resize(cap_); d_.resize(8 * (std::size_t)cap_);
            xy_.assign(2 * (std::size_t)p.maxTracks, 0.f); ids_.assign(p.maxTracks, 0);      // workspace, like orb_extractor.cpp:217-219: nothing is allocated per frame
        }
        if (mask != mask_ || (mask && maskTag != maskTag_)) { ctx_.check(ms_orb_set_valid_mask(orb_, mask), "ms_orb_set_valid_mask"); mask_ = mask; maskTag_ = maskTag; }
        const std::int32_t nt = (std::int32_t)std::min<std::size_t>(tracks.size(), p.maxTracks);
        for (int i = 0; i < nt; ++i) { xy_[2 * i] = tracks[i].x; xy_[2 * i + 1] = tracks[i].y; ids_[i] = tracks[i].id; }
        ctx_.check(ms_orb_extract(orb_, img.data, img.onDevice ? 1 : 0, 1, img.stride * (std::size_t)img.height, img.stride,
                                  p.maxTracks ? xy_.data() : nullptr, p.maxTracks ? ids_.data() : nullptr, p.maxTracks ? &nt : nullptr),
                   "ms_orb_extract");
        std::int32_t n = 0;
        ctx_.check(ms_orb_download(orb_, 0, x_.data(), y_.data(), a_.data(), o_.data(), d_.data(), t_.data(), &n), "ms_orb_download");
        keypts.clear(); keyptTrackIds.clear();
        keypts.reserve(n); keyptTrackIds.reserve(n);
        for (int i = 0; i < n; ++i) {
            KeyPoint kp{};
            kp.pt = {x_[i], y_[i]}; kp.angle = a_[i]; kp.octave = o_[i];
            for (int k = 0; k < 8; ++k) kp.descriptor[k] = d_[8 * (std::size_t)i + k];
            keypts.push_back(kp);
            keyptTrackIds.push_back(t_[i]);
        }
    }

public:
    std::vector<std::uint8_t> getLevel(std::size_t level, bool blurred, int &w, int &h) override {
        std::int32_t ww = 0, hh = 0;
        ctx_.check(ms_orb_level_size(orb_, (int)level, &ww, &hh), "ms_orb_level_size");
        std::vector<std::uint8_t> out((std::size_t)ww * hh);
        ctx_.check(ms_orb_download_level(orb_, 0, (int)level, blurred ? 1 : 0, out.data()), "ms_orb_download_level");
        w = ww; h = hh;
        return out;
    }

    std::size_t detections(std::vector<KeyPointVector> &perLevel) override {
        const unsigned L = settings_.parameters.orbScaleLevels;
        perLevel.assign(L, {});
        std::size_t total = 0;
        std::vector<std::int32_t> x((std::size_t)std::max(cap_, 1)), y(x.size()), s(x.size());
        for (unsigned l = 0; l < L; ++l) {
            std::int32_t n = 0;
            ctx_.check(ms_orb_download_detections(orb_, 0, (int)l, x.data(), y.data(), s.data(), &n), "ms_orb_download_detections");
            for (int i = 0; i < n; ++i) { KeyPoint kp{}; kp.pt = {(float)x[i], (float)y[i]}; kp.angle = 0; kp.octave = (int)l; perLevel[l].push_back(kp); }
            total += n;
        }
        return total;
    }

private:
    Context &ctx_;
    StaticSettings settings_;
    ms_orb *orb_ = nullptr;
    const std::uint8_t *mask_ = nullptr;
    std::uint64_t maskTag_ = 0;
    int cap_ = 0;
    std::vector<float> x_, y_, a_, xy_;
    std::vector<std::int32_t> o_, t_, ids_;
    std::vector<std::uint32_t> d_;
};
}  // namespace detail

inline std::unique_ptr<OrbExtractor> OrbExtractor::build(Context &ctx, const StaticSettings &settings) {
    return std::unique_ptr<OrbExtractor>(new detail::OrbExtractorImplementation(ctx, settings));
}

}  // namespace mi355slam
