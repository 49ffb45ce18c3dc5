// geometry.cpp -- host-side init-time tables of the ORB front end (product code).
//
// S1/S2 of the scope table: scale factors / sigma^2 (static_settings.cpp:9-24), per-level keypoint
// quotas (static_settings.cpp:39-60), level sizes (image_pyramid.cpp:76-78), the circular-patch
// half-width table (orb_extractor.cpp:174-186) and the fixed-point coefficient tables of
// cv::resize(INTER_LINEAR) on 8U (OpenCV semantics, restated; SURVEY 8a "OpenCV semantics").
// Everything here runs once per extractor on the host; the per-frame arithmetic is all on the GPU.
#include "ms_internal.h"
#include <cmath>

namespace msgeo {

void scale_factors(int levels, float f, float *out) {
    out[0] = 1.0f;
    for (int l = 1; l < levels; ++l) out[l] = f * out[l - 1];   // float32 product chain
}

void level_sigma_sq(int levels, float f, float *out) {
    float at_level = 1.0f;
    out[0] = 1.0f;
    for (int l = 1; l < levels; ++l) {
        at_level = f * at_level;
        out[l] = at_level * at_level;
    }
}

void level_quotas(int levels, float f, int max_kpts, int32_t *out) {
    const double inv = 1.0 / static_cast<double>(f);
    double want = max_kpts * (1.0 - inv) / (1.0 - std::pow(inv, static_cast<double>(levels)));
    int given = 0;
    for (int l = 0; l + 1 < levels; ++l) {
        out[l] = static_cast<int32_t>(std::round(want));
        given += out[l];
        want *= inv;
    }
    out[levels - 1] = std::max(max_kpts - given, 0);
}

void level_sizes(int levels, float f, int w0, int h0, int32_t *w, int32_t *h) {
    std::vector<float> s(levels);
    scale_factors(levels, f, s.data());
    w[0] = w0;
    h[0] = h0;
    for (int l = 1; l < levels; ++l) {
        const double scale = s[l];
        w[l] = static_cast<int32_t>(std::round(w0 * 1.0 / scale));
        h[l] = static_cast<int32_t>(std::round(h0 * 1.0 / scale));
    }
}

void umax(int32_t *u) {
    constexpr int half = 15;   // ORB_FAST_PATCH_HALF_SIZE, static_settings.hpp:16
    const unsigned vmax = static_cast<unsigned>(std::floor(half * std::sqrt(2.0) / 2 + 1));
    const unsigned vmin = static_cast<unsigned>(std::ceil(half * std::sqrt(2.0) / 2));
    for (unsigned v = 0; v <= vmax; ++v)
        u[v] = static_cast<int32_t>(std::round(std::sqrt(double(half * half) - double(v * v))));
    for (unsigned v = half, v0 = 0; vmin <= v; --v) {
        while (u[v0] == u[v0 + 1]) ++v0;
        u[v] = static_cast<int32_t>(v0);
        ++v0;
    }
}

static inline int16_t coef_q11(float c) {
    long r = std::lrintf(c * 2048.f);   // saturate_cast<short>(cvRound(.))
    if (r > 32767) r = 32767;
    if (r < -32768) r = -32768;
    return static_cast<int16_t>(r);
}

void resize_tables(int src_n, int dst_n, bool is_x, std::vector<int16_t> &ofs, std::vector<int16_t> &coef) {
    ofs.resize(dst_n);
    coef.resize(2 * static_cast<size_t>(dst_n));
    const double scale = 1.0 / (static_cast<double>(dst_n) / src_n);
    for (int d = 0; d < dst_n; ++d) {
        float frac = static_cast<float>((d + 0.5) * scale - 0.5);
        int s = static_cast<int>(std::floor(frac));
        frac -= static_cast<float>(s);
        if (is_x) {   // x: clamp the tap pair into the row; y clamps the ROW index instead (resize kernel)
            if (s < 0) { s = 0; frac = 0.f; }
            if (s >= src_n - 1) { s = src_n - 1; frac = 0.f; }
        }
        ofs[d] = static_cast<int16_t>(s);
        coef[2 * d] = coef_q11(1.f - frac);
        coef[2 * d + 1] = coef_q11(frac);
    }
}

}  // namespace msgeo

extern "C" {

int ms_scale_factors(int levels, float f, float *out) {
    if (levels < 1 || levels > MS_MAX_LEVELS || !out) return MS_ERR_INVALID;
    msgeo::scale_factors(levels, f, out);
    return MS_OK;
}
int ms_level_sigma_sq(int levels, float f, float *out) {
    if (levels < 1 || levels > MS_MAX_LEVELS || !out) return MS_ERR_INVALID;
    msgeo::level_sigma_sq(levels, f, out);
    return MS_OK;
}
int ms_level_quotas(int levels, float f, int max_kpts, int32_t *out) {
    if (levels < 1 || levels > MS_MAX_LEVELS || !out || max_kpts < 0) return MS_ERR_INVALID;
    msgeo::level_quotas(levels, f, max_kpts, out);
    return MS_OK;
}
int ms_level_sizes(int levels, float f, int width, int height, int32_t *w, int32_t *h) {
    if (levels < 1 || levels > MS_MAX_LEVELS || !w || !h || width < 1 || height < 1) return MS_ERR_INVALID;
    msgeo::level_sizes(levels, f, width, height, w, h);
    return MS_OK;
}

}  // extern "C"
