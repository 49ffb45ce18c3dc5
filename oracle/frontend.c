/*
 * oracle/frontend.c -- CPU ORACLE (test infrastructure only; see mso.h header).
 * Pyramid geometry, resize, blur, corner detector, orientation, descriptor, extractor driver.
 * Compile with -ffp-contract=off: the float32 steering arithmetic must not be fused (the reference
 * build is plain -O2 x86-64, CMakeLists.txt:4-5, i.e. SSE2 scalar float, no FMA).
 */
#include "mso.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* ------------------------------------------------------------------------------------------ */
/* S1: static_settings.cpp:9-15 -- s_l = f * s_{l-1}, all in float32                           */
void mso_scale_factors(int levels, float f, float *out) {
    for (int l = 0; l < levels; ++l) out[l] = 1.0f;
    for (int l = 1; l < levels; ++l) out[l] = f * out[l - 1];
}

/* S1: static_settings.cpp:16-24 -- sigma^2_l = (f^l)^2 in float32 */
void mso_level_sigma_sq(int levels, float f, float *out) {
    float s = 1.0f;
    for (int l = 0; l < levels; ++l) out[l] = 1.0f;
    for (int l = 1; l < levels; ++l) { s = f * s; out[l] = s * s; }
}

/* S2: static_settings.cpp:39-60 -- geometric quota, std::round, last level takes the remainder */
void mso_level_quotas(int levels, float f, int max_kpts, int *out) {
    const double sf = (double)f;
    double desired = max_kpts * (1.0 - 1.0 / sf) / (1.0 - pow(1.0 / sf, (double)levels));
    int total = 0;
    for (int l = 0; l < levels - 1; ++l) {
        out[l] = (int)round(desired);
        total += out[l];
        desired *= 1.0 / sf;
    }
    int rest = max_kpts - total;
    out[levels - 1] = rest > 0 ? rest : 0;
}

/* image_pyramid.cpp:76-78 -- size from BASE dims and the float32 scale factor, std::round */
void mso_level_sizes(int levels, float f, int w0, int h0, int *w, int *h) {
    float s[MSO_MAX_LEVELS];
    mso_scale_factors(levels, f, s);
    w[0] = w0; h[0] = h0;
    for (int l = 1; l < levels; ++l) {
        const double scale = (double)s[l];
        w[l] = (int)round(w0 * 1.0 / scale);
        h[l] = (int)round(h0 * 1.0 / scale);
    }
}

/* orb_extractor.cpp:174-186 -- circular patch half-widths */
void mso_umax(int *u) {
    const int hp = MSO_HALF_PATCH;
    const unsigned vmax = (unsigned)floor(hp * sqrt(2.0) / 2 + 1);
    const unsigned vmin = (unsigned)ceil(hp * sqrt(2.0) / 2);
    for (unsigned v = 0; v <= vmax; ++v) u[v] = (int)round(sqrt((double)(hp * hp) - (double)(v * v)));
    for (unsigned v = hp, v0 = 0; vmin <= v; --v) {
        while (u[v0] == u[v0 + 1]) ++v0;
        u[v] = (int)v0;
        ++v0;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* P1: cv::resize(..., INTER_LINEAR) on 8U, 1 channel (image_pyramid.cpp:79).                   */
/* OpenCV semantics restated (not in tree, version unpinned): source coord (d+0.5)*scale-0.5 in   */
/* float, 11-bit fixed-point weights via saturate_cast<short>(cvRound), horizontal pass in int,   */
/* vertical pass ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.                               */
static inline short sat_short_round(float v) {
    long r = lrintf(v);                 /* cvRound: round-half-to-even */
    if (r < -32768) r = -32768;
    if (r > 32767) r = 32767;
    return (short)r;
}

void mso_resize_tables(int sn, int dn, int is_x, int *ofs, short *coef) {
    const double inv_scale = (double)dn / sn;
    const double scale = 1.0 / inv_scale;
    for (int d = 0; d < dn; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (is_x) {
            if (s < 0) { f = 0.f; s = 0; }
            if (s >= sn - 1) { f = 0.f; s = sn - 1; }
        }
        ofs[d] = s;
        coef[2 * d] = sat_short_round((1.f - f) * 2048.f);
        coef[2 * d + 1] = sat_short_round(f * 2048.f);
    }
}

void mso_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride,
                          uint8_t *dst, int dw, int dh, int dstride) {
    int *xofs = (int *)malloc(sizeof(int) * dw), *yofs = (int *)malloc(sizeof(int) * dh);
    short *alpha = (short *)malloc(sizeof(short) * 2 * dw), *beta = (short *)malloc(sizeof(short) * 2 * dh);
    mso_resize_tables(sw, dw, 1, xofs, alpha);
    mso_resize_tables(sh, dh, 0, yofs, beta);
    int *row0 = (int *)malloc(sizeof(int) * dw), *row1 = (int *)malloc(sizeof(int) * dw);
    for (int dy = 0; dy < dh; ++dy) {
        int sy0 = yofs[dy], sy1 = yofs[dy] + 1;
        sy0 = sy0 < 0 ? 0 : (sy0 < sh ? sy0 : sh - 1);    /* clip(sy, 0, ssize.height) */
        sy1 = sy1 < 0 ? 0 : (sy1 < sh ? sy1 : sh - 1);
        const uint8_t *S0 = src + (size_t)sy0 * sstride, *S1 = src + (size_t)sy1 * sstride;
        for (int dx = 0; dx < dw; ++dx) {
            const int sx = xofs[dx];
            const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;  /* weight is 0 there */
            const int a0 = alpha[2 * dx], a1 = alpha[2 * dx + 1];
            row0[dx] = S0[sx] * a0 + S0[sx1] * a1;
            row1[dx] = S1[sx] * a0 + S1[sx1] * a1;
        }
        const int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        uint8_t *D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; ++dx) {
            int v = (((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2;
            D[dx] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    free(xofs); free(yofs); free(alpha); free(beta); free(row0); free(row1);
}

/* ------------------------------------------------------------------------------------------ */
/* P2: cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on 8U (image_pyramid.cpp:84).          */
/* Restated as OpenCV's fixed-point 8U path: 8.8 fixed-point taps of exp(-x^2/8)/sum, rounding     */
/* error diffused from the edge inwards with the centre taking the remainder so the taps sum to    */
/* 256; horizontal pass exact in 16 bits, vertical pass exact in 32 bits, one rounding at the end. */
const int mso_gauss7_q8[7] = {18, 34, 48, 56, 48, 34, 18};

void mso_gauss7_derive(int *k7) {   /* derivation of the constants above; checked by the tests */
    double g[7], sum = 0;
    for (int i = 0; i < 7; ++i) { double x = i - 3; g[i] = exp(-0.5 * x * x / 4.0); sum += g[i]; }
    double err = 0; int acc = 0;
    for (int i = 0; i < 3; ++i) {
        double v = g[i] / sum * 256.0 + err;
        int q = (int)floor(v + 0.5);
        err = v - q;
        k7[i] = k7[6 - i] = q;
        acc += 2 * q;
    }
    k7[3] = 256 - acc;
}

static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; }
    return i;
}

void mso_gauss7_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride) {
    uint16_t *tmp = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)w * h);
    const int *k = mso_gauss7_q8;
    for (int y = 0; y < h; ++y) {
        const uint8_t *S = src + (size_t)y * sstride;
        for (int x = 0; x < w; ++x) {
            unsigned a = 0;
            for (int i = -3; i <= 3; ++i) a += (unsigned)k[i + 3] * S[reflect101(x + i, w)];
            tmp[(size_t)y * w + x] = (uint16_t)a;      /* <= 255*256 */
        }
    }
    for (int y = 0; y < h; ++y) {
        uint8_t *D = dst + (size_t)y * dstride;
        for (int x = 0; x < w; ++x) {
            uint32_t a = 0;
            for (int j = -3; j <= 3; ++j) a += (uint32_t)k[j + 3] * tmp[(size_t)reflect101(y + j, h) * w + x];
            D[x] = (uint8_t)((a + 32768u) >> 16);
        }
    }
    free(tmp);
}

/* ------------------------------------------------------------------------------------------ */
/* D1: corner detector.  The reference delegates to tracker::FeatureDetector (feature_detector.cpp */
/* :89-98, not in tree), so the detector core below is THIS BUILD'S definition: FAST-9/16.          */
static const int fast_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int fast_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* score = max over the 16 arcs of 9 contiguous ring pixels of min_i (c - r_i) resp. min_i (r_i - c);
 * the pixel is a FAST-9 corner at threshold t  <=>  score > t.  (= OpenCV cornerScore + 1.) */
int mso_fast_score(const uint8_t *img, int stride, int x, int y) {
    const int c = img[(size_t)y * stride + x];
    int d[25];
    for (int i = 0; i < 16; ++i) d[i] = c - img[(size_t)(y + fast_dy[i]) * stride + x + fast_dx[i]];
    for (int i = 16; i < 25; ++i) d[i] = d[i - 16];
    int best = 0;
    for (int s = 0; s < 16; ++s) {
        int mn = d[s], mx = d[s];
        for (int i = 1; i < 9; ++i) { if (d[s + i] < mn) mn = d[s + i]; if (d[s + i] > mx) mx = d[s + i]; }
        if (mn > best) best = mn;          /* centre brighter than the whole arc */
        if (-mx > best) best = -mx;        /* centre darker than the whole arc */
    }
    return best;
}

static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

int mso_level_min_dist(float g, int w, int h) {      /* feature_detector.cpp:79-82 */
    const int minDim = w < h ? w : h;
    const double su = minDim / 720.0 * 0.8;
    return (int)floor(g * su + 0.5);
}

int mso_detect_level(const uint8_t *img, int w, int h, int stride, int threshold, int quota, int min_dist,
                     int *xs, int *ys, int *scores) {
    if (w < 7 || h < 7 || quota <= 0) return 0;
    uint8_t *sc = (uint8_t *)calloc((size_t)w * h, 1);
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x) {
            /* cheap exact reject (an arc of 9 always covers >= 2 of the 4 compass pixels) */
            const uint8_t *p = img + (size_t)y * stride + x;
            const int c = p[0], hi = c + threshold, lo = c - threshold;
            const int n = p[3 * stride], s_ = p[-3 * stride], e = p[3], w_ = p[-3];
            const int nb = (n > hi) + (s_ > hi) + (e > hi) + (w_ > hi), nd = (n < lo) + (s_ < lo) + (e < lo) + (w_ < lo);
            if (nb < 2 && nd < 2) continue;
            int s = mso_fast_score(img, stride, x, y);
            sc[(size_t)y * w + x] = (uint8_t)(s > threshold ? s : 0);
        }
    uint32_t *keys = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)w * h / 4 + 16);
    size_t nk = 0, cap = (size_t)w * h / 4 + 4;
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x) {
            const int s = sc[(size_t)y * w + x];
            if (!s) continue;
            int ok = 1;
            for (int j = -1; j <= 1 && ok; ++j)
                for (int i = -1; i <= 1; ++i) {
                    if (!i && !j) continue;
                    if (sc[(size_t)(y + j) * w + x + i] >= s) { ok = 0; break; }   /* strict maximum */
                }
            if (ok && nk < cap) keys[nk++] = ((uint32_t)(255 - s) << 24) | (uint32_t)(y * w + x);
        }
    qsort(keys, nk, sizeof(uint32_t), cmp_u32);
    if (min_dist >= 2) {            /* greedy minimum-distance walk over the best min(4*quota, 4096) corners */
        size_t M = (size_t)4 * quota; if (M > 4096) M = 4096; if (M > nk) M = nk;
        size_t kept = 0;
        for (size_t i = 0; i < M && kept < (size_t)quota; ++i) {
            const int idx = (int)(keys[i] & 0xFFFFFFu), x = idx % w, y = idx / w;
            int ok = 1;
            for (size_t j = 0; j < kept; ++j) {
                const int jd = (int)(keys[j] & 0xFFFFFFu), dx = x - jd % w, dy = y - jd / w;
                if (dx * dx + dy * dy < min_dist * min_dist) { ok = 0; break; }
            }
            if (ok) keys[kept++] = keys[i];
        }
        nk = kept;
    }
    if (nk > (size_t)quota) nk = (size_t)quota;     /* maxTracks = quota_l (feature_detector.cpp:39) */
    int n = 0;
    for (size_t i = 0; i < nk; ++i) {
        const int idx = (int)(keys[i] & 0xFFFFFFu), x = idx % w, y = idx / w;
        /* feature_detector.cpp:106-123: border margin applied AFTER detection */
        if (x < MSO_PATCH_RADIUS || y < MSO_PATCH_RADIUS || x >= w - MSO_PATCH_RADIUS || y >= h - MSO_PATCH_RADIUS) continue;
        xs[n] = x; ys[n] = y; if (scores) scores[n] = 255 - (int)(keys[i] >> 24);
        ++n;
    }
    free(keys); free(sc);
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* cv::fastAtan2 restated (OpenCV mathfuncs_core atan_f32; not in tree): degrees in [0,360).      */
float mso_fast_atan2(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* openvslam/trigonometric.h:11-46 */
static const float K_PI = 3.14159265358979f;
static inline float poly_cos(float v) {
    const float c1 = 0.99940307f, c2 = -0.49558072f, c3 = 0.03679168f;
    const float v2 = v * v;
    return c1 + v2 * (c2 + c3 * v2);
}
float mso_cos(float v) {
    const float PI_2 = K_PI / 2.0f, TWO_PI = 2.0f * K_PI, INV_TWO_PI = 1.0f / TWO_PI, THREE_PI_2 = 3.0f * PI_2;
    v = v - (float)(int)floorf(v * INV_TWO_PI) * TWO_PI;     /* cvFloor() returns int */
    v = (0.0f < v) ? v : -v;
    if (v < PI_2) return poly_cos(v);
    else if (v < K_PI) return -poly_cos(K_PI - v);
    else if (v < THREE_PI_2) return -poly_cos(v - K_PI);
    else return poly_cos(TWO_PI - v);
}
float mso_sin(float v) { return mso_cos(K_PI / 2.0f - v); }

/* O1: orb_extractor.cpp:245-275 -- intensity-centroid angle on the UNBLURRED level */
float mso_ic_angle(const uint8_t *img, int stride, int x, int y) {
    static int umax[MSO_HALF_PATCH + 1], init = 0;
    if (!init) { mso_umax(umax); init = 1; }
    int m01 = 0, m10 = 0;
    const uint8_t *center = img + (size_t)y * stride + x;
    for (int u = -MSO_HALF_PATCH; u <= MSO_HALF_PATCH; ++u) m10 += u * center[u];
    for (int v = 1; v <= MSO_HALF_PATCH; ++v) {
        unsigned v_sum = 0;
        const int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            const int vp = center[u + v * stride], vm = center[u - v * stride];
            v_sum += (unsigned)(vp - vm);
            m10 += u * (vp + vm);
        }
        m01 += v * (int)v_sum;
    }
    return mso_fast_atan2((float)m01, (float)m10);
}

const int8_t mso_orb_pattern[1024] = {
#include "orb_pattern.inc"
};

/* O2: orb_extractor.cpp:284-352 (scalar GET_VALUE path, :326-331) on the BLURRED level */
void mso_orb_descriptor(const uint8_t *blur, int stride, int x, int y, float angle_deg, uint32_t desc[8]) {
    const float angle = (float)(angle_deg * M_PI / 180.0);     /* double math, then -> f32 (:286) */
    const float ca = mso_cos(angle), sa = mso_sin(angle);
    const uint8_t *center = blur + (size_t)y * stride + x;
    uint8_t bytes[32];
    for (int i = 0; i < 32; ++i) {
        int val = 0;
        for (int b = 0; b < 8; ++b) {
            const int8_t *p = mso_orb_pattern + (i * 8 + b) * 4;
            const float x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];
            const int r1 = (int)lrintf(x1 * sa + y1 * ca), c1 = (int)lrintf(x1 * ca - y1 * sa);
            const int r2 = (int)lrintf(x2 * sa + y2 * ca), c2 = (int)lrintf(x2 * ca - y2 * sa);
            val |= (center[r1 * stride + c1] < center[r2 * stride + c2]) << b;
        }
        bytes[i] = (uint8_t)val;
    }
    memcpy(desc, bytes, 32);    /* bytes laid little-endian into uint32[8] (:285) */
}

/* ------------------------------------------------------------------------------------------ */
int mso_build_pyramid(const mso_orb_config *cfg, const uint8_t *img, int w, int h, int stride, mso_pyramid *P) {
    if (cfg->levels < 1 || cfg->levels > MSO_MAX_LEVELS) return -1;
    memset(P, 0, sizeof(*P));
    P->levels = cfg->levels;
    mso_level_sizes(cfg->levels, cfg->scale_factor, w, h, P->w, P->h);
    for (int l = 0; l < cfg->levels; ++l) {
        P->img[l] = (uint8_t *)malloc((size_t)P->w[l] * P->h[l]);
        P->blur[l] = (uint8_t *)malloc((size_t)P->w[l] * P->h[l]);
    }
    for (int y = 0; y < h; ++y) memcpy(P->img[0] + (size_t)y * w, img + (size_t)y * stride, w);   /* :75 */
    for (int l = 1; l < cfg->levels; ++l)                                                          /* :76-80 chained */
        mso_resize_linear_u8(P->img[l - 1], P->w[l - 1], P->h[l - 1], P->w[l - 1], P->img[l], P->w[l], P->h[l], P->w[l]);
    for (int l = 0; l < cfg->levels; ++l)                                                          /* :82-85 */
        mso_gauss7_u8(P->img[l], P->w[l], P->h[l], P->w[l], P->blur[l], P->w[l]);
    return 0;
}

void mso_free_pyramid(mso_pyramid *P) {
    for (int l = 0; l < P->levels; ++l) { free(P->img[l]); free(P->blur[l]); P->img[l] = P->blur[l] = NULL; }
}

static int mask_valid(const uint8_t *mask, int w, int h, float px, float py) {
    if (!mask) return 1;
    long xi = lrintf(px), yi = lrintf(py);
    if (xi < 0 || yi < 0 || xi >= w || yi >= h) return 0;
    return mask[(size_t)yi * w + xi] != 0;
}

/* O3: orb_extractor.cpp:73-164 */
int mso_orb_extract(const mso_orb_config *cfg, const uint8_t *img, int w, int h, int stride,
                    const uint8_t *valid_mask,
                    const float *track_xy, const int32_t *track_id, int n_tracks,
                    mso_keypoints *out, int capacity) {
    mso_pyramid P;
    if (mso_build_pyramid(cfg, img, w, h, stride, &P)) return -1;
    float sf[MSO_MAX_LEVELS]; int quota[MSO_MAX_LEVELS];
    mso_scale_factors(cfg->levels, cfg->scale_factor, sf);
    mso_level_quotas(cfg->levels, cfg->scale_factor, cfg->max_kpts, quota);
    int n = 0;
    /* tracker points first (:89-124) */
    for (int t = 0; t < n_tracks && n < capacity; ++t) {
        const int L = cfg->lk_track_level;
        const float px = track_xy[2 * t], py = track_xy[2 * t + 1], scale = sf[L];
        const int x = (int)lrintf(px / scale), y = (int)lrintf(py / scale), m = MSO_PATCH_RADIUS;
        if (x >= m && y >= m && x < P.w[L] - m && y < P.h[L] - m && mask_valid(valid_mask, w, h, px, py)) {
            const float ang = mso_ic_angle(P.img[L], P.w[L], x, y);
            mso_orb_descriptor(P.blur[L], P.w[L], x, y, ang, out->desc + 8 * (size_t)n);
            out->x[n] = px; out->y[n] = py; out->angle[n] = ang; out->octave[n] = L;
            out->track_id[n] = track_id ? track_id[t] : t;
            ++n;
        }
    }
    /* detected points, level-major (:126-163) */
    int maxq = 0;
    for (int l = 0; l < cfg->levels; ++l) if (quota[l] > maxq) maxq = quota[l];
    int *xs = (int *)malloc(sizeof(int) * (maxq + 1)), *ys = (int *)malloc(sizeof(int) * (maxq + 1));
    for (int l = 0; l < cfg->levels; ++l) {
        const int k = mso_detect_level(P.img[l], P.w[l], P.h[l], P.w[l], cfg->fast_threshold, quota[l],
                                       mso_level_min_dist(cfg->min_distance, P.w[l], P.h[l]), xs, ys, NULL);
        for (int i = 0; i < k && n < capacity; ++i) {
            const float fx = (float)xs[i], fy = (float)ys[i];
            if (!mask_valid(valid_mask, w, h, fx * sf[l], fy * sf[l])) continue;   /* dropInvalidKeypoints :221-237 */
            const float ang = mso_ic_angle(P.img[l], P.w[l], xs[i], ys[i]);
            mso_orb_descriptor(P.blur[l], P.w[l], xs[i], ys[i], ang, out->desc + 8 * (size_t)n);
            out->x[n] = fx * sf[l]; out->y[n] = fy * sf[l];                         /* :156 */
            out->angle[n] = ang; out->octave[n] = l; out->track_id[n] = -1;
            ++n;
        }
    }
    free(xs); free(ys);
    mso_free_pyramid(&P);
    out->n = n;
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* Synthetic frame: ramp + per-16x16-block texture + 4096 small blobs, integer only.  */
static inline uint32_t xorshift32(uint32_t s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

void mso_synth_frame(uint8_t *img, int w, int h, uint32_t seed, int shift_x, int shift_y) {
    int16_t *acc = (int16_t *)malloc(sizeof(int16_t) * (size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int ramp = (x * 96) / (w - 1) + (y * 64) / (h - 1);
            const int tx = x + shift_x, ty = y + shift_y;              /* texture translates with the sequence */
            const uint32_t bx = (uint32_t)(tx >> 4), by = (uint32_t)(ty >> 4);
            uint32_t s = seed ^ (bx * 73856093u) ^ (by * 19349663u);
            if (s == 0) s = 0x9E3779B9u;
            const int tex = (int)(xorshift32(s) & 63u);
            acc[(size_t)y * w + x] = (int16_t)(ramp + tex);
        }
    uint32_t s = seed * 2654435761u + 12345u;
    if (s == 0) s = 1;
    for (int k = 0; k < 4096; ++k) {
        s = xorshift32(s); const int cx = (int)(s % (uint32_t)w) - shift_x;
        s = xorshift32(s); const int cy = (int)(s % (uint32_t)h) - shift_y;
        s = xorshift32(s); const int sign = (s & 1u) ? 80 : -80;
        const int r = (k & 1) ? 1 : 2;                                  /* 3x3 (odd k) or 5x5 (even k) */
        for (int j = -r; j <= r; ++j)
            for (int i = -r; i <= r; ++i) {
                const int x = cx + i, y = cy + j;
                if (x < 0 || y < 0 || x >= w || y >= h) continue;
                acc[(size_t)y * w + x] = (int16_t)(acc[(size_t)y * w + x] + sign);
            }
    }
    for (size_t i = 0; i < (size_t)w * h; ++i) { int v = acc[i]; img[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
    free(acc);
}
