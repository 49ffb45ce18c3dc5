// orb.hip -- image pyramid + ORB extraction for gfx950 (MI355X), batched over frames.
//
// Replaces, behind the C ABI of include/mi355slam.h:
//   ImagePyramid::update            image_pyramid.cpp:68-86     -> k_resize (chained), k_blur
//   FeatureDetector::detect         feature_detector.cpp:68-134 -> k_fast, k_select
//   OrbExtractor::detectAndExtract  orb_extractor.cpp:73-164    -> k_tracks, k_describe
//
// Data layout in HBM (one extractor, batch of B frames):
//   level 0      : the caller's device image used in place (or uploaded once into the slab)
//   slab[f]      : levels 1..n-1 and ALL blurred levels of frame f, each plane h x pitch bytes,
//                  pitch = w rounded up to 64 B so every row starts 64-B aligned
//   cand[f][l]   : FAST corner keys of level l (post 3x3 NMS), unordered; capacity w*h/4 (the
//                  strict-maximum NMS cannot keep more than one pixel per 2x2 block)
//   det[f][l]    : the selected corners of level l in key order (score desc, y*w+x asc)
//   out[f]       : final keypoints, structure of arrays, tracker points first then level-major
//
// All pixel arithmetic is integer; the float32 steps of orientation/steering use explicit
// round-to-nearest intrinsics (no FMA contraction) so results are bit-identical to the reference's
// x86-64 SSE2 scalar build (CMakeLists.txt:4-5: -O2, no -march).
#include "ms_internal.h"
#include <cfloat>
#include <cmath>
#include <cstring>

namespace {

constexpr int kPatchRadius = MS_ORB_PATCH_RADIUS;   // 19
constexpr int kHalfPatch = 15;                       // ORB_FAST_PATCH_HALF_SIZE
constexpr int kMaxQuota = 4096;                      // per-level selection capacity (LDS sort)

struct LevelGeom {
    int32_t w, h, pitch, quota;
    int32_t min_dist;                 // per-level minimum keypoint distance (0/1 = none), feature_detector.cpp:79-82
    int32_t det_base;                 // first slot of this level in det arrays (prefix of quotas)
    int32_t cand_cap;                 // capacity of this level's candidate list (entries)
    int32_t btiles_x, btile_base;     // k_blur tile table (248 x 72 tiles: 4 waves x kBlurRows rows)
    int32_t ftiles_x, ftile_base;     // k_fast tile table (248 x 30 tiles: 8 waves x 4 position rows, 2 of them halo)
    uint32_t btiles_inv, ftiles_inv;  // ceil(2^32 / tiles_x): row of a tile = mulhi(t, inv), exact for t < 2^16
    uint64_t img_off, blur_off;       // byte offsets inside a frame slab
    uint64_t cand_off;                // entry offset inside a frame's candidate buffer
    float scale;                      // scaleFactors[l] (float32 chain)
};

struct PyrGeom {
    int32_t levels, lk_level, fast_threshold, max_kpts, max_tracks, capacity;
    int32_t det_stride;               // per-frame stride of the det arrays = max(max_kpts, sum of the level quotas): the quotas are rounded per level
                                      // (static_settings.cpp:52) and can add up to more than max_kpts (e.g. 15 levels / 160 keypoints -> 161)
    int32_t width, height, btiles_total, ftiles_total;
    uint64_t slab_stride, cand_stride;     // per frame: bytes / entries
    int32_t umax[16];
    LevelGeom L[MS_MAX_LEVELS];
};

// first tile id of every level, passed BY VALUE (kernel-argument SGPRs): finding a block's level must not be a chain of
// dependent loads from the geometry table (that chain was a third of k_fast's wave time)
struct TileMap { int32_t base[MS_MAX_LEVELS + 1]; };
__device__ __forceinline__ int tile_level(const TileMap &tm, int levels, int t) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < MS_MAX_LEVELS; ++k) l += t >= tm.base[k] ? 1 : 0;     // base[k >= levels] = tile total > t: a compare and an add-with-carry per level
    return l;
}

// What the tiled kernels (k_blur, k_fast) need about a level, also BY VALUE in the kernel arguments: one batch of scalar loads from the
// kernel-argument segment once the block knows its level, instead of a chain of dependent loads from the geometry table behind branches.
struct TileLevel { int32_t w, h, pitch, btiles_x, ftiles_x, cand_cap; uint32_t btiles_inv, ftiles_inv; uint64_t img_off, blur_off, cand_off; float scale; int32_t det_base; };
struct TileLevels { TileLevel L[MS_MAX_LEVELS]; uint64_t slab_stride, cand_stride; int32_t levels, fast_threshold; };

struct FrameSrc {          // where pyramid level 0 lives for this call
    const uint8_t *lvl0;
    uint64_t lvl0_frame_stride;
    int32_t lvl0_pitch;
    uint8_t *slab;
};

__device__ __forceinline__ const uint8_t *level_ptr(const FrameSrc &s, const PyrGeom *g, int f, int l, int &pitch) {
    if (l == 0) { pitch = s.lvl0_pitch; return s.lvl0 + (uint64_t)f * s.lvl0_frame_stride; }
    pitch = g->L[l].pitch;
    return s.slab + (uint64_t)f * g->slab_stride + g->L[l].img_off;
}
__device__ __forceinline__ uint8_t *blur_ptr(const FrameSrc &s, const PyrGeom *g, int f, int l) {
    return s.slab + (uint64_t)f * g->slab_stride + g->L[l].blur_off;
}

// ------------------------------------------------------------------------------------------------
// copy a non-aligned caller image into the slab's level-0 plane (only when in-place use is impossible)
__global__ __launch_bounds__(256) void k_copy_level0(const uint8_t *src, uint64_t frame_stride, uint64_t row_stride,
                                                     uint8_t *slab, uint64_t slab_stride, uint64_t off, int w, int h, int pitch) {
    const int f = blockIdx.z, y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x < w && y < h) slab[(uint64_t)f * slab_stride + off + (uint64_t)y * pitch + x] = src[(uint64_t)f * frame_stride + (uint64_t)y * row_stride + x];
}

// ------------------------------------------------------------------------------------------------
// P1: cv::resize INTER_LINEAR, 8U (image_pyramid.cpp:79).  Level l from level l-1 of the same frame.
// 4 destination pixels per lane, one dword store.  Per source row a lane fetches three aligned dwords
// (12 bytes cover the <= 8-byte tap window of its 4 pixels for scale factors up to 2) and funnel-shifts
// them into a 64-bit window; the per-column (offset, a0, a1) and per-row (row0, row1, b0, b1) tables are
// packed so a lane needs two 16-byte table loads.  HBM-bound by design: reads level l-1, writes level l.
// unaligned dword / qword accesses (global memory takes them)
struct __attribute__((packed, aligned(1))) U32u { uint32_t v; };
struct __attribute__((packed, aligned(1))) U64u { uint32_t lo, hi; };
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
typedef short s2_t __attribute__((ext_vector_type(2)));

// Everything k_resize needs about its two levels, by value in the kernel arguments: the launch is per level, and reading the
// geometry table instead put nine dependent scalar round trips in front of every wave's first pixel load.
struct ResizeArgs {
    const uint8_t *src; uint64_t src_frame_stride; int32_t src_pitch, sw;      // level l-1
    uint8_t *dst; uint64_t dst_frame_stride; int32_t dst_pitch, dw, dh;        // level l
};

struct ResizeTab {
    const int16_t *xtab;   // [dw][4] : sx, a0, a1, 0
    const int16_t *ytab;   // [dh][4] : sy0, sy1 (clamped rows), b0, b1
};

__device__ __forceinline__ int resize_px(uint64_t w0, uint64_t w1, int k, int a0, int a1, int b0, int b1) {
    const int r0 = (int)((w0 >> (8 * k)) & 255) * a0 + (int)((w0 >> (8 * k + 8)) & 255) * a1;
    const int r1 = (int)((w1 >> (8 * k)) & 255) * a0 + (int)((w1 >> (8 * k + 8)) & 255) * a1;
    const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    return min(max(v, 0), 255);
}

constexpr int kResizeRows = 5;   // destination rows per lane: the table fetch is paid once and 10 source windows (30 dwords) are in flight per lane

struct Window3 { uint32_t d0, d1, d2; };
typedef uint32_t u3_t __attribute__((ext_vector_type(3)));
typedef u3_t u3a_t __attribute__((aligned(4)));       // a 12-byte load from a dword-aligned address
__device__ __forceinline__ Window3 window_load(const uint8_t *row, int base, int last_dword) {
    return {*reinterpret_cast<const uint32_t *>(row + min(base, last_dword)), *reinterpret_cast<const uint32_t *>(row + min(base + 4, last_dword)),
            *reinterpret_cast<const uint32_t *>(row + min(base + 8, last_dword))};
}

template <bool WIDE>   // WIDE: tap window of 4 pixels may exceed 8 bytes (scale factor > 2) -> per-tap byte loads
__global__ __launch_bounds__(256) void k_resize(ResizeArgs R, ResizeTab T) {
    const struct { int w, h, pitch; } D = {R.dw, R.dh, R.dst_pitch};
    const int sw = R.sw;
    const int f = blockIdx.z;
    const int dy0 = (blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * kResizeRows;   // wave-uniform (SGPR): row addresses stay scalar
    const int dx0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    if (dy0 >= D.h || dx0 >= D.w) return;
    const int spitch = R.src_pitch;
    const uint8_t *S = R.src + (uint64_t)f * R.src_frame_stride;
    uint8_t *dst = R.dst + (uint64_t)f * R.dst_frame_stride;
    // both tables are padded to whole groups (last entry repeated), so a lane's four columns are two 16-byte loads
    const uint4 xa = reinterpret_cast<const uint4 *>(T.xtab)[dx0 >> 1], xb = reinterpret_cast<const uint4 *>(T.xtab)[(dx0 >> 1) + 1];
    const short4 xt[4] = {__builtin_bit_cast(short4, make_uint2(xa.x, xa.y)), __builtin_bit_cast(short4, make_uint2(xa.z, xa.w)),
                          __builtin_bit_cast(short4, make_uint2(xb.x, xb.y)), __builtin_bit_cast(short4, make_uint2(xb.z, xb.w))};
    short4 yt[kResizeRows];
#pragma unroll
    for (int r = 0; r < kResizeRows; ++r) yt[r] = reinterpret_cast<const short4 *>(T.ytab)[dy0 + r];
    if (!WIDE) {
        const int sx0 = xt[0].x, last = (sw - 1) & ~3, base = sx0 & ~3;
        // per column: the two taps as one v_dot2 operand (a0, a1), and a v_perm selector that lifts source bytes k, k+1 of the
        // row's 8-byte window into 16-bit lanes (k = sx - sx0 <= 6)
        uint32_t A[4], sel[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            A[i] = (uint32_t)(uint16_t)xt[i].y | ((uint32_t)(uint16_t)xt[i].z << 16);
            sel[i] = 0x0C010C00u + (uint32_t)(xt[i].x - sx0) * 0x00010001u;
        }
        Window3 W0[kResizeRows], W1[kResizeRows];
        // The kernel is bound by the NUMBER of vector-memory instructions (measured: dropping the arithmetic changes nothing, halving the
        // loads gives -36 %), so the three dwords of a window are ONE 12-byte load wherever no lane of the wave touches the row's last
        // dwords (a 12-byte load there would run past the row, and past the caller's buffer on the last row of level 0).
        const bool edge = __ballot(base + 8 > last) != 0;        // wave-uniform
        if (!edge || last >= 8) {
            // edge wave: the 12 bytes are fetched from min(base, last - 8) and moved down by whole dwords afterwards, which reproduces the
            // clamped dwords exactly (d_k = row[min(base + 4k, last)])
            const int from = edge ? min(base, last - 8) : base, shift = base - from;       // shift = 0, 4 or 8 bytes
#pragma unroll
            for (int r = 0; r < kResizeRows; ++r) {
                const uint8_t *p0 = S + (uint64_t)yt[r].x * spitch + from, *p1 = S + (uint64_t)yt[r].y * spitch + from;
                asm volatile("" : "+v"(p0), "+v"(p1));        // keeps the compiler from folding this branch into the clamped one below (equal values, three loads each)
                const u3_t a = *(const __attribute__((address_space(1))) u3a_t *)(p0), b = *(const __attribute__((address_space(1))) u3a_t *)(p1);   // global_load_dwordx3 (a generic pointer would become a flat load)
                W0[r] = {a.x, a.y, a.z};
                W1[r] = {b.x, b.y, b.z};
            }
            if (edge) {
#pragma unroll
                for (int r = 0; r < kResizeRows; ++r) {
                    W0[r] = {shift == 0 ? W0[r].d0 : shift == 4 ? W0[r].d1 : W0[r].d2, shift == 0 ? W0[r].d1 : W0[r].d2, W0[r].d2};
                    W1[r] = {shift == 0 ? W1[r].d0 : shift == 4 ? W1[r].d1 : W1[r].d2, shift == 0 ? W1[r].d1 : W1[r].d2, W1[r].d2};
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < kResizeRows; ++r) {
                W0[r] = window_load(S + (uint64_t)yt[r].x * spitch, base, last);
                W1[r] = window_load(S + (uint64_t)yt[r].y * spitch, base, last);
            }
        }
#pragma unroll
        for (int r = 0; r < kResizeRows; ++r) {
            // window = source bytes sx0 .. sx0+7 of each tap row (byte funnel of the three aligned dwords)
            const uint32_t lo0 = __builtin_amdgcn_alignbyte(W0[r].d1, W0[r].d0, (uint32_t)sx0), hi0 = __builtin_amdgcn_alignbyte(W0[r].d2, W0[r].d1, (uint32_t)sx0);
            const uint32_t lo1 = __builtin_amdgcn_alignbyte(W1[r].d1, W1[r].d0, (uint32_t)sx0), hi1 = __builtin_amdgcn_alignbyte(W1[r].d2, W1[r].d1, (uint32_t)sx0);
            const uint32_t b0 = (uint32_t)yt[r].z, b1 = (uint32_t)yt[r].w;
            uint32_t v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t r0 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, __builtin_amdgcn_perm(hi0, lo0, sel[i])), __builtin_bit_cast(us2_t, A[i]), 0u, false);
                const uint32_t r1 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, __builtin_amdgcn_perm(hi1, lo1, sel[i])), __builtin_bit_cast(us2_t, A[i]), 0u, false);
                v[i] = ((__umul24(b0, r0 >> 4) >> 16) + (__umul24(b1, r1 >> 4) >> 16) + 2) >> 2;      // <= 255 by construction (taps sum to 2048 +- 1)
            }
            const uint32_t packed = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
            if (dy0 + r < D.h) *reinterpret_cast<uint32_t *>(dst + (uint64_t)(dy0 + r) * D.pitch + dx0) = packed;
        }
    } else {
#pragma unroll
        for (int r = 0; r < kResizeRows; ++r) {
            const uint8_t *S0 = S + (uint64_t)yt[r].x * spitch, *S1 = S + (uint64_t)yt[r].y * spitch;
            uint32_t packed = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sx = xt[i].x, sx1 = min(sx + 1, sw - 1);
                const uint64_t w0 = (uint64_t)S0[sx] | ((uint64_t)S0[sx1] << 8), w1 = (uint64_t)S1[sx] | ((uint64_t)S1[sx1] << 8);
                packed |= (uint32_t)resize_px(w0, w1, 0, xt[i].y, xt[i].z, yt[r].z, yt[r].w) << (8 * i);
            }
            if (dy0 + r < D.h) *reinterpret_cast<uint32_t *>(dst + (uint64_t)(dy0 + r) * D.pitch + dx0) = packed;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// P2: cv::GaussianBlur 7x7 sigma 2, BORDER_REFLECT_101, 8U fixed point (image_pyramid.cpp:84).
// One launch covers every level of every frame (tile table in PyrGeom).  No LDS: a lane owns one dword
// (4 pixels) of a row; a wave owns a 256-pixel row segment and walks 8 output rows.
//   vertical pass   on the raw bytes, two pixels per instruction: even/odd bytes of the dword are two
//                   16-bit lanes (sums stay < 2^16 because the 8.8 taps add up to 256) -> v_pk_mad_u16
//   horizontal pass on the 16-bit column sums of the lane and its two neighbours (DPP wave shifts),
//                   v_dot2_u32_u16 with the rounding constant as the initial accumulator
// Lanes 0 and 63 only provide halo (248 outputs per 256 loaded pixels); REFLECT_101 is applied when the
// bytes are loaded, which commutes with the vertical pass.

__device__ __forceinline__ int reflect101(int i, int n) {
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * n - 2 - i : i;
    return min(max(i, 0), n - 1);
}

__device__ __forceinline__ uint32_t vsum7(const uint32_t *v) {   // 18,34,48,56,48,34,18 on two packed 16-bit lanes
    const us2_t k0 = {18, 18}, k1 = {34, 34}, k2 = {48, 48}, k3 = {56, 56};
    const us2_t a = __builtin_bit_cast(us2_t, v[0]) + __builtin_bit_cast(us2_t, v[6]);
    const us2_t b = __builtin_bit_cast(us2_t, v[1]) + __builtin_bit_cast(us2_t, v[5]);
    const us2_t c = __builtin_bit_cast(us2_t, v[2]) + __builtin_bit_cast(us2_t, v[4]);
    const us2_t d = __builtin_bit_cast(us2_t, v[3]);
    return __builtin_bit_cast(uint32_t, (us2_t)(a * k0 + b * k1 + c * k2 + d * k3));
}

__device__ __forceinline__ uint32_t dot2(uint32_t a, unsigned lo, unsigned hi, uint32_t acc) {
    const us2_t k = {(unsigned short)lo, (unsigned short)hi};
    return __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, a), k, acc, false);
}

// lane i <- lane i-1 / lane i+1 of the 64-wide wave in one VALU op (gfx9 DPP wave shifts); the end lane's value is unspecified
__device__ __forceinline__ uint32_t wave_from_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t wave_from_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /*wave_shl:1*/, 0xf, 0xf, false); }

// inclusive prefix sum over the 64 lanes: four row_shr steps inside each row of 16, then row_bcast 15 / 31 across rows
__device__ __forceinline__ int wave_scan_add(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111 /*row_shr:1*/, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112 /*row_shr:2*/, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114 /*row_shr:4*/, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118 /*row_shr:8*/, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142 /*row_bcast:15*/, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143 /*row_bcast:31*/, 0xc, 0xf, false);
    return v;
}

constexpr int kBlurSeg = 248;   // outputs per wave row segment
constexpr int kBlurRows = 18;   // output rows per wave (+ 6 halo rows loaded).  Measured in one session: 8 -> 0.61 ms, 14 -> 0.53, 16 -> 0.67 (64-row tiles: row starts collide), 18 -> 0.51, 28 -> 0.51
// (8 pixels per lane with 8-byte loads and stores -- half the memory instructions per pixel -- was built and measured: 0.60 ms against 0.51, at 6..10 rows per wave)

__global__ __launch_bounds__(256) void k_blur(FrameSrc src, TileLevels TL, TileMap tm) {
    int t = blockIdx.x;
    const int l = tile_level(tm, TL.levels, t);
    t -= tm.base[l];
    const TileLevel G = TL.L[l];                       // one batch of loads; everything below is arithmetic on it
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform (SGPR): the 14 row addresses are scalar work
    const int trow = G.btiles_x == 1 ? t : (int)__umulhi((uint32_t)t, G.btiles_inv);     // t / btiles_x without the division sequence (2^32 / 1 does not fit)
    const int x = (t - trow * G.btiles_x) * kBlurSeg - 4 + lane * 4;
    const int y0 = trow * (4 * kBlurRows) + wave * kBlurRows;
    const int f = blockIdx.y, w = G.w, h = G.h;
    if (y0 >= h) return;
    const int pitch = l == 0 ? src.lvl0_pitch : G.pitch;
    const uint8_t *img = l == 0 ? src.lvl0 + (uint64_t)f * src.lvl0_frame_stride : src.slab + (uint64_t)f * TL.slab_stride + G.img_off;
    uint32_t e[kBlurRows + 6], o[kBlurRows + 6];
    // the column test is the same for every row of a lane: ONE wave-uniform branch picks the plain-dword path for whole waves
    // (per-load branches cost scalar instructions, and the scalar unit is shared by the CU's four SIMDs)
    if (__ballot(!(x >= 0 && x + 3 < w)) == 0) {
        if (y0 >= 3 && y0 + kBlurRows + 2 < h) {           // no row is reflected: one 64-bit base, the pitch added per row (scalar work is shared by the CU)
            const uint8_t *rp = img + (int64_t)(y0 - 3) * pitch;
#pragma unroll
            for (int r = 0; r < kBlurRows + 6; ++r) {
                const uint32_t d = *reinterpret_cast<const uint32_t *>(rp + x);
                rp += pitch;
                e[r] = d & 0x00FF00FFu;
                o[r] = (d >> 8) & 0x00FF00FFu;
            }
        } else {
#pragma unroll
            for (int r = 0; r < kBlurRows + 6; ++r) {
                const uint32_t d = *reinterpret_cast<const uint32_t *>(img + (uint64_t)reflect101(y0 - 3 + r, h) * pitch + x);
                e[r] = d & 0x00FF00FFu;
                o[r] = (d >> 8) & 0x00FF00FFu;
            }
        }
    } else {
        // A wave on the left or right border (a third of all waves): BORDER_REFLECT_101 without per-byte loads.  Every reflected or
        // straddling pixel of the row lies in its first or last 8 bytes, so the wave fetches those two dwords (one address for all
        // lanes) and a border lane picks its 4 bytes out of them with ONE v_perm whose selector depends only on the lane, not on the row.
        const int xa = min(max(x, 0), w - 4);
        const bool is_l = x < 0, is_r = x > w - 4;
        uint32_t sel_r = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = x + i;
            sel_r |= (uint32_t)((p < w ? p - (w - 8) : (w + 6) - p) & 7) << (8 * i);       // byte of the window [w-8, w-1] that holds pixel reflect(p)
        }
        const bool wave_l = __ballot(is_l) != 0, wave_r = __ballot(is_r) != 0;
#pragma unroll
        for (int r = 0; r < kBlurRows + 6; ++r) {
            const uint8_t *row = img + (uint64_t)reflect101(y0 - 3 + r, h) * pitch;
            uint32_t d = reinterpret_cast<const U32u *>(row + xa)->v;
            if (wave_l) {                                        // pixels -4..-1 = pixels 4, 3, 2, 1
                const uint32_t f0 = reinterpret_cast<const U32u *>(row)->v, f1 = reinterpret_cast<const U32u *>(row + 4)->v;
                d = is_l ? __builtin_amdgcn_perm(f1, f0, 0x01020304u) : d;
            }
            if (wave_r) {
                const uint32_t e0 = reinterpret_cast<const U32u *>(row + (w - 8))->v, e1 = reinterpret_cast<const U32u *>(row + (w - 4))->v;
                d = is_r ? __builtin_amdgcn_perm(e1, e0, sel_r) : d;
            }
            e[r] = d & 0x00FF00FFu;
            o[r] = (d >> 8) & 0x00FF00FFu;
        }
    }
    uint8_t *dst = src.slab + (uint64_t)f * TL.slab_stride + G.blur_off;
#pragma unroll
    for (int j = 0; j < kBlurRows; ++j) {
        const uint32_t Ce = vsum7(e + j), Co = vsum7(o + j);            // columns (x, x+2) and (x+1, x+3)
        // neighbour lanes by DPP wave shifts (VALU, no LDS round trip); lanes 0 and 63 receive a don't-care and store nothing
        const uint32_t Le = wave_from_prev(Ce), Lo = wave_from_prev(Co);
        const uint32_t Re = wave_from_next(Ce), Ro = wave_from_next(Co);
        uint32_t o0 = dot2(Lo, 18, 48, 32768u); o0 = dot2(Le, 0, 34, o0); o0 = dot2(Ce, 56, 34, o0); o0 = dot2(Co, 48, 18, o0);
        uint32_t o1 = dot2(Le, 0, 18, 32768u); o1 = dot2(Lo, 0, 34, o1); o1 = dot2(Ce, 48, 48, o1); o1 = dot2(Co, 56, 34, o1); o1 = dot2(Re, 18, 0, o1);
        uint32_t o2 = dot2(Lo, 0, 18, 32768u); o2 = dot2(Ce, 34, 56, o2); o2 = dot2(Co, 48, 48, o2); o2 = dot2(Re, 34, 0, o2); o2 = dot2(Ro, 18, 0, o2);
        uint32_t o3 = dot2(Ce, 18, 48, 32768u); o3 = dot2(Co, 34, 56, o3); o3 = dot2(Re, 48, 18, o3); o3 = dot2(Ro, 34, 0, o3);
        const int y = y0 + j;
        if (y < h && lane >= 1 && lane <= 62 && x < w)
            *reinterpret_cast<uint32_t *>(dst + (uint64_t)y * G.pitch + x) = (o0 >> 16) | ((o1 >> 16) << 8) | ((o2 >> 16) << 16) | ((o3 >> 16) << 24);
    }
}

// ------------------------------------------------------------------------------------------------
// D1: FAST-9/16 corners + score + 3x3 strict NMS (this build's detector behind
// feature_detector.cpp:89-98).  One launch covers every level of every frame.
// Tile = 248 x 30 outputs; 256 x 32 score positions (1 px NMS halo, rounded to dwords), 4 position rows per wave, 8 waves:
// 34 KB of LDS, so 4 workgroups (32 waves) share a CU -- the kernel is latency-bound (dependent loads,
// five barriers, one returning atomic per tile), not ALU-bound, and needs the occupancy.
//   phase A1  compass pre-test on EVERY position, in registers, two pixels per instruction: a lane owns
//             one dword (4 pixels) of a row, the even/odd bytes are two 16-bit lanes; the sign bits of
//             (centre+t - ring) and (ring - (centre-t)) are formed for the ring pixels N, S, E, W.
//             An arc of 9 contiguous ring pixels always covers N or S and E or W (see compass_pass), so a
//             pixel without a brighter pixel in each opposite pair AND without a darker one in each cannot
//             be a corner (exact reject).  Survivors are compacted into an LDS list.
//   phase A2  dense over the survivors: the FAST score itself (max over the 16 arcs of 9 of min(c-r) / min(r-c)) on packed
//             16-bit lanes (two ring pixels per v_pk_min/max_i16), ring bytes through the vector cache; a pixel is a
//             corner iff score > threshold, so no separate mask test is needed; scores go to the LDS score tile
//   phase C   3x3 strict-maximum NMS, dense over the corner list, against the LDS score tile; survivors leave as 32-bit keys
//             ((255-score)<<24 | y*w+x) with ONE global atomic per tile
constexpr int kFastWaves = 8, kFastThreads = 64 * kFastWaves;     // waves per tile: 4 position rows each
constexpr int kFastSeg = 248, kFastRows = 4 * kFastWaves - 2, kFastPosRows = kFastRows + 2, kFastRowsPerWave = 4;
constexpr int kFastPositions = kFastPosRows * (kFastSeg + 2);   // scored positions of a tile: columns 3..252 of its position rows

__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (us2_t)(__builtin_bit_cast(us2_t, a) - __builtin_bit_cast(us2_t, b)));
}

__device__ __forceinline__ int mbcnt64(unsigned long long m) {      // number of set bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Compass pre-test on packed 16-bit lanes (pixels 0,2 of a lane's dword in "E" registers, pixels 1,3 in "O").
// One v_perm_b32 both shifts a 4-byte window out of a dword pair and zero-extends two of its bytes to 16-bit lanes.
// "At least two of N,S,E,W brighter than c+t" is "the SECOND LARGEST of the four exceeds c+t" (and likewise the second smallest
// for darker): two sorted pairs give both order statistics in 8 packed min/max, then one packed subtract each exposes the sign.
// Returns bit 15 of each 16-bit lane set where the pixel survives (the lower bits are not meaningful).
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s2_t, a), __builtin_bit_cast(s2_t, b))); }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s2_t, a), __builtin_bit_cast(s2_t, b))); }
__device__ __forceinline__ uint32_t compass_pass(uint32_t n, uint32_t s_, uint32_t e, uint32_t w, uint32_t hi, uint32_t lo) {
    // Round 2: a tighter exact reject for two operations less.  The complement of an arc of 9 is an arc of 7, which cannot hold two ring
    // pixels that are 8 apart: every arc of 9 contains N or S, and E or W.  So a brighter corner needs max(N, S) > c + t AND max(E, W) > c + t
    // (a darker one min(N, S) < c - t AND min(E, W) < c - t) -- one pixel from EACH opposite pair, not any two of the four.
    const uint32_t a = pk_max(n, s_), b = pk_min(n, s_), c = pk_max(e, w), d = pk_min(e, w);
    return pk_sub(hi, pk_min(a, c)) | pk_sub(pk_max(b, d), lo);
}

// FAST score = max over the 16 arcs of 9 of min(c - r) and of min(r - c), on packed 16-bit lanes: P[k] = (d[k], d[k+8]) with
// d = centre - ring pixel.  A 9-arc starting at k < 8 is the suffix d[k..7] of the first half plus the prefix d[8..8+k] of the
// second, the one starting at k+8 the mirror image, so running prefix / suffix minima of P (7 packed ops each) and one combine
// per k with the halves swapped (free operand select) give all 16 arc minima in 22 ops; the same for the maxima.
__device__ __forceinline__ int fast_ring_score(const uint32_t R[9]) {
    const uint32_t cc = R[8] * 0x00010001u;
    s2_t P[8], pmn[8], pmx[8], smn[8], smx[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) P[k] = __builtin_bit_cast(s2_t, pk_sub(cc, R[k]));
    pmn[0] = pmx[0] = P[0];
    smn[7] = smx[7] = P[7];
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        pmn[k] = __builtin_elementwise_min(pmn[k - 1], P[k]); pmx[k] = __builtin_elementwise_max(pmx[k - 1], P[k]);
        smn[7 - k] = __builtin_elementwise_min(smn[8 - k], P[7 - k]); smx[7 - k] = __builtin_elementwise_max(smx[8 - k], P[7 - k]);
    }
    s2_t bright = {-32768, -32768}, dark = {32767, 32767};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        bright = __builtin_elementwise_max(bright, __builtin_elementwise_min(smn[k], pmn[k].yx));   // arcs starting at k (.x) and k+8 (.y)
        dark = __builtin_elementwise_min(dark, __builtin_elementwise_max(smx[k], pmx[k].yx));
    }
    return max(max((int)bright.x, (int)bright.y), -min((int)dark.x, (int)dark.y));
}

// One step of a lane's list append: the mask's top bit is shifted out into the carry (v_add_co m, m, m), lanes whose bit was set
// write `entry` as a 16-bit value at LDS byte address `at` and advance it.  Exec is narrowed to the writers for the two
// instructions in between and restored, so unset pixels cost no select and no store.
__device__ __forceinline__ void list_append_top_bit(uint32_t &m, uint32_t &at, uint32_t entry) {
    unsigned long long saved;
    asm volatile("v_add_co_u32 %[m], vcc, %[m], %[m]\n\t"
                 "s_and_saveexec_b64 %[sv], vcc\n\t"
                 "ds_write_b16 %[at], %[e]\n\t"
                 "v_add_u32 %[at], 2, %[at]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [m] "+v"(m), [at] "+v"(at), [sv] "=&s"(saved)
                 : [e] "v"(entry)
                 : "vcc", "memory");
}

__global__ __launch_bounds__(kFastThreads) void k_fast(FrameSrc src, const PyrGeom *g, uint32_t *__restrict__ cand, int32_t *__restrict__ cand_count,
                                              const uint32_t *__restrict__ tile_tab, TileLevels TL) {
    __shared__ uint8_t s_sc[kFastPosRows][256];                                      // score tile (columns 2..253 are touched)
    __shared__ __attribute__((aligned(16))) uint16_t s_pre[kFastPositions + 8];     // compass survivors (+ dump slot); each wave's corners overwrite its own consumed slots
    __shared__ __attribute__((aligned(16))) uint8_t s_pix[(kFastPosRows + 6) * 256]; // the tile's pixels (image rows Y0-4 .. Y0+17) for the ring reads; NMS keys afterwards
    __shared__ int s_np, s_m, s_base;
    uint32_t *s_out = reinterpret_cast<uint32_t *>(s_pix);              // (rows + 6) * 64 keys >= 124 * rows / 2 possible NMS survivors
    // The scalar unit is shared by the CU's four SIMDs and every wave of a tile repeats the tile's scalar work, so that work is kept
    // short: the tile's level, row and column come packed from a host-built table (one scalar load instead of a 15-step search and
    // a division), and the ten row addresses of an interior wave are one 64-bit base plus the pitch (2 scalar adds per row instead
    // of two clamps, a 64-bit multiply and an add).
    const uint32_t te = tile_tab[blockIdx.x];
    const int l = (int)(te & 15u);
    const TileLevel G = TL.L[l];                       // one batch of loads; everything below is arithmetic on it
    const int X0 = (int)((te >> 4) & 0xFFFu) * kFastSeg, Y0 = (int)(te >> 16) * kFastRows;
    const int f = blockIdx.y, w = G.w, h = G.h, thr = TL.fast_threshold;
    const int pitch = l == 0 ? src.lvl0_pitch : G.pitch;
    const uint8_t *img = l == 0 ? src.lvl0 + (uint64_t)f * src.lvl0_frame_stride : src.slab + (uint64_t)f * TL.slab_stride + G.img_off;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform, in an SGPR: row addresses stay scalar
    // The wave's 10 image rows are requested FIRST, so their latency runs under the LDS clear and the barrier.
    const int x = X0 - 4 + 4 * lane;
    uint32_t rows[kFastRowsPerWave + 6];
    const int yw = Y0 - 1 + wave * kFastRowsPerWave;
    if (__ballot(!(x >= 0 && x + 3 < w)) == 0 && yw >= 3 && yw + kFastRowsPerWave + 2 < h) {     // whole wave inside the image: plain dword loads, one uniform branch
        const uint8_t *rp = img + (int64_t)(yw - 3) * pitch;
#pragma unroll
        for (int r = 0; r < kFastRowsPerWave + 6; ++r) { rows[r] = *reinterpret_cast<const uint32_t *>(rp + x); rp += pitch; }
    } else {
        // a wave on the image border: still ONE dword load per row and lane (a third of all waves are such waves; per-byte loads with
        // their branches made them cost three times an interior wave).  The address is clamped into the row, a lane that straddles the
        // right edge shifts its pixels down, pixels outside the image are zero.
        const int xa = min(max(x, 0), w - 4);
        const uint32_t sh = (uint32_t)(8 * (x - xa)) & 31u, keep = (x >= 0 && x < w) ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int r = 0; r < kFastRowsPerWave + 6; ++r)
            rows[r] = (reinterpret_cast<const U32u *>(img + (uint64_t)min(max(yw - 3 + r, 0), h - 1) * pitch + xa)->v >> sh) & keep;
    }
    if (tid == 0) { s_np = 0; s_m = 0; }
    for (int i = tid; i < kFastPosRows * 256 / 4; i += kFastThreads) reinterpret_cast<uint32_t *>(&s_sc[0][0])[i] = 0;
    // the rows go to LDS as well: wave w owns tile rows 4w .. 4w+3 (image rows Y0-4+4w ..), the last wave also the six below
#pragma unroll
    for (int r = 0; r < kFastRowsPerWave + 6; ++r)
        if (r < kFastRowsPerWave || wave == kFastWaves - 1) reinterpret_cast<uint32_t *>(s_pix)[(wave * kFastRowsPerWave + r) * 64 + lane] = rows[r];
    __syncthreads();
    // ---- phase A1: position rows pr = wave*4 .. wave*4+3  <->  image rows Y0-1+pr; columns X0-4+4*lane .. +3
    {
        const uint32_t T2 = (uint32_t)thr * 0x00010001u;
        uint32_t vmask = 0;                              // which of the lane's 4 pixels are valid positions (same for every row): bit 8i + 7
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = 4 * lane + i, px = x + i;
            vmask |= (uint32_t)(c >= 3 && c <= 252 && px >= 3 && px < w - 3) << (8 * i + 7);
        }
        uint32_t fr[kFastRowsPerWave];                   // survivors of row r: the top bit of byte i <-> pixel i
        uint32_t cnt4 = 0;                               // their number, row r in byte r (a row has at most 250 positions, so sums over lanes stay in the byte)
#pragma unroll
        for (int r = 0; r < kFastRowsPerWave; ++r) {
            const int y = Y0 - 1 + wave * kFastRowsPerWave + r;
            fr[r] = 0;
            if (y < 3 || y >= h - 3) continue;                           // wave-uniform
            const uint32_t C = rows[r + 3], Cn = rows[r + 6], Cs = rows[r];
            // neighbours' dwords by DPP wave shifts (one VALU op each); lanes 0 / 63 get a don't-care: their edge pixels are masked by vmask
            const uint32_t Lw = wave_from_prev(C), Rw = wave_from_next(C);
            const uint32_t ce = C & 0x00FF00FFu, co = (C >> 8) & 0x00FF00FFu;
            const uint32_t hie = ce + T2, hio = co + T2, loe = pk_sub(ce, T2), loo = pk_sub(co, T2);
            // byte selectors: 0-3 = bytes of the low dword, 4-7 = bytes of the high dword, 0x0C = zero
            const uint32_t pe = compass_pass(__builtin_amdgcn_perm(0u, Cn, 0x0C020C00u), __builtin_amdgcn_perm(0u, Cs, 0x0C020C00u),     // (0,+3), (0,-3): bytes 0,2
                                             __builtin_amdgcn_perm(Rw, C, 0x0C050C03u), __builtin_amdgcn_perm(C, Lw, 0x0C030C01u),      // (+3,0): x+3, x+5 of {Rw,C}; (-3,0): x-3, x-1 of {C,Lw}
                                             hie, loe);
            const uint32_t po = compass_pass(__builtin_amdgcn_perm(0u, Cn, 0x0C030C01u), __builtin_amdgcn_perm(0u, Cs, 0x0C030C01u),     // bytes 1,3
                                             __builtin_amdgcn_perm(Rw, C, 0x0C060C04u), __builtin_amdgcn_perm(C, Lw, 0x0C040C02u), hio, loo);
            fr[r] = __builtin_amdgcn_perm(po, pe, 0x07030501u) & vmask;  // the sign bytes of the four pixels side by side (pixel i in byte i)
            cnt4 += (uint32_t)__popc(fr[r]) << (8 * r);
        }
        // ONE list append per wave, in the order row by row, lane by lane (neighbouring list slots = neighbouring pixels of a row, which
        // keeps the ring reads of phase A2 off each other's LDS banks): a single DPP scan of the packed per-row counts gives every lane
        // its offset inside each row, the row totals come out of lane 63, one lane adds their sum to the tile's counter.
        const uint32_t incl = (uint32_t)wave_scan_add((int)cnt4);
        const uint32_t tot4 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63), excl = incl - cnt4;
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_np, (int)((tot4 & 255u) + ((tot4 >> 8) & 255u) + ((tot4 >> 16) & 255u) + (tot4 >> 24)));
        base = __builtin_amdgcn_readfirstlane(base);
        const uint32_t ebase = (uint32_t)((wave * kFastRowsPerWave) << 8) | (uint32_t)(4 * lane);
        const uint32_t pre_addr = (uint32_t)(uintptr_t)s_pre;
#pragma unroll
        for (int r = 0; r < kFastRowsPerWave; ++r) {
            uint32_t at = pre_addr + 2u * ((uint32_t)base + ((excl >> (8 * r)) & 255u));      // byte address of the lane's first slot of this row
            base += (int)((tot4 >> (8 * r)) & 255u);
            uint32_t f = fr[r];
            list_append_top_bit(f, at, ebase + (r << 8) + 3); f <<= 7;
            list_append_top_bit(f, at, ebase + (r << 8) + 2); f <<= 7;
            list_append_top_bit(f, at, ebase + (r << 8) + 1); f <<= 7;
            list_append_top_bit(f, at, ebase + (r << 8) + 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // ---- phase A2 + B: the survivors are SCORED directly (corner <=> score > threshold), on packed 16-bit lanes:
    //      P[k] = (d[k], d[k+8]) with d = centre - ring pixel; the sliding min / max of 9 over the circular ring is four
    //      v_pk_min_i16 / v_pk_max_i16 levels (windows 2, 4, 8, 9) with lane swaps providing the wrap-around.
    //      The ring comes from the tile's pixels in LDS (17 byte reads with immediate offsets from one base): as 7 wide global loads per
    //      survivor the texture addresser had ~30 cache lines to look up per instruction.  A wave appends its corners IN PLACE, into the
    //      slots of the survivor list it has already consumed (slots 64w + kFastThreads i + j belong to wave w), so no second list is needed.
    const int np = s_np;
    int ncw = 0;                                                         // corners of this wave so far (wave-uniform)
    for (int i = tid; i < np; i += kFastThreads) {
        const int e = s_pre[i], pr = e >> 8, c = e & 255;
        const uint8_t *q = s_pix + pr * 256 + (c - 3);                   // top-left of the 7x7 box: tile row pr + 3 is the centre's
        uint32_t R[9];
        {
            const int rdx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
            const int rdy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
            R[8] = q[3 * 256 + 3];
#pragma unroll
            for (int k = 0; k < 8; ++k) R[k] = (uint32_t)q[(3 + rdy[k]) * 256 + 3 + rdx[k]] | ((uint32_t)q[(3 + rdy[k + 8]) * 256 + 3 + rdx[k + 8]] << 16);
        }
        const int best = fast_ring_score(R);
        const bool corner = best > thr;
        const unsigned long long cm = __ballot(corner);
        if (corner) {
            const int slot = ncw + mbcnt64(cm);
            s_sc[pr][c] = (uint8_t)best;
            s_pre[64 * wave + kFastThreads * (slot >> 6) + (slot & 63)] = (uint16_t)e;
        }
        ncw += (int)__popcll(cm);
    }
    ncw = __builtin_amdgcn_readfirstlane(ncw);          // lanes that left the loop early missed the last updates; lane 0 stays to the end
    __syncthreads();
    // ---- phase C: 3x3 strict-maximum NMS, dense over the corner list (only outputs: px X0..X0+247 = columns 4..251,
    //      rows Y0..Y0+13 = position rows 1..14; the halo corners only serve as neighbours)
    for (int i = lane; i < ncw; i += 64) {                               // every wave walks its own corner sublist
        const int e = s_pre[64 * wave + kFastThreads * (i >> 6) + (i & 63)], pr = e >> 8, c = e & 255;
        const int px = X0 - 4 + c, y = Y0 - 1 + pr;
        if (pr < 1 || pr > kFastRows || c < 4 || c > 251 || px >= w || y >= h) continue;
        const int sc = s_sc[pr][c];
        const bool keep = sc > s_sc[pr - 1][c - 1] && sc > s_sc[pr - 1][c] && sc > s_sc[pr - 1][c + 1] && sc > s_sc[pr][c - 1] &&
                          sc > s_sc[pr][c + 1] && sc > s_sc[pr + 1][c - 1] && sc > s_sc[pr + 1][c] && sc > s_sc[pr + 1][c + 1];
        if (keep) s_out[atomicAdd(&s_m, 1)] = ((uint32_t)(255 - sc) << 24) | (uint32_t)(y * w + px);
    }
    __syncthreads();
    const int m = s_m;
    if (m == 0) return;
    if (tid == 0) s_base = atomicAdd(&cand_count[f * TL.levels + l], m);
    __syncthreads();
    for (int i = tid; i < m; i += kFastThreads) {
        const int pos = s_base + i;
        if (pos < G.cand_cap) cand[(uint64_t)f * TL.cand_stride + G.cand_off + pos] = s_out[i];
    }
}

// ------------------------------------------------------------------------------------------------
// D1 selection: per (frame, level) keep the `quota` smallest keys (= strongest corners, ties by raster
// index), sort them, then apply the 19 px border filter of feature_detector.cpp:106-123 and the camera
// validity mask (dropInvalidKeypoints, orb_extractor.cpp:221-237) with an ORDERED compaction.
// Radix select (4 x 8 bits, LDS histogram) -> LDS bitonic sort of <= 4096 keys.  Deterministic:
// the unordered candidate list only feeds order-insensitive steps.
// With a minimum distance (gfttMinDistance scaled per level, feature_detector.cpp:79-82) the best
// min(4*quota, 4096) corners are sorted and the sequential "keep a corner unless a kept one is closer than
// min_dist" walk is solved as a parallel fixed point: corners are hashed into an LDS grid (cell >= min_dist), a
// corner is rejected as soon as an earlier kept neighbour exists and kept once all its earlier neighbours are
// decided -- the unique fixed point is exactly the greedy result; the first `quota` kept corners are output.
constexpr int kGridCells = 8192;

// NTH: threads per block.  256 for a batch (thousands of blocks); 1024 when only a few frames are in flight -- a level's candidates are then walked by ONE block
// five times (four radix passes and the gather), and a single 720p frame's level 0 has tens of thousands of them: 54 us of a frame's 143 us with 256 threads.
template <bool SPACED, int NTH>   // SPACED: some level has a minimum keypoint distance (uses 66 KB more LDS for the grid)
__global__ __launch_bounds__(NTH) void k_select(const PyrGeom *g, const uint32_t *__restrict__ cand, int32_t *__restrict__ cand_count,
                                                const uint8_t *__restrict__ valid_mask,
                                                int16_t *__restrict__ det_x, int16_t *__restrict__ det_y, uint8_t *__restrict__ det_score,
                                                int32_t *__restrict__ det_count) {
    __shared__ uint32_t s_key[kMaxQuota];
    __shared__ int s_hist[256];
    __shared__ int s_cnt, s_digit, s_k, s_run, s_kept, s_open;
    __shared__ int s_w1[4], s_w2[4];
    __shared__ uint8_t s_state[SPACED ? kMaxQuota : 1];          // 0 undecided, 1 kept, 2 rejected
    const int l = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const LevelGeom &G = g->L[l];
    const uint32_t *keys = cand + (uint64_t)f * g->cand_stride + G.cand_off;
    const int n = min(cand_count[f * g->levels + l], G.cand_cap), n4 = n >> 2;
    const uint4 *keys4 = reinterpret_cast<const uint4 *>(keys);          // (every level's list starts 16-byte aligned: cand_cap is a multiple of 4)
    const int quota = G.quota, min_dist = G.min_dist;
    const bool spaced = SPACED && min_dist >= 2;
    const int want = spaced ? min(4 * quota, kMaxQuota) : quota;      // how many of the strongest corners are sorted
    uint32_t kth = 0xFFFFFFFFu;
    if (n > want && want > 0) {
        uint32_t prefix = 0, mask = 0;
        if (tid == 0) s_k = want;
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) s_hist[tid] = 0;
            __syncthreads();
            // one block walks a level's list, and a block's memory parallelism is what its waves keep in flight: 16-byte loads, two per thread (a
            // dword per lane and trip moved ~20 GB/s: 14 us per pass over a 720p frame's level 0)
            for (int i = tid; i < n4; i += 2 * NTH) {
                const uint4 q0 = keys4[i], q1 = i + NTH < n4 ? keys4[i + NTH] : uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                const uint32_t kk[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                const int lim = i + NTH < n4 ? 8 : 4;
#pragma unroll
                for (int u = 0; u < 8; ++u) if (u < lim && (kk[u] & mask) == prefix) atomicAdd(&s_hist[(kk[u] >> shift) & 255], 1);
            }
            for (int i = 4 * n4 + tid; i < n; i += NTH) { const uint32_t k = keys[i]; if ((k & mask) == prefix) atomicAdd(&s_hist[(k >> shift) & 255], 1); }
            __syncthreads();
            // the digit of the k-th smallest key: the first bin whose running count reaches k.  (One thread walking the 256 bins was ~10 us per pass --
            // a chain of dependent LDS reads -- and four passes of it were most of a single frame's 44 us in this kernel.)
            {
                const int kk = s_k;
                int h = tid < 256 ? s_hist[tid] : 0, c = h;                  // inclusive running count over the bins: wave scan, then the waves' totals
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(c, off, 64); if ((tid & 63) >= off) c += v; }
                if ((tid & 63) == 63 && tid < 256) s_w1[tid >> 6] = c;
                __syncthreads();
                if (tid < 256) {
                    for (int w = 0; w < (tid >> 6); ++w) c += s_w1[w];
                    const bool last = tid == 255;                            // (k never exceeds the total: the last bin takes what is left, like the walk did)
                    if ((c >= kk || last) && c - h < kk) { s_digit = tid; s_k = kk - (c - h); }
                }
            }
            __syncthreads();
            prefix |= (uint32_t)s_digit << shift;
            mask |= 255u << shift;
            __syncthreads();
        }
        kth = prefix;
    }
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    if (want > 0)
    {
        for (int i = tid; i < n4; i += 2 * NTH) {
            const uint4 q0 = keys4[i], q1 = i + NTH < n4 ? keys4[i + NTH] : uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            const uint32_t kk[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            const int lim = i + NTH < n4 ? 8 : 4;
#pragma unroll
            for (int u = 0; u < 8; ++u) if (u < lim && kk[u] <= kth) { const int p = atomicAdd(&s_cnt, 1); if (p < kMaxQuota) s_key[p] = kk[u]; }
        }
        for (int i = 4 * n4 + tid; i < n; i += NTH) { const uint32_t k = keys[i]; if (k <= kth) { const int p = atomicAdd(&s_cnt, 1); if (p < kMaxQuota) s_key[p] = k; } }
    }
    __syncthreads();
    const int m = min(s_cnt, kMaxQuota);
    int np2 = 1;
    while (np2 < m) np2 <<= 1;
    for (int i = m + tid; i < np2; i += NTH) s_key[i] = 0xFFFFFFFFu;
    __syncthreads();
    // bitonic sort.  A compare-exchange distance below 64 keeps both partners inside one wave's 64 consecutive elements (of every NTH-strided round), so
    // those steps only need the wave's own LDS order; the block meets after the steps at distance >= 64 and before the next stage starts with one
    // (45 steps for 512 keys, 6 block barriers instead of 45: a single frame's level is ONE block, and its barriers were most of its time)
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += NTH) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint32_t a = s_key[i], b = s_key[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { s_key[i] = b; s_key[ixj] = a; }
                }
            }
            if (j >= 64 || (j == 1 && k >= 64)) __syncthreads();
            else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        }
    __syncthreads();
    // minimum-distance walk as a fixed point
    if constexpr (SPACED) if (spaced) {
        __shared__ int16_t s_x[kMaxQuota], s_y[kMaxQuota], s_next[kMaxQuota];
        __shared__ int s_head[kGridCells];
        int cell = min_dist;
        while (((G.w + cell - 1) / cell) * ((G.h + cell - 1) / cell) > kGridCells) ++cell;
        const int gw = (G.w + cell - 1) / cell, gh = (G.h + cell - 1) / cell;
        for (int i = tid; i < gw * gh; i += NTH) s_head[i] = -1;
        __syncthreads();
        for (int i = tid; i < m; i += NTH) {
            const int idx = (int)(s_key[i] & 0xFFFFFFu), y = idx / G.w, x = idx - y * G.w;
            s_x[i] = (int16_t)x; s_y[i] = (int16_t)y; s_state[i] = 0;
        }
        __syncthreads();
        for (int i = tid; i < m; i += NTH)                      // cell lists (their internal order does not matter)
            s_next[i] = (int16_t)atomicExch(&s_head[(s_y[i] / cell) * gw + s_x[i] / cell], i);
        __syncthreads();
        const int d2 = min_dist * min_dist;
        for (int round = 0; round < kMaxQuota; ++round) {
            if (tid == 0) s_open = 0;
            __syncthreads();
            for (int i = tid; i < m; i += NTH) {
                if (s_state[i]) continue;
                const int x = s_x[i], y = s_y[i], cx = x / cell, cy = y / cell;
                bool rejected = false, waiting = false;
                for (int yy = max(cy - 1, 0); yy <= min(cy + 1, gh - 1) && !rejected; ++yy)
                    for (int xx = max(cx - 1, 0); xx <= min(cx + 1, gw - 1) && !rejected; ++xx)
                        for (int j = s_head[yy * gw + xx]; j >= 0; j = s_next[j]) {
                            if (j >= i) continue;                       // only corners earlier in key order matter
                            const int dx = x - s_x[j], dy = y - s_y[j];
                            if (dx * dx + dy * dy >= d2) continue;
                            const int st = s_state[j];
                            if (st == 1) { rejected = true; break; }
                            if (st == 0) waiting = true;
                        }
                if (rejected) s_state[i] = 2;
                else if (!waiting) s_state[i] = 1;
                else s_open = 1;
            }
            __syncthreads();
            const int open = s_open;
            __syncthreads();
            if (!open) break;
        }
    }
    // ordered compaction: first `quota` kept corners, then the border / validity filter
    if (tid == 0) { s_run = 0; s_kept = 0; }
    __syncthreads();
    const int W0 = g->width, H0 = g->height;
    // (the two prefix sums per chunk are over 0 / 1 flags: a ballot and two popcounts per wave, one barrier each for the waves' totals)
    const int wv = tid >> 6, ln = tid & 63;
    for (int base = 0; base < m; base += 256) {                  // chunks of 256 (threads beyond 256 only keep the barriers company)
        const int i = base + tid;
        int x = 0, y = 0, sc = 0, keep = 0;
        if (i < m && tid < 256) {
            const uint32_t k = s_key[i];
            const int idx = (int)(k & 0xFFFFFFu);
            y = idx / G.w; x = idx - y * G.w; sc = 255 - (int)(k >> 24);
            keep = spaced ? (s_state[i] == 1) : 1;
        }
        const unsigned long long bk = __ballot(keep != 0);
        if (ln == 0 && wv < 4) s_w1[wv] = __popcll(bk);
        __syncthreads();
        int incl = __popcll(bk & ((2ull << ln) - 1ull)), kept_chunk = 0;
        for (int w = 0; w < 4; ++w) { if (w < wv) incl += s_w1[w]; kept_chunk += s_w1[w]; }
        const int kept_before = s_kept, run = s_run;
        int ok = tid < 256 && keep && (kept_before + incl - 1 < quota);        // maxTracks = quota_l
        if (ok) {
            ok = x >= kPatchRadius && y >= kPatchRadius && x < G.w - kPatchRadius && y < G.h - kPatchRadius;
            if (ok && valid_mask) {
                const int mx = __float2int_rn(__fmul_rn((float)x, G.scale)), my = __float2int_rn(__fmul_rn((float)y, G.scale));
                ok = mx >= 0 && my >= 0 && mx < W0 && my < H0 && valid_mask[(uint64_t)my * W0 + mx] != 0;
            }
        }
        const unsigned long long bo = __ballot(ok != 0);
        if (ln == 0 && wv < 4) s_w2[wv] = __popcll(bo);
        __syncthreads();
        int incl2 = __popcll(bo & ((2ull << ln) - 1ull)), out_chunk = 0;
        for (int w = 0; w < 4; ++w) { if (w < wv) incl2 += s_w2[w]; out_chunk += s_w2[w]; }
        if (ok) {
            const uint64_t slot = (uint64_t)f * g->det_stride + G.det_base + run + incl2 - 1;
            det_x[slot] = (int16_t)x; det_y[slot] = (int16_t)y; det_score[slot] = (uint8_t)sc;
        }
        __syncthreads();                                         // everyone has read s_run / s_kept / the waves' totals
        if (tid == 0) { s_run = run + out_chunk; s_kept = kept_before + kept_chunk; }
        __syncthreads();
    }
    if (tid == 0) { det_count[f * g->levels + l] = s_run; cand_count[f * g->levels + l] = 0; }      // the list is consumed: k_fast of the next extract appends from zero (no memset launch per call)
}

// ------------------------------------------------------------------------------------------------
// Tracker features (orb_extractor.cpp:89-124): level lk_level, x = cvRound(pt.x/scale), margin 19,
// camera validity at (pt.x, pt.y); survivors keep their input order.  One wave per frame.
__global__ __launch_bounds__(64) void k_tracks(const PyrGeom *g, const float *__restrict__ track_xy, const int32_t *__restrict__ track_id,
                                               const int32_t *__restrict__ n_tracks, const uint8_t *__restrict__ valid_mask,
                                               int16_t *__restrict__ trk_x, int16_t *__restrict__ trk_y, float *__restrict__ trk_px,
                                               float *__restrict__ trk_py, int32_t *__restrict__ trk_id, int32_t *__restrict__ trk_count) {
    const int f = blockIdx.x, lane = threadIdx.x;
    const int n = n_tracks ? min(n_tracks[f], g->max_tracks) : 0;
    const LevelGeom &G = g->L[g->lk_level];
    int run = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        bool ok = false; int x = 0, y = 0; float px = 0.f, py = 0.f;
        if (i < n) {
            px = track_xy[((uint64_t)f * g->max_tracks + i) * 2];
            py = track_xy[((uint64_t)f * g->max_tracks + i) * 2 + 1];
            x = __float2int_rn(__fdiv_rn(px, G.scale));
            y = __float2int_rn(__fdiv_rn(py, G.scale));
            ok = x >= kPatchRadius && y >= kPatchRadius && x < G.w - kPatchRadius && y < G.h - kPatchRadius;
            if (ok && valid_mask) {
                const int mx = __float2int_rn(px), my = __float2int_rn(py);
                ok = mx >= 0 && my >= 0 && mx < g->width && my < g->height && valid_mask[(uint64_t)my * g->width + mx] != 0;
            }
        }
        const unsigned long long bal = __ballot(ok);
        if (ok) {
            const uint64_t slot = (uint64_t)f * g->max_tracks + run + __popcll(bal & ((1ull << lane) - 1ull));
            trk_x[slot] = (int16_t)x; trk_y[slot] = (int16_t)y; trk_px[slot] = px; trk_py[slot] = py;
            trk_id[slot] = track_id ? track_id[(uint64_t)f * g->max_tracks + i] : i;
        }
        run += __popcll(bal);
    }
    if (lane == 0) trk_count[f] = run;
}

// ------------------------------------------------------------------------------------------------
// O1 + O2: one wavefront per keypoint.
//   ic_angle (orb_extractor.cpp:245-275): integer moments over the radius-15 disc of the UNBLURRED
//   level, 961 candidate offsets strided over the 64 lanes, DPP/shuffle reduction, cv::fastAtan2.
//   descriptor (orb_extractor.cpp:284-352): lane j evaluates BRIEF tests j, j+64, j+128, j+192 on the
//   BLURRED level; four 64-bit ballots are the 256 descriptor bits (test t -> word t/32, bit t%32).

__device__ __forceinline__ float dev_fast_atan2(float y, float x) {   // cv::fastAtan2 (atan_f32), degrees
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, (float)DBL_EPSILON));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, (float)DBL_EPSILON));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

__device__ __forceinline__ float dev_poly_cos(float v) {   // openvslam/trigonometric.h:17-24
    const float c1 = 0.99940307f, c2 = -0.49558072f, c3 = 0.03679168f;
    const float v2 = __fmul_rn(v, v);
    return __fadd_rn(c1, __fmul_rn(v2, __fadd_rn(c2, __fmul_rn(c3, v2))));
}
__device__ __forceinline__ float dev_cos(float v) {        // openvslam/trigonometric.h:26-42
    const float PI = 3.14159265358979f, PI_2 = PI / 2.0f, TWO_PI = 2.0f * PI, INV_TWO_PI = 1.0f / TWO_PI, THREE_PI_2 = 3.0f * PI_2;
    v = __fsub_rn(v, __fmul_rn((float)(int)floorf(__fmul_rn(v, INV_TWO_PI)), TWO_PI));
    v = (0.0f < v) ? v : -v;
    if (v < PI_2) return dev_poly_cos(v);
    else if (v < PI) return -dev_poly_cos(__fsub_rn(PI, v));
    else if (v < THREE_PI_2) return -dev_poly_cos(__fsub_rn(v, PI));
    else return dev_poly_cos(__fsub_rn(TWO_PI, v));
}
__device__ __forceinline__ float dev_sin(float v) { return dev_cos(__fsub_rn(3.14159265358979f / 2.0f, v)); }

__device__ __forceinline__ int wave_sum(int v) {      // DPP inclusive scan, total broadcast from lane 63
    return __builtin_amdgcn_readlane(wave_scan_add(v), 63);
}

// Output slot -> detection, two dwords per slot (x | y << 16, level: coordinates up to the 32767 ms_orb_create admits), and the frame's keypoint total: k_describe's waves found their keypoint through a chain
// of dependent loads (the counts of all levels -> a 16-step search for the level -> the level's base in the geometry table -> the coordinates), ~150 scalar
// instructions and three round trips in front of the window fetch of EVERY wave.  One block per (level, frame) writes the level's run of the table once.
__global__ __launch_bounds__(256) void k_slots(const int16_t *__restrict__ det_x, const int16_t *__restrict__ det_y, const int32_t *__restrict__ det_count,
                                               const int32_t *__restrict__ trk_count, uint2 *__restrict__ slot_tab, int32_t *__restrict__ out_count,
                                               TileLevels TL, int det_stride, int capacity) {
    const int l = blockIdx.x, f = blockIdx.y;
    int base = trk_count[f];
    for (int k = 0; k < l; ++k) base += det_count[f * TL.levels + k];
    const int cnt = det_count[f * TL.levels + l];
    const uint64_t s0 = (uint64_t)f * det_stride + TL.L[l].det_base;
    for (int i = threadIdx.x; i < cnt; i += 256)
        if (base + i < capacity) slot_tab[(uint64_t)f * capacity + base + i] = uint2{(uint32_t)(uint16_t)det_x[s0 + i] | ((uint32_t)(uint16_t)det_y[s0 + i] << 16), (uint32_t)l};
    if (l == TL.levels - 1 && threadIdx.x == 0) out_count[f] = min(base + cnt, capacity);
}

// Keypoints a wave works on at once.  k_describe is bound by the memory system's throughput of scattered partial-line fetches: 2.0 GB
// of HBM traffic per 256-frame launch for 0.87 GB of patch bytes, in 0.53 ms = 3.8 TB/s.  Measured without effect on its time: 6, 7
// or 8 waves per SIMD; two keypoints side by side in a wave (0.64 ms, register pressure) or four one after another with the next
// one's patches prefetched (0.63 ms); 16-byte row pieces (3 loads per keypoint instead of 11); all outputs in one scattered store; a
// per-slot table that removes the slot -> counts -> coordinates -> geometry chain of dependent loads; and a quarter fewer VALU
// instructions (the moment sums below and the degree -> radian product).
constexpr int kDescPerWave = 1;

#ifndef MS_DESC_WAVES
#define MS_DESC_WAVES 8       // waves per SIMD the register allocation of k_describe aims at (the kernel is bound by how many keypoints are in flight)
#endif
struct DescKp { int x, y, oct, tid_out; float ox, oy; bool valid; };
typedef int v4i_t __attribute__((ext_vector_type(4)));
// r[m] in lane row q (the wave's four rows of 16 lanes)  ->  r[q] of lane row m: the 2 x 2 blocks trade places across the wave's halves
// (v_permlane32_swap: lanes 32-63 of the first operand <-> lanes 0-31 of the second), then each block is transposed across neighbouring rows
// (v_permlane16_swap: odd rows of the first <-> even rows of the second); lane maps checked on the device by tools/mfma_i8_probe.hip
__device__ __forceinline__ void transpose4_rows(uint32_t (&r)[4]) {
    auto s = __builtin_amdgcn_permlane32_swap(r[0], r[2], false, false); r[0] = s[0]; r[2] = s[1];
    s = __builtin_amdgcn_permlane32_swap(r[1], r[3], false, false); r[1] = s[0]; r[3] = s[1];
    s = __builtin_amdgcn_permlane16_swap(r[0], r[1], false, false); r[0] = s[0]; r[1] = s[1];
    s = __builtin_amdgcn_permlane16_swap(r[2], r[3], false, false); r[2] = s[0]; r[3] = s[1];
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MS_DESC_WAVES, 8))) void k_describe(FrameSrc src, TileLevels TL, const uint4 *__restrict__ moment_tab, const float4 *__restrict__ pattern_f,
                                                  const uint2 *__restrict__ slot_tab, int capacity, int max_tracks, int lk_level,
                                                  const int16_t *__restrict__ trk_x, const int16_t *__restrict__ trk_y, const float *__restrict__ trk_px,
                                                  const float *__restrict__ trk_py, const int32_t *__restrict__ trk_id, const int32_t *__restrict__ trk_count,
                                                  float *__restrict__ out_x, float *__restrict__ out_y, float *__restrict__ out_angle,
                                                  int32_t *__restrict__ out_octave, uint32_t *__restrict__ out_desc, int32_t *__restrict__ out_track,
                                                  const int32_t *__restrict__ out_count) {
    const int f = blockIdx.y, lane = threadIdx.x & 63;
    const int slot0 = (blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * kDescPerWave;   // wave-uniform: everything derived from it is scalar
    // A keypoint is a chain of dependent memory round trips (its slot -> coordinates -> patch -> angle -> BRIEF samples) and the
    // everything that does not depend on the keypoint is requested up front -- the lane's patch-offset table and the counts of
    // ALL levels in one batch (not one load per loop trip).
    // What does not depend on the keypoint and is needed late -- the blur's tap operands (right before the blur) and the 256 BRIEF point pairs (after the
    // orientation) -- was fetched from global memory where it was needed, behind the LDS fences no load may cross: two exposed round trips in the middle of every
    // wave's chain, and this kernel is bound by the LENGTH of that chain (ablation: prologue + window fetch alone 0.35 of 0.60 ms, no prologue chain -0.07 ms,
    // a quarter fewer vector instructions 0.00 ms).  They are block-wide tables in LDS now, loaded first thing, under the count loads.
    __shared__ uint4 s_tab[64 + 256];                        // [0..31] horizontal taps (dt, n), [32..63] vertical taps, [64..319] the point pairs [q][lane]
    if (threadIdx.x < 64) s_tab[threadIdx.x] = moment_tab[64 + threadIdx.x];
    s_tab[64 + threadIdx.x] = reinterpret_cast<const uint4 *>(pattern_f)[threadIdx.x];
    const uint4 dm = moment_tab[lane];                       // disc mask of the lane's four patch dwords
    const uint32_t dmask[4] = {dm.x, dm.y, dm.z, dm.w};
    // the frame's keypoint total and this wave's table entry (k_slots): both addresses are known from the block index, one round trip
    const int total = out_count[f], nt = trk_count[f];
    const uint2 ent = slot_tab[(uint64_t)f * capacity + min(slot0, capacity - 1)];
    __syncthreads();                                         // the tables are in LDS (every wave of the block is still here)
    if (slot0 >= total) return;
    DescKp K[kDescPerWave];
#pragma unroll
    for (int k = 0; k < kDescPerWave; ++k) {
        const int slot = slot0 + k;
        K[k].valid = slot < total;
        const int sl = K[k].valid ? slot : slot0;            // an absent second keypoint mirrors the first (its results are not stored)
        if (sl < nt) {
            const uint64_t s2 = (uint64_t)f * max_tracks + sl;
            K[k].x = trk_x[s2]; K[k].y = trk_y[s2]; K[k].ox = trk_px[s2]; K[k].oy = trk_py[s2]; K[k].oct = lk_level; K[k].tid_out = trk_id[s2];
        } else {
            const uint2 e2 = k == 0 ? ent : slot_tab[(uint64_t)f * capacity + sl];
            const int level = (int)e2.y;
            K[k].x = (int)(e2.x & 0xFFFFu); K[k].y = (int)(e2.x >> 16); K[k].oct = level; K[k].tid_out = -1;
            K[k].ox = __fmul_rn((float)K[k].x, TL.L[level].scale);     // orb_extractor.cpp:156
            K[k].oy = __fmul_rn((float)K[k].y, TL.L[level].scale);
        }
    }
    // ONE window of the (unblurred) level is copied into the wave's LDS slab with coalesced row loads: 45 rows x 48 B around the keypoint
    // (x-23 .. x+24, y-22 .. y+22; 9 wave-wide dword loads).  It holds the 31 x 31 orientation patch (copied out once more with everything
    // outside the radius-15 disc zeroed) AND everything the 39 x 39 BLURRED patch of the BRIEF tests depends on (19 + 3 pixels each way),
    // so the blur (image_pyramid.cpp:84: 7 x 7, sigma 2, REFLECT_101) is computed here, for the patch only, with k_blur's own arithmetic --
    // vertical pass on packed 16-bit lanes, horizontal pass with v_dot2 and one rounding -- and the blurred pyramid is no longer
    // written and read back for every frame (2 P bytes and a 0.49 ms kernel per 256-frame step; round 1 fetched 31 x 32 B of the level plus
    // 39 x 40 B of its blurred twin = 70 row pieces per keypoint, now 45).  k_blur still exists: ImagePyramid::getBlurredLevel runs it on demand.
    __shared__ __attribute__((aligned(16))) uint32_t s_patch[4][kDescPerWave][45 * 12 + 32 * 8 + 4];   // window + orientation patch (a whole number of 16-byte units: the window rows are read as ds_read_b128); the blurred patch takes the window's place
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kDescPerWave; ++k) {
        const TileLevel GL = TL.L[K[k].oct];               // (kernel arguments: no load from the geometry table behind the level)
        const int pitch = K[k].oct == 0 ? src.lvl0_pitch : GL.pitch;
        const uint8_t *img = K[k].oct == 0 ? src.lvl0 + (uint64_t)f * src.lvl0_frame_stride : src.slab + (uint64_t)f * TL.slab_stride + GL.img_off;
        const int w = GL.w, h = GL.h, kx = K[k].x, ky = K[k].y;
        uint32_t *win = &s_patch[wv][k][0], *pu = win + 45 * 12, *pb = win;      // (pass 1 of the blur has read the whole window into registers before pass 2 writes the first blurred dword)
        if (kx >= 23 && kx + 24 < w && ky >= 22 && ky + 22 < h) {          // wave-uniform: the whole window lies inside the level
            const uint8_t *corner = img + (int64_t)(ky - 22) * pitch + (kx - 23);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int i = lane + 64 * t, r = i / 12, c = i - 12 * r;
                if (i < 45 * 12) win[i] = reinterpret_cast<const U32u *>(corner + (uint32_t)r * (uint32_t)pitch + 4u * c)->v;
            }
        } else {                                                          // a keypoint within 24 pixels of the border (a few per cent): BORDER_REFLECT_101 byte by byte
#pragma unroll 1
            for (int t = 0; t < 9; ++t) {
                const int i = lane + 64 * t, r = i / 12, c = i - 12 * r;
                if (i >= 45 * 12) break;
                const uint8_t *row = img + (int64_t)reflect101(ky - 22 + r, h) * pitch;
                uint32_t v = 0;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) v |= (uint32_t)row[reflect101(kx - 23 + 4 * c + bb, w)] << (8 * bb);
                win[i] = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 4; ++t) {                                   // orientation patch: rows y-15 .. y+15 = window rows 7 .., columns x-15 .. = window dword 2 ..;
            const int i = lane + 64 * t, r = i >> 3, c4 = i & 7;        // everything outside the radius-15 disc zeroed (the mask of the 32nd row is zero)
            const uint32_t v = i < 31 * 8 ? win[(r + 7) * 12 + c4 + 2] : 0u;
            pu[i] = v & dmask[t];
        }
        {   // blurred patch rows y-19 .. y+19 (output row ro <- window rows ro .. ro+6), columns x-19 .. x+19 = window bytes 4 .. 42, ON THE MATRIX CORES (round 3):
            // a separable 7-tap filter is two products with banded constant matrices, out = V (P Hh), and every quantity of k_blur's fixed-point arithmetic is an
            // integer that i8 operands with i32 accumulators carry exactly.
            //   pass 1   T' = (P - 128) Hh: A = window rows as they lie in LDS (lane (i, kg): row 16 m + i, bytes 16 kg .. + 15, one ds_read_b128, xor 0x80 = -128
            //            as signed bytes; the bytes past column 47 and the rows past 44 meet zero weights), B = the taps by column (host table).  T' = T - 32768 fits 16 bits
            //   between  T' leaves the accumulators as its high and low bytes (4 v_perm per tile for both planes); the C layout holds 4 consecutive ROWS of a column per
            //            lane, the next A operand wants 16 -- a 4 x 4 transpose of dwords across the wave's four 16-lane rows: two v_permlane32_swap + two
            //            v_permlane16_swap per plane and column tile, no trip through LDS
            //   pass 2   out^T = T'^T V^T as 256 x (high bytes) + (low bytes - 128): two chained products per 16 x 16 tile (the first starts from 33024 =
            //            (rounding + the offsets' share) / 256, is shifted up by 8 and becomes the second's C), byte 2 of each sum is the blurred pixel; the
            //            C layout now holds 4 consecutive pixels of a ROW per lane = one dword of the patch
            // 27 v_mfma_i32_16x16x64_i8 + ~150 vector instructions per keypoint, where the packed-16-bit form (k_blur's, 60 lanes x 8 rows) took ~350: the kernel sits at
            // the VALU issue limit and the matrix pipe was idle.
            const int n = lane & 15, q = lane >> 4;
            // B operands of tile t: lane (n, kg) holds rows 16 kg .. + 15 of column 16 t + n -- non-zero only for kg - t = 0 or 1, and for columns <= 38
            auto taps = [&](int base, int t) {
                const int dt = q - t;
                const bool on = (dt == 0 || dt == 1) && (t < 2 || n <= 6);
                const uint4 zero4 = {0u, 0u, 0u, 0u};
                return __builtin_bit_cast(v4i_t, on ? s_tab[base + (dt & 1) * 16 + n] : zero4);
            };
            v4i_t Aw[3], Vb[3];                              // the window rows as pass 1's A operands (read BEFORE the first blurred dword takes the window's place), the vertical taps
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                uint4 a4 = *reinterpret_cast<const uint4 *>(win + (16 * m + n) * 12 + 4 * q);
                a4.x ^= 0x80808080u; a4.y ^= 0x80808080u; a4.z ^= 0x80808080u; a4.w ^= 0x80808080u;
                Aw[m] = __builtin_bit_cast(v4i_t, a4);
                Vb[m] = taps(32, m);
            }
            // one column tile at a time (pass 1 -> bytes -> transposes -> pass 2 -> store): the three tiles' intermediates side by side cost the kernel its eighth wave per SIMD
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const v4i_t Hb = taps(0, t), zero = {0, 0, 0, 0};
                uint32_t hd[4], ld[4];
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const v4i_t T = __builtin_amdgcn_mfma_i32_16x16x64_i8(Aw[m], Hb, zero, 0, 0, 0);
                    const uint32_t x01 = __builtin_amdgcn_perm((uint32_t)T[1], (uint32_t)T[0], 0x04000501u);      // [r0.b1, r1.b1, r0.b0, r1.b0]
                    const uint32_t x23 = __builtin_amdgcn_perm((uint32_t)T[3], (uint32_t)T[2], 0x04000501u);
                    hd[m] = __builtin_amdgcn_perm(x23, x01, 0x05040100u);
                    ld[m] = __builtin_amdgcn_perm(x23, x01, 0x07060302u) ^ 0x80808080u;
                }
                hd[3] = 0; ld[3] = 0;                                   // rows 48 .. 63: zero weights AND zero data
                transpose4_rows(hd); transpose4_rows(ld);
                const v4i_t A2h = {(int)hd[0], (int)hd[1], (int)hd[2], (int)hd[3]}, A2l = {(int)ld[0], (int)ld[1], (int)ld[2], (int)ld[3]};
#pragma unroll
                for (int yt = 0; yt < 3; ++yt) {
                    v4i_t acc = {33024, 33024, 33024, 33024};
                    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(A2h, Vb[yt], acc, 0, 0, 0);
                    acc <<= 8;
                    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(A2l, Vb[yt], acc, 0, 0, 0);
                    const int y = 16 * yt + n;
                    if (y < 39 && 4 * t + q < 10)                       // lane (n, q): pixels 16 t + 4 q .. + 3 of patch row y (byte 2 of each sum; the sums stay below 2^24)
                        pb[y * 10 + 4 * t + q] = __builtin_amdgcn_perm((uint32_t)acc[1], (uint32_t)acc[0], 0x0C0C0602u) | __builtin_amdgcn_perm((uint32_t)acc[3], (uint32_t)acc[2], 0x06020C0Cu);
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // O1: moments m10 = sum u I, m01 = sum v I over the disc (orb_extractor.cpp:245-275; integer sums, any order).  Lane = one column
    // u of one half of the rows (v = -15..0 | 1..16, the 32nd row is zero): 16 byte reads at constant offsets from one address,
    // S = sum I and J = sum j I; then m10 = u S and m01 = J + v0 S.  The disc shape is already in the data (zeros outside).
    const int half = lane >= 31 ? 1 : 0, col = lane - 31 * half;
    float angle_deg[kDescPerWave], ca[kDescPerWave], sa[kDescPerWave];
#pragma unroll
    for (int k = 0; k < kDescPerWave; ++k) {
        const uint8_t *colp = reinterpret_cast<const uint8_t *>(&s_patch[wv][k][45 * 12]) + half * (16 * 32) + col;
        int S = 0, J = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int I = colp[j * 32];
            S += I;
            J += j * I;
        }
        int m10 = lane < 62 ? (col - kHalfPatch) * S : 0;
        int m01 = lane < 62 ? J + (half ? 1 : -kHalfPatch) * S : 0;
        m10 = wave_sum(m10); m01 = wave_sum(m01);
        angle_deg[k] = dev_fast_atan2((float)m01, (float)m10);
        // float(angleDeg * M_PI / 180.0) in double (orb_extractor.cpp:286).  One multiplication by the double M_PI / 180.0 gives the same
        // float for EVERY float in [0, 361] (checked exhaustively, tests/test_oracle_frontend.py::test_degree_to_radian_constant_is_exact),
        // and it replaces a software double division per keypoint.
        const float angle = (float)__dmul_rn((double)angle_deg[k], M_PI / 180.0);
        ca[k] = dev_cos(angle); sa[k] = dev_sin(angle);
    }
    // O2: steered BRIEF on the blurred patch; the lane's point pairs are shared by the wave's keypoints
    unsigned long long bits[kDescPerWave][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 pt = __builtin_bit_cast(float4, s_tab[64 + q * 64 + lane]);             // the test's two points, already float
        const float x1 = pt.x, y1 = pt.y, x2 = pt.z, y2 = pt.w;
#pragma unroll
        for (int k = 0; k < kDescPerWave; ++k) {
            const uint8_t *pb = reinterpret_cast<const uint8_t *>(&s_patch[wv][k][0]);
            const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, sa[k]), __fmul_rn(y1, ca[k])));
            const int c1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, ca[k]), __fmul_rn(y1, sa[k])));
            const int r2 = __float2int_rn(__fadd_rn(__fmul_rn(x2, sa[k]), __fmul_rn(y2, ca[k])));
            const int c2 = __float2int_rn(__fsub_rn(__fmul_rn(x2, ca[k]), __fmul_rn(y2, sa[k])));
            bits[k][q] = __ballot(pb[(r1 + kPatchRadius) * 40 + (c1 + kPatchRadius)] < pb[(r2 + kPatchRadius) * 40 + (c2 + kPatchRadius)]);
        }
    }
#pragma unroll
    for (int k = 0; k < kDescPerWave; ++k) {
        if (!K[k].valid) continue;                              // wave-uniform
        const uint64_t o = (uint64_t)f * capacity + slot0 + k;
        if (lane < 8) out_desc[o * 8 + lane] = (uint32_t)(bits[k][lane >> 1] >> ((lane & 1) * 32));
        if (lane == 0) { out_x[o] = K[k].ox; out_y[o] = K[k].oy; out_angle[o] = angle_deg[k]; out_octave[o] = K[k].oct; out_track[o] = K[k].tid_out; }
    }
}

}  // namespace

// =================================================================================================
// host side of the extractor
// =================================================================================================
struct ms_orb {
    ms_ctx *ctx = nullptr;
    ms_orb_config cfg{};
    PyrGeom geom{};
    PyrGeom *d_geom = nullptr;
    TileMap blur_tiles{}, fast_tiles{};
    uint32_t *d_ftile_tab = nullptr;   // k_fast: level | column << 4 | row << 16 of every tile
    TileLevels tile_levels{};          // per-level geometry of the tiled kernels, passed by value
    uint8_t *d_slab = nullptr;
    uint32_t *d_cand = nullptr;
    int32_t *d_cand_count = nullptr, *d_det_count = nullptr, *d_trk_count = nullptr;
    size_t cand_count_bytes = 0;
    int16_t *d_det_x = nullptr, *d_det_y = nullptr, *d_trk_x = nullptr, *d_trk_y = nullptr;
    uint8_t *d_det_score = nullptr, *d_mask = nullptr;
    float *d_trk_px = nullptr, *d_trk_py = nullptr, *d_track_xy = nullptr;
    int32_t *d_trk_id = nullptr, *d_track_id = nullptr, *d_n_tracks = nullptr;
    // outputs
    float *d_x = nullptr, *d_y = nullptr, *d_angle = nullptr;
    int32_t *d_octave = nullptr, *d_track = nullptr, *d_count = nullptr;
    uint2 *d_slot_tab = nullptr;              // k_slots: output slot -> (x | y << 16, level)
    uint32_t *d_desc = nullptr;
    // resize tables per level (device)
    int16_t *d_xtab[MS_MAX_LEVELS] = {nullptr}, *d_ytab[MS_MAX_LEVELS] = {nullptr};
    bool wide[MS_MAX_LEVELS] = {false};
    uint4 *d_moment_tab = nullptr;            // k_describe: disc mask of the orientation patch, four dwords per lane
    float4 *d_pattern_f = nullptr;            // k_describe: the 256 BRIEF point pairs as floats
    // optional per-stage HIP events (ms_orb_set_profiling)
    bool profiling = false;
    bool blur_valid = false;           // the blurred planes of the slab hold the last batch (k_blur runs on demand only)
    // a ring of kProfRing event sets: a profiled call records into the next one, so the stage times of the last kProfRing calls can be read
    // after a run of calls -- without a host wait between them (ms_orb_stage_ms_back)
    static constexpr int kProfRing = 128;
    hipEvent_t evr[kProfRing][MS_ORB_STAGES + 1] = {{nullptr}};
    hipEvent_t *ev = evr[0];           // the set of the current / last profiled call
    long long prof_calls = 0;
    // host frames come in as up to kChunks pieces on a stream of their own: piece k+1 is copied while the kernels of piece k run
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copied[4] = {nullptr}, ev_free[4] = {nullptr}, ev_end = nullptr;
    int chunked_frames = 0;            // n_frames of the last chunked call (its ev_free[] mark the pieces of exactly that split); 0 = none
    bool end_recorded = false;
    // state of the last call
    FrameSrc last_src{};
    int last_frames = 0;
    bool lvl0_in_slab = false;
    uint64_t lvl0_off = 0;
    int lvl0_pitch = 0;
};

template <typename T>
static int dev_calloc(ms_ctx *c, T **p, size_t n) {
    MS_HIP(c, hipMalloc(reinterpret_cast<void **>(p), n * sizeof(T) ? n * sizeof(T) : 1));
    MS_HIP(c, hipMemsetAsync(*p, 0, n * sizeof(T) ? n * sizeof(T) : 1, c->stream));
    return MS_OK;
}
#define MS_TRY(x) do { int rc__ = (x); if (rc__ != MS_OK) return rc__; } while (0)

// KeyPoint::serialize (key_point.hpp:22-25): ar(pt.x, pt.y, angle, octave, octave, bearing, descriptor) -- 19 dwords per keypoint
// (x, y, angle f32; octave i32 twice; bearing 3 x f64; descriptor 8 x u32).  Thread = one dword of one record: coalesced stores.
__global__ __launch_bounds__(256) void k_pack_keypoints(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ angle,
                                                       const int32_t *__restrict__ octave, const uint32_t *__restrict__ desc, const double *__restrict__ bearing,
                                                       uint64_t base, int n, uint32_t *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 19) return;
    const int k = i / 19, w = i - 19 * k;
    const uint64_t s = base + (uint64_t)k;
    uint32_t v;
    if (w == 0) v = __float_as_uint(x[s]);
    else if (w == 1) v = __float_as_uint(y[s]);
    else if (w == 2) v = __float_as_uint(angle[s]);
    else if (w < 5) v = (uint32_t)octave[s];
    else if (w < 11) v = bearing ? reinterpret_cast<const uint32_t *>(bearing)[6 * (uint64_t)k + (w - 5)] : 0u;
    else v = desc[s * 8 + (w - 11)];
    out[i] = v;
}

extern "C" {

int ms_orb_create(ms_ctx *ctx, const ms_orb_config *cfg, ms_orb **out) {
    if (!ctx || !cfg || !out) return MS_ERR_INVALID;
    *out = nullptr;
    if (cfg->levels < 1 || cfg->levels > MS_MAX_LEVELS || cfg->max_kpts < 1 || cfg->max_batch < 1 || cfg->max_tracks < 0 ||
        cfg->lk_track_level < 0 || cfg->lk_track_level >= cfg->levels || cfg->fast_threshold < 1 || cfg->fast_threshold > 254 ||
        !(cfg->scale_factor > 1.0f) || !(cfg->min_distance >= 0.f) || cfg->width > 32767 || cfg->height > 32767 ||
        (int64_t)cfg->width * cfg->height >= (1 << 24))
        return ms_fail(ctx, MS_ERR_INVALID, "ms_orb_create: unsupported configuration");
    MS_HIP(ctx, hipSetDevice(ctx->device));
    ms_orb *o = new ms_orb();
    o->ctx = ctx;
    o->cfg = *cfg;
    PyrGeom &G = o->geom;
    G.levels = cfg->levels; G.lk_level = cfg->lk_track_level; G.fast_threshold = cfg->fast_threshold;
    G.max_kpts = cfg->max_kpts; G.max_tracks = cfg->max_tracks;
    G.width = cfg->width; G.height = cfg->height;
    int32_t w[MS_MAX_LEVELS], h[MS_MAX_LEVELS], quota[MS_MAX_LEVELS];
    float sf[MS_MAX_LEVELS];
    msgeo::level_sizes(cfg->levels, cfg->scale_factor, cfg->width, cfg->height, w, h);
    msgeo::level_quotas(cfg->levels, cfg->scale_factor, cfg->max_kpts, quota);
    msgeo::scale_factors(cfg->levels, cfg->scale_factor, sf);
    msgeo::umax(G.umax);
    uint64_t off = 0, coff = 0;
    int det_base = 0, bt = 0, ft = 0;
    for (int l = 0; l < cfg->levels; ++l) {
        if (w[l] < 2 * kPatchRadius + 2 || h[l] < 2 * kPatchRadius + 2 || quota[l] > kMaxQuota) {
            delete o;
            return ms_fail(ctx, MS_ERR_INVALID, "ms_orb_create: level %d is %dx%d (min 40x40) / quota %d (max %d)", l, w[l], h[l], quota[l], kMaxQuota);
        }
        if ((int64_t)ms_div_up(w[l], kFastSeg) * ms_div_up(h[l], kFastRows) >= 65536) {      // the kernels' mulhi tile division is exact below 2^16 tiles per level
            delete o;
            return ms_fail(ctx, MS_ERR_CAPACITY, "ms_orb_create: level %d (%dx%d) needs more than 65535 detector tiles", l, w[l], h[l]);
        }
        LevelGeom &L = G.L[l];
        L.w = w[l]; L.h = h[l]; L.pitch = (int)ms_align_up(w[l], 64); L.quota = quota[l]; L.scale = sf[l];
        {   // feature_detector.cpp:79-82: minDist = floor(gfttMinDistance * (min(w,h) / 720 * 0.8) + 0.5)
            const double su = std::min(w[l], h[l]) / 720.0 * 0.8;
            L.min_dist = (int)std::floor(cfg->min_distance * su + 0.5);
        }
        L.det_base = det_base; det_base += quota[l];
        L.img_off = off; off += (uint64_t)L.pitch * L.h;
        L.blur_off = off; off += (uint64_t)L.pitch * L.h;
        L.cand_cap = (int32_t)ms_align_up((size_t)(((w[l] + 1) / 2) * ((h[l] + 1) / 2) + 256), 4);   // strict 3x3 maxima: <= one per 2x2 block; a multiple of 4 so that every level's list starts 16-byte aligned (k_select reads it as uint4)
        L.cand_off = coff; coff += L.cand_cap;
        L.btiles_x = ms_div_up(w[l], kBlurSeg); L.btile_base = bt; bt += L.btiles_x * ms_div_up(h[l], 4 * kBlurRows);
        L.ftiles_x = ms_div_up(w[l], kFastSeg); L.ftile_base = ft; ft += L.ftiles_x * ms_div_up(h[l], kFastRows);
        L.btiles_inv = (uint32_t)(((1ull << 32) + L.btiles_x - 1) / L.btiles_x); L.ftiles_inv = (uint32_t)(((1ull << 32) + L.ftiles_x - 1) / L.ftiles_x);
    }
    G.btiles_total = bt; G.ftiles_total = ft;
    G.det_stride = std::max(cfg->max_kpts, det_base);       // the reference keeps per-level vectors, so a frame can hold sum(quota) > maxKeypoints points
    G.capacity = G.det_stride + cfg->max_tracks;
    for (int l = 0; l < cfg->levels; ++l) {
        const LevelGeom &L = G.L[l];
        o->tile_levels.L[l] = TileLevel{L.w, L.h, L.pitch, L.btiles_x, L.ftiles_x, L.cand_cap, L.btiles_inv, L.ftiles_inv, L.img_off, L.blur_off, L.cand_off, L.scale, L.det_base};
    }
    for (int l = 0; l <= MS_MAX_LEVELS; ++l) {
        o->blur_tiles.base[l] = l < cfg->levels ? G.L[l].btile_base : bt;
        o->fast_tiles.base[l] = l < cfg->levels ? G.L[l].ftile_base : ft;
    }
    G.slab_stride = ms_align_up(off, 256); G.cand_stride = coff;
    o->tile_levels.slab_stride = G.slab_stride; o->tile_levels.cand_stride = G.cand_stride; o->tile_levels.levels = G.levels; o->tile_levels.fast_threshold = G.fast_threshold;
    o->lvl0_off = G.L[0].img_off; o->lvl0_pitch = G.L[0].pitch;
    const size_t B = cfg->max_batch, cap = G.capacity;
    int rc = MS_OK;
    auto A = [&](int r) { if (rc == MS_OK) rc = r; };
    A(dev_calloc(ctx, &o->d_geom, 1));
    A(dev_calloc(ctx, &o->d_slab, B * G.slab_stride + 256));   // + slack (round 1's k_describe could touch one byte past the last plane; today's window loads stay inside the level)
    A(dev_calloc(ctx, &o->d_cand, B * G.cand_stride));
    A(dev_calloc(ctx, &o->d_cand_count, B * MS_MAX_LEVELS));
    o->cand_count_bytes = sizeof(int32_t) * B * MS_MAX_LEVELS;
    A(dev_calloc(ctx, &o->d_det_count, B * MS_MAX_LEVELS));
    A(dev_calloc(ctx, &o->d_trk_count, B));
    A(dev_calloc(ctx, &o->d_det_x, B * (size_t)G.det_stride));
    A(dev_calloc(ctx, &o->d_det_y, B * (size_t)G.det_stride));
    A(dev_calloc(ctx, &o->d_det_score, B * (size_t)G.det_stride));
    const size_t T = (size_t)std::max(cfg->max_tracks, 1);
    A(dev_calloc(ctx, &o->d_trk_x, B * T)); A(dev_calloc(ctx, &o->d_trk_y, B * T));
    A(dev_calloc(ctx, &o->d_trk_px, B * T)); A(dev_calloc(ctx, &o->d_trk_py, B * T));
    A(dev_calloc(ctx, &o->d_trk_id, B * T)); A(dev_calloc(ctx, &o->d_track_xy, B * T * 2));
    A(dev_calloc(ctx, &o->d_track_id, B * T)); A(dev_calloc(ctx, &o->d_n_tracks, B));
    A(dev_calloc(ctx, &o->d_x, B * cap)); A(dev_calloc(ctx, &o->d_y, B * cap)); A(dev_calloc(ctx, &o->d_angle, B * cap));
    A(dev_calloc(ctx, &o->d_octave, B * cap)); A(dev_calloc(ctx, &o->d_track, B * cap)); A(dev_calloc(ctx, &o->d_count, B)); A(dev_calloc(ctx, &o->d_slot_tab, B * cap));
    A(dev_calloc(ctx, &o->d_desc, B * cap * 8));
    if (rc == MS_OK && hipMemcpyAsync(o->d_geom, &o->geom, sizeof(PyrGeom), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = MS_ERR_HIP;
    for (int l = 1; l < cfg->levels && rc == MS_OK; ++l) {
        std::vector<int16_t> xo, xc, yo, yc;
        msgeo::resize_tables(w[l - 1], w[l], true, xo, xc);
        msgeo::resize_tables(h[l - 1], h[l], false, yo, yc);
        const int wpad = ms_div_up(w[l], 4) * 4, hpad = ms_div_up(h[l], kResizeRows) * kResizeRows;     // whole groups, last entry repeated
        std::vector<int16_t> xt(4 * (size_t)wpad), yt(4 * (size_t)hpad);
        for (int d = 0; d < w[l]; ++d) { xt[4 * d] = xo[d]; xt[4 * d + 1] = xc[2 * d]; xt[4 * d + 2] = xc[2 * d + 1]; xt[4 * d + 3] = 0; }
        for (int d = 0; d < h[l]; ++d) {      // clip(sy, 0, ssize.height) applied to both tap rows
            yt[4 * d] = (int16_t)std::min(std::max((int)yo[d], 0), h[l - 1] - 1);
            yt[4 * d + 1] = (int16_t)std::min(std::max((int)yo[d] + 1, 0), h[l - 1] - 1);
            yt[4 * d + 2] = yc[2 * d]; yt[4 * d + 3] = yc[2 * d + 1];
        }
        for (int d = w[l]; d < wpad; ++d) for (int k = 0; k < 4; ++k) xt[4 * d + k] = xt[4 * (w[l] - 1) + k];
        for (int d = h[l]; d < hpad; ++d) for (int k = 0; k < 4; ++k) yt[4 * d + k] = yt[4 * (h[l] - 1) + k];
        for (int d = 0; d + 3 < w[l]; d += 4) if (xo[d + 3] + 1 - xo[d] > 7) o->wide[l] = true;   // taps beyond an 8-byte window
        auto up = [&](int16_t **d, const std::vector<int16_t> &v) {
            if (rc != MS_OK) return;
            if (hipMalloc(reinterpret_cast<void **>(d), v.size() * 2) != hipSuccess ||
                hipMemcpy(*d, v.data(), v.size() * 2, hipMemcpyHostToDevice) != hipSuccess) rc = MS_ERR_HIP;
        };
        up(&o->d_xtab[l], xt); up(&o->d_ytab[l], yt);
    }
    if (rc == MS_OK) {                                   // k_fast's tile table: level | column << 4 | row << 16
        std::vector<uint32_t> tt;
        for (int l = 0; l < cfg->levels; ++l) {
            const int tx = G.L[l].ftiles_x, ty = ms_div_up(G.L[l].h, kFastRows);
            if (tx > 0xFFF || ty > 0xFFFF) { rc = MS_ERR_INVALID; break; }
            for (int r = 0; r < ty; ++r) for (int cx = 0; cx < tx; ++cx) tt.push_back((uint32_t)l | ((uint32_t)cx << 4) | ((uint32_t)r << 16));
        }
        if (rc == MS_OK && ((int)tt.size() != G.ftiles_total || hipMalloc(reinterpret_cast<void **>(&o->d_ftile_tab), tt.size() * 4 + 16) != hipSuccess ||
                            hipMemcpy(o->d_ftile_tab, tt.data(), tt.size() * 4, hipMemcpyHostToDevice) != hipSuccess)) rc = MS_ERR_HIP;
    }
    if (rc == MS_OK) {                                   // k_describe's lane tables
        static const int8_t pattern[1024] = {
#include "orb_pattern.inc"
        };
        // disc mask of the 31 x 31 orientation patch as k_describe stages it: dword i = lane + 64 t covers columns 4 (i & 7) .. + 3 of row
        // i >> 3; a byte is kept when |u| <= u_max[|v|] (orb_extractor.cpp:174-186, :259-271), the 32nd column and row are cleared
        std::vector<uint32_t> mt(64 * 4, 0u);
        for (int lane = 0; lane < 64; ++lane)
            for (int t = 0; t < 4; ++t) {
                const int i = lane + 64 * t, r = i >> 3, c4 = i & 7;
                if (r >= 31) continue;
                for (int b = 0; b < 4; ++b) {
                    const int u = 4 * c4 + b - kHalfPatch, v = r - kHalfPatch;
                    if (u <= kHalfPatch && std::abs(u) <= G.umax[std::abs(v)]) mt[lane * 4 + t] |= 0xFFu << (8 * b);
                }
            }
        // behind it, k_describe's blur as matrix operands (B of v_mfma_i32_16x16x64_i8: lane (n, kg) holds rows k = 16 kg .. + 15 of column 16 t + n, one byte each): the taps
        // 18 34 48 56 48 34 18 by patch column x -- window byte k contributes to x when 0 <= k - x - 1 <= 6 -- and by patch row y (window row k, 0 <= k - y <= 6).  Only
        // dt = kg - t = 0 and 1 can be non-zero, and the pattern depends on (dt, n) alone: [table 0 = horizontal, 1 = vertical][dt][n][4 dwords]
        {
            static const int w7[7] = {18, 34, 48, 56, 48, 34, 18};
            mt.resize(64 * 4 + 64 * 4, 0u);
            for (int tab = 0; tab < 2; ++tab)
                for (int dt = 0; dt < 2; ++dt)
                    for (int n = 0; n < 16; ++n)
                        for (int j = 0; j < 16; ++j) {
                            const int tap = 16 * dt + j - n - (tab == 0 ? 1 : 0);
                            const uint32_t v = (tap >= 0 && tap <= 6) ? (uint32_t)w7[tap] : 0u;
                            mt[64 * 4 + ((tab * 2 + dt) * 16 + n) * 4 + j / 4] |= v << (8 * (j % 4));
                        }
        }
        std::vector<float> pf(1024);
        for (int i = 0; i < 1024; ++i) pf[i] = (float)pattern[i];
        if (hipMalloc(reinterpret_cast<void **>(&o->d_moment_tab), mt.size() * 4) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&o->d_pattern_f), pf.size() * 4) != hipSuccess ||
            hipMemcpy(o->d_moment_tab, mt.data(), mt.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(o->d_pattern_f, pf.data(), pf.size() * 4, hipMemcpyHostToDevice) != hipSuccess) rc = MS_ERR_HIP;
    }
    if (rc == MS_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = MS_ERR_HIP;
    if (rc != MS_OK) { ms_orb_destroy(o); return ms_fail(ctx, rc, "ms_orb_create: device allocation failed"); }
    {   // the copy stream and its events (batches of host frames only; a one-frame extractor never uses them)
        bool ok = hipEventCreateWithFlags(&o->ev_end, hipEventDisableTiming) == hipSuccess;
        if (cfg->max_batch >= 32) {
            ok = ok && hipStreamCreateWithFlags(&o->copy_stream, hipStreamNonBlocking) == hipSuccess;
            for (int i = 0; i < 4 && ok; ++i)
                ok = hipEventCreateWithFlags(&o->ev_copied[i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&o->ev_free[i], hipEventDisableTiming) == hipSuccess;
        }
        if (!ok) { ms_orb_destroy(o); return ms_fail(ctx, MS_ERR_HIP, "ms_orb_create: stream / event creation failed"); }
    }
    *out = o;
    return MS_OK;
}

void ms_orb_destroy(ms_orb *o) {
    if (!o) return;
    (void)hipSetDevice(o->ctx->device);
    (void)hipStreamSynchronize(o->ctx->stream);
    void *ptrs[] = {o->d_geom, o->d_slab, o->d_cand, o->d_cand_count, o->d_det_count, o->d_trk_count, o->d_det_x, o->d_det_y,
                    o->d_det_score, o->d_mask, o->d_trk_x, o->d_trk_y, o->d_trk_px, o->d_trk_py, o->d_trk_id, o->d_track_xy,
                    o->d_track_id, o->d_n_tracks, o->d_x, o->d_y, o->d_angle, o->d_octave, o->d_track, o->d_count, o->d_desc, o->d_slot_tab};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (o->d_moment_tab) (void)hipFree(o->d_moment_tab);
    if (o->d_pattern_f) (void)hipFree(o->d_pattern_f);
    if (o->d_ftile_tab) (void)hipFree(o->d_ftile_tab);
    for (auto &set : o->evr) for (int i = 0; i <= MS_ORB_STAGES; ++i) if (set[i]) (void)hipEventDestroy(set[i]);
    if (o->copy_stream) { (void)hipStreamSynchronize(o->copy_stream); (void)hipStreamDestroy(o->copy_stream); }
    for (int i = 0; i < 4; ++i) { if (o->ev_copied[i]) (void)hipEventDestroy(o->ev_copied[i]); if (o->ev_free[i]) (void)hipEventDestroy(o->ev_free[i]); }
    if (o->ev_end) (void)hipEventDestroy(o->ev_end);

    for (int l = 0; l < MS_MAX_LEVELS; ++l) {
        if (o->d_xtab[l]) (void)hipFree(o->d_xtab[l]);
        if (o->d_ytab[l]) (void)hipFree(o->d_ytab[l]);
    }
    delete o;
}

int ms_orb_capacity(const ms_orb *o) { return o ? o->geom.capacity : MS_ERR_INVALID; }

int ms_orb_set_profiling(ms_orb *o, int enable) {
    if (!o) return MS_ERR_INVALID;
    ms_ctx *c = o->ctx;
    if (enable && !o->evr[0][0])
        for (auto &set : o->evr) for (int i = 0; i <= MS_ORB_STAGES; ++i) MS_HIP(c, hipEventCreate(&set[i]));
    o->profiling = enable != 0;
    return MS_OK;
}

int ms_orb_stage_ms_back(ms_orb *o, int calls_back, float *ms) {
    if (!o || !ms || calls_back < 0 || calls_back >= ms_orb::kProfRing) return MS_ERR_INVALID;
    ms_ctx *c = o->ctx;
    if (!o->evr[0][0] || calls_back >= o->prof_calls) return ms_fail(c, MS_ERR_INVALID, "ms_orb_stage_ms_back: no profiled call %d calls back", calls_back);
    hipEvent_t *set = o->evr[(o->prof_calls - 1 - calls_back) % ms_orb::kProfRing];
    MS_HIP(c, hipEventSynchronize(set[MS_ORB_STAGES]));
    for (int i = 0; i < MS_ORB_STAGES; ++i) MS_HIP(c, hipEventElapsedTime(&ms[i], set[i], set[i + 1]));
    return MS_OK;
}

int ms_orb_stage_ms(ms_orb *o, float *ms) {
    if (!o || !ms) return MS_ERR_INVALID;
    ms_ctx *c = o->ctx;
    if (!o->profiling || o->last_frames == 0) return ms_fail(c, MS_ERR_INVALID, "ms_orb_stage_ms: profiling is off or nothing ran");
    MS_HIP(c, hipEventSynchronize(o->ev[MS_ORB_STAGES]));
    for (int i = 0; i < MS_ORB_STAGES; ++i) MS_HIP(c, hipEventElapsedTime(&ms[i], o->ev[i], o->ev[i + 1]));
    return MS_OK;
}

int ms_orb_set_valid_mask(ms_orb *o, const uint8_t *mask) {
    if (!o) return MS_ERR_INVALID;
    ms_ctx *c = o->ctx;
    MS_HIP(c, hipStreamSynchronize(c->stream));
    if (!mask) { if (o->d_mask) { MS_HIP(c, hipFree(o->d_mask)); o->d_mask = nullptr; } return MS_OK; }
    const size_t n = (size_t)o->cfg.width * o->cfg.height;
    if (!o->d_mask) MS_HIP(c, hipMalloc(reinterpret_cast<void **>(&o->d_mask), n));
    MS_HIP(c, hipMemcpy(o->d_mask, mask, n, hipMemcpyHostToDevice));
    return MS_OK;
}

static int orb_extract_enqueue(ms_orb *o, const uint8_t *images, int on_device, int n_frames, size_t frame_stride, size_t row_stride,
                               const float *track_xy, const int32_t *track_id, const int32_t *n_tracks, bool &counters_dirty);

int ms_orb_extract(ms_orb *o, const uint8_t *images, int on_device, int n_frames, size_t frame_stride, size_t row_stride,
                   const float *track_xy, const int32_t *track_id, const int32_t *n_tracks) {
    if (!o || !images) return MS_ERR_INVALID;
    bool counters_dirty = false;
    const int rc = orb_extract_enqueue(o, images, on_device, n_frames, frame_stride, row_stride, track_xy, track_id, n_tracks, counters_dirty);
    // k_fast fills the candidate counters and k_select puts them back to zero; a call that fails in between must not leave them to the next one
    if (rc != MS_OK && counters_dirty) (void)hipMemsetAsync(o->d_cand_count, 0, o->cand_count_bytes, o->ctx->stream);
    return rc;
}

// the kernels of one extract over frames [f0, f0 + nf) of the batch: every per-frame array is linear in the frame index, so a piece of the batch
// is the same launches on offset pointers
static int orb_enqueue_kernels(ms_orb *o, FrameSrc src, int f0, int nf, bool have_tracks, bool have_ids, bool &counters_dirty) {
    ms_ctx *c = o->ctx;
    const PyrGeom &G = o->geom;
    hipStream_t st = c->stream;
    const size_t F = (size_t)f0, T = (size_t)o->cfg.max_tracks, L = (size_t)G.levels;
    src.lvl0 += F * src.lvl0_frame_stride; src.slab += F * G.slab_stride;
    uint32_t *cand = o->d_cand + F * G.cand_stride;
    int32_t *cand_count = o->d_cand_count + F * L, *det_count = o->d_det_count + F * L, *trk_count = o->d_trk_count + F;
    int16_t *det_x = o->d_det_x + F * G.det_stride, *det_y = o->d_det_y + F * G.det_stride;
    uint8_t *det_score = o->d_det_score + F * G.det_stride;
    int stage = 0;
    if (o->profiling) { o->ev = o->evr[o->prof_calls % ms_orb::kProfRing]; ++o->prof_calls; }
#define MS_STAGE_MARK() do { if (o->profiling) MS_HIP(c, hipEventRecord(o->ev[stage++], st)); } while (0)
    MS_STAGE_MARK();
    MsRange pyramid_range("pyramid");
    for (int l = 1; l < G.levels; ++l) {
        dim3 grid(ms_div_up(G.L[l].w, 256), ms_div_up(G.L[l].h, 4 * kResizeRows), nf);
        const ResizeTab RT{o->d_xtab[l], o->d_ytab[l]};
        ResizeArgs RA{};
        if (l == 1) { RA.src = src.lvl0; RA.src_frame_stride = src.lvl0_frame_stride; RA.src_pitch = src.lvl0_pitch; }
        else { RA.src = src.slab + G.L[l - 1].img_off; RA.src_frame_stride = G.slab_stride; RA.src_pitch = G.L[l - 1].pitch; }
        RA.sw = G.L[l - 1].w;
        RA.dst = src.slab + G.L[l].img_off; RA.dst_frame_stride = G.slab_stride; RA.dst_pitch = G.L[l].pitch; RA.dw = G.L[l].w; RA.dh = G.L[l].h;
        if (o->wide[l]) hipLaunchKernelGGL(k_resize<true>, grid, dim3(256), 0, st, RA, RT);
        else hipLaunchKernelGGL(k_resize<false>, grid, dim3(256), 0, st, RA, RT);
        MS_KERNEL_CHECK(c, "k_resize");
    }
    pyramid_range.end();
    MS_STAGE_MARK();
    // The blurred pyramid (image_pyramid.cpp:82-85) is not materialised per frame any more: its only consumer on the path, the BRIEF tests,
    // blurs its own 39 x 39 patches inside k_describe.  ImagePyramid::getBlurredLevel (ms_orb_download_level) runs k_blur on demand.
    o->blur_valid = false;
    MS_STAGE_MARK();
    MsRange detect_range("detect");
    counters_dirty = true;
    hipLaunchKernelGGL(k_fast, dim3(G.ftiles_total, nf), dim3(kFastThreads), 0, st, src, o->d_geom, cand, cand_count, o->d_ftile_tab, o->tile_levels);
    MS_KERNEL_CHECK(c, "k_fast");
    MS_STAGE_MARK();
    const bool few = nf * G.levels <= 64;                           // a frame or a handful: one block per level is the whole launch -- give it 1024 threads
    if (o->cfg.min_distance > 0.f) {
        if (few) hipLaunchKernelGGL((k_select<true, 1024>), dim3(G.levels, nf), dim3(1024), 0, st, o->d_geom, cand, cand_count, o->d_mask, det_x, det_y, det_score, det_count);
        else hipLaunchKernelGGL((k_select<true, 256>), dim3(G.levels, nf), dim3(256), 0, st, o->d_geom, cand, cand_count, o->d_mask, det_x, det_y, det_score, det_count);
    } else {
        if (few) hipLaunchKernelGGL((k_select<false, 1024>), dim3(G.levels, nf), dim3(1024), 0, st, o->d_geom, cand, cand_count, o->d_mask, det_x, det_y, det_score, det_count);
        else hipLaunchKernelGGL((k_select<false, 256>), dim3(G.levels, nf), dim3(256), 0, st, o->d_geom, cand, cand_count, o->d_mask, det_x, det_y, det_score, det_count);
    }
    MS_KERNEL_CHECK(c, "k_select");
    counters_dirty = false;
    MS_STAGE_MARK();
    if (o->cfg.max_tracks > 0) {                                    // (an extractor built without tracker features: d_trk_count stays at its initial zero)
        hipLaunchKernelGGL(k_tracks, dim3(nf), dim3(64), 0, st, o->d_geom, have_tracks ? o->d_track_xy + F * T * 2 : nullptr,
                           (have_tracks && have_ids) ? o->d_track_id + F * T : nullptr, have_tracks ? o->d_n_tracks + F : nullptr, o->d_mask,
                           o->d_trk_x + F * T, o->d_trk_y + F * T, o->d_trk_px + F * T, o->d_trk_py + F * T, o->d_trk_id + F * T, trk_count);
        MS_KERNEL_CHECK(c, "k_tracks");
    }
    MS_STAGE_MARK();
    detect_range.end();
    MsRange describe_range("describe");
    const size_t C = (size_t)G.capacity;
    hipLaunchKernelGGL(k_slots, dim3(G.levels, nf), dim3(256), 0, st, det_x, det_y, det_count, trk_count, o->d_slot_tab + F * C, o->d_count + F, o->tile_levels, G.det_stride, G.capacity);
    MS_KERNEL_CHECK(c, "k_slots");
    hipLaunchKernelGGL(k_describe, dim3(ms_div_up(G.capacity, 4 * kDescPerWave), nf), dim3(256), 0, st, src, o->tile_levels, o->d_moment_tab, o->d_pattern_f,
                       o->d_slot_tab + F * C, G.capacity, G.max_tracks, G.lk_level, o->d_trk_x + F * T, o->d_trk_y + F * T, o->d_trk_px + F * T, o->d_trk_py + F * T, o->d_trk_id + F * T, trk_count, o->d_x + F * C, o->d_y + F * C,
                       o->d_angle + F * C, o->d_octave + F * C, o->d_desc + F * C * 8, o->d_track + F * C, o->d_count + F);
    MS_KERNEL_CHECK(c, "k_describe");
    MS_STAGE_MARK();
#undef MS_STAGE_MARK
    return MS_OK;
}

static int orb_extract_enqueue(ms_orb *o, const uint8_t *images, int on_device, int n_frames, size_t frame_stride, size_t row_stride,
                               const float *track_xy, const int32_t *track_id, const int32_t *n_tracks, bool &counters_dirty) {
    ms_ctx *c = o->ctx;
    const PyrGeom &G = o->geom;
    if (n_frames < 1 || n_frames > o->cfg.max_batch) return ms_fail(c, MS_ERR_CAPACITY, "ms_orb_extract: n_frames %d outside [1,%d]", n_frames, o->cfg.max_batch);
    if (row_stride < (size_t)G.width || frame_stride < row_stride * (size_t)(G.height - 1) + G.width)
        return ms_fail(c, MS_ERR_INVALID, "ms_orb_extract: strides smaller than the frame");
    MS_HIP(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    MS_TRY(ms_ctx_order_after_downloads(c));               // asynchronous downloads of the previous outputs (ms_dev_download_async) finish before they are overwritten
    FrameSrc src{};
    src.slab = o->d_slab;
    const bool have_tracks = track_xy && n_tracks && o->cfg.max_tracks > 0;
    if (have_tracks) {
        const size_t T = o->cfg.max_tracks;
        MS_HIP(c, hipMemcpyAsync(o->d_track_xy, track_xy, (size_t)n_frames * T * 2 * sizeof(float), hipMemcpyHostToDevice, st));
        MS_HIP(c, hipMemcpyAsync(o->d_n_tracks, n_tracks, (size_t)n_frames * sizeof(int32_t), hipMemcpyHostToDevice, st));
        if (track_id) MS_HIP(c, hipMemcpyAsync(o->d_track_id, track_id, (size_t)n_frames * T * sizeof(int32_t), hipMemcpyHostToDevice, st));
    }
    // (d_cand_count is zero here: allocated zeroed, and k_select puts every counter it has consumed back to zero)
    const bool aligned = on_device && (reinterpret_cast<uintptr_t>(images) % 16 == 0) && (row_stride % 16 == 0) && (frame_stride % 16 == 0);
    if (aligned) {   // use the caller's frames in place as pyramid level 0 (image_pyramid.cpp:75 without the copy)
        src.lvl0 = images; src.lvl0_frame_stride = frame_stride; src.lvl0_pitch = (int)row_stride;
        o->lvl0_in_slab = false;
        MS_TRY(orb_enqueue_kernels(o, src, 0, n_frames, have_tracks, track_id != nullptr, counters_dirty));
    } else {
        uint8_t *dst = o->d_slab + o->lvl0_off;
        src.lvl0 = dst; src.lvl0_frame_stride = G.slab_stride; src.lvl0_pitch = o->lvl0_pitch;
        o->lvl0_in_slab = true;
        // frames whose rows are as far apart on the host as in the slab move as whole frames: one linear copy each, or ONE 2-D copy for a run of frames
        // ("rows" = whole frames, host pitch = frame_stride, slab pitch = slab_stride) -- 256 x 21 us of per-frame copies were 5.4 ms of a 9.7 ms step
        const bool whole = row_stride == (size_t)o->lvl0_pitch;
        const size_t frame_bytes = row_stride * (size_t)(G.height - 1) + G.width;
        auto copy_frames = [&](int f0, int nf, hipStream_t cs) -> int {
            if (whole && nf > 1) {
                MS_HIP(c, hipMemcpy2DAsync(dst + (size_t)f0 * G.slab_stride, G.slab_stride, images + (size_t)f0 * frame_stride, frame_stride, frame_bytes, nf, hipMemcpyHostToDevice, cs));
            } else {
                for (int f = f0; f < f0 + nf; ++f) {
                    if (whole) MS_HIP(c, hipMemcpyAsync(dst + (size_t)f * G.slab_stride, images + (size_t)f * frame_stride, frame_bytes, hipMemcpyHostToDevice, cs));
                    else MS_HIP(c, hipMemcpy2DAsync(dst + (size_t)f * G.slab_stride, o->lvl0_pitch, images + (size_t)f * frame_stride, row_stride, G.width, G.height, hipMemcpyHostToDevice, cs));
                }
            }
            return MS_OK;
        };
        const int kChunks = 4;
        const bool chunked = !on_device && !o->profiling && n_frames >= 8 * kChunks && o->copy_stream != nullptr;
        if (on_device) {
            dim3 grid(ms_div_up(G.width, 256), G.height, n_frames);
            hipLaunchKernelGGL(k_copy_level0, grid, dim3(256), 0, st, images, (uint64_t)frame_stride, (uint64_t)row_stride, o->d_slab,
                               G.slab_stride, o->lvl0_off, G.width, G.height, o->lvl0_pitch);
            MS_KERNEL_CHECK(c, "k_copy_level0");
            MS_TRY(orb_enqueue_kernels(o, src, 0, n_frames, have_tracks, track_id != nullptr, counters_dirty));
        } else if (!chunked) {
            MS_TRY(copy_frames(0, n_frames, st));
            MS_TRY(orb_enqueue_kernels(o, src, 0, n_frames, have_tracks, track_id != nullptr, counters_dirty));
        } else {
            // Piece k's kernels run under piece k+1's copy (the copy engine and the CUs work side by side; the design rule: copies on a stream of their own).
            // A piece of the slab may be overwritten once the PREVIOUS call's kernels on it are done: its ev_free when that call was split the same way,
            // else the end of that call.  With pinned host memory the call returns after enqueueing; pageable memory is staged by the runtime (correct, no overlap).
            const int per = ms_div_up(n_frames, kChunks);
            const bool same_split = o->chunked_frames == n_frames;      // (0 unless the LAST call was split: any other call resets it)
            if (!same_split && o->end_recorded) MS_HIP(c, hipStreamWaitEvent(o->copy_stream, o->ev_end, 0));
            for (int k = 0; k < kChunks; ++k) {
                const int f0 = k * per, nf = std::min(per, n_frames - f0);
                if (nf <= 0) break;
                if (same_split) MS_HIP(c, hipStreamWaitEvent(o->copy_stream, o->ev_free[k], 0));
                MS_TRY(copy_frames(f0, nf, o->copy_stream));
                MS_HIP(c, hipEventRecord(o->ev_copied[k], o->copy_stream));
                MS_HIP(c, hipStreamWaitEvent(st, o->ev_copied[k], 0));
                MS_TRY(orb_enqueue_kernels(o, src, f0, nf, have_tracks, track_id != nullptr, counters_dirty));
                MS_HIP(c, hipEventRecord(o->ev_free[k], st));
            }
            o->chunked_frames = n_frames;
        }
        if (!chunked) o->chunked_frames = 0;
    }
    if (aligned) o->chunked_frames = 0;
    if (o->ev_end) { MS_HIP(c, hipEventRecord(o->ev_end, st)); o->end_recorded = true; }
    o->last_src = src;
    o->last_frames = n_frames;
    return MS_OK;
}

int ms_keypoints_pack(ms_ctx *c, const ms_keypoints *view, int frame, int n, const double *bearing, uint8_t *records_host) {
    if (!c || !view || !view->x || !view->y || !view->angle || !view->octave || !view->desc || frame < 0 || n < 0 || n > view->capacity || (n && !records_host))
        return MS_ERR_INVALID;
    if (n == 0) return MS_OK;
    MS_HIP(c, hipSetDevice(c->device));
    void *scratch = nullptr;
    MS_TRY(ms_scratch(c, (size_t)n * MS_KEYPOINT_RECORD_BYTES, &scratch));
    hipLaunchKernelGGL(k_pack_keypoints, dim3(ms_div_up(n * 19, 256)), dim3(256), 0, c->stream, view->x, view->y, view->angle, view->octave, view->desc, bearing,
                       (uint64_t)frame * (uint64_t)view->capacity, n, static_cast<uint32_t *>(scratch));
    MS_KERNEL_CHECK(c, "k_pack_keypoints");
    MS_HIP(c, hipMemcpyAsync(records_host, scratch, (size_t)n * MS_KEYPOINT_RECORD_BYTES, hipMemcpyDeviceToHost, c->stream));
    MS_HIP(c, hipStreamSynchronize(c->stream));
    return MS_OK;
}

int ms_keypoints_unpack(const uint8_t *records, int n, float *x, float *y, float *angle, int32_t *octave, double *bearing, uint32_t *desc) {
    if (n < 0 || (n && !records)) return MS_ERR_INVALID;
    for (int k = 0; k < n; ++k) {
        const uint8_t *r = records + (size_t)k * MS_KEYPOINT_RECORD_BYTES;
        int32_t o1, o2;
        std::memcpy(&o1, r + 12, 4); std::memcpy(&o2, r + 16, 4);
        if (o1 != o2) return MS_ERR_INVALID;                    // KeyPoint::serialize writes `octave` twice; a record where they differ is corrupt
        if (x) std::memcpy(x + k, r, 4);
        if (y) std::memcpy(y + k, r + 4, 4);
        if (angle) std::memcpy(angle + k, r + 8, 4);
        if (octave) octave[k] = o1;
        if (bearing) std::memcpy(bearing + 3 * (size_t)k, r + 20, 24);
        if (desc) std::memcpy(desc + 8 * (size_t)k, r + 44, 32);
    }
    return MS_OK;
}

int ms_orb_device_view(ms_orb *o, ms_keypoints *v) {
    if (!o || !v) return MS_ERR_INVALID;
    v->capacity = o->geom.capacity; v->count = o->d_count; v->x = o->d_x; v->y = o->d_y; v->angle = o->d_angle;
    v->octave = o->d_octave; v->desc = o->d_desc; v->track_id = o->d_track;
    return MS_OK;
}

int ms_orb_download(ms_orb *o, int frame, float *x, float *y, float *angle, int32_t *octave, uint32_t *desc, int32_t *track_id, int32_t *n) {
    if (!o || !n) return MS_ERR_INVALID;
    ms_ctx *c = o->ctx;
    if (frame < 0 || frame >= o->last_frames) return ms_fail(c, MS_ERR_INVALID, "ms_orb_download: frame %d not in last batch", frame);
    MS_HIP(c, hipStreamSynchronize(c->stream));
    int32_t cnt = 0;
    MS_HIP(c, hipMemcpy(&cnt, o->d_count + frame, sizeof(int32_t), hipMemcpyDeviceToHost));
    const size_t cap = o->geom.capacity, b = (size_t)frame * cap, k = (size_t)cnt;
    if (x) MS_HIP(c, hipMemcpy(x, o->d_x + b, k * 4, hipMemcpyDeviceToHost));
    if (y) MS_HIP(c, hipMemcpy(y, o->d_y + b, k * 4, hipMemcpyDeviceToHost));
    if (angle) MS_HIP(c, hipMemcpy(angle, o->d_angle + b, k * 4, hipMemcpyDeviceToHost));
    if (octave) MS_HIP(c, hipMemcpy(octave, o->d_octave + b, k * 4, hipMemcpyDeviceToHost));
    if (desc) MS_HIP(c, hipMemcpy(desc, o->d_desc + b * 8, k * 32, hipMemcpyDeviceToHost));
    if (track_id) MS_HIP(c, hipMemcpy(track_id, o->d_track + b, k * 4, hipMemcpyDeviceToHost));
    *n = cnt;
    return MS_OK;
}

int ms_orb_level_size(const ms_orb *o, int level, int32_t *w, int32_t *h) {
    if (!o || level < 0 || level >= o->geom.levels || !w || !h) return MS_ERR_INVALID;
    *w = o->geom.L[level].w; *h = o->geom.L[level].h;
    return MS_OK;
}

int ms_orb_download_level(ms_orb *o, int frame, int level, int blurred, uint8_t *dst) {
    if (!o || !dst) return MS_ERR_INVALID;
    ms_ctx *c = o->ctx;
    if (frame < 0 || frame >= o->last_frames || level < 0 || level >= o->geom.levels) return ms_fail(c, MS_ERR_INVALID, "ms_orb_download_level: bad frame/level");
    MS_HIP(c, hipStreamSynchronize(c->stream));
    const LevelGeom &L = o->geom.L[level];
    const uint8_t *p; size_t pitch;
    if (blurred) {
        if (!o->blur_valid) {                                  // the blurred levels of the last batch, computed when somebody asks for them
            hipLaunchKernelGGL(k_blur, dim3(o->geom.btiles_total, o->last_frames), dim3(256), 0, c->stream, o->last_src, o->tile_levels, o->blur_tiles);
            MS_KERNEL_CHECK(c, "k_blur");
            MS_HIP(c, hipStreamSynchronize(c->stream));
            o->blur_valid = true;
        }
        p = o->d_slab + (size_t)frame * o->geom.slab_stride + L.blur_off; pitch = L.pitch;
    }
    else if (level == 0) { p = o->last_src.lvl0 + (size_t)frame * o->last_src.lvl0_frame_stride; pitch = o->last_src.lvl0_pitch; }
    else { p = o->d_slab + (size_t)frame * o->geom.slab_stride + L.img_off; pitch = L.pitch; }
    MS_HIP(c, hipMemcpy2D(dst, L.w, p, pitch, L.w, L.h, hipMemcpyDeviceToHost));
    return MS_OK;
}

int ms_orb_download_detections(ms_orb *o, int frame, int level, int32_t *x, int32_t *y, int32_t *score, int32_t *n) {
    if (!o || !n) return MS_ERR_INVALID;
    ms_ctx *c = o->ctx;
    if (frame < 0 || frame >= o->last_frames || level < 0 || level >= o->geom.levels) return ms_fail(c, MS_ERR_INVALID, "ms_orb_download_detections: bad frame/level");
    MS_HIP(c, hipStreamSynchronize(c->stream));
    int32_t cnt = 0;
    MS_HIP(c, hipMemcpy(&cnt, o->d_det_count + frame * o->geom.levels + level, 4, hipMemcpyDeviceToHost));
    const size_t b = (size_t)frame * o->geom.det_stride + o->geom.L[level].det_base;
    std::vector<int16_t> hx(cnt), hy(cnt); std::vector<uint8_t> hs(cnt);
    if (cnt) {
        MS_HIP(c, hipMemcpy(hx.data(), o->d_det_x + b, cnt * 2, hipMemcpyDeviceToHost));
        MS_HIP(c, hipMemcpy(hy.data(), o->d_det_y + b, cnt * 2, hipMemcpyDeviceToHost));
        MS_HIP(c, hipMemcpy(hs.data(), o->d_det_score + b, cnt, hipMemcpyDeviceToHost));
    }
    for (int i = 0; i < cnt; ++i) { if (x) x[i] = hx[i]; if (y) y[i] = hy[i]; if (score) score[i] = hs[i]; }
    *n = cnt;
    return MS_OK;
}

}  // extern "C"
