// match.hip -- descriptor matching on gfx950: Hamming best/second-best search and the two
// BoW-bucketed greedy matchers of keyframe_matcher.cpp with their exact sequential semantics.
//
//   ms_hamming_best2        scoring core of every matcher (compute_descriptor_distance_32,
//                           openvslam/match_base.h:18-39; update rule keyframe_matcher.cpp:106-112)
//   ms_ratio_test           accept rule keyframe_matcher.cpp:115-122
//   ms_match_loop_closure   matchForLoopClosures       keyframe_matcher.cpp:50-158
//   ms_match_triangulation  matchForTriangulationDBoW  keyframe_matcher.cpp:160-293
//
// The brute-force search is compute-bound, not HBM-bound (32 distance evaluations per input byte at 2000x2000).
// Unmasked it runs on the matrix cores (k_hamming_mfma: the distance matrix is an i8 product of +-1 bytes).  With bucket /
// validity masks it stays on the VALU (k_hamming_best2<true>): each lane keeps one 256-bit query in 8 VGPRs, targets are
// staged through LDS in tiles of 256 and read back as wave-uniform broadcasts (2 x ds_read_b128 per target), distance is
// 8 x (v_xor, v_bcnt_u32_b32-accumulate), best/second are tracked branch-free on packed (distance<<20 | index) keys.
#include "ms_internal.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t hamming8(const uint32_t q[8], const uint4 a, const uint4 b) {
    uint32_t d = __popc(q[0] ^ a.x);
    d += __popc(q[1] ^ a.y); d += __popc(q[2] ^ a.z); d += __popc(q[3] ^ a.w);
    d += __popc(q[4] ^ b.x); d += __popc(q[5] ^ b.y); d += __popc(q[6] ^ b.z); d += __popc(q[7] ^ b.w);
    return d;
}

struct HamArgs {
    const uint32_t *q, *t;            // descriptor pools
    int32_t q_stride, t_stride;       // rows between consecutive sets of a pool
    const int32_t *q_count, *t_count; // optional per-set row counts (device); NULL = nq / nt
    const int32_t *pair_q, *pair_t;   // optional set index of each pair; NULL = pair index
    int32_t nq, nt;                   // rows per set (upper bound when counts are given)
    const int32_t *qb, *tb;           // optional bucket ids, laid out like the pools
    const uint8_t *tv;                // optional target validity, laid out like the target pool
    int32_t *best_idx; uint16_t *best_dist, *second_dist;   // [n_pairs * q_stride]
};

template <bool MASKED>
__global__ __launch_bounds__(256) void k_hamming_best2(HamArgs A) {
    __shared__ uint4 s_t[256][2];
    __shared__ int32_t s_b[256];
    __shared__ uint8_t s_v[256];
    const int p = blockIdx.y, tid = threadIdx.x;
    const int qi = blockIdx.x * 256 + tid;
    const int qs = A.pair_q ? A.pair_q[p] : p, ts = A.pair_t ? A.pair_t[p] : p;
    const int nq = A.q_count ? min(A.q_count[qs], A.nq) : A.nq;
    const int nt = A.t_count ? min(A.t_count[ts], A.nt) : A.nt;
    const uint64_t o = (uint64_t)p * A.q_stride + qi;
    if (blockIdx.x * 256 >= nq) {                      // whole block beyond this pair's queries
        if (qi < A.q_stride) { A.best_idx[o] = -1; A.best_dist[o] = MS_HAMMING_MAX; A.second_dist[o] = MS_HAMMING_MAX; }
        return;
    }
    const uint4 *Q = reinterpret_cast<const uint4 *>(A.q + (uint64_t)qs * A.q_stride * 8);
    const uint4 *T = reinterpret_cast<const uint4 *>(A.t + (uint64_t)ts * A.t_stride * 8);
    uint32_t qr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int32_t my_bucket = 0;
    if (qi < nq) {
        const uint4 a = Q[2 * qi], b = Q[2 * qi + 1];
        qr[0] = a.x; qr[1] = a.y; qr[2] = a.z; qr[3] = a.w; qr[4] = b.x; qr[5] = b.y; qr[6] = b.z; qr[7] = b.w;
        if (MASKED && A.qb) my_bucket = A.qb[(uint64_t)qs * A.q_stride + qi];
    }
    uint32_t best = kNone, second = kNone;
    for (int base = 0; base < nt; base += 256) {
        __syncthreads();
        const int j = base + tid;
        if (j < nt) {
            s_t[tid][0] = T[2 * j]; s_t[tid][1] = T[2 * j + 1];
            if (MASKED) {
                s_b[tid] = A.tb ? A.tb[(uint64_t)ts * A.t_stride + j] : 0;
                s_v[tid] = A.tv ? A.tv[(uint64_t)ts * A.t_stride + j] : 1;
            }
        }
        __syncthreads();
        const int cnt = min(256, nt - base);
        for (int k = 0; k < cnt; ++k) {
            const uint32_t d = hamming8(qr, s_t[k][0], s_t[k][1]);
            uint32_t key = (d << 20) | (uint32_t)(base + k);
            if (MASKED) {
                const bool ok = s_v[k] != 0 && (!(A.qb && A.tb) || s_b[k] == my_bucket);
                key = ok ? key : kNone;
            }
            const uint32_t lo = min(best, key), hi = max(best, key);
            second = min(second, hi);
            best = lo;
        }
    }
    if (qi < A.q_stride) {
        const bool live = qi < nq;
        A.best_idx[o] = (!live || best == kNone) ? -1 : (int32_t)(best & 0xFFFFFu);
        A.best_dist[o] = (!live || best == kNone) ? (uint16_t)MS_HAMMING_MAX : (uint16_t)(best >> 20);
        A.second_dist[o] = (!live || second == kNone) ? (uint16_t)MS_HAMMING_MAX : (uint16_t)(second >> 20);
    }
}

// ------------------------------------------------------------------------------------------------
// The unmasked all-pairs search as an integer matrix product on the matrix cores.  Queries become +-1 bytes (bit 0 -> +1,
// bit 1 -> -1), targets stay 0 / 1 bytes; then  hamming(q, t) = popcount(q) + sum_k t_k * (1 - 2 q_k)  exactly, i.e. the i8 dot
// product plus a per-query constant that does not change the order of the targets.  A 32x32 tile of distances is eight
// v_mfma_i32_32x32x32_i8 (K = 256 bits).
//   * target bytes cost two full-rate VALU ops per dword: (word >> s) & 0x01010101 puts bits s, s+8, s+16, s+24 of a descriptor
//     word into the four bytes -- which bit sits in which k slot of a fragment is irrelevant as long as queries and targets use
//     the same rule, so the k order is chosen to make the expansion free of multiplies and byte shuffles;
//   * a wave owns 64 queries (two 32-wide column tiles) whose operand fragments live in 64 registers for the whole kernel;
//   * the workgroup (4 waves = 256 queries) expands 128 targets per stage into LDS, already in operand order
//     ([row tile][k step][lane][16 B]), so an A fragment is one conflict-free ds_read_b128;
//   * accumulator layout: column (= query) on the lane, 16 target rows in the registers, so best / second are tracked per lane,
//     both column tiles at once on packed 16-bit keys (2.5 VALU ops per pair instead of 21).  The two lane halves (rows 4h..)
//     of a column are merged once at the end, where popcount(q) turns the dot product back into the distance.
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
constexpr int kHmStage = 128;      // targets per LDS stage (four 32-row tiles, 32 KB of +-1 bytes)

__device__ __forceinline__ void best2_push(uint32_t &best, uint32_t &second, uint32_t key) {
    const uint32_t lo = min(best, key), hi = max(best, key);
    second = min(second, hi);
    best = lo;
}

__global__ __launch_bounds__(256) void k_hamming_mfma(HamArgs A) {
    __shared__ __attribute__((aligned(16))) uint32_t s_a[kHmStage * 64];
    const int p = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, h = lane >> 5;
    const int qs = A.pair_q ? A.pair_q[p] : p, ts = A.pair_t ? A.pair_t[p] : p;
    const int nq = A.q_count ? min(A.q_count[qs], A.nq) : A.nq;
    const int nt = A.t_count ? min(A.t_count[ts], A.nt) : A.nt;
    if (blockIdx.x * 256 >= nq) {                      // whole block beyond this pair's queries
        const int qi = blockIdx.x * 256 + tid;
        const uint64_t o = (uint64_t)p * A.q_stride + qi;
        if (qi < A.q_stride) { A.best_idx[o] = -1; A.best_dist[o] = MS_HAMMING_MAX; A.second_dist[o] = MS_HAMMING_MAX; }
        return;
    }
    const uint4 *Q = reinterpret_cast<const uint4 *>(A.q + (uint64_t)qs * A.q_stride * 8);
    const uint4 *T = reinterpret_cast<const uint4 *>(A.t + (uint64_t)ts * A.t_stride * 8);
    v4i_t bq[2][8];
    int popq[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int qi = blockIdx.x * 256 + wave * 64 + n * 32 + col;
        uint4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
        if (qi < nq) { lo = Q[2 * qi]; hi = Q[2 * qi + 1]; }
        const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int s = 0; s < 8; ++s)                   // lane half h holds bits 4h+j (+8, +16, +24) of word s in dword j, as +16 / -16
#pragma unroll
            for (int j = 0; j < 4; ++j) bq[n][s][j] = (int)((((w[s] >> (4 * h + j)) & 0x01010101u) * 0xE0u) ^ 0x10101010u);   // bit 0 -> +16, bit 1 -> -16: the accumulators come out as 16 * dot, ready to take a 4-bit row tag
        popq[n] = 0;
#pragma unroll
        for (int s = 0; s < 8; ++s) popq[n] += __popc(w[s]);
    }
    uint32_t best[2] = {kNone, kNone}, second[2] = {kNone, kNone};
    v16i_t tag;
#pragma unroll
    for (int r = 0; r < 16; ++r) tag[r] = 4096 + r;
    for (int base = 0; base < nt; base += kHmStage) {
        __syncthreads();
        {   // expand this stage's targets: thread -> target tid & 127, words 4*(tid>>7) .. +3 (one 16-byte load)
            const int tau = tid & 127, half = tid >> 7, j = base + tau;
            uint4 v = {0, 0, 0, 0};
            if (j < nt) v = T[2 * j + half];
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const uint32_t w = w4[k] >> (4 * hh);
                    *reinterpret_cast<v4i_t *>(&s_a[((tau >> 5) * 8 + 4 * half + k) * 256 + (hh * 32 + (tau & 31)) * 4]) =
                        v4i_t{(int)(w & 0x01010101u), (int)((w >> 1) & 0x01010101u), (int)((w >> 2) & 0x01010101u), (int)((w >> 3) & 0x01010101u)};
                }
        }
        __syncthreads();
        const int mtiles = min(kHmStage / 32, (nt - base + 31) >> 5);
        for (int m = 0; m < mtiles; ++m) {
            v16i_t acc[2];
            v4i_t a[8];                                               // all eight fragments of the tile are requested before the first MFMA waits
#pragma unroll
            for (int s = 0; s < 8; ++s) a[s] = *reinterpret_cast<const v4i_t *>(&s_a[(m * 8 + s) * 256 + lane * 4]);
            __builtin_amdgcn_sched_barrier(0);                        // keep the reads ahead of the chain (the scheduler would re-serialise them to save registers)
#pragma unroll
            for (int s = 0; s < 8; ++s) {          // the chain starts from the row tags (4096 + register index), so no add follows it
                acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], bq[0][s], s == 0 ? tag : acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], bq[1][s], s == 0 ? tag : acc[1], 0, 0, 0);
            }
            const int row0 = base + m * 32 + 4 * h;                   // this lane's rows: row0 + (reg & 3) + 8 * (reg >> 2)
            if (base + m * 32 + 32 <= nt) {                           // full tile (uniform)
                // both column tiles at once on packed 16-bit lanes: key16 = (dot + 256) * 16 + register index (< 2^14); the
                // register index orders a lane's rows, so the packed minimum is the lowest row among equal distances
                us2_t lb = {0xFFFF, 0xFFFF}, ls = {0xFFFF, 0xFFFF};
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t both = __builtin_amdgcn_perm((uint32_t)acc[1][r], (uint32_t)acc[0][r], 0x05040100u);   // low halves: (acc0, acc1)
                    const us2_t key = __builtin_bit_cast(us2_t, both);
                    const us2_t lo = __builtin_elementwise_min(lb, key), hi = __builtin_elementwise_max(lb, key);
                    ls = __builtin_elementwise_min(ls, hi);
                    lb = lo;
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const uint32_t kb = n ? lb.y : lb.x, ks = n ? ls.y : ls.x;      // (dot + 256) * 16 + r
                    const uint32_t rb = kb & 15u, rs = ks & 15u;
                    best2_push(best[n], second[n], ((kb >> 4) << 20) + (uint32_t)row0 + (rb & 3u) + 8u * (rb >> 2));
                    second[n] = min(second[n], ((ks >> 4) << 20) + (uint32_t)row0 + (rs & 3u) + 8u * (rs >> 2));
                }
            } else {                                                  // last tile of the set: rows beyond nt do not exist
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = row0 + (r & 3) + 8 * (r >> 2);
                        const uint32_t key = (((uint32_t)acc[n][r] - (uint32_t)r) << 16) + (uint32_t)row;  // (dot + 256) << 20
                        best2_push(best[n], second[n], row < nt ? key : kNone);
                    }
            }
        }
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {                                     // the other half of the wave holds the other rows of this column
        const uint32_t ob = __shfl_xor(best[n], 32, 64), os = __shfl_xor(second[n], 32, 64);
        const uint32_t b = min(best[n], ob), s2 = min(max(best[n], ob), min(second[n], os));
        const int qi = blockIdx.x * 256 + wave * 64 + n * 32 + col;
        if (h == 0 && qi < A.q_stride) {
            const uint64_t o = (uint64_t)p * A.q_stride + qi;
            const bool live = qi < nq;
            A.best_idx[o] = (!live || b == kNone) ? -1 : (int32_t)(b & 0xFFFFFu);
            A.best_dist[o] = (!live || b == kNone) ? (uint16_t)MS_HAMMING_MAX : (uint16_t)((int)(b >> 20) - 256 + popq[n]);
            A.second_dist[o] = (!live || s2 == kNone) ? (uint16_t)MS_HAMMING_MAX : (uint16_t)((int)(s2 >> 20) - 256 + popq[n]);
        }
    }
}

__global__ __launch_bounds__(256) void k_ratio_test(const int32_t *__restrict__ bi, const uint16_t *__restrict__ bd, const uint16_t *__restrict__ sd,
                                                    int n, float ratio, int max_dist, int32_t *__restrict__ match) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned b = bd[i], s = sd[i];
    bool ok = bi[i] >= 0 && b <= (unsigned)max_dist;                 // keyframe_matcher.cpp:115
    if (ok && __fmul_rn(ratio, (float)s) < (float)b) ok = false;     // keyframe_matcher.cpp:120
    match[i] = ok ? bi[i] : -1;
}

// ------------------------------------------------------------------------------------------------
// Representative descriptor of a map point (MapPoint::updateDescriptor, map_point.cpp:75-116): one workgroup per map
// point.  The observations' descriptors are staged in LDS (9-dword pitch), the n x n distance matrix is built in LDS
// (odd halfword pitch), then each row finds its median by bisection on the value range (9 counting passes) and the
// workgroup takes the (median, index) minimum -- the first index wins ties, exactly as the reference's ascending scan.
__global__ __launch_bounds__(256) void k_descriptor_medoid(const uint32_t *__restrict__ pool, const int32_t *__restrict__ start,
                                                           const int32_t *__restrict__ idx, int cap,
                                                           int32_t *__restrict__ best_local, int32_t *__restrict__ best_pool) {
    extern __shared__ uint32_t s_dyn[];
    __shared__ uint32_t s_best;
    const int p = blockIdx.x, tid = threadIdx.x;
    const int s0 = start[p], n = start[p + 1] - s0;
    if (n <= 0 || n > cap) {                                              // a list longer than the LDS was sized for: -2
        const int code = n <= 0 ? -1 : -2;
        if (tid == 0) { if (best_local) best_local[p] = code; if (best_pool) best_pool[p] = code; }
        return;
    }
    uint32_t *s_desc = s_dyn;                                              // [n][9]
    const int pitch = n | 1;
    uint16_t *s_mat = reinterpret_cast<uint16_t *>(s_dyn + 9 * n);         // [n][pitch]
    for (int e = tid; e < 8 * n; e += 256) s_desc[9 * (e >> 3) + (e & 7)] = pool[8 * (size_t)idx[s0 + (e >> 3)] + (e & 7)];
    if (tid == 0) s_best = 0xffffffffu;
    __syncthreads();
    for (int e = tid; e < n * n; e += 256) {
        const int j = e / n, i = e - j * n;                                // i varies across lanes, j is (nearly) uniform
        unsigned d = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) d += __popc(s_desc[9 * i + k] ^ s_desc[9 * j + k]);
        s_mat[i * pitch + j] = (uint16_t)d;
    }
    __syncthreads();
    const int kth = (n - 1) >> 1;                                          // (unsigned)(0.5 * (n - 1)), map_point.cpp:106
    for (int i = tid; i < n; i += 256) {
        int lo = 0, hi = 256;
        while (lo < hi) {                                                  // smallest v with #(row <= v) > kth
            const int mid = (lo + hi) >> 1;
            int cnt = 0;
            for (int j = 0; j < n; ++j) cnt += s_mat[i * pitch + j] <= mid;
            if (cnt > kth) hi = mid; else lo = mid + 1;
        }
        atomicMin(&s_best, ((uint32_t)lo << 16) | (uint32_t)i);
    }
    __syncthreads();
    if (tid == 0) {
        const int b = (s_best >> 16) < 256u ? (int)(s_best & 0xffffu) : 0;  // accepted only below MAX_HAMMING_DIST (:109)
        if (best_local) best_local[p] = b;
        if (best_pool) best_pool[p] = idx[s0 + b];
    }
}

__device__ __forceinline__ uint32_t rl_u(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ float rl_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double rl_d(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

constexpr int kGreedyCpl = 4;            // candidates per lane in k_greedy_nodes: nodes of up to 256 candidates stay in one wave's registers

// Wave-wide reductions on the DPP network (row shifts inside the 16-lane rows, then the two row broadcasts of gfx9): 6 steps of
// 1-2 register moves instead of 6 x ds_bpermute round trips through the LDS crossbar.  The result is read from lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_none(uint32_t v) {               // lanes the control gives no source keep kNone (the identity of min)
    return (uint32_t)__builtin_amdgcn_update_dpp((int)kNone, (int)v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void best2_step(uint32_t &best, uint32_t &second) {
    const uint32_t ob = dpp_or_none<CTRL, ROW_MASK>(best), os = dpp_or_none<CTRL, ROW_MASK>(second);
    const uint32_t l2 = min(best, ob), h2 = max(best, ob);
    second = min(min(second, os), h2);
    best = l2;
}
__device__ __forceinline__ void wave_best2(uint32_t &best, uint32_t &second) {      // every source lane enters each lane's result once: no key is counted twice
    best2_step<0x111, 0xF>(best, second);          // row_shr:1
    best2_step<0x112, 0xF>(best, second);          // row_shr:2
    best2_step<0x114, 0xF>(best, second);          // row_shr:4
    best2_step<0x118, 0xF>(best, second);          // row_shr:8   -> lane 15 of a row holds the row
    best2_step<0x142, 0xA>(best, second);          // row_bcast:15 into rows 1 and 3
    best2_step<0x143, 0xC>(best, second);          // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave
    best = rl_u(best, 63); second = rl_u(second, 63);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, dpp_or_none<0x111, 0xF>(v)); v = min(v, dpp_or_none<0x112, 0xF>(v)); v = min(v, dpp_or_none<0x114, 0xF>(v));
    v = min(v, dpp_or_none<0x118, 0xF>(v)); v = min(v, dpp_or_none<0x142, 0xA>(v)); v = min(v, dpp_or_none<0x143, 0xC>(v));
    return rl_u(v, 63);
}

// Top-4 candidate lists (ms_hamming_candidates_topk / ms_projection_topk): a lane keeps the four smallest keys it has seen, sorted; the wave then
// pops its minimum four times (keys are unique -- they carry the candidate's position -- so exactly one lane owns each minimum).
struct Top4 { uint32_t k0 = kNone, k1 = kNone, k2 = kNone, k3 = kNone; };
__device__ __forceinline__ void top4_push(Top4 &t, uint32_t x) {
    uint32_t a = min(t.k0, x); x = max(t.k0, x); t.k0 = a;
    a = min(t.k1, x); x = max(t.k1, x); t.k1 = a;
    a = min(t.k2, x); x = max(t.k2, x); t.k2 = a;
    t.k3 = min(t.k3, x);
}
__device__ __forceinline__ void top4_extract(Top4 &t, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t m = wave_min_u32(t.k0);
        out[r] = m;
        if (t.k0 == m && m != kNone) { t.k0 = t.k1; t.k1 = t.k2; t.k2 = t.k3; t.k3 = kNone; }
    }
}

// ------------------------------------------------------------------------------------------------
// One wavefront per query: lanes scan the query's own candidate list, butterfly merge of (best, second) keys.
// TOPK: instead of (best, second) the four best candidates -- bi / bo are then [nq][4] (keypoint index, octave), bd is [nq][4] (distance; unused entries
// -1 / 256), si [nq] the number of candidates that were scored (not skipped): a list with si <= 4 is the complete candidate set
template <bool TOPK>
__global__ __launch_bounds__(256) void k_hamming_candidates(const uint32_t *__restrict__ qd, int nq, const uint32_t *__restrict__ td,
                                                            const int32_t *__restrict__ cstart, const int32_t *__restrict__ cidx,
                                                            const uint8_t *__restrict__ skip, const int32_t *__restrict__ toct,
                                                            int32_t *__restrict__ bi, uint16_t *__restrict__ bd, uint16_t *__restrict__ sd,
                                                            int32_t *__restrict__ bo, int32_t *__restrict__ so, int32_t *__restrict__ si) {
    const int i = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= nq) return;
    const uint4 qa = reinterpret_cast<const uint4 *>(qd)[2 * i], qb = reinterpret_cast<const uint4 *>(qd)[2 * i + 1];
    const uint32_t qr[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
    const int s = cstart[i], e = cstart[i + 1];
    uint32_t best = kNone, second = kNone;
    Top4 top;
    int scored = 0;
    for (int r = s + lane; r < e; r += 64) {
        const int j = cidx[r];
        if (skip && skip[j]) continue;
        const uint4 ta = reinterpret_cast<const uint4 *>(td)[2 * j], tb = reinterpret_cast<const uint4 *>(td)[2 * j + 1];
        const uint32_t key = (hamming8(qr, ta, tb) << 20) | (uint32_t)(r - s);
        if (TOPK) { top4_push(top, key); ++scored; continue; }
        const uint32_t lo = min(best, key), hi = max(best, key);
        second = min(second, hi);
        best = lo;
    }
    if (TOPK) {
        uint32_t out[4];
        top4_extract(top, out);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) scored += __shfl_xor(scored, off, 64);
        if (lane < 4) {
            uint32_t k = out[0];
#pragma unroll
            for (int r = 1; r < 4; ++r) if (lane == r) k = out[r];
            const int j = k == kNone ? -1 : cidx[s + (int)(k & 0xFFFFFu)];
            bi[4 * (size_t)i + lane] = j;
            bd[4 * (size_t)i + lane] = k == kNone ? (uint16_t)MS_HAMMING_MAX : (uint16_t)(k >> 20);
            bo[4 * (size_t)i + lane] = (j >= 0 && toct) ? toct[j] : -1;
        }
        if (lane == 0) si[i] = scored;
        return;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t ob = __shfl_xor(best, off, 64), os = __shfl_xor(second, off, 64);
        const uint32_t lo = min(best, ob), hi = max(best, ob);
        second = min(min(second, os), hi);
        best = lo;
    }
    if (lane == 0) {
        const int jb = best == kNone ? -1 : cidx[s + (int)(best & 0xFFFFFu)], js = second == kNone ? -1 : cidx[s + (int)(second & 0xFFFFFu)];
        bi[i] = jb;
        bd[i] = best == kNone ? (uint16_t)MS_HAMMING_MAX : (uint16_t)(best >> 20);
        sd[i] = second == kNone ? (uint16_t)MS_HAMMING_MAX : (uint16_t)(second >> 20);
        if (bo) bo[i] = (jb >= 0 && toct) ? toct[jb] : -1;
        if (so) so[i] = (js >= 0 && toct) ? toct[js] : -1;
        if (si) si[i] = js;
    }
}

// ------------------------------------------------------------------------------------------------
// Radius query + scoring in one pass (FeatureSearch::getFeaturesAround, feature_search.cpp:33-48, feeding the scans of
// searchByProjection / replaceDuplication / findMatchesTranformedMps): one wavefront per query.  The keyframe's keypoints are
// sorted by y; the wave binary-searches the first y >= qy - r (std::lower_bound), then its lanes walk the range up to
// y <= qy + r, keep the points with dx*dx + dy*dy < r*r (float32, no contraction), and score the survivors exactly like
// k_hamming_candidates: candidates are ordered by their position in the sorted array, best = first minimum, second = next.
template <bool TOPK>       // as in k_hamming_candidates: bi / bd / bo become [nq][4] lists, si the number of scored candidates; ncand keeps its meaning
__global__ __launch_bounds__(256) void k_projection_candidates(const float *__restrict__ sx, const float *__restrict__ sy, const int32_t *__restrict__ sidx, int n,
                                                               const uint32_t *__restrict__ td, const int32_t *__restrict__ toct, const uint8_t *__restrict__ skip,
                                                               const float *__restrict__ qx, const float *__restrict__ qy, const float *__restrict__ qr_,
                                                               const int32_t *__restrict__ qlmin, const int32_t *__restrict__ qlmax,
                                                               const uint32_t *__restrict__ qd, int nq,
                                                               int32_t *__restrict__ bi, uint16_t *__restrict__ bd, uint16_t *__restrict__ sd,
                                                               int32_t *__restrict__ bo, int32_t *__restrict__ so, int32_t *__restrict__ si, int32_t *__restrict__ ncand) {
    const int i = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= nq) return;
    const uint4 qa = reinterpret_cast<const uint4 *>(qd)[2 * i], qb = reinterpret_cast<const uint4 *>(qd)[2 * i + 1];
    const uint32_t qr[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
    const float x = qx[i], y = qy[i], r = qr_[i];
    const float ylo = __fsub_rn(y, r), yhi = __fadd_rn(y, r), r2 = __fmul_rn(r, r);
    const int lmin = qlmin ? qlmin[i] : -0x7fffffff, lmax = qlmax ? qlmax[i] : 0x7fffffff;
    int lo = 0, hi = n;
    while (lo < hi) {                                   // first position whose y is not < qy - r
        const int mid = (lo + hi) >> 1;
        if (sy[mid] < ylo) lo = mid + 1; else hi = mid;
    }
    uint32_t best = kNone, second = kNone;
    Top4 top;
    int count = 0, scored = 0;
    for (int pos = lo + lane;; pos += 64) {
        const bool in_y = pos < n && sy[pos] <= yhi;
        if (__ballot(in_y) == 0) break;                 // sorted: nothing further can be inside
        if (!in_y) continue;
        const float dx = __fsub_rn(x, sx[pos]), dy = __fsub_rn(y, sy[pos]);
        if (!(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) < r2)) continue;
        const int j = sidx[pos];
        ++count;                                        // size of the reference's output vector
        if (skip && skip[j]) continue;
        if (toct && (toct[j] < lmin || toct[j] > lmax)) continue;
        const uint4 ta = reinterpret_cast<const uint4 *>(td)[2 * j], tb = reinterpret_cast<const uint4 *>(td)[2 * j + 1];
        const uint32_t key = (hamming8(qr, ta, tb) << 20) | (uint32_t)pos;
        if (TOPK) { top4_push(top, key); ++scored; continue; }
        const uint32_t l2 = min(best, key), h2 = max(best, key);
        second = min(second, h2);
        best = l2;
    }
    if (TOPK) {
        uint32_t out[4];
        top4_extract(top, out);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { scored += __shfl_xor(scored, off, 64); count += __shfl_xor(count, off, 64); }
        if (lane < 4) {
            uint32_t k = out[0];
#pragma unroll
            for (int r = 1; r < 4; ++r) if (lane == r) k = out[r];
            const int j = k == kNone ? -1 : sidx[k & 0xFFFFFu];
            bi[4 * (size_t)i + lane] = j;
            bd[4 * (size_t)i + lane] = k == kNone ? (uint16_t)MS_HAMMING_MAX : (uint16_t)(k >> 20);
            bo[4 * (size_t)i + lane] = (j >= 0 && toct) ? toct[j] : -1;
        }
        if (lane == 0) { si[i] = scored; if (ncand) ncand[i] = count; }
        return;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t ob = __shfl_xor(best, off, 64), os = __shfl_xor(second, off, 64);
        const uint32_t l2 = min(best, ob), h2 = max(best, ob);
        second = min(min(second, os), h2);
        best = l2;
        count += __shfl_xor(count, off, 64);
    }
    if (lane == 0) {
        const int jb = best == kNone ? -1 : sidx[best & 0xFFFFFu], js = second == kNone ? -1 : sidx[second & 0xFFFFFu];
        bi[i] = jb;
        bd[i] = best == kNone ? (uint16_t)MS_HAMMING_MAX : (uint16_t)(best >> 20);
        sd[i] = second == kNone ? (uint16_t)MS_HAMMING_MAX : (uint16_t)(second >> 20);
        if (bo) bo[i] = (jb >= 0 && toct) ? toct[jb] : -1;
        if (so) so[i] = (js >= 0 && toct) ? toct[js] : -1;
        if (si) si[i] = js;
        if (ncand) ncand[i] = count;
    }
}

// ------------------------------------------------------------------------------------------------
// Greedy BoW matchers: one wavefront per keyframe pair.  The walk over shared vocabulary nodes and
// over kf1's keypoints is sequential (targets are consumed as they are matched, so query i depends on
// the queries before it: keyframe_matcher.cpp:98-100,:128 / :224-226,:249); the 64 lanes scan the
// node's kf2 candidates in parallel and a butterfly reduction rebuilds exactly what the sequential
// scan would have kept (M1: first minimum + second minimum; M2: LAST minimum among epipolar inliers).
struct GreedyArgs {
    const ms_match_frame *f1, *f2;
    int32_t *const *matched;
    int32_t *n_matches;
    const double *E12;            // M2
    const float *scale_factors;   // M2
    float lowe_ratio, residual_deg_thr;
    int check_orientation;
    const int32_t *redo;          // sequential kernel only: when set, pair p runs only if redo[32 p + 31] != 0 (the node-parallel pass gave it up)
};

__device__ __forceinline__ int angle_bin(float a1, float a2) {   // match_angle_checker.h:72-83
    float d = __fsub_rn(a1, a2);
    if (d < 0.0) d = (float)__dadd_rn((double)d, 360.0);
    if (360.0 <= d) d = (float)__dsub_rn((double)d, 360.0);
    const float inv_len = 1.0f / 30;
    int b = __float2int_rn(__fmul_rn(d, inv_len));
    return (b < 0 || b >= 30) ? 29 : b;
}

__device__ __forceinline__ bool epipolar_ok(const double *b1, const double *b2, const double *E, float scale1, float thr_deg) {
    // keyframe_matcher.cpp:23-44, evaluated in the same operation order (no contraction)
    double n[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        n[i] = __dadd_rn(__dadd_rn(__dmul_rn(E[3 * i], b2[0]), __dmul_rn(E[3 * i + 1], b2[1])), __dmul_rn(E[3 * i + 2], b2[2]));
    const double nn = __dsqrt_rn(__dadd_rn(__dadd_rn(__dmul_rn(n[0], n[0]), __dmul_rn(n[1], n[1])), __dmul_rn(n[2], n[2])));
    const double dot = __dadd_rn(__dadd_rn(__dmul_rn(n[0], b1[0]), __dmul_rn(n[1], b1[1])), __dmul_rn(n[2], b1[2]));
    const double cr = __ddiv_rn(dot, nn);
    const double res = __dsub_rn(M_PI / 2.0, fabs(acos(cr)));
    const double thr = __ddiv_rn(__dmul_rn((double)thr_deg, M_PI), 180.0);
    return res < __dmul_rn(thr, (double)scale1);
}

template <bool TRIANGULATION>
__global__ __launch_bounds__(64) void k_match_greedy(GreedyArgs A) {
    __shared__ uint32_t s_used[1024];          // bitset over kf2 keypoints (n2 <= 32768)
    const int p = blockIdx.x, lane = threadIdx.x;
    if (A.redo && A.redo[32 * p + 31] == 0) return;
    const ms_match_frame F1 = A.f1[p], F2 = A.f2[p];
    int32_t *matched = A.matched[p];
    for (int i = lane; i < F1.n; i += 64) matched[i] = -1;
    for (int i = lane; i < 1024; i += 64) s_used[i] = 0;
    __syncthreads();
    const double *E = TRIANGULATION ? A.E12 + 9 * (uint64_t)p : nullptr;
    int hist[30];
#pragma unroll
    for (int b = 0; b < 30; ++b) hist[b] = 0;
    int num = 0, a = 0, b = 0;
    while (a < F1.bow.n_nodes && b < F2.bow.n_nodes) {                  // ordered-map merge
        const int ida = F1.bow.node_id[a], idb = F2.bow.node_id[b];
        if (ida < idb) { ++a; continue; }
        if (idb < ida) { ++b; continue; }
        const int s2 = F2.bow.node_start[b], e2 = F2.bow.node_start[b + 1];
        for (int pp = F1.bow.node_start[a]; pp < F1.bow.node_start[a + 1]; ++pp) {
            const int i1 = F1.bow.kp_idx[pp];
            if (!F1.usable[i1]) continue;
            const uint4 qa = reinterpret_cast<const uint4 *>(F1.desc)[2 * i1], qb = reinterpret_cast<const uint4 *>(F1.desc)[2 * i1 + 1];
            const uint32_t qr[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
            uint32_t best = kNone, second = kNone;
            for (int r = s2 + lane; r < e2; r += 64) {
                const int i2 = F2.bow.kp_idx[r];
                if (!F2.usable[i2]) continue;
                if (s_used[i2 >> 5] & (1u << (i2 & 31))) continue;
                const uint4 ta = reinterpret_cast<const uint4 *>(F2.desc)[2 * i2], tb = reinterpret_cast<const uint4 *>(F2.desc)[2 * i2 + 1];
                const uint32_t d = hamming8(qr, ta, tb);
                const uint32_t pos = (uint32_t)(r - s2);
                uint32_t key;
                if (TRIANGULATION) {
                    if (d > MS_HAMMING_THR_LOW) continue;                                        // :231
                    if (!epipolar_ok(F1.bearing + 3 * (uint64_t)i1, F2.bearing + 3 * (uint64_t)i2, E,
                                     A.scale_factors[F1.octave[i1]], A.residual_deg_thr)) continue;   // :237-239
                    key = (d << 20) | (0xFFFFFu - pos);                                           // ties: LAST wins
                } else {
                    key = (d << 20) | pos;                                                        // ties: FIRST wins
                }
                const uint32_t lo = min(best, key), hi = max(best, key);
                second = min(second, hi);
                best = lo;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t ob = __shfl_xor(best, off, 64), os = __shfl_xor(second, off, 64);
                const uint32_t lo = min(best, ob), hi = max(best, ob);
                second = min(min(second, os), hi);
                best = lo;
            }
            if (best == kNone) continue;
            const uint32_t bd = best >> 20;
            const uint32_t pos = TRIANGULATION ? 0xFFFFFu - (best & 0xFFFFFu) : (best & 0xFFFFFu);
            const int bi = F2.bow.kp_idx[s2 + (int)pos];
            if (!TRIANGULATION) {
                const uint32_t sd = second == kNone ? (uint32_t)MS_HAMMING_MAX : second >> 20;
                if (MS_HAMMING_THR_LOW < bd) continue;                                           // :115
                if (__fmul_rn(A.lowe_ratio, (float)sd) < (float)bd) continue;                    // :120
            }
            if (lane == 0) { matched[i1] = bi; s_used[bi >> 5] |= 1u << (bi & 31); }
            __syncthreads();
            ++num;
            if (A.check_orientation) {
                const int bin = angle_bin(F1.angle[i1], F2.angle[bi]);
#pragma unroll
                for (int k = 0; k < 30; ++k) hist[k] += (k == bin);
            }
        }
        ++a; ++b;
    }
    __syncthreads();
    if (A.check_orientation) {
        // top-3 bins by (size desc, bin asc); everything else is invalid (match_angle_checker.h:108-134)
        int v0 = -1, v1 = -1, v2 = -1;
#pragma unroll
        for (int rep = 0; rep < 3; ++rep) {
            int bb = -1, bc = -1;
#pragma unroll
            for (int k = 0; k < 30; ++k)
                if (k != v0 && k != v1 && hist[k] > bc) { bc = hist[k]; bb = k; }
            if (rep == 0) v0 = bb; else if (rep == 1) v1 = bb; else v2 = bb;
        }
        int removed = 0;
        for (int i = lane; i < F1.n; i += 64) {
            const int m = matched[i];
            if (m >= 0) {
                const int bin = angle_bin(F1.angle[i], F2.angle[m]);
                if (bin != v0 && bin != v1 && bin != v2) { matched[i] = -1; ++removed; }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) removed += __shfl_xor(removed, off, 64);
        num -= removed;
    }
    if (lane == 0) A.n_matches[p] = num;
}

// ------------------------------------------------------------------------------------------------
// The same two matchers, node-parallel.  A DBoW2 FeatureVector names every keypoint in exactly one node
// (TemplatedVocabulary::transform calls addFeature once per feature), so the targets a query can consume are
// the targets of its own node: the greedy order matters only INSIDE a node, and the nodes of a pair can run
// side by side.  One wavefront per (pair, node of kf1):
//   * the node's kf2 candidates live in the lanes' registers (descriptor, bearing, "free" flag; up to four per lane -- nodes with more
//     than 256 candidates go to k_greedy_big_nodes below),
//   * the node's kf1 queries are fetched 64 at a time, one per lane, and handed to the wave one after the other
//     with v_readlane -- the walk has no dependent memory access per query any more (the one-wave-per-pair
//     kernel above spent ~1 us per query in three of them),
//   * a node adds its number of matches to the pair's counter (integer atomic: order-free, deterministic).
// k_greedy_finish then builds the pair's rotation histogram from the matches and applies it.  Inputs that break the FeatureVector property (a keypoint
// listed in two shared nodes) are detected with per-keypoint counters and the pair is redone by the sequential
// kernel, so the result is the reference's for ANY input.
struct GreedyScratch {
    int32_t *own1, *own2;           // [n_pairs][stride]: node lists naming the keypoint (shared nodes only)
    int32_t stride1, stride2;
    int32_t *hist;                  // [n_pairs][32]: [30] matches, [31] redo flag (the rest unused)
    int32_t *big;                   // [0] nodes with more than 64 candidates found by k_greedy_nodes, [1] cursor of k_greedy_big_nodes, then int4 {pair, node of kf1, node of kf2, -} each
};

__global__ __launch_bounds__(256) void k_greedy_init(GreedyArgs A, GreedyScratch S) {
    const int p = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int n1 = A.f1[p].n, n2 = A.f2[p].n;
    if (i < n1) { A.matched[p][i] = -1; S.own1[(size_t)p * S.stride1 + i] = 0; }
    if (i < n2) S.own2[(size_t)p * S.stride2 + i] = 0;
    if (i < 32) S.hist[32 * p + i] = 0;
    if (p == 0 && i < 2) S.big[i] = 0;
}

template <bool TRIANGULATION>
__global__ __launch_bounds__(64) void k_greedy_nodes(GreedyArgs A, GreedyScratch S) {
    const int p = blockIdx.y, a = blockIdx.x, lane = threadIdx.x;
    const ms_match_frame F1 = A.f1[p], F2 = A.f2[p];
    if (a >= F1.bow.n_nodes) return;
    const int ida = F1.bow.node_id[a];
    int lo = 0, hi = F2.bow.n_nodes;
    while (lo < hi) {                            // the ordered-map merge of keyframe_matcher.cpp:70-147, per node
        const int mid = (lo + hi) >> 1;
        if (F2.bow.node_id[mid] < ida) lo = mid + 1; else hi = mid;
    }
    if (lo >= F2.bow.n_nodes || F2.bow.node_id[lo] != ida) return;
    const int s1 = F1.bow.node_start[a], e1 = F1.bow.node_start[a + 1];
    const int s2 = F2.bow.node_start[lo], n2 = F2.bow.node_start[lo + 1] - s2;
    if (n2 <= 0 || e1 <= s1) return;
    if (n2 > 64 * kGreedyCpl) {                  // more candidates than the lanes' registers hold: onto the work list of k_greedy_big_nodes
        if (lane == 0) { const int w = atomicAdd(&S.big[0], 1); reinterpret_cast<int4 *>(S.big + 4)[w] = make_int4(p, a, lo, 0); }
        return;
    }
    int32_t *own1 = S.own1 + (size_t)p * S.stride1, *own2 = S.own2 + (size_t)p * S.stride2, *hist = S.hist + 32 * p;
    int32_t *matched = A.matched[p];
    const uint4 *D1 = reinterpret_cast<const uint4 *>(F1.desc), *D2 = reinterpret_cast<const uint4 *>(F2.desc);
    bool bad = false;                            // a keypoint named twice, or an index outside the keyframe
    // the node's candidates: position 64 c + lane in slot c of this lane, for the whole walk; bit c of `free` = usable and not consumed yet
    int i2v[kGreedyCpl];
    uint4 ta[kGreedyCpl], tb[kGreedyCpl];
    double b2v[kGreedyCpl][3];
    uint32_t free = 0;
#pragma unroll
    for (int c = 0; c < kGreedyCpl; ++c) {
        const int r = 64 * c + lane;
        i2v[c] = -1; ta[c] = make_uint4(0, 0, 0, 0); tb[c] = ta[c]; b2v[c][0] = b2v[c][1] = b2v[c][2] = 0;
        if (64 * c >= n2) continue;              // uniform: small nodes pay for one slot
        if (r < n2) {
            i2v[c] = F2.bow.kp_idx[s2 + r];
            if ((unsigned)i2v[c] >= (unsigned)F2.n || atomicAdd(&own2[i2v[c]], 1) != 0) bad = true;
            else if (F2.usable[i2v[c]]) {
                free |= 1u << c;
                ta[c] = D2[2 * i2v[c]]; tb[c] = D2[2 * i2v[c] + 1];
                if (TRIANGULATION) { b2v[c][0] = F2.bearing[3 * (size_t)i2v[c]]; b2v[c][1] = F2.bearing[3 * (size_t)i2v[c] + 1]; b2v[c][2] = F2.bearing[3 * (size_t)i2v[c] + 2]; }
            }
        }
    }
    double E[9];
    if (TRIANGULATION) {
#pragma unroll
        for (int k = 0; k < 9; ++k) E[k] = A.E12[9 * (size_t)p + k];
    }
    int num = 0;
    for (int q0 = s1; q0 < e1 && !__any(bad); q0 += 64) {
        const int i1v = q0 + lane < e1 ? F1.bow.kp_idx[q0 + lane] : -1;
        bool q_ok = false;
        if (q0 + lane < e1) {
            if ((unsigned)i1v >= (unsigned)F1.n || atomicAdd(&own1[i1v], 1) != 0) bad = true;
            else q_ok = F1.usable[i1v] != 0;
        }
        uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
        float sc1v = 0.f;
        double b1v[3] = {0, 0, 0};
        if (q_ok) {
            qa = D1[2 * i1v]; qb = D1[2 * i1v + 1];
            if (TRIANGULATION) {
                b1v[0] = F1.bearing[3 * (size_t)i1v]; b1v[1] = F1.bearing[3 * (size_t)i1v + 1]; b1v[2] = F1.bearing[3 * (size_t)i1v + 2];
                sc1v = A.scale_factors[F1.octave[i1v]];
            }
        }
        if (__any(bad)) break;
        uint64_t qmask = __ballot(q_ok);
        while (qmask) {                                                       // the node's queries in list order (:76 / :200)
            const int k = __builtin_ctzll(qmask);
            qmask &= qmask - 1;
            const uint32_t qr[8] = {rl_u(qa.x, k), rl_u(qa.y, k), rl_u(qa.z, k), rl_u(qa.w, k), rl_u(qb.x, k), rl_u(qb.y, k), rl_u(qb.z, k), rl_u(qb.w, k)};
            double b1[3] = {0, 0, 0};
            float sc1 = 0.f;
            if (TRIANGULATION) { b1[0] = rl_d(b1v[0], k); b1[1] = rl_d(b1v[1], k); b1[2] = rl_d(b1v[2], k); sc1 = rl_f(sc1v, k); }
            uint32_t best = kNone, second = kNone;
#pragma unroll
            for (int c = 0; c < kGreedyCpl; ++c) {
                if (64 * c >= n2) continue;
                if (!((free >> c) & 1u)) continue;
                const uint32_t d = hamming8(qr, ta[c], tb[c]), pos = (uint32_t)(64 * c + lane);
                if (TRIANGULATION) {
                    if (d > MS_HAMMING_THR_LOW) continue;                                         // :231
                    if (!epipolar_ok(b1, b2v[c], E, sc1, A.residual_deg_thr)) continue;           // :237-239
                    best = min(best, (d << 20) | (0xFFFFFu - pos));                               // ties: LAST wins
                } else {
                    best2_push(best, second, (d << 20) | pos);                                    // ties: FIRST wins
                }
            }
            if (TRIANGULATION) best = wave_min_u32(best); else wave_best2(best, second);
            if (best == kNone) continue;
            const uint32_t bd = best >> 20;
            const uint32_t pos = TRIANGULATION ? 0xFFFFFu - (best & 0xFFFFFu) : (best & 0xFFFFFu);
            if (!TRIANGULATION) {
                const uint32_t sd = second == kNone ? (uint32_t)MS_HAMMING_MAX : second >> 20;
                if (MS_HAMMING_THR_LOW < bd) continue;                                            // :115
                if (__fmul_rn(A.lowe_ratio, (float)sd) < (float)bd) continue;                     // :120
            }
            const int i1 = (int)rl_u((uint32_t)i1v, k);
            const int pc = (int)(pos >> 6), pl = (int)(pos & 63u);
            int bi = 0;
#pragma unroll
            for (int c = 0; c < kGreedyCpl; ++c) if (c == pc) bi = (int)rl_u((uint32_t)i2v[c], pl);
            if (lane == 0) matched[i1] = bi;                                                      // :126 / :250
            if (lane == pl) free &= ~(1u << pc);                                                  // :128 / :249
            ++num;
        }
    }
    if (__any(bad)) { if (lane == 0) hist[31] = 1; return; }
    if (lane == 0 && num) atomicAdd(&hist[30], num);
}

// Nodes with more than 256 candidates (coarse vocabularies; one node = plain brute force): a 256-thread workgroup per node.  The candidates'
// descriptors and keypoint indices are staged in LDS once (the first kBigCap of them; a longer list is read from memory behind that), every
// thread owns the candidates at positions tid, tid + 256, ... and keeps their "consumed" flags in registers (a candidate without the usable flag
// starts consumed).  Queries are staged 64 at a time; for each one the four waves score their candidates, reduce (best, second) per wave, meet
// once (one barrier per query, the partial results double-buffered) and every thread merges the four partials to the same decision.
constexpr int kBigCap = 2048;
template <bool TRIANGULATION>
__global__ __launch_bounds__(256) void k_greedy_big_nodes(GreedyArgs A, GreedyScratch S) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_big[];
    uint4 *s_cd = reinterpret_cast<uint4 *>(s_big);                            // [kBigCap][2] candidate descriptors
    int32_t *s_ci = reinterpret_cast<int32_t *>(s_cd + 2 * kBigCap);           // [kBigCap] keypoint index of the candidate
    uint4 *s_q = reinterpret_cast<uint4 *>(s_ci + kBigCap);                    // [64][2] query descriptors of the current chunk
    double *s_qb = reinterpret_cast<double *>(s_q + 128);                      // [64][3] query bearings (M2)
    int32_t *s_qi = reinterpret_cast<int32_t *>(s_qb + 192);                   // [64] keypoint index
    float *s_qs = reinterpret_cast<float *>(s_qi + 64);                        // [64] scale factor of the query's octave (M2)
    uint32_t *s_part = reinterpret_cast<uint32_t *>(s_qs + 64);                // [2][4][2] per-wave (best, second), double-buffered
    __shared__ unsigned long long s_qmask;
    __shared__ int s_bad, s_work;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (;;) {                                                                   // work list: every workgroup leaves once the cursor has passed the count
    __syncthreads();
    if (tid == 0) s_work = atomicAdd(&S.big[1], 1);
    __syncthreads();
    if (s_work >= S.big[0]) return;
    const int4 item = reinterpret_cast<const int4 *>(S.big + 4)[s_work];
    const int p = item.x, a = item.y, lo = item.z;
    const ms_match_frame F1 = A.f1[p], F2 = A.f2[p];
    const int s1 = F1.bow.node_start[a], e1 = F1.bow.node_start[a + 1];
    const int s2 = F2.bow.node_start[lo], n2 = F2.bow.node_start[lo + 1] - s2;
    int32_t *own1 = S.own1 + (size_t)p * S.stride1, *own2 = S.own2 + (size_t)p * S.stride2, *hist = S.hist + 32 * p;
    int32_t *matched = A.matched[p];
    const uint4 *D1 = reinterpret_cast<const uint4 *>(F1.desc), *D2 = reinterpret_cast<const uint4 *>(F2.desc);
    if (tid == 0) s_bad = n2 > 32768 ? 1 : 0;
    __syncthreads();
    // stage the candidates; this thread's consumed flags: bit j of used[j >> 6] <-> position tid + 256 j
    unsigned long long used0 = 0, used1 = 0;                                   // (two scalars: an indexed pair would live in scratch memory)
    for (int r = tid, j = 0; r < min(n2, 32768); r += 256, ++j) {
        const int i2 = F2.bow.kp_idx[s2 + r];
        bool ok = true;
        if ((unsigned)i2 >= (unsigned)F2.n || atomicAdd(&own2[i2], 1) != 0) { s_bad = 1; ok = false; }
        else ok = F2.usable[i2] != 0;
        if (!ok) { if (j < 64) used0 |= 1ull << j; else used1 |= 1ull << (j - 64); }
        if (r < kBigCap) {
            s_ci[r] = i2;
            if (ok) { s_cd[2 * r] = D2[2 * i2]; s_cd[2 * r + 1] = D2[2 * i2 + 1]; }
        }
    }
#pragma unroll
    for (int j = 0; j < kBigCap / 256; ++j) if (tid + 256 * j >= n2) used0 |= 1ull << j;      // slots behind the end of the list are never free
    double E[9];
    if (TRIANGULATION) {
#pragma unroll
        for (int k = 0; k < 9; ++k) E[k] = A.E12[9 * (size_t)p + k];
    }
    __syncthreads();
    int num = 0, it = 0;
    for (int q0 = s1; q0 < e1 && !s_bad; q0 += 64) {
        __syncthreads();                                                       // the previous chunk's queries are no longer read
        if (wave == 0) {                                                       // stage up to 64 queries
            const int i1 = q0 + lane < e1 ? F1.bow.kp_idx[q0 + lane] : -1;
            bool q_ok = false;
            if (q0 + lane < e1) {
                if ((unsigned)i1 >= (unsigned)F1.n || atomicAdd(&own1[i1], 1) != 0) s_bad = 1;
                else q_ok = F1.usable[i1] != 0;
            }
            if (q_ok) {
                s_q[2 * lane] = D1[2 * i1]; s_q[2 * lane + 1] = D1[2 * i1 + 1];
                s_qi[lane] = i1;
                if (TRIANGULATION) {
                    s_qb[3 * lane] = F1.bearing[3 * (size_t)i1]; s_qb[3 * lane + 1] = F1.bearing[3 * (size_t)i1 + 1]; s_qb[3 * lane + 2] = F1.bearing[3 * (size_t)i1 + 2];
                    s_qs[lane] = A.scale_factors[F1.octave[i1]];
                }
            }
            const unsigned long long m = __ballot(q_ok);
            if (lane == 0) s_qmask = m;
        }
        __syncthreads();
        if (s_bad) break;
        unsigned long long qmask = s_qmask;
        while (qmask) {
            const int k = __builtin_ctzll(qmask);
            qmask &= qmask - 1;
            const uint4 qa = s_q[2 * k], qb = s_q[2 * k + 1];
            const uint32_t qr[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
            double b1[3] = {0, 0, 0};
            float sc1 = 0.f;
            if (TRIANGULATION) { b1[0] = s_qb[3 * k]; b1[1] = s_qb[3 * k + 1]; b1[2] = s_qb[3 * k + 2]; sc1 = s_qs[k]; }
            uint32_t best = kNone, second = kNone;
            auto score = [&](int r, int i2, const uint4 &da, const uint4 &db) {
                const uint32_t d = hamming8(qr, da, db);
                if (TRIANGULATION) {
                    if (d > MS_HAMMING_THR_LOW) return;                                             // :231
                    if (i2 < 0) i2 = s_ci[r];
                    const double bb[3] = {F2.bearing[3 * (size_t)i2], F2.bearing[3 * (size_t)i2 + 1], F2.bearing[3 * (size_t)i2 + 2]};
                    if (!epipolar_ok(b1, bb, E, sc1, A.residual_deg_thr)) return;                   // :237-239
                    best = min(best, (d << 20) | (0xFFFFFu - (uint32_t)r));                         // ties: LAST wins
                } else {
                    best2_push(best, second, (d << 20) | (uint32_t)r);                              // ties: FIRST wins
                }
            };
            {   // the staged candidates: all of this thread's LDS reads are issued before the first distance (fixed trip count, free slots only)
                const uint32_t lowfree = ~(uint32_t)used0;
                uint4 da[kBigCap / 256], db[kBigCap / 256];
#pragma unroll
                for (int j = 0; j < kBigCap / 256; ++j)
                    if ((lowfree >> j) & 1u) { da[j] = s_cd[2 * (tid + 256 * j)]; db[j] = s_cd[2 * (tid + 256 * j) + 1]; }
#pragma unroll
                for (int j = 0; j < kBigCap / 256; ++j)
                    if ((lowfree >> j) & 1u) score(tid + 256 * j, -1, da[j], db[j]);
            }
            for (int r = tid + kBigCap, j = kBigCap / 256; r < n2; r += 256, ++j) {                 // a list longer than the LDS holds: from memory
                if (((j < 64 ? used0 : used1) >> (j & 63)) & 1ull) continue;
                const int i2 = F2.bow.kp_idx[s2 + r];
                score(r, i2, D2[2 * i2], D2[2 * i2 + 1]);
            }
            if (TRIANGULATION) best = wave_min_u32(best); else wave_best2(best, second);
            uint32_t *part = s_part + 8 * (it & 1);
            ++it;
            if (lane == 0) { part[2 * wave] = best; part[2 * wave + 1] = second; }
            __syncthreads();
            best = part[0]; second = part[1];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const uint32_t ob = part[2 * w], os = part[2 * w + 1];
                const uint32_t l2 = min(best, ob), h2 = max(best, ob);
                second = min(min(second, os), h2);
                best = l2;
            }
            if (best == kNone) continue;
            const uint32_t bd = best >> 20;
            const uint32_t pos = TRIANGULATION ? 0xFFFFFu - (best & 0xFFFFFu) : (best & 0xFFFFFu);
            if (!TRIANGULATION) {
                const uint32_t sd = second == kNone ? (uint32_t)MS_HAMMING_MAX : second >> 20;
                if (MS_HAMMING_THR_LOW < bd) continue;                                              // :115
                if (__fmul_rn(A.lowe_ratio, (float)sd) < (float)bd) continue;                       // :120
            }
            if ((int)(pos & 255u) == tid) { const int j = (int)(pos >> 8); if (j < 64) used0 |= 1ull << j; else used1 |= 1ull << (j - 64); }   // :128 / :249
            ++num;
            if (tid == 0) matched[s_qi[k]] = pos < (uint32_t)kBigCap ? s_ci[pos] : F2.bow.kp_idx[s2 + (int)pos];    // :126 / :250
        }
    }
    __syncthreads();
    if (tid == 0) { if (s_bad) hist[31] = 1; else if (num) atomicAdd(&hist[30], num); }
  }
}
constexpr size_t kBigLds = sizeof(uint4) * 2 * kBigCap + 4 * kBigCap + sizeof(uint4) * 128 + sizeof(double) * 192 + 4 * 64 * 2 + 4 * 16;     // 76 KB: two workgroups per CU

__global__ __launch_bounds__(256) void k_greedy_finish(GreedyArgs A, GreedyScratch S) {
    // rotation consistency (match_angle_checker.h:60-134) over the pair's matches: the histogram is built here, from the matches the node passes left
    // (the bins do not depend on the order the matches were found in), then everything outside the three fullest bins is un-matched (:149-155 / :272-277)
    __shared__ int s_hist[30], s_removed;
    const int p = blockIdx.x, tid = threadIdx.x;
    const int32_t *cnt = S.hist + 32 * p;
    if (cnt[31]) return;                                                      // the sequential kernel redoes this pair
    const ms_match_frame F1 = A.f1[p], F2 = A.f2[p];
    int32_t *matched = A.matched[p];
    if (tid < 30) s_hist[tid] = 0;
    if (tid == 0) s_removed = 0;
    __syncthreads();
    if (A.check_orientation) {
        int8_t bins[8];                                                       // this thread's first 8 matches keep their bin; beyond (n1 > 2048) it is recomputed
#pragma unroll
        for (int j = 0; j < 8; ++j) bins[j] = -1;
        for (int i = tid, j = 0; i < F1.n; i += 256, ++j) {
            const int m = matched[i];
            if (m < 0) continue;
            const int bin = angle_bin(F1.angle[i], F2.angle[m]);
            atomicAdd(&s_hist[bin], 1);
            if (j < 8) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) if (jj == j) bins[jj] = (int8_t)bin;
            }
        }
        __syncthreads();
        int v0 = -1, v1 = -1, v2 = -1;                                        // top-3 bins by (size desc, bin asc), match_angle_checker.h:108-134
        for (int rep = 0; rep < 3; ++rep) {
            int bb = -1, bc = -1;
            for (int k = 0; k < 30; ++k)
                if (k != v0 && k != v1 && s_hist[k] > bc) { bc = s_hist[k]; bb = k; }
            if (rep == 0) v0 = bb; else if (rep == 1) v1 = bb; else v2 = bb;
        }
        int removed = 0;
        for (int i = tid, j = 0; i < F1.n; i += 256, ++j) {
            int bin = -1;
            if (j < 8) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) if (jj == j) bin = bins[jj];
                if (bin < 0) continue;
            } else {
                const int m = matched[i];
                if (m < 0) continue;
                bin = angle_bin(F1.angle[i], F2.angle[m]);
            }
            if (bin != v0 && bin != v1 && bin != v2) { matched[i] = -1; ++removed; }
        }
        if (removed) atomicAdd(&s_removed, removed);
        __syncthreads();
    }
    if (tid == 0) A.n_matches[p] = cnt[30] - s_removed;
}

}  // namespace

extern "C" {

static int launch_hamming(ms_ctx *c, const HamArgs &A, int n_pairs) {
    MsRange range("match");
    if (A.nq < 0 || A.nt < 0 || n_pairs < 0 || A.nt >= (1 << 20) || n_pairs > 65535 || A.q_stride < A.nq || A.t_stride < A.nt)
        return ms_fail(c, MS_ERR_INVALID, "hamming search: size out of range");
    if (A.q_stride == 0 || n_pairs == 0) return MS_OK;
    if ((A.qb == nullptr) != (A.tb == nullptr)) return ms_fail(c, MS_ERR_INVALID, "hamming search: give both bucket arrays or neither");
    if (reinterpret_cast<uintptr_t>(A.q) % 16 || reinterpret_cast<uintptr_t>(A.t) % 16 || (A.q_stride * 32) % 16)
        return ms_fail(c, MS_ERR_INVALID, "hamming search: descriptors must be 16-byte aligned");
    MS_HIP(c, hipSetDevice(c->device));
    dim3 grid(ms_div_up(A.q_stride, 256), n_pairs);
    if (A.qb || A.tv) hipLaunchKernelGGL(k_hamming_best2<true>, grid, dim3(256), 0, c->stream, A);
    else if (c->hamming_path == 1) hipLaunchKernelGGL(k_hamming_best2<false>, grid, dim3(256), 0, c->stream, A);
    else hipLaunchKernelGGL(k_hamming_mfma, grid, dim3(256), 0, c->stream, A);
    MS_KERNEL_CHECK(c, "k_hamming_best2");
    return MS_OK;
}

int ms_hamming_set_path(ms_ctx *c, int path) {
    if (!c || path < 0 || path > 1) return MS_ERR_INVALID;
    c->hamming_path = path;
    return MS_OK;
}

int ms_hamming_best2(ms_ctx *c, const uint32_t *q, int nq, const uint32_t *t, int nt, int n_pairs,
                     const int32_t *q_bucket, const int32_t *t_bucket, const uint8_t *t_valid,
                     int32_t *best_idx, uint16_t *best_dist, uint16_t *second_dist) {
    if (!c || !q || !t || !best_idx || !best_dist || !second_dist) return MS_ERR_INVALID;
    HamArgs A{};
    A.q = q; A.t = t; A.q_stride = nq; A.t_stride = nt; A.nq = nq; A.nt = nt;
    A.qb = q_bucket; A.tb = t_bucket; A.tv = t_valid;
    A.best_idx = best_idx; A.best_dist = best_dist; A.second_dist = second_dist;
    return launch_hamming(c, A, n_pairs);
}

int ms_hamming_best2_sets(ms_ctx *c, const uint32_t *q_pool, int q_stride, const int32_t *q_count,
                          const uint32_t *t_pool, int t_stride, const int32_t *t_count,
                          const int32_t *pair_q, const int32_t *pair_t, int n_pairs,
                          int32_t *best_idx, uint16_t *best_dist, uint16_t *second_dist) {
    if (!c || !q_pool || !t_pool || !best_idx || !best_dist || !second_dist) return MS_ERR_INVALID;
    HamArgs A{};
    A.q = q_pool; A.t = t_pool; A.q_stride = q_stride; A.t_stride = t_stride; A.nq = q_stride; A.nt = t_stride;
    A.q_count = q_count; A.t_count = t_count; A.pair_q = pair_q; A.pair_t = pair_t;
    A.best_idx = best_idx; A.best_dist = best_dist; A.second_dist = second_dist;
    return launch_hamming(c, A, n_pairs);
}

int ms_hamming_candidates(ms_ctx *c, const uint32_t *q_desc, int nq, const uint32_t *t_desc, const int32_t *cand_start, const int32_t *cand_idx,
                          const uint8_t *t_skip, const int32_t *t_octave, int32_t *best_idx, uint16_t *best_dist, uint16_t *second_dist,
                          int32_t *best_octave, int32_t *second_octave, int32_t *second_idx) {
    if (!c || !q_desc || !t_desc || !cand_start || !cand_idx || !best_idx || !best_dist || !second_dist || nq < 0) return MS_ERR_INVALID;
    if (nq == 0) return MS_OK;
    if (reinterpret_cast<uintptr_t>(q_desc) % 16 || reinterpret_cast<uintptr_t>(t_desc) % 16) return ms_fail(c, MS_ERR_INVALID, "ms_hamming_candidates: descriptors must be 16-byte aligned");
    MS_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_hamming_candidates<false>, dim3(ms_div_up(nq, 4)), dim3(256), 0, c->stream, q_desc, nq, t_desc, cand_start, cand_idx, t_skip, t_octave,
                       best_idx, best_dist, second_dist, best_octave, second_octave, second_idx);
    MS_KERNEL_CHECK(c, "k_hamming_candidates");
    return MS_OK;
}

int ms_hamming_candidates_topk(ms_ctx *c, const uint32_t *q_desc, int nq, const uint32_t *t_desc, const int32_t *cand_start, const int32_t *cand_idx,
                               const uint8_t *t_skip, const int32_t *t_octave, int32_t *top_idx, uint16_t *top_dist, int32_t *top_octave, int32_t *n_scored) {
    if (!c || !q_desc || !t_desc || !cand_start || !cand_idx || !top_idx || !top_dist || !top_octave || !n_scored || nq < 0) return MS_ERR_INVALID;
    if (nq == 0) return MS_OK;
    if (reinterpret_cast<uintptr_t>(q_desc) % 16 || reinterpret_cast<uintptr_t>(t_desc) % 16) return ms_fail(c, MS_ERR_INVALID, "ms_hamming_candidates_topk: descriptors must be 16-byte aligned");
    MsRange range("match");
    MS_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_hamming_candidates<true>, dim3(ms_div_up(nq, 4)), dim3(256), 0, c->stream, q_desc, nq, t_desc, cand_start, cand_idx, t_skip, t_octave,
                       top_idx, top_dist, static_cast<uint16_t *>(nullptr), top_octave, static_cast<int32_t *>(nullptr), n_scored);
    MS_KERNEL_CHECK(c, "k_hamming_candidates<topk>");
    return MS_OK;
}

int ms_feature_search_sort(const float *x, const float *y, int n, float *sorted_x, float *sorted_y, int32_t *sorted_idx) {
    if (n < 0 || (n && (!x || !y || !sorted_x || !sorted_y || !sorted_idx))) return MS_ERR_INVALID;
    std::vector<int32_t> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return y[a] < y[b]; });     // feature_search.cpp:26-29, ties by index
    for (int i = 0; i < n; ++i) { sorted_x[i] = x[order[i]]; sorted_y[i] = y[order[i]]; sorted_idx[i] = order[i]; }
    return MS_OK;
}

int ms_projection_candidates(ms_ctx *c, const float *sorted_x, const float *sorted_y, const int32_t *sorted_idx, int n_kp,
                             const uint32_t *t_desc, const int32_t *t_octave, const uint8_t *t_skip,
                             const float *q_x, const float *q_y, const float *q_radius, const int32_t *q_min_octave, const int32_t *q_max_octave,
                             const uint32_t *q_desc, int nq,
                             int32_t *best_idx, uint16_t *best_dist, uint16_t *second_dist, int32_t *best_octave, int32_t *second_octave,
                             int32_t *second_idx, int32_t *n_candidates) {
    if (!c || n_kp < 0 || nq < 0 || !q_x || !q_y || !q_radius || !q_desc || !best_idx || !best_dist || !second_dist) return MS_ERR_INVALID;
    if (n_kp && (!sorted_x || !sorted_y || !sorted_idx || !t_desc)) return MS_ERR_INVALID;
    if (n_kp >= (1 << 20)) return ms_fail(c, MS_ERR_CAPACITY, "ms_projection_candidates: %d keypoints (max %d)", n_kp, (1 << 20) - 1);
    if (nq == 0) return MS_OK;
    if (reinterpret_cast<uintptr_t>(q_desc) % 16 || reinterpret_cast<uintptr_t>(t_desc) % 16) return ms_fail(c, MS_ERR_INVALID, "ms_projection_candidates: descriptors must be 16-byte aligned");
    MS_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_projection_candidates<false>, dim3(ms_div_up(nq, 4)), dim3(256), 0, c->stream, sorted_x, sorted_y, sorted_idx, n_kp, t_desc, t_octave, t_skip,
                       q_x, q_y, q_radius, q_min_octave, q_max_octave, q_desc, nq, best_idx, best_dist, second_dist, best_octave, second_octave, second_idx, n_candidates);
    MS_KERNEL_CHECK(c, "k_projection_candidates");
    return MS_OK;
}

int ms_projection_topk(ms_ctx *c, const float *sorted_x, const float *sorted_y, const int32_t *sorted_idx, int n_kp,
                       const uint32_t *t_desc, const int32_t *t_octave, const uint8_t *t_skip,
                       const float *q_x, const float *q_y, const float *q_radius, const int32_t *q_min_octave, const int32_t *q_max_octave,
                       const uint32_t *q_desc, int nq, int32_t *top_idx, uint16_t *top_dist, int32_t *top_octave, int32_t *n_scored, int32_t *n_candidates) {
    if (!c || n_kp < 0 || nq < 0 || !q_x || !q_y || !q_radius || !q_desc || !top_idx || !top_dist || !top_octave || !n_scored) return MS_ERR_INVALID;
    if (n_kp && (!sorted_x || !sorted_y || !sorted_idx || !t_desc)) return MS_ERR_INVALID;
    if (n_kp >= (1 << 20)) return ms_fail(c, MS_ERR_CAPACITY, "ms_projection_topk: %d keypoints (max %d)", n_kp, (1 << 20) - 1);
    if (nq == 0) return MS_OK;
    if (reinterpret_cast<uintptr_t>(q_desc) % 16 || reinterpret_cast<uintptr_t>(t_desc) % 16) return ms_fail(c, MS_ERR_INVALID, "ms_projection_topk: descriptors must be 16-byte aligned");
    MsRange range("match");
    MS_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_projection_candidates<true>, dim3(ms_div_up(nq, 4)), dim3(256), 0, c->stream, sorted_x, sorted_y, sorted_idx, n_kp, t_desc, t_octave, t_skip,
                       q_x, q_y, q_radius, q_min_octave, q_max_octave, q_desc, nq, top_idx, top_dist, static_cast<uint16_t *>(nullptr), top_octave,
                       static_cast<int32_t *>(nullptr), n_scored, n_candidates);
    MS_KERNEL_CHECK(c, "k_projection_candidates<topk>");
    return MS_OK;
}

int ms_angle_check(const float *delta_angle, const int32_t *ids, int n, int32_t *invalid_ids) {
    if (n < 0 || (n && (!delta_angle || !ids || !invalid_ids))) return MS_ERR_INVALID;
    std::vector<int> bin(n);
    int count[30] = {0};
    for (int i = 0; i < n; ++i) {                       // match_angle_checker.h:72-83
        float d = delta_angle[i];
        if (d < 0.0) d = (float)((double)d + 360.0);
        if (360.0 <= d) d = (float)((double)d - 360.0);
        int b = (int)std::lrintf(d * (1.0f / 30));
        bin[i] = (b < 0 || b >= 30) ? 29 : b;
        count[bin[i]]++;
    }
    int top[3] = {-1, -1, -1};
    for (int r = 0; r < 3; ++r) {
        int bc = -1;
        for (int k = 0; k < 30; ++k) if (k != top[0] && k != top[1] && count[k] > bc) { bc = count[k]; top[r] = k; }
    }
    int m = 0;
    for (int b = 0; b < 30; ++b) {
        if (b == top[0] || b == top[1] || b == top[2]) continue;
        for (int i = 0; i < n; ++i) if (bin[i] == b) invalid_ids[m++] = ids[i];
    }
    return m;
}

int ms_ratio_test(ms_ctx *c, const int32_t *best_idx, const uint16_t *best_dist, const uint16_t *second_dist,
                  int n, float lowe_ratio, int max_dist, int32_t *match) {
    if (!c || !best_idx || !best_dist || !second_dist || !match || n < 0) return MS_ERR_INVALID;
    if (n == 0) return MS_OK;
    MS_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_ratio_test, dim3(ms_div_up(n, 256)), dim3(256), 0, c->stream, best_idx, best_dist, second_dist, n, lowe_ratio, max_dist, match);
    MS_KERNEL_CHECK(c, "k_ratio_test");
    return MS_OK;
}

int ms_descriptor_medoid(ms_ctx *c, const uint32_t *desc_pool, const int32_t *obs_start, const int32_t *obs_idx, int n_points,
                         int max_obs, int32_t *best_local, int32_t *best_pool) {
    if (!c || !desc_pool || !obs_start || !obs_idx || n_points < 0 || max_obs < 0 || (!best_local && !best_pool)) return MS_ERR_INVALID;
    if (n_points == 0) return MS_OK;
    if (max_obs > MS_MEDOID_MAX_OBS) return ms_fail(c, MS_ERR_CAPACITY, "descriptor medoid: %d observations (max %d)", max_obs, MS_MEDOID_MAX_OBS);
    MS_HIP(c, hipSetDevice(c->device));
    const int n = max_obs > 0 ? max_obs : 1;
    const size_t lds = sizeof(uint32_t) * 9 * (size_t)n + sizeof(uint16_t) * (size_t)n * (n | 1) + 16;
    static bool attr_done = false;
    if (!attr_done) {
        MS_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_descriptor_medoid), hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL(k_descriptor_medoid, dim3(n_points), dim3(256), lds, c->stream, desc_pool, obs_start, obs_idx, n, best_local, best_pool);
    MS_KERNEL_CHECK(c, "k_descriptor_medoid");
    return MS_OK;
}

static int launch_greedy(ms_ctx *c, bool tri, const ms_match_frame *p1, const ms_match_frame *p2, int n_pairs,
                         const double *E12, const float *sf, float thr_deg, float ratio, int check_orientation,
                         int32_t *const *matched, int32_t *n_matches) {
    MsRange range("match");
    if (!c || !p1 || !p2 || !matched || !n_matches || n_pairs < 0) return MS_ERR_INVALID;
    if (n_pairs == 0) return MS_OK;
    if (n_pairs > 65535) return ms_fail(c, MS_ERR_CAPACITY, "greedy matcher: %d pairs in one call (max 65535)", n_pairs);
    int max_n1 = 0, max_n2 = 0, max_nodes = 0;
    for (int p = 0; p < n_pairs; ++p) {
        if (p2[p].n > 32768 || p1[p].n < 0 || p2[p].n < 0) return ms_fail(c, MS_ERR_CAPACITY, "greedy matcher: kf2 has %d keypoints (max 32768)", p2[p].n);
        if (reinterpret_cast<uintptr_t>(p1[p].desc) % 16 || reinterpret_cast<uintptr_t>(p2[p].desc) % 16)
            return ms_fail(c, MS_ERR_INVALID, "greedy matcher: descriptors must be 16-byte aligned");
        if (p1[p].bow.n_nodes < 0 || p2[p].bow.n_nodes < 0) return ms_fail(c, MS_ERR_INVALID, "greedy matcher: negative node count");
        max_n1 = std::max(max_n1, p1[p].n); max_n2 = std::max(max_n2, p2[p].n); max_nodes = std::max(max_nodes, p1[p].bow.n_nodes);
    }
    MS_HIP(c, hipSetDevice(c->device));
    // one argument table, one copy: [kf1 structs][kf2 structs][output pointers], then the device-only part of the scratch
    const size_t fb = sizeof(ms_match_frame) * (size_t)n_pairs, mb = sizeof(int32_t *) * (size_t)n_pairs;
    const size_t tab = ms_align_up(2 * fb + mb, 256), hb = ms_align_up(sizeof(int32_t) * 32 * (size_t)n_pairs, 256);
    const size_t o1 = ms_align_up(sizeof(int32_t) * (size_t)n_pairs * std::max(max_n1, 1), 256), o2 = ms_align_up(sizeof(int32_t) * (size_t)n_pairs * std::max(max_n2, 1), 256);
    const bool node_parallel = c->greedy_path != 1;
    const size_t bl = ms_align_up(16 * ((size_t)n_pairs * std::max(max_nodes, 1) + 1), 256);
    void *scr = nullptr;
    int rc = ms_scratch(c, tab + (node_parallel ? hb + o1 + o2 + bl : 0), &scr);
    if (rc != MS_OK) return rc;
    char *base = static_cast<char *>(scr);
    std::vector<char> host(2 * fb + mb);
    std::memcpy(host.data(), p1, fb); std::memcpy(host.data() + fb, p2, fb); std::memcpy(host.data() + 2 * fb, matched, mb);
    MS_HIP(c, hipMemcpyAsync(base, host.data(), host.size(), hipMemcpyHostToDevice, c->stream));     // pageable source: staged before the call returns
    GreedyArgs A{};
    A.f1 = reinterpret_cast<const ms_match_frame *>(base);
    A.f2 = reinterpret_cast<const ms_match_frame *>(base + fb);
    A.matched = reinterpret_cast<int32_t *const *>(base + 2 * fb);
    A.n_matches = n_matches; A.E12 = E12; A.scale_factors = sf; A.lowe_ratio = ratio; A.residual_deg_thr = thr_deg;
    A.check_orientation = check_orientation;
    if (node_parallel) {
        GreedyScratch S{};
        S.hist = reinterpret_cast<int32_t *>(base + tab);
        S.own1 = reinterpret_cast<int32_t *>(base + tab + hb);
        S.own2 = reinterpret_cast<int32_t *>(base + tab + hb + o1);
        S.big = reinterpret_cast<int32_t *>(base + tab + hb + o1 + o2);
        S.stride1 = std::max(max_n1, 1); S.stride2 = std::max(max_n2, 1);
        hipLaunchKernelGGL(k_greedy_init, dim3(std::max(ms_div_up(std::max(max_n1, max_n2), 256), 1), n_pairs), dim3(256), 0, c->stream, A, S);
        if (max_nodes > 0) {
            if (tri) hipLaunchKernelGGL(k_greedy_nodes<true>, dim3(max_nodes, n_pairs), dim3(64), 0, c->stream, A, S);
            else hipLaunchKernelGGL(k_greedy_nodes<false>, dim3(max_nodes, n_pairs), dim3(64), 0, c->stream, A, S);
            // a node can only hold more than 256 of kf2's keypoints if kf2 has that many
            if (max_n2 > 64 * kGreedyCpl) {
                bool *attr_done = c->greedy_attr_done;
                if (!attr_done[tri]) {
                    if (tri) MS_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_greedy_big_nodes<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBigLds));
                    else MS_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_greedy_big_nodes<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBigLds));
                    attr_done[tri] = true;
                }
                // as many workgroups as are resident at once (two per CU at 76 KB of LDS each); they share the work list, and all leave at once when it is empty
                const int wgs = c->greedy_path == 2 ? 1 : (int)std::min<size_t>((size_t)n_pairs * max_nodes, (size_t)std::max(c->n_cu, 1) * 2);
                if (tri) hipLaunchKernelGGL(k_greedy_big_nodes<true>, dim3(wgs), dim3(256), kBigLds, c->stream, A, S);
                else hipLaunchKernelGGL(k_greedy_big_nodes<false>, dim3(wgs), dim3(256), kBigLds, c->stream, A, S);
            }
        }
        hipLaunchKernelGGL(k_greedy_finish, dim3(n_pairs), dim3(256), 0, c->stream, A, S);
        A.redo = S.hist;                                   // pairs the node pass gave up (a keypoint in two shared nodes): exact sequential walk
    }
    if (tri) hipLaunchKernelGGL(k_match_greedy<true>, dim3(n_pairs), dim3(64), 0, c->stream, A);
    else hipLaunchKernelGGL(k_match_greedy<false>, dim3(n_pairs), dim3(64), 0, c->stream, A);
    MS_KERNEL_CHECK(c, "k_match_greedy");
    return MS_OK;
}

int ms_match_set_path(ms_ctx *c, int path) {
    if (!c || path < 0 || path > 2) return MS_ERR_INVALID;
    c->greedy_path = path;
    return MS_OK;
}

int ms_match_loop_closure(ms_ctx *c, const ms_match_frame *pairs1, const ms_match_frame *pairs2, int n_pairs,
                          float lowe_ratio, int check_orientation, int32_t *const *matched, int32_t *n_matches) {
    return launch_greedy(c, false, pairs1, pairs2, n_pairs, nullptr, nullptr, 0.f, lowe_ratio, check_orientation, matched, n_matches);
}

int ms_match_triangulation(ms_ctx *c, const ms_match_frame *pairs1, const ms_match_frame *pairs2, int n_pairs,
                           const double *E12, const float *scale_factors, float residual_deg_thr,
                           int check_orientation, int32_t *const *matched, int32_t *n_matches) {
    if (!E12 || !scale_factors) return MS_ERR_INVALID;
    return launch_greedy(c, true, pairs1, pairs2, n_pairs, E12, scale_factors, residual_deg_thr, 0.f, check_orientation, matched, n_matches);
}

}  // extern "C"
