// ba.hip -- local bundle adjustment on gfx950: the whole Levenberg-Marquardt loop on the device.
//
// Replaces the numerical core the reference delegates to g2o (bundle_adjuster.cpp:149-154, :322-323,
// :372-373, :376-379): EdgeSE3ProjectXYZ residuals + 2x6 / 2x3 Jacobians + Huber weights, the damped
// normal equations, their solution and g2o's LM accept/reject schedule.  Problem construction (which
// vertices/edges, information matrices, the two-stage schedule, outlier deletion, write-back) stays with
// the host wrapper (slam-module_amd/host/bundle_adjuster.hpp), exactly as bundle_adjuster.cpp does it.
//
// Design: ONE 512-thread workgroup per problem, persistent for all iterations, problems of a batch in
// parallel across the 256 CUs.  A local-BA window is small for this chip (C4: 20 k observations, 6.6 MB
// per iteration), so throughput comes from batching windows and from never returning to the host inside
// a solve -- not from spreading one window over the chip (hard part 5).
//   linearise   per point (thread = point): Hll (3x3), bl, and Hpl (6x3) per observation
//               per pose  (wave = pose)   : Hpp (6x6), bp by a fixed-order wave reduction; SE3 edges one thread each
//   Schur       S = Hpp + lambda I - sum_p Hpl (Hll + lambda I)^-1 Hpl^T as ordered sums of 6x6 block products: the
//               host sorts the observation pairs of every point by pose pair once; a wave per pose pair stages the
//               144-byte records of 32 pairs into LDS (coalesced 16-byte pieces) and multiplies them from there
//   solve       blocked left-looking Cholesky (16-column panels, v_mfma_f64_16x16x4_f64 update from L2, rhs carried as
//               an extra row), wave-level back substitution in LDS, point back-substitution, SE3 exponential update
//   LM          g2o's schedule: lambda0 = 1e-5 max diag, rho = dF / (dx.(lambda dx + b) + 1e-3),
//               lambda *= max(1/3, min(2/3, 1-(2 rho-1)^3)) or lambda *= nu, nu *= 2, <= 10 trials
// The 6x6 / 6x3 / 3x3 block work is plain fp64 VALU (no MFMA: it is sparse block work, not a dense contraction); only the
// dense Cholesky panel update uses the f64 MFMA.  The Schur sums have a fixed order; the few SE3-edge atomics do not,
// so results may differ in the last bits run-to-run; parity is judged at 1e-5 on the residuals.
#include "ms_internal.h"
#include <sys/prctl.h>
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <type_traits>
#include <utility>

#define g_ba_host_allocs g_ms_host_allocs      // (the library-wide counter of ms_internal.h)

namespace {

constexpr int NT = 512;          // threads per problem
constexpr int NW = NT / 64;      // waves
constexpr int NB = 16;           // Cholesky panel width
constexpr int CH = 32;           // Schur work items (pairs of observations of one point) per chunk
constexpr int kMaxFreePoses = 176;   // Cholesky panel (6*176+1) x 16 doubles + solution vector must fit the LDS budget; beyond it the panel lives in global memory
constexpr int kMaxFreePosesTeam = 2048;   // ... and the factorisation is spread over the team (cholesky_factor_team); the solution vector (6*2048 doubles) still fits the LDS
constexpr int kMaxTeam = 64;         // workgroups that may share one problem
constexpr size_t kBaCacheSlack = 8;  // a kept device block serves a request of at least 1/8 of its size
constexpr size_t kBaCacheMaxBytes = (size_t)1 << 30;   // blocks kept per context for the next ms_ba_create: 1 GiB in total, larger ones are freed at destroy
#ifndef MS_BA_LDS_KB
#define MS_BA_LDS_KB 150                  // -DMS_BA_LDS_KB=78 -DMS_FS_OB=32 -DMS_BA_WAVES_PER_EU=4: the two-windows-per-CU build of tools/ba_occupancy_probe.py
#endif
constexpr size_t kLdsBytes = (size_t)MS_BA_LDS_KB * 1024;  // max(Schur staging 8 x 9 KB, Cholesky panel (n+1) x 16 doubles + x (n doubles))

// One set of passes of the fused Schur phase.  A pass owns the free-pose rows [row0, row1) of S: their envelope part lives in an
// LDS tile while every point that observes one of these poses adds its block products; points are packed into batches of
// whole points with at most 64 observations (one per lane), and each batch lists its (observation a, observation b) pairs.
struct FsSet {
    int32_t n_pass;
    const int32_t *row0, *row1;              // [n_pass]
    const int32_t *batch_start;              // [n_pass + 1] first batch of a pass
    const int32_t *b_obs_start, *b_run_start;    // [n_batch + 1]: lane slots / first pair element (uint16 units, 8-aligned) of a batch
    const int32_t *b_fmt;                    // [n_batch] 0: the batch's pairs are chunks of 8; n > 0: n single pairs, one per lane (batches with <= 64 pairs);
                                             // < 0: G points with the SAME k poses, lane = point * k + pose slot -- no pair list in memory, the kernel enumerates the pairs
                                             // (slot a >= a0, slot b <= a, all points) itself: -1 - fmt = G | k << 7 | a0 << 12 (a0: first slot inside the pass's rows)
    const int32_t *pobs;                     // per lane slot: observation, pose vertex, point, free pose index (int4); a batch owns FS_OB slots, the unused ones hold observation -1
    const double *puv;                       // per lane slot: u, v, information, 0 (the observation's constants side by side: two 16-byte loads at an address that needs no index)
    const uint16_t *pairs;                   // chunks of 8 pairs (a_lane | b_lane << 8, 0xFFFF = none): the pairs of a chunk fall into the same block (pose a,
                                             // pose b); a batch's chunks are sorted by block
    const int32_t *rowoff;                   // per pass, concatenated: offset (doubles) of pose row r0 + i inside the pass's tile
    const int32_t *yoff;                     // [n_pass][2]: offset of the pass's rhs segment (6 doubles per row) = size of its matrix part; start of the pass in rowoff
    int32_t by_points;                       // 0: a pass owns its rows of S (every point that touches them is visited; plain stores).  1: a pass owns a SET OF POINTS
                                             // (each point visited once in the whole launch); the passes' row ranges overlap and their sums meet in S through atomics
};

struct BaProb {
    int32_t n_pose, n_point, n_obs, n_edge, np_free, n6, max_iters;
    int32_t team;                            // workgroups that share this problem (1 = the whole solve in one workgroup)
    int32_t chol_team;                       // of these, the workgroups that share the distributed Cholesky (systems beyond kMaxFreePoses)
    int32_t debug_reject;                    // test hook (ms_ba_debug_force_reject): the first n damped trials of k_ba_lm count as rejected
    double huber;
    // state
    double *pose, *pose_bk, *point, *point_bk;
    const double *pose0, *point0;            // initial estimates (solve() restarts from these)
    const int32_t *pidx;                     // pose vertex -> free index or -1
    const uint8_t *point_fixed;              // may be null
    const int32_t *obs_pose, *obs_point;
    const double *obs_uv, *obs_info;
    const int32_t *pt_start, *pt_obs;        // observations grouped by point
    const int32_t *fstart, *fobs;            // observations grouped by FREE pose index; behind them, fobs[fstart[np_free] .. n_obs), those of the fixed poses
    const int32_t *free2pose;                // free index -> pose vertex
    const int32_t *fo_lo;                    // the observations in fobs order as a stream: (point, observation | point fixed << 31, pose vertex, free pose index or -1) ...
    const double *fo_uvi;                    // ... and (u, v, information, 0): coalesced loads at addresses that need no index (linearise_stream)
    // Schur work list: pairs (a,b) of observations of one free point with free poses fb <= fa, sorted by (fa,fb),
    // cut into chunks of CH items of the same pose pair (padding = -1); segments = runs of chunks of one pair
    int32_t n_chunks, n_seg;
    const int32_t *chunk_items;              // [n_chunks*CH*2]
    const int32_t *seg_start, *seg_pair;     // [n_seg+1], [n_seg] (fa<<16 | fb)
    const int32_t *edge_i, *edge_j;
    const double *edge_meas, *edge_info;
    // work
    double *Hpp, *S, *bp, *dp, *y, *Hll, *bl, *Hinv, *Hpl, *dl, *chi2_obs, *Y;
    const double *zrow;                      // n6 + 16 zeros
    double *dinv;                            // [n6] reciprocals of the Cholesky diagonal
    double *panG;                            // [(n6+1) x 16] Cholesky panel in global memory, only for systems beyond kMaxFreePoses (else null)
    const int32_t *act_start, *act_blk;      // with panG: per 16-column panel the row tiles (relative to the panel) whose envelope reaches it, CSR
    const int32_t *env16;                    // [n6/16 + 2] envelope of S per 16-row block: first structurally non-zero column (0 for the rhs row's block)
    // team state (team > 1): arrival counter (monotonic, one 128-B line), per-workgroup partial sums [2][team][2], solve status
    uint32_t *bar;
    double *red;
    int32_t *flag;
    // fused Schur pass (schur_fused): two pass sets -- [0] coarse (as few passes as the LDS tile allows: one workgroup walks them
    // all), [1] fine (about one pose row per pass: the workgroups of a team take them round-robin); null when the problem uses the
    // record-based path (a point with more than 64 free observations, or a pose row wider than the LDS tile)
    FsSet fs[2];
    int32_t fused;                           // 1: schur_fused / point_backsub_fused (no Hpl / Y records exist); 0: the record-based path
    // windowed Cholesky (cholesky_window): the active front of the factorisation as W x W tiles of 16 x 16 in LDS; 0 = front too wide
    int32_t cw_meta_lds;                     // 1: the five index arrays below are staged in LDS for the factorisation (they fit beside the tiles)
    int32_t cw_W, cw_zglobal;                // tiles per side of the window; 1: the rhs vector stays in global memory (it does not fit the LDS beside the tiles)
    const int32_t *cw_slot;                  // [nblk] LDS slot (row and column index in the tile grid) of 16-row block b while it is active
    const int32_t *cw_act_start, *cw_act;    // per panel p: the other active blocks (block | slot << 16), ascending
    const int32_t *cw_load_start, *cw_load;  // per panel p: tiles that enter the window (bi | si << 16, bj | sj << 16)
    const int32_t *fs_cs;                    // [np_free] first scalar column of pose row fa held in the tile (6 * first coupled pose), then [np_free] the first column of the row in which Hpp
                                             // can be non-zero (6 * first pose coupled by a pose-pose edge; Hpp is the diagonal blocks and the edges' blocks, lower triangle only)
    // one free pose + free points (k_ba_one_pose): the observations sorted by point -- pose vertex, original index, (u, v, information) --, the per-point
    // records [28][n_point] and the team's partial sums [2][team][64]; null for every other shape
    const int32_t *op_pose, *op_o;
    const double *op_uvi;
    double *op_rec, *op_red;
    // results
    double *pack;                            // k_ba_pose_only: the block ms_ba_download reads ([16 stats][7 n_pose][3 n_point][n_obs chi2]) is written by the solver itself (null: k_ba_pack_result does it)
    double *stats;                           // [16]: iters, trials, stop, lambda, chi2_init, chi2_final, ok, -, then cycles per phase:
                                             //       8 eval, 9 linearise, 10 Schur, 11 Cholesky+backsub, 12 points+update, 13 total
};

// Out-of-line device functions see plain pointers as generic (flat) pointers; flat loads count on BOTH memory counters, so
// every LDS wait would also wait for prefetched global data.  The hot loops therefore use explicitly address-space-typed pointers.
#define MS_GLOBAL __attribute__((address_space(1)))
#define MS_LDS __attribute__((address_space(3)))
typedef double d2_t __attribute__((ext_vector_type(2)));   // builtin vectors: loadable from any address space (HIP's double2 struct is not)
typedef int i2_t __attribute__((ext_vector_type(2)));
typedef int i4_t __attribute__((ext_vector_type(4)));
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
// A pointer that is the same in every lane, as an SGPR pair in the global address space: loads through it are global_load with a scalar base (no 64-bit vector
// address arithmetic, and -- unlike the flat loads a plain pointer read out of BaProb turns into -- they count on vmcnt only, so an LDS wait does not wait for them)
template <class T>
__device__ __forceinline__ const MS_GLOBAL T *uglobal(const T *p) {
    const uint64_t a = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return (const MS_GLOBAL T *)(((uint64_t)hi << 32) | lo);
}

// ---------------------------------------------------------------- SE3 helpers (g2o / Eigen conventions)
__device__ __forceinline__ void q_normalize(double *q) {
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
__device__ __forceinline__ void q_mul(const double *a, const double *b, double *r) {
    r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void q_rot(const double *q, const double *v, double *r) {
    const double ux = 2 * (q[1] * v[2] - q[2] * v[1]), uy = 2 * (q[2] * v[0] - q[0] * v[2]), uz = 2 * (q[0] * v[1] - q[1] * v[0]);
    r[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
    r[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
    r[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}
__device__ __forceinline__ void q_to_R(const double *q, double *R) {
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0], tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
__device__ void R_to_q(const double *m, double *q) {
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
        double qq[4];
        qq[i] = 0.5 * t; t = 0.5 / t;
        qq[3] = (m[3 * k + j] - m[3 * j + k]) * t;
        qq[j] = (m[3 * j + i] + m[3 * i + j]) * t;
        qq[k] = (m[3 * k + i] + m[3 * i + k]) * t;
        q[0] = qq[0]; q[1] = qq[1]; q[2] = qq[2]; q[3] = qq[3];
    }
}
__device__ void se3_mul(const double *a, const double *b, double *r) {
    double t[3], q[4];
    q_rot(a, b + 4, t);
    q_mul(a, b, q);
    r[0] = q[0]; r[1] = q[1]; r[2] = q[2]; r[3] = q[3];
    r[4] = t[0] + a[4]; r[5] = t[1] + a[5]; r[6] = t[2] + a[6];
    q_normalize(r);
}
__device__ void se3_inv(const double *a, double *r) {
    double t[3];
    r[0] = -a[0]; r[1] = -a[1]; r[2] = -a[2]; r[3] = a[3];
    q_rot(r, a + 4, t);
    r[4] = -t[0]; r[5] = -t[1]; r[6] = -t[2];
}
__device__ __forceinline__ void mat3_mul(const double *A, const double *B, double *C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ void skew(const double *v, double *S) { S[0] = 0; S[1] = -v[2]; S[2] = v[1]; S[3] = v[2]; S[4] = 0; S[5] = -v[0]; S[6] = -v[1]; S[7] = v[0]; S[8] = 0; }

__device__ void se3_exp(const double *u, double *pose) {          // SE3Quat::exp
    const double theta = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    double O[9], O2[9], R[9], V[9];
    skew(u, O); mat3_mul(O, O, O2);
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (theta < 0.00001) {
#pragma unroll
        for (int i = 0; i < 9; ++i) { R[i] = I[i] + O[i] + 0.5 * O2[i]; V[i] = I[i] + 0.5 * O[i] + O2[i] / 6.; }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / (theta * theta * theta);
#pragma unroll
        for (int i = 0; i < 9; ++i) { R[i] = I[i] + a * O[i] + b * O2[i]; V[i] = I[i] + b * O[i] + c * O2[i]; }
    }
    R_to_q(R, pose);
#pragma unroll
    for (int i = 0; i < 3; ++i) pose[4 + i] = V[3 * i] * u[3] + V[3 * i + 1] * u[4] + V[3 * i + 2] * u[5];
    q_normalize(pose);
}
__device__ void se3_log(const double *pose, double *out) {        // SE3Quat::log
    double R[9], O[9], O2[9], Vi[9], om[3];
    q_to_R(pose, R);
    const double d = 0.5 * (R[0] + R[4] + R[8] - 1);
    const double dR[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (fabs(d) > 0.99999) {
        for (int i = 0; i < 3; ++i) om[i] = 0.5 * dR[i];
        skew(om, O); mat3_mul(O, O, O2);
        for (int i = 0; i < 9; ++i) Vi[i] = I[i] - 0.5 * O[i] + (1. / 12.) * O2[i];
    } else {
        const double theta = acos(d);
        for (int i = 0; i < 3; ++i) om[i] = theta / (2 * sqrt(1 - d * d)) * dR[i];
        skew(om, O); mat3_mul(O, O, O2);
        const double k = (1 - theta / (2 * tan(theta / 2))) / (theta * theta);
        for (int i = 0; i < 9; ++i) Vi[i] = I[i] - 0.5 * O[i] + k * O2[i];
    }
    for (int i = 0; i < 3; ++i) { out[i] = om[i]; out[3 + i] = Vi[3 * i] * pose[4] + Vi[3 * i + 1] * pose[5] + Vi[3 * i + 2] * pose[6]; }
}
__device__ void se3_adj(const double *pose, double *A) {          // SE3Quat::adj, row-major 6x6
    double R[9], T[9], TR[9];
    q_to_R(pose, R); skew(pose + 4, T); mat3_mul(T, R, TR);
    for (int i = 0; i < 36; ++i) A[i] = 0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { A[6 * i + j] = R[3 * i + j]; A[6 * (i + 3) + j + 3] = R[3 * i + j]; A[6 * (i + 3) + j] = TR[3 * i + j]; }
}

// EdgeSE3ProjectXYZ with fx=fy=1, cx=cy=0 (bundle_adjuster.cpp:59-62)
template <bool JAC>
__device__ __forceinline__ void proj_edge(const double *pose, const double *X, const double *uv, double *e, double *Jp, double *Jl) {
    double p[3];
    q_rot(pose, X, p);
    p[0] += pose[4]; p[1] += pose[5]; p[2] += pose[6];
    const double x = p[0], y = p[1], z = p[2];
    // ONE division per observation: u = x / z and v = y / z as products with 1 / z, and the Jacobian entries in u, v (the reference divides nine times,
    // types_six_dof_expmap.cpp; an IEEE fp64 division is ~18 instructions here, and this function is a third of the linearisation and of the Schur pass's
    // first half).  Each entry differs from the reference's by an ulp or two -- far inside the 1e-5 the residuals are held to.
    const double rz = 1.0 / z, u = x * rz, v = y * rz;
    e[0] = uv[0] - u; e[1] = uv[1] - v;
    if (JAC) {
        double R[9];
        q_to_R(pose, R);
        const double iz = -rz;
#pragma unroll
        for (int j = 0; j < 3; ++j) { Jl[j] = iz * (R[j] - u * R[6 + j]); Jl[3 + j] = iz * (R[3 + j] - v * R[6 + j]); }
        Jp[0] = u * v; Jp[1] = -(1 + u * u); Jp[2] = v; Jp[3] = iz; Jp[4] = 0; Jp[5] = u * rz;
        Jp[6] = 1 + v * v; Jp[7] = -(u * v); Jp[8] = -u; Jp[9] = 0; Jp[10] = iz; Jp[11] = v * rz;
    }
}
__device__ void pose_edge(const double *Ti, const double *Tj, const double *M, double *e, double *Ji, double *Jj, bool jac) {
    double Tjinv[7], A[7], B[7];
    se3_inv(Tj, Tjinv); se3_mul(Tjinv, M, A); se3_mul(A, Ti, B);
    se3_log(B, e);
    if (!jac) return;
    se3_adj(A, Ji);
    double Tiinv[7], Minv[7], C[7];
    se3_inv(Ti, Tiinv); se3_inv(M, Minv); se3_mul(Tiinv, Minv, C);
    se3_adj(C, Jj);
    for (int i = 0; i < 36; ++i) Jj[i] = -Jj[i];
}
__device__ __forceinline__ void huber(double chi2, double delta, double &rho0, double &w) {
    const double dsqr = delta * delta;
    if (delta <= 0 || chi2 <= dsqr) { rho0 = chi2; w = 1; }
    else { const double s = sqrt(chi2); rho0 = 2 * s * delta - dsqr; w = delta / s; }
}

// 1 / sqrt(d) for the pivots of the small Cholesky factorisations: v_rsq_f64 and two Newton steps (one cubic, one quadratic: full double precision) -- ~10
// instructions where sqrt and a division are ~60 on the solver's critical path
__device__ __forceinline__ double rsqrt_d(double d) {
    double y = __builtin_amdgcn_rsq(d);
    double e = fma(-d * y, y, 1.0);
    y = fma(y * e, fma(e, 0.375, 0.5), y);
    e = fma(-d * y, y, 1.0);
    return fma(y * e, 0.5, y);
}

// 144-byte (18 double) per-observation records are moved as nine 16-byte pieces: a lane's record sits in its own cache
// lines, so the vector-memory cost is per instruction, not per byte
__device__ __forceinline__ void load18(const double *p, double *v) {
    const double2 *q = reinterpret_cast<const double2 *>(p);
#pragma unroll
    for (int k = 0; k < 9; ++k) { const double2 u = q[k]; v[2 * k] = u.x; v[2 * k + 1] = u.y; }
}
__device__ __forceinline__ void store18(double *p, const double *v) {
    double2 *q = reinterpret_cast<double2 *>(p);
#pragma unroll
    for (int k = 0; k < 9; ++k) q[k] = double2{v[2 * k], v[2 * k + 1]};
}
__device__ __forceinline__ void load6(const double *p, double *v) {
    const double2 *q = reinterpret_cast<const double2 *>(p);
#pragma unroll
    for (int k = 0; k < 3; ++k) { const double2 u = q[k]; v[2 * k] = u.x; v[2 * k + 1] = u.y; }
}
__device__ __forceinline__ void store6(double *p, const double *v) {
    double2 *q = reinterpret_cast<double2 *>(p);
#pragma unroll
    for (int k = 0; k < 3; ++k) q[k] = double2{v[2 * k], v[2 * k + 1]};
}

// ---------------------------------------------------------------- block reductions (fixed order)
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double readlane_d(double v, int l) {          // wave-uniform broadcast of lane l's double (l uniform)
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ void lds_addd(MS_LDS double *p, double v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);        // ds_add_f64
}
__device__ double block_sum(double v, double *s_red) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += s_red[w];
    return t;
}
__device__ double block_max(double v, double *s_red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = s_red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) t = fmax(t, s_red[w]);
    return t;
}


// ---------------------------------------------------------------- a team of workgroups on one problem
// With team > 1 the phases below stride over all threads / waves of the team and every hand-off between phases is a barrier
// across the team's workgroups.  Visibility follows the one recipe that is valid across XCDs (per-XCD L2s are not coherent):
// every wave drains its stores (vmcnt 0) -> workgroup barrier -> one lane: agent-scope RELEASE fence (L2 write-back) ->
// vmcnt 0 -> relaxed agent atomic on the arrival counter; then that lane polls the counter (relaxed, agent scope), issues ONE
// agent-scope ACQUIRE fence (L1 invalidate), waits for it, and the workgroup barrier releases the other waves.  The counter only
// grows (barrier k completes at k * team arrivals), so there is no reset to race with.  All workgroups of a team must be resident:
// the host keeps problems * team <= CUs with one workgroup per CU (LDS), so they all become resident.  A poll gives up after kGiveUpTicks (2 s of the
// constant 100 MHz s_memrealtime clock) WITHOUT PROGRESS and marks the problem failed instead of hanging the GPU.  Progress = the arrival counter
// moved, or the team's heartbeat word did: a workgroup that works alone through a long phase while the others wait (the lead's factorisation of
// a large window) beats once per panel, so a legitimately long phase is not mistaken for a lost team.
#define BA_IDS                                                                                                   \
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;                                               \
    const int T_ = P.team, rank_ = T_ > 1 ? (int)(blockIdx.x % (unsigned)T_) : 0;                                \
    const int gt = rank_ * NT + tid, GT = T_ * NT, gw = rank_ * NW + wave, GW = T_ * NW;                         \
    /* per-item loops with long bodies (a thread per point, a wave per edge): consecutive items go to DIFFERENT workgroups -- 2000 points */ \
    /* on the first 2000 threads of a team are four workgroups at two waves per SIMD while 28 workgroups watch */                          \
    const int gts = tid * T_ + rank_, gws = (NW - 1 - wave) * T_ + rank_;                                        \
    (void)lane; (void)wave; (void)gt; (void)GT; (void)gw; (void)GW; (void)gts; (void)gws;

constexpr unsigned long long kGiveUpTicks = 200000000ull;      // 2 s of wall_clock64() (s_memrealtime, 100 MHz) without progress
constexpr int kBeatWord = 16;                                  // P.bar[kBeatWord]: the team's heartbeat (P.bar[0] / P.bar[32] are the two arrival counters)

// one lane of a workgroup that works alone while its team waits: "still here"
__device__ __forceinline__ void team_heartbeat(const BaProb &P) {
    if (P.team > 1) (void)__hip_atomic_fetch_add(P.bar + kBeatWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// FENCES = false: no agent-scope release / acquire (no L2 write-back, no invalidate) -- for a kernel whose workgroups exchange NOTHING but values written and
// read with agent-scope atomics (k_ba_one_pose: the partial sums); the barrier then only orders those: every wave's stores are acknowledged (vmcnt 0) before
// its workgroup arrives, and the values are loaded after the arrival counter was seen complete
template <bool FENCES = true>
__device__ __forceinline__ void group_sync(const BaProb &P, uint32_t *counter, uint32_t T) {
    if (T == 1) { __syncthreads(); return; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (FENCES) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        const uint32_t a = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t target = (a / T + 1u) * T;
        uint32_t seen = a + 1u, beat = __hip_atomic_load(P.bar + kBeatWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long t_last = wall_clock64();
        for (;;) {
            const uint32_t cur = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int32_t)(cur - target) >= 0) break;
            __builtin_amdgcn_s_sleep(8);
            // give up after 2 s without progress (or as soon as another workgroup has): every later barrier then falls through at once
            const uint32_t b = __hip_atomic_load(P.bar + kBeatWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long now = wall_clock64();
            if (cur != seen || b != beat) { seen = cur; beat = b; t_last = now; }
            if (now - t_last > kGiveUpTicks || __hip_atomic_load(P.flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                __hip_atomic_store(P.flag + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        if (FENCES) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    __syncthreads();
}
#ifndef MS_TEAM_NO_FENCE
#define MS_TEAM_NO_FENCE 0          /* -DMS_TEAM_NO_FENCE=1: k_ba_lm's team barriers without the agent-scope write-back / invalidate (WRONG results; timing A/B of tools/beside_probe.py) */
#endif
__device__ __noinline__ void team_sync(const BaProb &P) { group_sync<!MS_TEAM_NO_FENCE>(P, P.bar, (uint32_t)P.team); }
__device__ __noinline__ void team_sync_light(const BaProb &P, int team) { group_sync<false>(P, P.bar, (uint32_t)team); }
// barrier of the first P.chol_team workgroups only (the distributed factorisation), on a counter of its own (P.bar + 32: another 128-byte line)
__device__ __noinline__ void chol_sync(const BaProb &P) { group_sync(P, P.bar + 32, (uint32_t)P.chol_team); }

// Sum / maximum over the team, the same value (bit for bit) in every workgroup: partials are combined in rank order.
// `which` alternates per call site sequence so a fast workgroup cannot overwrite partials a slow one still reads.
__device__ __noinline__ double team_reduce(const BaProb &P, double v, double *s_red, bool is_max, int &seq) {
    const double b = is_max ? block_max(v, s_red) : block_sum(v, s_red);
    if (P.team == 1) return b;
    const int T = P.team, rank = (int)(blockIdx.x % (unsigned)T);
    double *slot = P.red + (size_t)(seq & 1) * 2 * T;
    ++seq;
    if (threadIdx.x == 0) slot[rank] = b;
    team_sync(P);
    // one lane combines the partials and hands the result on through LDS: every thread of the workgroup gets the SAME value even when a barrier
    // has given up and the slots are still being written (control flow that depends on it -- rho > 0 -- must stay uniform: it contains barriers)
    if (threadIdx.x == 0) {
        double t = slot[0];
        for (int r = 1; r < T; ++r) t = is_max ? fmax(t, slot[r]) : t + slot[r];
        s_red[0] = t;
    }
    __syncthreads();
    return s_red[0];                                   // (the next block_sum / block_max passes a barrier before it writes s_red)
}

// Two sums over the team behind ONE barrier (the robust chi2 of the new state and the step's gain denominator are needed together)
__device__ __noinline__ double team_reduce2(const BaProb &P, double v1, double v2, double *s_red, int &seq, double &out2) {
    const double b1 = block_sum(v1, s_red), b2 = block_sum(v2, s_red);
    if (P.team == 1) { out2 = b2; return b1; }
    const int T = P.team, rank = (int)(blockIdx.x % (unsigned)T);
    double *slot = P.red + (size_t)(seq & 1) * 2 * T;
    ++seq;
    if (threadIdx.x == 0) { slot[rank] = b1; slot[T + rank] = b2; }
    team_sync(P);
    if (threadIdx.x == 0) {                            // as in team_reduce: one reader, the workgroup takes the values from LDS
        double t1 = slot[0], t2 = slot[T];
        for (int r = 1; r < T; ++r) { t1 += slot[r]; t2 += slot[T + r]; }
        s_red[0] = t1; s_red[1] = t2;
    }
    __syncthreads();
    out2 = s_red[1];
    return s_red[0];
}

// robust chi2 of the current state (activeRobustChi2); optionally stores the plain chi2 per observation
__device__ __noinline__ double eval_chi2(const BaProb &P_, double *s_red_, bool store, int &seq, double extra = 0.0, double *extra_sum = nullptr) {
    const BaProb &P = P_;
    double *s_red = s_red_;
    BA_IDS
    double acc = 0;
    for (int o = gt; o < P.n_obs; o += GT) {
        double e[2];
        proj_edge<false>(P.pose + 7 * (size_t)P.obs_pose[o], P.point + 3 * (size_t)P.obs_point[o], P.obs_uv + 2 * (size_t)o, e, nullptr, nullptr);
        const double chi2 = P.obs_info[o] * (e[0] * e[0] + e[1] * e[1]);
        double r, w;
        huber(chi2, P.huber, r, w);
        if (store) P.chi2_obs[o] = chi2;
        acc += r;
    }
    for (int k = gt; k < P.n_edge; k += GT) {
        double e[6];
        pose_edge(P.pose + 7 * (size_t)P.edge_i[k], P.pose + 7 * (size_t)P.edge_j[k], P.edge_meas + 7 * (size_t)k, e, nullptr, nullptr, false);
        const double *W = P.edge_info + 36 * (size_t)k;
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) acc += e[i] * W[6 * i + j] * e[j];
    }
    if (extra_sum) return team_reduce2(P, acc, extra, s_red, seq, *extra_sum);     // a second, unrelated sum rides on the same barrier
    return team_reduce(P, acc, s_red, false, seq);
}

// ---------------------------------------------------------------- observation data one and two steps ahead
// The thread-per-point and wave-per-pose loops walk index chains (list entry -> observation -> pose / point -> values): three dependent
// memory round trips per observation with nothing to hide them at 2 waves per SIMD.  They keep the indices of the observation two steps
// ahead and the values of the next one in registers instead (a two-deep software pipeline), so a step waits for arithmetic, not for memory.
struct PtObs { int o, pi; double pose[7], uv[2], info; };
__device__ __forceinline__ void ptobs_idx(const BaProb &P, int ii, int end, int &o, int &pi) {
    o = -1; pi = 0;
    if (ii < end) { o = P.pt_obs[ii]; pi = P.obs_pose[o]; }
}
__device__ __forceinline__ void ptobs_data(const BaProb &P, int o, int pi, PtObs &d) {
    d.o = o; d.pi = pi;
    if (o >= 0) {
#pragma unroll
        for (int q = 0; q < 7; ++q) d.pose[q] = P.pose[7 * (size_t)pi + q];
        d.uv[0] = P.obs_uv[2 * (size_t)o]; d.uv[1] = P.obs_uv[2 * (size_t)o + 1]; d.info = P.obs_info[o];
    }
}
struct PoseObs { int o, l; double X[3], uv[2], info; };
__device__ __forceinline__ void poseobs_idx(const BaProb &P, int ii, int end, int &o, int &l) {
    o = -1; l = 0;
    if (ii < end) { o = P.fobs[ii]; l = P.obs_point[o]; }
}
__device__ __forceinline__ void poseobs_data(const BaProb &P, int o, int l, PoseObs &d) {
    d.o = o; d.l = l;
    if (o >= 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) d.X[q] = P.point[3 * (size_t)l + q];
        d.uv[0] = P.obs_uv[2 * (size_t)o]; d.uv[1] = P.obs_uv[2 * (size_t)o + 1]; d.info = P.obs_info[o];
    }
}

// ---------------------------------------------------------------- linearisation of one workgroup's window from coalesced streams (round 4)
// One workgroup per window (the chip-filling batch), point table in LDS.  Round 3's per-pose pass walked index chains -- list entry -> observation -> point ->
// position, uv, information: per-lane gathers through flat loads, two deep -- and took 12 k cycles per 64 observations where its arithmetic is ~1.5 k: 256 windows
// stream 270 MB per pass from HBM, and a wave had one or two gather groups in flight.  Here the observations of a pose are a static stream in pose order,
// (point, observation) and (u, v, information) side by side: coalesced loads whose addresses depend on nothing, requested three and two steps ahead; only the
// point's position is still a gather (one step ahead, from the 48 KB the window's points occupy).  Units of work are half poses, handed out from an LDS counter
// (the waves of a SIMD do not run at the same speed); a unit's 27 sums leave with global atomics (Hpp's diagonal blocks and bp are zero at this point).
__device__ __noinline__ double linearise_stream(const BaProb &P_, double *lds_v) {      // returns this thread's share of the robust chi2 of the observations (the caller adds the shares up when it wants them)
    __shared__ int s_ls_next;
    const BaProb &P = *(const BaProb *)uglobal(&P_);
    const int tid = threadIdx.x, lane = tid & 63;
    MS_LDS double *ptab = (MS_LDS double *)(uintptr_t)__builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(MS_LDS double *)lds_v);      // [n_point][H 6 | b 3]
    const int n_point = P.n_point, np = P.np_free, n6 = P.n6, n_obs = P.n_obs;
    const MS_GLOBAL i4_t *rec = (const MS_GLOBAL i4_t *)uglobal(P.fo_lo);
    const MS_GLOBAL d2_t *uvi = (const MS_GLOBAL d2_t *)uglobal(P.fo_uvi);
    const MS_GLOBAL double *gpoint = uglobal(P.point), *gpose = uglobal(P.pose);
    const MS_GLOBAL int32_t *fstart = uglobal(P.fstart), *free2pose = uglobal(P.free2pose);
    MS_GLOBAL double *Hpp = (MS_GLOBAL double *)uglobal(P.Hpp), *bp = (MS_GLOBAL double *)uglobal(P.bp);
    const double hub = P.huber;
    double chi2_share = 0;
    for (int i = tid; i < 9 * n_point; i += NT) ptab[i] = 0;
    if (tid == 0) s_ls_next = 0;
    __syncthreads();
    // one observation: residual, both Jacobians, Huber weight; the point's Hll / bl terms into the LDS table, the pose's terms into A / g when it is free
    auto point_terms = [&](int l, bool fixed, const double (&Jl)[6], const double (&e)[2], double wi) {
        if (fixed) return;
        double Jw[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) Jw[a] = wi * Jl[a];
        MS_LDS double *t = ptab + 9 * l;
#if MS_LIN_ABL == 1           /* timing ablation (wrong results): plain stores instead of LDS atomics */
        t[0] = fma(Jw[0], Jl[0], Jw[3] * Jl[3]); t[1] = fma(Jw[0], Jl[1], Jw[3] * Jl[4]); t[2] = fma(Jw[0], Jl[2], Jw[3] * Jl[5]);
        t[3] = fma(Jw[1], Jl[1], Jw[4] * Jl[4]); t[4] = fma(Jw[1], Jl[2], Jw[4] * Jl[5]); t[5] = fma(Jw[2], Jl[2], Jw[5] * Jl[5]);
        for (int a = 0; a < 3; ++a) t[6 + a] = -fma(Jw[a], e[0], Jw[3 + a] * e[1]);
        return;
#endif
        lds_addd(t + 0, fma(Jw[0], Jl[0], Jw[3] * Jl[3])); lds_addd(t + 1, fma(Jw[0], Jl[1], Jw[3] * Jl[4])); lds_addd(t + 2, fma(Jw[0], Jl[2], Jw[3] * Jl[5]));
        lds_addd(t + 3, fma(Jw[1], Jl[1], Jw[4] * Jl[4])); lds_addd(t + 4, fma(Jw[1], Jl[2], Jw[4] * Jl[5])); lds_addd(t + 5, fma(Jw[2], Jl[2], Jw[5] * Jl[5]));
#pragma unroll
        for (int a = 0; a < 3; ++a) lds_addd(t + 6 + a, -fma(Jw[a], e[0], Jw[3 + a] * e[1]));
    };
    // the observations of FIXED poses (the tail of the stream): point terms only, a thread each
    for (int ix = fstart[np] + tid; ix < n_obs; ix += NT) {
        const i4_t r = rec[ix];
        const d2_t uvv = uvi[2 * ix];
        const double info = ((const MS_GLOBAL double *)uvi)[4 * ix + 2];
        const int pi = r.z;
        double pose[7], X[3], e[2], Jp[12], Jl[6];
#pragma unroll
        for (int q = 0; q < 7; ++q) pose[q] = gpose[7 * pi + q];
#pragma unroll
        for (int q = 0; q < 3; ++q) X[q] = gpoint[3 * r.x + q];
        const double uv[2] = {uvv.x, uvv.y};
        proj_edge<true>(pose, X, uv, e, Jp, Jl);
        double rho, w;
        huber(info * (e[0] * e[0] + e[1] * e[1]), hub, rho, w);
        chi2_share += rho;
        point_terms(r.x, r.y < 0, Jl, e, w * info);
    }
    // the free poses, a wave each (handed out from a counter: the waves of a SIMD do not run at the same speed)
    for (;;) {
        int fp = 0;
        if (lane == 0) fp = atomicAdd(&s_ls_next, 1);
        fp = __builtin_amdgcn_readfirstlane(fp);
        if (fp >= np) break;
        const int pi = free2pose[fp];
        double pose[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) pose[q] = gpose[7 * pi + q];
        const int lo = fstart[fp], hi = fstart[fp + 1];
        double A[21], g[6];
#pragma unroll
        for (int a = 0; a < 21; ++a) A[a] = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) g[a] = 0;
        // every load of the pipeline is unconditional (indices clamped into the pose's run): a load inside a branch makes the compiler's wait for ANY later value a
        // wait for everything in flight, and the pipeline collapses into one round trip per step
        struct Ob { int l; bool fixed; double u, v, info; };
        auto fetch_ob = [&](int ii, Ob &d) {
            const int ic = min(ii, hi - 1);
            const i2_t r = ((const MS_GLOBAL i2_t *)rec)[2 * ic];
            d.l = r.x; d.fixed = r.y < 0;
            const d2_t a = uvi[2 * ic];
            d.u = a.x; d.v = a.y; d.info = ((const MS_GLOBAL double *)uvi)[4 * ic + 2];
        };
        auto fetch_x = [&](int l, double (&X)[3]) {
#pragma unroll
            for (int q = 0; q < 3; ++q) X[q] = gpoint[3 * l + q];
        };
        if (hi > lo) {
            Ob o0, o1, o2;
            double X0[3], X1[3];
            int ii = lo + lane;
            fetch_ob(ii, o0); fetch_ob(ii + 64, o1); fetch_ob(ii + 128, o2);
            fetch_x(o0.l, X0);
            for (; ii - lane < hi; ii += 64) {
                Ob o3;
                fetch_ob(ii + 192, o3);
                fetch_x(o1.l, X1);
                if (ii < hi) {
                    double e[2], Jp[12], Jl[6];
                    const double uv[2] = {o0.u, o0.v};
                    proj_edge<true>(pose, X0, uv, e, Jp, Jl);
                    double rho, w;
                    huber(o0.info * (e[0] * e[0] + e[1] * e[1]), hub, rho, w);
                    chi2_share += rho;
                    const double wi = w * o0.info;
                    double Jw[12];
#pragma unroll
                    for (int a = 0; a < 12; ++a) Jw[a] = wi * Jp[a];
                    int k = 0;
#pragma unroll
                    for (int a = 0; a < 6; ++a) {
                        g[a] = fma(-Jw[a], e[0], fma(-Jw[6 + a], e[1], g[a]));
#pragma unroll
                        for (int b = a; b < 6; ++b) { A[k] = fma(Jw[a], Jp[b], fma(Jw[6 + a], Jp[6 + b], A[k])); ++k; }
                    }
                    point_terms(o0.l, o0.fixed, Jl, e, wi);
                }
                o0 = o1; o1 = o2; o2 = o3;
                X0[0] = X1[0]; X0[1] = X1[1]; X0[2] = X1[2];
            }
        }
#pragma unroll
        for (int a = 0; a < 21; ++a) A[a] = wave_sum_d(A[a]);
#pragma unroll
        for (int a = 0; a < 6; ++a) g[a] = wave_sum_d(g[a]);
        if (lane == 0) {                                           // the pose's diagonal block and gradient: this wave alone writes them (plain stores)
            int k = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                bp[6 * fp + a] = g[a];
#pragma unroll
                for (int b = a; b < 6; ++b) { Hpp[(size_t)(6 * fp + a) * n6 + 6 * fp + b] = A[k]; Hpp[(size_t)(6 * fp + b) * n6 + 6 * fp + a] = A[k]; ++k; }
            }
        }
    }
    __syncthreads();
    MS_GLOBAL double *Hll = (MS_GLOBAL double *)uglobal(P.Hll), *bl = (MS_GLOBAL double *)uglobal(P.bl);
    for (int i = tid; i < 9 * n_point; i += NT) {                  // the point table leaves the LDS before the SE3 edges take it over
        const int l = i / 9, c = i - 9 * l;
        if (c < 6) Hll[6 * l + c] = ptab[i]; else bl[3 * l + c - 6] = ptab[i];
    }
    return chi2_share;
}

// robust chi2 of the current state for a window on ONE workgroup, from the same streams (eval_chi2 below walks observation -> pose / point index chains, per-lane
// gathers two dependent round trips deep): a flat loop over the stream, records two steps and point positions one step ahead, the poses in an LDS table
__device__ __noinline__ double eval_stream(const BaProb &P_, double *lds_v, double *s_red_v, bool store, double extra, double *extra_sum) {
    const BaProb &P = *(const BaProb *)uglobal(&P_);
    const int tid = threadIdx.x;
    MS_LDS double *ptab = (MS_LDS double *)(uintptr_t)__builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(MS_LDS double *)lds_v);      // [7 n_pose]
    const int n_obs = P.n_obs, n_pose = P.n_pose;
    const MS_GLOBAL i4_t *rec = (const MS_GLOBAL i4_t *)uglobal(P.fo_lo);
    const MS_GLOBAL d2_t *uvi = (const MS_GLOBAL d2_t *)uglobal(P.fo_uvi);
    const MS_GLOBAL double *gpoint = uglobal(P.point), *gpose = uglobal(P.pose);
    MS_GLOBAL double *chi2_obs = (MS_GLOBAL double *)uglobal(P.chi2_obs);
    const double hub = P.huber;
    for (int i = tid; i < 7 * n_pose; i += NT) ptab[i] = gpose[i];
    __syncthreads();
    double acc = 0;
    // the SE3 edges are a long dependent chain per edge (inverse, two products, the logarithm: ~30 k cycles in one lane) -- the LAST wave takes them, a lane each,
    // while the other waves stream the observations; all eight used to wait for the first wave's edges behind the reduction's barrier
    const int n_edge = P.n_edge;
    const bool split = n_edge > 0 && n_edge <= 64 && n_obs >= 4 * NT;
    const int OT = split ? NT - 64 : NT;                          // threads that walk the stream
    const bool edge_wave = split && tid >= OT;
    struct Ob { int l, o, pi; double u, v, info; };
    auto fetch_ob = [&](int ix, Ob &d) {
        const int ic = min(ix, n_obs - 1);
        const i4_t r = rec[ic];
        d.l = r.x; d.o = r.y & 0x7fffffff; d.pi = r.z;
        const d2_t a = uvi[2 * ic];
        d.u = a.x; d.v = a.y; d.info = ((const MS_GLOBAL double *)uvi)[4 * ic + 2];
    };
    if (n_obs > 0 && !edge_wave) {
        Ob o0, o1, o2;
        double X0[3], X1[3];
        int ix = tid;
        fetch_ob(ix, o0); fetch_ob(ix + OT, o1); fetch_ob(ix + 2 * OT, o2);
#pragma unroll
        for (int q = 0; q < 3; ++q) X0[q] = gpoint[3 * o0.l + q];
        for (; ix - tid < n_obs; ix += OT) {
            Ob o3;
            fetch_ob(ix + 3 * OT, o3);
#pragma unroll
            for (int q = 0; q < 3; ++q) X1[q] = gpoint[3 * o1.l + q];
            if (ix < n_obs) {
                double pose[7], e[2];
#pragma unroll
                for (int q = 0; q < 7; ++q) pose[q] = ptab[7 * o0.pi + q];
                const double uv[2] = {o0.u, o0.v};
                proj_edge<false>(pose, X0, uv, e, nullptr, nullptr);
                const double chi2 = o0.info * (e[0] * e[0] + e[1] * e[1]);
                double r, w;
                huber(chi2, hub, r, w);
                if (store) chi2_obs[o0.o] = chi2;
                acc += r;
            }
            o0 = o1; o1 = o2; o2 = o3;
            X0[0] = X1[0]; X0[1] = X1[1]; X0[2] = X1[2];
        }
    }
    for (int k = split ? tid - OT : tid; k >= 0 && k < n_edge; k += NT) {
        double e[6];
        pose_edge(P.pose + 7 * (size_t)P.edge_i[k], P.pose + 7 * (size_t)P.edge_j[k], P.edge_meas + 7 * (size_t)k, e, nullptr, nullptr, false);
        const double *W = P.edge_info + 36 * (size_t)k;
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) acc += e[i] * W[6 * i + j] * e[j];
    }
    const double total = block_sum(acc, s_red_v);
    if (extra_sum) *extra_sum = block_sum(extra, s_red_v);
    return total;
}

// dl = (Hll + lambda I)^-1 (bl - sum_a W_a^T dp_a) for a window on ONE workgroup, from the streams: every observation of a free pose takes its W^T dp off an LDS
// table that starts as bl (three ds_add_f64 per observation), then a thread per point solves the 3 x 3 system.  (point_backsub_fused's thread-per-point loop walks
// pt_obs -> observation -> pose index chains and fetches dp per observation.)
__device__ __noinline__ void backsub_stream(const BaProb &P_, double lambda_, double *lds_v) {
    const BaProb &P = *(const BaProb *)uglobal(&P_);
    const double lambda = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(lambda_)), __builtin_amdgcn_readfirstlane(__double2loint(lambda_)));
    const int tid = threadIdx.x;
    const int n_point = P.n_point, n_pose = P.n_pose, np = P.np_free;
    MS_LDS double *rtab = (MS_LDS double *)(uintptr_t)__builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(MS_LDS double *)lds_v);      // [3 n_point] | [7 n_pose] | [6 np_free]
    MS_LDS double *ptab = rtab + 3 * n_point, *xtab = ptab + 7 * n_pose;
    const MS_GLOBAL i4_t *rec = (const MS_GLOBAL i4_t *)uglobal(P.fo_lo);
    const MS_GLOBAL d2_t *uvi = (const MS_GLOBAL d2_t *)uglobal(P.fo_uvi);
    const MS_GLOBAL double *gpoint = uglobal(P.point), *gpose = uglobal(P.pose), *gbl = uglobal(P.bl), *gdp = uglobal(P.dp), *gHll = uglobal(P.Hll);
    const MS_GLOBAL uint8_t *pfix = P.point_fixed ? uglobal(P.point_fixed) : nullptr;
    MS_GLOBAL double *gdl = (MS_GLOBAL double *)uglobal(P.dl);
    const double hub = P.huber;
    const int n_free_obs = P.fstart[np];                           // the stream's observations of free poses come first
    for (int i = tid; i < 3 * n_point; i += NT) rtab[i] = gbl[i];
    for (int i = tid; i < 7 * n_pose; i += NT) ptab[i] = gpose[i];
    for (int i = tid; i < 6 * np; i += NT) xtab[i] = gdp[i];
    __syncthreads();
    struct Ob { int l, pi, fa; double u, v, info; };                // fa < 0: the observation's point is fixed (nothing to move)
    auto fetch_ob = [&](int ix, Ob &d) {
        const int ic = min(ix, n_free_obs - 1);
        const i4_t r = rec[ic];
        d.l = r.x; d.pi = r.z; d.fa = r.y < 0 ? -1 : r.w;
        const d2_t a = uvi[2 * ic];
        d.u = a.x; d.v = a.y; d.info = ((const MS_GLOBAL double *)uvi)[4 * ic + 2];
    };
    if (n_free_obs > 0) {
        Ob o0, o1, o2;
        double X0[3], X1[3];
        int ix = tid;
        fetch_ob(ix, o0); fetch_ob(ix + NT, o1); fetch_ob(ix + 2 * NT, o2);
#pragma unroll
        for (int q = 0; q < 3; ++q) X0[q] = gpoint[3 * o0.l + q];
        for (; ix - tid < n_free_obs; ix += NT) {
            Ob o3;
            fetch_ob(ix + 3 * NT, o3);
#pragma unroll
            for (int q = 0; q < 3; ++q) X1[q] = gpoint[3 * o1.l + q];
            if (ix < n_free_obs && o0.fa >= 0) {
                double pose[7], x[6], e[2], Jp[12], Jl[6];
#pragma unroll
                for (int q = 0; q < 7; ++q) pose[q] = ptab[7 * o0.pi + q];
#pragma unroll
                for (int q = 0; q < 6; ++q) x[q] = xtab[6 * o0.fa + q];
                const double uv[2] = {o0.u, o0.v};
                proj_edge<true>(pose, X0, uv, e, Jp, Jl);
                double rho, w;
                huber(o0.info * (e[0] * e[0] + e[1] * e[1]), hub, rho, w);
                const double wi = w * o0.info;
                double s0 = 0, s1 = 0;
#pragma unroll
                for (int a = 0; a < 6; ++a) { s0 = fma(Jp[a], x[a], s0); s1 = fma(Jp[6 + a], x[a], s1); }
                s0 *= wi; s1 *= wi;
#pragma unroll
                for (int c = 0; c < 3; ++c) lds_addd(rtab + 3 * o0.l + c, -fma(Jl[c], s0, Jl[3 + c] * s1));
            }
            o0 = o1; o1 = o2; o2 = o3;
            X0[0] = X1[0]; X0[1] = X1[1]; X0[2] = X1[2];
        }
    }
    __syncthreads();
    for (int l = tid; l < n_point; l += NT) {
        if (pfix && pfix[l]) { gdl[3 * l] = 0; gdl[3 * l + 1] = 0; gdl[3 * l + 2] = 0; continue; }
        const double r0 = rtab[3 * l], r1 = rtab[3 * l + 1], r2 = rtab[3 * l + 2];
        const MS_GLOBAL d2_t *h2 = (const MS_GLOBAL d2_t *)(gHll + 6 * l);
        const d2_t ha = h2[0], hb = h2[1], hc = h2[2];
        const double a = ha.x + lambda, b = ha.y, c = hb.x, d = hb.y + lambda, e2 = hc.x, f = hc.y + lambda;
        const double A = d * f - e2 * e2, B = c * e2 - b * f, C = b * e2 - c * d;
        const double id = 1.0 / (a * A + b * B + c * C);
        const double h0 = A * id, h1 = B * id, h2v = C * id, h3 = (a * f - c * c) * id, h4 = (b * c - a * e2) * id, h5 = (a * d - b * b) * id;
        gdl[3 * l] = h0 * r0 + h1 * r1 + h2v * r2;
        gdl[3 * l + 1] = h1 * r0 + h3 * r1 + h4 * r2;
        gdl[3 * l + 2] = h2v * r0 + h4 * r1 + h5 * r2;
    }
    __syncthreads();
}

// ---------------------------------------------------------------- linearisation
#ifndef MS_LIN_ABL
#define MS_LIN_ABL 0
#endif
#ifndef MS_BA_NO_FUSED_TRIAL
#define MS_BA_NO_FUSED_TRIAL 0  // 1: k_ba_lm never fuses a trial's chi2 with the next linearisation (A/B builds)
#endif
#ifndef MS_LIN_NO_STREAM
#define MS_LIN_NO_STREAM 0          /* -DMS_LIN_NO_STREAM=1: round 3's per-pose pass over index chains (A/B runs) */
#endif
#ifdef MS_LIN_PROF
__device__ long long g_lin[32 * 8 * 6];
extern "C" int ms_debug_linprof(long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lin), sizeof(g_lin)); }
#define LINP(i) do { if (lane == 0 && rank_ < 32) g_lin[(rank_ * 8 + wave) * 6 + i] = clock64() - lin_t0; } while (0)
#else
#define LINP(i) do { } while (0)
#endif
// chi2_share (optional): the linearisation visits every edge at the current state anyway -- when the caller wants the robust chi2 of that state (a trial of the fused
// schedule in k_ba_lm), every thread's share of it is added here (streamed, one-workgroup path only: the caller checks lin_streams())
__device__ __forceinline__ bool lin_streams(const BaProb &P) {      // build_system takes the streamed path (linearise_stream) for this problem on ONE workgroup
    return P.fused && 9 * (size_t)P.n_point + 64 <= kLdsBytes / 8 && P.fo_lo != nullptr && !MS_LIN_NO_STREAM;
}
__device__ __noinline__ void build_system(const BaProb &P_, double *lds_, double *chi2_share = nullptr) {
    const BaProb &P = P_;
    BA_IDS
    const int n6 = P.n6;
#ifdef MS_LIN_PROF
    const long long lin_t0 = clock64();
#endif
    if (P.fused) {
        // Hpp is its diagonal blocks and the lower blocks of the pose-pose edges: only those row pieces are written below, read by the Schur pass and cleared here (the
        // rest of the n6 x n6 array keeps the zeros it was created with; round 3 cleared all of it, 720 KB per window and iteration)
        const MS_GLOBAL int32_t *hcs = uglobal(P.fs_cs) + P.np_free;
        MS_GLOBAL double *Hz = (MS_GLOBAL double *)uglobal(P.Hpp);
        for (int it = gt >> 4; it < n6; it += GT / 16) {
            const int f = it / 6;
            for (int c = hcs[f] + (gt & 15); c < 6 * f + 6; c += 16) Hz[(size_t)it * n6 + c] = 0;
        }
    } else
    for (size_t i = gt; i < (size_t)n6 * n6; i += GT) P.Hpp[i] = 0;
    for (int i = gt; i < n6; i += GT) P.bp[i] = 0;
    LINP(0);
    team_sync(P);
    LINP(1);
    // A team on a window whose sums fit the LDS: a THREAD PER OBSERVATION, once.  Workgroup r takes the points [n r / T, n (r+1) / T) and with them a
    // contiguous run of the point-major observation list; each thread evaluates its observation (error, both Jacobians) and adds the point's
    // Hll / bl terms and the pose's diagonal-block / gradient terms into two LDS tables with ds_add_f64.  The point table goes out with plain
    // stores (the workgroup owns its points), the pose table with one global atomic per non-zero entry.  The thread-per-point loop below took
    // 51 k cycles for its 62 points per workgroup (ten observations one after another), the wave-per-pose-slice pass after it another 50 k --
    // each a walk over all observations with most of the lanes idle.  The SE3 edges keep their waves (the last ones of every workgroup), side by side with this.
    const int lin_l0 = T_ > 1 ? (int)((long long)P.n_point * rank_ / T_) : 0, lin_l1 = T_ > 1 ? (int)((long long)P.n_point * (rank_ + 1) / T_) : 0;
    const bool obs_par = T_ > 1 && P.fused && (size_t)NW * CH * 36 + 9 * (size_t)((P.n_point + T_ - 1) / T_ + 1) + 27 * (size_t)P.np_free + 64 <= kLdsBytes / 8;
    if (obs_par) {
        const int edge_waves_w = min(P.n_edge, GW / 2);        // the same split as below: waves with gws < edge_waves_w evaluate edges
        MS_LDS double *ptab = (MS_LDS double *)lds_ + NW * CH * 36, *qtab = ptab + 9 * (lin_l1 - lin_l0);     // [points][H 6 | b 3], [free poses][A 21 | g 6]
        __shared__ int s_lin_arrive;
        const int n_ptab = 9 * (lin_l1 - lin_l0), n_qtab = 27 * P.np_free;
        for (int i = tid; i < n_ptab + n_qtab; i += NT) ptab[i] = 0;
        if (tid == 0) s_lin_arrive = 0;
        __syncthreads();
        const bool edge_wave = gws < edge_waves_w;
        // the waves that have no edge: the observations of the workgroup's points, one per thread
        int n_work_waves = 0;
        for (int w = 0; w < NW; ++w) n_work_waves += ((NW - 1 - w) * T_ + rank_ < edge_waves_w) ? 0 : 1;
        int my_slot = 0;
        for (int w = 0; w < wave; ++w) my_slot += ((NW - 1 - w) * T_ + rank_ < edge_waves_w) ? 0 : 1;
        if (!edge_wave) {
            const int o_lo = P.pt_start[lin_l0], o_hi = P.pt_start[lin_l1];
            for (int idx = o_lo + my_slot * 64 + lane; idx < o_hi; idx += 64 * n_work_waves) {
                const int o = P.pt_obs[idx], pi = P.obs_pose[o], l = P.obs_point[o];
                double pose[7], X[3], uv[2];
#pragma unroll
                for (int q = 0; q < 7; ++q) pose[q] = P.pose[7 * (size_t)pi + q];
#pragma unroll
                for (int q = 0; q < 3; ++q) X[q] = P.point[3 * (size_t)l + q];
                uv[0] = P.obs_uv[2 * (size_t)o]; uv[1] = P.obs_uv[2 * (size_t)o + 1];
                const double info = P.obs_info[o];
                const int fp = P.pidx[pi];
                const bool lfree = !(P.point_fixed && P.point_fixed[l]);
                double e[2], Jp[12], Jl[6];
                proj_edge<true>(pose, X, uv, e, Jp, Jl);
                const double chi2 = info * (e[0] * e[0] + e[1] * e[1]);
                double r, w;
                huber(chi2, P.huber, r, w);
                const double wi = w * info;
                if (lfree) {
                    MS_LDS double *t = ptab + 9 * (l - lin_l0);
                    lds_addd(t + 0, wi * (Jl[0] * Jl[0] + Jl[3] * Jl[3])); lds_addd(t + 1, wi * (Jl[0] * Jl[1] + Jl[3] * Jl[4])); lds_addd(t + 2, wi * (Jl[0] * Jl[2] + Jl[3] * Jl[5]));
                    lds_addd(t + 3, wi * (Jl[1] * Jl[1] + Jl[4] * Jl[4])); lds_addd(t + 4, wi * (Jl[1] * Jl[2] + Jl[4] * Jl[5])); lds_addd(t + 5, wi * (Jl[2] * Jl[2] + Jl[5] * Jl[5]));
#pragma unroll
                    for (int a = 0; a < 3; ++a) lds_addd(t + 6 + a, -(Jl[a] * e[0] + Jl[3 + a] * e[1]) * wi);
                }
                if (fp >= 0) {
                    MS_LDS double *t = qtab + 27 * fp;
                    int k = 0;
#pragma unroll
                    for (int a = 0; a < 6; ++a) {
                        lds_addd(t + 21 + a, -(Jp[a] * e[0] + Jp[6 + a] * e[1]) * wi);
#pragma unroll
                        for (int b = a; b < 6; ++b) lds_addd(t + k++, wi * (Jp[a] * Jp[b] + Jp[6 + a] * Jp[6 + b]));
                    }
                }
            }
            // the working waves meet on an LDS counter (the edge waves are busy for ~20 k cycles more: a workgroup barrier would wait for them)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) {
                __hip_atomic_fetch_add(&s_lin_arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                while (__hip_atomic_load(&s_lin_arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < n_work_waves) __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const int wt = my_slot * 64 + lane, WT = 64 * n_work_waves;
            for (int i = wt; i < n_ptab; i += WT) {
                const int lp = i / 9, c = i - 9 * lp, l = lin_l0 + lp;
                if (c < 6) P.Hll[6 * (size_t)l + c] = ptab[i]; else P.bl[3 * (size_t)l + c - 6] = ptab[i];
            }
            for (int i = wt; i < n_qtab; i += WT) {
                const double v = qtab[i];
                if (v == 0) continue;
                const int fp = i / 27, c = i - 27 * fp;
                if (c >= 21) { atomicAdd(&P.bp[6 * fp + c - 21], v); continue; }
                int a = 0, rem = c;
                while (rem >= 6 - a) { rem -= 6 - a; ++a; }
                const int b = a + rem;
                atomicAdd(&P.Hpp[(size_t)(6 * fp + a) * n6 + 6 * fp + b], v);
                if (b != a) atomicAdd(&P.Hpp[(size_t)(6 * fp + b) * n6 + 6 * fp + a], v);
            }
        }
    }
    // One workgroup per window (a chip-filling batch) with a point table that fits the LDS (9 doubles per point: 2000 points = 141 KB): the per-point walk below is
    // dropped -- the per-pose pass visits every observation of a free pose anyway (lanes = the observations of ONE pose: all different points), so it adds the
    // point's Hll / bl terms into the table with ds_add_f64 while it is there; the observations of FIXED poses get a sweep of their own.  The two walks evaluated
    // every observation twice: 260 k + 340 k cycles of a 760 k-cycle linearisation.
    const bool one_pass = T_ == 1 && !obs_par && P.fused && 9 * (size_t)P.n_point + 64 <= kLdsBytes / 8;
    MS_LDS double *ptab1 = (MS_LDS double *)lds_;
    const bool stream = one_pass && P.fo_lo != nullptr && !MS_LIN_NO_STREAM;
    if (stream) { const double share = linearise_stream(P, lds_); if (chi2_share) *chi2_share += share; }
    if (one_pass && !stream) {
        for (int i = tid; i < 9 * P.n_point; i += NT) ptab1[i] = 0;
        __syncthreads();
        for (int ix = P.fstart[P.np_free] + tid; ix < P.n_obs; ix += NT) {      // observations whose pose is fixed (the tail of fobs): point terms only
            const int o = P.fobs[ix], pi = P.obs_pose[o], l = P.obs_point[o];
            if (P.point_fixed && P.point_fixed[l]) continue;
            double pose[7], X[3], uv[2];
#pragma unroll
            for (int q = 0; q < 7; ++q) pose[q] = P.pose[7 * (size_t)pi + q];
#pragma unroll
            for (int q = 0; q < 3; ++q) X[q] = P.point[3 * (size_t)l + q];
            uv[0] = P.obs_uv[2 * (size_t)o]; uv[1] = P.obs_uv[2 * (size_t)o + 1];
            double e[2], Jp[12], Jl[6];
            proj_edge<true>(pose, X, uv, e, Jp, Jl);
            const double info = P.obs_info[o], chi2 = info * (e[0] * e[0] + e[1] * e[1]);
            double r, w;
            huber(chi2, P.huber, r, w);
            const double wi = w * info;
            MS_LDS double *t = ptab1 + 9 * l;
            lds_addd(t + 0, wi * (Jl[0] * Jl[0] + Jl[3] * Jl[3])); lds_addd(t + 1, wi * (Jl[0] * Jl[1] + Jl[3] * Jl[4])); lds_addd(t + 2, wi * (Jl[0] * Jl[2] + Jl[3] * Jl[5]));
            lds_addd(t + 3, wi * (Jl[1] * Jl[1] + Jl[4] * Jl[4])); lds_addd(t + 4, wi * (Jl[1] * Jl[2] + Jl[4] * Jl[5])); lds_addd(t + 5, wi * (Jl[2] * Jl[2] + Jl[5] * Jl[5]));
#pragma unroll
            for (int a = 0; a < 3; ++a) lds_addd(t + 6 + a, -(Jl[a] * e[0] + Jl[3 + a] * e[1]) * wi);
        }
    }
    // per point: Hll, bl, Hpl
    for (int l = (obs_par || one_pass) ? P.n_point : gts; l < P.n_point; l += GT) {
        const bool lfree = !(P.point_fixed && P.point_fixed[l]);
        double H[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
        const double X[3] = {P.point[3 * (size_t)l], P.point[3 * (size_t)l + 1], P.point[3 * (size_t)l + 2]};
        const int iend = P.pt_start[l + 1];
        int ii = P.pt_start[l], o1, pi1;
        PtObs cur;
        ptobs_idx(P, ii, iend, o1, pi1); ptobs_data(P, o1, pi1, cur); ptobs_idx(P, ii + 1, iend, o1, pi1);
        for (; ii < iend; ++ii) {
            PtObs nxt;
            ptobs_data(P, o1, pi1, nxt);
            int o2, pi2;
            ptobs_idx(P, ii + 2, iend, o2, pi2);
            const int o = cur.o, pi = cur.pi;
            double e[2], Jp[12], Jl[6];
            proj_edge<true>(cur.pose, X, cur.uv, e, Jp, Jl);
            const double info = cur.info, chi2 = info * (e[0] * e[0] + e[1] * e[1]);
            double r, w;
            huber(chi2, P.huber, r, w);
            const double wi = w * info;
            if (lfree) {
#pragma unroll
                for (int a = 0; a < 3; ++a) b[a] += -(Jl[a] * e[0] + Jl[3 + a] * e[1]) * wi;
                H[0] += wi * (Jl[0] * Jl[0] + Jl[3] * Jl[3]); H[1] += wi * (Jl[0] * Jl[1] + Jl[3] * Jl[4]); H[2] += wi * (Jl[0] * Jl[2] + Jl[3] * Jl[5]);
                H[3] += wi * (Jl[1] * Jl[1] + Jl[4] * Jl[4]); H[4] += wi * (Jl[1] * Jl[2] + Jl[4] * Jl[5]); H[5] += wi * (Jl[2] * Jl[2] + Jl[5] * Jl[5]);
                if (P.pidx[pi] >= 0 && !P.fused) {      // the record-based Schur path keeps Hpl; the fused pass recomputes it
                    double W[18];
#pragma unroll
                    for (int a = 0; a < 6; ++a)
#pragma unroll
                        for (int c = 0; c < 3; ++c) W[3 * a + c] = wi * (Jp[a] * Jl[c] + Jp[6 + a] * Jl[3 + c]);
                    store18(P.Hpl + 18 * (size_t)o, W);
                }
            }
            cur = nxt; o1 = o2; pi1 = pi2;
        }
        store6(P.Hll + 6 * (size_t)l, H);
#pragma unroll
        for (int a = 0; a < 3; ++a) P.bl[3 * (size_t)l + a] = b[a];
    }
    LINP(2);
    // per free pose: Hpp diagonal block + bp (a wave per pose, lanes over its observations).  A team cuts every pose's observations into
    // slices so that all its waves have work, adds the slices' sums with fp64 atomics, and lets the first waves take the SE3 edges at the
    // same time (one wave per edge, ~27 k cycles each): both only ADD into Hpp / bp, so no barrier is needed between them.
    const int edge_waves = T_ > 1 ? min(P.n_edge, GW / 2) : 0;
    const int PW = GW - edge_waves, pw = gws - edge_waves;      // (gws: the edges land on the last waves of every workgroup, the point loop above kept the first ones busy)
    const int n_slice = T_ > 1 ? max(1, min(8, PW / max(P.np_free, 1))) : 1;
    for (int u = pw; !obs_par && !stream && pw >= 0 && u < P.np_free * n_slice; u += PW) {
        const int fp = u / n_slice, sl = u - fp * n_slice;
        const int pi = P.free2pose[fp];
        double pose[7];
#pragma unroll
        for (int a = 0; a < 7; ++a) pose[a] = P.pose[7 * (size_t)pi + a];
        double A[21], g[6];
#pragma unroll
        for (int a = 0; a < 21; ++a) A[a] = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) g[a] = 0;
        const int f_lo = P.fstart[fp], f_len = P.fstart[fp + 1] - f_lo;
        const int iend = f_lo + (int)((long long)f_len * (sl + 1) / n_slice);
        int ii = f_lo + (int)((long long)f_len * sl / n_slice) + lane, o1, l1;
        PoseObs cur;
        poseobs_idx(P, ii, iend, o1, l1); poseobs_data(P, o1, l1, cur); poseobs_idx(P, ii + 64, iend, o1, l1);
        for (; ii < iend; ii += 64) {
            PoseObs nxt;
            poseobs_data(P, o1, l1, nxt);
            int o2, l2;
            poseobs_idx(P, ii + 128, iend, o2, l2);
            double e[2], Jp[12], Jl[6];
            proj_edge<true>(pose, cur.X, cur.uv, e, Jp, Jl);
            const double info = cur.info, chi2 = info * (e[0] * e[0] + e[1] * e[1]);
            double r, w;
            huber(chi2, P.huber, r, w);
            const double wi = w * info;
            int k = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                g[a] += -(Jp[a] * e[0] + Jp[6 + a] * e[1]) * wi;
#pragma unroll
                for (int b = a; b < 6; ++b) A[k++] += wi * (Jp[a] * Jp[b] + Jp[6 + a] * Jp[6 + b]);
            }
            if (one_pass && cur.o >= 0 && !(P.point_fixed && P.point_fixed[cur.l])) {
                MS_LDS double *t = ptab1 + 9 * cur.l;
                lds_addd(t + 0, wi * (Jl[0] * Jl[0] + Jl[3] * Jl[3])); lds_addd(t + 1, wi * (Jl[0] * Jl[1] + Jl[3] * Jl[4])); lds_addd(t + 2, wi * (Jl[0] * Jl[2] + Jl[3] * Jl[5]));
                lds_addd(t + 3, wi * (Jl[1] * Jl[1] + Jl[4] * Jl[4])); lds_addd(t + 4, wi * (Jl[1] * Jl[2] + Jl[4] * Jl[5])); lds_addd(t + 5, wi * (Jl[2] * Jl[2] + Jl[5] * Jl[5]));
#pragma unroll
                for (int a = 0; a < 3; ++a) lds_addd(t + 6 + a, -(Jl[a] * e[0] + Jl[3 + a] * e[1]) * wi);
            }
            cur = nxt; o1 = o2; l1 = l2;
        }
#pragma unroll
        for (int a = 0; a < 21; ++a) A[a] = wave_sum_d(A[a]);
#pragma unroll
        for (int a = 0; a < 6; ++a) g[a] = wave_sum_d(g[a]);
        if (lane == 0) {
            int k = 0;
            for (int a = 0; a < 6; ++a) {
                if (T_ > 1) atomicAdd(&P.bp[6 * fp + a], g[a]); else P.bp[6 * fp + a] = g[a];
                for (int b = a; b < 6; ++b) {
                    if (T_ > 1) { atomicAdd(&P.Hpp[(size_t)(6 * fp + a) * n6 + 6 * fp + b], A[k]); if (b != a) atomicAdd(&P.Hpp[(size_t)(6 * fp + b) * n6 + 6 * fp + a], A[k]); }
                    else { P.Hpp[(size_t)(6 * fp + a) * n6 + 6 * fp + b] = A[k]; P.Hpp[(size_t)(6 * fp + b) * n6 + 6 * fp + a] = A[k]; }
                    ++k;
                }
            }
        }
    }
    LINP(3);
    if (one_pass && !stream) {                                     // the point table leaves the LDS before the SE3 edges take it over
        __syncthreads();
        for (int i = tid; i < 9 * P.n_point; i += NT) {
            const int l = i / 9, c = i - 9 * l;
            if (c < 6) P.Hll[6 * (size_t)l + c] = ptab1[i]; else P.bl[3 * (size_t)l + c - 6] = ptab1[i];
        }
    }
    if (T_ == 1) team_sync(P);
    // EdgeSE3Expmap edges (odometry chain, loop closures, orientation prior).  One workgroup: in rounds of 128, a LANE per edge evaluates it (error, both
    // 6x6 Jacobians; the edges of a round run side by side instead of one after another) and parks the result in LDS, then the
    // round's gradient entries (edge, side, a) and block entries (edge, side s, side t, a, b) of Js^T (W Jt) are spread over all threads,
    // each formed with the operation order of a scalar loop and added with fp64 atomics (neighbouring edges share a pose block).
    // A wave per edge (every lane computing the same edge) took 190 k cycles per linearisation at 49 edges -- as long as 20 k observations.
    if (T_ > 1) {     // a team: one WAVE per edge, the team's waves side by side (every lane evaluates the edge; lane (a, b) < 36 forms the block entries)
        MS_LDS double *slab = (MS_LDS double *)lds_ + (size_t)wave * (CH * 36);      // [Ji 36][Jj 36][W 36][We 6]
        for (int k = gws; gws < edge_waves && k < P.n_edge; k += edge_waves) {
            const int vi = P.edge_i[k], vj = P.edge_j[k], fi = P.pidx[vi], fj = P.pidx[vj];
            if (fi < 0 && fj < 0) continue;                                       // wave-uniform
            double e[6], Ji[36], Jj[36];
            pose_edge(P.pose + 7 * (size_t)vi, P.pose + 7 * (size_t)vj, P.edge_meas + 7 * (size_t)k, e, Ji, Jj, true);
            const double *W = P.edge_info + 36 * (size_t)k;
            if (lane < 36) { slab[lane] = Ji[lane]; slab[36 + lane] = Jj[lane]; slab[72 + lane] = W[lane]; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < 6) { double v = 0; for (int b2 = 0; b2 < 6; ++b2) v += slab[72 + 6 * lane + b2] * e[b2]; slab[108 + lane] = -v; }      // We = -W e
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int a2 = lane / 6, b3 = lane - 6 * a2;
            for (int sidx = 0; sidx < 2; ++sidx) {
                const int fs = sidx ? fj : fi;
                if (fs < 0) continue;
                const MS_LDS double *Js = slab + 36 * sidx;
                if (lane < 6) { double v = 0; for (int r2 = 0; r2 < 6; ++r2) v += Js[6 * r2 + lane] * slab[108 + r2]; atomicAdd(&P.bp[6 * fs + lane], v); }
                for (int tidx = 0; tidx < 2; ++tidx) {
                    const int ft = tidx ? fj : fi;
                    if (ft < 0 || ft > fs) continue;                           // (lower block triangle only: nothing reads the blocks above the diagonal)
                    const MS_LDS double *Jt = slab + 36 * tidx;
                    if (lane < 36) {
                        double v = 0;
                        for (int r2 = 0; r2 < 6; ++r2) {
                            double m = 0;                                         // (W Jt)[r2][b3]
                            for (int c2 = 0; c2 < 6; ++c2) m += slab[72 + 6 * r2 + c2] * Jt[6 * c2 + b3];
                            v += Js[6 * r2 + a2] * m;
                        }
                        atomicAdd(&P.Hpp[(size_t)(6 * fs + a2) * n6 + 6 * ft + b3], v);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        constexpr int ES = 80, EB = kLdsBytes >= 128 * ES * 8 ? 128 : 64;                                          // edges per round; doubles per edge: e 6 | Ji 36 | Jj 36 | pad
        MS_LDS double *eb = (MS_LDS double *)lds_;
        const int rounds = (P.n_edge + EB - 1) / EB;
        for (int rd = rank_; rd < rounds; rd += T_) {
            const int k0 = rd * EB, kn = min(EB, P.n_edge - k0);
            if (tid < kn) {
                const int k = k0 + tid, vi = P.edge_i[k], vj = P.edge_j[k];
                const bool touches = P.pidx[vi] >= 0 || P.pidx[vj] >= 0;
                if (!touches && chi2_share) {                                     // an edge between fixed poses: a constant, but part of the chi2
                    double e[6];
                    pose_edge(P.pose + 7 * (size_t)vi, P.pose + 7 * (size_t)vj, P.edge_meas + 7 * (size_t)k, e, nullptr, nullptr, false);
                    const double *W = P.edge_info + 36 * (size_t)k;
                    double c2 = 0;
                    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) c2 += e[i] * W[6 * i + j] * e[j];
                    *chi2_share += c2;
                }
                if (touches) {
                    double e[6], Ji[36], Jj[36];
                    pose_edge(P.pose + 7 * (size_t)vi, P.pose + 7 * (size_t)vj, P.edge_meas + 7 * (size_t)k, e, Ji, Jj, true);
                    if (chi2_share) {
                        const double *W = P.edge_info + 36 * (size_t)k;
                        double c2 = 0;
                        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) c2 += e[i] * W[6 * i + j] * e[j];
                        *chi2_share += c2;
                    }
                    MS_LDS double *d = eb + tid * ES;
#pragma unroll
                    for (int q = 0; q < 6; ++q) d[q] = e[q];
#pragma unroll
                    for (int q = 0; q < 36; ++q) { d[6 + q] = Ji[q]; d[42 + q] = Jj[q]; }
                }
            }
            __syncthreads();
            for (int it = tid; it < kn * 12; it += NT) {                          // bp[fs] += Js^T (-W e)
                const int ke = it / 12, r = it - 12 * ke, side = r / 6, a2 = r - 6 * side, k = k0 + ke;
                const int fs = P.pidx[side ? P.edge_j[k] : P.edge_i[k]];
                if (fs < 0) continue;
                const MS_LDS double *d = eb + ke * ES, *Js = d + 6 + 36 * side;
                const double *W = P.edge_info + 36 * (size_t)k;
                double v = 0;
                for (int r2 = 0; r2 < 6; ++r2) {
                    double we = 0;
                    for (int b2 = 0; b2 < 6; ++b2) we += W[6 * r2 + b2] * d[b2];
                    v += Js[6 * r2 + a2] * -we;
                }
                atomicAdd(&P.bp[6 * fs + a2], v);
            }
            // Hpp[fs][ft] += Js^T (W Jt): a thread per (edge, side pair, column b3) forms the column (W Jt)[:, b3] once and the six entries of the block's column
            // from it -- every entry with the operation order of the scalar loop (a thread per ENTRY recomputed the column for each of the six: 68 k cycles)
            for (int it = tid; it < kn * 24; it += NT) {
                const int ke = it / 24, r = it - 24 * ke, st = r / 6, b3 = r - 6 * st, k = k0 + ke;
                const int fs = P.pidx[(st >> 1) ? P.edge_j[k] : P.edge_i[k]], ft = P.pidx[(st & 1) ? P.edge_j[k] : P.edge_i[k]];
                if (fs < 0 || ft < 0 || ft > fs) continue;                    // (lower block triangle only: nothing reads the blocks above the diagonal)
                const MS_LDS double *d = eb + ke * ES, *Js = d + 6 + 36 * (st >> 1), *Jt = d + 6 + 36 * (st & 1);
                const double *W = P.edge_info + 36 * (size_t)k;
                double m[6];
#pragma unroll
                for (int r2 = 0; r2 < 6; ++r2) {
                    double mm = 0;                                                // (W Jt)[r2][b3]
#pragma unroll
                    for (int c2 = 0; c2 < 6; ++c2) mm += W[6 * r2 + c2] * Jt[6 * c2 + b3];
                    m[r2] = mm;
                }
#pragma unroll
                for (int a2 = 0; a2 < 6; ++a2) {
                    double v = 0;
#pragma unroll
                    for (int r2 = 0; r2 < 6; ++r2) v += Js[6 * r2 + a2] * m[r2];
                    atomicAdd(&P.Hpp[(size_t)(6 * fs + a2) * n6 + 6 * ft + b3], v);
                }
            }
            __syncthreads();
        }
    }
    LINP(4);
    team_sync(P);
    LINP(5);
}

// The (observation a, observation b) pairs of a chunk, one per record piece this lane moves (9 pieces per lane) ...
__device__ __forceinline__ void schur_fetch_items(const MS_GLOBAL int32_t *chunk_items, int ch, int lane, i2_t (&ab)[9]) {
    const MS_GLOBAL i2_t *items = reinterpret_cast<const MS_GLOBAL i2_t *>(chunk_items) + (size_t)ch * CH;
#pragma unroll
    for (int k = 0; k < 9; ++k) ab[k] = items[(lane + 64 * k) / 18];
}
// ... and the 16-byte pieces of their Y_a / Hpl_b records
__device__ __forceinline__ void schur_fetch_records(const MS_GLOBAL double *Y, const MS_GLOBAL double *Hpl, int lane, const i2_t (&ab)[9], d2_t (&v)[9]) {
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int piece = (lane + 64 * k) % 18;
        v[k] = d2_t{0.0, 0.0};
        if (ab[k].x >= 0) v[k] = piece < 9 ? reinterpret_cast<const MS_GLOBAL d2_t *>(Y + 18 * (size_t)ab[k].x)[piece]
                                           : reinterpret_cast<const MS_GLOBAL d2_t *>(Hpl + 18 * (size_t)ab[k].y)[piece - 9];
    }
}

// ---------------------------------------------------------------- damped solve
// The damped solve, phase by phase.  The phases are compiled out of line (noinline): each gets its own register
// allocation, so the staging registers of the Schur loop are not spilled because another phase needs many registers
// (measured: 67.4 -> 65.7 Mcycles per C4 solve; fully inlined without the register prefetch: 112 Mcycles).
__device__ __noinline__ void schur_prepare(const BaProb &P_, double lambda) {
    const BaProb &P = P_;
    BA_IDS
    const int n = P.n6;
    (void)n;
    // (Hll + lambda I)^-1, closed form
    for (int l = gt; l < P.n_point; l += GT) {
        if (P.point_fixed && P.point_fixed[l]) continue;
        double H[6];
        load6(P.Hll + 6 * (size_t)l, H);
        const double a = H[0] + lambda, b = H[1], c = H[2], d = H[3] + lambda, e = H[4], f = H[5] + lambda;
        const double A = d * f - e * e, B = c * e - b * f, C = b * e - c * d;
        const double det = a * A + b * B + c * C;
        if (!(fabs(det) > 0) || !isfinite(det)) P.flag[0] = 0;
        const double id = 1.0 / det;
        const double Hi[6] = {A * id, B * id, C * id, (a * f - c * c) * id, (b * c - a * e) * id, (a * d - b * b) * id};
        store6(P.Hinv + 6 * (size_t)l, Hi);
    }
    team_sync(P);
    // Schur complement S = Hpp + lambda I - sum_l Hpl (Hll + lambda I)^-1 Hpl^T (lower block triangle), rhs y = bp - sum Y bl.
    // Deterministic and atomic-free: every (pose pair) block is the ordered sum of per-chunk partial blocks.
    for (int o = gt; o < P.n_obs; o += GT) {                         // Y_o = Hpl_o * Hinv_l
        const int l = P.obs_point[o];
        if (P.pidx[P.obs_pose[o]] < 0 || (P.point_fixed && P.point_fixed[l])) continue;
        double W[18], h[6], Yo[18];
        load18(P.Hpl + 18 * (size_t)o, W);
        load6(P.Hinv + 6 * (size_t)l, h);
        const double Hm[9] = {h[0], h[1], h[2], h[1], h[3], h[4], h[2], h[4], h[5]};
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) Yo[3 * r + c] = W[3 * r] * Hm[c] + W[3 * r + 1] * Hm[3 + c] + W[3 * r + 2] * Hm[6 + c];
        store18(P.Y + 18 * (size_t)o, Yo);
    }
    {   // S = Hpp + lambda I on the lower block triangle: thread = (row, 64-column strip), 4 loads in flight
        const double *__restrict__ Hp = P.Hpp;
        double *__restrict__ Sp0 = P.S;
        const int strips = (n + 63) / 64;
        for (int w = gw; w < n * strips; w += GW) {
            const int r = w / strips, c = (w - r * strips) * 64 + (tid & 63);
            if (c < n && c / 6 <= r / 6) Sp0[(size_t)r * n + c] = Hp[(size_t)r * n + c] + (c == r ? lambda : 0.0);
        }
    }
    team_sync(P);
    for (int fp = gw; fp < P.np_free; fp += GW) {                   // rhs: fixed-order wave reduction per pose
        double gsum[6] = {0, 0, 0, 0, 0, 0};
        for (int ii = P.fstart[fp] + lane; ii < P.fstart[fp + 1]; ii += 64) {
            const int o = P.fobs[ii], l = P.obs_point[o];
            if (P.point_fixed && P.point_fixed[l]) continue;
            double Yo[18];
            load18(P.Y + 18 * (size_t)o, Yo);
            const double *bl = P.bl + 3 * (size_t)l;
#pragma unroll
            for (int r = 0; r < 6; ++r) gsum[r] += Yo[3 * r] * bl[0] + Yo[3 * r + 1] * bl[1] + Yo[3 * r + 2] * bl[2];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) gsum[r] = wave_sum_d(gsum[r]);
        if (lane < 6) P.y[6 * fp + lane] = P.bp[6 * fp + lane] - gsum[lane];
    }
    team_sync(P);
}

__device__ __noinline__ void schur_segments(const BaProb &P_, double *lds_) {
    const BaProb &P = P_;
    double *lds = lds_;
    BA_IDS
    const int n = P.n6;
    (void)n;
    // One wave per pose-pair segment.  Per chunk of 32 items the wave stages the 32 Y_a and 32 Hpl_b records
    // (144 B each) into its private LDS slab with coalesced 16-byte pieces (9 lanes per record), then lane
    // (t = lane/2, half = lane%2) multiplies item t's record pair into 18 of the 36 block entries and keeps the
    // running sum in registers across the segment's chunks; one fixed-order butterfly over t ends the segment.
    {
        MS_LDS double *stage = (MS_LDS double *)lds + (size_t)wave * (CH * 36);      // [32 items][18 Y | 18 W]
        const MS_GLOBAL int32_t *chunk_items = (const MS_GLOBAL int32_t *)P.chunk_items, *seg_start = (const MS_GLOBAL int32_t *)P.seg_start,
                                *seg_pair = (const MS_GLOBAL int32_t *)P.seg_pair;
        const MS_GLOBAL double *Yp = (const MS_GLOBAL double *)P.Y, *Hplp = (const MS_GLOBAL double *)P.Hpl;
        MS_GLOBAL double *Sp = (MS_GLOBAL double *)P.S;
        const int n_seg = P.n_seg;
        for (int seg = gw; seg < n_seg; seg += GW) {
            double acc[18];
#pragma unroll
            for (int q = 0; q < 18; ++q) acc[q] = 0;
            const int t = lane >> 1, half = lane & 1;
            // software pipeline, two deep: while chunk ch is multiplied the record pieces of chunk ch+1 and the item indices of
            // chunk ch+2 are in flight, so neither of the two dependent L2 round trips of a chunk is on the critical path
            const int ch_begin = seg_start[seg], ch_end = seg_start[seg + 1];
            d2_t cur[9], nxt[9];
            i2_t idx[9];
            schur_fetch_items(chunk_items, ch_begin, lane, idx);
            schur_fetch_records(Yp, Hplp, lane, idx, cur);
            if (ch_begin + 1 < ch_end) schur_fetch_items(chunk_items, ch_begin + 1, lane, idx);
            for (int ch = ch_begin; ch < ch_end; ++ch) {
#pragma unroll
                for (int k = 0; k < 9; ++k) reinterpret_cast<MS_LDS d2_t *>(stage)[lane + 64 * k] = cur[k];
                if (ch + 1 < ch_end) {
                    schur_fetch_records(Yp, Hplp, lane, idx, nxt);
                    if (ch + 2 < ch_end) schur_fetch_items(chunk_items, ch + 2, lane, idx);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const MS_LDS double *rec = stage + t * 36;
                double ya[9], wb[18];
#pragma unroll
                for (int q = 0; q < 9; ++q) ya[q] = rec[9 * half + q];           // rows 3*half .. 3*half+2 of Y
#pragma unroll
                for (int q = 0; q < 18; ++q) wb[q] = rec[18 + q];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 6; ++c) acc[6 * r + c] += ya[3 * r] * wb[3 * c] + ya[3 * r + 1] * wb[3 * c + 1] + ya[3 * r + 2] * wb[3 * c + 2];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 9; ++k) cur[k] = nxt[k];
            }
#pragma unroll
            for (int q = 0; q < 18; ++q) {
#pragma unroll
                for (int off = 2; off < 64; off <<= 1) acc[q] += __shfl_xor(acc[q], off, 64);
            }
            if (lane < 2) {                                           // lane = half: rows 3*half .. 3*half+2 of the block
                const int fa = seg_pair[seg] >> 16, fb = seg_pair[seg] & 0xFFFF;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 6; ++c) Sp[(size_t)(6 * fa + 3 * half + r) * n + 6 * fb + c] -= acc[6 * r + c];
            }
        }
    }
    team_sync(P);
}

// ---------------------------------------------------------------- fused Schur pass (point-major, S accumulated in LDS)
// S = Hpp + lambda I - sum_l W_l (Hll_l + lambda I)^-1 W_l^T and y = bp - sum_l W_l (Hll_l + lambda I)^-1 bl_l without ever
// writing the per-observation records to memory.  The record-based path above re-fetches every 144-byte record ~10 times
// (once per partner observation of its point): 32 MB per damped solve at C4, 7x the algorithmic bytes, which is what bounds a
// batch of 256 windows.  Here a pass keeps the envelope part of a range of pose rows of S in an LDS tile; every point that
// observes one of these poses is visited once: a wave takes a batch of whole points (<= 64 observations, one per lane), each
// lane recomputes its observation's Jacobians and writes Z_a = W_a L^-T (Hll + lambda I = L L^T, 3x3) into the wave's LDS
// slab, then the lanes take the batch's (a, b) pairs and subtract Z_a Z_b^T (= W_a (Hll + lambda I)^-1 W_b^T) from block
// (pose a, pose b) of the tile with LDS atomics.  Per damped solve the pass reads the 32 bytes of every observation (once per
// pass its point touches) instead of 288 bytes per PAIR.  The sums are no longer in a fixed order (LDS atomics), like the SE3-edge sums.
#ifndef MS_FS_OB
#define MS_FS_OB 64
#endif
constexpr int FS_OB = MS_FS_OB;                                  // observations (lanes) per batch
constexpr int FS_ZD = 14;                                        // doubles per slab entry: Z_a = Jp_a^T G_a is kept as its factors, G (2 x 3) and the 8 entries of Jp that are neither zero nor repeated
constexpr int kFsStageDoubles = NW * FS_OB * FS_ZD;              // slabs of the 8 waves: 57,344 B
constexpr int kFsMetaDoubles = NW * FS_OB / 2;                   // free-pose index per lane: 2,048 B
constexpr int kFsPoseTab = 128;                                  // poses (all vertices, free or fixed) whose 7 doubles each sit in LDS during the pass; a window with more reads them from memory
constexpr int kFsRowTab = 320;                                   // pose rows a pass can hold at most (a row needs 42 doubles of the tile at least): their (first column, tile offset) pairs sit in LDS
constexpr int kFsTileDoubles = (int)(kLdsBytes / 8) - kFsStageDoubles - kFsMetaDoubles - 7 * kFsPoseTab - 2 * kFsRowTab;     // 10,240 doubles = 80 KB
static_assert(kFsTileDoubles / 42 <= kFsRowTab, "row table too small for the tile");

__device__ __forceinline__ void lds_sub(MS_LDS double *p, double v) {
    (void)__hip_atomic_fetch_add(p, -v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);       // ds_add_f64
}

// Stamps inside schur_fused (build with BA_EXTRA=-DMS_FS_PROF, read with tools/ba_schur_prof.py): per-wave cycle sums of workgroup 0, kept in registers and written once per call:
// 0 hand-out, 1 top of the batch (index loads issued), 2 Jacobians + slab stores, 3 pair products, 4 last flush, 5 batches, 6 next batch's lane values requested, 7 single pairs / enumerated pairs
#ifdef MS_FS_PROF
__device__ long long g_fsprof[NW * 8];
extern "C" int ms_debug_fsprof(long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fsprof), sizeof(g_fsprof)); }
#define FSP_DECL long long fst = clock64(), fsacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define FSP(i) do { const long long _t = clock64(); fsacc[i] += _t - fst; fst = _t; } while (0)
#if MS_FS_PROF > 1
#define FSP_WAIT_VM asm volatile("s_waitcnt vmcnt(0)" ::: "memory")      /* -DMS_FS_PROF=2: the wait for the batch's loads gets a stamp of its own (and is forced to sit there) */
#else
#define FSP_WAIT_VM do { } while (0)
#endif
#define FSP_FLUSH do { if (blockIdx.x == 0 && lane == 0) { for (int i = 0; i < 8; ++i) g_fsprof[wave * 8 + i] += fsacc[i]; } } while (0)
#else
#define FSP_DECL
#define FSP(i) do { } while (0)
#define FSP_WAIT_VM do { } while (0)
#define FSP_FLUSH do { } while (0)
#endif
#ifndef MS_FS_PROCEDURAL
#define MS_FS_PROCEDURAL 1
#endif
#ifndef MS_FS_ABL
#define MS_FS_ABL 0
#endif
constexpr bool kProceduralPairs = MS_FS_PROCEDURAL != 0;         // batches of equal pose sets carry no pair list (-DMS_FS_PROCEDURAL=0: lists for every batch, for A/B runs)
template <bool PROCEDURAL, bool POSE_LDS>    // PROCEDURAL: the pass set has batches without pair lists (fmt < 0).  Two copies of the function: the enumeration code in the pair loop cost the
                              // list-only launches (256 windows, one workgroup each) 5 % through register allocation alone, whether or not it ever ran
                              // POSE_LDS: every pose vertex of the window fits the LDS table (n_pose <= kFsPoseTab).  A template parameter, not a branch: where a value may come from
                              // LDS or from memory the compiler's wait for it covers both counters in full, and with them every load that was meant to stay in flight
__device__ __noinline__ void schur_fused(const BaProb &P_, double lambda_, double *lds_v, long long *cyc) {
    __shared__ unsigned long long s_fs_q[NW];              // per wave: its range of the pass's batches still to do, next | end << 32 (the owner takes from the front, a wave that has run dry from the end)
    // The arguments of an out-of-line function arrive in vector registers, and everything derived from them counts as per-lane data: the problem's fields, the LDS
    // regions, the pass's bounds ... ~30 registers of "the same value in every lane", in a function that needs every one of its 256.  Say that they are uniform.
    const BaProb &P = *(const BaProb *)uglobal(&P_);
    const double lambda = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(lambda_)), __builtin_amdgcn_readfirstlane(__double2loint(lambda_)));
    double *lds_ = (double *)(MS_LDS double *)(uintptr_t)__builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(MS_LDS double *)lds_v);
    BA_IDS
    const int n = P.n6;
    const FsSet &F = P.fs[T_ > 1 ? 1 : 0];
    MS_LDS double *stage = (MS_LDS double *)lds_ + (size_t)wave * (FS_OB * FS_ZD);
    MS_LDS int32_t *meta = (MS_LDS int32_t *)((MS_LDS double *)lds_ + kFsStageDoubles) + wave * FS_OB;
    MS_LDS double *ptab = (MS_LDS double *)lds_ + kFsStageDoubles + kFsMetaDoubles;          // [7 n_pose] when n_pose <= kFsPoseTab
    MS_LDS i4_t *rowtab = (MS_LDS i4_t *)(ptab + 7 * kFsPoseTab);                          // [r1 - r0] of the pass: first scalar column of the row's envelope part, offset of the row in the tile, first column of its part of Hpp
    MS_LDS double *tile = (MS_LDS double *)rowtab + 2 * kFsRowTab;
    constexpr bool pose_lds = POSE_LDS;
    const MS_GLOBAL int32_t *cs = uglobal(P.fs_cs), *hcs = cs + P.np_free, *env = uglobal(P.env16);
    const MS_GLOBAL i4_t *pobs4 = (const MS_GLOBAL i4_t *)uglobal(F.pobs);
    const MS_GLOBAL u4_t *chunks = (const MS_GLOBAL u4_t *)uglobal(F.pairs);
    const MS_GLOBAL uint16_t *pairs16 = uglobal(F.pairs);
    const MS_GLOBAL int32_t *b_fmt = uglobal(F.b_fmt);
    const MS_GLOBAL int32_t *b_run = uglobal(F.b_run_start);
    const MS_GLOBAL d2_t *puv = (const MS_GLOBAL d2_t *)uglobal(F.puv);
    const MS_GLOBAL double *Hpp = uglobal(P.Hpp);
    const MS_GLOBAL double *gpose = uglobal(P.pose), *gpoint = uglobal(P.point);
    MS_GLOBAL double *Sg = (MS_GLOBAL double *)P.S;
    const bool by_pts = F.by_points != 0;
    const double huber_delta = P.huber;
    if (by_pts) {
        // the passes own points, not rows: S <- Hpp + lambda I (envelope part; zeros from the 16-row block's envelope up to it) and y <- bp first, by the whole
        // team, then every pass takes its points' products off them with atomics
        for (int row = gw; row < n; row += GW) {
            const int fa = row / 6, c0 = cs[fa], len = 6 * fa + 6 - c0, z0 = env[row >> 4] & ~15;
            const MS_GLOBAL double *hrow = Hpp + (size_t)row * n + c0;
            MS_GLOBAL double *srow = Sg + (size_t)row * n;
            const int h0 = hcs[fa];
            for (int c = z0 + lane; c < h0; c += 64) srow[c] = 0.0;             // (Hpp holds nothing left of the row's first edge block: not read)
            for (int c = h0 - c0 + lane; c < len; c += 64) srow[c0 + c] = hrow[c] + (c0 + c == row ? lambda : 0.0);
        }
        for (int i = gt; i < n; i += GT) P.y[i] = P.bp[i];
        team_sync(P);
    }
    // One workgroup per window: the 3 x 3 factor Hll + lambda I = L L^T and L^-1 bl once per POINT (into Hinv / dl, both unused by the fused path at this point; the
    // first pass's barrier below orders them) instead of once per observation inside the batches -- three square roots and three divisions, ~80 dependent
    // instructions that every one of a point's ~10 lanes repeated.  A team keeps them in the lanes: its passes would need a team barrier in between.
    const bool pre = T_ == 1;
    if (pre) {
        for (int l = tid; l < P.n_point; l += NT) {
            if (P.point_fixed && P.point_fixed[l]) continue;
            double H[6];
            load6(P.Hll + 6 * (size_t)l, H);
            const double b0 = P.bl[3 * (size_t)l], b1 = P.bl[3 * (size_t)l + 1], b2 = P.bl[3 * (size_t)l + 2];
            const double a = H[0] + lambda, d = H[3] + lambda, f = H[5] + lambda;
            const double i11 = rsqrt_d(a), l21 = H[1] * i11, l31 = H[2] * i11;
            const double d2 = d - l21 * l21, i22 = rsqrt_d(d2), l32 = (H[4] - l31 * l21) * i22;
            const double d3 = f - l31 * l31 - l32 * l32, i33 = rsqrt_d(d3);
            if (!(a > 0) || !(d2 > 0) || !(d3 > 0) || !isfinite(i11 * i22 * i33)) P.flag[0] = 0;
            const double u0 = b0 * i11, u1 = (b1 - l21 * u0) * i22, u2 = (b2 - l31 * u0 - l32 * u1) * i33;
            const double Li[6] = {i11, l21, l31, i22, l32, i33};
            store6(P.Hinv + 6 * (size_t)l, Li);
            P.dl[3 * (size_t)l] = u0; P.dl[3 * (size_t)l + 1] = u1; P.dl[3 * (size_t)l + 2] = u2;
        }
    }
    for (int pass = rank_; pass < F.n_pass; pass += T_) {
        const int r0 = F.row0[pass], r1 = F.row1[pass], yoff = F.yoff[2 * pass];
        const MS_GLOBAL int32_t *rowoff = (const MS_GLOBAL int32_t *)F.rowoff + F.yoff[2 * pass + 1] - r0;      // rowoff[fa] for the rows of this pass
        const long long tp0 = clock64();
        // tile <- Hpp (the envelope part of the pass's rows) + lambda I, rhs segment <- bp (or zeros: a pass that owns points only collects their sums).  Hpp is its diagonal
        // blocks and the blocks of the pose-pose edges, nothing else: the tile is cleared in LDS and only the row pieces from the first such block on are fetched, 16 lanes
        // per row (round 3 copied the whole envelope row by row, each row two dependent round trips: 47 k cycles per pass for 80 KB that are mostly zeros)
        const int tile_used = yoff + 6 * (r1 - r0);
        for (int i = tid; i < tile_used; i += NT) tile[i] = (i >= yoff && !by_pts) ? P.bp[6 * r0 + i - yoff] : 0.0;
        if (pose_lds) for (int i = tid; i < 7 * P.n_pose; i += NT) ptab[i] = gpose[i];
        for (int i = tid; i < r1 - r0; i += NT) rowtab[i] = i4_t{cs[r0 + i], rowoff[r0 + i], hcs[r0 + i], 0};
        if (tid < NW) {                                            // the pass's batches in NW contiguous ranges
            const int pbs = F.batch_start[pass], nbp = F.batch_start[pass + 1] - pbs;
            const unsigned lo = (unsigned)(pbs + (long long)nbp * tid / NW), hi = (unsigned)(pbs + (long long)nbp * (tid + 1) / NW);
            s_fs_q[tid] = (unsigned long long)lo | ((unsigned long long)hi << 32);
        }
        __syncthreads();
        if (!by_pts) {
            for (int it = tid >> 4; it < 6 * (r1 - r0); it += NT / 16) {
                const int f = it / 6, i = it - 6 * f, fa = r0 + f, row = 6 * fa + i;
                const i4_t rt = rowtab[f];
                const int len = 6 * fa + 6 - rt.x;
                MS_LDS double *trow = tile + rt.y + i * len - rt.x;                 // trow[c]: scalar column c of the row
                const MS_GLOBAL double *hrow = Hpp + (size_t)row * n;
                for (int c = rt.z + (tid & 15); c < 6 * fa + 6; c += 16) trow[c] = hrow[c] + (c == row ? lambda : 0.0);
            }
            __syncthreads();
        }
        const long long tp1 = clock64();
        cyc[5] += tp1 - tp0;
        // Every wave owns a contiguous range of the pass's batches: the pass's points are sorted by their set of poses, so consecutive batches mostly repeat the
        // same blocks and a lane keeps the sum of "its" block in registers from batch to batch -- the block goes to the tile (36 LDS fp64 atomics per lane, ~2 lanes
        // per cycle) only when the lane's block changes.  A wave that has finished its range takes batches off the END of another wave's (the waves of a SIMD do not
        // run at the same speed: the older one has priority, 27 against 19 batches of a C4 pass), so the owner's run stays contiguous.  (Round 3 handed out runs of
        // three batches from one counter: a wave's consecutive runs were then far apart, every run started with a flush, and the flushes -- with two dependent global
        // loads for the row's tile offset in front of them -- were 3 k of the 8 k cycles a batch's products took.)
        const int pb1 = F.batch_start[pass + 1];
        double acc[36];
#pragma unroll
        for (int q = 0; q < 36; ++q) acc[q] = 0;
        int key = -1;                                              // fa << 16 | fb of the block held in acc
        double yacc[6] = {0, 0, 0, 0, 0, 0};                       // the lane's running part of the rhs of pose ykey
        int ykey = -1;
        // runs per hand-out: a row pass of a team has ~75 batches per workgroup (runs of 1 / 2 / 3 / 4 / 5 / 7 gave 2.22 / 2.08 / 2.00 / 1.99 / 1.96 / 1.98 ms per C4 window:
        // fewer block flushes against a longer tail); a pass that owns points has ~10, one at a time
        FSP_DECL
        // One stream of batches per wave, every batch's operands requested while the batch before it is worked on (round 4: with everything fetched at the top of
        // a batch a wave of the 256-window launch waited 5.2 k cycles per batch for memory -- 270 MB of windows stream from HBM in every pass --, 29 % of the pass):
        // the next batch's lane records and run table entry go out before the Jacobians, its point and observation values (15 doubles per lane) before the pair
        // products.  A batch owns FS_OB lane slots of pobs / puv whatever it holds (padding: observation -1), so no address in this chain depends on a loaded value.
        auto next_batch = [&]() -> int {                           // the wave's next batch, or pb1 when the pass has none left for it (the same value in every lane)
            int got = pb1;
            if (lane == 0) {
                // next | end << 32 moves with ONE 64-bit atomic add per attempt (the owner adds 1 to the front, a thief takes 1 off the end): both halves come back from
                // the same instant, so "was there a batch left" needs no compare-and-swap loop; attempts on an empty range overshoot it, which only keeps it empty
                for (int k = 0; k < NW && got == pb1; ++k) {
                    const int v = (wave + (k & 1) * (NW / 2) + (k >> 1)) % NW;      // itself first, then its SIMD partner, then the others
                    const unsigned long long old = atomicAdd(&s_fs_q[v], k == 0 ? 1ull : 0xFFFFFFFF00000000ull);
                    const int h = (int)(unsigned)old, t = (int)(unsigned)(old >> 32);
                    if (h < t) got = k == 0 ? h : t - 1;
                }
            }
            return __builtin_amdgcn_readfirstlane(got);
        };
        struct LaneData { double X[3], bl[3], H[6], uv[2], info; };
        const MS_GLOBAL double *hsrc = uglobal(pre ? P.Hinv : P.Hll), *blsrc = uglobal(pre ? P.dl : P.bl);
        // (every load of the chain is unconditional -- padding slots name point 0 / pose 0, a batch index past the pass is clamped into it: a load inside a branch
        //  turns the compiler's wait for any LATER value into a wait for everything in flight, and the chain collapses into one round trip per batch)
        auto fetch_lane = [&](const i4_t &r, int slot, LaneData &d) {
            const int l = r.z;
#pragma unroll
            for (int q = 0; q < 3; ++q) { d.X[q] = gpoint[3 * l + q]; d.bl[q] = blsrc[3 * l + q]; }
            const MS_GLOBAL d2_t *h2 = (const MS_GLOBAL d2_t *)(hsrc + 6 * l);
#pragma unroll
            for (int q = 0; q < 3; ++q) { const d2_t u = h2[q]; d.H[2 * q] = u.x; d.H[2 * q + 1] = u.y; }
            const d2_t a = puv[2 * slot];
            d.uv[0] = a.x; d.uv[1] = a.y; d.info = ((const MS_GLOBAL double *)puv)[4 * slot + 2];       // (8 bytes, not the whole 16: a destination register nobody reads is re-used at once, and that write waits for the load)
        };
        int b = next_batch();
        i4_t rec = {-1, 0, 0, -1};
        LaneData cur = {};
        int fmt = 0, run0 = 0, run1 = 0;
        if (b < pb1) { rec = pobs4[FS_OB * b + lane]; fmt = b_fmt[b]; run0 = b_run[b]; run1 = b_run[b + 1]; fetch_lane(rec, FS_OB * b + lane, cur); }
        __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0), as an instruction the compiler's counter tracking sees: otherwise the loop's first use of run0 waits for "everything" in EVERY
                                                                   // iteration (the first batch's lane values are younger than it), and with it for the loads the loop has just issued
        FSP(0);
        while (b < pb1) {
            // this batch's lane values were requested before the previous batch's pair products: they are here.  Saying so BEFORE the next loads go out keeps the compiler's
            // wait in front of the Jacobians from covering those as well (the loads below sit in branches, so it would wait for "all of them")
            __builtin_amdgcn_s_waitcnt(0x0F70);
            const bool act = rec.x >= 0;
            const int pi = rec.y, fa = rec.w;
            const int bn = next_batch();
            double pose[7];
            if constexpr (pose_lds) {
#pragma unroll
                for (int q = 0; q < 7; ++q) pose[q] = ptab[7 * pi + q];
            } else {
#pragma unroll
                for (int q = 0; q < 7; ++q) pose[q] = gpose[7 * pi + q];
            }
            const int run_lo = run0 >> 3, run_hi = fmt ? run_lo : (run1 >> 3);
            u4_t pk = chunks[min(run_lo + lane, max(run_hi - 1, run_lo))];
            if (!(run_lo + lane < run_hi)) pk = u4_t{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            unsigned single = pairs16[run0 + min(lane, max(fmt - 1, 0))];
            if (!(fmt > 0 && lane < fmt)) single = 0xFFFFu;
            const int bc = min(bn, pb1 - 1);
            i4_t rec_n = pobs4[FS_OB * bc + lane];
            const int fmt_n = b_fmt[bc], run0_n = b_run[bc], run1_n = b_run[bc + 1];
            if (bn >= pb1) rec_n.x = -1;
            FSP_WAIT_VM; FSP(1);
            const double (&X)[3] = cur.X, (&blv)[3] = cur.bl, (&H)[6] = cur.H, (&uv)[2] = cur.uv;
            const double info = cur.info;
            if (act) {
                double e[2], Jp[12], Jl[6];
                proj_edge<true>(pose, X, uv, e, Jp, Jl);
                const double chi2 = info * (e[0] * e[0] + e[1] * e[1]);
                double rho, w;
                huber(chi2, huber_delta, rho, w);
                const double wi = w * info;
                // Hll + lambda I = L L^T: one workgroup per window has it from the pre-pass (H = the reciprocal diagonal and the off-diagonal entries of L, blv = L^-1 bl)
                double i11, l21, l31, i22, l32, i33, u0, u1, u2;
                if (pre) { i11 = H[0]; l21 = H[1]; l31 = H[2]; i22 = H[3]; l32 = H[4]; i33 = H[5]; u0 = blv[0]; u1 = blv[1]; u2 = blv[2]; }
                else {
                    const double a = H[0] + lambda, d = H[3] + lambda, f = H[5] + lambda;
                    i11 = rsqrt_d(a); l21 = H[1] * i11; l31 = H[2] * i11;
                    const double d2 = d - l21 * l21;
                    i22 = rsqrt_d(d2); l32 = (H[4] - l31 * l21) * i22;
                    const double d3 = f - l31 * l31 - l32 * l32;
                    i33 = rsqrt_d(d3);
                    if (!(a > 0) || !(d2 > 0) || !(d3 > 0) || !isfinite(i11 * i22 * i33)) P.flag[0] = 0;
                    u0 = blv[0] * i11; u1 = (blv[1] - l21 * u0) * i22; u2 = (blv[2] - l31 * u0 - l32 * u1) * i33;      // L^-1 bl
                }
                // Z = W L^-T with W = wi Jp^T Jl (6 x 3) has rank 2: Z = Jp^T G, G = wi Jl L^-T (2 x 3).  The slab keeps the factors -- 14 doubles per observation instead of
                // 18, 18 operations here instead of 90 -- and a block product becomes Jp_a^T (G_a G_b^T) Jp_b: 92 multiply-adds instead of 108 (Jp has two structural zeros)
                double G[6], sr[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const double w0 = wi * Jl[3 * r], w1 = wi * Jl[3 * r + 1], w2 = wi * Jl[3 * r + 2];
                    const double g0 = w0 * i11, g1 = (w1 - g0 * l21) * i22, g2 = (w2 - g0 * l31 - g1 * l32) * i33;                     // row r of wi Jl L^-T
                    G[3 * r] = g0; G[3 * r + 1] = g1; G[3 * r + 2] = g2;
                    sr[r] = g0 * u0 + g1 * u1 + g2 * u2;
                }
                MS_LDS d2_t *ent = (MS_LDS d2_t *)(stage + lane * FS_ZD);
                ent[0] = d2_t{G[0], G[1]}; ent[1] = d2_t{G[2], G[3]}; ent[2] = d2_t{G[4], G[5]};
                ent[3] = d2_t{Jp[0], Jp[1]}; ent[4] = d2_t{Jp[2], Jp[3]}; ent[5] = d2_t{Jp[5], Jp[6]}; ent[6] = d2_t{Jp[8], Jp[11]};      // (Jp[4] = Jp[9] = 0, Jp[7] = -Jp[0], Jp[10] = Jp[3])
                if (fa >= r0 && fa < r1) {                                    // W (Hll + lambda I)^-1 bl = Jp^T (G L^-1 bl), summed in the lane while its pose stays the same (lane = point x pose slot)
                    if (fa != ykey) {
                        if (ykey >= 0) {
#pragma unroll
                            for (int r = 0; r < 6; ++r) lds_sub(tile + yoff + 6 * (ykey - r0) + r, yacc[r]);
                        }
#pragma unroll
                        for (int r = 0; r < 6; ++r) yacc[r] = 0;
                        ykey = fa;
                    }
#pragma unroll
                    for (int r = 0; r < 6; ++r) yacc[r] = fma(Jp[r], sr[0], fma(Jp[6 + r], sr[1], yacc[r]));
                }
                meta[lane] = fa;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            FSP(2);
            LaneData nxt;
            fetch_lane(rec_n, FS_OB * bc + lane, nxt);
            FSP(6);
            auto take_block = [&](unsigned ab) {                      // the block of pair ab becomes the lane's current one: the old sum goes out first
                const int k2 = (meta[ab & 255u] << 16) | meta[ab >> 8];
                if (k2 != key) {
                    if (key >= 0) {
                        const int fa2 = key >> 16, fb2 = key & 0xFFFF;
                        const i4_t rt = rowtab[fa2 - r0];
                        const int len = 6 * fa2 + 6 - rt.x;
                        MS_LDS double *blk = tile + rt.y + 6 * fb2 - rt.x;
#pragma unroll
                        for (int i = 0; i < 6; ++i)
#pragma unroll
                            for (int j = 0; j < 6; ++j) { if (!(MS_FS_ABL & 8)) lds_sub(blk + i * len + j, acc[6 * i + j]); acc[6 * i + j] = 0; }
                    }
                    key = k2;
                }
            };
            auto add_pair = [&](unsigned ab) {
                const MS_LDS d2_t *za = (const MS_LDS d2_t *)(stage + (ab & 255u) * FS_ZD), *zb = (const MS_LDS d2_t *)(stage + (ab >> 8) * FS_ZD);
                double A[FS_ZD], B[FS_ZD];
#if (MS_FS_ABL & 3) == 2          /* timing ablation (wrong results): no slab reads */
#pragma unroll
                for (int q = 0; q < FS_ZD; ++q) { A[q] = acc[q] * 1e-300; B[q] = acc[q + 14] * 1e-300; }
                (void)za; (void)zb;
#else
#pragma unroll
                for (int q = 0; q < FS_ZD / 2; ++q) { const d2_t u = za[q], v = zb[q]; A[2 * q] = u.x; A[2 * q + 1] = u.y; B[2 * q] = v.x; B[2 * q + 1] = v.y; }
#endif
#if (MS_FS_ABL & 3) == 1          /* timing ablation (wrong results): slab reads, no products */
#pragma unroll
                for (int q = 0; q < FS_ZD; ++q) acc[q] += A[q] + B[q];
                return;
#endif
                double M[4], T0[6], T1[6];                                 // M = G_a G_b^T, T = M Jp_b
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c) M[2 * r + c] = A[3 * r] * B[3 * c] + A[3 * r + 1] * B[3 * c + 1] + A[3 * r + 2] * B[3 * c + 2];
                const double *qa = A + 6, *qb = B + 6;                     // Jp row 0 = (q0, q1, q2, q3, 0, q4), row 1 = (q5, -q0, q6, 0, q3, q7)
                T0[0] = M[0] * qb[0] + M[1] * qb[5]; T0[1] = M[0] * qb[1] - M[1] * qb[0]; T0[2] = M[0] * qb[2] + M[1] * qb[6]; T0[3] = M[0] * qb[3]; T0[4] = M[1] * qb[3]; T0[5] = M[0] * qb[4] + M[1] * qb[7];
                T1[0] = M[2] * qb[0] + M[3] * qb[5]; T1[1] = M[2] * qb[1] - M[3] * qb[0]; T1[2] = M[2] * qb[2] + M[3] * qb[6]; T1[3] = M[2] * qb[3]; T1[4] = M[3] * qb[3]; T1[5] = M[2] * qb[4] + M[3] * qb[7];
#pragma unroll
                for (int j = 0; j < 6; ++j) {                               // (chained: `acc += x + y` is a multiply, a multiply-add and an add -- contraction does not reassociate)
                    acc[j] = fma(qa[0], T0[j], fma(qa[5], T1[j], acc[j]));
                    acc[6 + j] = fma(qa[1], T0[j], fma(-qa[0], T1[j], acc[6 + j]));
                    acc[12 + j] = fma(qa[2], T0[j], fma(qa[6], T1[j], acc[12 + j]));
                    acc[18 + j] = fma(qa[3], T0[j], acc[18 + j]);
                    acc[24 + j] = fma(qa[3], T1[j], acc[24 + j]);
                    acc[30 + j] = fma(qa[4], T0[j], fma(qa[7], T1[j], acc[30 + j]));
                }
            };
            if (PROCEDURAL && fmt < 0) {
                // equal pose sets: block q of the batch is (slot a, slot b2), a0 <= a < k, b2 <= a, in that order; its G pairs (one per point) are cut into pieces of
                // `cap` so that the pieces of all blocks spread over the lanes -- the order and the cut the host's pair lists had, without the lists
                const int enc = -1 - fmt, G = enc & 127, kq = (enc >> 7) & 31, a0q = enc >> 12;
                const int nblk = kq * (kq + 1) / 2 - a0q * (a0q + 1) / 2, n_pairs = G * nblk;
                const int cap = min(8, max(1, (n_pairs + 63) >> 6)), ppb = (G + cap - 1) / cap;          // pairs per piece, pieces per block
                for (int piece = lane; piece < nblk * ppb; piece += 64) {
                    const int q = piece / ppb, g0 = (piece - q * ppb) * cap, g1 = min(G, g0 + cap);
                    int a = a0q, t2 = q;
                    while (t2 > a) { t2 -= a + 1; ++a; }                      // q-th block: row a, column t2
                    for (int gp = g0; gp < g1; ++gp) {
                        const unsigned ab = (unsigned)(gp * kq + a) | ((unsigned)(gp * kq + t2) << 8);
                        if (gp == g0) take_block(ab);
                        add_pair(ab);
                    }
                }
            }
            if (single != 0xFFFFu) { take_block(single); add_pair(single); }
            FSP(7);
            for (int run = run_lo + lane; run < run_hi; run += 64) {
                u4_t nx = chunks[min(run + 64, run_hi - 1)];
                if (!(run + 64 < run_hi)) nx = u4_t{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                take_block(pk.x & 0xFFFFu);
#pragma unroll 1
                for (int k = 0; k < 8; ++k) {
                    const unsigned wd = k < 2 ? pk.x : (k < 4 ? pk.y : (k < 6 ? pk.z : pk.w));
                    const unsigned ab = (k & 1) ? wd >> 16 : wd & 0xFFFFu;
                    if (ab == 0xFFFFu) break;
                    add_pair(ab);
                }
                pk = nx;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            FSP(3);
#ifdef MS_FS_PROF
            fsacc[5] += 1;
#endif
            rec = rec_n; cur = nxt; fmt = fmt_n; run0 = run0_n; run1 = run1_n; b = bn;
        }
        if (key >= 0) {
            const int fa2 = key >> 16, fb2 = key & 0xFFFF;
            const i4_t rt = rowtab[fa2 - r0];
            const int len = 6 * fa2 + 6 - rt.x;
            MS_LDS double *blk = tile + rt.y + 6 * fb2 - rt.x;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) lds_sub(blk + i * len + j, acc[6 * i + j]);
        }
        if (ykey >= 0) {
#pragma unroll
            for (int r = 0; r < 6; ++r) lds_sub(tile + yoff + 6 * (ykey - r0) + r, yacc[r]);
        }
        FSP(4); FSP_FLUSH;
        const long long tp2 = clock64();
        __syncthreads();
        cyc[6] += clock64() - tp2;
        // flush: the rows' envelope part to S, zeros between the 16-row block's envelope (where the factorisation starts reading) and it
        for (int rr = wave; rr < 6 * (r1 - r0); rr += NW) {
            const int fa = r0 + rr / 6, i = rr - 6 * (rr / 6), row = 6 * fa + i, c0 = cs[fa], len = 6 * fa + 6 - c0, z0 = env[row >> 4] & ~15;
            const MS_LDS double *trow = tile + rowoff[fa] + i * len;
            MS_GLOBAL double *srow = Sg + (size_t)row * n;
            if (by_pts) { for (int c = lane; c < len; c += 64) { const double v = trow[c]; if (v != 0) atomicAdd(P.S + (size_t)row * n + c0 + c, v); } continue; }
            for (int c = z0 + lane; c < c0; c += 64) srow[c] = 0.0;
            for (int c = lane; c < len; c += 64) srow[c0 + c] = trow[c];
        }
        for (int i = tid; i < 6 * (r1 - r0); i += NT) {
            if (by_pts) { const double v = tile[yoff + i]; if (v != 0) atomicAdd(&P.y[6 * r0 + i], v); }
            else P.y[6 * r0 + i] = tile[yoff + i];
        }
        __syncthreads();
    }
    team_sync(P);
}

// dl = (Hll + lambda I)^-1 (bl - sum_a W_a^T dp_a) with W_a = w info Jp^T Jl recomputed from the (unchanged) state:
// W_a^T x = w info (Jl_row0 (Jp_row0 . x) + Jl_row1 (Jp_row1 . x))
__device__ __noinline__ void point_backsub_fused(const BaProb &P_, double lambda, double *lds_) {
    const BaProb &P = P_;
    BA_IDS
    // a team: a thread per OBSERVATION again (like the linearisation): workgroup r owns the points [n r / T, n (r+1) / T), starts an LDS table
    // from their bl, every observation takes its W^T dp off it with ds_add_f64, then a thread per point solves the 3 x 3 system
    const int bs_l0 = T_ > 1 ? (int)((long long)P.n_point * rank_ / T_) : 0, bs_l1 = T_ > 1 ? (int)((long long)P.n_point * (rank_ + 1) / T_) : 0;
    const bool obs_par = T_ > 1 && 3 * (size_t)((P.n_point + T_ - 1) / T_ + 1) <= kLdsBytes / 8;
    if (obs_par) {
        MS_LDS double *rtab = (MS_LDS double *)lds_;
        for (int i = tid; i < 3 * (bs_l1 - bs_l0); i += NT) rtab[i] = P.bl[3 * (size_t)bs_l0 + i];
        __syncthreads();
        const int o_lo = P.pt_start[bs_l0], o_hi = P.pt_start[bs_l1];
        for (int idx = o_lo + tid; idx < o_hi; idx += NT) {
            const int o = P.pt_obs[idx], pi = P.obs_pose[o], l = P.obs_point[o], fa = P.pidx[pi];
            if (fa < 0 || (P.point_fixed && P.point_fixed[l])) continue;       // an observation from a fixed pose moves nothing here
            double pose[7], X[3], uv[2], x[6];
#pragma unroll
            for (int q = 0; q < 7; ++q) pose[q] = P.pose[7 * (size_t)pi + q];
#pragma unroll
            for (int q = 0; q < 3; ++q) X[q] = P.point[3 * (size_t)l + q];
            uv[0] = P.obs_uv[2 * (size_t)o]; uv[1] = P.obs_uv[2 * (size_t)o + 1];
            const double info = P.obs_info[o];
            load6(P.dp + 6 * fa, x);
            double e[2], Jp[12], Jl[6];
            proj_edge<true>(pose, X, uv, e, Jp, Jl);
            const double chi2 = info * (e[0] * e[0] + e[1] * e[1]);
            double rho, w;
            huber(chi2, P.huber, rho, w);
            const double wi = w * info;
            double s0 = 0, s1 = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) { s0 += Jp[a] * x[a]; s1 += Jp[6 + a] * x[a]; }
#pragma unroll
            for (int c = 0; c < 3; ++c) lds_addd(rtab + 3 * (l - bs_l0) + c, -(wi * (Jl[c] * s0 + Jl[3 + c] * s1)));
        }
        __syncthreads();
        for (int l = bs_l0 + tid; l < bs_l1; l += NT) {
            double *dq = P.dl + 3 * (size_t)l;
            if (P.point_fixed && P.point_fixed[l]) { dq[0] = dq[1] = dq[2] = 0; continue; }
            const double r[3] = {rtab[3 * (l - bs_l0)], rtab[3 * (l - bs_l0) + 1], rtab[3 * (l - bs_l0) + 2]};
            double H[6];
            load6(P.Hll + 6 * (size_t)l, H);
            const double a = H[0] + lambda, b = H[1], c = H[2], d = H[3] + lambda, e2 = H[4], f = H[5] + lambda;
            const double A = d * f - e2 * e2, B = c * e2 - b * f, C = b * e2 - c * d;
            const double id = 1.0 / (a * A + b * B + c * C);
            const double h0 = A * id, h1 = B * id, h2 = C * id, h3 = (a * f - c * c) * id, h4 = (b * c - a * e2) * id, h5 = (a * d - b * b) * id;
            dq[0] = h0 * r[0] + h1 * r[1] + h2 * r[2];
            dq[1] = h1 * r[0] + h3 * r[1] + h4 * r[2];
            dq[2] = h2 * r[0] + h4 * r[1] + h5 * r[2];
        }
        team_sync(P);
        return;
    }
    for (int l = gts; l < P.n_point; l += GT) {
        double *dq = P.dl + 3 * (size_t)l;
        if (P.point_fixed && P.point_fixed[l]) { dq[0] = dq[1] = dq[2] = 0; continue; }
        double r[3] = {P.bl[3 * (size_t)l], P.bl[3 * (size_t)l + 1], P.bl[3 * (size_t)l + 2]};
        const double X[3] = {P.point[3 * (size_t)l], P.point[3 * (size_t)l + 1], P.point[3 * (size_t)l + 2]};
        const int iend = P.pt_start[l + 1];
        int ii = P.pt_start[l], o1, pi1;
        PtObs cur;
        ptobs_idx(P, ii, iend, o1, pi1); ptobs_data(P, o1, pi1, cur); ptobs_idx(P, ii + 1, iend, o1, pi1);
        int fa = cur.o >= 0 ? P.pidx[cur.pi] : -1;
        for (; ii < iend; ++ii) {
            PtObs nxt;
            ptobs_data(P, o1, pi1, nxt);
            const int fa_n = o1 >= 0 ? P.pidx[pi1] : -1;
            int o2, pi2;
            ptobs_idx(P, ii + 2, iend, o2, pi2);
            double x[6] = {0, 0, 0, 0, 0, 0};
            if (fa >= 0) load6(P.dp + 6 * fa, x);
            double e[2], Jp[12], Jl[6];
            proj_edge<true>(cur.pose, X, cur.uv, e, Jp, Jl);
            const double info = cur.info, chi2 = info * (e[0] * e[0] + e[1] * e[1]);
            double rho, w;
            huber(chi2, P.huber, rho, w);
            const double wi = fa >= 0 ? w * info : 0.0;                  // an observation from a fixed pose moves nothing here
            double s0 = 0, s1 = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) { s0 += Jp[a] * x[a]; s1 += Jp[6 + a] * x[a]; }
#pragma unroll
            for (int c = 0; c < 3; ++c) r[c] -= wi * (Jl[c] * s0 + Jl[3 + c] * s1);
            cur = nxt; o1 = o2; pi1 = pi2; fa = fa_n;
        }
        double H[6];
        load6(P.Hll + 6 * (size_t)l, H);
        const double a = H[0] + lambda, b = H[1], c = H[2], d = H[3] + lambda, e2 = H[4], f = H[5] + lambda;
        const double A = d * f - e2 * e2, B = c * e2 - b * f, C = b * e2 - c * d;
        const double id = 1.0 / (a * A + b * B + c * C);
        const double h0 = A * id, h1 = B * id, h2 = C * id, h3 = (a * f - c * c) * id, h4 = (b * c - a * e2) * id, h5 = (a * d - b * b) * id;
        dq[0] = h0 * r[0] + h1 * r[1] + h2 * r[2];
        dq[1] = h1 * r[0] + h3 * r[1] + h4 * r[2];
        dq[2] = h2 * r[0] + h4 * r[1] + h5 * r[2];
    }
    team_sync(P);
}

typedef double d4_t __attribute__((ext_vector_type(4)));

// Left-looking update of the 16-column panel at c0 with the finished columns [0, c0), row tiles tile0, tile0 + tstride, ...
// (two per trip): result rows go to pan[(row - c0) * NB + col], in LDS for the single-workgroup factorisation, in global
// memory when the tiles of a panel are spread over a team.
template <bool LIST>   // LIST: walk the host-built list of the panel's active row tiles instead of testing every tile's envelope (large systems)
__device__ __forceinline__ void chol_panel_update(const BaProb &P, int c0, int nb, int m, double *pan, int tile0, int tstride, int lane) {
    const int n = P.n6;
    // left-looking update of the 16-column panel with the finished columns [0, c0): a dense
    // (m x c0)(c0 x 16) product -> v_mfma_f64_16x16x4_f64, one 16-row tile per wave, operands straight
    // from L2 (each lane streams 4 consecutive doubles of its A row and of its B row per 16-k chunk;
    // the k order inside a chunk is permuted identically for A and B, which leaves the sum unchanged)
    // Each wave works on TWO row tiles at once (independent accumulators) and two k-chunks per trip, so 16 double2
    // loads are in flight per group of 16 MFMAs instead of 4 per 4.
    const MS_GLOBAL double *Sg = (const MS_GLOBAL double *)P.S, *yg = (const MS_GLOBAL double *)P.y, *zg = (const MS_GLOBAL double *)P.zrow;
    // Envelope: rows of a 16-row block have no entries left of env[block] (and Cholesky creates none), so a row tile whose
    // envelope starts right of this panel is skipped outright and the k loop of the others starts at the later of the two
    // envelopes.  Skipped rows never enter the panel buffer; the substitution and the write-back skip them the same way.
    const MS_GLOBAL int32_t *env = (const MS_GLOBAL int32_t *)P.env16;
    const int pblk = c0 / NB, penv = env[pblk];
    const int q = lane >> 4, jb = lane & 15;
    const MS_GLOBAL d2_t *bp2 = reinterpret_cast<const MS_GLOBAL d2_t *>(((jb < nb) ? Sg + (size_t)(c0 + jb) * n : zg) + 4 * q);
    const MS_GLOBAL int32_t *al = LIST ? (const MS_GLOBAL int32_t *)P.act_blk + P.act_start[pblk] : nullptr;
    const int n_act = LIST ? P.act_start[pblk + 1] - P.act_start[pblk] : 0, beyond = (m + 15) / 16 + 1;
    for (int jt = tile0; LIST ? jt < n_act : jt * 16 < m; jt += 2 * tstride) {
        const int rt = LIST ? al[jt] : jt, rt2 = LIST ? (jt + tstride < n_act ? al[jt + tstride] : beyond) : jt + tstride;
        const int e1 = env[pblk + rt], e2 = rt2 * 16 < m ? env[pblk + rt2] : 0x7fffffff;
        const bool act1 = e1 <= c0 + NB - 1, act2 = e2 <= c0 + NB - 1;
        if (!act1 && !act2) continue;                       // wave-uniform
        const int ia = rt * 16 + (lane & 15), ib = rt2 * 16 + (lane & 15);
        const MS_GLOBAL double *arow = (c0 + ia < n) ? Sg + (size_t)(c0 + ia) * n : (c0 + ia == n ? yg : zg);
        const MS_GLOBAL double *brow2 = (c0 + ib < n) ? Sg + (size_t)(c0 + ib) * n : (c0 + ib == n ? yg : zg);
        const MS_GLOBAL d2_t *ap = reinterpret_cast<const MS_GLOBAL d2_t *>(arow + 4 * q), *ap2 = reinterpret_cast<const MS_GLOBAL d2_t *>(brow2 + 4 * q);
        // the panel's own entries (what the products are subtracted from) are requested first: their round trip runs under the k loop
        double sv[2][4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                const int ii = (which ? rt2 : rt) * 16 + (lane >> 4) + 4 * reg, j = lane & 15;
                sv[which][reg] = (ii < m && j < nb) ? ((c0 + ii < n) ? P.S[(size_t)(c0 + ii) * n + c0 + j] : P.y[c0 + j]) : 0.0;
            }
        d4_t acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
        int kk = max(penv, min(act1 ? e1 : 0x7fffffff, act2 ? e2 : 0x7fffffff)) & ~15;
        for (; kk + 32 <= c0; kk += 32) {
            const d2_t a0 = ap[kk / 2], a1 = ap[kk / 2 + 1], a2 = ap[kk / 2 + 8], a3 = ap[kk / 2 + 9];
            const d2_t e0 = ap2[kk / 2], e1 = ap2[kk / 2 + 1], e2 = ap2[kk / 2 + 8], e3 = ap2[kk / 2 + 9];
            const d2_t b0 = bp2[kk / 2], b1 = bp2[kk / 2 + 1], b2 = bp2[kk / 2 + 8], b3 = bp2[kk / 2 + 9];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b0.x, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(e0.x, b0.x, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b0.y, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(e0.y, b0.y, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b1.x, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(e1.x, b1.x, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b1.y, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(e1.y, b1.y, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.x, b2.x, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(e2.x, b2.x, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.y, b2.y, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(e2.y, b2.y, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a3.x, b3.x, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(e3.x, b3.x, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a3.y, b3.y, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(e3.y, b3.y, acc2, 0, 0, 0);
        }
        for (; kk < c0; kk += 16) {
            const d2_t a0 = ap[kk / 2], a1 = ap[kk / 2 + 1], e0 = ap2[kk / 2], e1 = ap2[kk / 2 + 1], b0 = bp2[kk / 2], b1 = bp2[kk / 2 + 1];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b0.x, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(e0.x, b0.x, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b0.y, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(e0.y, b0.y, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b1.x, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(e1.x, b1.x, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b1.y, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(e1.y, b1.y, acc2, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {       // C/D layout of the f64 MFMA: row = (lane>>4) + 4*reg, col = lane&15
            const int j = lane & 15;
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                const int ii = (which ? rt2 : rt) * 16 + (lane >> 4) + 4 * reg;
                if (ii < m) {
                    pan[ii * NB + j] = sv[which][reg] - (which ? acc2[reg] : acc[reg]);
                }
            }
        }
    }
}

// d -= bcast(s, lane C of each 16-lane row) * t in ONE instruction: gfx950's only 64-bit DPP control is row_newbcast, and v_fmac_f64 takes it --
// the rank-1 updates of the register Cholesky below were two v_readlane + s_nop + v_fma each (4 issue slots and a trip through the SGPR file).
// `fresh`: s was written by the preceding VALU instruction (DPP reads need two wait states after a VALU write; inline asm is opaque to the
// compiler's hazard recogniser).
template <int C, bool FRESH>
__device__ __forceinline__ void fmac_neg_rowbcast(double &d, double s, double t) {
    if constexpr (FRESH) asm("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(s), "v"(t), "n"(C));
    else asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(s), "v"(t), "n"(C));
}
template <int C>
__device__ __forceinline__ double rowbcast_d(double s) {      // lane C of each 16-lane row to the whole row
    double d;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(s), "n"(C));
    return d;
}
template <int J, int... Cs>
__device__ __forceinline__ void chol_rank1(double (&r)[NB], std::integer_sequence<int, Cs...>) {
    (fmac_neg_rowbcast<J + 1 + Cs, Cs == 0>(r[J + 1 + Cs], r[J], r[J]), ...);      // r[c] -= L[c][j] L[i][j], c = j+1 .. 15
}
template <int J>
__device__ __forceinline__ void chol_step(double (&r)[NB], int nb, int ln, double &di, bool &ok) {
    const double d = rowbcast_d<J>(r[J]);                      // the pivot, in every lane
    // positive and finite (normal or denormal), else the block is not positive definite: carry on with 1.0 so that nothing overflows
    const bool okj = __builtin_amdgcn_class(d, 0x180);         // +denormal | +normal
    if (J < nb && !okj) ok = false;
    // 1 / sqrt(d): hardware estimate + two Newton steps (full double precision to ~1 ulp; sqrt + division were 2/3 of this chain)
    const double dd = okj ? d : 1.0;
    double y = __builtin_amdgcn_rsq(dd);
    y = fma(y * 0.5, fma(-dd * y, y, 1.0), y);
    y = fma(y * 0.5, fma(-dd * y, y, 1.0), y);
    r[J] = (ln == J ? dd : r[J]) * y;
    if (ln == J) di = y;
    chol_rank1<J>(r, std::make_integer_sequence<int, NB - 1 - J>{});
}
template <int... Js>
__device__ __forceinline__ void chol_steps(double (&r)[NB], int nb, int ln, double &di, bool &ok, std::integer_sequence<int, Js...>) {
    (chol_step<Js>(r, nb, ln, di, ok), ...);
}

// the 16 x 16 triangle of the back substitution as a lane recurrence, last row first: x_j = rr_j di_j is final in lane j when step j comes
// (col[j] is zero for lanes >= j), and rr -= col[j] x_j takes it as a row broadcast inside the fmac
template <int... Js>
__device__ __forceinline__ void backsub_steps(double &rr, double di, const double (&col)[NB], std::integer_sequence<int, Js...>) {
    (fmac_neg_rowbcast<NB - 1 - Js, true>(rr, rr * di, col[NB - 1 - Js]), ...);
}

// rr -= sum over r of t1[r] * x_r, x_r = lane r of x in each 16-lane row (the tile next to the diagonal in the back substitution)
template <int... Rs>
__device__ __forceinline__ void backsub_adjacent(double &rr, double x, const double (&t1)[NB], std::integer_sequence<int, Rs...>) {
    (fmac_neg_rowbcast<Rs, Rs == 0>(rr, x, t1[Rs]), ...);
}

// the block already in registers (lane ln and its mirrors: row ln); rows / columns from nb on must hold the identity
__device__ __forceinline__ bool chol_factor_regs(double (&r)[NB], int nb, int ln, double &di) {
    bool ok = true;
    di = 1.0;
    asm volatile("s_nop 4");                                   // (an EXEC write by a VALU compare right before the first DPP read would need five wait states)
    chol_steps(r, nb, ln, di, ok, std::make_integer_sequence<int, NB>{});
    return ok;
}

// x L11^T = a for one row per lane (a[] in, x out), L11 as rows in r[] of the lanes of the same 16-lane row, di = 1 / L[ln][ln]: column form,
// x_j = a_j / L_jj, then a_k -= L_kj x_j for k > j with L_kj broadcast inside the fmac -- no operand comes from memory
template <int J, int... Ks>
__device__ __forceinline__ void tri_solve_step(double (&a)[NB], const double (&r)[NB], double di, std::integer_sequence<int, Ks...>) {
    a[J] *= rowbcast_d<J>(di);
    (fmac_neg_rowbcast<J + 1 + Ks, false>(a[J + 1 + Ks], r[J], a[J]), ...);
}
template <int... Js>
__device__ __forceinline__ void tri_solve_rows(double (&a)[NB], const double (&r)[NB], double di, std::integer_sequence<int, Js...>) {
    (tri_solve_step<Js>(a, r, di, std::make_integer_sequence<int, NB - 1 - Js>{}), ...);
}
// c[C] -= sum over j of x_C[j] x[j]: column C of X X^T for the lane's row (x_C = the row held by lane C of the same 16-lane row)
template <int C, int... Js>
__device__ __forceinline__ void syrk_col(double &cc, const double (&x)[NB], std::integer_sequence<int, Js...>) {
    (fmac_neg_rowbcast<C, false>(cc, x[Js], x[Js]), ...);
}
template <int... Cs>
__device__ __forceinline__ void syrk_rows(double (&c)[NB], const double (&x)[NB], std::integer_sequence<int, Cs...>) {
    (syrk_col<Cs>(c[Cs], x, std::make_integer_sequence<int, NB>{}), ...);
}

// Factor the nb x nb diagonal block held as pan[row * LD + col], in the registers of one wave; returns false when a pivot is not
// positive.  On return lane i (and its mirrors i + 16, + 32, + 48) holds row i of L in r[], di = 1 / L[i][i].
template <int LD = NB>
__device__ __forceinline__ bool chol_factor_diag(const double *pan, int nb, int lane, double (&r)[NB], double &di) {
    // factor the nb x nb diagonal block in REGISTERS: lane i holds row i (16 doubles); the pivot and the column entries L[c][j] reach the other
    // lanes as DPP row broadcasts.  Same operations in the same order as the textbook loop over LDS it replaces (26.7 k cycles per block there,
    // ~5 k with v_readlane broadcasts, half of that with the broadcast inside the fmac); a short last block is padded with identity.
    // Only the lower triangle (c <= lane) is meaningful on return: the entries above the diagonal take part in the same
    // instructions unmasked (a mask per column cost more than the whole update) and hold garbage nobody reads.
    const int ln = lane & 15;                                  // the four 16-lane rows of the wave hold the same 16 matrix rows: a row broadcast serves each alike
    if (nb == NB) {
#pragma unroll
        for (int c = 0; c < NB; ++c) r[c] = pan[ln * LD + c];
    } else {
#pragma unroll
        for (int c = 0; c < NB; ++c) r[c] = (ln < nb && c < nb) ? pan[ln * LD + c] : (ln == c ? 1.0 : 0.0);
    }
    return chol_factor_regs(r, nb, ln, di);
}

// Back substitution L^T x = z with xs[n] in LDS (z = the forward-substituted rhs in P.y on entry; P.dp on exit).
__device__ __forceinline__ void chol_back_substitute(const BaProb &P, double *xs, int tid, int lane, int wave) {
    const int n = P.n6;
    // back substitution L^T x = z (z = the forward-substituted rhs, now in P.y), panels in reverse, in LDS.  Per panel: wave 0
    // solves the 16x16 triangle as a lane recurrence (lane k owns x_k; one shuffle and one multiply by the stored reciprocal
    // per step), then every thread k < c0 subtracts the panel's 16 rows from z_k -- rows of L are contiguous in memory, so
    // these are coalesced, independent loads with no reduction (the column-dot formulation needed a wave sum per column).
    for (int i = tid; i < n; i += NT) xs[i] = P.y[i];
    __syncthreads();
    const int last = ((n - 1) / NB) * NB;
    for (int c0 = last; c0 >= 0; c0 -= NB) {
        const int nb = min(NB, n - c0);
        if (tid == NT - 1 && ((c0 / NB) & 7) == 0) team_heartbeat(P);
        if (wave == 0) {
            double col[NB];                                    // lane k: column c0+k of the diagonal block, L[c0+j][c0+k] for j > k
#pragma unroll
            for (int j = 0; j < NB; ++j) col[j] = (j < nb && lane < j) ? P.S[(size_t)(c0 + j) * n + c0 + lane] : 0.0;
            const double di = lane < nb ? P.dinv[c0 + lane] : 0.0;
            double r = lane < nb ? xs[c0 + lane] : 0.0, xk = 0;
#pragma unroll
            for (int j = NB - 1; j >= 0; --j) {
                if (j < nb) {                                  // uniform
                    const double xj = readlane_d(r, j) * readlane_d(di, j);
                    if (lane == j) xk = xj;
                    r -= col[j] * xj;                          // col[j] is zero for lanes >= j
                }
            }
            if (lane < nb) xs[c0 + lane] = xk;
        }
        __syncthreads();
        for (int k = (P.env16[c0 / NB] & ~15) + tid; k < c0; k += NT) {     // columns left of the panel rows' envelope hold zeros
            double zk = xs[k];
#pragma unroll
            for (int j = 0; j < NB; ++j) if (j < nb) zk -= P.S[(size_t)(c0 + j) * n + k] * xs[c0 + j];
            xs[k] = zk;
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += NT) P.dp[i] = xs[i];
    __syncthreads();
}

__device__ __noinline__ void cholesky_solve(const BaProb &P_, double *lds_) {
    const BaProb &P = P_;
    double *lds = lds_;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = P.n6;
    // blocked left-looking Cholesky of S (lower), rhs y carried as row n
    double *pan = lds;                              // [(n+1)][NB]
    double *tvec = lds + (size_t)(n + 1) * NB;      // [NB]
    const MS_GLOBAL int32_t *env = (const MS_GLOBAL int32_t *)P.env16;
    for (int c0 = 0; c0 < n; c0 += NB) {
        const int nb = min(NB, n - c0), m = n - c0 + 1, cnt = m * NB;
        if (tid == 0) team_heartbeat(P);
        chol_panel_update<false>(P, c0, nb, m, pan, wave, NW, lane);
        __syncthreads();
        if (wave == 0) {
            double r[NB], di;
            const bool ok = chol_factor_diag(pan, nb, lane, r, di);
#pragma unroll
            for (int c = 0; c < NB; ++c) if (lane < nb && c < nb) pan[lane * NB + c] = r[c];
            // 1 / L[j][j], once per column: the substitutions below and the back substitution multiply instead of dividing
            if (lane < NB) tvec[lane] = di;
            if (lane < nb) P.dinv[c0 + lane] = di;
            if (!ok && lane == 0) P.flag[0] = 0;
        }
        __syncthreads();
        for (int i = nb + tid; i < m; i += NT) {     // rows below: x L11^T = a
            if (env[(c0 + i) / NB] > c0 + NB - 1) continue;     // outside the envelope: structurally zero, not in the panel
            double x[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (j < nb) {
                    double s = pan[i * NB + j];
                    for (int k = 0; k < j; ++k) s -= x[k] * pan[j * NB + k];
                    x[j] = s * tvec[j];
                }
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) if (j < nb) pan[i * NB + j] = x[j];
        }
        __syncthreads();
        for (int idx = tid; idx < cnt; idx += NT) {
            const int i = idx / NB, j = idx - i * NB;
            if (j < nb && (i >= j || i >= nb) && env[(c0 + i) / NB] <= c0 + NB - 1) {
                if (c0 + i < n) P.S[(size_t)(c0 + i) * n + c0 + j] = pan[idx]; else P.y[c0 + j] = pan[idx];
            }
        }
        __syncthreads();
    }
    chol_back_substitute(P, tvec + NB, tid, lane, wave);
}

// ---------------------------------------------------------------- windowed right-looking Cholesky (the whole front in LDS)
// The left-looking factorisation above feeds every panel update from L2 (operands straight from memory) and spends ~36 k cycles per
// 16-column panel, most of it waiting.  A sliding window's reduced camera matrix is banded (C4: half-bandwidth 60 of 300), so the part
// of the matrix a right-looking factorisation is working on -- the 16-row blocks whose envelope has reached the current panel and
// that are not factored yet -- is a handful of blocks: they are kept as W x W tiles of 16 x 16 doubles in LDS (slots assigned by the
// host, a block keeps its slot while it is active), every tile is read from S once when its later block enters, updated in LDS
// (v_mfma_f64_16x16x4_f64, operands from LDS), and written once when its column is factored.  The rhs is forward-substituted along.
// Stamps inside cholesky_window (build with -DMS_CW_PROF, read with tools/ba_chol_prof.py): per-thread cycle sums in registers, thread 0's and
// thread 64's written out once per call (a stamp that went through memory cost a round trip of its own and drained the prefetches it was meant to time).
#ifdef MS_CW_PROF
__device__ long long g_cwprof[48];
#define CWP_DECL long long cwt = clock64(), cwacc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define CWP(i) do { const long long _t = clock64(); cwacc[i] += _t - cwt; cwt = _t; } while (0)
#define CWP_FLUSH do { if (tid == 0) { for (int i = 0; i < 24; ++i) g_cwprof[i] += cwacc[i]; g_cwprof[7] += 1; } if (tid == 64) { for (int i = 0; i < 24; ++i) g_cwprof[24 + i] += cwacc[i]; } } while (0)
#else
#define CWP_DECL
#define CWP(i) do { } while (0)
#define CWP_FLUSH do { } while (0)
#endif
constexpr int CT_LD = 18;                 // doubles per tile row (16 + 2: the 32-byte operand reads of 16 rows fall into different banks)
constexpr int CT = 16 * CT_LD;            // doubles per tile
// ML: the per-panel index arrays (slots, active blocks, entering tiles) are copied to LDS first -- every step below starts from them, and as global
//     loads their two dependent round trips per step were a fifth of the factorisation of a 50-keyframe window
template <bool ZG, bool ML>   // ZG: the rhs / solution vector stays in global memory (P.y): systems whose n doubles do not fit the LDS beside the tiles (global BA of a long trajectory)
__device__ __noinline__ void cholesky_window(const BaProb &P_, double *lds_) {
    const BaProb &P = P_;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = P.n6, W = P.cw_W, nblk = (n + 15) / 16;
    MS_LDS double *tiles = (MS_LDS double *)lds_;
    typedef typename std::conditional<ZG, MS_GLOBAL double *, MS_LDS double *>::type zptr_t;
    zptr_t z;                                                 // [n] rhs, becomes L^-1 y, then the solution
    MS_LDS double *tvec;                                      // [16] reciprocal pivots of the current panel
    if constexpr (ZG) { z = (MS_GLOBAL double *)P.y; tvec = tiles + W * W * CT; }
    else { z = tiles + W * W * CT; tvec = z + ((n + 15) & ~15); }
    MS_LDS double *LT = tvec + 16 + 16 * W + 32;               // [16][16] L11 transposed (after part and dv, below)
    MS_LDS double *dvec = LT + 256;                            // ML: [n] reciprocal pivots (a global store by the factoring wave would meet the panel's barrier with its round trip still out)
    const MS_GLOBAL double *Sg = (const MS_GLOBAL double *)P.S;
    MS_GLOBAL double *Sw = (MS_GLOBAL double *)P.S;
    const MS_GLOBAL int32_t *gslot = (const MS_GLOBAL int32_t *)P.cw_slot, *gact_start = (const MS_GLOBAL int32_t *)P.cw_act_start, *gact = (const MS_GLOBAL int32_t *)P.cw_act,
                            *gload_start = (const MS_GLOBAL int32_t *)P.cw_load_start, *gloads = (const MS_GLOBAL int32_t *)P.cw_load;
    typedef typename std::conditional<ML, const MS_LDS int32_t *, const MS_GLOBAL int32_t *>::type mptr_t;
    mptr_t slot, act_start, act, load_start, loads;
    if constexpr (ML) {
        MS_LDS int32_t *mb = (MS_LDS int32_t *)(LT + 256 + ((n + 15) & ~15));
        const int nact = gact_start[nblk], nload = gload_start[nblk];
        MS_LDS int32_t *m_slot = mb, *m_as = m_slot + nblk, *m_ls = m_as + nblk + 1, *m_act = m_ls + nblk + 1, *m_ld = m_act + nact;
        for (int i = tid; i < nblk; i += NT) m_slot[i] = gslot[i];
        for (int i = tid; i <= nblk; i += NT) { m_as[i] = gact_start[i]; m_ls[i] = gload_start[i]; }
        for (int i = tid; i < nact; i += NT) m_act[i] = gact[i];
        for (int i = tid; i < 2 * nload; i += NT) m_ld[i] = gloads[i];
        slot = m_slot; act_start = m_as; load_start = m_ls; act = m_act; loads = m_ld;
        __syncthreads();
    } else { slot = gslot; act_start = gact_start; act = gact; load_start = gload_start; loads = gloads; }
    if constexpr (!ZG) for (int i = tid; i < ((n + 15) & ~15); i += NT) z[i] = i < n ? P.y[i] : 0.0;
    if constexpr (ML) { if (tid < 16) dvec[((n + 15) & ~15) - 16 + tid] = 0.0; }      // (the pad of the last block; the pivots overwrite the rest)
    CWP_DECL
    auto fetch_tiles = [&](int pnl, int skip) {               // tiles entering the window at panel pnl: S -> LDS, one tile per wave and trip (waves 1..7; wave 0 factors)
        for (int e = load_start[pnl] + skip + wave - 1; e < load_start[pnl + 1]; e += NW - 1) {
            const int ea = loads[2 * e], eb = loads[2 * e + 1], bi = ea & 0xFFFF, si = ea >> 16, bj = eb & 0xFFFF, sj = eb >> 16;
            MS_LDS double *t = tiles + (si * W + sj) * CT;
            const int c = lane & 15, gc = 16 * bj + c;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = (lane >> 4) + 4 * k, gr = 16 * bi + r;
                t[r * CT_LD + c] = (gr < n && gc < n) ? Sg[(size_t)gr * n + gc] : 0.0;
            }
        }
    };
    // factor the diagonal tile of panel pnl (wave 0): L11 back into the tile, reciprocal pivots to tvec / dinv
    auto factor_diag = [&](int pnl) {
        const int c0 = 16 * pnl, nb = min(16, n - c0), sp = slot[pnl];
        MS_LDS double *Lpp = tiles + (sp * W + sp) * CT;
        double r[NB], di;
        CWP(3);
        const bool ok = chol_factor_diag<CT_LD>((const double *)Lpp, nb, lane, r, di);
        CWP(14);
        // L11 into the tile (the garbage above the diagonal goes along: nothing reads it) and transposed into LT: column j of L11 as a contiguous row for phase A
        if (lane < NB) {
#pragma unroll
            for (int c = 0; c < NB / 2; ++c) reinterpret_cast<MS_LDS d2_t *>(Lpp + lane * CT_LD)[c] = d2_t{r[2 * c], r[2 * c + 1]};
#pragma unroll
            for (int c = 0; c < NB; ++c) LT[c * NB + lane] = r[c];
            tvec[lane] = di;
        }
        if (lane < nb) { if constexpr (ML) dvec[c0 + lane] = di; else P.dinv[c0 + lane] = di; }
        if (!ok && lane == 0) P.flag[0] = 0;
    };
    auto update_pair = [&](int si, int sj, int sp) {         // tile(si, sj) -= tile(si, sp) tile(sj, sp)^T
        const int row = lane & 15, qd = lane >> 4;
        const MS_LDS d2_t *A = reinterpret_cast<const MS_LDS d2_t *>(tiles + (si * W + sp) * CT + row * CT_LD + 4 * qd);
        const MS_LDS d2_t *B = reinterpret_cast<const MS_LDS d2_t *>(tiles + (sj * W + sp) * CT + row * CT_LD + 4 * qd);
        const d2_t a0v = A[0], a1v = A[1], b0v = B[0], b1v = B[1];
        d4_t acc = {0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0v.x, b0v.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0v.y, b0v.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1v.x, b1v.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1v.y, b1v.y, acc, 0, 0, 0);
        MS_LDS double *C = tiles + (si * W + sj) * CT;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) C[(qd + 4 * reg) * CT_LD + row] -= acc[reg];
    };
    if constexpr (!ZG && ML) {
        // Everything the factorisation touches per panel is in LDS or registers.  Wave 0 carries the serial chain in REGISTERS from one diagonal
        // block to the next: with L11(p) still in r[] it solves the tile next to the diagonal, L[p+1,p] (one row per lane, L11 entries as row
        // broadcasts), forms block (p+1,p+1) -= L[p+1,p] L[p+1,p]^T the same way and factors it -- no LDS round trip, no barrier inside the chain.
        // Waves 1, 2 solve the other tiles of the column and the rhs against L11(p) read back from its tile (16 rows per 16-lane row);
        // after the panel's first barrier waves 1..7 do the trailing pairs on the matrix cores, the rhs update, the write-back and the fetch
        // of the tiles that enter next, beside wave 0's update + factor.  Two barriers per panel.
        const int ln = lane & 15, grp = lane >> 4;
        double r[NB], di = 1.0;                                // wave 0: L11 of the current panel, row ln (the same in the four 16-lane rows)
#pragma unroll
        for (int c = 0; c < NB; ++c) r[c] = 0.0;
        auto publish = [&](int pnl, const double (&rr)[NB], double dd, bool ok) {       // L11 -> its tile, reciprocal pivots -> tvec / dvec
            const int c0 = 16 * pnl, nb = min(16, n - c0), sp = slot[pnl];
            MS_LDS double *Lpp = tiles + (sp * W + sp) * CT;
            if (lane < NB) {
#pragma unroll
                for (int c = 0; c < NB / 2; ++c) reinterpret_cast<MS_LDS d2_t *>(Lpp + lane * CT_LD)[c] = d2_t{rr[2 * c], rr[2 * c + 1]};
                tvec[lane] = dd;
                if (lane < nb) dvec[c0 + lane] = dd;
            }
            if (!ok && lane == 0) P.flag[0] = 0;
        };
        auto global_diag_rows = [&](int blk, double (&c)[NB]) {                       // rows of a diagonal block straight from S, identity beyond n
            const int nb1 = min(16, n - 16 * blk);
#pragma unroll
            for (int cc = 0; cc < NB; ++cc) c[cc] = (ln < nb1 && cc < nb1) ? Sg[(size_t)(16 * blk + ln) * n + 16 * blk + cc] : (ln == cc ? 1.0 : 0.0);
        };
        if (nblk > 0 && wave > 0) fetch_tiles(0, 0);
        if (nblk > 0 && wave == 0) {
            global_diag_rows(0, r);
            const bool ok = chol_factor_regs(r, min(16, n), ln, di);
            publish(0, r, di, ok);
        }
        __syncthreads();
        CWP(0);
        for (int p = 0; p < nblk; ++p) {
            if (tid == NT - 1 && (p & 7) == 0) team_heartbeat(P);            // (a lane of the last wave: wave 0 carries the pivot chain)
            const int c0 = 16 * p, nb = min(16, n - c0), sp = slot[p];
            const int a0 = act_start[p], m = act_start[p + 1] - a0;
            const bool has_next = p + 1 < nblk, next_active = m > 0 && (act[a0] & 0xFFFF) == p + 1;
            const int s1 = has_next ? slot[p + 1] : 0, nb1 = has_next ? min(16, n - c0 - 16) : 0;
            double x[NB], cg[NB];                              // wave 0: the solved row of L[p+1,p]; the rows of block p+1 when it only enters now
            double pf[4] = {0, 0, 0, 0};
            int pf_dst = -1;
            if (wave == 0) {
                if (has_next && next_active) {
                    MS_LDS d2_t *row2 = reinterpret_cast<MS_LDS d2_t *>(tiles + (s1 * W + sp) * CT + ln * CT_LD);
#pragma unroll
                    for (int j = 0; j < NB / 2; ++j) { const d2_t v = row2[j]; x[2 * j] = v.x; x[2 * j + 1] = v.y; }
                    tri_solve_rows(x, r, di, std::make_integer_sequence<int, NB>{});
                    if (lane < NB) {
#pragma unroll
                        for (int j = 0; j < NB / 2; ++j) row2[j] = d2_t{x[2 * j], x[2 * j + 1]};
                    }
                    // block (p+1,p+1) has had every earlier panel's update since the last barrier: this panel's goes on top right away, from the rows just
                    // solved -- X X^T on the matrix core with both operands out of x[] (lane (row, g) supplies X[row][4 kk + g]); the result comes back as
                    // four columns of the lane's own row (g, g+4, g+8, g+12: the product is symmetric), is taken off the tile in LDS and the whole row read back
                    {
                        d4_t acc = {0, 0, 0, 0};
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const double xs = grp == 0 ? x[4 * kk] : grp == 1 ? x[4 * kk + 1] : grp == 2 ? x[4 * kk + 2] : x[4 * kk + 3];
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xs, xs, acc, 0, 0, 0);
                        }
                        MS_LDS double *Crow = tiles + (s1 * W + s1) * CT + ln * CT_LD;
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) Crow[grp + 4 * reg] -= acc[reg];
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        const MS_LDS d2_t *C2 = reinterpret_cast<const MS_LDS d2_t *>(Crow);
#pragma unroll
                        for (int j = 0; j < NB / 2; ++j) { const d2_t v = C2[j]; cg[2 * j] = v.x; cg[2 * j + 1] = v.y; }
                    }
                    if (nb1 < NB) {
#pragma unroll
                        for (int cc = 0; cc < NB; ++cc) cg[cc] = (ln < nb1 && cc < nb1) ? cg[cc] : (ln == cc ? 1.0 : 0.0);
                    }
                } else if (has_next) global_diag_rows(p + 1, cg);
            } else {
                // the tiles that enter at the next panel are requested FIRST and parked in registers: their trip to L2 runs beside everything below
                // (slots released a panel ago -- the host delays the reuse -- so the LDS stores at the end of the panel overwrite nothing in use)
                const int le = has_next ? load_start[p + 2] : 0, e0 = has_next ? load_start[p + 1] + wave - 1 : 0;
                if (e0 < le) {
                    const int ea = loads[2 * e0], eb = loads[2 * e0 + 1], bi = ea & 0xFFFF, bj = eb & 0xFFFF, gc = 16 * bj + ln;
                    pf_dst = ((ea >> 16) * W + (eb >> 16)) * CT + grp * CT_LD + ln;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const int gr = 16 * bi + grp + 4 * k; if (gr < n && gc < n) pf[k] = Sg[(size_t)gr * n + gc]; }
                }
                // the other tiles of column p and the rhs: entry e of the panel's list per 16-lane row (e == m: the rhs)
                const int first = next_active ? 1 : 0, e = 4 * (wave - 1) + grp + first;
                if (4 * (wave - 1) + first <= m) {
                    const MS_LDS d2_t *L2 = reinterpret_cast<const MS_LDS d2_t *>(tiles + (sp * W + sp) * CT + ln * CT_LD);
                    double lr[NB], a[NB];
#pragma unroll
                    for (int j = 0; j < NB / 2; ++j) { const d2_t v = L2[j]; lr[2 * j] = v.x; lr[2 * j + 1] = v.y; }
                    const double ldi = tvec[ln];
                    MS_LDS double *row = e < m ? tiles + ((act[a0 + e] >> 16) * W + sp) * CT + ln * CT_LD : (MS_LDS double *)z + c0;
                    MS_LDS d2_t *row2 = reinterpret_cast<MS_LDS d2_t *>(row);
#pragma unroll
                    for (int j = 0; j < NB / 2; ++j) { const d2_t v = row2[j]; a[2 * j] = v.x; a[2 * j + 1] = v.y; }
                    tri_solve_rows(a, lr, ldi, std::make_integer_sequence<int, NB>{});
                    if (e < m || (e == m && ln == 0)) {
#pragma unroll
                        for (int j = 0; j < NB / 2; ++j) row2[j] = d2_t{a[2 * j], a[2 * j + 1]};
                    }
                }
            }
            CWP(1);
            __syncthreads();
            CWP(2);
            if (wave == 0) {
                if (has_next) {
#pragma unroll
                    for (int cc = 0; cc < NB; ++cc) r[cc] = cg[cc];
                    const bool ok = chol_factor_regs(r, nb1, ln, di);
                    publish(p + 1, r, di, ok);
                }
                CWP(4);
            } else {
                // trailing update: pair q of the lower triangle of the m active blocks (row-major: (0,0), (1,0), (1,1), (2,0) ...) goes to wave 1 + q % 7;
                // (0,0) is the next diagonal tile when block p+1 is active already -- wave 0 has it
                const int npair = m * (m + 1) / 2;
                for (int q = wave - 1; q < npair; q += NW - 1) {
                    int i = (int)((__builtin_sqrtf(8.0f * q + 1.0f) - 1.0f) * 0.5f);
                    if ((i + 1) * (i + 2) / 2 <= q) ++i;               // (guard the float root at the triangle's corners)
                    if (i * (i + 1) / 2 > q) --i;
                    const int j = q - i * (i + 1) / 2;
                    if (next_active && q == 0) continue;
                    update_pair(act[a0 + i] >> 16, act[a0 + j] >> 16, sp);
                }
                CWP(19);
                const int t7 = tid - 64, N7 = NT - 64;
                for (int idx = t7; idx < 16 * m; idx += N7) {         // the rhs below the panel: z_b -= L[b,p] z_p
                    const int ea = act[a0 + (idx >> 4)], gr = 16 * (ea & 0xFFFF) + (idx & 15);
                    if (gr >= n) continue;
                    const MS_LDS d2_t *row2 = reinterpret_cast<const MS_LDS d2_t *>(tiles + ((ea >> 16) * W + sp) * CT + (idx & 15) * CT_LD);
                    double sacc = z[gr];
                    d2_t rv[NB / 2];
#pragma unroll
                    for (int k = 0; k < NB / 2; ++k) rv[k] = row2[k];
#pragma unroll
                    for (int k = 0; k < NB; ++k) sacc -= ((k & 1) ? rv[k / 2].y : rv[k / 2].x) * z[c0 + k];
                    z[gr] = sacc;
                }
                CWP(20);
                // column p of L goes out: tile wave-1 (+7 ...) of the panel's m + 1, four rows of 16 per lane and trip
                for (int t = wave - 1; t <= m; t += NW - 1) {
                    const int ea = t == 0 ? (p | (sp << 16)) : act[a0 + t - 1];
                    const MS_LDS double *src = tiles + ((ea >> 16) * W + sp) * CT + grp * CT_LD + ln;
                    double v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = src[4 * k * CT_LD];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int rr = grp + 4 * k, gr = 16 * (ea & 0xFFFF) + rr;
                        if (gr < n && ln < nb && (t > 0 || ln <= rr)) Sw[(size_t)gr * n + c0 + ln] = v[k];
                    }
                }
                CWP(21);
                if (pf_dst >= 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) tiles[pf_dst + 4 * k * CT_LD] = pf[k];
                }
                CWP(22);
                if (has_next) fetch_tiles(p + 1, NW - 1);             // a front that brings in more than seven tiles at once: the rest the plain way
            }
            __syncthreads();
            CWP(5);
        }
    } else {
        if (nblk > 0 && wave > 0) fetch_tiles(0, 0);
        if (nblk > 0 && wave == 0) {                              // panel 0's diagonal tile is wave 0's, like every later one
            MS_LDS double *t = tiles + (slot[0] * W + slot[0]) * CT;
            const int c = lane & 15;
    #pragma unroll
            for (int k = 0; k < 4; ++k) { const int r = (lane >> 4) + 4 * k; t[r * CT_LD + c] = (r < n && c < n) ? Sg[(size_t)r * n + c] : 0.0; }
        }
        __syncthreads();
        if (wave == 0 && nblk > 0) factor_diag(0);
        __syncthreads();
        CWP(0);
        for (int p = 0; p < nblk; ++p) {
            if (tid == NT - 1 && (p & 7) == 0) team_heartbeat(P);
            const int c0 = 16 * p, nb = min(16, n - c0), sp = slot[p];
            MS_LDS double *Lpp = tiles + (sp * W + sp) * CT;
            // A. rows below: x L11^T = a (a thread per row), and the panel's part of the rhs
            const int a0 = act_start[p], m = act_start[p + 1] - a0;
            // (column form: x_j = a_j / L_jj, then a_k -= x_j L_kj for k > j -- the same operations on every a_k in the same order as the dot-product
            //  form, but the 15 - j updates of a step are independent of each other; column j of L11 is row j of LT)
            auto solve_row = [&](double (&a)[NB]) {
                if (nb == NB) {
                    // row j + 1 of LT is requested before step j computes (the scheduler, left alone, put each row's reads right before their use: 16 exposed LDS latencies)
                    const MS_LDS d2_t *LT2 = reinterpret_cast<const MS_LDS d2_t *>(LT), *tv2 = reinterpret_cast<const MS_LDS d2_t *>(tvec);
                    d2_t tv[NB / 2], buf[2][NB / 2];
    #pragma unroll
                    for (int k = 0; k < NB / 2; ++k) tv[k] = tv2[k];
    #pragma unroll
                    for (int k = 0; k < NB / 2; ++k) buf[0][k] = LT2[k];
    #pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        if (j + 1 < NB) {
    #pragma unroll
                            for (int k = (j + 2) / 2; k < NB / 2; ++k) buf[(j + 1) & 1][k] = LT2[(j + 1) * (NB / 2) + k];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        a[j] *= (j & 1) ? tv[j / 2].y : tv[j / 2].x;
    #pragma unroll
                        for (int k = j + 1; k < NB; ++k) a[k] -= a[j] * ((k & 1) ? buf[j & 1][k / 2].y : buf[j & 1][k / 2].x);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {                                          // the short last block
    #pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        if (j < nb) {
                            double sacc = a[j];
    #pragma unroll
                            for (int k = 0; k < j; ++k) sacc -= a[k] * Lpp[j * CT_LD + k];
                            a[j] = sacc * tvec[j];
                        }
                    }
                }
            };
            // (the panel's piece of the rhs is one more row when it lives in LDS: it rides in the wave that has lanes to spare)
            for (int idx = tid; idx < 16 * m + (ZG ? 0 : 1); idx += NT) {
                MS_LDS double *row = tiles;
                if (idx < 16 * m) row = tiles + ((act[a0 + (idx >> 4)] >> 16) * W + sp) * CT + (idx & 15) * CT_LD;
                else if constexpr (!ZG) row = z + c0;
                double x[NB];
                MS_LDS d2_t *row2 = reinterpret_cast<MS_LDS d2_t *>(row);
    #pragma unroll
                for (int j = 0; j < NB / 2; ++j) { const d2_t v = row2[j]; x[2 * j] = v.x; x[2 * j + 1] = v.y; }
                CWP(15);
                solve_row(x);
                CWP(16);
                if (nb == NB) {
    #pragma unroll
                    for (int j = 0; j < NB / 2; ++j) row2[j] = d2_t{x[2 * j], x[2 * j + 1]};
                } else {
    #pragma unroll
                    for (int j = 0; j < NB; ++j) if (j < nb) row[j] = x[j];
                }
                CWP(17);
            }
            if (ZG && tid == NT - 1) {
                double x[NB];
    #pragma unroll
                for (int j = 0; j < NB; ++j) x[j] = j < nb ? z[c0 + j] : 0.0;
                solve_row(x);
    #pragma unroll
                for (int j = 0; j < NB; ++j) if (j < nb) z[c0 + j] = x[j];
            }
            CWP(1);
            __syncthreads();
            CWP(2);
            // B. look-ahead: wave 0 brings the NEXT diagonal tile up to date (its update by this panel, or its first fetch) and factors it at
            //    once -- the serial pivot chain runs beside the trailing update, the rhs update, the write-back of column p and the fetch of the
            //    tiles that enter at the next panel, which the other seven waves share
            const bool next_active = m > 0 && (act[a0] & 0xFFFF) == p + 1;       // block p+1 is in the window already (else it enters at p+1)
            if (wave == 0) {
                if (p + 1 < nblk) {
                    const int s1 = slot[p + 1];
                    if (next_active) update_pair(s1, s1, sp);
                    else {
                        MS_LDS double *t = tiles + (s1 * W + s1) * CT;
                        const int c = lane & 15, gc = 16 * (p + 1) + c;
    #pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int r = (lane >> 4) + 4 * k, gr = 16 * (p + 1) + r;
                            t[r * CT_LD + c] = (gr < n && gc < n) ? Sg[(size_t)gr * n + gc] : 0.0;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    CWP(3);
                    factor_diag(p + 1);
                    CWP(4);
                }
            } else {
                // the tiles that enter at the next panel are requested FIRST and parked in registers: their trip to L2 runs beside the updates
                // and the write-back below (slots released a panel ago -- the host delays the reuse -- so the LDS stores at the end overwrite nothing in use)
                const int le = p + 1 < nblk ? load_start[p + 2] : 0, e0 = p + 1 < nblk ? load_start[p + 1] + wave - 1 : 0;
                double pf[4] = {0, 0, 0, 0};
                int pf_dst = -1;
                if (e0 < le) {
                    const int ea = loads[2 * e0], eb = loads[2 * e0 + 1], bi = ea & 0xFFFF, bj = eb & 0xFFFF, c = lane & 15, gc = 16 * bj + c;
                    pf_dst = ((ea >> 16) * W + (eb >> 16)) * CT + (lane >> 4) * CT_LD + c;
    #pragma unroll
                    for (int k = 0; k < 4; ++k) { const int gr = 16 * bi + (lane >> 4) + 4 * k; if (gr < n && gc < n) pf[k] = Sg[(size_t)gr * n + gc]; }
                }
                CWP(18);
                // trailing update: pair q of the lower triangle of the m active blocks (row-major: (0,0), (1,0), (1,1), (2,0) ...) goes to wave 1 + q % 7;
                // (0,0) is the next diagonal tile when block p+1 is active already -- wave 0 has it
                {
                    const int npair = m * (m + 1) / 2;
                    for (int q = wave - 1; q < npair; q += NW - 1) {
                        int i = (int)((__builtin_sqrtf(8.0f * q + 1.0f) - 1.0f) * 0.5f);
                        if ((i + 1) * (i + 2) / 2 <= q) ++i;               // (guard the float root at the triangle's corners)
                        if (i * (i + 1) / 2 > q) --i;
                        const int j = q - i * (i + 1) / 2;
                        if (next_active && q == 0) continue;
                        update_pair(act[a0 + i] >> 16, act[a0 + j] >> 16, sp);
                    }
                }
                CWP(19);
                const int t7 = tid - 64, N7 = NT - 64;
                for (int idx = t7; idx < 16 * m; idx += N7) {         // the rhs below the panel: z_b -= L[b,p] z_p
                    const int ea = act[a0 + (idx >> 4)], gr = 16 * (ea & 0xFFFF) + (idx & 15);
                    if (gr >= n) continue;
                    const MS_LDS double *row = tiles + ((ea >> 16) * W + sp) * CT + (idx & 15) * CT_LD;
                    double sacc = z[gr];
                    if (!ZG && nb == NB) {
                        const MS_LDS d2_t *row2 = reinterpret_cast<const MS_LDS d2_t *>(row);
                        d2_t rv[NB / 2];
    #pragma unroll
                        for (int k = 0; k < NB / 2; ++k) rv[k] = row2[k];
    #pragma unroll
                        for (int k = 0; k < NB; ++k) sacc -= ((k & 1) ? rv[k / 2].y : rv[k / 2].x) * z[c0 + k];
                    } else {
                        for (int k = 0; k < nb; ++k) sacc -= row[k] * z[c0 + k];
                    }
                    z[gr] = sacc;
                }
                CWP(20);
                // column p of L goes out: tile wave-1 (+7 ...) of the panel's m + 1, four rows of 16 per lane and trip
                for (int t = wave - 1; t <= m; t += NW - 1) {
                    const int ea = t == 0 ? (p | (sp << 16)) : act[a0 + t - 1], c = lane & 15;
                    const MS_LDS double *src = tiles + ((ea >> 16) * W + sp) * CT + (lane >> 4) * CT_LD + c;
                    double v[4];
    #pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = src[4 * k * CT_LD];
    #pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int r = (lane >> 4) + 4 * k, gr = 16 * (ea & 0xFFFF) + r;
                        if (gr < n && c < nb && (t > 0 || c <= r)) Sw[(size_t)gr * n + c0 + c] = v[k];
                    }
                }
                CWP(21);
                if (pf_dst >= 0) {
    #pragma unroll
                    for (int k = 0; k < 4; ++k) tiles[pf_dst + 4 * k * CT_LD] = pf[k];
                }
                CWP(22);
                if (p + 1 < nblk) fetch_tiles(p + 1, NW - 1);     // a front that brings in more than seven tiles at once: the rest the plain way
            }
            if (wave != 0) CWP(23);
            __syncthreads();
            CWP(5);
        }
    }
    // back substitution L^T x = z, panels in reverse, again in LDS: column p of L (the tiles this loop wrote out above) comes back one panel
    // ahead of its use (double buffer in the tile area), a thread per (active block, column) forms L[b,p]^T x_b, wave 0 solves the 16 x 16
    // triangle as a lane recurrence.  Two barriers per panel; the version that reads rows of S from L2 spent a third of the factorisation's time here.
    MS_LDS double *part = tvec + 16;                          // [16 * W] partial dot products, then [2][16] reciprocal pivots
    MS_LDS double *dv = part + 16 * W;
    auto fetch_col = [&](int pnl, int buf, int t0, bool with_dv) {          // tiles t0 + wave, + NW, ... of column pnl
        const int fa0 = act_start[pnl], fm = act_start[pnl + 1] - fa0;
        for (int t = t0 + wave; t <= fm; t += NW) {
            const int b = t == 0 ? pnl : (act[fa0 + t - 1] & 0xFFFF);
            MS_LDS double *dst = tiles + (buf * W + t) * CT;
            const int c = lane & 15, gc = 16 * pnl + c;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = (lane >> 4) + 4 * k, gr = 16 * b + r;
                dst[r * CT_LD + c] = (gr < n && gc < n && (t > 0 || c < r)) ? Sg[(size_t)gr * n + gc] : 0.0;   // (the diagonal tile: strictly lower part -- the pivots come as reciprocals in dv)
            }
        }
        if (with_dv && t0 == 0 && wave == NW - 1 && lane < 16) dv[16 * buf + lane] = 16 * pnl + lane < n ? (ML ? dvec[16 * pnl + lane] : P.dinv[16 * pnl + lane]) : 0.0;
    };
    __syncthreads();                                          // the last panel's stores to S are done (same workgroup: visible through L1 after the barrier's waitcnt)
    if constexpr (!ZG && ML) {
        // Everything in LDS (a sliding window): the chain x_{p+1} -> x_p runs in wave 0 alone, ONE barrier per panel.  Only the tile next to the
        // diagonal, L[p+1,p], needs the x that has just been found: wave 0 takes it from its own registers (x_{p+1} sits in lane k of each 16-lane
        // row; the product is 16 row-broadcast fmacs) and goes straight into the triangle recurrence.  The tiles further down the column, L[b,p],
        // b >= p+2, were applied to z_p a step earlier by wave 1, while wave 0 was busy with panel p+1; the columns arrive two steps ahead of
        // wave 0 (three buffers of W tiles in the tile area), parked in registers while they travel.
        const int ln = lane & 15;
        for (int q = nblk - 1; q >= nblk - 2 && q >= 0; --q) fetch_col(q, q % 3, 0, false);
        __syncthreads();
        CWP(6);
        double xprev = 0;
        bool adjacent = false;                                 // block p+1 is coupled to panel p (the last panel has nothing below it)
        for (int p = nblk - 1; p >= 0; --p) {
            const int c0 = 16 * p;
            double pf[4] = {0, 0, 0, 0};
            int pf_dst = -1;
            auto request_tile = [&]() {                        // column p-2: tile `wave` of it into registers now, into its buffer at the end of the step
                if (p <= 1) return;
                const int q = p - 2, fa0 = act_start[q], fm = act_start[q + 1] - fa0;
                if (wave <= fm) {
                    const int b = wave == 0 ? q : (act[fa0 + wave - 1] & 0xFFFF), c = ln, gc = 16 * q + c;
                    pf_dst = ((q % 3) * W + wave) * CT + (lane >> 4) * CT_LD + c;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const int r = (lane >> 4) + 4 * k, gr = 16 * b + r; if (gr < n && gc < n && (wave > 0 || c < r)) pf[k] = Sg[(size_t)gr * n + gc]; }
                }
            };
            CWP(8);
            if (wave == 0) {
                // everything this step reads is addressed without the index lists (they were read a step ago): the loads go out together
                const MS_LDS double *T0 = tiles + ((p % 3) * W) * CT, *T1 = T0 + CT;
                double rr = z[c0 + ln], col[NB], t1[NB];
                const double di = dvec[c0 + ln];
#pragma unroll
                for (int j = 0; j < NB; ++j) col[j] = T0[j * CT_LD + ln];          // L[c0 + j][c0 + lane], zero on and above the diagonal
                if (adjacent) {
#pragma unroll
                    for (int r = 0; r < NB; ++r) t1[r] = T1[r * CT_LD + ln];
                }
                int a0n = 0, mn = 0, firstn = -1;              // the next step's coupling flag: its two dependent index reads hide behind this step's arithmetic
                if (p > 0) { a0n = act_start[p - 1]; mn = act_start[p] - a0n; if (mn > 0) firstn = act[a0n] & 0xFFFF; }
                if (adjacent) backsub_adjacent(rr, xprev, t1, std::make_integer_sequence<int, NB>{});
                backsub_steps(rr, di, col, std::make_integer_sequence<int, NB>{});
                xprev = rr * di;
                if (lane < NB) z[c0 + lane] = xprev;
                adjacent = firstn == p;
                request_tile();
                CWP(10);
            } else {
                request_tile();
                if (wave == 1 && p > 0) {
                    // z_{p-1} -= sum over the blocks b >= p+1 of column p-1 of L[b,p-1]^T x_b (all of them final): the four 16-lane rows share the tiles out,
                    // their sums meet in `part`
                    const int q = p - 1, a0 = act_start[q], m = act_start[q + 1] - a0;
                    const int first = (m > 0 && (act[a0] & 0xFFFF) == q + 1) ? 1 : 0, rq = lane >> 4;
                    double sum = 0;
                    for (int e = first + rq; e < m; e += 4) {
                        const int b = act[a0 + e] & 0xFFFF;
                        const MS_LDS double *T = tiles + ((q % 3) * W + e + 1) * CT;
#pragma unroll
                        for (int r = 0; r < 16; ++r) sum += T[r * CT_LD + ln] * z[16 * b + r];
                    }
                    part[lane] = sum;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (lane < NB) z[16 * q + lane] -= (part[lane] + part[16 + lane]) + (part[32 + lane] + part[48 + lane]);
                    CWP(10);
                }
            }
            if (pf_dst >= 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) tiles[pf_dst + 4 * k * CT_LD] = pf[k];
            }
            if (p > 1) fetch_col(p - 2, (p - 2) % 3, NW, false);
            CWP(11);
            __syncthreads();
            CWP(12);
        }
        for (int i = tid; i < n; i += NT) P.dp[i] = z[i];
        __syncthreads();
        CWP(13);
        CWP_FLUSH;
        return;
    }
    if (nblk > 0) fetch_col(nblk - 1, 0, 0, true);
    __syncthreads();
    CWP(6);
    for (int p = nblk - 1; p >= 0; --p) {
        const int c0 = 16 * p, nb = min(16, n - c0), buf = (nblk - 1 - p) & 1;
        const int a0 = act_start[p], m = act_start[p + 1] - a0;
        // column p-1 is requested now and parked in registers (tile `wave` of it; a column of more than NW tiles fetches the rest the plain way):
        // it reaches the other buffer at the end of this step, after a whole step of latency cover
        double pf[4] = {0, 0, 0, 0}, pf_dv = 0;
        int pf_dst = -1;
        if (p > 0) {
            const int fa0 = act_start[p - 1], fm = act_start[p] - fa0;
            if (wave <= fm) {
                const int b = wave == 0 ? p - 1 : (act[fa0 + wave - 1] & 0xFFFF), c = lane & 15, gc = 16 * (p - 1) + c;
                pf_dst = ((buf ^ 1) * W + wave) * CT + (lane >> 4) * CT_LD + c;
#pragma unroll
                for (int k = 0; k < 4; ++k) { const int r = (lane >> 4) + 4 * k, gr = 16 * b + r; if (gr < n && gc < n && (wave > 0 || c < r)) pf[k] = Sg[(size_t)gr * n + gc]; }
            }
            if (wave == NW - 1 && lane < 16 && 16 * (p - 1) + lane < n) pf_dv = ML ? dvec[16 * (p - 1) + lane] : P.dinv[16 * (p - 1) + lane];
        }
        if (tid < 16 * m) {
            const int t = 1 + (tid >> 4), c = tid & 15, b = act[a0 + t - 1] & 0xFFFF;
            const MS_LDS double *T = tiles + (buf * W + t) * CT;
            double sum = 0;
            if (!ZG || 16 * b + 16 <= n) {                     // (the LDS copy of z is padded with zeros to a whole block, and so are the tiles)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += T[r * CT_LD + c] * z[16 * b + r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) { const int gr = 16 * b + r; if (gr < n) sum += T[r * CT_LD + c] * z[gr]; }
            }
            part[tid] = sum;
        }
        CWP(8);
        __syncthreads();
        CWP(9);
        if (wave == 0) {
            const MS_LDS double *T0 = tiles + (buf * W) * CT;
            double rr = 0, di = 0, col[NB];
            if (lane < nb) { rr = z[c0 + lane]; for (int t = 0; t < m; ++t) rr -= part[16 * t + lane]; di = dv[16 * buf + lane]; }
            if (nb == NB) {                                    // (the tile holds zeros on and above its diagonal: col[j] is zero for lanes >= j)
#pragma unroll
                for (int j = 0; j < NB; ++j) col[j] = T0[j * CT_LD + (lane & 15)];
                backsub_steps(rr, di, col, std::make_integer_sequence<int, NB>{});
            } else {
#pragma unroll
                for (int j = 0; j < NB; ++j) col[j] = (j < nb && lane < j) ? T0[j * CT_LD + lane] : 0.0;       // L[c0 + j][c0 + lane], j > lane
#pragma unroll
                for (int j = NB - 1; j >= 0; --j) {
                    if (j < nb) {                              // uniform
                        const double xj = readlane_d(rr * di, j);
                        rr -= col[j] * xj;
                    }
                }
            }
            if (lane < nb) z[c0 + lane] = rr * di;
        }
        CWP(10);
        if (pf_dst >= 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) tiles[pf_dst + 4 * k * CT_LD] = pf[k];
        }
        if (p > 0) {
            if (wave == NW - 1 && lane < 16) dv[16 * (buf ^ 1) + lane] = pf_dv;
            fetch_col(p - 1, buf ^ 1, NW, false);
        }
        CWP(11);
        __syncthreads();
        CWP(12);
    }
    for (int i = tid; i < n; i += NT) P.dp[i] = z[i];
    __syncthreads();
    CWP(13);
    CWP_FLUSH;
}
#ifdef MS_CW_PROF
extern "C" int ms_debug_cwprof(long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cwprof), sizeof(long long) * 48); }
#endif

// The same factorisation for systems whose panel does not fit the LDS (more than kMaxFreePoses free poses: global bundle
// adjustment, bundle_adjuster.cpp:493-604), spread over the team: the row tiles of a panel are updated by all waves of all
// workgroups into a panel buffer in global memory; after a team barrier EVERY workgroup factors the 16 x 16 diagonal block
// for itself (4 k cycles of redundant work instead of a second 6 us barrier), solves its share of the rows below against it and
// writes them straight into S.  Two team barriers per panel.  The arithmetic per entry is the same as in cholesky_solve.
__device__ __noinline__ void cholesky_factor_team(const BaProb &P_, double *lds_) {
    const BaProb &P = P_;
    BA_IDS
    const int n = P.n6, CT = P.chol_team;            // workgroups that take part (the rest wait at the caller's team barrier)
    if (rank_ >= CT) return;
    const int cgt = rank_ * NT + tid, CGT = CT * NT, cgw = rank_ * NW + wave, CGW = CT * NW;
    double *pan = P.panG;                           // [(n+1)][NB] in global memory
    double *sd = lds_;                              // [NB][NB] factored diagonal block
    double *tvec = lds_ + NB * NB;                  // [NB] reciprocal pivots
    for (int c0 = 0; c0 < n; c0 += NB) {
        const int nb = min(NB, n - c0), m = n - c0 + 1;
        chol_panel_update<true>(P, c0, nb, m, pan, cgw, CGW, lane);
        chol_sync(P);
        if (wave == 0) {
            double r[NB], di;
            const bool ok = chol_factor_diag(pan, nb, lane, r, di);
#pragma unroll
            for (int c = 0; c < NB; ++c) if (lane < NB) sd[lane * NB + c] = r[c];
            if (lane < NB) tvec[lane] = di;
            if (rank_ == 0) {
#pragma unroll
                for (int c = 0; c < NB; ++c) if (lane < nb && c <= lane) P.S[(size_t)(c0 + lane) * n + c0 + c] = r[c];
                if (lane < nb) P.dinv[c0 + lane] = di;
                if (!ok && lane == 0) P.flag[0] = 0;
            }
        }
        __syncthreads();
        const MS_GLOBAL int32_t *al = (const MS_GLOBAL int32_t *)P.act_blk + P.act_start[c0 / NB];
        const int n_act = P.act_start[c0 / NB + 1] - P.act_start[c0 / NB];
        for (int idx = cgt; idx < n_act * 16; idx += CGT) {      // rows below, active tiles only: x L11^T = a
            const int i = al[idx >> 4] * 16 + (idx & 15);
            if (i < nb || i >= m) continue;
            double x[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (j < nb) {
                    double s = pan[(size_t)i * NB + j];
                    for (int k = 0; k < j; ++k) s -= x[k] * sd[j * NB + k];
                    x[j] = s * tvec[j];
                }
            }
            double *dst = c0 + i < n ? P.S + (size_t)(c0 + i) * n + c0 : P.y + c0;
#pragma unroll
            for (int j = 0; j < NB; ++j) if (j < nb) dst[j] = x[j];
        }
        chol_sync(P);
    }
}

__device__ __noinline__ void point_backsub(const BaProb &P_) {
    const BaProb &P = P_;
    BA_IDS
    // point back-substitution: dl = Hinv (bl - sum_a Hpl_a^T dp_a)
    for (int l = gt; l < P.n_point; l += GT) {
        double *d = P.dl + 3 * (size_t)l;
        if (P.point_fixed && P.point_fixed[l]) { d[0] = d[1] = d[2] = 0; continue; }
        double r[3] = {P.bl[3 * (size_t)l], P.bl[3 * (size_t)l + 1], P.bl[3 * (size_t)l + 2]};
        for (int ii = P.pt_start[l]; ii < P.pt_start[l + 1]; ++ii) {
            const int o = P.pt_obs[ii], fa = P.pidx[P.obs_pose[o]];
            if (fa < 0) continue;
            double W[18], x[6];
            load18(P.Hpl + 18 * (size_t)o, W);
            load6(P.dp + 6 * fa, x);
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int a = 0; a < 6; ++a) r[c] -= W[3 * a + c] * x[a];
        }
        double h[6];
        load6(P.Hinv + 6 * (size_t)l, h);
        d[0] = h[0] * r[0] + h[1] * r[1] + h[2] * r[2];
        d[1] = h[1] * r[0] + h[3] * r[1] + h[4] * r[2];
        d[2] = h[2] * r[0] + h[4] * r[1] + h[5] * r[2];
    }
    team_sync(P);
}

// Returns false (uniformly) when a pivot is not positive / a point block is singular.
__device__ bool solve_step(const BaProb &P, double lambda, double *lds, long long *cyc) {
    long long t0 = clock64();
    const bool lead = P.team == 1 || blockIdx.x % (unsigned)P.team == 0;
    // P.flag[0] ("this damped solve is sound") is 1 on entry: the caller sets it behind a barrier that every reader of the previous value has passed
    const bool fused = P.fused != 0;
    if (fused) {
        const bool pl = P.n_pose <= kFsPoseTab;
        if (P.fs[P.team > 1 ? 1 : 0].by_points) { if (pl) schur_fused<true, true>(P, lambda, lds, cyc); else schur_fused<true, false>(P, lambda, lds, cyc); }
        else { if (pl) schur_fused<false, true>(P, lambda, lds, cyc); else schur_fused<false, false>(P, lambda, lds, cyc); }
    }
    else {
        schur_prepare(P, lambda);
        { const long long t1 = clock64(); cyc[6] += t1 - t0; }
        schur_segments(P, lds);
    }
    { const long long t1 = clock64(); cyc[2] += t1 - t0; t0 = t1; }
    if (P.cw_W > 0) {                                  // the active front fits the LDS: right-looking, one workgroup (banded systems of any size)
        if (lead) { if (P.cw_zglobal) cholesky_window<true, false>(P, lds); else if (P.cw_meta_lds) cholesky_window<false, true>(P, lds); else cholesky_window<false, false>(P, lds); }
    } else if (P.panG) {                               // too large for an LDS panel: factor across the team, substitute back in one workgroup
        cholesky_factor_team(P, lds);
        if (lead) chol_back_substitute(P, lds, threadIdx.x, threadIdx.x & 63, threadIdx.x >> 6);
    } else if (lead) cholesky_solve(P, lds);           // dense front: left-looking, panel in LDS, operands from L2
    team_sync(P);
    { const long long t1 = clock64(); cyc[3] += t1 - t0; t0 = t1; }
    __shared__ int s_sound;                                // one reader: the verdict is the same in every thread of the workgroup whatever else writes the flag
    if (threadIdx.x == 0) s_sound = __hip_atomic_load(P.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool ok = s_sound != 0;
    __syncthreads();
    if (!ok) return false;
    if (fused && P.team == 1 && P.fo_lo != nullptr && 3 * (size_t)P.n_point + 7 * (size_t)P.n_pose + 6 * (size_t)P.np_free <= kLdsBytes / 8 && !MS_LIN_NO_STREAM) backsub_stream(P, lambda, lds);
    else if (fused) point_backsub_fused(P, lambda, lds); else point_backsub(P);
    cyc[4] += clock64() - t0;
    return true;
}

// before a team launch: arrival counters and the gave-up marker back to zero (the counter is monotonic within a launch)
__global__ void k_ba_team_reset(const BaProb *probs, int n, int gave_up) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { probs[i].bar[0] = 0; probs[i].bar[32] = 0; probs[i].bar[kBeatWord] = 0; probs[i].flag[1] = gave_up; }
}

// one problem's results side by side -- [stats 16][pose 7 n_pose][point 3 n_point][chi2 n_obs] -- so that ms_ba_download is ONE copy instead of four
__global__ __launch_bounds__(256) void k_ba_pack_result(const BaProb *probs, int i, double *dst, int with_chi2) {
    const BaProb &P = probs[i];
    const int np7 = 7 * P.n_pose, nl3 = 3 * P.n_point, total = 16 + np7 + nl3 + (with_chi2 ? P.n_obs : 0);
    for (int k = blockIdx.x * 256 + threadIdx.x; k < total; k += gridDim.x * 256) {
        double v;
        if (k < 16) v = (k == 7 && P.team > 1 && P.flag[1] != 0) ? 1.0 : P.stats[k];      // [7] "a team barrier gave up": the marker itself, read here, after the launch (a lane of the solve kernel
                                                                                           //     may have looked before a workgroup on another XCD had set it)
        else if (k < 16 + np7) v = P.pose[k - 16];
        else if (k < 16 + np7 + nl3) v = P.point[k - 16 - np7];
        else v = P.chi2_obs[k - 16 - np7 - nl3];
        dst[k] = v;
    }
}

// after a team launch: did ANY problem of the batch see a barrier give up?  (one word, in the first problem's flag line)
__global__ void k_ba_collect_gave_up(const BaProb *probs, int n) {
    int any = 0;
    for (int i = threadIdx.x; i < n; i += 64) any |= probs[i].flag[1];
    any = __any(any != 0);
    if (threadIdx.x == 0) probs[0].flag[2] = any;
}

// the solved state of src's problems becomes the initial state of dst's (ms_ba_copy_state); poses dst has beyond src's take src's pose extra[p]
__global__ __launch_bounds__(256) void k_ba_copy_state(const BaProb *dst, const BaProb *src, const int32_t *extra) {
    const BaProb &D = dst[blockIdx.x], &S = src[blockIdx.x];
    double *pose0 = const_cast<double *>(D.pose0), *point0 = const_cast<double *>(D.point0);
    const int ex = extra ? extra[blockIdx.x] : 0;
    for (int i = threadIdx.x; i < 7 * D.n_pose; i += 256) {
        const int k = i / 7, from = k < S.n_pose ? k : ex;
        pose0[i] = S.pose[7 * (size_t)from + (i - 7 * k)];
    }
    for (int i = threadIdx.x; i < 3 * D.n_point; i += 256) point0[i] = S.point[i];
}

// grid = problems x team workgroups; workgroup b works on problem b / team
#ifdef MS_BA_WAVES_PER_EU
#define MS_BA_OCC __attribute__((amdgpu_waves_per_eu(MS_BA_WAVES_PER_EU, MS_BA_WAVES_PER_EU)))
#else
#define MS_BA_OCC
#endif
// alt (one workgroup per problem, streamed phases; else nullptr): the same problems with the linearisation's outputs -- Hpp, bp, Hll, bl -- pointing at a SECOND set of
// arrays.  The chi2 of a trial state and the linearisation of the next iteration walk the same observations at the same state when the trial is accepted: a trial
// then linearises into the set that is not in use and takes its chi2 from the same pass (build_system's chi2_share); accepted, the two descriptors change roles,
// rejected, the old set is still there.  One pass over the observations per trial instead of two per iteration.
__global__ __launch_bounds__(NT) MS_BA_OCC void k_ba_lm(const BaProb *probs, int team, const BaProb *alt) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double s_red[NW];
    const BaProb &P = probs[blockIdx.x / (unsigned)team];      // fields stay in constant memory: uniform scalar loads, no private copy
    BA_IDS
    const int n6 = P.n6;
    int seq = 0;                                               // team_reduce call counter (same in every workgroup)
    // restart from the initial estimates
    for (int i = gt; i < 7 * P.n_pose; i += GT) { P.pose[i] = P.pose0[i]; P.pose_bk[i] = P.pose0[i]; }      // (the saved copy of a FIXED pose is never written again)
    for (int i = gt; i < 3 * P.n_point; i += GT) P.point[i] = P.point0[i];
    if (gt == 0) P.flag[0] = 1;
    team_sync(P);
    double lambda = 0, ni = 2;
    int it = 0, trials = 0, stop = 0;
    long long cyc[7] = {0, 0, 0, 0, 0, 0, 0};
    const long long t_begin = clock64();
    // one workgroup per window, streams built, pose table in LDS: the streamed phases (eval_stream, backsub_stream)
    const bool ev_stream = P.team == 1 && P.fo_lo != nullptr && 7 * (size_t)P.n_pose <= kLdsBytes / 8 && !MS_LIN_NO_STREAM;
    const bool fuse = ev_stream && alt != nullptr && lin_streams(P) && !MS_BA_NO_FUSED_TRIAL;
    const BaProb *Pc = &P, *Pa = fuse ? &alt[blockIdx.x] : &P;            // the descriptor whose linearisation belongs to the accepted state / the other one
    // chi2 of the state as it stands + its linearisation into *Q, one pass (fused schedule); extra / extra_sum as in eval_stream
    auto linearise_and_chi2 = [&](const BaProb &Q, double extra, double *extra_sum) {
        double share = 0;
        build_system(Q, lds, &share);
        const double total = block_sum(share, s_red);
        if (extra_sum) *extra_sum = block_sum(extra, s_red);
        return total;
    };
    const double chi2_init = fuse ? linearise_and_chi2(*Pc, 0.0, nullptr) : ev_stream ? eval_stream(P, lds, s_red, false, 0.0, nullptr) : eval_chi2(P, s_red, false, seq);
    double chi2_carried = chi2_init;
    for (it = 0; it < P.max_iters; ++it) {
        long long tt = clock64();
        // g2o evaluates activeRobustChi2 again here; the state is the one the last accepted (or restored) trial left, so it is the same number
        double current = chi2_carried, temp = current;
        { const long long t1 = clock64(); cyc[0] += t1 - tt; tt = t1; }
        if (!fuse) build_system(P, lds);                 // (fused schedule: the state's linearisation exists already -- the initial pass, or the trial that was accepted)
        cyc[1] += clock64() - tt;
        if (it == 0) {                                   // computeLambdaInit
            double md = 0;
            for (int i = gt; i < n6; i += GT) md = fmax(md, fabs(Pc->Hpp[(size_t)i * n6 + i]));
            for (int l = gt; l < P.n_point; l += GT)
                if (!(P.point_fixed && P.point_fixed[l])) { const double *h = Pc->Hll + 6 * (size_t)l; md = fmax(md, fmax(fabs(h[0]), fmax(fabs(h[3]), fabs(h[5])))); }
            lambda = 1e-5 * team_reduce(P, md, s_red, true, seq); ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        do {
            // push() happens where the state is changed: the update below saves every value it overwrites (a trial whose system cannot be solved changes nothing
            // and needs neither the copy nor the restore) -- one walk over the state per trial instead of three (save, update, gain denominator)
            const bool ok2 = solve_step(*Pc, lambda, lds, cyc);
            tt = clock64();
            // re-arm the flag for the next damped solve: on the good path every workgroup has read it before point_backsub's barrier; on the
            // (rare) failed path an extra barrier separates the reads from the write.  The barriers that follow order it before the next solve's phases.
            if (!ok2) team_sync(P);
            if (gt == 0) P.flag[0] = 1;
            double sc = 0, scale = 0;
            if (ok2) {
                for (int fp = gt; fp < P.np_free; fp += GT) {
                    const int pi = P.free2pose[fp];
                    double ex[7], r[7], old[7];
                    for (int a = 0; a < 7; ++a) { old[a] = P.pose[7 * (size_t)pi + a]; P.pose_bk[7 * (size_t)pi + a] = old[a]; }
                    se3_exp(P.dp + 6 * fp, ex);
                    se3_mul(ex, old, r);
                    for (int a = 0; a < 7; ++a) P.pose[7 * (size_t)pi + a] = r[a];
                }
                {
                    const MS_GLOBAL double *gdl = uglobal(P.dl), *gbl = uglobal(Pc->bl);
                    MS_GLOBAL double *gpt = (MS_GLOBAL double *)uglobal(P.point), *gbk = (MS_GLOBAL double *)uglobal(P.point_bk);
                    const int n3 = 3 * P.n_point;
#pragma unroll 4
                    for (int i = gt; i < n3; i += GT) {
                        const double old = gpt[i], d = gdl[i];
                        gbk[i] = old; gpt[i] = old + d;
                        sc = fma(d, fma(lambda, d, gbl[i]), sc);
                    }
                }
                team_sync(P);
            }
            cyc[4] += clock64() - tt; tt = clock64();
            if (ok2) for (int i = gt; i < n6; i += GT) sc += P.dp[i] * (lambda * P.dp[i] + Pc->bp[i]);
            temp = ok2 ? (fuse ? linearise_and_chi2(*Pa, sc, &scale) : ev_stream ? eval_stream(P, lds, s_red, false, sc, &scale) : eval_chi2(P, s_red, false, seq, sc, &scale)) : DBL_MAX;     // chi2 of the new state and the gain denominator behind one team barrier
            scale += 1e-3;
            cyc[0] += clock64() - tt;
            rho = (current - temp) / scale;
            if (trials < P.debug_reject) rho = -1.0;                // (test hook: drives the ten-rejections Terminate path)
            if (rho > 0 && isfinite(temp)) {
                const double t3 = 2 * rho - 1;
                double alpha = 1. - t3 * t3 * t3;                                          // (pow(x, 3) of the reference to an ulp or two)
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha);
                ni = 2; current = temp; chi2_carried = temp;
                if (fuse) { const BaProb *t = Pc; Pc = Pa; Pa = t; }                      // the trial's linearisation is the accepted state's
            } else {
                lambda *= ni; ni *= 2;
                if (ok2) {                                                                  // pop(): only a trial that moved the state has something to take back
                    for (int i = gt; i < 7 * P.n_pose; i += GT) P.pose[i] = P.pose_bk[i];
                    for (int i = gt; i < 3 * P.n_point; i += GT) P.point[i] = P.point_bk[i];
                }
                team_sync(P);
                if (!isfinite(lambda)) break;
            }
            ++qmax; ++trials;
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0 || !isfinite(lambda)) { stop = 1; ++it; break; }
    }
    const double chi2_final = ev_stream ? eval_stream(P, lds, s_red, true, 0.0, nullptr) : eval_chi2(P, s_red, true, seq);
    if (gt == 0) {
        const bool hung = P.team > 1 && P.flag[1] != 0;        // a team barrier gave up: the result is not to be trusted
        P.stats[0] = it; P.stats[1] = trials; P.stats[2] = stop; P.stats[3] = lambda; P.stats[4] = chi2_init; P.stats[5] = hung ? NAN : chi2_final;
        P.stats[6] = (isfinite(chi2_final) && !hung) ? 1 : 0; P.stats[7] = hung ? 1 : 0;
        for (int k = 0; k < 5; ++k) P.stats[8 + k] = (double)cyc[k];
        P.stats[13] = (double)(clock64() - t_begin); P.stats[14] = (double)cyc[5]; P.stats[15] = (double)cyc[6];
    }
}

// ---------------------------------------------------------------- pose-only problems: poseBundleAdjust (bundle_adjuster.cpp:396-491)
// ONE free pose, every point fixed: the projection edges of the current frame (+ the odometry edge to the fixed previous keyframe, :440-447) give a 6 x 6
// system per trial.  The general kernel walks such a problem through ~20 phases with a workgroup barrier and a round trip to memory each (0.47 ms for
// 12 iterations of a 6-dof problem -- and poseBundleAdjust runs on EVERY non-keyframe, mapper_helpers.cpp:1043-1050); here a 256-thread workgroup keeps
// the trial in registers: one sweep over the observations accumulates H (21 entries), b and the robust chi2 per thread, one reduction, lane 0 factors the
// 6 x 6 matrix, one more sweep gives the chi2 of the moved pose.  Same LM schedule, same arithmetic per edge as k_ba_lm.
// Round 4: the stamps showed 45 % of the kernel in its reductions (28 sums x 6 ds_bpermute steps, three barriers per sweep), 20 % in the observations and 18 % in
// the one SE3 edge's logarithm on the thread that also had the most observations -- and two sweeps per iteration where one is enough:
//   * every sweep linearises.  The trial's sweep at the moved pose already holds H and b of the NEXT iteration when the trial is accepted (g2o linearises the
//     same state again, optimizable_graph / sparse_optimizer computeActiveErrors + linearizeSystem), and a rejected trial keeps the old H, b in registers: 1 + trials
//     sweeps instead of 2 + iterations + trials, and the per-observation chi2 of the accepted state stays in registers, so the closing sweep goes too;
//   * the 28 sums go through LDS transposed: every thread writes its 28 partial sums (value-major rows of PO_ROW doubles), 8 lanes per value add 32 entries each
//     and meet over the DPP network (3 steps): two barriers, ~100 instructions;
//   * waves 0-2 take the observations (PO_K per thread in registers), wave 3 takes the SE3 edges, so the logarithm runs beside the observations.  One wave per
//     SIMD: the whole 512-register file is each wave's (a 512-thread version, two waves per SIMD, spilled ~200 dwords per trial and was no faster than round 3's).
constexpr int PO_NT = 256, PO_OT = 192, PO_K = 8, PO_MAXE = 8, PO_NV = 28;       // threads; threads with observations; observations a thread keeps in registers; SE3 edges; sums per sweep
constexpr int PO_ROW = PO_NT + 8;                            // row stride (doubles) of the transposed partial sums: 8 mod 32, so the 8 values x 8 parts a wave reads spread over all banks
constexpr size_t kPoLdsBytes = (size_t)PO_NV * PO_ROW * sizeof(double);
static_assert(8 * PO_NV <= PO_NT, "stage 2 of the reduction: 8 lanes per sum");
template <int CTRL> __device__ __forceinline__ double dpp_d(double v) {               // the double of the lane the DPP control names (all source lanes active)
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// The SE3 edges of a problem with ONE free pose (k_ba_pose_only, k_ba_one_pose), once per solve, by ONE wave (all its lanes): an edge between fixed poses is a constant chi2
// (returned: every lane's share); one that touches
// the free pose leaves its fixed side's transform, G = -J^T W and J^T W J = -G J in LDS.  Only the free side's Jacobian is formed (Ji = adj(Tj^-1 M) or
// Jj = -adj(Ti^-1 M^-1): no logarithm), and the two 6 x 6 products are spread over the lanes -- as the work of one thread (two Jacobians in scratch, 970 dependent
// multiply-adds) this was 44 k cycles, a third of the whole kernel.
__device__ __noinline__ double po_edges_setup(const BaProb &P, int pi, int lane, int *s_ne, int *s_eSide, double *s_eC, double *s_eM, double *s_eG, double *s_eW, double *s_Hc, double *s_J) {
    double cacc = 0;
    int ne = 0;
    for (int k = lane; k < P.n_edge; k += 64) {                             // the edges between fixed poses (a window's stage 1 has ~50): one per lane
        const int vi = P.edge_i[k], vj = P.edge_j[k];
        if (vi == pi || vj == pi) continue;
        const double *W = P.edge_info + 36 * (size_t)k;
        double e[6];
        pose_edge(P.pose0 + 7 * (size_t)vi, P.pose0 + 7 * (size_t)vj, P.edge_meas + 7 * (size_t)k, e, nullptr, nullptr, false);
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) cacc += e[i] * W[6 * i + j] * e[j];
    }
    // the edges of the free pose: found 64 at a time (a ballot over the lanes' edges -- a scan with one dependent load per edge was 35 us for a window's 50 edges), then
    // taken by the wave together
    for (int k0 = 0; k0 < P.n_edge; k0 += 64) {
      const int kl = k0 + lane;
      unsigned long long touching = __ballot(kl < P.n_edge && (P.edge_i[kl] == pi || P.edge_j[kl] == pi));
      while (touching) {
        const int k = k0 + __builtin_ctzll(touching);
        touching &= touching - 1;
        const int vi = P.edge_i[k], vj = P.edge_j[k];
        const double *W = P.edge_info + 36 * (size_t)k, *M = P.edge_meas + 7 * (size_t)k, *Ti = P.pose0 + 7 * (size_t)vi, *Tj = P.pose0 + 7 * (size_t)vj;
        const int sl = ne++, side = vi == pi ? 0 : 1;
        double C7[7], J[36];
        if (side == 0) { double Tjinv[7]; se3_inv(Tj, Tjinv); se3_mul(Tjinv, M, C7); se3_adj(C7, J); }
        else { double Tiinv[7], Minv[7]; se3_inv(Ti, Tiinv); se3_inv(M, Minv); se3_mul(Tiinv, Minv, C7); se3_adj(C7, J); for (int i = 0; i < 36; ++i) J[i] = -J[i]; }
        if (lane == 0) {
            s_eSide[sl] = side;
            if (side == 0) for (int i = 0; i < 7; ++i) s_eC[8 * sl + i] = C7[i];
            else for (int i = 0; i < 7; ++i) { s_eC[8 * sl + i] = Ti[i]; s_eM[8 * sl + i] = M[i]; }
            for (int i = 0; i < 36; ++i) s_J[i] = J[i];
        }
        if (lane < 36) s_eW[36 * sl + lane] = W[lane];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (lane < 36) {                                                    // G = -J^T W
            const int a2 = lane / 6, c = lane % 6;
            double v = 0;
            for (int r2 = 0; r2 < 6; ++r2) v += s_J[6 * r2 + a2] * s_eW[36 * sl + 6 * r2 + c];
            s_eG[36 * sl + lane] = -v;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (lane < 21) {                                                    // upper triangle of J^T W J = -G J, entry (a, b) at a (11 - a) / 2 + b
            int a2 = 0, rest = lane;
            while (rest >= 6 - a2) { rest -= 6 - a2; ++a2; }
            const int b2 = a2 + rest;
            double v = 0;
            for (int c = 0; c < 6; ++c) v += s_eG[36 * sl + 6 * a2 + c] * s_J[6 * c + b2];
            s_Hc[lane] -= v;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
      }
    }
    if (lane == 0) *s_ne = ne;
    return cacc;
}

#ifndef MS_PO_PROF
#define MS_PO_PROF 0            // 1: cycle stamps per phase in stats[8 .. 12) (tools/pose_only_prof.py; each stamp is an s_memtime and a wait, ~10 % of the kernel together)
#endif
#if MS_PO_PROF
#define PO_CLOCK() clock64()
#else
#define PO_CLOCK() 0ll
#endif
__global__ __launch_bounds__(PO_NT) void k_ba_pose_only(const BaProb *probs) {
    extern __shared__ __attribute__((aligned(16))) double po_red[];         // [PO_NV][PO_ROW]
    __shared__ __attribute__((aligned(16))) double s_sum[2][32];       // the sums of the last two sweeps: [cur] belongs to the accepted state (its H and b), the other to the trial
    __shared__ double s_eC[PO_MAXE][8], s_eM[PO_MAXE][8], s_eG[PO_MAXE][36], s_eW[PO_MAXE][36], s_Hc[24], s_const;
    __shared__ int s_eSide[PO_MAXE], s_ne, s_any;
    __shared__ double s_J[36];
    const long long t_kernel = clock64();
    const BaProb &P = probs[blockIdx.x];
    const int tid = threadIdx.x, pi = P.free2pose[0];
    // the block ms_ba_download fetches -- [16 stats][7 n_pose poses][3 n_point points][n_obs chi2] -- is filled as the values arise (a single problem: BaProb::pack)
    MS_GLOBAL double *pk = (MS_GLOBAL double *)uglobal(P.pack);
    MS_GLOBAL double *pk_pose = pk + 16, *pk_point = pk_pose + 7 * P.n_pose, *pk_chi2 = pk_point + 3 * P.n_point;
    for (int i = tid; i < 7 * P.n_pose; i += PO_NT) { const double v = P.pose0[i]; P.pose[i] = v; if (pk) pk_pose[i] = v; }
    for (int i = tid; i < 3 * P.n_point; i += PO_NT) { const double v = P.point0[i]; P.point[i] = v; if (pk) pk_point[i] = v; }
    if (tid < 24) s_Hc[tid] = 0;
    if (tid == 0) { s_ne = 0; s_const = 0; s_any = P.n_obs; }
    __syncthreads();
    {   // the first observation of the free pose
        int mine = P.n_obs;
        for (int o = tid; o < P.n_obs; o += PO_NT) if (P.obs_pose[o] == pi) { mine = o; break; }
        if (mine < P.n_obs) atomicMin(&s_any, mine);
    }
    __syncthreads();
    double pose[7];
#pragma unroll
    for (int a = 0; a < 7; ++a) pose[a] = P.pose0[7 * (size_t)pi + a];
    // ---- what never changes: the observations of the free pose go into registers (PO_K per thread of waves 0-2), the others' chi2 and the edges between fixed
    //      poses into one constant; an edge that touches the free pose leaves its fixed side's transform, -J^T W and J^T W J in LDS (its Jacobian does not depend
    //      on the free pose: Ji = adj(Tj^-1 M), Jj = -adj(Ti^-1 M^-1)), so a sweep only takes its logarithm
    double X[PO_K][3], uv[PO_K][2], info[PO_K], c2[PO_K], c2t[PO_K], cacc = 0;
    unsigned have = 0;
    const int o_any = s_any;                                                // an observation of the free pose (n_obs: there is none)
    double Xd[3] = {0, 0, 1}, uvd[2] = {0, 0};
    if (o_any < P.n_obs) {
        const int l = P.obs_point[o_any];
        Xd[0] = P.point0[3 * (size_t)l]; Xd[1] = P.point0[3 * (size_t)l + 1]; Xd[2] = P.point0[3 * (size_t)l + 2];
        uvd[0] = P.obs_uv[2 * (size_t)o_any]; uvd[1] = P.obs_uv[2 * (size_t)o_any + 1];
    }
#pragma unroll
    for (int j = 0; j < PO_K; ++j) {
        const int o = tid + PO_OT * j;
        X[j][0] = Xd[0]; X[j][1] = Xd[1]; X[j][2] = Xd[2]; uv[j][0] = uvd[0]; uv[j][1] = uvd[1]; info[j] = c2[j] = c2t[j] = 0;
        if (tid < PO_OT && o < P.n_obs) {
            const int po = P.obs_pose[o], l = P.obs_point[o];
            if (po == pi) {
                have |= 1u << j;
                X[j][0] = P.point0[3 * (size_t)l]; X[j][1] = P.point0[3 * (size_t)l + 1]; X[j][2] = P.point0[3 * (size_t)l + 2];
                uv[j][0] = P.obs_uv[2 * (size_t)o]; uv[j][1] = P.obs_uv[2 * (size_t)o + 1]; info[j] = P.obs_info[o];
            }
        }
    }
    // slots this wave sweeps: observation tid + PO_OT j exists for j < (n_obs - tid) / PO_OT, most for the wave's first lane (wave 3, the edges' wave, and a problem
    // without observations of the free pose: none)
    const int w0 = __builtin_amdgcn_readfirstlane(tid & ~63);
    const int jn = (w0 < PO_OT && o_any < P.n_obs && P.n_obs > w0) ? min(PO_K, (P.n_obs - w0 + PO_OT - 1) / PO_OT) : 0;
    for (int o = tid; o < P.n_obs; o += PO_NT) {                            // observations from FIXED poses: constant
        const int po = P.obs_pose[o];
        if (po == pi) continue;
        double e[2], r, w;
        proj_edge<false>(P.pose0 + 7 * (size_t)po, P.point0 + 3 * (size_t)P.obs_point[o], P.obs_uv + 2 * (size_t)o, e, nullptr, nullptr);
        const double chi2 = P.obs_info[o] * (e[0] * e[0] + e[1] * e[1]);
        huber(chi2, P.huber, r, w);
        P.chi2_obs[o] = chi2;
        if (pk) pk_chi2[o] = chi2;
        cacc += r;
    }
    if (tid >= PO_OT) cacc += po_edges_setup(P, pi, tid - PO_OT, &s_ne, s_eSide, &s_eC[0][0], &s_eM[0][0], &s_eG[0][0], &s_eW[0][0], s_Hc, s_J);   // (wave 3, beside the observation loads)
    cacc = wave_sum_d(cacc);
    if ((tid & 63) == 0) lds_addd((MS_LDS double *)&s_const, cacc);
    __syncthreads();
    const int ne = s_ne;
    const bool overflow = P.n_obs > PO_OT * PO_K;                           // more observations than the registers take: those come from memory in every sweep
    long long pc[4] = {0, 0, 0, 0};                                         // cycles of thread 0: a sweep's observations, (thread PO_OT:) its SE3 edges, its reduction, the 6 x 6 solve + exp
    const long long t_begin = clock64();
    int cur = 1;                                                            // which half of s_sum belongs to the accepted state
    // ---- one sweep over the free pose's edges at `pose`: the robust chi2 (returned), the upper triangle of H and b (left in s_sum[1 - cur][0 .. 27)), the chi2 per observation in c2t
    auto sweep = [&]() {
        double A[21], g[6], acc = 0;
        long long ts = PO_CLOCK();
#pragma unroll
        for (int a = 0; a < 21; ++a) A[a] = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) g[a] = 0;
        auto one = [&](const double *Xo, const double *uvo, double inf) {
            double e[2], Jp[12], Jl[6];
            proj_edge<true>(pose, Xo, uvo, e, Jp, Jl);
            const double chi2 = inf * (e[0] * e[0] + e[1] * e[1]);
            double r, w;
            huber(chi2, P.huber, r, w);
            acc += r;
            const double wi = w * inf;
            int k = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                g[a] += -(Jp[a] * e[0] + Jp[6 + a] * e[1]) * wi;
#pragma unroll
                for (int b2 = a; b2 < 6; ++b2) A[k++] += wi * (Jp[a] * Jp[b2] + Jp[6 + a] * Jp[6 + b2]);
            }
            return chi2;
        };
        // slots in pairs, both evaluations in one basic block so that the scheduler interleaves their chains (a lone wave per SIMD has nothing else to hide the
        // fp64 latency with); a slot without an observation holds a copy of a real one with information 0: finite arithmetic, nothing added
#pragma unroll
        for (int j = 0; j < PO_K; j += 2) {
            if (j + 1 < jn) { c2t[j] = one(X[j], uv[j], info[j]); c2t[j + 1] = one(X[j + 1], uv[j + 1], info[j + 1]); }
            else if (j < jn) c2t[j] = one(X[j], uv[j], info[j]);
        }
        if (overflow && tid < PO_OT)
            for (int o = tid + PO_OT * PO_K; o < P.n_obs; o += PO_OT) {
                if (P.obs_pose[o] != pi) continue;
                const double Xo[3] = {P.point0[3 * (size_t)P.obs_point[o]], P.point0[3 * (size_t)P.obs_point[o] + 1], P.point0[3 * (size_t)P.obs_point[o] + 2]};
                const double uvo[2] = {P.obs_uv[2 * (size_t)o], P.obs_uv[2 * (size_t)o + 1]};
                (void)one(Xo, uvo, P.obs_info[o]);
            }
        if (MS_PO_PROF && tid == 0) { const long long t1 = PO_CLOCK(); pc[0] += t1 - ts; }
        if (tid >= PO_OT && tid - PO_OT < ne) {                             // wave 3: the SE3 edges of the free pose
            const int k = tid - PO_OT;
            double Bm[7], e[6], We[6];
            if (s_eSide[k] == 0) se3_mul(s_eC[k], pose, Bm);                                      // (Tj^-1 M) Ti
            else { double Tjinv[7], A2[7]; se3_inv(pose, Tjinv); se3_mul(Tjinv, s_eM[k], A2); se3_mul(A2, s_eC[k], Bm); }
            se3_log(Bm, e);
            for (int i = 0; i < 6; ++i) { double v = 0; for (int j = 0; j < 6; ++j) v += s_eW[k][6 * i + j] * e[j]; We[i] = v; }
            for (int i = 0; i < 6; ++i) acc += e[i] * We[i];
            for (int a = 0; a < 6; ++a) { double v = 0; for (int c = 0; c < 6; ++c) v += s_eG[k][6 * a + c] * e[c]; g[a] += v; }
            if (MS_PO_PROF && k == 0) pc[1] += PO_CLOCK() - ts;
        }
        ts = PO_CLOCK();
        {
            MS_LDS double *row = (MS_LDS double *)po_red + tid;
#pragma unroll
            for (int a = 0; a < 21; ++a) row[a * PO_ROW] = A[a];
#pragma unroll
            for (int a = 0; a < 6; ++a) row[(21 + a) * PO_ROW] = g[a];
            row[27 * PO_ROW] = acc;
        }
        __syncthreads();
        if (tid < 8 * PO_NV) {                                              // lanes 8 v .. 8 v + 7 add row v up, each 32 entries, then among themselves
            const int v = tid >> 3, part = tid & 7;
            const MS_LDS double *row = (const MS_LDS double *)po_red + v * PO_ROW + part;
            double s0 = 0, s1 = 0;
#pragma unroll
            for (int k = 0; k < PO_NT / 8; k += 2) { s0 += row[8 * k]; s1 += row[8 * k + 8]; }
            double s = s0 + s1;
            s += dpp_d<0xB1>(s);                                            // quad_perm [1 0 3 2]
            s += dpp_d<0x4E>(s);                                            // quad_perm [2 3 0 1]
            s += dpp_d<0x141>(s);                                           // row_half_mirror
            if (part == 0) s_sum[1 - cur][v] = s + (v < 21 ? s_Hc[v] : (v == 27 ? s_const : 0.0));
        }
        __syncthreads();
        if (MS_PO_PROF && tid == 0) pc[2] += PO_CLOCK() - ts;
        return s_sum[1 - cur][27];
    };
    double lambda = 0, ni = 2;
    int it = 0, trials = 0, stop = 0;
    auto accept = [&]() {                                                   // the swept state becomes the current one: its H, b (the other half of s_sum) and per-observation chi2
        cur = 1 - cur;
#pragma unroll
        for (int j = 0; j < PO_K; ++j) c2[j] = c2t[j];
    };
    const double chi2_init = sweep();
    accept();
    double chi2_carried = chi2_init;
    for (it = 0; it < P.max_iters; ++it) {
        double current = chi2_carried, temp = current;
        if (it == 0) {                                                       // computeLambdaInit: 1e-5 x the largest diagonal entry
            const double *H = s_sum[cur];
            const double md = fmax(fmax(fmax(fabs(H[0]), fabs(H[6])), fmax(fabs(H[11]), fabs(H[15]))), fmax(fabs(H[18]), fabs(H[20])));
            lambda = 1e-5 * md; ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        do {
            // (H + lambda I) dp = b: Cholesky of the 6 x 6 matrix, by every thread for itself (fully unrolled: 21 + 6 registers, ~150 dependent operations --
            // cheaper than one thread doing it behind a barrier and a trip through LDS)
            double Lm[21], dpv[6], b[6];                                     // lower triangle, row-major: Lm[i (i + 1) / 2 + j]
            bool ok2 = true;
            const long long tc = PO_CLOCK();
            const double *H = s_sum[cur];
#pragma unroll
            for (int a = 0; a < 6; ++a) b[a] = H[21 + a];
            {
                int k = 0;
#pragma unroll
                for (int a = 0; a < 6; ++a)
#pragma unroll
                    for (int c = a; c < 6; ++c) { Lm[c * (c + 1) / 2 + a] = H[k] + (a == c ? lambda : 0.0); ++k; }
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                double d = Lm[j * (j + 1) / 2 + j];
#pragma unroll
                for (int k = 0; k < j; ++k) d -= Lm[j * (j + 1) / 2 + k] * Lm[j * (j + 1) / 2 + k];
                if (!(d > 0) || !isfinite(d)) ok2 = false;
                const double inv = rsqrt_d(d);
                Lm[j * (j + 1) / 2 + j] = inv;                               // the reciprocal pivot is what the substitutions use
#pragma unroll
                for (int i = j + 1; i < 6; ++i) {
                    double v = Lm[i * (i + 1) / 2 + j];
#pragma unroll
                    for (int k = 0; k < j; ++k) v -= Lm[i * (i + 1) / 2 + k] * Lm[j * (j + 1) / 2 + k];
                    Lm[i * (i + 1) / 2 + j] = v * inv;
                }
            }
            {
                double y[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    double v = b[i];
#pragma unroll
                    for (int k = 0; k < i; ++k) v -= Lm[i * (i + 1) / 2 + k] * y[k];
                    y[i] = v * Lm[i * (i + 1) / 2 + i];
                }
#pragma unroll
                for (int i = 5; i >= 0; --i) {
                    double v = y[i];
#pragma unroll
                    for (int k = i + 1; k < 6; ++k) v -= Lm[k * (k + 1) / 2 + i] * dpv[k];
                    dpv[i] = v * Lm[i * (i + 1) / 2 + i];
                }
            }
            double bk[7], sc = 0;
#pragma unroll
            for (int a = 0; a < 7; ++a) bk[a] = pose[a];                                    // push()
            if (ok2) {
                double dp[6], ex[7];
#pragma unroll
                for (int a = 0; a < 6; ++a) { dp[a] = dpv[a]; sc += dp[a] * (lambda * dp[a] + b[a]); }
                se3_exp(dp, ex);
                se3_mul(ex, bk, pose);                                                      // every thread moves its own copy of the pose: the same arithmetic, the same result
                if (MS_PO_PROF && tid == 0) pc[3] += PO_CLOCK() - tc;
                temp = sweep();
            } else temp = DBL_MAX;
            const double scale = sc + 1e-3;
            rho = (current - temp) / scale;
            if (trials < P.debug_reject) rho = -1.0;                // (test hook: drives the ten-rejections Terminate path)
            if (rho > 0 && isfinite(temp)) {
                const double t3 = 2 * rho - 1;
                double alpha = 1. - t3 * t3 * t3;                                          // (pow(x, 3) of the reference to an ulp or two, without the ~300 instructions of pow)
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha);
                ni = 2; current = temp; chi2_carried = temp;
                accept();                                                                   // (the trial's half of s_sum becomes the current one; the next sweep writes the other, behind its first barrier)
            } else {
                lambda *= ni; ni *= 2;
#pragma unroll
                for (int a = 0; a < 7; ++a) pose[a] = bk[a];                                // pop()
                if (!isfinite(lambda)) break;
            }
            ++qmax; ++trials;
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0 || !isfinite(lambda)) { stop = 1; ++it; break; }
    }
    // the chi2 per observation of the accepted state: from the registers; the observations beyond them are evaluated once more
#pragma unroll
    for (int j = 0; j < PO_K; ++j) if ((have >> j) & 1u) { P.chi2_obs[tid + PO_OT * j] = c2[j]; if (pk) pk_chi2[tid + PO_OT * j] = c2[j]; }
    if (overflow && tid < PO_OT)
        for (int o = tid + PO_OT * PO_K; o < P.n_obs; o += PO_OT) {
            if (P.obs_pose[o] != pi) continue;
            double e[2], Jp[12], Jl[6];
            proj_edge<true>(pose, P.point0 + 3 * (size_t)P.obs_point[o], P.obs_uv + 2 * (size_t)o, e, Jp, Jl);     // (the arithmetic of the sweeps)
            P.chi2_obs[o] = P.obs_info[o] * (e[0] * e[0] + e[1] * e[1]);
            if (pk) pk_chi2[o] = P.chi2_obs[o];
        }
    const double chi2_final = chi2_carried;                                 // the sweep that was accepted last evaluated exactly this state
    if (tid == PO_OT) { P.stats[9] = (double)pc[1]; if (pk) pk[9] = (double)pc[1]; }
    if (tid == 0) {
        // [13], [14]: cycles of the iterations and of what came before them (the only stamps with MS_PO_PROF off); [9] is the edge wave's (above)
        const double st[16] = {(double)it, (double)trials, (double)stop, lambda, chi2_init, chi2_final, isfinite(chi2_final) ? 1.0 : 0.0, 0.0,
                               (double)pc[0], 0.0, (double)pc[2], (double)pc[3], 0.0, (double)(clock64() - t_begin), (double)(t_begin - t_kernel), 0.0};
        for (int a = 0; a < 7; ++a) { P.pose[7 * (size_t)pi + a] = pose[a]; if (pk) pk_pose[7 * (size_t)pi + a] = pose[a]; }
#pragma unroll
        for (int q = 0; q < 16; ++q) if (q != 9) { P.stats[q] = st[q]; if (pk) pk[q] = st[q]; }
    }
}


// ---------------------------------------------------------------- one free pose + free points: stage 1 of localBundleAdjust (bundle_adjuster.cpp:251-252, :268, :322-333)
// ONE free keyframe, every other keyframe fixed, the map points free: the reduced camera system is 6 x 6 whatever the window's size.  The general kernel
// still walked it through its whole machinery (pass sets, LDS tile, windowed Cholesky of one block, ~7 team barriers per damped solve): 0.60 ms for the
// 8 iterations of a C4 window on 32 workgroups, 3.3 ms for 256 such windows with one workgroup each.  Here a point belongs to a GROUP of G = 1 .. 8
// neighbouring lanes (G chosen so that the launch's lanes cover the points): the group's lanes share the point's observations out, their sums meet over
// the DPP network.  With a lane group per point (ONE: a single window on a team) the point -- position, Hll, bl, W = sum of Jp^T w Jl over its observations
// in the free keyframe -- stays in the group's REGISTERS from the first linearisation to the last trial; a batch (one workgroup per window, several points
// per group) keeps these records in memory that only the owning workgroup touches.  Per iteration: linearise, and in the same pass every group forms its
// point's W (Hll + lambda I)^-1 W^T and W (Hll + lambda I)^-1 bl (3 x 3 Cholesky, as in schur_fused).  Only ~1 point in 4 is seen by the free keyframe, so
// these 27-value contributions are SPARSE: the contributing lanes write them side by side into the wave's LDS slab, the wave adds the entries up and issues
// one atomic per value (27 uniform-address atomics per observation cost a wave reduction each: 5 x the time of everything else).  ONE reduction over the
// team, the 6 x 6 Cholesky in every thread (as in k_ba_pose_only), then each group back-substitutes its point, moves it, and evaluates the robust chi2 of
// its observations at the trial state; a second reduction (chi2, gain denominator) decides.  Trial points are separate from the accepted ones: no backup /
// restore pass.  Observation data comes from copies sorted by point (pose index, u, v, information side by side: one round trip instead of three dependent
// ones), the poses from an LDS table.  Same LM schedule and the same arithmetic per edge as k_ba_lm.
constexpr int OP_NT = 512, OP_NW = OP_NT / 64, OP_NV = 64, OP_MAX_LDS_POSES = 512;
#ifndef OP_PROF_GL
#define OP_PROF_GL 0
#endif
constexpr int OP_S0 = 32, OP_BAD = 59, OP_MAX = 63;          // accumulators: 0..20 Hpp, 21..26 bp | 32..52 Schur matrix terms, 53..58 rhs terms, 59 unsound point blocks | 63 largest diagonal entry
__host__ __device__ constexpr int op_slab_cap(int lgG) { return lgG == 0 ? 64 : 32; }     // entries of a wave's staging slab
constexpr size_t kOpLdsBytes = (((size_t)7 * OP_MAX_LDS_POSES + 2) + (size_t)OP_NW * 27 * 64) * sizeof(double);     // dynamic LDS at most: the pose table + the slabs (136 KB; ~11 KB are static)

template <int CTRL>
__device__ __forceinline__ double dpp_move_d(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// sum over a group of 2^lg neighbouring lanes (aligned), the result in every lane of the group: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror
__device__ __forceinline__ double group_sum_d(double v, int lg) {
    if (lg >= 1) v += dpp_move_d<0xB1>(v);
    if (lg >= 2) v += dpp_move_d<0x4E>(v);
    if (lg >= 3) v += dpp_move_d<0x141>(v);
    return v;
}
__device__ __forceinline__ int group_sum_i(int v, int lg) {
    if (lg >= 1) v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
    if (lg >= 2) v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);
    if (lg >= 3) v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);
    return v;
}

// Hll + lambda I = L L^T (3 x 3, packed 00 01 02 11 12 22); returns false when a pivot is not positive
struct Chol3 { double i11, l21, l31, i22, l32, i33; };
__device__ __forceinline__ bool chol3(const double *H, double lambda, Chol3 &c) {
    const double a = H[0] + lambda, d = H[3] + lambda, f = H[5] + lambda;
    c.i11 = rsqrt_d(a); c.l21 = H[1] * c.i11; c.l31 = H[2] * c.i11;
    const double d2 = d - c.l21 * c.l21;
    c.i22 = rsqrt_d(d2); c.l32 = (H[4] - c.l31 * c.l21) * c.i22;
    const double d3 = f - c.l31 * c.l31 - c.l32 * c.l32;
    c.i33 = rsqrt_d(d3);
    return a > 0 && d2 > 0 && d3 > 0 && isfinite(c.i11 * c.i22 * c.i33);
}

// Sparse sums: a few lanes of a wave hold 27 values each whose sums go into acc[0 .. 27).  The contributors write their values side by side into the wave's
// slab ([27][cap], value-major: neighbouring entries in neighbouring banks; entry = a running index over the wave's contributors), then the wave adds the
// entries up -- value k by lanes 2k and 2k + 1 (even / odd entries) -- and issues one atomic per value.  A contributor beyond the slab's capacity adds its
// values with plain atomics.
__device__ __forceinline__ void slab_put(MS_LDS double *slab, int cap, MS_LDS double *acc, int idx, const double (&v)[27]) {
    if (idx < cap) {
#pragma unroll
        for (int k = 0; k < 27; ++k) slab[k * cap + idx] = v[k];
    } else {
#pragma unroll
        for (int k = 0; k < 27; ++k) lds_addd(acc + k, v[k]);
    }
}
__device__ __forceinline__ void slab_sum(MS_LDS double *slab, int cap, MS_LDS double *acc, int n, int lane) {      // wave-uniform call, n wave-uniform
    if (n == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int ns = min(n, cap);
    if (lane < 54) {
        const int k = lane >> 1;
        double t = 0;
        for (int e = lane & 1; e < ns; e += 2) t += slab[k * cap + e];
        t += dpp_move_d<0xB1>(t);
        if ((lane & 1) == 0) lds_addd(acc + k, t);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ int lanes_below(unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

struct OpPoint { double X[3], H[6], bl[3], W[18]; int nf, i0, i1; bool pfree; };       // nf: the point's observations in the free keyframe (group total)

template <bool ONE>     // ONE: the launch has a lane group per point -- the point's record lives in registers for the whole solve
__global__ __launch_bounds__(OP_NT) void k_ba_one_pose(const BaProb *probs, int team, int lgG, int pose_doubles) {
    extern __shared__ __attribute__((aligned(16))) double op_lds[];      // [7 n_pose] the poses (when they fit), then the waves' staging slabs [8][27][cap]
    __shared__ double s_acc[OP_NV], s_sum[OP_NV], s_w[OP_NW * 2], s_part[OP_NW * OP_NV];
    __shared__ double s_eC[PO_MAXE][8], s_eM[PO_MAXE][8], s_eG[PO_MAXE][36], s_eW[PO_MAXE][36], s_Hc[24], s_const;
    __shared__ int s_eSide[PO_MAXE], s_ne;
    __shared__ double s_J[36];
    const long long t_kernel = clock64();
    const BaProb &P = probs[blockIdx.x / (unsigned)team];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, rank = team > 1 ? (int)(blockIdx.x % (unsigned)team) : 0;
    const int G = 1 << lgG, gl = rank * OP_NT + tid, sub = gl & (G - 1), slot = gl >> lgG, nslot = (team * OP_NT) >> lgG;
    const int pi = P.free2pose[0], n_point = P.n_point, cap = op_slab_cap(lgG);
    const bool lds_poses = pose_doubles > 0;                              // (the host's decision for the whole launch: every problem's poses fit the table then)
    MS_LDS double *slab = (MS_LDS double *)op_lds + pose_doubles + wave * 27 * cap;
    MS_LDS double *acc = (MS_LDS double *)s_acc;
    const MS_GLOBAL int32_t *o_pose = (const MS_GLOBAL int32_t *)P.op_pose, *o_idx = (const MS_GLOBAL int32_t *)P.op_o, *pt_start = (const MS_GLOBAL int32_t *)P.pt_start;
    const MS_GLOBAL double *o_uvi = (const MS_GLOBAL double *)P.op_uvi;
    MS_GLOBAL double *rec = (MS_GLOBAL double *)P.op_rec;                 // (not ONE) [28][n_point]: Hll 0-5, bl 6-8, W 9-26, 27 = observations in the free keyframe, -1 for a fixed point
    MS_GLOBAL double *point = (MS_GLOBAL double *)P.point, *trial = (MS_GLOBAL double *)P.point_bk;
    const MS_GLOBAL uint8_t *pfix = (const MS_GLOBAL uint8_t *)P.point_fixed;
    const int nrounds = ONE ? 1 : (n_point + nslot - 1) / nslot;
    int seq = 0;
    // (the free pose's seven entries are written ONCE, at the end, by the launch's first lane: the team's barriers carry no cache maintenance, and two workgroups on
    //  different XCDs writing the same line -- one here, one at the end -- would leave the order of the two write-backs to chance)
    for (int i = gl; i < 7 * P.n_pose; i += team * OP_NT) if (i / 7 != pi) P.pose[i] = P.pose0[i];
    if (lds_poses) for (int i = tid; i < 7 * P.n_pose; i += OP_NT) op_lds[i] = P.pose0[i];
    OpPoint ps;
    double Xt[3] = {0, 0, 0};                                             // ONE: the trial point
    auto point_begin = [&](int l, bool from_point0) {                     // the point's observation range, its flag and its position
        ps.i0 = ps.i1 = 0; ps.pfree = false; ps.nf = 0; ps.X[0] = ps.X[1] = ps.X[2] = 0;
        if (l < n_point) {
            ps.i0 = pt_start[l]; ps.i1 = pt_start[l + 1]; ps.pfree = !(pfix && pfix[l]);
            const MS_GLOBAL double *src = from_point0 ? (const MS_GLOBAL double *)P.point0 : point;
            ps.X[0] = src[3 * (size_t)l]; ps.X[1] = src[3 * (size_t)l + 1]; ps.X[2] = src[3 * (size_t)l + 2];
        }
    };
    if (ONE) point_begin(slot, true);
    else for (int l = slot; l < n_point; l += nslot) if (sub == 0) { point[3 * (size_t)l] = P.point0[3 * (size_t)l]; point[3 * (size_t)l + 1] = P.point0[3 * (size_t)l + 1]; point[3 * (size_t)l + 2] = P.point0[3 * (size_t)l + 2]; }
    if (tid < OP_NV) s_acc[tid] = 0;
    if (tid < 24) s_Hc[tid] = 0;
    if (tid == 0) { s_ne = 0; s_const = 0; }
    __syncthreads();
    double pose[7];
#pragma unroll
    for (int a = 0; a < 7; ++a) pose[a] = P.pose0[7 * (size_t)pi + a];
    // ---- the SE3 edges: an edge between fixed poses is a constant, one that touches the free pose leaves its fixed side's transform, -J^T W and J^T W J
    //      in LDS (its Jacobian does not depend on the free pose), exactly as in k_ba_pose_only.  They belong to the LAST wave of the team's LAST workgroup:
    //      an edge is ~1 700 dependent instructions (quaternion products, the SE3 logarithm) in one lane, 17 k cycles per sweep, and on the team's first wave
    //      it sat in front of that wave's points -- the last lanes of a launch have the fewest points (none when the lanes outnumber them)
    const bool edge_wg = rank == team - 1;
    if (edge_wg && tid >= OP_NT - 64) {                                     // (the last wave: the one that takes the edges in every sweep)
        double cc = po_edges_setup(P, pi, tid - (OP_NT - 64), &s_ne, s_eSide, &s_eC[0][0], &s_eM[0][0], &s_eG[0][0], &s_eW[0][0], s_Hc, s_J);
        cc = wave_sum_d(cc);
        if (tid == OP_NT - 64) lds_addd((MS_LDS double *)&s_const, cc);
    }
    __syncthreads();
    const int ne = edge_wg ? s_ne : 0, et = tid - (OP_NT - 64);             // lane et of the last wave takes edge et
    // The workgroups of a team exchange nothing but their partial sums (points, records and trial points belong to the workgroup that owns the lane group):
    // these go through agent-scope atomic stores and loads, which are coherent across the XCDs' L2s by themselves, so the team barriers carry no L2 write-back
    // and no invalidate (team_sync_light) -- the observation data stays in the caches from sweep to sweep
    auto put = [](double *q, double v) { __hip_atomic_store(reinterpret_cast<unsigned long long *>(q), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto get = [](const double *q) { return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); };
    // ---- sums over the team: the LDS accumulators s_acc[lo .. hi) (contributions of this workgroup) -> s_sum[lo .. hi), the same bits in every
    //      workgroup (partials combined in rank order); slot OP_MAX is combined as a maximum
    // after the barrier every thread fetches one or two of the partials (value tid & 63 of ranks tid >> 6, + 8: the loads of one thread wait for each other),
    // the eight rows meet in LDS
    auto team_combine = [&](const double *part, double mine, bool mine_slot) {
        const int k = tid & 63, r0 = tid >> 6;
        double v = 0;
        for (int r = r0; r < team; r += OP_NW) { const double x = get(part + (size_t)r * OP_NV + k); v = k == OP_MAX ? fmax(v, x) : v + x; }
        s_part[r0 * OP_NV + k] = v;
        __syncthreads();
        if (mine_slot) {
            double t = s_part[tid];
            for (int r = 1; r < OP_NW; ++r) t = tid == OP_MAX ? fmax(t, s_part[r * OP_NV + tid]) : t + s_part[r * OP_NV + tid];
            mine = t;
        }
        return mine;
    };
    auto vec_reduce = [&](int lo, int hi, bool with_max) {
        __syncthreads();
        const bool mine_slot = (tid >= lo && tid < hi) || (with_max && tid == OP_MAX);
        double mine = 0;
        if (mine_slot) { mine = s_acc[tid]; s_acc[tid] = 0; }
        if (team > 1) {
            double *part = P.op_red + (size_t)(seq & 1) * team * OP_NV;
            if (mine_slot) put(part + (size_t)rank * OP_NV + tid, mine);
            team_sync_light(P, team);
            mine = team_combine(part, mine, mine_slot);
        }
        ++seq;
        if (mine_slot) s_sum[tid] = mine;
        __syncthreads();
    };
    // two plain sums (fixed order: lanes, waves, ranks) -> s_sum[28], s_sum[29]
    auto pair_reduce = [&](double a, double b) {
        a = wave_sum_d(a); b = wave_sum_d(b);
        __syncthreads();
        if (lane == 0) { s_w[2 * wave] = a; s_w[2 * wave + 1] = b; }
        __syncthreads();
        double mine = 0;
        const bool mine_slot = tid == 28 || tid == 29;
        if (mine_slot) { for (int w = 0; w < OP_NW; ++w) mine += s_w[2 * w + tid - 28]; }
        if (team > 1) {
            double *part = P.op_red + (size_t)(seq & 1) * team * OP_NV;
            if (mine_slot) put(part + (size_t)rank * OP_NV + tid, mine);
            team_sync_light(P, team);
            mine = team_combine(part, mine, mine_slot);
        }
        ++seq;
        if (mine_slot) s_sum[tid] = mine;
        __syncthreads();
    };
    auto load_pose = [&](int pc, const double (&cur)[7], double (&pz)[7]) {
        if (pc == pi) {
#pragma unroll
            for (int a = 0; a < 7; ++a) pz[a] = cur[a];
        } else if (lds_poses) {                                           // (explicit address spaces: a flat load would wait for the prefetched global data as well)
            const MS_LDS double *q = (const MS_LDS double *)op_lds + 7 * pc;
#pragma unroll
            for (int a = 0; a < 7; ++a) pz[a] = q[a];
        } else {
            const MS_GLOBAL double *q = (const MS_GLOBAL double *)P.pose0 + 7 * (size_t)pc;
#pragma unroll
            for (int a = 0; a < 7; ++a) pz[a] = q[a];
        }
    };
    // ---- robust chi2 of this lane's share of the observations [i0, i1) of a point at X, with the free pose at `cur`
    auto chi2_share = [&](int i0, int i1, const double (&X)[3], const double (&cur)[7], bool store) {
        double sum = 0;
        int ii = i0 + sub, pn = 0;
        double un = 0, vn = 0, fn = 0;
        if (ii < i1) { pn = o_pose[ii]; un = o_uvi[3 * (size_t)ii]; vn = o_uvi[3 * (size_t)ii + 1]; fn = o_uvi[3 * (size_t)ii + 2]; }
        while (ii < i1) {
            const int pc = pn, icur = ii;
            const double uvc[2] = {un, vn}, infc = fn;
            ii += G;
            if (ii < i1) { pn = o_pose[ii]; un = o_uvi[3 * (size_t)ii]; vn = o_uvi[3 * (size_t)ii + 1]; fn = o_uvi[3 * (size_t)ii + 2]; }
            double pz[7], e[2], r, w;
            load_pose(pc, cur, pz);
            proj_edge<false>(pz, X, uvc, e, nullptr, nullptr);
            const double chi2 = infc * (e[0] * e[0] + e[1] * e[1]);
            huber(chi2, P.huber, r, w);
            if (store) P.chi2_obs[o_idx[icur]] = chi2;
            sum += r;
        }
        return sum;
    };
    // the SE3 edges at the free pose `cur` (lanes < ne of the team's first workgroup): chi2, and with lin the gradient into the accumulators
    // (an edge's error at the accepted pose is the error of the trial that was accepted: it is kept, e_acc, and the linearisation only multiplies it)
    double e_acc[6] = {0, 0, 0, 0, 0, 0}, e_try[6] = {0, 0, 0, 0, 0, 0};
    auto edge_part = [&](const double (&cur)[7], bool lin) {
        double sum = 0;
        if (et >= 0 && et < ne) {
            if (lin) {
#pragma unroll
                for (int a = 0; a < 6; ++a) { double v = 0; for (int c = 0; c < 6; ++c) v += s_eG[et][6 * a + c] * e_acc[c]; lds_addd(acc + 21 + a, v); }
            } else {
                double Bm[7], We[6];
                if (s_eSide[et] == 0) se3_mul(s_eC[et], cur, Bm);
                else { double Tjinv[7], A2[7]; se3_inv(cur, Tjinv); se3_mul(Tjinv, s_eM[et], A2); se3_mul(A2, s_eC[et], Bm); }
                se3_log(Bm, e_try);
                for (int i = 0; i < 6; ++i) { double v = 0; for (int j = 0; j < 6; ++j) v += s_eW[et][6 * i + j] * e_try[j]; We[i] = v; }
                for (int i = 0; i < 6; ++i) sum += e_try[i] * We[i];
            }
        }
        if (!lin && edge_wg && tid == 0) sum += s_const;
        return sum;
    };
    auto total_chi2 = [&](const double (&cur)[7], bool store) {          // at the accepted state
        double sum = 0;
        for (int rnd = 0; rnd < nrounds; ++rnd) {
            if (!ONE) point_begin(slot + rnd * nslot, false);
            sum += chi2_share(ps.i0, ps.i1, ps.X, cur, store);
        }
        sum += edge_part(cur, false);
        pair_reduce(sum, 0.0);
        return s_sum[28];
    };
    // ---- the point's part of the damped system: -W (Hll + lambda I)^-1 W^T (upper triangle, 21) and -W (Hll + lambda I)^-1 bl (6)
    auto schur_point = [&](double lambda, double (&Sv)[27]) {
        Chol3 c;
        const bool sound = chol3(ps.H, lambda, c);
        const double u0 = ps.bl[0] * c.i11, u1 = (ps.bl[1] - c.l21 * u0) * c.i22, u2 = (ps.bl[2] - c.l31 * u0 - c.l32 * u1) * c.i33;
        double Z[18];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const double z0 = ps.W[3 * r] * c.i11, z1 = (ps.W[3 * r + 1] - z0 * c.l21) * c.i22, z2 = (ps.W[3 * r + 2] - z0 * c.l31 - z1 * c.l32) * c.i33;
            Z[3 * r] = z0; Z[3 * r + 1] = z1; Z[3 * r + 2] = z2;
            Sv[21 + r] = -(z0 * u0 + z1 * u1 + z2 * u2);
        }
        int k = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b2 = a; b2 < 6; ++b2) { Sv[k] = -(Z[3 * a] * Z[3 * b2] + Z[3 * a + 1] * Z[3 * b2 + 1] + Z[3 * a + 2] * Z[3 * b2 + 2]); ++k; }
        return sound;
    };
    // one point's Schur terms into the accumulators (the group's first lane contributes); wave-uniform call
    auto schur_stage = [&](bool live, double lambda) {
        double Sv[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) Sv[k] = 0;
        bool contributes = false;
        if (live && sub == 0 && ps.pfree) {
            if (ps.nf > 0) { contributes = true; if (!schur_point(lambda, Sv)) lds_addd(acc + OP_BAD, 1.0); }
            else { Chol3 c; if (!chol3(ps.H, lambda, c)) lds_addd(acc + OP_BAD, 1.0); }
        }
        const unsigned long long m = __ballot(contributes);
        if (contributes) slab_put(slab, cap, acc + OP_S0, lanes_below(m), Sv);
        slab_sum(slab, cap, acc + OP_S0, (int)__popcll(m), lane);
    };
    auto rec_store = [&](int l) {
#pragma unroll
        for (int q = 0; q < 6; ++q) rec[(size_t)q * n_point + l] = ps.H[q];
#pragma unroll
        for (int q = 0; q < 3; ++q) rec[(size_t)(6 + q) * n_point + l] = ps.bl[q];
        if (ps.nf > 0) {
#pragma unroll
            for (int q = 0; q < 18; ++q) rec[(size_t)(9 + q) * n_point + l] = ps.W[q];
        }
        rec[(size_t)27 * n_point + l] = ps.pfree ? (double)ps.nf : -1.0;
    };
    auto rec_load = [&](int l) {
        const double nfv = rec[(size_t)27 * n_point + l];
        ps.nf = nfv > 0 ? (int)nfv : 0;
#pragma unroll
        for (int q = 0; q < 6; ++q) ps.H[q] = rec[(size_t)q * n_point + l];
#pragma unroll
        for (int q = 0; q < 3; ++q) ps.bl[q] = rec[(size_t)(6 + q) * n_point + l];
        if (nfv > 0) {
#pragma unroll
            for (int q = 0; q < 18; ++q) ps.W[q] = rec[(size_t)(9 + q) * n_point + l];
        }
    };
    double lambda = 0, ni = 2;
    int it = 0, trials = 0, stop = 0;
    long long cyc[6] = {0, 0, 0, 0, 0, 0}, tq = clock64();                 // 0 chi2 sweeps, 1 linearise (+ Schur terms), 2 the reduction after it, 3 6 x 6 solve, 4 points + trial chi2, 5 the reduction after it
    const long long t_begin = tq;
#define OP_LAP(i) do { const long long t_ = clock64(); cyc[i] += t_ - tq; tq = t_; } while (0)
    const double chi2_init = total_chi2(pose, false);
#pragma unroll
    for (int a = 0; a < 6; ++a) e_acc[a] = e_try[a];
    OP_LAP(0);
    double chi2_carried = chi2_init;
    for (it = 0; it < P.max_iters; ++it) {
        double current = chi2_carried, temp = current;
        // ---- linearisation: per point Hll, bl, W; the free pose's block and gradient (A: 21 + 6, this lane's part).  From the second iteration on lambda is
        //      known and the points' Schur terms follow in the same pass: one reduction instead of two
        double md = 0;
        for (int rnd = 0; rnd < nrounds; ++rnd) {
            const int l = slot + rnd * nslot;
            const bool live = l < n_point;
            if (!ONE) point_begin(l, false);
#pragma unroll
            for (int q = 0; q < 6; ++q) ps.H[q] = 0;
#pragma unroll
            for (int q = 0; q < 3; ++q) ps.bl[q] = 0;
#pragma unroll
            for (int q = 0; q < 18; ++q) ps.W[q] = 0;
            const int i1 = ps.i1;
            int jj = i1, jj2 = i1;                                       // this lane's first and second observation in the free keyframe
            bool more = false;                                           // ... and there are further ones (found by scanning: never in a SLAM window, where a keyframe sees a point once)
            double fu = 0, fv = 0, finf = 0;                             // the first one's measurement stays in registers (a reload after the loop is a round trip to L2)
            bool fkept = false;
            if (ps.pfree) {                                              // the point's own block and gradient: every observation of the share
                int ii = ps.i0 + sub, pn = 0;
                double un = 0, vn = 0, fn = 0;
                if (ii < i1) { pn = o_pose[ii]; un = o_uvi[3 * (size_t)ii]; vn = o_uvi[3 * (size_t)ii + 1]; fn = o_uvi[3 * (size_t)ii + 2]; }
                while (ii < i1) {
                    const int pc = pn;
                    const double uvc[2] = {un, vn}, infc = fn;
                    if (pc == pi) { if (jj == i1) { jj = ii; fu = un; fv = vn; finf = fn; fkept = true; } else if (jj2 == i1) jj2 = ii; else more = true; }
                    ii += G;
                    if (ii < i1) { pn = o_pose[ii]; un = o_uvi[3 * (size_t)ii]; vn = o_uvi[3 * (size_t)ii + 1]; fn = o_uvi[3 * (size_t)ii + 2]; }
                    double pz[7], e[2], Jp[12], Jl[6], r, w;
                    load_pose(pc, pose, pz);
                    proj_edge<true>(pz, ps.X, uvc, e, Jp, Jl);
                    const double chi2 = infc * (e[0] * e[0] + e[1] * e[1]);
                    huber(chi2, P.huber, r, w);
                    const double wi = w * infc;
                    ps.H[0] += wi * (Jl[0] * Jl[0] + Jl[3] * Jl[3]); ps.H[1] += wi * (Jl[0] * Jl[1] + Jl[3] * Jl[4]); ps.H[2] += wi * (Jl[0] * Jl[2] + Jl[3] * Jl[5]);
                    ps.H[3] += wi * (Jl[1] * Jl[1] + Jl[4] * Jl[4]); ps.H[4] += wi * (Jl[1] * Jl[2] + Jl[4] * Jl[5]); ps.H[5] += wi * (Jl[2] * Jl[2] + Jl[5] * Jl[5]);
#pragma unroll
                    for (int c = 0; c < 3; ++c) ps.bl[c] -= wi * (Jl[c] * e[0] + Jl[3 + c] * e[1]);
                }
            } else { jj = ps.i0 + sub; while (jj < i1 && o_pose[jj] != pi) jj += G; more = true; }     // (a fixed point: its share is scanned)
            // the observations in the free keyframe once more (one lane in four has one; a wave-uniform loop): the pose's block and gradient go straight into the
            // wave's slab, W = Jp^T w Jl stays with the point -- keeping 27 more sums in registers through the loop above made the kernel spill
            int nf = 0, cnt = 0;
            for (;;) {
                const bool has = jj < i1;
                const unsigned long long m = __ballot(has);
                if (m == 0) break;
                if (has) {
                    if (!fkept) { fu = o_uvi[3 * (size_t)jj]; fv = o_uvi[3 * (size_t)jj + 1]; finf = o_uvi[3 * (size_t)jj + 2]; }
                    fkept = false;
                    const double uvc[2] = {fu, fv}, infc = finf;
                    double e[2], Jp[12], Jl[6], r, w, A[27];
                    proj_edge<true>(pose, ps.X, uvc, e, Jp, Jl);
                    const double chi2 = infc * (e[0] * e[0] + e[1] * e[1]);
                    huber(chi2, P.huber, r, w);
                    const double wi = w * infc;
                    int k = 0;
#pragma unroll
                    for (int a = 0; a < 6; ++a) {
                        A[21 + a] = -(Jp[a] * e[0] + Jp[6 + a] * e[1]) * wi;
#pragma unroll
                        for (int b2 = a; b2 < 6; ++b2) { A[k] = wi * (Jp[a] * Jp[b2] + Jp[6 + a] * Jp[6 + b2]); ++k; }
                    }
                    slab_put(slab, cap, acc, cnt + lanes_below(m), A);
                    if (ps.pfree) {
                        ++nf;
#pragma unroll
                        for (int r2 = 0; r2 < 6; ++r2)
#pragma unroll
                            for (int c = 0; c < 3; ++c) ps.W[3 * r2 + c] += wi * (Jp[r2] * Jl[c] + Jp[6 + r2] * Jl[3 + c]);
                    }
                    if (more && jj2 == i1) { jj += G; while (jj < i1 && o_pose[jj] != pi) jj += G; }      // the rare cases: scan on
                    else { jj = jj2; jj2 = i1; }
                }
                cnt += (int)__popcll(m);
            }
            slab_sum(slab, cap, acc, cnt, lane);
            ps.nf = group_sum_i(nf, lgG);
#pragma unroll
            for (int q = 0; q < 6; ++q) ps.H[q] = group_sum_d(ps.H[q], lgG);
#pragma unroll
            for (int q = 0; q < 3; ++q) ps.bl[q] = group_sum_d(ps.bl[q], lgG);
            if (ps.nf > 0) {
#pragma unroll
                for (int q = 0; q < 18; ++q) ps.W[q] = group_sum_d(ps.W[q], lgG);
            }
            if (live && sub == 0) {
                if (!ONE) rec_store(l);
                if (ps.pfree) md = fmax(md, fmax(fabs(ps.H[0]), fmax(fabs(ps.H[3]), fabs(ps.H[5]))));
            }
            if (it > 0) schur_stage(live, lambda);
        }
        (void)edge_part(pose, true);
        OP_LAP(1);
        if (edge_wg && tid < 21) lds_addd(acc + tid, s_Hc[tid]);          // the edges' constant J^T W J: into the sums, so that every workgroup of the team gets it
        if (it == 0) {
            for (int off = 32; off > 0; off >>= 1) md = fmax(md, __shfl_xor(md, off, 64));
            if (lane == 0) (void)atomicMax(reinterpret_cast<unsigned long long *>(&s_acc[OP_MAX]), (unsigned long long)__double_as_longlong(md));
        }
        vec_reduce(0, it == 0 ? 27 : OP_BAD + 1, it == 0);
        OP_LAP(2);
        const double *Hp = s_sum, *bp = s_sum + 21;                         // (they stay in LDS: 27 doubles in every thread's registers were a part of the kernel's spills)
        if (it == 0) {                                                       // computeLambdaInit: 1e-5 x the largest diagonal entry of the whole system
            const double mdp = fmax(fmax(fmax(fabs(Hp[0]), fabs(Hp[6])), fmax(fabs(Hp[11]), fabs(Hp[15]))), fmax(fabs(Hp[18]), fabs(Hp[20])));
            lambda = 1e-5 * fmax(mdp, s_sum[OP_MAX]); ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        bool have_schur = it > 0;                                            // the Schur terms at this lambda are in s_sum already
        do {
            if (!have_schur) {
                for (int rnd = 0; rnd < nrounds; ++rnd) {
                    const int l = slot + rnd * nslot;
                    const bool live = l < n_point;
                    if (!ONE) { ps.pfree = false; ps.nf = 0; if (live && sub == 0) { ps.pfree = !(pfix && pfix[l]); if (ps.pfree) rec_load(l); } }
                    schur_stage(live, lambda);
                }
                OP_LAP(1);
                vec_reduce(OP_S0, OP_BAD + 1, false);
                OP_LAP(2);
            }
            have_schur = false;
            // (S + lambda I) dp = y: the 6 x 6 Cholesky by every thread for itself, as in k_ba_pose_only
            double Lm[21], dpv[6], yv[6];
            bool ok2 = s_sum[OP_BAD] == 0.0;
            {
                int k = 0;
#pragma unroll
                for (int a = 0; a < 6; ++a)
#pragma unroll
                    for (int c = a; c < 6; ++c) { Lm[c * (c + 1) / 2 + a] = Hp[k] + s_sum[OP_S0 + k] + (a == c ? lambda : 0.0); ++k; }
#pragma unroll
                for (int a = 0; a < 6; ++a) yv[a] = bp[a] + s_sum[OP_S0 + 21 + a];
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                double d = Lm[j * (j + 1) / 2 + j];
#pragma unroll
                for (int k = 0; k < j; ++k) d -= Lm[j * (j + 1) / 2 + k] * Lm[j * (j + 1) / 2 + k];
                if (!(d > 0) || !isfinite(d)) ok2 = false;
                const double inv = rsqrt_d(d);
                Lm[j * (j + 1) / 2 + j] = inv;
#pragma unroll
                for (int i = j + 1; i < 6; ++i) {
                    double v = Lm[i * (i + 1) / 2 + j];
#pragma unroll
                    for (int k = 0; k < j; ++k) v -= Lm[i * (i + 1) / 2 + k] * Lm[j * (j + 1) / 2 + k];
                    Lm[i * (i + 1) / 2 + j] = v * inv;
                }
            }
            {
                double y2[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    double v = yv[i];
#pragma unroll
                    for (int k = 0; k < i; ++k) v -= Lm[i * (i + 1) / 2 + k] * y2[k];
                    y2[i] = v * Lm[i * (i + 1) / 2 + i];
                }
#pragma unroll
                for (int i = 5; i >= 0; --i) {
                    double v = y2[i];
#pragma unroll
                    for (int k = i + 1; k < 6; ++k) v -= Lm[k * (k + 1) / 2 + i] * dpv[k];
                    dpv[i] = v * Lm[i * (i + 1) / 2 + i];
                }
            }
            double trial_pose[7];
#pragma unroll
            for (int a = 0; a < 7; ++a) trial_pose[a] = pose[a];
            OP_LAP(3);
            if (ok2) {
                double ex[7], sc = 0, sum = 0;
                se3_exp(dpv, ex);
                se3_mul(ex, pose, trial_pose);
                if (gl == 0) { for (int a = 0; a < 6; ++a) sc += dpv[a] * (lambda * dpv[a] + bp[a]); }
                // ---- the points follow: dl = (Hll + lambda I)^-1 (bl - W^T dp), the trial point, its share of the gain denominator and the chi2 of its observations
                for (int rnd = 0; rnd < nrounds; ++rnd) {
                    const int l = slot + rnd * nslot;
                    const bool live = l < n_point;
                    if (!ONE) { point_begin(l, false); if (live && ps.pfree) rec_load(l); }
                    double Xn[3] = {ps.X[0], ps.X[1], ps.X[2]};
                    if (live && ps.pfree) {
                        double r0 = ps.bl[0], r1 = ps.bl[1], r2 = ps.bl[2];
                        if (ps.nf > 0) {
#pragma unroll
                            for (int a = 0; a < 6; ++a) { r0 -= ps.W[3 * a] * dpv[a]; r1 -= ps.W[3 * a + 1] * dpv[a]; r2 -= ps.W[3 * a + 2] * dpv[a]; }
                        }
                        Chol3 c;
                        (void)chol3(ps.H, lambda, c);
                        const double v0 = r0 * c.i11, v1 = (r1 - c.l21 * v0) * c.i22, v2 = (r2 - c.l31 * v0 - c.l32 * v1) * c.i33;       // L^-1 r
                        const double d2 = v2 * c.i33, d1 = (v1 - c.l32 * d2) * c.i22, d0 = (v0 - c.l21 * d1 - c.l31 * d2) * c.i11;         // L^-T (L^-1 r)
                        if (sub == 0) sc += d0 * (lambda * d0 + ps.bl[0]) + d1 * (lambda * d1 + ps.bl[1]) + d2 * (lambda * d2 + ps.bl[2]);
                        Xn[0] += d0; Xn[1] += d1; Xn[2] += d2;
                    }
                    if (ONE) { Xt[0] = Xn[0]; Xt[1] = Xn[1]; Xt[2] = Xn[2]; }
                    else if (live && sub == 0) { trial[3 * (size_t)l] = Xn[0]; trial[3 * (size_t)l + 1] = Xn[1]; trial[3 * (size_t)l + 2] = Xn[2]; }
                    sum += chi2_share(ps.i0, ps.i1, Xn, trial_pose, false);
                }
                sum += edge_part(trial_pose, false);
                OP_LAP(4);
                pair_reduce(sum, sc);
                OP_LAP(5);
                temp = s_sum[28];
            } else temp = DBL_MAX;
            const double scale = (ok2 ? s_sum[29] : 0.0) + 1e-3;
            rho = (current - temp) / scale;
            if (trials < P.debug_reject) rho = -1.0;                // (test hook: drives the ten-rejections Terminate path)
            if (rho > 0 && isfinite(temp)) {
                const double t3 = 2 * rho - 1;
                double alpha = 1. - t3 * t3 * t3;                                          // (pow(x, 3) of the reference to an ulp or two)
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha);
                ni = 2; current = temp; chi2_carried = temp;
#pragma unroll
                for (int a = 0; a < 7; ++a) pose[a] = trial_pose[a];
#pragma unroll
                for (int a = 0; a < 6; ++a) e_acc[a] = e_try[a];
                if (ONE) { ps.X[0] = Xt[0]; ps.X[1] = Xt[1]; ps.X[2] = Xt[2]; }
                else {
                    for (int l = slot; l < n_point; l += nslot)
                        if (sub == 0) { point[3 * (size_t)l] = trial[3 * (size_t)l]; point[3 * (size_t)l + 1] = trial[3 * (size_t)l + 1]; point[3 * (size_t)l + 2] = trial[3 * (size_t)l + 2]; }
                    __syncthreads();                                        // the group's other lanes read the moved points next
                }
            } else {
                lambda *= ni; ni *= 2;
                if (!isfinite(lambda)) break;
            }
            ++qmax; ++trials;
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0 || !isfinite(lambda)) { stop = 1; ++it; break; }
    }
    tq = clock64();
    const double chi2_final = total_chi2(pose, true);
    OP_LAP(0);
    if (ONE && slot < n_point && sub == 0) { point[3 * (size_t)slot] = ps.X[0]; point[3 * (size_t)slot + 1] = ps.X[1]; point[3 * (size_t)slot + 2] = ps.X[2]; }
    if (gl == 0) {
        const bool hung = team > 1 && __hip_atomic_load(P.flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;      // a team barrier gave up: the result is not to be trusted
        for (int a = 0; a < 7; ++a) P.pose[7 * (size_t)pi + a] = pose[a];
        P.stats[0] = it; P.stats[1] = trials; P.stats[2] = stop; P.stats[3] = lambda; P.stats[4] = chi2_init; P.stats[5] = hung ? NAN : chi2_final;
        P.stats[6] = (isfinite(chi2_final) && !hung) ? 1 : 0; P.stats[7] = hung ? 1 : 0;
        P.stats[15] = 0;
    }
    if (gl == (OP_PROF_GL < team * OP_NT ? OP_PROF_GL : 0)) {              // the phase stamps of one wave (-DOP_PROF_GL=<global lane>: another wave than the first)
        P.stats[8] = (double)cyc[0]; P.stats[9] = (double)cyc[1]; P.stats[10] = (double)cyc[2]; P.stats[11] = (double)cyc[3]; P.stats[12] = (double)cyc[4];
        P.stats[13] = (double)(clock64() - t_begin); P.stats[14] = (double)cyc[5];
        P.stats[15] = (double)(t_begin - t_kernel);                           // what comes before the first sweep (not part of [13])
    }
#undef OP_LAP
}

}  // namespace

// =================================================================================================
// host side
// =================================================================================================
struct ms_ba {
    ms_ctx *ctx = nullptr;
    int n = 0;
    std::vector<BaProb> host;          // device pointers inside
    std::vector<ms_ba_problem> dims;   // sizes only
    BaProb *d_probs = nullptr;
    // k_ba_lm's fused trial schedule: the same descriptors with Hpp / bp / Hll / bl pointing at a second set of arrays (behind d_probs on the device); alt_ptrs[i].Hpp == nullptr:
    // problem i has no second set and the whole handle runs the plain schedule
    struct AltPtrs { double *Hpp, *bp, *Hll, *bl; };
    std::vector<AltPtrs> alt_ptrs;
    std::vector<BaProb> host_alt;
    BaProb *d_probs_alt = nullptr;
    bool alt_ok = false;
    char *d_arena = nullptr;
    size_t arena_bytes = 0;
    int team = 0;                      // workgroups per problem of the next launch (ms_ba_set_team; 0 = automatic)
    int factor_team = 0;               // of these, workgroups in the distributed Cholesky of a large system (ms_ba_set_factor_team; 0 = automatic)
    std::vector<double> chol_tiles;    // per problem: row tiles a Cholesky panel touches on average
    int cus = 0;
    bool pose_only = false;            // every problem has ONE free pose and only fixed points: k_ba_pose_only instead of k_ba_lm (poseBundleAdjust)
    bool one_pose = false;             // every problem has ONE free pose and at least one free point: k_ba_one_pose (stage 1 of localBundleAdjust)
    int one_pose_lg = 0;               // log2 of the lanes per point of the last k_ba_one_pose launch
    bool last_one_pose = false;        // the last launch was k_ba_one_pose (its fallback after a barrier gave up is the same kernel with one workgroup)
    int launched_team = 1;             // team size of the last launch
    bool team_checked = true;          // the last team launch has been looked at (every problem's gave-up marker) and, if need be, repeated
    int debug_fail_barriers = 0;       // test hook: team barriers give up at once (ms_ba_debug_fail_team_barriers)
    int team_fallbacks = 0;            // launches repeated with one workgroup per problem after a team barrier gave up
    int solves = 0;                    // launches so far (ms_ba_copy_state refuses a source that has never been solved)
    // a single small problem (poseBundleAdjust): its results are packed and copied into h_result right behind the solver launch, so that ms_ba_download only
    // waits for ev_done and reads them -- no pack launch and device-to-host round trip after the wait (0.03 ms of a 0.08 ms solve)
    void *h_result = nullptr;          // page-locked, stays with the handle OBJECT (pooled per context)
    size_t h_result_bytes = 0, pack_doubles = 0;
    double *d_pack = nullptr;          // in the arena; nullptr: no eager results for this handle
    int32_t *h_verdict = nullptr;      // page-locked word, stays with the handle object: "a team barrier gave up somewhere in the last launch", collected behind every team launch
    bool verdict_eager = false;        // h_verdict belongs to the last launch
    bool eager = false;                // h_result holds the last launch's results
    bool self_packed = false;          // the last launch was k_ba_pose_only with BaProb::pack set: d_pack is already what k_ba_pack_result would write
    bool work_dirty = false;           // a pose-only handle whose work areas [work_lo[p], work_hi[p]) were never cleared: the general kernel needs them zero (ms_ba_solve)
    std::vector<size_t> work_lo, work_hi;
    bool quiet = false;                // everything this handle put on the stream is known to have finished (ev_done was seen, nothing enqueued since): ms_ba_destroy need not wait
    hipEvent_t ev_done = nullptr;      // the end of this handle's last launch (what reads its results waits for it ON THE HOST, politely: ba_wait_event)
    bool pending = false;
};

// Team launches of this process, per device: what is (or may still be) running, so that the workgroups of all concurrent team launches
// together never exceed the CUs (see ms_ba_solve)
struct TeamLaunch { hipEvent_t ev; hipStream_t stream; int wgs; bool live; };
static std::mutex g_team_mu;
static std::vector<TeamLaunch> g_team_live[64];
static int g_team_query_errors = 0;        // hipEventQuery answers other than success / not-ready seen by the admission list (guarded by g_team_mu)
#define MS_TRY_BA(x) do { int rc__ = (x); if (rc__ != MS_OK) return rc__; } while (0)

constexpr size_t kBaEagerMax = (size_t)64 << 10;        // results of a single problem up to this size are packed and copied behind every launch (ms_ba struct: h_result).
                                                        // (Tried at 512 KB, i.e. for a whole C4 window too: -0.025 ms for the window alone, but the front end of the same
                                                        //  sequence, running beside it, fell from 2.7-3.0 k to 2.0-2.2 k frames/s -- tools/together_ab.sh; kept for small problems)
constexpr size_t kBaStageMax = (size_t)4 << 20;        // creates whose inputs fit are uploaded from the context's page-locked staging block without a wait
static size_t ba_eager_max() { static const size_t v = std::getenv("MS_BA_EAGER_MAX") ? (size_t)std::atoll(std::getenv("MS_BA_EAGER_MAX")) : kBaEagerMax; return v; }     // (experiment knob)
// SE3 edges that touch pose `pi` (k_ba_pose_only / k_ba_one_pose keep their constants in PO_MAXE LDS slots; edges between fixed poses need none)
static int ba_edges_at_free_pose(const ms_ba_problem &Q, int pi) {
    int t = 0;
    for (int k = 0; k < Q.n_pose_edge; ++k) t += Q.edge_i[k] == pi || Q.edge_j[k] == pi;
    return t;
}
// the descriptors of the second linearisation set: host_alt[i] = host[i] with the four output arrays exchanged (rebuilt whenever host[] is about to be uploaded: the
// two must agree in everything else -- team sizes included)
static void ba_make_alt(ms_ba *B) {
    B->host_alt.assign(B->host.begin(), B->host.end());
    B->alt_ok = !B->host.empty() && B->alt_ptrs.size() == B->host.size();
    for (size_t i = 0; i < B->host_alt.size() && B->alt_ok; ++i) {
        const ms_ba::AltPtrs &a = B->alt_ptrs[i];
        if (!a.Hpp) { B->alt_ok = false; break; }
        BaProb &q = B->host_alt[i];
        q.Hpp = a.Hpp; q.bp = a.bp; q.Hll = a.Hll; q.bl = a.bl;
    }
}
// (both descriptor sets, in stream order)
static hipError_t ba_upload_descriptors(ms_ba *B, hipStream_t st) {
    ba_make_alt(B);
    hipError_t e = hipMemcpyAsync(B->d_probs, B->host.data(), sizeof(BaProb) * B->n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(B->d_probs_alt, B->host_alt.data(), sizeof(BaProb) * B->n, hipMemcpyHostToDevice, st);
    return e;
}
// The end of a handle's last launch, as an event the handle owns: whatever reads the launch's results waits for it ON THE HOST, politely (ba_wait_event), before it
// enqueues anything -- not with hipStreamSynchronize, and not with a wait packet behind the launch (round 4, tools/hog_probe.py: a stream with packets queued behind
// a 2 ms kernel slowed the front end of ANOTHER sequence 10 ... 90 x, one that holds one launch at a time 3 ... 15 x).
static int ba_launch_done(ms_ctx *c, ms_ba *B) {
    B->quiet = false;
    if (!B->ev_done) { MS_HIP(c, hipEventCreateWithFlags(&B->ev_done, hipEventDisableTiming)); ++g_ba_host_allocs; }
    MS_HIP(c, hipEventRecord(B->ev_done, c->stream));
    B->pending = true;
    return MS_OK;
}
static void ba_delete_object(ms_ba *B) {
    if (B->ev_done) (void)hipEventDestroy(B->ev_done);
    if (B->h_result) (void)hipHostFree(B->h_result);
    if (B->h_verdict) (void)hipHostFree(B->h_verdict);
    delete B;
}
// A host wait for a solver launch (milliseconds) that leaves the processor to the other sequences' threads: hipStreamSynchronize / hipEventSynchronize spin, and
// seven threads spinning in them made the thread that drives another sequence's front end 3 ... 15 x slower (tools/hog_probe.py: x 3.3 beside two streams whose
// threads wait in hipStreamSynchronize, x 1.09 beside the same streams with the threads asleep) -- on this runtime a waiting thread is not free for the others.
// Two phases: for the first ~120 us the event is polled with a yield in between (poseBundleAdjust's launch is 0.08 ms: a frame must not pay a timer's granularity
// for it), after that the thread sleeps 20 us at a time, with its timer slack set to 1 us once (the default 50 us slack turned every 25 us sleep into ~75 us:
// 0.18 ms in ms_ba_download for a 0.08 ms kernel, tools/pose_path_probe.py).
static int ba_wait_event(ms_ctx *c, hipEvent_t ev) {
    thread_local bool slack_set = false;
    static const int spin_us = std::getenv("MS_WAIT_SPIN_US") ? std::atoi(std::getenv("MS_WAIT_SPIN_US")) : 120;          // (experiment knobs)
    static const int sleep_us = std::getenv("MS_WAIT_SLEEP_US") ? std::atoi(std::getenv("MS_WAIT_SLEEP_US")) : 20;
    static const bool fine_slack = !std::getenv("MS_WAIT_COARSE");
    const auto t0 = std::chrono::steady_clock::now();
    for (bool spinning = spin_us > 0;;) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return MS_OK;
        if (q != hipErrorNotReady) { (void)hipGetLastError(); return ms_fail(c, MS_ERR_HIP, "waiting for a solver launch failed: %s", hipGetErrorString(q)); }
        if (spinning) {
            std::this_thread::yield();
            spinning = std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(spin_us);
        } else {
            if (!slack_set && fine_slack) { (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL); slack_set = true; }      // (nanoseconds; the calling thread only)
            std::this_thread::sleep_for(std::chrono::microseconds(sleep_us));
        }
    }
}
static int ba_wait_pending(ms_ba *B) {
    if (!B->pending) return MS_OK;
    MS_TRY_BA(ba_wait_event(B->ctx, B->ev_done));
    B->pending = false;
    B->quiet = true;
    return MS_OK;
}

// ---------------------------------------------------------------- host scratch of ms_ba_create: no allocation per window after warm-up
// Everything ms_ba_create builds on the host (index structures, batch lists, the streams: ~40 arrays per problem) lives in a per-thread bump arena that keeps its
// memory from call to call: a sliding-window BA per keyframe allocates nothing once the first windows have sized it.  (A batch that does not fit overflows into
// plain blocks, freed at the next call.)  g_ba_host_allocs counts every allocation this file makes on the host or the device -- arena growth, overflow blocks,
// handle objects, device blocks, events -- so a test can hold "none after warm-up" against it (ms_debug_host_allocs; tests/host_shim_smoke.cpp).
struct BaArena {
    static constexpr size_t kKeepMax = (size_t)64 << 20;            // what a thread keeps between calls: enough for a handful of windows per call, not for a 256-window batch
    char *base = nullptr;
    size_t cap = 0, used = 0, total = 0;
    std::vector<void *> overflow;
    void begin() {                                                  // start of a create: nothing of the previous call is alive any more
        const size_t want = std::min(kKeepMax, total + total / 4);
        if (want > cap) { std::free(base); cap = ms_align_up(want, (size_t)1 << 20); base = static_cast<char *>(std::malloc(cap)); ++g_ba_host_allocs; }
        used = 0; total = 0;
    }
    void end() {                                                    // end of a create: what did not fit goes back at once (the arena itself stays)
        for (void *p : overflow) std::free(p);
        overflow.clear();
    }
    void *take(size_t bytes, size_t align) {
        const size_t at = ms_align_up(used, align);
        total = ms_align_up(total, align) + bytes;
        if (base && at + bytes <= cap) { used = at + bytes; return base + at; }
        void *p = std::malloc(bytes ? bytes : 1);
        ++g_ba_host_allocs;
        overflow.push_back(p);
        return p;
    }
    ~BaArena() { end(); std::free(base); }
};
static thread_local BaArena tl_ba_arena;
template <class T>
struct BaAlloc {
    typedef T value_type;
    BaAlloc() = default;
    template <class U> BaAlloc(const BaAlloc<U> &) {}
    T *allocate(size_t n) { return static_cast<T *>(tl_ba_arena.take(n * sizeof(T), alignof(T) < 16 ? 16 : alignof(T))); }
    void deallocate(T *, size_t) {}
    template <class U> bool operator==(const BaAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const BaAlloc<U> &) const { return false; }
};
template <class T> using bvec = std::vector<T, BaAlloc<T>>;

extern "C" {

int ms_ba_create(ms_ctx *c, const ms_ba_problem *problems, int n, ms_ba **out) {
    MsRange range("ms_ba_create");
    if (!c || !problems || !out || n < 1) return MS_ERR_INVALID;
    *out = nullptr;
    MS_HIP(c, hipSetDevice(c->device));
    tl_ba_arena.begin();
    struct ArenaEnd { ~ArenaEnd() { tl_ba_arena.end(); } } arena_end;      // (after every container of this call is gone: declared first, destroyed last)
    const bool tm_on = std::getenv("MS_BA_TIMING") != nullptr;           // prints the host index build and the allocation + upload time of every create to stderr
    auto tm_now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tm0 = tm_now();
    double tm_part[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tm_mark = tm0;                  // CSR + envelope | record-based Schur lists | Cholesky panel lists | fused Schur batches | windowed Cholesky tables
    auto tm_lap = [&](int k) { if (tm_on) { const double t = tm_now(); tm_part[k] += t - tm_mark; tm_mark = t; } };
    // pass 1: sizes + host-side structure (free-pose index, CSR by point and by free pose)
    struct FsHost { bvec<int32_t> row0, row1, batch_start, b_obs_start, b_run_start, b_fmt, pobs, rowoff, yoff; bvec<uint16_t> pairs; bvec<double> puv; bool by_points = false; };
    struct Prep { bvec<int32_t> cw_slot, cw_act_start, cw_act, cw_load_start, cw_load; int cw_W = 0; bool cw_zglobal = false, cw_meta_lds = false; bvec<int32_t> pidx, free2pose, pt_start, pt_obs, fstart, fobs, fo_lo; bvec<double> fo_uvi; bvec<int32_t> chunk_items, seg_start, seg_pair, env16, act_start, act_blk, fs_cs; FsHost fs[2];
                  bool fused = false; int fs_only = 0; int np_free = 0, n_chunks = 0, n_seg = 0; double chol_tiles = 0;
                  bool one_pose = false; bvec<int32_t> op_pose, op_o; bvec<double> op_uvi; };
    bvec<Prep> prep(n);
    size_t total = 0;
    auto bump = [&](size_t bytes) { size_t o = total; total += ms_align_up(bytes ? bytes : 8, 256); return o; };
    struct Off { size_t pose, pose_bk, pose0, point, point_bk, point0, pidx, pfix, obs_pose, obs_point, obs_uv, obs_info, pt_start, pt_obs, fstart, fobs, fo_lo, fo_uvi,
                 free2pose, edge_i, edge_j, edge_meas, edge_info, Hpp, S, bp, dp, y, Hll, bl, Hinv, Hpl, dl, chi2, stats, chunk_items, seg_start, seg_pair, Y, zrow, bar, red, flag, dinv, env16, panG, act_start, act_blk, fs_cs, fs_row0[2], fs_row1[2], fs_batch[2], fs_bobs[2], fs_brun[2], fs_bfmt[2], fs_pobs[2], fs_puv[2], fs_pairs[2], fs_rowoff[2], fs_yoff[2], cw_slot, cw_act_start, cw_act, cw_load_start, cw_load, op_pose, op_o, op_uvi, op_rec, op_red, pack, Hpp2, bp2, Hll2, bl2, desc; };
    bvec<Off> off(n);
    bvec<size_t> in_lo(n), in_hi(n);
    for (int p = 0; p < n; ++p) {
        const ms_ba_problem &Q = problems[p];
        if (Q.n_pose < 1 || Q.n_point < 0 || Q.n_obs < 0 || Q.n_pose_edge < 0 || !Q.pose || !Q.pose_fixed || (Q.n_point && !Q.point) ||
            (Q.n_obs && (!Q.obs_pose || !Q.obs_point || !Q.obs_uv || !Q.obs_info)) || (Q.n_pose_edge && (!Q.edge_i || !Q.edge_j || !Q.edge_meas || !Q.edge_info)))
            return ms_fail(c, MS_ERR_INVALID, "ms_ba_create: problem %d has missing arrays", p);
        Prep &R = prep[p];
        R.pidx.assign(Q.n_pose, -1);
        for (int i = 0; i < Q.n_pose; ++i) if (!Q.pose_fixed[i]) { R.pidx[i] = R.np_free++; R.free2pose.push_back(i); }
        if (R.np_free > kMaxFreePosesTeam) return ms_fail(c, MS_ERR_CAPACITY, "ms_ba_create: %d free poses (max %d in this version)", R.np_free, kMaxFreePosesTeam);
        for (int o = 0; o < Q.n_obs; ++o)
            if (Q.obs_pose[o] < 0 || Q.obs_pose[o] >= Q.n_pose || Q.obs_point[o] < 0 || Q.obs_point[o] >= Q.n_point)
                return ms_fail(c, MS_ERR_INVALID, "ms_ba_create: observation %d of problem %d indexes outside the problem", o, p);
        for (int k = 0; k < Q.n_pose_edge; ++k)
            if (Q.edge_i[k] < 0 || Q.edge_i[k] >= Q.n_pose || Q.edge_j[k] < 0 || Q.edge_j[k] >= Q.n_pose)
                return ms_fail(c, MS_ERR_INVALID, "ms_ba_create: pose edge %d of problem %d indexes outside the problem", k, p);
        R.pt_start.assign(Q.n_point + 1, 0);
        for (int o = 0; o < Q.n_obs; ++o) R.pt_start[Q.obs_point[o] + 1]++;
        for (int l = 0; l < Q.n_point; ++l) R.pt_start[l + 1] += R.pt_start[l];
        R.pt_obs.resize(Q.n_obs);
        { bvec<int32_t> cur(R.pt_start.begin(), R.pt_start.end() - 1); for (int o = 0; o < Q.n_obs; ++o) R.pt_obs[cur[Q.obs_point[o]]++] = o; }
        R.fstart.assign(R.np_free + 1, 0);
        for (int o = 0; o < Q.n_obs; ++o) { const int f = R.pidx[Q.obs_pose[o]]; if (f >= 0) R.fstart[f + 1]++; }
        for (int f = 0; f < R.np_free; ++f) R.fstart[f + 1] += R.fstart[f];
        R.fobs.resize(R.fstart[R.np_free]);
        { bvec<int32_t> cur(R.fstart.begin(), R.fstart.end() - 1); for (int o = 0; o < Q.n_obs; ++o) { const int f = R.pidx[Q.obs_pose[o]]; if (f >= 0) R.fobs[cur[f]++] = o; } }
        for (int o = 0; o < Q.n_obs; ++o) if (R.pidx[Q.obs_pose[o]] < 0) R.fobs.push_back(o);      // behind them: the observations of FIXED poses (fobs[fstart[np_free] .. n_obs))
        // the same order as a stream of values (linearise_stream, eval_stream, backsub_stream): read by launches with ONE workgroup per window -- the same rule
        // ms_ba_solve applies; a handle that will get teams (a window per keyframe) neither builds nor uploads them
        const bool want_streams = std::min(std::max(1, std::min(kMaxTeam, c->n_cu / std::max(n, 1))), std::min(32, std::max(1, Q.n_obs / 512))) == 1;
        if (want_streams) { R.fo_lo.resize(4 * (size_t)Q.n_obs); R.fo_uvi.assign(4 * (size_t)Q.n_obs, 0.0); }
        for (int ii = 0; want_streams && ii < Q.n_obs; ++ii) {
            const int o = R.fobs[ii];
            R.fo_lo[4 * (size_t)ii] = Q.obs_point[o]; R.fo_lo[4 * (size_t)ii + 1] = o | ((Q.point_fixed && Q.point_fixed[Q.obs_point[o]]) ? (int32_t)0x80000000 : 0); R.fo_lo[4 * (size_t)ii + 2] = Q.obs_pose[o]; R.fo_lo[4 * (size_t)ii + 3] = R.pidx[Q.obs_pose[o]];
            R.fo_uvi[4 * (size_t)ii] = Q.obs_uv[2 * (size_t)o]; R.fo_uvi[4 * (size_t)ii + 1] = Q.obs_uv[2 * (size_t)o + 1]; R.fo_uvi[4 * (size_t)ii + 2] = Q.obs_info[o];
        }
        {   // one free pose + at least one free point (stage 1 of localBundleAdjust): k_ba_one_pose reads the observations in point order, indices and values side by side
            bool any_free_point = false;
            for (int l = 0; l < Q.n_point && !any_free_point; ++l) any_free_point = !(Q.point_fixed && Q.point_fixed[l]);
            int touching = 0;
            if (R.np_free == 1) for (int k = 0; k < Q.n_pose_edge; ++k) touching += Q.edge_i[k] == R.free2pose[0] || Q.edge_j[k] == R.free2pose[0];
            R.one_pose = R.np_free == 1 && any_free_point && Q.n_pose_edge <= OP_NT && touching <= PO_MAXE;
            if (R.one_pose) {
                R.op_pose.resize(Q.n_obs); R.op_o.resize(Q.n_obs); R.op_uvi.resize(3 * (size_t)Q.n_obs);
                for (int ii = 0; ii < Q.n_obs; ++ii) {
                    const int o = R.pt_obs[ii];
                    R.op_pose[ii] = Q.obs_pose[o]; R.op_o[ii] = o;
                    R.op_uvi[3 * (size_t)ii] = Q.obs_uv[2 * (size_t)o]; R.op_uvi[3 * (size_t)ii + 1] = Q.obs_uv[2 * (size_t)o + 1]; R.op_uvi[3 * (size_t)ii + 2] = Q.obs_info[o];
                }
            }
        }
        // envelope of the reduced camera matrix at pose level: the first free pose each free pose is coupled with (a shared point or a
        // pose-pose edge), and the free observations of every free point, sorted by free pose (flat arrays: the fused Schur pass is built from them)
        bvec<int> first(R.np_free), hfirst(R.np_free);          // hfirst: the first free pose a pose is coupled with by a pose-pose EDGE (Hpp has nothing left of that block)
        bvec<int32_t> fp_start(Q.n_point + 1, 0), fp_f, fp_o;
        {
            for (int f = 0; f < R.np_free; ++f) first[f] = hfirst[f] = f;
            int max_k = 0;
            for (int l = 0; l < Q.n_point; ++l) {
                fp_start[l] = (int32_t)fp_f.size();
                if (Q.point_fixed && Q.point_fixed[l]) continue;
                const size_t b0 = fp_f.size();
                for (int ii = R.pt_start[l]; ii < R.pt_start[l + 1]; ++ii) {
                    const int o = R.pt_obs[ii], f = R.pidx[Q.obs_pose[o]];
                    if (f < 0) continue;
                    size_t at = fp_f.size();
                    fp_f.push_back(f); fp_o.push_back(o);
                    while (at > b0 && fp_f[at - 1] > f) { std::swap(fp_f[at - 1], fp_f[at]); std::swap(fp_o[at - 1], fp_o[at]); --at; }     // insertion: lists are short and mostly sorted
                }
                const int kk = (int)(fp_f.size() - b0);
                max_k = std::max(max_k, kk);
                if (kk) { const int fmin = fp_f[b0]; for (size_t a = b0; a < fp_f.size(); ++a) first[fp_f[a]] = std::min(first[fp_f[a]], fmin); }
            }
            fp_start[Q.n_point] = (int32_t)fp_f.size();
            for (int k = 0; k < Q.n_pose_edge; ++k) {
                const int fi = R.pidx[Q.edge_i[k]], fj = R.pidx[Q.edge_j[k]];
                if (fi >= 0 && fj >= 0) { first[std::max(fi, fj)] = std::min(first[std::max(fi, fj)], std::min(fi, fj)); hfirst[std::max(fi, fj)] = std::min(hfirst[std::max(fi, fj)], std::min(fi, fj)); }
            }
            bool ok = R.np_free > 0 && max_k <= FS_OB;
            for (int f = 0; f < R.np_free && ok; ++f) if (36 * (f - first[f] + 1) + 6 > kFsTileDoubles) ok = false;
            R.fused = ok;
        }
        tm_lap(0);
        if (!R.fused) {   // record-based Schur work list: for every free point, every ordered pair (a, b) of its observations with free poses fb <= fa,
            // counting-sorted by (fa, fb), then cut into chunks of CH items of one pose pair
            const int np = R.np_free;
            bvec<int32_t> count((size_t)np * np + 1, 0);
            auto for_items = [&](auto &&fn) {
                for (int l = 0; l < Q.n_point; ++l) {
                    if (Q.point_fixed && Q.point_fixed[l]) continue;
                    for (int ia = R.pt_start[l]; ia < R.pt_start[l + 1]; ++ia) {
                        const int a = R.pt_obs[ia], fa = R.pidx[Q.obs_pose[a]];
                        if (fa < 0) continue;
                        for (int ib = R.pt_start[l]; ib < R.pt_start[l + 1]; ++ib) {
                            const int b = R.pt_obs[ib], fb = R.pidx[Q.obs_pose[b]];
                            if (fb < 0 || fb > fa) continue;
                            fn(fa * np + fb, a, b);
                        }
                    }
                }
            };
            for_items([&](int key, int, int) { count[key + 1]++; });
            bvec<int32_t> kstart(count);
            for (size_t k = 1; k < kstart.size(); ++k) kstart[k] += kstart[k - 1];
            bvec<int32_t> sorted(2 * (size_t)kstart.back()), cur(kstart.begin(), kstart.end() - 1);
            for_items([&](int key, int a, int b) { const int pos = cur[key]++; sorted[2 * (size_t)pos] = a; sorted[2 * (size_t)pos + 1] = b; });
            R.seg_start.push_back(0);
            for (int key = 0; key < np * np; ++key) {
                const int lo = kstart[key], hi = kstart[key + 1];
                if (hi == lo) continue;
                for (int i = lo; i < hi; i += CH) {
                    for (int t = 0; t < CH; ++t) {
                        const bool in = i + t < hi;
                        R.chunk_items.push_back(in ? sorted[2 * (size_t)(i + t)] : -1);
                        R.chunk_items.push_back(in ? sorted[2 * (size_t)(i + t) + 1] : -1);
                    }
                    ++R.n_chunks;
                }
                R.seg_pair.push_back(((key / np) << 16) | (key % np));
                R.seg_start.push_back(R.n_chunks);
                ++R.n_seg;
            }
        }
        {   // envelope of the reduced camera matrix: the first free pose each free pose is coupled with (a shared point or a
            // pose-pose edge).  Cholesky creates no fill left of it, so the factorisation skips everything outside.
            const int np = R.np_free, n6i = 6 * np;
            R.env16.assign(n6i / 16 + 2, 0);
            for (int b = 0; b < (int)R.env16.size(); ++b) {
                int e = n6i;
                for (int r = 16 * b; r < 16 * b + 16; ++r) e = r < n6i ? std::min(e, 6 * first[r / 6]) : 0;     // the rhs row (r = n6) is dense
                R.env16[b] = std::min(e, n6i);
            }
            // row tiles a Cholesky panel touches on average (those whose envelope reaches the panel): sizes the factorisation's sub-team
            const int nblk = n6i / 16 + 1;
            long long act = 0;
            for (int pb = 0; pb * 16 < n6i; ++pb) for (int b = pb; b <= nblk; ++b) act += R.env16[(size_t)std::min(b, (int)R.env16.size() - 1)] <= pb * 16 + 15;
            R.chol_tiles = n6i ? (double)act / ((n6i + 15) / 16) : 0.0;
            if (np > kMaxFreePoses) {                   // the distributed factorisation walks these lists
                R.act_start.push_back(0);
                for (int pb = 0; pb * 16 < n6i; ++pb) {
                    const int m_rows = n6i - pb * 16 + 1;
                    for (int rt = 0; rt * 16 < m_rows; ++rt) if (R.env16[(size_t)std::min(pb + rt, (int)R.env16.size() - 1)] <= pb * 16 + 15) R.act_blk.push_back(rt);
                    R.act_start.push_back((int32_t)R.act_blk.size());
                }
            }
        }
        tm_lap(2);
        {   // fused Schur pass (schur_fused): pose rows -> passes whose envelope part fits the LDS tile, points -> batches of <= 64 observations
            const int np = R.np_free;
            R.fs_cs.resize(2 * (size_t)np);                              // [np] first column of the row's envelope part, then [np] first column of the row's part of Hpp
            for (int f = 0; f < np; ++f) { R.fs_cs[f] = 6 * first[f]; R.fs_cs[(size_t)np + f] = 6 * hfirst[f]; }
            const bool ok = R.fused;
            bvec<int32_t> stamp(Q.n_point, -1);
            bvec<std::pair<uint64_t, int32_t>> pts;               // (signature of the point's pose set, point)
            bvec<std::pair<int32_t, uint16_t>> bp2;                // (block key, pair) of the open batch
            // Only the set the launches will use is built (the other one aliases it: still correct, only slower, should ms_ba_set_team
            // ask for the other regime later): a batch that fills the chip always runs one workgroup per problem, a handful of windows
            // gets teams -- the same rule ms_ba_solve applies.
            const int max_team = std::max(1, std::min(kMaxTeam, c->n_cu / std::max(n, 1)));
            const int pred_team = std::min(max_team, std::min(32, std::max(1, Q.n_obs / 512)));      // what ms_ba_solve picks on its own
            R.fs_only = pred_team == 1 ? 0 : 1;
            for (int set = 0; set < 2 && ok; ++set) {
                if (set != R.fs_only) continue;
                FsHost &F = R.fs[set];
                // coarse set (set 0): as many rows per pass as the tile takes
                const int want_passes = set == 0 ? 1 : std::max(1, std::min(pred_team, np));
                auto add_pass = [&](int r, int r1) {                      // rows [r, r1): their envelope parts side by side in the tile, then the rhs segment
                    F.row0.push_back(r); F.row1.push_back(r1);
                    F.yoff.push_back(0); F.yoff.push_back((int32_t)F.rowoff.size());
                    int off2 = 0;
                    for (int f = r; f < r1; ++f) { F.rowoff.push_back(off2); off2 += 36 * (f - first[f] + 1); }
                    F.yoff[F.yoff.size() - 2] = off2;
                };
                bvec<int32_t> need_pre(np + 1, 0);                        // tile doubles of rows [0, f): a pass's need is a difference
                for (int f = 0; f < np; ++f) need_pre[f + 1] = need_pre[f] + 36 * (f - first[f] + 1) + 6;
                auto tile_need = [&](int r, int r1) { return r1 > r ? need_pre[r1] - need_pre[r] : 0; };
                // set 1 (teams): the POINTS are dealt out, not the rows -- a pass is a run of points (in the order of their first pose) with about 1 / team of the
                // block products; its rows are the poses those points see (they overlap with the neighbours': the sums meet in S through atomics).  Every
                // observation is then linearised once per damped solve.  (Row passes made each of the 32 workgroups re-evaluate every point that touches its
                // one or two rows: 8x the observations, 75 batches per workgroup instead of 10.)
                bvec<bvec<int32_t>> group_pts;
                F.by_points = false;
                if (set == 1) {
                    bvec<std::pair<uint32_t, int32_t>> order;      // (first pose << 16 | last pose, point)
                    double total_cost = 0;
                    for (int l = 0; l < Q.n_point; ++l) {
                        const int k = fp_start[l + 1] - fp_start[l];
                        if (k == 0) continue;
                        order.emplace_back(((uint32_t)fp_f[fp_start[l]] << 16) | (uint32_t)fp_f[fp_start[l + 1] - 1], l);
                        total_cost += 8.0 * k + 0.5 * k * (k + 1);
                    }
                    {   // by (first pose, last pose), then point: a counting sort over the np x np key space (std::sort of 2000 keys was 0.05 ms of the 0.45 ms create)
                        const size_t nk = (size_t)np * np;
                        if (order.size() > 64 && nk <= 65536) {
                            bvec<int32_t> cnt(nk + 1, 0);
                            auto key_of = [&](const std::pair<uint32_t, int32_t> &e) { return (size_t)(e.first >> 16) * np + (e.first & 0xFFFF); };
                            for (const auto &e : order) cnt[key_of(e) + 1]++;
                            for (size_t q = 0; q < nk; ++q) cnt[q + 1] += cnt[q];
                            bvec<std::pair<uint32_t, int32_t>> sorted(order.size());
                            for (const auto &e : order) sorted[cnt[key_of(e)]++] = e;      // (order is in point order: equal keys stay in it)
                            order.swap(sorted);
                        } else std::sort(order.begin(), order.end());
                    }
                    double acc_cost = 0;
                    int g_lo = np, g_hi = 0;
                    group_pts.emplace_back();
                    for (const auto &e : order) {
                        const int l = e.second, k = fp_start[l + 1] - fp_start[l], lo = std::min(g_lo, (int)(e.first >> 16)), hi = std::max(g_hi, (int)(e.first & 0xFFFF) + 1);
                        const bool full = !group_pts.back().empty() && (tile_need(lo, hi) > kFsTileDoubles ||
                                          ((int)group_pts.size() < want_passes && acc_cost >= total_cost * (double)group_pts.size() / want_passes));
                        if (full) { add_pass(g_lo, g_hi); group_pts.emplace_back(); g_lo = (int)(e.first >> 16); g_hi = (int)(e.first & 0xFFFF) + 1; }
                        else { g_lo = lo; g_hi = hi; }
                        group_pts.back().push_back(l);
                        acc_cost += 8.0 * k + 0.5 * k * (k + 1);
                    }
                    if (!group_pts.back().empty()) add_pass(g_lo, g_hi); else group_pts.pop_back();
                    F.by_points = true;
                    // a point seen from poses so far apart that the rows between them do not fit the tile (scattered covisibility): row passes for this window
                    for (size_t ps = 0; ps < F.row0.size(); ++ps) if (tile_need(F.row0[ps], F.row1[ps]) > kFsTileDoubles) F.by_points = false;
                    if (!F.by_points) { F.row0.clear(); F.row1.clear(); F.yoff.clear(); F.rowoff.clear(); group_pts.clear(); }
                }
                tm_lap(5);
                if (!F.by_points) {
                    int r = 0;
                    while (r < np) {                                      // greedy row ranges under the tile budget
                        int r1 = r, used = 0;
                        while (r1 < np) {
                            const int need = 36 * (r1 - first[r1] + 1) + 6;
                            if (used + need > kFsTileDoubles) break;
                            used += need; ++r1;
                        }
                        add_pass(r, r1);
                        r = r1;
                    }
                }
                F.batch_start.push_back(0); F.b_obs_start.push_back(0); F.b_run_start.push_back(0);
                F.pobs.reserve(4 * fp_f.size() * 2); F.pairs.reserve(8 * fp_f.size());
                std::fill(stamp.begin(), stamp.end(), -1);
                for (size_t ps = 0; ps < F.row0.size(); ++ps) {
                    const int r0 = F.row0[ps], r1 = F.row1[ps];
                    pts.clear();
                    if (F.by_points) {
                        for (int l : group_pts[ps]) {
                            uint64_t h = 1469598103934665603ull; int fmin = 0x7fff, fmax = 0, kk = 0;
                            for (int jj = fp_start[l]; jj < fp_start[l + 1]; ++jj) { h = (h ^ (uint64_t)fp_f[jj]) * 1099511628211ull; fmin = std::min(fmin, (int)fp_f[jj]); fmax = fp_f[jj]; ++kk; }
                            pts.emplace_back(((uint64_t)fmin << 48) | ((uint64_t)fmax << 32) | ((uint64_t)(kk & 0xFF) << 24) | (h & 0xFFFFFFull), l);
                        }
                    } else
                    for (int fa = r0; fa < r1; ++fa)
                        for (int ii = R.fstart[fa]; ii < R.fstart[fa + 1]; ++ii) {
                            const int l = Q.obs_point[R.fobs[ii]];
                            if (stamp[l] == (int32_t)ps || (Q.point_fixed && Q.point_fixed[l])) continue;
                            stamp[l] = (int32_t)ps;
                            // signature of the point's free poses below r1 (later poses have no pair with a row of this pass): points with the same
                            // set get the same key, and keys order roughly by position in the window
                            uint64_t h = 1469598103934665603ull; int fmin = 0x7fff, fmax = 0, kk = 0;
                            for (int jj = fp_start[l]; jj < fp_start[l + 1] && fp_f[jj] < r1; ++jj) { h = (h ^ (uint64_t)fp_f[jj]) * 1099511628211ull; fmin = std::min(fmin, (int)fp_f[jj]); fmax = fp_f[jj]; ++kk; }
                            pts.emplace_back(((uint64_t)fmin << 48) | ((uint64_t)fmax << 32) | ((uint64_t)(kk & 0xFF) << 24) | (h & 0xFFFFFFull), l);
                        }
                    tm_lap(6);
                    std::sort(pts.begin(), pts.end());                    // points with the same set of poses next to each other: their pairs fall into the same blocks
                    tm_lap(7);
                    int in_batch = 0;
                    bool uniform = true;                                  // every point of the open batch has the same pose set
                    uint64_t batch_key = 0;
                    int batch_k = 0, batch_first = -1;                    // poses per point / first point (index into pts) of the open batch
                    auto emit_chunks = [&](size_t n_pairs, auto &&pair_at, auto &&key_at) {
                        // chunks of <= 8 pairs of one block; when the batch has few pairs the chunks get shorter so that more lanes share them
                        const size_t cap = std::min<size_t>(8, std::max<size_t>(1, (n_pairs + 63) / 64));
                        size_t in_chunk = 0;
                        for (size_t i = 0; i < n_pairs; ++i) {
                            if (i && (key_at(i) != key_at(i - 1) || in_chunk == cap)) { while (F.pairs.size() % 8) F.pairs.push_back((uint16_t)0xFFFF); in_chunk = 0; }
                            F.pairs.push_back(pair_at(i)); ++in_chunk;
                        }
                        while (F.pairs.size() % 8) F.pairs.push_back((uint16_t)0xFFFF);
                    };
                    auto pad_slots = [&]() {                              // a batch owns FS_OB lane slots (the kernel addresses batch b at slot FS_OB b): the rest hold "no observation"
                        const size_t at = F.pobs.size(), slots = at / 4, padded = (slots + FS_OB - 1) / FS_OB * FS_OB;
                        if (padded == slots) return;
                        F.pobs.resize(4 * padded);
                        for (size_t q = slots; q < padded; ++q) { F.pobs[4 * q] = -1; F.pobs[4 * q + 1] = 0; F.pobs[4 * q + 2] = 0; F.pobs[4 * q + 3] = -1; }
                    };
                    auto close_batch = [&](int pt_end) {
                        if (in_batch == 0) return;
                        int single_fmt = 0;
                        if (uniform) {
                            // the usual case: G points with the same k poses -> pairs in (a, b) position order are already grouped by block, blocks ascending
                            const int G = pt_end - batch_first, k = batch_k, j0 = fp_start[pts[batch_first].second];
                            int a0 = 0;
                            while (a0 < k && fp_f[j0 + a0] < r0) ++a0;
                            // (only for pass sets that own points = team launches: one window per keyframe, where the index build is a fifth of the call; a 256-window launch keeps the
                            //  lists -- enumerating costs its Schur pass 4 %, 11.4 against 10.9 ms, and its handles are built once.  schur_fused<true> runs exactly these sets)
                            if (F.by_points && k <= 31 && G <= 127 && kProceduralPairs) {      // the kernel enumerates the pairs of such a batch itself: nothing to build, nothing to upload
                                F.b_fmt.push_back(-1 - (G | (k << 7) | (a0 << 12)));
                                pad_slots();
                                F.b_obs_start.push_back((int32_t)(F.pobs.size() / 4)); F.b_run_start.push_back((int32_t)F.pairs.size());
                                in_batch = 0;
                                return;
                            }
                            const size_t n_pairs = (size_t)G * ((size_t)k * (k + 1) / 2 - (size_t)a0 * (a0 + 1) / 2);
                            const size_t cap = std::min<size_t>(8, std::max<size_t>(1, (n_pairs + 63) / 64));
                            single_fmt = n_pairs <= 64 ? (int)n_pairs : 0;
                            for (int a = a0; a < k; ++a)
                                for (int b2 = 0; b2 <= a; ++b2) {
                                    if (single_fmt) { for (int gp = 0; gp < G; ++gp) F.pairs.push_back((uint16_t)((gp * k + a) | ((gp * k + b2) << 8))); continue; }
                                    size_t in_chunk = 0;
                                    for (int gp = 0; gp < G; ++gp) {
                                        if (in_chunk == cap) { while (F.pairs.size() % 8) F.pairs.push_back((uint16_t)0xFFFF); in_chunk = 0; }
                                        F.pairs.push_back((uint16_t)((gp * k + a) | ((gp * k + b2) << 8))); ++in_chunk;
                                    }
                                    while (F.pairs.size() % 8) F.pairs.push_back((uint16_t)0xFFFF);
                                }
                        } else {
                            bp2.clear();
                            int base = 0;
                            for (int q = batch_first; q < pt_end; ++q) {
                                const int l = pts[q].second, j0 = fp_start[l];
                                int kk = 0;
                                while (j0 + kk < fp_start[l + 1] && fp_f[j0 + kk] < r1) ++kk;
                                for (int a = 0; a < kk; ++a) {
                                    const int fa = fp_f[j0 + a];
                                    if (fa < r0) continue;
                                    for (int b2 = 0; b2 < kk; ++b2)
                                        if (fp_f[j0 + b2] <= fa) bp2.emplace_back((fa << 16) | fp_f[j0 + b2], (uint16_t)((base + a) | ((base + b2) << 8)));
                                }
                                base += kk;
                            }
                            std::stable_sort(bp2.begin(), bp2.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
                            single_fmt = bp2.size() <= 64 ? (int)bp2.size() : 0;
                            if (single_fmt) for (const auto &e : bp2) F.pairs.push_back(e.second);
                            else emit_chunks(bp2.size(), [&](size_t i) { return bp2[i].second; }, [&](size_t i) { return bp2[i].first; });
                        }
                        while (F.pairs.size() % 8) F.pairs.push_back((uint16_t)0xFFFF);
                        F.b_fmt.push_back(single_fmt);
                        pad_slots();
                        F.b_obs_start.push_back((int32_t)(F.pobs.size() / 4)); F.b_run_start.push_back((int32_t)F.pairs.size());
                        in_batch = 0;
                    };
                    for (int q = 0; q < (int)pts.size(); ++q) {
                        const int l = pts[q].second, j0 = fp_start[l];
                        int kk = 0;
                        while (j0 + kk < fp_start[l + 1] && fp_f[j0 + kk] < r1) ++kk;
                        if (in_batch + kk > FS_OB) close_batch(q);
                        if (in_batch == 0) { uniform = true; batch_key = pts[q].first; batch_k = kk; batch_first = q; }
                        else if (pts[q].first != batch_key || kk != batch_k) uniform = false;
                        if (uniform && q > batch_first) {                 // equal signatures: make sure the sets really are equal (the key holds a 24-bit hash)
                            const int jb = fp_start[pts[batch_first].second];
                            for (int a = 0; a < kk; ++a) if (fp_f[j0 + a] != fp_f[jb + a]) { uniform = false; break; }
                        }
                        {
                            const size_t at = F.pobs.size();
                            F.pobs.resize(at + 4 * (size_t)kk);
                            int32_t *w = F.pobs.data() + at;
                            for (int a = 0; a < kk; ++a, w += 4) { const int o = fp_o[j0 + a]; w[0] = o; w[1] = Q.obs_pose[o]; w[2] = l; w[3] = fp_f[j0 + a]; }
                        }
                        in_batch += kk;
                    }
                    close_batch((int)pts.size());
                    F.batch_start.push_back((int32_t)F.b_obs_start.size() - 1);
                    tm_lap(8);
                }
                F.b_obs_start.push_back(F.b_obs_start.back());
                F.puv.assign(F.pobs.size(), 0.0);                         // per lane slot: u, v, information, 0
                for (size_t i = 0; i < F.pobs.size() / 4; ++i) {
                    const int o = F.pobs[4 * i];
                    if (o >= 0) { F.puv[4 * i] = Q.obs_uv[2 * (size_t)o]; F.puv[4 * i + 1] = Q.obs_uv[2 * (size_t)o + 1]; F.puv[4 * i + 2] = Q.obs_info[o]; }
                }
                tm_lap(9);
            }
        }
        if (tm_on) { for (int k = 5; k <= 9; ++k) tm_part[3] += tm_part[k]; }
        tm_lap(3);
        {   // windowed Cholesky (cholesky_window): active 16-row blocks per panel, LDS slots, tiles entering per panel
            const int np = R.np_free, n6i = 6 * np, nblk = (n6i + 15) / 16;
            bvec<int> ent(nblk, 0);
            for (int b = 0; b < nblk; ++b) {
                int e = n6i;
                for (int r = 16 * b; r < std::min(16 * b + 16, n6i); ++r) e = std::min(e, 6 * first[r / 6]);
                ent[b] = std::min(e / 16, b);
            }
            for (int b = nblk - 2; b >= 0; --b) ent[b] = std::min(ent[b], b);      // (a block is active at its own panel at the latest)
            R.cw_slot.assign(nblk, 0);
            bvec<int> free_slots, active;
            int W = 0;
            bvec<bvec<int>> entering(nblk);
            for (int b = 0; b < nblk; ++b) entering[ent[b]].push_back(b);
            R.cw_act_start.push_back(0); R.cw_load_start.push_back(0);
            for (int pnl = 0; pnl < nblk; ++pnl) {
                if (pnl > 0) active.erase(std::find(active.begin(), active.end(), pnl - 1));      // block pnl-1 is factored ...
                if (pnl > 1) free_slots.push_back(R.cw_slot[pnl - 2]);       // ... its slot is reused one panel later: panel pnl's tiles are fetched while pnl-1 is still updating
                std::sort(free_slots.begin(), free_slots.end(), std::greater<int>());
                for (int b : entering[pnl]) {
                    int sl;
                    if (!free_slots.empty()) { sl = free_slots.back(); free_slots.pop_back(); } else sl = W++;
                    R.cw_slot[b] = sl;
                    active.push_back(b);
                }
                std::sort(active.begin(), active.end());
                for (int bi : active)
                    for (int bj : active) {
                        if (bj > bi) break;
                        if (bi == bj && bi == pnl) continue;                       // the panel's own diagonal tile is fetched by the factoring wave
                        if (ent[bi] == pnl || ent[bj] == pnl) { R.cw_load.push_back(bi | (R.cw_slot[bi] << 16)); R.cw_load.push_back(bj | (R.cw_slot[bj] << 16)); }
                    }
                R.cw_load_start.push_back((int32_t)R.cw_load.size() / 2);
                for (int b : active) if (b != pnl) R.cw_act.push_back(b | (R.cw_slot[b] << 16));
                R.cw_act_start.push_back((int32_t)R.cw_act.size());
            }
            if (W < 3) W = 3;                                                // the back substitution keeps three columns (<= W tiles each) in the W x W tile area
            const size_t fixed = ((size_t)W * W * CT + 16 + 16 * (size_t)W + 32 + 256 + 16) * sizeof(double), zbytes = (size_t)((n6i + 15) & ~15) * sizeof(double);
            const bool fits = R.fused && nblk < 65536 && W >= 1 && W < 256;
            R.cw_W = fits && fixed <= kLdsBytes ? W : 0;
            R.cw_zglobal = fixed + zbytes > kLdsBytes;                       // a long trajectory: the tiles fit, the 8 n bytes of the rhs do not -- it stays in global memory
            const size_t meta = 4 * (R.cw_slot.size() + R.cw_act_start.size() + R.cw_load_start.size() + R.cw_act.size() + R.cw_load.size());
            R.cw_meta_lds = R.cw_W > 0 && fixed + 2 * zbytes + meta <= kLdsBytes;      // (and the reciprocal pivots: another n doubles)
        }
        const size_t n6 = 6 * (size_t)R.np_free, D = sizeof(double);
        Off &O = off[p];
        // inputs first, contiguous: they go up in ONE host->device copy per problem
        in_lo[p] = total;
        O.pose0 = bump(7 * Q.n_pose * D); O.point0 = bump(3 * Q.n_point * D);
        O.pidx = bump(4 * Q.n_pose); O.pfix = bump(Q.n_point);
        O.obs_pose = bump(4 * Q.n_obs); O.obs_point = bump(4 * Q.n_obs); O.obs_uv = bump(2 * Q.n_obs * D); O.obs_info = bump(Q.n_obs * D);
        O.pt_start = bump(4 * (Q.n_point + 1)); O.pt_obs = bump(4 * Q.n_obs); O.fstart = bump(4 * (R.np_free + 1)); O.fobs = bump(4 * R.fobs.size()); O.fo_lo = bump(4 * R.fo_lo.size()); O.fo_uvi = bump(8 * R.fo_uvi.size());
        O.free2pose = bump(4 * R.np_free); O.edge_i = bump(4 * Q.n_pose_edge); O.edge_j = bump(4 * Q.n_pose_edge);
        O.edge_meas = bump(7 * Q.n_pose_edge * D); O.edge_info = bump(36 * Q.n_pose_edge * D);
        O.chunk_items = bump(4 * R.chunk_items.size()); O.seg_start = bump(4 * R.seg_start.size()); O.seg_pair = bump(4 * R.seg_pair.size());
        O.env16 = bump(4 * R.env16.size());
        O.act_start = bump(4 * R.act_start.size()); O.act_blk = bump(4 * R.act_blk.size());
        O.fs_cs = bump(4 * R.fs_cs.size());
        O.cw_slot = bump(4 * R.cw_slot.size()); O.cw_act_start = bump(4 * R.cw_act_start.size()); O.cw_act = bump(4 * R.cw_act.size());
        O.cw_load_start = bump(4 * R.cw_load_start.size()); O.cw_load = bump(4 * R.cw_load.size());
        for (int set = 0; set < 2; ++set) {
            const FsHost &F = R.fs[set];
            O.fs_row0[set] = bump(4 * F.row0.size()); O.fs_row1[set] = bump(4 * F.row1.size()); O.fs_batch[set] = bump(4 * F.batch_start.size());
            O.fs_bobs[set] = bump(4 * F.b_obs_start.size()); O.fs_brun[set] = bump(4 * F.b_run_start.size()); O.fs_bfmt[set] = bump(4 * F.b_fmt.size()); O.fs_pobs[set] = bump(4 * F.pobs.size()); O.fs_puv[set] = bump(8 * F.puv.size());
            O.fs_pairs[set] = bump(2 * F.pairs.size() + 16); O.fs_rowoff[set] = bump(4 * F.rowoff.size()); O.fs_yoff[set] = bump(4 * F.yoff.size());
        }
        O.op_pose = bump(4 * R.op_pose.size()); O.op_o = bump(4 * R.op_o.size()); O.op_uvi = bump(sizeof(double) * R.op_uvi.size());
        O.desc = n == 1 ? bump(2 * sizeof(BaProb)) : 0;             // a single problem's two descriptors travel with its inputs: ONE copy per create
        in_hi[p] = total;
        O.pose = bump(7 * Q.n_pose * D); O.pose_bk = bump(7 * Q.n_pose * D); O.point = bump(3 * Q.n_point * D); O.point_bk = bump(3 * Q.n_point * D);
        O.Hpp = bump(n6 * n6 * D); O.S = bump(n6 * n6 * D); O.bp = bump(n6 * D); O.dp = bump(n6 * D); O.y = bump(n6 * D);
        O.Hll = bump(6 * Q.n_point * D); O.bl = bump(3 * Q.n_point * D); O.Hinv = bump(6 * Q.n_point * D); O.Hpl = bump(R.fused ? 8 : 18 * ((size_t)Q.n_obs + 1) * D);
        O.dl = bump(3 * Q.n_point * D); O.chi2 = bump(Q.n_obs * D); O.stats = bump(16 * D);
        O.Y = bump(R.fused ? 8 : 18 * ((size_t)Q.n_obs + 1) * D); O.zrow = bump((n6 + 16) * D);
        O.dinv = bump((n6 + 16) * D);
        O.panG = R.np_free > kMaxFreePoses ? bump((n6 + 17) * NB * D) : 0;
        O.bar = bump(256); O.red = bump(4 * kMaxTeam * D); O.flag = bump(256);     // team state on lines of their own (bump aligns to 256 B)
        O.op_rec = bump(R.one_pose ? 28 * (size_t)Q.n_point * D : 8); O.op_red = bump(R.one_pose ? 2 * (size_t)kMaxTeam * OP_NV * D : 8);
        // the second set of the linearisation's outputs (k_ba_lm's fused trial schedule): only for handles whose launches run one workgroup per problem on the streams
        const bool alt_set = !R.fo_lo.empty() && R.fused;
        O.Hpp2 = bump(alt_set ? n6 * n6 * D : 8); O.bp2 = bump(alt_set ? n6 * D : 8); O.Hll2 = bump(alt_set ? 6 * Q.n_point * D : 8); O.bl2 = bump(alt_set ? 3 * Q.n_point * D : 8);
        // a single small problem gets its results packed behind every launch (ba_after_launch): status, poses, points, chi2 per observation
        const size_t pack_doubles = 16 + 7 * (size_t)Q.n_pose + 3 * (size_t)Q.n_point + (size_t)Q.n_obs;
        O.pack = bump(n == 1 && pack_doubles * D <= ba_eager_max() ? pack_doubles * D : 8);
    }
    const double tm1 = tm_now();
    ms_ba *B = nullptr;
    for (void *&slot : c->ba_handle_pool) if (slot) { B = static_cast<ms_ba *>(slot); slot = nullptr; break; }      // a destroyed handle's object: its vectors keep their capacity, its event stays
    if (!B) { B = new ms_ba(); ++g_ba_host_allocs; }
    B->ctx = c; B->n = n;
    {   // ONE device block per handle (arena + the problem descriptors behind it), taken from the context's cache of destroyed handles when one is large enough
        const size_t probs_at = ms_align_up(total, 256), need = probs_at + 2 * sizeof(BaProb) * n;
        // best fit among the kept blocks, but never one more than kBaCacheSlack times the request: a small window must not sit on the
        // gigabytes a global-BA handle left behind (that block waits for the next large request, or goes when the cache is trimmed)
        int best = -1;
        for (int i = 0; i < 4; ++i)
            if (c->ba_cache[i].p && c->ba_cache[i].bytes >= need && c->ba_cache[i].bytes / kBaCacheSlack <= need &&
                (best < 0 || c->ba_cache[i].bytes < c->ba_cache[best].bytes)) best = i;
        if (best >= 0) { B->d_arena = static_cast<char *>(c->ba_cache[best].p); B->arena_bytes = c->ba_cache[best].bytes; c->ba_cache[best] = {}; }
        else {
            hipError_t e = hipMalloc(reinterpret_cast<void **>(&B->d_arena), need);
            ++g_ba_host_allocs;
            if (e != hipSuccess) {                                  // out of memory with blocks kept for later: give them back and try once more
                (void)hipGetLastError();
                (void)hipStreamSynchronize(c->stream);
                for (auto &b : c->ba_cache) if (b.p) { (void)hipFree(b.p); b = {}; }
                e = hipMalloc(reinterpret_cast<void **>(&B->d_arena), need);
            }
            if (e != hipSuccess) { (void)hipGetLastError(); ba_delete_object(B); return ms_fail(c, MS_ERR_HIP, "ms_ba_create: cannot allocate %zu bytes: %s", need, hipGetErrorString(e)); }
            B->arena_bytes = need;
        }
        B->d_probs = reinterpret_cast<BaProb *>(B->d_arena + (n == 1 ? off[0].desc : probs_at));
        B->d_probs_alt = B->d_probs + n;
    }
    // small creates (a pose-only problem per frame, a window per keyframe) stage all their inputs in the context's own page-locked block and do not wait for the copies
    size_t stage_need = 2 * sizeof(BaProb) * n, stage_at = 0;
    for (int p = 0; p < n; ++p) stage_need += ms_align_up(in_hi[p] - in_lo[p], (size_t)256);
    const bool staged = stage_need <= kBaStageMax;
    if (staged) {
        if (c->ba_stage_busy) { (void)hipEventSynchronize(c->ba_stage_ev); c->ba_stage_busy = false; }      // (the previous create's upload: long done)
        hipError_t e = hipSuccess;
        if (!c->ba_stage_ev) { e = hipEventCreateWithFlags(&c->ba_stage_ev, hipEventDisableTiming); ++g_ba_host_allocs; }
        if (e == hipSuccess && c->ba_stage_bytes < stage_need) {
            if (c->ba_stage) (void)hipHostFree(c->ba_stage);
            c->ba_stage = nullptr; c->ba_stage_bytes = 0;
            const size_t want = std::min(kBaStageMax, ms_align_up(stage_need + stage_need / 4, (size_t)1 << 16));
            e = hipHostMalloc(&c->ba_stage, want, hipHostMallocDefault); ++g_ba_host_allocs;
            if (e == hipSuccess) c->ba_stage_bytes = want;
        }
        if (e != hipSuccess) { (void)hipGetLastError(); ms_ba_destroy(B); return ms_fail(c, MS_ERR_HIP, "ms_ba_create: staging block: %s", hipGetErrorString(e)); }
    }
    {   // the eager results of a single small problem: a page-locked block that stays with the handle object
        const size_t pack_doubles = n == 1 ? 16 + 7 * (size_t)problems[0].n_pose + 3 * (size_t)problems[0].n_point + (size_t)problems[0].n_obs : 0;
        B->pack_doubles = pack_doubles * sizeof(double) <= ba_eager_max() ? pack_doubles : 0;
        if (B->pack_doubles * sizeof(double) > B->h_result_bytes) {
            if (B->h_result) (void)hipHostFree(B->h_result);
            B->h_result = nullptr; B->h_result_bytes = 0;
            const size_t want = ms_align_up(B->pack_doubles * sizeof(double) * 2, (size_t)4096);
            const hipError_t e = hipHostMalloc(&B->h_result, want, hipHostMallocDefault); ++g_ba_host_allocs;
            if (e != hipSuccess) { (void)hipGetLastError(); ms_ba_destroy(B); return ms_fail(c, MS_ERR_HIP, "ms_ba_create: result block: %s", hipGetErrorString(e)); }
            B->h_result_bytes = want;
        }
    }
    {   // poseBundleAdjust-shaped handles (one free pose, no free point, <= PO_MAXE edges): k_ba_pose_only reads nothing it has not been given or written itself, so the
        // arena is not cleared now -- ms_ba_solve clears the work areas (everything behind the inputs) should the general kernel ever run on the handle
        bool all_po = true;
        for (int p = 0; p < n && all_po; ++p) {
            all_po = prep[p].np_free == 1 && problems[p].point_fixed != nullptr && ba_edges_at_free_pose(problems[p], prep[p].free2pose.empty() ? -1 : prep[p].free2pose[0]) <= PO_MAXE;
            for (int l = 0; l < problems[p].n_point && all_po; ++l) all_po = problems[p].point_fixed[l] != 0;
        }
        B->work_dirty = all_po;
        if (!all_po) {
            const hipError_t e = hipMemsetAsync(B->d_arena, 0, total, c->stream);
            if (e != hipSuccess) { ms_ba_destroy(B); return ms_fail(c, MS_ERR_HIP, "ms_ba_create: clearing the arena failed: %s", hipGetErrorString(e)); }
        }
        B->work_lo.assign(in_hi.begin(), in_hi.begin() + n); B->work_hi.resize(n);
        for (int p = 0; p < n; ++p) B->work_hi[p] = p + 1 < n ? in_lo[p + 1] : total;
    }
    B->host.resize(n); B->dims.assign(problems, problems + n);
    for (int p = 0; p < n; ++p) {
        const ms_ba_problem &Q = problems[p]; const Prep &R = prep[p]; const Off &O = off[p]; const size_t D = sizeof(double);
        // the problem's input block is assembled in the context's page-locked staging and goes up in ONE copy the copy engine reads directly
        const size_t stage_bytes = in_hi[p] - in_lo[p];
        char *stage = nullptr;
        if (staged) { stage = static_cast<char *>(c->ba_stage) + stage_at; stage_at += ms_align_up(stage_bytes, (size_t)256); }
        else {
            if (ms_pinned(c, stage_bytes) != MS_OK) { ms_ba_destroy(B); return MS_ERR_HIP; }
            stage = static_cast<char *>(c->pinned);
        }
        std::memset(stage, 0, stage_bytes);
        auto up = [&](size_t o, const void *src, size_t bytes) { if (bytes) std::memcpy(stage + (o - in_lo[p]), src, bytes); };
        up(O.pose0, Q.pose, 7 * Q.n_pose * D); up(O.point0, Q.point, 3 * Q.n_point * D);
        up(O.pidx, R.pidx.data(), 4 * Q.n_pose); if (Q.point_fixed) up(O.pfix, Q.point_fixed, Q.n_point);
        up(O.obs_pose, Q.obs_pose, 4 * Q.n_obs); up(O.obs_point, Q.obs_point, 4 * Q.n_obs); up(O.obs_uv, Q.obs_uv, 2 * Q.n_obs * D); up(O.obs_info, Q.obs_info, Q.n_obs * D);
        up(O.pt_start, R.pt_start.data(), 4 * (Q.n_point + 1)); up(O.pt_obs, R.pt_obs.data(), 4 * Q.n_obs);
        up(O.fstart, R.fstart.data(), 4 * (R.np_free + 1)); up(O.fobs, R.fobs.data(), 4 * R.fobs.size()); up(O.fo_lo, R.fo_lo.data(), 4 * R.fo_lo.size()); up(O.fo_uvi, R.fo_uvi.data(), 8 * R.fo_uvi.size()); up(O.free2pose, R.free2pose.data(), 4 * R.np_free);
        up(O.chunk_items, R.chunk_items.data(), 4 * R.chunk_items.size()); up(O.seg_start, R.seg_start.data(), 4 * R.seg_start.size()); up(O.seg_pair, R.seg_pair.data(), 4 * R.seg_pair.size());
        up(O.env16, R.env16.data(), 4 * R.env16.size());
        up(O.act_start, R.act_start.data(), 4 * R.act_start.size()); up(O.act_blk, R.act_blk.data(), 4 * R.act_blk.size());
        up(O.fs_cs, R.fs_cs.data(), 4 * R.fs_cs.size());
        up(O.cw_slot, R.cw_slot.data(), 4 * R.cw_slot.size()); up(O.cw_act_start, R.cw_act_start.data(), 4 * R.cw_act_start.size()); up(O.cw_act, R.cw_act.data(), 4 * R.cw_act.size());
        up(O.cw_load_start, R.cw_load_start.data(), 4 * R.cw_load_start.size()); up(O.cw_load, R.cw_load.data(), 4 * R.cw_load.size());
        for (int set = 0; set < 2; ++set) {
            const FsHost &F = R.fs[set];
            up(O.fs_row0[set], F.row0.data(), 4 * F.row0.size()); up(O.fs_row1[set], F.row1.data(), 4 * F.row1.size()); up(O.fs_batch[set], F.batch_start.data(), 4 * F.batch_start.size());
            up(O.fs_bobs[set], F.b_obs_start.data(), 4 * F.b_obs_start.size()); up(O.fs_brun[set], F.b_run_start.data(), 4 * F.b_run_start.size()); up(O.fs_bfmt[set], F.b_fmt.data(), 4 * F.b_fmt.size());
            up(O.fs_pobs[set], F.pobs.data(), 4 * F.pobs.size()); up(O.fs_puv[set], F.puv.data(), 8 * F.puv.size()); up(O.fs_pairs[set], F.pairs.data(), 2 * F.pairs.size());
            up(O.fs_rowoff[set], F.rowoff.data(), 4 * F.rowoff.size()); up(O.fs_yoff[set], F.yoff.data(), 4 * F.yoff.size());
        }
        up(O.op_pose, R.op_pose.data(), 4 * R.op_pose.size()); up(O.op_o, R.op_o.data(), 4 * R.op_o.size()); up(O.op_uvi, R.op_uvi.data(), sizeof(double) * R.op_uvi.size());
        up(O.edge_i, Q.edge_i, 4 * Q.n_pose_edge); up(O.edge_j, Q.edge_j, 4 * Q.n_pose_edge); up(O.edge_meas, Q.edge_meas, 7 * Q.n_pose_edge * D); up(O.edge_info, Q.edge_info, 36 * Q.n_pose_edge * D);
        BaProb &H = B->host[p];
        char *a = B->d_arena;
        H.n_pose = Q.n_pose; H.n_point = Q.n_point; H.n_obs = Q.n_obs; H.n_edge = Q.n_pose_edge; H.np_free = R.np_free; H.n6 = 6 * R.np_free;
        H.max_iters = Q.max_iters; H.huber = Q.huber_delta;
#define PTR(T, f) reinterpret_cast<T *>(a + O.f)
        H.pose = PTR(double, pose); H.pose_bk = PTR(double, pose_bk); H.pose0 = PTR(double, pose0);
        H.point = PTR(double, point); H.point_bk = PTR(double, point_bk); H.point0 = PTR(double, point0);
        H.pidx = PTR(int32_t, pidx); H.point_fixed = Q.point_fixed ? PTR(uint8_t, pfix) : nullptr;
        H.obs_pose = PTR(int32_t, obs_pose); H.obs_point = PTR(int32_t, obs_point); H.obs_uv = PTR(double, obs_uv); H.obs_info = PTR(double, obs_info);
        H.pt_start = PTR(int32_t, pt_start); H.pt_obs = PTR(int32_t, pt_obs); H.fstart = PTR(int32_t, fstart); H.fobs = PTR(int32_t, fobs); H.fo_lo = R.fo_lo.empty() ? nullptr : PTR(int32_t, fo_lo); H.fo_uvi = R.fo_lo.empty() ? nullptr : PTR(double, fo_uvi); H.free2pose = PTR(int32_t, free2pose);
        H.edge_i = PTR(int32_t, edge_i); H.edge_j = PTR(int32_t, edge_j); H.edge_meas = PTR(double, edge_meas); H.edge_info = PTR(double, edge_info);
        H.Hpp = PTR(double, Hpp); H.S = PTR(double, S); H.bp = PTR(double, bp); H.dp = PTR(double, dp); H.y = PTR(double, y);
        H.Hll = PTR(double, Hll); H.bl = PTR(double, bl); H.Hinv = PTR(double, Hinv); H.Hpl = PTR(double, Hpl); H.dl = PTR(double, dl);
        H.chi2_obs = PTR(double, chi2); H.stats = PTR(double, stats);
        H.n_chunks = R.n_chunks; H.n_seg = R.n_seg; H.chunk_items = PTR(int32_t, chunk_items); H.seg_start = PTR(int32_t, seg_start); H.seg_pair = PTR(int32_t, seg_pair);
        H.env16 = PTR(int32_t, env16);
        H.act_start = PTR(int32_t, act_start); H.act_blk = PTR(int32_t, act_blk);
        H.fs_cs = PTR(int32_t, fs_cs);
        H.cw_W = R.cw_W; H.cw_zglobal = R.cw_zglobal ? 1 : 0; H.cw_meta_lds = R.cw_meta_lds ? 1 : 0; H.cw_slot = PTR(int32_t, cw_slot); H.cw_act_start = PTR(int32_t, cw_act_start); H.cw_act = PTR(int32_t, cw_act);
        H.cw_load_start = PTR(int32_t, cw_load_start); H.cw_load = PTR(int32_t, cw_load);
        for (int set = 0; set < 2; ++set) {
            FsSet &F = H.fs[set];
            if (!R.fused) { std::memset(&F, 0, sizeof(F)); continue; }
            if (set != R.fs_only) continue;
            F.n_pass = (int32_t)R.fs[set].row0.size();
            F.row0 = PTR(int32_t, fs_row0[set]); F.row1 = PTR(int32_t, fs_row1[set]); F.batch_start = PTR(int32_t, fs_batch[set]);
            F.b_obs_start = PTR(int32_t, fs_bobs[set]); F.b_run_start = PTR(int32_t, fs_brun[set]); F.b_fmt = PTR(int32_t, fs_bfmt[set]); F.pobs = PTR(int32_t, fs_pobs[set]); F.puv = PTR(double, fs_puv[set]);
            F.pairs = PTR(uint16_t, fs_pairs[set]); F.rowoff = PTR(int32_t, fs_rowoff[set]); F.yoff = PTR(int32_t, fs_yoff[set]); F.by_points = R.fs[set].by_points ? 1 : 0;
        }
        if (R.fused) H.fs[1 - R.fs_only] = H.fs[R.fs_only];          // the set that was not built aliases the one that was
        H.fused = R.fused ? 1 : 0;
        H.Y = PTR(double, Y); H.zrow = PTR(double, zrow);
        H.dinv = PTR(double, dinv);
        H.panG = R.np_free > kMaxFreePoses ? PTR(double, panG) : nullptr;
        H.bar = PTR(uint32_t, bar); H.red = PTR(double, red); H.flag = PTR(int32_t, flag); H.team = 1; H.chol_team = 1; H.debug_reject = 0;
        H.op_pose = R.one_pose ? PTR(int32_t, op_pose) : nullptr; H.op_o = R.one_pose ? PTR(int32_t, op_o) : nullptr; H.op_uvi = R.one_pose ? PTR(double, op_uvi) : nullptr;
        H.op_rec = R.one_pose ? PTR(double, op_rec) : nullptr; H.op_red = R.one_pose ? PTR(double, op_red) : nullptr;
        if (p == 0) B->one_pose = R.one_pose; else B->one_pose = B->one_pose && R.one_pose;
        if (n == 1 && B->pack_doubles) B->d_pack = PTR(double, pack);
        H.pack = (n == 1 && B->pack_doubles) ? PTR(double, pack) : nullptr;
        B->alt_ptrs.push_back(!R.fo_lo.empty() && R.fused ? ms_ba::AltPtrs{PTR(double, Hpp2), PTR(double, bp2), PTR(double, Hll2), PTR(double, bl2)} : ms_ba::AltPtrs{nullptr, nullptr, nullptr, nullptr});
        B->chol_tiles.push_back(R.chol_tiles);
        {   // poseBundleAdjust-shaped: one free pose, no free point
            bool po = R.np_free == 1 && ba_edges_at_free_pose(problems[p], R.free2pose[0]) <= PO_MAXE;      // (edges between fixed poses are constants: any number)
            for (int l = 0; l < problems[p].n_point && po; ++l) po = problems[p].point_fixed && problems[p].point_fixed[l];
            if (p == 0) B->pose_only = po; else B->pose_only = B->pose_only && po;
        }
#undef PTR
        if (n == 1) {                                       // the two descriptors ride in the input range (O.desc)
            ba_make_alt(B);
            std::memcpy(stage + (O.desc - in_lo[p]), &B->host[0], sizeof(BaProb));
            std::memcpy(stage + (O.desc - in_lo[p]) + sizeof(BaProb), &B->host_alt[0], sizeof(BaProb));
        }
        // (the shared staging block is written again for the next problem: wait; the context's own block holds every problem's inputs side by side: no wait)
        if (hipMemcpyAsync(B->d_arena + in_lo[p], stage, stage_bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            (!staged && hipStreamSynchronize(c->stream) != hipSuccess)) { ms_ba_destroy(B); return ms_fail(c, MS_ERR_HIP, "ms_ba_create: upload failed"); }
    }
    {
        hipError_t e;
        ba_make_alt(B);
        if (n == 1) {                                       // (went up with the inputs)
            e = staged ? hipEventRecord(c->ba_stage_ev, c->stream) : hipSuccess;
            if (staged) c->ba_stage_busy = e == hipSuccess;
        } else if (staged) {                                // the descriptors (both sets, side by side like on the device) follow the inputs out of the same block; its next user waits for ba_stage_ev
            char *at = static_cast<char *>(c->ba_stage) + stage_at;
            std::memcpy(at, B->host.data(), sizeof(BaProb) * n);
            std::memcpy(at + sizeof(BaProb) * n, B->host_alt.data(), sizeof(BaProb) * n);
            e = hipMemcpyAsync(B->d_probs, at, 2 * sizeof(BaProb) * n, hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) e = hipEventRecord(c->ba_stage_ev, c->stream);
            c->ba_stage_busy = e == hipSuccess;
        } else {
            e = hipMemcpy(B->d_probs, B->host.data(), sizeof(BaProb) * n, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(B->d_probs_alt, B->host_alt.data(), sizeof(BaProb) * n, hipMemcpyHostToDevice);
        }
        // the solvers' dynamic LDS sizes: once per device and process
        static std::atomic<unsigned long long> attr_done{0};
        if (e == hipSuccess && !((attr_done.load() >> (c->device & 63)) & 1ull)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_ba_lm), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void *>(k_ba_pose_only), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPoLdsBytes) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void *>(k_ba_one_pose<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kOpLdsBytes) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void *>(k_ba_one_pose<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kOpLdsBytes) != hipSuccess) e = hipErrorUnknown;
            else attr_done.fetch_or(1ull << (c->device & 63));
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            ms_ba_destroy(B);
            return ms_fail(c, MS_ERR_HIP, "ms_ba_create: device setup failed");
        }
    }
    if (tm_on && std::getenv("MS_BA_TIMING")[0] == '2') std::fprintf(stderr, "  fused-pass batches: passes %.3f, point keys %.3f, sort %.3f, batches %.3f, values %.3f ms\n", tm_part[5], tm_part[6], tm_part[7], tm_part[8], tm_part[9]);
    if (tm_on) std::fprintf(stderr, "ms_ba_create: prep %.3f ms (CSR + envelope %.3f, record lists + panel lists %.3f, fused-pass batches %.3f, rest %.3f), alloc+upload %.3f ms, arena %.1f MB\n",
                            tm1 - tm0, tm_part[0], tm_part[2], tm_part[3], tm1 - tm0 - tm_part[0] - tm_part[2] - tm_part[3], tm_now() - tm1, total / 1e6);
    *out = B;
    return MS_OK;
}

void ms_ba_destroy(ms_ba *B) {
    if (!B) return;
    (void)hipSetDevice(B->ctx->device);
    if (B->pending) (void)ba_wait_event(B->ctx, B->ev_done);
    // (a handle whose last launch has been waited for and that enqueued nothing since leaves the stream alone: hipStreamSynchronize on a stream the runtime has
    //  not itself seen idle costs ~15 us, per frame for poseBundleAdjust)
    const bool synced = !B->quiet || B->pending;
    if (synced) (void)hipStreamSynchronize(B->ctx->stream);
    if (synced) {   // every team launch of this stream has finished: its entries leave the admission list now, before the stream itself can go away (an event
        // queried after its stream was destroyed answered "operation not permitted on an event last recorded in a capturing stream" once in ~20 runs)
        std::lock_guard<std::mutex> lk(g_team_mu);
        for (TeamLaunch &t : g_team_live[B->ctx->device & 63]) if (t.live && t.stream == B->ctx->stream) t.live = false;
    }
    if (B->d_arena) {                                               // back to the context's cache; when that is full the smallest block goes
        ms_ctx *c = B->ctx;
        int slot = -1, small = 0;
        for (int i = 0; i < 4; ++i) if (!c->ba_cache[i].p) { slot = i; break; }
        for (int i = 1; i < 4; ++i) if (c->ba_cache[i].bytes < c->ba_cache[small].bytes) small = i;
        const bool evict = slot < 0 && c->ba_cache[small].bytes < B->arena_bytes;      // full: the new block replaces the smallest one if it is larger
        size_t kept = 0;
        for (int i = 0; i < 4; ++i) if (!(evict && i == small)) kept += c->ba_cache[i].bytes;
        // the cap is tested BEFORE anything is evicted: a block that will not be kept anyway (a global-BA sized arena) must not cost the cache a warm local-BA block
        // (hipFree synchronises the whole device, and the next per-keyframe create would have to hipMalloc again)
        if ((slot >= 0 || evict) && kept + B->arena_bytes <= kBaCacheMaxBytes) {
            if (evict) { (void)hipFree(c->ba_cache[small].p); slot = small; }
            c->ba_cache[slot].p = B->d_arena; c->ba_cache[slot].bytes = B->arena_bytes;
        } else (void)hipFree(B->d_arena);
    }
    {   // the handle OBJECT goes back to the context as well (emptied; its vectors' capacity and its event are what the next window per keyframe re-uses)
        ms_ctx *c = B->ctx;
        void **slot = nullptr;
        for (void *&sl : c->ba_handle_pool) if (!sl) { slot = &sl; break; }
        if (slot) {
            const hipEvent_t ev = B->ev_done;
            void *const hres = B->h_result; const size_t hres_bytes = B->h_result_bytes;
            int32_t *const hver = B->h_verdict;
            std::vector<BaProb> host = std::move(B->host), host_alt = std::move(B->host_alt); std::vector<ms_ba_problem> dims = std::move(B->dims); std::vector<double> tiles = std::move(B->chol_tiles);
            std::vector<ms_ba::AltPtrs> alt_ptrs = std::move(B->alt_ptrs);
            std::vector<size_t> wlo = std::move(B->work_lo), whi = std::move(B->work_hi);
            host.clear(); host_alt.clear(); dims.clear(); tiles.clear(); alt_ptrs.clear(); wlo.clear(); whi.clear();
            *B = ms_ba();
            B->work_lo = std::move(wlo); B->work_hi = std::move(whi);
            B->host = std::move(host); B->host_alt = std::move(host_alt); B->alt_ptrs = std::move(alt_ptrs); B->dims = std::move(dims); B->chol_tiles = std::move(tiles); B->ev_done = ev;
            B->h_result = hres; B->h_result_bytes = hres_bytes; B->h_verdict = hver;
            *slot = B;
            return;
        }
    }
    ba_delete_object(B);
}

int ms_ba_set_team(ms_ba *B, int workgroups_per_problem) {
    if (!B || workgroups_per_problem < 0 || workgroups_per_problem > kMaxTeam) return MS_ERR_INVALID;
    B->team = workgroups_per_problem;
    return MS_OK;
}

int ms_ba_set_factor_team(ms_ba *B, int workgroups) {
    if (!B || workgroups < 0 || workgroups > kMaxTeam) return MS_ERR_INVALID;
    B->factor_team = workgroups;
    return MS_OK;
}

// what follows every solver launch of ms_ba_solve: the eager results of a small single problem, then the handle's completion event
static int ba_after_launch(ms_ctx *c, ms_ba *B) {
    B->eager = false; B->verdict_eager = false;
    static const bool no_eager_verdict = std::getenv("MS_BA_NO_EAGER_VERDICT") != nullptr;                       // (experiment knob)
    if (!B->team_checked && (B->n > 1 || B->last_one_pose) && !no_eager_verdict) {      // a team launch: its verdict travels behind it, so that whoever needs it (ms_ba_copy_state between the
        if (!B->h_verdict) {                                       // two stages, ms_ba_download of a batch) finds it on the host once the launch's event has been seen
                                                                   // (a single window's stage 2 is read by ms_ba_download, whose packed results carry the marker: nothing extra)
            MS_HIP(c, hipHostMalloc(reinterpret_cast<void **>(&B->h_verdict), 64, hipHostMallocDefault)); ++g_ba_host_allocs;
        }
        hipLaunchKernelGGL(k_ba_collect_gave_up, dim3(1), dim3(64), 0, c->stream, B->d_probs, B->n);
        MS_KERNEL_CHECK(c, "k_ba_collect_gave_up");
        MS_HIP(c, hipMemcpyAsync(B->h_verdict, B->host[0].flag + 2, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        B->verdict_eager = true;
    }
    if (B->d_pack && !B->last_one_pose) {                          // (stage 1 of a window is handed on by ms_ba_copy_state, not downloaded: nothing to pack)
        if (!B->self_packed) hipLaunchKernelGGL(k_ba_pack_result, dim3((unsigned)std::min<size_t>(ms_div_up((int)B->pack_doubles, 256), 256)), dim3(256), 0, c->stream, B->d_probs, 0, B->d_pack, 1);
        MS_KERNEL_CHECK(c, "k_ba_pack_result");
        MS_HIP(c, hipMemcpyAsync(B->h_result, B->d_pack, B->pack_doubles * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        B->eager = true;
    }
    return ba_launch_done(c, B);
}

int ms_ba_solve(ms_ba *B) {
    MsRange range("ms_ba_solve");
    if (!B) return MS_ERR_INVALID;
    ms_ctx *c = B->ctx;
    MS_HIP(c, hipSetDevice(c->device));
    if (!B->cus) B->cus = c->n_cu;
    const hipStream_t ls = c->stream;
    // A workgroup takes more than half a CU's LDS, so one per CU: a team is only possible while problems x team fits the chip
    // (all its workgroups must be resident for the barriers), hence the cooperative launch below.  Automatic choice: as many
    // workgroups per problem as fit, at most 32 (beyond that the single-workgroup Cholesky dominates).
    if (B->pose_only && B->team <= 1 && !std::getenv("MS_BA_NO_POSE_KERNEL")) {      // (an explicit team request keeps the general kernel: tests compare the two)
        hipLaunchKernelGGL(k_ba_pose_only, dim3(B->n), dim3(PO_NT), kPoLdsBytes, ls, B->d_probs);
        MS_KERNEL_CHECK(c, "k_ba_pose_only");
        B->launched_team = 1; B->team_checked = true; B->last_one_pose = false; ++B->solves;
        B->self_packed = B->d_pack != nullptr;
        return ba_after_launch(c, B);
    }
    if (B->work_dirty) {                                   // a pose-only handle on the general kernel (an explicit team, MS_BA_NO_POSE_KERNEL): its work areas were never cleared
        for (size_t p = 0; p < B->work_lo.size(); ++p) MS_HIP(c, hipMemsetAsync(B->d_arena + B->work_lo[p], 0, B->work_hi[p] - B->work_lo[p], ls));
        B->work_dirty = false;
    }
    int most_obs = 0, most_points = 0, most_poses = 0;
    for (const auto &h : B->host) { most_obs = std::max(most_obs, h.n_obs); most_points = std::max(most_points, h.n_point); most_poses = std::max(most_poses, h.n_pose); }
    // stage 1 of localBundleAdjust (one free pose, free points): a kernel of its own, with or without a team
    const bool one_pose = B->one_pose && !std::getenv("MS_BA_NO_ONE_POSE_KERNEL");
    int team = B->team == 0 ? (one_pose ? std::min(8, std::max(1, most_obs / 2048)) : std::min(32, std::max(1, most_obs / 512))) : B->team;      // small problems are latency-bound on the barriers
    team = std::max(1, std::min(team, B->cus / std::max(B->n, 1)));
    int lgG = 0;                                                          // lanes per point: as many as the launch has to spare, at most 8
    if (one_pose) {
        if (const char *e = std::getenv("MS_BA_ONE_POSE_LANES")) { const int g = std::atoi(e); lgG = g >= 8 ? 3 : g >= 4 ? 2 : g >= 2 ? 1 : 0; }
        else while (lgG < 3 && (long long)(team * OP_NT >> (lgG + 1)) >= most_points) ++lgG;
    }
    const int op_pose_doubles = most_poses <= OP_MAX_LDS_POSES ? (7 * most_poses + 1) & ~1 : 0;
    auto launch_lm = [&](int tm) {
        if (one_pose) {
            const size_t lds = ((size_t)op_pose_doubles + (size_t)OP_NW * 27 * op_slab_cap(lgG)) * sizeof(double);
            if ((long long)(tm * OP_NT >> lgG) >= most_points) hipLaunchKernelGGL(k_ba_one_pose<true>, dim3(B->n * tm), dim3(OP_NT), lds, ls, B->d_probs, tm, lgG, op_pose_doubles);
            else hipLaunchKernelGGL(k_ba_one_pose<false>, dim3(B->n * tm), dim3(OP_NT), lds, ls, B->d_probs, tm, lgG, op_pose_doubles);
        } else hipLaunchKernelGGL(k_ba_lm, dim3(B->n * tm), dim3(NT), kLdsBytes, ls, B->d_probs, tm, (tm == 1 && B->alt_ok) ? B->d_probs_alt : static_cast<const BaProb *>(nullptr));
    };
    B->last_one_pose = one_pose; B->one_pose_lg = lgG;
    // the distributed factorisation is barrier-bound on banded systems: it gets one workgroup per 16 row tiles a panel touches
    bool changed = team != B->host[0].team;
    for (int i = 0; i < B->n; ++i) {
        const int ct = std::max(1, std::min(team, B->factor_team > 0 ? B->factor_team : (int)std::ceil(B->chol_tiles[(size_t)i] / 16.0)));
        changed |= ct != B->host[(size_t)i].chol_team;
        B->host[(size_t)i].chol_team = ct;
    }
    if (changed) {
        for (auto &h : B->host) h.team = team;
        MS_HIP(c, ba_upload_descriptors(B, ls));
    }
    // A plain launch: every workgroup needs more than half a CU's LDS and problems x team <= CUs, so all of a team's workgroups
    // become resident as soon as CUs are free.  What the spin barriers additionally need is that no OTHER team launch holds CUs
    // while waiting for its own missing workgroups (two half-resident teams would wait for each other until the give-up): the
    // team launches of one process are therefore admitted per device so that the workgroups of all of them together fit the CUs --
    // a new launch waits (on the device, through events) for as many of the oldest running ones of OTHER streams as it takes; eight
    // 32-workgroup solves of eight contexts run side by side, two 160-workgroup ones run one after the other.  Launches without a
    // team (one workgroup per problem: every batch) never wait for anything and are not counted.  A kernel of another stream that
    // keeps CUs busy only delays a team; should a barrier still give up (~1 s without progress), ms_ba_download solves the batch
    // again with one workgroup per problem.
    if (team > 1) {
        std::lock_guard<std::mutex> lk(g_team_mu);
        std::vector<TeamLaunch> &live = g_team_live[c->device & 63];
        const int need = B->n * team;
        const int team_cap = B->cus;                                   // (a lower cap -- 224, 192, 128 of the 256 CUs left to the team launches of a device -- changed nothing for 8 sequences, round 4)
        int in_use = 0;
        for (TeamLaunch &t : live) {                                   // retire what has finished; launches of this stream precede the new one anyway
            if (t.live) {                                              // anything but "not ready" retires the entry (an error: the stream it was recorded on is gone)
                const hipError_t q = hipEventQuery(t.ev);
                if (q != hipErrorNotReady) {
                    t.live = false;
                    if (q != hipSuccess) {                             // not swallowed: counted, and the text stays with this context
                        ++g_team_query_errors;
                        (void)ms_fail(c, MS_ERR_HIP, "team admission: hipEventQuery of an earlier launch failed: %s (entry retired; %d so far)", hipGetErrorString(q), g_team_query_errors);
                        (void)hipGetLastError();
                    }
                }
            }
            if (t.live && t.stream != ls) in_use += t.wgs;
        }
        for (TeamLaunch &t : live) {                                   // oldest first: wait for as many as it takes to make room
            if (in_use + need <= team_cap) break;
            if (!t.live || t.stream == ls) continue;
            MS_HIP(c, hipStreamWaitEvent(ls, t.ev, 0));                // on the device
            in_use -= t.wgs;
        }
        hipLaunchKernelGGL(k_ba_team_reset, dim3(ms_div_up(B->n, 64)), dim3(64), 0, ls, B->d_probs, B->n, B->debug_fail_barriers);
        launch_lm(team);
        MS_KERNEL_CHECK(c, "k_ba_lm");
        TeamLaunch *slot = nullptr;
        for (TeamLaunch &t : live) if (!t.live) { slot = &t; break; }
        if (!slot) {
            if (live.size() >= 64) {                                   // (never in practice: 64 unfinished team launches) -- fall back to waiting for the oldest
                MS_HIP(c, hipEventSynchronize(live.front().ev));
                slot = &live.front();
            } else {
                live.push_back(TeamLaunch{nullptr, nullptr, 0, false});
                slot = &live.back();
                MS_HIP(c, hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming));
                ++g_ba_host_allocs;
            }
        }
        MS_HIP(c, hipEventRecord(slot->ev, ls));
        slot->stream = ls; slot->wgs = need; slot->live = true;
        if (slot != &live.back()) std::rotate(slot, slot + 1, &live.back() + 1);      // keep the list in launch order (oldest first)
    } else {
        launch_lm(team);
        MS_KERNEL_CHECK(c, "k_ba_lm");
    }
    B->launched_team = team;
    B->team_checked = team == 1;
    B->self_packed = false;
    ++B->solves;
    return ba_after_launch(c, B);
}

// one workgroup per problem, no team barriers: the fallback after a team barrier gave up
static int ba_relaunch_single(ms_ba *B) {
    ms_ctx *c = B->ctx;
    const hipStream_t ls = c->stream;
    for (auto &h : B->host) { h.team = 1; h.chol_team = 1; }
    B->eager = false; B->verdict_eager = false; B->self_packed = false;   // (what h_result / h_verdict hold belongs to the void launch)
    MS_HIP(c, ba_upload_descriptors(B, ls));
    if (B->last_one_pose) {
        int most_poses = 0, most_points = 0;
        for (const auto &h : B->host) { most_poses = std::max(most_poses, h.n_pose); most_points = std::max(most_points, h.n_point); }
        const int pd = most_poses <= OP_MAX_LDS_POSES ? (7 * most_poses + 1) & ~1 : 0;
        const size_t lds = ((size_t)pd + (size_t)OP_NW * 27 * op_slab_cap(0)) * sizeof(double);
        if (most_points <= OP_NT) hipLaunchKernelGGL(k_ba_one_pose<true>, dim3(B->n), dim3(OP_NT), lds, ls, B->d_probs, 1, 0, pd);
        else hipLaunchKernelGGL(k_ba_one_pose<false>, dim3(B->n), dim3(OP_NT), lds, ls, B->d_probs, 1, 0, pd);
    } else hipLaunchKernelGGL(k_ba_lm, dim3(B->n), dim3(NT), kLdsBytes, ls, B->d_probs, 1, B->alt_ok ? B->d_probs_alt : static_cast<const BaProb *>(nullptr));
    MS_KERNEL_CHECK(c, "k_ba_lm");
    MS_TRY_BA(ba_launch_done(c, B));
    MS_TRY_BA(ba_wait_pending(B));
    MS_HIP(c, hipStreamSynchronize(c->stream));
    B->launched_team = 1;
    B->team_checked = true;
    ++B->team_fallbacks;
    return MS_OK;
}

// The last launch was a team launch nobody has looked at yet: did a barrier give up anywhere in it?  The markers are collected by a kernel of their own AFTER the launch
// (a marker stored by a workgroup on another XCD is only certain to be visible once its kernel has ended -- also for a single problem, whose in-kernel stats[7] is
// read by one lane of the first workgroup while the others may still be storing), and a launch with one set is repeated with one workgroup per problem before
// anything of it is handed on -- to the caller (ms_ba_download) or to another handle (ms_ba_copy_state).
static int ba_team_verdict(ms_ba *B) {
    if (B->team_checked) return MS_OK;          // (a launch without teams has no verdict to wait for: what follows it on the stream is ordered behind it anyway)
    MS_TRY_BA(ba_wait_pending(B));
    ms_ctx *c = B->ctx;
    int any = 0;
    if (B->verdict_eager) any = *B->h_verdict;                    // collected and copied behind the launch: the wait above covered it
    else {
        hipLaunchKernelGGL(k_ba_collect_gave_up, dim3(1), dim3(64), 0, c->stream, B->d_probs, B->n);
        MS_KERNEL_CHECK(c, "k_ba_collect_gave_up");
        MS_HIP(c, hipMemcpyAsync(&any, B->host[0].flag + 2, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        MS_HIP(c, hipStreamSynchronize(c->stream));
    }
    if (any) MS_TRY_BA(ba_relaunch_single(B));
    B->team_checked = true;
    return MS_OK;
}

int ms_ba_download(ms_ba *B, int i, double *pose, double *point, double *chi2_per_obs, ms_ba_result *res) {
    MsRange range("ms_ba_download");
    if (!B || i < 0 || i >= B->n) return MS_ERR_INVALID;
    ms_ctx *c = B->ctx;
    MS_HIP(c, hipSetDevice(c->device));
    const BaProb &H = B->host[i];
    // status first: a failed solve must not overwrite the caller's arrays (the host mirrors pass the window itself as output)
    // A team barrier that gave up anywhere in the batch voids the whole launch (the workgroups of one launch share the chip): before the FIRST
    // problem of a team launch is handed out, every problem's marker is looked at and the batch is solved again without teams if one is set --
    // never after some results have already been returned.
    MS_TRY_BA(ba_wait_pending(B));
    if (B->n > 1) MS_TRY_BA(ba_team_verdict(B));          // (a single problem's marker arrives with its results below: no extra round trip)
    // everything the caller asked for in ONE device-to-host copy (status, poses, points, per-observation chi2 packed side by side by a small kernel, into the
    // context's pinned staging block): four blocking copies were 0.09 ms of a 2.3 ms window
    const bool want_chi2 = chi2_per_obs && H.n_obs;
    const size_t n_st = 16, n_pose7 = 7 * (size_t)H.n_pose, n_pt3 = 3 * (size_t)H.n_point, n_all = n_st + n_pose7 + n_pt3 + (want_chi2 ? (size_t)H.n_obs : 0);
    void *scr = nullptr;
    if (!B->eager) {
        MS_TRY_BA(ms_scratch(c, n_all * sizeof(double), &scr));
        MS_TRY_BA(ms_pinned(c, n_all * sizeof(double)));
    }
    const double *st = nullptr;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (B->eager) st = static_cast<const double *>(B->h_result);       // packed and copied behind the launch: the wait above covered it (chi2 always included)
        else {
            if (!scr) { MS_TRY_BA(ms_scratch(c, n_all * sizeof(double), &scr)); MS_TRY_BA(ms_pinned(c, n_all * sizeof(double))); }
            hipLaunchKernelGGL(k_ba_pack_result, dim3((unsigned)std::min<size_t>(ms_div_up((int)std::min<size_t>(n_all, 1u << 30), 256), 256)), dim3(256), 0, c->stream, B->d_probs, i, static_cast<double *>(scr), want_chi2 ? 1 : 0);
            MS_KERNEL_CHECK(c, "k_ba_pack_result");
            MS_HIP(c, hipMemcpyAsync(c->pinned, scr, n_all * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            MS_HIP(c, hipStreamSynchronize(c->stream));
            st = static_cast<const double *>(c->pinned);
        }
        if (B->team_checked || st[7] == 0) break;            // a team barrier gave up: solve again without a team, then fetch again
        MS_TRY_BA(ba_relaunch_single(B));
    }
    B->team_checked = true;
    if (res) {
        for (int k = 0; k < 8; ++k) res->phase_cycles[k] = st[8 + k];
        res->iterations = (int)st[0]; res->trials = (int)st[1]; res->stopped_early = (int)st[2]; res->final_lambda = st[3];
        res->chi2_initial = st[4]; res->chi2_final = st[5];
    }
    if (st[6] == 0) return ms_fail(c, MS_ERR_NUMERIC, "ms_ba_download: problem %d ended in a non-finite state", i);
    if (pose) std::memcpy(pose, st + n_st, n_pose7 * sizeof(double));
    if (point && H.n_point) std::memcpy(point, st + n_st + n_pose7, n_pt3 * sizeof(double));
    if (want_chi2) std::memcpy(chi2_per_obs, st + n_st + n_pose7 + n_pt3, (size_t)H.n_obs * sizeof(double));
    return MS_OK;
}

int ms_ba_copy_state(ms_ba *dst, const ms_ba *src, const int32_t *extra_pose_src) {
    if (!dst || !src || dst == src || dst->ctx != src->ctx || dst->n != src->n) return MS_ERR_INVALID;
    ms_ctx *c = dst->ctx;
    bool need_extra = false;
    for (int i = 0; i < dst->n; ++i) {
        const BaProb &D = dst->host[(size_t)i], &S = src->host[(size_t)i];
        if (D.n_point != S.n_point || D.n_pose < S.n_pose) return ms_fail(c, MS_ERR_INVALID, "ms_ba_copy_state: problem %d: %d / %d poses, %d / %d points", i, D.n_pose, S.n_pose, D.n_point, S.n_point);
        if (D.n_pose > S.n_pose) {
            need_extra = true;
            if (!extra_pose_src || extra_pose_src[i] < 0 || extra_pose_src[i] >= S.n_pose) return ms_fail(c, MS_ERR_INVALID, "ms_ba_copy_state: problem %d has %d poses more than its source and no valid pose to copy them from", i, D.n_pose - S.n_pose);
        }
    }
    MS_HIP(c, hipSetDevice(c->device));
    if (src->solves == 0) return ms_fail(c, MS_ERR_INVALID, "ms_ba_copy_state: the source handle has never been solved");
    // a source whose last launch ran on teams is looked at first (and repeated without teams if a barrier gave up): nothing of a void launch becomes anybody's initial state
    MS_TRY_BA(ba_team_verdict(const_cast<ms_ba *>(src)));
    int32_t *d_extra = nullptr;
    if (need_extra) {
        void *scr = nullptr;
        MS_TRY_BA(ms_scratch(c, sizeof(int32_t) * (size_t)dst->n, &scr));
        d_extra = static_cast<int32_t *>(scr);
        MS_HIP(c, hipMemcpyAsync(d_extra, extra_pose_src, sizeof(int32_t) * (size_t)dst->n, hipMemcpyHostToDevice, c->stream));
    }
    dst->quiet = false; const_cast<ms_ba *>(src)->quiet = false;          // (both arenas are in use on the stream again)
    hipLaunchKernelGGL(k_ba_copy_state, dim3(dst->n), dim3(256), 0, c->stream, dst->d_probs, src->d_probs, d_extra);
    MS_KERNEL_CHECK(c, "k_ba_copy_state");
    return MS_OK;
}

int ms_ba_team_fallbacks(const ms_ba *B) { return B ? B->team_fallbacks : MS_ERR_INVALID; }
int ms_ba_admission_errors(void) { std::lock_guard<std::mutex> lk(g_team_mu); return g_team_query_errors; }
int ms_ba_debug_fail_team_barriers(ms_ba *B, int on) { if (!B) return MS_ERR_INVALID; B->debug_fail_barriers = on ? 1 : 0; return MS_OK; }
int ms_ba_debug_force_reject(ms_ba *B, int first_trials) {
    if (!B || first_trials < 0 || first_trials > 255) return MS_ERR_INVALID;
    if (B->pose_only || B->one_pose) return ms_fail(B->ctx, MS_ERR_INVALID, "ms_ba_debug_force_reject: only the general solver (k_ba_lm) has the hook");
    for (auto &h : B->host) h.debug_reject = first_trials;
    MS_HIP(B->ctx, hipSetDevice(B->ctx->device));
    MS_HIP(B->ctx, ba_upload_descriptors(B, B->ctx->stream));          // (both sets: the fused trial schedule reads whichever is in use)
    return MS_OK;
}

int ms_ba_solve_host(ms_ctx *c, const ms_ba_problem *problem, double *pose_out, double *point_out, double *chi2_per_obs, ms_ba_result *res) {
    ms_ba *B = nullptr;
    int rc = ms_ba_create(c, problem, 1, &B);
    if (rc != MS_OK) return rc;
    rc = ms_ba_solve(B);
    if (rc == MS_OK) rc = ms_ba_download(B, 0, pose_out, point_out, chi2_per_obs, res);
    ms_ba_destroy(B);
    return rc;
}

}  // extern "C"

void ms_ba_release_pool(ms_ctx *c) {
    for (void *&sl : c->ba_handle_pool) if (sl) { ba_delete_object(static_cast<ms_ba *>(sl)); sl = nullptr; }
    if (c->ba_stage) { if (c->ba_stage_busy) (void)hipEventSynchronize(c->ba_stage_ev); (void)hipHostFree(c->ba_stage); c->ba_stage = nullptr; c->ba_stage_bytes = 0; }
    if (c->ba_stage_ev) { (void)hipEventDestroy(c->ba_stage_ev); c->ba_stage_ev = nullptr; }
}
