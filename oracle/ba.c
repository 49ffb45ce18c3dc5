/*
 * oracle/ba.c -- CPU ORACLE (test infrastructure only; see mso.h header).
 *
 * Local bundle adjustment as the reference runs it through g2o (bundle_adjuster.cpp:141-394):
 *   vertices  VertexSE3Expmap (pose, world->camera, quaternion + translation), VertexSBAPointXYZ
 *   edges     EdgeSE3ProjectXYZ with fx=fy=1, cx=cy=0 (bundle_adjuster.cpp:43-63), Huber delta sqrt(5.991)
 *             EdgeSE3Expmap for odometry / loop / orientation priors (bundle_adjuster.cpp:65-111, :341-370)
 *   solver    OptimizationAlgorithmLevenberg over BlockSolverX + LinearSolverEigen, NO marginalisation
 *             (bundle_adjuster.cpp:149-154, :269 commented out)
 *
 * g2o and Eigen are not in the reference tree and no version is pinned, so this file restates their
 * published algorithms (g2o types_six_dof_expmap.{h,cpp}, se3quat.h, optimization_algorithm_levenberg.cpp,
 * robust_kernel_impl.cpp, base_binary_edge.hpp): PARITY UNPINNED against the reference's iterates.
 * The damped normal equations are solved either as the full (poses + points) dense system, which is what
 * g2o's un-marginalised BlockSolverX assembles, or through the Schur complement on the points (same step in
 * exact arithmetic; this is what the GPU path does).  tests/ check the two agree.
 */
#include "mso.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- small SE3 / quaternion helpers (Eigen / g2o conventions: q = (x,y,z,w)) ---- */
static void q_normalize(double *q) {                 /* SE3Quat::normalizeRotation */
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void q_mul(const double *a, const double *b, double *r) {   /* Eigen quaternion product a*b */
    r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
static void q_rot(const double *q, const double *v, double *r) {   /* Eigen: v + w*uv + qv x uv, uv = 2 qv x v */
    const double ux = 2 * (q[1] * v[2] - q[2] * v[1]), uy = 2 * (q[2] * v[0] - q[0] * v[2]), uz = 2 * (q[0] * v[1] - q[1] * v[0]);
    r[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
    r[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
    r[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}
static void q_to_R(const double *q, double *R) {      /* Eigen toRotationMatrix, row-major */
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0], tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
static void R_to_q(const double *m, double *q) {      /* Eigen quaternion from rotation matrix */
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
        q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (m[3 * k + j] - m[3 * j + k]) * t;
        q[j] = (m[3 * j + i] + m[3 * i + j]) * t;
        q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
    }
}
/* pose p = (q[4], t[3]); r = a * b (SE3Quat::operator*) */
static void se3_mul(const double *a, const double *b, double *r) {
    double t[3], q[4];
    q_rot(a, b + 4, t);
    q_mul(a, b, q);
    r[0] = q[0]; r[1] = q[1]; r[2] = q[2]; r[3] = q[3];
    r[4] = t[0] + a[4]; r[5] = t[1] + a[5]; r[6] = t[2] + a[6];
    q_normalize(r);
}
static void se3_inv(const double *a, double *r) {     /* SE3Quat::inverse */
    double t[3];
    r[0] = -a[0]; r[1] = -a[1]; r[2] = -a[2]; r[3] = a[3];
    q_rot(r, a + 4, t);
    r[4] = -t[0]; r[5] = -t[1]; r[6] = -t[2];
}
static void mat3_mul(const double *A, const double *B, double *C) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += A[3 * i + k] * B[3 * k + j]; C[3 * i + j] = s; }
}
static void skew(const double *v, double *S) { S[0] = 0; S[1] = -v[2]; S[2] = v[1]; S[3] = v[2]; S[4] = 0; S[5] = -v[0]; S[6] = -v[1]; S[7] = v[0]; S[8] = 0; }

/* SE3Quat::exp(update), update = (omega, upsilon) */
void mso_se3_exp(const double *u, double *pose) {
    const double *om = u, *up = u + 3;
    const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    double O[9], O2[9], R[9], V[9];
    skew(om, O); mat3_mul(O, O, O2);
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (theta < 0.00001) {
        for (int i = 0; i < 9; ++i) { R[i] = I[i] + O[i] + 0.5 * O2[i]; V[i] = I[i] + 0.5 * O[i] + O2[i] / 6.; }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / (theta * theta * theta);
        for (int i = 0; i < 9; ++i) { R[i] = I[i] + a * O[i] + b * O2[i]; V[i] = I[i] + b * O[i] + c * O2[i]; }
    }
    R_to_q(R, pose);
    for (int i = 0; i < 3; ++i) pose[4 + i] = V[3 * i] * up[0] + V[3 * i + 1] * up[1] + V[3 * i + 2] * up[2];
    q_normalize(pose);
}

/* SE3Quat::log() */
void mso_se3_log(const double *pose, double *out) {
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

/* SE3Quat::adj(): [[R,0],[ [t]x R, R ]] with (rotation, translation) ordering; 6x6 row-major */
static void se3_adj(const double *pose, double *A) {
    double R[9], T[9], TR[9];
    q_to_R(pose, R); skew(pose + 4, T); mat3_mul(T, R, TR);
    memset(A, 0, 36 * sizeof(double));
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { A[6 * i + j] = R[3 * i + j]; A[6 * (i + 3) + j + 3] = R[3 * i + j]; A[6 * (i + 3) + j] = TR[3 * i + j]; }
}

/* ---- residuals and Jacobians ---- */
/* EdgeSE3ProjectXYZ (fx=fy=1,cx=cy=0): e = z - proj(T X); Jp 2x6, Jl 2x3 (types_six_dof_expmap.cpp linearizeOplus) */
static void proj_edge(const double *pose, const double *X, const double *uv, double *e, double *Jp, double *Jl) {
    double p[3];
    q_rot(pose, X, p);
    p[0] += pose[4]; p[1] += pose[5]; p[2] += pose[6];
    const double x = p[0], y = p[1], z = p[2];
    e[0] = uv[0] - x / z; e[1] = uv[1] - y / z;
    if (!Jp) return;
    const double z2 = z * z;
    double R[9];
    q_to_R(pose, R);
    const double tmp[6] = {1, 0, -x / z, 0, 1, -y / z};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j)
        Jl[3 * i + j] = -1. / z * (tmp[3 * i] * R[j] + tmp[3 * i + 1] * R[3 + j] + tmp[3 * i + 2] * R[6 + j]);
    Jp[0] = x * y / z2; Jp[1] = -(1 + (x * x / z2)); Jp[2] = y / z; Jp[3] = -1. / z; Jp[4] = 0; Jp[5] = x / z2;
    Jp[6] = (1 + y * y / z2); Jp[7] = -x * y / z2; Jp[8] = -x / z; Jp[9] = 0; Jp[10] = -1. / z; Jp[11] = y / z2;
}

/* EdgeSE3Expmap: e = log(Tj^-1 * M * Ti); Ji = (Tj^-1 M).adj(), Jj = -(Ti^-1 M^-1).adj() */
static void pose_edge(const double *Ti, const double *Tj, const double *M, double *e, double *Ji, double *Jj) {
    double Tjinv[7], A[7], B[7];
    se3_inv(Tj, Tjinv);
    se3_mul(Tjinv, M, A);            /* invTj_Tij */
    se3_mul(A, Ti, B);
    mso_se3_log(B, e);
    if (!Ji) return;
    se3_adj(A, Ji);
    double Tiinv[7], Minv[7], C[7];
    se3_inv(Ti, Tiinv); se3_inv(M, Minv); se3_mul(Tiinv, Minv, C);
    se3_adj(C, Jj);
    for (int i = 0; i < 36; ++i) Jj[i] = -Jj[i];
}

void mso_ba_proj_edge(const double *pose, const double *X, const double *uv, double *e, double *Jp, double *Jl) { proj_edge(pose, X, uv, e, Jp, Jl); }
void mso_ba_pose_edge(const double *Ti, const double *Tj, const double *M, double *e, double *Ji, double *Jj) { pose_edge(Ti, Tj, M, e, Ji, Jj); }
void mso_se3_mul(const double *a, const double *b, double *r) { se3_mul(a, b, r); }

static inline void huber(double chi2, double delta, double *rho0, double *w) {   /* RobustKernelHuber::robustify */
    const double dsqr = delta * delta;
    if (delta <= 0 || chi2 <= dsqr) { *rho0 = chi2; *w = 1; }
    else { const double s = sqrt(chi2); *rho0 = 2 * s * delta - dsqr; *w = delta / s; }
}

/* in-place dense Cholesky (lower), returns 0 on success */
static int chol(double *A, int n) {
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0) || !isfinite(d)) return -1;
        d = sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    return 0;
}
static void chol_solve(const double *L, int n, double *b) {
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * b[k]; b[i] = s / L[(size_t)i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * b[k]; b[i] = s / L[(size_t)i * n + i]; }
}
static int inv3_sym(const double *H, double *Hi) {    /* H: xx,xy,xz,yy,yz,zz */
    const double a = H[0], b = H[1], c = H[2], d = H[3], e = H[4], f = H[5];
    const double A = d * f - e * e, B = c * e - b * f, C = b * e - c * d;
    const double det = a * A + b * B + c * C;
    if (!(fabs(det) > 0) || !isfinite(det)) return -1;
    const double id = 1.0 / det;
    Hi[0] = A * id; Hi[1] = B * id; Hi[2] = C * id; Hi[3] = (a * f - c * c) * id; Hi[4] = (b * c - a * e) * id; Hi[5] = (a * d - b * b) * id;
    return 0;
}

typedef struct {
    int np, nl;                 /* free poses / points */
    int *pidx, *lidx;           /* vertex -> free index or -1 */
    double *Hpp;                /* (6np)^2 dense: pose block incl. pose-pose edges */
    double *bp;                 /* 6np */
    double *Hll;                /* 6 per point (sym) */
    double *bl;                 /* 3 per point */
    double *Hpl;                /* 18 per observation (6x3), only where both free */
} normal_eq;

static double robust_chi2(const mso_ba_problem *P, const double *pose, const double *point, double *chi2_obs) {
    double total = 0;
    for (int o = 0; o < P->n_obs; ++o) {
        double e[2];
        proj_edge(pose + 7 * (size_t)P->obs_pose[o], point + 3 * (size_t)P->obs_point[o], P->obs_uv + 2 * (size_t)o, e, NULL, NULL);
        const double chi2 = P->obs_info[o] * (e[0] * e[0] + e[1] * e[1]);
        double r, w;
        huber(chi2, P->huber_delta, &r, &w);
        if (chi2_obs) chi2_obs[o] = chi2;
        total += r;
    }
    for (int k = 0; k < P->n_edge; ++k) {
        double e[6];
        pose_edge(pose + 7 * (size_t)P->edge_i[k], pose + 7 * (size_t)P->edge_j[k], P->edge_meas + 7 * (size_t)k, e, NULL, NULL);
        const double *W = P->edge_info + 36 * (size_t)k;
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) total += e[i] * W[6 * i + j] * e[j];
    }
    return total;
}

static void build_system(const mso_ba_problem *P, const double *pose, const double *point, normal_eq *N) {
    const int n6 = 6 * N->np;
    memset(N->Hpp, 0, sizeof(double) * (size_t)n6 * n6);
    memset(N->bp, 0, sizeof(double) * n6);
    memset(N->Hll, 0, sizeof(double) * 6 * (size_t)P->n_point);
    memset(N->bl, 0, sizeof(double) * 3 * (size_t)P->n_point);
    for (int o = 0; o < P->n_obs; ++o) {
        const int pi = P->obs_pose[o], li = P->obs_point[o];
        const int fp = N->pidx[pi], fl = N->lidx[li];
        double e[2], Jp[12], Jl[6];
        proj_edge(pose + 7 * (size_t)pi, point + 3 * (size_t)li, P->obs_uv + 2 * (size_t)o, e, Jp, Jl);
        const double info = P->obs_info[o], chi2 = info * (e[0] * e[0] + e[1] * e[1]);
        double r, w;
        huber(chi2, P->huber_delta, &r, &w);
        const double wi = w * info;                                /* robustInformation = rho[1] * information */
        if (fp >= 0) {
            for (int a = 0; a < 6; ++a) {
                N->bp[6 * fp + a] += -(Jp[a] * e[0] + Jp[6 + a] * e[1]) * wi;
                for (int b = 0; b < 6; ++b) N->Hpp[(size_t)(6 * fp + a) * n6 + 6 * fp + b] += wi * (Jp[a] * Jp[b] + Jp[6 + a] * Jp[6 + b]);
            }
        }
        if (fl >= 0) {
            double *H = N->Hll + 6 * (size_t)li, *b = N->bl + 3 * (size_t)li;
            for (int a = 0; a < 3; ++a) b[a] += -(Jl[a] * e[0] + Jl[3 + a] * e[1]) * wi;
            H[0] += wi * (Jl[0] * Jl[0] + Jl[3] * Jl[3]); H[1] += wi * (Jl[0] * Jl[1] + Jl[3] * Jl[4]); H[2] += wi * (Jl[0] * Jl[2] + Jl[3] * Jl[5]);
            H[3] += wi * (Jl[1] * Jl[1] + Jl[4] * Jl[4]); H[4] += wi * (Jl[1] * Jl[2] + Jl[4] * Jl[5]); H[5] += wi * (Jl[2] * Jl[2] + Jl[5] * Jl[5]);
        }
        if (fp >= 0 && fl >= 0) {
            double *W = N->Hpl + 18 * (size_t)o;
            for (int a = 0; a < 6; ++a) for (int c = 0; c < 3; ++c) W[3 * a + c] = wi * (Jp[a] * Jl[c] + Jp[6 + a] * Jl[3 + c]);
        }
    }
    for (int k = 0; k < P->n_edge; ++k) {
        const int vi = P->edge_i[k], vj = P->edge_j[k], fi = N->pidx[vi], fj = N->pidx[vj];
        if (fi < 0 && fj < 0) continue;
        double e[6], Ji[36], Jj[36], We[6];
        pose_edge(pose + 7 * (size_t)vi, pose + 7 * (size_t)vj, P->edge_meas + 7 * (size_t)k, e, Ji, Jj);
        const double *W = P->edge_info + 36 * (size_t)k;
        for (int a = 0; a < 6; ++a) { double s = 0; for (int b = 0; b < 6; ++b) s += W[6 * a + b] * e[b]; We[a] = -s; }   /* omega_r */
        const double *J[2] = {Ji, Jj}; const int f[2] = {fi, fj};
        for (int s = 0; s < 2; ++s) {
            if (f[s] < 0) continue;
            for (int a = 0; a < 6; ++a) { double v = 0; for (int r2 = 0; r2 < 6; ++r2) v += J[s][6 * r2 + a] * We[r2]; N->bp[6 * f[s] + a] += v; }
            for (int t = 0; t < 2; ++t) {
                if (f[t] < 0) continue;
                for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) {
                    double v = 0;
                    for (int r2 = 0; r2 < 6; ++r2) for (int c2 = 0; c2 < 6; ++c2) v += J[s][6 * r2 + a] * W[6 * r2 + c2] * J[t][6 * c2 + b];
                    N->Hpp[(size_t)(6 * f[s] + a) * n6 + 6 * f[t] + b] += v;
                }
            }
        }
    }
}

/* Solve (H + lambda I) dx = b.  dp: 6np, dl: 3 per point (vertex-indexed).  Returns 0 on success. */
static int solve_schur(const mso_ba_problem *P, const normal_eq *N, double lambda, double *dp, double *dl) {
    const int n6 = 6 * N->np;
    double *S = (double *)malloc(sizeof(double) * (size_t)(n6 ? n6 : 1) * (n6 ? n6 : 1));
    double *Hi = (double *)malloc(sizeof(double) * 6 * (size_t)(P->n_point ? P->n_point : 1));
    int rc = 0;
    memcpy(S, N->Hpp, sizeof(double) * (size_t)n6 * n6);
    for (int i = 0; i < n6; ++i) { S[(size_t)i * n6 + i] += lambda; dp[i] = N->bp[i]; }
    for (int l = 0; l < P->n_point; ++l) {
        if (N->lidx[l] < 0) continue;
        double H[6];
        memcpy(H, N->Hll + 6 * (size_t)l, sizeof(H));
        H[0] += lambda; H[3] += lambda; H[5] += lambda;
        if (inv3_sym(H, Hi + 6 * (size_t)l)) rc = -1;
    }
    /* per-point lists of observations */
    int *start = (int *)calloc((size_t)P->n_point + 2, sizeof(int)), *list = (int *)malloc(sizeof(int) * (size_t)(P->n_obs ? P->n_obs : 1));
    for (int o = 0; o < P->n_obs; ++o) start[P->obs_point[o] + 2]++;
    for (int l = 0; l < P->n_point; ++l) start[l + 2] += start[l + 1];
    for (int o = 0; o < P->n_obs; ++o) list[start[P->obs_point[o] + 1]++] = o;
    for (int l = 0; l < P->n_point && rc == 0; ++l) {
        if (N->lidx[l] < 0) continue;
        const double *h = Hi + 6 * (size_t)l, *bl = N->bl + 3 * (size_t)l;
        const double Hinv[9] = {h[0], h[1], h[2], h[1], h[3], h[4], h[2], h[4], h[5]};
        for (int ia = start[l]; ia < start[l + 1]; ++ia) {
            const int oa = list[ia], fa = N->pidx[P->obs_pose[oa]];
            if (fa < 0) continue;
            const double *Wa = N->Hpl + 18 * (size_t)oa;
            double Y[18];                                           /* Wa * Hinv (6x3) */
            for (int a = 0; a < 6; ++a) for (int c = 0; c < 3; ++c) Y[3 * a + c] = Wa[3 * a] * Hinv[c] + Wa[3 * a + 1] * Hinv[3 + c] + Wa[3 * a + 2] * Hinv[6 + c];
            for (int a = 0; a < 6; ++a) dp[6 * fa + a] -= Y[3 * a] * bl[0] + Y[3 * a + 1] * bl[1] + Y[3 * a + 2] * bl[2];
            for (int ib = start[l]; ib < start[l + 1]; ++ib) {
                const int ob = list[ib], fb = N->pidx[P->obs_pose[ob]];
                if (fb < 0) continue;
                const double *Wb = N->Hpl + 18 * (size_t)ob;
                for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b)
                    S[(size_t)(6 * fa + a) * n6 + 6 * fb + b] -= Y[3 * a] * Wb[3 * b] + Y[3 * a + 1] * Wb[3 * b + 1] + Y[3 * a + 2] * Wb[3 * b + 2];
            }
        }
    }
    if (rc == 0 && n6 > 0) { if (chol(S, n6)) rc = -1; else chol_solve(S, n6, dp); }
    if (rc == 0) {
        for (int l = 0; l < P->n_point; ++l) {
            double *d = dl + 3 * (size_t)l;
            d[0] = d[1] = d[2] = 0;
            if (N->lidx[l] < 0) continue;
            double r[3] = {N->bl[3 * (size_t)l], N->bl[3 * (size_t)l + 1], N->bl[3 * (size_t)l + 2]};
            for (int ia = start[l]; ia < start[l + 1]; ++ia) {
                const int oa = list[ia], fa = N->pidx[P->obs_pose[oa]];
                if (fa < 0) continue;
                const double *Wa = N->Hpl + 18 * (size_t)oa;
                for (int c = 0; c < 3; ++c) for (int a = 0; a < 6; ++a) r[c] -= Wa[3 * a + c] * dp[6 * fa + a];
            }
            const double *h = Hi + 6 * (size_t)l;
            d[0] = h[0] * r[0] + h[1] * r[1] + h[2] * r[2];
            d[1] = h[1] * r[0] + h[3] * r[1] + h[4] * r[2];
            d[2] = h[2] * r[0] + h[4] * r[1] + h[5] * r[2];
        }
    }
    free(S); free(Hi); free(start); free(list);
    return rc;
}

/* the same step from the full (poses + points) dense system -- what g2o's un-marginalised solver assembles */
static int solve_full(const mso_ba_problem *P, const normal_eq *N, double lambda, double *dp, double *dl) {
    const int n6 = 6 * N->np, n = n6 + 3 * N->nl;
    double *H = (double *)calloc((size_t)(n ? n : 1) * (n ? n : 1), sizeof(double)), *b = (double *)calloc(n ? n : 1, sizeof(double));
    for (int i = 0; i < n6; ++i) { memcpy(H + (size_t)i * n, N->Hpp + (size_t)i * n6, sizeof(double) * n6); b[i] = N->bp[i]; }
    for (int l = 0; l < P->n_point; ++l) {
        const int fl = N->lidx[l];
        if (fl < 0) continue;
        const double *h = N->Hll + 6 * (size_t)l;
        const int o = n6 + 3 * fl;
        const double M[9] = {h[0], h[1], h[2], h[1], h[3], h[4], h[2], h[4], h[5]};
        for (int a = 0; a < 3; ++a) { b[o + a] = N->bl[3 * (size_t)l + a]; for (int c = 0; c < 3; ++c) H[(size_t)(o + a) * n + o + c] = M[3 * a + c]; }
    }
    for (int ob = 0; ob < P->n_obs; ++ob) {
        const int fp = N->pidx[P->obs_pose[ob]], fl = N->lidx[P->obs_point[ob]];
        if (fp < 0 || fl < 0) continue;
        const double *W = N->Hpl + 18 * (size_t)ob;
        for (int a = 0; a < 6; ++a) for (int c = 0; c < 3; ++c) {
            H[(size_t)(6 * fp + a) * n + n6 + 3 * fl + c] += W[3 * a + c];
            H[(size_t)(n6 + 3 * fl + c) * n + 6 * fp + a] += W[3 * a + c];
        }
    }
    for (int i = 0; i < n; ++i) H[(size_t)i * n + i] += lambda;
    int rc = chol(H, n);
    if (rc == 0) {
        chol_solve(H, n, b);
        memcpy(dp, b, sizeof(double) * n6);
        for (int l = 0; l < P->n_point; ++l) {
            const int fl = N->lidx[l];
            for (int a = 0; a < 3; ++a) dl[3 * (size_t)l + a] = fl < 0 ? 0 : b[n6 + 3 * fl + a];
        }
    }
    free(H); free(b);
    return rc;
}

/* OptimizationAlgorithmLevenberg::solve wrapped in SparseOptimizer::optimize(max_iters) */
/* flags: bit 0 = the un-marginalised system (what the reference solves, bundle_adjuster.cpp:269 commented out); bit 1 = per-observation chi2 as g2o's edge->chi2() would
 * return it after optimize(): the errors of the LAST computeActiveErrors(), i.e. of the last trial even when that trial was rejected and the vertices restored
 * (bundle_adjuster.cpp:376-379 reads exactly that; it differs from the accepted state only when the solve ends on rejected trials -- Terminate); bits 8..15 = test
 * hook shared with the GPU solver: the first n damped trials count as rejected whatever their gain (drives the ten-rejections Terminate path deterministically) */
int mso_ba_solve(mso_ba_problem *P, double *chi2_per_obs, mso_ba_stats *st, int flags) {
    const int full_system = flags & 1, stale_chi2 = (flags >> 1) & 1, force_reject = (flags >> 8) & 0xFF;
    normal_eq N;
    memset(&N, 0, sizeof(N));
    N.pidx = (int *)malloc(sizeof(int) * (size_t)(P->n_pose ? P->n_pose : 1));
    N.lidx = (int *)malloc(sizeof(int) * (size_t)(P->n_point ? P->n_point : 1));
    for (int i = 0; i < P->n_pose; ++i) N.pidx[i] = P->pose_fixed[i] ? -1 : N.np++;
    for (int i = 0; i < P->n_point; ++i) N.lidx[i] = (P->point_fixed && P->point_fixed[i]) ? -1 : N.nl++;
    const int n6 = 6 * N.np;
    N.Hpp = (double *)malloc(sizeof(double) * (size_t)(n6 ? n6 : 1) * (n6 ? n6 : 1));
    N.bp = (double *)malloc(sizeof(double) * (size_t)(n6 ? n6 : 1));
    N.Hll = (double *)malloc(sizeof(double) * 6 * (size_t)(P->n_point ? P->n_point : 1));
    N.bl = (double *)malloc(sizeof(double) * 3 * (size_t)(P->n_point ? P->n_point : 1));
    N.Hpl = (double *)calloc(18 * (size_t)(P->n_obs ? P->n_obs : 1), sizeof(double));
    double *dp = (double *)malloc(sizeof(double) * (size_t)(n6 ? n6 : 1)), *dl = (double *)malloc(sizeof(double) * 3 * (size_t)(P->n_point ? P->n_point : 1));
    double *pose_bk = (double *)malloc(sizeof(double) * 7 * (size_t)(P->n_pose ? P->n_pose : 1)), *point_bk = (double *)malloc(sizeof(double) * 3 * (size_t)(P->n_point ? P->n_point : 1));
    double *pose_tr = (double *)malloc(sizeof(double) * 7 * (size_t)(P->n_pose ? P->n_pose : 1)), *point_tr = (double *)malloc(sizeof(double) * 3 * (size_t)(P->n_point ? P->n_point : 1));      /* the state of the last trial (g2o's edges keep its errors) */
    int have_trial = 0;
    double lambda = 0, ni = 2;
    int it = 0, trials_total = 0, stop = 0;
    st->chi2_init = robust_chi2(P, P->pose, P->point, NULL);
    for (it = 0; it < P->max_iters; ++it) {
        double current = robust_chi2(P, P->pose, P->point, NULL), temp = current;
        build_system(P, P->pose, P->point, &N);
        if (it == 0) {                                              /* computeLambdaInit: tau * max diagonal */
            double md = 0;
            for (int i = 0; i < n6; ++i) md = fmax(md, fabs(N.Hpp[(size_t)i * n6 + i]));
            for (int l = 0; l < P->n_point; ++l) if (N.lidx[l] >= 0) { const double *h = N.Hll + 6 * (size_t)l; md = fmax(md, fmax(fabs(h[0]), fmax(fabs(h[3]), fabs(h[5])))); }
            lambda = 1e-5 * md; ni = 2;
        }
        double rho = 0; int qmax = 0;
        do {
            memcpy(pose_bk, P->pose, sizeof(double) * 7 * (size_t)P->n_pose);             /* push() */
            memcpy(point_bk, P->point, sizeof(double) * 3 * (size_t)P->n_point);
            const int ok2 = (full_system ? solve_full(P, &N, lambda, dp, dl) : solve_schur(P, &N, lambda, dp, dl)) == 0;
            if (ok2) {                                              /* update(x): oplus on every free vertex */
                for (int i = 0; i < P->n_pose; ++i) if (N.pidx[i] >= 0) { double ex[7], r[7]; mso_se3_exp(dp + 6 * N.pidx[i], ex); se3_mul(ex, P->pose + 7 * (size_t)i, r); memcpy(P->pose + 7 * (size_t)i, r, sizeof(r)); }
                for (int l = 0; l < P->n_point; ++l) if (N.lidx[l] >= 0) for (int a = 0; a < 3; ++a) P->point[3 * (size_t)l + a] += dl[3 * (size_t)l + a];
            }
            temp = ok2 ? robust_chi2(P, P->pose, P->point, NULL) : DBL_MAX;
            if (ok2) { memcpy(pose_tr, P->pose, sizeof(double) * 7 * (size_t)P->n_pose); memcpy(point_tr, P->point, sizeof(double) * 3 * (size_t)P->n_point); have_trial = 1; }
            rho = current - temp;
            double scale = 0;                                       /* computeScale: sum x_j (lambda x_j + b_j) */
            if (ok2) {
                for (int i = 0; i < n6; ++i) scale += dp[i] * (lambda * dp[i] + N.bp[i]);
                for (int l = 0; l < P->n_point; ++l) if (N.lidx[l] >= 0) for (int a = 0; a < 3; ++a) scale += dl[3 * (size_t)l + a] * (lambda * dl[3 * (size_t)l + a] + N.bl[3 * (size_t)l + a]);
            }
            scale += 1e-3;
            rho /= scale;
            if (trials_total < force_reject) rho = -1.0;
            if (rho > 0 && isfinite(temp)) {
                double alpha = 1. - pow((2 * rho - 1), 3);
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha);
                ni = 2; current = temp;                              /* discardTop() */
            } else {
                lambda *= ni; ni *= 2;
                memcpy(P->pose, pose_bk, sizeof(double) * 7 * (size_t)P->n_pose);         /* pop() */
                memcpy(P->point, point_bk, sizeof(double) * 3 * (size_t)P->n_point);
                if (!isfinite(lambda)) break;
            }
            ++qmax; ++trials_total;
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0 || !isfinite(lambda)) { stop = 1; ++it; break; }       /* Terminate */
    }
    st->iters = it; st->lambda = lambda; st->trials_total = trials_total; st->stop_reason = stop;
    st->chi2_final = robust_chi2(P, P->pose, P->point, chi2_per_obs);
    if (stale_chi2 && have_trial && chi2_per_obs) (void)robust_chi2(P, pose_tr, point_tr, chi2_per_obs);
    free(pose_tr); free(point_tr);
    free(N.pidx); free(N.lidx); free(N.Hpp); free(N.bp); free(N.Hll); free(N.bl); free(N.Hpl); free(dp); free(dl); free(pose_bk); free(point_bk);
    return 0;
}
