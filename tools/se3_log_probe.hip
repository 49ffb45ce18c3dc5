// se3_log_probe.hip -- how long does ONE lane take for SE3Quat::log the way g2o writes it (rotation matrix, acos, tan) and from the quaternion directly
// (theta = 2 atan2(|v|, w), tan(theta / 2) = |v| / w: one inverse tangent instead of an inverse cosine and a tangent)?  The odometry edge's logarithm is a
// dependent chain in one lane and sits in front of every trial of poseBundleAdjust / stage 1 of localBundleAdjust.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=fast tools/se3_log_probe.hip -o tools/variants/se3_log_probe && tools/variants/se3_log_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__device__ __forceinline__ void q_to_R(const double *q, double *R) {
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0], tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy; R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx; R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
__device__ __forceinline__ void mat3_mul(const double *A, const double *B, double *C) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ void skew(const double *v, double *S) { S[0] = 0; S[1] = -v[2]; S[2] = v[1]; S[3] = v[2]; S[4] = 0; S[5] = -v[0]; S[6] = -v[1]; S[7] = v[0]; S[8] = 0; }
__device__ void log_g2o(const double *pose, double *out) {
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
// the same value from the unit quaternion (w >= 0 after normalisation): cos(theta) = w^2 - |v|^2, sin(theta) = 2 |v| w, tan(theta / 2) = |v| / w, omega = theta v / |v|
__device__ void log_quat(const double *pose, double *out) {
    const double x = pose[0], y = pose[1], z = pose[2], w = pose[3];
    const double n2 = x * x + y * y + z * z, d = w * w - n2;
    double om[3], k;
    if (fabs(d) > 0.99999) {
        om[0] = 2 * w * x; om[1] = 2 * w * y; om[2] = 2 * w * z;
        k = 1. / 12.;
    } else {
        const double n = sqrt(n2), theta = 2 * atan2(n, w), s = theta / n;
        om[0] = s * x; om[1] = s * y; om[2] = s * z;
        k = (1 - 0.5 * theta * w / n) / (theta * theta);
    }
    // V^-1 t = t - 0.5 om x t + k om x (om x t)
    const double t0 = pose[4], t1 = pose[5], t2 = pose[6];
    const double c0 = om[1] * t2 - om[2] * t1, c1 = om[2] * t0 - om[0] * t2, c2 = om[0] * t1 - om[1] * t0;
    const double e0 = om[1] * c2 - om[2] * c1, e1 = om[2] * c0 - om[0] * c2, e2 = om[0] * c1 - om[1] * c0;
    out[0] = om[0]; out[1] = om[1]; out[2] = om[2];
    out[3] = t0 - 0.5 * c0 + k * e0; out[4] = t1 - 0.5 * c1 + k * e1; out[5] = t2 - 0.5 * c2 + k * e2;
}
template <int V>
__global__ void k(const double *in, double *out, long long *cyc, int reps) {
    double p[7], o[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 7; ++i) p[i] = in[7 * threadIdx.x + i];
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        if (V == 0) log_g2o(p, o); else log_quat(p, o);
        p[4] += 1e-9 * o[3]; p[5] += 1e-9 * o[0];            // the next evaluation depends on this one (a chain, as in the solver)
    }
    cyc[threadIdx.x] = clock64() - t0;
    for (int i = 0; i < 6; ++i) out[6 * threadIdx.x + i] = o[i];
}
int main() {
    const int n = 64, reps = 200;
    double h[7 * n], o0[6 * n], o1[6 * n];
    for (int i = 0; i < n; ++i) {
        const double ang = 0.002 + 3.1 * i / n, ax[3] = {0.3, -0.5, 0.8};
        const double an = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        for (int k2 = 0; k2 < 3; ++k2) h[7 * i + k2] = ax[k2] / an * sin(ang / 2);
        h[7 * i + 3] = cos(ang / 2); h[7 * i + 4] = 0.3; h[7 * i + 5] = -0.2; h[7 * i + 6] = 0.1;
    }
    double *di, *dout; long long *dc;
    hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o0)); hipMalloc(&dc, 8 * n);
    hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    long long c0[n], c1[n];
    k<0><<<1, n>>>(di, dout, dc, reps); hipMemcpy(o0, dout, sizeof(o0), hipMemcpyDeviceToHost); hipMemcpy(c0, dc, 8 * n, hipMemcpyDeviceToHost);
    k<1><<<1, n>>>(di, dout, dc, reps); hipMemcpy(o1, dout, sizeof(o1), hipMemcpyDeviceToHost); hipMemcpy(c1, dc, 8 * n, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 6 * n; ++i) worst = fmax(worst, fabs(o0[i] - o1[i]));
    std::printf("SE3 log, one wave (64 different rotations 0.002 ... 3.1 rad), clock ticks per evaluation: g2o form %.0f, quaternion form %.0f; largest difference %.3g\n",
                (double)c0[0] / reps, (double)c1[0] / reps, worst);
    return 0;
}
