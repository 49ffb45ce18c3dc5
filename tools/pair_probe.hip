// pair_probe.hip -- what bounds the block products of the fused Schur pass (schur_fused in ba.hip) at two waves per SIMD?
// One 512-thread workgroup per CU (150 KB of dynamic LDS keeps a second one out, like k_ba_lm), every lane walks N "pairs": read Z_a and Z_b from the wave's LDS slab,
// acc[6][6] += Z_a Z_b^T.  Variants:
//   0  as in ba.hip round 3: 18 ds_read_b128, wait, 108 v_fma_f64
//   1  the same with the NEXT pair's operands requested before the products of the current one (72 more registers)
//   2  rank-2 form: Z = Jp^T G (Jp 2x6 with two structural zeros, G 2x3): 14 doubles per observation, M = G_a G_b^T (12), T = M Jp_b (20), acc += Jp_a^T T (60)
//   3  rank-2 form with the next pair's operands in flight
//   4  no LDS reads at all (operands constant in registers): the FMA pipe alone
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=fast tools/pair_probe.hip -o tools/pair_probe ; run: tools/pair_probe [threads per workgroup = 512]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define MS_LDS __attribute__((address_space(3)))
typedef double d2_t __attribute__((ext_vector_type(2)));

template <int V>
__global__ __launch_bounds__(512) void k_pairs(double *out, int n_pairs, long long *cycles) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    MS_LDS double *stage = (MS_LDS double *)lds + wave * (64 * 18);
    for (int i = lane; i < 64 * 18; i += 64) stage[i] = 1e-3 * (i % 97) + 0.5;
    __syncthreads();
    double acc[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) acc[q] = 0;
    // the lane's block (a, b) as in a batch of 6 points x 10 poses: pair g reads entries g * 10 + a and g * 10 + b
    int a = 0, t = lane % 55;
    while (t > a) { t -= a + 1; ++a; }
    const int b = t;
    const long long t0 = clock64();
    if (V == 0 || V == 1) {
        double A[18], B[18], An[18], Bn[18];
        auto rd = [&](int g, double (&X)[18], double (&Y)[18]) {
            const MS_LDS d2_t *za = (const MS_LDS d2_t *)(stage + ((g % 6) * 10 + a) * 18), *zb = (const MS_LDS d2_t *)(stage + ((g % 6) * 10 + b) * 18);
#pragma unroll
            for (int q = 0; q < 9; ++q) { const d2_t u = za[q], v = zb[q]; X[2 * q] = u.x; X[2 * q + 1] = u.y; Y[2 * q] = v.x; Y[2 * q + 1] = v.y; }
        };
        if (V == 1) rd(0, A, B);
        for (int g = 0; g < n_pairs; ++g) {
            if (V == 0) rd(g, A, B);
            else rd(g + 1, An, Bn);
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[6 * i + j] += A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
            if (V == 1) {
#pragma unroll
                for (int q = 0; q < 18; ++q) { A[q] = An[q]; B[q] = Bn[q]; }
            }
        }
    } else if (V == 2 || V == 3) {
        // slab entry: G[2][3] (6), then p = {uv, 1 + u^2, v, iz, u rz, 1 + v^2, u, v rz} (8): Jp row 0 = (p0, -p1, p2, p3, 0, p4), row 1 = (p5, -p0, -p6, 0, p3, p7)
        double A[14], B[14], An[14], Bn[14];
        auto rd = [&](int g, double (&X)[14], double (&Y)[14]) {
            const MS_LDS d2_t *za = (const MS_LDS d2_t *)(stage + ((g % 6) * 10 + a) * 14), *zb = (const MS_LDS d2_t *)(stage + ((g % 6) * 10 + b) * 14);
#pragma unroll
            for (int q = 0; q < 7; ++q) { const d2_t u = za[q], v = zb[q]; X[2 * q] = u.x; X[2 * q + 1] = u.y; Y[2 * q] = v.x; Y[2 * q + 1] = v.y; }
        };
        if (V == 3) rd(0, A, B);
        for (int g = 0; g < n_pairs; ++g) {
            if (V == 2) rd(g, A, B);
            else rd(g + 1, An, Bn);
            double M[4], T0[6], T1[6];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) M[2 * r + c] = A[3 * r] * B[3 * c] + A[3 * r + 1] * B[3 * c + 1] + A[3 * r + 2] * B[3 * c + 2];
            const double *pb = B + 6, *pa = A + 6;
            // T = M Jp_b (2 x 6)
            T0[0] = M[0] * pb[0] + M[1] * pb[5]; T0[1] = -(M[0] * pb[1] + M[1] * pb[0]); T0[2] = M[0] * pb[2] - M[1] * pb[6]; T0[3] = M[0] * pb[3]; T0[4] = M[1] * pb[3]; T0[5] = M[0] * pb[4] + M[1] * pb[7];
            T1[0] = M[2] * pb[0] + M[3] * pb[5]; T1[1] = -(M[2] * pb[1] + M[3] * pb[0]); T1[2] = M[2] * pb[2] - M[3] * pb[6]; T1[3] = M[2] * pb[3]; T1[4] = M[3] * pb[3]; T1[5] = M[2] * pb[4] + M[3] * pb[7];
            // acc += Jp_a^T T
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                acc[j] = fma(pa[0], T0[j], fma(pa[5], T1[j], acc[j]));
                acc[6 + j] = fma(-pa[1], T0[j], fma(-pa[0], T1[j], acc[6 + j]));
                acc[12 + j] = fma(pa[2], T0[j], fma(-pa[6], T1[j], acc[12 + j]));
                acc[18 + j] = fma(pa[3], T0[j], acc[18 + j]);
                acc[24 + j] = fma(pa[3], T1[j], acc[24 + j]);
                acc[30 + j] = fma(pa[4], T0[j], fma(pa[7], T1[j], acc[30 + j]));
            }
            if (V == 3) {
#pragma unroll
                for (int q = 0; q < 14; ++q) { A[q] = An[q]; B[q] = Bn[q]; }
            }
        }
    } else {
        double A[18], B[18];
#pragma unroll
        for (int q = 0; q < 18; ++q) { A[q] = stage[lane * 18 + q]; B[q] = stage[((lane + 7) & 63) * 18 + q]; }
        for (int g = 0; g < n_pairs; ++g) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[6 * i + j] += A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
            asm volatile("" : "+v"(A[0]), "+v"(B[0]));
        }
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int q = 0; q < 36; ++q) s += acc[q];
    out[(size_t)blockIdx.x * blockDim.x + tid] = s;
    if (lane == 0 && blockIdx.x == 0) cycles[wave] = t1 - t0;
}

template <int V>
static void run(const char *name, int threads, int n_pairs, double *d_out, long long *d_cyc) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_pairs<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_pairs<V><<<256, threads, 150 * 1024>>>(d_out, n_pairs, d_cyc);
    hipEventRecord(e0);
    k_pairs<V><<<256, threads, 150 * 1024>>>(d_out, n_pairs, d_cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    long long cyc[8]; hipMemcpy(cyc, d_cyc, sizeof(cyc), hipMemcpyDeviceToHost);
    const int waves = threads / 64;
    std::printf("%-44s %2d waves per CU: %7.3f ms, cycles per pair-step per wave:", name, waves, ms);
    for (int w = 0; w < waves; ++w) std::printf(" %5.0f", (double)cyc[w] / n_pairs);
    std::printf("\n");
}

int main(int argc, char **argv) {
    const int n_pairs = 20000;
    double *d_out; long long *d_cyc;
    hipMalloc(&d_out, 256 * 512 * sizeof(double)); hipMalloc(&d_cyc, 8 * sizeof(long long));
    for (int threads : {256, 512}) {
        run<4>("FMA pipe alone (108 v_fma_f64, no LDS)", threads, n_pairs, d_out, d_cyc);
        run<0>("18 ds_read_b128 + 108 fma (round 3)", threads, n_pairs, d_out, d_cyc);
        run<1>("  ... next pair's operands in flight", threads, n_pairs, d_out, d_cyc);
        run<2>("rank-2: 14 ds_read_b128 + 92 fma", threads, n_pairs, d_out, d_cyc);
        run<3>("  ... next pair's operands in flight", threads, n_pairs, d_out, d_cyc);
    }
    return 0;
}
