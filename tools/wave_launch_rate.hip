// How fast can the chip start (and retire) waves?  Empty kernels of 1 M waves in workgroups of 1, 4 and 8 waves, with and without a
// static LDS allocation.  Build: hipcc --offload-arch=gfx950 -O3 tools/wave_launch_rate.hip -o tools/wave_launch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS> __global__ void k_empty(int *p) {
    __shared__ int s[LDS > 0 ? LDS : 1];
    if (LDS > 0 && threadIdx.x == 0) s[blockIdx.x % LDS] = 1;
    if (p && threadIdx.x == 12345) p[0] = LDS > 0 ? s[0] : 1;
}
struct Big { int v[256]; };      // 1 KB of kernel arguments, read with a block-dependent index (like k_fast's per-level geometry)
__global__ void k_args(Big b, int *p) {
    const int x = b.v[blockIdx.x & 255], y = b.v[(x + blockIdx.x) & 255];
    if (p && y == 12345) p[0] = 1;
}
template <int LDS> static void run(int threads, long waves) {
    const int wpb = threads / 64;
    const unsigned blocks = (unsigned)(waves / wpb);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_empty<LDS>, dim3(blocks), dim3(threads), 0, 0, (int *)nullptr);
    hipEventRecord(a);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_empty<LDS>, dim3(blocks), dim3(threads), 0, 0, (int *)nullptr);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
    printf("%4d threads/workgroup, %6d B LDS, %ld waves: %.3f ms = %.2f waves/ns\n", threads, LDS * 4, waves, ms, waves / (ms * 1e6));
}
int main() {
    const long W = 1 << 20;
    for (int t : {64, 256, 512, 1024}) run<0>(t, W);
    for (int t : {256, 512}) run<4352>(t, W);      // 17 KB
    for (int t : {256, 512}) run<8704>(t, W);      // 34 KB
    {
        Big b{}; for (int i = 0; i < 256; ++i) b.v[i] = i * 7;
        const unsigned blocks = (unsigned)(W / 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_args, dim3(blocks), dim3(512), 0, 0, b, (int *)nullptr);
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_args, dim3(blocks), dim3(512), 0, 0, b, (int *)nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf(" 512 threads/workgroup, 1 KB of kernel arguments with two dependent scalar reads, %ld waves: %.3f ms = %.2f waves/ns\n", W, ms, W / (ms * 1e6));
    }
    return 0;
}
