// cu_hog.hip -- a kernel with the residency footprint of one k_ba_lm workgroup (512 threads, 256 VGPRs per lane, 150 KB of LDS: nothing else fits the CU beside it)
// that does NOTHING: no memory traffic, no fences, no barriers -- it sleeps for a given time.  tools/hog_probe.py runs the front end of one sequence beside N x 32
// of them and beside N real local-BA teams: if the front end's small kernels slow down alike, the cause is the CUs the teams take away, not what they do to the caches.
//   hipcc -O2 --offload-arch=gfx950 -shared -fPIC tools/cu_hog.hip -o tools/variants/libcuhog.so
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(512) void k_cu_hog(long long ticks, double *sink) {
    extern __shared__ double lds[];
    double r[120];
#pragma unroll
    for (int i = 0; i < 120; ++i) r[i] = threadIdx.x * 1e-3 + i;
    if (ticks < 0) {                                       // silent: no clock read either -- -ticks rounds of s_sleep 127 (127 x 64 cycles = 3.4 us each)
        for (long long k = 0; k < -ticks; ++k) {
            __builtin_amdgcn_s_sleep(127);
#pragma unroll
            for (int i = 0; i < 120; ++i) asm volatile("" : "+v"(r[i]));
        }
    } else {
        const long long t0 = wall_clock64();               // 100 MHz
        while (wall_clock64() - t0 < ticks) {
            __builtin_amdgcn_s_sleep(32);
#pragma unroll
            for (int i = 0; i < 120; ++i) asm volatile("" : "+v"(r[i]));      // all 240 registers stay live across the loop
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 120; ++i) s += r[i];
    if (s == 12345.678) { lds[threadIdx.x] = s; sink[0] = lds[0]; }
}
extern "C" int hog_launch(int workgroups, int usec, void *stream, double *sink) {
    static bool once = false;
    if (!once) { if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_cu_hog), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return -1; once = true; }
    hipLaunchKernelGGL(k_cu_hog, dim3(workgroups), dim3(512), 150 * 1024, (hipStream_t)stream, usec < 0 ? (long long)usec * 10 / 34 : (long long)usec * 100, sink);      // usec < 0: the silent loop, ~|usec| long
    return (int)hipGetLastError();
}
