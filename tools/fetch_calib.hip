// fetch_calib.hip -- what FETCH_SIZE / WRITE_SIZE count for loads of a known size (tools/profile_set.sh runs it under rocprofv3 --pmc).
// Three streaming kernels read the same 1 GiB buffer once with 4-, 8- and 16-byte loads per lane (coalesced) and one reads 256 MiB of it with 8-byte
// loads at a 64-byte stride per lane (one double per cache-line half: the scattered pattern of the bundle adjuster's gathers); each writes 4 bytes per
// workgroup.  bytes_known / (counter x 1024) is the correction factor for that access width on this machine.
//   hipcc --offload-arch=gfx950 -O2 tools/fetch_calib.hip -o tools/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <typename T>
__global__ __launch_bounds__(256) void k_calib_stream(const T *__restrict__ src, size_t n, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const T v = src[i];
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc ^= w[k];
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;      // (never: keeps the loads alive)
    if (threadIdx.x == 0) out[blockIdx.x] = 1;
}
__global__ __launch_bounds__(256) void k_calib_strided8(const double *__restrict__ src, size_t n_lines, uint32_t *out) {
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * 256) acc += src[8 * i];     // one double per 64 bytes
    if (acc == 1.2345) out[blockIdx.x] = 7;
    if (threadIdx.x == 0) out[blockIdx.x] = 1;
}

template <typename T>
__global__ __launch_bounds__(256) void k_calib_write(T *__restrict__ dst, size_t n, T v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    void *buf = nullptr; uint32_t *out = nullptr;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4 * 65536));
    CK(hipMemset(buf, 1, bytes));
    const int grid = 8192;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_calib_stream<uint32_t>, dim3(grid), dim3(256), 0, 0, static_cast<const uint32_t *>(buf), bytes / 4, out);
        hipLaunchKernelGGL(k_calib_stream<uint2>, dim3(grid), dim3(256), 0, 0, static_cast<const uint2 *>(buf), bytes / 8, out);
        hipLaunchKernelGGL(k_calib_stream<uint4>, dim3(grid), dim3(256), 0, 0, static_cast<const uint4 *>(buf), bytes / 16, out);
        hipLaunchKernelGGL(k_calib_strided8, dim3(grid), dim3(256), 0, 0, static_cast<const double *>(buf), bytes / 64, out);
    }
    for (int rep = 0; rep < 3; ++rep) {                    // stores: the same 1 GiB written once with 4- and 16-byte stores per lane
        hipLaunchKernelGGL(k_calib_write<uint32_t>, dim3(grid), dim3(256), 0, 0, static_cast<uint32_t *>(buf), bytes / 4, 1u);
        hipLaunchKernelGGL(k_calib_write<uint4>, dim3(grid), dim3(256), 0, 0, static_cast<uint4 *>(buf), bytes / 16, make_uint4(1, 2, 3, 4));
    }
    CK(hipDeviceSynchronize());
    std::printf("known bytes per launch: stream 4 / 8 / 16 B per lane = %zu each; strided 8 B = %zu useful, %zu in touched 64-byte halves, %zu in touched 128-byte lines\n",
                bytes, bytes / 8, bytes, bytes);
    return 0;
}
