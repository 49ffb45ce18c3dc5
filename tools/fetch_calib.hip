// fetch_calib.hip -- what FETCH_SIZE / WRITE_SIZE count for loads of a known size (tools/profile_set.sh runs it under rocprofv3 --pmc).
// Three streaming kernels read the same 1 GiB buffer once with 4-, 8- and 16-byte loads per lane (coalesced); strided kernels read one double per 64, per 128 and
// per 256 bytes of it (128 / 64 / 32 MiB of doubles: a gather that uses a fraction of every line it touches -- the bundle adjuster's point / pose gathers); one reads
// 48-byte row pieces at byte-granular pseudo-random alignment, one piece per 256 bytes (k_describe's window rows).  Each writes 4 bytes per workgroup.
// bytes_known / (counter x 1024) is the correction factor for that pattern on this machine -- "known" = the 64-byte halves / 128-byte lines the pattern touches,
// both are printed, and the per-pattern factors go to pmc_calibration.json (one global factor 2.0 was more than had been measured).
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

template <int STRIDE_DOUBLES>
__global__ __launch_bounds__(256) void k_calib_stride(const double *__restrict__ src, size_t n_items, uint32_t *out) {
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_items; i += (size_t)gridDim.x * 256) acc += src[(size_t)STRIDE_DOUBLES * i];
    if (acc == 1.2345) out[blockIdx.x] = 7;
    if (threadIdx.x == 0) out[blockIdx.x] = 1;
}
// 48 bytes (three 16-byte loads, as unaligned dword groups) at offset 256 i + (hash(i) mod 200): row pieces that straddle 64-byte halves the way k_describe's do
__global__ __launch_bounds__(256) void k_calib_rows48(const uint8_t *__restrict__ src, size_t n_items, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_items; i += (size_t)gridDim.x * 256) {
        const uint32_t h = (uint32_t)(i * 2654435761u) >> 8;
        const uint8_t *p = src + 256 * i + (h % 200u) / 4 * 4;
        for (int k = 0; k < 12; ++k) acc ^= reinterpret_cast<const uint32_t *>(p)[k];
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
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
        hipLaunchKernelGGL(k_calib_stride<16>, dim3(grid), dim3(256), 0, 0, static_cast<const double *>(buf), bytes / 128, out);
        hipLaunchKernelGGL(k_calib_stride<32>, dim3(grid), dim3(256), 0, 0, static_cast<const double *>(buf), bytes / 256, out);
        hipLaunchKernelGGL(k_calib_rows48, dim3(grid), dim3(256), 0, 0, static_cast<const uint8_t *>(buf), bytes / 256 - 1, out);
    }
    for (int rep = 0; rep < 3; ++rep) {                    // stores: the same 1 GiB written once with 4- and 16-byte stores per lane
        hipLaunchKernelGGL(k_calib_write<uint32_t>, dim3(grid), dim3(256), 0, 0, static_cast<uint32_t *>(buf), bytes / 4, 1u);
        hipLaunchKernelGGL(k_calib_write<uint4>, dim3(grid), dim3(256), 0, 0, static_cast<uint4 *>(buf), bytes / 16, make_uint4(1, 2, 3, 4));
    }
    CK(hipDeviceSynchronize());
    std::printf("known bytes per launch: stream 4 / 8 / 16 B per lane = %zu each; 8 B per 64 B: %zu useful, %zu in touched 64-byte halves, %zu in touched 128-byte lines; "
                "8 B per 128 B: %zu useful, %zu in halves, %zu in lines; 8 B per 256 B: %zu useful, %zu in halves, %zu in lines; 48-byte rows (one per 256 B): %zu useful\n",
                bytes, bytes / 8, bytes, bytes, bytes / 16, bytes / 2, bytes, bytes / 32, bytes / 4, bytes / 2, (bytes / 256 - 1) * 48);
    return 0;
}
