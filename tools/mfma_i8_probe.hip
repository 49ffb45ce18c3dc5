// Checks, on the device, the operand / result maps this tree assumes for v_mfma_i32_16x16x64_i8 and the lane maps of v_permlane16_swap / v_permlane32_swap
// (k_describe's patch blur on the matrix cores is built on them).  hipcc --offload-arch=gfx950 tools/mfma_i8_probe.hip -o /tmp/mfma_i8_probe && /tmp/mfma_i8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int v4i_t __attribute__((ext_vector_type(4)));

__global__ void k_probe(const int8_t *A, const int8_t *B, int *D, int *sw) {     // A [16][64], B [64][16] row-major
    const int l = threadIdx.x, i = l & 15, kg = l >> 4;
    v4i_t a, b, c = {0, 0, 0, 0};
    for (int d = 0; d < 4; ++d) {
        uint32_t wa = 0, wb = 0;
        for (int j = 0; j < 4; ++j) {
            wa |= (uint32_t)(uint8_t)A[i * 64 + 16 * kg + 4 * d + j] << (8 * j);
            wb |= (uint32_t)(uint8_t)B[(16 * kg + 4 * d + j) * 16 + i] << (8 * j);
        }
        a[d] = (int)wa; b[d] = (int)wb;
    }
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
    int x = l, y = 100 + l;
    auto r32 = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    sw[l * 4 + 0] = r32[0]; sw[l * 4 + 1] = r32[1];
    auto r16 = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    sw[l * 4 + 2] = r16[0]; sw[l * 4 + 3] = r16[1];
}

int main() {
    int8_t hA[16 * 64], hB[64 * 16];
    srand(7);
    for (auto &v : hA) v = (int8_t)(rand() % 256 - 128);
    for (auto &v : hB) v = (int8_t)(rand() % 256 - 128);
    int8_t *dA, *dB; int *dD, *dS;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 256 * 4); hipMalloc(&dS, 256 * 4);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, dS);
    int hD[256], hS[256];
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost); hipMemcpy(hS, dS, sizeof hS, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int row = (l >> 4) * 4 + r, col = l & 15;
            int want = 0;
            for (int k = 0; k < 64; ++k) want += (int)hA[row * 64 + k] * (int)hB[k * 16 + col];
            if (hD[l * 4 + r] != want) ++bad;
        }
    printf("mfma_i32_16x16x64_i8: A[l&15][16(l>>4)+j], B[16(l>>4)+j][l&15], D[4(l>>4)+reg][l&15]: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    int bad32 = 0, bad16 = 0;
    for (int l = 0; l < 64; ++l) {
        // permlane32_swap(vdst = x, src = y): lanes 32-63 of x <-> lanes 0-31 of y
        const int w0 = l < 32 ? l : 100 + (l - 32), w1 = l < 32 ? (l + 32) : 100 + l;
        if (hS[l * 4] != w0 || hS[l * 4 + 1] != w1) ++bad32;
        // permlane16_swap: odd rows of x <-> even rows of y
        const int row = l >> 4;
        const int v0 = (row & 1) ? 100 + (l - 16) : l, v1 = (row & 1) ? 100 + l : (l + 16);
        if (hS[l * 4 + 2] != v0 || hS[l * 4 + 3] != v1) ++bad16;
    }
    printf("permlane32_swap: %s, permlane16_swap: %s\n", bad32 ? "WRONG" : "ok", bad16 ? "WRONG" : "ok");
    if (bad32 || bad16) for (int l = 0; l < 64; l += 8) printf("lane %2d: p32 (%d, %d) p16 (%d, %d)\n", l, hS[l * 4], hS[l * 4 + 1], hS[l * 4 + 2], hS[l * 4 + 3]);
    return bad || bad32 || bad16;
}
