// Issue-rate microbenchmark for the packed / dot / permute VALU ops the front-end kernels lean on (gfx950).
// Each wave runs N independent chains of one instruction; 8 waves per SIMD, every CU busy.  Prints cycles per
// wave-instruction per SIMD (2.0 = full rate for a wave64 on the 32-lane SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int kIters = 4096, kChains = 8;
template <int OP> __global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
    uint32_t v[kChains];
    unsigned long long w[kChains];
    for (int i = 0; i < kChains; ++i) { v[i] = seed + threadIdx.x * 7 + i; w[i] = 0x3ff0000000000000ull + v[i]; }
    uint32_t b = seed ^ 0x12345u, c = seed + 3;
    unsigned long long bw = 0x3ff0000000001234ull + seed, cw = 0x3fe0000000001234ull + seed;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < kChains; ++i) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 3) asm volatile("v_or_b32 %0, %0, %1" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 4) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(v[i]));
            if (OP == 5) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(v[i]));
            if (OP == 6) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 7) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 9) asm volatile("v_min_u32 %0, %0, %1" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 10) asm volatile("v_max_i32 %0, %0, %1" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 11) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (OP == 12) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (OP == 13) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (OP == 14) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 15) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 16) asm volatile("v_bfe_u32 %0, %0, 4, 8" : "+v"(v[i]));
            if (OP == 17) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (OP == 18) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (OP == 19) asm volatile("v_alignbit_b32 %0, %0, %1, 8" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 20) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 21) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (OP == 22) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 23) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (OP == 25) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 26) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (OP == 27) asm volatile("v_sad_u16 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (OP == 28) asm volatile("v_msad_u8 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (OP == 29) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 30) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 31) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (OP == 32) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 33) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 34) asm volatile("v_pk_lshrrev_b16 %0, 15, %0 op_sel_hi:[0,1]" : "+v"(v[i]));
            if (OP == 35) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 36) asm volatile("v_dot2_u32_u16 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (OP == 37) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (OP == 38) asm volatile("v_dot4_i32_i8 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (OP == 39) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 40) asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 41) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 42) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(v[i]));
            if (OP == 43) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (OP == 44) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 45) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[i]) : "v"(bw), "v"(cw));
            if (OP == 46) asm volatile("v_add_f64 %0, %0, %1" : "+v"(w[i]) : "v"(bw));
            if (OP == 47) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(w[i]) : "v"(bw), "v"(cw));
            if (OP == 48) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
            if (OP == 49) asm volatile("v_sqrt_f32 %0, %0" : "+v"(v[i]));
            if (OP == 50) asm volatile("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(v[i]) : "v"(b) : "vcc");
            if (OP == 51) asm volatile("v_sub_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_2" : "+v"(v[i]) : "v"(b) : "vcc");
        }
    }
    uint32_t r = c;
    for (int i = 0; i < kChains; ++i) r ^= v[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
    if (r == 0xdeadbeef) out[threadIdx.x] = r;
}
template <int OP> int run(const char *name, uint32_t *d_out, int cus) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = cus * 8;                 // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 2u);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double insts_per_simd = 8.0 * kIters * kChains;      // wave-instructions issued by one SIMD
    printf("%-20s %7.3f ms  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
    return 0;
}
int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("%s, %d CUs, clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    uint32_t *d; CHECK(hipMalloc(&d, 4096));
    const int cus = p.multiProcessorCount;
    run<0>("v_add_u32", d, cus);
    run<1>("v_xor_b32", d, cus);
    run<2>("v_and_b32", d, cus);
    run<3>("v_or_b32", d, cus);
    run<4>("v_lshlrev_b32", d, cus);
    run<5>("v_lshrrev_b32", d, cus);
    run<6>("v_sub_u32", d, cus);
    run<7>("v_mov_b32", d, cus);
    run<8>("v_cndmask_b32", d, cus);
    run<9>("v_min_u32", d, cus);
    run<10>("v_max_i32", d, cus);
    run<11>("v_and_or_b32", d, cus);
    run<12>("v_or3_b32", d, cus);
    run<13>("v_add3_u32", d, cus);
    run<14>("v_lshl_or_b32", d, cus);
    run<15>("v_lshl_add_u32", d, cus);
    run<16>("v_bfe_u32", d, cus);
    run<17>("v_min3_u32", d, cus);
    run<18>("v_med3_i32", d, cus);
    run<19>("v_alignbit_b32", d, cus);
    run<20>("v_alignbyte_b32", d, cus);
    run<21>("v_perm_b32", d, cus);
    run<22>("v_bcnt_u32_b32", d, cus);
    run<23>("v_mul_u32_u24", d, cus);
    run<24>("v_mad_u32_u24", d, cus);
    run<25>("v_mul_lo_u32", d, cus);
    run<26>("v_sad_u8", d, cus);
    run<27>("v_sad_u16", d, cus);
    run<28>("v_msad_u8", d, cus);
    run<29>("v_pk_add_u16", d, cus);
    run<30>("v_pk_sub_i16", d, cus);
    run<31>("v_pk_mad_u16", d, cus);
    run<32>("v_pk_min_i16", d, cus);
    run<33>("v_pk_max_u16", d, cus);
    run<34>("v_pk_lshrrev_b16", d, cus);
    run<35>("v_pk_mul_lo_u16", d, cus);
    run<36>("v_dot2_u32_u16", d, cus);
    run<37>("v_dot4_u32_u8", d, cus);
    run<38>("v_dot4_i32_i8", d, cus);
    run<39>("v_mov_dpp wave_shr", d, cus);
    run<40>("v_add_dpp row_shr", d, cus);
    run<41>("v_cmp_lt_u32", d, cus);
    run<42>("v_cvt_f32_u32", d, cus);
    run<43>("v_fma_f32", d, cus);
    run<44>("v_mul_f32", d, cus);
    run<45>("v_pk_fma_f32", d, cus);
    run<46>("v_add_f64", d, cus);
    run<47>("v_fma_f64", d, cus);
    run<48>("v_rcp_f32", d, cus);
    run<49>("v_sqrt_f32", d, cus);
    run<50>("v_and_b32 sdwa", d, cus);
    run<51>("v_sub_u16 sdwa", d, cus);
    return 0;
}
