// sdwa_probe.hip -- what do SDWA forms cost on gfx950?  (k_resize got SLOWER when 40 shifts per lane were folded into v_mul_u32_u24_sdwa.)
// Each kernel runs N iterations of 8 independent chains of one instruction form per lane, 4 waves per SIMD; prints wave-cycles per instruction.
//   hipcc -O3 --offload-arch=gfx950 tools/sdwa_probe.hip -o tools/sdwa_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAINS 8
#define DEF(NAME, ASM)                                                                                   \
__global__ __launch_bounds__(256) void NAME(uint32_t *out, int n, long long *cyc) {                      \
    uint32_t v[CHAINS], b = threadIdx.x * 2654435761u | 1u;                                              \
    for (int k = 0; k < CHAINS; ++k) v[k] = threadIdx.x + k * 977u;                                      \
    const long long t0 = clock64();                                                                      \
    for (int i = 0; i < n; ++i) {                                                                        \
        _Pragma("unroll") for (int k = 0; k < CHAINS; ++k) asm volatile(ASM : "+v"(v[k]) : "v"(b));     \
    }                                                                                                    \
    const long long t1 = clock64();                                                                      \
    uint32_t s = 0; for (int k = 0; k < CHAINS; ++k) s ^= v[k];                                          \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                                             \
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                           \
}
DEF(k_add,        "v_add_u32_e32 %0, %0, %1")
DEF(k_add_sdwa,   "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1")
DEF(k_add_sdwa_b, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0")
DEF(k_mul24,      "v_mul_u32_u24_e32 %0, %0, %1")
DEF(k_mul24_sdwa, "v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
DEF(k_lshr,       "v_lshrrev_b32_e32 %0, 1, %0")
DEF(k_perm,       "v_perm_b32 %0, %0, %1, %1")
DEF(k_dot2,       "v_dot2_u32_u16 %0, %0, %1, %0")
DEF(k_mad24,      "v_mad_u32_u24 %0, %0, %1, %0")
DEF(k_alignbyte,  "v_alignbyte_b32 %0, %0, %1, %1")
DEF(k_lshl_add,   "v_lshl_add_u32 %0, %0, 3, %1")
DEF(k_add3,       "v_add3_u32 %0, %0, %1, %1")
DEF(k_pk_add,     "v_pk_add_u16 %0, %0, %1")
DEF(k_pk_mad,     "v_pk_mad_u16 %0, %0, %1, %0")
DEF(k_bfe,        "v_bfe_u32 %0, %0, 4, 16")
DEF(k_mul_lo,     "v_mul_lo_u32 %0, %0, %1")
DEF(k_mul_hi24,   "v_mul_hi_u32_u24_e32 %0, %0, %1")
DEF(k_min3,       "v_min3_u32 %0, %0, %1, %1")
DEF(k_and_or,     "v_and_or_b32 %0, %0, %1, %1")
DEF(k_cndmask,    "v_cndmask_b32_e32 %0, %0, %1, vcc")
DEF(k_mov_dpp,    "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEF(k_add_dpp,    "v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEF(k_sad,        "v_sad_u8 %0, %0, %1, %0")
DEF(k_pk_min,     "v_pk_min_u16 %0, %0, %1")
DEF(k_pk_max_i,   "v_pk_max_i16 %0, %0, %1")
DEF(k_pk_sub_sat, "v_pk_sub_u16 %0, %0, %1 clamp")
DEF(k_seq_shift_mul,   "v_dot2_u32_u16 %0, %0, %1, 0\n v_lshrrev_b32_e32 %0, 4, %0\n v_mul_u32_u24_e32 %0, %0, %1")
DEF(k_seq_sdwa_mul,    "v_dot2_u32_u16 %0, %0, %1, 0\n v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
DEF(k_seq_sdwa_mul_s,  "v_dot2_u32_u16 %0, %0, %1, 0\n v_mul_u32_u24_sdwa %0, %0, s4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
DEF(k_seq_mul_s,       "v_dot2_u32_u16 %0, %0, %1, 0\n v_mul_u32_u24_e32 %0, s4, %0")
DEF(k_seq_add_sdwa,    "v_add_u32_e32 %0, %0, %1\n v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1")
DEF(k_seq_add_add,     "v_add_u32_e32 %0, %0, %1\n v_add_u32_e32 %0, %0, %1")
template <class K> void run(const char *name, K kern, uint32_t *d_out, long long *d_cyc) {
    const int n = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<256 * 4, 256>>>(d_out, 100, d_cyc);
    hipEventRecord(e0); kern<<<256 * 4, 256>>>(d_out, n, d_cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
    // 4 blocks of 4 waves per CU = 4 waves per SIMD; wave-instructions per SIMD = 4 waves x n x CHAINS
    std::printf("%-28s %8.3f ms   SIMD cycles per wave-instruction %.2f (clock64 of one wave: %.2f per instruction)\n", name, ms, ms * 1e-3 * 2.4e9 / (4.0 * n * CHAINS), (double)c / ((double)n * CHAINS));
}
int main() {
    uint32_t *d_out; long long *d_cyc; hipMalloc(&d_out, 1024 * 256 * 4); hipMalloc(&d_cyc, 8);
#define R(k) run(#k, k, d_out, d_cyc)
    R(k_add); R(k_add_sdwa); R(k_add_sdwa_b); R(k_mul24); R(k_mul24_sdwa); R(k_lshr); R(k_perm); R(k_dot2); R(k_mad24); R(k_alignbyte); R(k_lshl_add); R(k_add3);
    R(k_pk_add); R(k_pk_mad); R(k_bfe); R(k_mul_lo); R(k_mul_hi24); R(k_min3); R(k_and_or); R(k_cndmask); R(k_mov_dpp); R(k_add_dpp); R(k_sad); R(k_pk_min); R(k_pk_max_i); R(k_pk_sub_sat);
    std::printf("two- and three-instruction sequences (per SEQUENCE):\n");
    R(k_seq_shift_mul); R(k_seq_sdwa_mul); R(k_seq_sdwa_mul_s); R(k_seq_mul_s); R(k_seq_add_sdwa); R(k_seq_add_add);
    return 0;
}
