// cumask_probe.hip -- where do the workgroups of a kernel land when its stream was created with a CU mask (hipExtStreamCreateWithCUMask)?
// Every workgroup records XCC_ID and HW_ID; the host prints, per mask, how many distinct (XCD, SE, SH, CU) places were used and how they spread over the XCDs.
//   hipcc -O2 --offload-arch=gfx950 tools/cumask_probe.hip -o tools/variants/cumask_probe && tools/variants/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <set>
#include <vector>
__global__ void k_where(uint32_t *out, int spin) {
    if (threadIdx.x == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);     // stay a little so that the grid spreads over every place it may use
}
static std::set<uint32_t> run(const char *name, const std::vector<uint32_t> &mask) {
    hipStream_t s;
    hipError_t e = mask.empty() ? hipStreamCreate(&s) : hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { std::printf("%-34s stream creation failed: %s\n", name, hipGetErrorString(e)); return {}; }
    const int n = 4096;
    uint32_t *d; hipMalloc(&d, 8 * n);
    hipLaunchKernelGGL(k_where, dim3(n), dim3(64), 0, s, d, 2000);
    hipStreamSynchronize(s);
    std::vector<uint32_t> h(2 * n); hipMemcpy(h.data(), d, 8 * n, hipMemcpyDeviceToHost);
    std::set<uint32_t> places; std::map<int, std::set<uint32_t>> per_xcc;
    for (int i = 0; i < n; ++i) {
        const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
        const uint32_t cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;      // gfx9 HW_ID layout
        const uint32_t place = xcc << 16 | se << 8 | sh << 4 | cu;
        places.insert(place); per_xcc[(int)xcc].insert(place);
    }
    std::printf("%-34s %3zu distinct CUs used;", name, places.size());
    for (auto &kv : per_xcc) std::printf(" xcd%d:%zu", kv.first, kv.second.size());
    std::printf("\n");
    hipFree(d); hipStreamDestroy(s);
    return places;
}
static size_t common(const std::set<uint32_t> &a, const std::set<uint32_t> &b) { size_t n = 0; for (uint32_t x : a) n += b.count(x); return n; }
int main() {
    run("no mask", {});
    auto lo = run("bits 0..127", {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0});
    auto hi = run("bits 128..255", {0, 0, 0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu});
    std::printf("   CUs in both: %zu\n", common(lo, hi));
    auto q0 = run("bits 0..63", {0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0, 0, 0}), q1 = run("bits 64..127", {0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0});
    auto q2 = run("bits 128..191", {0, 0, 0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0}), q3 = run("bits 192..255", {0, 0, 0, 0, 0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu});
    std::printf("   quarters in common: 0&1 %zu  0&2 %zu  0&3 %zu  1&2 %zu  1&3 %zu  2&3 %zu\n", common(q0, q1), common(q0, q2), common(q0, q3), common(q1, q2), common(q1, q3), common(q2, q3));
    auto ev = run("even bits", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u});
    auto od = run("odd bits", {0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu});
    std::printf("   even & odd in common: %zu\n", common(ev, od));
    for (uint32_t x : q0) std::printf(" %05x", x);
    std::printf("\n");
    for (uint32_t x : q2) std::printf(" %05x", x);
    std::printf("\n");
    return 0;
}
