// launch_rate_probe.hip -- how many kernel launches per second does ONE process issue from T host threads (a stream each)?  C5 on one GPU is 8 threads x ~20 runtime
// calls per frame; two processes of four sequences reach 1.6 x the frames of one process of eight.
//   hipcc -O2 --offload-arch=gfx950 tools/launch_rate_probe.hip -o tools/launch_rate_probe -lpthread ; tools/launch_rate_probe
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void k_nop(int *p) { if (p && threadIdx.x == 1024) *p = 1; }
int main() {
    for (int T : {1, 2, 4, 8, 16}) {
        std::atomic<long long> total{0};
        std::atomic<int> go{0};
        std::vector<std::thread> th;
        const double secs = 0.5;
        for (int t = 0; t < T; ++t) th.emplace_back([&]() {
            hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, (int *)nullptr);
            (void)hipStreamSynchronize(st);
            while (!go.load()) std::this_thread::yield();
            const auto t0 = std::chrono::steady_clock::now();
            long long n = 0;
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
                for (int i = 0; i < 16; ++i) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, (int *)nullptr);
                n += 16;
                if ((n & 255) == 0) (void)hipStreamSynchronize(st);            // (bounded queue depth, like a frame's sync)
            }
            (void)hipStreamSynchronize(st);
            total += n;
            (void)hipStreamDestroy(st);
        });
        go = 1;
        for (auto &x : th) x.join();
        std::printf("%2d threads: %8.0f launches/s in total, %7.0f per thread (%.2f us of host time per launch and thread)\n", T, total / secs, total / secs / T, secs * 1e6 * T / total);
    }
    // the same with the 13 launches of a frame's extract captured into a graph: nodes per second, T threads replaying their own executable graph
    for (int T : {1, 8, 16}) {
        std::atomic<long long> total{0};
        std::atomic<int> go{0};
        std::vector<std::thread> th;
        const double secs = 0.5;
        for (int t = 0; t < T; ++t) th.emplace_back([&]() {
            hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            hipGraph_t g; hipGraphExec_t ge;
            (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            for (int i = 0; i < 13; ++i) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, (int *)nullptr);
            if (hipStreamEndCapture(st, &g) != hipSuccess || hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { std::printf("graph capture failed\n"); return; }
            for (int i = 0; i < 20; ++i) (void)hipGraphLaunch(ge, st);
            (void)hipStreamSynchronize(st);
            while (!go.load()) std::this_thread::yield();
            const auto t0 = std::chrono::steady_clock::now();
            long long n = 0;
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
                (void)hipGraphLaunch(ge, st); n += 13;
                if ((n % (13 * 16)) == 0) (void)hipStreamSynchronize(st);
            }
            (void)hipStreamSynchronize(st);
            total += n;
            (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipStreamDestroy(st);
        });
        go = 1;
        for (auto &x : th) x.join();
        std::printf("%2d threads, 13-node graphs: %8.0f kernel nodes/s in total (%.0f graph launches/s)\n", T, total / secs, total / secs / 13);
    }
    return 0;
}
