// Microbenchmark: cost of a dependent kernel boundary on gfx950 -- plain stream launches vs one hipGraph launch,
// for a chain of short kernels like the MPM substep (6 dependent kernels per substep).
// hipcc --offload-arch=gfx950 -O3 dispatch_gap.hip -o dispatch_gap.bin && ./dispatch_gap.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>

__global__ void k_small(float* a, int n, int work) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = a[i];
    for (int k = 0; k < work; ++k) v = v * 1.0001f + 0.5f;
    a[i] = v;
}
int main() {
    const int n = 500000, chain = 600;
    float* a; hipMalloc(&a, n * 4); hipMemset(a, 0, n * 4);
    hipStream_t s; hipStreamCreate(&s);
    for (int work : {0, 2000}) {
        for (int wgs : {64, 1954}) {
            auto launch_chain = [&] { for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(k_small, dim3(wgs), dim3(256), 0, s, a, n, work); };
            launch_chain(); hipStreamSynchronize(s);
            auto t0 = std::chrono::steady_clock::now();
            launch_chain(); hipStreamSynchronize(s);
            double us_plain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / chain;
            hipGraph_t g; hipGraphExec_t ge;
            hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
            launch_chain();
            hipStreamEndCapture(s, &g);
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            hipGraphLaunch(ge, s); hipStreamSynchronize(s);
            t0 = std::chrono::steady_clock::now();
            hipGraphLaunch(ge, s); hipStreamSynchronize(s);
            double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / chain;
            printf("work=%5d wgs=%5d : per kernel %7.2f us (stream)  %7.2f us (hipGraph)\n", work, wgs, us_plain, us_graph);
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
    }
    return 0;
}
