// Microbenchmark: global float atomic throughput vs address pattern within a wave instruction (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
// PAT 0: lane -> base + lane (64 consecutive floats)   1: base + 4*lane (AoS component)   2: random
// each wave owns a 'tile' of 4096 floats somewhere in a 64 MB array; tiles of different waves overlap with prob ~0
template <int PAT>
__global__ __launch_bounds__(256) void k(float* g, size_t n, int iters) {
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    size_t base = ((size_t)wave * 2654435761u) % (n - 65536);
    base &= ~(size_t)63;
    for (int it = 0; it < iters; ++it) {
        size_t a;
        if (PAT == 0) a = base + (size_t)it * 64 + lane;
        else if (PAT == 1) a = base + (size_t)it * 256 + 4 * lane;
        else a = ((size_t)(wave * 64 + lane) * 2654435761u + (size_t)it * 40503u) % n;
        atomicAdd(&g[a], 1.0f);
    }
}
template <class F> float timeit(F f) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    size_t n = 16u << 20; float* g; (void)hipMalloc(&g, n * 4); (void)hipMemset(g, 0, n * 4);
    const int blocks = 2048, iters = 128;
    double ops = (double)blocks * 256 * iters;
    float a = timeit([&] { hipLaunchKernelGGL((k<0>), dim3(blocks), dim3(256), 0, 0, g, n, iters); });
    float b = timeit([&] { hipLaunchKernelGGL((k<1>), dim3(blocks), dim3(256), 0, 0, g, n, iters); });
    float c = timeit([&] { hipLaunchKernelGGL((k<2>), dim3(blocks), dim3(256), 0, 0, g, n, iters); });
    printf("consecutive floats : %.3f ms  %.1f G atomics/s\n", a, ops / a * 1e-6);
    printf("stride-4 (AoS comp): %.3f ms  %.1f G atomics/s\n", b, ops / b * 1e-6);
    printf("random             : %.3f ms  %.1f G atomics/s\n", c, ops / c * 1e-6);
    return 0;
}
