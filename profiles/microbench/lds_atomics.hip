// Microbenchmark: cost of LDS / global atomic flavours on gfx950, to pick the scatter strategy of k_p2g.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomics.hip -o lds_atomics && ./lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int ITER = 2048;
// MODE: 0 f32 add, 1 u32 add, 2 u64 add, 3 plain rmw (ds_read+ds_write), 4 f64 add
// PATTERN: lanes active = ACT (of 64), address = (lane*stride + it*7) % 1024
template <int MODE, int ACT, int STRIDE>
__global__ __launch_bounds__(256) void k_lds(float* out) {
    __shared__ unsigned long long buf[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) buf[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    if (lane < ACT) {
        for (int it = 0; it < ITER; ++it) {
            // STRIDE >= 100: (STRIDE-100)-way same-address conflicts: lanes share an address in groups
            int a = STRIDE >= 100 ? ((lane / (STRIDE - 100)) * 5 + it * 7) & 1023 : (lane * STRIDE + it * 7) & 1023;
            if (MODE == 0) atomicAdd(reinterpret_cast<float*>(&buf[a]), 1.0f);
            else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned int*>(&buf[a]), 1u);
            else if (MODE == 2) atomicAdd(&buf[a], 1ull);
            else if (MODE == 4) atomicAdd(reinterpret_cast<double*>(&buf[a]), 1.0);
            else { volatile float* p = reinterpret_cast<volatile float*>(&buf[a]); *p = *p + 1.0f; }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)buf[3];
}
template <int MODE>
__global__ __launch_bounds__(256) void k_glob(float* g, unsigned long long* gi, int n) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    for (int it = 0; it < 64; ++it) {
        int a = (int)(((unsigned)t * 2654435761u + it * 40503u) % (unsigned)n);
        if (MODE == 0) atomicAdd(&g[a], 1.0f);
        else atomicAdd(&gi[a], 1ull);
    }
}
template <class F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
#define RUN(MODE, ACT, STRIDE, name)                                                                    \
    { float ms = timeit([&] { hipLaunchKernelGGL((k_lds<MODE, ACT, STRIDE>), dim3(1024), dim3(256), 0, 0, out); }); \
      double per = ms * 1e-3 * 2.4e9 / (ITER * (4.0 * 1024 / 256 / 1.0)); /* cycles per wave-instr per CU, 4 waves/blk, 4 blk/CU rounds */ \
      printf("%-28s act=%2d stride=%2d : %8.3f ms  -> %.1f ns per wave-instruction per CU-slot\n", name, ACT, STRIDE, ms, ms * 1e6 / (ITER * 16.0)); (void)per; }
int main() {
    float* out; hipMalloc(&out, 1 << 20);
    RUN(0, 64, 1, "lds f32 add distinct");
    RUN(0, 64, 0, "lds f32 add same addr");
    RUN(0, 12, 5, "lds f32 add 12 lanes");
    RUN(0, 1, 1, "lds f32 add 1 lane");
    RUN(1, 64, 1, "lds u32 add distinct");
    RUN(1, 64, 0, "lds u32 add same addr");
    RUN(1, 12, 5, "lds u32 add 12 lanes");
    RUN(2, 64, 1, "lds u64 add distinct");
    RUN(2, 64, 0, "lds u64 add same addr");
    RUN(2, 12, 5, "lds u64 add 12 lanes");
    RUN(4, 64, 1, "lds f64 add distinct");
    RUN(4, 12, 5, "lds f64 add 12 lanes");
    RUN(4, 64, 108, "lds f64 add 8-way groups");
    RUN(4, 64, 104, "lds f64 add 4-way groups");
    RUN(4, 64, 116, "lds f64 add 16-way groups");
    RUN(2, 64, 108, "lds u64 add 8-way groups");
    RUN(0, 64, 108, "lds f32 add 8-way groups");
    RUN(3, 64, 1, "lds plain rmw distinct");
    RUN(3, 12, 5, "lds plain rmw 12 lanes");
    int n = 1 << 20; float* g; unsigned long long* gi; hipMalloc(&g, n * 4); hipMalloc(&gi, n * 8);
    hipMemset(g, 0, n * 4); hipMemset(gi, 0, n * 8);
    for (int nn : {1 << 20, 1 << 14, 256}) {
        float a = timeit([&] { hipLaunchKernelGGL((k_glob<0>), dim3(4096), dim3(256), 0, 0, g, gi, nn); });
        float b = timeit([&] { hipLaunchKernelGGL((k_glob<1>), dim3(4096), dim3(256), 0, 0, g, gi, nn); });
        printf("global atomics over %8d addresses: f32 %.3f ms (%.1f G/s)   u64 %.3f ms (%.1f G/s)\n", nn, a, 4096.0 * 256 * 64 / a * 1e-6, b, 4096.0 * 256 * 64 / b * 1e-6);
    }
    return 0;
}
