// Microbenchmark: streaming NARR SoA float arrays (one value per particle per array) -- the particle-frame
// access pattern of the MPM kernels -- with 4 B/lane vs 16 B/lane accesses and different workgroup counts.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int NARR = 24;
// one particle per thread, dword accesses
__global__ __launch_bounds__(256) void k_dword(const float* __restrict__ in, float* __restrict__ out, int n, size_t stride) {
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    float v[NARR];
#pragma unroll
    for (int a = 0; a < NARR; ++a) v[a] = in[a * stride + p];
#pragma unroll
    for (int a = 0; a < NARR; ++a) out[a * stride + p] = v[a] * 1.0001f;
}
// four particles per thread, dwordx4 accesses
__global__ __launch_bounds__(256) void k_x4(const float4* __restrict__ in, float4* __restrict__ out, int n4, size_t stride4) {
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n4) return;
    float4 v[NARR];
#pragma unroll
    for (int a = 0; a < NARR; ++a) v[a] = in[a * stride4 + p];
#pragma unroll
    for (int a = 0; a < NARR; ++a) { float4 t = v[a]; t.x *= 1.0001f; out[a * stride4 + p] = t; }
}
// read-only / write-only variants (dword)
__global__ __launch_bounds__(256) void k_rd(const float* __restrict__ in, float* __restrict__ out, int n, size_t stride) {
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    float s = 0;
#pragma unroll
    for (int a = 0; a < NARR; ++a) s += in[a * stride + p];
    if (s == -1.f) out[p] = s;
}
__global__ __launch_bounds__(256) void k_wr(float* __restrict__ out, int n, size_t stride) {
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
#pragma unroll
    for (int a = 0; a < NARR; ++a) out[a * stride + p] = (float)p;
}
template <class F> float timeit(F f) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int i = 0; i < 10; ++i) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main() {
    for (int n : {500224, 4000000}) {
        size_t stride = n;
        float *in, *out; (void)hipMalloc(&in, stride * NARR * 4); (void)hipMalloc(&out, stride * NARR * 4);
        (void)hipMemset(in, 0, stride * NARR * 4);
        double mb = 2.0 * n * NARR * 4 * 1e-6;
        float a = timeit([&] { hipLaunchKernelGGL(k_dword, dim3((n + 255) / 256), dim3(256), 0, 0, in, out, n, stride); });
        float b = timeit([&] { hipLaunchKernelGGL(k_x4, dim3((n / 4 + 255) / 256), dim3(256), 0, 0, (const float4*)in, (float4*)out, n / 4, stride / 4); });
        float c = timeit([&] { hipLaunchKernelGGL(k_rd, dim3((n + 255) / 256), dim3(256), 0, 0, in, out, n, stride); });
        float d = timeit([&] { hipLaunchKernelGGL(k_wr, dim3((n + 255) / 256), dim3(256), 0, 0, out, n, stride); });
        printf("n=%8d  %6.1f MB copy: dword %.1f us (%.2f TB/s)   x4 %.1f us (%.2f TB/s)   read-only %.1f us (%.2f TB/s)  write-only %.1f us (%.2f TB/s)\n",
               n, mb, a * 1e3, mb / a * 1e-3, b * 1e3, mb / b * 1e-3, c * 1e3, mb / 2 / c * 1e-3, d * 1e3, mb / 2 / d * 1e-3);
        (void)hipFree(in); (void)hipFree(out);
    }
    return 0;
}
