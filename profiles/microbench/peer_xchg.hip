// What one device-side halo exchange costs on the GPU, piece by piece (self loop-back, one process):
//   copy of the exchange planes into coarse- / fine-grained device memory, the "all workgroups done" counter, the arrival
//   counter store + poll, and a consumer kernel reading the received planes from coarse- / fine-grained memory.
// hipcc --offload-arch=gfx950 -O3 peer_xchg.hip -o peer_xchg && ./peer_xchg
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE, int SLEEP = 1>   // 0 copy | 1 + fence + done counter | 2 + arrival store + poll | 3 like 2, no per-thread system fence (barrier + one
                                      // release per workgroup) | 4 like 2 with "s_waitcnt vmcnt(0)" per wave instead of any fence (NOT enough: fine-grained
                                      // memory is cached) | 5 like 4 with write-through stores (sc0 sc1) for data and counter: no L2 write-back anywhere
__global__ __launch_bounds__(256) void k_x(const float4* src, float4* dst, size_t nv, unsigned* done, unsigned* arrive, unsigned seq) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    if (MODE == 5) for (size_t j = tid; j < nv; j += nth) { typedef float __attribute__((ext_vector_type(4))) vec16; vec16 v = ((const vec16*)src)[j]; asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst + j), "v"(v) : "memory"); }
    else for (size_t j = tid; j < nv; j += nth) dst[j] = src[j];
    if (MODE == 0) return;
    if (MODE >= 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (MODE != 3) __threadfence_system();
    __syncthreads();
    __shared__ int last;
    if (threadIdx.x == 0) {
        if (MODE == 3) __threadfence_system();
        last = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0) {
        *done = 0;
        if (MODE == 5) {
            asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(arrive), "v"(seq) : "memory");
            while ((int)(__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) __builtin_amdgcn_s_sleep(SLEEP);
        } else if (MODE >= 2) {
            __threadfence_system();
            __hip_atomic_store(arrive, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            while ((int)(__hip_atomic_load(arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) __builtin_amdgcn_s_sleep(SLEEP);
            __threadfence_system();
        }
    }
}
__global__ __launch_bounds__(256) void k_read(const float4* src, float* out, size_t nv) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    float a = 0;
    for (size_t j = tid; j < nv; j += nth) { float4 v = src[j]; a += v.x + v.y + v.z + v.w; }
    if (a == 12345.f) out[0] = a;
}

int main() {
    const size_t bytes = 1769472;     // 2 faces x 2 block planes x 24 x 18 blocks x 64 nodes x 4 comps x 4 B (config 3 cut four ways)
    const size_t nv = bytes / 16;
    float4 *src, *coarse, *fine; unsigned *done, *arr_c, *arr_f; float* out;
    CHK(hipMalloc(&src, bytes)); CHK(hipMalloc(&coarse, bytes)); CHK(hipMalloc(&done, 256)); CHK(hipMalloc(&arr_c, 256)); CHK(hipMalloc(&out, 256));
    CHK(hipExtMallocWithFlags((void**)&fine, bytes, hipDeviceMallocFinegrained));
    CHK(hipExtMallocWithFlags((void**)&arr_f, 256, hipDeviceMallocFinegrained));
    CHK(hipMemset(src, 1, bytes)); CHK(hipMemset(done, 0, 256)); CHK(hipMemset(arr_c, 0, 256)); CHK(hipMemset(arr_f, 0, 256));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int reps = 200;
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 10; ++i) launch(i + 1);
        hipEventRecord(e0, 0);
        for (int i = 0; i < reps; ++i) launch(100 + i);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-78s %6.2f us per launch\n", name, 1e3 * ms / reps);
        return 0;
    };
    for (int nwg : {27, 64, 108, 256}) {
        printf("-- %d workgroups, %.2f MB\n", nwg, bytes * 1e-6);
        timeit("copy -> coarse", [&](unsigned s) { hipLaunchKernelGGL(k_x<0>, dim3(nwg), dim3(256), 0, 0, src, coarse, nv, done, arr_c, s); });
        timeit("copy -> fine-grained", [&](unsigned s) { hipLaunchKernelGGL(k_x<0>, dim3(nwg), dim3(256), 0, 0, src, fine, nv, done, arr_f, s); });
        timeit("copy -> fine + system fence per thread + done counter", [&](unsigned s) { hipLaunchKernelGGL(k_x<1>, dim3(nwg), dim3(256), 0, 0, src, fine, nv, done, arr_f, s); });
        timeit("copy -> fine + fences + done + arrival store / poll (fine), sleep 64", [&](unsigned s) { hipLaunchKernelGGL((k_x<2, 64>), dim3(nwg), dim3(256), 0, 0, src, fine, nv, done, arr_f, s); });
        timeit("copy -> fine + fences + done + arrival store / poll (fine), sleep 1", [&](unsigned s) { hipLaunchKernelGGL(k_x<2>, dim3(nwg), dim3(256), 0, 0, src, fine, nv, done, arr_f, s); });
        timeit("copy -> fine + barrier + one fence + done + arrival (fine), sleep 1", [&](unsigned s) { hipLaunchKernelGGL(k_x<3>, dim3(nwg), dim3(256), 0, 0, src, fine, nv, done, arr_f, s); });
        timeit("copy -> fine + s_waitcnt vmcnt(0) + barrier + done + arrival (fine), sleep 1", [&](unsigned s) { hipLaunchKernelGGL((k_x<4>), dim3(nwg), dim3(256), 0, 0, src, fine, nv, done, arr_f, s); });
        timeit("copy -> fine, write-through stores + s_waitcnt + barrier + done + arrival, sleep 1", [&](unsigned s) { hipLaunchKernelGGL((k_x<5>), dim3(nwg), dim3(256), 0, 0, src, fine, nv, done, arr_f, s); });
        timeit("copy -> coarse + barrier + one fence + done + arrival (fine), sleep 1", [&](unsigned s) { hipLaunchKernelGGL(k_x<3>, dim3(nwg), dim3(256), 0, 0, src, coarse, nv, done, arr_f, s); });
        timeit("copy -> coarse + everything coarse, sleep 1", [&](unsigned s) { hipLaunchKernelGGL(k_x<3>, dim3(nwg), dim3(256), 0, 0, src, coarse, nv, done, arr_c, s); });
    }
    printf("-- consumer\n");
    timeit("read 1.77 MB from coarse (256 workgroups)", [&](unsigned) { hipLaunchKernelGGL(k_read, dim3(256), dim3(256), 0, 0, coarse, out, nv); });
    timeit("read 1.77 MB from fine-grained (256 workgroups)", [&](unsigned) { hipLaunchKernelGGL(k_read, dim3(256), dim3(256), 0, 0, fine, out, nv); });
    timeit("empty-ish kernel (1 workgroup copy of 4 KB)", [&](unsigned s) { hipLaunchKernelGGL(k_x<0>, dim3(1), dim3(256), 0, 0, src, coarse, (size_t)256, done, arr_c, s); });
    return 0;
}
