// Microbenchmark: how long does the FIRST global load of a freshly started workgroup take on gfx950 when the
// kernel looks like the particle kernels (1954 workgroups x 256 threads, SoA f64 positions, some work, stores)?
// hipcc --offload-arch=gfx950 -O3 first_load_latency.hip -o first_load_latency.bin && ./first_load_latency.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

// work: number of dependent fma (x4 cycles each roughly); lds_kb: static LDS to limit occupancy like the real kernels
template <int LDS_KB>
__global__ __launch_bounds__(256) void k(const double* X, float* out, long long* lat, int N, int Np, int work, int natom, float* grid) {
    __shared__ char pad[LDS_KB * 1024];
    const int p = blockIdx.x * 256 + threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    double x0 = 0, x1 = 0, x2 = 0;
    if (p < N) { x0 = X[p]; x1 = X[Np + p]; x2 = X[2 * Np + p]; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    float a = (float)(x0 + x1 + x2);
    for (int i = 0; i < work; ++i) a = a * 1.0001f + 0.5f;
    pad[threadIdx.x] = (char)a;
    __syncthreads();
    // tail: scattered float atomics like a tile flush
    for (int i = 0; i < natom; ++i) {
        unsigned idx = ((unsigned)(blockIdx.x * 977 + i * 256 + threadIdx.x) * 2654435761u) >> 12;   // 1M floats
        atomicAdd(&grid[idx], a);
    }
    if (p < N) out[p] = a + pad[(threadIdx.x + 1) & 255];
    if (threadIdx.x == 0) lat[blockIdx.x] = t1 - t0;
}

int main() {
    const int N = 500000, Np = 500224, nwg = Np / 256, NF = 64;
    double* X; float *out, *grid; long long* lat;
    hipMalloc(&X, (size_t)NF * 3 * Np * 8); hipMalloc(&out, (size_t)Np * 4); hipMalloc(&lat, nwg * 8); hipMalloc(&grid, 4 << 20);
    hipMemset(X, 0, (size_t)NF * 3 * Np * 8); hipMemset(grid, 0, 4 << 20);
    std::vector<long long> h(nwg);
    auto run = [&](const char* name, auto kern, int work, int natom, bool cold) {
        double sum = 0; long long p50 = 0, p90 = 0, mx = 0; float ms_tot = 0;
        for (int it = 0; it < 8; ++it) {
            const double* Xf = X + (size_t)(cold ? (it * 7) % NF : 0) * 3 * Np;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, 0, Xf, out, lat, N, Np, work, natom, grid);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (it < 2) continue;
            ms_tot += ms;
            hipMemcpy(h.data(), lat, nwg * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            for (auto v : h) sum += v;
            p50 += h[nwg / 2]; p90 += h[nwg * 9 / 10]; mx = std::max(mx, h.back());
        }
        printf("%-34s work=%6d atomics/thr=%2d %s: kernel %6.1f us | first-load latency cycles: mean %7.0f p50 %6lld p90 %6lld max %7lld\n",
               name, work, natom, cold ? "cold" : "warm", ms_tot / 6 * 1e3, sum / (6.0 * nwg), p50 / 6, p90 / 6, mx);
    };
    for (int cold = 0; cold < 2; ++cold) {
        run("lds 0 KB", k<1>, 0, 0, cold);
        run("lds 0 KB", k<1>, 2000, 0, cold);
        run("lds 48 KB (3 WG/CU)", k<48>, 2000, 0, cold);
        run("lds 48 KB (3 WG/CU)", k<48>, 2000, 4, cold);
        run("lds 48 KB (3 WG/CU)", k<48>, 2000, 12, cold);
        run("lds 32 KB (4-5 WG/CU)", k<32>, 4000, 8, cold);
        run("lds 64 KB (2 WG/CU)", k<64>, 3000, 0, cold);
    }
    return 0;
}
