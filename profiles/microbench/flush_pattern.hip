// Microbenchmark: cost of the LDS-tile flush (global float atomics) for two tile layouts, and what it does to the
// first-load latency of the workgroups that start behind it on the same CU (gfx950).
//   box   : tile = bounding box of the stencil nodes (6x6x6 at an arbitrary origin); lane i -> i-th node of the box
//   block : tile = whole 4^3 blocks (2x2x2 blocks covering the same box); lane i -> i-th node of a block
// Grid memory is 4^3-blocked (64 consecutive floats per block), 4 SoA components, as in plmpm_kernels.h.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics flush_pattern.hip -o flush_pattern.bin && ./flush_pattern.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__device__ __forceinline__ int node_index(int nb, int i, int j, int k) {
    return ((((k >> 2) * nb + (j >> 2)) * nb + (i >> 2)) << 6) | ((k & 3) << 4) | ((j & 3) << 2) | (i & 3);
}
template <int MODE>
__global__ __launch_bounds__(256) void k(const double* X, float* out, long long* lat, int N, int Np, int work, float* grid, size_t G) {
    __shared__ char pad[40 * 1024];
    const int nb = 32;
    const int p = blockIdx.x * 256 + threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    double x0 = 0, x1 = 0, x2 = 0;
    if (p < N) { x0 = X[p]; x1 = X[Np + p]; x2 = X[2 * Np + p]; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    float a = (float)(x0 + x1 + x2) + 1.0f;
    for (int i = 0; i < work; ++i) a = a * 1.0001f + 0.5f;
    pad[threadIdx.x] = (char)a;
    __syncthreads();
    // this workgroup's box origin: pseudo-random inside a 40^3 region (like the cube), not block aligned
    unsigned h = blockIdx.x * 2654435761u;
    int ox = 40 + (h & 31), oy = 10 + ((h >> 5) & 31), oz = 40 + ((h >> 10) & 31);
    if (MODE == 0) {
        for (int i = threadIdx.x; i < 216; i += 256) {
            int lz = i / 36, r = i - lz * 36, ly = r / 6, lx = r - ly * 6;
            int idx = node_index(nb, ox + lx, oy + ly, oz + lz);
            for (int c = 0; c < 4; ++c) atomicAdd(&grid[c * G + idx], a);
        }
    } else {
        // blocks covering the box: up to 3 per axis; flush whole blocks, skipping nodes outside the box (zero in LDS)
        int bx0 = ox >> 2, by0 = oy >> 2, bz0 = oz >> 2, nx = ((ox + 5) >> 2) - bx0 + 1, ny = ((oy + 5) >> 2) - by0 + 1, nz = ((oz + 5) >> 2) - bz0 + 1;
        int nblk = nx * ny * nz;
        for (int i = threadIdx.x; i < nblk * 64; i += 256) {
            int b = i >> 6, l = i & 63;
            int bz = b / (nx * ny), r = b - bz * nx * ny, by = r / nx, bx = r - by * nx;
            int gx = ((bx0 + bx) << 2) | (l & 3), gy = ((by0 + by) << 2) | ((l >> 2) & 3), gz = ((bz0 + bz) << 2) | (l >> 4);
            bool in = gx >= ox && gx < ox + 6 && gy >= oy && gy < oy + 6 && gz >= oz && gz < oz + 6;
            int idx = ((((bz0 + bz) * nb + (by0 + by)) * nb + (bx0 + bx)) << 6) | l;
            if (in) for (int c = 0; c < 4; ++c) atomicAdd(&grid[c * G + idx], a);
        }
    }
    if (p < N) out[p] = a + pad[(threadIdx.x + 1) & 255];
    if (threadIdx.x == 0) lat[blockIdx.x] = t1 - t0;
}
int main() {
    const int N = 500000, Np = 500224, nwg = Np / 256;
    const size_t G = 128 * 128 * 128;
    double* X; float *out, *grid; long long* lat;
    hipMalloc(&X, (size_t)3 * Np * 8); hipMalloc(&out, (size_t)Np * 4); hipMalloc(&lat, nwg * 8); hipMalloc(&grid, 4 * G * 4);
    hipMemset(X, 0, (size_t)3 * Np * 8); hipMemset(grid, 0, 4 * G * 4);
    std::vector<long long> h(nwg);
    auto run = [&](const char* name, auto kern, int work) {
        double sum = 0; long long p50 = 0, p90 = 0; float ms_tot = 0;
        for (int it = 0; it < 8; ++it) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, 0, X, out, lat, N, Np, work, grid, G);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (it < 2) continue;
            ms_tot += ms;
            hipMemcpy(h.data(), lat, nwg * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            for (auto v : h) sum += v;
            p50 += h[nwg / 2]; p90 += h[nwg * 9 / 10];
        }
        printf("%-10s work=%5d: kernel %6.1f us | first-load latency cycles: mean %7.0f p50 %6lld p90 %6lld\n", name, work, ms_tot / 6 * 1e3,
               sum / (6.0 * nwg), p50 / 6, p90 / 6);
    };
    for (int work : {500, 2000, 4000}) {
        run("box", k<0>, work);
        run("block", k<1>, work);
    }
    return 0;
}
