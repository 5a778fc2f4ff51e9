// Microbenchmark (round 4): the tile flush of the scatter kernels -- every workgroup adds the 4 sums (mass, momentum) of the ~216 nodes
// of its box to the grid with global fp32 atomics; a node lies in ~4.5 boxes.  grid_in is four arrays (SoA: the 4 atomics of a node go
// to 4 cache lines); would one array of float4 (AoS: the same line) make the flush cheaper?
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics flush_layout.hip -o flush_layout && ./flush_layout
// Model: 128^3 grid in 4^3 blocks (the engine's layout), 1 954 workgroups whose 6x6x6 boxes tile a 40^3-cell body with the overlap of
// the real thing (box origins on a 3.7-cell lattice), every box node added once per launch.  Reported: us per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int N = 128, NB = N / 4;
__device__ __forceinline__ int node_index(int x, int y, int z) {
    return ((((z >> 2) * NB + (y >> 2)) * NB + (x >> 2)) << 6) | ((z & 3) << 4) | ((y & 3) << 2) | (x & 3);
}
template <int MODE>   // 0 SoA (4 arrays), 1 AoS (float4 per node), 2 AoS with one 64-bit... (not available for fp32 pairs): same as 1
__global__ __launch_bounds__(256) void k(float* g, const int* org, int e) {
    const int o0 = org[3 * blockIdx.x], o1 = org[3 * blockIdx.x + 1], o2 = org[3 * blockIdx.x + 2];
    const size_t G = (size_t)N * N * N;
    for (int i = threadIdx.x; i < e * e * e; i += 256) {
        const int lx = i % e, ly = (i / e) % e, lz = i / (e * e);
        const int idx = node_index(o0 + lx, o1 + ly, o2 + lz);
        const float a = 1.0f + i;
        if (MODE == 0) { atomicAdd(g + idx, a); atomicAdd(g + G + idx, a); atomicAdd(g + 2 * G + idx, a); atomicAdd(g + 3 * G + idx, a); }
        else { float* q = g + 4 * (size_t)idx; atomicAdd(q, a); atomicAdd(q + 1, a); atomicAdd(q + 2, a); atomicAdd(q + 3, a); }
    }
}
int main() {
    const int e = 6;
    std::vector<int> org;
    // box origins on a lattice of pitch 3.7 cells inside [44, 84)^3: ~ (40 / 3.7)^3 = 1 260 boxes; repeat part of them to reach 1 954
    for (int r = 0; (int)org.size() / 3 < 1954; ++r)
        for (float z = 44; z < 84 - e && (int)org.size() / 3 < 1954; z += 3.7f)
            for (float y = 20; y < 60 - e; y += 3.7f)
                for (float x = 44; x < 84 - e; x += 3.7f) { org.push_back((int)x + r); org.push_back((int)y); org.push_back((int)z); }
    const int nwg = 1954;
    int* dorg; float* g;
    hipMalloc(&dorg, org.size() * 4); hipMemcpy(dorg, org.data(), org.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&g, (size_t)N * N * N * 16); hipMemset(g, 0, (size_t)N * N * N * 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            auto go = [&] { if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(256), 0, 0, g, dorg, e); else hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(256), 0, 0, g, dorg, e); };
            go(); hipDeviceSynchronize();
            hipEventRecord(a); for (int i = 0; i < 50; ++i) go(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("%s  %d workgroups x %d nodes x 4 atomics: %.2f us per launch\n", mode ? "AoS float4" : "SoA 4 arrays", nwg, e * e * e, ms * 1000 / 50);
        }
    return 0;
}
