// Microbenchmark (round 5): "cells as the unit of work with the particle data transposed once" -- would the
// scatter phase be 1.5x faster (THE BAR, stated before measuring) in a cell-task formulation than in the product's form?
//
// (a)  the product's scatter phase: one particle per lane, in-wave bitonic sort on the cell key (so that same-cell lanes are
//      adjacent), per stencil node 4 values summed over the lanes of a run with 4 v_fmac_f32_dpp steps each (432 DPP per wave),
//      converted, and added to an f64 LDS tile by the run's head lane (108 ds_add_f64 per wave, ~10 active lanes each).
//      [(a0): the same without the sort -- what profiles/microbench/scatter_mfma_bound.hip (round 4) timed as "(a)".]
// (c)  the cell-task form: the workgroup ranks its 256 particles by cell (one ds_add_rtn_u32 per particle, a 64-entry scan,
//      4 barriers), every particle writes ONE record {wx[3], wy[3], wz[3], b[3], A[9]} (24 floats) to its ranked LDS slot, and
//      then a lane is a TASK (cell, i, j): it loops over the cell's particles, reads their records (broadcast reads: the nine
//      tasks of a cell read the same record), accumulates the 3 nodes k = 0..2 of its (i, j) column x {m, mv_x, mv_y, mv_z} in
//      registers (28 VALU per particle for 3 nodes -- no cross-lane reduction, no sort) and adds its 12 sums to the f64 LDS tile
//      at the end (12 ds_add_f64 per task, all lanes active).  LDS: 24 KiB of records + a 14 KiB tile (448 nodes) instead of the
//      32 KiB tile: four workgroups per CU as in the product.
// Cells per workgroup and particles per cell follow the benchmark: ~36 cells of Poisson(7.1)-like counts per 256 particles
// (seeded per workgroup); a workgroup's ~324 tasks are dealt round-robin to its 256 lanes.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scatter_celltask.hip -o scatter_celltask.bin && ./scatter_celltask.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int REP = 64;          // scatters per wave per launch (amortises the launch)
constexpr int MAXC = 64;         // cells per workgroup (capacity)

template <int D> __device__ __forceinline__ float row_shl(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + D, 0xf, 0xf, true));
}
__device__ __forceinline__ void seg_sum4(float& a, float& b, float& c, float& d, float m1, float m2, float m4, float m8) {
#define STEP(x, m, n) "v_fmac_f32_dpp " x ", " x ", " m " row_shl:" n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
    asm("s_nop 1\n" STEP("%0", "%4", "1") STEP("%1", "%4", "1") STEP("%2", "%4", "1") STEP("%3", "%4", "1")
        STEP("%0", "%5", "2") STEP("%1", "%5", "2") STEP("%2", "%5", "2") STEP("%3", "%5", "2")
        STEP("%0", "%6", "4") STEP("%1", "%6", "4") STEP("%2", "%6", "4") STEP("%3", "%6", "4")
        STEP("%0", "%7", "8") STEP("%1", "%7", "8") STEP("%2", "%7", "8") STEP("%3", "%7", "8")
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m1), "v"(m2), "v"(m4), "v"(m8));
#undef STEP
}
// the product's in-wave sort (plmpm_kernels.h: wave_sort_lanes32): bitonic network on (key << 6 | lane), 18 of 21 steps in DPP
template <int CTRL> __device__ __forceinline__ unsigned dpp_u32(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
template <int J> __device__ __forceinline__ unsigned partner_xor(unsigned v, int lane) {
    if (J == 1) return dpp_u32<0xB1>(v);
    if (J == 2) return dpp_u32<0x4E>(v);
    if (J == 4) { unsigned a = dpp_u32<0x124>(v), b = dpp_u32<0x12C>(v); return (lane & 4) ? a : b; }
    if (J == 8) return dpp_u32<0x128>(v);
    return (unsigned)__shfl_xor((int)v, J);
}
template <int K, int J> __device__ __forceinline__ unsigned bitonic_step(unsigned v, int lane) {
    const unsigned o = partner_xor<J>(v, lane);
    const unsigned mn = v < o ? v : o, mx = v < o ? o : v;
    return (((lane & J) == 0) == ((lane & K) == 0)) ? mn : mx;
}
template <int K> __device__ __forceinline__ unsigned bitonic_merge(unsigned v, int lane) {
    if (K >= 64) v = bitonic_step<K, 32>(v, lane);
    if (K >= 32) v = bitonic_step<K, 16>(v, lane);
    if (K >= 16) v = bitonic_step<K, 8>(v, lane);
    if (K >= 8) v = bitonic_step<K, 4>(v, lane);
    if (K >= 4) v = bitonic_step<K, 2>(v, lane);
    return bitonic_step<K, 1>(v, lane);
}
__device__ __forceinline__ int wave_sort_lanes32(unsigned key, int lane) {
    unsigned v = (key << 6) | (unsigned)lane;
    v = bitonic_merge<2>(v, lane); v = bitonic_merge<4>(v, lane); v = bitonic_merge<8>(v, lane);
    v = bitonic_merge<16>(v, lane); v = bitonic_merge<32>(v, lane); v = bitonic_merge<64>(v, lane);
    return (int)(v & 63u);
}

// cells[wg][0] = number of cells, cells[wg][1 + c] = particles of cell c (sum = 256)
__device__ __forceinline__ int cell_of(const int* cw, int t) {
    int c = 0, acc = cw[1];
    while (t >= acc) acc += cw[1 + ++c];
    return c;
}

// (a) -------------------------------------------------------------------------------------------------------------
template <bool SORT>
__global__ __launch_bounds__(256, 4) void k_dpp(const float* in, const int* cells, float* out) {
    __shared__ double tile[1024 * 4];
    for (int i = threadIdx.x; i < 4096; i += 256) tile[i] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, t = blockIdx.x * 256 + threadIdx.x;
    const int* cw = cells + blockIdx.x * (MAXC + 1);
    const int cell = cell_of(cw, threadIdx.x);
    // run structure: lanes of equal cell, clipped at the 16-lane rows
    const int prev = __shfl_up(cell, 1);
    const bool head = (lane & 15) == 0 || cell != prev;
    const unsigned long long heads = __ballot(head);
    const unsigned long long higher = lane == 63 ? 0ull : (heads >> (lane + 1));
    const int end = higher ? lane + __ffsll((long long)higher) - 1 : 63;
    const float m1 = lane + 1 <= end, m2 = lane + 2 <= end, m4 = lane + 4 <= end, m8 = lane + 8 <= end;
    float w[3][3], q0[3], ax[3], ay[3], az[3];
    for (int k = 0; k < 9; ++k) w[k / 3][k % 3] = in[t * 24 + k];
    for (int k = 0; k < 3; ++k) { q0[k] = in[t * 24 + 9 + k]; ax[k] = in[t * 24 + 12 + k]; ay[k] = in[t * 24 + 15 + k]; az[k] = in[t * 24 + 18 + k]; }
    float sink = 0.f;
    for (int r = 0; r < REP; ++r) {
        if (SORT) {
            // the sort and the three position shuffles that follow it in the product (the values are made to depend on r)
            const int src = wave_sort_lanes32((unsigned)(cell * 64 + ((lane * 37 + r) & 63)), lane);
            sink += __shfl(q0[0], src) + __shfl(q0[1], src) + __shfl(q0[2], src);
        }
#pragma unroll 1
        for (int i = 0; i < 3; ++i) {
            const float wi = i == 0 ? w[0][0] : (i == 1 ? w[1][0] : w[2][0]);
            float qi[3] = {q0[0] + i * ax[0], q0[1] + i * ax[1], q0[2] + i * ax[2]};
#pragma unroll 1
            for (int j = 0; j < 3; ++j) {
                float qj[3] = {qi[0] + j * ay[0], qi[1] + j * ay[1], qi[2] + j * ay[2]};
                const float wij = wi * (j == 0 ? w[0][1] : (j == 1 ? w[1][1] : w[2][1]));
                for (int l = 0; l < 3; ++l) {
                    const float wt = wij * w[l][2];
                    float a0 = wt * 1.5e-5f, a1 = wt * (qj[0] + l * az[0]), a2 = wt * (qj[1] + l * az[1]), a3 = wt * (qj[2] + l * az[2]);
                    seg_sum4(a0, a1, a2, a3, m1, m2, m4, m8);
                    if (head) {
                        double* q = tile + 4 * ((cell * 3 + l * 100 + j * 10 + i + r) & 1023);
                        atomicAdd(q, (double)a0); atomicAdd(q + 1, (double)a1); atomicAdd(q + 2, (double)a2); atomicAdd(q + 3, (double)a3);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = (float)tile[threadIdx.x * 4 + 1] + sink;
}

// (c) cell tasks ---------------------------------------------------------------------------------------------------
constexpr int TILE_C = 448;
__global__ __launch_bounds__(256, 4) void k_celltask(const float* in, const int* cells, float* out) {
    __shared__ double tile[TILE_C * 4];                    // 14 KiB
    __shared__ __attribute__((aligned(16))) float rec[256 * 24];      // 24 KiB: one record per particle, ranked by cell
    __shared__ int cnt[MAXC], start[MAXC];
    for (int i = threadIdx.x; i < TILE_C * 4; i += 256) tile[i] = 0.0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, t = blockIdx.x * 256 + threadIdx.x;
    const int* cw = cells + blockIdx.x * (MAXC + 1);
    const int ncell = cw[0];
    // the particles arrive in storage order: nearly, not exactly, cell-sorted (a permutation inside the workgroup)
    const int mycell = cell_of(cw, (threadIdx.x * 37 + 11) & 255);
    float w[3][3], q0[3], ax[3], ay[3], az[3];
    for (int k = 0; k < 9; ++k) w[k / 3][k % 3] = in[t * 24 + k];
    for (int k = 0; k < 3; ++k) { q0[k] = in[t * 24 + 9 + k]; ax[k] = in[t * 24 + 12 + k]; ay[k] = in[t * 24 + 15 + k]; az[k] = in[t * 24 + 18 + k]; }
    __syncthreads();
    for (int r = 0; r < REP; ++r) {
        // ---- rank the workgroup's particles by cell
        if (threadIdx.x < MAXC) cnt[threadIdx.x] = 0;
        __syncthreads();
        const int slot = atomicAdd(&cnt[mycell], 1);                        // ds_add_rtn_u32
        __syncthreads();
        if (wave == 0) {                                                    // exclusive scan of the 64 counts
            int v = cnt[lane], s = v;
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(s, d); if (lane >= d) s += o; }
            start[lane] = s - v;
        }
        __syncthreads();
        // ---- one record per particle at its ranked slot: wx wy (6) pad (2) | wz b0 | b1 b2 A0 A1 | A2..A5 | A6 A7 A8 pad
        {
            float* q = rec + (start[mycell] + slot) * 24;
            const float rr = 1.0f + 1e-6f * r;
            *reinterpret_cast<v4f*>(q) = v4f{w[0][0] * rr, w[1][0], w[2][0], w[0][1]};
            *reinterpret_cast<v4f*>(q + 4) = v4f{w[1][1], w[2][1], 0.f, 0.f};
            *reinterpret_cast<v4f*>(q + 8) = v4f{w[0][2], w[1][2], w[2][2], q0[0]};
            *reinterpret_cast<v4f*>(q + 12) = v4f{q0[1], q0[2], ax[0], ay[0]};
            *reinterpret_cast<v4f*>(q + 16) = v4f{az[0], ax[1], ay[1], az[1]};
            *reinterpret_cast<v4f*>(q + 20) = v4f{ax[2], ay[2], az[2], 0.f};
        }
        __syncthreads();
        // ---- tasks (cell, i, j): the column k = 0..2 of 3 nodes x 4 values, summed over the cell's particles in registers
        for (int task = threadIdx.x; task < ncell * 9; task += 256) {
            const int c = task / 9, ij = task - 9 * c, i = ij / 3, j = ij - 3 * i;
            const float fi = (float)i, fj = (float)j;
            const int n = cnt[c];
            const float* q = rec + start[c] * 24;
            float am[3] = {0.f, 0.f, 0.f}, a0[3] = {0.f, 0.f, 0.f}, a1[3] = {0.f, 0.f, 0.f}, a2[3] = {0.f, 0.f, 0.f};
#pragma unroll 1
            for (int p = 0; p < n; ++p, q += 24) {
                const float wx = q[i], wy = q[3 + j];
                const v4f r2 = *reinterpret_cast<const v4f*>(q + 8), r3 = *reinterpret_cast<const v4f*>(q + 12);
                const v4f r4 = *reinterpret_cast<const v4f*>(q + 16), r5 = *reinterpret_cast<const v4f*>(q + 20);
                const float wxy = wx * wy;
                // b + i A[:, 0] + j A[:, 1]
                float v0 = r2.w + fi * r3.z + fj * r3.w, v1 = r3.x + fi * r4.y + fj * r4.z, v2 = r3.y + fi * r5.x + fj * r5.y;
                const float w0 = wxy * r2.x, w1 = wxy * r2.y, w2 = wxy * r2.z;
                am[0] += w0; a0[0] += w0 * v0; a1[0] += w0 * v1; a2[0] += w0 * v2;
                v0 += r4.x; v1 += r4.w; v2 += r5.z;
                am[1] += w1; a0[1] += w1 * v0; a1[1] += w1 * v1; a2[1] += w1 * v2;
                v0 += r4.x; v1 += r4.w; v2 += r5.z;
                am[2] += w2; a0[2] += w2 * v0; a1[2] += w2 * v1; a2[2] += w2 * v2;
            }
            for (int l = 0; l < 3; ++l) {
                double* d = tile + 4 * ((c * 3 + l * 100 + j * 10 + i + r) % TILE_C);
                atomicAdd(d, (double)(am[l] * 1.5e-5f)); atomicAdd(d + 1, (double)a0[l]); atomicAdd(d + 2, (double)a1[l]); atomicAdd(d + 3, (double)a2[l]);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = (float)tile[threadIdx.x * 4 + 1];
}

template <class F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int k = 0; k < 3; ++k) {
        hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
    }
    return best;
}
int main() {
    const int wgs = 1024;                                   // 4 workgroups per CU, as the scatter kernels run
    float *in, *out; int* cells;
    hipMalloc(&in, (size_t)wgs * 256 * 24 * 4); hipMalloc(&out, wgs * 64 * 4); hipMalloc(&cells, (size_t)wgs * (MAXC + 1) * 4);
    std::vector<float> h((size_t)wgs * 256 * 24);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.1f + 0.001f * (float)(i % 977);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    // Poisson(7.1)-like cell counts summing to 256 per workgroup
    std::vector<int> hc((size_t)wgs * (MAXC + 1), 0);
    srand(1);
    double tasks = 0, ncells = 0, longest = 0;
    for (int g = 0; g < wgs; ++g) {
        int left = 256, c = 0, mx = 0;
        while (left > 0 && c < MAXC) {
            int n = 0;                                      // sum of 71 Bernoulli(0.1): mean 7.1, variance 6.4
            for (int k = 0; k < 71; ++k) n += (rand() % 10) == 0;
            n = n < 1 ? 1 : n;
            n = n > left ? left : n;
            if (c == MAXC - 1) n = left;
            hc[(size_t)g * (MAXC + 1) + 1 + c++] = n; left -= n; mx = n > mx ? n : mx;
        }
        hc[(size_t)g * (MAXC + 1)] = c;
        tasks += 9.0 * c; ncells += c; longest += mx;
    }
    hipMemcpy(cells, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
    printf("workgroups %d: %.1f cells and %.0f tasks per 256 particles on average, longest cell %.1f\n", wgs, ncells / wgs, tasks / wgs, longest / wgs);
    const float a0 = timeit([&] { hipLaunchKernelGGL(k_dpp<false>, dim3(wgs), dim3(256), 0, 0, in, cells, out); });
    const float a = timeit([&] { hipLaunchKernelGGL(k_dpp<true>, dim3(wgs), dim3(256), 0, 0, in, cells, out); });
    const float c = timeit([&] { hipLaunchKernelGGL(k_celltask, dim3(wgs), dim3(256), 0, 0, in, cells, out); });
    // per wave and scatter, with all 16 waves of a CU busy: ms * 1e6 ns / REP / (waves per CU = 16)
    printf("(a0) DPP segmented reduction + 108 ds_add_f64, no sort : %8.3f ms = %7.1f ns per 64-particle scatter per CU\n", a0, a0 * 1e6 / REP / 16.0);
    printf("(a)  the same behind the in-wave sort (the product)    : %8.3f ms = %7.1f ns per 64-particle scatter per CU\n", a, a * 1e6 / REP / 16.0);
    printf("(c)  cell tasks (rank by cell, records in LDS, no DPP) : %8.3f ms = %7.1f ns per 64-particle scatter per CU\n", c, c * 1e6 / REP / 16.0);
    printf("ratio (a) / (c) = %.2f  (the bar for building it: 1.5, stated before the measurement)\n", a / c);
    return 0;
}
