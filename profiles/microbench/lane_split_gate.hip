// Round 6 gate: would a kernel family with SEVERAL LANES PER PARTICLE shorten a workgroup's lifetime in the
// latency regime (<= 1-2 workgroups per CU: every rank at N >= 2 on 128^3 / 500k)?
//
// THE BAR, stated before measuring: the gather + scatter phases of the new lane map, for the SAME 256 particles on a CU that has
// nothing else to do, take <= 0.5 x the time of today's one-particle-per-lane form.
//
// (a)  today's form, one 256-thread workgroup = 256 particles, one wave per SIMD: 27-node sum-factorised gather of 3 components
//      through an LDS tile (27 ds_read_b128, ~280 multiply-adds), then the product's scatter (27 nodes x 4 values, 4
//      v_fmac_f32_dpp steps each, convert, ds_add_f64 by the run's head lane) behind the in-wave sort.
// (q4) four lanes per particle (lane & 3 = component: {m | v_x, mv_x | v_y, mv_y | v_z, mv_z}), 64 particles per workgroup, so
//      the same 256 particles are FOUR workgroups (4 waves per SIMD, each with a chain a quarter as long): the gather reads one
//      component per lane (27 ds_read_b32, ~93 multiply-adds, lane 3 idles), the scatter computes one component per lane (27
//      weights + 27 affine steps + 27 products), sums it over the same-cell particles of the 16-lane row with TWO DPP steps
//      (row_shl:4, :8 -- runs are clipped at 4 particles), converts, and one ds_add_f64 per node carries all four components.
// Both do REP rounds; time per round = what a CU needs for 256 particles' gather + scatter phases.  Also printed: the same with
// 4 x the workgroups (the throughput regime), where the q4 form's extra weight arithmetic and shorter runs are pure cost.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp lane_split_gate.hip -o lane_split_gate.bin && ./lane_split_gate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int REP = 64;
constexpr int MAXC = 64;

#define STEP(x, m, n) "v_fmac_f32_dpp " x ", " x ", " m " row_shl:" n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
__device__ __forceinline__ void seg_sum4(float& a, float& b, float& c, float& d, float m1, float m2, float m4, float m8) {
    asm("s_nop 1\n" STEP("%0", "%4", "1") STEP("%1", "%4", "1") STEP("%2", "%4", "1") STEP("%3", "%4", "1")
        STEP("%0", "%5", "2") STEP("%1", "%5", "2") STEP("%2", "%5", "2") STEP("%3", "%5", "2")
        STEP("%0", "%6", "4") STEP("%1", "%6", "4") STEP("%2", "%6", "4") STEP("%3", "%6", "4")
        STEP("%0", "%7", "8") STEP("%1", "%7", "8") STEP("%2", "%7", "8") STEP("%3", "%7", "8")
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m1), "v"(m2), "v"(m4), "v"(m8));
}
// three values of one lane, two steps: the same-component lanes of the row's other particles are 4 and 8 lanes to the right
__device__ __forceinline__ void quad_sum3(float& a, float& b, float& c, float m4, float m8) {
    asm("s_nop 1\n" STEP("%0", "%3", "4") STEP("%1", "%3", "4") STEP("%2", "%3", "4") "s_nop 0\n"
        STEP("%0", "%4", "8") STEP("%1", "%4", "8") STEP("%2", "%4", "8")
        : "+v"(a), "+v"(b), "+v"(c) : "v"(m4), "v"(m8));
}
#undef STEP
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
__device__ __forceinline__ int cell_of(const int* cw, int t) {
    int c = 0, acc = cw[1];
    while (t >= acc) acc += cw[1 + ++c];
    return c;
}

// (a) one particle per lane ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 4) void k_lane(const float* in, const int* cells, float* out) {
    __shared__ double tile[1024 * 4];
    v4f* tile_v = reinterpret_cast<v4f*>(tile);
    const int lane = threadIdx.x & 63, t = blockIdx.x * 256 + threadIdx.x;
    const int* cw = cells + (blockIdx.x & 1023) * (MAXC + 1);
    const int cell = cell_of(cw, threadIdx.x);
    const int prev = __shfl_up(cell, 1);
    const bool head = (lane & 15) == 0 || cell != prev;
    const unsigned long long heads = __ballot(head);
    const unsigned long long higher = lane == 63 ? 0ull : (heads >> (lane + 1));
    const int end = higher ? lane + __ffsll((long long)higher) - 1 : 63;
    const float m1 = lane + 1 <= end, m2 = lane + 2 <= end, m4 = lane + 4 <= end, m8 = lane + 8 <= end;
    float w[3][3], zw[3][3], q0[3], ax[3], ay[3], az[3];
    for (int k = 0; k < 9; ++k) { w[k / 3][k % 3] = in[t * 24 + k]; zw[k / 3][k % 3] = w[k / 3][k % 3] * (k / 3 - 0.5f); }
    for (int k = 0; k < 3; ++k) { q0[k] = in[t * 24 + 9 + k]; ax[k] = in[t * 24 + 12 + k]; ay[k] = in[t * 24 + 15 + k]; az[k] = in[t * 24 + 18 + k]; }
    float sink = 0.f;
    for (int r = 0; r < REP; ++r) {
        // ---- gather phase: tile of {v_x, v_y, v_z, 0}, sum-factorised contraction (mpm_math.h: g2p_particle)
        for (int i = threadIdx.x; i < 1024; i += 256) tile_v[i] = v4f{0.1f + i * 1e-3f, 0.2f, 0.3f + r, 0.f};
        __syncthreads();
        float vn[3] = {0.f, 0.f, 0.f}, Cn[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int b0 = (cell * 3 + r) & 511;
#pragma unroll 1
        for (int i = 0; i < 3; ++i) {
            float Sww[3] = {0.f, 0.f, 0.f}, Szw[3] = {0.f, 0.f, 0.f}, Swz[3] = {0.f, 0.f, 0.f};
            for (int j = 0; j < 3; ++j) {
                float Rw[3] = {0.f, 0.f, 0.f}, Rz[3] = {0.f, 0.f, 0.f};
                for (int l = 0; l < 3; ++l) {
                    const v4f g = tile_v[b0 + l * 64 + j * 8 + i];
                    Rw[0] += w[l][2] * g.x; Rw[1] += w[l][2] * g.y; Rw[2] += w[l][2] * g.z;
                    Rz[0] += zw[l][2] * g.x; Rz[1] += zw[l][2] * g.y; Rz[2] += zw[l][2] * g.z;
                }
                for (int a = 0; a < 3; ++a) { Sww[a] += w[j][1] * Rw[a]; Szw[a] += zw[j][1] * Rw[a]; Swz[a] += w[j][1] * Rz[a]; }
            }
            const float wi = i == 0 ? w[0][0] : (i == 1 ? w[1][0] : w[2][0]), zi = i == 0 ? zw[0][0] : (i == 1 ? zw[1][0] : zw[2][0]);
            for (int a = 0; a < 3; ++a) { vn[a] += wi * Sww[a]; Cn[3 * a] += zi * Sww[a]; Cn[3 * a + 1] += wi * Szw[a]; Cn[3 * a + 2] += wi * Swz[a]; }
        }
        for (int a = 0; a < 3; ++a) q0[a] += 1e-6f * (vn[a] + Cn[3 * a] + Cn[3 * a + 1] + Cn[3 * a + 2]);
        __syncthreads();
        for (int i = threadIdx.x; i < 4096; i += 256) tile[i] = 0.0;
        __syncthreads();
        // ---- scatter phase (the product's form, behind the in-wave sort)
        const int src = wave_sort_lanes32((unsigned)(cell * 64 + ((lane * 37 + r) & 63)), lane);
        sink += __shfl(q0[0], src) + __shfl(q0[1], src) + __shfl(q0[2], src);
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
        __syncthreads();
    }
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = (float)tile[threadIdx.x * 4 + 1] + sink + q0[0];
}

// (q4) four lanes per particle -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 4) void k_quad(const float* in, const int* cells, float* out) {
    __shared__ double tile[448 * 4];                       // a 64-particle workgroup's box is small: 14 KiB
    float* tile_f = reinterpret_cast<float*>(tile);
    const int lane = threadIdx.x & 63, comp = lane & 3;
    const int part = threadIdx.x >> 2;                     // particle of this workgroup (0 .. 63)
    const int t = (blockIdx.x * 64 + part) & (1024 * 256 - 1);
    const int* cw = cells + ((blockIdx.x >> 2) & 1023) * (MAXC + 1);
    const int cell = cell_of(cw, (blockIdx.x & 3) * 64 + part);
    // runs of same-cell particles inside the 16-lane row (4 particles): this lane adds the lane 4 / 8 to the right
    const int prev = __shfl_up(cell, 4);
    const bool head = (lane & 12) == 0 || cell != prev;
    const int c1 = __shfl_down(cell, 4), c2 = __shfl_down(cell, 8), c3 = __shfl_down(cell, 12);
    const int pr = (lane >> 2) & 3;                        // particle slot inside the row
    const int len = 1 + ((pr < 3 && c1 == cell) ? 1 + ((pr < 2 && c2 == cell) ? 1 + ((pr < 1 && c3 == cell) ? 1 : 0) : 0) : 0);
    const float m4 = len >= 2 ? 1.f : 0.f, m8 = len >= 3 ? 1.f : 0.f;        // (run of 4: head adds +4, then +8 picks up +8 and +12)
    float w[3][3], zw[3][3];
    for (int k = 0; k < 9; ++k) { w[k / 3][k % 3] = in[t * 24 + k]; zw[k / 3][k % 3] = w[k / 3][k % 3] * (k / 3 - 0.5f); }
    // this lane's component of the affine momentum: q0, ax, ay, az (component 0 = mass: constant)
    float q0 = comp ? in[t * 24 + 8 + comp] : 1.5e-5f, ax = comp ? in[t * 24 + 11 + comp] : 0.f, ay = comp ? in[t * 24 + 14 + comp] : 0.f,
          az = comp ? in[t * 24 + 17 + comp] : 0.f;
    float sink = 0.f;
    for (int r = 0; r < REP; ++r) {
        for (int i = threadIdx.x; i < 448 * 4; i += 256) tile_f[i] = 0.1f + i * 1e-3f + r;
        __syncthreads();
        // ---- gather: one component per lane (lane 3 of the quad repeats component 2: it has nothing of its own to do)
        const int b0 = ((cell * 3 + r) & 255) * 4 + (comp < 3 ? comp : 2);
        float vn = 0.f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
#pragma unroll 1
        for (int i = 0; i < 3; ++i) {
            float Sww = 0.f, Szw = 0.f, Swz = 0.f;
            for (int j = 0; j < 3; ++j) {
                float Rw = 0.f, Rz = 0.f;
                for (int l = 0; l < 3; ++l) {
                    const float g = tile_f[b0 + 4 * (l * 36 + j * 6 + i)];
                    Rw += w[l][2] * g; Rz += zw[l][2] * g;
                }
                Sww += w[j][1] * Rw; Szw += zw[j][1] * Rw; Swz += w[j][1] * Rz;
            }
            const float wi = i == 0 ? w[0][0] : (i == 1 ? w[1][0] : w[2][0]), zi = i == 0 ? zw[0][0] : (i == 1 ? zw[1][0] : zw[2][0]);
            vn += wi * Sww; C0 += zi * Sww; C1 += wi * Szw; C2 += wi * Swz;
        }
        q0 += 1e-6f * (vn + C0 + C1 + C2);
        __syncthreads();
        for (int i = threadIdx.x; i < 448 * 4; i += 256) tile[i] = 0.0;
        __syncthreads();
        // ---- scatter: the in-wave sort (quads travel together: the key carries the particle), then one component per lane
        const int src = wave_sort_lanes32((unsigned)((cell * 16 + ((part * 5 + r) & 15)) * 4 + comp), lane);
        sink += __shfl(q0, src);
#pragma unroll 1
        for (int i = 0; i < 3; ++i) {
            const float wi = i == 0 ? w[0][0] : (i == 1 ? w[1][0] : w[2][0]);
            const float qi = q0 + i * ax;
            for (int j = 0; j < 3; ++j) {
                const float wij = wi * w[j][1], qj = qi + j * ay;
                float a0 = wij * w[0][2] * qj, a1 = wij * w[1][2] * (qj + az), a2 = wij * w[2][2] * (qj + 2.f * az);
                quad_sum3(a0, a1, a2, m4, m8);
                if (head) {
                    double* q = tile + 4 * ((cell * 3 + j * 10 + i + r) % 400) + comp;
                    atomicAdd(q, (double)a0); atomicAdd(q + 4 * 16, (double)a1); atomicAdd(q + 4 * 32, (double)a2);
                }
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = (float)tile[threadIdx.x * 4 + 1] + sink + q0;
}

template <class F> float timeit(F f) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int k = 0; k < 5; ++k) {
        (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
    }
    return best;
}
int main() {
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    float *in, *out; int* cells;
    (void)hipMalloc(&in, (size_t)1024 * 256 * 24 * 4); (void)hipMalloc(&out, (size_t)4096 * 4 * 64 * 4); (void)hipMalloc(&cells, (size_t)1024 * (MAXC + 1) * 4);
    std::vector<float> h((size_t)1024 * 256 * 24);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.1f + 0.001f * (float)(i % 977);
    (void)hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<int> hc((size_t)1024 * (MAXC + 1), 0);
    srand(1);
    for (int g = 0; g < 1024; ++g) {
        int left = 256, c = 0;
        while (left > 0 && c < MAXC) {
            int n = 0;
            for (int k = 0; k < 71; ++k) n += (rand() % 10) == 0;
            n = n < 1 ? 1 : n; n = n > left ? left : n;
            if (c == MAXC - 1) n = left;
            hc[(size_t)g * (MAXC + 1) + 1 + c++] = n; left -= n;
        }
        hc[(size_t)g * (MAXC + 1)] = c;
    }
    (void)hipMemcpy(cells, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
    printf("# %s, %d CUs; gather + scatter phases of 256 particles per CU-slot, REP = %d rounds per launch; us per round\n", pr.gcnArchName, cus, REP);
    for (int per_cu = 1; per_cu <= 4; per_cu *= 2) {
        const int sets = cus * per_cu;                     // 256-particle sets in flight: per_cu per CU
        const float a = timeit([&] { hipLaunchKernelGGL(k_lane, dim3(sets), dim3(256), 0, 0, in, cells, out); });
        const float q = timeit([&] { hipLaunchKernelGGL(k_quad, dim3(sets * 4), dim3(256), 0, 0, in, cells, out); });
        printf("%d x 256 particles per CU:  (a) one particle per lane %8.2f us   (q4) four lanes per particle %8.2f us   ratio q4 / a = %.2f%s\n",
               per_cu, a * 1e3 / REP, q * 1e3 / REP, q / a, per_cu == 1 ? "   <- the gate: bar 0.50" : "");
    }
    return 0;
}
