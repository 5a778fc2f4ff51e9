// Microbenchmark (round 4): could the matrix pipe do the scatter's stencil contraction?
//
// (a) the product's form: one particle per lane, lanes grouped by cell; per stencil node 4 values (mass, 3 momentum
//     components, affine in the node offset) are multiplied by the node's weight, summed over the lanes of a run with 4
//     v_fmac_f32_dpp steps each (432 DPP per wave), converted and added to an f64 LDS tile by the run's head lane (108 ds_add_f64).
// (b) a LOWER BOUND of the cheapest MFMA form we found: out[(i,j)][(l,c)] = sum_p A[(i,j)][p] B[p][(l,c)] per cell with
//     A = wx_i wy_j (9 rows -> one 16-row M tile), B = wz_l x {m, q0, ax, ay, az}[c] (39 columns -> three 16-column N tiles),
//     K = the cell's particles in groups of 4 (v_mfma_f32_16x16x4_f32).  The operands are per-particle values that live in
//     ONE lane's registers and must reach the lanes (k, m) / (k, n) of the fragment: one trip through LDS (12 ds_write_b128 per
//     lane, 4 ds_read_b32 per K-group).  Counted here: the 48 per-particle products, the LDS transpose, 3 MFMA per K-group,
//     and per cell the affine recombination of the 39 columns into 4 components (27 DPP multiply-adds), 12 converts and
//     12 ds_add_f64 of the three result tiles.  NOT counted (so the real thing costs more): run bookkeeping (variable K-group
//     counts per cell are made wave-uniform by padding), zero-padding of ragged groups, address arithmetic of the tile.
// Cells per wave and particles per cell are those of the benchmark: ~9 runs per wave after row clipping, 64 particles.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scatter_mfma_bound.hip -o scatter_mfma_bound.bin && ./scatter_mfma_bound.bin
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int REP = 64;          // scatters per wave per launch (amortises the launch)

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

// (a) -------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 4) void k_dpp(const float* in, float* out) {
    __shared__ double tile[1024 * 4];
    for (int i = threadIdx.x; i < 4096; i += 256) tile[i] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, t = blockIdx.x * 256 + threadIdx.x;
    // run structure: a head every ~7 lanes (ragged), clipped at the 16-lane rows
    const int runlen = 5 + (lane * 7 + (lane >> 4)) % 5;
    const bool head = (lane & 15) == 0 || (lane % runlen) == 0;
    const unsigned long long heads = __ballot(head);
    const unsigned long long higher = lane == 63 ? 0ull : (heads >> (lane + 1));
    const int end = higher ? lane + __ffsll((long long)higher) - 1 : 63;
    const float m1 = lane + 1 <= end, m2 = lane + 2 <= end, m4 = lane + 4 <= end, m8 = lane + 8 <= end;
    float w[3][3], q0[3], ax[3], ay[3], az[3];
    for (int k = 0; k < 9; ++k) w[k / 3][k % 3] = in[t * 24 + k];
    for (int k = 0; k < 3; ++k) { q0[k] = in[t * 24 + 9 + k]; ax[k] = in[t * 24 + 12 + k]; ay[k] = in[t * 24 + 15 + k]; az[k] = in[t * 24 + 18 + k]; }
    const int cell = (lane / 7) * 3 + (blockIdx.x & 7);
    for (int r = 0; r < REP; ++r) {
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
                        double* q = tile + 4 * ((cell + l * 100 + j * 10 + i + r) & 1023);
                        atomicAdd(q, (double)a0); atomicAdd(q + 1, (double)a1); atomicAdd(q + 2, (double)a2); atomicAdd(q + 3, (double)a3);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = (float)tile[threadIdx.x * 4 + 1];
}

// (b) lower bound ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 4) void k_mfma(const float* in, float* out) {
    __shared__ double tile[1024 * 4];
    __shared__ float stage[4][64 * 48];                    // per wave: 64 particles x 48 operand values
    for (int i = threadIdx.x; i < 4096; i += 256) tile[i] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, t = blockIdx.x * 256 + threadIdx.x;
    float w[3][3], c13[13];
    for (int k = 0; k < 9; ++k) w[k / 3][k % 3] = in[t * 24 + k];
    for (int k = 0; k < 13; ++k) c13[k] = in[t * 24 + 9 + k];
    float* st = stage[wave];
    const int kq = lane >> 4, mn = lane & 15;
    for (int r = 0; r < REP; ++r) {
        // per-particle operand values: A = wx_i wy_j (9), B = wz_l x coefficient (39): 48 products, one particle per lane
        float v[48];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) v[i * 3 + j] = w[i][0] * w[j][1];
        for (int l = 0; l < 3; ++l) for (int c = 0; c < 13; ++c) v[9 + l * 13 + c] = w[l][2] * c13[c];
        // transpose through LDS: particle-major rows (12 x 16-byte stores per lane)
        for (int k = 0; k < 12; ++k) *reinterpret_cast<v4f*>(st + lane * 48 + 4 * k) = v4f{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
        __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): own wave's stores
        // ~9 cells per wave, 2.5 K-groups of 4 particles each on average: 22 groups, 3 N tiles
#pragma unroll 1
        for (int cellr = 0; cellr < 9; ++cellr) {
            v4f acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
            const int groups = 2 + (cellr & 1);            // 2, 3, 2, 3, ... = 22 per wave
            for (int g = 0; g < groups; ++g) {
                const int p = (cellr * 7 + g * 4 + kq) & 63;
                const float a = st[p * 48 + (mn < 9 ? mn : 0)];
                const float b0 = st[p * 48 + 9 + mn], b1 = st[p * 48 + 25 + (mn < 7 ? mn : 0)], b2 = st[p * 48 + 32 + mn];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b2, acc2, 0, 0, 0);
            }
            // affine recombination of the 39 columns into {m, mv_x, mv_y, mv_z} per node: 27 DPP multiply-adds per cell
            float o = (float)(mn & 3);
            for (int k = 0; k < 4; ++k) {
                acc0[k] += o * row_shl<4>(acc0[k]) + o * row_shl<8>(acc0[k]);
                acc1[k] += o * row_shl<4>(acc1[k]) + o * row_shl<8>(acc1[k]);
                acc2[k] += o * row_shl<4>(acc2[k]) + o * row_shl<8>(acc2[k]);
            }
            if (mn < 4 && kq < 3)                          // rows (i, j) 0..8 live in 3 of the 4 row groups; 12 adds per cell
                for (int k = 0; k < 4; ++k) {
                    double* q = tile + 4 * ((cellr * 3 + kq * 40 + k * 10 + r) & 1023) + mn;
                    atomicAdd(q, (double)acc0[k]); atomicAdd(q + 400, (double)acc1[k]); atomicAdd(q + 800, (double)acc2[k]);
                }
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = (float)tile[threadIdx.x * 4 + 1];
}

template <class F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    const int wgs = 1024;                                   // 4 workgroups per CU, as the scatter kernels run
    float *in, *out;
    hipMalloc(&in, (size_t)wgs * 256 * 24 * 4); hipMalloc(&out, wgs * 64 * 4);
    float* h = new float[(size_t)wgs * 256 * 24];
    for (size_t i = 0; i < (size_t)wgs * 256 * 24; ++i) h[i] = 0.1f + 0.001f * (float)(i % 977);
    hipMemcpy(in, h, (size_t)wgs * 256 * 24 * 4, hipMemcpyHostToDevice);
    const float a = timeit([&] { hipLaunchKernelGGL(k_dpp, dim3(wgs), dim3(256), 0, 0, in, out); });
    const float b = timeit([&] { hipLaunchKernelGGL(k_mfma, dim3(wgs), dim3(256), 0, 0, in, out); });
    // per wave and scatter, with all 16 waves of a CU busy: ms * 1e6 ns / REP / (waves per CU = 16)
    printf("(a) DPP segmented reduction + 108 ds_add_f64 : %8.3f ms = %7.1f ns per 64-particle scatter per CU\n", a, a * 1e6 / REP / 16.0);
    printf("(b) MFMA form, lower bound                   : %8.3f ms = %7.1f ns per 64-particle scatter per CU\n", b, b * 1e6 / REP / 16.0);
    printf("ratio (a) / (b) = %.2f  (the bar for building it was 2)\n", a / b);
    return 0;
}
