// Microbenchmarks behind the round-2 design decisions (MI355X, gfx950):
//  A. copy peak of this box (float4 copy of 1 GiB): the "measured" HBM roof bench.py reports next to the 8 TB/s spec.
//  B. cost of a dependent kernel boundary behind a kernel that leaves B bytes of freshly written data in the L2s,
//     with plain / nontemporal / sc1 (write-through) stores: is the ~4.6 us per boundary of the substep an L2
//     write-back at kernel end?
//  C. issue cost per wave-instruction of the VALU / LDS instructions the scatter loops are made of.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <class F> float timeit(F f, int reps = 20) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / reps;
}

// ---- A
__global__ __launch_bounds__(256) void k_copy(const f4* __restrict__ in, f4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
__global__ __launch_bounds__(256) void k_read(const f4* __restrict__ in, float* __restrict__ out, size_t n) {
    f4 s = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += in[i];
    if (s.x + s.y + s.z + s.w == -1.f) out[0] = s.x;
}

// ---- B: MODE 0 plain, 1 nontemporal, 2 sc1 (write-through), 3 sc0 sc1
template <int MODE> __global__ __launch_bounds__(256) void k_write(f4* __restrict__ out, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        f4 val = {v, v + 1, v + 2, (float)i};
        if (MODE == 0) out[i] = val;
        else if (MODE == 1) __builtin_nontemporal_store(val, &out[i]);
        else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&out[i]), "v"(val) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(&out[i]), "v"(val) : "memory");
    }
}
__global__ void k_touch(const float* __restrict__ in, float* __restrict__ out) {
    if (threadIdx.x == 0 && in[blockIdx.x * 1024] == -123.f) out[blockIdx.x] = 1.f;
}

// ---- C: per-wave instruction issue cost (all SIMDs busy: 8 waves per SIMD, long unrolled chains)
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int KIND> __global__ __launch_bounds__(256) void k_valu(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = 0.999f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pm = {m, m};
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {        // 64 x 8 independent v_fma_f32
            REP64(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                               "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (KIND == 1) { // 64 x 4 v_pk_fma_f32 (8 flops-lanes each) + nothing else
            REP64(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                               "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));)
        } else if (KIND == 2) { // 64 x 8 v_fmac_f32_dpp row_shl:1
            REP64(asm volatile("v_fmac_f32_dpp %0, %0, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %1, %1, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_fmac_f32_dpp %2, %2, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %3, %3, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_fmac_f32_dpp %4, %4, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %5, %5, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_fmac_f32_dpp %6, %6, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %7, %7, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (KIND == 3) { // 64 x 8 v_fma_f64 (4 chains, twice)
            REP64(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n"
                               "v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)m));)
        } else if (KIND == 4) { // 64 x 8 v_cvt_f64_f32
            REP64(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n"
                               "v_cvt_f64_f32 %0, %5\n v_cvt_f64_f32 %1, %6\n v_cvt_f64_f32 %2, %7\n v_cvt_f64_f32 %3, %4\n"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
        } else if (KIND == 5) { // 64 x 8 v_mul_f32 (plain)
            REP64(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                               "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (r == -1.2345f) out[threadIdx.x] = r;
}
// LDS f64 atomics with `active` lanes per wave (distinct addresses), 512 per wave per iteration
__global__ __launch_bounds__(256) void k_ldsatom(float* out, int iters, int active) {
    __shared__ double tile[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) tile[i] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < active)
        for (int it = 0; it < iters; ++it)
#pragma unroll 16
            for (int k = 0; k < 512; ++k) atomicAdd(&tile[(wave * 1024 + lane * 16 + (k & 15) + ((k >> 4) & 3) * 256) & 4095], 1.0);
    __syncthreads();
    if (tile[threadIdx.x] == -1.0) out[0] = 1.f;
}

int main() {
    float* scratch; (void)hipMalloc(&scratch, 1 << 20);
    // ---- A
    {
        const size_t bytes = (size_t)1 << 30, n = bytes / 16;
        f4 *in, *out; (void)hipMalloc(&in, bytes); (void)hipMalloc(&out, bytes);
        (void)hipMemset(in, 0, bytes);
        for (int wg : {2048, 4096, 8192, 16384}) {
            float c = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(wg), dim3(256), 0, 0, in, out, n); }, 10);
            float r = timeit([&] { hipLaunchKernelGGL(k_read, dim3(wg), dim3(256), 0, 0, in, scratch, n); }, 10);
            printf("A copy_peak wg=%5d: copy %.3f ms = %.0f GB/s (read+write)   read-only %.3f ms = %.0f GB/s\n", wg, c, 2.0 * bytes / c * 1e-6, r,
                   1.0 * bytes / r * 1e-6);
        }
        (void)hipFree(in); (void)hipFree(out);
    }
    // ---- B
    {
        const size_t maxb = (size_t)256 << 20;
        f4* buf; (void)hipMalloc(&buf, maxb);
        for (size_t mb : {1, 8, 32, 100}) {
            const size_t n = (mb << 20) / 16;
            const int wg = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
            float t[4], tp[4];
            t[0] = timeit([&] { hipLaunchKernelGGL(k_write<0>, dim3(wg), dim3(256), 0, 0, buf, n, 1.f); });
            t[1] = timeit([&] { hipLaunchKernelGGL(k_write<1>, dim3(wg), dim3(256), 0, 0, buf, n, 1.f); });
            t[2] = timeit([&] { hipLaunchKernelGGL(k_write<2>, dim3(wg), dim3(256), 0, 0, buf, n, 1.f); });
            t[3] = timeit([&] { hipLaunchKernelGGL(k_write<3>, dim3(wg), dim3(256), 0, 0, buf, n, 1.f); });
            tp[0] = timeit([&] { hipLaunchKernelGGL(k_write<0>, dim3(wg), dim3(256), 0, 0, buf, n, 1.f); hipLaunchKernelGGL(k_touch, dim3(256), dim3(64), 0, 0, (const float*)buf, scratch); });
            tp[1] = timeit([&] { hipLaunchKernelGGL(k_write<1>, dim3(wg), dim3(256), 0, 0, buf, n, 1.f); hipLaunchKernelGGL(k_touch, dim3(256), dim3(64), 0, 0, (const float*)buf, scratch); });
            tp[2] = timeit([&] { hipLaunchKernelGGL(k_write<2>, dim3(wg), dim3(256), 0, 0, buf, n, 1.f); hipLaunchKernelGGL(k_touch, dim3(256), dim3(64), 0, 0, (const float*)buf, scratch); });
            tp[3] = timeit([&] { hipLaunchKernelGGL(k_write<3>, dim3(wg), dim3(256), 0, 0, buf, n, 1.f); hipLaunchKernelGGL(k_touch, dim3(256), dim3(64), 0, 0, (const float*)buf, scratch); });
            printf("B write %3zu MB  back-to-back writes us: plain %.1f nt %.1f sc1 %.1f sc0sc1 %.1f | write+dependent tiny kernel us: plain %.1f nt %.1f sc1 %.1f sc0sc1 %.1f\n",
                   mb, t[0] * 1e3, t[1] * 1e3, t[2] * 1e3, t[3] * 1e3, tp[0] * 1e3, tp[1] * 1e3, tp[2] * 1e3, tp[3] * 1e3);
        }
        float e = timeit([&] { hipLaunchKernelGGL(k_touch, dim3(256), dim3(64), 0, 0, (const float*)buf, scratch); hipLaunchKernelGGL(k_touch, dim3(256), dim3(64), 0, 0, (const float*)buf, scratch); });
        printf("B two dependent tiny kernels: %.1f us\n", e * 1e3);
        (void)hipFree(buf);
    }
    // ---- C: 256 CUs x 8 workgroups x 4 waves = 8 waves per SIMD
    {
        const int iters = 40, wg = 256 * 8;
        const char* names[6] = {"v_fma_f32", "v_pk_fma_f32", "v_fmac_f32_dpp", "v_fma_f64", "v_cvt_f64_f32", "v_mul_f32"};
        float t[6];
        t[0] = timeit([&] { hipLaunchKernelGGL(k_valu<0>, dim3(wg), dim3(256), 0, 0, scratch, iters, 1.f); }, 5);
        t[1] = timeit([&] { hipLaunchKernelGGL(k_valu<1>, dim3(wg), dim3(256), 0, 0, scratch, iters, 1.f); }, 5);
        t[2] = timeit([&] { hipLaunchKernelGGL(k_valu<2>, dim3(wg), dim3(256), 0, 0, scratch, iters, 1.f); }, 5);
        t[3] = timeit([&] { hipLaunchKernelGGL(k_valu<3>, dim3(wg), dim3(256), 0, 0, scratch, iters, 1.f); }, 5);
        t[4] = timeit([&] { hipLaunchKernelGGL(k_valu<4>, dim3(wg), dim3(256), 0, 0, scratch, iters, 1.f); }, 5);
        t[5] = timeit([&] { hipLaunchKernelGGL(k_valu<5>, dim3(wg), dim3(256), 0, 0, scratch, iters, 1.f); }, 5);
        // instructions per SIMD = 8 waves x iters x 64 x 8
        const double per_simd = 8.0 * iters * 64 * 8;
        for (int k = 0; k < 6; ++k)
            printf("C %-16s %.3f ms -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", names[k], t[k], t[k] * 1e6 / per_simd,
                   t[k] * 1e6 / per_simd * 2.4);
        for (int active : {1, 8, 16, 32, 64}) {
            float a = timeit([&] { hipLaunchKernelGGL(k_ldsatom, dim3(256 * 4), dim3(256), 0, 0, scratch, 8, active); }, 5);
            // per CU: 4 workgroups x 4 waves x 8 x 512 instructions
            printf("C ds_add_f64 active=%2d: %.3f ms -> %.2f ns per wave-instruction per CU\n", active, a, a * 1e6 / (16.0 * 8 * 512));
        }
    }
    return 0;
}
