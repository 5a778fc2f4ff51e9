// Round 3: does a packed fp32 instruction (v_pk_fma_f32) need more independent work than a plain one to keep a SIMD busy?
// Chains of DEPENDENT instructions (CH independent chains per wave), W waves per SIMD on every SIMD of the chip:
// prints cycles per wave-instruction per SIMD (at the measured clock: duration x 2.4 GHz nominal).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int KIND, int CH> __global__ __launch_bounds__(64) void k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, m = 0.999f;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a3}, p3 = {a0, a2}, pm = {m, m};
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0 && CH == 1) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n" : "+v"(a0) : "v"(m));) }
        if (KIND == 0 && CH == 2) { REP64(asm volatile("v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2\n v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2\n" : "+v"(a0), "+v"(a1) : "v"(m));) }
        if (KIND == 0 && CH == 4) { REP64(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));) }
        if (KIND == 1 && CH == 1) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n" : "+v"(p0) : "v"(pm));) }
        if (KIND == 1 && CH == 2) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2\n v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2\n" : "+v"(p0), "+v"(p1) : "v"(pm));) }
        if (KIND == 1 && CH == 4) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));) }
        if (KIND == 2 && CH == 1) { REP64(asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n" : "+v"(p0) : "v"(pm));) }
        if (KIND == 2 && CH == 4) { REP64(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));) }
        if (KIND == 3 && CH == 4) { REP64(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));) }
        if (KIND == 4 && CH == 4) { REP64(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));) }
        if (KIND == 5 && CH == 4) { REP64(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");) }
        if (KIND == 6 && CH == 4) { REP64(asm volatile("v_pk_mov_b32 %0, %0, %4\n v_pk_mov_b32 %1, %1, %4\n v_pk_mov_b32 %2, %2, %4\n v_pk_mov_b32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));) }
    }
    float r = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (r == -1.f) out[0] = r;
}
template <int KIND, int CH> void run(const char* name, float* d) {
    int dev = 0; hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, dev);
    const int cus = pr.multiProcessorCount, iters = 200;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    printf("%-34s", name);
    for (int w = 1; w <= 8; w *= 2) {                 // waves per SIMD (4 SIMDs per CU, one wave per workgroup)
        const int grid = cus * 4 * w;
        hipLaunchKernelGGL((k<KIND, CH>), dim3(grid), dim3(64), 0, 0, d, 10, 1.f);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k<KIND, CH>), dim3(grid), dim3(64), 0, 0, d, iters, 1.f);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        const double instr = (double)iters * 64 * 4 * w;       // wave-instructions per SIMD
        printf("  W=%d: %6.2f", w, ms * 1e-3 * 2.4e9 / instr);
    }
    printf("   cycles/instr/SIMD @2.4GHz\n");
}
int main() {
    float* d; (void)hipMalloc(&d, 1024);
    run<0, 1>("v_fma_f32     1 dependent chain", d);
    run<0, 2>("v_fma_f32     2 chains", d);
    run<0, 4>("v_fma_f32     4 chains", d);
    run<1, 1>("v_pk_fma_f32  1 dependent chain", d);
    run<1, 2>("v_pk_fma_f32  2 chains", d);
    run<1, 4>("v_pk_fma_f32  4 chains", d);
    run<2, 1>("v_pk_mul_f32  1 dependent chain", d);
    run<2, 4>("v_pk_mul_f32  4 chains", d);
    run<3, 4>("v_pk_add_f32  4 chains", d);
    run<4, 4>("v_mul_f32     4 chains", d);
    run<5, 4>("v_cndmask_b32 4 chains", d);
    run<6, 4>("v_pk_mov_b32  4 chains", d);
    return 0;
}
