// Round 6: cycles per wave-instruction per SIMD at the REAL shader clock, for the instruction classes the
// particle kernels issue.  Every wave brackets its instruction stream with s_memtime (shader cycles) and s_memrealtime (100 MHz):
// cycles come from the first, the clock the stream actually ran at from the ratio of the two -- no nominal frequency anywhere.
// W workgroups of 256 threads per CU = W waves per SIMD (a workgroup's 4 waves go to the 4 SIMDs of one CU); the placement is
// checked from HW_ID.  Streams are 4 independent chains unless the name says otherwise, 256 instructions per loop iteration.
//   build: hipcc --offload-arch=gfx950 -O2 valu_calibration.hip -o valu_calibration.bin
//   run:   ./valu_calibration.bin            (table)        ./valu_calibration.bin pmc   (one launch per stream at W = 4, for rocprofv3 --pmc)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <map>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define A4(ins, tail) asm volatile(ins " %0, %0" tail "\n" ins " %1, %1" tail "\n" ins " %2, %2" tail "\n" ins " %3, %3" tail "\n" \
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(m2), "v"(m3));
#define I4(ins, tail) asm volatile(ins " %0, %0" tail "\n" ins " %1, %1" tail "\n" ins " %2, %2" tail "\n" ins " %3, %3" tail "\n" \
                                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(im), "v"(im2));

struct Rec { unsigned long long cyc, rt; unsigned hwid, xcc; };

enum Kind {
    FMA_FWD1, FMA_2SRC, FMA_3SRC, FMAC, MUL, ADD, MAXF, FMAC_DPP, MOV_DPP, ADD_DPP, AND_DPP, CVT_F64_F32, CVT_F32_I32, ADD_U32, LSHL_ADD, MAD_U24, MUL_LO, AND_B32,
    CNDMASK, CMP_CND, LSHL_ADD_U64, ASHR_I32, MAD_U64_U32, BFE_U32, RCP, RSQ, SQRT, MOV, PK_FMA, PK_MUL, FMA_F64, ADD_F64, MIX_FMA_MUL, MIX_PRODUCT, BANK_SAME, BANK_DIFF, FMA_1W_DEP, NKIND
};
static const char* kname[NKIND] = {
    "v_fma_f32 d,d,m,m   ONE chain (result forwarded)", "v_fma_f32 d,d,m,m   (2 distinct VGPR sources)", "v_fma_f32 d,d,m,m2  (3 distinct VGPR sources)",
    "v_fmac_f32 d,m,m2   (2 sources + accumulator)", "v_mul_f32 d,d,m", "v_add_f32 d,d,m", "v_max_f32 d,d,m",
    "v_fmac_f32_dpp d,d,m row_shl:1 (product's reduce step)", "v_mov_b32_dpp d,d row_shl:1", "v_add_f32_dpp d,d,m row_shl:1", "v_and_b32_dpp d,d,m row_shl:1", "v_cvt_f64_f32", "v_cvt_f32_i32", "v_add_u32 d,d,m", "v_lshl_add_u32 d,d,2,m",
    "v_mad_u32_u24 d,d,m,m2", "v_mul_lo_u32 d,d,m", "v_and_b32 d,d,m", "v_cndmask_b32 d,d,m,vcc", "v_cmp_lt_f32 + v_cndmask pair (per pair)", "v_lshl_add_u64 d,d,2,m64 (64-bit address add)",
    "v_ashrrev_i32 d,31,d (sign extension)", "v_mad_u64_u32", "v_bfe_u32 d,d,m,5",
    "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_mov_b32 d,m", "v_pk_fma_f32 (two FMAs)", "v_pk_mul_f32 (two MULs)", "v_fma_f64", "v_add_f64",
    "mix: fma,mul,fma,add", "mix: product-like (10 fma, 4 mul, 2 add, 4 int, 2 cvt, 2 mov per 24)", "v_fma_f32, 3 sources in ONE VGPR bank (v8,v12,v16)",
    "v_fma_f32, 3 sources in 3 banks (v9,v14,v19)", "v_fma_f32 ONE chain, 1 wave/SIMD only (dependent-issue latency)"};

template <int KIND> __global__ __launch_bounds__(256) void k(Rec* rec, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, m = 0.999f + seed * 1e-9f, m2 = 1.001f + seed * 1e-9f, m3 = 0.5f + seed * 1e-9f;
    int i0 = threadIdx.x + (int)seed, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, im = 3 + (int)seed, im2 = 5 + (int)seed;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a3}, p3 = {a0, a2}, pm = {m, m};
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, dm = m;
    unsigned long long t0, t1, w0, w1;
    asm volatile("s_memrealtime %0\n s_memtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(w0), "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        if (KIND == FMA_FWD1 || KIND == FMA_1W_DEP) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n" : "+v"(a0) : "v"(m));) }
        if (KIND == FMA_2SRC) { REP64(A4("v_fma_f32", ", %4, %4")) }
        if (KIND == FMA_3SRC) { REP64(A4("v_fma_f32", ", %4, %5")) }
        if (KIND == FMAC) { REP64(asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(m2));) }
        if (KIND == MUL) { REP64(A4("v_mul_f32", ", %4")) }
        if (KIND == ADD) { REP64(A4("v_add_f32", ", %4")) }
        if (KIND == MAXF) { REP64(A4("v_max_f32", ", %4")) }
        if (KIND == FMAC_DPP) { REP64(A4("v_fmac_f32_dpp", ", %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")) }
        if (KIND == ADD_DPP) { REP64(A4("v_add_f32_dpp", ", %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")) }
        if (KIND == AND_DPP) { REP64(I4("v_and_b32_dpp", ", %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")) }
        if (KIND == MOV_DPP) { REP64(A4("v_mov_b32_dpp", " row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")) }
        if (KIND == CVT_F64_F32) { REP64(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
        if (KIND == CVT_F32_I32) { REP64(asm volatile("v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_f32_i32 %2, %6\n v_cvt_f32_i32 %3, %7\n" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(i0), "v"(i1), "v"(i2), "v"(i3));) }
        if (KIND == ADD_U32) { REP64(I4("v_add_u32", ", %4")) }
        if (KIND == LSHL_ADD) { REP64(I4("v_lshl_add_u32", ", 2, %4")) }
        if (KIND == MAD_U24) { REP64(I4("v_mad_u32_u24", ", %4, %5")) }
        if (KIND == MUL_LO) { REP64(I4("v_mul_lo_u32", ", %4")) }
        if (KIND == AND_B32) { REP64(I4("v_and_b32", ", %4")) }
        if (KIND == CNDMASK) { REP64(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");) }
        if (KIND == CMP_CND) { REP64(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc\n" : "+v"(a0), "+v"(a1) : "v"(m), "v"(m2) : "vcc");) }
        if (KIND == LSHL_ADD_U64) { REP64(asm volatile("v_lshl_add_u64 %0, %0, 2, %4\n v_lshl_add_u64 %1, %1, 2, %4\n v_lshl_add_u64 %2, %2, 2, %4\n v_lshl_add_u64 %3, %3, 2, %4\n" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm));) }
        if (KIND == ASHR_I32) { REP64(asm volatile("v_ashrrev_i32 %0, 31, %4\n v_ashrrev_i32 %1, 31, %5\n v_ashrrev_i32 %2, 31, %4\n v_ashrrev_i32 %3, 31, %5\n" : "=v"(i0), "=v"(i1), "=v"(i2), "=v"(i3) : "v"(im), "v"(im2));) }
        if (KIND == MAD_U64_U32) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(im), "v"(im2) : "vcc");) }
        if (KIND == BFE_U32) { REP64(I4("v_bfe_u32", ", %4, 5")) }
        if (KIND == RCP) { REP64(A4("v_rcp_f32", "")) }
        if (KIND == RSQ) { REP64(A4("v_rsq_f32", "")) }
        if (KIND == SQRT) { REP64(A4("v_sqrt_f32", "")) }
        if (KIND == MOV) { REP64(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4\n" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(m));) }
        if (KIND == PK_FMA) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));) }
        if (KIND == PK_MUL) { REP64(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));) }
        if (KIND == FMA_F64) { REP64(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm));) }
        if (KIND == ADD_F64) { REP64(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm));) }
        if (KIND == MIX_FMA_MUL) { REP64(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_mul_f32 %1, %1, %4\n v_fma_f32 %2, %2, %4, %5\n v_add_f32 %3, %3, %5\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(m2));) }
        if (KIND == MIX_PRODUCT) {          // 24 instructions in the dynamic proportions of k_g2p_p2g (r02_notes.md): x 11 = 264 per iteration (counted as such)
            REP8(asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_mul_f32 %2, %2, %8\n v_add_u32 %4, %4, %10\n v_fma_f32 %3, %3, %8, %9\n v_fmac_f32 %0, %8, %9\n"
                "v_cvt_f64_f32 %6, %1\n v_fma_f32 %2, %2, %8, %9\n v_mul_f32 %3, %3, %9\n v_lshl_add_u32 %5, %5, 2, %10\n v_fma_f32 %0, %0, %9, %8\n v_add_f32 %1, %1, %8\n"
                "v_mov_b32 %7, %2\n v_fma_f32 %3, %3, %8, %9\n v_fmac_f32 %1, %9, %8\n v_mul_f32 %0, %0, %8\n v_and_b32 %4, %4, %10\n v_fma_f32 %2, %2, %9, %8\n"
                "v_cvt_f32_i32 %7, %5\n v_add_f32 %3, %3, %9\n v_fmac_f32 %2, %8, %9\n v_mul_f32 %1, %1, %8\n v_add_u32 %5, %5, %10\n v_mov_b32 %7, %0\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "=v"(d0), "=v"(m3) : "v"(m), "v"(m2), "v"(im));)
            REP8(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %3, %2\n v_fma_f32 %1, %1, %3, %2\n v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n" : "+v"(a0), "+v"(a1) : "v"(m), "v"(m2));)
        }
        if (KIND == BANK_SAME) { REP64(asm volatile("v_fma_f32 v20, v8, v12, v16\n v_fma_f32 v21, v8, v12, v16\n v_fma_f32 v22, v8, v12, v16\n v_fma_f32 v23, v8, v12, v16\n" ::: "v8", "v12", "v16", "v20", "v21", "v22", "v23");) }
        if (KIND == BANK_DIFF) { REP64(asm volatile("v_fma_f32 v20, v9, v14, v19\n v_fma_f32 v21, v9, v14, v19\n v_fma_f32 v22, v9, v14, v19\n v_fma_f32 v23, v9, v14, v19\n" ::: "v9", "v14", "v19", "v20", "v21", "v22", "v23");) }
    }
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(w1) :: "memory");
    float r = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(d0 + d1 + d2 + d3) + (float)(i0 + i1 + i2 + i3) + m3;
    if ((threadIdx.x & 63) == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        Rec q; q.cyc = t1 - t0; q.rt = w1 - w0; q.hwid = hw; q.xcc = xcc;
        rec[blockIdx.x * 4 + (threadIdx.x >> 6)] = q;
    }
    if (r == -1.2345f) rec[0].cyc = (unsigned long long)r;
}

static int g_cus;
template <int KIND> double run_one(Rec* d, int w, int iters, double* ghz, double* spread, int* maxw) {
    const int grid = g_cus * w;
    std::vector<Rec> h(grid * 4);
    hipLaunchKernelGGL((k<KIND>), dim3(grid), dim3(256), 0, 0, d, 4, 1.f);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k<KIND>), dim3(grid), dim3(256), 0, 0, d, iters, 1.f);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d, h.size() * sizeof(Rec), hipMemcpyDeviceToHost);
    // waves per (xcc, se, cu, simd): HW_ID bits [3:0] wave, [5:4] simd, [11:8] cu, [12] sh, [15:13] se
    std::map<unsigned, int> per;
    double sc = 0, sr = 0, mn = 1e30, mx = 0;
    for (auto& q : h) {
        per[((q.xcc & 0xf) << 16) | (q.hwid & 0xff30)]++;
        sc += (double)q.cyc; sr += (double)q.rt; mn = std::min(mn, (double)q.cyc); mx = std::max(mx, (double)q.cyc);
    }
    int mw = 0;
    for (auto& e : per) mw = std::max(mw, e.second);
    *maxw = mw;
    const double per_iter = KIND == MIX_PRODUCT ? 8 * 24 + 8 * 8 : 256;
    const double instr = per_iter * iters, mean = sc / h.size();
    *ghz = sc / (sr * 10.0);                 // cycles per ns: s_memrealtime ticks are 10 ns
    *spread = (mx - mn) / mean;
    return mean / (instr * w);               // cycles per wave-instruction per SIMD
}
template <int KIND> void row(Rec* d, bool pmc) {
    printf("%-58s", kname[KIND]);
    const int ws[4] = {1, 2, 4, 8};
    for (int i = 0; i < 4; ++i) {
        if (pmc && ws[i] != 4) continue;
        if (KIND == FMA_1W_DEP && ws[i] != 1) continue;
        double ghz, sp; int mw;
        const double c = run_one<KIND>(d, ws[i], 200, &ghz, &sp, &mw);
        printf("  W=%d: %5.2f cyc @%4.2f GHz%s", ws[i], c, ghz, mw != ws[i] ? "(!placement)" : "");
    }
    printf("\n");
    fflush(stdout);
}
template <int K> struct All { static void go(Rec* d, bool pmc) { row<K>(d, pmc); All<K + 1>::go(d, pmc); } };
template <> struct All<NKIND> { static void go(Rec*, bool) {} };

int main(int argc, char** argv) {
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    g_cus = pr.multiProcessorCount;
    const bool pmc = argc > 1 && !strcmp(argv[1], "pmc");
    printf("# %s, %d CUs; cycles per wave-instruction per SIMD from s_memtime, clock = s_memtime / s_memrealtime (10 ns ticks); W waves per SIMD\n", pr.gcnArchName, g_cus);
    Rec* d; (void)hipMalloc(&d, sizeof(Rec) * g_cus * 8 * 4);
    All<0>::go(d, pmc);
    return 0;
}
