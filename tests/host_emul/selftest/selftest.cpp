// TEST INFRASTRUCTURE: self-test of the CPU interpreter (tests/host_emul/hipemu) on kernels small enough to state the expected result
// by hand -- the lane maps of every DPP control the product kernels use, shuffles, ballot with returned lanes, readlane /
// readfirstlane, the workgroup barrier with LDS, atomics from workgroups on several OS threads, and the two aborts (a divergent wave
// operation; a barrier not reached by every thread).  `selftest` prints "ok"; `selftest diverge` / `selftest deadlock` must abort.
#include <hip/hip_runtime.h>

#include <vector>

static int g_fail = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++g_fail; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

__global__ void k_dpp(int* out) {
    const int lane = threadIdx.x & 63, v = 1000 + lane;
    int k = 0;
#define ROW(ctrl, rowmask, bankmask, bc) out[(k++) * 64 + lane] = __builtin_amdgcn_update_dpp(-1, v, ctrl, rowmask, bankmask, bc)
    ROW(0xB1, 0xf, 0xf, true);       // 0 quad_perm [1,0,3,2]
    ROW(0x4E, 0xf, 0xf, true);       // 1 quad_perm [2,3,0,1]
    ROW(0x101, 0xf, 0xf, true);      // 2 row_shl:1, bound_ctrl: out of the row reads 0
    ROW(0x104, 0xf, 0xf, false);     // 3 row_shl:4, no bound_ctrl: keeps `old`
    ROW(0x111, 0xf, 0xf, false);     // 4 row_shr:1
    ROW(0x118, 0xf, 0xf, false);     // 5 row_shr:8
    ROW(0x124, 0xf, 0xf, true);      // 6 row_ror:4
    ROW(0x12C, 0xf, 0xf, true);      // 7 row_ror:12
    ROW(0x128, 0xf, 0xf, true);      // 8 row_ror:8
    ROW(0x142, 0xa, 0xf, false);     // 9 row_bcast:15 into rows 1 and 3
    ROW(0x143, 0xc, 0xf, false);     // 10 row_bcast:31 into rows 2 and 3
    ROW(0x140, 0xf, 0xf, true);      // 11 row_mirror
    ROW(0x111, 0xf, 0x5, false);     // 12 row_shr:1, banks 0 and 2 only (lanes 4k+0..3 -> bank (lane >> 2) & 3)
#undef ROW
}

__global__ void k_wave(int* out) {
    const int lane = threadIdx.x & 63;
    out[0 * 64 + lane] = __shfl_xor(lane * 3, 16);
    out[1 * 64 + lane] = __shfl_up(lane, 1);
    out[2 * 64 + lane] = __shfl(lane + 7, 63 - lane);
    const unsigned long long b = __ballot(lane % 3 == 0);
    out[3 * 64 + lane] = (int)(b & 0xffffffffu);
    out[4 * 64 + lane] = (int)(b >> 32);
    if (lane >= 48) return;                                   // the upper quarter leaves: exec mask = lanes 0..47 from here on
    const unsigned long long b2 = __ballot(1);
    out[5 * 64 + lane] = (int)(b2 >> 32);                     // 0x0000ffff
    out[6 * 64 + lane] = __builtin_amdgcn_readlane(lane * 2, 5);
    out[7 * 64 + lane] = __builtin_amdgcn_readfirstlane(lane + 100);
    out[8 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x101, 0xf, 0xf, true);      // lane 47 reads lane 48, which has returned: 0
    out[9 * 64 + lane] = __any(lane == 47) + 2 * __all(lane < 48);
}

__global__ void k_block(int* out, int n) {
    __shared__ int part[4];
    __shared__ int total;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    int v = blockIdx.x * 256 + t < n ? 1 : 0;
    for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);    // wave sum, every lane
    if (lane == 0) part[wave] = v;
    __syncthreads();
    if (t == 0) total = part[0] + part[1] + part[2] + part[3];
    __syncthreads();
    if (t == 0) atomicAdd(out, total);
    if (lane == 1) atomicAdd(out + 1, 1);                      // 4 per workgroup
    float* f = reinterpret_cast<float*>(out + 2);
    atomicAdd(f, 0.5f);                                       // every thread: floating-point atomics from all OS threads at once
}

__global__ void k_diverge(int* out) {
    const int lane = threadIdx.x & 63;
    if (lane & 1) out[lane] = __shfl_xor(lane, 2);            // half the wave at a shuffle ...
    else out[lane] = (int)__ballot(1);                        // ... the other half at a ballot
}
__global__ void k_deadlock(int* out) {
    // half a wave at the workgroup barrier, the other half at a wave operation: each waits for the other
    if ((threadIdx.x & 63) < 32) __syncthreads();
    else out[threadIdx.x] = (int)__ballot(1);
    out[0] = 1;
}

// (for the ThreadSanitizer build: a forgotten barrier -- thread t + 1 reads what thread t wrote into LDS with nothing in between)
__global__ void k_race(int* out) {
    __shared__ int x[256];
    x[threadIdx.x] = threadIdx.x;
    out[threadIdx.x] = x[(threadIdx.x + 1) & 255];
}

int main(int argc, char** argv) {
    int* out = nullptr;
    hipMalloc(&out, 64 * 64 * sizeof(int));
    if (argc > 1 && !strcmp(argv[1], "race")) { hipLaunchKernelGGL(k_race, dim3(1), dim3(256), 0, nullptr, out); printf("ran\n"); return 0; }
    if (argc > 1 && !strcmp(argv[1], "diverge")) { hipLaunchKernelGGL(k_diverge, dim3(1), dim3(64), 0, nullptr, out); return 0; }
    if (argc > 1 && !strcmp(argv[1], "deadlock")) { hipLaunchKernelGGL(k_deadlock, dim3(1), dim3(256), 0, nullptr, out); return 0; }

    hipLaunchKernelGGL(k_dpp, dim3(1), dim3(64), 0, nullptr, out);
    for (int l = 0; l < 64; ++l) {
        const int row = l & ~15, r = l & 15, V = 1000;
        CHECK(out[0 * 64 + l] == V + (l ^ 1), "quad_perm 1032 lane %d: %d", l, out[0 * 64 + l]);
        CHECK(out[1 * 64 + l] == V + (l ^ 2), "quad_perm 2301 lane %d: %d", l, out[1 * 64 + l]);
        CHECK(out[2 * 64 + l] == (r + 1 <= 15 ? V + l + 1 : 0), "row_shl:1 lane %d: %d", l, out[2 * 64 + l]);
        CHECK(out[3 * 64 + l] == (r + 4 <= 15 ? V + l + 4 : -1), "row_shl:4 lane %d: %d", l, out[3 * 64 + l]);
        CHECK(out[4 * 64 + l] == (r >= 1 ? V + l - 1 : -1), "row_shr:1 lane %d: %d", l, out[4 * 64 + l]);
        CHECK(out[5 * 64 + l] == (r >= 8 ? V + l - 8 : -1), "row_shr:8 lane %d: %d", l, out[5 * 64 + l]);
        CHECK(out[6 * 64 + l] == V + row + ((r - 4) & 15), "row_ror:4 lane %d: %d", l, out[6 * 64 + l]);
        CHECK(out[7 * 64 + l] == V + row + ((r - 12) & 15), "row_ror:12 lane %d: %d", l, out[7 * 64 + l]);
        CHECK(out[8 * 64 + l] == V + (l ^ 8), "row_ror:8 = xor 8 lane %d: %d", l, out[8 * 64 + l]);
        CHECK(out[9 * 64 + l] == (((l >> 4) & 1) ? V + row - 1 : -1), "row_bcast:15 lane %d: %d", l, out[9 * 64 + l]);
        CHECK(out[10 * 64 + l] == (l >= 32 ? V + 31 : -1), "row_bcast:31 lane %d: %d", l, out[10 * 64 + l]);
        CHECK(out[11 * 64 + l] == V + row + 15 - r, "row_mirror lane %d: %d", l, out[11 * 64 + l]);
        const bool bank_on = (((l >> 2) & 3) == 0) || (((l >> 2) & 3) == 2);
        CHECK(out[12 * 64 + l] == (bank_on && r >= 1 ? V + l - 1 : -1), "row_shr:1 bank_mask 0x5 lane %d: %d", l, out[12 * 64 + l]);
    }

    hipLaunchKernelGGL(k_wave, dim3(1), dim3(64), 0, nullptr, out);
    unsigned long long want = 0;
    for (int l = 0; l < 64; ++l) if (l % 3 == 0) want |= 1ULL << l;
    for (int l = 0; l < 64; ++l) {
        CHECK(out[0 * 64 + l] == (l ^ 16) * 3, "shfl_xor lane %d", l);
        CHECK(out[1 * 64 + l] == (l ? l - 1 : 0), "shfl_up lane %d", l);
        CHECK(out[2 * 64 + l] == 63 - l + 7, "shfl lane %d", l);
        CHECK((unsigned)out[3 * 64 + l] == (unsigned)(want & 0xffffffffu) && (unsigned)out[4 * 64 + l] == (unsigned)(want >> 32), "ballot lane %d", l);
        if (l < 48) {
            CHECK(out[5 * 64 + l] == 0xffff, "ballot after a partial return, lane %d: 0x%x", l, out[5 * 64 + l]);
            CHECK(out[6 * 64 + l] == 10 && out[7 * 64 + l] == 100, "readlane / readfirstlane lane %d", l);
            CHECK(out[8 * 64 + l] == ((l & 15) == 15 ? 0 : l + 1), "row_shl:1 next to returned lanes, lane %d: %d", l, out[8 * 64 + l]);
            CHECK(out[9 * 64 + l] == 3, "any / all over the live lanes, lane %d: %d", l, out[9 * 64 + l]);
        }
    }

    const int n = 200 * 256 - 77, nwg = 200;
    hipMemset(out, 0, 16);
    hipLaunchKernelGGL(k_block, dim3(nwg), dim3(256), 0, nullptr, out, n);
    CHECK(out[0] == n, "block sums: %d, want %d", out[0], n);
    CHECK(out[1] == 4 * nwg, "integer atomics: %d", out[1]);
    CHECK(*reinterpret_cast<float*>(out + 2) == 0.5f * 256 * nwg, "float atomics: %g", *reinterpret_cast<float*>(out + 2));
    if (g_fail) { printf("%d failures\n", g_fail); return 1; }
    printf("ok\n");
    return 0;
}
