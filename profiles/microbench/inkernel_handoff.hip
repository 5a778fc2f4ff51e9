// One launch, two roles: the first NHEAD workgroups PRODUCE a buffer (like grid_op writing grid_v_out), the others CONSUME
// it (like a particle kernel's tile fill) after spinning on a counter.  Questions: (1) what must producer / consumer do for
// the consumers -- on other XCDs, with their own L2 -- to read fresh data, after having READ THE OLD CONTENTS of the same
// lines in the same kernel (worst case for staleness); (2) how long after the producers finish do the consumers get going.
// hipcc --offload-arch=gfx950 -O3 inkernel_handoff.hip -o inkernel_handoff && ./inkernel_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float __attribute__((ext_vector_type(4))) vec16;

// PROD: 0 plain stores + agent release fence by one thread per workgroup | 1 write-through stores (sc0 sc1) + s_waitcnt
// CONS: 0 nothing | 1 agent acquire fence after the wait | 2 nothing, but loads with sc1 (via inline asm, waited one by one)
template <int PROD, int CONS>
__global__ __launch_bounds__(256) void k(vec16* buf, size_t nv, int nhead, unsigned* counter, unsigned expect, float val, unsigned* stale, long long* tstamp) {
    if ((int)blockIdx.x < nhead) {
        const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)nhead * 256;
        for (size_t j = tid; j < nv; j += nth) {
            vec16 v = {val, val, val, val};
            if (PROD == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(buf + j), "v"(v) : "memory");
            else buf[j] = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (PROD == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == expect) tstamp[0] = wall_clock64();
        }
        return;
    }
    // consumer: touch the OLD contents first (they are now in this CU's L1 / this XCD's L2), then wait, then read again
    const size_t ctid = (size_t)(blockIdx.x - nhead) * 256 + threadIdx.x, cnth = (size_t)(gridDim.x - nhead) * 256;
    float acc = 0;
    for (size_t j = ctid; j < nv; j += cnth) acc += buf[j].x;
    if (acc == 1234567.f) stale[1] = 1;
    if (threadIdx.x == 0) {
        while ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - expect) < 0) __builtin_amdgcn_s_sleep(1);
        if (blockIdx.x == nhead) tstamp[1] = wall_clock64();
    }
    __syncthreads();
    if (CONS == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    unsigned bad = 0;
    for (size_t j = ctid; j < nv; j += cnth) {
        vec16 v;
        if (CONS == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(buf + j) : "memory");
        else v = buf[j];
        bad += (v.x != val) + (v.w != val);
    }
    if (bad) atomicAdd(stale, bad);
}

template <int PROD, int CONS>
int run(const char* name, vec16* buf, size_t nv, unsigned* counter, unsigned* stale, long long* ts) {
    const int nhead = 128, ncons = 512, reps = 200;
    unsigned expect = 0;
    hipMemset(counter, 0, 4); hipMemset(stale, 0, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double gap = 0;
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) {
        expect += nhead;
        hipLaunchKernelGGL((k<PROD, CONS>), dim3(nhead + ncons), dim3(256), 0, 0, buf, nv, nhead, counter, expect, (float)(r + 1), stale, ts);
    }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned h[2]; long long t[2];
    hipMemcpy(h, stale, 8, hipMemcpyDeviceToHost); hipMemcpy(t, ts, 16, hipMemcpyDeviceToHost);
    gap = (t[1] - t[0]) * 10.0;      // 100 MHz clock -> ns (last launch)
    printf("%-86s %6.2f us per launch, stale values read: %u of %zu, last producer -> first consumer released: %.0f ns\n", name, 1e3 * ms / reps, h[0], (size_t)reps * nv * 2, gap);
    return 0;
}

int main() {
    const size_t bytes = 1242400 / 16 * 16;       // 77 650 active nodes x 16 B: one frame's grid_v_out
    const size_t nv = bytes / 16;
    vec16* buf; unsigned *counter, *stale; long long* ts;
    CHK(hipMalloc(&buf, bytes)); CHK(hipMalloc(&counter, 256)); CHK(hipMalloc(&stale, 256)); CHK(hipMalloc(&ts, 256));
    CHK(hipMemset(buf, 0, bytes));
    run<0, 0>("plain stores + agent release per producer workgroup | consumer: no fence", buf, nv, counter, stale, ts);
    run<0, 1>("plain stores + agent release per producer workgroup | consumer: agent acquire fence", buf, nv, counter, stale, ts);
    run<1, 0>("write-through stores (sc0 sc1) + s_waitcnt          | consumer: no fence", buf, nv, counter, stale, ts);
    run<1, 1>("write-through stores (sc0 sc1) + s_waitcnt          | consumer: agent acquire fence", buf, nv, counter, stale, ts);
    run<1, 2>("write-through stores (sc0 sc1) + s_waitcnt          | consumer: sc1 loads", buf, nv, counter, stale, ts);
    run<0, 2>("plain stores + agent release per producer workgroup | consumer: sc1 loads", buf, nv, counter, stale, ts);
    return 0;
}
