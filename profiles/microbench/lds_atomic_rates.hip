// Microbenchmark (round 4): what does one LDS atomic wave-instruction cost on gfx950, by flavour and by the number of active
// lanes, at bank-conflict-free addresses?  The scatter kernels issue 108 ds_add_f64 per wave with ~10 active lanes each
// (one head lane per cell run): ablating them (PLB_ABLATE=1) takes 11-13 us out of a 49 us kernel on identical inputs
// (profiles/r04_notes.md), so this instruction is the single most expensive thing in the forward kernel.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic_rates.hip -o lds_atomic_rates && ./lds_atomic_rates
// Every wave of 16 per CU (4 workgroups x 4 waves, as in k_g2p_p2g) issues ITER atomics; reported: ns per wave-instruction
// per CU (throughput with all 16 waves issuing) and the same for ONE wave per CU (latency-ish).
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int ITER = 4096;
enum { F32 = 0, F64 = 1, U32 = 2, U64 = 3, RMW64 = 4, F32x2 = 5, RMW128 = 6, RMW128PRE = 7 };

// lane -> node: active lanes take distinct nodes, 32 B apart (Vec4<double> tile) or 16 B apart (Vec4<float> tile)
template <int MODE, int ACT>
__global__ __launch_bounds__(256) void k(float* out, int waves_active) {
    __shared__ double buf[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // active lanes spread over the wave like run heads (every 64/ACT-th lane)
    const bool on = (lane % (64 / ACT)) == 0 && wave < waves_active;
    const int slot = lane / (64 / ACT);
    if (on) {
        for (int it = 0; it < ITER; it += 4) {
            // 4 components of one node, then the next node of the stencil: the access shape of the scatter loop
            const int node = (slot * 3 + (it >> 2) * 7 + wave * 64) & 255;
            if (MODE == RMW128 || MODE == RMW128PRE) {
                // the 4 components of a node as ONE float4 read-modify-write (a per-wave private tile needs no atomics: in one step of
                // the scatter loop the head lanes of a wave address different nodes).  PRE: the read is issued 16 dependent VALU
                // instructions (the DPP reduction's length) before its value is needed
                typedef float f4 __attribute__((ext_vector_type(4)));
                volatile f4* q = reinterpret_cast<volatile f4*>(buf) + node;
                f4 v = *q;
                float a = 1.0f + it;
                if (MODE == RMW128PRE) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) asm volatile("v_fmac_f32 %0, %0, %0" : "+v"(a));
                }
                v.x += a; v.y += a; v.z += a; v.w += a;
                *q = v;
                continue;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (MODE == F32) atomicAdd(reinterpret_cast<float*>(buf) + node * 4 + c, 1.0f);
                else if (MODE == F64) atomicAdd(buf + node * 4 + c, 1.0);
                else if (MODE == U32) atomicAdd(reinterpret_cast<unsigned*>(buf) + node * 4 + c, 1u);
                else if (MODE == U64) atomicAdd(reinterpret_cast<unsigned long long*>(buf) + node * 4 + c, 1ull);
                else if (MODE == RMW64) { volatile double* p = buf + node * 4 + c; *p = *p + 1.0; }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)buf[5];
}
template <class F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
template <int MODE, int ACT> void run(const char* name, float* out) {
    // 1024 workgroups = 4 per CU resident at once (LDS 16 KB each): one round
    const float all = timeit([&] { hipLaunchKernelGGL((k<MODE, ACT>), dim3(1024), dim3(256), 0, 0, out, 4); });
    const float one = timeit([&] { hipLaunchKernelGGL((k<MODE, ACT>), dim3(256), dim3(256), 0, 0, out, 1); });
    printf("%-10s active lanes %2d : 16 waves/CU %7.3f ms = %6.2f ns per wave-instruction per CU | 1 wave/CU %7.3f ms = %6.2f ns per instruction\n",
           name, ACT, all, all * 1e6 / (ITER * 16.0), one, one * 1e6 / ITER);
}
int main() {
    float* out; hipMalloc(&out, 1 << 20);
#define ROW(M, name) run<M, 1>(name, out); run<M, 8>(name, out); run<M, 16>(name, out); run<M, 32>(name, out); run<M, 64>(name, out);
    ROW(F64, "ds_add_f64") ROW(F32, "ds_add_f32") ROW(U32, "ds_add_u32") ROW(U64, "ds_add_u64") ROW(RMW64, "rd+wr b64") ROW(RMW128, "rmw b128/4") ROW(RMW128PRE, "rmw128+16v")
    return 0;
}
