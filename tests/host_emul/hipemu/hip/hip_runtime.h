// TEST INFRASTRUCTURE ONLY -- never loaded, linked or imported by plasticinelab_amd.
//
// A CPU interpreter for the subset of HIP that libplmpm.so is written in, so that the DEVICE SOURCE ITSELF
// (plasticinelab_amd/csrc/*.hip, plmpm_kernels.h: tiles, DPP sorts and segmented reductions, atomics, block flags, launch
// logic of the C ABI) can be executed and checked against the oracle in the CPU-only test tier, where no GPU exists.
// g++ compiles the unchanged .hip files against this header instead of <hip/hip_runtime.h> (tests/host_emul/Makefile ->
// tests/host_emul/libplmpm_emul.so); tests select that library explicitly (tests/emul_engine.py).  It is orders of magnitude
// slower than the GPU and is not a fallback: plasticinelab_amd/_lib.py only ever loads the hipcc build.
//
// Execution model: a launch runs its workgroups one after another (PLMPM_EMUL_THREADS=n: on n OS threads at once, and ALL of them at
// once -- one OS thread per workgroup -- when the launch has at most 16: kernels whose workgroups wait for each other, the halo
// exchange folded into the grid kernels, need every workgroup resident); the threads of a workgroup are fibers (one small stack
// each) scheduled round-robin on the workgroup's OS thread.  A fiber runs until it reaches a synchronisation point:
//   * __syncthreads / s_barrier: waits for every thread of the workgroup that has not returned yet;
//   * a wave-level operation (DPP move, shuffle, ballot, readlane): the 64 lanes of the wave deposit their operand, wait for
//     each other, and read their partners' -- lock-step semantics of a 64-wide wavefront with the exec mask = lanes that have
//     not returned.  Every live lane of the wave must reach the SAME operation (checked: a divergent collective aborts with a
//     message instead of silently reading stale registers as the hardware would).
// What is NOT reproduced: lock-step between two such points.  A lane runs ahead of its wave until the next wave-level operation or
// barrier, so a kernel in which one lane overwrites what the other lanes of its wave read earlier in program order, with nothing
// wave-wide in between, behaves differently here (k_clear_active did: lane 0 cleared the block flag the other 63 lanes were about
// to test; it now takes the flag through v_readfirstlane).
// DPP controls (quad_perm, row_shl / shr / ror, row_bcast15 / 31, row_mirror, row / bank masks, bound_ctrl) follow the CDNA3/4
// ISA manual.  Atomics are real atomics (workgroups on several OS threads, ranks in several processes).  `__shared__` is storage
// shared by the fibers of the running workgroup (per OS thread).  Inline assembly cannot be interpreted: the few asm helpers of plmpm_kernels.h have host definitions here
// (PLB_HOST_EMUL), and the fused v_fmac_f32_dpp forms fall back to their generic C++ templates (same sums, mul + add).
#pragma once
#ifndef PLB_HOST_EMUL
#define PLB_HOST_EMUL 1
#endif

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
// LDS: storage shared by the fibers of the running workgroup -- per OS thread, because with PLMPM_EMUL_THREADS the workgroups of a launch
// run on several OS threads at once (each thread one workgroup at a time)
#define __shared__ static thread_local
#define __launch_bounds__(...)
// dynamic LDS (plmpm_kernels.h: PLB_DYN_LDS): one static arena
#define PLB_DYN_LDS(type, name) type* name = reinterpret_cast<type*>(hipemu::dyn_lds_arena())

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { unsigned x, y, z, w; };
struct float4 { float x, y, z, w; };

// ---------------------------------------------------------------------------------------------- runtime API (host side)
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotSupported = 801, hipErrorUnknown = 999 };
typedef struct hipemu_stream* hipStream_t;
struct hipemu_event { double t_ms; };
typedef hipemu_event* hipEvent_t;
typedef struct hipemu_graph* hipGraph_t;
typedef struct hipemu_graph_exec* hipGraphExec_t;
struct hipIpcMemHandle_t { char reserved[64]; };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamCaptureModeThreadLocal = 1 };
enum { hipDeviceMallocFinegrained = 1, hipDeviceMallocUncached = 3, hipHostMallocMapped = 2, hipIpcMemLazyEnablePeerAccess = 1 };

namespace hipemu {
void* device_alloc(size_t bytes);
void device_free(void* p);
void* shared_alloc(size_t bytes);
bool shared_handle(void* p, char* handle64);
void* shared_open(const char* handle64);
void shared_close(void* p);
void external_wait();
void device_memset(void* p, int v, size_t n);
double now_ms();
}
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : e == hipErrorNotSupported ? "not supported by the CPU interpreter" : "error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
template <class P> inline hipError_t hipMalloc(P** p, size_t bytes) { *p = (P*)hipemu::device_alloc(bytes); return *p ? hipSuccess : hipErrorOutOfMemory; }
// hipExtMallocWithFlags (the library asks for it only for memory a NEIGHBOUR RANK maps: the receive areas of the device-side halo
// exchange) is POSIX shared memory here, and the IPC handle carries its name: ranks that are separate processes of one test exchange
// halos through it exactly as ranks on one GPU do through hipIpc -- peer writes, arrival counters, bounded waits and all.
template <class P> inline hipError_t hipExtMallocWithFlags(P** p, size_t bytes, unsigned) { *p = (P*)hipemu::shared_alloc(bytes); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class P> inline hipError_t hipHostMalloc(P** p, size_t bytes, unsigned = 0) { return hipMalloc(p, bytes); }
inline hipError_t hipFree(void* p) { hipemu::device_free(p); return hipSuccess; }
inline hipError_t hipHostFree(void* p) { hipemu::device_free(p); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { hipemu::device_memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { hipemu::device_memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event{0.0}; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t_ms = hipemu::now_ms(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t_ms - a->t_ms); return hipSuccess; }
// graphs: launches execute at once, so there is nothing to capture -- the one caller (plmpm_replay, a measurement tool) gets an error
inline hipError_t hipStreamBeginCapture(hipStream_t, int) { return hipErrorNotSupported; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*) { return hipErrorNotSupported; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return hipErrorNotSupported; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p) { return hipemu::shared_handle(p, h->reserved) ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned) { *p = hipemu::shared_open(h.reserved); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipIpcCloseMemHandle(void* p) { hipemu::shared_close(p); return hipSuccess; }

// ---------------------------------------------------------------------------------------------- the interpreter
namespace hipemu {

struct Idx { unsigned x, y, z; };
struct Fiber {
    void* sp;                 // saved stack pointer while the fiber is not running
    Idx tid;                  // threadIdx
    int wave, lane;
    bool done;
    void* asan_fake;          // AddressSanitizer builds: the fiber's fake-stack handle while it is switched out
    void* tsan_fiber;         // ThreadSanitizer builds: the fiber's own TSan context (every thread of a workgroup is a thread to TSan)
};
extern thread_local Fiber* cur;       // the running fiber (one OS thread runs one workgroup at a time)
extern thread_local Idx g_block;
extern Idx g_bdim, g_gdim;
void* dyn_lds_arena();

// what the 64 lanes of the calling wave passed to this operation, and which of them are still alive
struct Gathered { const uint64_t* v; uint64_t live; };
Gathered wave_gather(uint64_t mine, unsigned op);
void block_barrier();

void run_grid(dim3 grid, dim3 block, size_t lds, void (*thunk)(void*), void* ctx);

template <class K, class... A> inline void launch_k(dim3 grid, dim3 block, size_t lds, K kern, A... args) {
    auto body = [&]() { kern(args...); };
    typedef decltype(body) B;
    run_grid(grid, block, lds, [](void* c) { (*static_cast<B*>(c))(); }, &body);
}

template <class T> inline uint64_t bits_of(T v) { static_assert(sizeof(T) <= 8, "wave operand"); uint64_t b = 0; memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }

enum : unsigned { OP_BALLOT = 1u << 24, OP_SHFL = 2u << 24, OP_SHFL_XOR = 3u << 24, OP_SHFL_UP = 4u << 24, OP_SHFL_DOWN = 5u << 24,
                  OP_DPP = 6u << 24, OP_READLANE = 7u << 24 };

int dpp_source_lane(int lane, int ctrl);          // -1: no valid source (out of the row / unsupported position)

inline int update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const Gathered g = wave_gather(bits_of(src), OP_DPP | (unsigned)ctrl);
    const int lane = cur->lane;
    if (!((row_mask >> (lane >> 4)) & 1) || !((bank_mask >> ((lane >> 2) & 3)) & 1)) return old;
    const int s = dpp_source_lane(lane, ctrl);
    if (s < 0 || !((g.live >> s) & 1)) return bound_ctrl ? 0 : old;
    return from_bits<int>(g.v[s]);
}
}  // namespace hipemu

#define threadIdx (hipemu::cur->tid)
#define blockIdx (hipemu::g_block)
#define blockDim (hipemu::g_bdim)
#define gridDim (hipemu::g_gdim)
#define warpSize 64
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) hipemu::launch_k(dim3(grid), dim3(block), (size_t)(lds), kern, __VA_ARGS__)

inline void __syncthreads() { hipemu::block_barrier(); }
inline unsigned long long __ballot(int pred) {
    const hipemu::Gathered g = hipemu::wave_gather(pred != 0, hipemu::OP_BALLOT);
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) if (((g.live >> l) & 1) && g.v[l]) m |= 1ULL << l;
    return m;
}
inline int __any(int pred) { return __ballot(pred) != 0; }
inline int __all(int pred) {
    const hipemu::Gathered g = hipemu::wave_gather(pred != 0, hipemu::OP_BALLOT);
    for (int l = 0; l < 64; ++l) if (((g.live >> l) & 1) && !g.v[l]) return 0;
    return 1;
}
template <class T> inline T __shfl(T v, int src) { const hipemu::Gathered g = hipemu::wave_gather(hipemu::bits_of(v), hipemu::OP_SHFL); return hipemu::from_bits<T>(g.v[src & 63]); }
template <class T> inline T __shfl_xor(T v, int m) { const hipemu::Gathered g = hipemu::wave_gather(hipemu::bits_of(v), hipemu::OP_SHFL_XOR); return hipemu::from_bits<T>(g.v[(hipemu::cur->lane ^ m) & 63]); }
template <class T> inline T __shfl_up(T v, unsigned d) {
    const hipemu::Gathered g = hipemu::wave_gather(hipemu::bits_of(v), hipemu::OP_SHFL_UP);
    const int s = hipemu::cur->lane - (int)d;
    return s < 0 ? v : hipemu::from_bits<T>(g.v[s]);
}
template <class T> inline T __shfl_down(T v, unsigned d) {
    const hipemu::Gathered g = hipemu::wave_gather(hipemu::bits_of(v), hipemu::OP_SHFL_DOWN);
    const int s = hipemu::cur->lane + (int)d;
    return s > 63 ? v : hipemu::from_bits<T>(g.v[s]);
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl) hipemu::update_dpp((int)(old), (int)(src), (ctrl), (row_mask), (bank_mask), (bound_ctrl))
inline int hipemu_readlane(int v, int lane) { const hipemu::Gathered g = hipemu::wave_gather(hipemu::bits_of(v), hipemu::OP_READLANE); return hipemu::from_bits<int>(g.v[lane & 63]); }
#define __builtin_amdgcn_readlane(v, lane) hipemu_readlane((v), (lane))
inline int hipemu_readfirstlane(int v) {
    const hipemu::Gathered g = hipemu::wave_gather(hipemu::bits_of(v), hipemu::OP_READLANE | 1u);
    return hipemu::from_bits<int>(g.v[__builtin_ctzll(g.live)]);
}
#define __builtin_amdgcn_readfirstlane(v) hipemu_readfirstlane((v))
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_rsqf(x) (1.0f / sqrtf(x))
#define __builtin_amdgcn_sqrtf(x) sqrtf(x)
// s_sleep is what a spin-wait on ANOTHER agent (a neighbour rank's arrival counter) does between two polls: the fiber lets the other
// fibers -- and the other processes -- run, and the scheduler does not take the spinning for a deadlock (the waits are bounded by the
// kernels' own wall-clock timeouts)
#define __builtin_amdgcn_s_sleep(n) hipemu::external_wait()
#define __builtin_amdgcn_s_getreg(x) 0
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline long long __double_as_longlong(double v) { return hipemu::bits_of(v); }
inline double __longlong_as_double(long long v) { return hipemu::from_bits<double>((uint64_t)v); }
inline float __logf(float x) { return logf(x); }
inline float __expf(float x) { return expf(x); }
inline long long wall_clock64() { return (long long)(hipemu::now_ms() * 1e5); }        // 100 MHz
inline long long clock64() { return wall_clock64(); }

// ---- atomics: real ones (workgroups may run on several OS threads, ranks in several processes)
template <class T> inline T hipemu_fetch_op(T* p, T v, T (*op)(T, T)) {
    // compare-and-swap on the word's integer image (integer atomics are what the sanitizers understand)
    typedef typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type W;
    static_assert(sizeof(T) == sizeof(W), "4- or 8-byte operands");
    W* q = reinterpret_cast<W*>(p);
    W o = __atomic_load_n(q, __ATOMIC_RELAXED);
    for (;;) {
        T ov;
        memcpy(&ov, &o, sizeof(T));
        const T nv = op(ov, v);
        W n;
        memcpy(&n, &nv, sizeof(T));
        if (__atomic_compare_exchange_n(q, &o, n, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED)) return ov;
    }
}
inline float atomicAdd(float* p, float v) { return hipemu_fetch_op<float>(p, v, [](float a, float b) { return a + b; }); }
inline double atomicAdd(double* p, double v) { return hipemu_fetch_op<double>(p, v, [](double a, double b) { return a + b; }); }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicMax(T* p, T v) { return hipemu_fetch_op<T>(p, v, [](T a, T b) { return b > a ? b : a; }); }
template <class T> inline T atomicMin(T* p, T v) { return hipemu_fetch_op<T>(p, v, [](T a, T b) { return b < a ? b : a; }); }
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
// the scoped atomics are what ranks in different processes talk through (shared-memory receive areas): real atomics
template <class T> inline T hipemu_atomic_load(const T* p) { T v; __atomic_load(const_cast<T*>(p), &v, __ATOMIC_SEQ_CST); return v; }
template <class T, class V> inline void hipemu_atomic_store(T* p, V v) { T w = (T)v; __atomic_store(p, &w, __ATOMIC_SEQ_CST); }
#define __hip_atomic_load(p, order, scope) hipemu_atomic_load(p)
#define __hip_atomic_store(p, v, order, scope) hipemu_atomic_store((p), (v))
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)
#define __hip_atomic_compare_exchange_strong(p, e, d, o1, o2, scope) __atomic_compare_exchange_n((p), (e), (d), false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)

// HIP puts min / max of the arithmetic types in the global namespace
#define HIPEMU_MINMAX(T) inline T min(T a, T b) { return b < a ? b : a; } inline T max(T a, T b) { return a < b ? b : a; }
HIPEMU_MINMAX(int) HIPEMU_MINMAX(unsigned) HIPEMU_MINMAX(long) HIPEMU_MINMAX(unsigned long) HIPEMU_MINMAX(long long)
HIPEMU_MINMAX(unsigned long long) HIPEMU_MINMAX(float) HIPEMU_MINMAX(double)
#undef HIPEMU_MINMAX

// ---- buffer descriptors (PLB_BUFIO builds): base + byte range; raw loads / stores at voffset + soffset.  Stricter than the
// hardware, which returns 0 / drops the store out of range: an access outside the descriptor's range aborts.
struct hipemu_rsrc { char* base; unsigned bytes; };
typedef hipemu_rsrc __amdgpu_buffer_rsrc_t;
inline hipemu_rsrc hipemu_make_rsrc(void* p, int stride, int num, int) { (void)stride; return hipemu_rsrc{(char*)p, (unsigned)num}; }
template <class W> inline W* hipemu_buf_addr(hipemu_rsrc r, unsigned voff, unsigned soff) {
    if ((size_t)voff + soff + sizeof(W) > r.bytes) { fprintf(stderr, "hipemu: buffer access at %u + %u outside %u bytes\n", voff, soff, r.bytes); abort(); }
    return reinterpret_cast<W*>(r.base + voff + soff);
}
typedef unsigned hipemu_u2 __attribute__((vector_size(8)));
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, num, flags) hipemu_make_rsrc((void*)(p), (stride), (num), (flags))
#define __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, aux) (*hipemu_buf_addr<unsigned>((r), (voff), (soff)))
#define __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, aux) (*hipemu_buf_addr<hipemu_u2>((r), (voff), (soff)))
#define __builtin_amdgcn_raw_buffer_store_b32(v, r, voff, soff, aux) ((void)(*hipemu_buf_addr<unsigned>((r), (voff), (soff)) = (v)))
#define __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, aux) ((void)(*hipemu_buf_addr<hipemu_u2>((r), (voff), (soff)) = (v)))

// ---- host definitions of the inline-assembly helpers of plmpm_kernels.h (their device forms are guarded by PLB_HOST_EMUL)
namespace plb {
inline void lds_barrier() { hipemu::block_barrier(); }
inline void store_flag(int* p, int v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }      // (the same-value block marks of the scatters: see plmpm_kernels.h)
inline void wait_lds() {}
inline void wait_vmem() {}
template <class P, class V> inline void store_through(P* p, V v) { hipemu_atomic_store(p, v); }
}  // namespace plb
