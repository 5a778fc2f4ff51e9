// HIP kernels of the differentiable MLS-MPM substep for gfx950 (MI355X).
//
// Data layout in HBM (T = float on the fast path, double on the parity path):
//   particle frame f : [x0 x1 x2 : double x Npad] [v0..2, C00..22, E00..22 : T x Npad]   (SoA, E = F - I)
//   adjoint frame    : [xa0..2, va0..2, Ca00..22, Ea00..22 : T x Npad]                    (2 ping-pong frames)
//   grid             : 4x4x4-node blocks, block index (bz*nb + by)*nb + bx, node (lz*16 + ly*4 + lx);
//                      grid_out = T4 {v_x, v_y, v_z, 0} and grid_in_adj = T4 {m', mv'} are AoS (gathered 16 B at a
//                      time); the two grids that are *accumulated* with atomics, grid_in {m, mv_x, mv_y, mv_z} and
//                      grid_out_adj {v'_x, v'_y, v'_z}, are SoA: float atomics run at ~300 G/s on consecutive
//                      addresses but ~80 G/s at a 16-byte stride (profiles/microbench/global_atomics.hip).
//                      flags[block] marks blocks touched this substep.
// Particles are stored along the Hilbert curve of their cells (plmpm_set_frame(resort=1); re-sorted on the device
// every cfg.resort_steps env steps), so a 256-thread workgroup's particles cover a small box of cells: the
// scatter / gather kernels stage that box in LDS (tile path) and fall back to direct global atomics when a
// workgroup's bounding box does not fit.  The box of every workgroup is kept per frame (Dev::tiles).  Inside a
// wave, lanes are re-assigned by a DPP bitonic sort on the cell so that same-cell lanes are adjacent, their
// contributions are pre-reduced with DPP row shifts and one lane per cell does the LDS atomics.
#pragma once
#include <hip/hip_runtime.h>
#include "mpm_grid.h"
#include "plmpm_profiling.h"

namespace plb {

#ifndef PLB_KBLOCK
#define PLB_KBLOCK 256
#endif
constexpr int kBlock = PLB_KBLOCK;   // threads per workgroup in particle kernels: 256 particles per box.  Measured optimum (round 4: 512 -> -11 %,
                                     // 128 -> -18 % substeps/s).  What depends on it: the wave round-robin of the grid kernels (a power of two
                                     // of waves), the per-wave box slots (kSred), and the row padding Npad -- the helper kernels of the library
                                     // (re-sort, loss adjoint, frame I/O) run 256-thread workgroups over Npad / 256 chunks, so Npad is padded to
                                     // a multiple of BOTH (round 4's -DPLB_KBLOCK=128 build padded to 128 only: the last 128 rows never
                                     // reached those kernels, which is the "other loss" its notes report)
static_assert(kBlock >= 64 && kBlock <= 1024 && (kBlock & (kBlock - 1)) == 0, "kBlock: a power of two of 64-lane waves");
constexpr int kRowPad = kBlock > 256 ? kBlock : 256;       // frames are padded to whole workgroups of either size
constexpr int kSred = (kBlock / 64) * 6 + 8;               // LDS ints of block_tile_publish / _collect: one box per wave
// minimum waves per SIMD the register allocator must leave room for (second __launch_bounds__ argument)
#ifndef PLB_P2G_WAVES
#define PLB_P2G_WAVES 4
#endif
#ifndef PLB_P2G_GRAD_WAVES
#define PLB_P2G_GRAD_WAVES 2
#endif
// the float64 instantiations of the two scatter kernels: 4 = the fp32 bound (128 VGPRs = 64 doubles: 350 B of scratch, ~200 scratch
// operations per wave in the device listing), 2 = 256 VGPRs; which one is faster is for a measurement to say (round 6: not measured)
#ifndef PLB_P2G_WAVES_F64
#define PLB_P2G_WAVES_F64 PLB_P2G_WAVES
#endif
constexpr int kMaxPrim = 8;
#ifndef PLB_GRID_WG
#define PLB_GRID_WG 512
#endif
constexpr int kGridWG = PLB_GRID_WG;         // workgroups of the persistent grid kernels (4 waves each, one wave per block)
// LDS tile capacity (nodes) of the scatter/gather kernels: 16 KiB per tile for either scalar type
template <class T> struct TileCap;
#ifndef PLB_TILECAP
#define PLB_TILECAP 1024
#endif
template <> struct TileCap<float> { static constexpr int nodes = PLB_TILECAP; };
template <> struct TileCap<double> { static constexpr int nodes = 512; };


// Workgroup barrier that only orders LDS traffic.  __syncthreads() also drains the wave's global-memory counter
// (vmcnt(0)): every load still in flight AND every store just issued -- a full memory round trip per barrier for a wave
// that has just written its particle state.  The barriers of the particle kernels only publish LDS data (tiles, box
// reductions), so they wait for the LDS counter alone and leave loads / stores in flight across them.
#ifndef PLB_LDS_BAR
#define PLB_LDS_BAR 1
#endif
// (the inline-assembly helpers -- this barrier, the counter waits, the write-through stores, the fused DPP steps of seg_sum -- are the
// only device code a C++ compiler for the host cannot read: tests/host_emul/hipemu, the CPU interpreter the device source is
// executed on in the GPU-less test tier, defines PLB_HOST_EMUL and brings its own)
#ifndef PLB_HOST_EMUL
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }       // this wave's LDS traffic has landed
__device__ __forceinline__ void wait_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }        // ... its global loads / stores are acknowledged
#endif
__device__ __forceinline__ void wg_barrier() {
    if (PLB_LDS_BAR) lds_barrier();
    else __syncthreads();
}

// a kernel's dynamic LDS as an array `name` of `type`
#ifndef PLB_DYN_LDS
#define PLB_DYN_LDS(type, name) extern __shared__ type name[]
#endif

template <class T> struct Vec4 { T x, y, z, w; };
template <> struct __attribute__((aligned(16))) Vec4<float> { float x, y, z, w; };
template <> struct __attribute__((aligned(32))) Vec4<double> { double x, y, z, w; };

struct PrimStatic {
    int shape, movable;
    double par[3];
    double friction;
};

// everything a kernel needs to find its data
template <class T> struct Dev {
    SimP<T> P;
    int N, Npad;                     // particles of this frame's storage epoch, padded capacity (SoA stride)
    int go[3];                       // grid window: origin node (multiples of 4) ...
    int nbx, nby, nbz;               // ... and extent in 4^3 blocks; only this box of the n^3 grid is allocated
    int twg;                         // stride of the per-frame workgroup tile table (capacity / 256)
    int fgl, fs;                     // block flags are stored per grid workgroup: flags[(blk & (2^fgl - 1)) * fs + (blk >> fgl)]
    int nprim;
    int z0, z1;                      // owned z-slab (nodes): pose adjoints / loss sums only count these
    int rlo[3], rhi[3];              // reach: stencil bases must satisfy rlo <= base, base + 2 < rhi (window, and slab + halo in z)
    int* err;                        // device error word (bit 0: a particle left the reach box)
    size_t frame_bytes;
    char* state;                     // particle frames
    T* adj[2];                       // ping-pong adjoint frames
    T *mu, *lam, *ys;
    int mats_uniform;                // 1: every particle has the material mat_u = (mu, lam, yield stress); the arrays are not read
    T mat_u[3];
    T* gin[4];                       // grid_m, grid_v_in x/y/z (SoA, accumulated)
    T* goa[3];                       // grid_v_out.grad x/y/z (SoA, accumulated)
    Vec4<T>*grid_out, *grid_in_adj;  // AoS
    int* flags;
    int* tiles;                      // [(F+1)][workgroups][8]: stencil box of each 256-particle workgroup, per frame
    int* contact;                    // [0] = n, [1..n] = blocks whose pose adjoints are still due (grid_op.grad -> p2g.grad)
    const PrimT<T>* ptab;            // [(F+1)][kMaxPrim] the primitives as a grid node sees them during substep f (k_build_prims, once
                                     // per env step behind the kinematics chain): the grid kernels read them straight from here --
                                     // uniform addresses, no LDS copy, no per-workgroup set-up
    unsigned long long* trace;       // profiling builds only
    long long* det;                  // deterministic mode only (else null): two-limb fixed-point accumulators, [8][det_stride]
    size_t det_stride;               //   component c of a node: hi limb det[c * stride + idx], lo limb det[(4 + c) * stride + idx]
    // primitives (double): pos[(F+1)][P][3], rot[(F+1)][P][4], gap[(F+1)][P] (Chopsticks) and adjoints
    const double *ppos, *prot, *pgap;
    double *ppos_a, *prot_a, *pgap_a;
    PrimStatic prim[kMaxPrim];
};

// Component c of row p of a particle SoA array set -- a frame's positions (3 doubles) or scalars (v, C, E: 21 T), an adjoint
// frame (24 T), v[f+1] kept aside (3 T): element c * Np + p of `base`.
// PLB_BUFIO = 1 goes through a buffer descriptor: wave-uniform base, the component's byte offset in an SGPR (scalar unit), the row
// as one 32-bit byte offset per lane shared by every access of the kernel -- no 64-bit vector address arithmetic, which the
// flat form pays with a sign extension and a v_lshl_add_u64 per access (~100-170 of a particle kernel's vector instructions).
// An array set must then stay below 4 GiB (plmpm_create checks the capacity).
#ifndef PLB_BUFIO
#define PLB_BUFIO 0
#endif
template <class E> struct Soa {
    const E* base;
    int Np;
#if PLB_BUFIO
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned stride;                 // bytes between components
    __device__ __forceinline__ Soa(const E* b, int np, int ncomp)
        : base(b), Np(np), rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<E*>(b), 0, (int)((unsigned)np * (unsigned)ncomp * (unsigned)sizeof(E)), 0x00020000)),
          stride((unsigned)np * (unsigned)sizeof(E)) {}
    __device__ __forceinline__ E ld(int c, int p) const {
        if constexpr (sizeof(E) == 4) return __builtin_bit_cast(E, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (unsigned)p * 4u, (unsigned)c * stride, 0));
        else return __builtin_bit_cast(E, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (unsigned)p * 8u, (unsigned)c * stride, 0));
    }
    __device__ __forceinline__ void st(int c, int p, E v) const {
        typedef unsigned u2 __attribute__((vector_size(8)));
        if constexpr (sizeof(E) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, (unsigned)p * 4u, (unsigned)c * stride, 0);
        else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), rsrc, (unsigned)p * 8u, (unsigned)c * stride, 0);
    }
#else
    __device__ __forceinline__ Soa(const E* b, int np, int) : base(b), Np(np) {}
    __device__ __forceinline__ E ld(int c, int p) const { return base[c * Np + p]; }
    __device__ __forceinline__ void st(int c, int p, E v) const { const_cast<E*>(base)[c * Np + p] = v; }
#endif
};
template <class T> __device__ __forceinline__ Soa<double> frame_xs(const Dev<T>& D, int f) {
    return Soa<double>(reinterpret_cast<const double*>(D.state + (size_t)f * D.frame_bytes), D.Npad, 3);
}
template <class T> __device__ __forceinline__ Soa<T> frame_rs(const Dev<T>& D, int f) {
    return Soa<T>(reinterpret_cast<const T*>(D.state + (size_t)f * D.frame_bytes + (size_t)3 * 8 * D.Npad), D.Npad, 21);
}
template <class T> __device__ __forceinline__ Soa<T> adj_s(const Dev<T>& D, int which) { return Soa<T>(D.adj[which], D.Npad, 24); }

template <class T> __device__ __forceinline__ void load_materials(const Dev<T>& D, int p, T& mu, T& lam, T& ys) {
    if (D.mats_uniform) { mu = D.mat_u[0]; lam = D.mat_u[1]; ys = D.mat_u[2]; }
    else { mu = D.mu[p]; lam = D.lam[p]; ys = D.ys[p]; }
}
template <class T> __device__ __forceinline__ const double* frame_x(const Dev<T>& D, int f) {
    return reinterpret_cast<const double*>(D.state + (size_t)f * D.frame_bytes);
}
template <class T> __device__ __forceinline__ double* frame_x_w(const Dev<T>& D, int f) {
    return reinterpret_cast<double*>(D.state + (size_t)f * D.frame_bytes);
}
template <class T> __device__ __forceinline__ T* frame_r(const Dev<T>& D, int f) {
    return reinterpret_cast<T*>(D.state + (size_t)f * D.frame_bytes + (size_t)3 * 8 * D.Npad);
}


// ---- deterministic accumulation (plmpm_config.deterministic) ---------------------------------------------------------
// Floating-point atomics add in whatever order the waves arrive, so two runs of the same rollout differ in the last
// bits.  The deterministic engine accumulates every sum that more than one wave contributes to in two 64-bit integer
// limbs instead -- hi counts units of 2^-24, lo the remainder in units of 2^-76 -- and integer adds commute exactly:
// the result is the same for every arrival order, every re-sort and every workgroup schedule.  A contribution a is
// split without error (|a| < 2^38; anything finer than 2^-76 is rounded once, per contribution, the same way in every
// run), so the sums are also more accurate than the fp32 / fp64 atomics they replace.  Up to 2^12 contributions per
// accumulator before the lo limb could wrap.
__device__ __forceinline__ void det_split(double a, long long& hi, long long& lo) {
    const double h = rint(a * 0x1p24);
    hi = (long long)h;
    lo = (long long)rint((a - h * 0x1p-24) * 0x1p76);
}
__device__ __forceinline__ void det_add(long long* hi, long long* lo, double a) {
    if (a == 0.0) return;
    long long h, l;
    det_split(a, h, l);
    if (h) atomicAdd(reinterpret_cast<unsigned long long*>(hi), (unsigned long long)h);
    if (l) atomicAdd(reinterpret_cast<unsigned long long*>(lo), (unsigned long long)l);
}
__device__ __forceinline__ double det_value(long long hi, long long lo) { return (double)hi * 0x1p-24 + (double)lo * 0x1p-76; }
// component c (< 4) of grid node idx
template <class T> __device__ __forceinline__ void det_add_node(const Dev<T>& D, int c, int idx, double a) {
    det_add(D.det + (size_t)c * D.det_stride + idx, D.det + (size_t)(4 + c) * D.det_stride + idx, a);
}
// one component of an LDS limb tile on to the global limbs
template <class T> __device__ __forceinline__ void det_flush_node(const Dev<T>& D, int c, int idx, long long hi, long long lo) {
    if (hi) atomicAdd(reinterpret_cast<unsigned long long*>(D.det + (size_t)c * D.det_stride + idx), (unsigned long long)hi);
    if (lo) atomicAdd(reinterpret_cast<unsigned long long*>(D.det + (size_t)(4 + c) * D.det_stride + idx), (unsigned long long)lo);
}
// limbs -> the T arrays the scatter would have added into (dst[c] += sum, limbs cleared); dense sweep of the window
template <class T> __global__ void k_det_resolve(long long* det, size_t G, T* d0, T* d1, T* d2, T* d3) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < G; i += (size_t)gridDim.x * blockDim.x) {
#define PLB_DET_RESOLVE(c, dst) if (dst) { long long* h = det + (size_t)(c) * G + i; long long* l = det + (size_t)(4 + (c)) * G + i; \
            const long long hv = *h, lv = *l; if (hv | lv) { dst[i] += (T)det_value(hv, lv); *h = 0; *l = 0; } }
        PLB_DET_RESOLVE(0, d0) PLB_DET_RESOLVE(1, d1) PLB_DET_RESOLVE(2, d2) PLB_DET_RESOLVE(3, d3)
#undef PLB_DET_RESOLVE
    }
}

// node (i, j, k) of the grid -> index inside the allocated window (origin a multiple of 4, so the low bits are the node's)
template <class T> __device__ __forceinline__ int node_index(const Dev<T>& D, int i, int j, int k) {
    return (((((k - D.go[2]) >> 2) * D.nby + ((j - D.go[1]) >> 2)) * D.nbx + ((i - D.go[0]) >> 2)) << 6) | ((k & 3) << 4) | ((j & 3) << 2) | (i & 3);
}
// Block flags live where the grid kernels read them: workgroup g of the 2^fgl persistent grid workgroups owns blocks
// g, g + 2^fgl, g + 2 * 2^fgl, ... (interleaved: the active blocks are spatially clustered) and finds their flags
// side by side, one coalesced load per 64 blocks (indexed by block they were 64 cache lines per wave load).
template <class T> __device__ __forceinline__ int flag_slot(const Dev<T>& D, int blk) {
    return (blk & ((1 << D.fgl) - 1)) * D.fs + (blk >> D.fgl);
}
// A scatter marks the blocks it touches.  Several workgroups of one launch may mark the same block: plain stores of the same value, a
// race by design (nobody reads a flag in the launch that sets it).  One function, so that the tests' CPU interpreter -- which runs the
// workgroups on OS threads under ThreadSanitizer -- can make exactly these stores relaxed atomics and report every OTHER conflict.
#ifndef PLB_HOST_EMUL
__device__ __forceinline__ void store_flag(int* p, int v) { *p = v; }
#endif
template <class T> __device__ __forceinline__ void mark_block(const Dev<T>& D, int blk) { store_flag(&D.flags[flag_slot(D, blk)], 1); }
// block index inside the window -> node coordinates of lane `lane` of the wave that owns the block
template <class T> __device__ __forceinline__ void block_nodes(const Dev<T>& D, int blk, int lane, int* I) {
    const int bx = blk % D.nbx, by = (blk / D.nbx) % D.nby, bz = blk / (D.nbx * D.nby);
    I[0] = D.go[0] + ((bx << 2) | (lane & 3)); I[1] = D.go[1] + ((by << 2) | ((lane >> 2) & 3)); I[2] = D.go[2] + ((bz << 2) | (lane >> 4));
}

// WAVE: every wave fills sp itself (lanes 0 .. nprim-1 of each wave write the same bytes) and only waits for its own
// LDS writes -- no workgroup barrier, so no wave waits for another wave's loads before it may go on
template <class T, bool WAVE = false> __device__ __forceinline__ void load_prims(const Dev<T>& D, int f, PrimT<T>* sp) {
    // called by all threads of a workgroup; sp in LDS
    int t = WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    if (t < D.nprim) {
        PrimT<T> p;
        p.shape = D.prim[t].shape; p.movable = D.prim[t].movable; p.friction = (T)D.prim[t].friction;
        for (int i = 0; i < 3; ++i) p.par[i] = D.prim[t].par[i];
        if (p.shape == SHAPE_CHOPSTICKS) p.par[2] = D.pgap[(size_t)f * D.nprim + t];
        p.rb = prim_bounding_radius(p.shape, p.par);
        const double* a = D.ppos + ((size_t)f * D.nprim + t) * 3;
        const double* b = D.ppos + ((size_t)(f + 1) * D.nprim + t) * 3;
        const double* c = D.prot + ((size_t)f * D.nprim + t) * 4;
        const double* d = D.prot + ((size_t)(f + 1) * D.nprim + t) * 4;
        for (int i = 0; i < 3; ++i) { p.pos[i] = a[i]; p.pos1[i] = b[i]; }
        for (int i = 0; i < 4; ++i) { p.rot[i] = c[i]; p.rot1[i] = d[i]; }
        sp[t] = p;
    }
    if (WAVE) wait_lds();
}

// one record per (frame, primitive): what load_prims assembles, kept in HBM for the fused-grid fills
template <class T> __global__ void k_build_prims(Dev<T> D, int first, int n, PrimT<T>* ptab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * D.nprim) return;
    const int f = first + i / D.nprim, t = i % D.nprim;
    PrimT<T> p;
    p.shape = D.prim[t].shape; p.movable = D.prim[t].movable; p.friction = (T)D.prim[t].friction;
    for (int k = 0; k < 3; ++k) p.par[k] = D.prim[t].par[k];
    if (p.shape == SHAPE_CHOPSTICKS) p.par[2] = D.pgap[(size_t)f * D.nprim + t];
    p.rb = prim_bounding_radius(p.shape, p.par);
    const double* a = D.ppos + ((size_t)f * D.nprim + t) * 3;
    const double* b = D.ppos + ((size_t)(f + 1) * D.nprim + t) * 3;
    const double* c = D.prot + ((size_t)f * D.nprim + t) * 4;
    const double* d = D.prot + ((size_t)(f + 1) * D.nprim + t) * 4;
    for (int k = 0; k < 3; ++k) { p.pos[k] = a[k]; p.pos1[k] = b[k]; }
    for (int k = 0; k < 4; ++k) { p.rot[k] = c[k]; p.rot1[k] = d[k]; }
    ptab[(size_t)f * kMaxPrim + t] = p;
}

// Grid kernels run kGridWG persistent workgroups.  Workgroup g owns blocks g, g + G, g + 2G, ... (G = gridDim.x): the
// active blocks are spatially clustered, so this interleaving hands every workgroup about the same number of them.
// A wave reads 64 of its workgroup's flags at a time (one strided load), ballots, and the workgroup's 4 waves
// take the set bits round-robin.  (One wave per block of the whole grid with an early return for the ~95 % empty
// ones needs ~16 occupancy rounds of flag loads per launch at 128^3; a separate compaction kernel costs a 5 us
// dispatch on the critical path.)
// Multi-GPU: the block planes around a slab face hold partial sums on both neighbours.  Each rank sends its copy of
// those planes straight out of the grid arrays (the blocked layout keeps a range of block planes contiguous per SoA
// component: no pack kernel) and the grid kernels add the received copy on first touch (no unpack kernel).
//   buf[f]: [ncomp][(bb - ba) * nbx * nby * 64] scalars, block planes [ba[f], bb[f]) of the window
struct HaloIn {
    int n;                           // faces with a neighbour (0 on a single GPU)
    int ba[2], bb[2];
    const void* buf[2];
    const int* valid[2];             // device-side exchange: one word per block of the face's planes -- 0: the neighbour sent nothing
                                     // for it (its copy of the block is empty; buf holds stale values there); nullptr: all sent
    int part;                        // 0: every active block | 1: only blocks outside the exchanged planes (they do not
                                     // need the neighbours' values: this pass can run while the halos are in flight) | 2: only
                                     // the blocks of the exchanged planes
};
template <class T> __device__ __forceinline__ int halo_face_of(const Dev<T>& D, const HaloIn& H, int blk) {
    const int bz = blk / (D.nbx * D.nby);
    for (int f = 0; f < H.n; ++f)
        if (bz >= H.ba[f] && bz < H.bb[f]) return f;
    return -1;
}
// did the neighbour on face f send block blk (which must lie in face f's planes)?
// (system-scope loads -- global_load ... sc0 sc1: what a neighbour on another GPU wrote into the receive area must not be
// served from a stale cache line of this GPU; a few hundred KB per launch, so the cache bypass costs nothing)
template <class T> __device__ __forceinline__ bool halo_sent(const Dev<T>& D, const HaloIn& H, int f, int blk) {
    return !H.valid[f] || __hip_atomic_load(H.valid[f] + (blk - H.ba[f] * D.nbx * D.nby), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
}
// received value of component c at node (blk, lane); blk must lie in face f's planes
template <class T> __device__ __forceinline__ T halo_value(const Dev<T>& D, const HaloIn& H, int f, int c, int blk, int lane) {
    const size_t per = (size_t)(H.bb[f] - H.ba[f]) * D.nbx * D.nby * 64;
    return __hip_atomic_load((const T*)H.buf[f] + ((size_t)c * per + ((size_t)(blk - H.ba[f] * D.nbx * D.nby) << 6) + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// first 64 flags of this workgroup: issued by the grid kernels before they wait for the primitives' poses, so that the
// two memory latencies overlap
template <class T> __device__ __forceinline__ int first_flags(const Dev<T>& D) {
    const int lane = threadIdx.x & 63;
    return lane < D.fs ? D.flags[blockIdx.x * D.fs + lane] : 0;
}
// CAND: blocks of the exchanged planes are candidates whatever their flag
// part: H.part, or -- the fused exchange + grid kernels make two passes over the same (kernel-argument) HaloIn -- the pass's own
template <bool CAND, class T, class Body> __device__ __forceinline__ void for_each_active_block(const Dev<T>& D, const HaloIn& H, int part, int flags0, Body&& body) {
    const int nblk = D.nbx * D.nby * D.nbz, g = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k0 = 0; k0 < D.fs; k0 += 64) {
        const int k = k0 + lane, b = g + (k << D.fgl);
        const bool in = k < D.fs && b < nblk;
        const int fl = k0 == 0 ? flags0 : (in ? D.flags[g * D.fs + k] : 0);
        // blocks in the exchanged planes are candidates whatever their flag: the neighbour's particles may reach
        // nodes that none of ours do (the body skips a candidate whose summed mass is zero everywhere)
        const bool face = H.n > 0 && halo_face_of(D, H, b) >= 0;
        const unsigned long long m = __ballot(in && (fl != 0 || (CAND && face)) && (part == 0 || (part == 2) == face));
        __syncthreads();             // bodies may clear flags: every wave must have taken the same snapshot first
        int rank = 0;
        for (unsigned long long r = m; r; r &= r - 1, ++rank)
            if ((rank & (kBlock / 64 - 1)) == wave) body(g + ((k0 + __ffsll((long long)r) - 1) << D.fgl));
    }
}

// ---- device-side halo exchange (plmpm_peer.hip), as a kernel of its own or folded into the grid kernel that consumes it ----
// One exchange of a halo field: this rank's copy of the exchanged block planes goes straight into the neighbours' receive
// areas (IPC-mapped, uncached), an arrival counter is published behind the data, and the neighbours' counters are waited for.
//   receive area of a face:  [256 B: arrival counter][half 0][half 1],  half = [one validity word per 4^3 block, padded to
//                            256 B][ncomp x count scalars];  exchange k of a field writes half k & 1 (plmpm_peer.hip: why two)
struct PeerXchg {
    int n, ncomp;                // faces with a neighbour (0: no exchange), components of the field
    const void* src[4];          // component c of this rank's grid array (block 0 of the window)
    char* dst[2];                // the half of the neighbour's receive area this exchange writes: validity words, then data
    size_t data_ofs[2];          // bytes from dst to the data
    int blk0[2], nblk[2];        // first block and number of blocks of the face's planes
    const int* flags;            // activity flags of the frame's blocks (flag_slot layout); nullptr: every block is sent
    int fgl, fs;
    unsigned* arrive_remote[2];  // the neighbour's counter for this rank's planes
    unsigned* arrive_local[2];   // this rank's counters
    unsigned seq;
    unsigned* done;              // [0]: workgroups of the running launch that have finished their sends; [8]: tag of the last exchange whose
                                 // neighbours have all arrived (fused kernels: what the workgroups other than the poller wait on)
    unsigned tag;                // this exchange's tag (unique per engine, never 0)
    int* status;                 // pinned host word the host reads
    int* status_dev;             // device copy of it: what the kernels look at (a system-scope load of host memory crosses PCIe: +1.4 us per exchange)
    int code;                    // field << 16 | 1
    long long timeout_ticks;     // of the 100 MHz wall clock
    float spoil;                 // 1; a test hook (plmpm_debug_peer_spoil) scales what face 0 sends, to prove that a wrong halo is NOTICED
};
// Stores written THROUGH the L2 (sc0 sc1: system-scope write-through): once the store is acknowledged the data is where the
// neighbour -- another XCD, another process, another GPU -- reads it, and no L2 write-back is needed before the arrival
// counter moves.  (A release fence at system scope writes back everything the PREVIOUS kernels left dirty in this XCD's L2
// -- megabytes of particle state after a particle kernel: 11 us per exchange instead of 6; a system fence per thread: 25.)
#ifndef PLB_HOST_EMUL
__device__ __forceinline__ void store_through(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store_through(double* p, double v) { asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store_through(int* p, int v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store_through(unsigned* p, unsigned v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
#endif

// the block `bl` of face i's planes (wave-uniform): validity word (lane 0), and -- if the block carries something -- its 64 nodes
// per component, read from this rank's grid and stored through to the neighbour
template <class T> __device__ __forceinline__ void xchg_send_block(const PeerXchg& X, int i, int bl, int lane) {
    const int blk = X.blk0[i] + bl;
    const bool on = X.flags ? (X.flags[(blk & ((1 << X.fgl) - 1)) * X.fs + (blk >> X.fgl)] != 0) : true;      // flag_slot
    if (lane == 0) store_through((int*)X.dst[i] + bl, on ? 1 : 0);
    if (!on) return;
    T* data = (T*)(X.dst[i] + X.data_ofs[i]);
    const size_t count = (size_t)X.nblk[i] << 6;
    T v[4];
    for (int c = 0; c < X.ncomp; ++c) {
        v[c] = ((const T*)X.src[c])[((size_t)blk << 6) + lane];
        if (i == 0 && X.spoil != 1.0f) v[c] *= (T)X.spoil;
    }
    for (int c = 0; c < X.ncomp; ++c) store_through(data + (size_t)c * count + ((size_t)bl << 6) + lane, v[c]);
}
// all threads of a workgroup call, behind their sends: every wave waits for the acknowledgements of its own stores, the
// workgroup counts itself done, and the LAST workgroup of the launch -- every copy has landed -- publishes the arrival
// counters.  Returns true in that workgroup (all threads).
__device__ __forceinline__ bool xchg_publish(const PeerXchg& X) {
    __shared__ int s_last;
    wait_vmem();
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool last = __hip_atomic_fetch_add(X.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
        if (last) {
            __hip_atomic_store(X.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the next launch is stream-ordered behind this one
            // (published even when an earlier exchange timed out: a live neighbour is not kept waiting by THIS rank)
            if (X.n > 0) store_through(X.arrive_remote[0], X.seq);          // (constant indices: kernel-argument fields read with
            if (X.n > 1) store_through(X.arrive_remote[1], X.seq);          //  run-time indices become dependent loads inside the kernel)
        }
        s_last = last;
    }
    __syncthreads();
    return s_last != 0;
}
// threads 0 .. n-1 of the calling wave(s) wait for the neighbours' counters (one lane per face).  A neighbour that has stopped
// is reported after timeout_ticks -- and never waited for again: an earlier timeout (status_dev != 0) makes the launches still
// queued behind it drain in microseconds, not in n x PLMPM_PEER_TIMEOUT, before the host gets to look at the status word.
__device__ __forceinline__ void xchg_wait_lanes(const PeerXchg& X) {
    if ((int)threadIdx.x < X.n) {
        const int i = threadIdx.x;
        unsigned* const mine = i == 0 ? X.arrive_local[0] : X.arrive_local[1];
        const bool dead = __hip_atomic_load(X.status_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        const long long t0 = wall_clock64();
        for (; !dead;) {
            const unsigned got = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((int)(got - X.seq) >= 0) break;
            if (wall_clock64() - t0 > X.timeout_ticks) {         // the neighbour is gone: report, do not hang the GPU
                __hip_atomic_store(X.status, X.code | (i << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(X.status_dev, X.code | (i << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        wait_vmem();
    }
}
// Folded into a persistent grid kernel (k_grid_op_x / k_grid_op_grad_x): workgroup g sends the blocks of the exchanged planes
// that it OWNS (g, g + G, ...: the blocks it will itself sum and process once the neighbour's copy is there -- program order
// inside the workgroup keeps a block's send ahead of the write-back of its complete sums, no other workgroup touches it), one
// wave per block; then the launch's last workgroup publishes.  Every workgroup of the launch must be resident for the wait
// that follows to end (the launch is sized for that: plmpm_peer.hip).
// The wait is hierarchical: the publishing workgroup alone polls the neighbours' counters (uncached memory: hundreds of
// workgroups polling one uncached word queue up at its memory channel -- measured: 40 us per launch instead of 10) and then
// raises a tag in ordinary device memory, which the other workgroups poll in the L2 (xchg_wait_arrived).
template <class T> __device__ __forceinline__ void xchg_push_owned(const Dev<T>& D, const PeerXchg& X) {
    const int g = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i >= X.n) break;
        const int lo = X.blk0[i], hi = X.blk0[i] + X.nblk[i];
        int k = lo > g ? (lo - g + (1 << D.fgl) - 1) >> D.fgl : 0;
        for (int r = 0, b = g + (k << D.fgl); b < hi; ++r, ++k, b = g + (k << D.fgl))
            if ((r & (kBlock / 64 - 1)) == wave) xchg_send_block<T>(X, i, b - lo, lane);
    }
    if (xchg_publish(X)) {           // the launch's poller: neighbours' arrivals (or a timeout), then the tag for everybody else
        xchg_wait_lanes(X);
        __syncthreads();
        // (relaxed: the tag carries no data -- what arrived is read from uncached memory with system-scope loads; a release
        // here would write back everything the previous kernels left dirty in this XCD's L2)
        if (threadIdx.x == 0) __hip_atomic_store(X.done + 8, X.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// all threads of every workgroup call, in front of the blocks of the exchanged planes.  Bounded like the poller's wait (twice its
// limit: the poller raises the tag after ITS timeout at the latest): a launch whose workgroups are not all resident -- more grid
// workgroups than the GPU holds at once, so that the publishing workgroup never gets to run -- reports face 0xff instead of hanging.
__device__ __forceinline__ void xchg_wait_arrived(const PeerXchg& X) {
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(X.done + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != X.tag) {
            if (wall_clock64() - t0 > 2 * X.timeout_ticks) {
                // only if nobody has reported yet: the poller's code names the field AND the face that stayed silent
                int none = 0;
                if (__hip_atomic_compare_exchange_strong(X.status_dev, &none, X.code | (0xff << 8), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                    __hip_atomic_store(X.status, X.code | (0xff << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

// The 3-wide stencil at `base` must stay inside the reach box: the allocated grid window, and in z also this rank's
// slab + halo.  A particle that leaves it raises the error word (the caller fails the step); its base is clamped
// into the box so that every index computed from it stays inside the allocation.
template <class T> __device__ __forceinline__ bool clamp_to_reach(const Dev<T>& D, int* base) {
    bool out = false;
    for (int d = 0; d < 3; ++d) {
        const int b = min(max(base[d], D.rlo[d]), D.rhi[d] - 3);
        out |= b != base[d];
        base[d] = b;
    }
    return out;
}

// XCD-aware workgroup -> particle chunk map.  Workgroup b of a launch runs on XCD b % 8 (observed, MI355X_MICROARCH.md:
// "block b runs on XCD b % 8"; a speed assumption only -- the map is a bijection whatever the placement), and every XCD
// has its own L2.  Particles are stored along the Hilbert curve, so neighbouring 256-particle chunks have overlapping
// stencil boxes; with the identity map those neighbours sit on eight different L2s.  Here XCD k takes the k-th
// CONTIGUOUS eighth of the chunks: the tile fills of neighbouring workgroups hit the same L2.
#ifndef PLB_XCD_MAP
#define PLB_XCD_MAP 0            // measured (round 4): no effect either way (48.9 / 31.2 / 43.9 us with it against 48.6 / 30.9 / 43.9): the identity stays
#endif
__device__ __forceinline__ int xcd_chunk(int b, int n) {
    if (!PLB_XCD_MAP) return b;
    const int q = n >> 3, r = n & 7, k = b & 7;
    return k * q + min(k, r) + (b >> 3);
}

// ------------------------------------------------------------------------------------------------
// workgroup bounding box of stencil bases -> LDS tile geometry
struct Tile {
    int o[3];      // origin node
    int e[3];      // extent in nodes
    int ok;        // fits in the LDS tile
};

// linear tile index -> (lz, ly, lx).  Tiles hold < 2^11 nodes, so the two divisions are a float multiply by a
// reciprocal plus one correction each (~8 VALU instructions) instead of the ~30-instruction integer division hipcc
// expands -- these run once per tile loop in every wave of the particle kernels.
__device__ __forceinline__ int small_div(int i, int d) {
    int q = (int)((float)i * __builtin_amdgcn_rcpf((float)d));      // v_rcp_f32: within 1 ulp, the two corrections make it exact
    q += ((q + 1) * d <= i) ? 1 : 0;
    q -= (q * d > i) ? 1 : 0;
    return q;
}
__device__ __forceinline__ void tile_coords(int i, int ex, int exy, int& lz, int& ly, int& lx) {
    lz = small_div(i, exy);
    const int r = i - lz * exy;
    ly = small_div(r, ex);
    lx = r - ly * ex;
}

// wave-wide min / max of an int: DPP scan inside the rows, row_bcast to chain the rows, result read from lane 63
// (six VALU steps instead of six dependent trips through the LDS crossbar)
template <bool MAX> __device__ __forceinline__ int wave_minmax(int v) {
#define PLB_MM(ctrl, rowmask)                                                                      \
    { int o = __builtin_amdgcn_update_dpp(v, v, ctrl, rowmask, 0xf, false); v = MAX ? max(v, o) : min(v, o); }
    PLB_MM(0x111, 0xf) PLB_MM(0x112, 0xf) PLB_MM(0x114, 0xf) PLB_MM(0x118, 0xf)      // row_shr 1, 2, 4, 8: lane 15 of each row is complete
    PLB_MM(0x142, 0xa)                                                                   // row_bcast:15 -> rows 1, 3
    PLB_MM(0x143, 0xc)                                                                   // row_bcast:31 -> rows 2, 3
#undef PLB_MM
    return __builtin_amdgcn_readlane(v, 63);
}
// wave-wide sum of a double, complete in lane 63: same six DPP steps (each half of the double moved separately)
template <int CTRL, int ROWMASK> __device__ __forceinline__ double dpp_move_f64(double v) {
    long long b = __builtin_bit_cast(long long, v);
    int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, ROWMASK, 0xf, true);
    int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROWMASK, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
    v += dpp_move_f64<0x111, 0xf>(v);
    v += dpp_move_f64<0x112, 0xf>(v);
    v += dpp_move_f64<0x114, 0xf>(v);
    v += dpp_move_f64<0x118, 0xf>(v);
    v += dpp_move_f64<0x142, 0xa>(v);
    v += dpp_move_f64<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ int wave_min(int v) { return wave_minmax<false>(v); }
__device__ __forceinline__ int wave_max(int v) { return wave_minmax<true>(v); }

// Wave-local reassignment of particles to lanes so that particles sharing a stencil base sit in adjacent lanes
// (bitonic sort of (cell key, lane) over the 64 lanes).  The storage order is only sorted at episode reset; as
// particles drift, same-cell lanes stop being neighbours and the run-based pre-reduction below would leave
// same-address LDS atomics behind (measured: scatter kernels 2x slower after 8 env steps).  Returns the lane whose
// particle this lane should process.
__device__ __forceinline__ int wave_sort_lanes(long long key) {
    const int lane = threadIdx.x & 63;
    unsigned long long v = ((unsigned long long)key << 6) | (unsigned)lane;
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffu), j), hi = __shfl_xor((unsigned)(v >> 32), j);
            unsigned long long o = ((unsigned long long)hi << 32) | lo;
            const bool up = (lane & k) == 0;              // ascending block
            const bool lower = (lane & j) == 0;           // this lane keeps the smaller element
            const bool take = (lower == up) ? (o < v) : (o > v);
            v = take ? o : v;
        }
    }
    return (int)(v & 63);
}

// 32-bit flavour of the same sort (cell index << 6 | lane fits 32 bits up to n_grid = 256): the 18 exchange steps
// with a partner inside the 16-lane row are DPP moves (VALU only), only the three cross-row steps go through the LDS
// crossbar.  The shuffle version spends ~4k cycles of pure latency per wave (42 dependent ds_bpermute round trips,
// wave trace in profiles/); this one ~1k.
template <int CTRL> __device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
template <int J> __device__ __forceinline__ unsigned partner_xor(unsigned v, int lane) {
    if (J == 1) return dpp_u32<0xB1>(v);                       // quad_perm [1,0,3,2]
    if (J == 2) return dpp_u32<0x4E>(v);                       // quad_perm [2,3,0,1]
    if (J == 4) { unsigned a = dpp_u32<0x124>(v), b = dpp_u32<0x12C>(v); return (lane & 4) ? a : b; }   // row_ror:4 / :12
    if (J == 8) return dpp_u32<0x128>(v);                      // row_ror:8 == xor 8 inside a row of 16
    return (unsigned)__shfl_xor((int)v, J);
}
template <int K, int J> __device__ __forceinline__ unsigned bitonic_step(unsigned v, int lane) {
    const unsigned o = partner_xor<J>(v, lane);
    const unsigned mn = v < o ? v : o, mx = v < o ? o : v;
    const bool keep_min = ((lane & J) == 0) == ((lane & K) == 0);
    return keep_min ? mn : mx;
}
template <int K> __device__ __forceinline__ unsigned bitonic_merge(unsigned v, int lane) {
    if (K >= 64) v = bitonic_step<K, 32>(v, lane);
    if (K >= 32) v = bitonic_step<K, 16>(v, lane);
    if (K >= 16) v = bitonic_step<K, 8>(v, lane);
    if (K >= 8) v = bitonic_step<K, 4>(v, lane);
    if (K >= 4) v = bitonic_step<K, 2>(v, lane);
    return bitonic_step<K, 1>(v, lane);
}
__device__ __forceinline__ int wave_sort_lanes32(unsigned key) {          // key < 2^26
    const int lane = threadIdx.x & 63;
    unsigned v = (key << 6) | (unsigned)lane;
    v = bitonic_merge<2>(v, lane);
    v = bitonic_merge<4>(v, lane);
    v = bitonic_merge<8>(v, lane);
    v = bitonic_merge<16>(v, lane);
    v = bitonic_merge<32>(v, lane);
    v = bitonic_merge<64>(v, lane);
    return (int)(v & 63u);
}

// Wave-level segmented reduction.  Particles are stored cell-sorted, so lanes that share a stencil base
// form runs; the 27 x 4 per-particle contributions of a run are summed with shuffles and only the run's
// head lane touches LDS / HBM atomics (same-address atomics serialise, shuffles do not).
// Runs are additionally cut at 16-lane DPP rows so the whole reduction is 4 row-shift steps of pure VALU.
template <class T> struct Seg {
    T m1, m2, m4, m8;   // 1 where the lane `d` to the right still belongs to this lane's run
    bool head;          // first lane of a (row-clipped) run
};
#ifndef PLB_SEG_STEPS
#define PLB_SEG_STEPS 4          // 4: runs clipped at the 16-lane DPP rows; 3: at 8 lanes (one reduction step less, a few more LDS atomics)
#endif
template <class T> __device__ __forceinline__ Seg<T> wave_segments(int key) {
    const int lane = threadIdx.x & 63;
    int prev = __shfl_up(key, 1);
    Seg<T> s;
    s.head = ((lane & ((1 << PLB_SEG_STEPS) - 1)) == 0) || (key != prev);
    unsigned long long heads = __ballot(s.head);
    unsigned long long higher = lane == 63 ? 0ULL : (heads >> (lane + 1));
    const int end = higher ? lane + __ffsll((long long)higher) - 1 : 63;
    s.m1 = lane + 1 <= end ? T(1) : T(0);
    s.m2 = lane + 2 <= end ? T(1) : T(0);
    s.m4 = lane + 4 <= end ? T(1) : T(0);
    s.m8 = lane + 8 <= end ? T(1) : T(0);
    return s;
}
// value of the lane D to the right inside the 16-lane row (0 outside the row)
template <int D> __device__ __forceinline__ float row_shl(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + D, 0xf, 0xf, true));
}
template <int D> __device__ __forceinline__ double row_shl(double v) {
    long long b = __builtin_bit_cast(long long, v);
    int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x100 + D, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x100 + D, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo);
}
template <class T> __device__ __forceinline__ T seg_sum(T v, const Seg<T>& s) {
    v += row_shl<1>(v) * s.m1;
    v += row_shl<2>(v) * s.m2;
    v += row_shl<4>(v) * s.m4;
#if PLB_SEG_STEPS >= 4
    v += row_shl<8>(v) * s.m8;
#endif
    return v;
}
// Several values at once.  For float the four steps are single fused v_fmac_f32_dpp instructions (hipcc does
// not fold v_mov_b32_dpp into the multiply-add by itself); interleaving the values keeps >= 2 instructions
// between a write and the DPP read of the same register, the leading s_nop covers the producer before the block.
template <class T> __device__ __forceinline__ void seg_sum4(T& a, T& b, T& c, T& d, const Seg<T>& s) {
    a = seg_sum(a, s); b = seg_sum(b, s); c = seg_sum(c, s); d = seg_sum(d, s);
}
template <class T> __device__ __forceinline__ void seg_sum3(T& a, T& b, T& c, const Seg<T>& s) {
    a = seg_sum(a, s); b = seg_sum(b, s); c = seg_sum(c, s);
}
#ifndef PLB_HOST_EMUL
#define PLB_DPP_STEP(x, m, n) "v_fmac_f32_dpp " x ", " x ", " m " row_shl:" n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
template <> __device__ __forceinline__ void seg_sum4<float>(float& a, float& b, float& c, float& d, const Seg<float>& s) {
    asm("s_nop 1\n"
        PLB_DPP_STEP("%0", "%4", "1") PLB_DPP_STEP("%1", "%4", "1") PLB_DPP_STEP("%2", "%4", "1") PLB_DPP_STEP("%3", "%4", "1")
        PLB_DPP_STEP("%0", "%5", "2") PLB_DPP_STEP("%1", "%5", "2") PLB_DPP_STEP("%2", "%5", "2") PLB_DPP_STEP("%3", "%5", "2")
        PLB_DPP_STEP("%0", "%6", "4") PLB_DPP_STEP("%1", "%6", "4") PLB_DPP_STEP("%2", "%6", "4") PLB_DPP_STEP("%3", "%6", "4")
#if PLB_SEG_STEPS >= 4
        PLB_DPP_STEP("%0", "%7", "8") PLB_DPP_STEP("%1", "%7", "8") PLB_DPP_STEP("%2", "%7", "8") PLB_DPP_STEP("%3", "%7", "8")
#endif
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s.m1), "v"(s.m2), "v"(s.m4), "v"(s.m8));
}
template <> __device__ __forceinline__ void seg_sum3<float>(float& a, float& b, float& c, const Seg<float>& s) {
    asm("s_nop 1\n"
        PLB_DPP_STEP("%0", "%3", "1") PLB_DPP_STEP("%1", "%3", "1") PLB_DPP_STEP("%2", "%3", "1") "s_nop 0\n"
        PLB_DPP_STEP("%0", "%4", "2") PLB_DPP_STEP("%1", "%4", "2") PLB_DPP_STEP("%2", "%4", "2") "s_nop 0\n"
        PLB_DPP_STEP("%0", "%5", "4") PLB_DPP_STEP("%1", "%5", "4") PLB_DPP_STEP("%2", "%5", "4")
#if PLB_SEG_STEPS >= 4
        "s_nop 0\n"
        PLB_DPP_STEP("%0", "%6", "8") PLB_DPP_STEP("%1", "%6", "8") PLB_DPP_STEP("%2", "%6", "8")
#endif
        : "+v"(a), "+v"(b), "+v"(c) : "v"(s.m1), "v"(s.m2), "v"(s.m4), "v"(s.m8));
}
#undef PLB_DPP_STEP
#endif

// all threads call; valid == false for padding lanes.  sred: LDS int[kSred], written once per kernel.
// In two halves so that a kernel can put its own barrier between them: every wave publishes its box ...
__device__ __forceinline__ void block_tile_publish(const int* base, bool valid, int* sred) {
    int lo[3], hi[3];
    for (int d = 0; d < 3; ++d) {
        lo[d] = wave_min(valid ? base[d] : 0x7fffffff);
        hi[d] = wave_max(valid ? base[d] : -0x7fffffff);
    }
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        for (int d = 0; d < 3; ++d) { sred[wave * 6 + d] = lo[d]; sred[wave * 6 + 3 + d] = hi[d]; }
}
// ... and, after a __syncthreads(), every thread combines them
__device__ __forceinline__ Tile block_tile_collect(const int* sred, int cap) {
    Tile t;
    int nodes = 1;
    for (int d = 0; d < 3; ++d) {
        int l = sred[d], h = sred[3 + d];
        for (int w = 1; w < kBlock / 64; ++w) { l = min(l, sred[w * 6 + d]); h = max(h, sred[w * 6 + 3 + d]); }
        t.o[d] = l;
        t.e[d] = h - l + 3;
        nodes *= t.e[d];
    }
    t.ok = (nodes > 0 && nodes <= cap) ? 1 : 0;
    return t;
}
__device__ __forceinline__ Tile block_tile(const int* base, bool valid, int* sred, int cap) {
    block_tile_publish(base, valid, sred);
    __syncthreads();
    return block_tile_collect(sred, cap);
}

// The box of frame f is computed once, by the kernel that scatters frame f (it has to reduce over the workgroup
// anyway), and kept per frame: every later kernel over the same frame -- g2p, g2p.grad, p2g.grad -- reads 6 ints
// and can start filling its LDS tile at once, in parallel with its particle loads, instead of load -> reduce ->
// barrier -> fill.
template <class T> __device__ __forceinline__ void store_tile(const Dev<T>& D, int f, const Tile& t, int wg = blockIdx.x) {
    if (threadIdx.x < 6) {
        // (selects, not t.o[threadIdx.x]: a per-lane index into the struct sends all of it through scratch memory -- two scratch
        // stores in every wave of every scatter launch and a dependent scratch load in front of this store)
        const int i = threadIdx.x;
        const int v = i == 0 ? t.o[0] : i == 1 ? t.o[1] : i == 2 ? t.o[2] : i == 3 ? t.e[0] : i == 4 ? t.e[1] : t.e[2];
        int* q = D.tiles + ((size_t)f * D.twg + wg) * 8;
        q[i] = v;
    }
}
template <class T> __device__ __forceinline__ Tile load_tile(const Dev<T>& D, int f, int cap, int wg = blockIdx.x) {
    const int* q = D.tiles + ((size_t)f * D.twg + wg) * 8;
    Tile t;
    int nodes = 1;
    for (int d = 0; d < 3; ++d) { t.o[d] = q[d]; t.e[d] = q[3 + d]; nodes *= t.e[d]; }
    t.ok = (nodes > 0 && nodes <= cap) ? 1 : 0;
    return t;
}


// Sorted particle load in two halves so that independent memory traffic can be issued in between.
struct SortLoad { double x0[3]; long long key; };
template <class T> __device__ __forceinline__ SortLoad sorted_begin(const Dev<T>& D, const Soa<double>& X, int wg = blockIdx.x) {
    const int p0 = wg * kBlock + threadIdx.x;
    SortLoad s;
    s.x0[0] = s.x0[1] = s.x0[2] = 0.5;
    if (p0 < D.N) for (int d = 0; d < 3; ++d) s.x0[d] = X.ld(d, p0);
    return s;
}
template <class T>
__device__ __forceinline__ bool sorted_finish(const Dev<T>& D, SortLoad& s, int& p, double* x, int* base, bool flag_err = false, int wg = blockIdx.x) {
    const int p0 = wg * kBlock + threadIdx.x;
    s.key = (1LL << 40);                                            // padding lanes last
    if (p0 < D.N) {
        int b[3];
        for (int d = 0; d < 3; ++d) b[d] = (int)(s.x0[d] * (double)D.P.inv_dx - 0.5);
        s.key = ((long long)b[2] * D.P.n + b[1]) * D.P.n + b[0];
    }
    // padding lanes carry the largest key either way; the 32-bit network needs (cells << 6) to fit
    // (skipping the sort when the lanes already come in few runs of equal keys was measured in round 2: the extra LDS
    // atomics of the shorter runs cost more than the sort, profiles/r02_notes.md)
    const int src = (D.P.n <= 256 ? wave_sort_lanes32(p0 < D.N ? (unsigned)s.key : 0x3ffffffu) : wave_sort_lanes(s.key));
    p = (p0 & ~63) + src;
    for (int d = 0; d < 3; ++d) x[d] = __shfl(s.x0[d], src);       // the position travels with the sort
    for (int d = 0; d < 3; ++d) base[d] = (int)(x[d] * (double)D.P.inv_dx - 0.5);
    if (clamp_to_reach(D, base) && p < D.N && flag_err) atomicOr(D.err, 1);
    return p < D.N;
}

// Load this lane's particle after the wave-local sort by stencil base: p = particle index, x = position,
// base = stencil base.  Padding lanes (beyond N) sort to the end and return false.
template <class T>
__device__ __forceinline__ bool load_sorted_particle(const Dev<T>& D, const Soa<double>& X, int& p, double* x, int* base, bool flag_err = false, int wg = blockIdx.x) {
    const int p0 = wg * kBlock + threadIdx.x;
    long long key = (1LL << 40);                                    // padding lanes last
    double x0[3] = {0.5, 0.5, 0.5};
    if (p0 < D.N) {
        int b[3];
        for (int d = 0; d < 3; ++d) { x0[d] = X.ld(d, p0); b[d] = (int)(x0[d] * (double)D.P.inv_dx - 0.5); }
        key = ((long long)b[2] * D.P.n + b[1]) * D.P.n + b[0];
    }
    const int src = D.P.n <= 256 ? wave_sort_lanes32(p0 < D.N ? (unsigned)key : 0x3ffffffu) : wave_sort_lanes(key);
    p = (p0 & ~63) + src;
    // the position travels with the sort (shuffles) instead of a second, dependent trip to memory
    for (int d = 0; d < 3; ++d) x[d] = __shfl(x0[d], src);
    for (int d = 0; d < 3; ++d) base[d] = (int)(x[d] * (double)D.P.inv_dx - 0.5);
    if (clamp_to_reach(D, base) && p < D.N && flag_err) atomicOr(D.err, 1);
    return p < D.N;
}

// ------------------------------------------------------------------------------------------------
// p2g: compute_F_tmp + svd + von Mises + stress + APIC scatter      (mpm_simulator.py:82-90,157-184)
// WRITE_F: store F[f+1] (forward) or not (recompute in substep_grad).
// DET: deterministic mode -- the LDS tile and the global flush accumulate integer limbs (8 per node instead of 4
// doubles: half the tile capacity), see det_add.
template <class T, bool WRITE_F, bool DET = false>
__global__ __launch_bounds__(kBlock, sizeof(T) == 4 ? PLB_P2G_WAVES : PLB_P2G_WAVES_F64) void k_p2g(Dev<T> D, int f) {
    __shared__ int sred[kSred];
    // accumulate in double: on gfx950 ds_add_f64 is ~5x cheaper per instruction than ds_add_f32
    // (profiles/microbench/lds_atomics.hip), and the node sums lose no precision
    __shared__ Vec4<double> tile[TileCap<T>::nodes];
    const Soa<double> X = frame_xs(D, f);
    const Soa<T> R = frame_rs(D, f);
    int p, base[3];
    double x[3];
    const int wgi = xcd_chunk((int)blockIdx.x, (int)gridDim.x);
    const bool valid = load_sorted_particle(D, X, p, x, base, WRITE_F, wgi);
    Tile tl = block_tile(base, valid, sred, DET ? TileCap<T>::nodes / 2 : TileCap<T>::nodes);
    store_tile(D, f, tl, wgi);
    const int tn = tl.e[0] * tl.e[1] * tl.e[2];
    if (tl.ok) {
        for (int i = threadIdx.x; i < (DET ? 2 * tn : tn); i += kBlock) tile[i] = Vec4<double>{0.0, 0.0, 0.0, 0.0};
        __syncthreads();
    }
    {
        // every lane runs the arithmetic (padding lanes on dummy data) so the shuffles below are well defined
        T v[3] = {T(0), T(0), T(0)}, C[9], E[9], En[9];
        for (int d = 0; d < 9; ++d) { C[d] = T(0); E[d] = T(0); }
        T mu = T(1), lam = T(1), ys = T(1);
        if (valid) {
            for (int d = 0; d < 3; ++d) v[d] = R.ld(d, p);
            for (int d = 0; d < 9; ++d) { C[d] = R.ld(3 + d, p); E[d] = R.ld(12 + d, p); }
            load_materials(D, p, mu, lam, ys);
        }
        const Seg<T> sg = wave_segments<T>(valid ? (base[2] * D.P.n + base[1]) * D.P.n + base[0] : -1);
        const bool emitter = sg.head && valid;
        int b2[3];
        if (tl.ok) {
            const int ox = base[0] - tl.o[0], oy = base[1] - tl.o[1], oz = base[2] - tl.o[2];
            const int ex = tl.e[0], exy = tl.e[0] * tl.e[1];
            p2g_particle<T, double>(D.P, x, v, C, E, mu, lam, ys, En, b2, [&](int i, int j, int l, T mass, const T* mom) {
                T a0 = mass, a1 = mom[0], a2 = mom[1], a3 = mom[2];
                seg_sum4(a0, a1, a2, a3, sg);
                if (emitter) {
                    if constexpr (DET) {          // node: 4 hi limbs, then 4 lo limbs
                        long long* q = reinterpret_cast<long long*>(tile) + 8 * ((oz + l) * exy + (oy + j) * ex + (ox + i));
                        det_add(q, q + 4, (double)a0); det_add(q + 1, q + 5, (double)a1); det_add(q + 2, q + 6, (double)a2); det_add(q + 3, q + 7, (double)a3);
                    } else {
                        double* q = reinterpret_cast<double*>(&tile[(oz + l) * exy + (oy + j) * ex + (ox + i)]);
                        atomicAdd(q, (double)a0); atomicAdd(q + 1, (double)a1); atomicAdd(q + 2, (double)a2); atomicAdd(q + 3, (double)a3);
                    }
                }
            });
        } else {
            p2g_particle<T, double>(D.P, x, v, C, E, mu, lam, ys, En, b2, [&](int i, int j, int l, T mass, const T* mom) {
                T a0 = mass, a1 = mom[0], a2 = mom[1], a3 = mom[2];
                seg_sum4(a0, a1, a2, a3, sg);
                if (emitter) {
                    int idx = node_index(D, base[0] + i, base[1] + j, base[2] + l);
                    if constexpr (DET) {
                        det_add_node(D, 0, idx, (double)a0); det_add_node(D, 1, idx, (double)a1);
                        det_add_node(D, 2, idx, (double)a2); det_add_node(D, 3, idx, (double)a3);
                    } else {
                        atomicAdd(&D.gin[0][idx], a0); atomicAdd(&D.gin[1][idx], a1);
                        atomicAdd(&D.gin[2][idx], a2); atomicAdd(&D.gin[3][idx], a3);
                    }
                    mark_block(D, idx >> 6);
                }
            });
        }
        if (WRITE_F && valid) {
            const Soa<T> R1 = frame_rs(D, f + 1);
            for (int d = 0; d < 9; ++d) R1.st(12 + d, p, En[d]);
        }
    }
    if (tl.ok) {
        __syncthreads();
        const int ex = tl.e[0], exy = tl.e[0] * tl.e[1];
        for (int i = threadIdx.x; i < tn; i += kBlock) {
            if constexpr (DET) {
                long long q[8];
                long long any = 0;
                for (int c = 0; c < 8; ++c) {
                    q[c] = reinterpret_cast<const long long*>(tile)[8 * i + c];
                    any |= q[c];
                }
                if (any) {
                    int lz, ly, lx;
                    tile_coords(i, ex, exy, lz, ly, lx);
                    int idx = node_index(D, tl.o[0] + lx, tl.o[1] + ly, tl.o[2] + lz);
                    for (int c = 0; c < 4; ++c) det_flush_node(D, c, idx, q[c], q[4 + c]);
                    mark_block(D, idx >> 6);
                }
                continue;
            }
            Vec4<double> a = tile[i];
            if (a.x != 0.0 || a.y != 0.0 || a.z != 0.0 || a.w != 0.0) {
                int lz, ly, lx;
                tile_coords(i, ex, exy, lz, ly, lx);
                int idx = node_index(D, tl.o[0] + lx, tl.o[1] + ly, tl.o[2] + lz);
                atomicAdd(&D.gin[0][idx], (T)a.x); atomicAdd(&D.gin[1][idx], (T)a.y);
                atomicAdd(&D.gin[2][idx], (T)a.z); atomicAdd(&D.gin[3][idx], (T)a.w);
                mark_block(D, idx >> 6);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// grid_op (mpm_simulator.py:189-221) over active 4^3 blocks; one wave per block.
// CLEAR: forward pass -- consume grid_in (zero it and the flag for the next substep).
// XCHG: the device-side exchange of grid_m / grid_v_in is part of this launch -- send the owned blocks of the exchanged planes |
// grid_op on the blocks outside them, while the neighbours' copies are in flight | wait for the arrival | the blocks of the
// exchanged planes.  One launch instead of an exchange kernel + a grid kernel, and the interior hides the arrival.  Measured
// (round 5, a middle rank of 8 at config 3, one block plane thick = no interior at all, profiles/r05_slab_host_cost.txt): 19.5 us
// against 10.4 + 8.5 forward, 17.0 against 10.4 + 10.1 in reverse -- the launch saved is paid back by the hand-shake's own
// latency chain (store acknowledgements -> counter -> publish -> poll -> tag -> system-scope loads of the received planes: 6 us
// with nothing to hide it behind).  Two things these kernels must NOT do, both measured at +20 us per launch: keep a mutable
// copy of HaloIn (dynamic indexing sends it to LDS / scratch: the pass number is an argument of for_each_active_block instead),
// and index kernel-argument arrays with run-time indices (dependent scalar loads inside the kernel: the faces are unrolled).
template <class T, bool CLEAR, bool XCHG>
__device__ __forceinline__ void grid_op_body(const Dev<T>& D, int f, const HaloIn& H, const PeerXchg& X) {
    // the primitives of this substep: ready-made records in HBM (k_build_prims), read through uniform addresses -- no
    // assembly from the pose arrays, no LDS copy and no barrier in front of the first block (this kernel is one chain of
    // latencies: flags -> grid loads -> node arithmetic -> stores)
    const PrimT<T>* sp = D.ptab + (size_t)f * kMaxPrim;
    const int fl0 = first_flags(D);
    const int lane = threadIdx.x & 63;
    auto body = [&](int blk) {
        const int idx = (blk << 6) | lane;
        int I[3];
        block_nodes(D, blk, lane, I);
        T m = D.gin[0][idx];
        T mv[3] = {D.gin[1][idx], D.gin[2][idx], D.gin[3][idx]}, vo[3];
        const int hf = H.n > 0 ? halo_face_of(D, H, blk) : -1;
        if (hf >= 0) {
            // symmetric sum exchange: our partial sums + the neighbour's.  The complete sums go back into this
            // frame's stored grid_in (substep_grad recomputes grid_op from it), and a block that only the neighbour's
            // particles reach becomes active here too.  (The block plane of a slab that is one plane thick lies in the
            // exchange range of both faces: both neighbours' copies are added.)
            const int bz = blk / (D.nbx * D.nby);
            for (int hq = hf; hq < H.n; ++hq) {
                if (bz < H.ba[hq] || bz >= H.bb[hq] || !halo_sent(D, H, hq, blk)) continue;
                m += halo_value(D, H, hq, 0, blk, lane);
                for (int c = 0; c < 3; ++c) mv[c] += halo_value(D, H, hq, 1 + c, blk, lane);
            }
            if (!__any(m != T(0))) return;
            if (!CLEAR) {
                D.gin[0][idx] = m; D.gin[1][idx] = mv[0]; D.gin[2][idx] = mv[1]; D.gin[3][idx] = mv[2];
                if (lane == 0) D.flags[flag_slot(D, blk)] = 1;
            }
        }
        grid_node_fwd<T>(D.P, I, m, mv, D.nprim, sp, vo);
        D.grid_out[idx] = Vec4<T>{vo[0], vo[1], vo[2], T(0)};
        if (CLEAR) {
            D.gin[0][idx] = T(0); D.gin[1][idx] = T(0); D.gin[2][idx] = T(0); D.gin[3][idx] = T(0);
            if (lane == 0) D.flags[flag_slot(D, blk)] = 0;
        }
    };
    if constexpr (XCHG) {
        xchg_push_owned<T>(D, X);
        for_each_active_block<true>(D, H, 1, fl0, body);
        xchg_wait_arrived(X);
        for_each_active_block<true>(D, H, 2, fl0, body);
    } else for_each_active_block<true>(D, H, H.part, fl0, body);
}
template <class T, bool CLEAR>
__global__ __launch_bounds__(kBlock) void k_grid_op(Dev<T> D, int f, HaloIn H) { grid_op_body<T, CLEAR, false>(D, f, H, PeerXchg{}); }
template <class T>
__global__ __launch_bounds__(kBlock) void k_grid_op_x(Dev<T> D, int f, HaloIn H, PeerXchg X) { grid_op_body<T, false, true>(D, f, H, X); }

// ------------------------------------------------------------------------------------------------
// g2p (mpm_simulator.py:223-242): gather v_out through an LDS tile, write x,v,C of frame f+1
template <class T>
__global__ __launch_bounds__(kBlock) void k_g2p(Dev<T> D, int f) {
    __shared__ Vec4<T> tile[TileCap<T>::nodes];
    const int wgi = xcd_chunk((int)blockIdx.x, (int)gridDim.x);
    const int p = wgi * kBlock + threadIdx.x;
    const bool valid = p < D.N;
    const Soa<double> X = frame_xs(D, f);
    const Tile tl = load_tile(D, f, TileCap<T>::nodes, wgi);        // written by the scatter of this frame
    double x[3] = {0.5, 0.5, 0.5};
    if (valid) { x[0] = X.ld(0, p); x[1] = X.ld(1, p); x[2] = X.ld(2, p); }
    const int ex = tl.e[0], exy = tl.e[0] * tl.e[1], tn = exy * tl.e[2];
    if (tl.ok) {
        for (int i = threadIdx.x; i < tn; i += kBlock) {
            int lz, ly, lx;
                tile_coords(i, ex, exy, lz, ly, lx);
            tile[i] = D.grid_out[node_index(D, tl.o[0] + lx, tl.o[1] + ly, tl.o[2] + lz)];
        }
        __syncthreads();
    }
    int base[3];
    for (int d = 0; d < 3; ++d) base[d] = (int)(x[d] * (double)D.P.inv_dx - 0.5);
    clamp_to_reach(D, base);
    if (!valid) return;
    double xn[3];
    T vn[3], Cn[9];
#if PLB_PK_GATHER & 2
    if constexpr (sizeof(T) == 4) {                  // the gather on packed pairs (mpm_math.h: g2p_particle_pk)
        if (tl.ok) {
            const int ox = base[0] - tl.o[0], oy = base[1] - tl.o[1], oz = base[2] - tl.o[2];
            g2p_particle_pk<double>(D.P, x, xn, vn, Cn, [&](int i, int j, int l, plb_f2& axy, plb_f2& azw) {
                const Vec4<float> a = tile[(oz + l) * exy + (oy + j) * ex + (ox + i)];
                axy = pk2(a.x, a.y); azw = pk2(a.z, a.w);
            });
        } else {
            g2p_particle_pk<double>(D.P, x, xn, vn, Cn, [&](int i, int j, int l, plb_f2& axy, plb_f2& azw) {
                const Vec4<float> a = D.grid_out[node_index(D, base[0] + i, base[1] + j, base[2] + l)];
                axy = pk2(a.x, a.y); azw = pk2(a.z, a.w);
            });
        }
    } else
#endif
    if (tl.ok) {
        const int ox = base[0] - tl.o[0], oy = base[1] - tl.o[1], oz = base[2] - tl.o[2];
        g2p_particle<T, double>(D.P, x, xn, vn, Cn, [&](int i, int j, int l, T* gv) {
            Vec4<T> a = tile[(oz + l) * exy + (oy + j) * ex + (ox + i)];
            gv[0] = a.x; gv[1] = a.y; gv[2] = a.z;
        });
    } else {
        g2p_particle<T, double>(D.P, x, xn, vn, Cn, [&](int i, int j, int l, T* gv) {
            Vec4<T> a = D.grid_out[node_index(D, base[0] + i, base[1] + j, base[2] + l)];
            gv[0] = a.x; gv[1] = a.y; gv[2] = a.z;
        });
    }
    const Soa<double> X1 = frame_xs(D, f + 1);
    const Soa<T> R1 = frame_rs(D, f + 1);
    for (int d = 0; d < 3; ++d) { X1.st(d, p, xn[d]); R1.st(d, p, vn[d]); }
    for (int d = 0; d < 9; ++d) R1.st(3 + d, p, Cn[d]);
}

// ------------------------------------------------------------------------------------------------
// Fused forward kernel: g2p of substep f-1 followed by p2g of substep f for the same particle.  x,v,C of frame f
// are written once and go straight on (in registers) into the scatter; the LDS region first holds the
// grid_v_out(f-1) tile, then -- after the gather -- is reused for the f64 accumulation tile of grid_in(f).
// D is built for frame f (grid_in / flags of f); vout_prev is grid_v_out of substep f-1.
template <class T, bool DET = false>
__global__ __launch_bounds__(kBlock, sizeof(T) == 4 ? PLB_P2G_WAVES : PLB_P2G_WAVES_F64) void k_g2p_p2g(Dev<T> D, int f, const Vec4<T>* vout_prev) {
    __shared__ int sred[kSred];
    __shared__ Vec4<double> tile[TileCap<T>::nodes];
    Vec4<T>* tile_v = reinterpret_cast<Vec4<T>*>(tile);          // first use of the same LDS
    // ---------------- g2p(f-1): gather
    // lanes are sorted by the stencil base of frame f-1; particles move less than a cell per substep, so the
    // runs are (almost) the same for the scatter of frame f
    const Soa<double> X0 = frame_xs(D, f - 1);
    int p, base0[3];
    double x0[3];
    PT_BEGIN();
    // box of frame f-1 (stored by the kernel that scattered it; capacity: the same LDS bytes in Vec4<T> nodes):
    // the tile fill is issued right behind the position loads and overlaps with them and with the sort
    const int wgi = xcd_chunk((int)blockIdx.x, (int)gridDim.x);
    Tile ta = load_tile(D, f - 1, (int)(TileCap<T>::nodes * sizeof(Vec4<double>) / sizeof(Vec4<T>)), wgi);
    SortLoad sl = sorted_begin(D, X0, wgi);
    {
        const int ex = ta.e[0], exy = ta.e[0] * ta.e[1], tn = exy * ta.e[2];
        if (ta.ok)
            for (int i = threadIdx.x; i < tn; i += kBlock) {
                int lz, ly, lx;
                tile_coords(i, ex, exy, lz, ly, lx);
                tile_v[i] = vout_prev[node_index(D, ta.o[0] + lx, ta.o[1] + ly, ta.o[2] + lz)];
            }
    }
    PT_MARK(0);
    const bool valid = sorted_finish(D, sl, p, x0, base0, false, wgi);
    // state that does not depend on the gather: issue these loads now so they fly during the gather
    T E[9];
    for (int d = 0; d < 9; ++d) E[d] = T(0);
    T mu = T(1), lam = T(1), ys = T(1);
    const Soa<T> R1 = frame_rs(D, f);                            // frame f: E is read, x / v / C are written
    if (valid) {
        for (int d = 0; d < 9; ++d) E[d] = R1.ld(12 + d, p);
        load_materials(D, p, mu, lam, ys);
    }
    wg_barrier();                                               // tile_v complete
    PT_MARK(1);
    double x[3] = {0.5, 0.5, 0.5};
    T v[3] = {T(0), T(0), T(0)}, C[9];
    for (int d = 0; d < 9; ++d) C[d] = T(0);
    if (valid) {
#if PLB_PK_GATHER & 2
        if constexpr (sizeof(T) == 4) {              // the gather on packed pairs (mpm_math.h: g2p_particle_pk)
            if (ta.ok) {
                const int ex = ta.e[0], exy = ta.e[0] * ta.e[1];
                const int ox = base0[0] - ta.o[0], oy = base0[1] - ta.o[1], oz = base0[2] - ta.o[2];
                g2p_particle_pk<double>(D.P, x0, x, v, C, [&](int i, int j, int l, plb_f2& axy, plb_f2& azw) {
                    const Vec4<float> a = tile_v[(oz + l) * exy + (oy + j) * ex + (ox + i)];
                    axy = pk2(a.x, a.y); azw = pk2(a.z, a.w);
                });
            } else {
                g2p_particle_pk<double>(D.P, x0, x, v, C, [&](int i, int j, int l, plb_f2& axy, plb_f2& azw) {
                    const Vec4<float> a = vout_prev[node_index(D, base0[0] + i, base0[1] + j, base0[2] + l)];
                    axy = pk2(a.x, a.y); azw = pk2(a.z, a.w);
                });
            }
        } else
#endif
        if (ta.ok) {
            const int ex = ta.e[0], exy = ta.e[0] * ta.e[1];
            const int ox = base0[0] - ta.o[0], oy = base0[1] - ta.o[1], oz = base0[2] - ta.o[2];
            g2p_particle<T, double>(D.P, x0, x, v, C, [&](int i, int j, int l, T* gv) {
                Vec4<T> a = tile_v[(oz + l) * exy + (oy + j) * ex + (ox + i)];
                gv[0] = a.x; gv[1] = a.y; gv[2] = a.z;
            });
        } else {
            g2p_particle<T, double>(D.P, x0, x, v, C, [&](int i, int j, int l, T* gv) {
                Vec4<T> a = vout_prev[node_index(D, base0[0] + i, base0[1] + j, base0[2] + l)];
                gv[0] = a.x; gv[1] = a.y; gv[2] = a.z;
            });
        }
        const Soa<double> X1 = frame_xs(D, f);
        for (int d = 0; d < 3; ++d) { X1.st(d, p, x[d]); R1.st(d, p, v[d]); }
        for (int d = 0; d < 9; ++d) R1.st(3 + d, p, C[d]);
    }
    PT_MARK(2);
    // ---------------- p2g(f): scatter
    int base[3];
    for (int d = 0; d < 3; ++d) base[d] = (int)(x[d] * (double)D.P.inv_dx - 0.5);
    if (clamp_to_reach(D, base) && valid) atomicOr(D.err, 1);
    block_tile_publish(base, valid, sred);
    wg_barrier();                                                        // everyone is done reading tile_v, and has published its box
    Tile tl = block_tile_collect(sred, DET ? TileCap<T>::nodes / 2 : TileCap<T>::nodes);
    store_tile(D, f, tl, wgi);
    const int tn = tl.e[0] * tl.e[1] * tl.e[2];
#ifdef PLB_DEBUG_COUNTERS      // tile statistics for plmpm_debug_counters (same-address atomics: keep out of production builds)
    if (threadIdx.x == 0) { atomicAdd(D.err + (tl.ok ? 2 : 1), 1); if (tl.ok) atomicAdd(D.err + 3, tn); }
#endif
    if (tl.ok) {
        for (int i = threadIdx.x; i < (DET ? 2 * tn : tn); i += kBlock) tile[i] = Vec4<double>{0.0, 0.0, 0.0, 0.0};
        wg_barrier();
    }
    PT_MARK(3);
    {
        T En[9];
        const Seg<T> sg = wave_segments<T>(valid ? (base[2] * D.P.n + base[1]) * D.P.n + base[0] : -1);
        const bool emitter = sg.head && valid;
        int b2[3];
        if (tl.ok) {
            const int ox = base[0] - tl.o[0], oy = base[1] - tl.o[1], oz = base[2] - tl.o[2];
            const int ex = tl.e[0], exy = tl.e[0] * tl.e[1];
            P2GCoef<T> K;
            T w[3][3];
            p2g_prepare<T, double>(D.P, x, v, C, E, mu, lam, ys, En, b2, w, K);          // (= p2g_particle, with a phase mark in between)
            PT_MARK(4);
            p2g_emit<T>(w, K, [&](int i, int j, int l, T mass, const T* mom) {
                T a0 = mass, a1 = mom[0], a2 = mom[1], a3 = mom[2];
                seg_sum4(a0, a1, a2, a3, sg);
                if (emitter) {
                    if constexpr (DET) {          // node: 4 hi limbs, then 4 lo limbs
                        long long* q = reinterpret_cast<long long*>(tile) + 8 * ((oz + l) * exy + (oy + j) * ex + (ox + i));
                        det_add(q, q + 4, (double)a0); det_add(q + 1, q + 5, (double)a1); det_add(q + 2, q + 6, (double)a2); det_add(q + 3, q + 7, (double)a3);
                    } else {
                        double* q = reinterpret_cast<double*>(&tile[(oz + l) * exy + (oy + j) * ex + (ox + i)]);
                        atomicAdd(q, (double)a0); atomicAdd(q + 1, (double)a1); atomicAdd(q + 2, (double)a2); atomicAdd(q + 3, (double)a3);
                    }
                }
            });
        } else {
            p2g_particle<T, double>(D.P, x, v, C, E, mu, lam, ys, En, b2, [&](int i, int j, int l, T mass, const T* mom) {
                T a0 = mass, a1 = mom[0], a2 = mom[1], a3 = mom[2];
                seg_sum4(a0, a1, a2, a3, sg);
                if (emitter) {
                    int idx = node_index(D, base[0] + i, base[1] + j, base[2] + l);
                    if constexpr (DET) {
                        det_add_node(D, 0, idx, (double)a0); det_add_node(D, 1, idx, (double)a1);
                        det_add_node(D, 2, idx, (double)a2); det_add_node(D, 3, idx, (double)a3);
                    } else {
                        atomicAdd(&D.gin[0][idx], a0); atomicAdd(&D.gin[1][idx], a1);
                        atomicAdd(&D.gin[2][idx], a2); atomicAdd(&D.gin[3][idx], a3);
                    }
                    mark_block(D, idx >> 6);
                }
            });
        }
        if (valid) {
            const Soa<T> R2 = frame_rs(D, f + 1);
            for (int d = 0; d < 9; ++d) R2.st(12 + d, p, En[d]);
        }
    }
    PT_MARK(5);
    if (tl.ok) {
        wg_barrier();
        const int ex = tl.e[0], exy = tl.e[0] * tl.e[1];
        for (int i = threadIdx.x; i < tn; i += kBlock) {
            if constexpr (DET) {
                long long q[8];
                long long any = 0;
                for (int c = 0; c < 8; ++c) {
                    q[c] = reinterpret_cast<const long long*>(tile)[8 * i + c];
                    any |= q[c];
                }
                if (any) {
                    int lz, ly, lx;
                    tile_coords(i, ex, exy, lz, ly, lx);
                    int idx = node_index(D, tl.o[0] + lx, tl.o[1] + ly, tl.o[2] + lz);
                    for (int c = 0; c < 4; ++c) det_flush_node(D, c, idx, q[c], q[4 + c]);
                    mark_block(D, idx >> 6);
                }
                continue;
            }
            Vec4<double> a = tile[i];
            if (a.x != 0.0 || a.y != 0.0 || a.z != 0.0 || a.w != 0.0) {
                int lz, ly, lx;
                tile_coords(i, ex, exy, lz, ly, lx);
                int idx = node_index(D, tl.o[0] + lx, tl.o[1] + ly, tl.o[2] + lz);
                atomicAdd(&D.gin[0][idx], (T)a.x); atomicAdd(&D.gin[1][idx], (T)a.y);
                atomicAdd(&D.gin[2][idx], (T)a.z); atomicAdd(&D.gin[3][idx], (T)a.w);
                mark_block(D, idx >> 6);
            }
        }
    }
    PT_MARK(6);
    PT_END(D, 0);
}

// ------------------------------------------------------------------------------------------------
// g2p.grad: scatter grid_v_out.grad, x[f].grad partial -> adjoint frame `dst`.  vnext: see below (nullptr = frame f+1)
// g2p.grad at 4 waves per SIMD: with the j loop of g2p_particle_grad rolled (PLB_ROLL_G2PG_J, mpm_math.h) the fp32 kernel
// needs 128 VGPRs without scratch instead of 168, and four 37.5 KiB workgroups fit a CU's LDS: 37.0 -> 35.0 us
// (round 2; 5 waves = 96 VGPRs + 100 B scratch and a smaller tile was not faster)
#ifndef PLB_G2PG_WAVES
#define PLB_G2PG_WAVES 4
#endif
#ifndef PLB_G2PG_CAP
#define PLB_G2PG_CAP 960
#endif
#ifndef PLB_G2PG_FIX8
#define PLB_G2PG_FIX8 1
#endif
template <class T, bool DET = false>
__global__ __launch_bounds__(kBlock, sizeof(T) == 4 ? PLB_G2PG_WAVES : 1) void k_g2p_grad(Dev<T> D, int f, int src, int dst, const T* vnext) {
    const int wg = xcd_chunk((int)blockIdx.x, (int)gridDim.x);
    // 960 nodes x (16 + 24) bytes = 37.5 KiB: four workgroups per CU (128 VGPRs = 4 waves per SIMD, see PLB_G2PG_WAVES)
    constexpr int CAP = sizeof(T) == 4 ? PLB_G2PG_CAP : 480;
    __shared__ Vec4<T> tile[CAP];                    // v_out values
    __shared__ double tile_a[CAP * 3];               // v_out adjoint accumulation (f64, see k_p2g)
    const Soa<double> X = frame_xs(D, f);
    int p, base[3];
    double x[3];
    PT_BEGIN();
    // (in front of the kernel's first store: behind one, hipcc no longer proves the descriptor unclobbered and fetches it with a vector
    // load + readfirstlane instead of s_load)
    const Tile tl = load_tile(D, f, DET ? CAP / 2 : CAP, wg);       // stored by the scatter of this frame (DET: 6 limbs per node in tile_a)
    if (wg == 0 && threadIdx.x == 0) D.contact[0] = 0;     // the list k_grid_op_grad(f) is about to fill
    SortLoad sl = sorted_begin(D, X, wg);

    const int ex = tl.e[0], exy = tl.e[0] * tl.e[1], tn = exy * tl.e[2];
    // Fixed-shape tile: a box of at most 8 nodes per axis (most are: the mean box is ~6^3 nodes) is laid out in LDS with
    // the CONSTANT strides 8 / 64 instead of its own extents -- the 27 + 27 tile addresses of a particle's stencil become
    // immediates off one base register, the fill / flush loops get their node coordinates from shifts instead of
    // divisions.  fp32 engine (the f64 tiles hold fewer than 512 nodes); workgroup-uniform choice.  Measured (round 2):
    // 35.0 -> 34.3 us here; the same in k_g2p_p2g's gather and in k_p2g_grad changed nothing and in k_g2p_p2g's scatter
    // cost 2 us (512 instead of ~230 tile slots to zero and flush), so only this kernel has it.
    const bool fix8 = PLB_G2PG_FIX8 && !DET && sizeof(T) == 4 && CAP >= 512 && tl.ok && tl.e[0] <= 8 && tl.e[1] <= 8 && tl.e[2] <= 8;
    const int n8 = tl.e[2] << 6;                                    // slots of the z planes in use
    if (fix8) {
        for (int i = threadIdx.x; i < n8; i += kBlock) {
            const int lx = i & 7, ly = (i >> 3) & 7, lz = i >> 6;
            if (lx < tl.e[0] && ly < tl.e[1] && lz < tl.e[2])
                tile[i] = D.grid_out[node_index(D, tl.o[0] + lx, tl.o[1] + ly, tl.o[2] + lz)];
            tile_a[3 * i] = 0.0; tile_a[3 * i + 1] = 0.0; tile_a[3 * i + 2] = 0.0;
        }
    } else if (tl.ok)
        for (int i = threadIdx.x; i < tn; i += kBlock) {
            int lz, ly, lx;
                tile_coords(i, ex, exy, lz, ly, lx);
            tile[i] = D.grid_out[node_index(D, tl.o[0] + lx, tl.o[1] + ly, tl.o[2] + lz)];
            tile_a[3 * i] = 0.0; tile_a[3 * i + 1] = 0.0; tile_a[3 * i + 2] = 0.0;
            if (DET) { tile_a[3 * (tn + i)] = 0.0; tile_a[3 * (tn + i) + 1] = 0.0; tile_a[3 * (tn + i) + 2] = 0.0; }   // lo limbs
        }
    PT_MARK(0);
    const bool valid = sorted_finish(D, sl, p, x, base, false, wg);
    // padding lanes run the gather / scatter arithmetic on dummy data (the wave-level reductions need every lane): their stencil is
    // parked on the tile's origin, so that what they read lies inside the tile (an LDS read outside the allocation returns zero on
    // the hardware and nothing of it is kept -- but it is an out-of-bounds index all the same: found by the sanitizer run of the
    // CPU interpreter, round 6)
    if (!valid) { base[0] = tl.o[0]; base[1] = tl.o[1]; base[2] = tl.o[2]; }
    PT_MARK(1);
    {
        // v[f+1]: normally the stored frame; after a re-sort of frame f+1 the copy kept in this frame's particle order
        const Soa<T> R1(vnext ? vnext : frame_r(D, f + 1), D.Npad, vnext ? 3 : 21);
        const Soa<T> A1 = adj_s(D, src);
        T vn[3] = {T(0), T(0), T(0)}, xna[3] = {T(0), T(0), T(0)}, vna[3] = {T(0), T(0), T(0)}, Cna[9], xa[3];
        for (int d = 0; d < 9; ++d) Cna[d] = T(0);
        if (valid) {
            for (int d = 0; d < 3; ++d) { vn[d] = R1.ld(d, p); xna[d] = A1.ld(d, p); vna[d] = A1.ld(3 + d, p); }
            for (int d = 0; d < 9; ++d) Cna[d] = A1.ld(6 + d, p);
        }
        __syncthreads();                             // tile / tile_a complete (the loads above are in flight)
        PT_MARK(2);
        const Seg<T> sg = wave_segments<T>(valid ? (base[2] * D.P.n + base[1]) * D.P.n + base[0] : -1);
        const bool emitter = sg.head && valid;
        if constexpr (DET) {
            // deterministic mode: integer limbs in tile_a (hi at [3 i + c], lo at [3 (tn + i) + c]) or, without a tile,
            // in the global limb grid
            const int ox = base[0] - tl.o[0], oy = base[1] - tl.o[1], oz = base[2] - tl.o[2];
            long long* la = reinterpret_cast<long long*>(tile_a);
            g2p_particle_grad<T, double>(D.P, x, vn, xna, vna, Cna, xa,
                [&](int i, int j, int l, T* gv) {
                    gv[0] = gv[1] = gv[2] = T(0);
                    if (tl.ok) {
                        Vec4<T> a = tile[(oz + l) * exy + (oy + j) * ex + (ox + i)];
                        gv[0] = a.x; gv[1] = a.y; gv[2] = a.z;
                    } else if (valid) {
                        Vec4<T> a = D.grid_out[node_index(D, base[0] + i, base[1] + j, base[2] + l)];
                        gv[0] = a.x; gv[1] = a.y; gv[2] = a.z;
                    }
                },
                [&](int i, int j, int l, const T* ga) {
                    T a0 = ga[0], a1 = ga[1], a2 = ga[2];
                    seg_sum3(a0, a1, a2, sg);
                    if (!emitter) return;
                    if (tl.ok) {
                        const int node = (oz + l) * exy + (oy + j) * ex + (ox + i);
                        long long *h = la + 3 * node, *lo = la + 3 * (tn + node);
                        det_add(h, lo, (double)a0); det_add(h + 1, lo + 1, (double)a1); det_add(h + 2, lo + 2, (double)a2);
                    } else {
                        int idx = node_index(D, base[0] + i, base[1] + j, base[2] + l);
                        det_add_node(D, 0, idx, (double)a0); det_add_node(D, 1, idx, (double)a1); det_add_node(D, 2, idx, (double)a2);
                    }
                });
        } else if (fix8) {
            const int b0 = ((base[2] - tl.o[2]) << 6) + ((base[1] - tl.o[1]) << 3) + (base[0] - tl.o[0]);
            const Vec4<T>* tb = tile + b0;
            double* ab = tile_a + 3 * b0;
            g2p_particle_grad<T, double>(D.P, x, vn, xna, vna, Cna, xa,
                [&](int i, int j, int l, T* gv) {
                    Vec4<T> a = tb[(l << 6) + (j << 3) + i];
                    gv[0] = a.x; gv[1] = a.y; gv[2] = a.z;
                },
                [&](int i, int j, int l, const T* ga) {
                    T a0 = ga[0], a1 = ga[1], a2 = ga[2];
                    seg_sum3(a0, a1, a2, sg);
                    if (emitter) {
                        double* q = ab + 3 * ((l << 6) + (j << 3) + i);
                        atomicAdd(q, (double)a0); atomicAdd(q + 1, (double)a1); atomicAdd(q + 2, (double)a2);
                    }
                });
        } else if (tl.ok) {
            const int ox = base[0] - tl.o[0], oy = base[1] - tl.o[1], oz = base[2] - tl.o[2];
            g2p_particle_grad<T, double>(D.P, x, vn, xna, vna, Cna, xa,
                [&](int i, int j, int l, T* gv) {
                    Vec4<T> a = tile[(oz + l) * exy + (oy + j) * ex + (ox + i)];
                    gv[0] = a.x; gv[1] = a.y; gv[2] = a.z;
                },
                [&](int i, int j, int l, const T* ga) {
                    T a0 = ga[0], a1 = ga[1], a2 = ga[2];
                    seg_sum3(a0, a1, a2, sg);
                    if (emitter) {
                        double* q = &tile_a[3 * ((oz + l) * exy + (oy + j) * ex + (ox + i))];
                        atomicAdd(q, (double)a0); atomicAdd(q + 1, (double)a1); atomicAdd(q + 2, (double)a2);
                    }
                });
        } else {
            g2p_particle_grad<T, double>(D.P, x, vn, xna, vna, Cna, xa,
                [&](int i, int j, int l, T* gv) {
                    gv[0] = gv[1] = gv[2] = T(0);
                    if (valid) {
                        Vec4<T> a = D.grid_out[node_index(D, base[0] + i, base[1] + j, base[2] + l)];
                        gv[0] = a.x; gv[1] = a.y; gv[2] = a.z;
                    }
                },
                [&](int i, int j, int l, const T* ga) {
                    T a0 = ga[0], a1 = ga[1], a2 = ga[2];
                    seg_sum3(a0, a1, a2, sg);
                    if (emitter) {
                        int idx = node_index(D, base[0] + i, base[1] + j, base[2] + l);
                        atomicAdd(&D.goa[0][idx], a0); atomicAdd(&D.goa[1][idx], a1); atomicAdd(&D.goa[2][idx], a2);
                    }
                });
        }
        if (valid) {
            const Soa<T> A0 = adj_s(D, dst);
            for (int d = 0; d < 3; ++d) A0.st(d, p, xa[d]);
        }
    }
    PT_MARK(3);
    if (fix8) {
        __syncthreads();
        for (int i = threadIdx.x; i < n8; i += kBlock) {
            const double ax = tile_a[3 * i], ay = tile_a[3 * i + 1], az = tile_a[3 * i + 2];
            if (ax != 0.0 || ay != 0.0 || az != 0.0) {
                int idx = node_index(D, tl.o[0] + (i & 7), tl.o[1] + ((i >> 3) & 7), tl.o[2] + (i >> 6));
                atomicAdd(&D.goa[0][idx], (T)ax); atomicAdd(&D.goa[1][idx], (T)ay); atomicAdd(&D.goa[2][idx], (T)az);
            }
        }
    } else if (tl.ok) {
        __syncthreads();
        for (int i = threadIdx.x; i < tn; i += kBlock) {
            if constexpr (DET) {
                const long long* h = reinterpret_cast<const long long*>(tile_a) + 3 * i;
                const long long* lo = reinterpret_cast<const long long*>(tile_a) + 3 * (tn + i);
                if (h[0] | h[1] | h[2] | lo[0] | lo[1] | lo[2]) {
                    int lz, ly, lx;
                    tile_coords(i, ex, exy, lz, ly, lx);
                    int idx = node_index(D, tl.o[0] + lx, tl.o[1] + ly, tl.o[2] + lz);
                    for (int c = 0; c < 3; ++c) det_flush_node(D, c, idx, h[c], lo[c]);
                }
                continue;
            }
            const double ax = tile_a[3 * i], ay = tile_a[3 * i + 1], az = tile_a[3 * i + 2];
            if (ax != 0.0 || ay != 0.0 || az != 0.0) {
                int lz, ly, lx;
                tile_coords(i, ex, exy, lz, ly, lx);
                int idx = node_index(D, tl.o[0] + lx, tl.o[1] + ly, tl.o[2] + lz);
                atomicAdd(&D.goa[0][idx], (T)ax); atomicAdd(&D.goa[1][idx], (T)ay); atomicAdd(&D.goa[2][idx], (T)az);
            }
        }
    }
    PT_MARK(4);
    PT_END(D, 10);
}

// ------------------------------------------------------------------------------------------------
// grid_op.grad: grid_out_adj -> grid_in_adj, pose adjoints; clears grid_out_adj, grid_in and the block flag.
// One wave, one 4^3 block.  POSE = false computes the velocity adjoint only and returns true when an owned node of
// the block touches a movable primitive: its pose adjoints are then still due and the block's inputs are left in
// place for the POSE = true pass, which clears them.
template <class T, bool POSE>
__device__ __forceinline__ bool grid_block_bwd(const Dev<T>& D, const HaloIn& H, int blk, int lane, const PrimT<T>* sp, double* sacc, int* shit) {
    const int idx = (blk << 6) | lane;
    int I[3];
    block_nodes(D, blk, lane, I);
    T gm = D.gin[0][idx];
    T mv[3] = {D.gin[1][idx], D.gin[2][idx], D.gin[3][idx]};
    T va[3] = {D.goa[0][idx], D.goa[1][idx], D.goa[2][idx]}, ma, mva[3];
    // multi-GPU: add the neighbour's share of grid_v_out.grad on the exchanged planes (POSE = false pass only: a
    // deferred block gets the sum written back below, so the POSE = true pass reads complete values)
    const int hf = (!POSE && H.n > 0) ? halo_face_of(D, H, blk) : -1;
    if (hf >= 0) {
        const int bz = blk / (D.nbx * D.nby);
        for (int hq = hf; hq < H.n; ++hq) {
            if (bz < H.ba[hq] || bz >= H.bb[hq] || !halo_sent(D, H, hq, blk)) continue;
            for (int c = 0; c < 3; ++c) va[c] += halo_value(D, H, hq, c, blk, lane);
        }
    }
    const bool owned = I[2] >= D.z0 && I[2] < D.z1;     // halo nodes are computed on two ranks: count once
    bool due = false;
    grid_node_bwd<T, POSE>(D.P, I, gm, mv, D.nprim, sp, va, &ma, mva, [&](int q, const PoseAdj<T>& pa, bool hit) {
        // every lane of the wave gets here for every primitive: sum the 15 pose-adjoint components across the
        // wave (DPP scans) and let one lane touch LDS (64 lanes hitting the same 14 addresses with
        // ds_add_f64 serialise badly)
        const bool h = hit && owned;
        if constexpr (!POSE) { due |= h; return; }
        else {
            if (!__any(h)) return;
            double vals[15];
            for (int d = 0; d < 3; ++d) { vals[d] = h ? pa.pos[d] : 0.0; vals[7 + d] = h ? pa.pos1[d] : 0.0; }
            for (int d = 0; d < 4; ++d) { vals[3 + d] = h ? pa.rot[d] : 0.0; vals[10 + d] = h ? pa.rot1[d] : 0.0; }
            vals[14] = h ? pa.gap : 0.0;
            const int nc = sp[q].shape == SHAPE_CHOPSTICKS ? 15 : 14;
            for (int c = 0; c < nc; ++c) vals[c] = wave_sum_to_lane63(vals[c]);
            if (lane == 63) {
                double* o = &sacc[q * 15];
                for (int c = 0; c < nc; ++c) atomicAdd(&o[c], vals[c]);
                *shit = 1;
            }
        }
    });
    const bool defer = !POSE && __any(due);
    if (!POSE) D.grid_in_adj[idx] = Vec4<T>{mva[0], mva[1], mva[2], ma};      // vector part first: (x, y) is an aligned register pair for the packed gather
    if (defer && hf >= 0) { D.goa[0][idx] = va[0]; D.goa[1][idx] = va[1]; D.goa[2][idx] = va[2]; }
    if (!defer) {
        D.goa[0][idx] = T(0); D.goa[1][idx] = T(0); D.goa[2][idx] = T(0);
        // this frame's grid is consumed: leave grid_in / flags clean for the next scatter into them.  grid_in_adj
        // is never cleared -- p2g.grad only reads nodes of active blocks, which are all rewritten every substep.
        D.gin[0][idx] = T(0); D.gin[1][idx] = T(0); D.gin[2][idx] = T(0); D.gin[3][idx] = T(0);
        if (lane == 0) D.flags[flag_slot(D, blk)] = 0;
    }
    return defer;
}

// Persistent over the active blocks.  The double-precision pose adjoints of the blocks in contact (a few dozen waves,
// several microseconds each: the whole tail of this kernel when done here) are handed to spare workgroups of the
// p2g.grad launch that follows, through D.contact.
#ifndef PLB_GOG_WAVES
#define PLB_GOG_WAVES 1          // 4 (128 VGPRs + 60 B scratch) measured: 14.1 -> 17.8 us
#endif
template <class T, bool XCHG>
__device__ __forceinline__ void grid_op_grad_body(const Dev<T>& D, int f, const HaloIn& H, const PeerXchg& X) {
    const PrimT<T>* sp = D.ptab + (size_t)f * kMaxPrim;          // see k_grid_op
    const int fl0 = first_flags(D);
    const int lane = threadIdx.x & 63;
    // the forward grid_op marked every block of the exchanged planes that carries mass, so the flags alone are
    // complete here: no halo candidates
    auto body = [&](int blk) {
        if (grid_block_bwd<T, false>(D, H, blk, lane, sp, nullptr, nullptr) && lane == 0)
            D.contact[1 + atomicAdd(&D.contact[0], 1)] = blk;
    };
    if constexpr (XCHG) {              // the exchange of grid_v_out.grad folded in, exactly as in k_grid_op_x
        xchg_push_owned<T>(D, X);
        for_each_active_block<false>(D, H, 1, fl0, body);
        xchg_wait_arrived(X);
        for_each_active_block<false>(D, H, 2, fl0, body);
    } else for_each_active_block<false>(D, H, H.part, fl0, body);
}
template <class T>
__global__ __launch_bounds__(kBlock, sizeof(T) == 4 ? PLB_GOG_WAVES : 1) void k_grid_op_grad(Dev<T> D, int f, HaloIn H) {
    grid_op_grad_body<T, false>(D, f, H, PeerXchg{});
}
template <class T>
__global__ __launch_bounds__(kBlock, sizeof(T) == 4 ? PLB_GOG_WAVES : 1) void k_grid_op_grad_x(Dev<T> D, int f, HaloIn H, PeerXchg X) {
    grid_op_grad_body<T, true>(D, f, H, X);
}

// the pose-adjoint workgroups of the p2g.grad launch: blocks listed in D.contact, one wave each
template <class T>
__device__ __forceinline__ void pose_adjoint_blocks(const Dev<T>& D, int f, int wg, int nwg, PrimT<T>* sp) {
    __shared__ double sacc[kMaxPrim * 15];
    __shared__ int shit;
    load_prims(D, f, sp);
    if (threadIdx.x < kMaxPrim * 15) sacc[threadIdx.x] = 0.0;
    if (threadIdx.x == 0) shit = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, count = D.contact[0];
    for (int i = wg * (kBlock / 64) + (threadIdx.x >> 6); i < count; i += nwg * (kBlock / 64))
        grid_block_bwd<T, true>(D, HaloIn{}, D.contact[1 + i], lane, sp, sacc, &shit);
    __syncthreads();
    if (shit && threadIdx.x < D.nprim * 15) {
        int q = threadIdx.x / 15, c = threadIdx.x % 15;
        double v = sacc[threadIdx.x];
        if (v != 0.0) {
            // c: 0-2 pos[f], 3-6 rot[f], 7-9 pos[f+1], 10-13 rot[f+1], 14 gap[f]
            if (c < 3) atomicAdd(&D.ppos_a[((size_t)f * D.nprim + q) * 3 + c], v);
            else if (c < 7) atomicAdd(&D.prot_a[((size_t)f * D.nprim + q) * 4 + (c - 3)], v);
            else if (c < 10) atomicAdd(&D.ppos_a[((size_t)(f + 1) * D.nprim + q) * 3 + (c - 7)], v);
            else if (c < 14) atomicAdd(&D.prot_a[((size_t)(f + 1) * D.nprim + q) * 4 + (c - 10)], v);
            else atomicAdd(&D.pgap_a[(size_t)f * D.nprim + q], v);
        }
    }
}

// Deterministic mode: the list of blocks in contact is filled through an atomic counter, so its order -- and with it
// the order of the floating-point pose sums -- varies from run to run.  Here ONE wave walks the list in increasing
// block order (k_p2g_grad is then launched without pose workgroups).
template <class T>
__global__ __launch_bounds__(64) void k_pose_adjoint_det(Dev<T> D, int f) {
    __shared__ PrimT<T> sp[kMaxPrim];
    __shared__ double sacc[kMaxPrim * 15];
    __shared__ int shit;
    load_prims(D, f, sp);
    for (int i = threadIdx.x; i < kMaxPrim * 15; i += 64) sacc[i] = 0.0;
    if (threadIdx.x == 0) shit = 0;
    __syncthreads();
    const int lane = threadIdx.x, count = D.contact[0];
    int last = -1;
    for (int it = 0; it < count; ++it) {
        int best = 0x7fffffff;
        for (int i = lane; i < count; i += 64) { const int b = D.contact[1 + i]; if (b > last && b < best) best = b; }
        best = wave_min(best);
        grid_block_bwd<T, true>(D, HaloIn{}, best, lane, sp, sacc, &shit);
        last = best;
    }
    __syncthreads();
    if (!shit) return;
    for (int t = threadIdx.x; t < D.nprim * 15; t += 64) {
        const int q = t / 15, c = t % 15;
        const double v = sacc[t];
        if (v == 0.0) continue;
        if (c < 3) D.ppos_a[((size_t)f * D.nprim + q) * 3 + c] += v;
        else if (c < 7) D.prot_a[((size_t)f * D.nprim + q) * 4 + (c - 3)] += v;
        else if (c < 10) D.ppos_a[((size_t)(f + 1) * D.nprim + q) * 3 + (c - 7)] += v;
        else if (c < 14) D.prot_a[((size_t)(f + 1) * D.nprim + q) * 4 + (c - 10)] += v;
        else D.pgap_a[(size_t)f * D.nprim + q] += v;
    }
}

// ------------------------------------------------------------------------------------------------
// p2g.grad + svd_grad + compute_F_tmp.grad: gather grid_in_adj, finish adjoint frame `dst`
// (Fetching the particle state this kernel needs behind its gather at the START by LDS-DMA -- global_load_lds, 36 words per
// particle -- was built and measured in round 4: parity-green, 47.1 / 49.0 us against 42.7 / 43.3; profiles/r04_ablation_hooks.patch.)
template <class T>
__global__ __launch_bounds__(kBlock, sizeof(T) == 4 ? PLB_P2G_GRAD_WAVES : 1) void k_p2g_grad(Dev<T> D, int f, int src, int dst, int npose) {
    __shared__ Vec4<T> tile[TileCap<T>::nodes];
    __shared__ PrimT<T> sp[kMaxPrim];
    // the first `npose` workgroups finish grid_op.grad (pose adjoints of the blocks in contact) under cover of the
    // particle workgroups
    if ((int)blockIdx.x < npose) { pose_adjoint_blocks<T>(D, f, (int)blockIdx.x, npose, sp); return; }
    const int chunk = xcd_chunk((int)blockIdx.x - npose, (int)gridDim.x - npose);
    const int p = chunk * kBlock + threadIdx.x;
    const bool valid = p < D.N;
    const Soa<double> X = frame_xs(D, f);
    const Soa<T> R = frame_rs(D, f);
    PT_BEGIN();
    Tile tl = load_tile(D, f, TileCap<T>::nodes, chunk);   // stored by the scatter of this frame
    double x[3] = {0.5, 0.5, 0.5};
    if (valid) { x[0] = X.ld(0, p); x[1] = X.ld(1, p); x[2] = X.ld(2, p); }
    const int ex = tl.e[0], exy = tl.e[0] * tl.e[1], tn = exy * tl.e[2];
    if (tl.ok) {
        for (int i = threadIdx.x; i < tn; i += kBlock) {
            int lz, ly, lx;
                tile_coords(i, ex, exy, lz, ly, lx);
            tile[i] = D.grid_in_adj[node_index(D, tl.o[0] + lx, tl.o[1] + ly, tl.o[2] + lz)];
        }
        __syncthreads();
    }
    int base[3];
    for (int d = 0; d < 3; ++d) base[d] = (int)(x[d] * (double)D.P.inv_dx - 0.5);
    clamp_to_reach(D, base);
    PT_MARK(1);
    if (!valid) return;
    // the 27-node gather needs the position only: the other 42 words of particle state are fetched after it, so
    // that they are not live across the loop (the kernel then fits 3 waves per SIMD instead of 2)
    P2GGather<T> G;
#if PLB_PK_GATHER & 1
    // fp32 engine: the gather on packed pairs (mpm_math.h: p2g_gather_grad_pk) -- 381 v_pk_fma_f32 + 39 plain instead of ~800 plain
    if constexpr (sizeof(T) == 4) {
        if (tl.ok) {
            const int ox = base[0] - tl.o[0], oy = base[1] - tl.o[1], oz = base[2] - tl.o[2];
            p2g_gather_grad_pk<double>(D.P, x, G, [&](int i, int j, int l, plb_f2& axy, plb_f2& azw) {
                const Vec4<float> a = tile[(oz + l) * exy + (oy + j) * ex + (ox + i)];
                axy = pk2(a.x, a.y); azw = pk2(a.z, a.w);
            });
        } else {
            p2g_gather_grad_pk<double>(D.P, x, G, [&](int i, int j, int l, plb_f2& axy, plb_f2& azw) {
                const Vec4<float> a = D.grid_in_adj[node_index(D, base[0] + i, base[1] + j, base[2] + l)];
                axy = pk2(a.x, a.y); azw = pk2(a.z, a.w);
            });
        }
    } else
#endif
    if (tl.ok) {
        const int ox = base[0] - tl.o[0], oy = base[1] - tl.o[1], oz = base[2] - tl.o[2];
        p2g_gather_grad<T, double>(D.P, x, G, [&](int i, int j, int l, T* g) {
            Vec4<T> a = tile[(oz + l) * exy + (oy + j) * ex + (ox + i)];
            g[0] = a.w; g[1] = a.x; g[2] = a.y; g[3] = a.z;
        });
    } else {
        p2g_gather_grad<T, double>(D.P, x, G, [&](int i, int j, int l, T* g) {
            Vec4<T> a = D.grid_in_adj[node_index(D, base[0] + i, base[1] + j, base[2] + l)];
            g[0] = a.w; g[1] = a.x; g[2] = a.y; g[3] = a.z;
        });
    }
    PT_MARK(2);
    T v[3], C[9], E[9], Ena[9], xa[3], va[3], Ca[9], Ea[9];
    const Soa<T> A1 = adj_s(D, src), A0 = adj_s(D, dst);
    T mu, lam, ys;
    for (int d = 0; d < 3; ++d) { v[d] = R.ld(d, p); xa[d] = A0.ld(d, p); }
    for (int d = 0; d < 9; ++d) { C[d] = R.ld(3 + d, p); E[d] = R.ld(12 + d, p); Ena[d] = A1.ld(15 + d, p); }
    load_materials(D, p, mu, lam, ys);
    p2g_finish_grad<T>(D.P, G, v, C, E, mu, lam, ys, Ena, xa, va, Ca, Ea);
    PT_MARK(3);
    for (int d = 0; d < 3; ++d) { A0.st(d, p, xa[d]); A0.st(3 + d, p, va[d]); }
    for (int d = 0; d < 9; ++d) { A0.st(6 + d, p, Ca[d]); A0.st(15 + d, p, Ea[d]); }
    PT_MARK(4);
    PT_END(D, 20);
}

// compute_grid_m_kernel (mpm_simulator.py:382-392): mass-only scatter for the loss, same LDS-tile +
// wave pre-reduction scheme as k_p2g.  gm is a dense blocked T grid (zeroed by the caller).
template <class T, bool DET = false>
__global__ __launch_bounds__(kBlock) void k_grid_mass(Dev<T> D, int f, T* gm) {
    __shared__ int sred[kSred];
    __shared__ double tile[TileCap<T>::nodes * 4];
    const Soa<double> X = frame_xs(D, f);
    int p, base[3];
    double x[3];
    int base_true[3];
    const bool valid = load_sorted_particle(D, X, p, x, base, true);
    T fx[3], w[3][3];
    stencil<T, double>(x, D.P.inv_dx, base_true, fx, w, nullptr);
    Tile tl = block_tile(base, valid, sred, DET ? TileCap<T>::nodes * 2 : TileCap<T>::nodes * 4);     // DET: hi limb at [i], lo at [tn + i]
    const int ex = tl.e[0], exy = tl.e[0] * tl.e[1], tn = exy * tl.e[2];
    if (tl.ok) {
        for (int i = threadIdx.x; i < (DET ? 2 * tn : tn); i += kBlock) tile[i] = 0.0;
        __syncthreads();
    }
    const Seg<T> sg = wave_segments<T>(valid ? (base[2] * D.P.n + base[1]) * D.P.n + base[0] : -1);
    const bool emitter = sg.head && valid;
    const int ox = base[0] - tl.o[0], oy = base[1] - tl.o[1], oz = base[2] - tl.o[2];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            for (int l = 0; l < 3; ++l) {
                T m = seg_sum(w[i][0] * w[j][1] * w[l][2] * D.P.p_mass, sg);
                if (emitter) {
                    if (tl.ok && DET) {
                        long long* q = reinterpret_cast<long long*>(tile) + (oz + l) * exy + (oy + j) * ex + (ox + i);
                        det_add(q, q + tn, (double)m);
                    } else if (tl.ok) atomicAdd(&tile[(oz + l) * exy + (oy + j) * ex + (ox + i)], (double)m);
                    else if constexpr (DET) det_add_node(D, 0, node_index(D, base[0] + i, base[1] + j, base[2] + l), (double)m);
                    else atomicAdd(&gm[node_index(D, base[0] + i, base[1] + j, base[2] + l)], m);
                }
            }
    if (tl.ok) {
        __syncthreads();
        for (int i = threadIdx.x; i < tn; i += kBlock) {
            if constexpr (DET) {
                const long long* q = reinterpret_cast<const long long*>(tile) + i;
                if (q[0] | q[tn]) {
                    int lz, ly, lx;
                    tile_coords(i, ex, exy, lz, ly, lx);
                    det_flush_node(D, 0, node_index(D, tl.o[0] + lx, tl.o[1] + ly, tl.o[2] + lz), q[0], q[tn]);
                }
                continue;
            }
            double a = tile[i];
            if (a != 0.0) {
                int lz, ly, lx;
                tile_coords(i, ex, exy, lz, ly, lx);
                atomicAdd(&gm[node_index(D, tl.o[0] + lx, tl.o[1] + ly, tl.o[2] + lz)], (T)a);
            }
        }
    }
}

// zero grid_in / flags of the active blocks (a stored frame that is scattered into again without a backward
// pass in between, and plmpm_grid_stats; the reverse pass clears inside k_grid_op_grad)
template <class T>
__global__ __launch_bounds__(kBlock) void k_clear_active(Dev<T> D) {
    const int blk = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int nblk = D.nbx * D.nby * D.nbz;
    if (blk >= nblk) return;
    // one flag per wave: taken through the scalar unit (a wave-uniform branch -- and a wave-level operation between the 64 lanes'
    // reads of the flag and lane 0's write to it below, which the tests' CPU interpreter of this source needs: its lanes are not
    // in lock-step between two such operations)
    if (__builtin_amdgcn_readfirstlane(D.flags[flag_slot(D, blk)]) == 0) return;
    const int lane = threadIdx.x & 63;
    const int idx = (blk << 6) | lane;
    D.gin[0][idx] = T(0); D.gin[1][idx] = T(0); D.gin[2][idx] = T(0); D.gin[3][idx] = T(0);
    if (lane == 0) D.flags[flag_slot(D, blk)] = 0;
}

}  // namespace plb
