// Two particles per lane: the fp32 particle kernels with packed arithmetic (gfx950 / MI355X).
//
// A wave64 fp32 instruction occupies a 16-lane SIMD for 4 cycles; the particle kernels are bound by exactly that
// (profiles/r02_notes.md: vector pipes 75-89 % busy, ~6.8k vector instructions per 64 particles and substep).  Only the
// packed forms -- v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, two operations per lane in ~5 cycles -- reach the
// vector peak, and they want their operands in the two halves of a 64-bit register pair.  Pairing values of ONE
// particle costs moves (tried in round 2); here a lane carries TWO particles and every value of the per-particle
// arithmetic (mpm_math.h, instantiated for the pack types P2 / D2 / I2) is a pair by construction.
//
// Layout: a workgroup is 128 threads = 2 waves and owns the same 256 consecutive particles -- and the same entry of
// the per-frame tile table -- as a 256-thread workgroup of the scalar kernels, so scalar and packed kernels can be
// mixed freely over the frames of a rollout.  Wave w, slot s (the half of the pair) holds particles
// [w * 128 + s * 64, + 64): each slot of a wave is one "virtual wave" of the scalar kernels -- it is sorted by cell,
// cut into runs and pre-reduced with the same DPP steps, one slot after the other (the cross-lane steps have no packed
// form) -- while the arithmetic in between runs once for both.  256 VGPRs at 2 waves per SIMD: the same number of
// particles in flight per SIMD as the scalar kernels at 128 VGPRs and 4 waves.
#pragma once
#include "plmpm_kernels.h"

namespace plb {

constexpr int kBlockPk = 128;        // threads per workgroup (two particles each)
constexpr int kWavesPk = kBlockPk / 64;

// one slot of a wave: its 64 particles sorted by stencil base, exactly as sorted_begin / sorted_finish do for a scalar wave
struct SlotLoad { double x0[3]; };
template <class T> __device__ __forceinline__ int slot_first(int slot) {       // first particle of this wave's slot
    return (int)blockIdx.x * kBlock + ((int)threadIdx.x >> 6) * 128 + slot * 64;
}
template <class T> __device__ __forceinline__ SlotLoad slot_begin(const Dev<T>& D, const double* X, int slot) {
    const int p0 = slot_first<T>(slot) + (threadIdx.x & 63);
    SlotLoad s;
    s.x0[0] = s.x0[1] = s.x0[2] = 0.5;
    if (p0 < D.N) for (int d = 0; d < 3; ++d) s.x0[d] = X[d * D.Npad + p0];
    return s;
}
// -> p: the particle this lane processes in this slot, x its position, base its (clamped) stencil base
template <class T> __device__ __forceinline__ bool slot_finish(const Dev<T>& D, const SlotLoad& s, int slot, int& p, double* x, int* base) {
    const int first = slot_first<T>(slot), p0 = first + (threadIdx.x & 63);
    long long key = (1LL << 40);                                    // padding lanes last
    if (p0 < D.N) {
        int b[3];
        for (int d = 0; d < 3; ++d) b[d] = (int)(s.x0[d] * (double)D.P.inv_dx - 0.5);
        key = ((long long)b[2] * D.P.n + b[1]) * D.P.n + b[0];
    }
    const int src = D.P.n <= 256 ? wave_sort_lanes32(p0 < D.N ? (unsigned)key : 0x3ffffffu) : wave_sort_lanes(key);
    p = first + src;
    for (int d = 0; d < 3; ++d) x[d] = __shfl(s.x0[d], src);
    for (int d = 0; d < 3; ++d) base[d] = (int)(x[d] * (double)D.P.inv_dx - 0.5);
    clamp_to_reach(D, base);
    return p < D.N;
}

// box of the workgroup's stencil bases over both slots (block_tile_publish / _collect for two waves of two slots)
__device__ __forceinline__ void pk_tile_publish(const I2* base, bool v0, bool v1, int* sred) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int d = 0; d < 3; ++d) {
        const int lo = wave_min(min(v0 ? base[d].x : 0x7fffffff, v1 ? base[d].y : 0x7fffffff));
        const int hi = wave_max(max(v0 ? base[d].x : -0x7fffffff, v1 ? base[d].y : -0x7fffffff));
        if (lane == 0) { sred[wave * 6 + d] = lo; sred[wave * 6 + 3 + d] = hi; }
    }
}
__device__ __forceinline__ Tile pk_tile_collect(const int* sred, int cap) {
    Tile t;
    int nodes = 1;
    for (int d = 0; d < 3; ++d) {
        int l = sred[d], h = sred[3 + d];
        for (int w = 1; w < kWavesPk; ++w) { l = min(l, sred[w * 6 + d]); h = max(h, sred[w * 6 + 3 + d]); }
        t.o[d] = l;
        t.e[d] = h - l + 3;
        nodes *= t.e[d];
    }
    t.ok = (nodes > 0 && nodes <= cap) ? 1 : 0;
    return t;
}
template <class T> __device__ __forceinline__ void pk_store_tile(const Dev<T>& D, int f, const Tile& t) {
    if (threadIdx.x < 6) {
        int* q = D.tiles + ((size_t)f * D.twg + blockIdx.x) * 8;
        q[threadIdx.x] = threadIdx.x < 3 ? t.o[threadIdx.x] : t.e[threadIdx.x - 3];
    }
}

// ------------------------------------------------------------------------------------------------
// g2p(f-1) + p2g(f), two particles per lane (see k_g2p_p2g for the scalar kernel this mirrors step by step).
// FG (fused-grid engines): grid_op of substep f-1 evaluated in the tile fill.
template <bool FG = false>
__global__ __launch_bounds__(kBlockPk, 2) void k_g2p_p2g_pk(Dev<float> D, int f, PrevGrid<float> G0) {
    typedef float T;
    __shared__ int sred[32];
    __shared__ Vec4<double> tile[TileCap<T>::nodes];
    const PrimT<T>* sp = D.ptab + (size_t)(f - 1) * kMaxPrim;
    Vec4<T>* tile_v = reinterpret_cast<Vec4<T>*>(tile);
    const Vec4<T>* vout_prev = G0.vout;
    const int Np = D.Npad;
    // ---------------- g2p(f-1): gather
    const double* X0 = frame_x(D, f - 1);
    PT_BEGIN();
    const Tile ta = load_tile(D, f - 1, (int)(TileCap<T>::nodes * sizeof(Vec4<double>) / sizeof(Vec4<T>)));
    const SlotLoad sl0 = slot_begin(D, X0, 0), sl1 = slot_begin(D, X0, 1);
    {
        const int ex = ta.e[0], exy = ta.e[0] * ta.e[1], tn = exy * ta.e[2];
        if constexpr (FG) {
            for (int i = threadIdx.x; i < tn; i += kBlockPk) {
                int lz, ly, lx, idx;
                tile_coords(i, ex, exy, lz, ly, lx);
                const Vec4<T> a = fg_node_vout<NEAR_STORE>(D, G0.gin, sp, ta.o[0] + lx, ta.o[1] + ly, ta.o[2] + lz, const_cast<Vec4<T>*>(vout_prev), D.nprim, &idx);
                if (ta.ok) tile_v[i] = a; else const_cast<Vec4<T>*>(vout_prev)[idx] = a;
            }
        } else if (ta.ok)
            for (int i = threadIdx.x; i < tn; i += kBlockPk) {
                int lz, ly, lx;
                tile_coords(i, ex, exy, lz, ly, lx);
                tile_v[i] = vout_prev[node_index(D, ta.o[0] + lx, ta.o[1] + ly, ta.o[2] + lz)];
            }
    }
    PT_MARK(0);
    int p0, p1, b0[3], b1[3];
    double xa[3], xb[3];
    const bool v0 = slot_finish(D, sl0, 0, p0, xa, b0), v1 = slot_finish(D, sl1, 1, p1, xb, b1);
    // state that does not depend on the gather: in flight during it
    P2 E[9], mu, lam, ys;
    {
        const T* R = frame_r(D, f);
        const int q0 = v0 ? p0 : 0, q1 = v1 ? p1 : 0;                 // padding slots read row 0 (and store nothing)
        for (int d = 0; d < 9; ++d) E[d] = P2(R[(12 + d) * Np + q0], R[(12 + d) * Np + q1]);
        mu = P2(D.mu[q0], D.mu[q1]); lam = P2(D.lam[q0], D.lam[q1]); ys = P2(D.ys[q0], D.ys[q1]);
    }
    if (FG && !ta.ok) __syncthreads();                               // v_out went through HBM
    else wg_barrier();                                               // tile_v complete
    PT_MARK(1);
    D2 x0p[3], x[3];
    P2 v[3], C[9];
    for (int d = 0; d < 3; ++d) x0p[d] = D2(xa[d], xb[d]);
    if (ta.ok) {
        const int ex = ta.e[0], exy = ta.e[0] * ta.e[1];
        // padding slots gather from node 0 of the tile
        const Vec4<T>* t0 = tile_v + (v0 ? (b0[2] - ta.o[2]) * exy + (b0[1] - ta.o[1]) * ex + (b0[0] - ta.o[0]) : 0);
        const Vec4<T>* t1 = tile_v + (v1 ? (b1[2] - ta.o[2]) * exy + (b1[1] - ta.o[1]) * ex + (b1[0] - ta.o[0]) : 0);
        g2p_particle<P2, D2>(D.P, x0p, x, v, C, [&](int i, int j, int l, P2* gv) {
            const int o = l * exy + j * ex + i;
            const Vec4<T> a = t0[o], b = t1[o];
            gv[0] = P2(a.x, b.x); gv[1] = P2(a.y, b.y); gv[2] = P2(a.z, b.z);
        });
    } else {
        g2p_particle<P2, D2>(D.P, x0p, x, v, C, [&](int i, int j, int l, P2* gv) {
            const Vec4<T> a = vout_prev[node_index(D, b0[0] + i, b0[1] + j, b0[2] + l)];
            const Vec4<T> b = vout_prev[node_index(D, b1[0] + i, b1[1] + j, b1[2] + l)];
            gv[0] = P2(a.x, b.x); gv[1] = P2(a.y, b.y); gv[2] = P2(a.z, b.z);
        });
    }
    {
        double* X1 = frame_x_w(D, f);
        T* R1 = frame_r(D, f);
        if (v0) {
            for (int d = 0; d < 3; ++d) { X1[d * Np + p0] = x[d].x; R1[d * Np + p0] = v[d].lo(); }
            for (int d = 0; d < 9; ++d) R1[(3 + d) * Np + p0] = C[d].lo();
        }
        if (v1) {
            for (int d = 0; d < 3; ++d) { X1[d * Np + p1] = x[d].y; R1[d * Np + p1] = v[d].hi(); }
            for (int d = 0; d < 9; ++d) R1[(3 + d) * Np + p1] = C[d].hi();
        }
    }
    PT_MARK(2);
    // ---------------- p2g(f): scatter
    I2 base[3];
    {
        int c0[3], c1[3];
        for (int d = 0; d < 3; ++d) { c0[d] = (int)(x[d].x * (double)D.P.inv_dx - 0.5); c1[d] = (int)(x[d].y * (double)D.P.inv_dx - 0.5); }
        if ((clamp_to_reach(D, c0) && v0) | (clamp_to_reach(D, c1) && v1)) atomicOr(D.err, 1);
        for (int d = 0; d < 3; ++d) base[d] = I2(c0[d], c1[d]);
    }
    pk_tile_publish(base, v0, v1, sred);
    wg_barrier();                                                        // everyone is done reading tile_v, and has published its box
    const Tile tl = pk_tile_collect(sred, TileCap<T>::nodes);
    pk_store_tile(D, f, tl);
    const int tn = tl.e[0] * tl.e[1] * tl.e[2];
    if (tl.ok) {
        for (int i = threadIdx.x; i < tn; i += kBlockPk) tile[i] = Vec4<double>{0.0, 0.0, 0.0, 0.0};
        wg_barrier();
    }
    PT_MARK(3);
    {
        P2 En[9];
        const Seg<T> sg0 = wave_segments<T>(v0 ? (base[2].x * D.P.n + base[1].x) * D.P.n + base[0].x : -1);
        const Seg<T> sg1 = wave_segments<T>(v1 ? (base[2].y * D.P.n + base[1].y) * D.P.n + base[0].y : -1);
        const bool em0 = sg0.head && v0, em1 = sg1.head && v1;
        I2 b2[3];
        if (tl.ok) {
            const int ex = tl.e[0], exy = tl.e[0] * tl.e[1];
            Vec4<double>* t0 = tile + ((base[2].x - tl.o[2]) * exy + (base[1].x - tl.o[1]) * ex + (base[0].x - tl.o[0]));
            Vec4<double>* t1 = tile + ((base[2].y - tl.o[2]) * exy + (base[1].y - tl.o[1]) * ex + (base[0].y - tl.o[0]));
            p2g_particle<P2, D2>(D.P, x, v, C, E, mu, lam, ys, En, b2, [&](int i, int j, int l, P2 mass, const P2* mom) {
                T a0 = mass.lo(), a1 = mom[0].lo(), a2 = mom[1].lo(), a3 = mom[2].lo();
                T c0 = mass.hi(), c1 = mom[0].hi(), c2 = mom[1].hi(), c3 = mom[2].hi();
                PLB_ABLATE_STOP(4, a0 + a1 + a2 + a3 + c0 + c1 + c2 + c3, tile);
                seg_sum4(a0, a1, a2, a3, sg0);
                seg_sum4(c0, c1, c2, c3, sg1);
                PLB_ABLATE_STOP(1, a0 + a1 + a2 + a3 + c0 + c1 + c2 + c3, tile);
                const int o = l * exy + j * ex + i;
                if (em0) {
                    double* q = reinterpret_cast<double*>(t0 + o);
                    atomicAdd(q, (double)a0); atomicAdd(q + 1, (double)a1); atomicAdd(q + 2, (double)a2); atomicAdd(q + 3, (double)a3);
                }
                if (em1) {
                    double* q = reinterpret_cast<double*>(t1 + o);
                    atomicAdd(q, (double)c0); atomicAdd(q + 1, (double)c1); atomicAdd(q + 2, (double)c2); atomicAdd(q + 3, (double)c3);
                }
            });
        } else {
            p2g_particle<P2, D2>(D.P, x, v, C, E, mu, lam, ys, En, b2, [&](int i, int j, int l, P2 mass, const P2* mom) {
                T a0 = mass.lo(), a1 = mom[0].lo(), a2 = mom[1].lo(), a3 = mom[2].lo();
                T c0 = mass.hi(), c1 = mom[0].hi(), c2 = mom[1].hi(), c3 = mom[2].hi();
                seg_sum4(a0, a1, a2, a3, sg0);
                seg_sum4(c0, c1, c2, c3, sg1);
                if (em0) {
                    const int idx = node_index(D, base[0].x + i, base[1].x + j, base[2].x + l);
                    atomicAdd(&D.gin[0][idx], a0); atomicAdd(&D.gin[1][idx], a1); atomicAdd(&D.gin[2][idx], a2); atomicAdd(&D.gin[3][idx], a3);
                    D.flags[flag_slot(D, idx >> 6)] = 1;
                }
                if (em1) {
                    const int idx = node_index(D, base[0].y + i, base[1].y + j, base[2].y + l);
                    atomicAdd(&D.gin[0][idx], c0); atomicAdd(&D.gin[1][idx], c1); atomicAdd(&D.gin[2][idx], c2); atomicAdd(&D.gin[3][idx], c3);
                    D.flags[flag_slot(D, idx >> 6)] = 1;
                }
            });
        }
        T* R2 = frame_r(D, f + 1);
        if (v0) for (int d = 0; d < 9; ++d) R2[(12 + d) * Np + p0] = En[d].lo();
        if (v1) for (int d = 0; d < 9; ++d) R2[(12 + d) * Np + p1] = En[d].hi();
    }
    PT_MARK(4);
    if (tl.ok) {
        wg_barrier();
        const int ex = tl.e[0], exy = tl.e[0] * tl.e[1];
        for (int i = threadIdx.x; i < tn; i += kBlockPk) {
            const Vec4<double> a = tile[i];
            if (a.x != 0.0 || a.y != 0.0 || a.z != 0.0 || a.w != 0.0) {
                int lz, ly, lx;
                tile_coords(i, ex, exy, lz, ly, lx);
                const int idx = node_index(D, tl.o[0] + lx, tl.o[1] + ly, tl.o[2] + lz);
                atomicAdd(&D.gin[0][idx], (T)a.x); atomicAdd(&D.gin[1][idx], (T)a.y);
                atomicAdd(&D.gin[2][idx], (T)a.z); atomicAdd(&D.gin[3][idx], (T)a.w);
                D.flags[flag_slot(D, idx >> 6)] = 1;
            }
        }
    }
    PT_MARK(5);
    PT_END(D, 0);
}

}  // namespace plb
