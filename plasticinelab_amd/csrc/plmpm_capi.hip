// C-ABI implementation (include/plmpm.h) on top of the kernels in plmpm_kernels.h.
// Host code here only carves workspaces, moves host<->device state and sequences launches; every
// arithmetic step of the hot path runs in a HIP kernel.  There is no CPU fallback.
#include "plmpm_internal.h"

static thread_local std::string g_err;
int plmpm_fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}
// ---------------------------------------------------------------------------------------------
// small kernels: state I/O, primitive chains, loss
// staging layout (double, original particle order): x[N*3] v[N*3] F[N*9] C[N*9]
template <class T>
__global__ void k_unpack_frame(Dev<T> D, int f, const double* st, const int* perm, int has_x, int has_v, int has_F, int has_C) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.Npad) return;
    double* X = frame_x_w(D, f);
    T* R = frame_r(D, f);
    const int N = D.N, Np = D.Npad;
    if (i >= N) {       // padding lanes: harmless values
        for (int d = 0; d < 3; ++d) { X[d * Np + i] = 0.5; R[d * Np + i] = T(0); }
        for (int d = 0; d < 18; ++d) R[(3 + d) * Np + i] = T(0);
        return;
    }
    int o = perm[i];
    const double *sx = st, *sv = st + (size_t)3 * N, *sF = st + (size_t)6 * N, *sC = st + (size_t)15 * N;
    if (has_x) for (int d = 0; d < 3; ++d) X[d * Np + i] = sx[(size_t)3 * o + d];
    if (has_v) for (int d = 0; d < 3; ++d) R[d * Np + i] = (T)sv[(size_t)3 * o + d];
    if (has_C) for (int d = 0; d < 9; ++d) R[(3 + d) * Np + i] = (T)sC[(size_t)9 * o + d];
    if (has_F) for (int d = 0; d < 9; ++d) R[(12 + d) * Np + i] = (T)(sF[(size_t)9 * o + d] - ((d % 4 == 0) ? 1.0 : 0.0));
}
template <class T>
__global__ void k_pack_frame(Dev<T> D, int f, double* st, const int* perm) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.N) return;
    const double* X = frame_x(D, f);
    const T* R = frame_r(D, f);
    const int N = D.N, Np = D.Npad;
    int o = perm[i];
    double *sx = st, *sv = st + (size_t)3 * N, *sF = st + (size_t)6 * N, *sC = st + (size_t)15 * N;
    for (int d = 0; d < 3; ++d) { sx[(size_t)3 * o + d] = X[d * Np + i]; sv[(size_t)3 * o + d] = (double)R[d * Np + i]; }
    for (int d = 0; d < 9; ++d) {
        sC[(size_t)9 * o + d] = (double)R[(3 + d) * Np + i];
        sF[(size_t)9 * o + d] = (double)R[(12 + d) * Np + i] + ((d % 4 == 0) ? 1.0 : 0.0);
    }
}
// adjoint frame <-> staging (same staging layout: xa, va, Fa, Ca)
template <class T>
__global__ void k_adj_io(Dev<T> D, int which, double* st, const int* perm, int add_from_staging, int has_x, int has_v, int has_F, int has_C) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.N) return;
    T* A = D.adj[which];
    const int N = D.N, Np = D.Npad;
    int o = perm[i];
    double *sx = st, *sv = st + (size_t)3 * N, *sF = st + (size_t)6 * N, *sC = st + (size_t)15 * N;
    if (add_from_staging) {
        if (has_x) for (int d = 0; d < 3; ++d) A[d * Np + i] += (T)sx[(size_t)3 * o + d];
        if (has_v) for (int d = 0; d < 3; ++d) A[(3 + d) * Np + i] += (T)sv[(size_t)3 * o + d];
        if (has_C) for (int d = 0; d < 9; ++d) A[(6 + d) * Np + i] += (T)sC[(size_t)9 * o + d];
        if (has_F) for (int d = 0; d < 9; ++d) A[(15 + d) * Np + i] += (T)sF[(size_t)9 * o + d];
    } else {
        for (int d = 0; d < 3; ++d) { sx[(size_t)3 * o + d] = (double)A[d * Np + i]; sv[(size_t)3 * o + d] = (double)A[(3 + d) * Np + i]; }
        for (int d = 0; d < 9; ++d) { sC[(size_t)9 * o + d] = (double)A[(6 + d) * Np + i]; sF[(size_t)9 * o + d] = (double)A[(15 + d) * Np + i]; }
    }
}
template <class T> __global__ void k_set_mats(Dev<T> D, const double* st, const int* perm) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.Npad) return;
    int o = i < D.N ? perm[i] : perm[0];
    D.mu[i] = (T)st[o]; D.lam[i] = (T)st[(size_t)D.N + o]; D.ys[i] = (T)st[(size_t)2 * D.N + o];
}
__global__ void k_copy_frame(char* state, size_t frame_bytes, int src, int dst) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n16 = frame_bytes / 16;
    const uint4* s = reinterpret_cast<const uint4*>(state + (size_t)src * frame_bytes);
    uint4* d = reinterpret_cast<uint4*>(state + (size_t)dst * frame_bytes);
    if (i < n16) d[i] = s[i];
}

// global += local; local = 0   (pose adjoints after the cross-rank sum)
// dst += src (the neighbour's copy of exchanged block planes, fields that no grid kernel adds on first touch)
// (src may be a receive area of the device-side exchange, written by a neighbour GPU: system-scope load, as every other reader
// of those areas -- plmpm_peer.hip -- so that the fine-grained fallback allocation cannot serve a stale cache line either)
template <class T> __global__ void k_add_region(T* dst, const T* src, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <class T> __global__ void k_grid_stats(Dev<T> D, unsigned long long* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t G = (size_t)D.nbx * D.nby * D.nbz * 64;
    if (i < G && D.gin[0][i] > T(0)) atomicAdd(&out[0], 1ULL);
    if (i < G / 64 && D.flags[flag_slot(D, (int)i)]) atomicAdd(&out[1], 1ULL);
}

static const HaloIn kNoHalo = {0, {0, 0}, {0, 0}, {nullptr, nullptr}, 0};
#ifndef PLB_POSE_WG
#define PLB_POSE_WG 16
#endif
// workgroups at the head of every k_p2g_grad launch that finish grid_op.grad's pose adjoints (64 waves: the blocks in
// contact with a manipulator number a few dozen)
constexpr int kPoseWG = PLB_POSE_WG;
static inline int nblocks_grid(const plmpm_sim* s) { return (s->nblk + (kBlock / 64) - 1) / (kBlock / 64); }
// persistent grid kernels: a fixed number of workgroups, each striding over its share of the block flags
static inline int nwg_grid(const plmpm_sim* s) { return s->gwg; }

template <class T> static int substep_fwd(plmpm_sim* s, int f) {
    Dev<T> D = make_dev<T>(s, f);
    if (s->store) {
        if (s->dirty[f]) LAUNCH(s, K_CLEAR, (k_clear_active<T>), dim3(nblocks_grid(s)), D);   // frame reused without a backward pass
        LAUNCH_P2G(s, K_P2G, true, D, f);
        LAUNCH(s, K_GRID_OP, (k_grid_op<T, false>), dim3(nwg_grid(s)), D, f, s->halo_in[PLMPM_HALO_GRID_IN]);                    // keep grid_in for substep_grad
        s->dirty[f] = 1;
    } else {
        LAUNCH_P2G(s, K_P2G, true, D, f);
        LAUNCH(s, K_GRID_OP, (k_grid_op<T, true>), dim3(nwg_grid(s)), D, f, kNoHalo);
    }
    LAUNCH(s, K_G2P, (k_g2p<T>), dim3(nblocks_particles(s, f)), D, f);
    return 0;
}
template <class T> static int substep_bwd(plmpm_sim* s, int f) {
    Dev<T> D = make_dev<T>(s, f);
    const int src = (f + 1) & 1, dst = f & 1;
    // frame f+1 re-sorted by the env step that starts there: its v in THIS frame's order was kept aside
    const T* vnext = s->frame_epoch[f + 1] != s->frame_epoch[f] ? (const T*)(s->vend + (size_t)s->frame_epoch[f + 1] * 3 * s->Npad * s->tsz) : nullptr;
    if (!(s->store && s->dirty[f])) {        // this frame's grid is not resident: recompute it (mpm_simulator.py:265-268)
        LAUNCH_P2G(s, K_P2G_RE, false, D, f);
        LAUNCH(s, K_GRID_OP_RE, (k_grid_op<T, false>), dim3(nwg_grid(s)), D, f, kNoHalo);
    }
    LAUNCH_G2P_GRAD(s, D, f, src, dst, vnext);
    LAUNCH(s, K_GRID_OP_GRAD, (k_grid_op_grad<T>), dim3(nwg_grid(s)), D, f, s->halo_in[PLMPM_HALO_GRID_OUT_ADJ]);
    LAUNCH_P2G_GRAD(s, D, f, src, dst);
    if (s->store) s->dirty[f] = 0;           // k_grid_op_grad left grid_in / flags of this frame clean
    s->adj_frame[dst] = f;
    return 0;
}

// Whole env step forward in store mode: p2g(f0) | grid_op(f0) | [g2p(f-1)+p2g(f) fused | grid_op(f)] ... | g2p(last)
template <class T> static int step_fwd_fused(plmpm_sim* s, int first, int n) {
    for (int f = first; f < first + n; ++f) {
        Dev<T> D = make_dev<T>(s, f);
        if (s->dirty[f]) LAUNCHG_CLEAR(s, D);
        if (f == first) {
            LAUNCH_P2G(s, K_P2G, true, D, f);
        } else {
            const Vec4<T>* vprev = (const Vec4<T>*)(s->vstore + (size_t)(f - 1) * s->gstride);
            LAUNCH_G2P_P2G(s, D, f, vprev);
        }
        LAUNCH(s, K_GRID_OP, (k_grid_op<T, false>), dim3(nwg_grid(s)), D, f, s->halo_in[PLMPM_HALO_GRID_IN]);
        s->dirty[f] = 1;
    }
    Dev<T> D = make_dev<T>(s, first + n - 1);
    LAUNCH(s, K_G2P, (k_g2p<T>), dim3(nblocks_particles(s, first + n - 1)), D, first + n - 1);
    return 0;
}

// phase-split variants used by the multi-GPU driver (store_grid mode only)
template <class T> static int phase_p2g(plmpm_sim* s, int f) {
    Dev<T> D = make_dev<T>(s, f);
    if (s->dirty[f]) LAUNCH(s, K_CLEAR, (k_clear_active<T>), dim3(nblocks_grid(s)), D);
    LAUNCH_P2G(s, K_P2G, true, D, f);
    s->dirty[f] = 1;
    return 0;
}
// part: 0 every active block | 1 only the blocks outside the exchanged planes (nothing else: the halos may still be in
// flight) | 2 the blocks of the exchanged planes, then g2p
// X (device-side exchange of grid_m / grid_v_in folded into the launch, part = 0 only): k_grid_op_x
template <class T> static int phase_grid_g2p(plmpm_sim* s, int f, bool defer_g2p = false, int part = 0, const PeerXchg* X = nullptr) {
    Dev<T> D = make_dev<T>(s, f);
    s->frame_epoch[f + 1] = s->frame_epoch[f];                 // g2p (now or fused into the next p2g) writes frame f + 1 in this order
    HaloIn H = s->halo_in[PLMPM_HALO_GRID_IN];
    H.part = part;
    if (X && X->n > 0) LAUNCH(s, K_GRID_OP_X, (k_grid_op_x<T>), dim3(nwg_grid(s)), D, f, H, *X);
    else LAUNCH(s, K_GRID_OP, (k_grid_op<T, false>), dim3(nwg_grid(s)), D, f, H);
    if (part == 1) return 0;
    if (!defer_g2p) LAUNCH(s, K_G2P, (k_g2p<T>), dim3(nblocks_particles(s, f)), D, f);
    return 0;
}
// g2p(f-1), deferred by the previous phase_grid_g2p, fused with p2g(f) exactly as in step_fwd_fused
template <class T> static int phase_g2p_p2g(plmpm_sim* s, int f) {
    Dev<T> D = make_dev<T>(s, f);
    if (s->dirty[f]) LAUNCHG_CLEAR(s, D);
    const Vec4<T>* vprev = (const Vec4<T>*)(s->vstore + (size_t)(f - 1) * s->gstride);
    LAUNCH_G2P_P2G(s, D, f, vprev);
    s->dirty[f] = 1;
    return 0;
}
template <class T> static int phase_grad_scatter(plmpm_sim* s, int f) {
    Dev<T> D = make_dev<T>(s, f);
    const T* vnext = s->frame_epoch[f + 1] != s->frame_epoch[f] ? (const T*)(s->vend + (size_t)s->frame_epoch[f + 1] * 3 * s->Npad * s->tsz) : nullptr;
    LAUNCH_G2P_GRAD(s, D, f, (f + 1) & 1, f & 1, vnext);
    return 0;
}
template <class T> static int phase_grad_gather(plmpm_sim* s, int f, int part = 0, const PeerXchg* X = nullptr) {
    Dev<T> D = make_dev<T>(s, f);
    HaloIn H = s->halo_in[PLMPM_HALO_GRID_OUT_ADJ];
    H.part = part;
    if (X && X->n > 0) LAUNCH(s, K_GRID_OP_GRAD_X, (k_grid_op_grad_x<T>), dim3(nwg_grid(s)), D, f, H, *X);
    else LAUNCH(s, K_GRID_OP_GRAD, (k_grid_op_grad<T>), dim3(nwg_grid(s)), D, f, H);
    if (part == 1) return 0;
    LAUNCH_P2G_GRAD(s, D, f, (f + 1) & 1, f & 1);
    s->dirty[f] = 0;
    s->adj_frame[f & 1] = f;
    s->adj_epoch[f & 1] = s->frame_epoch[f];
    return 0;
}


template <class T> static int set_materials_t(plmpm_sim* s, int epoch) {
    const bool filled = s->mats_filled;
    s->mats_epoch = epoch;
    if (s->dist && epoch != 0) return 0;          // slab engines: epochs > 0 got their materials with the migrating rows
    if (s->mats_uniform && filled && !s->dist) return 0;      // every particle the same: a permutation changes nothing
    s->mats_filled = true;
    Dev<T> D = make_dev<T>(s);
    hipLaunchKernelGGL((k_set_mats<T>), dim3(s->Npad / 256), dim3(256), 0, s->stream, D, s->mats_master, perm_of(s, epoch));
    return 0;
}

template <class T> static int unpack_t(plmpm_sim* s, int f, int hx, int hv, int hF, int hC) {
    Dev<T> D = make_dev<T>(s, f);
    hipLaunchKernelGGL((k_unpack_frame<T>), dim3(s->Npad / 256), dim3(256), 0, s->stream, D, f, s->staging, perm_of(s, s->frame_epoch[f]), hx, hv, hF, hC);
    return 0;
}

template <class T> static int pack_t(plmpm_sim* s, int f) {
    Dev<T> D = make_dev<T>(s, f);
    hipLaunchKernelGGL((k_pack_frame<T>), dim3(s->Npad / 256), dim3(256), 0, s->stream, D, f, s->staging, perm_of(s, s->frame_epoch[f]));
    return 0;
}

template <class T> static int adj_io_t(plmpm_sim* s, int which, int add, int hx, int hv, int hF, int hC, int epoch = -1) {
    Dev<T> D = make_dev<T>(s);
    if (epoch < 0) epoch = s->adj_epoch[which];
    D.N = s->epochN[epoch];
    hipLaunchKernelGGL((k_adj_io<T>), dim3(s->Npad / 256), dim3(256), 0, s->stream, D, which, s->staging, perm_of(s, epoch), add, hx, hv, hF, hC);
    return 0;
}

template <class T> __global__ void k_hilbert_keys(Dev<T> D, int f, int bits, unsigned* keys, int* idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.Npad) return;
    idx[i] = i;
    if (i >= D.N) { keys[i] = 1u << (3 * bits); return; }          // one past the largest cell key: padding sorts last
    const double* X = frame_x(D, f);
    int b[3];
    for (int d = 0; d < 3; ++d) {
        b[d] = (int)(X[d * D.Npad + i] * (double)D.P.inv_dx - 0.5);
        b[d] = min(max(b[d], 0), D.P.n - 1);
    }
    keys[i] = hilbert_key_dev((unsigned)b[0], (unsigned)b[1], (unsigned)b[2], bits);
}
// frame f gathered through `order` (new slot i <- old slot order[i]) into `out`; v of the old frame kept in `vend`
template <class T> __global__ void k_permute_frame(Dev<T> D, int f, const int* order, char* out, T* vend, const int* perm_old, int* perm_new) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.Npad) return;
    const int Np = D.Npad, j = order[i];
    const double* X = frame_x(D, f);
    const T* R = frame_r(D, f);
    double* Xo = reinterpret_cast<double*>(out);
    T* Ro = reinterpret_cast<T*>(out + (size_t)3 * 8 * Np);
    for (int d = 0; d < 3; ++d) Xo[d * Np + i] = X[d * Np + j];
    for (int d = 0; d < 21; ++d) Ro[d * Np + i] = R[d * Np + j];
    for (int d = 0; d < 3; ++d) vend[d * Np + i] = R[d * Np + i];          // old order, same slot
    if (i < D.N) perm_new[i] = perm_old[j];
}
// Counting sort by cell along the curve: a particle's key IS its bin, so the order is histogram -> exclusive scan ->
// scatter through a per-bin cursor -- 4 launches instead of the ~20 of the library radix sort (0.13 ms for 500k pairs,
// launch-bound).  The order inside a cell is whatever the cursor hands out (the in-wave sort does not care); the
// deterministic engine and grids with more than 2^25 cells take the radix sort.
template <class T> __global__ void k_cell_hist(Dev<T> D, int f, int bits, unsigned* keys, unsigned* hist) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.N) return;
    const double* X = frame_x(D, f);
    int b[3];
    for (int d = 0; d < 3; ++d) b[d] = min(max((int)(X[d * D.Npad + i] * (double)D.P.inv_dx - 0.5), 0), D.P.n - 1);
    const unsigned k = hilbert_key_dev((unsigned)b[0], (unsigned)b[1], (unsigned)b[2], bits);
    keys[i] = k;
    atomicAdd(&hist[k], 1u);
}
__global__ void k_cell_scatter(int n, int npad, const unsigned* keys, unsigned* cursor, int* order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    if (i >= n) { order[i] = i; return; }                   // padding slots stay where they are
    order[atomicAdd(&cursor[keys[i]], 1u)] = i;
}
template <class T> static int resort_frame_t(plmpm_sim* s, int f, int epoch) {
    Dev<T> D = make_dev<T>(s, f);
    s->epochN[epoch] = D.N;
    const int nb = s->Npad / 256;
    int bits = 1;
    while ((1 << bits) < s->n) ++bits;
    if (s->cell_bins && !s->det) {
        HIPCHK(hipMemsetAsync(s->cell_hist, 0, s->cell_bins * 4, s->stream));
        hipLaunchKernelGGL((k_cell_hist<T>), dim3(nb), dim3(256), 0, s->stream, D, f, bits, s->skey[0], s->cell_hist);
        if (plmpm_exclusive_scan(s->sort_tmp, s->sort_tmp_bytes, s->cell_hist, s->cell_hist, s->cell_bins, s->stream) != 0) return fail("resort: device scan failed");
        hipLaunchKernelGGL(k_cell_scatter, dim3(nb), dim3(256), 0, s->stream, D.N, s->Npad, s->skey[0], s->cell_hist, s->sidx[1]);
        T* vend = (T*)(s->vend + (size_t)epoch * 3 * s->Npad * s->tsz);
        hipLaunchKernelGGL((k_permute_frame<T>), dim3(nb), dim3(256), 0, s->stream, D, f, s->sidx[1], s->frame_tmp, vend,
                           perm_of(s, s->frame_epoch[f]), perm_of(s, epoch));
        HIPCHK(hipMemcpyAsync(s->state + (size_t)f * s->frame_bytes, s->frame_tmp, s->frame_bytes, hipMemcpyDeviceToDevice, s->stream));
        s->frame_epoch[f] = epoch;
        return 0;
    }
    // (keys without the lowest 1 / 2 curve levels -- fewer radix passes -- were measured: the kernels lose more to the
    // coarser order than the sort saves, profiles/r02_notes.md)
    hipLaunchKernelGGL((k_hilbert_keys<T>), dim3(nb), dim3(256), 0, s->stream, D, f, bits, s->skey[0], s->sidx[0]);
    if (plmpm_sort_pairs(s->sort_tmp, s->sort_tmp_bytes, s->skey[0], s->skey[1], s->sidx[0], s->sidx[1], s->Npad, 3 * bits + 1, s->stream) != 0)
        return fail("resort: device sort failed");
    T* vend = (T*)(s->vend + (size_t)epoch * 3 * s->Npad * s->tsz);
    hipLaunchKernelGGL((k_permute_frame<T>), dim3(nb), dim3(256), 0, s->stream, D, f, s->sidx[1], s->frame_tmp, vend,
                       perm_of(s, s->frame_epoch[f]), perm_of(s, epoch));
    HIPCHK(hipMemcpyAsync(s->state + (size_t)f * s->frame_bytes, s->frame_tmp, s->frame_bytes, hipMemcpyDeviceToDevice, s->stream));
    s->frame_epoch[f] = epoch;
    return 0;
}
// adjoint frame `which` from the storage order of epoch `from` to that of epoch `to`: slot j of `to` holds the particle
// (caller index perm_to[j]) that sat in slot inv_from[perm_to[j]] of `from` -- one gather pass over the 24 rows.  (Through
// the float64 caller-order staging buffer, as state I/O goes, this was two passes and 0.18 ms per re-sort boundary.)
__global__ void k_inv_perm(const int* perm, int n, int* inv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) inv[perm[i]] = i;
}
template <class T> __global__ void k_adj_regather(const T* in, T* out, const int* inv_from, const int* perm_to, int n, int Np) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Np) return;
    if (j >= n) { for (int d = 0; d < 24; ++d) out[(size_t)d * Np + j] = T(0); return; }
    const int i = inv_from[perm_to[j]];
    for (int d = 0; d < 24; ++d) out[(size_t)d * Np + j] = in[(size_t)d * Np + i];
}
template <class T> static int convert_adjoint_t(plmpm_sim* s, int which, int from, int to) {
    if (from == to) return 0;
    if (s->dist) return fail("slab engine: the adjoint frame is in storage epoch %d but epoch %d is needed -- particles migrated in between; "
                             "run plmpm_migrate_adjoint_begin / _finish on the boundary frame first", from, to);
    const int nb = s->Npad / 256;
    int* inv = s->sidx[0];                               // sort scratch: idle during the reverse sweep
    hipLaunchKernelGGL(k_inv_perm, dim3(nb), dim3(256), 0, s->stream, perm_of(s, from), s->N, inv);
    hipLaunchKernelGGL((k_adj_regather<T>), dim3(nb), dim3(256), 0, s->stream, (const T*)s->adj[which], (T*)s->frame_tmp, inv, perm_of(s, to), s->N, s->Npad);
    HIPCHK(hipMemcpyAsync(s->adj[which], s->frame_tmp, (size_t)24 * s->Npad * s->tsz, hipMemcpyDeviceToDevice, s->stream));
    s->adj_epoch[which] = to;
    return 0;
}

int plmpm_convert_adjoint(plmpm_sim* s, int which, int from, int to) { return DISPATCH(s, convert_adjoint_t, s, which, from, to); }
template <class T> static int grid_stats_t(plmpm_sim* s, int f, unsigned long long* d_out) {
    Dev<T> D = make_dev<T>(s);
    D.N = s->epochN[s->frame_epoch[f]];
    // recompute the scatter of frame f without consuming it, count, then clear
    LAUNCH_P2G(s, K_P2G_RE, false, D, f);
    hipLaunchKernelGGL((k_grid_stats<T>), dim3((unsigned)((s->G + 255) / 256)), dim3(256), 0, s->stream, D, d_out);
    hipLaunchKernelGGL((k_clear_active<T>), dim3(nblocks_grid(s)), dim3(kBlock), 0, s->stream, D);
    return 0;
}

extern "C" {

const char* plmpm_last_error(void) { return g_err.c_str(); }
int plmpm_version(void) { return 1; }
int plmpm_build_flags(void) { return (PLB_FAST ? 2 : 0) | (PLB_XCD_MAP ? 4 : 0) | (PLB_BUFIO ? 8 : 0); }

int plmpm_create(const plmpm_config* cfg, const plmpm_primitive* prims, plmpm_handle* out) {
    REQUIRE(cfg && out, "null argument");
    REQUIRE(cfg->dtype == PLMPM_F32 || cfg->dtype == PLMPM_F64, "dtype must be PLMPM_F32 or PLMPM_F64");
    REQUIRE(cfg->n_grid >= 8 && cfg->n_grid % 4 == 0, "n_grid must be a multiple of 4 (got %d)", cfg->n_grid);
    REQUIRE(cfg->n_particles > 0, "n_particles must be positive");
    REQUIRE(cfg->n_primitives >= 0 && cfg->n_primitives <= PLMPM_MAX_PRIMITIVES, "at most %d primitives", PLMPM_MAX_PRIMITIVES);
    REQUIRE(cfg->max_frames >= 1, "max_frames must be >= 1");
    REQUIRE(cfg->n_primitives == 0 || prims, "prims is null");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device visible: this engine has no CPU path");
    plmpm_sim* s = new plmpm_sim();
    s->cfg = *cfg;
    if (s->cfg.slab_z1 <= s->cfg.slab_z0) { s->cfg.slab_z0 = 0; s->cfg.slab_z1 = cfg->n_grid; }
    s->P = cfg->n_primitives;
    s->act_ofs[0] = 0;
    for (int p = 0; p < s->P; ++p) {
        s->prims[p] = prims[p];
        if (prims[p].action_dim < 0 || prims[p].action_dim > PLMPM_MAX_ACTION_DIM) { delete s; return fail("bad action_dim"); }
        if (prims[p].kinematics < PLMPM_KIN_DEFAULT || prims[p].kinematics > PLMPM_KIN_CHOPSTICKS) { delete s; return fail("unknown kinematics %d", prims[p].kinematics); }
        if ((prims[p].shape == PLMPM_CHOPSTICKS) != (prims[p].kinematics == PLMPM_KIN_CHOPSTICKS)) {
            delete s;
            return fail("primitive %d: the Chopsticks shape and PLMPM_KIN_CHOPSTICKS go together", p);
        }
        if (prims[p].shape == PLMPM_CHOPSTICKS && prims[p].action_dim != 7) {
            delete s;
            return fail("primitive %d: Chopsticks take a 7-dim action (3 linear, 3 angular, 1 grasp; primitives.py:92)", p);
        }
        if (prims[p].shape < PLMPM_SPHERE || prims[p].shape > PLMPM_CHOPSTICKS) { delete s; return fail("primitive %d: unknown shape %d", p, prims[p].shape); }
        s->act_ofs[p + 1] = s->act_ofs[p] + prims[p].action_dim;
    }
    s->act_total = s->act_ofs[s->P];
    s->N = cfg->n_particles;
    const int cap = std::max(cfg->particle_capacity, cfg->n_particles);
    s->Npad = (int)align_up(cap, kRowPad);
#if PLB_BUFIO
    // particle arrays are addressed through 32-bit buffer offsets (plmpm_kernels.h: Soa): the largest array set, an adjoint frame
    // of 24 scalars per row, must stay below 4 GiB
    if ((size_t)s->Npad * 24 * (cfg->dtype == PLMPM_F64 ? 8 : 4) >= ((size_t)1 << 32)) {
        delete s;
        return fail("particle capacity %d is too large for this build's 32-bit buffer addressing (-DPLB_BUFIO=1)", cap);
    }
#endif
    s->n = cfg->n_grid; s->Gfull = (size_t)s->n * s->n * s->n;
    // grid window: the box of 4^3 blocks that is allocated and swept (all-zero grid_lo / grid_hi = the whole grid)
    for (int d = 0; d < 3; ++d) {
        int lo = cfg->grid_lo[d], hi = cfg->grid_hi[d];
        if (hi <= lo) { lo = 0; hi = s->n; }
        lo = std::max(0, lo) / 4 * 4;
        hi = std::min(s->n, (hi + 3) / 4 * 4);
        if (hi - lo < 4) { delete s; return fail("grid window axis %d is empty: [%d, %d)", d, cfg->grid_lo[d], cfg->grid_hi[d]); }
        s->go[d] = lo; s->nbw[d] = (hi - lo) / 4;
    }
    s->nblk = s->nbw[0] * s->nbw[1] * s->nbw[2]; s->G = (size_t)s->nblk * 64;
    // persistent grid kernels: a power-of-two number of workgroups (<= kGridWG, about one wave per 1-4 blocks), each
    // with its blocks' flags side by side
    s->gwg = 1; s->gwg_log2 = 0;
    const int gwg_max = cfg->grid_workgroups > 0 ? std::min(cfg->grid_workgroups, kGridWG) : kGridWG;
    while (s->gwg * 2 <= gwg_max && s->gwg * 2 * (kBlock / 64) <= s->nblk) { s->gwg *= 2; ++s->gwg_log2; }
    s->fs = (s->nblk + s->gwg - 1) / s->gwg;
    s->nflag = s->gwg * s->fs;
    s->F = cfg->max_frames;
    s->tsz = cfg->dtype == PLMPM_F64 ? 8 : 4;
    s->frame_bytes = (size_t)s->Npad * (24 + 21 * s->tsz);
    size_t P1 = std::max(s->P, 1);
    s->ws.state_bytes = (size_t)(s->F + 1) * s->frame_bytes;
    s->ws.adjoint_bytes = align_up(2 * 24 * s->Npad * s->tsz, 256) + 3 * align_up(s->Npad * s->tsz, 256) + align_up((size_t)s->Npad * 4, 256);
    s->ws.grid_bytes = 4 * align_up(s->G * 4 * s->tsz, 256) + align_up((size_t)s->nflag * 4, 256) + 3 * align_up(s->G * s->tsz, 256);
    s->dist = s->cfg.slab_z0 > 0 || s->cfg.slab_z1 < cfg->n_grid || cfg->slab_halo > 0;
    s->resort = cfg->resort_steps > 0 && !s->dist && cfg->substeps > 0;
    // storage epochs: single GPU one per re-sort (+2: the alternating pair of copy-mode episodes); slab engines one per
    // migration, at most one per env step
    s->n_epochs = s->resort ? s->F / (cfg->substeps * cfg->resort_steps) + 3 : 1;
    if (s->dist && cfg->substeps > 0) s->n_epochs = s->F / cfg->substeps + 4;
    s->epochN.assign(s->n_epochs + 1, s->N);
    s->mig.assign(s->n_epochs + 1, plmpm_sim::MigInfo());
    s->frame_epoch.assign(s->F + 2, 0);
    const bool sorts = s->resort || s->dist;
    s->sort_cap = s->dist ? s->Npad + s->Npad / 2 : s->Npad;          // slab engines sort stayers + arrivals
    s->mig_max_rows = s->Npad / 4;
    s->sort_tmp_bytes = sorts ? plmpm_sort_temp_bytes(s->sort_cap) : 0;
    s->cell_bins = 0;
    if (s->resort) {                                       // counting-sort re-sort: one bin per cell of the 2^bits cube
        int cb = 1;
        while ((1 << cb) < s->n) ++cb;
        if (3 * cb <= 25) {
            s->cell_bins = (size_t)1 << (3 * cb);
            s->sort_tmp_bytes = std::max(s->sort_tmp_bytes, plmpm_scan_temp_bytes(s->cell_bins));
            s->ws.adjoint_bytes += align_up(s->cell_bins * 4, 256);
        }
    }
    s->ws.adjoint_bytes += align_up((size_t)3 * s->Npad * 8, 256);                   // material master copy
    if (sorts)
        s->ws.adjoint_bytes += align_up((size_t)(s->n_epochs - 1) * s->Npad * 4, 256) + align_up((size_t)s->n_epochs * 3 * s->Npad * s->tsz, 256)
                               + 4 * align_up((size_t)s->sort_cap * 4, 256) + align_up(s->sort_tmp_bytes, 256) + align_up(s->frame_bytes, 256);
    if (s->dist)
        s->ws.adjoint_bytes += 3 * align_up((size_t)s->n_epochs * s->Npad * 4, 256)            // gid_store, mig_src, mig_leave
                               + align_up((size_t)s->n_epochs * 3 * s->Npad * s->tsz, 256)    // mats_store
                               + 2 * align_up((size_t)s->Npad * 4, 256) + 256                  // mig_dest, iota, mig_cnt
                               + 2 * align_up((size_t)s->mig_max_rows * 28 * 8, 256);          // packed rows of the leavers, per direction
    s->store = cfg->store_grid != 0;
    s->gstride = align_up(s->G * 4 * s->tsz, 256);
    if (s->store) s->ws.grid_bytes += 2 * (size_t)s->F * s->gstride + align_up((size_t)s->F * s->nflag * 4, 256);
    s->ws.grid_bytes += align_up((size_t)(s->F + 1) * (s->Npad / kBlock) * 8 * 4, 256) + align_up((size_t)(s->nblk + 1) * 4, 256);
    s->ws.grid_bytes += align_up((size_t)(s->F + 1) * kMaxPrim * sizeof(PrimT<double>), 256);
    s->dirty.assign(s->F + 1, 0);
    s->ws.misc_bytes = 2 * align_up((size_t)(s->F + 1) * P1 * 7 * 8, 256) + 2 * align_up((size_t)(s->F + 1) * P1 * 8 * 8, 256)  // poses(+adj), padded
                       + 4 * align_up((size_t)(s->F + 1) * P1 * 3 * 8, 256)                       // v,w (+adj)
                       + 4 * align_up((size_t)(s->F + 1) * P1 * 8, 256)                           // gap, gap_vel (+adj)
                       + 2 * align_up((size_t)(s->F + 1) * P1 * PLMPM_MAX_ACTION_DIM * 8, 256)    // action buffers (+adj)
                       + align_up(LS_COUNT * 8, 256) + align_up((size_t)s->Npad * 24 * 8, 256) + 512;
    if (s->dist) s->ws.misc_bytes += 3 * align_up((size_t)(s->F + 1) * P1 * 4 * 8, 256);
    s->det = cfg->deterministic != 0;
    if (s->det) {
        s->ws.grid_bytes += align_up(s->G * 8 * 8, 256);                                  // [8][G] integer limbs
        s->ws.misc_bytes += align_up((size_t)(LS_COUNT + kMaxPrim * 8) * 2 * 8, 256);
    }
    memset(s->halo_in, 0, sizeof s->halo_in);
    s->perm.resize(s->N);
    for (int i = 0; i < s->N; ++i) s->perm[i] = i;
    *out = s;
    return 0;
}

int plmpm_destroy(plmpm_handle s) {
    if (s) {
        for (auto e : s->ev_pool) (void)hipEventDestroy(e);
        for (void* p : s->peer_mapped) (void)hipIpcCloseMemHandle(p);
        for (void* p : s->peer_allocs) (void)hipFree(p);
        if (s->peer_done) (void)hipFree(s->peer_done);
        if (s->peer_status) (void)hipHostFree(s->peer_status);
    }
    delete s;
    return 0;
}

int plmpm_workspace_bytes(plmpm_handle s, plmpm_workspace* out) {
    REQUIRE(s && out, "null argument");
    *out = s->ws;
    return 0;
}

int plmpm_bind_workspace(plmpm_handle s, void* state, void* adjoint, void* grid, void* misc) {
    REQUIRE(s && state && adjoint && grid && misc, "null argument");
    REQUIRE(((uintptr_t)state | (uintptr_t)adjoint | (uintptr_t)grid | (uintptr_t)misc) % 256 == 0, "workspaces must be 256-byte aligned");
    s->state = (char*)state; s->adjw = (char*)adjoint; s->gridw = (char*)grid; s->miscw = (char*)misc;
    char* p = s->adjw;
    auto take = [&](size_t bytes) { char* r = p; p += align_up(bytes, 256); return r; };
    s->adj[0] = take((size_t)24 * s->Npad * s->tsz);
    s->adj[1] = s->adj[0] + (size_t)24 * s->Npad * s->tsz;
    p = s->adjw + align_up(2 * 24 * s->Npad * s->tsz, 256);
    s->mu = take(s->Npad * s->tsz); s->lam = take(s->Npad * s->tsz); s->ys = take(s->Npad * s->tsz);
    s->perm_d = (int*)take((size_t)s->Npad * 4);
    s->mats_master = (double*)take((size_t)3 * s->Npad * 8);
    if (s->resort || s->dist) {
        s->perm_store = (int*)take((size_t)(s->n_epochs - 1) * s->Npad * 4);
        s->vend = take((size_t)s->n_epochs * 3 * s->Npad * s->tsz);
        for (int i = 0; i < 2; ++i) { s->skey[i] = (unsigned*)take((size_t)s->sort_cap * 4); s->sidx[i] = (int*)take((size_t)s->sort_cap * 4); }
        s->sort_tmp = take(s->sort_tmp_bytes);
        if (s->cell_bins) s->cell_hist = (unsigned*)take(s->cell_bins * 4);
        s->frame_tmp = take(s->frame_bytes);
    }
    if (s->dist) {
        s->gid_store = (int*)take((size_t)s->n_epochs * s->Npad * 4);
        s->mig_src = (int*)take((size_t)s->n_epochs * s->Npad * 4);
        s->mig_leave = (int*)take((size_t)s->n_epochs * s->Npad * 4);
        s->mats_store = take((size_t)s->n_epochs * 3 * s->Npad * s->tsz);
        s->mig_dest = (int*)take((size_t)s->Npad * 4);
        s->iota = (int*)take((size_t)s->Npad * 4);
        s->mig_cnt = (int*)take(256);
        for (int i = 0; i < 2; ++i) s->mig_send[i] = (double*)take((size_t)s->mig_max_rows * 28 * 8);
    }
    REQUIRE((size_t)(p - s->adjw) <= s->ws.adjoint_bytes, "internal: adjoint workspace overflow");
    p = s->gridw;
    s->grid_in = take(s->G * 4 * s->tsz); s->grid_out = take(s->G * 4 * s->tsz);
    s->grid_out_adj = take(s->G * 4 * s->tsz); s->grid_in_adj = take(s->G * 4 * s->tsz);
    s->flags = (int*)take((size_t)s->nflag * 4);
    s->loss_gm = take(s->G * s->tsz); s->loss_td = take(s->G * s->tsz); s->loss_ts = take(s->G * s->tsz);
    if (s->store) {
        s->gstore = take((size_t)s->F * s->gstride);
        s->vstore = take((size_t)s->F * s->gstride);
        s->fstore = (int*)take((size_t)s->F * s->nflag * 4);
    }
    s->tiles = (int*)take((size_t)(s->F + 1) * (s->Npad / kBlock) * 8 * 4);
    s->contact = (int*)take((size_t)(s->nblk + 1) * 4);
    s->ptab = take((size_t)(s->F + 1) * kMaxPrim * sizeof(PrimT<double>));
    s->det_grid = s->det ? (long long*)take(s->G * 8 * 8) : nullptr;
    REQUIRE((size_t)(p - s->gridw) <= s->ws.grid_bytes, "internal: grid workspace overflow");
    p = s->miscw;
    size_t P1 = std::max(s->P, 1), F1 = s->F + 1;
    s->ppos = (double*)take(F1 * P1 * 3 * 8); s->prot = (double*)take(F1 * P1 * 4 * 8);
    s->ppos_a = (double*)take(F1 * P1 * 3 * 8); s->prot_a = (double*)take(F1 * P1 * 4 * 8);
    s->pv = (double*)take(F1 * P1 * 3 * 8); s->pw = (double*)take(F1 * P1 * 3 * 8);
    s->pv_a = (double*)take(F1 * P1 * 3 * 8); s->pw_a = (double*)take(F1 * P1 * 3 * 8);
    s->pgap = (double*)take(F1 * P1 * 8); s->pgap_a = (double*)take(F1 * P1 * 8);
    s->pgv = (double*)take(F1 * P1 * 8); s->pgv_a = (double*)take(F1 * P1 * 8);
    s->act = (double*)take(F1 * P1 * PLMPM_MAX_ACTION_DIM * 8); s->act_a = (double*)take(F1 * P1 * PLMPM_MAX_ACTION_DIM * 8);
    s->lscal = (double*)take(LS_COUNT * 8);
    s->staging = (double*)take((size_t)s->Npad * 24 * 8);
    s->err_d = (int*)take(512);
    if (s->dist) { s->ppos_l = (double*)take(F1 * P1 * 3 * 8); s->prot_l = (double*)take(F1 * P1 * 4 * 8); s->pgap_l = (double*)take(F1 * P1 * 8); }
    s->det_small = s->det ? (long long*)take((size_t)(LS_COUNT + kMaxPrim * 8) * 2 * 8) : nullptr;
    REQUIRE((size_t)(p - s->miscw) <= s->ws.misc_bytes, "internal: misc workspace overflow");
    // initial contents: zero grids / adjoints / primitive buffers, identity order
    HIPCHK(hipMemsetAsync(s->state, 0, s->ws.state_bytes, s->stream));       // frames are first touched here, not inside a caller's timed region
    HIPCHK(hipMemsetAsync(s->adjw, 0, s->ws.adjoint_bytes, s->stream));
    HIPCHK(hipMemsetAsync(s->gridw, 0, s->ws.grid_bytes, s->stream));
    HIPCHK(hipMemsetAsync(s->miscw, 0, s->ws.misc_bytes, s->stream));
    HIPCHK(hipMemcpyAsync(s->perm_d, s->perm.data(), (size_t)s->N * 4, hipMemcpyHostToDevice, s->stream));
    if (s->dist) {
        std::vector<int> io(s->Npad);
        for (int i = 0; i < s->Npad; ++i) io[i] = i;
        HIPCHK(hipMemcpyAsync(s->iota, io.data(), (size_t)s->Npad * 4, hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->gid_store, io.data(), (size_t)s->N * 4, hipMemcpyHostToDevice, s->stream));      // default ids: the row index
        HIPCHK(hipStreamSynchronize(s->stream));          // `io` goes out of scope
        s->ids0.assign(io.begin(), io.begin() + s->N);
    }
    // unit quaternions everywhere so an unset primitive frame is still a valid pose
    std::vector<double> rot(F1 * P1 * 4, 0.0);
    for (size_t i = 0; i < F1 * P1; ++i) rot[4 * i] = 1.0;
    HIPCHK(hipMemcpyAsync(s->prot, rot.data(), rot.size() * 8, hipMemcpyHostToDevice, s->stream));
    if (s->resort || s->dist) {
        // first use of the library sort loads its code object (~20 ms): pay that here, not in the first re-sorted step
        if (plmpm_sort_pairs(s->sort_tmp, s->sort_tmp_bytes, s->skey[0], s->skey[1], s->sidx[0], s->sidx[1], s->Npad, 8, s->stream) != 0)
            return fail("device sort unavailable");
    }
    HIPCHK(hipStreamSynchronize(s->stream));
    s->bound = true;
    return 0;
}

int plmpm_set_stream(plmpm_handle s, void* hip_stream) {
    REQUIRE(s, "null handle");
    s->stream = (hipStream_t)hip_stream;
    return 0;
}


int plmpm_set_materials(plmpm_handle s, const double* mu, const double* lam, const double* ys) {
    NEED_BOUND(s);
    REQUIRE(mu && lam && ys, "null argument");
    size_t nb = (size_t)s->N * 8;
    // the common case -- one material for the whole body -- needs no re-gather when the storage order changes
    bool uni = true;
    for (int i = 1; i < s->N && uni; ++i) uni = mu[i] == mu[0] && lam[i] == lam[0] && ys[i] == ys[0];
    s->mats_uniform = uni;
    if (s->N > 0) { s->mats_u[0] = mu[0]; s->mats_u[1] = lam[0]; s->mats_u[2] = ys[0]; }      // (a slab rank may hold no rows)
    s->mats_filled = false;
    HIPCHK(hipMemcpyAsync(s->mats_master, mu, nb, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(s->mats_master + s->N, lam, nb, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(s->mats_master + 2 * (size_t)s->N, ys, nb, hipMemcpyHostToDevice, s->stream));
    s->have_mats = true;
    DISPATCH(s, set_materials_t, s, s->mats_epoch);
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}


// Index of cell (b0,b1,b2) along the 3-D Hilbert curve over a 2^bits cube (Skilling, "Programming the Hilbert
// curve", AIP Conf. Proc. 707, 2004).  Consecutive indices are face-adjacent cells, so any run of the sorted
// particle list -- a wavefront, a workgroup -- covers a compact box of cells.
static uint64_t hilbert_index(int b0, int b1, int b2, int bits) {
    uint32_t X[3] = {(uint32_t)b0, (uint32_t)b1, (uint32_t)b2};
    const uint32_t M = 1u << (bits - 1);
    for (uint32_t Q = M; Q > 1; Q >>= 1) {
        const uint32_t P = Q - 1;
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) X[0] ^= P;
            else { uint32_t t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    for (int i = 1; i < 3; ++i) X[i] ^= X[i - 1];
    uint32_t t = 0;
    for (uint32_t Q = M; Q > 1; Q >>= 1) if (X[2] & Q) t ^= Q - 1;
    for (int i = 0; i < 3; ++i) X[i] ^= t;
    uint64_t h = 0;
    for (int bit = bits - 1; bit >= 0; --bit)
        for (int i = 0; i < 3; ++i) h = (h << 1) | ((X[i] >> bit) & 1u);
    return h;
}

// Storage order chosen at reset: particles sorted along the Hilbert curve of their stencil-base cell (ties keep
// caller order).  With a Morton order ~4 % of the 256-particle workgroups straddle a long jump of the curve and
// their stencil box overflows the LDS tile; along the Hilbert curve the mean box is 230 nodes instead of 410 and
// 0.3 % overflow (config-3 cloud).
static void compute_order(plmpm_sim* s, const double* x) {
    const int n = s->n;
    int bits = 1;
    while ((1 << bits) < n) ++bits;
    std::vector<std::pair<uint64_t, int32_t>> key(s->N);
    for (int i = 0; i < s->N; ++i) {
        int b[3];
        for (int d = 0; d < 3; ++d) {
            b[d] = (int)(x[(size_t)3 * i + d] * n - 0.5);
            b[d] = std::min(std::max(b[d], 0), n - 1);
        }
        key[i] = {hilbert_index(b[0], b[1], b[2], bits), (int32_t)i};
    }
    std::sort(key.begin(), key.end());
    for (int i = 0; i < s->N; ++i) s->perm[i] = key[i].second;
}

int plmpm_set_frame(plmpm_handle s, int frame, const double* x, const double* v, const double* F, const double* C, int resort) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    if (resort) {
        REQUIRE(x && v && F && C, "resort needs the full state (all of x, v, F, C)");
        compute_order(s, x);
        HIPCHK(hipMemcpyAsync(s->perm_d, s->perm.data(), (size_t)s->N * 4, hipMemcpyHostToDevice, s->stream));
        std::fill(s->frame_epoch.begin(), s->frame_epoch.end(), 0);          // a new episode: every frame in the reset order
        s->epochN[0] = s->N;
        s->next_epoch = 1;
        s->steps_since_sort = 0;
        s->mats_epoch = 0;                       // whatever epoch the last rollout ended in: the materials that follow are epoch 0's
        if (s->dist) {       // global ids in the reset order
            std::vector<int32_t> g(s->N);
            for (int i = 0; i < s->N; ++i) g[i] = s->ids0[s->perm[i]];
            HIPCHK(hipMemcpyAsync(s->gid_store, g.data(), (size_t)s->N * 4, hipMemcpyHostToDevice, s->stream));
            HIPCHK(hipStreamSynchronize(s->stream));
        }
        if (s->have_mats) DISPATCH(s, set_materials_t, s, 0);
    }
    const size_t N = s->epochN[s->frame_epoch[frame]];
    if (x) HIPCHK(hipMemcpyAsync(s->staging, x, N * 3 * 8, hipMemcpyHostToDevice, s->stream));
    if (v) HIPCHK(hipMemcpyAsync(s->staging + 3 * N, v, N * 3 * 8, hipMemcpyHostToDevice, s->stream));
    if (F) HIPCHK(hipMemcpyAsync(s->staging + 6 * N, F, N * 9 * 8, hipMemcpyHostToDevice, s->stream));
    if (C) HIPCHK(hipMemcpyAsync(s->staging + 15 * N, C, N * 9 * 8, hipMemcpyHostToDevice, s->stream));
    DISPATCH(s, unpack_t, s, frame, x != nullptr, v != nullptr, F != nullptr, C != nullptr);
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

int plmpm_get_frame(plmpm_handle s, int frame, double* x, double* v, double* F, double* C) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    const size_t N = s->epochN[s->frame_epoch[frame]];
    DISPATCH(s, pack_t, s, frame);
    if (x) HIPCHK(hipMemcpyAsync(x, s->staging, N * 3 * 8, hipMemcpyDeviceToHost, s->stream));
    if (v) HIPCHK(hipMemcpyAsync(v, s->staging + 3 * N, N * 3 * 8, hipMemcpyDeviceToHost, s->stream));
    if (F) HIPCHK(hipMemcpyAsync(F, s->staging + 6 * N, N * 9 * 8, hipMemcpyDeviceToHost, s->stream));
    if (C) HIPCHK(hipMemcpyAsync(C, s->staging + 15 * N, N * 9 * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

int plmpm_copy_frame(plmpm_handle s, int source, int target) {
    NEED_BOUND(s);
    NEED_FRAME(s, source);
    NEED_FRAME(s, target);
    size_t n16 = s->frame_bytes / 16;
    hipLaunchKernelGGL(k_copy_frame, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s->stream, s->state, s->frame_bytes, source, target);
    s->frame_epoch[target] = s->frame_epoch[source];
    if (s->P > 0) {
        HIPCHK(hipMemcpyAsync(s->ppos + (size_t)target * s->P * 3, s->ppos + (size_t)source * s->P * 3, (size_t)s->P * 3 * 8, hipMemcpyDeviceToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->prot + (size_t)target * s->P * 4, s->prot + (size_t)source * s->P * 4, (size_t)s->P * 4 * 8, hipMemcpyDeviceToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->pgap + (size_t)target * s->P, s->pgap + (size_t)source * s->P, (size_t)s->P * 8, hipMemcpyDeviceToDevice, s->stream));
    }
    return 0;
}

__global__ __launch_bounds__(256) void k_copy16(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
__global__ __launch_bounds__(256) void k_read16(const uint4* __restrict__ in, unsigned* __restrict__ out, size_t n) {
    uint4 a = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { uint4 v = in[i]; a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w; }
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x9e3779b9u) out[0] = a.x;
}
int plmpm_measure_hbm(void* src, void* dst, size_t bytes, int reps, void* hip_stream, double* copy_gbs, double* read_gbs) {
    REQUIRE(src && dst && bytes >= (1u << 20) && reps > 0 && copy_gbs && read_gbs, "measure_hbm: bad arguments");
    hipStream_t st = (hipStream_t)hip_stream;
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    const size_t n = bytes / 16;
    double best[2] = {0, 0};
    for (int k = 0; k < 2; ++k)
        for (int r = 0; r < reps + 1; ++r) {
            HIPCHK(hipEventRecord(a, st));
            if (k == 0) hipLaunchKernelGGL(k_copy16, dim3(16384), dim3(256), 0, st, (const uint4*)src, (uint4*)dst, n);
            else hipLaunchKernelGGL(k_read16, dim3(2048), dim3(256), 0, st, (const uint4*)src, (unsigned*)dst, n);
            HIPCHK(hipEventRecord(b, st));
            HIPCHK(hipEventSynchronize(b));
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, a, b));
            if (r > 0) best[k] = std::max(best[k], (k == 0 ? 2.0 : 1.0) * (double)bytes / (ms * 1e6));      // first run warms up
        }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *copy_gbs = best[0]; *read_gbs = best[1];
    return 0;
}
int plmpm_set_softness(plmpm_handle s, double softness) {
    REQUIRE(s, "null handle");
    s->softness = softness;
    return 0;
}

int plmpm_substep(plmpm_handle s, int frame) {
    NEED_BOUND(s);
    REQUIRE(frame >= 0 && frame < s->F, "substep: frame %d out of range", frame);
    plmpm_launch_fk(s, frame, 1);
    if (s->have_mats && s->mats_epoch != s->frame_epoch[frame]) DISPATCH(s, set_materials_t, s, s->frame_epoch[frame]);
    s->frame_epoch[frame + 1] = s->frame_epoch[frame];
    DISPATCH(s, substep_fwd, s, frame);
    HIPCHK(hipGetLastError());
    return 0;
}

int plmpm_step(plmpm_handle s, int first_frame, int n_substeps) {
    NEED_BOUND(s);
    REQUIRE(first_frame >= 0 && n_substeps > 0 && first_frame + n_substeps <= s->F, "step: frames [%d,%d] exceed max_frames %d",
            first_frame, first_frame + n_substeps, s->F);
    plmpm_launch_fk(s, first_frame, n_substeps);
    // re-sort the step's first frame along the Hilbert curve (not the episode's first step: set_frame just sorted it)
    const int span = std::max(s->cfg.substeps, 1) * std::max(s->cfg.resort_steps, 1);       // frames per storage order
    int epoch = first_frame / span;
    bool sort_now = s->resort && first_frame > 0 && n_substeps == s->cfg.substeps && first_frame % span == 0 && epoch < s->n_epochs;
    if (s->resort && first_frame == 0 && s->steps_since_sort >= s->cfg.resort_steps && s->n_epochs >= 3) {
        // copy-mode episodes (Gym step(): every env step starts again at frame 0): same cadence, two alternating epochs
        epoch = s->frame_epoch[0] == 1 ? 2 : 1;
        sort_now = true;
    }
    if (sort_now && !s->prof_no_resort && s->frame_epoch[first_frame] != epoch) {
        if (DISPATCH(s, resort_frame_t, s, first_frame, epoch)) return -1;
        s->steps_since_sort = 0;
    }
    ++s->steps_since_sort;
    const int e = s->frame_epoch[first_frame];
    if (s->have_mats && s->mats_epoch != e) DISPATCH(s, set_materials_t, s, e);
    for (int f = first_frame + 1; f <= first_frame + n_substeps; ++f) s->frame_epoch[f] = e;      // before the launches: they size by epoch
    if (s->store && n_substeps > 1) DISPATCH(s, step_fwd_fused, s, first_frame, n_substeps);
    else for (int f = first_frame; f < first_frame + n_substeps; ++f) DISPATCH(s, substep_fwd, s, f);
    HIPCHK(hipGetLastError());
    return 0;
}
int plmpm_set_resort(plmpm_handle s, int on) {
    REQUIRE(s, "null handle");
    s->prof_no_resort = !on;
    return 0;
}

int plmpm_grad_begin(plmpm_handle s, int last_frame) {
    NEED_BOUND(s);
    NEED_FRAME(s, last_frame);
    HIPCHK(hipMemsetAsync(s->adj[0], 0, (size_t)2 * 24 * s->Npad * s->tsz, s->stream));
    size_t P1 = std::max(s->P, 1), F1 = s->F + 1;
    HIPCHK(hipMemsetAsync(s->ppos_a, 0, F1 * P1 * 3 * 8, s->stream));
    HIPCHK(hipMemsetAsync(s->prot_a, 0, F1 * P1 * 4 * 8, s->stream));
    HIPCHK(hipMemsetAsync(s->pv_a, 0, F1 * P1 * 3 * 8, s->stream));
    HIPCHK(hipMemsetAsync(s->pw_a, 0, F1 * P1 * 3 * 8, s->stream));
    HIPCHK(hipMemsetAsync(s->pgap_a, 0, F1 * P1 * 8, s->stream));
    HIPCHK(hipMemsetAsync(s->pgv_a, 0, F1 * P1 * 8, s->stream));
    HIPCHK(hipMemsetAsync(s->act_a, 0, F1 * P1 * PLMPM_MAX_ACTION_DIM * 8, s->stream));
    if (s->dist) {
        HIPCHK(hipMemsetAsync(s->ppos_l, 0, F1 * P1 * 3 * 8, s->stream));
        HIPCHK(hipMemsetAsync(s->prot_l, 0, F1 * P1 * 4 * 8, s->stream));
        HIPCHK(hipMemsetAsync(s->pgap_l, 0, F1 * P1 * 8, s->stream));
    }
    s->adj_frame[last_frame & 1] = last_frame;
    s->adj_frame[(last_frame + 1) & 1] = -1;
    s->adj_epoch[0] = s->adj_epoch[1] = s->frame_epoch[last_frame];     // all zero: any order
    return 0;
}

// Segment-checkpointed backward (plb/optimizer/long_term_gradient.ipynb cell 2, copy_and_clear): the adjoint held
// for frame `from_frame` (start of the later segment) becomes the adjoint of `to_frame` (end of the earlier
// segment, whose forward was just re-run); primitive pose adjoints move with it, every other pose / velocity /
// action adjoint is cleared.
__global__ void k_move_double(double* dst, const double* src, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
int plmpm_segment_carry(plmpm_handle s, int from_frame, int to_frame) {
    NEED_BOUND(s);
    NEED_FRAME(s, from_frame);
    NEED_FRAME(s, to_frame);
    REQUIRE(s->adj_frame[from_frame & 1] == from_frame, "segment_carry: adjoint of frame %d is not resident", from_frame);
    if ((from_frame & 1) != (to_frame & 1))
        HIPCHK(hipMemcpyAsync(s->adj[to_frame & 1], s->adj[from_frame & 1], (size_t)24 * s->Npad * s->tsz, hipMemcpyDeviceToDevice, s->stream));
    s->adj_epoch[to_frame & 1] = s->adj_epoch[from_frame & 1];
    s->adj_frame[to_frame & 1] = to_frame;
    s->adj_frame[(to_frame + 1) & 1] = -1;
    size_t P1 = std::max(s->P, 1), F1 = s->F + 1;
    if (s->P > 0) {
        double* tmp = s->staging;                       // 8 * P doubles of scratch (staging holds >= 24 * N)
        hipLaunchKernelGGL(k_move_double, dim3(1), dim3(64), 0, s->stream, tmp, s->ppos_a + (size_t)from_frame * s->P * 3, s->P * 3);
        hipLaunchKernelGGL(k_move_double, dim3(1), dim3(64), 0, s->stream, tmp + 32, s->prot_a + (size_t)from_frame * s->P * 4, s->P * 4);
        hipLaunchKernelGGL(k_move_double, dim3(1), dim3(64), 0, s->stream, tmp + 64, s->pgap_a + (size_t)from_frame * s->P, s->P);
        HIPCHK(hipMemsetAsync(s->ppos_a, 0, F1 * P1 * 3 * 8, s->stream));
        HIPCHK(hipMemsetAsync(s->prot_a, 0, F1 * P1 * 4 * 8, s->stream));
        HIPCHK(hipMemsetAsync(s->pgap_a, 0, F1 * P1 * 8, s->stream));
        hipLaunchKernelGGL(k_move_double, dim3(1), dim3(64), 0, s->stream, s->ppos_a + (size_t)to_frame * s->P * 3, tmp, s->P * 3);
        hipLaunchKernelGGL(k_move_double, dim3(1), dim3(64), 0, s->stream, s->prot_a + (size_t)to_frame * s->P * 4, tmp + 32, s->P * 4);
        hipLaunchKernelGGL(k_move_double, dim3(1), dim3(64), 0, s->stream, s->pgap_a + (size_t)to_frame * s->P, tmp + 64, s->P);
    }
    HIPCHK(hipMemsetAsync(s->pv_a, 0, F1 * P1 * 3 * 8, s->stream));
    HIPCHK(hipMemsetAsync(s->pw_a, 0, F1 * P1 * 3 * 8, s->stream));
    HIPCHK(hipMemsetAsync(s->pgv_a, 0, F1 * P1 * 8, s->stream));
    HIPCHK(hipMemsetAsync(s->act_a, 0, F1 * P1 * PLMPM_MAX_ACTION_DIM * 8, s->stream));
    HIPCHK(hipGetLastError());
    return 0;
}

// reverse of the substep that starts at `frame`: bring the incoming adjoint (of frame+1) and the materials into the
// storage order of `frame`, then run it
static int bwd_prepare(plmpm_sim* s, int frame) {
    const int e = s->frame_epoch[frame], slot = (frame + 1) & 1;
    if (s->adj_epoch[slot] != e && DISPATCH(s, convert_adjoint_t, s, slot, s->adj_epoch[slot], e)) return -1;
    if (s->have_mats && s->mats_epoch != e) DISPATCH(s, set_materials_t, s, e);
    s->adj_epoch[frame & 1] = e;
    return 0;
}
int plmpm_substep_grad(plmpm_handle s, int frame) {
    NEED_BOUND(s);
    REQUIRE(frame >= 0 && frame < s->F, "substep_grad: frame %d out of range", frame);
    REQUIRE(s->adj_frame[(frame + 1) & 1] == frame + 1, "substep_grad(%d): adjoint of frame %d is not resident (call grad_begin / go in reverse order)", frame, frame + 1);
    if (bwd_prepare(s, frame)) return -1;
    DISPATCH(s, substep_bwd, s, frame);
    HIPCHK(hipGetLastError());
    return 0;
}

int plmpm_step_grad(plmpm_handle s, int first_frame, int n_substeps, int step) {
    NEED_BOUND(s);
    REQUIRE(first_frame >= 0 && n_substeps > 0 && first_frame + n_substeps <= s->F, "step_grad: bad frame range");
    for (int f = first_frame + n_substeps - 1; f >= first_frame; --f) {
        REQUIRE(s->adj_frame[(f + 1) & 1] == f + 1, "step_grad: adjoint of frame %d is not resident", f + 1);
        if (bwd_prepare(s, f)) return -1;
        DISPATCH(s, substep_bwd, s, f);
    }
    if (s->P > 0) plmpm_launch_fk_grad(s, first_frame, n_substeps, step);
    HIPCHK(hipGetLastError());
    return 0;
}

int plmpm_add_frame_grad(plmpm_handle s, int frame, const double* xa, const double* va, const double* Fa, const double* Ca) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(s->adj_frame[frame & 1] == frame, "add_frame_grad: adjoint of frame %d is not resident", frame);
    const size_t N = s->epochN[s->frame_epoch[frame]];
    if (s->adj_epoch[frame & 1] != s->frame_epoch[frame] &&
        DISPATCH(s, convert_adjoint_t, s, frame & 1, s->adj_epoch[frame & 1], s->frame_epoch[frame])) return -1;
    if (xa) HIPCHK(hipMemcpyAsync(s->staging, xa, N * 3 * 8, hipMemcpyHostToDevice, s->stream));
    if (va) HIPCHK(hipMemcpyAsync(s->staging + 3 * N, va, N * 3 * 8, hipMemcpyHostToDevice, s->stream));
    if (Fa) HIPCHK(hipMemcpyAsync(s->staging + 6 * N, Fa, N * 9 * 8, hipMemcpyHostToDevice, s->stream));
    if (Ca) HIPCHK(hipMemcpyAsync(s->staging + 15 * N, Ca, N * 9 * 8, hipMemcpyHostToDevice, s->stream));
    DISPATCH(s, adj_io_t, s, frame & 1, 1, xa != nullptr, va != nullptr, Fa != nullptr, Ca != nullptr);
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}
int plmpm_get_frame_grad(plmpm_handle s, int frame, double* xa, double* va, double* Fa, double* Ca) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(s->adj_frame[frame & 1] == frame, "get_frame_grad: adjoint of frame %d is not resident", frame);
    const size_t N = s->epochN[s->adj_epoch[frame & 1]];
    DISPATCH(s, adj_io_t, s, frame & 1, 0, 1, 1, 1, 1);
    if (xa) HIPCHK(hipMemcpyAsync(xa, s->staging, N * 3 * 8, hipMemcpyDeviceToHost, s->stream));
    if (va) HIPCHK(hipMemcpyAsync(va, s->staging + 3 * N, N * 3 * 8, hipMemcpyDeviceToHost, s->stream));
    if (Fa) HIPCHK(hipMemcpyAsync(Fa, s->staging + 6 * N, N * 9 * 8, hipMemcpyDeviceToHost, s->stream));
    if (Ca) HIPCHK(hipMemcpyAsync(Ca, s->staging + 15 * N, N * 9 * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

int plmpm_grid_stats(plmpm_handle s, int frame, int64_t* active_nodes, int64_t* active_blocks) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    unsigned long long* d_out;
    unsigned long long h[2] = {0, 0};
    HIPCHK(hipMalloc(&d_out, 16));
    HIPCHK(hipMemsetAsync(d_out, 0, 16, s->stream));
    DISPATCH(s, grid_stats_t, s, frame, d_out);
    HIPCHK(hipMemcpyAsync(h, d_out, 16, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    hipFree(d_out);
    if (active_nodes) *active_nodes = (int64_t)h[0];
    if (active_blocks) *active_blocks = (int64_t)h[1];
    return 0;
}

int plmpm_tile_boxes(plmpm_handle s, int frame, int32_t* out, int max_workgroups, int* n_workgroups) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    const int nwg = nblocks_particles(s, frame);
    if (n_workgroups) *n_workgroups = nwg;
    if (!out) return 0;
    REQUIRE(max_workgroups >= nwg, "tile_boxes: room for %d workgroups, need %d", max_workgroups, nwg);
    std::vector<int> h((size_t)nwg * 8);
    HIPCHK(hipMemcpyAsync(h.data(), s->tiles + (size_t)frame * nwg * 8, h.size() * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    for (int w = 0; w < nwg; ++w)
        for (int k = 0; k < 6; ++k) out[w * 6 + k] = h[(size_t)w * 8 + k];
    return 0;
}

int plmpm_p2g(plmpm_handle s, int frame, int chain) {
    REQUIRE(s && s->bound, "workspace not bound");
    REQUIRE(s->store, "the phase-split substep needs store_grid = 1");
    REQUIRE(frame >= 0 && frame < s->F, "p2g: frame out of range");
    if (chain) {
        REQUIRE(frame >= 1 && s->g2p_deferred == frame - 1, "p2g(%d, chain): frame %d's g2p was not deferred (deferred: %d)", frame, frame - 1, s->g2p_deferred);
        s->g2p_deferred = -1;
        DISPATCH(s, phase_g2p_p2g, s, frame);
    } else {
        REQUIRE(s->g2p_deferred < 0, "frame %d's g2p is deferred: call plmpm_p2g(frame + 1, chain = 1) next", s->g2p_deferred);
        DISPATCH(s, phase_p2g, s, frame);
    }
    HIPCHK(hipGetLastError());
    return 0;
}
int plmpm_grid_interior(plmpm_handle s, int frame) {
    NEED_BOUND(s);
    REQUIRE(s->store && frame >= 0 && frame < s->F, "grid_interior: bad call");
    REQUIRE(s->halo_in[PLMPM_HALO_GRID_IN].n > 0, "grid_interior: no halo planes are registered for PLMPM_HALO_GRID_IN (nothing to overlap with)");
    DISPATCH(s, phase_grid_g2p, s, frame, false, 1);
    HIPCHK(hipGetLastError());
    s->interior_fwd = frame;
    return 0;
}
int plmpm_grid_g2p(plmpm_handle s, int frame, int chain) {
    NEED_BOUND(s);
    REQUIRE(s->store && frame >= 0 && frame < s->F, "grid_g2p: bad call");
    REQUIRE(!chain || frame + 1 < s->F, "grid_g2p: only a substep with a successor chains");
    const int part = s->interior_fwd == frame ? 2 : 0;         // the interior blocks were done by plmpm_grid_interior
    s->interior_fwd = -1;
    DISPATCH(s, phase_grid_g2p, s, frame, chain != 0, part);
    HIPCHK(hipGetLastError());
    if (chain) s->g2p_deferred = frame;
    return 0;
}
int plmpm_grad_scatter(plmpm_handle s, int frame) {
    NEED_BOUND(s);
    REQUIRE(s->store && frame >= 0 && frame < s->F, "grad_scatter: bad call");
    REQUIRE(s->dirty[frame], "grad_scatter(%d): the frame's grid is not resident (run the forward substep first)", frame);
    REQUIRE(s->adj_frame[(frame + 1) & 1] == frame + 1, "grad_scatter(%d): adjoint of frame %d is not resident", frame, frame + 1);
    REQUIRE(s->adj_epoch[(frame + 1) & 1] == s->frame_epoch[frame], "grad_scatter(%d): the adjoint of frame %d is in storage epoch %d, this substep ran in "
            "epoch %d (particles migrated at that frame: run plmpm_migrate_adjoint_begin / _finish first)", frame, frame + 1,
            s->adj_epoch[(frame + 1) & 1], s->frame_epoch[frame]);
    DISPATCH(s, phase_grad_scatter, s, frame);
    HIPCHK(hipGetLastError());
    return 0;
}
int plmpm_grad_gather_interior(plmpm_handle s, int frame) {
    NEED_BOUND(s);
    REQUIRE(s->store && frame >= 0 && frame < s->F, "grad_gather_interior: bad call");
    REQUIRE(s->halo_in[PLMPM_HALO_GRID_OUT_ADJ].n > 0, "grad_gather_interior: no halo planes are registered for PLMPM_HALO_GRID_OUT_ADJ");
    DISPATCH(s, phase_grad_gather, s, frame, 1);
    HIPCHK(hipGetLastError());
    s->interior_bwd = frame;
    return 0;
}
int plmpm_grad_gather(plmpm_handle s, int frame) {
    NEED_BOUND(s);
    REQUIRE(s->store && frame >= 0 && frame < s->F, "grad_gather: bad call");
    const int part = s->interior_bwd == frame ? 2 : 0;
    s->interior_bwd = -1;
    DISPATCH(s, phase_grad_gather, s, frame, part);
    HIPCHK(hipGetLastError());
    return 0;
}
// ---- halos: zero-copy exchange of whole block planes ---------------------------------------------------------------
}  // extern "C"
int plmpm_grid_g2p_xchg(plmpm_sim* s, int frame, int chain, const PeerXchg* X) {
    NEED_BOUND(s);
    REQUIRE(s->store && frame >= 0 && frame < s->F, "grid_g2p: bad call");
    REQUIRE(!chain || frame + 1 < s->F, "grid_g2p: only a substep with a successor chains");
    REQUIRE(s->interior_fwd != frame, "grid_g2p with the exchange folded in: the interior blocks were already done by plmpm_grid_interior");
    DISPATCH(s, phase_grid_g2p, s, frame, chain != 0, 0, X);
    HIPCHK(hipGetLastError());
    if (chain) s->g2p_deferred = frame;
    return 0;
}
int plmpm_grad_gather_xchg(plmpm_sim* s, int frame, const PeerXchg* X) {
    NEED_BOUND(s);
    REQUIRE(s->store && frame >= 0 && frame < s->F, "grad_gather: bad call");
    REQUIRE(s->interior_bwd != frame, "grad_gather with the exchange folded in: the interior blocks were already done by plmpm_grad_gather_interior");
    DISPATCH(s, phase_grad_gather, s, frame, 0, X);
    HIPCHK(hipGetLastError());
    return 0;
}
int plmpm_halo_field(plmpm_sim* s, int field, int frame, char** base, int* ncomp) {
    if (field == PLMPM_HALO_GRID_IN) {
        REQUIRE(s->store && frame >= 0 && frame < s->F, "halo: grid_in needs store_grid and a valid frame");
        *base = s->gstore + (size_t)frame * s->gstride; *ncomp = 4;
    } else if (field == PLMPM_HALO_GRID_OUT_ADJ) { *base = s->grid_out_adj; *ncomp = 3; }
    else if (field == PLMPM_HALO_LOSS_MASS) { *base = s->loss_gm; *ncomp = 1; }
    else return fail("unknown halo field %d", field);
    return 0;
}
#define halo_field plmpm_halo_field
extern "C" {
int plmpm_grid_window(plmpm_handle s, int32_t* origin3, int32_t* blocks3) {
    REQUIRE(s && origin3 && blocks3, "null argument");
    for (int d = 0; d < 3; ++d) { origin3[d] = s->go[d]; blocks3[d] = s->nbw[d]; }
    return 0;
}
int plmpm_halo_region(plmpm_handle s, int field, int frame, int comp, int bz_a, int bz_b, void** dev_ptr, size_t* count) {
    NEED_BOUND(s);
    REQUIRE(dev_ptr && count, "null argument");
    char* base; int nc;
    if (halo_field(s, field, frame, &base, &nc)) return -1;
    REQUIRE(comp >= 0 && comp < nc, "halo_region: field %d has %d components", field, nc);
    const int ra = bz_a - s->go[2] / 4, rb = bz_b - s->go[2] / 4;       // block planes relative to the window
    REQUIRE(ra >= 0 && rb <= s->nbw[2] && ra < rb, "halo_region: block planes [%d,%d) outside this rank's grid window (planes [%d,%d))",
            bz_a, bz_b, s->go[2] / 4, s->go[2] / 4 + s->nbw[2]);
    const size_t plane = (size_t)s->nbw[0] * s->nbw[1] * 64;
    *dev_ptr = base + ((size_t)comp * s->G + (size_t)ra * plane) * s->tsz;
    *count = (size_t)(rb - ra) * plane;
    return 0;
}
int plmpm_halo_set_recv(plmpm_handle s, int field, int n_faces, const int* bz_a, const int* bz_b, void* const* recv) {
    REQUIRE(s, "null handle");
    REQUIRE(field >= 0 && field < 3 && n_faces >= 0 && n_faces <= 2, "halo_set_recv: a z-slab has at most 2 faces");
    HaloIn& H = s->halo_in[field];
    memset(&H, 0, sizeof H);
    for (int i = 0; i < n_faces; ++i) {
        const int ra = bz_a[i] - s->go[2] / 4, rb = bz_b[i] - s->go[2] / 4;
        REQUIRE(recv[i] && ra >= 0 && rb <= s->nbw[2] && ra < rb, "halo_set_recv: block planes [%d,%d) outside the grid window", bz_a[i], bz_b[i]);
        H.ba[i] = ra; H.bb[i] = rb; H.buf[i] = recv[i];
    }
    H.n = n_faces;
    return 0;
}
int plmpm_halo_apply(plmpm_handle s, int field, int frame) {
    NEED_BOUND(s);
    REQUIRE(field == PLMPM_HALO_LOSS_MASS, "halo_apply: the substep fields are added by grid_op / grid_op.grad themselves");
    char* base; int nc;
    if (halo_field(s, field, frame, &base, &nc)) return -1;
    const HaloIn& H = s->halo_in[field];
    const size_t plane = (size_t)s->nbw[0] * s->nbw[1] * 64;
    for (int i = 0; i < H.n; ++i) {
        const size_t cnt = (size_t)(H.bb[i] - H.ba[i]) * plane;
        for (int c = 0; c < nc; ++c) {
            char* dst = base + ((size_t)c * s->G + (size_t)H.ba[i] * plane) * s->tsz;
            const char* src = (const char*)H.buf[i] + (size_t)c * cnt * s->tsz;
            if (s->cfg.dtype == PLMPM_F64) hipLaunchKernelGGL((k_add_region<double>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s->stream, (double*)dst, (const double*)src, cnt);
            else hipLaunchKernelGGL((k_add_region<float>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s->stream, (float*)dst, (const float*)src, cnt);
        }
    }
    HIPCHK(hipGetLastError());
    return 0;
}

}  // extern "C"
extern "C" {
int plmpm_debug_counters(plmpm_handle s, int* out4) {
    NEED_BOUND(s);
    HIPCHK(hipMemcpyAsync(out4, s->err_d, 16, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipMemsetAsync(s->err_d + 1, 0, 12, s->stream));
    return 0;
}
// test hook: the list of blocks whose pose adjoints are due (D.contact).  seed >= 0 first overwrites it with `seed` entries
// naming block 0 (what a reverse substep that never reset the list would have left behind); *count = entries now listed
int plmpm_debug_contact(plmpm_handle s, int seed, int* count) {
    NEED_BOUND(s);
    REQUIRE(count && seed <= s->nblk, "bad argument");
    if (seed >= 0) {
        HIPCHK(hipMemsetAsync(s->contact, 0, (size_t)(seed + 1) * 4, s->stream));
        HIPCHK(hipMemcpyAsync(s->contact, &seed, 4, hipMemcpyHostToDevice, s->stream));
    }
    HIPCHK(hipMemcpyAsync(count, s->contact, 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}
#ifdef PLB_PHASE_TIMING
extern "C" int plmpm_debug_trace(plmpm_handle s, unsigned long long* out, size_t n) {       // profiling builds only
    NEED_BOUND(s);
    HIPCHK(hipMemcpyAsync(out, s->staging, n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}
#endif
int plmpm_check_error(plmpm_handle s, int* flags) {
    NEED_BOUND(s);
    REQUIRE(flags, "null argument");
    HIPCHK(hipMemcpyAsync(flags, s->err_d, 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (*flags) HIPCHK(hipMemsetAsync(s->err_d, 0, 4, s->stream));
    return 0;
}

int plmpm_profile_enable(plmpm_handle s, int on) {
    REQUIRE(s, "null handle");
    s->prof = on != 0;
    s->ev_used.clear();
    s->ev_next = 0;
    return 0;
}
int plmpm_profile_kernel_count(void) { return K_COUNT; }
const char* plmpm_profile_kernel_name(int id) { return (id >= 0 && id < K_COUNT) ? kKernelNames[id] : ""; }
int plmpm_profile_read(plmpm_handle s, double* total_ms, int64_t* launches) {
    REQUIRE(s && total_ms && launches, "null argument");
    HIPCHK(hipStreamSynchronize(s->stream));
    for (int i = 0; i < K_COUNT; ++i) { total_ms[i] = 0; launches[i] = 0; }
    for (auto& u : s->ev_used) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, s->ev_pool[u.second], s->ev_pool[u.second + 1]));
        total_ms[u.first] += ms;
        launches[u.first] += 1;
    }
    s->ev_used.clear();
    s->ev_next = 0;
    return 0;
}

}  // extern "C"
// Profiling aid: launch ONE hot-path kernel `reps` times on the state the engine is in and return its mean duration
// (HIP events on the launch stream).  Every build of the library lays plmpm_sim out identically, so a handle created
// by the default build can be replayed through an experiment build loaded next to it (profiles/tools/replay_ab.py):
// variants of a kernel are timed on bit-identical inputs, in one process.  The replayed launches accumulate into the
// grids they scatter to and overwrite the frame they produce -- the rollout is not usable afterwards.
//   kind 0: g2p(f-1)+p2g(f) fused forward kernel   1: g2p.grad(f)   2: p2g.grad(f)   3: p2g(f) alone
template <class T> static int replay_t(plmpm_sim* s, int kind, int f, int reps, double* us) {
    Dev<T> D = make_dev<T>(s, f);
    const int src = (f + 1) & 1, dst = f & 1;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    for (int r = -2; r < reps; ++r) {                     // two untimed launches first
        if (r == 0) HIPCHK(hipEventRecord(e0, s->stream));
        if (kind == 0) {
            const Vec4<T>* vprev = (const Vec4<T>*)(s->vstore + (size_t)(f - 1) * s->gstride);
            LAUNCH_G2P_P2G(s, D, f, vprev);
        } else if (kind == 1) {
            LAUNCH_G2P_GRAD(s, D, f, src, dst, (const T*)nullptr);
        } else if (kind == 2) {
            LAUNCH_P2G_GRAD(s, D, f, src, dst);
        } else {
            LAUNCH_P2G(s, K_P2G, true, D, f);
        }
    }
    HIPCHK(hipEventRecord(e1, s->stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *us = 1e3 * ms / reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIPCHK(hipGetLastError());
    return 0;
}
// The same for a whole env step: the forward substep loop (dir 0) or the reverse one (dir 1) of frames [first, first + n)
// launched `reps` times eagerly (graph 0) or as `reps` replays of ONE captured hipGraph (graph 1) -- what the launch
// boundaries cost on the stream and inside a graph, on identical work.  Timing only: every repetition re-executes the same
// frames (the scatters accumulate).
template <class T> static int replay_step_t(plmpm_sim* s, int graph, int dir, int first, int n, int reps, double* us) {
    auto body = [&]() {
        if (dir == 0) {
            for (int f = first; f < first + n; ++f) s->dirty[f] = 0;          // no clear launches: the same launch list every time
            step_fwd_fused<T>(s, first, n);
        } else {
            for (int f = first + n - 1; f >= first; --f) {
                s->dirty[f] = 1;                                              // the frame's grids are resident: no forward recompute
                s->adj_frame[(f + 1) & 1] = f + 1;
                substep_bwd<T>(s, f);
            }
        }
    };
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    if (graph) {
        HIPCHK(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
        body();
        HIPCHK(hipStreamEndCapture(s->stream, &g));
        HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    }
    for (int r = -2; r < reps; ++r) {
        if (r == 0) HIPCHK(hipEventRecord(e0, s->stream));
        if (graph) HIPCHK(hipGraphLaunch(ge, s->stream));
        else body();
    }
    HIPCHK(hipEventRecord(e1, s->stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *us = 1e3 * ms / reps;
    if (ge) (void)hipGraphExecDestroy(ge);
    if (g) (void)hipGraphDestroy(g);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" {
int plmpm_replay_step(plmpm_handle s, int graph, int dir, int first, int n, int reps, double* mean_us) {
    NEED_BOUND(s);
    REQUIRE(mean_us && reps > 0 && first >= 0 && n > 1 && first + n <= s->F, "replay_step: bad arguments");
    REQUIRE(s->store && !s->det && !s->prof, "replay_step: needs the per-frame grid store and the default engine, profiling off");
    return DISPATCH(s, replay_step_t, s, graph, dir, first, n, reps, mean_us);
}
int plmpm_replay(plmpm_handle s, int kind, int frame, int reps, double* mean_us) {
    NEED_BOUND(s);
    REQUIRE(mean_us && reps > 0 && kind >= 0 && kind <= 3, "replay: bad arguments");
    REQUIRE(frame >= (kind == 0 ? 1 : 0) && frame < s->F, "replay: frame %d out of range", frame);
    REQUIRE(s->store, "replay: needs the per-frame grid store and the default engine");
    return DISPATCH(s, replay_t, s, kind, frame, reps, mean_us);
}

int plmpm_get_order(plmpm_handle s, int32_t* perm) {
    REQUIRE(s && perm, "null argument");
    memcpy(perm, s->perm.data(), (size_t)s->N * 4);
    return 0;
}

}  // extern "C"

