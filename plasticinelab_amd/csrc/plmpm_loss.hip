// Loss (plb/engine/losses/loss.py:81-298): mass scatter, density / sdf / contact terms, their adjoint, the target SDF sweep.
#include "plmpm_internal.h"

// ---- loss -----------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double* sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double r = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double block_max(double v, double* sh) {
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double r = sh[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = fmax(r, sh[i]);
    __syncthreads();
    return r;
}
// density / sdf losses (loss.py:145-153) + IoU sums (loss.py:239-254)
// dl: deterministic mode only (else null) -- two integer limbs per loss scalar, see det_add (plmpm_kernels.h)
__device__ __forceinline__ void ls_add(double* ls, long long* dl, int slot, double v) {
    if (dl) det_add(dl + 2 * slot, dl + 2 * slot + 1, v);
    else atomicAdd(&ls[slot], v);
}
template <class T> __global__ void k_loss_reduce(size_t G, int nbxy, int gz, int z0, int z1, const T* gm, const T* td, const T* ts, double* ls, long long* dl) {
    __shared__ double sh[8];
    double dens = 0, sdf = 0, mx = 0, dot = 0, sum = 0;
    const unsigned nb2 = (unsigned)nbxy;                      // blocks per z-plane of the window (32-bit division)
    // four nodes (one x-row of a 4^3 block: the same z) per thread and iteration, 16-byte loads (32 for double): the sweep
    // runs one wave per SIMD (256 workgroups: each ends in five same-address double atomics, more of them cost more than
    // they win), so it lives on loads in flight -- with one 4-byte load per array and iteration it took 35 us for 25 MB
    const Vec4<T>* gm4 = reinterpret_cast<const Vec4<T>*>(gm);
    const Vec4<T>* td4 = reinterpret_cast<const Vec4<T>*>(td);
    const Vec4<T>* ts4 = reinterpret_cast<const Vec4<T>*>(ts);
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < G / 4; q += (size_t)gridDim.x * blockDim.x) {
        const size_t i = q * 4;
        int z = gz + (int)((unsigned)(i >> 6) / nb2) * 4 + (int)((i & 63) >> 4);
        if (z < z0 || z >= z1) continue;                      // nodes owned by another rank
        const Vec4<T> g4 = gm4[q], t4 = td4[q], s4 = ts4[q];
        const double g[4] = {(double)g4.x, (double)g4.y, (double)g4.z, (double)g4.w}, t[4] = {(double)t4.x, (double)t4.y, (double)t4.z, (double)t4.w};
        const double sv[4] = {(double)s4.x, (double)s4.y, (double)s4.z, (double)s4.w};
        for (int k = 0; k < 4; ++k) { dens += fabs(g[k] - t[k]); sdf += sv[k] * g[k]; mx = fmax(mx, g[k]); dot += g[k] * t[k]; sum += g[k]; }
    }
    dens = block_sum(dens, sh); sdf = block_sum(sdf, sh); dot = block_sum(dot, sh); sum = block_sum(sum, sh);
    mx = block_max(mx, sh);
    if (threadIdx.x == 0) {
        ls_add(ls, dl, LS_DENSITY, dens); ls_add(ls, dl, LS_SDF, sdf); ls_add(ls, dl, LS_DOT, dot); ls_add(ls, dl, LS_SUMGM, sum);
        atomicMax(reinterpret_cast<unsigned long long*>(&ls[LS_MAXGM]), (unsigned long long)__double_as_longlong(mx));
    }
}
// contact distance passes (loss.py:116-135).  mode 0: hard min, 1: soft normaliser, 2: soft weighted sum.
// Grid-stride over the particles with a bounded number of workgroups, one result per workgroup and primitive: the
// partial results all land on the same word, and same-address atomics cost ~5 ns EACH on this chip (one per wave
// made this kernel 180 us at 500k particles).
__device__ __forceinline__ double block_min(double v, double* sh) {
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double r = sh[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = fmin(r, sh[i]);
    __syncthreads();
    return r;
}
template <class T> __global__ void k_contact(Dev<T> D, int f, int mode, double* ls, long long* dl) {
    __shared__ double sh[8];
    const double* X = frame_x(D, f);
    for (int q = 0; q < D.nprim; ++q) {
        if (!D.prim[q].movable) continue;
        const PrimT<T> pr = prim_at(D, q, f);
        const double dn = mode == 2 ? ls[LS_DNORM + q] : 1.0;
        double acc = mode == 0 ? 1e30 : 0.0;
        for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < D.N; p += gridDim.x * blockDim.x) {
            const double x[3] = {X[p], X[D.Npad + p], X[2 * D.Npad + p]};
            const double d = fmax(prim_sdf(pr, x), 0.0);
            if (mode == 0) acc = fmin(acc, d);
            else {
                const double sw = 1.0 / (1.0 + d * d * 10000.0);
                acc += mode == 1 ? sw : d * sw / dn;
            }
        }
        if (mode == 0) {
            const double m = block_min(acc, sh);
            // non-negative doubles order like their bit patterns; skip the atomic when it cannot lower the minimum
            if (threadIdx.x == 0 && m < ls[LS_MIND + q])
                atomicMin(reinterpret_cast<unsigned long long*>(&ls[LS_MIND + q]), (unsigned long long)__double_as_longlong(m));
        } else {
            const double v = block_sum(acc, sh);
            if (threadIdx.x == 0) ls_add(ls, dl, (mode == 1 ? LS_DNORM : LS_MIND) + q, v);
        }
    }
}
// compute_loss_kernel_grad (loss.py:210-237) per particle: density + sdf through grid_m, contact through sdf.
template <class T>
__global__ void k_loss_grad(Dev<T> D, int f, int which, const T* gm, const T* td, const T* ts, const double* ls,
                            double w_sdf, double w_density, double w_contact, int soft, long long* dl, int argmin) {
    __shared__ double sacc[kMaxPrim * 8];
    __shared__ long long sdet[kMaxPrim * 8 * 2];          // deterministic mode: integer limbs instead of sacc
    if (threadIdx.x < kMaxPrim * 8) { sacc[threadIdx.x] = 0.0; sdet[2 * threadIdx.x] = 0; sdet[2 * threadIdx.x + 1] = 0; }
    __syncthreads();
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < D.N) {
        const double* X = frame_x(D, f);
        double x[3] = {X[p], X[D.Npad + p], X[2 * D.Npad + p]};
        int base[3];
        T fx[3], w[3][3], dw[3][3];
        stencil<T, double>(x, D.P.inv_dx, base, fx, w, dw);
        clamp_to_reach(D, base);
        double fxa[3] = {0, 0, 0};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                for (int l = 0; l < 3; ++l) {
                    int idx = node_index(D, base[0] + i, base[1] + j, base[2] + l);
                    double diff = (double)gm[idx] - (double)td[idx];
                    double sg = diff > 0 ? 1.0 : (diff < 0 ? -1.0 : 0.0);          // d|x|/dx with sgn(0) = 0
                    double ga = (w_density * sg + w_sdf * (double)ts[idx]) * (double)D.P.p_mass;
                    fxa[0] += ga * (double)(dw[i][0] * w[j][1] * w[l][2]);
                    fxa[1] += ga * (double)(w[i][0] * dw[j][1] * w[l][2]);
                    fxa[2] += ga * (double)(w[i][0] * w[j][1] * dw[l][2]);
                }
        double xa[3] = {fxa[0] * (double)D.P.inv_dx, fxa[1] * (double)D.P.inv_dx, fxa[2] * (double)D.P.inv_dx};
        for (int q = 0; q < D.nprim; ++q) {
            if (!D.prim[q].movable) continue;
            PrimT<T> pr = prim_at(D, q, f);
            double sd = prim_sdf(pr, x);
            if (!max_to_lhs(sd, 0.0, D.P.tie_first)) continue;          // max(sdf, 0): adjoint to sdf iff 0 < sdf
            double md = ls[LS_MIND + q];
            double coef;
            // hard contact, ti.atomic_min(min_dist, d) (loss.py:123-128): differentiated as an add by Taichi 0.7.x as far as
            // it is known (unverified) -- every particle gets min_dist's adjoint; plmpm_config.contact_min_adjoint
            // = 1 sends it to the particle(s) that attain the minimum instead (the mathematical derivative)
            if (!soft) { if (argmin && fmax(sd, 0.0) != md) continue; coef = w_contact * 2.0 * md; }
            else {
                double dn = ls[LS_DNORM + q];
                double den = 1.0 + sd * sd * 10000.0;
                double sw = 1.0 / den, dsw = -20000.0 * sd / (den * den);
                coef = w_contact * 2.0 * md * (sw + sd * dsw - md * dsw) / dn;
            }
            double pa[3] = {0, 0, 0}, ra[4] = {0, 0, 0, 0}, ga = 0.0;
            if (pr.shape == SHAPE_SPHERE) {                 // d sdf/dx = (x - c)/len ; d sdf/dc = -that
                double dvec[3] = {x[0] - pr.pos[0], x[1] - pr.pos[1], x[2] - pr.pos[2]};
                double L = len14(dvec[0], dvec[1], dvec[2]);
                for (int d = 0; d < 3; ++d) { double g = coef * dvec[d] / L; xa[d] += g; pa[d] = -g; }
            } else {                                        // sdf = sdf_local(inv_trans(x, pos, rot))
                double loc[3], iq[4], na0[3] = {0, 0, 0}, loca[3] = {0, 0, 0};
                inv_trans(x, pr.pos, pr.rot, loc, iq);
                shape_local_adj(pr.shape, pr.par, loc, coef, na0, loca, &ga, D.P.tie_first);       // minmax_tie applies to the shape SDFs' max / min here as in collide
                inv_trans_adj(x, pr.pos, pr.rot, iq, loca, pa, ra);
                for (int d = 0; d < 3; ++d) xa[d] -= pa[d];  // d/dx = -d/dpos
            }
            if (dl) {
                for (int d = 0; d < 3; ++d) det_add(&sdet[2 * (q * 8 + d)], &sdet[2 * (q * 8 + d) + 1], pa[d]);
                for (int d = 0; d < 4; ++d) det_add(&sdet[2 * (q * 8 + 3 + d)], &sdet[2 * (q * 8 + 3 + d) + 1], ra[d]);
                det_add(&sdet[2 * (q * 8 + 7)], &sdet[2 * (q * 8 + 7) + 1], ga);
            } else {
                for (int d = 0; d < 3; ++d) if (pa[d] != 0.0) atomicAdd(&sacc[q * 8 + d], pa[d]);
                for (int d = 0; d < 4; ++d) if (ra[d] != 0.0) atomicAdd(&sacc[q * 8 + 3 + d], ra[d]);
                if (ga != 0.0) atomicAdd(&sacc[q * 8 + 7], ga);
            }
        }
        T* A = D.adj[which];
        for (int d = 0; d < 3; ++d) A[d * D.Npad + p] += (T)xa[d];
    }
    __syncthreads();
    if (dl) {
        // the workgroup's integer sums go on into the global limbs (slot LS_COUNT + q * 8 + c); k_det_small_resolve adds
        // the totals into the pose adjoints
        if (threadIdx.x < D.nprim * 8 * 2 && sdet[threadIdx.x] != 0)
            atomicAdd(reinterpret_cast<unsigned long long*>(dl + 2 * LS_COUNT + threadIdx.x), (unsigned long long)sdet[threadIdx.x]);
    } else if (threadIdx.x < D.nprim * 8) {
        double v = sacc[threadIdx.x];
        int q = threadIdx.x / 8, c = threadIdx.x % 8;
        if (v != 0.0) {
            if (c < 3) atomicAdd(&D.ppos_a[((size_t)f * D.nprim + q) * 3 + c], v);
            else if (c < 7) atomicAdd(&D.prot_a[((size_t)f * D.nprim + q) * 4 + (c - 3)], v);
            else atomicAdd(&D.pgap_a[(size_t)f * D.nprim + q], v);
        }
    }
}
// deterministic mode: the integer limbs of the loss scalars and of k_loss_grad's pose adjoints -> their double targets
template <class T> __global__ void k_det_small_resolve(Dev<T> D, int f, long long* dl, double* ls) {
    const int t = threadIdx.x;
    if (t >= LS_COUNT + D.nprim * 8) return;
    const long long hi = dl[2 * t], lo = dl[2 * t + 1];
    if (!(hi | lo)) return;
    dl[2 * t] = 0; dl[2 * t + 1] = 0;
    const double v = det_value(hi, lo);
    if (t < LS_COUNT) { ls[t] += v; return; }
    const int q = (t - LS_COUNT) / 8, c = (t - LS_COUNT) % 8;
    if (c < 3) D.ppos_a[((size_t)f * D.nprim + q) * 3 + c] += v;
    else if (c < 7) D.prot_a[((size_t)f * D.nprim + q) * 4 + (c - 3)] += v;
    else D.pgap_a[(size_t)f * D.nprim + q] += v;
}
// target SDF sweep (loss.py:81-101), double, linear [i][j][k] layout
__global__ void k_sdf_sweep(int n, double dx, double inf, const double* dens, const double* sdf_c, const double* np_c,
                            double* sdf, double* npn, int* changed) {
    size_t I = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t G = (size_t)n * n * n;
    if (I >= G) return;
    int k = I % n, j = (I / n) % n, i = I / ((size_t)n * n);
    double gx = i * dx, gy = j * dx, gz = k * dx;
    double best = inf, bx = npn[3 * I], by = npn[3 * I + 1], bz = npn[3 * I + 2];
    if (dens[I] > 1e-4) { best = 0.0; bx = gx; by = gy; bz = gz; }
    else {
        for (int a = -3; a < 3; ++a)
            for (int b = -3; b < 3; ++b)
                for (int c = -3; c < 3; ++c) {
                    int vi = i + a, vj = j + b, vk = k + c;
                    if (vi < 0 || vj < 0 || vk < 0 || vi >= n || vj >= n || vk >= n) continue;
                    if (a == 0 && b == 0 && c == 0) continue;
                    size_t V = ((size_t)vi * n + vj) * n + vk;
                    if (sdf_c[V] < inf) {
                        double ex = gx - np_c[3 * V], ey = gy - np_c[3 * V + 1], ez = gz - np_c[3 * V + 2];
                        double dist = sqrt(ex * ex + ey * ey + ez * ez + 1e-8);
                        if (dist < best) { best = dist; bx = np_c[3 * V]; by = np_c[3 * V + 1]; bz = np_c[3 * V + 2]; }
                    }
                }
    }
    if (best != sdf_c[I] || bx != np_c[3 * I] || by != np_c[3 * I + 1] || bz != np_c[3 * I + 2]) *changed = 1;
    sdf[I] = best; npn[3 * I] = bx; npn[3 * I + 1] = by; npn[3 * I + 2] = bz;
}
// host grids are dense (n,n,n) [i][j][k]; the device holds the blocked window.  Upload: the window's part of the dense
// grid; download: the dense grid is zeroed first, the window's nodes written over it.
template <class T> __global__ void k_upload_grid(Dev<T> D, int n, const double* lin, T* blocked) {
    size_t I = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= (size_t)D.nbx * D.nby * D.nbz * 64) return;
    int nd[3];
    block_nodes(D, (int)(I >> 6), (int)(I & 63), nd);
    blocked[I] = (T)lin[((size_t)nd[0] * n + nd[1]) * n + nd[2]];
}
template <class T> __global__ void k_download_grid(Dev<T> D, int n, const T* blocked, double* lin) {
    size_t I = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= (size_t)D.nbx * D.nby * D.nbz * 64) return;
    int nd[3];
    block_nodes(D, (int)(I >> 6), (int)(I & 63), nd);
    lin[((size_t)nd[0] * n + nd[1]) * n + nd[2]] = (double)blocked[I];
}
template <class T> static int upload_grid_t(plmpm_sim* s, const double* lin_d, char* dst) {
    hipLaunchKernelGGL((k_upload_grid<T>), dim3((unsigned)((s->G + 255) / 256)), dim3(256), 0, s->stream, make_dev<T>(s), s->n, lin_d, (T*)dst);
    return 0;
}

template <class T> static int download_grid_t(plmpm_sim* s, const char* src, double* lin_d) {
    (void)hipMemsetAsync(lin_d, 0, s->Gfull * 8, s->stream);
    hipLaunchKernelGGL((k_download_grid<T>), dim3((unsigned)((s->G + 255) / 256)), dim3(256), 0, s->stream, make_dev<T>(s), s->n, (const T*)src, lin_d);
    return 0;
}

template <class T> static int loss_scatter_t(plmpm_sim* s, int f) {
    Dev<T> D = make_dev<T>(s, f);
    hipMemsetAsync(s->loss_gm, 0, s->G * s->tsz, s->stream);
    if (nblocks_particles(s, f) > 0) {
        if (s->det) {
            hipLaunchKernelGGL((k_grid_mass<T, true>), dim3(nblocks_particles(s, f)), dim3(kBlock), 0, s->stream, D, f, (T*)s->loss_gm);
            DET_RESOLVE(s, (T*)s->loss_gm, (T*)nullptr, (T*)nullptr, (T*)nullptr);
        } else
            hipLaunchKernelGGL((k_grid_mass<T>), dim3(nblocks_particles(s, f)), dim3(kBlock), 0, s->stream, D, f, (T*)s->loss_gm);
    }
    return 0;
}

// mode 0: hard min, 1: soft normaliser, 2: soft weighted sum (needs the global normaliser in lscal)
template <class T> static int loss_contact_pass_t(plmpm_sim* s, int f, int mode) {
    Dev<T> D = make_dev<T>(s, f);
    bool any = false;
    for (int p = 0; p < s->P; ++p) any |= s->prims[p].action_dim > 0;
    if (any) {
        hipLaunchKernelGGL((k_contact<T>), dim3(std::min(s->Npad / 256, 512)), dim3(256), 0, s->stream, D, f, mode, s->lscal, s->det_small);
        if (s->det) hipLaunchKernelGGL((k_det_small_resolve<T>), dim3(1), dim3(128), 0, s->stream, D, f, s->det_small, s->lscal);
    }
    return 0;
}
// loss scalars to their start values, on the device (no host buffer to keep alive, no synchronisation)
__global__ void k_ls_init(double* ls, int soft) {
    const int i = threadIdx.x;
    if (i < LS_COUNT) ls[i] = (!soft && i >= LS_MIND && i < LS_MIND + kMaxPrim) ? 100000.0 : 0.0;      // loss.py:189-191
}
static int loss_reset_scalars(plmpm_sim* s) {
    hipLaunchKernelGGL(k_ls_init, dim3(1), dim3(64), 0, s->stream, s->lscal, s->soft_contact ? 1 : 0);
    return 0;
}

template <class T> static int loss_reduce_t(plmpm_sim* s) {
    hipLaunchKernelGGL((k_loss_reduce<T>), dim3(256), dim3(256), 0, s->stream, s->G, s->nbw[0] * s->nbw[1], s->go[2], s->cfg.slab_z0, s->cfg.slab_z1,
                       (const T*)s->loss_gm, (const T*)s->loss_td, (const T*)s->loss_ts, s->lscal, s->det_small);
    if (s->det) hipLaunchKernelGGL((k_det_small_resolve<T>), dim3(1), dim3(128), 0, s->stream, make_dev<T>(s), 0, s->det_small, s->lscal);
    return 0;
}

template <class T> static int loss_grad_t(plmpm_sim* s, int f) {
    Dev<T> D = make_dev<T>(s, f);
    hipLaunchKernelGGL((k_loss_grad<T>), dim3(s->Npad / 256), dim3(256), 0, s->stream, D, f, f & 1, (const T*)s->loss_gm,
                       (const T*)s->loss_td, (const T*)s->loss_ts, s->lscal, s->w_sdf, s->w_density, s->w_contact, s->soft_contact, s->det_small,
                       s->cfg.contact_min_adjoint);
    if (s->det) hipLaunchKernelGGL((k_det_small_resolve<T>), dim3(1), dim3(128), 0, s->stream, D, f, s->det_small, s->lscal);
    return 0;
}

extern "C" {
int plmpm_loss_contact_scalars(plmpm_handle s, double* min_dist, double* dist_norm) {
    NEED_BOUND(s);
    double ls[LS_COUNT];
    HIPCHK(hipMemcpyAsync(ls, s->lscal, sizeof ls, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    for (int q = 0; q < s->P; ++q) {
        if (min_dist) min_dist[q] = ls[LS_MIND + q];
        if (dist_norm) dist_norm[q] = ls[LS_DNORM + q];
    }
    return 0;
}
// measured HBM roof of this device: float4 copy of `bytes` (read + write), best of `reps`; GB/s of bytes moved
// ---- loss -----------------------------------------------------------------------------------

int plmpm_loss_set_target(plmpm_handle s, const double* density) {
    NEED_BOUND(s);
    REQUIRE(density, "null density");
    const size_t G = s->Gfull;                   // the sweeps run on the dense n^3 grid; only the window's part is kept
    double *d_dens, *d_sdf[2], *d_np[2];
    int* d_changed;
    HIPCHK(hipMalloc(&d_dens, G * 8));
    for (int i = 0; i < 2; ++i) { HIPCHK(hipMalloc(&d_sdf[i], G * 8)); HIPCHK(hipMalloc(&d_np[i], G * 24)); }
    HIPCHK(hipMalloc(&d_changed, 4));
    HIPCHK(hipMemcpyAsync(d_dens, density, G * 8, hipMemcpyHostToDevice, s->stream));
    std::vector<double> inf(G, 1000.0);
    HIPCHK(hipMemcpyAsync(d_sdf[0], inf.data(), G * 8, hipMemcpyHostToDevice, s->stream));     // target_sdf_copy.fill(inf)
    HIPCHK(hipMemsetAsync(d_np[0], 0, G * 24, s->stream));
    HIPCHK(hipMemsetAsync(d_np[1], 0, G * 24, s->stream));
    int cur = 0, last = 0;
    for (int it = 0; it < 2 * s->n; ++it) {                                                  // loss.py:103-106
        HIPCHK(hipMemsetAsync(d_changed, 0, 4, s->stream));
        // nearest_point persists across sweeps where nothing improves: carry the previous field over
        HIPCHK(hipMemcpyAsync(d_np[1 - cur], d_np[cur], G * 24, hipMemcpyDeviceToDevice, s->stream));
        hipLaunchKernelGGL(k_sdf_sweep, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, s->stream, s->n, 1.0 / s->n, 1000.0,
                           d_dens, d_sdf[cur], d_np[cur], d_sdf[1 - cur], d_np[1 - cur], d_changed);
        int changed = 0;
        HIPCHK(hipMemcpyAsync(&changed, d_changed, 4, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        cur = 1 - cur;
        last = cur;
        if (!changed) break;
    }
    DISPATCH(s, upload_grid_t, s, d_dens, s->loss_td);
    DISPATCH(s, upload_grid_t, s, d_sdf[last], s->loss_ts);
    HIPCHK(hipStreamSynchronize(s->stream));
    hipFree(d_dens); hipFree(d_changed);
    for (int i = 0; i < 2; ++i) { hipFree(d_sdf[i]); hipFree(d_np[i]); }
    s->target_max = 0; s->target_sum = 0; s->target_outside = 0;
    const int n = s->n;
    for (size_t i = 0; i < G; ++i) {
        s->target_max = std::max(s->target_max, density[i]);
        s->target_sum += density[i];
        if (density[i] != 0.0) {
            // |grid_m - target| of an owned node outside the grid window is |0 - target|: a constant of the density loss
            const int k = (int)(i % n), j = (int)((i / n) % n), ii = (int)(i / ((size_t)n * n));
            const int nd[3] = {ii, j, k};
            bool inside = true;
            for (int d = 0; d < 3; ++d) inside &= nd[d] >= s->go[d] && nd[d] < s->go[d] + 4 * s->nbw[d];
            if (!inside && k >= s->cfg.slab_z0 && k < s->cfg.slab_z1) s->target_outside += std::fabs(density[i]);
        }
    }
    s->have_target = true;
    return 0;
}

int plmpm_loss_set_weights(plmpm_handle s, double sdf, double density, double contact, int soft_contact) {
    REQUIRE(s, "null handle");
    s->w_sdf = sdf; s->w_density = density; s->w_contact = contact; s->soft_contact = soft_contact;
    return 0;
}


int plmpm_loss_scatter(plmpm_handle s, int frame) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    DISPATCH(s, loss_scatter_t, s, frame);
    HIPCHK(hipGetLastError());
    return 0;
}

int plmpm_loss_partials(plmpm_handle s, int frame, int phase, double* out32) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(s->have_target, "loss: no target density set");
    REQUIRE(out32 && (phase == 0 || phase == 1), "bad arguments");
    if (phase == 0) {
        if (loss_reset_scalars(s)) return -1;
        DISPATCH(s, loss_reduce_t, s);
        DISPATCH(s, loss_contact_pass_t, s, frame, s->soft_contact ? 1 : 0);
    } else {
        REQUIRE(s->soft_contact, "phase 1 only exists for the soft contact loss");
        DISPATCH(s, loss_contact_pass_t, s, frame, 2);
    }
    HIPCHK(hipMemcpyAsync(out32, s->lscal, LS_COUNT * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (phase == 0) out32[LS_DENSITY] += s->target_outside;
    return 0;
}

int plmpm_loss_set_globals(plmpm_handle s, const double* in32) {
    NEED_BOUND(s);
    REQUIRE(in32, "null argument");
    HIPCHK(hipMemcpyAsync(s->lscal, in32, LS_COUNT * 8, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

int plmpm_loss_finish(plmpm_handle s, const double* ls, double* out6) {
    REQUIRE(s && ls && out6, "null argument");
    double contact = 0;
    for (int p = 0; p < s->P; ++p)
        if (s->prims[p].action_dim > 0) contact += ls[LS_MIND + p] * ls[LS_MIND + p];          // loss.py:137-140
    double ma = ls[LS_MAXGM], mb = s->target_max;
    double I = ls[LS_DOT] / ma / mb, U = ls[LS_SUMGM] / ma + s->target_sum / mb;                 // loss.py:252-254
    out6[0] = contact * s->w_contact + ls[LS_DENSITY] * s->w_density + ls[LS_SDF] * s->w_sdf;   // loss.py:158-162
    out6[1] = ls[LS_SDF]; out6[2] = ls[LS_DENSITY]; out6[3] = contact; out6[4] = I / (U - I); out6[5] = 0;
    return 0;
}

int plmpm_loss_backward_local(plmpm_handle s, int frame) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(s->have_target, "loss: no target density set");
    REQUIRE(s->adj_frame[frame & 1] == frame, "loss_backward: adjoint of frame %d is not resident", frame);
    if (s->adj_epoch[frame & 1] != s->frame_epoch[frame] &&
        plmpm_convert_adjoint(s, frame & 1, s->adj_epoch[frame & 1], s->frame_epoch[frame])) return -1;
    DISPATCH(s, loss_grad_t, s, frame);
    HIPCHK(hipGetLastError());
    return 0;
}

// single-rank composition of the phases above
static int loss_globals_single(plmpm_sim* s, int frame, double* ls) {
    if (plmpm_loss_scatter(s, frame)) return -1;
    if (plmpm_loss_partials(s, frame, 0, ls)) return -1;
    if (s->soft_contact) {
        if (plmpm_loss_set_globals(s, ls)) return -1;
        if (plmpm_loss_partials(s, frame, 1, ls)) return -1;
    }
    return 0;
}

int plmpm_loss_forward(plmpm_handle s, int frame, double* out6) {
    NEED_BOUND(s);
    REQUIRE(out6, "null output");
    double ls[LS_COUNT];
    if (loss_globals_single(s, frame, ls)) return -1;
    return plmpm_loss_finish(s, ls, out6);
}

int plmpm_loss_backward(plmpm_handle s, int frame) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(s->have_target, "loss: no target density set");
    // recompute grid_m and the contact scalars (loss.py:210-237) -- all of it stays on the device: the adjoint needs the
    // mass grid and the per-primitive contact scalars, not the density / sdf sums, and nothing of it on the host
    if (plmpm_loss_scatter(s, frame)) return -1;
    if (loss_reset_scalars(s)) return -1;
    DISPATCH(s, loss_contact_pass_t, s, frame, s->soft_contact ? 1 : 0);
    if (s->soft_contact) DISPATCH(s, loss_contact_pass_t, s, frame, 2);
    return plmpm_loss_backward_local(s, frame);
}

int plmpm_get_grid_mass(plmpm_handle s, int frame, double* out) {
    NEED_BOUND(s);
    NEED_FRAME(s, frame);
    REQUIRE(out, "null output");
    double* d_lin;
    HIPCHK(hipMalloc(&d_lin, s->Gfull * 8));
    DISPATCH(s, loss_scatter_t, s, frame);
    DISPATCH(s, download_grid_t, s, s->loss_gm, d_lin);
    HIPCHK(hipMemcpyAsync(out, d_lin, s->Gfull * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    hipFree(d_lin);
    return 0;
}
int plmpm_loss_get_target_sdf(plmpm_handle s, double* out) {
    NEED_BOUND(s);
    REQUIRE(out && s->have_target, "no target set");
    double* d_lin;
    HIPCHK(hipMalloc(&d_lin, s->Gfull * 8));
    DISPATCH(s, download_grid_t, s, s->loss_ts, d_lin);
    HIPCHK(hipMemcpyAsync(out, d_lin, s->Gfull * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    hipFree(d_lin);
    return 0;
}

}  // extern "C"
