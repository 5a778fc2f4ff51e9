// TEST INFRASTRUCTURE ONLY -- never loaded by plasticinelab_amd.
// Runs the *same* per-particle / per-node functions the HIP kernels call
// (plasticinelab_amd/csrc/mpm_math.h, mpm_grid.h) in plain serial loops on the
// host, so the hand-derived adjoints can be checked against the oracle in the
// CPU-only test tier.  The GPU kernels' indexing, tiling and atomics are covered
// by the -m gpu tests.
#include <vector>
#include <cstring>
#include "../../plasticinelab_amd/csrc/mpm_grid.h"

using namespace plb;

extern "C" {
struct emul_cfg {
    int n_grid, use_float, n_prim;
    double dt, p_vol, p_mass, gravity[3], ground_friction, svd_clamp, softness;
};
struct emul_prim {
    int shape, movable;
    double par[3], friction, pos[3], rot[4], pos1[3], rot1[4];
};
}

template <class T> static SimP<T> make_simp(const emul_cfg& c) {
    SimP<T> P;
    P.n = c.n_grid;
    double dx = 1.0 / c.n_grid;
    P.dx = (T)dx; P.inv_dx = (T)c.n_grid; P.dt = (T)c.dt; P.p_mass = (T)c.p_mass;
    P.kappa = (T)(-c.dt * c.p_vol * 4 * c.n_grid * (double)c.n_grid);
    for (int i = 0; i < 3; ++i) P.grav[i] = (T)(c.dt * c.gravity[i] * 30);
    P.x_hi = (T)(1.0 - 3 * dx);
    P.ground_friction = (T)c.ground_friction; P.svd_clamp = (T)c.svd_clamp; P.softness = (T)c.softness;
    P.tie_first = 0;
    return P;
}
template <class T> static std::vector<PrimT<T>> make_prims(const emul_cfg& c, const emul_prim* p) {
    std::vector<PrimT<T>> out(c.n_prim);
    for (int i = 0; i < c.n_prim; ++i) {
        out[i].shape = p[i].shape; out[i].movable = p[i].movable; out[i].friction = (T)p[i].friction;
        out[i].rb = prim_bounding_radius(p[i].shape, p[i].par);
        for (int k = 0; k < 3; ++k) { out[i].par[k] = p[i].par[k]; out[i].pos[k] = p[i].pos[k]; out[i].pos1[k] = p[i].pos1[k]; }
        for (int k = 0; k < 4; ++k) { out[i].rot[k] = p[i].rot[k]; out[i].rot1[k] = p[i].rot1[k]; }
    }
    return out;
}

template <class T> struct Grid {
    int n; std::vector<T> m, mv, vout;
    explicit Grid(int n_) : n(n_), m((size_t)n_ * n_ * n_, T(0)), mv((size_t)n_ * n_ * n_ * 3, T(0)), vout((size_t)n_ * n_ * n_ * 3, T(0)) {}
    size_t idx(int i, int j, int k) const { return ((size_t)i * n + j) * n + k; }
};

#ifndef EMUL_XFLOAT
typedef double XP;   // positions stay double on the fp32 path (as in the HIP kernels)
#define XT double
#else
#define XT T         // experiment: fp32 positions
#endif
template <class T> static void load_particle(int p, const double* x, const double* v, const double* C, const double* F,
                                             XT* xp, T* vp, T* Cp, T* Ep) {
    for (int d = 0; d < 3; ++d) { xp[d] = (XT)x[3 * p + d]; vp[d] = (T)v[3 * p + d]; }
    for (int d = 0; d < 9; ++d) { Cp[d] = (T)C[9 * p + d]; Ep[d] = (T)(F[9 * p + d] - ((d % 4 == 0) ? 1.0 : 0.0)); }
}

template <class T> static void forward_grid(const SimP<T>& P, const std::vector<PrimT<T>>& prims, int N,
                                            const double* x, const double* v, const double* C, const double* F,
                                            const double* mu, const double* lam, const double* ys, Grid<T>& g, double* F1) {
    for (int p = 0; p < N; ++p) {
        XT xp[3]; T vp[3], Cp[9], Ep[9], En[9];
        int base[3];
        load_particle<T>(p, x, v, C, F, xp, vp, Cp, Ep);
        p2g_particle<T, XT>(P, xp, vp, Cp, Ep, (T)mu[p], (T)lam[p], (T)ys[p], En, base,
                        [&](int i, int j, int l, T mass, const T* mom) {
                            size_t I = g.idx(base[0] + i, base[1] + j, base[2] + l);
                            g.m[I] += mass;
                            for (int a = 0; a < 3; ++a) g.mv[3 * I + a] += mom[a];
                        });
        if (F1) for (int d = 0; d < 9; ++d) F1[9 * p + d] = (double)En[d] + ((d % 4 == 0) ? 1.0 : 0.0);
    }
    int n = P.n;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) for (int k = 0; k < n; ++k) {
        size_t I = g.idx(i, j, k);
        int Iv[3] = {i, j, k};
        grid_node_fwd<T>(P, Iv, g.m[I], &g.mv[3 * I], (int)prims.size(), prims.data(), &g.vout[3 * I]);
    }
}

template <class T> static int substep_t(const emul_cfg& c, const emul_prim* pr, int N, const double* x, const double* v,
                                        const double* C, const double* F, const double* mu, const double* lam,
                                        const double* ys, double* x1, double* v1, double* C1, double* F1) {
    SimP<T> P = make_simp<T>(c);
    auto prims = make_prims<T>(c, pr);
    Grid<T> g(c.n_grid);
    forward_grid<T>(P, prims, N, x, v, C, F, mu, lam, ys, g, F1);
    for (int p = 0; p < N; ++p) {
        XT xp[3] = {(XT)x[3 * p], (XT)x[3 * p + 1], (XT)x[3 * p + 2]}, xn[3]; T vn[3], Cn[9];
        int base[3]; T fx[3], w[3][3];
        stencil<T, XT>(xp, P.inv_dx, base, fx, w, nullptr);
        g2p_particle<T, XT>(P, xp, xn, vn, Cn, [&](int i, int j, int l, T* gv) {
            size_t I = g.idx(base[0] + i, base[1] + j, base[2] + l);
            for (int a = 0; a < 3; ++a) gv[a] = g.vout[3 * I + a];
        });
        for (int d = 0; d < 3; ++d) { x1[3 * p + d] = xn[d]; v1[3 * p + d] = vn[d]; }
        for (int d = 0; d < 9; ++d) C1[9 * p + d] = Cn[d];
    }
    return 0;
}

template <class T> static int substep_grad_t(const emul_cfg& c, const emul_prim* pr, int N, const double* x, const double* v,
                                             const double* C, const double* F, const double* mu, const double* lam,
                                             const double* ys, const double* v1, const double* x1a, const double* v1a,
                                             const double* C1a, const double* F1a, double* xa, double* va, double* Ca,
                                             double* Fa, double* pose_adj) {
    SimP<T> P = make_simp<T>(c);
    auto prims = make_prims<T>(c, pr);
    int n = c.n_grid;
    Grid<T> g(n);
    forward_grid<T>(P, prims, N, x, v, C, F, mu, lam, ys, g, nullptr);
    std::vector<T> vout_a((size_t)n * n * n * 3, T(0)), in_a((size_t)n * n * n * 4, T(0));
    std::vector<T> xa_t((size_t)N * 3);
    // g2p.grad
    for (int p = 0; p < N; ++p) {
        XT xp[3]; T vn[3], xna[3], vna[3], Cna[9], xat[3];
        for (int d = 0; d < 3; ++d) { xp[d] = (XT)x[3 * p + d]; vn[d] = (T)v1[3 * p + d]; xna[d] = (T)x1a[3 * p + d]; vna[d] = (T)v1a[3 * p + d]; }
        for (int d = 0; d < 9; ++d) Cna[d] = (T)C1a[9 * p + d];
        int base[3]; T fx[3], w[3][3];
        stencil<T, XT>(xp, P.inv_dx, base, fx, w, nullptr);
        g2p_particle_grad<T, XT>(P, xp, vn, xna, vna, Cna, xat,
            [&](int i, int j, int l, T* gv) {
                size_t I = g.idx(base[0] + i, base[1] + j, base[2] + l);
                for (int a = 0; a < 3; ++a) gv[a] = g.vout[3 * I + a];
            },
            [&](int i, int j, int l, const T* ga) {
                size_t I = g.idx(base[0] + i, base[1] + j, base[2] + l);
                for (int a = 0; a < 3; ++a) vout_a[3 * I + a] += ga[a];
            });
        for (int d = 0; d < 3; ++d) xa_t[3 * p + d] = xat[d];
    }
    // grid_op.grad
    std::vector<double> padj((size_t)c.n_prim * 15, 0.0);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) for (int k = 0; k < n; ++k) {
        size_t I = g.idx(i, j, k);
        int Iv[3] = {i, j, k};
        T ma, mva[3], ma2, mva2[3];
        // as on the GPU: the velocity adjoint comes from the POSE = false pass (k_grid_op_grad), the pose adjoints
        // from the POSE = true pass that the spare workgroups of k_p2g_grad run on the blocks in contact
        grid_node_bwd<T, false>(P, Iv, g.m[I], &g.mv[3 * I], (int)prims.size(), prims.data(), &vout_a[3 * I], &ma, mva,
            [&](int, const PoseAdj<T>&, bool) {});
        grid_node_bwd<T, true>(P, Iv, g.m[I], &g.mv[3 * I], (int)prims.size(), prims.data(), &vout_a[3 * I], &ma2, mva2,
            [&](int p, const PoseAdj<T>& pa, bool hit) {
                if (!hit) return;
                double* o = &padj[(size_t)p * 15];
                for (int d = 0; d < 3; ++d) { o[d] += pa.pos[d]; o[7 + d] += pa.pos1[d]; }
                for (int d = 0; d < 4; ++d) { o[3 + d] += pa.rot[d]; o[10 + d] += pa.rot1[d]; }
                o[14] += pa.gap;
            });
        in_a[4 * I] = ma; in_a[4 * I + 1] = mva[0]; in_a[4 * I + 2] = mva[1]; in_a[4 * I + 3] = mva[2];
    }
    if (pose_adj) memcpy(pose_adj, padj.data(), padj.size() * sizeof(double));
    // p2g.grad (+ svd_grad + compute_F_tmp.grad)
    for (int p = 0; p < N; ++p) {
        XT xp[3]; T vp[3], Cp[9], Ep[9], Ena[9], xat[3], vat[3], Cat[9], Eat[9];
        load_particle<T>(p, x, v, C, F, xp, vp, Cp, Ep);
        for (int d = 0; d < 9; ++d) Ena[d] = (T)F1a[9 * p + d];
        for (int d = 0; d < 3; ++d) xat[d] = xa_t[3 * p + d];
        int base[3]; T fx[3], w[3][3];
        stencil<T, XT>(xp, P.inv_dx, base, fx, w, nullptr);
        p2g_particle_grad<T, XT>(P, xp, vp, Cp, Ep, (T)mu[p], (T)lam[p], (T)ys[p], Ena, xat, vat, Cat, Eat,
            [&](int i, int j, int l, T* gg) {
                size_t I = g.idx(base[0] + i, base[1] + j, base[2] + l);
                for (int a = 0; a < 4; ++a) gg[a] = in_a[4 * I + a];
            });
        for (int d = 0; d < 3; ++d) { xa[3 * p + d] = xat[d]; va[3 * p + d] = vat[d]; }
        for (int d = 0; d < 9; ++d) { Ca[9 * p + d] = Cat[d]; Fa[9 * p + d] = Eat[d]; }
    }
    return 0;
}

extern "C" {
int emul_substep(const emul_cfg* c, const emul_prim* pr, int N, const double* x, const double* v, const double* C,
                 const double* F, const double* mu, const double* lam, const double* ys, double* x1, double* v1,
                 double* C1, double* F1) {
    return c->use_float ? substep_t<float>(*c, pr, N, x, v, C, F, mu, lam, ys, x1, v1, C1, F1)
                        : substep_t<double>(*c, pr, N, x, v, C, F, mu, lam, ys, x1, v1, C1, F1);
}
int emul_substep_grad(const emul_cfg* c, const emul_prim* pr, int N, const double* x, const double* v, const double* C,
                      const double* F, const double* mu, const double* lam, const double* ys, const double* v1,
                      const double* x1a, const double* v1a, const double* C1a, const double* F1a, double* xa, double* va,
                      double* Ca, double* Fa, double* pose_adj) {
    return c->use_float
        ? substep_grad_t<float>(*c, pr, N, x, v, C, F, mu, lam, ys, v1, x1a, v1a, C1a, F1a, xa, va, Ca, Fa, pose_adj)
        : substep_grad_t<double>(*c, pr, N, x, v, C, F, mu, lam, ys, v1, x1a, v1a, C1a, F1a, xa, va, Ca, Fa, pose_adj);
}
void emul_fk_fwd(const double* pos, const double* rot, const double* v, const double* w, const double* lo,
                 const double* hi, double* pos1, double* rot1) { fk_fwd_d(pos, rot, v, w, lo, hi, pos1, rot1); }
void emul_fk_rollingpin_fwd(const double* pos, const double* rot, const double* v, const double* lo, const double* hi,
                            double* pos1, double* rot1) { fk_rollingpin_fwd_d(pos, rot, v, lo, hi, pos1, rot1); }
void emul_fk_rollingpin_bwd(const double* pos, const double* rot, const double* v, const double* lo, const double* hi,
                            const double* pos1_a, const double* rot1_a, double* pos_a, double* rot_a, double* v_a) {
    fk_rollingpin_bwd_d(pos, rot, v, lo, hi, pos1_a, rot1_a, pos_a, rot_a, v_a);
}
void emul_fk_chopsticks_fwd(const double* pos, const double* rot, const double* v, const double* w, double gap,
                            double gap_vel, double min_gap, const double* lo, const double* hi, double* pos1,
                            double* rot1, double* gap1) {
    fk_chopsticks_fwd_d(pos, rot, v, w, gap, gap_vel, min_gap, lo, hi, pos1, rot1, gap1);
}
void emul_fk_chopsticks_bwd(const double* pos, const double* rot, const double* v, const double* w, double gap,
                            double gap_vel, double min_gap, const double* lo, const double* hi, const double* pos1_a,
                            const double* rot1_a, double gap1_a, double* pos_a, double* rot_a, double* gap_a,
                            double* v_a, double* w_a, double* gap_vel_a) {
    fk_chopsticks_bwd_d(pos, rot, v, w, gap, gap_vel, min_gap, lo, hi, pos1_a, rot1_a, gap1_a, pos_a, rot_a, gap_a,
                        v_a, w_a, gap_vel_a);
}
void emul_fk_bwd(const double* pos, const double* rot, const double* v, const double* w, const double* lo,
                 const double* hi, const double* pos1_a, const double* rot1_a, double* pos_a, double* rot_a,
                 double* v_a, double* w_a) { fk_bwd_d(pos, rot, v, w, lo, hi, pos1_a, rot1_a, pos_a, rot_a, v_a, w_a); }
}

// constitutive block alone: the wave-uniform elastic fast path of mpm_math.h (here: one particle = one "wave") against the
// Jacobi path on the same inputs.  allow_fast = 0: Jacobi path only; took_fast[p] reports the path taken.
template <class T> static void constitutive_t(int allow_fast, int n, const double* Et_, const double* mu, const double* lam, const double* ys,
                                              double clamp, const double* GS_, const double* GF_, double* stress_, double* En_, double* Fta_, int* took_fast) {
    for (int p = 0; p < n; ++p) {
        T Et[9], GS[9], GF[9], stress[9], En[9], Fta[9];
        for (int i = 0; i < 9; ++i) { Et[i] = (T)Et_[9 * p + i]; GS[i] = (T)GS_[9 * p + i]; GF[i] = (T)GF_[9 * p + i]; }
        Elastic<T> el;
        const bool fast = allow_fast && elastic_try(Et, (T)mu[p], (T)ys[p], (T)clamp, true, el);
        if (fast) {
            for (int i = 0; i < 9; ++i) En[i] = Et[i];
            elastic_stress(Et, el, (T)mu[p], (T)lam[p], stress);
            elastic_vjp(Et, el, (T)mu[p], (T)lam[p], GS, GF, Fta);
        } else {
            Consti<T> k;
            constitutive_fwd(Et, (T)mu[p], (T)lam[p], (T)ys[p], k, En, stress, 0);
            constitutive_vjp(k, (T)mu[p], (T)lam[p], (T)clamp, GS, GF, Fta);
        }
        took_fast[p] = fast ? 1 : 0;
        for (int i = 0; i < 9; ++i) { stress_[9 * p + i] = stress[i]; En_[9 * p + i] = En[i]; Fta_[9 * p + i] = Fta[i]; }
    }
}
extern "C" void emul_constitutive(int use_float, int allow_fast, int n, const double* Et, const double* mu, const double* lam, const double* ys,
                                  double clamp, const double* GS, const double* GF, double* stress, double* En, double* Fta, int* took_fast) {
    if (use_float) constitutive_t<float>(allow_fast, n, Et, mu, lam, ys, clamp, GS, GF, stress, En, Fta, took_fast);
    else constitutive_t<double>(allow_fast, n, Et, mu, lam, ys, clamp, GS, GF, stress, En, Fta, took_fast);
}


// The packed-pair form of the p2g.grad gather (mpm_math.h: p2g_gather_grad_pk, the -DPLB_PK_GATHER=1 build of the fp32 engine) against
// the plain form, on n random stencils: positions x[n][3] in (0.1, 0.9), nodal adjoints g[n][27][4] = {grid_m.grad, grid_v_in.grad[3]}.
// Returns the largest difference of the 51 gathered sums relative to the largest sum of its particle.
extern "C" double emul_gather_pk_check(int n_grid, double p_mass, int n, const double* x, const double* g) {
    SimP<float> P{};
    P.n = n_grid; P.dx = 1.0f / n_grid; P.inv_dx = (float)n_grid; P.p_mass = (float)p_mass;
    double worst = 0.0;
    for (int s = 0; s < n; ++s) {
        double xp[3] = {x[3 * s], x[3 * s + 1], x[3 * s + 2]};
        const double* gs = g + (size_t)s * 108;
        P2GGather<float> A, B;
        p2g_gather_grad<float, double>(P, xp, A, [&](int i, int j, int l, float* q) {
            for (int a = 0; a < 4; ++a) q[a] = (float)gs[4 * (9 * i + 3 * j + l) + a];
        });
        p2g_gather_grad_pk<double>(P, xp, B, [&](int i, int j, int l, plb_f2& axy, plb_f2& azw) {
            const double* q = gs + 4 * (9 * i + 3 * j + l);
            axy = pk2((float)q[1], (float)q[2]); azw = pk2((float)q[3], (float)q[0]);
        });
        const float* a = reinterpret_cast<const float*>(&A);
        const float* b = reinterpret_cast<const float*>(&B);
        double scale = 0.0, diff = 0.0;
        for (int k = 0; k < 51; ++k) { scale = std::max(scale, (double)std::fabs(a[k])); diff = std::max(diff, (double)std::fabs(a[k] - b[k])); }
        worst = std::max(worst, diff / std::max(scale, 1e-30));          // (51 sums of different kinds: the largest sets the scale)
    }
    return worst;
}

// g2p_particle_pk against g2p_particle (fp32): n stencils, grid velocities gv[n][27][3]; largest difference of (v', C', x') relative
// to the scale of the stencil's input.
extern "C" double emul_g2p_pk_check(int n_grid, double dt, int n, const double* x, const double* gv) {
    SimP<float> P{};
    P.n = n_grid; P.dx = 1.0f / n_grid; P.inv_dx = (float)n_grid; P.dt = (float)dt;
    double worst = 0.0;
    for (int s = 0; s < n; ++s) {
        double xp[3] = {x[3 * s], x[3 * s + 1], x[3 * s + 2]}, xa[3], xb[3];
        const double* gs = gv + (size_t)s * 81;
        float va[3], Ca[9], vb[3], Cb[9];
        g2p_particle<float, double>(P, xp, xa, va, Ca, [&](int i, int j, int l, float* q) {
            for (int a = 0; a < 3; ++a) q[a] = (float)gs[3 * (9 * i + 3 * j + l) + a];
        });
        g2p_particle_pk<double>(P, xp, xb, vb, Cb, [&](int i, int j, int l, plb_f2& axy, plb_f2& azw) {
            const double* q = gs + 3 * (9 * i + 3 * j + l);
            axy = pk2((float)q[0], (float)q[1]); azw = pk2((float)q[2], 0.f);
        });
        // scales of the INPUT (the weights sum to one, |z| <= 1.5 cells): sums of random-sign terms may cancel to nothing
        double gmax = 0, diff = 0;
        for (int k = 0; k < 81; ++k) gmax = std::max(gmax, std::fabs(gs[k]));
        const double scale_v = gmax, scale_c = 4.0 * n_grid * gmax;
        for (int k = 0; k < 3; ++k) diff = std::max(diff, std::fabs((double)va[k] - vb[k]) / std::max(scale_v, 1e-30));
        for (int k = 0; k < 9; ++k) diff = std::max(diff, std::fabs((double)Ca[k] - Cb[k]) / std::max(scale_c, 1e-30));
        for (int k = 0; k < 3; ++k) diff = std::max(diff, std::fabs(xa[k] - xb[k]));
        worst = std::max(worst, diff);
    }
    return worst;
}

// svd_finish's rebuild of the weakest column of U for nearly singular F (sig_min < 1e-3), as it was written before round 6 -- a loop
// over the columns with a branch per column -- kept HERE as the reference the straight-line form in mpm_math.h is checked against
// (the two must agree exactly: same operations on the same values).
template <class T> static void svd_finish_column_loop(const T* Et, Svd3<T>& r) {
    const T* V = r.V;
    for (int i = 0; i < 3; ++i) {
        T sg = t_fsqrt(t_max(T(1) + r.lam[i], T(0)));
        r.sig[i] = sg;
        r.s[i] = r.lam[i] * t_rcp(T(1) + sg);
    }
    for (int i = 0; i < 3; ++i) {
        T inv = t_rcp(t_max(r.sig[i], T(1e-30)));
        for (int k = 0; k < 3; ++k)
            r.U[3 * k + i] = (V[3 * k + i] + Et[3 * k] * V[i] + Et[3 * k + 1] * V[3 + i] + Et[3 * k + 2] * V[6 + i]) * inv;
    }
    T smin = t_min(r.sig[0], t_min(r.sig[1], r.sig[2]));
    bool todo = smin < T(1e-3);
    if (todo) {
        T Fm[9];
        for (int i = 0; i < 9; ++i) Fm[i] = Et[i];
        Fm[0] += T(1); Fm[4] += T(1); Fm[8] += T(1);
        T sgn = det3(Fm) < T(0) ? T(-1) : T(1);
        for (int k = 0; k < 3; ++k) {
            const bool hit = todo && (r.sig[k] == smin);
            if (hit) {
                const int a = (k + 1) % 3, b = (k + 2) % 3;
                T ua[3] = {r.U[a], r.U[3 + a], r.U[6 + a]}, ub[3] = {r.U[b], r.U[3 + b], r.U[6 + b]}, uc[3];
                cross3(ua, ub, uc);
                T nrm = t_sqrt(dot3(uc, uc));
                T sc = nrm > T(0) ? sgn / nrm : T(0);
                r.U[k] = uc[0] * sc; r.U[3 + k] = uc[1] * sc; r.U[6 + k] = uc[2] * sc;
                todo = false;
            }
        }
    }
}
template <class T> static double svd_singular_check_t(int n, const double* Et, int* rebuilt) {
    double worst = 0.0;
    *rebuilt = 0;
    for (int s = 0; s < n; ++s) {
        T e[9];
        for (int i = 0; i < 9; ++i) e[i] = (T)Et[9 * s + i];
        Svd3<T> a, b;
        svd_jacobi(e, a.lam, a.V);
        b = a;
        svd_finish(e, a);
        svd_finish_column_loop(e, b);
        if (std::min(a.sig[0], std::min(a.sig[1], a.sig[2])) < T(1e-3)) ++*rebuilt;
        for (int i = 0; i < 9; ++i) worst = std::max(worst, (double)std::fabs(a.U[i] - b.U[i]));
        for (int i = 0; i < 3; ++i) worst = std::max(worst, (double)std::fabs(a.sig[i] - b.sig[i]) + (double)std::fabs(a.s[i] - b.s[i]));
    }
    return worst;
}
extern "C" double emul_svd_singular_check(int use_float, int n, const double* Et, int* rebuilt) {
    return use_float ? svd_singular_check_t<float>(n, Et, rebuilt) : svd_singular_check_t<double>(n, Et, rebuilt);
}
