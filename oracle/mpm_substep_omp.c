/* C / OpenMP float64 restatement of ONE MPM substep and its reverse -- the CPU baseline of bench.py.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY: nothing under plasticinelab_amd/ may load this; bench.py's cpu_baseline leg
 * times it, tests/test_oracle_omp.py checks it against oracle/plb_oracle.py (torch autograd) on small cases.
 *
 * PARITY UNPINNED against Taichi (see oracle/plb_oracle.py): the reference's CPU path is the Taichi CPU backend of
 * /root/reference/plb/engine/mpm_simulator.py, which cannot be installed here.  This file follows the same source,
 * kernel by kernel, in the reference's own data layout (AoS particle arrays, dense n^3 grids, a dense clear_grid
 * sweep, atomic scatters) and the reverse pass in the reference's own schedule (substep_grad, :260-278: recompute
 * clear_grid / compute_F_tmp / svd / p2g / grid_op, then g2p.grad, grid_op.grad, p2g.grad, svd_grad,
 * compute_F_tmp.grad) with hand-written statement-level adjoints where Taichi generates them:
 *   compute_F_tmp  :82-85       svd :87-90 (Jacobi; ti.svd itself is third-party)     backward_svd :97-115 (+ clamp :143-151)
 *   compute_von_mises :124-141  p2g :157-184       grid_op :189-221      g2p :223-242
 *   Primitive.collide primive_base.py:91-115 with Sphere.sdf / normal primitives.py:22-28 and collider_v :82-89
 * Taichi autodiff rules assumed (unverified): max(a,b) sends the adjoint to a iff b < a, min(a,b) to a iff
 * a < b, cast(int) has zero gradient, `if` differentiates the taken side only.
 * Scope: Sphere primitives whose rotation stays the identity (action dim 3: every BASELINE workload) -- collider_v then
 * reduces to (pos[f+1] - pos[f]) / dt and only position adjoints exist.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXP 8

typedef struct {
    int n, n_particles, n_prim;
    double dx, inv_dx, dt, p_vol, p_mass;
    double gravity[3];
    double ground_friction, softness;
    double radius[MAXP], friction[MAXP];
} plb_omp_cfg;

/* ------------------------------------------------------------------ small 3x3 helpers (row major) */
static void mm(const double* a, const double* b, double* c) {             /* c = a b */
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += a[3 * i + k] * b[3 * k + j]; c[3 * i + j] = s; }
}
static void mm_nt(const double* a, const double* b, double* c) {          /* c = a b^T */
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += a[3 * i + k] * b[3 * j + k]; c[3 * i + j] = s; }
}
static void mm_tn(const double* a, const double* b, double* c) {          /* c = a^T b */
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += a[3 * k + i] * b[3 * k + j]; c[3 * i + j] = s; }
}
static double det3(const double* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
static void cofactor(const double* m, double* c) {                        /* dJ/dF */
    c[0] = m[4] * m[8] - m[5] * m[7]; c[1] = m[5] * m[6] - m[3] * m[8]; c[2] = m[3] * m[7] - m[4] * m[6];
    c[3] = m[2] * m[7] - m[1] * m[8]; c[4] = m[0] * m[8] - m[2] * m[6]; c[5] = m[1] * m[6] - m[0] * m[7];
    c[6] = m[1] * m[5] - m[2] * m[4]; c[7] = m[2] * m[3] - m[0] * m[5]; c[8] = m[0] * m[4] - m[1] * m[3];
}

/* svd (mpm_simulator.py:87-90): F = U diag(sig) V^T by cyclic Jacobi on F^T F; V a rotation, det U = sign det F */
static void svd3(const double* F, double* U, double* sig, double* V) {
    double A[9];
    mm_tn(F, F, A);
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 40; ++sweep) {
        double off = fabs(A[1]) + fabs(A[2]) + fabs(A[5]);
        if (off < 1e-300 || off < 1e-17 * (fabs(A[0]) + fabs(A[4]) + fabs(A[8]))) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double apq = A[3 * p + q];
                if (apq == 0.0) continue;
                double theta = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {             /* A <- A J */
                    double akp = A[3 * k + p], akq = A[3 * k + q];
                    A[3 * k + p] = c * akp - s * akq; A[3 * k + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {             /* A <- J^T A */
                    double apk = A[3 * p + k], aqk = A[3 * q + k];
                    A[3 * p + k] = c * apk - s * aqk; A[3 * q + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    double vkp = V[3 * k + p], vkq = V[3 * k + q];
                    V[3 * k + p] = c * vkp - s * vkq; V[3 * k + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 3; ++i) sig[i] = sqrt(fmax(A[4 * i], 0.0));
    double FV[9];
    mm(F, V, FV);
    for (int i = 0; i < 3; ++i) {
        double inv = sig[i] > 1e-300 ? 1.0 / sig[i] : 0.0;
        for (int k = 0; k < 3; ++k) U[3 * k + i] = FV[3 * k + i] * inv;
    }
}

static double clampsvd(double a) { return a >= 0 ? fmax(a, 1e-6) : fmin(a, -1e-6); }       /* :143-151 */

/* backward_svd (:97-115): gu, gv 3x3, gs the diagonal adjoint; returns the adjoint of F_tmp */
static void backward_svd(const double* gu, const double* gs, const double* gv, const double* u, const double* sig, const double* v, double* out) {
    double S[9] = {sig[0], 0, 0, 0, sig[1], 0, 0, 0, sig[2]}, GS[9] = {gs[0], 0, 0, 0, gs[1], 0, 0, 0, gs[2]};
    double t1[9], t2[9], t3[9], sigma_term[9];
    mm(u, GS, t1); mm_nt(t1, v, sigma_term);
    double s2[3] = {sig[0] * sig[0], sig[1] * sig[1], sig[2] * sig[2]}, Fm[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Fm[3 * i + j] = i == j ? 0.0 : 1.0 / clampsvd(s2[j] - s2[i]);
    /* u_term = u ((F * (u^T gu - gu^T u)) sig) v^T */
    double a[9], b[9];
    mm_tn(u, gu, a); mm_tn(gu, u, b);
    for (int i = 0; i < 9; ++i) t1[i] = Fm[i] * (a[i] - b[i]);
    mm(t1, S, t2); mm(u, t2, t3);
    double u_term[9];
    mm_nt(t3, v, u_term);
    /* v_term = u (sig ((F * (v^T gv - gv^T v)) v^T)) */
    mm_tn(v, gv, a); mm_tn(gv, v, b);
    for (int i = 0; i < 9; ++i) t1[i] = Fm[i] * (a[i] - b[i]);
    mm_nt(t1, v, t2); mm(S, t2, t3);
    double v_term[9];
    mm(u, t3, v_term);
    for (int i = 0; i < 9; ++i) out[i] = u_term[i] + v_term[i] + sigma_term[i];
}

/* quadratic B-spline stencil (:160-163) */
static void stencil(const plb_omp_cfg* c, const double* x, int* base, double* fx, double w[3][3], double dw[3][3]) {
    for (int d = 0; d < 3; ++d) {
        double xs = x[d] * c->inv_dx;
        base[d] = (int)(xs - 0.5);
        double f = xs - (double)base[d];
        fx[d] = f;
        w[0][d] = 0.5 * (1.5 - f) * (1.5 - f); w[1][d] = 0.75 - (f - 1.0) * (f - 1.0); w[2][d] = 0.5 * (f - 0.5) * (f - 0.5);
        dw[0][d] = -(1.5 - f); dw[1][d] = -2.0 * (f - 1.0); dw[2][d] = f - 0.5;
    }
}

/* per-particle constitutive state shared by p2g forward and backward */
typedef struct {
    double Ft[9], U[9], sig[3], V[9];
    int yielded;
    double s[3], e[3], ehat[3], nrm, k, sp[3];   /* clamped sig, log, deviator, its norm, delta_gamma / norm, exp of the projected strain */
    double newF[9], J, r[9], stress[9], affine[9];
} pstate;

static void particle_fwd(const plb_omp_cfg* c, const double* C, const double* F, double mu, double lam, double ys, pstate* q) {
    double ICdt[9];
    for (int i = 0; i < 9; ++i) ICdt[i] = c->dt * C[i] + ((i % 4 == 0) ? 1.0 : 0.0);
    mm(ICdt, F, q->Ft);                                               /* compute_F_tmp */
    svd3(q->Ft, q->U, q->sig, q->V);                                  /* svd */
    double mean = 0;                                                  /* compute_von_mises */
    for (int i = 0; i < 3; ++i) { q->s[i] = 0.05 < q->sig[i] ? q->sig[i] : 0.05; q->e[i] = log(q->s[i]); mean += q->e[i]; }
    mean /= 3.0;
    double n2 = 1e-8;
    for (int i = 0; i < 3; ++i) { q->ehat[i] = q->e[i] - mean; n2 += q->ehat[i] * q->ehat[i]; }
    q->nrm = sqrt(n2);
    double dg = q->nrm - ys / (2.0 * mu);
    q->yielded = dg > 0;
    if (q->yielded) {
        q->k = dg / q->nrm;
        double USp[9];
        for (int i = 0; i < 3; ++i) q->sp[i] = exp(q->e[i] - q->k * q->ehat[i]);
        for (int r = 0; r < 3; ++r) for (int i = 0; i < 3; ++i) USp[3 * r + i] = q->U[3 * r + i] * q->sp[i];
        mm_nt(USp, q->V, q->newF);
    } else
        memcpy(q->newF, q->Ft, sizeof q->newF);
    q->J = det3(q->newF);                                             /* p2g :168-173 */
    mm_nt(q->U, q->V, q->r);
    double A[9], AFt[9];
    for (int i = 0; i < 9; ++i) A[i] = q->newF[i] - q->r[i];
    mm_nt(A, q->newF, AFt);
    const double kappa = -c->dt * c->p_vol * 4.0 * c->inv_dx * c->inv_dx;
    for (int i = 0; i < 9; ++i) {
        q->stress[i] = 2.0 * mu * AFt[i] + ((i % 4 == 0) ? lam * q->J * (q->J - 1.0) : 0.0);
        q->affine[i] = kappa * q->stress[i] + c->p_mass * C[i];
    }
}

static inline size_t node(const plb_omp_cfg* c, int i, int j, int k) { return ((size_t)i * c->n + j) * c->n + k; }

/* clear_grid + compute_F_tmp + svd + p2g (:60-70, :82-90, :157-184); F1 may be NULL (recompute in substep_grad) */
static void p2g_all(const plb_omp_cfg* c, const double* x, const double* v, const double* C, const double* F, const double* mu, const double* lam,
                    const double* ys, double* gm, double* gv, double* F1) {
    const size_t G = (size_t)c->n * c->n * c->n;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < G; ++i) { gm[i] = 0; gv[3 * i] = gv[3 * i + 1] = gv[3 * i + 2] = 0; }
#pragma omp parallel for schedule(static)
    for (int p = 0; p < c->n_particles; ++p) {
        pstate q;
        particle_fwd(c, C + 9 * (size_t)p, F + 9 * (size_t)p, mu[p], lam[p], ys[p], &q);
        if (F1) memcpy(F1 + 9 * (size_t)p, q.newF, sizeof q.newF);
        int base[3];
        double fx[3], w[3][3], dw[3][3];
        stencil(c, x + 3 * (size_t)p, base, fx, w, dw);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {
            const double dpos[3] = {(i - fx[0]) * c->dx, (j - fx[1]) * c->dx, (k - fx[2]) * c->dx};
            const double weight = w[i][0] * w[j][1] * w[k][2];
            const size_t I = node(c, base[0] + i, base[1] + j, base[2] + k);
            for (int a = 0; a < 3; ++a) {
                const double val = weight * (c->p_mass * v[3 * (size_t)p + a] + q.affine[3 * a] * dpos[0] + q.affine[3 * a + 1] * dpos[1] + q.affine[3 * a + 2] * dpos[2]);
#pragma omp atomic
                gv[3 * I + a] += val;
            }
#pragma omp atomic
            gm[I] += weight * c->p_mass;
        }
    }
}

/* Sphere collide forward (primive_base.py:91-115, primitives.py:22-28); returns 1 when the branch is taken */
typedef struct { double d[3], L, dist, e, infl, D[3], cv[3], iv[3], nc, gvt[3], nrm, t, flag; } cstate;
static int collide_fwd(const plb_omp_cfg* c, int q, const double* pos, const double* pos1, const double* gp, const double* vin, double* vout, cstate* s) {
    for (int a = 0; a < 3; ++a) s->d[a] = gp[a] - pos[a];
    s->L = sqrt(s->d[0] * s->d[0] + s->d[1] * s->d[1] + s->d[2] * s->d[2] + 1e-14);
    s->dist = s->L - c->radius[q];
    s->e = exp(-s->dist * c->softness);
    s->infl = s->e < 1.0 ? s->e : 1.0;
    if (!((c->softness > 0 && s->infl > 0.1) || s->dist <= 0)) { for (int a = 0; a < 3; ++a) vout[a] = vin[a]; return 0; }
    for (int a = 0; a < 3; ++a) { s->D[a] = s->d[a] / s->L; s->cv[a] = (pos1[a] - pos[a]) / c->dt; s->iv[a] = vin[a] - s->cv[a]; }
    s->nc = s->iv[0] * s->D[0] + s->iv[1] * s->D[1] + s->iv[2] * s->D[2];
    const double mn = s->nc < 0.0 ? s->nc : 0.0;
    for (int a = 0; a < 3; ++a) s->gvt[a] = s->iv[a] - mn * s->D[a];
    const double g2 = s->gvt[0] * s->gvt[0] + s->gvt[1] * s->gvt[1] + s->gvt[2] * s->gvt[2];
    s->nrm = sqrt(g2 + 1e-8);
    s->t = s->nrm + s->nc * c->friction[q];
    const double m = s->t < 0.0 ? 0.0 : s->t;                       /* max(0, t): adjoint to t iff !(t < 0) */
    s->flag = (s->nc < 0 && sqrt(g2) > 1e-30) ? 1.0 : 0.0;
    for (int a = 0; a < 3; ++a) {
        const double fr = s->gvt[a] / s->nrm * m;
        const double g = fr * s->flag + s->gvt[a] * (1.0 - s->flag);
        vout[a] = s->cv[a] + s->iv[a] * (1.0 - s->infl) + g * s->infl;
    }
    return 1;
}
/* adjoint of the taken branch: va_out -> va_in, pos / pos1 adjoints accumulated into pa / p1a */
static void collide_bwd(const plb_omp_cfg* c, int q, const cstate* s, const double* va_out, double* va_in, double* pa, double* p1a) {
    const double m = s->t < 0.0 ? 0.0 : s->t;
    double cva[3], iva[3], ga[3], gvta[3] = {0, 0, 0}, Da[3] = {0, 0, 0}, infla = 0, nca = 0, nrma = 0;
    for (int a = 0; a < 3; ++a) {
        const double fr = s->gvt[a] / s->nrm * m;
        const double g = fr * s->flag + s->gvt[a] * (1.0 - s->flag);
        cva[a] = va_out[a];
        iva[a] = (1.0 - s->infl) * va_out[a];
        infla += va_out[a] * (g - s->iv[a]);
        ga[a] = s->infl * va_out[a];
    }
    if (s->flag != 0.0) {                     /* g = gvt / nrm * m */
        double ma = 0, ua[3];
        for (int a = 0; a < 3; ++a) { ma += ga[a] * s->gvt[a] / s->nrm; ua[a] = m * ga[a]; }
        const double ta = (s->t < 0.0) ? 0.0 : ma;
        nrma += ta; nca += c->friction[q] * ta;
        double dotug = 0;
        for (int a = 0; a < 3; ++a) { gvta[a] += ua[a] / s->nrm; dotug += ua[a] * s->gvt[a]; }
        nrma += -dotug / (s->nrm * s->nrm);
        for (int a = 0; a < 3; ++a) gvta[a] += nrma * s->gvt[a] / s->nrm;
    } else
        for (int a = 0; a < 3; ++a) gvta[a] += ga[a];
    /* gvt = iv - min(nc, 0) D */
    const double mn = s->nc < 0.0 ? s->nc : 0.0;
    double mna = 0;
    for (int a = 0; a < 3; ++a) { iva[a] += gvta[a]; mna -= gvta[a] * s->D[a]; Da[a] -= mn * gvta[a]; }
    if (s->nc < 0.0) nca += mna;
    for (int a = 0; a < 3; ++a) { iva[a] += nca * s->D[a]; Da[a] += nca * s->iv[a]; }
    for (int a = 0; a < 3; ++a) { va_in[a] = iva[a]; cva[a] -= iva[a]; }
    /* influence = min(exp(-dist softness), 1) */
    const double ea = (s->e < 1.0) ? infla : 0.0;
    const double dista = -c->softness * s->e * ea;
    /* D = d / L, dist = L - radius, L = sqrt(d.d + 1e-14) */
    double da[3], La = dista, dotDd = 0;
    for (int a = 0; a < 3; ++a) { da[a] = Da[a] / s->L; dotDd += Da[a] * s->d[a]; }
    La += -dotDd / (s->L * s->L);
    for (int a = 0; a < 3; ++a) da[a] += La * s->d[a] / s->L;
    for (int a = 0; a < 3; ++a) {
        pa[a] -= da[a];                       /* d = gp - pos */
        p1a[a] += cva[a] / c->dt;             /* cv = (pos1 - pos) / dt */
        pa[a] -= cva[a] / c->dt;
    }
}

/* one node of grid_op (:189-221).  With `va` != NULL also the reverse: va = grid_v_out.grad[I] -> gva (grid_v_in.grad),
 * gma (grid_m.grad), pose adjoints accumulated into pa / p1a ([P][3]). */
static void grid_node(const plb_omp_cfg* c, const int* I, double m, const double* vin, const double* ppos, const double* ppos1, double* vout,
                      const double* va, double* gva, double* gma, double* pa, double* p1a) {
    if (!(m > 1e-12)) {
        if (vout) vout[0] = vout[1] = vout[2] = 0.0;
        if (va) { gva[0] = gva[1] = gva[2] = 0.0; *gma = 0.0; }
        return;
    }
    const int n = c->n, P = c->n_prim;
    double v[MAXP + 2][3];
    cstate cs[MAXP];
    int hit[MAXP];
    for (int a = 0; a < 3; ++a) v[0][a] = vin[a] / m + c->dt * c->gravity[a] * 30.0;
    const double gp[3] = {I[0] * c->dx, I[1] * c->dx, I[2] * c->dx};
    for (int q = 0; q < P; ++q) hit[q] = collide_fwd(c, q, ppos + 3 * q, ppos1 + 3 * q, gp, v[q], v[q + 1], &cs[q]);
    /* boundary: the three axes act one after the other on the updated vector */
    double b[4][3];                     /* state before axis d */
    int lo[3], hi[3];
    double lin[3], lit[3], qv[3];
    for (int a = 0; a < 3; ++a) b[0][a] = v[P][a];
    for (int d = 0; d < 3; ++d) {
        double cur[3] = {b[d][0], b[d][1], b[d][2]};
        lo[d] = I[d] < 3 && cur[d] < 0;
        if (lo[d]) {
            if (d != 1 || c->ground_friction == 0) cur[d] = 0;
            else if (c->ground_friction < 10) {
                lin[d] = cur[1] + 1e-30;
                double vit[3] = {cur[0] - I[0] * 1e-30, cur[1] - lin[d] - I[1] * 1e-30, cur[2] - I[2] * 1e-30};
                lit[d] = sqrt(vit[0] * vit[0] + vit[1] * vit[1] + vit[2] * vit[2] + 1e-8);
                qv[d] = 1.0 + c->ground_friction * lin[d] / lit[d];
                const double s = 0.0 < qv[d] ? qv[d] : 0.0;
                for (int a = 0; a < 3; ++a) cur[a] = s * (vit[a] + I[a] * 1e-30);
                cur[1] = 0;
            } else cur[0] = cur[1] = cur[2] = 0;
        }
        hi[d] = I[d] > n - 3 && cur[d] > 0;
        if (hi[d]) cur[d] = 0;
        for (int a = 0; a < 3; ++a) b[d + 1][a] = cur[a];
    }
    if (vout) for (int a = 0; a < 3; ++a) vout[a] = b[3][a];
    if (!va) return;
    /* ---------------- reverse */
    double g[3] = {va[0], va[1], va[2]};
    for (int d = 2; d >= 0; --d) {
        if (hi[d]) g[d] = 0;
        if (lo[d]) {
            if (d != 1 || c->ground_friction == 0) g[d] = 0;
            else if (c->ground_friction < 10) {
                const double* cur = b[d];
                double vit[3] = {cur[0] - I[0] * 1e-30, cur[1] - lin[d] - I[1] * 1e-30, cur[2] - I[2] * 1e-30};
                const double s = 0.0 < qv[d] ? qv[d] : 0.0;
                g[1] = 0;                                             /* v_out[1] = 0 */
                double sa = 0, vita[3];
                for (int a = 0; a < 3; ++a) { sa += g[a] * (vit[a] + I[a] * 1e-30); vita[a] = s * g[a]; }
                const double qa = (0.0 < qv[d]) ? sa : 0.0;          /* max(q, 0): to q iff 0 < q */
                double lina = c->ground_friction * qa / lit[d];
                const double lita = -c->ground_friction * lin[d] * qa / (lit[d] * lit[d]);
                for (int a = 0; a < 3; ++a) vita[a] += lita * vit[a] / lit[d];
                lina -= vita[1];                                      /* vit = v - lin n - I 1e-30 */
                g[0] = vita[0]; g[1] = vita[1] + lina; g[2] = vita[2];
            } else g[0] = g[1] = g[2] = 0;
        }
    }
    for (int q = P - 1; q >= 0; --q)
        if (hit[q]) {
            double gin[3];
            collide_bwd(c, q, &cs[q], g, gin, pa + 3 * q, p1a + 3 * q);
            g[0] = gin[0]; g[1] = gin[1]; g[2] = gin[2];
        }
    /* v0 = v_in / m + const */
    double ma = 0;
    for (int a = 0; a < 3; ++a) { gva[a] = g[a] / m; ma -= g[a] * vin[a] / (m * m); }
    *gma = ma;
}

/* ------------------------------------------------------------------ the C ABI of this file */
/* scratch: gm[G], gv_in[3G], gv_out[3G] doubles (caller-provided, G = n^3) */
int plb_omp_substep_fwd(const plb_omp_cfg* c, const double* ppos, const double* ppos1, const double* x, const double* v, const double* C,
                        const double* F, const double* mu, const double* lam, const double* ys, double* x1, double* v1, double* C1, double* F1,
                        double* gm, double* gvin, double* gvout) {
    const int n = c->n;
    const size_t G = (size_t)n * n * n;
    p2g_all(c, x, v, C, F, mu, lam, ys, gm, gvin, F1);
#pragma omp parallel for schedule(static)
    for (size_t I = 0; I < G; ++I) {
        const int idx[3] = {(int)(I / ((size_t)n * n)), (int)((I / n) % n), (int)(I % n)};
        grid_node(c, idx, gm[I], gvin + 3 * I, ppos, ppos1, gvout + 3 * I, NULL, NULL, NULL, NULL, NULL);
    }
#pragma omp parallel for schedule(static)
    for (int p = 0; p < c->n_particles; ++p) {                       /* g2p :223-242 */
        int base[3];
        double fx[3], w[3][3], dw[3][3], nv[3] = {0, 0, 0}, nC[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        stencil(c, x + 3 * (size_t)p, base, fx, w, dw);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {
            const double dpos[3] = {i - fx[0], j - fx[1], k - fx[2]};
            const double* g = gvout + 3 * node(c, base[0] + i, base[1] + j, base[2] + k);
            const double weight = w[i][0] * w[j][1] * w[k][2];
            for (int a = 0; a < 3; ++a) {
                nv[a] += weight * g[a];
                for (int b = 0; b < 3; ++b) nC[3 * a + b] += 4.0 * c->inv_dx * weight * g[a] * dpos[b];
            }
        }
        for (int a = 0; a < 3; ++a) {
            v1[3 * (size_t)p + a] = nv[a];
            const double y = x[3 * (size_t)p + a] + c->dt * nv[a], hi = 1.0 - 3.0 * c->dx;
            const double mn = y < hi ? y : hi;
            x1[3 * (size_t)p + a] = 0.0 < mn ? mn : 0.0;
        }
        memcpy(C1 + 9 * (size_t)p, nC, sizeof nC);
    }
    return 0;
}

/* substep_grad (:260-278).  In: adjoints of frame f+1 (x1a, v1a, C1a, F1a).  Out: adjoints of frame f (xa, va, Ca, Fa,
 * overwritten) and the pose adjoints pa (pos[f]) / p1a (pos[f+1]), [P][3], accumulated into.
 * scratch: gm[G], gvin[3G], gvout[3G], gvout_a[3G], gvin_a[3G], gm_a[G]. */
int plb_omp_substep_bwd(const plb_omp_cfg* c, const double* ppos, const double* ppos1, const double* x, const double* v, const double* C,
                        const double* F, const double* mu, const double* lam, const double* ys, const double* x1a, const double* v1a,
                        const double* C1a, const double* F1a, double* xa, double* va, double* Ca, double* Fa, double* pa, double* p1a,
                        double* gm, double* gvin, double* gvout, double* gvout_a, double* gvin_a, double* gm_a) {
    const int n = c->n, N = c->n_particles, P = c->n_prim;
    const size_t G = (size_t)n * n * n;
    /* forward recompute: clear_grid, compute_F_tmp, svd, p2g, grid_op */
    p2g_all(c, x, v, C, F, mu, lam, ys, gm, gvin, NULL);
#pragma omp parallel for schedule(static)
    for (size_t I = 0; I < G; ++I) {
        const int idx[3] = {(int)(I / ((size_t)n * n)), (int)((I / n) % n), (int)(I % n)};
        grid_node(c, idx, gm[I], gvin + 3 * I, ppos, ppos1, gvout + 3 * I, NULL, NULL, NULL, NULL, NULL);
        gvout_a[3 * I] = gvout_a[3 * I + 1] = gvout_a[3 * I + 2] = 0.0;     /* clear_grid zeroes the grads too */
    }
    /* g2p.grad */
#pragma omp parallel for schedule(static)
    for (int p = 0; p < N; ++p) {
        int base[3];
        double fx[3], w[3][3], dw[3][3], nv[3] = {0, 0, 0};
        const double* xp = x + 3 * (size_t)p;
        stencil(c, xp, base, fx, w, dw);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {       /* new_v is needed for the clamp gate */
            const double* g = gvout + 3 * node(c, base[0] + i, base[1] + j, base[2] + k);
            const double weight = w[i][0] * w[j][1] * w[k][2];
            for (int a = 0; a < 3; ++a) nv[a] += weight * g[a];
        }
        double nva[3], fxa[3] = {0, 0, 0};
        for (int a = 0; a < 3; ++a) {
            const double y = xp[a] + c->dt * nv[a], hi = 1.0 - 3.0 * c->dx;
            const double mn = y < hi ? y : hi;
            const double gate = (y < hi && 0.0 < mn) ? 1.0 : 0.0;     /* max(min(y, hi), 0) */
            xa[3 * (size_t)p + a] = gate * x1a[3 * (size_t)p + a];
            nva[a] = v1a[3 * (size_t)p + a] + c->dt * gate * x1a[3 * (size_t)p + a];
        }
        const double* Cn = C1a + 9 * (size_t)p;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {
            const double dpos[3] = {i - fx[0], j - fx[1], k - fx[2]};
            const size_t I = node(c, base[0] + i, base[1] + j, base[2] + k);
            const double* g = gvout + 3 * I;
            const double weight = w[i][0] * w[j][1] * w[k][2];
            double wa = 0, dposa[3] = {0, 0, 0};
            for (int a = 0; a < 3; ++a) {
                double ga = nva[a];
                for (int b = 0; b < 3; ++b) {
                    ga += 4.0 * c->inv_dx * Cn[3 * a + b] * dpos[b];
                    wa += 4.0 * c->inv_dx * Cn[3 * a + b] * g[a] * dpos[b];
                    dposa[b] += 4.0 * c->inv_dx * weight * Cn[3 * a + b] * g[a];
                }
                wa += nva[a] * g[a];
#pragma omp atomic
                gvout_a[3 * I + a] += weight * ga;
            }
            fxa[0] += wa * dw[i][0] * w[j][1] * w[k][2] - dposa[0];
            fxa[1] += wa * w[i][0] * dw[j][1] * w[k][2] - dposa[1];
            fxa[2] += wa * w[i][0] * w[j][1] * dw[k][2] - dposa[2];
        }
        for (int a = 0; a < 3; ++a) xa[3 * (size_t)p + a] += c->inv_dx * fxa[a];
    }
    /* grid_op.grad (pose adjoints: per-thread partial sums, then one reduction) */
    int nt = 1;
#ifdef _OPENMP
    nt = omp_get_max_threads();
#endif
    double* part = (double*)calloc((size_t)nt * 2 * MAXP * 3, sizeof(double));
#pragma omp parallel
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        double* mypa = part + (size_t)tid * 2 * MAXP * 3;
        double* myp1a = mypa + MAXP * 3;
#pragma omp for schedule(static)
        for (size_t I = 0; I < G; ++I) {
            const int idx[3] = {(int)(I / ((size_t)n * n)), (int)((I / n) % n), (int)(I % n)};
            grid_node(c, idx, gm[I], gvin + 3 * I, ppos, ppos1, NULL, gvout_a + 3 * I, gvin_a + 3 * I, gm_a + I, mypa, myp1a);
        }
    }
    for (int t = 0; t < nt; ++t)
        for (int i = 0; i < P * 3; ++i) { pa[i] += part[(size_t)t * 2 * MAXP * 3 + i]; p1a[i] += part[(size_t)t * 2 * MAXP * 3 + MAXP * 3 + i]; }
    free(part);
    /* p2g.grad + svd_grad + compute_F_tmp.grad */
    const double kappa = -c->dt * c->p_vol * 4.0 * c->inv_dx * c->inv_dx;
#pragma omp parallel for schedule(static)
    for (int p = 0; p < N; ++p) {
        pstate q;
        const double *Cp = C + 9 * (size_t)p, *Fp = F + 9 * (size_t)p, *vp = v + 3 * (size_t)p;
        particle_fwd(c, Cp, Fp, mu[p], lam[p], ys[p], &q);
        int base[3];
        double fx[3], w[3][3], dw[3][3];
        stencil(c, x + 3 * (size_t)p, base, fx, w, dw);
        double vacc[3] = {0, 0, 0}, affa[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, fxa[3] = {0, 0, 0};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {
            const double dpos[3] = {(i - fx[0]) * c->dx, (j - fx[1]) * c->dx, (k - fx[2]) * c->dx};
            const double weight = w[i][0] * w[j][1] * w[k][2];
            const size_t I = node(c, base[0] + i, base[1] + j, base[2] + k);
            const double* ga = gvin_a + 3 * I;
            double wa = gm_a[I] * c->p_mass, dposa[3] = {0, 0, 0};
            for (int a = 0; a < 3; ++a) {
                const double val = c->p_mass * vp[a] + q.affine[3 * a] * dpos[0] + q.affine[3 * a + 1] * dpos[1] + q.affine[3 * a + 2] * dpos[2];
                wa += ga[a] * val;
                vacc[a] += weight * c->p_mass * ga[a];
                for (int b = 0; b < 3; ++b) { affa[3 * a + b] += weight * ga[a] * dpos[b]; dposa[b] += weight * ga[a] * q.affine[3 * a + b]; }
            }
            fxa[0] += wa * dw[i][0] * w[j][1] * w[k][2] - c->dx * dposa[0];
            fxa[1] += wa * w[i][0] * dw[j][1] * w[k][2] - c->dx * dposa[1];
            fxa[2] += wa * w[i][0] * w[j][1] * dw[k][2] - c->dx * dposa[2];
        }
        for (int a = 0; a < 3; ++a) { xa[3 * (size_t)p + a] += c->inv_dx * fxa[a]; va[3 * (size_t)p + a] = vacc[a]; }
        double Cacc[9], stra[9];
        for (int i = 0; i < 9; ++i) { Cacc[i] = c->p_mass * affa[i]; stra[i] = kappa * affa[i]; }
        /* stress = 2 mu (newF - r) newF^T + I lam J (J - 1) */
        double A[9], Aa[9], nFa[9], t[9];
        for (int i = 0; i < 9; ++i) A[i] = q.newF[i] - q.r[i];
        mm(stra, q.newF, Aa);
        mm_tn(stra, A, t);
        for (int i = 0; i < 9; ++i) { Aa[i] *= 2.0 * mu[p]; nFa[i] = 2.0 * mu[p] * t[i] + Aa[i] + F1a[9 * (size_t)p + i]; }
        const double Ja = lam[p] * (2.0 * q.J - 1.0) * (stra[0] + stra[4] + stra[8]);
        double cof[9];
        cofactor(q.newF, cof);
        for (int i = 0; i < 9; ++i) nFa[i] += Ja * cof[i];
        double Ua[9], Va[9], siga[3] = {0, 0, 0}, Fta[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        /* r = U V^T, r.grad = -A.grad */
        double ra[9];
        for (int i = 0; i < 9; ++i) ra[i] = -Aa[i];
        mm(ra, q.V, Ua);
        mm_tn(ra, q.U, Va);
        /* compute_von_mises.grad */
        if (q.yielded) {
            double VS[9], US[9], M[9];
            for (int r = 0; r < 3; ++r) for (int i = 0; i < 3; ++i) { VS[3 * r + i] = q.V[3 * r + i] * q.sp[i]; US[3 * r + i] = q.U[3 * r + i] * q.sp[i]; }
            mm(nFa, VS, t);
            for (int i = 0; i < 9; ++i) Ua[i] += t[i];
            mm_tn(nFa, US, t);
            for (int i = 0; i < 9; ++i) Va[i] += t[i];
            mm_tn(q.U, nFa, t); mm(t, q.V, M);                        /* sp.grad = diag(U^T newF.grad V) */
            double ga[3], ehata[3], ka = 0, ea[3], sum = 0;
            for (int i = 0; i < 3; ++i) { ga[i] = M[4 * i] * q.sp[i]; ea[i] = ga[i]; ehata[i] = -q.k * ga[i]; ka -= ga[i] * q.ehat[i]; }
            const double cc = ys[p] / (2.0 * mu[p]);
            const double nrma = ka * cc / (q.nrm * q.nrm);            /* k = 1 - c / nrm */
            for (int i = 0; i < 3; ++i) { ehata[i] += nrma * q.ehat[i] / q.nrm; sum += ehata[i]; }
            for (int i = 0; i < 3; ++i) {
                ea[i] += ehata[i] - sum / 3.0;
                const double sa = ea[i] / q.s[i];
                if (0.05 < q.sig[i]) siga[i] += sa;                  /* ti.max(sig, 0.05) */
            }
        } else
            for (int i = 0; i < 9; ++i) Fta[i] += nFa[i];
        double bs[9];
        backward_svd(Ua, siga, Va, q.U, q.sig, q.V, bs);             /* svd_grad */
        for (int i = 0; i < 9; ++i) Fta[i] += bs[i];
        /* compute_F_tmp.grad: F_tmp = (I + dt C) F */
        mm_nt(Fta, Fp, t);
        for (int i = 0; i < 9; ++i) Ca[9 * (size_t)p + i] = Cacc[i] + c->dt * t[i];
        double ICdt[9];
        for (int i = 0; i < 9; ++i) ICdt[i] = c->dt * Cp[i] + ((i % 4 == 0) ? 1.0 : 0.0);
        mm_tn(ICdt, Fta, Fa + 9 * (size_t)p);
    }
    return 0;
}

int plb_omp_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void plb_omp_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
