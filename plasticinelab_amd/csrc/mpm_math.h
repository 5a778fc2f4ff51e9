// Per-particle / per-node arithmetic of the differentiable MLS-MPM substep.
//
// Everything here is a templated inline function on a scalar type T (float on
// the fast path, double for the parity path).  The HIP kernels in
// mpm_kernels.hip call these from device code; tests/host_emul compiles the
// same header with g++ to check the hand-derived adjoints against the CPU
// oracle without a GPU (test infrastructure only -- the product never runs
// this on the host).
//
// What is computed follows /root/reference/plb/engine/mpm_simulator.py and
// plb/engine/primitive/*.py (cited per function); HOW is different:
//   * the deformation gradient is carried as E = F - I so that small strains
//     keep full relative precision in fp32;
//   * the 3x3 SVD is a Jacobi eigen-solve of F^T F - I (accurate sigma - 1);
//   * the adjoint never forms dU / dV: stress and the return-mapped F are
//     isotropic functions of F_tmp, so the VJP is written in the singular
//     basis with divided differences.  The reference's +-1e-6 clamp in
//     backward_svd (mpm_simulator.py:143-151) is reproduced as a pairwise
//     attenuation min(1, |s_j^2 - s_i^2| / clamp) so results match it.
#pragma once
#include <math.h>
#include <stdint.h>
#include <type_traits>

#if defined(__HIPCC__)
#define PLB_HD __host__ __device__ __forceinline__
#define PLB_ROLL _Pragma("unroll 1")
#ifndef PLB_ROLL_G2PG_I
#define PLB_ROLL_G2PG_I PLB_ROLL
#endif
#ifndef PLB_ROLL_G2PG_J
#define PLB_ROLL_G2PG_J PLB_ROLL        // rolled: 128 instead of 168 VGPRs, which buys g2p.grad its 4th wave per SIMD (round 2:
#endif                                  // 37.0 -> 35.0 us); unrolled at 3 waves was the round-1 choice.  Unrolling i as well is slower
#ifndef PLB_ROLL_P2G_I
#define PLB_ROLL_P2G_I PLB_ROLL
#endif
#ifndef PLB_ROLL_G2P_I
#define PLB_ROLL_G2P_I PLB_ROLL
#endif
#ifndef PLB_ROLL_P2G_J
#define PLB_ROLL_P2G_J PLB_ROLL         // rolled: 8 B less scratch in the fused forward kernel, 55.1 -> 53.9 us (round 2)
#endif
#ifndef PLB_ROLL_G2P_J
#define PLB_ROLL_G2P_J
#endif
#ifndef PLB_ROLL_GATH_I
#define PLB_ROLL_GATH_I PLB_ROLL
#endif
#ifndef PLB_ROLL_GATH_J
#define PLB_ROLL_GATH_J
#endif
#define PLB_UNROLL _Pragma("unroll")
#else
#define PLB_HD inline
#define PLB_ROLL
#define PLB_ROLL_G2PG_I
#define PLB_ROLL_G2PG_J
#define PLB_ROLL_GATH_I
#define PLB_ROLL_GATH_J
#define PLB_ROLL_P2G_I
#define PLB_ROLL_G2P_I
#define PLB_ROLL_P2G_J
#define PLB_ROLL_G2P_J
#define PLB_UNROLL
#endif

namespace plb {

// ---------------------------------------------------------------- lane values
// The arithmetic below is written for a "lane value" T (float or double) without data-dependent branches on values:
// `sel(cond, a, b)` picks, `any(cond)` guards work that is rare.  (Round 3 also instantiated it for packs of two
// particles per lane -- v_pk_*_f32 -- which needed 25 % fewer vector instructions and ran 25 % slower at half the
// occupancy; that variant left the tree in round 5, see profiles/r03_notes.md and git history.)
// what belongs to a lane value T: its scalar, its condition type, the integer pack of its stencil base
template <class T> struct Lane { typedef T scalar; typedef bool mask; typedef int ivec; };

PLB_HD float sel(bool c, float a, float b) { return c ? a : b; }
PLB_HD double sel(bool c, double a, double b) { return c ? a : b; }
PLB_HD int sel(bool c, int a, int b) { return c ? a : b; }
PLB_HD bool any(bool c) { return c; }
// conversions between the value kinds of one lane layout
template <class To, class From> PLB_HD To cvt(From v) { return (To)v; }
PLB_HD int to_int(double v) { return (int)v; }
PLB_HD int to_int(float v) { return (int)v; }

// ---------------------------------------------------------------- scalar helpers
template <class T> PLB_HD T t_sqrt(T x);
template <> PLB_HD float t_sqrt<float>(float x) { return sqrtf(x); }
template <> PLB_HD double t_sqrt<double>(double x) { return sqrt(x); }
template <class T> PLB_HD T t_exp(T x);
template <> PLB_HD float t_exp<float>(float x) { return expf(x); }
template <> PLB_HD double t_exp<double>(double x) { return exp(x); }
template <class T> PLB_HD T t_log(T x);
template <> PLB_HD float t_log<float>(float x) { return logf(x); }
template <> PLB_HD double t_log<double>(double x) { return log(x); }
template <class T> PLB_HD T t_log1p(T x);
template <> PLB_HD float t_log1p<float>(float x) { return log1pf(x); }
template <> PLB_HD double t_log1p<double>(double x) { return log1p(x); }
template <class T> PLB_HD T t_expm1(T x);
template <> PLB_HD float t_expm1<float>(float x) { return expm1f(x); }
template <> PLB_HD double t_expm1<double>(double x) { return expm1(x); }
// 1-ulp hardware reciprocal / rsqrt / sqrt on the fp32 device path (v_rcp_f32, v_rsq_f32, v_sqrt_f32) instead
// of the ~10-instruction IEEE sequences; exact operations for double and on the host.
template <class T> PLB_HD T t_rcp(T x) { return T(1) / x; }
template <> PLB_HD float t_rcp<float>(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
template <class T> PLB_HD T t_rsqrt(T x) { return T(1) / t_sqrt(x); }
template <> PLB_HD float t_rsqrt<float>(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(x);
#else
    return 1.0f / sqrtf(x);
#endif
}
template <class T> PLB_HD T t_fsqrt(T x) { return t_sqrt(x); }
template <> PLB_HD float t_fsqrt<float>(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
// log(1+s) / exp(e)-1 with full relative accuracy at small arguments; short series on the fp32 device path
template <class T> PLB_HD T t_log1p_fast(T s) { return t_log1p(s); }
template <class T> PLB_HD T t_expm1_fast(T e) { return t_expm1(e); }
#if defined(__HIP_DEVICE_COMPILE__)
template <> PLB_HD float t_log1p_fast<float>(float s) {
    if (fabsf(s) < 0.25f) {                       // log1p(s) = 2 atanh(s / (2 + s)), |z| < 0.143
        float z = s * __builtin_amdgcn_rcpf(2.0f + s), z2 = z * z;
        return 2.0f * z * (1.0f + z2 * (1.0f / 3 + z2 * (1.0f / 5 + z2 * (1.0f / 7 + z2 * (1.0f / 9 + z2 * (1.0f / 11))))));
    }
    return __logf(1.0f + s);
}
template <> PLB_HD float t_expm1_fast<float>(float e) {
    if (fabsf(e) < 0.25f)
        return e * (1.0f + e * (0.5f + e * (1.0f / 6 + e * (1.0f / 24 + e * (1.0f / 120 + e * (1.0f / 720 + e * (1.0f / 5040)))))));
    return __expf(e) - 1.0f;
}
#endif
template <class T> PLB_HD T t_abs(T x) { return sel(x < T(0), -x, x); }
template <class T> PLB_HD T t_max(T a, T b) { return sel(a > b, a, b); }
template <class T> PLB_HD T t_min(T a, T b) { return sel(a < b, a, b); }
// pick one of three by a loop index that stays a run-time value (outer stencil loop is kept rolled on the GPU)
template <class T> PLB_HD T sel3(int i, T a, T b, T c) { return i == 0 ? a : (i == 1 ? b : c); }

template <class T> struct Tol;
template <> struct Tol<float> {
    static constexpr int sweeps = 4;
    static PLB_HD float dd() { return 2e-2f; }
    static PLB_HD float small_angle() { return 1e-6f; }
};
template <> struct Tol<double> {
    static constexpr int sweeps = 8;
    static PLB_HD double dd() { return 1e-4; }
    static PLB_HD double small_angle() { return 1e-12; }
};

// ---------------------------------------------------------------- 3x3 helpers (row major)
template <class T> PLB_HD void mat_mul(const T* a, const T* b, T* c) {          // c = a b
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
template <class T> PLB_HD void mat_mul_tn(const T* a, const T* b, T* c) {       // c = a^T b
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[3 * i + j] = a[i] * b[j] + a[3 + i] * b[3 + j] + a[6 + i] * b[6 + j];
}
template <class T> PLB_HD void mat_mul_nt(const T* a, const T* b, T* c) {       // c = a b^T
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[3 * j] + a[3 * i + 1] * b[3 * j + 1] + a[3 * i + 2] * b[3 * j + 2];
}
template <class T> PLB_HD T det3(const T* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
template <class T> PLB_HD void cross3(const T* a, const T* b, T* c) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
template <class T> PLB_HD T dot3(const T* a, const T* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// ---------------------------------------------------------------- simulation constants
template <class T> struct SimP {
    int n;              // n_grid                                (mpm_simulator.py:19)
    T dx, inv_dx, dt;   //                                        (:21-22)
    T p_mass;           // (dx/2)^2, the 2-D formula kept in 3-D  (:23-24)
    T kappa;            // -dt * p_vol * 4 * inv_dx^2             (:173)
    T grav[3];          // dt * gravity * 30                      (:194)
    T x_hi;             // 1 - 3 dx                               (:242)
    T ground_friction;  //                                        (:204-217)
    T svd_clamp;        // 1e-6 reproduces backward_svd's clamp; 0 = exact derivative
    T softness;         // Primitive.softness                     (primive_base.py:29)
    int tie_first;      // adjoint routing of max / min on exact ties (plmpm_config.minmax_tie): 0 second operand, 1 first
};

// Adjoint routing of Taichi's max(lhs, rhs) / min(lhs, rhs): the adjoint goes to ONE operand.  tie_first = 0: to lhs iff
// it is strictly the winner, ties to rhs (what Taichi 0.7.x's auto_diff.cpp is remembered to do -- unverified);
// tie_first = 1: ties to lhs.  Every max / min on the differentiated path routes through these two, so that a
// Taichi-generated golden vector that disagrees costs a flag (plmpm_config.minmax_tie), not a rewrite.
template <class A> PLB_HD auto max_to_lhs(A lhs, A rhs, int tie_first) -> decltype(lhs < rhs) { return tie_first ? !(lhs < rhs) : (rhs < lhs); }
template <class A> PLB_HD auto min_to_lhs(A lhs, A rhs, int tie_first) -> decltype(lhs < rhs) { return tie_first ? !(rhs < lhs) : (lhs < rhs); }

// ---------------------------------------------------------------- quadratic B-spline stencil
// base = trunc(x*inv_dx - 0.5) (Taichi cast(int) truncates); fx = x*inv_dx - base;
// w[k][d] for offset k in {0,1,2}; dw = d w / d fx.      (mpm_simulator.py:160-163)
// X is the position type: positions are carried in double even on the fp32 path, so that fx (and with
// it every weight) keeps full fp32 relative precision instead of the ~4e-6 an fp32 x*inv_dx would leave.
template <class T, class X> PLB_HD void stencil(const X* x, typename Lane<T>::scalar inv_dx, typename Lane<T>::ivec* base, T* fx, T (*w)[3], T (*dw)[3]) {
    for (int d = 0; d < 3; ++d) {
        X xs = x[d] * cvt<X>(inv_dx);
        base[d] = to_int(xs - X(0.5));
        T f = cvt<T>(xs - cvt<X>(base[d]));
        fx[d] = f;
        w[0][d] = T(0.5) * (T(1.5) - f) * (T(1.5) - f);
        w[1][d] = T(0.75) - (f - T(1)) * (f - T(1));
        w[2][d] = T(0.5) * (f - T(0.5)) * (f - T(0.5));
        if (dw) {
            dw[0][d] = -(T(1.5) - f);
            dw[1][d] = T(-2) * (f - T(1));
            dw[2][d] = f - T(0.5);
        }
    }
}

// ---------------------------------------------------------------- SVD of I + Et
template <class T> struct Svd3 {
    T U[9], V[9];
    T sig[3];   // singular values
    T s[3];     // sig - 1, accurate for small strain
    T lam[3];   // sig^2 - 1 (eigenvalues of F^T F - I)
};

template <class T> PLB_HD void jacobi_pair(T& app, T& aqq, T& apq, T& arp, T& arq, T* V, int p, int q) {
    // tan of the rotation angle that zeroes apq, smaller root: t = 2 apq / (d + sign(d) sqrt(d^2 + 4 apq^2)).
    // Branch-free on purpose (divergent branches here cost the GPU more than the arithmetic they skip); the
    // form has no theta = d / (2 apq) intermediate, so a vanishing apq cannot overflow it, and apq == 0 gives
    // t = 0, c = 1, s = 0: an exact no-op.
    const T d = aqq - app;
    const T h = t_fsqrt(d * d + T(4) * apq * apq);
    const T den = d + sel(d >= T(0), h, -h);
    const T t = sel(h > T(0), T(2) * apq * t_rcp(den), T(0));
    const T c = t_rsqrt(t * t + T(1));
    const T s = t * c;
    app -= t * apq;
    aqq += t * apq;
    apq = T(0);
    const T rp = c * arp - s * arq, rq = s * arp + c * arq;
    arp = rp; arq = rq;
    for (int k = 0; k < 3; ++k) {
        const T vp = V[3 * k + p], vq = V[3 * k + q];
        V[3 * k + p] = c * vp - s * vq;
        V[3 * k + q] = s * vp + c * vq;
    }
}

// Et = F_tmp - I.  Standard convention: sig >= 0, V a rotation, det U = sign det F.
// (ti.svd's own convention is third-party and unverified; results downstream are
// invariant to it whenever det F_tmp > 0.)                 (mpm_simulator.py:87-90)
// The decomposition in two halves.  svd_jacobi: the eigen-pairs (lam, V) of F^T F - I (the Jacobi sweeps); svd_finish:
// singular values and U from them.  (Keeping (lam, V) of the forward pass per particle and frame so that p2g.grad skips
// the sweeps was measured in round 2: no gain -- that kernel sits as close to its HBM bound as to its VALU bound, and
// the 24 MB of extra reads cost what the sweeps do; profiles/r02_notes.md.)
template <class T> PLB_HD void svd_jacobi(const T* Et, T* lam3, T* V) {
    // A = Et + Et^T + Et^T Et = F^T F - I
    T a00 = T(2) * Et[0] + Et[0] * Et[0] + Et[3] * Et[3] + Et[6] * Et[6];
    T a11 = T(2) * Et[4] + Et[1] * Et[1] + Et[4] * Et[4] + Et[7] * Et[7];
    T a22 = T(2) * Et[8] + Et[2] * Et[2] + Et[5] * Et[5] + Et[8] * Et[8];
    T a01 = Et[1] + Et[3] + Et[0] * Et[1] + Et[3] * Et[4] + Et[6] * Et[7];
    T a02 = Et[2] + Et[6] + Et[0] * Et[2] + Et[3] * Et[5] + Et[6] * Et[8];
    T a12 = Et[5] + Et[7] + Et[1] * Et[2] + Et[4] * Et[5] + Et[7] * Et[8];
    V[0] = V[4] = V[8] = T(1);
    V[1] = V[2] = V[3] = V[5] = V[6] = V[7] = T(0);
#if defined(__HIPCC__) && !defined(PLB_SVD_UNROLL)
#pragma unroll 1
#endif
    for (int sw = 0; sw < Tol<T>::sweeps; ++sw) {
        jacobi_pair(a00, a11, a01, a02, a12, V, 0, 1);
        jacobi_pair(a00, a22, a02, a01, a12, V, 0, 2);
        jacobi_pair(a11, a22, a12, a01, a02, V, 1, 2);
    }
    lam3[0] = a00; lam3[1] = a11; lam3[2] = a22;
}
template <class T> PLB_HD void svd_finish(const T* Et, Svd3<T>& r) {
    const T* V = r.V;
    for (int i = 0; i < 3; ++i) {
        T sg = t_fsqrt(t_max(T(1) + r.lam[i], T(0)));
        r.sig[i] = sg;
        r.s[i] = r.lam[i] * t_rcp(T(1) + sg);
    }
    // U = F V Sigma^-1, F = I + Et
    for (int i = 0; i < 3; ++i) {
        T inv = t_rcp(t_max(r.sig[i], T(1e-30)));
        for (int k = 0; k < 3; ++k)
            r.U[3 * k + i] = (V[3 * k + i] + Et[3 * k] * V[i] + Et[3 * k + 1] * V[3 + i] + Et[3 * k + 2] * V[6 + i]) * inv;
    }
    // nearly singular F: rebuild the weakest column -- the FIRST one whose singular value is the smallest -- from the cross product
    // of the other two.  Straight-line selects on purpose: written as a loop over the columns with a branch per column, hipcc
    // turned the "which column" into a run-time index into a private array, i.e. scratch memory -- two scratch stores and four
    // dependent scratch loads in EVERY wave of the scatter kernels for a branch that is never taken on a sane state, and a
    // 7-word scratch frame (round 6, found in the device listing).
    T smin = t_min(r.sig[0], t_min(r.sig[1], r.sig[2]));
    typename Lane<T>::mask todo = smin < T(1e-3);
    if (any(todo)) {
        T Fm[9];
        for (int i = 0; i < 9; ++i) Fm[i] = Et[i];
        Fm[0] += T(1); Fm[4] += T(1); Fm[8] += T(1);
        T sgn = sel(det3(Fm) < T(0), T(-1), T(1));
        const typename Lane<T>::mask hit0 = todo && (r.sig[0] == smin);
        const typename Lane<T>::mask hit1 = todo && !hit0 && (r.sig[1] == smin);
        const typename Lane<T>::mask hit2 = todo && !hit0 && !hit1 && (r.sig[2] == smin);
        // column k is rebuilt from columns a = (k + 1) % 3 and b = (k + 2) % 3
        T ua[3], ub[3], uc[3];
        for (int row = 0; row < 3; ++row) {
            const T c0 = r.U[3 * row], c1 = r.U[3 * row + 1], c2 = r.U[3 * row + 2];
            ua[row] = sel(hit0, c1, sel(hit1, c2, c0));
            ub[row] = sel(hit0, c2, sel(hit1, c0, c1));
        }
        cross3(ua, ub, uc);
        T nrm = t_sqrt(dot3(uc, uc));
        T sc = sel(nrm > T(0), sgn / nrm, T(0));
        for (int row = 0; row < 3; ++row) {
            const T v = uc[row] * sc;
            r.U[3 * row] = sel(hit0, v, r.U[3 * row]);
            r.U[3 * row + 1] = sel(hit1, v, r.U[3 * row + 1]);
            r.U[3 * row + 2] = sel(hit2, v, r.U[3 * row + 2]);
        }
    }
}

template <class T> PLB_HD void svd_eform(const T* Et, Svd3<T>& r) {
    svd_jacobi(Et, r.lam, r.V);
    svd_finish(Et, r);
}

// ---------------------------------------------------------------- elastic fast path (round 4)
// A wave in which NO lane can yield does not need the singular value decomposition at all.  Without yielding
// compute_von_mises (mpm_simulator.py:124-141) returns F_tmp itself (sig >= 0.05 cannot bind either), and the stress of
// p2g (:164-171), 2 mu (F - R) F^T + lam J (J - 1) I, needs only the rotation R = U V^T of the polar decomposition
// F = R S -- a 3-step Newton iteration on I + E instead of 12 Jacobi rotations, log / exp and the U h U^T product -- and
// its VJP has a closed form in (R, S) (elastic_vjp below) instead of the divided differences in the singular basis.
// The decision is wave-uniform (one vote per wave, both code paths stay free of divergence) and made from
// A = F^T F - I alone, with SUFFICIENT conditions:
//   no yield:   ||dev eps|| <= ||dev A||_F / (2 (1 + lam_min)),  eps = log sig = log(1 + lam) / 2,  lam_min >= -||A||_F,
//               and the reference adds 1e-8 under the root (norm <= ||dev eps|| + 1e-4);
//   backward_svd's clamp (mpm_simulator.py:143-151) inactive: every pair of eigenvalues of A at least `clamp` apart.
//               With B = dev A, p = ||B||_F^2, q = det B the eigenvalues are 2 r cos(theta + 2 pi k / 3), r = sqrt(p / 6),
//               cos 3 theta = q / (2 r^3), and the smallest gap is >= (2 / 3) sqrt(p t), t = 1 - |q| / (2 r^3);
//   or the clamp removes the rotation term altogether (all gaps << clamp: F_tmp a multiple of a rotation, e.g. the
//               undeformed state): the divided-difference VJP then equals the closed form WITHOUT its dR term.
// Everything in between (and every wave with an inverted, crushed or possibly yielding particle) takes the Jacobi path.
#ifndef PLB_FAST
#define PLB_FAST 0          // measured (round 4, profiles/r04_notes.md): parity-green and not faster -- the kernels do not follow their
#endif                      // constitutive instruction count (the whole block is 3.3 us of the forward kernel's 48.5); opt-in: -DPLB_FAST=1
template <class T> struct TolFast;
template <> struct TolFast<float> {
    static constexpr int it_small = 3, it_big = 5;          // Newton steps for ||A|| <= 0.1 / < 0.9: error e -> e^2 / 2
    static PLB_HD float gap_margin() { return 4e-6f; }      // round-off of p t, relative to (||A|| + ||dev A||) ||dev A||
    static PLB_HD float pristine() { return 1e-6f; }        // largest attenuation factor treated as 0
};
template <> struct TolFast<double> {
    static constexpr int it_small = 4, it_big = 7;
    static PLB_HD double gap_margin() { return 1e-13; }
    static PLB_HD double pristine() { return 1e-11; }
};
PLB_HD bool wave_all(bool c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __all(c ? 1 : 0) != 0;
#else
    return c;
#endif
}
template <class T> struct Elastic {
    T Y[9];      // R - I, R the rotation of the polar decomposition of F_tmp = R S
    T D[6];      // S - I (symmetric): 00 11 22 01 02 12
    T Jm1;       // det F_tmp - 1 (= det S - 1)
    T rotw;      // 1, or 0 where the reference's clamp removes the dR term of the VJP
};
// per-lane test; `small`: ||A||_F <= 0.1 (three Newton steps suffice)
template <class T> PLB_HD bool elastic_lane_ok(const T* Et, T mu, T ys, T svd_clamp, bool need_gap, T* A6, T& rotw, bool& small) {
    const T a00 = T(2) * Et[0] + Et[0] * Et[0] + Et[3] * Et[3] + Et[6] * Et[6];
    const T a11 = T(2) * Et[4] + Et[1] * Et[1] + Et[4] * Et[4] + Et[7] * Et[7];
    const T a22 = T(2) * Et[8] + Et[2] * Et[2] + Et[5] * Et[5] + Et[8] * Et[8];
    const T a01 = Et[1] + Et[3] + Et[0] * Et[1] + Et[3] * Et[4] + Et[6] * Et[7];
    const T a02 = Et[2] + Et[6] + Et[0] * Et[2] + Et[3] * Et[5] + Et[6] * Et[8];
    const T a12 = Et[5] + Et[7] + Et[1] * Et[2] + Et[4] * Et[5] + Et[7] * Et[8];
    const T off2 = T(2) * (a01 * a01 + a02 * a02 + a12 * a12);
    const T a = t_fsqrt(a00 * a00 + a11 * a11 + a22 * a22 + off2);
    const T m = (a00 + a11 + a22) * (T(1) / T(3));
    const T b00 = a00 - m, b11 = a11 - m, b22 = a22 - m;
    const T p = b00 * b00 + b11 * b11 + b22 * b22 + off2;
    const T sp = t_fsqrt(p);
    // det F_tmp - 1 = tr E + tr cof E + det E
    const T c00 = Et[4] * Et[8] - Et[5] * Et[7], c11 = Et[0] * Et[8] - Et[2] * Et[6], c22 = Et[0] * Et[4] - Et[1] * Et[3];
    const T detE = Et[0] * c00 - Et[1] * (Et[3] * Et[8] - Et[5] * Et[6]) + Et[2] * (Et[3] * Et[7] - Et[4] * Et[6]);
    const T Jm1 = (Et[0] + Et[4] + Et[8]) + (c00 + c11 + c22) + detE;          // only its sign matters here (an inverted particle)
    A6[0] = a00; A6[1] = a11; A6[2] = a22; A6[3] = a01; A6[4] = a02; A6[5] = a12;
    small = a <= T(0.1);
    rotw = T(1);
    const T c = ys * t_rcp(T(2) * mu);
    bool ok = a < T(0.9) && Jm1 > T(-0.9) && sp < (c * T(1 - 1e-5) - T(1e-4)) * T(2) * (T(1) - a);
    if (need_gap && svd_clamp > T(0)) {
        const T q = b00 * (b11 * b22 - a12 * a12) - a01 * (a01 * b22 - a12 * a02) + a02 * (a01 * a12 - b11 * a02);
        const T pt = p > T(0) ? p - T(7.348469228349534) * t_abs(q) * t_rsqrt(p) : T(0);     // p t, 6^1.5 / 2
        const bool apart = pt >= T(2.25) * svd_clamp * svd_clamp + TolFast<T>::gap_margin() * (a + sp) * sp;
        const T lim = svd_clamp * TolFast<T>::pristine();
        const bool pristine = T(2) * p <= lim * lim;               // largest gap <= sqrt(2 p)
        rotw = apart ? T(1) : T(0);
        ok = ok && (apart || pristine);
    }
    return ok;
}
// Newton iteration for the polar rotation in deviation form: X = I + Y, Y <- (Y + (X^-T - I)) / 2 with
// X^-T - I = (cof Y - Y^T - (tr cof Y + det Y) I) / det X -- every term is O(Y), so R - I keeps its relative precision.
template <class T> PLB_HD void polar_newton(const T* Et, int iters, T* Y) {
    for (int i = 0; i < 9; ++i) Y[i] = Et[i];
    PLB_ROLL
    for (int it = 0; it < iters; ++it) {
        T cf[9];
        cf[0] = Y[4] * Y[8] - Y[5] * Y[7]; cf[1] = Y[5] * Y[6] - Y[3] * Y[8]; cf[2] = Y[3] * Y[7] - Y[4] * Y[6];
        cf[3] = Y[2] * Y[7] - Y[1] * Y[8]; cf[4] = Y[0] * Y[8] - Y[2] * Y[6]; cf[5] = Y[1] * Y[6] - Y[0] * Y[7];
        cf[6] = Y[1] * Y[5] - Y[2] * Y[4]; cf[7] = Y[2] * Y[3] - Y[0] * Y[5]; cf[8] = Y[0] * Y[4] - Y[1] * Y[3];
        const T trc = cf[0] + cf[4] + cf[8];
        const T det = Y[0] * cf[0] + Y[1] * cf[1] + Y[2] * cf[2];
        const T half_inv = T(0.5) * t_rcp(T(1) + (Y[0] + Y[4] + Y[8]) + trc + det);
        const T s = -(trc + det);
        T Z[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) Z[3 * r + c] = (cf[3 * r + c] - Y[3 * c + r] + (r == c ? s : T(0))) * half_inv;
        for (int i = 0; i < 9; ++i) Y[i] = T(0.5) * Y[i] + Z[i];
    }
}
// wave-uniform: true when every lane may take the fast path; then e holds the lane's rotation and stretch.
// The stretch deviation D = S - I: D0 = sym(R^T F) - I carries the round-off of R (eps x the rotation angle); one
// correction step against A = S^2 - I = 2 D + D^2, which the E-form gives to eps x angle^2, brings it down to that
// (D = D0 + (A - 2 D0 - D0^2) / 2) -- so a rotated, barely strained particle keeps the accuracy of the Jacobi path.
template <class T> PLB_HD bool elastic_try(const T* Et, T mu, T ys, T svd_clamp, bool need_gap, Elastic<T>& e) {
    bool small;
    T A[6];
    const bool ok = elastic_lane_ok(Et, mu, ys, svd_clamp, need_gap, A, e.rotw, small);
    if (!wave_all(ok)) return false;
    polar_newton(Et, wave_all(small) ? TolFast<T>::it_small : TolFast<T>::it_big, e.Y);
    const T* Y = e.Y;
    T M[9];
    mat_mul_tn(Y, Et, M);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) M[3 * r + c] += Y[3 * c + r] + Et[3 * r + c];
    const T d00 = M[0], d11 = M[4], d22 = M[8], d01 = T(0.5) * (M[1] + M[3]), d02 = T(0.5) * (M[2] + M[6]), d12 = T(0.5) * (M[5] + M[7]);
    T* D = e.D;
    D[0] = d00 + T(0.5) * (A[0] - T(2) * d00 - (d00 * d00 + d01 * d01 + d02 * d02));
    D[1] = d11 + T(0.5) * (A[1] - T(2) * d11 - (d01 * d01 + d11 * d11 + d12 * d12));
    D[2] = d22 + T(0.5) * (A[2] - T(2) * d22 - (d02 * d02 + d12 * d12 + d22 * d22));
    D[3] = d01 + T(0.5) * (A[3] - T(2) * d01 - (d00 * d01 + d01 * d11 + d02 * d12));
    D[4] = d02 + T(0.5) * (A[4] - T(2) * d02 - (d00 * d02 + d01 * d12 + d02 * d22));
    D[5] = d12 + T(0.5) * (A[5] - T(2) * d12 - (d01 * d02 + d11 * d12 + d12 * d22));
    // det S - 1 = tr D + tr cof D + det D
    const T c00 = D[1] * D[2] - D[5] * D[5], c11 = D[0] * D[2] - D[4] * D[4], c22 = D[0] * D[1] - D[3] * D[3];
    const T detD = D[0] * c00 - D[3] * (D[3] * D[2] - D[5] * D[4]) + D[4] * (D[3] * D[5] - D[1] * D[4]);
    e.Jm1 = (D[0] + D[1] + D[2]) + (c00 + c11 + c22) + detD;
    return true;
}
// stress of p2g (unscaled, as constitutive_fwd returns it): 2 mu (F - R) F^T + lam J (J - 1) I with
// (F - R) F^T = R D (R S)^T = R (D + D^2) R^T
template <class T> PLB_HD void elastic_stress(const T* Et, const Elastic<T>& e, T mu, T lam, T* stress) {
    const T* D = e.D;
    const T* Y = e.Y;
    T Tm[9];
    Tm[0] = D[0] + (D[0] * D[0] + D[3] * D[3] + D[4] * D[4]);
    Tm[4] = D[1] + (D[3] * D[3] + D[1] * D[1] + D[5] * D[5]);
    Tm[8] = D[2] + (D[4] * D[4] + D[5] * D[5] + D[2] * D[2]);
    Tm[1] = Tm[3] = D[3] + (D[0] * D[3] + D[3] * D[1] + D[4] * D[5]);
    Tm[2] = Tm[6] = D[4] + (D[0] * D[4] + D[3] * D[5] + D[4] * D[2]);
    Tm[5] = Tm[7] = D[5] + (D[3] * D[4] + D[1] * D[5] + D[5] * D[2]);
    T YT[9], YTY[9];
    mat_mul(Y, Tm, YT);
    mat_mul_nt(YT, Y, YTY);
    const T vol = lam * (T(1) + e.Jm1) * e.Jm1;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) stress[3 * r + c] = T(2) * mu * (Tm[3 * r + c] + YT[3 * r + c] + YT[3 * c + r] + YTY[3 * r + c]);
    stress[0] += vol; stress[4] += vol; stress[8] += vol;
}
// VJP of (new_F = F_tmp, stress) w.r.t. F_tmp on the elastic branch, G = GS:
//   d<G, F F^T>  = ((G + G^T) F) : dF
//   d<G, R F^T>  = <G F, dR> + (G^T R) : dF,   <Q, dR> = (2 R [y]x) : dF,  (tr S I - S) y = axl(skw(R^T Q)),  S = R^T F
//   d<G, J (J - 1) I> = tr G (2 J - 1) cof F : dF
// (dR = R W with W S + S W = R^T dF - dF^T R; the solve is the 3x3 system above.)  rotw = 0 drops the dR term.
template <class T> PLB_HD void elastic_vjp(const T* Et, const Elastic<T>& e, T mu, T lam, const T* GS, const T* GF, T* Ft_adj) {
    const T* Y = e.Y;
    T Q[9], t9[9];
    mat_mul(GS, Et, t9);
    for (int i = 0; i < 9; ++i) Q[i] = GS[i] + t9[i];                         // G F
    // skew part of R^T Q = Q + Y^T Q (only the three axial components)
    T M[9];
    mat_mul_tn(Y, Q, M);
    for (int i = 0; i < 9; ++i) M[i] += Q[i];
    const T tx = T(0.5) * (M[7] - M[5]), ty = T(0.5) * (M[2] - M[6]), tz = T(0.5) * (M[3] - M[1]);
    // H = tr S I - S = (2 + tr D) I - D,  D = S - I
    const T trD = e.D[0] + e.D[1] + e.D[2];
    const T h00 = T(2) + trD - e.D[0], h11 = T(2) + trD - e.D[1], h22 = T(2) + trD - e.D[2], h01 = -e.D[3], h02 = -e.D[4], h12 = -e.D[5];
    // y = H^-1 t by the adjugate (H is symmetric positive definite: its eigenvalues are the pairwise sums of the stretches)
    const T k00 = h11 * h22 - h12 * h12, k01 = h02 * h12 - h01 * h22, k02 = h01 * h12 - h02 * h11;
    const T k11 = h00 * h22 - h02 * h02, k12 = h01 * h02 - h00 * h12, k22 = h00 * h11 - h01 * h01;
    const T idet = e.rotw * t_rcp(h00 * k00 + h01 * k01 + h02 * k02);
    const T y0 = (k00 * tx + k01 * ty + k02 * tz) * idet, y1 = (k01 * tx + k11 * ty + k12 * tz) * idet, y2 = (k02 * tx + k12 * ty + k22 * tz) * idet;
    // W = [y]x ;  R W = W + Y W
    const T W[9] = {T(0), -y2, y1, y2, T(0), -y0, -y1, y0, T(0)};
    T YW[9], GtY[9], SF[9];
    mat_mul(Y, W, YW);
    mat_mul_tn(GS, Y, GtY);                                                 // G^T Y
    T Sg[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Sg[3 * r + c] = GS[3 * r + c] + GS[3 * c + r];
    mat_mul(Sg, Et, SF);                                                    // (G + G^T) E
    // cof F = (1 + tr E) I - E^T + cof E
    T cf[9];
    cf[0] = Et[4] * Et[8] - Et[5] * Et[7]; cf[1] = Et[5] * Et[6] - Et[3] * Et[8]; cf[2] = Et[3] * Et[7] - Et[4] * Et[6];
    cf[3] = Et[2] * Et[7] - Et[1] * Et[8]; cf[4] = Et[0] * Et[8] - Et[2] * Et[6]; cf[5] = Et[1] * Et[6] - Et[0] * Et[7];
    cf[6] = Et[1] * Et[5] - Et[2] * Et[4]; cf[7] = Et[2] * Et[3] - Et[0] * Et[5]; cf[8] = Et[0] * Et[4] - Et[1] * Et[3];
    const T trE = Et[0] + Et[4] + Et[8];
    const T kv = lam * (GS[0] + GS[4] + GS[8]) * (T(2) * (T(1) + e.Jm1) - T(1));
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            const int i = 3 * r + c;
            const T cofF = cf[i] - Et[3 * c + r] + (r == c ? T(1) + trE : T(0));
            //        (G + G^T) F          - 2 R W                 - G^T R
            Ft_adj[i] = GF[i] + T(2) * mu * ((Sg[i] + SF[i]) - T(2) * (W[i] + YW[i]) - (GS[3 * c + r] + GtY[i])) + kv * cofF;
        }
}

// ---------------------------------------------------------------- constitutive model
// compute_von_mises + stress of p2g               (mpm_simulator.py:124-141, :164-171)
template <class T> struct Consti {
    Svd3<T> svd;
    typename Lane<T>::mask yield;
    T g[3];      // singular values of new_F
    T gm1[3];    // g - 1
    T eps[3], eh[3], epsn[3];
    T nrm, c;    // ||dev eps|| (with the 1e-8 eps), yield_stress / (2 mu)
    T J, Jm1;    // det(new_F), det - 1
    T h[3];      // principal (unscaled) stress: 2 mu g (g-1) + lam J (J-1)
    typename Lane<T>::mask unc[3]; // sig > 0.05 (not clamped)
};

// Et = F_tmp - I.  Outputs: En = new_F - I (what is stored as F[f+1]), stress (unscaled).
template <class T> PLB_HD void constitutive_fwd(const T* Et, T mu, T lam, T ys, Consti<T>& k, T* En, T* stress, int tie_first = 0) {
    svd_eform(Et, k.svd);
    const Svd3<T>& S = k.svd;
    T mean = T(0);
    for (int i = 0; i < 3; ++i) {
        k.unc[i] = max_to_lhs(S.sig[i], T(0.05), tie_first);          // ti.max(sig, 0.05): adjoint to sig iff 0.05 < sig (ties: Q10)
        k.eps[i] = sel(k.unc[i], t_log1p_fast(S.s[i]), T(-2.995732273553991));     // log(0.05)
        mean += k.eps[i];
    }
    mean *= T(1) / T(3);
    T n2 = T(1e-8);
    for (int i = 0; i < 3; ++i) { k.eh[i] = k.eps[i] - mean; n2 += k.eh[i] * k.eh[i]; }
    k.nrm = t_fsqrt(n2);
    k.c = ys * t_rcp(T(2) * mu);
    k.yield = (k.nrm - k.c) > T(0);
    // elastic branch (every lane value starts from it; the values that yield are overwritten below)
    for (int i = 0; i < 3; ++i) { k.g[i] = S.sig[i]; k.gm1[i] = S.s[i]; k.epsn[i] = k.eps[i]; }
    for (int i = 0; i < 9; ++i) En[i] = Et[i];
    T detsign;
    {   // det(F_tmp) sign: only negative for inverted elements
        T Fm[9];
        for (int i = 0; i < 9; ++i) Fm[i] = Et[i];
        Fm[0] += T(1); Fm[4] += T(1); Fm[8] += T(1);
        detsign = sel(det3(Fm) < T(0), T(-1), T(1));
    }
    if (any(k.yield)) {
        T f = (k.nrm - k.c) * t_rcp(k.nrm);
        T gy[3];
        for (int i = 0; i < 3; ++i) {
            const T en = k.eps[i] - f * k.eh[i];
            const T gm = t_expm1_fast(en);
            gy[i] = T(1) + gm;
            k.epsn[i] = sel(k.yield, en, k.epsn[i]);
            k.gm1[i] = sel(k.yield, gm, k.gm1[i]);
            k.g[i] = sel(k.yield, gy[i], k.g[i]);
        }
        T US[9], Ey[9];
        for (int r = 0; r < 3; ++r)
            for (int i = 0; i < 3; ++i) US[3 * r + i] = S.U[3 * r + i] * gy[i];
        mat_mul_nt(US, S.V, Ey);
        Ey[0] -= T(1); Ey[4] -= T(1); Ey[8] -= T(1);
        for (int i = 0; i < 9; ++i) En[i] = sel(k.yield, Ey[i], En[i]);
        detsign = sel(k.yield, sel(det3(S.U) < T(0), T(-1), T(1)), detsign);     // det V = +1
    }
    T e1 = k.gm1[0] + k.gm1[1] + k.gm1[2];
    T e2 = k.gm1[0] * k.gm1[1] + k.gm1[0] * k.gm1[2] + k.gm1[1] * k.gm1[2];
    T e3 = k.gm1[0] * k.gm1[1] * k.gm1[2];
    T Jp1m = e1 + e2 + e3;                      // prod(g) - 1
    k.J = sel(detsign > T(0), T(1) + Jp1m, -(T(1) + Jp1m));
    k.Jm1 = sel(detsign > T(0), Jp1m, k.J - T(1));
    T vol = lam * k.J * k.Jm1;
    for (int i = 0; i < 3; ++i) k.h[i] = T(2) * mu * k.g[i] * k.gm1[i] + vol;
    T UH[9];
    for (int r = 0; r < 3; ++r)
        for (int i = 0; i < 3; ++i) UH[3 * r + i] = S.U[3 * r + i] * k.h[i];
    mat_mul_nt(UH, S.U, stress);
}

// stable divided differences
template <class T> PLB_HD T dd_exp(T a, T b) {       // (e^a - e^b)/(a - b)
    T d = a - b;
    if (t_abs(d) < Tol<T>::dd()) return t_exp(b) * (T(1) + d * (T(0.5) + d * (T(1) / T(6) + d * (T(1) / T(24)))));
    return (t_exp(a) - t_exp(b)) * t_rcp(d);
}
template <class T> PLB_HD T dd_log(T a, T b) {       // (log a - log b)/(a - b), a,b > 0
    T ib = t_rcp(b);
    T t = (a - b) * ib;
    if (t_abs(t) < Tol<T>::dd()) return (T(1) - t * (T(0.5) - t * (T(1) / T(3) - t * T(0.25)))) * ib;
    return t_log1p_fast(t) * t_rcp(t * b);
}


// VJP of (new_F, stress) w.r.t. F_tmp: returns Ft_adj = d<GF,new_F>/dFt + d<GS,stress>/dFt.
// Replaces p2g.grad's U/sig/V adjoints + svd_grad    (mpm_simulator.py:92-115, :276-277)
// One body for the elastic and the yielding branch: M (the adjoint in the singular basis) is built for the elastic
// branch, the lane values that yield overwrite it -- that part only runs when some value of the wave yields.
template <class T> PLB_HD void constitutive_vjp(const Consti<T>& k, T mu, T lam, typename Lane<T>::scalar svd_clamp_s,
                                                const T* GS, const T* GF, T* Ft_adj) {
    const Svd3<T>& S = k.svd;
    const T svd_clamp = T(svd_clamp_s);
    T tmp[9], Gs[9], Gf[9], M[9];
    mat_mul_tn(S.U, GS, tmp); mat_mul(tmp, S.U, Gs);            // U^T GS U
    const T* sig = S.sig;
    T att[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            T dl = t_abs(S.lam[j] - S.lam[i]);
            att[i][j] = sel((svd_clamp > T(0)) && (dl < svd_clamp), dl * t_rcp(svd_clamp), T(1));
        }
    T J = k.J;
    {
        T dvol = lam * (T(2) * J - T(1));
        for (int q = 0; q < 3; ++q) {
            T Jq = sel(sig[q] < T(1e-20), T(0), J * t_rcp(sig[q]));      // dJ/dsig_q
            M[4 * q] = T(2) * mu * (T(2) * sig[q] - T(1)) * Gs[4 * q] + dvol * Jq * (Gs[0] + Gs[4] + Gs[8]);
        }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                if (i == j) continue;
                T ssum = sig[i] + sig[j];
                T inv = sel(ssum > T(1e-20), t_rcp(ssum), T(0));
                T kij = T(2) * mu * (ssum - T(1)) * inv;
                M[3 * i + j] = kij * sig[j] * (Gs[3 * i + j] + Gs[3 * j + i])
                    + (T(1) - att[i][j]) * T(2) * mu * (Gs[3 * i + j] * sig[j] - Gs[3 * j + i] * sig[i]) * inv;
            }
    }
    if (!any(k.yield)) {
        T UM[9];
        mat_mul(S.U, M, UM); mat_mul_nt(UM, S.V, Ft_adj);
        for (int i = 0; i < 9; ++i) Ft_adj[i] += GF[i];
        return;
    }
    mat_mul_tn(S.U, GF, tmp); mat_mul(tmp, S.V, Gf);            // U^T GF V
    // d g_i / d sig_q
    T dg[3][3], dJ[3];
    T inrm = t_rcp(k.nrm);
    T cn = k.c * inrm, cn3 = k.c * inrm * inrm * inrm;
    for (int q = 0; q < 3; ++q) {
        T dsc = sel(k.unc[q], t_rcp(sig[q]), T(0));                // d eps_q / d sig_q
        dJ[q] = T(0);
        for (int i = 0; i < 3; ++i) {
            T de = T(1) / T(3) + cn * ((i == q ? T(1) : T(0)) - T(1) / T(3)) - cn3 * k.eh[i] * k.eh[q];
            dg[i][q] = k.g[i] * de * dsc;
            dJ[q] += (J * t_rcp(k.g[i])) * dg[i][q];
        }
    }
    T dvol = lam * (T(2) * J - T(1));
    for (int q = 0; q < 3; ++q) {
        T acc = T(0);
        for (int i = 0; i < 3; ++i) {
            T dh = T(2) * mu * (T(2) * k.g[i] - T(1)) * dg[i][q] + dvol * dJ[q];
            acc += dh * Gs[4 * i] + dg[i][q] * Gf[4 * i];
        }
        M[4 * q] = sel(k.yield, acc, M[4 * q]);
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            if (i == j) continue;
            const typename Lane<T>::mask both = k.unc[i] && k.unc[j];
            const T ds = sig[i] - sig[j];
            T ddg = sel(ds != T(0), (k.g[i] - k.g[j]) * t_rcp(ds), T(0));
            if (any(both)) ddg = sel(both, dd_exp(k.epsn[i], k.epsn[j]) * cn * dd_log(sig[i], sig[j]), ddg);
            T ssum = sig[i] + sig[j];
            T inv = sel(ssum > T(1e-20), t_rcp(ssum), T(0));
            T sumr = (k.g[i] + k.g[j]) * inv;
            T a = T(0.5) * (ddg + sumr), b = T(0.5) * (ddg - sumr);
            T kk = T(2) * mu * (k.g[i] + k.g[j] - T(1)) * ddg * inv;
            M[3 * i + j] = sel(k.yield, att[i][j] * (kk * sig[j] * (Gs[3 * i + j] + Gs[3 * j + i]) + a * Gf[3 * i + j] + b * Gf[3 * j + i]), M[3 * i + j]);
        }
    T UM[9];
    mat_mul(S.U, M, UM); mat_mul_nt(UM, S.V, Ft_adj);
    for (int i = 0; i < 9; ++i) Ft_adj[i] += sel(k.yield, T(0), GF[i]);
}

// ---------------------------------------------------------------- particle <-> grid
// compute_F_tmp in E-form: Et = E + dt C + dt C E        (mpm_simulator.py:82-85)
template <class T> PLB_HD void f_tmp_eform(const T* C, const T* E, typename Lane<T>::scalar dt, T* Et) {
    T CE[9];
    mat_mul(C, E, CE);
    for (int i = 0; i < 9; ++i) Et[i] = E[i] + dt * (C[i] + CE[i]);
}

// p2g body (mpm_simulator.py:157-184) in two halves, so that a kernel can put work between them (the scatter kernels reduce a
// bound of the momentum coefficients over the workgroup before the first contribution leaves a lane):
//   p2g_prepare: compute_F_tmp, the constitutive model, new E (F[f+1] - I) in En, and the particle's scatter coefficients;
//   p2g_emit:    Emit(k0,k1,k2, mass, mom[3]) for the 27 offsets.
// The momentum per unit weight at stencil offset o is affine in o:  q(o) = m v + A (o - fx) dx
//   = q0 + o_x ax + o_y ay + o_z az,   q0 = m v - A fx dx,  a_d = A[:,d] dx   (3 adds per node instead of a mat-vec)
template <class T> struct P2GCoef {        // (the weights w[k][d] travel beside it: the rolled stencil loops index them at run time)
    T q0[3], ax[3], ay[3], az[3];
    T pm;                   // mass per unit weight (p_mass; the fixed-point scatter folds its scale in)
};
template <class T, class X>
PLB_HD void p2g_prepare(const SimP<typename Lane<T>::scalar>& P, const X* x, const T* v, const T* C, const T* E,
                        T mu, T lam, T ys, T* En, typename Lane<T>::ivec* base, T (*w)[3], P2GCoef<T>& K) {
    T fx[3];
    stencil<T, X>(x, P.inv_dx, base, fx, w, nullptr);
    T Et[9], stress[9], A[9];
    f_tmp_eform(C, E, P.dt, Et);
    bool fast = false;
    if constexpr (PLB_FAST && std::is_floating_point<T>::value) {
        // a wave without a lane that can yield: polar rotation instead of the SVD (elastic fast path above); the forward
        // pass has no use for the eigenvalue gaps
        Elastic<T> el;
        fast = elastic_try(Et, mu, ys, T(0), false, el);
        if (fast) {
            for (int i = 0; i < 9; ++i) En[i] = Et[i];
            elastic_stress(Et, el, mu, lam, stress);
        }
    }
    if (!fast) {
        Consti<T> k;
        constitutive_fwd(Et, mu, lam, ys, k, En, stress, P.tie_first);
    }
    for (int i = 0; i < 9; ++i) A[i] = P.kappa * stress[i] + P.p_mass * C[i];
    for (int a = 0; a < 3; ++a) {
        K.ax[a] = A[3 * a] * P.dx; K.ay[a] = A[3 * a + 1] * P.dx; K.az[a] = A[3 * a + 2] * P.dx;
        K.q0[a] = P.p_mass * v[a] - (K.ax[a] * fx[0] + K.ay[a] * fx[1] + K.az[a] * fx[2]);
    }
    K.pm = T(P.p_mass);
}
template <class T, class Emit>
PLB_HD void p2g_emit(const T (*w)[3], const P2GCoef<T>& K, Emit&& emit) {
    const T q0[3] = {K.q0[0], K.q0[1], K.q0[2]}, ax[3] = {K.ax[0], K.ax[1], K.ax[2]}, ay[3] = {K.ay[0], K.ay[1], K.ay[2]},
            az[3] = {K.az[0], K.az[1], K.az[2]};
    PLB_ROLL_P2G_I
    for (int i = 0; i < 3; ++i) {
        const T wi = sel3(i, w[0][0], w[1][0], w[2][0]);
        const T fi = T(i);
        T qi[3] = {q0[0] + fi * ax[0], q0[1] + fi * ax[1], q0[2] + fi * ax[2]};
        PLB_ROLL_P2G_J
        for (int j = 0; j < 3; ++j) {
            T qj[3] = {qi[0] + T(j) * ay[0], qi[1] + T(j) * ay[1], qi[2] + T(j) * ay[2]};
            const T wij = wi * w[j][1];
            for (int l = 0; l < 3; ++l) {
                T wt = wij * w[l][2];
                T mom[3] = {wt * (qj[0] + T(l) * az[0]), wt * (qj[1] + T(l) * az[1]), wt * (qj[2] + T(l) * az[2])};
                emit(i, j, l, wt * K.pm, mom);
            }
        }
    }
}
// Returns new E (F[f+1] - I) in En.
template <class T, class X, class Emit>
PLB_HD void p2g_particle(const SimP<typename Lane<T>::scalar>& P, const X* x, const T* v, const T* C, const T* E,
                         T mu, T lam, T ys, T* En, typename Lane<T>::ivec* base, Emit&& emit) {
    P2GCoef<T> K;
    T w[3][3];
    p2g_prepare<T, X>(P, x, v, C, E, mu, lam, ys, En, base, w, K);
    p2g_emit<T>(w, K, emit);
}

// g2p body (mpm_simulator.py:223-242).  Fetch(k0,k1,k2, gv[3]) reads grid_v_out.
template <class T, class X, class Fetch>
PLB_HD void g2p_particle(const SimP<typename Lane<T>::scalar>& P, const X* x, X* xn, T* vn, T* Cn, Fetch&& fetch) {
    typename Lane<T>::ivec base[3];
    T fx[3], w[3][3];
    stencil<T, X>(x, P.inv_dx, base, fx, w, nullptr);
    for (int a = 0; a < 3; ++a) vn[a] = T(0);
    for (int a = 0; a < 9; ++a) Cn[a] = T(0);
#ifndef PLB_G2P_PLAIN
    // v' = sum_o w_o g_o and C'[a][b] = sum_o w_o g_o[a] dp_o[b] are sums over the 27 nodes of (a fetched value) x (one
    // 1-D factor per axis: w or z = (k - fx) w), so they are contracted one axis at a time -- z inside, then y, then
    // x -- as the reverse gather of p2g.grad does: ~280 multiply-adds per particle instead of ~430.
    T zw[3][3];                                        // (k - fx[d]) * w[k][d]
    for (int k = 0; k < 3; ++k)
        for (int d = 0; d < 3; ++d) zw[k][d] = (T(k) - fx[d]) * w[k][d];
    PLB_ROLL_G2P_I
    for (int i = 0; i < 3; ++i) {
        T Sww[3] = {T(0), T(0), T(0)}, Szw[3] = {T(0), T(0), T(0)}, Swz[3] = {T(0), T(0), T(0)};
        PLB_ROLL_G2P_J
        for (int j = 0; j < 3; ++j) {
            T Rw[3] = {T(0), T(0), T(0)}, Rz[3] = {T(0), T(0), T(0)};
            for (int l = 0; l < 3; ++l) {
                T gv[3];
                fetch(i, j, l, gv);
                for (int a = 0; a < 3; ++a) { Rw[a] += w[l][2] * gv[a]; Rz[a] += zw[l][2] * gv[a]; }
            }
            for (int a = 0; a < 3; ++a) { Sww[a] += w[j][1] * Rw[a]; Szw[a] += zw[j][1] * Rw[a]; Swz[a] += w[j][1] * Rz[a]; }
        }
        const T wi = sel3(i, w[0][0], w[1][0], w[2][0]), zi = sel3(i, zw[0][0], zw[1][0], zw[2][0]);
        for (int a = 0; a < 3; ++a) {
            vn[a] += wi * Sww[a];
            Cn[3 * a] += zi * Sww[a]; Cn[3 * a + 1] += wi * Szw[a]; Cn[3 * a + 2] += wi * Swz[a];
        }
    }
#else
    PLB_ROLL_G2P_I
    for (int i = 0; i < 3; ++i) {
        const T wi = sel3(i, w[0][0], w[1][0], w[2][0]);
        for (int j = 0; j < 3; ++j)
            for (int l = 0; l < 3; ++l) {
                T gv[3];
                fetch(i, j, l, gv);
                T wt = wi * w[j][1] * w[l][2];
                T dp[3] = {T(i) - fx[0], T(j) - fx[1], T(l) - fx[2]};
                for (int a = 0; a < 3; ++a) {
                    T wg = wt * gv[a];
                    vn[a] += wg;
                    Cn[3 * a] += wg * dp[0]; Cn[3 * a + 1] += wg * dp[1]; Cn[3 * a + 2] += wg * dp[2];
                }
            }
    }
#endif
    for (int a = 0; a < 9; ++a) Cn[a] *= T(4) * P.inv_dx;
    for (int d = 0; d < 3; ++d) {
        X y = x[d] + cvt<X>(P.dt) * cvt<X>(vn[d]);
        X hi = X(1) - X(3) / cvt<X>(P.n);
        xn[d] = t_max(t_min(y, hi), X(0));
    }
}

// g2p adjoint.  Inputs: x[f], the stored v[f+1] (= new_v, needed for the clamp gate), adjoints of
// (x,v,C)[f+1]; Fetch reads grid_v_out, Emit(k0,k1,k2, gv_adj[3]) scatters into grid_v_out.grad.
// Output xa = contribution to x[f].grad.
template <class T, class X, class Fetch, class Emit>
PLB_HD void g2p_particle_grad(const SimP<typename Lane<T>::scalar>& P, const X* x, const T* vn, const T* xn_a, const T* vn_a,
                              const T* Cn_a, T* xa, Fetch&& fetch, Emit&& emit) {
    typename Lane<T>::ivec base[3];
    T fx[3], w[3][3], dw[3][3];
    stencil<T, X>(x, P.inv_dx, base, fx, w, dw);
    T nva[3];
    for (int d = 0; d < 3; ++d) {
        X y = x[d] + cvt<X>(P.dt) * cvt<X>(vn[d]);
        X hi = X(1) - X(3) / cvt<X>(P.n);
        // max(min(y, hi), 0): adjoint reaches y iff y < hi and 0 < min(y,hi)   (Taichi min/max rule)
        T gate = sel(min_to_lhs(y, hi, P.tie_first) && max_to_lhs(t_min(y, hi), X(0), P.tie_first), T(1), T(0));
        xa[d] = gate * xn_a[d];
        nva[d] = vn_a[d] + P.dt * gate * xn_a[d];
    }
    // new_v.grad + C.grad reach node o through t_a(o) = nva[a] + c4 (Cn_a[a] . dp_o), dp_o = o - fx: affine in the
    // node offset, so it is stepped along the loops instead of re-evaluated (3 adds per node).
    // Its dp-derivative, -c4 sum_o w_o gv_o[a] Cn_a[a][b], needs sum_o w_o gv_o = new_v: the stored v[f+1] (vn).
    const T c4 = T(4) * P.inv_dx;
    T tc[9], t0[3], fxa[3] = {T(0), T(0), T(0)};
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) tc[3 * a + b] = c4 * Cn_a[3 * a + b];
        t0[a] = nva[a] - (tc[3 * a] * fx[0] + tc[3 * a + 1] * fx[1] + tc[3 * a + 2] * fx[2]);
        for (int b = 0; b < 3; ++b) fxa[b] -= vn[a] * tc[3 * a + b];
    }
    T ti[3] = {t0[0], t0[1], t0[2]};
    PLB_ROLL_G2PG_I
    for (int i = 0; i < 3; ++i) {
        const T wi = sel3(i, w[0][0], w[1][0], w[2][0]);
        const T dwi = sel3(i, dw[0][0], dw[1][0], dw[2][0]);
        T tj[3] = {ti[0], ti[1], ti[2]};
        PLB_ROLL_G2PG_J
        for (int j = 0; j < 3; ++j) {
            const T wj = sel3(j, w[0][1], w[1][1], w[2][1]);
            const T dwj = sel3(j, dw[0][1], dw[1][1], dw[2][1]);
            const T wij = wi * wj, gx = dwi * wj, gy = wi * dwj;
            T t[3] = {tj[0], tj[1], tj[2]};
            for (int l = 0; l < 3; ++l) {
                T gv[3];
                fetch(i, j, l, gv);
                const T wt = wij * w[l][2];
                const T ga[3] = {wt * t[0], wt * t[1], wt * t[2]};
                const T wa = gv[0] * t[0] + gv[1] * t[1] + gv[2] * t[2];
                emit(i, j, l, ga);
                fxa[0] += wa * (gx * w[l][2]);
                fxa[1] += wa * (gy * w[l][2]);
                fxa[2] += wa * (wij * dw[l][2]);
                for (int a = 0; a < 3; ++a) t[a] += tc[3 * a + 2];
            }
            for (int a = 0; a < 3; ++a) tj[a] += tc[3 * a + 1];
        }
        for (int a = 0; a < 3; ++a) ti[a] += tc[3 * a];
    }
    for (int d = 0; d < 3; ++d) xa[d] += P.inv_dx * fxa[d];
}

// p2g adjoint + svd_grad + compute_F_tmp.grad for one particle, in two halves.
//   p2g_gather_grad: everything that needs the grid -- uses only the particle's position, so a kernel can run it
//                    before the rest of the particle state is loaded (fewer live registers across the 27-node loop);
//   p2g_finish_grad: the constitutive VJP and the chain through F_tmp.
// Fetch(k0,k1,k2, g[4]) reads {grid_m.grad, grid_v_in.grad[3]}.  En_a = F[f+1].grad.
// xa_io already holds the g2p contribution and is accumulated into; va, Ca, Ea are written.
template <class T> struct P2GGather {
    T va[3];     // m sum_o w_o gva_o
    T Aa[9];     // sum_o w_o gva_o[a] dp_o[b]
    T M[27];     // sum_o gva_o[a] dp_o[b] dw_o/dfx_d
    T sm[3];     // sum_o gm_o dw_o/dfx_d
    T sv[9];     // sum_o gva_o[a] dw_o/dfx_d   ([3 d + a])
};
template <class T, class X, class Fetch>
PLB_HD void p2g_gather_grad(const SimP<typename Lane<T>::scalar>& P, const X* x, P2GGather<T>& G, Fetch&& fetch) {
    typename Lane<T>::ivec base[3];
    T fx[3], w[3][3], dw[3][3];
    stencil<T, X>(x, P.inv_dx, base, fx, w, dw);
    // Gather pass first, with accumulators that do not need the affine matrix A = kappa*stress + m C:
    //   va   = m sum_o w_o gva_o                 Aa[a][b] = sum_o w_o gva_o[a] dp_o[b]
    //   s1[d] = sum_o (m gm_o + gva_o . m v) dw_o/dfx_d
    //   M[a][b][d] = sum_o gva_o[a] dp_o[b] dw_o/dfx_d        (so that sum_o (gva_o^T A dp_o) dw_o = A : M)
    // (w_o = w_i w_j w_l, dw_o/dfx_d = the same product with dw on axis d, dp_o = (o - fx) dx.)
    // The SVD / return mapping then runs after the loop and its ~60 registers are not live across it.
    // Every accumulator is a sum over the 27 nodes of (one fetched field) x (a product of one 1-D factor per axis),
    // the factor being W = w, D = dw/dfx, ZW = (k - fx) w or ZD = (k - fx) dw.  The sums are therefore contracted
    // one axis at a time (z inside, then y, then x): ~800 multiply-adds per particle instead of ~1700.
    //   fields: g[0] = grid_m.grad (types W, D only), g[1..3] = grid_v_in.grad
    {
        enum { W = 0, Dd = 1, ZW = 2, ZD = 3 };
        // (type on y, type on z) pairs needed after the z- and y-contractions, and the (type on x, pair) combos
        // that make the final accumulators.  Everything below indexes these tables with compile-time constants.
        constexpr int kPairY[9] = {W, Dd, W, ZW, ZD, ZW, W, Dd, W};
        constexpr int kPairZ[9] = {W, W, Dd, W, W, Dd, ZW, ZW, ZD};
        //            va  Aa0 Aa1 Aa2 M00 M01 M02 M10 M11 M12 M20 M21 M22 s0  s1  s2
        constexpr int kCombX[16] = {W, ZW, W, W, ZD, ZW, ZW, Dd, W, W, Dd, W, W, Dd, W, W};
        constexpr int kCombP[16] = {0, 0, 3, 6, 0, 1, 2, 3, 4, 5, 6, 7, 8, 0, 1, 2};
        T cz[4][3];                                    // factors along z, all three stencil offsets
        for (int n = 0; n < 3; ++n) {
            const T z = T(n) - fx[2];
            cz[W][n] = w[n][2]; cz[Dd][n] = dw[n][2]; cz[ZW][n] = z * w[n][2]; cz[ZD][n] = z * dw[n][2];
        }
        T Fv[16][3], Fm[3];                            // final sums: vector field, mass field
        for (int c = 0; c < 16; ++c) Fv[c][0] = Fv[c][1] = Fv[c][2] = T(0);
        Fm[0] = Fm[1] = Fm[2] = T(0);
        PLB_ROLL_GATH_I
        for (int i = 0; i < 3; ++i) {
            T Sv[9][3], Sm[3];                         // sums over (j, l) for this i
            for (int q = 0; q < 9; ++q) Sv[q][0] = Sv[q][1] = Sv[q][2] = T(0);
            Sm[0] = Sm[1] = Sm[2] = T(0);
            PLB_ROLL_GATH_J
            for (int j = 0; j < 3; ++j) {
                T Rv[4][3], Rm[2];                     // sums over l for this (i, j)
                for (int t = 0; t < 4; ++t) Rv[t][0] = Rv[t][1] = Rv[t][2] = T(0);
                Rm[0] = Rm[1] = T(0);
                for (int l = 0; l < 3; ++l) {
                    T g[4];
                    fetch(i, j, l, g);
                    Rm[W] += cz[W][l] * g[0];
                    Rm[Dd] += cz[Dd][l] * g[0];
                    for (int t = 0; t < 4; ++t)
                        for (int a = 0; a < 3; ++a) Rv[t][a] += cz[t][l] * g[1 + a];
                }
                const T yw = sel3(j, w[0][1], w[1][1], w[2][1]), yd = sel3(j, dw[0][1], dw[1][1], dw[2][1]);
                const T zy = T(j) - fx[1];
                const T cy[4] = {yw, yd, zy * yw, zy * yd};
                for (int q = 0; q < 9; ++q)
                    for (int a = 0; a < 3; ++a) Sv[q][a] += cy[kPairY[q]] * Rv[kPairZ[q]][a];
                Sm[0] += cy[W] * Rm[W]; Sm[1] += cy[Dd] * Rm[W]; Sm[2] += cy[W] * Rm[Dd];
            }
            const T xw = sel3(i, w[0][0], w[1][0], w[2][0]), xd = sel3(i, dw[0][0], dw[1][0], dw[2][0]);
            const T zx = T(i) - fx[0];
            const T cx[4] = {xw, xd, zx * xw, zx * xd};
            for (int c = 0; c < 16; ++c)
                for (int a = 0; a < 3; ++a) Fv[c][a] += cx[kCombX[c]] * Sv[kCombP[c]][a];
            Fm[0] += cx[Dd] * Sm[0]; Fm[1] += cx[W] * Sm[1]; Fm[2] += cx[W] * Sm[2];
        }
        // dp = (k - fx) dx carries the dx
        for (int a = 0; a < 3; ++a) {
            G.va[a] = P.p_mass * Fv[0][a];
            for (int b = 0; b < 3; ++b) {
                G.Aa[3 * a + b] = P.dx * Fv[1 + b][a];
                for (int d = 0; d < 3; ++d) G.M[9 * a + 3 * b + d] = P.dx * Fv[4 + 3 * b + d][a];
            }
            for (int d = 0; d < 3; ++d) G.sv[3 * d + a] = Fv[13 + d][a];
        }
        for (int d = 0; d < 3; ++d) G.sm[d] = Fm[d];
    }
}

// ---- the same gather on packed fp32 pairs (fp32 engine, -DPLB_PK_GATHER=1) -------------------------------------------------------
// v_pk_fma_f32 does two multiply-adds per instruction in about the issue time of one and a quarter plain ones, and either half of
// a 64-bit source can be broadcast to both results by op_sel -- for free, IF the value already sits in a register pair (hipcc
// folds a splat of an element of a pair into op_sel; a splat of a lone scalar costs a move, which is what sank round 2's attempt:
// profiles/r02_notes.md).  So here every factor and every running sum lives in a pair from the start:
//   node level   fetched {mv'_x, mv'_y | mv'_z, m'} x z-factor pairs (W, D), (ZW, ZD): 7 packed instead of 14 plain multiply-adds
//   y level      14 packed + 2 plain instead of 30,        x level   22 packed + 7 plain instead of 51
// = 381 packed + 39 plain instead of ~800 plain per particle.  Same sums in the same order per accumulator (each is still one
// chain of 27 fused multiply-adds), so results agree with p2g_gather_grad to the last bit on the device; the host build (g++, no
// vector extension) emulates the pairs for tests/test_host_emul.py.
#ifndef PLB_PK_GATHER
#define PLB_PK_GATHER 0          // bit 0: the p2g.grad gather, bit 1: the g2p gathers (k_g2p, k_g2p_p2g); off until measured
#endif
#if defined(__clang__)
typedef float plb_f2 __attribute__((ext_vector_type(2)));
PLB_HD plb_f2 pk_fma(plb_f2 a, plb_f2 b, plb_f2 c) { return __builtin_elementwise_fma(a, b, c); }
#else
struct plb_f2 { float x, y; };
PLB_HD plb_f2 pk_fma(plb_f2 a, plb_f2 b, plb_f2 c) { plb_f2 r; r.x = fmaf(a.x, b.x, c.x); r.y = fmaf(a.y, b.y, c.y); return r; }
#endif
PLB_HD plb_f2 pk2(float a, float b) { plb_f2 r; r.x = a; r.y = b; return r; }
PLB_HD plb_f2 pk_lo(plb_f2 v) { return pk2(v.x, v.x); }          // both halves = the low / the high element (op_sel on the device)
PLB_HD plb_f2 pk_hi(plb_f2 v) { return pk2(v.y, v.y); }
// Fetch(k0, k1, k2, axy, azw): axy = {grid_v_in.grad x, y}, azw = {grid_v_in.grad z, grid_m.grad}
template <class X, class Fetch>
PLB_HD void p2g_gather_grad_pk(const SimP<float>& P, const X* x, P2GGather<float>& G, Fetch&& fetch) {
    int base[3];
    float fx[3], w[3][3], dw[3][3];
    stencil<float, X>(x, P.inv_dx, base, fx, w, dw);
    // z factors of the three offsets as pairs: (W, D) and (ZW, ZD)
    plb_f2 cz01[3], cz23[3];
    for (int n = 0; n < 3; ++n) {
        const float z = float(n) - fx[2];
        cz01[n] = pk2(w[n][2], dw[n][2]);
        cz23[n] = pk2(z * w[n][2], z * dw[n][2]);
    }
    const plb_f2 zero = pk2(0.f, 0.f);
    // final sums.  (x, y) components of combination c: Fxy[c]  (c as in p2g_gather_grad: va | Aa0..2 | M00..M22 | s0..2).
    // z components in the pairs the x level can produce with one instruction each:
    //   Fz_0_13 = (c0, c13)  Fz_1_4 = (c1, c4)  Fz_14_5 = (c14, c5)  Fz_15_6 = (c15, c6)  Fz_2_7 = (c2, c7)  Fz_3_10 = (c3, c10); c8, c9, c11, c12 alone
    plb_f2 Fxy[16];
    for (int c = 0; c < 16; ++c) Fxy[c] = zero;
    plb_f2 Fz_0_13 = zero, Fz_1_4 = zero, Fz_14_5 = zero, Fz_15_6 = zero, Fz_2_7 = zero, Fz_3_10 = zero;
    float Fz8 = 0.f, Fz9 = 0.f, Fz11 = 0.f, Fz12 = 0.f, Fm0 = 0.f, Fm1 = 0.f, Fm2 = 0.f;
    PLB_ROLL_GATH_I
    for (int i = 0; i < 3; ++i) {
        // sums over (j, l) for this i.  (x, y): Sxy[q], q = the (type on y, type on z) pair of p2g_gather_grad.  z components:
        //   Sz01 = (q0, q1)  Sz34 = (q3, q4)  Sz67 = (q6, q7)  Sz25 = (q2, q5)  Sz8;   mass: Sm02 = (Sm0, Sm2), Sm1
        plb_f2 Sxy[9];
        for (int q = 0; q < 9; ++q) Sxy[q] = zero;
        plb_f2 Sz01 = zero, Sz34 = zero, Sz67 = zero, Sz25 = zero, Sm02 = zero;
        float Sz8 = 0.f, Sm1 = 0.f;
        PLB_ROLL_GATH_J
        for (int j = 0; j < 3; ++j) {
            plb_f2 Rxy0 = zero, Rxy1 = zero, Rxy2 = zero, Rxy3 = zero;      // (x, y) x z type W, D, ZW, ZD
            plb_f2 Rz01 = zero, Rz23 = zero, Rm01 = zero;                    // z component x (W, D), (ZW, ZD); mass x (W, D)
            for (int l = 0; l < 3; ++l) {
                plb_f2 axy, azw;
                fetch(i, j, l, axy, azw);
                Rxy0 = pk_fma(axy, pk_lo(cz01[l]), Rxy0);
                Rxy1 = pk_fma(axy, pk_hi(cz01[l]), Rxy1);
                Rxy2 = pk_fma(axy, pk_lo(cz23[l]), Rxy2);
                Rxy3 = pk_fma(axy, pk_hi(cz23[l]), Rxy3);
                Rz01 = pk_fma(cz01[l], pk_lo(azw), Rz01);
                Rz23 = pk_fma(cz23[l], pk_lo(azw), Rz23);
                Rm01 = pk_fma(cz01[l], pk_hi(azw), Rm01);
            }
            const float yw = sel3(j, w[0][1], w[1][1], w[2][1]), yd = sel3(j, dw[0][1], dw[1][1], dw[2][1]);
            const float zy = float(j) - fx[1];
            const plb_f2 cy01 = pk2(yw, yd), cy23 = pk2(zy * yw, zy * yd), cy02 = pk2(yw, zy * yw);
            // (type on y, type on z) of q: (W,W) (D,W) (W,D) (ZW,W) (ZD,W) (ZW,D) (W,ZW) (D,ZW) (W,ZD)
            Sxy[0] = pk_fma(Rxy0, pk_lo(cy01), Sxy[0]);
            Sxy[1] = pk_fma(Rxy0, pk_hi(cy01), Sxy[1]);
            Sxy[2] = pk_fma(Rxy1, pk_lo(cy01), Sxy[2]);
            Sxy[3] = pk_fma(Rxy0, pk_lo(cy23), Sxy[3]);
            Sxy[4] = pk_fma(Rxy0, pk_hi(cy23), Sxy[4]);
            Sxy[5] = pk_fma(Rxy1, pk_lo(cy23), Sxy[5]);
            Sxy[6] = pk_fma(Rxy2, pk_lo(cy01), Sxy[6]);
            Sxy[7] = pk_fma(Rxy2, pk_hi(cy01), Sxy[7]);
            Sxy[8] = pk_fma(Rxy3, pk_lo(cy01), Sxy[8]);
            Sz01 = pk_fma(cy01, pk_lo(Rz01), Sz01);                          // q0 = yW zW, q1 = yD zW
            Sz34 = pk_fma(cy23, pk_lo(Rz01), Sz34);                          // q3 = yZW zW, q4 = yZD zW
            Sz67 = pk_fma(cy01, pk_lo(Rz23), Sz67);                          // q6 = yW zZW, q7 = yD zZW
            Sz25 = pk_fma(cy02, pk_hi(Rz01), Sz25);                          // q2 = yW zD, q5 = yZW zD
            Sz8 += yw * Rz23.y;                                              // q8 = yW zZD
            Sm02 = pk_fma(Rm01, pk_lo(cy01), Sm02);                          // Sm0 = yW mW, Sm2 = yW mD
            Sm1 += yd * Rm01.x;                                              // Sm1 = yD mW
        }
        const float xw = sel3(i, w[0][0], w[1][0], w[2][0]), xd = sel3(i, dw[0][0], dw[1][0], dw[2][0]);
        const float zx = float(i) - fx[0];
        const plb_f2 cx01 = pk2(xw, xd), cx23 = pk2(zx * xw, zx * xd), cx02 = pk2(xw, zx * xw);
        //              c:  0   1   2  3   4   5   6  7  8  9  10 11 12 13 14 15
        // type on x        W   ZW  W  W   ZD  ZW  ZW D  W  W  D  W  W  D  W  W
        // pair q           0   0   3  6   0   1   2  3  4  5  6  7  8  0  1  2
        Fxy[0] = pk_fma(Sxy[0], pk_lo(cx01), Fxy[0]);
        Fxy[1] = pk_fma(Sxy[0], pk_lo(cx23), Fxy[1]);
        Fxy[2] = pk_fma(Sxy[3], pk_lo(cx01), Fxy[2]);
        Fxy[3] = pk_fma(Sxy[6], pk_lo(cx01), Fxy[3]);
        Fxy[4] = pk_fma(Sxy[0], pk_hi(cx23), Fxy[4]);
        Fxy[5] = pk_fma(Sxy[1], pk_lo(cx23), Fxy[5]);
        Fxy[6] = pk_fma(Sxy[2], pk_lo(cx23), Fxy[6]);
        Fxy[7] = pk_fma(Sxy[3], pk_hi(cx01), Fxy[7]);
        Fxy[8] = pk_fma(Sxy[4], pk_lo(cx01), Fxy[8]);
        Fxy[9] = pk_fma(Sxy[5], pk_lo(cx01), Fxy[9]);
        Fxy[10] = pk_fma(Sxy[6], pk_hi(cx01), Fxy[10]);
        Fxy[11] = pk_fma(Sxy[7], pk_lo(cx01), Fxy[11]);
        Fxy[12] = pk_fma(Sxy[8], pk_lo(cx01), Fxy[12]);
        Fxy[13] = pk_fma(Sxy[0], pk_hi(cx01), Fxy[13]);
        Fxy[14] = pk_fma(Sxy[1], pk_lo(cx01), Fxy[14]);
        Fxy[15] = pk_fma(Sxy[2], pk_lo(cx01), Fxy[15]);
        Fz_0_13 = pk_fma(cx01, pk_lo(Sz01), Fz_0_13);                        // c0 = xW q0, c13 = xD q0
        Fz_1_4 = pk_fma(cx23, pk_lo(Sz01), Fz_1_4);                          // c1 = xZW q0, c4 = xZD q0
        Fz_14_5 = pk_fma(cx02, pk_hi(Sz01), Fz_14_5);                        // c14 = xW q1, c5 = xZW q1
        Fz_15_6 = pk_fma(cx02, pk_lo(Sz25), Fz_15_6);                        // c15 = xW q2, c6 = xZW q2
        Fz_2_7 = pk_fma(cx01, pk_lo(Sz34), Fz_2_7);                          // c2 = xW q3, c7 = xD q3
        Fz_3_10 = pk_fma(cx01, pk_lo(Sz67), Fz_3_10);                        // c3 = xW q6, c10 = xD q6
        Fz8 += xw * Sz34.y;                                                  // c8 = xW q4
        Fz9 += xw * Sz25.y;                                                  // c9 = xW q5
        Fz11 += xw * Sz67.y;                                                 // c11 = xW q7
        Fz12 += xw * Sz8;                                                    // c12 = xW q8
        Fm0 += xd * Sm02.x; Fm1 += xw * Sm1; Fm2 += xw * Sm02.y;
    }
    const float Fz[16] = {Fz_0_13.x, Fz_1_4.x, Fz_2_7.x, Fz_3_10.x, Fz_1_4.y, Fz_14_5.y, Fz_15_6.y, Fz_2_7.y,
                          Fz8, Fz9, Fz_3_10.y, Fz11, Fz12, Fz_0_13.y, Fz_14_5.x, Fz_15_6.x};
    for (int a = 0; a < 3; ++a) {
        auto comp = [&](int c) { return a == 0 ? Fxy[c].x : (a == 1 ? Fxy[c].y : Fz[c]); };
        G.va[a] = P.p_mass * comp(0);
        for (int b = 0; b < 3; ++b) {
            G.Aa[3 * a + b] = P.dx * comp(1 + b);
            for (int d = 0; d < 3; ++d) G.M[9 * a + 3 * b + d] = P.dx * comp(4 + 3 * b + d);
        }
        for (int d = 0; d < 3; ++d) G.sv[3 * d + a] = comp(13 + d);
    }
    G.sm[0] = Fm0; G.sm[1] = Fm1; G.sm[2] = Fm2;
}

// g2p on packed pairs (fp32 engine, -DPLB_PK_GATHER=1): the gather of g2p_particle with every factor and running sum in a register
// pair -- 132 packed + 15 plain multiply-adds instead of 279 plain.  Fetch(k0, k1, k2, axy, azw): axy = {v_out x, y}, azw.x = v_out z.
template <class X, class Fetch>
PLB_HD void g2p_particle_pk(const SimP<float>& P, const X* x, X* xn, float* vn, float* Cn, Fetch&& fetch) {
    int base[3];
    float fx[3], w[3][3];
    stencil<float, X>(x, P.inv_dx, base, fx, w, nullptr);
    plb_f2 cz[3];                                          // (w, z w) along z for the three offsets
    for (int k = 0; k < 3; ++k) cz[k] = pk2(w[k][2], (float(k) - fx[2]) * w[k][2]);
    const plb_f2 zero = pk2(0.f, 0.f);
    // v' = sum w g;  C'[a][0] = sum (zx w) g_a, C'[a][1] = sum (zy w) g_a, C'[a][2] = sum (zz w) g_a
    plb_f2 Vxy = zero, C0xy = zero, C1xy = zero, C2xy = zero;     // (a = 0, a = 1) of v', C'[a][0], C'[a][1], C'[a][2]
    plb_f2 VC0z = zero;                                           // a = 2: (v'_z, C'[2][0])
    float C1z = 0.f, C2z = 0.f;                                   // C'[2][1], C'[2][2]
    PLB_ROLL_G2P_I
    for (int i = 0; i < 3; ++i) {
        plb_f2 Swwxy = zero, Szwxy = zero, Swzxy = zero, SwwSzw_z = zero;
        float Swz_z = 0.f;
        PLB_ROLL_G2P_J
        for (int j = 0; j < 3; ++j) {
            plb_f2 Rwxy = zero, Rzxy = zero, RwRz_z = zero;
            for (int l = 0; l < 3; ++l) {
                plb_f2 axy, azw;
                fetch(i, j, l, axy, azw);
                Rwxy = pk_fma(axy, pk_lo(cz[l]), Rwxy);
                Rzxy = pk_fma(axy, pk_hi(cz[l]), Rzxy);
                RwRz_z = pk_fma(cz[l], pk_lo(azw), RwRz_z);
            }
            const float wy = sel3(j, w[0][1], w[1][1], w[2][1]);
            const plb_f2 cy = pk2(wy, (float(j) - fx[1]) * wy);
            Swwxy = pk_fma(Rwxy, pk_lo(cy), Swwxy);
            Szwxy = pk_fma(Rwxy, pk_hi(cy), Szwxy);
            Swzxy = pk_fma(Rzxy, pk_lo(cy), Swzxy);
            SwwSzw_z = pk_fma(cy, pk_lo(RwRz_z), SwwSzw_z);
            Swz_z += wy * RwRz_z.y;
        }
        const float wi = sel3(i, w[0][0], w[1][0], w[2][0]);
        const plb_f2 cx = pk2(wi, (float(i) - fx[0]) * wi);
        Vxy = pk_fma(Swwxy, pk_lo(cx), Vxy);
        C0xy = pk_fma(Swwxy, pk_hi(cx), C0xy);
        C1xy = pk_fma(Szwxy, pk_lo(cx), C1xy);
        C2xy = pk_fma(Swzxy, pk_lo(cx), C2xy);
        VC0z = pk_fma(cx, pk_lo(SwwSzw_z), VC0z);
        C1z += wi * SwwSzw_z.y;
        C2z += wi * Swz_z;
    }
    const float c4 = 4.f * P.inv_dx;
    vn[0] = Vxy.x; vn[1] = Vxy.y; vn[2] = VC0z.x;
    Cn[0] = c4 * C0xy.x; Cn[1] = c4 * C1xy.x; Cn[2] = c4 * C2xy.x;
    Cn[3] = c4 * C0xy.y; Cn[4] = c4 * C1xy.y; Cn[5] = c4 * C2xy.y;
    Cn[6] = c4 * VC0z.y; Cn[7] = c4 * C1z; Cn[8] = c4 * C2z;
    for (int d = 0; d < 3; ++d) {
        X y = x[d] + cvt<X>(P.dt) * cvt<X>(vn[d]);
        X hi = X(1) - X(3) / cvt<X>(P.n);
        xn[d] = t_max(t_min(y, hi), X(0));
    }
}

template <class T>
PLB_HD void p2g_finish_grad(const SimP<typename Lane<T>::scalar>& P, const P2GGather<T>& G, const T* v, const T* C, const T* E, T mu, T lam, T ys,
                            const T* En_a, T* xa_io, T* va, T* Ca, T* Ea) {
    T Et[9], En[9], stress[9], A[9];
    f_tmp_eform(C, E, P.dt, Et);
    bool fast = false;
    Elastic<T> el;
    Consti<T> k;
    if constexpr (PLB_FAST && std::is_floating_point<T>::value)
        fast = elastic_try(Et, mu, ys, T(P.svd_clamp), true, el);         // wave-uniform (elastic fast path above)
    if (fast) elastic_stress(Et, el, mu, lam, stress);
    else constitutive_fwd(Et, mu, lam, ys, k, En, stress, P.tie_first);
    for (int i = 0; i < 9; ++i) A[i] = P.kappa * stress[i] + P.p_mass * C[i];
    const T* M = G.M;
    const T* Aa = G.Aa;
    for (int a = 0; a < 3; ++a) va[a] = G.va[a];
    T fxa[3];
    for (int d = 0; d < 3; ++d) {
        // s1[d] = sum_o (m gm_o + gva_o . m v) dw_o/dfx_d
        T acc = P.p_mass * (G.sm[d] + G.sv[3 * d] * v[0] + G.sv[3 * d + 1] * v[1] + G.sv[3 * d + 2] * v[2]);
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) acc += A[3 * a + b] * M[9 * a + 3 * b + d];
            acc -= P.dx * A[3 * a + d] * va[a] * t_rcp(T(P.p_mass));       // d/d dp: sum_o w_o A^T gva_o, dp = (k - fx) dx
        }
        fxa[d] = acc;
    }
    for (int d = 0; d < 3; ++d) xa_io[d] += P.inv_dx * fxa[d];
    T GS[9], Fta[9];
    for (int i = 0; i < 9; ++i) { Ca[i] = P.p_mass * Aa[i]; GS[i] = P.kappa * Aa[i]; }
    if constexpr (PLB_FAST && std::is_floating_point<T>::value) {
        if (fast) elastic_vjp(Et, el, mu, lam, GS, En_a, Fta);
        else constitutive_vjp(k, mu, lam, P.svd_clamp, GS, En_a, Fta);
    } else constitutive_vjp(k, mu, lam, P.svd_clamp, GS, En_a, Fta);
    // F_tmp = (I + dt C) F :  C.grad += dt Fta F^T ;  F.grad = (I + dt C)^T Fta
    T Fm[9];
    for (int i = 0; i < 9; ++i) Fm[i] = E[i];
    Fm[0] += T(1); Fm[4] += T(1); Fm[8] += T(1);
    T t1[9];
    mat_mul_nt(Fta, Fm, t1);
    for (int i = 0; i < 9; ++i) Ca[i] += P.dt * t1[i];
    mat_mul_tn(C, Fta, t1);
    for (int i = 0; i < 9; ++i) Ea[i] = Fta[i] + P.dt * t1[i];
}


template <class T, class X, class Fetch>
PLB_HD void p2g_particle_grad(const SimP<typename Lane<T>::scalar>& P, const X* x, const T* v, const T* C, const T* E,
                              T mu, T lam, T ys, const T* En_a, T* xa_io, T* va, T* Ca, T* Ea,
                              Fetch&& fetch) {
    P2GGather<T> G;
    p2g_gather_grad<T, X>(P, x, G, fetch);
    p2g_finish_grad<T>(P, G, v, C, E, mu, lam, ys, En_a, xa_io, va, Ca, Ea);
}

// mass-only scatter weights for the loss (mpm_simulator.py:382-392) are stencil() + products.

}  // namespace plb
