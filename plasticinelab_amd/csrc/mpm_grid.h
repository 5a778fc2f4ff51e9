// Grid-node arithmetic: momentum->velocity, gravity, rigid-primitive contact
// (collide), box boundary with ground friction -- forward and hand-derived
// adjoint -- plus the serial primitive kinematics chain.
//
// Follows /root/reference/plb/engine/mpm_simulator.py:189-221 (grid_op),
// plb/engine/primitive/primive_base.py:57-121,184-192 (sdf/normal/collider_v/
// collide/forward_kinematics/set_velocity), primitives.py:17-34 (Sphere),
// :36-61 (Capsule), :83-154 (Chopsticks), :157-183 (Cylinder), :193-213 (Torus), :223-251 (Box) and
// primitive/utils.py:3-48 (quaternions).  Shared by the HIP kernels and by
// tests/host_emul (see mpm_math.h).
#pragma once
#include "mpm_math.h"

namespace plb {

enum ShapeKind { SHAPE_SPHERE = 0, SHAPE_CAPSULE = 1, SHAPE_CYLINDER = 2, SHAPE_TORUS = 3, SHAPE_BOX = 4, SHAPE_CHOPSTICKS = 5 };

// One primitive as a grid node sees it during substep f: static description + pose at f and f+1.
// Rigid-body geometry (signed distance, normal, collider velocity and the pose adjoints) is always
// evaluated in double, also on the fp32 path: collider_v is (new_pos - grid_pos)/dt, a difference of
// O(1) positions divided by 1e-4, which in fp32 costs ~4 digits of the contact velocity.  It only runs
// on the few nodes touching a manipulator.
template <class T> struct PrimT {
    int shape;
    int movable;             // action_dim > 0: pose adjoints are wanted
    double par[3];           // Sphere: radius | Capsule: h, r | Cylinder: h(=radius), r(=half height) | Torus: tx, ty | Box: size
                             // | Chopsticks: h, r, gap[f] (the kernels fill par[2] from the gap trajectory)
    T friction;
    float rb;                // radius of a sphere around pos that contains the shape (cheap fp32 cull)
    double pos[3], rot[4];   // pose at frame f
    double pos1[3], rot1[4]; // pose at frame f+1
};

template <class T> struct PoseAdj {
    double pos[3], rot[4], pos1[3], rot1[4];
    double gap;              // Chopsticks: adjoint of gap[f]
    PLB_HD void zero() {
        for (int i = 0; i < 3; ++i) pos[i] = pos1[i] = 0.0;
        for (int i = 0; i < 4; ++i) rot[i] = rot1[i] = 0.0;
        gap = 0.0;
    }
};

// ------------------------------------------------------------------ quaternions (utils.py:7-47)
template <class T> PLB_HD void qrot(const T* q, const T* v, T* out) {
    T uv[3], uuv[3];
    cross3(q + 1, v, uv);
    cross3(q + 1, uv, uuv);
    for (int i = 0; i < 3; ++i) out[i] = v[i] + T(2) * (q[0] * uv[i] + uuv[i]);
}
template <class T> PLB_HD void qconj(const T* q, T* c) { c[0] = q[0]; c[1] = -q[1]; c[2] = -q[2]; c[3] = -q[3]; }
// adjoint of qrot w.r.t. q:  qa += d<g, qrot(q,v)>/dq
template <class T> PLB_HD void qrot_adj_q(const T* q, const T* v, const T* g, T* qa) {
    const T* u = q + 1;
    T uv[3], vg[3];
    cross3(u, v, uv);
    cross3(v, g, vg);
    qa[0] += T(2) * dot3(uv, g);
    T uvd = dot3(u, v), gu = dot3(g, u), gvd = dot3(g, v);
    for (int i = 0; i < 3; ++i)
        qa[1 + i] += T(2) * q[0] * vg[i] + T(2) * (g[i] * uvd + v[i] * gu - T(2) * u[i] * gvd);
}
// inv_trans (utils.py:43-47): out = qrot(normalize(conj(rot)), p - pos); iq returned for reuse
template <class T> PLB_HD void inv_trans(const T* p, const T* pos, const T* rot, T* out, T* iq) {
    T c[4];
    qconj(rot, c);
    T inv = T(1) / t_sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3]);
    for (int i = 0; i < 4; ++i) iq[i] = c[i] * inv;
    T d[3] = {p[0] - pos[0], p[1] - pos[1], p[2] - pos[2]};
    qrot(iq, d, out);
}
// adjoint of inv_trans w.r.t. (pos, rot) given adjoint g of its output
template <class T> PLB_HD void inv_trans_adj(const T* p, const T* pos, const T* rot, const T* iq, const T* g,
                                             T* pos_a, T* rot_a) {
    T d[3] = {p[0] - pos[0], p[1] - pos[1], p[2] - pos[2]};
    T iqa[4] = {T(0), T(0), T(0), T(0)};
    qrot_adj_q(iq, d, g, iqa);
    T ciq[4], da[3];
    qconj(iq, ciq);
    qrot(ciq, g, da);
    for (int i = 0; i < 3; ++i) pos_a[i] -= da[i];
    T nrm = t_sqrt(rot[0] * rot[0] + rot[1] * rot[1] + rot[2] * rot[2] + rot[3] * rot[3]);
    T dotv = iq[0] * iqa[0] + iq[1] * iqa[1] + iq[2] * iqa[2] + iq[3] * iqa[3];
    T ca[4];
    for (int i = 0; i < 4; ++i) ca[i] = (iqa[i] - iq[i] * dotv) / nrm;
    rot_a[0] += ca[0]; rot_a[1] -= ca[1]; rot_a[2] -= ca[2]; rot_a[3] -= ca[3];
}

// ------------------------------------------------------------------ shape SDFs in the local frame
template <class T> PLB_HD T len14(T a, T b) { return t_sqrt(a * a + b * b + T(1e-14)); }
template <class T> PLB_HD T len14(T a, T b, T c) { return t_sqrt(a * a + b * b + c * c + T(1e-14)); }

template <class T> PLB_HD T capsule_sdf(const T* par, const T* p) {          // primitives.py:42-47
    T y = p[1] + par[0] / T(2);
    y -= t_min(t_max(y, T(0)), par[0]);
    return len14(p[0], y, p[2]) - par[1];
}
template <class T> PLB_HD void capsule_normal(const T* par, const T* p, T* n) {   // primitives.py:49-54
    T y = p[1] + par[0] / T(2);
    y -= t_min(t_max(y, T(0)), par[0]);
    T inv = T(1) / len14(p[0], y, p[2]);
    n[0] = p[0] * inv; n[1] = y * inv; n[2] = p[2] * inv;
}
// reverse mode of (capsule_sdf, capsule_normal) w.r.t. p; min/max follow Taichi's adjoint routing
template <class T> PLB_HD void capsule_adj(const T* par, const T* p, T da, const T* na, T* pa, int tf = 0) {
    T t = p[1] + par[0] / T(2);
    T mx = t_max(t, T(0));                          // max(t, 0): adjoint to t iff 0 < t
    T dcl = (max_to_lhs(t, T(0), tf) && min_to_lhs(mx, par[0], tf)) ? T(1) : T(0);   // min(mx, h): adjoint to mx iff mx < h
    T cl = t_min(mx, par[0]);
    T y = t - cl, dy = T(1) - dcl;                  // y' and dy'/dy
    T L = len14(p[0], y, p[2]);
    T n[3] = {p[0] / L, y / L, p[2] / L};
    T nd = n[0] * na[0] + n[1] * na[1] + n[2] * na[2];
    T g[3];                                         // adjoint of p2 = (x, y', z)
    for (int i = 0; i < 3; ++i) g[i] = da * n[i] + (na[i] - n[i] * nd) / L;
    pa[0] += g[0]; pa[1] += g[1] * dy; pa[2] += g[2];
}

template <class T> PLB_HD T shape_sdf_local(int shape, const T* par, const T* p) {
    switch (shape) {
    case SHAPE_CAPSULE: return capsule_sdf(par, p);
    case SHAPE_CHOPSTICKS: {                              // primitives.py:111-117: min over two capsules, gap = par[2]
        T qa[3] = {p[0] - par[2] / T(2), p[1] + par[0] / T(2), p[2]};
        T qb[3] = {p[0] + par[2] / T(2), qa[1], p[2]};
        T a = capsule_sdf(par, qa), b = capsule_sdf(par, qb);
        return a < b ? a : b;
    }
    case SHAPE_CYLINDER: {                                // primitives.py:163-167 (h = radius, r = half height)
        T d0 = t_abs(len14(p[0], p[2])) - par[0], d1 = t_abs(p[1]) - par[1];
        return t_min(t_max(d0, d1), T(0)) + len14(t_max(d0, T(0)), t_max(d1, T(0)));
    }
    case SHAPE_TORUS: {                                   // primitives.py:199-202
        T q0 = len14(p[0], p[2]) - par[0];
        return len14(q0, p[1]) - par[1];
    }
    case SHAPE_BOX: {                                     // primitives.py:232-238
        T q[3] = {t_abs(p[0]) - par[0], t_abs(p[1]) - par[1], t_abs(p[2]) - par[2]};
        T o = len14(t_max(q[0], T(0)), t_max(q[1], T(0)), t_max(q[2], T(0)));
        return o + t_min(t_max(q[0], t_max(q[1], q[2])), T(0));
    }
    default: return T(0);
    }
}

template <class T> PLB_HD void shape_normal_local(int shape, const T* par, const T* p, T* n) {
    switch (shape) {
    case SHAPE_CHOPSTICKS: {                              // primitives.py:119-129: the normal of the nearer capsule
        T qa[3] = {p[0] - par[2] / T(2), p[1] + par[0] / T(2), p[2]};
        T qb[3] = {p[0] + par[2] / T(2), qa[1], p[2]};
        T a = capsule_sdf(par, qa), b = capsule_sdf(par, qb);
        capsule_normal(par, a <= b ? qa : qb, n);
        return;
    }
    case SHAPE_CAPSULE: capsule_normal(par, p, n); return;
    case SHAPE_CYLINDER: {                                // primitives.py:169-183
        T l = len14(p[0], p[2]);
        T d0 = l - par[0], d1 = t_abs(p[1]) - par[1];
        T f = d0 > d1 ? T(1) : T(0);
        T ins = t_max(d0, d1) <= T(0) ? T(1) : T(0);
        T n20 = t_max(d0, T(0)) + ins * f, n21 = t_max(d1, T(0)) + ins * (T(1) - f);
        T inv2 = T(1) / len14(n20, n21);
        n20 *= inv2; n21 *= inv2;
        T sgn = p[1] >= T(0) ? T(1) : T(-1);
        T a = p[0] / l * n20, b = n21 * sgn, c = p[2] / l * n20;
        T inv = T(1) / len14(a, b, c);
        n[0] = a * inv; n[1] = b * inv; n[2] = c * inv;
        return;
    }
    case SHAPE_TORUS: {                                   // primitives.py:204-213
        T l = len14(p[0], p[2]);
        T q0 = l - par[0], q1 = p[1];
        T invq = T(1) / len14(q0, q1);
        T n0 = q0 * invq, n1 = q1 * invq;
        T a = p[0] / l * n0, b = n1, c = p[2] / l * n0;
        T inv = T(1) / len14(a, b, c);
        n[0] = a * inv; n[1] = b * inv; n[2] = c * inv;
        return;
    }
    case SHAPE_BOX: {                                     // primitives.py:240-251 (central differences, d = 1e-4)
        const T d = T(1e-4);
        T g[3];
        for (int i = 0; i < 3; ++i) {
            T pi[3] = {p[0], p[1], p[2]}, pd[3] = {p[0], p[1], p[2]};
            pi[i] += d; pd[i] -= d;
            g[i] = (T(0.5) / d) * (shape_sdf_local(shape, par, pi) - shape_sdf_local(shape, par, pd));
        }
        T inv = T(1) / len14(g[0], g[1], g[2]);
        n[0] = g[0] * inv; n[1] = g[1] * inv; n[2] = g[2] * inv;
        return;
    }
    default: n[0] = n[1] = n[2] = T(0);
    }
}

// gradient of the Box distance (primitives.py:232-238) w.r.t. p, with Taichi's max/min adjoint routing:
// g += scale * d sdf / d p
template <class T> PLB_HD void box_sdf_grad(const T* par, const T* p, T scale, T* g, int tf = 0) {
    T q[3] = {t_abs(p[0]) - par[0], t_abs(p[1]) - par[1], t_abs(p[2]) - par[2]};
    T e[3] = {t_max(q[0], T(0)), t_max(q[1], T(0)), t_max(q[2], T(0))};
    T Le = len14(e[0], e[1], e[2]);
    T qa[3];
    for (int i = 0; i < 3; ++i) qa[i] = max_to_lhs(q[i], T(0), tf) ? scale * e[i] / Le : T(0);     // max(q_i, 0): to q_i iff 0 < q_i
    T mm = t_max(q[1], q[2]);                        // max(q1, q2): to q1 iff q2 < q1, else to q2
    T m3 = t_max(q[0], mm);                          // max(q0, mm): to q0 iff mm < q0, else to mm
    if (min_to_lhs(m3, T(0), tf)) {                  // min(m3, 0): to m3 iff m3 < 0
        if (max_to_lhs(q[0], mm, tf)) qa[0] += scale;
        else if (max_to_lhs(q[1], q[2], tf)) qa[1] += scale;
        else qa[2] += scale;
    }
    for (int i = 0; i < 3; ++i) g[i] += qa[i] * (p[i] > T(0) ? T(1) : (p[i] < T(0) ? T(-1) : T(0)));
}

// Reverse mode of (shape_sdf_local, shape_normal_local) w.r.t. the local point p: given the adjoints
// da of the distance and na[3] of the normal, accumulate pa[3] += d<da*sdf + na.n>/dp.  Hand-derived for the
// shapes the reference's tasks actually move (Capsule: writer.yml, Torus: torus.yml); min/max follow Taichi's
// adjoint routing.  Returns false for shapes without a derived adjoint.  `ga` (Chopsticks only)
// accumulates the adjoint of the gap par[2].
template <class T> PLB_HD bool shape_local_adj(int shape, const T* par, const T* p, T da, const T* na, T* pa,
                                               double* ga = nullptr, int tf = 0) {
    switch (shape) {
    case SHAPE_CHOPSTICKS: {
        // sdf = ti.min(a, b): adjoint to a iff a < b, else to b.  normal: m = (a <= b) carries no gradient and
        // selects a's normal on ties.  Each branch is a Capsule evaluated at q = p + (0, h/2, 0) -/+ (gap/2, 0, 0).
        T qa[3] = {p[0] - par[2] / T(2), p[1] + par[0] / T(2), p[2]};
        T qb[3] = {p[0] + par[2] / T(2), qa[1], p[2]};
        T a = capsule_sdf(par, qa), b = capsule_sdf(par, qb);
        const T zero3[3] = {T(0), T(0), T(0)};
        const bool sdf_a = min_to_lhs(a, b, tf), nrm_a = a <= b;
        T ga_[3] = {T(0), T(0), T(0)}, gb_[3] = {T(0), T(0), T(0)};
        capsule_adj(par, qa, sdf_a ? da : T(0), nrm_a ? na : zero3, ga_, tf);
        capsule_adj(par, qb, sdf_a ? T(0) : da, nrm_a ? zero3 : na, gb_, tf);
        for (int i = 0; i < 3; ++i) pa[i] += ga_[i] + gb_[i];
        if (ga) *ga += (double)((gb_[0] - ga_[0]) / T(2));
        return true;
    }
    case SHAPE_CAPSULE: capsule_adj(par, p, da, na, pa, tf); return true;
    case SHAPE_TORUS: {
        T l = len14(p[0], p[2]);
        T q0 = l - par[0], q1 = p[1];
        T Lq = len14(q0, q1);
        T n20 = q0 / Lq, n21 = q1 / Lq;
        T x20 = p[0] / l, x21 = p[2] / l;
        T n3[3] = {x20 * n20, n21, x21 * n20};
        T L3 = len14(n3[0], n3[1], n3[2]);
        T n[3] = {n3[0] / L3, n3[1] / L3, n3[2] / L3};
        T nd = n[0] * na[0] + n[1] * na[1] + n[2] * na[2];
        T n3a[3];
        for (int i = 0; i < 3; ++i) n3a[i] = (na[i] - n[i] * nd) / L3;
        T x20a = n3a[0] * n20, x21a = n3a[2] * n20;
        T n20a = n3a[0] * x20 + n3a[2] * x21, n21a = n3a[1];
        // n2 = q / Lq (normal) and sdf = Lq - ty
        T n2d = n20 * n20a + n21 * n21a;
        T q0a = (n20a - n20 * n2d) / Lq + da * n20;
        T q1a = (n21a - n21 * n2d) / Lq + da * n21;
        T la = q0a;
        // x2 = (x, z) / l
        T xa = x20a / l, za = x21a / l;
        la -= (x20a * p[0] + x21a * p[2]) / (l * l);
        // l = len14(x, z)
        xa += la * p[0] / l; za += la * p[2] / l;
        pa[0] += xa; pa[1] += q1a; pa[2] += za;
        return true;
    }
    case SHAPE_CYLINDER: {                                // primitives.py:163-183 (h = radius, r = half height)
        const T l = len14(p[0], p[2]);
        const T d0 = l - par[0], d1 = t_abs(p[1]) - par[1];
        // ---- normal
        const T f = d0 > d1 ? T(1) : T(0);
        const T ins = t_max(d0, d1) <= T(0) ? T(1) : T(0);
        const T m0 = t_max(d0, T(0)) + ins * f, m1 = t_max(d1, T(0)) + ins * (T(1) - f);
        const T L2 = len14(m0, m1);
        const T n20 = m0 / L2, n21 = m1 / L2;
        const T sgn = p[1] >= T(0) ? T(1) : T(-1);
        const T x20 = p[0] / l, x21 = p[2] / l;
        const T u[3] = {x20 * n20, n21 * sgn, x21 * n20};
        const T L3 = len14(u[0], u[1], u[2]);
        const T n[3] = {u[0] / L3, u[1] / L3, u[2] / L3};
        const T nd = n[0] * na[0] + n[1] * na[1] + n[2] * na[2];
        T ua[3];
        for (int i = 0; i < 3; ++i) ua[i] = (na[i] - n[i] * nd) / L3;
        const T x20a = ua[0] * n20, x21a = ua[2] * n20;
        const T n20a = ua[0] * x20 + ua[2] * x21, n21a = ua[1] * sgn;
        const T n2d = n20 * n20a + n21 * n21a;
        const T m0a = (n20a - n20 * n2d) / L2, m1a = (n21a - n21 * n2d) / L2;
        T d0a = max_to_lhs(d0, T(0), tf) ? m0a : T(0), d1a = max_to_lhs(d1, T(0), tf) ? m1a : T(0);           // max(d, 0): to d iff 0 < d
        T xa = x20a / l, za = x21a / l;
        T la = -(x20a * p[0] + x21a * p[2]) / (l * l);
        // ---- distance: min(max(d0, d1), 0) + len14(max(d0, 0), max(d1, 0))
        const T e0 = t_max(d0, T(0)), e1 = t_max(d1, T(0));
        const T Le = len14(e0, e1);
        if (max_to_lhs(d0, T(0), tf)) d0a += da * e0 / Le;
        if (max_to_lhs(d1, T(0), tf)) d1a += da * e1 / Le;
        if (min_to_lhs(t_max(d0, d1), T(0), tf)) { if (max_to_lhs(d0, d1, tf)) d0a += da; else d1a += da; }       // min(mx, 0), max(d0, d1)
        la += d0a;                                        // d0 = |l| - h, l > 0
        xa += la * p[0] / l; za += la * p[2] / l;
        pa[0] += xa; pa[2] += za;
        pa[1] += d1a * (p[1] > T(0) ? T(1) : (p[1] < T(0) ? T(-1) : T(0)));
        return true;
    }
    case SHAPE_BOX: {                                     // primitives.py:232-251: normal by central differences, d = 1e-4
        const T d = T(1e-4);
        T g[3], pp[3], pm[3];
        for (int i = 0; i < 3; ++i) {
            for (int k = 0; k < 3; ++k) { pp[k] = p[k]; pm[k] = p[k]; }
            pp[i] += d; pm[i] -= d;
            g[i] = (T(0.5) / d) * (shape_sdf_local(shape, par, pp) - shape_sdf_local(shape, par, pm));
        }
        const T Lg = len14(g[0], g[1], g[2]);
        const T n[3] = {g[0] / Lg, g[1] / Lg, g[2] / Lg};
        const T nd = n[0] * na[0] + n[1] * na[1] + n[2] * na[2];
        for (int i = 0; i < 3; ++i) {
            const T gia = (na[i] - n[i] * nd) / Lg * (T(0.5) / d);
            for (int k = 0; k < 3; ++k) { pp[k] = p[k]; pm[k] = p[k]; }
            pp[i] += d; pm[i] -= d;
            box_sdf_grad(par, pp, gia, pa, tf);
            box_sdf_grad(par, pm, -gia, pa, tf);
        }
        box_sdf_grad(par, p, da, pa, tf);
        return true;
    }
    default: return false;
    }
}

// world-frame signed distance / normal (Primitive.sdf / normal; Sphere overrides ignore rotation)
template <class T> PLB_HD double prim_sdf(const PrimT<T>& pr, const double* gp) {
    if (pr.shape == SHAPE_SPHERE)
        return len14(gp[0] - pr.pos[0], gp[1] - pr.pos[1], gp[2] - pr.pos[2]) - pr.par[0];
    double loc[3], iq[4];
    inv_trans(gp, pr.pos, pr.rot, loc, iq);
    return shape_sdf_local(pr.shape, pr.par, loc);
}
template <class T> PLB_HD void prim_normal(const PrimT<T>& pr, const double* gp, double* D) {
    if (pr.shape == SHAPE_SPHERE) {
        double d[3] = {gp[0] - pr.pos[0], gp[1] - pr.pos[1], gp[2] - pr.pos[2]};
        double inv = 1.0 / len14(d[0], d[1], d[2]);
        D[0] = d[0] * inv; D[1] = d[1] * inv; D[2] = d[2] * inv;
        return;
    }
    double loc[3], iq[4], nl[3];
    inv_trans(gp, pr.pos, pr.rot, loc, iq);
    shape_normal_local(pr.shape, pr.par, loc, nl);
    qrot(pr.rot, nl, D);
}

// ------------------------------------------------------------------ collide (primive_base.py:82-115)
template <class T> struct CollideTmp {
    T infl, ex, D[3], cv[3], iv[3], nc, gvt[3], gn, e, flag;
    double dist, rel[3], iq[4];
};

// radius of a sphere centred on the primitive position that contains the whole shape
PLB_HD float prim_bounding_radius(int shape, const double* par) {
    switch (shape) {
    case SHAPE_SPHERE: return (float)par[0];
    case SHAPE_CAPSULE: return (float)(par[0] * 0.5 + par[1]);
    case SHAPE_CHOPSTICKS: return (float)(sqrt(par[0] * par[0] + 0.25 * par[2] * par[2]) + par[1]);
    case SHAPE_CYLINDER: return (float)sqrt(par[0] * par[0] + par[1] * par[1]);
    case SHAPE_TORUS: return (float)(par[0] + par[1]);
    default: return (float)sqrt(par[0] * par[0] + par[1] * par[1] + par[2] * par[2]);
    }
}

// conservative fp32 cull: sdf(p) >= |p - pos| - rb, and contact needs sdf <= max(0, ln(10)/softness).
// Nodes that fail this test take exactly the branch the full evaluation would take (no contact);
// the 1e-3 margin dwarfs fp32 rounding, so results are unchanged.  Skips the double-precision geometry
// for the ~98 % of active nodes that are nowhere near a manipulator.
template <class T> PLB_HD bool prim_within_reach(const PrimT<T>& pr, T softness, const double* gp) {
    float dx = (float)(gp[0] - pr.pos[0]), dy = (float)(gp[1] - pr.pos[1]), dz = (float)(gp[2] - pr.pos[2]);
    float reach = pr.rb + (softness > T(0) ? 2.302585093f / (float)softness : 0.0f) + 1e-3f;
    return !(dx * dx + dy * dy + dz * dz > reach * reach);
}
// node I passes that cull for at least one primitive
template <class T> PLB_HD bool node_near_any(const SimP<T>& P, const int* I, int nprim, const PrimT<T>* prims) {
    const double inv_n = 1.0 / (double)P.n;
    const double gp[3] = {I[0] * inv_n, I[1] * inv_n, I[2] * inv_n};
    bool near = false;
    for (int p = 0; p < nprim; ++p) near |= prim_within_reach(prims[p], P.softness, gp);
    return near;
}
template <class T> PLB_HD bool collide_eval(const PrimT<T>& pr, T softness, T dt, const double* gp, const T* v,
                                            CollideTmp<T>& c, T* vnew) {
    if (!prim_within_reach(pr, softness, gp)) return false;
    c.dist = prim_sdf(pr, gp);
    T ex = t_exp((T)(-c.dist * (double)softness));
    c.ex = ex;
    c.infl = ex < T(1) ? ex : T(1);
    if (!((softness > T(0) && c.infl > T(0.1)) || c.dist <= 0.0)) return false;
    double Dd[3], np[3];
    prim_normal(pr, gp, Dd);
    inv_trans(gp, pr.pos, pr.rot, c.rel, c.iq);                   // collider_v :82-89
    qrot(pr.rot1, c.rel, np);
    const double inv_dt = 1.0 / (double)dt;               // one reciprocal instead of three double-precision divisions
    for (int i = 0; i < 3; ++i) {
        c.D[i] = (T)Dd[i];
        c.cv[i] = (T)((np[i] + pr.pos1[i] - gp[i]) * inv_dt);
        c.iv[i] = v[i] - c.cv[i];
    }
    c.nc = dot3(c.iv, c.D);
    T mn = c.nc < T(0) ? c.nc : T(0);
    for (int i = 0; i < 3; ++i) c.gvt[i] = c.iv[i] - mn * c.D[i];
    T g2 = dot3(c.gvt, c.gvt);
    c.gn = t_sqrt(g2 + T(1e-8));
    c.e = c.gn + c.nc * pr.friction;
    T m = c.e < T(0) ? T(0) : c.e;                                // max(0, e)
    c.flag = (c.nc < T(0) && t_sqrt(g2) > T(1e-30)) ? T(1) : T(0);
    for (int i = 0; i < 3; ++i) {
        T fr = c.gvt[i] / c.gn * m;
        T g2v = fr * c.flag + c.gvt[i] * (T(1) - c.flag);
        vnew[i] = c.cv[i] + c.iv[i] * (T(1) - c.infl) + g2v * c.infl;
    }
    return true;
}

// adjoint of one collide.  v: velocity before this collide; vn_a: adjoint of its output.
// Writes v_a; accumulates pose adjoints into pa (only for movable Spheres -- other movable
// shapes need d sdf/d pose, d normal/d pose, which are not derived yet).
template <class T> PLB_HD void collide_grad_from(const PrimT<T>& pr, T softness, T dt, const double* gp,
                                                 const CollideTmp<T>& c, const T* vn_a, T* v_a, PoseAdj<T>* pa, int tf = 0);
template <class T> PLB_HD bool collide_grad(const PrimT<T>& pr, T softness, T dt, const double* gp, const T* v,
                                            const T* vn_a, T* v_a, PoseAdj<T>* pa, int tf = 0) {
    CollideTmp<T> c;
    T vnew[3];
    if (!collide_eval(pr, softness, dt, gp, v, c, vnew)) {
        for (int i = 0; i < 3; ++i) v_a[i] = vn_a[i];
        return false;
    }
    collide_grad_from(pr, softness, dt, gp, c, vn_a, v_a, pa, tf);
    return true;
}
// the adjoint proper, from the intermediates `c` of a collide_eval that hit
template <class T> PLB_HD void collide_grad_from(const PrimT<T>& pr, T softness, T dt, const double* gp,
                                                 const CollideTmp<T>& c, const T* vn_a, T* v_a, PoseAdj<T>* pa, int tf) {
    T m = c.e < T(0) ? T(0) : c.e;
    T cva[3], iva[3], g2a[3], gvta[3], Da[3] = {T(0), T(0), T(0)};
    T infla = T(0);
    for (int i = 0; i < 3; ++i) {
        T fr = c.gvt[i] / c.gn * m;
        T g2v = fr * c.flag + c.gvt[i] * (T(1) - c.flag);
        cva[i] = vn_a[i];
        iva[i] = (T(1) - c.infl) * vn_a[i];
        g2a[i] = c.infl * vn_a[i];
        infla += vn_a[i] * (g2v - c.iv[i]);
    }
    // g2v = fric*flag + gvt*(1-flag); fric = gvt * (m / gn)
    T fra_dot_gvt = T(0);
    for (int i = 0; i < 3; ++i) {
        T fra = c.flag * g2a[i];
        gvta[i] = (T(1) - c.flag) * g2a[i] + fra * (m / c.gn);
        fra_dot_gvt += fra * c.gvt[i];
    }
    T gna = -fra_dot_gvt * m / (c.gn * c.gn);
    T ma = fra_dot_gvt / c.gn;
    T ea = max_to_lhs(T(0), c.e, tf) ? T(0) : ma;            // max(0, e): e is the second operand -- it gets the adjoint unless 0 wins
    gna += ea;
    T nca = pr.friction * ea;
    for (int i = 0; i < 3; ++i) gvta[i] += gna * c.gvt[i] / c.gn;
    // gvt = iv - min(nc,0) D
    T mn = c.nc < T(0) ? c.nc : T(0);
    T mna = T(0);
    for (int i = 0; i < 3; ++i) { iva[i] += gvta[i]; mna -= gvta[i] * c.D[i]; Da[i] -= mn * gvta[i]; }
    if (min_to_lhs(c.nc, T(0), tf)) nca += mna;              // min(nc, 0)
    for (int i = 0; i < 3; ++i) { iva[i] += nca * c.D[i]; Da[i] += nca * c.iv[i]; }
    for (int i = 0; i < 3; ++i) { v_a[i] = iva[i]; cva[i] -= iva[i]; }
    if (!pa || !pr.movable) return;
    // ---- pose adjoints
    // influence = min(exp(-dist*soft), 1): adjoint to the exp iff exp < 1
    double dista = 0.0;
    if (min_to_lhs(c.ex, T(1), tf)) dista = -(double)softness * (double)c.infl * (double)infla;     // infl = min(exp(..), 1)
    // collider velocity
    const double inv_dt = 1.0 / (double)dt;
    double npa[3] = {(double)cva[0] * inv_dt, (double)cva[1] * inv_dt, (double)cva[2] * inv_dt};
    for (int i = 0; i < 3; ++i) pa->pos1[i] += npa[i];
    qrot_adj_q(pr.rot1, c.rel, npa, pa->rot1);
    double cr1[4], rela[3];
    qconj(pr.rot1, cr1);
    qrot(cr1, npa, rela);
    inv_trans_adj(gp, pr.pos, pr.rot, c.iq, rela, pa->pos, pa->rot);
    if (pr.shape == SHAPE_SPHERE) {
        // dist = len14(gp - c) - r ; D = (gp - c)/len14
        double d[3] = {gp[0] - pr.pos[0], gp[1] - pr.pos[1], gp[2] - pr.pos[2]};
        double iL = 1.0 / len14(d[0], d[1], d[2]);
        double Dd[3] = {d[0] * iL, d[1] * iL, d[2] * iL};
        double Dad[3] = {(double)Da[0], (double)Da[1], (double)Da[2]};
        double dD = dot3(Dad, Dd);
        for (int i = 0; i < 3; ++i) {
            double da = dista * Dd[i] + (Dad[i] - Dd[i] * dD) * iL;     // adjoint of d = gp - c
            pa->pos[i] -= da;
        }
    } else {
        // dist = sdf_local(loc), D = qrot(rot, n_local(loc)), loc = inv_trans(gp, pos, rot)
        double Dad[3] = {(double)Da[0], (double)Da[1], (double)Da[2]};
        double nl[3], cr[4], nla[3], loca[3] = {0.0, 0.0, 0.0};
        shape_normal_local(pr.shape, pr.par, c.rel, nl);
        qrot_adj_q(pr.rot, nl, Dad, pa->rot);            // d D / d rot (direct)
        qconj(pr.rot, cr);
        qrot(cr, Dad, nla);                              // adjoint of n_local (qrot is linear in its vector argument)
        shape_local_adj(pr.shape, pr.par, c.rel, dista, nla, loca, &pa->gap, tf);
        inv_trans_adj(gp, pr.pos, pr.rot, c.iq, loca, pa->pos, pa->rot);
    }
}

// ------------------------------------------------------------------ box boundary (mpm_simulator.py:200-219)
// one axis d; v updated in place.
template <class T> PLB_HD void boundary_axis(const SimP<T>& P, const int* I, int d, T* v) {
    if (I[d] < 3 && v[d] < T(0)) {
        if (d != 1 || P.ground_friction == T(0)) v[d] = T(0);
        else if (P.ground_friction < T(10)) {
            T lin = v[1];
            T lit = t_sqrt(v[0] * v[0] + v[2] * v[2] + T(1e-8));
            T s = t_max(T(1) + P.ground_friction * lin / lit, T(0));
            v[0] *= s; v[2] *= s; v[1] = T(0);
        } else v[0] = v[1] = v[2] = T(0);
    }
    if (I[d] > P.n - 3 && v[d] > T(0)) v[d] = T(0);
}
// adjoint: vb = value before the axis stage, a = adjoint (in: of output, out: of input)
template <class T> PLB_HD void boundary_axis_grad(const SimP<T>& P, const int* I, int d, const T* vb, T* a) {
    T vm[3] = {vb[0], vb[1], vb[2]};
    bool lo = I[d] < 3 && vb[d] < T(0);
    T s = T(1), lit = T(1), e = T(1);
    int mode = 0;
    if (lo) {
        if (d != 1 || P.ground_friction == T(0)) { mode = 1; vm[d] = T(0); }
        else if (P.ground_friction < T(10)) {
            mode = 2;
            lit = t_sqrt(vb[0] * vb[0] + vb[2] * vb[2] + T(1e-8));
            e = T(1) + P.ground_friction * vb[1] / lit;
            s = t_max(e, T(0));
            vm[0] *= s; vm[2] *= s; vm[1] = T(0);
        } else { mode = 3; vm[0] = vm[1] = vm[2] = T(0); }
    }
    if (I[d] > P.n - 3 && vm[d] > T(0)) a[d] = T(0);
    if (mode == 1) a[d] = T(0);
    else if (mode == 3) a[0] = a[1] = a[2] = T(0);
    else if (mode == 2) {
        T sa = a[0] * vb[0] + a[2] * vb[2];
        T ax = s * a[0], az = s * a[2];
        T ea = max_to_lhs(e, T(0), P.tie_first) ? sa : T(0);            // max(e, 0): adjoint to e iff 0 < e
        T lina = P.ground_friction * ea / lit;
        T lita = -P.ground_friction * vb[1] * ea / (lit * lit);
        ax += lita * vb[0] / lit;
        az += lita * vb[2] / lit;
        a[0] = ax; a[1] = lina; a[2] = az;
    }
}

// ------------------------------------------------------------------ grid_op for one node
// m, mv: grid_m / grid_v_in.  Returns false (and vout = 0) when the node is empty (m <= 1e-12).
// touch (optional): set when the node is in contact with a movable primitive (its pose adjoints are due in the reverse pass).
template <class T>
PLB_HD bool grid_node_fwd(const SimP<T>& P, const int* I, T m, const T* mv, int nprim, const PrimT<T>* prims, T* vout,
                          bool* touch = nullptr) {
    if (!(m > T(1e-12))) { vout[0] = vout[1] = vout[2] = T(0); return false; }
    T inv = T(1) / m;
    T v[3] = {inv * mv[0] + P.grav[0], inv * mv[1] + P.grav[1], inv * mv[2] + P.grav[2]};
    // grid_pos = I * dx (mpm_simulator.py:197); one reciprocal (loop-invariant: the compiler hoists it out of a caller's node
    // loop) instead of three double-precision divisions per node -- ~150 instructions each time a node is evaluated
    const double inv_n = 1.0 / (double)P.n;
    double gp[3] = {I[0] * inv_n, I[1] * inv_n, I[2] * inv_n};
    for (int p = 0; p < nprim; ++p) {
        CollideTmp<T> c;
        T vn[3];
        if (collide_eval(prims[p], P.softness, P.dt, gp, v, c, vn)) {
            v[0] = vn[0]; v[1] = vn[1]; v[2] = vn[2];
            if (touch && prims[p].movable) *touch = true;
        }
    }
    for (int d = 0; d < 3; ++d) boundary_axis(P, I, d, v);
    vout[0] = v[0]; vout[1] = v[1]; vout[2] = v[2];
    return true;
}

// grid_op adjoint for one node.  vout_a: adjoint of grid_v_out.  Outputs m_a, mv_a.  sink(p, PoseAdj, hit) is
// called for EVERY primitive by EVERY caller (hit = false, zero adjoint when the node does not touch it), so a
// GPU sink may use wave-wide collectives.
// POSE = false leaves the pose adjoints out (sink still gets `hit`): the GPU path computes them off the critical
// path, for the few blocks that touch a manipulator.
template <class T, bool POSE = true, class Sink>
PLB_HD void grid_node_bwd(const SimP<T>& P, const int* I, T m, const T* mv, int nprim, const PrimT<T>* prims,
                          const T* vout_a, T* m_a, T* mv_a, Sink&& sink) {
    *m_a = T(0); mv_a[0] = mv_a[1] = mv_a[2] = T(0);
    // Control flow is kept convergent around sink(): every caller walks the same primitive loop and calls it
    // once per primitive, the node-specific work sits in `if (live)` blocks in between.
    const bool live = m > T(1e-12);
    const T inv = live ? T(1) / m : T(0);
    T v0[3] = {inv * mv[0] + P.grav[0], inv * mv[1] + P.grav[1], inv * mv[2] + P.grav[2]};
    // grid_pos = I * dx (mpm_simulator.py:197); one reciprocal (loop-invariant: the compiler hoists it out of a caller's node
    // loop) instead of three double-precision divisions per node -- ~150 instructions each time a node is evaluated
    const double inv_n = 1.0 / (double)P.n;
    double gp[3] = {I[0] * inv_n, I[1] * inv_n, I[2] * inv_n};
    T a[3] = {T(0), T(0), T(0)};
    // the LAST primitive the node touches keeps its collide intermediates from the forward sweep, so the common case
    // (a node in contact with one manipulator) evaluates the double-precision geometry once, not three times
    // (not in the POSE pass, which runs off the critical path and is short of registers)
    constexpr bool KEEP = !POSE;
    int last = -1;
    CollideTmp<T> cl;
    if (live) {
        // forward to the state after all collides
        T vc[3] = {v0[0], v0[1], v0[2]};
        for (int p = 0; p < nprim; ++p) {
            CollideTmp<T> c;
            T vn[3];
            if (collide_eval(prims[p], P.softness, P.dt, gp, vc, c, vn)) {
                vc[0] = vn[0]; vc[1] = vn[1]; vc[2] = vn[2];
                if (KEEP) { cl = c; last = p; }
            }
        }
        // boundary stages
        T vb0[3] = {vc[0], vc[1], vc[2]}, vb1[3], vb2[3];
        T t[3] = {vc[0], vc[1], vc[2]};
        boundary_axis(P, I, 0, t); vb1[0] = t[0]; vb1[1] = t[1]; vb1[2] = t[2];
        boundary_axis(P, I, 1, t); vb2[0] = t[0]; vb2[1] = t[1]; vb2[2] = t[2];
        a[0] = vout_a[0]; a[1] = vout_a[1]; a[2] = vout_a[2];
        boundary_axis_grad(P, I, 2, vb2, a);
        boundary_axis_grad(P, I, 1, vb1, a);
        boundary_axis_grad(P, I, 0, vb0, a);
    }
    // collides in reverse; primitives after `last` did not touch the node, `last` reuses its intermediates, and for
    // the ones before it (a node between two manipulators) the velocity entering the collide is recomputed from v0
    for (int p = nprim - 1; p >= 0; --p) {
        PoseAdj<T> pa;
        pa.zero();
        bool hit = false;
        CollideTmp<T> c;
        if (KEEP && live && p == last) { c = cl; hit = true; }
        else if (live && (!KEEP || p < last)) {
            T vin[3] = {v0[0], v0[1], v0[2]}, vn[3];
            for (int q = 0; q < p; ++q) {
                CollideTmp<T> cq;
                if (collide_eval(prims[q], P.softness, P.dt, gp, vin, cq, vn)) { vin[0] = vn[0]; vin[1] = vn[1]; vin[2] = vn[2]; }
            }
            hit = collide_eval(prims[p], P.softness, P.dt, gp, vin, c, vn);
        }
        if (hit) {
            T va[3];
            collide_grad_from(prims[p], P.softness, P.dt, gp, c, a, va, POSE ? &pa : nullptr, P.tie_first);
            a[0] = va[0]; a[1] = va[1]; a[2] = va[2];
        }
        sink(p, pa, hit && prims[p].movable);
    }
    // v0 = mv / m + g
    if (live)
        for (int i = 0; i < 3; ++i) { mv_a[i] = a[i] * inv; *m_a -= a[i] * mv[i] * inv * inv; }
}

// ------------------------------------------------------------------ primitive kinematics (always double)
// forward_kinematics (primive_base.py:117-121): pos' = clamp(pos + v), rot' = normalize(w2quat(w) (x) rot)
PLB_HD void w2quat_d(const double* a, double* q) {          // utils.py:29-41
    double w = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    q[0] = 1.0; q[1] = q[2] = q[3] = 0.0;
    if (w > 1e-9) {
        double s = sin(w / 2) / w;
        q[0] = cos(w / 2); q[1] = a[0] * s; q[2] = a[1] * s; q[3] = a[2] * s;
    }
}
PLB_HD void qmul_raw_d(const double* q, const double* r, double* o) {   // utils.py:19-26 (Hamilton q (x) r)
    o[0] = r[0] * q[0] - r[1] * q[1] - r[2] * q[2] - r[3] * q[3];
    o[1] = r[0] * q[1] + r[1] * q[0] - r[2] * q[3] + r[3] * q[2];
    o[2] = r[0] * q[2] + r[1] * q[3] + r[2] * q[0] - r[3] * q[1];
    o[3] = r[0] * q[3] - r[1] * q[2] + r[2] * q[1] + r[3] * q[0];
}
PLB_HD void fk_fwd_d(const double* pos, const double* rot, const double* v, const double* w,
                     const double* lo, const double* hi, double* pos1, double* rot1) {
    for (int i = 0; i < 3; ++i) {
        double y = pos[i] + v[i];
        double mn = y < hi[i] ? y : hi[i];
        pos1[i] = lo[i] < mn ? mn : lo[i];
    }
    double q[4], o[4];
    w2quat_d(w, q);
    qmul_raw_d(q, rot, o);
    double inv = 1.0 / sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
    for (int i = 0; i < 4; ++i) rot1[i] = o[i] * inv;
}
// o = normalize(q (x) r):  given o_a, accumulate q_a, r_a
PLB_HD void qmul_adj_d(const double* q, const double* r, const double* o_a, double* q_a, double* r_a) {
    double o[4];
    qmul_raw_d(q, r, o);
    // one reciprocal instead of eight divisions: this runs in the serial double-precision chain of k_fk_chain_grad, where
    // every division is ~40 dependent instructions on a single lane
    const double inv = 1.0 / sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
    double on[4], dotv = 0;
    for (int i = 0; i < 4; ++i) { on[i] = o[i] * inv; dotv += on[i] * o_a[i]; }
    double oa[4];
    for (int i = 0; i < 4; ++i) oa[i] = (o_a[i] - on[i] * dotv) * inv;
    r_a[0] += oa[0] * q[0] + oa[1] * q[1] + oa[2] * q[2] + oa[3] * q[3];
    r_a[1] += -oa[0] * q[1] + oa[1] * q[0] + oa[2] * q[3] - oa[3] * q[2];
    r_a[2] += -oa[0] * q[2] - oa[1] * q[3] + oa[2] * q[0] + oa[3] * q[1];
    r_a[3] += -oa[0] * q[3] + oa[1] * q[2] - oa[2] * q[1] + oa[3] * q[0];
    q_a[0] += oa[0] * r[0] + oa[1] * r[1] + oa[2] * r[2] + oa[3] * r[3];
    q_a[1] += -oa[0] * r[1] + oa[1] * r[0] - oa[2] * r[3] + oa[3] * r[2];
    q_a[2] += -oa[0] * r[2] + oa[1] * r[3] + oa[2] * r[0] - oa[3] * r[1];
    q_a[3] += -oa[0] * r[3] - oa[1] * r[2] + oa[2] * r[1] + oa[3] * r[0];
}
PLB_HD void qmul_d(const double* q, const double* r, double* o) {
    qmul_raw_d(q, r, o);
    double inv = 1.0 / sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
    for (int i = 0; i < 4; ++i) o[i] *= inv;
}
// q = w2quat(a): given q_a, accumulate a_a
PLB_HD void w2quat_adj_d(const double* a, const double* qa, double* a_a) {
    double th = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (th > 1e-9) {
        double n[3] = {a[0] / th, a[1] / th, a[2] / th};
        double sh = sin(th / 2), ch = cos(th / 2);
        double nq = n[0] * qa[1] + n[1] * qa[2] + n[2] * qa[3];
        for (int i = 0; i < 3; ++i)
            a_a[i] += -0.5 * sh * n[i] * qa[0] + (sh / th) * (qa[1 + i] - n[i] * nq) + 0.5 * ch * n[i] * nq;
    }
}

// adjoint: pos1_a, rot1_a in; accumulates pos_a, rot_a (pose at f); writes v_a, w_a (overwrite)
PLB_HD void fk_bwd_d(const double* pos, const double* rot, const double* v, const double* w,
                     const double* lo, const double* hi, const double* pos1_a, const double* rot1_a,
                     double* pos_a, double* rot_a, double* v_a, double* w_a, int tf = 0) {
    for (int i = 0; i < 3; ++i) {
        double y = pos[i] + v[i];
        double mn = y < hi[i] ? y : hi[i];
        // max(min(y,hi),lo): to y iff y < hi and lo < min
        double gate = (min_to_lhs(y, hi[i], tf) && max_to_lhs(mn, lo[i], tf)) ? 1.0 : 0.0;
        pos_a[i] += gate * pos1_a[i];
        v_a[i] = gate * pos1_a[i];
    }
    double q[4], qa[4] = {0, 0, 0, 0};
    w2quat_d(w, q);
    qmul_adj_d(q, rot, rot1_a, qa, rot_a);
    w_a[0] = w_a[1] = w_a[2] = 0.0;
    w2quat_adj_d(w, qa, w_a);
}

// Chopsticks.forward_kinematics (primitives.py:94-98): gap closes by gap_vel down to minimal_gap, position as the
// base class, rotation updated in the BODY frame: rot[f+1] = qmul(rot[f], w2quat(w)).
PLB_HD void fk_chopsticks_fwd_d(const double* pos, const double* rot, const double* v, const double* w, double gap,
                                double gap_vel, double min_gap, const double* lo, const double* hi, double* pos1,
                                double* rot1, double* gap1) {
    for (int i = 0; i < 3; ++i) {
        double y = pos[i] + v[i];
        double mn = y < hi[i] ? y : hi[i];
        pos1[i] = lo[i] < mn ? mn : lo[i];
    }
    double g = gap - gap_vel;
    *gap1 = min_gap < g ? g : min_gap;
    double q[4];
    w2quat_d(w, q);
    qmul_d(rot, q, rot1);
}
PLB_HD void fk_chopsticks_bwd_d(const double* pos, const double* rot, const double* v, const double* w, double gap,
                                double gap_vel, double min_gap, const double* lo, const double* hi,
                                const double* pos1_a, const double* rot1_a, double gap1_a, double* pos_a,
                                double* rot_a, double* gap_a, double* v_a, double* w_a, double* gap_vel_a, int tf = 0) {
    for (int i = 0; i < 3; ++i) {
        double y = pos[i] + v[i];
        double mn = y < hi[i] ? y : hi[i];
        double gate = (min_to_lhs(y, hi[i], tf) && max_to_lhs(mn, lo[i], tf)) ? 1.0 : 0.0;
        pos_a[i] += gate * pos1_a[i];
        v_a[i] = gate * pos1_a[i];
    }
    double ggate = max_to_lhs(gap - gap_vel, min_gap, tf) ? 1.0 : 0.0;      // max(lhs, rhs): adjoint to lhs iff rhs < lhs
    *gap_a += ggate * gap1_a;
    *gap_vel_a = -ggate * gap1_a;
    double q[4], qa[4] = {0, 0, 0, 0};
    w2quat_d(w, q);
    qmul_adj_d(rot, q, rot1_a, rot_a, qa);
    w_a[0] = w_a[1] = w_a[2] = 0.0;
    w2quat_adj_d(w, qa, w_a);
}

// RollingPin.forward_kinematics (primitives.py:66-80): v = (roll about own axis, turn about world y, move in y)
PLB_HD void fk_rollingpin_fwd_d(const double* pos, const double* rot, const double* v, const double* lo,
                                const double* hi, double* pos1, double* rot1) {
    const double dw = v[0], dth = v[1], dy = v[2];
    const double e[3] = {0.0, -1.0, 0.0};
    double yd[3];
    qrot(rot, e, yd);
    const double xd[3] = {0.03 * dw * yd[2], dy, -0.03 * dw * yd[0]};     // ((0,1,0) x y_dir) * dw * 0.03, y := dy
    for (int i = 0; i < 3; ++i) {
        double y = pos[i] + xd[i];
        double mn = y < hi[i] ? y : hi[i];
        pos1[i] = lo[i] < mn ? mn : lo[i];
    }
    const double a1[3] = {0.0, dw, 0.0}, a2[3] = {0.0, -dth, 0.0};
    double q1[4], q2[4], inner[4];
    w2quat_d(a1, q1); w2quat_d(a2, q2);
    qmul_d(rot, q1, inner);
    qmul_d(q2, inner, rot1);
}
PLB_HD void fk_rollingpin_bwd_d(const double* pos, const double* rot, const double* v, const double* lo,
                                const double* hi, const double* pos1_a, const double* rot1_a,
                                double* pos_a, double* rot_a, double* v_a, int tf = 0) {
    const double dw = v[0], dth = v[1], dy = v[2];
    const double e[3] = {0.0, -1.0, 0.0};
    double yd[3];
    qrot(rot, e, yd);
    const double xd[3] = {0.03 * dw * yd[2], dy, -0.03 * dw * yd[0]};
    double xda[3];
    for (int i = 0; i < 3; ++i) {
        double y = pos[i] + xd[i];
        double mn = y < hi[i] ? y : hi[i];
        double gate = (min_to_lhs(y, hi[i], tf) && max_to_lhs(mn, lo[i], tf)) ? 1.0 : 0.0;
        pos_a[i] += gate * pos1_a[i];
        xda[i] = gate * pos1_a[i];
    }
    double dwa = 0.03 * (yd[2] * xda[0] - yd[0] * xda[2]);
    const double dya = xda[1];
    const double yda[3] = {-0.03 * dw * xda[2], 0.0, 0.03 * dw * xda[0]};
    qrot_adj_q(rot, e, yda, rot_a);
    const double a1[3] = {0.0, dw, 0.0}, a2[3] = {0.0, -dth, 0.0};
    double q1[4], q2[4], inner[4];
    w2quat_d(a1, q1); w2quat_d(a2, q2);
    qmul_d(rot, q1, inner);
    double q2a[4] = {0, 0, 0, 0}, inner_a[4] = {0, 0, 0, 0}, q1a[4] = {0, 0, 0, 0};
    qmul_adj_d(q2, inner, rot1_a, q2a, inner_a);
    qmul_adj_d(rot, q1, inner_a, rot_a, q1a);
    double a1a[3] = {0, 0, 0}, a2a[3] = {0, 0, 0};
    w2quat_adj_d(a1, q1a, a1a);
    w2quat_adj_d(a2, q2a, a2a);
    dwa += a1a[1];
    v_a[0] = dwa; v_a[1] = -a2a[1]; v_a[2] = dya;
}

}  // namespace plb
