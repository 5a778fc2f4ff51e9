// TEST INFRASTRUCTURE ONLY -- never loaded by plasticinelab_amd.
// The per-particle arithmetic of the HIP kernels (plasticinelab_amd/csrc/mpm_math.h) is written once for a "lane value" T:
// a scalar, or a pack of two particles per GPU lane (P2 / D2 / I2).  This program instantiates both on the host and
// checks that the pack computes, component by component, exactly what two scalar float evaluations compute (same
// operations in the same order; compiled without fp contraction): forward (p2g incl. return mapping and the nearly
// singular branch), the reverse gather + constitutive VJP, g2p and its adjoint.
#include "../../plasticinelab_amd/csrc/mpm_grid.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
using namespace plb;
// host check: the P2 instantiation must equal two scalar float evaluations bit for bit
static float rnd() { return (float)rand() / RAND_MAX * 2 - 1; }
int main() {
    SimP<float> P; P.n = 64; P.dx = 1.f/64; P.inv_dx = 64; P.dt = 1e-4f; P.p_mass = 1e-4f; P.kappa = -1e-4f*1e-6f*4*64*64; P.grav[0]=0;P.grav[1]=-3e-2f;P.grav[2]=0; P.x_hi = 1-3.f/64; P.ground_friction=0; P.svd_clamp=1e-6f; P.softness=666; P.tie_first=0;
    int bad = 0;
    for (int it = 0; it < 20000; ++it) {
        double x[2][3]; float v[2][3], C[2][9], E[2][9], mu[2], lam[2], ys[2];
        for (int s = 0; s < 2; ++s) {
            for (int d = 0; d < 3; ++d) { x[s][d] = 0.3 + 0.4 * (double)rand() / RAND_MAX; v[s][d] = rnd(); }
            float amp = (it % 4 == 0) ? 0.3f : 0.02f;
            for (int d = 0; d < 9; ++d) { C[s][d] = rnd() * 50; E[s][d] = rnd() * amp; }
            if (it % 97 == 0) { for (int d = 0; d < 9; ++d) E[s][d] = 0; E[s][0] = -0.9995f; }     // nearly singular
            mu[s] = 2083; lam[s] = 1388; ys[s] = (it % 3 == 0) ? 50.f : 1e9f;
        }
        // scalar
        float En[2][9]; int base[2][3]; float acc[2][4] = {{0}};
        for (int s = 0; s < 2; ++s)
            p2g_particle<float, double>(P, x[s], v[s], C[s], E[s], mu[s], lam[s], ys[s], En[s], base[s], [&](int i, int j, int l, float m, const float* mom) {
                float w = 1.f + i + 3 * j + 9 * l; acc[s][0] += w * m; for (int a = 0; a < 3; ++a) acc[s][1 + a] += w * mom[a]; });
        // packed
        D2 xp[3]; P2 vp[3], Cp[9], Ep[9], Enp[9]; I2 bp[3]; P2 accp[4] = {P2(0), P2(0), P2(0), P2(0)};
        for (int d = 0; d < 3; ++d) { xp[d] = D2(x[0][d], x[1][d]); vp[d] = P2(v[0][d], v[1][d]); }
        for (int d = 0; d < 9; ++d) { Cp[d] = P2(C[0][d], C[1][d]); Ep[d] = P2(E[0][d], E[1][d]); }
        p2g_particle<P2, D2>(P, xp, vp, Cp, Ep, P2(mu[0], mu[1]), P2(lam[0], lam[1]), P2(ys[0], ys[1]), Enp, bp, [&](int i, int j, int l, P2 m, const P2* mom) {
            float w = 1.f + i + 3 * j + 9 * l; accp[0] += P2(w) * m; for (int a = 0; a < 3; ++a) accp[1 + a] += P2(w) * mom[a]; });
        for (int d = 0; d < 9; ++d) if (memcmp(&En[0][d], &Enp[d], 0) || En[0][d] != Enp[d].lo() || En[1][d] != Enp[d].hi()) ++bad;
        for (int c = 0; c < 4; ++c) if (acc[0][c] != accp[c].lo() || acc[1][c] != accp[c].hi()) ++bad;
        for (int d = 0; d < 3; ++d) if (base[0][d] != bp[d].x || base[1][d] != bp[d].y) ++bad;
        // reverse: gather + finish
        P2GGather<float> G[2]; P2GGather<P2> Gp;
        float Ena[2][9], xa[2][3], va[2][3], Ca[2][9], Ea[2][9];
        for (int s = 0; s < 2; ++s) { for (int d = 0; d < 9; ++d) Ena[s][d] = rnd(); for (int d = 0; d < 3; ++d) xa[s][d] = rnd(); }
        auto field = [&](int s, int i, int j, int l, int c) { return (float)sin(1.0 + s + 2 * i + 3 * j + 5 * l + 7 * c); };
        for (int s = 0; s < 2; ++s) {
            p2g_gather_grad<float, double>(P, x[s], G[s], [&](int i, int j, int l, float* g) { for (int c = 0; c < 4; ++c) g[c] = field(s, i, j, l, c); });
            p2g_finish_grad<float>(P, G[s], v[s], C[s], E[s], mu[s], lam[s], ys[s], Ena[s], xa[s], va[s], Ca[s], Ea[s]);
        }
        P2 Enap[9], xap[3], vap[3], Cap[9], Eap[9];
        for (int d = 0; d < 9; ++d) Enap[d] = P2(Ena[0][d], Ena[1][d]);
        // xa was accumulated in place above: rebuild the inputs
        float xa0[2][3]; srand(it + 12345); 
        (void)xa0;
        p2g_gather_grad<P2, D2>(P, xp, Gp, [&](int i, int j, int l, P2* g) { for (int c = 0; c < 4; ++c) g[c] = P2(field(0, i, j, l, c), field(1, i, j, l, c)); });
        for (int d = 0; d < 3; ++d) xap[d] = P2(0.f);
        float xz[2][3] = {{0, 0, 0}, {0, 0, 0}}, va2[2][3], Ca2[2][9], Ea2[2][9];
        for (int s = 0; s < 2; ++s) p2g_finish_grad<float>(P, G[s], v[s], C[s], E[s], mu[s], lam[s], ys[s], Ena[s], xz[s], va2[s], Ca2[s], Ea2[s]);
        p2g_finish_grad<P2>(P, Gp, vp, Cp, Ep, P2(mu[0], mu[1]), P2(lam[0], lam[1]), P2(ys[0], ys[1]), Enap, xap, vap, Cap, Eap);
        for (int d = 0; d < 3; ++d) if (xz[0][d] != xap[d].lo() || xz[1][d] != xap[d].hi() || va2[0][d] != vap[d].lo() || va2[1][d] != vap[d].hi()) ++bad;
        for (int d = 0; d < 9; ++d) if (Ca2[0][d] != Cap[d].lo() || Ca2[1][d] != Cap[d].hi() || Ea2[0][d] != Eap[d].lo() || Ea2[1][d] != Eap[d].hi()) ++bad;
        // g2p and its adjoint
        double xn[2][3]; float vn[2][3], Cn[2][9]; D2 xnp[3]; P2 vnp[3], Cnp[9];
        for (int s = 0; s < 2; ++s) g2p_particle<float, double>(P, x[s], xn[s], vn[s], Cn[s], [&](int i, int j, int l, float* g) { for (int c = 0; c < 3; ++c) g[c] = field(s, i, j, l, c); });
        g2p_particle<P2, D2>(P, xp, xnp, vnp, Cnp, [&](int i, int j, int l, P2* g) { for (int c = 0; c < 3; ++c) g[c] = P2(field(0, i, j, l, c), field(1, i, j, l, c)); });
        for (int d = 0; d < 3; ++d) if (xn[0][d] != xnp[d].x || xn[1][d] != xnp[d].y || vn[0][d] != vnp[d].lo() || vn[1][d] != vnp[d].hi()) ++bad;
        for (int d = 0; d < 9; ++d) if (Cn[0][d] != Cnp[d].lo() || Cn[1][d] != Cnp[d].hi()) ++bad;
        float xga[2][3], ea[2][4] = {{0}}; P2 xgap[3], eap[4] = {P2(0), P2(0), P2(0), P2(0)};
        for (int s = 0; s < 2; ++s) g2p_particle_grad<float, double>(P, x[s], vn[s], xa[s], va2[s], Ca2[s], xga[s],
            [&](int i, int j, int l, float* g) { for (int c = 0; c < 3; ++c) g[c] = field(s, i, j, l, c); },
            [&](int i, int j, int l, const float* ga) { float w = 1.f + i + 3 * j + 9 * l; for (int c = 0; c < 3; ++c) ea[s][c] += w * ga[c]; });
        P2 xna[3]; for (int d = 0; d < 3; ++d) xna[d] = P2(xa[0][d], xa[1][d]);
        g2p_particle_grad<P2, D2>(P, xp, vnp, xna, vap, Cap, xgap,
            [&](int i, int j, int l, P2* g) { for (int c = 0; c < 3; ++c) g[c] = P2(field(0, i, j, l, c), field(1, i, j, l, c)); },
            [&](int i, int j, int l, const P2* ga) { float w = 1.f + i + 3 * j + 9 * l; for (int c = 0; c < 3; ++c) eap[c] += P2(w) * ga[c]; });
        for (int d = 0; d < 3; ++d) if (xga[0][d] != xgap[d].lo() || xga[1][d] != xgap[d].hi() || ea[0][d] != eap[d].lo() || ea[1][d] != eap[d].hi()) ++bad;
    }
    printf("mismatches: %d\n", bad);
    return bad != 0;
}
