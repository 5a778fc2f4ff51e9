/* CPU oracle (TEST INFRASTRUCTURE ONLY; parity unpinned -- see oracle/plb_oracle.py).
 *
 * Plain-C restatement of Loss.update_target / update_target_sdf,
 * /root/reference/plb/engine/losses/loss.py:81-106: 2*n Jacobi sweeps in which
 * every non-solid node adopts, from the 6^3 window of offsets -3..2 around it
 * (ti.ndrange order, first axis slowest, strict '<'), the nearest-point record
 * of the neighbour that gives the smallest distance.  Sweeps stop early once a
 * whole sweep changes nothing (the remaining ones would be no-ops).
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC sdf_sweep.c -o libplb_oracle_c.so -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* returns number of sweeps executed */
int plb_oracle_target_sdf(const double *density, int n, double dx, double inf,
                          int max_sweeps, double *sdf_out, double *np_out)
{
    const size_t G = (size_t)n * n * n;
    double *sdf_c = (double *)malloc(G * sizeof(double));
    double *np_c = (double *)calloc(G * 3, sizeof(double));
    double *sdf = sdf_out, *npnt = np_out;
    for (size_t i = 0; i < G; ++i) sdf_c[i] = inf;
    memset(npnt, 0, G * 3 * sizeof(double));
    int sweep = 0;
    for (; sweep < max_sweeps; ++sweep) {
        int changed = 0;
#pragma omp parallel for collapse(2) reduction(| : changed)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                for (int k = 0; k < n; ++k) {
                    size_t I = ((size_t)i * n + j) * n + k;
                    double gx = i * dx, gy = j * dx, gz = k * dx;
                    double best = inf, bx = npnt[3 * I], by = npnt[3 * I + 1], bz = npnt[3 * I + 2];
                    if (density[I] > 1e-4) {
                        best = 0.0; bx = gx; by = gy; bz = gz;
                    } else {
                        for (int a = -3; a < 3; ++a)
                            for (int b = -3; b < 3; ++b)
                                for (int c = -3; c < 3; ++c) {
                                    int vi = i + a, vj = j + b, vk = k + c;
                                    if (vi < 0 || vj < 0 || vk < 0 || vi >= n || vj >= n || vk >= n) continue;
                                    if (a == 0 && b == 0 && c == 0) continue;
                                    size_t V = ((size_t)vi * n + vj) * n + vk;
                                    if (sdf_c[V] < inf) {
                                        double ddx = gx - np_c[3 * V], ddy = gy - np_c[3 * V + 1], ddz = gz - np_c[3 * V + 2];
                                        double dist = sqrt(ddx * ddx + ddy * ddy + ddz * ddz + 1e-8);
                                        if (dist < best) { best = dist; bx = np_c[3 * V]; by = np_c[3 * V + 1]; bz = np_c[3 * V + 2]; }
                                    }
                                }
                    }
                    if (best != sdf_c[I] || bx != np_c[3 * I] || by != np_c[3 * I + 1] || bz != np_c[3 * I + 2]) changed = 1;
                    sdf[I] = best; npnt[3 * I] = bx; npnt[3 * I + 1] = by; npnt[3 * I + 2] = bz;
                }
        memcpy(sdf_c, sdf, G * sizeof(double));
        memcpy(np_c, npnt, G * 3 * sizeof(double));
        if (!changed) { ++sweep; break; }
    }
    free(sdf_c); free(np_c);
    return sweep;
}
