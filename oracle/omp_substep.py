"""ctypes loader of oracle/libplb_oracle_omp.so (mpm_substep_omp.c): one MPM substep forward + reverse in C / OpenMP,
float64 -- the CPU baseline bench.py times on the GPU box's host cores.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (like everything under oracle/): nothing under plasticinelab_amd/ loads this.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAXP = 8


class Cfg(C.Structure):
    _fields_ = [("n", C.c_int), ("n_particles", C.c_int), ("n_prim", C.c_int), ("dx", C.c_double), ("inv_dx", C.c_double),
                ("dt", C.c_double), ("p_vol", C.c_double), ("p_mass", C.c_double), ("gravity", C.c_double * 3),
                ("ground_friction", C.c_double), ("softness", C.c_double), ("radius", C.c_double * MAXP), ("friction", C.c_double * MAXP)]


_lib = None


def load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.path.join(_HERE, "libplb_oracle_omp.so"))
        _lib.plb_omp_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OmpSubstep:
    """Holds the config and the dense scratch grids of one scene (Sphere manipulators with fixed rotation)."""

    def __init__(self, n_grid, dt, p_vol, p_mass, gravity, ground_friction, softness, radii, frictions, n_particles):
        self.lib = load()
        c = Cfg()
        c.n, c.n_particles, c.n_prim = int(n_grid), int(n_particles), len(radii)
        c.dx, c.inv_dx, c.dt, c.p_vol, c.p_mass = 1.0 / n_grid, float(n_grid), dt, p_vol, p_mass
        c.gravity = (C.c_double * 3)(*gravity)
        c.ground_friction, c.softness = float(ground_friction), float(softness)
        c.radius = (C.c_double * MAXP)(*(list(radii) + [0.0] * (MAXP - len(radii))))
        c.friction = (C.c_double * MAXP)(*(list(frictions) + [0.0] * (MAXP - len(frictions))))
        self.cfg, self.N, self.P = c, int(n_particles), len(radii)
        G = int(n_grid) ** 3
        self.gm, self.gvin, self.gvout = np.zeros(G), np.zeros(3 * G), np.zeros(3 * G)
        self.gvout_a, self.gvin_a, self.gm_a = np.zeros(3 * G), np.zeros(3 * G), np.zeros(G)

    def threads(self, n=None):
        if n is not None:
            self.lib.plb_omp_set_threads(int(n))
        return self.lib.plb_omp_threads()

    def forward(self, pos, pos1, x, v, Cm, F, mu, lam, ys):
        N = self.N
        out = [np.empty((N, 3)), np.empty((N, 3)), np.empty((N, 3, 3)), np.empty((N, 3, 3))]
        args = [np.ascontiguousarray(a, np.float64) for a in (pos, pos1, x, v, Cm, F, mu, lam, ys)]
        self.lib.plb_omp_substep_fwd(C.byref(self.cfg), *[_p(a) for a in args], *[_p(o) for o in out], _p(self.gm), _p(self.gvin), _p(self.gvout))
        return out

    def backward(self, pos, pos1, x, v, Cm, F, mu, lam, ys, x1a, v1a, C1a, F1a):
        N, P = self.N, self.P
        out = [np.empty((N, 3)), np.empty((N, 3)), np.empty((N, 3, 3)), np.empty((N, 3, 3))]
        pa, p1a = np.zeros((max(P, 1), 3)), np.zeros((max(P, 1), 3))
        args = [np.ascontiguousarray(a, np.float64) for a in (pos, pos1, x, v, Cm, F, mu, lam, ys, x1a, v1a, C1a, F1a)]
        self.lib.plb_omp_substep_bwd(C.byref(self.cfg), *[_p(a) for a in args], *[_p(o) for o in out], _p(pa), _p(p1a),
                                     _p(self.gm), _p(self.gvin), _p(self.gvout), _p(self.gvout_a), _p(self.gvin_a), _p(self.gm_a))
        return out, pa[:P], p1a[:P]
