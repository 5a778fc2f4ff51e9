"""ctypes wrapper around tests/host_emul (CPU emulation of the HIP per-particle/per-node math)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_emul")
SHAPE_ID = {"Sphere": 0, "Capsule": 1, "Cylinder": 2, "Torus": 3, "Box": 4, "RollingPin": 1, "Chopsticks": 5}


class EmulCfg(C.Structure):
    _fields_ = [("n_grid", C.c_int), ("use_float", C.c_int), ("n_prim", C.c_int),
                ("dt", C.c_double), ("p_vol", C.c_double), ("p_mass", C.c_double),
                ("gravity", C.c_double * 3), ("ground_friction", C.c_double),
                ("svd_clamp", C.c_double), ("softness", C.c_double)]


class EmulPrim(C.Structure):
    _fields_ = [("shape", C.c_int), ("movable", C.c_int), ("par", C.c_double * 3), ("friction", C.c_double),
                ("pos", C.c_double * 3), ("rot", C.c_double * 4), ("pos1", C.c_double * 3), ("rot1", C.c_double * 4)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "libplb_host_emul.so")
        src = os.path.join(HERE, "emul.cpp")
        hdrs = [os.path.join(HERE, "..", "..", "plasticinelab_amd", "csrc", h) for h in ("mpm_math.h", "mpm_grid.h")]
        if not os.path.exists(so) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in [src] + hdrs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", src, "-o", so])
        _lib = C.CDLL(so)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def prim_par(p):
    if p.shape == "Sphere":
        return (p.radius, 0.0, 0.0)
    if p.shape in ("Capsule", "Cylinder", "RollingPin"):
        return (p.h, p.r, 0.0)
    if p.shape == "Torus":
        return (p.tx, p.ty, 0.0)
    if p.shape == "Box":
        return tuple(p.size)
    if p.shape == "Chopsticks":
        return (p.h, p.r, p.init_gap)          # par[2] = gap[f]; make_prims overwrites it with the pose's gap
    raise NotImplementedError(p.shape)


def make_cfg(sim, n_prim, softness, use_float=False, svd_clamp=1e-6):
    c = EmulCfg()
    c.n_grid, c.use_float, c.n_prim = sim.n_grid, int(use_float), n_prim
    c.dt, c.p_vol, c.p_mass = sim.dt, sim.p_vol, sim.p_mass
    c.gravity = (C.c_double * 3)(*sim.gravity)
    c.ground_friction, c.svd_clamp, c.softness = sim.ground_friction, svd_clamp, softness
    return c


def make_prims(prims, poses_f, poses_f1):
    arr = (EmulPrim * max(len(prims), 1))()
    for i, (p, pose, pose1) in enumerate(zip(prims, poses_f, poses_f1)):
        (pf, rf), (pf1, rf1) = pose[:2], pose1[:2]
        arr[i].shape, arr[i].movable = SHAPE_ID[p.shape], int(p.action_dim > 0)
        par = list(prim_par(p))
        if len(pose) > 2:
            par[2] = float(pose[2])
        arr[i].par = (C.c_double * 3)(*par)
        arr[i].friction = p.friction
        arr[i].pos = (C.c_double * 3)(*np.asarray(pf, float))
        arr[i].rot = (C.c_double * 4)(*np.asarray(rf, float))
        arr[i].pos1 = (C.c_double * 3)(*np.asarray(pf1, float))
        arr[i].rot1 = (C.c_double * 4)(*np.asarray(rf1, float))
    return arr


def substep(cfg, parr, state, mats):
    x, v, Cm, F = [np.ascontiguousarray(a, np.float64) for a in state]
    mu, lam, ys = [np.ascontiguousarray(a, np.float64) for a in mats]
    N = x.shape[0]
    x1, v1, C1, F1 = np.empty_like(x), np.empty_like(v), np.empty_like(Cm), np.empty_like(F)
    lib().emul_substep(C.byref(cfg), parr, N, _p(x), _p(v), _p(Cm), _p(F), _p(mu), _p(lam), _p(ys),
                       _p(x1), _p(v1), _p(C1), _p(F1))
    return x1, v1, C1, F1


def substep_grad(cfg, parr, state, mats, v1, out_adj):
    x, v, Cm, F = [np.ascontiguousarray(a, np.float64) for a in state]
    mu, lam, ys = [np.ascontiguousarray(a, np.float64) for a in mats]
    x1a, v1a, C1a, F1a = [np.ascontiguousarray(a, np.float64) for a in out_adj]
    v1 = np.ascontiguousarray(v1, np.float64)
    N = x.shape[0]
    xa, va, Ca, Fa = np.empty_like(x), np.empty_like(v), np.empty_like(Cm), np.empty_like(F)
    pose = np.zeros((cfg.n_prim, 15))      # pos[f] rot[f] pos[f+1] rot[f+1] gap[f]
    lib().emul_substep_grad(C.byref(cfg), parr, N, _p(x), _p(v), _p(Cm), _p(F), _p(mu), _p(lam), _p(ys), _p(v1),
                            _p(x1a), _p(v1a), _p(C1a), _p(F1a), _p(xa), _p(va), _p(Ca), _p(Fa), _p(pose))
    return (xa, va, Ca, Fa), pose


def fk_fwd(pos, rot, v, w, lo, hi):
    a = [np.ascontiguousarray(t, np.float64) for t in (pos, rot, v, w, lo, hi)]
    pos1, rot1 = np.empty(3), np.empty(4)
    lib().emul_fk_fwd(*[_p(t) for t in a], _p(pos1), _p(rot1))
    return pos1, rot1


def fk_bwd(pos, rot, v, w, lo, hi, pos1_a, rot1_a):
    a = [np.ascontiguousarray(t, np.float64) for t in (pos, rot, v, w, lo, hi, pos1_a, rot1_a)]
    pos_a, rot_a, v_a, w_a = np.zeros(3), np.zeros(4), np.zeros(3), np.zeros(3)
    lib().emul_fk_bwd(*[_p(t) for t in a], _p(pos_a), _p(rot_a), _p(v_a), _p(w_a))
    return pos_a, rot_a, v_a, w_a


def fk_rollingpin_fwd(pos, rot, v, lo, hi):
    a = [np.ascontiguousarray(t, np.float64) for t in (pos, rot, v, lo, hi)]
    pos1, rot1 = np.empty(3), np.empty(4)
    lib().emul_fk_rollingpin_fwd(*[_p(t) for t in a], _p(pos1), _p(rot1))
    return pos1, rot1


def fk_rollingpin_bwd(pos, rot, v, lo, hi, pos1_a, rot1_a):
    a = [np.ascontiguousarray(t, np.float64) for t in (pos, rot, v, lo, hi, pos1_a, rot1_a)]
    pos_a, rot_a, v_a = np.zeros(3), np.zeros(4), np.zeros(3)
    lib().emul_fk_rollingpin_bwd(*[_p(t) for t in a], _p(pos_a), _p(rot_a), _p(v_a))
    return pos_a, rot_a, v_a


def fk_chopsticks_fwd(pos, rot, v, w, gap, gap_vel, min_gap, lo, hi):
    a = [np.ascontiguousarray(t, np.float64) for t in (pos, rot, v, w)]
    b = [np.ascontiguousarray(t, np.float64) for t in (lo, hi)]
    pos1, rot1, gap1 = np.empty(3), np.empty(4), C.c_double()
    lib().emul_fk_chopsticks_fwd(*[_p(t) for t in a], C.c_double(gap), C.c_double(gap_vel), C.c_double(min_gap),
                                 *[_p(t) for t in b], _p(pos1), _p(rot1), C.byref(gap1))
    return pos1, rot1, gap1.value


def fk_chopsticks_bwd(pos, rot, v, w, gap, gap_vel, min_gap, lo, hi, pos1_a, rot1_a, gap1_a):
    a = [np.ascontiguousarray(t, np.float64) for t in (pos, rot, v, w)]
    b = [np.ascontiguousarray(t, np.float64) for t in (lo, hi, pos1_a, rot1_a)]
    pos_a, rot_a, v_a, w_a = np.zeros(3), np.zeros(4), np.zeros(3), np.zeros(3)
    gap_a, gv_a = C.c_double(0.0), C.c_double(0.0)
    lib().emul_fk_chopsticks_bwd(*[_p(t) for t in a], C.c_double(gap), C.c_double(gap_vel), C.c_double(min_gap),
                                 *[_p(t) for t in b], C.c_double(gap1_a), _p(pos_a), _p(rot_a), C.byref(gap_a),
                                 _p(v_a), _p(w_a), C.byref(gv_a))
    return pos_a, rot_a, gap_a.value, v_a, w_a, gv_a.value


def constitutive(Et, mu, lam, ys, GS, GF, clamp=1e-6, use_float=False, allow_fast=True):
    """The constitutive block of p2g / p2g.grad for n particles (Et = F_tmp - I): (stress, new_F - I, F_tmp adjoint, fast?)
    through the Jacobi path or -- allow_fast, per particle -- the elastic fast path of mpm_math.h."""
    n = len(Et)
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (Et, np.broadcast_to(mu, (n,)), np.broadcast_to(lam, (n,)),
                                                              np.broadcast_to(ys, (n,)), GS, GF)]
    stress, En, Fta = np.empty((n, 3, 3)), np.empty((n, 3, 3)), np.empty((n, 3, 3))
    fast = np.zeros(n, np.int32)
    lib().emul_constitutive(int(use_float), int(allow_fast), n, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), C.c_double(clamp),
                            _p(a[4]), _p(a[5]), _p(stress), _p(En), _p(Fta), _p(fast))
    return stress, En, Fta, fast.astype(bool)


def gather_pk_check(n_grid, p_mass, x, g):
    """Largest relative difference between the packed-pair and the plain form of the p2g.grad gather over the stencils (x[n,3], g[n,27,4])."""
    L = lib()
    L.emul_gather_pk_check.restype = C.c_double
    x, g = np.ascontiguousarray(x, np.float64), np.ascontiguousarray(g, np.float64)
    return float(L.emul_gather_pk_check(int(n_grid), C.c_double(p_mass), len(x), _p(x), _p(g)))


def g2p_pk_check(n_grid, dt, x, gv):
    """Largest relative difference between g2p_particle_pk and g2p_particle over the stencils (x[n,3], gv[n,27,3])."""
    L = lib()
    L.emul_g2p_pk_check.restype = C.c_double
    x, gv = np.ascontiguousarray(x, np.float64), np.ascontiguousarray(gv, np.float64)
    return float(L.emul_g2p_pk_check(int(n_grid), C.c_double(dt), len(x), _p(x), _p(gv)))


def svd_singular_check(Et, use_float):
    """(largest difference between svd_finish and its pre-round-6 column-loop form over the matrices Et[n,3,3] = F - I, how many of
    them took the nearly-singular branch)."""
    L = lib()
    L.emul_svd_singular_check.restype = C.c_double
    Et = np.ascontiguousarray(Et, np.float64)
    k = C.c_int(0)
    d = float(L.emul_svd_singular_check(int(use_float), len(Et), _p(Et), C.byref(k)))
    return d, k.value
