"""CPU oracle: float64 restatement of the PlasticineLab MPM hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``plasticinelab_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker.

PARITY UNPINNED.  The reference's arithmetic lives in Taichi 0.7.14 kernels
(``/root/reference/plb/engine/mpm_simulator.py`` et al.); Taichi is not
installable in the build container and the reference ships no fixtures for
this path.  This file follows the reference source line
by line (citations ``file:line`` on every function), the forward pass is plain
torch float64, and the reverse pass is ``torch.autograd`` over that forward
with three hand-written pieces where Taichi's autodiff is known to differ from
torch's:

* ``ti_max`` / ``ti_min``  -- Taichi routes the adjoint of ``max(a, b)`` to
  ``a`` iff ``b < a`` (ties go to ``b``); ``min`` symmetric  [UNVERIFIED, from
  Taichi 0.7.x ``auto_diff.cpp``; only differs from torch on exact ties].
* ``SvdRef``  -- forward is an exact SVD, backward is the reference's own
  ``backward_svd`` (mpm_simulator.py:97-115) including its +-1e-6 clamp.
* ``AtomicMinAsAdd``  -- Taichi 0.7.x differentiates every ``AtomicOpStmt`` as
  if it were an add, so the hard contact loss (loss.py:123-128) sends the
  ``min_dist`` adjoint to *every* particle  [UNVERIFIED].

It is independent of the hand-derived HIP adjoints by construction (autograd
versus closed forms), which is what makes it a useful checker.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

DT = torch.float64


# --------------------------------------------------------------------------- #
# configuration (mpm_simulator.py:6-34)
# --------------------------------------------------------------------------- #
@dataclass
class SimCfg:
    n_particles: int
    quality: float = 1.0
    yield_stress: float = 50.0
    E: float = 5e3
    nu: float = 0.2
    ground_friction: float = 1.5
    gravity: Tuple[float, float, float] = (0.0, -1.0, 0.0)
    max_steps: int = 1024
    dim: int = 3
    # derived
    n_grid: int = field(init=False)
    dx: float = field(init=False)
    inv_dx: float = field(init=False)
    dt: float = field(init=False)
    p_vol: float = field(init=False)
    p_mass: float = field(init=False)
    mu: float = field(init=False)
    lam: float = field(init=False)
    substeps: int = field(init=False)

    def __post_init__(self):
        q = self.quality * 0.5                       # :16-17 (dim == 3)
        self.n_grid = int(128 * q)                   # :19
        self.dx, self.inv_dx = 1 / self.n_grid, float(self.n_grid)   # :21
        self.dt = 0.5e-4 / q                         # :22
        self.p_vol = (self.dx * 0.5) ** 2            # :23  (2-D formula kept in 3-D)
        self.p_mass = self.p_vol * 1                 # :24
        E, nu = self.E, self.nu
        self.mu = E / (2 * (1 + nu))                 # :28
        self.lam = E * nu / ((1 + nu) * (1 - 2 * nu))
        self.substeps = int(2e-3 // self.dt)         # :34


@dataclass
class PrimCfg:
    """primive_base.py:209-224 + per-shape default_config."""
    shape: str = "Sphere"
    init_pos: Tuple[float, float, float] = (0.3, 0.3, 0.3)
    init_rot: Tuple[float, float, float, float] = (1.0, 0.0, 0.0, 0.0)
    lower_bound: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    upper_bound: Tuple[float, float, float] = (1.0, 1.0, 1.0)
    friction: float = 0.9
    action_dim: int = 0
    action_scale: Tuple[float, ...] = ()
    radius: float = 1.0          # Sphere
    h: float = 0.06              # Capsule / Cylinder (Cylinder default h=.2,r=.1)
    r: float = 0.03
    tx: float = 0.2              # Torus
    ty: float = 0.1
    size: Tuple[float, float, float] = (0.1, 0.1, 0.1)   # Box
    minimal_gap: float = 0.06    # Chopsticks (primitives.py:150-151)
    init_gap: float = 0.06


# --------------------------------------------------------------------------- #
# Taichi autodiff semantics
# --------------------------------------------------------------------------- #
# The two unverified pieces of Taichi semantics are switches here as in the engine (plmpm_config.contact_min_adjoint /
# minmax_tie, include/plmpm.h): when Taichi-generated vectors arrive (tests/golden/make_taichi_golden.py) and disagree,
# the fix is a flag.  Defaults = what Taichi 0.7.x is remembered to do.
SEMANTICS = {"minmax_tie": "second",            # adjoint of max / min on an exact tie: "second" operand | "first"
             "contact_min_adjoint": "add"}      # ti.atomic_min in the hard contact loss: "add" (every particle) | "argmin"


def ti_max(a, b):
    """max(lhs, rhs): adjoint to lhs iff rhs < lhs, else to rhs (ties to lhs with minmax_tie = "first")."""
    a, b = torch.broadcast_tensors(torch.as_tensor(a, dtype=DT), torch.as_tensor(b, dtype=DT))
    return torch.where(b <= a if SEMANTICS["minmax_tie"] == "first" else b < a, a, b)


def ti_min(a, b):
    """min(lhs, rhs): adjoint to lhs iff lhs < rhs, else to rhs (ties to lhs with minmax_tie = "first")."""
    a, b = torch.broadcast_tensors(torch.as_tensor(a, dtype=DT), torch.as_tensor(b, dtype=DT))
    return torch.where(a <= b if SEMANTICS["minmax_tie"] == "first" else a < b, a, b)


def _svd_clamp(a):
    """mpm_simulator.py:143-151."""
    return torch.where(a >= 0, torch.clamp(a, min=1e-6), torch.clamp(a, max=-1e-6))


class SvdRef(torch.autograd.Function):
    """ti.svd forward (mpm_simulator.py:87-90) + backward_svd (:97-115).

    Taichi's ``ti.svd`` is third-party; any exact SVD gives the same new_F /
    stress (they are invariant to the sign/ordering convention as long as
    det(F_tmp) > 0).  sig is returned as the (N,3) diagonal.
    """

    @staticmethod
    def forward(ctx, Fm):
        U, S, Vh = torch.linalg.svd(Fm)
        V = Vh.transpose(-1, -2).contiguous()
        ctx.save_for_backward(U, S, V)
        return U, S, V

    @staticmethod
    def backward(ctx, gu, gsig, gv):
        u, sig, v = ctx.saved_tensors
        vt, ut = v.transpose(-1, -2), u.transpose(-1, -2)
        Sig = torch.diag_embed(sig)
        sigma_term = u @ torch.diag_embed(gsig) @ vt
        s = sig ** 2
        diff = s[:, None, :] - s[:, :, None]            # [i,j] = s[j]-s[i]
        Fm = 1.0 / _svd_clamp(diff)
        eye = torch.eye(3, dtype=DT).bool()
        Fm = torch.where(eye, torch.zeros_like(Fm), Fm)
        u_term = u @ ((Fm * (ut @ gu - gu.transpose(-1, -2) @ u)) @ Sig) @ vt
        v_term = u @ (Sig @ ((Fm * (vt @ gv - gv.transpose(-1, -2) @ v)) @ vt))
        return u_term + v_term + sigma_term


class AtomicMinAsAdd(torch.autograd.Function):
    """ti.atomic_min(dest, val) whose adjoint is ``val.grad += dest.grad`` for
    every contributing val (Taichi 0.7.x AtomicOpStmt adjoint) [UNVERIFIED]."""

    @staticmethod
    def forward(ctx, init, vals):
        ctx.n = vals.shape[0]
        out = torch.minimum(torch.as_tensor(init, dtype=DT), vals.min())
        ctx.hit = (vals == out)                   # contact_min_adjoint = "argmin": only the particle(s) attaining the minimum
        return out

    @staticmethod
    def backward(ctx, g):
        if SEMANTICS["contact_min_adjoint"] == "argmin":
            return None, g * ctx.hit.to(DT)
        return None, g.expand(ctx.n).clone()


# --------------------------------------------------------------------------- #
# quaternion helpers (primitive/utils.py:3-48)
# --------------------------------------------------------------------------- #
def length8(x):                       # utils.py:3-5
    return torch.sqrt((x * x).sum(-1) + 1e-8)


def length14(x):                      # primitives.py:8-10
    return torch.sqrt((x * x).sum(-1) + 1e-14)


def qrot(rot, v):                     # utils.py:7-13
    qvec = rot[..., 1:4].expand(v.shape)
    uv = torch.linalg.cross(qvec, v)
    uuv = torch.linalg.cross(qvec, uv)
    return v + 2 * (rot[..., 0:1] * uv + uuv)


def qmul(q, r):                       # utils.py:19-27
    t = r[:, None] * q[None, :]       # terms = r.outer_product(q)
    w = t[0, 0] - t[1, 1] - t[2, 2] - t[3, 3]
    x = t[0, 1] + t[1, 0] - t[2, 3] + t[3, 2]
    y = t[0, 2] + t[1, 3] + t[2, 0] - t[3, 1]
    z = t[0, 3] - t[1, 2] + t[2, 1] + t[3, 0]
    out = torch.stack([w, x, y, z])
    return out / torch.sqrt((out * out).sum())


def w2quat(axis_angle):               # utils.py:29-41
    w = torch.sqrt((axis_angle * axis_angle).sum())
    if float(w) > 1e-9:
        v = (axis_angle / w) * torch.sin(w / 2)
        return torch.cat([torch.cos(w / 2)[None], v])
    return torch.tensor([1.0, 0.0, 0.0, 0.0], dtype=DT)


def inv_trans(pos, position, rotation):   # utils.py:43-47
    iq = torch.stack([rotation[0], -rotation[1], -rotation[2], -rotation[3]])
    iq = iq / torch.sqrt((iq * iq).sum())
    return qrot(iq, pos - position)


# --------------------------------------------------------------------------- #
# primitive shapes (primitives.py)
# --------------------------------------------------------------------------- #
def _capsule_of(p: PrimCfg):
    return PrimCfg(shape="Capsule", h=p.h, r=p.r)


def _shape_sdf(p: PrimCfg, gp, gap=None):
    """local-frame _sdf; primitives.py:42-47 (Capsule), :163-167 (Cylinder),
    :199-202 (Torus), :232-238 (Box)."""
    if p.shape == "Chopsticks":                  # primitives.py:111-117: two capsules gap apart, hanging below the pose
        cap = _capsule_of(p)
        delta = torch.stack([gap / 2, torch.zeros((), dtype=DT), torch.zeros((), dtype=DT)])
        q = gp - torch.tensor([0.0, -p.h / 2, 0.0], dtype=DT)
        return ti_min(_shape_sdf(cap, q - delta), _shape_sdf(cap, q + delta))
    if p.shape in ("Capsule", "RollingPin"):     # RollingPin(Capsule), primitives.py:64
        y = gp[:, 1] + p.h / 2
        y = y - ti_min(ti_max(y, 0.0), p.h)
        p2 = torch.stack([gp[:, 0], y, gp[:, 2]], -1)
        return length14(p2) - p.r
    if p.shape == "Cylinder":
        l = length14(torch.stack([gp[:, 0], gp[:, 2]], -1))
        d = torch.abs(torch.stack([l, gp[:, 1]], -1)) - torch.tensor([p.h, p.r], dtype=DT)
        return ti_min(ti_max(d[:, 0], d[:, 1]), 0.0) + length14(ti_max(d, 0.0))
    if p.shape == "Torus":
        q = torch.stack([length14(torch.stack([gp[:, 0], gp[:, 2]], -1)) - p.tx, gp[:, 1]], -1)
        return length14(q) - p.ty
    if p.shape == "Box":
        q = torch.abs(gp) - torch.tensor(p.size, dtype=DT)
        out = length14(ti_max(q, 0.0))
        return out + ti_min(ti_max(q[:, 0], ti_max(q[:, 1], q[:, 2])), 0.0)
    raise NotImplementedError(p.shape)


def _shape_normal(p: PrimCfg, gp, gap=None):
    """local-frame _normal; primitives.py:49-54, :169-183, :204-213, :240-251; Chopsticks :119-129."""
    if p.shape == "Chopsticks":
        cap = _capsule_of(p)
        delta = torch.stack([gap / 2, torch.zeros((), dtype=DT), torch.zeros((), dtype=DT)])
        q = gp - torch.tensor([0.0, -p.h / 2, 0.0], dtype=DT)
        a, b = _shape_sdf(cap, q - delta), _shape_sdf(cap, q + delta)
        m = (a <= b).to(DT)[:, None].detach()          # ti.cast of a comparison: no gradient
        return m * _shape_normal(cap, q - delta) + (1 - m) * _shape_normal(cap, q + delta)
    if p.shape in ("Capsule", "RollingPin"):
        y = gp[:, 1] + p.h / 2
        y = y - ti_min(ti_max(y, 0.0), p.h)
        p2 = torch.stack([gp[:, 0], y, gp[:, 2]], -1)
        return p2 / length14(p2)[:, None]
    if p.shape == "Cylinder":
        pp = torch.stack([gp[:, 0], gp[:, 2]], -1)
        l = length14(pp)
        d = torch.stack([l, torch.abs(gp[:, 1])], -1) - torch.tensor([p.h, p.r], dtype=DT)
        f = (d[:, 0] > d[:, 1]).to(DT)
        inside = (ti_max(d[:, 0], d[:, 1]) <= 0.0).to(DT)
        n2 = ti_max(d, 0.0) + inside[:, None] * torch.stack([f, 1 - f], -1)
        n2_ = n2 / length14(n2)[:, None]
        p2 = pp / l[:, None]
        sgn = (gp[:, 1] >= 0).to(DT) * 2 - 1
        n3 = torch.stack([p2[:, 0] * n2_[:, 0], n2_[:, 1] * sgn, p2[:, 1] * n2_[:, 0]], -1)
        return n3 / length14(n3)[:, None]
    if p.shape == "Torus":
        x = torch.stack([gp[:, 0], gp[:, 2]], -1)
        l = length14(x)
        q = torch.stack([l - p.tx, gp[:, 1]], -1)
        n2 = q / length14(q)[:, None]
        x2 = x / l[:, None]
        n3 = torch.stack([x2[:, 0] * n2[:, 0], n2[:, 1], x2[:, 1] * n2[:, 0]], -1)
        return n3 / length14(n3)[:, None]
    if p.shape == "Box":
        d = 1e-4
        cols = []
        for i in range(3):
            e = torch.zeros(3, dtype=DT)
            e[i] = d
            cols.append((0.5 / d) * (_shape_sdf(p, gp + e) - _shape_sdf(p, gp - e)))
        n = torch.stack(cols, -1)
        return n / length14(n)[:, None]
    raise NotImplementedError(p.shape)


def prim_sdf(p: PrimCfg, pos_f, rot_f, gp, gap_f=None):
    """Primitive.sdf (primive_base.py:57-60); Sphere override primitives.py:22-24."""
    if p.shape == "Sphere":
        return length14(gp - pos_f) - p.radius
    return _shape_sdf(p, inv_trans(gp, pos_f, rot_f), gap_f)


def prim_normal(p: PrimCfg, pos_f, rot_f, gp, gap_f=None):
    """Primitive.normal (primive_base.py:75-80); Sphere override primitives.py:26-28."""
    if p.shape == "Sphere":
        d = gp - pos_f
        return d / length14(d)[:, None]
    return qrot(rot_f, _shape_normal(p, inv_trans(gp, pos_f, rot_f), gap_f))


def collider_v(pos_f, rot_f, pos_f1, rot_f1, gp, dt):
    """primive_base.py:82-89."""
    rel = inv_trans(gp, pos_f, rot_f)
    new_pos = qrot(rot_f1, rel) + pos_f1
    return (new_pos - gp) / dt


def collide(p: PrimCfg, softness: float, pos_f, rot_f, pos_f1, rot_f1, gp, v_out, dt, gap_f=None):
    """Primitive.collide, primive_base.py:91-115.  gp, v_out: (A,3).  ``gap_f``: Chopsticks gap[f]."""
    dist = prim_sdf(p, pos_f, rot_f, gp, gap_f)
    influence = ti_min(torch.exp(-dist * softness), 1.0)
    mask = (dist <= 0)
    if softness > 0:
        mask = mask | (influence > 0.1)
    idx = torch.nonzero(mask).squeeze(-1)
    if idx.numel() == 0:
        return v_out
    g, v, infl = gp[idx], v_out[idx], influence[idx]
    D = prim_normal(p, pos_f, rot_f, g, gap_f)
    cv = collider_v(pos_f, rot_f, pos_f1, rot_f1, g, dt)
    input_v = v - cv
    nc = (input_v * D).sum(-1)
    gvt = input_v - ti_min(nc, 0.0)[:, None] * D
    gnorm = length8(gvt)
    gfric = gvt / gnorm[:, None] * ti_max(0.0, gnorm + nc * p.friction)[:, None]
    flag = ((nc < 0) & (torch.sqrt((gvt * gvt).sum(-1)) > 1e-30)).to(DT)[:, None].detach()
    gvt = gfric * flag + gvt * (1 - flag)
    v_new = cv + input_v * (1 - infl)[:, None] + gvt * infl[:, None]
    return v_out.index_put((idx,), v_new)


def forward_kinematics(p: PrimCfg, pos, rot, v, w, gap=None, gap_vel=None):
    """primive_base.py:117-121 (world-frame rotation update); RollingPin override primitives.py:66-80;
    Chopsticks override primitives.py:94-98 (body-frame rotation update + gap) returns (pos, rot, gap)."""
    lo = torch.tensor(p.lower_bound, dtype=DT)
    hi = torch.tensor(p.upper_bound, dtype=DT)
    if p.shape == "Chopsticks":
        new_gap = ti_max(gap - gap_vel, p.minimal_gap)
        new_pos = ti_max(ti_min(pos + v, hi), lo)
        return new_pos, qmul(rot, w2quat(w)), new_gap
    if p.shape == "RollingPin":                      # primitives.py:66-80
        dw, dth, dy = v[0], v[1], v[2]               # roll about own y, turn about world y, move down
        y_dir = qrot(rot, torch.tensor([0.0, -1.0, 0.0], dtype=DT))
        x_dir = torch.linalg.cross(torch.tensor([0.0, 1.0, 0.0], dtype=DT), y_dir) * dw * 0.03
        x_dir = torch.stack([x_dir[0], dy, x_dir[2]])
        zero = torch.zeros((), dtype=DT)
        new_rot = qmul(w2quat(torch.stack([zero, -dth, zero])), qmul(rot, w2quat(torch.stack([zero, dw, zero]))))
        new_pos = ti_max(ti_min(pos + x_dir, hi), lo)
        return new_pos, new_rot
    new_pos = ti_max(ti_min(pos + v, hi), lo)
    new_rot = qmul(w2quat(w), rot)
    return new_pos, new_rot


def set_velocity(p: PrimCfg, action, n_substeps):
    """primive_base.py:184-192: per-substep (v, w) from one env-step action; Chopsticks (primitives.py:101-109)
    adds the gap velocity as a third entry."""
    scale = torch.tensor(p.action_scale, dtype=DT)
    v = action[:3] * scale[:3] / n_substeps
    if p.action_dim > 3:
        w = action[3:6] * scale[3:6] / n_substeps
    else:
        w = torch.zeros(3, dtype=DT)
    if p.shape == "Chopsticks":
        return v, w, action[6] * scale[6] / n_substeps
    return v, w


# --------------------------------------------------------------------------- #
# MPM kernels (mpm_simulator.py)
# --------------------------------------------------------------------------- #
def _stencil(cfg: SimCfg, x):
    """base / fx / w of mpm_simulator.py:160-163 (also :226-228, :385-387).
    ``cast(int)`` truncates toward zero (Q1) and carries no gradient."""
    base = torch.trunc(x.detach() * cfg.inv_dx - 0.5).to(torch.int64)
    fx = x * cfg.inv_dx - base.to(DT)
    w = [0.5 * (1.5 - fx) ** 2, 0.75 - (fx - 1) ** 2, 0.5 * (fx - 0.5) ** 2]
    return base, fx, w


def _flat(cfg: SimCfg, I):
    n = cfg.n_grid
    return (I[:, 0] * n + I[:, 1]) * n + I[:, 2]


def compute_F_tmp(cfg: SimCfg, C, F):
    """mpm_simulator.py:82-85."""
    return (torch.eye(3, dtype=DT) + cfg.dt * C) @ F


def compute_von_mises(F, U, sig, V, yield_stress, mu):
    """mpm_simulator.py:124-141.  sig: (N,3) diagonal."""
    sigc = ti_max(sig, 0.05)
    eps = torch.log(sigc)
    eps_hat = eps - eps.sum(-1, keepdim=True) / 3
    eps_hat_norm = torch.sqrt((eps_hat * eps_hat).sum(-1) + 1e-8)
    delta_gamma = eps_hat_norm - yield_stress / (2 * mu)
    yields = delta_gamma > 0
    eps2 = eps - (delta_gamma / eps_hat_norm)[:, None] * eps_hat
    Fy = U @ torch.diag_embed(torch.exp(eps2)) @ V.transpose(-1, -2)
    return torch.where(yields[:, None, None], Fy, F), yields


def p2g(cfg: SimCfg, x, v, C, F_tmp, U, sig, V, mu, lam, ys):
    """mpm_simulator.py:157-184 -> (F_next, grid_v_in (G,3), grid_m (G,))."""
    n = cfg.n_grid
    base, fx, w = _stencil(cfg, x)
    new_F, _ = compute_von_mises(F_tmp, U, sig, V, ys, mu)
    J = torch.linalg.det(new_F)
    r = U @ V.transpose(-1, -2)
    eye = torch.eye(3, dtype=DT)
    stress = 2 * mu[:, None, None] * (new_F - r) @ new_F.transpose(-1, -2) \
        + eye * (lam * J * (J - 1))[:, None, None]
    stress = (-cfg.dt * cfg.p_vol * 4 * cfg.inv_dx * cfg.inv_dx) * stress
    affine = stress + cfg.p_mass * C
    gv = torch.zeros(n ** 3, 3, dtype=DT)
    gm = torch.zeros(n ** 3, dtype=DT)
    for i in range(3):
        for j in range(3):
            for k in range(3):
                off = torch.tensor([i, j, k])
                dpos = (off.to(DT) - fx) * cfg.dx
                weight = w[i][:, 0] * w[j][:, 1] * w[k][:, 2]
                idx = _flat(cfg, base + off)
                contrib = weight[:, None] * (cfg.p_mass * v + (affine @ dpos[:, :, None]).squeeze(-1))
                gv = gv.index_add(0, idx, contrib)
                gm = gm.index_add(0, idx, weight * cfg.p_mass)
    return new_F, gv, gm


def grid_op(cfg: SimCfg, prims: Sequence[PrimCfg], softness, poses_f, poses_f1, gv_in, gm):
    """mpm_simulator.py:189-221.  poses_*: list of (pos, rot) per primitive."""
    n = cfg.n_grid
    act = torch.nonzero(gm.detach() > 1e-12).squeeze(-1)
    I = torch.stack([act // (n * n), (act // n) % n, act % n], -1)
    v_out = (1 / gm[act])[:, None] * gv_in[act]
    v_out = v_out + cfg.dt * torch.tensor(cfg.gravity, dtype=DT) * 30
    gp = I.to(DT) * cfg.dx
    for p, pf, pf1 in zip(prims, poses_f, poses_f1):      # a pose is (pos, rot) or, for Chopsticks, (pos, rot, gap)
        v_out = collide(p, softness, pf[0], pf[1], pf1[0], pf1[1], gp, v_out, cfg.dt, *pf[2:])
    bound = 3
    If = I.to(DT)
    for d in range(3):
        lo = (I[:, d] < bound) & (v_out[:, d].detach() < 0)
        if d != 1 or cfg.ground_friction == 0:
            col = torch.where(lo, torch.zeros_like(v_out[:, d]), v_out[:, d])
            v_out = torch.cat([v_out[:, :d], col[:, None], v_out[:, d + 1:]], -1)
        elif cfg.ground_friction < 10:
            normal = torch.zeros(3, dtype=DT)
            normal[d] = 1.0
            lin = (v_out * normal).sum(-1) + 1e-30
            vit = v_out - lin[:, None] * normal - If * 1e-30
            lit = torch.sqrt((vit * vit).sum(-1) + 1e-8)
            vf = ti_max(1.0 + cfg.ground_friction * lin / lit, 0.0)[:, None] * (vit + If * 1e-30)
            vf = torch.cat([vf[:, :1], torch.zeros_like(vf[:, 1:2]), vf[:, 2:]], -1)
            v_out = torch.where(lo[:, None], vf, v_out)
        else:
            v_out = torch.where(lo[:, None], torch.zeros_like(v_out), v_out)
        hi = (I[:, d] > n - bound) & (v_out[:, d].detach() > 0)
        col = torch.where(hi, torch.zeros_like(v_out[:, d]), v_out[:, d])
        v_out = torch.cat([v_out[:, :d], col[:, None], v_out[:, d + 1:]], -1)
    out = torch.zeros(n ** 3, 3, dtype=DT)
    return out.index_put((act,), v_out)


def g2p(cfg: SimCfg, x, gv_out):
    """mpm_simulator.py:223-242 -> (x', v', C')."""
    base, fx, w = _stencil(cfg, x)
    new_v = torch.zeros_like(x)
    new_C = torch.zeros(x.shape[0], 3, 3, dtype=DT)
    for i in range(3):
        for j in range(3):
            for k in range(3):
                off = torch.tensor([i, j, k])
                dpos = off.to(DT) - fx
                g_v = gv_out[_flat(cfg, base + off)]
                weight = w[i][:, 0] * w[j][:, 1] * w[k][:, 2]
                new_v = new_v + weight[:, None] * g_v
                new_C = new_C + 4 * cfg.inv_dx * weight[:, None, None] * (g_v[:, :, None] * dpos[:, None, :])
    new_x = ti_max(ti_min(x + cfg.dt * new_v, 1.0 - 3 * cfg.dx), 0.0)
    return new_x, new_v, new_C


def substep(cfg, prims, softness, state, mats, poses_f, poses_f1):
    """mpm_simulator.py:245-257 (forward_kinematics is done by the caller)."""
    x, v, C, F = state
    mu, lam, ys = mats
    F_tmp = compute_F_tmp(cfg, C, F)
    U, sig, V = SvdRef.apply(F_tmp)
    F_next, gv_in, gm = p2g(cfg, x, v, C, F_tmp, U, sig, V, mu, lam, ys)
    gv_out = grid_op(cfg, prims, softness, poses_f, poses_f1, gv_in, gm)
    x1, v1, C1 = g2p(cfg, x, gv_out)
    return (x1, v1, C1, F_next)


def compute_grid_m(cfg: SimCfg, x):
    """mpm_simulator.py:382-392."""
    n = cfg.n_grid
    base, fx, w = _stencil(cfg, x)
    gm = torch.zeros(n ** 3, dtype=DT)
    for i in range(3):
        for j in range(3):
            for k in range(3):
                off = torch.tensor([i, j, k])
                weight = w[i][:, 0] * w[j][:, 1] * w[k][:, 2]
                gm = gm.index_add(0, _flat(cfg, base + off), weight * cfg.p_mass)
    return gm


# --------------------------------------------------------------------------- #
# loss (losses/loss.py)
# --------------------------------------------------------------------------- #
def update_target_sdf_numpy(target_density: np.ndarray, dx: float, inf: float = 1000.0,
                            max_sweeps: Optional[int] = None):
    """loss.py:81-106: 2*n_grid Jacobi sweeps of a 6^3 (offsets -3..2) nearest
    point propagation.  Vectorised over nodes, sequential over the 215 offsets
    in ``ti.ndrange`` order so the strict ``<`` tie-breaking is kept.  Stops
    early once a sweep changes nothing (later sweeps are then no-ops)."""
    n = target_density.shape[0]
    sweeps = 2 * n if max_sweeps is None else max_sweeps
    gi = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1)
    grid_pos = gi.astype(np.float64) * dx
    solid = target_density > 1e-4
    sdf_copy = np.full((n, n, n), inf)
    np_copy = np.zeros((n, n, n, 3))
    sdf = np.full((n, n, n), inf)
    npnt = np.zeros((n, n, n, 3))
    offsets = [(a, b, c) for a in range(-3, 3) for b in range(-3, 3) for c in range(-3, 3)
               if (a, b, c) != (0, 0, 0)]
    for _ in range(sweeps):
        sdf = np.full((n, n, n), inf)
        # nearest_point keeps its previous value where nothing improves (field persists)
        sdf[solid] = 0.0
        npnt[solid] = grid_pos[solid]
        for (a, b, c) in offsets:
            # neighbour v = I + offset: valid region slices
            sl_dst, sl_src = [], []
            for o in (a, b, c):
                if o >= 0:
                    sl_dst.append(slice(0, n - o)); sl_src.append(slice(o, n))
                else:
                    sl_dst.append(slice(-o, n)); sl_src.append(slice(0, n + o))
            sl_dst, sl_src = tuple(sl_dst), tuple(sl_src)
            nb_sdf = sdf_copy[sl_src]
            nb_np = np_copy[sl_src]
            d = grid_pos[sl_dst] - nb_np
            dist = np.sqrt((d * d).sum(-1) + 1e-8)
            better = (nb_sdf < inf) & (dist < sdf[sl_dst]) & (~solid[sl_dst])
            sub_sdf = sdf[sl_dst]
            sub_np = npnt[sl_dst]
            sub_sdf[better] = dist[better]
            sub_np[better] = nb_np[better]
            sdf[sl_dst] = sub_sdf
            npnt[sl_dst] = sub_np
        if np.array_equal(sdf, sdf_copy) and np.array_equal(npnt, np_copy):
            break
        sdf_copy = sdf.copy()
        np_copy = npnt.copy()
    return sdf, npnt


def soft_weight(d):                  # loss.py:112-114
    return 1 / (1 + d * d * 10000)


@dataclass
class LossCfg:
    sdf_weight: float = 10.0
    density_weight: float = 10.0
    contact_weight: float = 1.0
    soft_contact: bool = False


def compute_loss(cfg: SimCfg, lcfg: LossCfg, prims, x, poses_f, target_density, target_sdf):
    """Loss.compute_loss_kernel, loss.py:186-208 (one call; returns the increment
    added to ``loss`` plus the per-term values)."""
    gm = compute_grid_m(cfg, x)
    density_loss = torch.abs(gm - target_density).sum()          # :145-148
    sdf_loss = (target_sdf * gm).sum()                           # :150-153
    contact_loss = torch.zeros((), dtype=DT)
    movable = [(p, pose) for p, pose in zip(prims, poses_f) if p.action_dim > 0]   # :20-24
    for p, pose in movable:
        d = ti_max(prim_sdf(p, pose[0], pose[1], x, *pose[2:]), 0.0)
        if lcfg.soft_contact:                                    # :116-121, :130-135
            sw = soft_weight(d)
            dist_norm = sw.sum()
            min_dist = (d * sw / dist_norm).sum()
        else:                                                    # :123-128
            min_dist = AtomicMinAsAdd.apply(100000.0, ti_max(d, 0.0))
        contact_loss = contact_loss + min_dist ** 2              # :137-140
    loss = contact_loss * lcfg.contact_weight + density_loss * lcfg.density_weight \
        + sdf_loss * lcfg.sdf_weight                             # :158-162
    return loss, dict(contact_loss=contact_loss, density_loss=density_loss, sdf_loss=sdf_loss, grid_m=gm)


def iou(gm, target):                 # loss.py:239-254
    ma, mb = gm.max(), target.max()
    I = (gm * target).sum() / ma / mb
    U = gm.sum() / ma + target.sum() / mb
    return I / (U - I)


# --------------------------------------------------------------------------- #
# env-step / rollout drivers
# --------------------------------------------------------------------------- #
def env_step(cfg, prims, softness, state, mats, poses, action):
    """MPMSimulator.step(is_copy=False, action) (mpm_simulator.py:365-376):
    set_action (clip to [-1,1] without gating the gradient, Q12), then
    ``substeps`` x [forward_kinematics; substep]."""
    with torch.no_grad():
        clipped = action.detach().clamp(-1, 1)
    act = action + (clipped - action.detach())       # value clipped, identity gradient
    vel = []
    ofs = 0
    for p in prims:
        vel.append(set_velocity(p, act[ofs:ofs + p.action_dim], cfg.substeps) if p.action_dim > 0
                   else (torch.zeros(3, dtype=DT), torch.zeros(3, dtype=DT)))
        ofs += p.action_dim
    for _ in range(cfg.substeps):
        nxt = [forward_kinematics(p, pose[0], pose[1], vw[0], vw[1], *pose[2:], *vw[2:])
               for p, pose, vw in zip(prims, poses, vel)]
        state = substep(cfg, prims, softness, state, mats, poses, nxt)
        poses = nxt
    return state, poses


def rollout_loss_and_grad(cfg, lcfg, prims, softness, state0, mats, poses0, actions,
                          target_density, target_sdf, want_grad=True):
    """Solver.forward (plb/optimizer/solver.py:31-44): total loss over the
    horizon and d loss / d actions, float64.  Backward is done one env step at
    a time from saved step-boundary states (same result as one tape, less
    memory)."""
    H = actions.shape[0]
    states, poses_l = [tuple(t.detach() for t in state0)], [[tuple(t.detach() for t in po) for po in poses0]]
    total = 0.0
    info = []
    with torch.no_grad():
        for i in range(H):
            s, po = env_step(cfg, prims, softness, states[-1], mats, poses_l[-1], actions[i])
            l, parts = compute_loss(cfg, lcfg, prims, s[0], po, target_density, target_sdf)
            total += float(l)
            info.append({k: float(v) for k, v in parts.items() if k != "grid_m"})
            states.append(tuple(t.detach() for t in s))
            poses_l.append([tuple(t.detach() for t in q) for q in po])
    if not want_grad:
        return total, None, states, poses_l, info
    grad = torch.zeros_like(actions)
    adj_state = [torch.zeros_like(t) for t in states[-1]]
    adj_pose = [tuple(torch.zeros_like(t) for t in po) for po in poses_l[-1]]
    for i in reversed(range(H)):
        s_in = tuple(t.clone().requires_grad_(True) for t in states[i])
        p_in = [tuple(t.clone().requires_grad_(True) for t in po) for po in poses_l[i]]
        a_in = actions[i].clone().requires_grad_(True)
        s_out, p_out = env_step(cfg, prims, softness, s_in, mats, p_in, a_in)
        l, _ = compute_loss(cfg, lcfg, prims, s_out[0], p_out, target_density, target_sdf)
        obj = l
        for t, a in zip(s_out, adj_state):
            obj = obj + (t * a).sum()
        for po, ap in zip(p_out, adj_pose):
            for t, a in zip(po, ap):
                obj = obj + (t * a).sum()
        inputs = list(s_in) + [t for pr in p_in for t in pr] + [a_in]
        gs = torch.autograd.grad(obj, inputs, allow_unused=True)
        gs = [torch.zeros_like(t) if g is None else g for g, t in zip(gs, inputs)]
        adj_state = gs[:4]
        adj_pose, o = [], 4
        for po in p_in:
            adj_pose.append(tuple(gs[o:o + len(po)]))
            o += len(po)
        grad[i] = gs[-1]
    return total, grad, states, poses_l, info


def init_state(x0: np.ndarray):
    """MPMSimulator.reset (mpm_simulator.py:330-341)."""
    N = x0.shape[0]
    x = torch.as_tensor(x0, dtype=DT).clone()
    return (x, torch.zeros(N, 3, dtype=DT), torch.zeros(N, 3, 3, dtype=DT),
            torch.eye(3, dtype=DT).expand(N, 3, 3).clone())


def init_poses(prims):
    """(pos, rot) per primitive; Chopsticks carry their gap as a third entry (primitives.py:131-133)."""
    return [(torch.tensor(p.init_pos, dtype=DT), torch.tensor(p.init_rot, dtype=DT))
            + ((torch.tensor(p.init_gap, dtype=DT),) if p.shape == "Chopsticks" else ()) for p in prims]


def materials(cfg: SimCfg, ys=None):
    N = cfg.n_particles
    ys = cfg.yield_stress if ys is None else ys
    return (torch.full((N,), cfg.mu, dtype=DT), torch.full((N,), cfg.lam, dtype=DT),
            torch.as_tensor(ys, dtype=DT).expand(N).clone())
