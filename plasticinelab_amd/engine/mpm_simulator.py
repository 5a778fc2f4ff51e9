"""MI355X drop-in for plb.engine.mpm_simulator.MPMSimulator.

Same constructor, attributes and methods as the reference class
(/root/reference/plb/engine/mpm_simulator.py:4-392); the arithmetic runs in the
HIP engine (plasticinelab_amd/csrc) through the C ABI (include/plmpm.h).  State
crosses this boundary exactly as in the reference: float64 numpy arrays
``[x (N,3), v (N,3), F (N,3,3), C (N,3,3), primitive states...]``.

Differences that are deliberate and visible:
* ``cfg.dtype`` stays the I/O dtype ("float64"); the arithmetic type of the GPU
  hot path is ``cfg.compute_dtype`` ("float32" default, "float64" for parity runs).
* adjoints of only two frames are resident (the reference keeps ``.grad`` for all
  ``max_steps`` frames); ``substep_grad`` must therefore be called in reverse
  frame order after ``grad_begin`` -- which is what every reference caller does.
"""
from __future__ import annotations

import numpy as np

from .core import Engine


class _FrameField:
    """`field[f]` -> (N, ...) array of frame f, `field[f, i]` -> one particle's value, `field.to_numpy()` -> frames
    0 .. cur stacked (what `ti.field.to_numpy()` of the reference's (max_steps, n_particles) fields holds, up to the frames
    computed so far).  Reads go through plmpm_get_frame (a device -> host copy): for inspection, not for hot loops."""

    def __init__(self, sim, name):
        self._sim, self._name = sim, name

    def __getitem__(self, key):
        if isinstance(key, tuple):
            f, rest = key[0], key[1:]
            return self._sim.engine.get_frame(int(f), want=(self._name,))[self._name][rest if len(rest) > 1 else rest[0]]
        return self._sim.engine.get_frame(int(key), want=(self._name,))[self._name]

    def to_numpy(self):
        return np.stack([self[f] for f in range(self._sim.cur + 1)])


class MPMSimulator:
    def __init__(self, cfg, primitives=(), compute_dtype=None, device=None, slab=None, slab_halo=0, grid_window=None,
                 particle_capacity=None):
        dim = self.dim = cfg.dim
        assert dim == 3, "only the 3-D path exists (the reference's 2-D branches are dead code, SURVEY section 2)"
        assert cfg.dtype == "float64"                         # mpm_simulator.py:8 (host I/O dtype)
        self.dtype = np.float64
        self.compute_dtype = compute_dtype or cfg.get("compute_dtype", "float32")
        self._yield_stress = cfg.yield_stress
        self.ground_friction = cfg.ground_friction
        self.default_gravity = tuple(cfg.gravity)
        self.n_primitive = len(primitives)

        quality = cfg.quality * 0.5                            # :15-17
        self.n_particles = int(cfg.n_particles)
        self.n_grid = int(128 * quality)                       # :19
        self.dx, self.inv_dx = 1 / self.n_grid, float(self.n_grid)
        self.dt = 0.5e-4 / quality                             # :22
        self.p_vol, self.p_rho = (self.dx * 0.5) ** 2, 1       # :23
        self.p_mass = self.p_vol * self.p_rho
        E, nu = cfg.E, cfg.nu
        self._mu, self._lam = E / (2 * (1 + nu)), E * nu / ((1 + nu) * (1 - 2 * nu))   # :28
        self.max_steps = int(cfg.max_steps)
        self.substeps = int(2e-3 // self.dt)                   # :34
        self.res = (self.n_grid,) * 3
        self.primitives = primitives
        self.cur = 0

        descr = [p.describe() for p in primitives]
        self.engine = Engine(n_grid=self.n_grid, n_particles=self.n_particles, max_frames=self.max_steps,
                             substeps=self.substeps, dt=self.dt, p_vol=self.p_vol, p_mass=self.p_mass,
                             gravity=self.default_gravity, ground_friction=self.ground_friction, primitives=descr,
                             dtype=self.compute_dtype, svd_grad_clamp=float(cfg.get("svd_grad_clamp", 1e-6)),
                             device=device, slab=slab, slab_halo=slab_halo,
                             store_grid=cfg.get("store_grid", "auto"),
                             grid_window=grid_window if grid_window is not None else cfg.get("grid_window", None),
                             particle_capacity=particle_capacity, deterministic=bool(cfg.get("deterministic", False)),
                             contact_min_adjoint=str(cfg.get("contact_min_adjoint", "add")), minmax_tie=str(cfg.get("minmax_tie", "second")),
                             grid_workgroups=int(cfg.get("grid_workgroups", 0)))
        if hasattr(primitives, "_bind"):
            primitives._bind(self.engine)
        self._mats = None

    # ------------------------------------------------------------------ setup
    def initialize(self):                                      # :53-57
        self.set_materials(self._mu, self._lam, self._yield_stress)

    def set_materials(self, mu, lam, yield_stress):
        """Per-particle Lame parameters / yield stress (the reference's mu/lam/yield_stress fields)."""
        self._mats = (mu, lam, yield_stress)
        self.engine.set_materials(mu, lam, yield_stress)

    # ------------------------------------------------------------------ hot path
    def substep(self, s):                                      # :245-257
        self.engine.substep(s)

    def grad_begin(self, last_frame):
        """What ``ti.Tape.__enter__`` does for this path: clear every adjoint; ``last_frame`` is the seed frame."""
        self.engine.grad_begin(last_frame)

    def substep_grad(self, s):                                 # :260-278
        self.engine.substep_grad(s)

    def step(self, is_copy, action=None):                      # :365-376
        start = 0 if is_copy else self.cur
        self.cur = start + self.substeps
        if action is not None:
            self.primitives.set_action(start // self.substeps, self.substeps, action)
        self.engine.step(start, self.substeps)
        if is_copy:
            self.engine.copy_frame(self.cur, 0)
            self.cur = 0

    def step_grad(self, first_frame, step):
        """Reverse of one ``step``: substep_grad over its frames, then forward_kinematics.grad and
        set_velocity.grad of env step ``step`` (what the Tape replays, solver.py:36-44)."""
        self.engine.step_grad(first_frame, self.substeps, step)

    # ------------------------------------------------------------------ state I/O
    def get_state(self, f):                                    # :314-323
        fr = self.engine.get_frame(f)
        out = [fr["x"], fr["v"], fr["F"], fr["C"]]
        for p in self.primitives:
            out.append(p.get_state(f))
        return out

    def set_state(self, f, state, resort=None):                # :325-328
        if resort is None:
            resort = (f == 0)
        self.engine.set_frame(f, x=state[0], v=state[1], F=state[2], C_=state[3], resort=resort)
        if resort and self._mats is not None:
            self.engine.set_materials(*self._mats)             # material arrays follow the storage order
        for s, p in zip(state[4:], self.primitives):
            p.set_state(f, s)

    def reset(self, x):                                        # :330-341
        N = self.n_particles
        x = np.asarray(x, np.float64).reshape(N, 3)
        F = np.broadcast_to(np.eye(3), (N, 3, 3)).copy()
        self.engine.set_frame(0, x=x, v=np.zeros((N, 3)), F=F, C_=np.zeros((N, 3, 3)), resort=True)
        if self._mats is not None:
            self.engine.set_materials(*self._mats)
        self.cur = 0

    # The reference exposes its particle state as Taichi fields (`sim.x[f, i]`, `sim.v`, `sim.C`, `sim.F`: mpm_simulator.py:35-38)
    # and a few callers read them as such (losses/loss.py:18-19, the notebook).  Read-only views with the same indexing:
    @property
    def x(self):
        return _FrameField(self, "x")

    @property
    def v(self):
        return _FrameField(self, "v")

    @property
    def C(self):
        return _FrameField(self, "C")

    @property
    def F(self):
        return _FrameField(self, "F")

    def get_x(self, f):                                        # :349-352
        return self.engine.get_frame(f, want=("x",))["x"]

    def get_v(self, f):                                        # :360-363
        return self.engine.get_frame(f, want=("v",))["v"]

    def compute_grid_m_kernel(self, f):                        # :382-392
        """Mass-only scatter of frame ``f``; returns the (n,n,n) float64 grid (the reference fills ``grid_m``)."""
        self._grid_m = self.engine.grid_mass(f)
        return self._grid_m

    @property
    def grid_m(self):
        return getattr(self, "_grid_m", None)
