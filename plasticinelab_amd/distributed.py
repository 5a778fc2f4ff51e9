"""Multi-GPU MPM: z-slab domain decomposition, one process per GPU, halos over RCCL/xGMI.

What is built (round 1)
* The grid is cut into z-slabs of node layers; rank r owns nodes ``z in [z_r, z_{r+1})`` and -- for the whole
  rollout -- the particles whose stencil base lay in that slab at reset (**fixed ownership**).  The reference has
  no multi-GPU path at all; this is new design.
* Every rank allocates the full n^3 index space (only touched blocks cost traffic; 288 GB of HBM make the
  allocation irrelevant) but scatters only its own particles.  After ``p2g`` the ``2*halo`` node planes around each
  slab face hold partial sums on both neighbours; one **symmetric sum exchange** per face (both sides send their
  copy, both add what they receive) makes them complete on both, so the pointwise ``grid_op`` and the ``g2p``
  gather need no further communication.  The reverse pass mirrors it on ``grid_v_out.grad``.
* Pose adjoints are counted on owned nodes only, summed over ranks once per env step, then the (tiny, serial)
  kinematics-chain adjoint runs redundantly everywhere, so every rank ends with the full action gradient.
* The loss sums over owned nodes / local particles and all-reduces a 32-double record.

Limits, stated plainly
* No particle migration yet: a particle whose stencil leaves ``[z_r - halo, z_{r+1} + halo)`` raises
  (``Engine.check_error``).  Ownership goes by the stencil centre, so ``halo`` node layers (default 4) allow
  ``halo - 1`` layers of drift either way during a rollout; slabs must be at least ``2*halo`` thick, which caps the rank count for thin bodies (config 3: the cube spans 40 layers).
* The per-substep exchange is host-driven (two ``plmpm_*`` phase calls + one ``batch_isend_irecv``), so small
  grids are latency-bound; overlapping the halo with interior blocks is future work.

The communication layer is backend-agnostic (``nccl`` = RCCL on the GPUs; ``gloo`` for the CPU tests and for
two ranks sharing one GPU, staged through host memory).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

# how each slot of the 32-double loss record combines across ranks (plmpm_loss_partials)
_SUM_SLOTS = (0, 1, 3, 4)
_MAX_SLOTS = (2,)


@dataclass
class SlabLayout:
    n_grid: int
    bounds: Tuple[int, ...]          # len world+1, node z indices, bounds[0] = 0, bounds[-1] = n_grid
    halo: int

    @property
    def world(self):
        return len(self.bounds) - 1

    def slab(self, rank):
        return self.bounds[rank], self.bounds[rank + 1]

    def owner_of(self, base_z: np.ndarray) -> np.ndarray:
        """Owner = slab holding the stencil's CENTRE node (base + 1): the 3-wide stencil then reaches exactly one
        layer beyond either face at reset, leaving ``halo - 1`` layers of drift margin on both sides."""
        return np.clip(np.searchsorted(np.asarray(self.bounds[1:-1]), np.asarray(base_z) + 1, side="right"), 0, self.world - 1)

    def faces(self, rank) -> List[Tuple[int, int, int]]:
        """(neighbour rank, za, zb): node planes exchanged with each neighbour."""
        out = []
        z0, z1 = self.slab(rank)
        if rank > 0:
            out.append((rank - 1, max(z0 - self.halo, 0), min(z0 + self.halo, self.n_grid)))
        if rank < self.world - 1:
            out.append((rank + 1, max(z1 - self.halo, 0), min(z1 + self.halo, self.n_grid)))
        return out

    @staticmethod
    def stencil_base_z(x: np.ndarray, n_grid: int) -> np.ndarray:
        return (np.asarray(x)[:, 2] * n_grid - 0.5).astype(np.int64)      # trunc, as in the kernels

    @classmethod
    def balanced(cls, x: np.ndarray, n_grid: int, world: int, halo: int = 4) -> "SlabLayout":
        """Slab faces at particle-count quantiles of the stencil base z, widened so every slab is >= 2*halo thick."""
        if world == 1:
            return cls(n_grid, (0, n_grid), 0)
        bz = np.sort(cls.stencil_base_z(x, n_grid)) + 1        # stencil centres
        cuts = [int(bz[min(len(bz) - 1, (len(bz) * r) // world)]) for r in range(1, world)]
        lo, hi = int(bz[0]), int(bz[-1]) + 3
        min_th = max(2 * halo, 2)
        # enforce monotone faces with the minimum thickness, sweeping up then down
        faces = [0] + cuts + [n_grid]
        for i in range(1, world):
            faces[i] = max(faces[i], faces[i - 1] + min_th)
        for i in range(world - 1, 0, -1):
            faces[i] = min(faces[i], faces[i + 1] - min_th)
        if any(faces[i + 1] - faces[i] < min_th for i in range(world)):
            raise ValueError(f"cannot cut {n_grid} layers into {world} slabs of >= {min_th} layers (body spans z {lo}..{hi})")
        return cls(n_grid, tuple(faces), halo)


class HaloComm:
    """Symmetric sum exchange with the z-neighbours through ``torch.distributed`` point-to-point ops."""

    def __init__(self, layout: SlabLayout, rank: int, group=None):
        self.layout, self.rank, self.group = layout, rank, group
        self.stage_host = dist.get_backend(group) == "gloo"      # gloo P2P wants host tensors
        # small host records (loss sums) are reduced on the device when the backend is RCCL
        self.scalar_device = torch.device("cpu") if self.stage_host else torch.device("cuda", torch.cuda.current_device())

    def _wire(self, t: torch.Tensor) -> torch.Tensor:
        return t.cpu() if (self.stage_host and t.is_cuda) else t

    def exchange(self, *fields):
        """fields: (pack, unpack_add) pairs, ``pack(za, zb) -> tensor`` and ``unpack_add(za, zb, tensor)``.
        All fields of all faces travel in ONE batch of point-to-point ops (one latency per substep phase)."""
        faces = self.layout.faces(self.rank)
        if not faces:
            return
        work, ops = [], []
        for nbr, za, zb in faces:
            for pack, unpack_add in fields:
                s = self._wire(pack(za, zb)).contiguous()
                r = torch.empty_like(s)
                work.append((za, zb, r, s, unpack_add))
                ops.append(dist.P2POp(dist.isend, s, nbr, self.group))
                ops.append(dist.P2POp(dist.irecv, r, nbr, self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for za, zb, r, _s, unpack_add in work:
            unpack_add(za, zb, r)

    def all_reduce_(self, t: torch.Tensor, op=dist.ReduceOp.SUM):
        if self.layout.world == 1:
            return t
        if self.stage_host and t.is_cuda:
            h = t.cpu()
            dist.all_reduce(h, op=op, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op, group=self.group)
        return t

    def reduce_loss_record(self, rec: np.ndarray, soft_contact: bool, phase: int) -> np.ndarray:
        """Combine the 32-double partial record of ``plmpm_loss_partials`` across ranks."""
        if self.layout.world == 1:
            return rec
        t = torch.as_tensor(rec, dtype=torch.float64).clone().to(self.scalar_device)
        out = t.clone()
        s = t.clone(); dist.all_reduce(s, op=dist.ReduceOp.SUM, group=self.group)
        if phase == 0:
            mx = t.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=self.group)
            mn = t.clone(); dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=self.group)
            for i in _SUM_SLOTS:
                out[i] = s[i]
            for i in _MAX_SLOTS:
                out[i] = mx[i]
            if soft_contact:
                out[16:24] = s[16:24]                 # dist_norm
            else:
                out[8:16] = mn[8:16]                  # hard contact: min over all particles
        else:
            out[8:16] = s[8:16]                       # soft contact: weighted sum (local sums used the global norm)
        return out.cpu().numpy()


class SlabEngine:
    """Proxy around one rank's ``Engine`` that turns ``step`` / ``step_grad`` / ``loss_*`` into the phase-split,
    halo-exchanging versions.  ``MPMSimulator``, ``Loss`` and ``Tape`` work on it unchanged."""

    def __init__(self, engine, layout: SlabLayout, rank: int, group=None):
        self._e, self.layout, self.rank = engine, layout, rank
        self.comm = HaloComm(layout, rank, group)
        self.soft_contact = False

    def __getattr__(self, name):                   # everything not overridden goes straight to the engine
        return getattr(self._e, name)

    # ---- halos
    def _halo(self, field, f):
        """Symmetric sum exchange of one halo field.  (The block flags of grid_in need no exchange of their own:
        plmpm_halo_unpack_add marks the block of every node that receives a non-zero value.)"""
        e = self._e
        self.comm.exchange((lambda za, zb: e.halo_pack(field, f, za, zb),
                            lambda za, zb, buf: e.halo_unpack_add(field, f, za, zb, buf.to(e.device))))

    # ---- hot path
    def step(self, first, n):
        e = self._e
        e.fk(first, n)
        for f in range(first, first + n):
            e.p2g(f)
            self._halo(e.HALO_GRID_IN, f)
            e.grid_g2p(f)

    def substep(self, f):
        self.step(f, 1)

    def step_grad(self, first, n, step):
        e = self._e
        for f in range(first + n - 1, first - 1, -1):
            e.grad_scatter(f)
            self._halo(e.HALO_GRID_OUT_ADJ, f)
            e.grad_gather(f)
        for view in e.pose_grad_views(first, n + 1):      # position, rotation, (Chopsticks) gap adjoints
            self.comm.all_reduce_(view)
        e.chain_grad(first, n, step)

    def substep_grad(self, f):
        raise NotImplementedError("multi-GPU runs differentiate whole env steps (step_grad)")

    # ---- loss
    def loss_set_weights(self, sdf, density, contact, soft_contact):
        self.soft_contact = bool(soft_contact)
        self._e.loss_set_weights(sdf, density, contact, soft_contact)

    def _loss_globals(self, f):
        e = self._e
        e.loss_scatter(f)
        self._halo(e.HALO_LOSS_MASS, f)
        g = self.comm.reduce_loss_record(e.loss_partials(f, 0), self.soft_contact, 0)
        if self.soft_contact:
            e.loss_set_globals(g)
            g = self.comm.reduce_loss_record(e.loss_partials(f, 1), True, 1)
        return g

    def loss_forward(self, f):
        g = self._loss_globals(f)
        # the error word is combined over the ranks before anyone raises: a rank that bailed out alone would leave
        # the others waiting in their next exchange
        flags = torch.tensor([float(self._e.error_flags())], dtype=torch.float64, device=self.comm.scalar_device)
        self.comm.all_reduce_(flags, op=dist.ReduceOp.MAX)
        self._e.check_error(int(flags.item()))
        return self._e.loss_finish(g)

    def loss_backward(self, f):
        g = self._loss_globals(f)
        self._e.loss_set_globals(g)
        self._e.loss_backward_local(f)

    def grid_mass(self, f):
        raise NotImplementedError("grid_mass on a slab engine returns only this rank's partial grid")


def make_slab_env(cfg, rank: int, world: int, *, halo: int = 4, compute_dtype=None, device=None, group=None,
                  target_fn: Optional[Callable] = None, particles: Optional[np.ndarray] = None):
    """Build this rank's ``TaichiEnv`` over its slab of the scene in ``cfg`` (every rank samples the same seed-0
    particle cloud and keeps its own part).  ``target_fn(all_particles, sim) -> (n,n,n) grid`` may supply the loss
    target.  Returns (env, layout, owned_index)."""
    from .engine import taichi_env as te
    from .engine.losses import Loss
    from .engine.mpm_simulator import MPMSimulator
    from .engine.primitives import Primitives
    from .engine.shapes import Shapes

    x_all, colors = Shapes(cfg.SHAPES).get()
    if particles is not None:                       # caller-chosen cloud (e.g. a subsample), same on every rank
        x_all = np.ascontiguousarray(particles, np.float64)
        colors = np.zeros(len(x_all), np.int32)
    quality = cfg.SIMULATOR.quality * 0.5
    n_grid = int(128 * quality)
    layout = SlabLayout.balanced(x_all, n_grid, world, halo)
    owner = layout.owner_of(SlabLayout.stencil_base_z(x_all, n_grid))
    mine = np.nonzero(owner == rank)[0]
    if len(mine) == 0:
        raise ValueError(f"rank {rank} owns no particles")

    env = te.TaichiEnv.__new__(te.TaichiEnv)
    env.cfg = cfg.ENV
    env.primitives = Primitives(cfg.PRIMITIVES, max_timesteps=int(cfg.SIMULATOR.max_steps))
    env.shapes = None
    env.init_particles, env.particle_colors = np.ascontiguousarray(x_all[mine]), colors[mine]
    env.all_particles = x_all
    cfg.SIMULATOR.n_particles = len(mine)
    cfg.SIMULATOR["store_grid"] = True
    env.n_particles = len(mine)
    z0, z1 = layout.slab(rank)
    sim = MPMSimulator(cfg.SIMULATOR, env.primitives, compute_dtype=compute_dtype, device=device,
                       slab=(z0, z1), slab_halo=layout.halo if world > 1 else 0)
    if world > 1:
        sim.engine = SlabEngine(sim.engine, layout, rank, group)
        env.primitives._bind(sim.engine)
    env.simulator = sim
    env.renderer = None
    env.loss = Loss(cfg.ENV.loss, sim)
    env._is_copy, env._tape = True, None
    env.initialize()
    if target_fn is not None:
        env.loss.load_target_density(grids=target_fn(x_all, sim))
    return env, layout, mine
