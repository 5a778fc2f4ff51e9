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


@dataclass
class HaloFace:
    nbr: int
    za: int
    zb: int
    send: torch.Tensor
    recv: torch.Tensor
    wire_send: torch.Tensor          # = send / recv unless the backend wants host memory
    wire_recv: torch.Tensor


@dataclass
class HaloPlan:
    faces: List[HaloFace]
    ops: list


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
        All fields of all faces travel in ONE batch of point-to-point ops (one latency per substep phase).
        General-purpose form (fresh buffers every call); the per-substep hot path uses ``plan`` / ``run``."""
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

    def plan(self, pack) -> "HaloPlan":
        """Persistent buffers and point-to-point op list of one halo field: ``pack(za, zb)`` returns the face's first
        message, whose storage becomes the face's send buffer for the rest of the run.  Built once per field so that
        a substep costs two kernel launches per face and one ``batch_isend_irecv`` -- no allocation, no op
        construction (the host issue rate, not the link, bounds the slab path at 128^3)."""
        faces, ops = [], []
        for nbr, za, zb in self.layout.faces(self.rank):
            send = pack(za, zb).contiguous()
            recv = torch.empty_like(send)
            if self.stage_host and send.is_cuda:
                ws, wr = (torch.empty(send.shape, dtype=send.dtype, pin_memory=True) for _ in range(2))
            else:
                ws, wr = send, recv
            faces.append(HaloFace(nbr, za, zb, send, recv, ws, wr))
            ops.append(dist.P2POp(dist.isend, ws, nbr, self.group))
            ops.append(dist.P2POp(dist.irecv, wr, nbr, self.group))
        return HaloPlan(faces, ops)

    def run(self, plan: "HaloPlan"):
        """Send every face's ``send`` buffer, receive into its ``recv`` buffer."""
        if not plan.faces:
            return
        for fc in plan.faces:
            if fc.wire_send is not fc.send:
                fc.wire_send.copy_(fc.send)                      # synchronous device -> pinned host copy
        for req in dist.batch_isend_irecv(plan.ops):
            req.wait()
        for fc in plan.faces:
            if fc.wire_recv is not fc.recv:
                fc.recv.copy_(fc.wire_recv, non_blocking=True)   # ordered on the engine's stream before the unpack

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

    def __init__(self, engine, layout: SlabLayout, rank: int, group=None, comm: Optional[HaloComm] = None):
        self._e, self.layout, self.rank = engine, layout, rank
        self.comm = comm if comm is not None else HaloComm(layout, rank, group)
        self.soft_contact = False
        self._plans = {}                           # halo field -> HaloPlan (persistent buffers, built on first use)

    def __getattr__(self, name):                   # everything not overridden goes straight to the engine
        return getattr(self._e, name)

    # ---- halos
    def _halo(self, field, f):
        """Symmetric sum exchange of one halo field.  (The block flags of grid_in need no exchange of their own:
        plmpm_halo_unpack_add marks the block of every node that receives a non-zero value.)"""
        e = self._e
        plan = self._plans.get(field)
        if plan is None:
            plan = self._plans[field] = self.comm.plan(lambda za, zb: e.halo_pack(field, f, za, zb))
        else:
            for fc in plan.faces:
                e.halo_pack(field, f, fc.za, fc.zb, out=fc.send)
        self.comm.run(plan)
        for fc in plan.faces:
            e.halo_unpack_add(field, f, fc.za, fc.zb, fc.recv)

    def _phase(self, field, f, pre, post, chain_in=False, chain_out=False):
        """One exchange-split phase of a substep: pre(f) | halo sum exchange of ``field`` | post(f).  After the first
        call the face buffers are persistent and each side of the exchange is ONE library call (two kernel launches
        before the exchange, two after).  Returns whether g2p(f) was left pending for the next phase (``chain_out``
        honoured)."""
        e = self._e
        plan = self._plans.get(field)
        if plan is None:
            pre(f)
            self._halo(field, f)                # builds the plan
            post(f)
            return False
        e.slab_pre(field, f, plan.faces, chain=chain_in)
        self.comm.run(plan)
        e.slab_post(field, f, plan.faces, chain=chain_out)
        return chain_out

    # ---- hot path
    def step(self, first, n):
        e = self._e
        e.fk(first, n)
        pending = False                         # g2p(f - 1) deferred: it runs fused with p2g(f), as on one GPU
        for f in range(first, first + n):
            pending = self._phase(e.HALO_GRID_IN, f, e.p2g, e.grid_g2p, chain_in=pending, chain_out=f + 1 < first + n)

    def substep(self, f):
        self.step(f, 1)

    def step_grad(self, first, n, step):
        e = self._e
        for f in range(first + n - 1, first - 1, -1):
            self._phase(e.HALO_GRID_OUT_ADJ, f, e.grad_scatter, e.grad_gather)
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
                  target_fn: Optional[Callable] = None, particles: Optional[np.ndarray] = None,
                  layout: Optional[SlabLayout] = None, comm: Optional[HaloComm] = None,
                  xy_margin: Optional[int] = None):
    """Build this rank's ``TaichiEnv`` over its slab of the scene in ``cfg`` (every rank samples the same seed-0
    particle cloud and keeps its own part).  ``target_fn(all_particles, sim) -> (n,n,n) grid`` may supply the loss
    target.  ``xy_margin`` (node layers) shrinks the exchanged halo planes to the xy bounding box of the whole
    particle cloud at reset plus that margin -- a body that moves further than the margin raises, like one that
    drifts out of its slab; ``None`` sends whole planes.  ``layout`` / ``comm`` override the balanced cut and the torch.distributed communicator (measurement
    tools: profiles/tools/slab_host_cost.py).  Returns (env, layout, owned_index)."""
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
    if layout is None:
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
        if xy_margin is not None:
            b = (x_all[:, :2] * n_grid - 0.5).astype(np.int64)           # stencil bases, same on every rank
            lo = np.maximum(b.min(0) - int(xy_margin), 0)
            hi = np.minimum(b.max(0) + 3 + int(xy_margin), n_grid)
            sim.engine.set_halo_window(lo[0], hi[0], lo[1], hi[1])
        sim.engine = SlabEngine(sim.engine, layout, rank, group, comm)
        env.primitives._bind(sim.engine)
    env.simulator = sim
    env.renderer = None
    env.loss = Loss(cfg.ENV.loss, sim)
    env._is_copy, env._tape = True, None
    env.initialize()
    if target_fn is not None:
        env.loss.load_target_density(grids=target_fn(x_all, sim))
    return env, layout, mine
