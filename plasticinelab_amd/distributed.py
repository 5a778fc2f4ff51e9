"""Multi-GPU MPM: z-slab domain decomposition, one process per GPU, halos and migrating particles over RCCL/xGMI.

The reference has no multi-GPU path; this is new design.

* **Slabs.**  The grid is cut into z-slabs whose faces sit on multiples of 4 node layers (the grid is stored in 4^3
  blocks, z block index slowest).  Rank r owns the nodes ``z in [z_r, z_{r+1})`` and the particles whose stencil
  CENTRE node lies in that range.
* **Memory.**  A rank allocates only a *window* of the grid -- the xy bounding box of the body plus a margin, and in z
  its slab plus the exchanged planes -- and keeps the per-frame grid store on that window (``plmpm_config.grid_lo /
  grid_hi``): 32 B x window nodes per frame instead of 32 B x n^3.  That is what lets 256^3 and 512^3 rollouts stay
  in store mode (no forward recompute in ``substep_grad``).
* **Halos, zero copy.**  After ``p2g`` the block planes next to a face hold partial sums on both neighbours.  One
  block plane either side of the face (4 node layers: the one-layer stencil reach plus three layers of drift between
  migrations) is exchanged: each rank sends its copy straight out of the grid arrays -- a range of block planes is
  contiguous per SoA component, so there is no pack kernel -- and ``grid_op`` adds the received copy on first touch,
  so there is no unpack kernel either (symmetric sum exchange: both sides end with complete sums, ``g2p`` needs no
  further communication).  The reverse pass mirrors it on ``grid_v_out.grad``.  A fwd+bwd substep is the same five
  kernel launches as on one GPU plus two ``batch_isend_irecv`` calls.
* **Migration.**  Every ``migrate_every`` env steps, before the step starts, the rows whose stencil centre has left the
  slab are packed on the device and sent to the neighbour; arrivals are merged in and the whole set is re-sorted
  along the Hilbert curve into a new storage epoch (this is also the slab engines' cell re-sort).  The reverse sweep
  sends the adjoint rows of the arrivals back where they came from, so gradients are those of the single-GPU run.
  Particle identity is a global id that travels with the row.
* Pose adjoints are counted on owned nodes only, summed over ranks once per env step, then the (tiny, serial)
  kinematics-chain adjoint runs redundantly everywhere, so every rank ends with the full action gradient.  The loss
  sums over owned nodes / local particles and all-reduces a 32-double record.

Limits, stated plainly: slabs are whole block planes.  At the default reach (a particle may drift 3 layers past its slab
between two migrations; it raises otherwise) a slab is at least two planes thick; when a body is too thin for that --
config 3's cube spans ten planes, so at most 5 such ranks -- the layout falls back to a reach of 2 layers (one of drift)
and slabs of ONE plane, whose plane is exchanged with both neighbours (``min_thickness``): up to one rank per block plane
of the body, at the price of a coarser load balance (8 ranks over ten planes: the largest slab holds 2/10 of the
particles).

* **Who drives the exchange.**  Two transports, same planes, same kernels consuming them.  *Peer writes* (opt-in:
  ``peer=True`` / ``PLB_PEER_HALOS=1``; ``bench.py`` opts in at N > 1 and checks them against the library transport on one
  env step before it times anything): every rank allocates its receive areas in uncached (else fine-grained)
  device memory, the neighbours map them through IPC handles (exchanged once, over ``torch.distributed``), and an
  exchange is ONE kernel on the engine's stream -- copy into the neighbours' areas, publish an arrival counter, wait
  for theirs (``csrc/plmpm_peer.hip``).  The substep loops of an env step are then native (``plmpm_slab_step`` /
  ``plmpm_slab_step_grad``) and only enqueue: no Python, no communication-library call, no request object per substep.
  *torch.distributed point-to-point* (``batch_isend_irecv`` of zero-copy views; the fallback, and what gloo-staged test
  ranks without IPC use): issued by the host each substep; ``SlabEngine(overlap=True)`` runs the grid kernels of the
  blocks outside the exchanged planes while it is in flight.

The communication layer is backend-agnostic (``nccl`` = RCCL on the GPUs; ``gloo`` for the CPU tests and for ranks
sharing one GPU, staged through host memory).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

import os

import numpy as np
import torch
import torch.distributed as dist

# how each slot of the 32-double loss record combines across ranks (plmpm_loss_partials)
_SUM_SLOTS = (0, 1, 3, 4)
_MAX_SLOTS = (2,)

HALO_PLANES = 1          # block planes exchanged either side of a face (4 node layers)
MIN_THICKNESS = 8        # node layers of a slab at the default reach (halo = 4); see min_thickness


def min_thickness(halo: int) -> int:
    """Thinnest slab (node layers, a multiple of 4) for particles that may reach ``halo`` layers beyond their slab.

    A node layer must not collect contributions from more than two ranks -- the exchange is pairwise: each rank adds
    the copy of its neighbour across the face, nothing is forwarded -- so the reach of the slab below and of the slab
    above may not meet inside a slab: thickness >= 2 * halo.  halo = 4 (one layer of stencil + three of drift between
    migrations): two block planes, the exchange planes of the two faces are disjoint.  halo = 2 (one of stencil + one
    of drift: particles move a fraction of a cell per env step, so this holds with a migration every env step): a
    slab may be a single block plane, which then lies in the exchange range of BOTH faces -- it is sent to both
    neighbours and the grid kernels add both received copies."""
    if halo not in (2, 4):
        raise ValueError(f"slab reach must be 2 or 4 node layers (got {halo})")
    return 4 * ((2 * halo + 3) // 4)


@dataclass
class SlabLayout:
    n_grid: int
    bounds: Tuple[int, ...]          # len world+1, node z indices (multiples of 4), bounds[0] = 0, bounds[-1] = n_grid
    halo: int = 4 * HALO_PLANES      # node layers a rank's particles may reach beyond its slab

    @property
    def world(self):
        return len(self.bounds) - 1

    def slab(self, rank):
        return self.bounds[rank], self.bounds[rank + 1]

    def owner_of(self, base_z: np.ndarray) -> np.ndarray:
        """Owner = slab holding the stencil's CENTRE node (base + 1): the 3-wide stencil then reaches exactly one
        layer beyond either face when ownership is (re)assigned."""
        return np.clip(np.searchsorted(np.asarray(self.bounds[1:-1]), np.asarray(base_z) + 1, side="right"), 0, self.world - 1)

    def faces(self, rank) -> List[Tuple[int, int, int]]:
        """(neighbour rank, bz_a, bz_b): block planes exchanged with each neighbour (down first, then up)."""
        out = []
        z0, z1 = self.slab(rank)
        if rank > 0:
            out.append((rank - 1, z0 // 4 - HALO_PLANES, z0 // 4 + HALO_PLANES))
        if rank < self.world - 1:
            out.append((rank + 1, z1 // 4 - HALO_PLANES, z1 // 4 + HALO_PLANES))
        return out

    @staticmethod
    def stencil_base_z(x: np.ndarray, n_grid: int) -> np.ndarray:
        return (np.asarray(x)[:, 2] * n_grid - 0.5).astype(np.int64)      # trunc, as in the kernels

    @classmethod
    def balanced(cls, x: np.ndarray, n_grid: int, world: int, halo: Optional[int] = 4) -> "SlabLayout":
        """Slab faces at particle-count quantiles of the stencil centre z, rounded to multiples of 4 and pushed apart
        so that every slab is >= min_thickness(halo) layers thick.  ``halo=None``: 4 if the body can be cut that way,
        else 2 (thin slabs: up to one rank per block plane of the body)."""
        if world == 1:
            return cls(n_grid, (0, n_grid), 0)
        if halo is None:
            try:
                return cls.balanced(x, n_grid, world, 4)
            except ValueError:
                return cls.balanced(x, n_grid, world, 2)
        thick = min_thickness(halo)
        cz = cls.stencil_base_z(x, n_grid) + 1                 # stencil centres
        # particles per block plane; a face may sit on any multiple of 4.  Choose the world - 1 faces, at least
        # MIN_THICKNESS apart, that minimise the largest slab's particle count (dynamic programme over the planes)
        # with no slab left empty.
        nbp = n_grid // 4
        cnt = np.bincount(np.clip(cz // 4, 0, nbp - 1), minlength=nbp).astype(np.int64)
        cum = np.concatenate([[0], np.cumsum(cnt)])           # cum[j] = particles in planes [0, j)
        gap = thick // 4
        INF = np.iinfo(np.int64).max
        # best[k][j]: smallest possible maximum load of k slabs covering planes [0, j), the k-th ending at face j
        best = np.full((world + 1, nbp + 1), INF, dtype=np.int64)
        prev = np.zeros((world + 1, nbp + 1), dtype=np.int64)
        best[0][0] = 0
        for k in range(1, world + 1):
            for j in range(k * gap, nbp + 1):
                for i in range((k - 1) * gap, j - gap + 1):
                    if best[k - 1][i] == INF:
                        continue
                    load = cum[j] - cum[i]
                    if load <= 0:
                        continue                              # an empty slab is useless
                    m = max(best[k - 1][i], load)
                    if m < best[k][j]:
                        best[k][j], prev[k][j] = m, i
        if best[world][nbp] == INF:
            raise ValueError(f"cannot cut the body (stencil centres z {int(cz.min())}..{int(cz.max())}) into {world} non-empty slabs of >= "
                             f"{thick} layers with faces on multiples of 4")
        faces, j = [n_grid], nbp
        for k in range(world, 0, -1):
            j = int(prev[k][j])
            faces.append(4 * j)
        faces = faces[::-1]
        return cls(n_grid, tuple(faces), halo)


class HaloComm:
    """Point-to-point traffic with the two z-neighbours through ``torch.distributed``: the per-substep symmetric sum
    exchange of block planes (zero-copy send views, persistent receive buffers, op lists cached per frame) and the
    row exchanges of particle migration."""

    def __init__(self, layout: SlabLayout, rank: int, group=None, peer: Optional[bool] = None):
        self.layout, self.rank, self.group = layout, rank, group
        # peer writes are OPT-IN (peer=True, or PLB_PEER_HALOS=1; PLB_PEER_HALOS=0 forbids them): the device-side exchange has
        # only ever been validated with all ranks on one GPU, and a visibility bug across real GPUs would give wrong halos
        # silently, not an error.  Callers that can afford the check opt in and compare transports first (bench.py at N > 1:
        # one env step fwd + bwd through each, same loss and action gradient or the library transport is used).
        env = os.environ.get("PLB_PEER_HALOS", "")
        self.want_peer = (env == "1") if env in ("0", "1") else bool(peer)
        self.peer_ready = False
        self.stage_host = dist.get_backend(group) == "gloo"      # gloo P2P wants host tensors
        # small host records (loss sums) are reduced on the device when the backend is RCCL
        self.scalar_device = torch.device("cpu") if self.stage_host else torch.device("cuda", torch.cuda.current_device())
        self.backend = dist.get_backend(group)
        self._recv = {}          # field -> [tensor [ncomp, count] per face]
        self._ops = {}           # (field, frame) -> cached P2POp list
        self.down = rank - 1 if rank > 0 else None
        self.up = rank + 1 if rank < layout.world - 1 else None

    # ---- halos
    def attach(self, engine, field, f=0):
        """Allocate and register the receive buffers of ``field`` (once)."""
        faces = self.layout.faces(self.rank)
        bufs = []
        for _nbr, a, b in faces:
            cnt = engine.halo_views(field, f, a, b)[0].numel()
            bufs.append(torch.zeros(engine.halo_ncomp(field), cnt, dtype=engine.torch_dtype, device=engine.device))
        engine.halo_set_recv(field, [(a, b) for _n, a, b in faces], bufs)
        self._recv[field] = bufs
        self._engine = engine

    def setup_peer(self, engine):
        """Collective, once: receive areas for the three halo fields in fine-grained device memory, IPC handles swapped
        with the neighbours, their areas mapped here.  True if EVERY rank got through (otherwise nobody uses the peer
        path: the point-to-point exchange below works without it)."""
        if self.peer_ready or not self.want_peer or self.layout.world < 2:
            return self.peer_ready
        faces = self.layout.faces(self.rank)
        fields = (engine.HALO_GRID_IN, engine.HALO_GRID_OUT_ADJ, engine.HALO_LOSS_MASS)
        mine, local, err = {}, {}, ""
        # What the first run on real neighbours needs to know when something goes wrong (bench.py: transport_check.preflight): which
        # device every rank sits on, whether the runtime lets this device reach the neighbours', what kind of memory the receive
        # areas got, whether the IPC handles opened, and whether ONE hand-off of the exchange's own pattern crossed each face.
        device = torch.cuda.current_device() if torch.cuda.is_available() else -1
        pre = self.preflight = {"rank": self.rank, "gpu": list(gpu_identity()) + [device], "neighbours": {}, "alloc": None, "ipc_open": None, "ping": None}
        try:
            for field in fields:
                for nbr, a, b in faces:
                    ptr, handle = engine.peer_alloc(field, a, b)
                    local[(field, nbr)] = ptr
                    mine[(field, nbr)] = (handle, a, b)
            pre["alloc"] = engine.peer_memory_kind() if hasattr(engine, "peer_memory_kind") else "ok"
        except Exception as e:                                   # noqa: BLE001 -- e.g. no IPC in this environment
            err = f"{type(e).__name__}: {e}"
            pre["alloc"] = "FAILED: " + err[:160]
        everyone = [None] * self.layout.world
        dist.all_gather_object(everyone, {"rank": self.rank, "areas": mine, "err": err, "gpu": pre["gpu"]}, group=self.group)
        ok = not any(r["err"] for r in everyone)
        for nbr, _a, _b in faces:
            theirs = everyone[nbr]["gpu"]
            same_host = theirs[0] == pre["gpu"][0]
            if theirs[:2] == pre["gpu"][:2]:
                access = "same device"
            elif same_host and device >= 0 and theirs[2] >= 0:
                try:
                    access = bool(torch.cuda.can_device_access_peer(device, theirs[2]))      # hipDeviceCanAccessPeer
                except Exception as e:                           # noqa: BLE001
                    access = f"query failed: {type(e).__name__}"
            else:
                access = "other host" if not same_host else "no device"
            pre["neighbours"][str(nbr)] = {"gpu": theirs, "can_access_peer": access}
        remote = {}
        if ok:
            try:
                for field in fields:
                    for nbr, a, b in faces:
                        handle, ra, rb = everyone[nbr]["areas"][(field, self.rank)]
                        assert (ra, rb) == (a, b), "both sides of a face exchange the same block planes"
                        remote[(field, nbr)] = engine.peer_open(handle)
            except Exception as e:                               # noqa: BLE001
                err = f"{type(e).__name__}: {e}"
        pre["ipc_open"] = ("ok" if ok and not err else "FAILED: " + (err or "a receive area could not be allocated on another rank")[:160])
        flags = [None] * self.layout.world
        dist.all_gather_object(flags, err, group=self.group)
        if any(flags):
            self.peer_error = next(f for f in flags if f)
            return False
        for field in fields:
            engine.halo_peer_setup(field, [(a, b) for _n, a, b in faces], [local[(field, n)] for n, _a, _b in faces],
                                   [remote[(field, n)] for n, _a, _b in faces])
        # one hand-off across every face, the exchange kernels' own pattern on the real areas (plmpm_peer_ping); a face that stays
        # silent is reported, not fatal: bench.py's transport check decides what the halos travel on
        if hasattr(engine, "peer_ping"):
            try:
                got = engine.peer_ping(engine.HALO_GRID_IN, 0x600D0001, float(os.environ.get("PLB_PEER_PING_TIMEOUT", "5")))
                pre["ping"] = {str(nbr): {"arrived": got[i][0], "wait_us": round(got[i][1], 2)} for i, (nbr, _a, _b) in enumerate(faces)}
            except Exception as e:                               # noqa: BLE001
                pre["ping"] = "FAILED: " + f"{type(e).__name__}: {e}"[:160]
            dist.barrier(group=self.group)                       # nobody resets or exchanges before every token has been seen
        spoil = os.environ.get("PLB_TEST_PEER_SPOIL", "")       # test hook "rank:factor": that rank sends wrong halos through face 0
        if spoil and int(spoil.split(":")[0]) == self.rank:
            engine.debug_peer_spoil(float(spoil.split(":")[1]))
        self.peer_ready = self.peer_mapped = True
        return True

    def use_peer(self, on: bool):
        """Switch between the two transports of an engine whose peer areas are set up (bench.py's transport check)."""
        self.peer_ready = bool(on) and bool(getattr(self, "peer_mapped", False))
        if not self.peer_ready:
            # every peer exchange pointed the grid kernels' halo input at the peer areas: back to the registered buffers
            faces = self.layout.faces(self.rank)
            for field, bufs in self._recv.items():
                self._engine.halo_set_recv(field, [(a, b) for _n, a, b in faces], bufs)
        return self.peer_ready

    def exchange(self, engine, field, f):
        """Send this rank's copy of the exchanged block planes of ``field`` (frame ``f``) to the neighbours and receive
        theirs into the registered buffers.  The grid kernels (or ``halo_apply``) add them."""
        if self.peer_ready:
            engine.halo_peer_exchange(field, f)          # one kernel on the engine's stream: push, publish, wait
            return
        self.exchange_finish(self.exchange_start(engine, field, f))

    def exchange_finish(self, reqs):
        """Make the engine's stream wait for an exchange begun with ``exchange_start``."""
        for req in reqs or ():
            req.wait()

    def exchange_start(self, engine, field, f):
        """Begin the exchange and return its requests without waiting: over RCCL the transfers run on the communicator's
        own stream, ordered behind what the engine's stream has enqueued so far, and the engine may go on enqueueing
        work that does not touch the exchanged planes (``Engine.grid_interior``) until ``exchange_finish``.  (gloo, the
        test transport, goes through host memory and is complete on return.)"""
        faces = self.layout.faces(self.rank)
        if not faces:
            return []
        if self.peer_ready:
            engine.halo_peer_exchange(field, f)
            return []
        if field not in self._recv:
            self.attach(engine, field, f)
        recv = self._recv[field]
        if self.stage_host:          # gloo: through host memory (tests; ranks sharing one GPU)
            ops, hosts = [], []
            for (nbr, a, b), rb in zip(faces, recv):
                hs = torch.stack([v for v in engine.halo_views(field, f, a, b)]).cpu()
                hr = torch.empty_like(hs)
                hosts.append((hr, rb, hs))
                ops.append(dist.P2POp(dist.isend, hs, nbr, self.group))
                ops.append(dist.P2POp(dist.irecv, hr, nbr, self.group))
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            for hr, rb, _hs in hosts:
                rb.copy_(hr)                                     # ordered on the engine's stream before the grid kernel
            return []
        key = (field, f if field == engine.HALO_GRID_IN else -1)
        ops = self._ops.get(key)
        if ops is None:
            ops = []
            for (nbr, a, b), rb in zip(faces, recv):
                for c, v in enumerate(engine.halo_views(field, f, a, b)):
                    ops.append(dist.P2POp(dist.isend, v, nbr, self.group))
                    ops.append(dist.P2POp(dist.irecv, rb[c], nbr, self.group))
            self._ops[key] = ops
        return dist.batch_isend_irecv(ops)

    # ---- rows (migration)
    def _wire(self, t):
        return t.cpu() if (self.stage_host and t is not None and t.is_cuda) else t

    def exchange_counts(self, n_down: int, n_up: int) -> Tuple[int, int]:
        """Tell the neighbours how many rows come their way; returns (rows arriving from below, from above)."""
        ops, got = [], {}
        for nbr, n, key in ((self.down, n_down, "d"), (self.up, n_up, "u")):
            if nbr is None:
                continue
            s = torch.tensor([n], dtype=torch.int64, device=self.scalar_device)
            r = torch.zeros(1, dtype=torch.int64, device=self.scalar_device)
            got[key] = (r, s)
            ops.append(dist.P2POp(dist.isend, s, nbr, self.group))
            ops.append(dist.P2POp(dist.irecv, r, nbr, self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return (int(got["d"][0].item()) if "d" in got else 0, int(got["u"][0].item()) if "u" in got else 0)

    def exchange_rows(self, send_down, send_up, n_recv_down: int, n_recv_up: int, width: int, device):
        """Send whole rows (float64 device tensors or None) down / up; receive ``n_recv_*`` rows of ``width`` doubles."""
        ops, keep, out = [], [], [None, None]
        for i, (nbr, snd, nrecv) in enumerate(((self.down, send_down, n_recv_down), (self.up, send_up, n_recv_up))):
            if nbr is None:
                continue
            if snd is not None and snd.numel() > 0:
                w = self._wire(snd.contiguous())
                keep.append(w)
                ops.append(dist.P2POp(dist.isend, w, nbr, self.group))
            if nrecv > 0:
                r = torch.empty(nrecv * width, dtype=torch.float64, device="cpu" if self.stage_host else device)
                out[i] = r
                ops.append(dist.P2POp(dist.irecv, r, nbr, self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return [None if r is None else r.to(device) for r in out]

    # ---- reductions
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


def _collective(exc):
    """Mark an exception that every rank raises together (the ranks agreed on it through a reduction first): whoever catches
    it may run collectives in its clean-up, which is NOT true of an exception one rank raises on its own."""
    exc.collective = True
    return exc


class SlabEngine:
    """Proxy around one rank's ``Engine`` that turns ``step`` / ``step_grad`` / ``loss_*`` into the phase-split,
    halo-exchanging, migrating versions.  ``MPMSimulator``, ``Loss`` and ``Tape`` work on it unchanged."""

    def __init__(self, engine, layout: SlabLayout, rank: int, group=None, comm: Optional[HaloComm] = None, migrate_every: int = 1,
                 overlap: bool = False, peer: Optional[bool] = None):
        self._e, self.layout, self.rank = engine, layout, rank
        # overlap: grid_op / grid_op.grad of the blocks outside the exchanged planes run while the halos are in flight
        # (one more launch per phase: worth it when a rank's kernels are long against the exchange -- configs 4 and 5 --
        # not when the host is what bounds the substep, as at 128^3 cut four ways)
        self.overlap = bool(overlap) and bool(layout.faces(rank))
        self.comm = comm if comm is not None else HaloComm(layout, rank, group, peer=peer)
        # device-side exchange (peer writes): the substep loops are then the native ones and the host only enqueues
        self.native_loops = bool(getattr(self.comm, "setup_peer", None) and self.comm.setup_peer(engine))
        self._overlap_asked = self.overlap
        if self.native_loops:
            self.overlap = False                   # the exchange is a kernel in the engine's own stream
        self.soft_contact = False
        self.migrate_every = int(migrate_every)        # env steps between two migrations (0: never -- fixed ownership)
        self._since_migration = 0
        self.migrations = 0                             # statistics: migrations done, rows sent away
        self.rows_moved = 0

    def __getattr__(self, name):                   # everything not overridden goes straight to the engine
        return getattr(self._e, name)

    def use_transport(self, kind: str) -> str:
        """"peer": device-side exchange + native substep loops (only if the peer areas were set up); "p2p": the
        torch.distributed point-to-point exchange driven from Python.  Returns the transport now in use.  Collective in
        effect: every rank must make the same choice before the next step."""
        on = self.comm.use_peer(kind == "peer") if hasattr(self.comm, "use_peer") else False
        self.native_loops = on
        self.overlap = False if on else self._overlap_asked
        return self.transport

    @property
    def transport(self) -> str:
        if self.native_loops:
            how = "exchange folded into the grid kernels" if self._e.peer_fused() else "device-side exchange kernel"
            return f"peer-write ({self._e.peer_memory_kind()} IPC-mapped receive areas, {how})"
        return {"nccl": "rccl-p2p (batch_isend_irecv)", "gloo": "gloo-p2p staged through host memory"}.get(self.comm.backend, self.comm.backend)

    def reset_exchange(self):
        """Collective, two phases with a barrier behind each: (0) every rank waits for the exchange kernels it has enqueued --
        behind the barrier nobody can still publish an old sequence number into a neighbour's arrival counter; (1) every rank
        clears its counters, sequence numbers and status word -- behind the second barrier all sides start again at sequence
        number 1.  Called at every episode reset / segment re-entry, so that an engine recovers from a timed-out or interrupted
        exchange instead of staying out of step (a single clear + barrier could be overwritten by a neighbour still draining)."""
        if not getattr(self.comm, "peer_mapped", False):
            return
        for phase in (0, 1):
            self._e.halo_peer_reset(phase)
            t = torch.zeros(1, dtype=torch.float64, device=self.comm.scalar_device)
            self.comm.all_reduce_(t)               # barrier on either backend

    # ---- state
    def set_frame(self, f, x=None, v=None, F=None, C_=None, resort=False):
        self._e.set_frame(f, x=x, v=v, F=F, C_=C_, resort=resort)
        if resort:
            self._since_migration = 0              # a new episode: ownership as assigned by the caller
            self.reset_exchange()

    def get_frame_by_id(self, f, want=("x", "v", "F", "C")):
        """(global ids ascending, rows in that order): the canonical view of a frame whatever its storage epoch."""
        fr, ids = self._e.get_frame(f, want=want), self._e.get_ids(f)
        o = np.argsort(ids, kind="stable")
        return ids[o], {k: (None if a is None else a[o]) for k, a in fr.items()}

    def _check(self):
        """Combine the device error word over the ranks before anyone raises: a rank that bailed out alone would
        leave the others waiting in their next exchange."""
        flags = torch.tensor([float(self._e.error_flags()), float(self._e.peer_status() if self.native_loops else 0)],
                             dtype=torch.float64, device=self.comm.scalar_device)
        self.comm.all_reduce_(flags, op=dist.ReduceOp.MAX)
        # (what is raised from here is raised on EVERY rank -- the flags were combined first: marked `collective`, so that a
        # caller's clean-up knows it may run collectives of its own, optimizer/checkpoint.py)
        if flags[1].item():
            st = int(flags[1].item())
            raise _collective(RuntimeError(f"halo exchange: an arrival timed out on some rank (status 0x{st:x}: field {st >> 16}, face {(st >> 8) & 255}) "
                                           "-- a neighbouring rank stopped enqueueing; results of this env step are not valid"))
        try:
            self._e.check_error(int(flags[0].item()))
        except Exception as exc:
            raise _collective(exc)

    # ---- segment checkpoints (optimizer/checkpoint.py): this rank's population at a frame, and re-entry with it
    def checkpoint(self, f):
        """Host copy of everything that defines this rank's rows at frame ``f``: global ids, state, materials, and how
        long ago the last migration was.  Rows are in the frame's own row order (``get_frame`` / ``get_ids``)."""
        e = self._e
        fr = e.get_frame(f)
        mu, lam, ys = e.get_materials(f)
        return dict(ids=e.get_ids(f).copy(), x=fr["x"], v=fr["v"], F=fr["F"], C=fr["C"], mu=mu, lam=lam, ys=ys,
                    since=self._since_migration)

    def reenter(self, ck, collective=True):
        """Make frame 0 the checkpointed population (a new episode for the engine: epoch 0 again, rows in the checkpoint's
        order, Hilbert-sorted storage); the next ``step`` migrates exactly as the run the checkpoint was taken from did.
        ``collective=False``: this rank alone (clean-up after a failure the other ranks do not know of) -- the exchange is NOT
        re-synchronised; the next ``set_state`` / ``reenter`` all ranks take part in does that."""
        e = self._e
        e.set_population(len(ck["ids"]))
        e.set_ids(ck["ids"])
        e.set_frame(0, x=ck["x"], v=ck["v"], F=ck["F"], C_=ck["C"], resort=True)
        e.set_materials(ck["mu"], ck["lam"], ck["ys"])
        self._since_migration = ck["since"]
        if collective:
            self.reset_exchange()

    def adjoint_to_reentry_rows(self, f=0):
        """After the reverse sweep has reached frame ``f`` = the frame a segment re-entered at: undo the migration the
        segment's first step began with, so that ``get_frame_grad(f)`` is in the checkpoint's rows (collective)."""
        if self.layout.world > 1 and self._e.frame_info(f)[2] > 0:
            self._migrate_adjoint(f)

    def _agree(self, err, where):
        """Raise on every rank if any rank has an exception to report (``err``), the failing rank its own."""
        bad = torch.tensor([0.0 if err is None else 1.0], dtype=torch.float64, device=self.comm.scalar_device)
        self.comm.all_reduce_(bad, op=dist.ReduceOp.MAX)
        if err is not None:
            raise _collective(err)
        if bad.item() > 0:
            raise _collective(RuntimeError(f"rank {self.rank}: another rank failed in {where}"))

    # ---- migration
    def migrate(self, f):
        """Hand the rows of frame ``f`` whose stencil centre left this slab to the neighbours, take theirs, re-sort."""
        e = self._e
        self._check()                              # nobody migrates a state that is already wrong
        # A host-side failure inside migrate_begin / migrate_finish ("N rows leave at once", "out of storage epochs",
        # "raise particle_capacity") happens on ONE rank; the others would wait for it forever in the row exchange or in
        # the next halo.  So every rank learns of it before anyone raises (_agree), and all raise together.
        err, nd, nu, rows_d, rows_u = None, 0, 0, None, None
        try:
            (nd, nu), (rows_d, rows_u) = e.migrate_begin(f)
        except Exception as ex:                    # noqa: BLE001
            err = ex
        self._agree(err, "migrate_begin")
        in_d, in_u = self.comm.exchange_counts(nd, nu)
        got_d, got_u = self.comm.exchange_rows(rows_d, rows_u, in_d, in_u, e.MIG_ROW, e.device)
        try:
            e.migrate_finish(f, got_d, got_u)
        except Exception as ex:                    # noqa: BLE001
            err = ex
        self._agree(err, "migrate_finish")
        self._since_migration = 0
        self.migrations += 1
        self.rows_moved += nd + nu

    def _migrate_adjoint(self, f):
        e = self._e
        (_sd, _su), (rd, ru), (rows_d, rows_u) = e.migrate_adjoint_begin(f)
        got_d, got_u = self.comm.exchange_rows(rows_d, rows_u, rd, ru, e.MIG_ADJ_ROW, e.device)
        e.migrate_adjoint_finish(f, got_d, got_u)

    # ---- hot path
    def step(self, first, n):
        e = self._e
        if self.layout.world > 1 and self.migrate_every > 0 and self._since_migration >= self.migrate_every:
            self.migrate(first)
        self._since_migration += 1
        if self.native_loops:
            e.slab_step(first, n)
            return
        e.fk(first, n)
        pending = False                         # g2p(f - 1) deferred: it runs fused with p2g(f), as on one GPU
        for f in range(first, first + n):
            e.p2g(f, chain=pending)
            if self.overlap:
                reqs = self.comm.exchange_start(e, e.HALO_GRID_IN, f)
                e.grid_interior(f)
                self.comm.exchange_finish(reqs)
            else:
                self.comm.exchange(e, e.HALO_GRID_IN, f)
            pending = f + 1 < first + n
            e.grid_g2p(f, chain=pending)

    def substep(self, f):
        self.step(f, 1)

    def step_grad(self, first, n, step):
        e = self._e
        last = first + n
        if self.layout.world > 1:
            adj_epoch = e.frame_info(last)[2]
            if adj_epoch >= 0 and adj_epoch != e.frame_info(last - 1)[1]:
                self._migrate_adjoint(last)     # particles migrated at `last`: adjoint rows go back where they came from
        if self.native_loops:
            e.slab_step_grad(first, n)
        else:
            for f in range(last - 1, first - 1, -1):
                e.grad_scatter(f)
                if self.overlap:
                    reqs = self.comm.exchange_start(e, e.HALO_GRID_OUT_ADJ, f)
                    e.grad_gather_interior(f)
                    self.comm.exchange_finish(reqs)
                else:
                    self.comm.exchange(e, e.HALO_GRID_OUT_ADJ, f)
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
        if self.layout.world > 1:
            self.comm.exchange(e, e.HALO_LOSS_MASS, f)
            e.halo_apply(e.HALO_LOSS_MASS, f)
        g = self.comm.reduce_loss_record(e.loss_partials(f, 0), self.soft_contact, 0)
        if self.soft_contact:
            e.loss_set_globals(g)
            g = self.comm.reduce_loss_record(e.loss_partials(f, 1), True, 1)
        return g

    def loss_forward(self, f):
        g = self._loss_globals(f)
        self._check()
        return self._e.loss_finish(g)

    def loss_backward(self, f):
        g = self._loss_globals(f)
        self._e.loss_set_globals(g)
        self._e.loss_backward_local(f)

    def grid_mass(self, f):
        raise NotImplementedError("grid_mass on a slab engine returns only this rank's partial grid")


def slab_window(x_all: np.ndarray, n_grid: int, layout: SlabLayout, rank: int, xy_margin: int, z_margin: Optional[int] = None):
    """Grid window (lo3, hi3) in nodes of one rank: the xy bounding box of the whole cloud's stencils plus
    ``xy_margin`` layers (the same on every rank, so that both sides of a face see the same planes), and in z the slab
    plus the exchanged block planes, clipped to where the body (plus ``z_margin``) can be."""
    z_margin = xy_margin if z_margin is None else z_margin
    b = (np.asarray(x_all) * n_grid - 0.5).astype(np.int64)
    lo = np.maximum(b.min(0) - [xy_margin, xy_margin, z_margin], 0)
    hi = np.minimum(b.max(0) + 3 + [xy_margin, xy_margin, z_margin], n_grid)
    z0, z1 = layout.slab(rank)
    if rank > 0:                                   # interior face below: the exchanged planes, nothing further down
        lo[2] = max(0, z0 - 4 * HALO_PLANES)
    if rank < layout.world - 1:
        hi[2] = min(n_grid, z1 + 4 * HALO_PLANES)
    return [int(v) for v in lo], [int(v) for v in hi]


def fused_grid_workgroups(fused: bool, world: int, ranks_per_gpu: int) -> int:
    """plmpm_config.grid_workgroups of a slab engine (0 = the library's default of 512).  Exchange kernels (the default): nothing
    waits inside a grid kernel, no cap.  Fused exchange + grid kernels: every grid workgroup of every rank on a GPU must be resident
    at once and 512 fit (k_grid_op_grad: two 256-thread workgroups per CU x 256 CUs) -- one rank per GPU runs 256 of them (no margin
    at all would mean that one CU reserved or masked by anybody turns the wait into a timeout), ranks that share a GPU (tests,
    emulations) half of their share of 512, as a power of two, at least 8."""
    if not fused or world <= 1:
        return 0
    if ranks_per_gpu <= 1:
        return 256
    cap = 8
    while cap * 2 <= 256 // ranks_per_gpu:
        cap *= 2
    return cap


def ranks_sharing_my_gpu(group=None) -> int:
    """How many processes of the torch.distributed world run on the GPU this process has selected -- from an all_gather of
    (host name, device identity), not from a guess: `procs / device_count` says 1 when several ranks are bound to one device
    while more devices are visible.  Collective over ``group`` when a process group exists; 1 otherwise (a
    one-process emulation of a rank has the GPU to itself)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    mine = gpu_identity()
    everyone = [None] * dist.get_world_size(group)
    dist.all_gather_object(everyone, mine, group=group)
    return max(1, sum(1 for other in everyone if other == mine))


def gpu_identity():
    """(host, device) of this process's current GPU; the device by its UUID where the runtime reports one (two processes
    may see the same physical GPU under different indices through HIP_VISIBLE_DEVICES), else by index."""
    import socket
    if not torch.cuda.is_available():
        return (socket.gethostname(), "cpu")
    idx = torch.cuda.current_device()
    try:
        dev = str(torch.cuda.get_device_properties(idx).uuid)
    except Exception:
        dev = f"index{idx}"
    return (socket.gethostname(), dev)


def make_slab_env(cfg, rank: int, world: int, *, halo: Optional[int] = None, compute_dtype=None, device=None, group=None,
                  target_fn: Optional[Callable] = None, particles: Optional[np.ndarray] = None,
                  layout: Optional[SlabLayout] = None, comm: Optional[HaloComm] = None,
                  xy_margin: Optional[int] = 12, migrate_every: int = 1, capacity_factor: float = 1.5,
                  yield_stress: Optional[np.ndarray] = None, overlap: bool = False, peer: Optional[bool] = None):
    """Build this rank's ``TaichiEnv`` over its slab of the scene in ``cfg`` (every rank samples the same seed-0
    particle cloud and keeps its own part).  ``target_fn(all_particles, sim) -> (n,n,n) grid`` may supply the loss
    target.  ``xy_margin`` (node layers): the grid window is the bounding box of the whole cloud at reset plus that
    margin -- a body that moves further raises, like one that drifts out of its slab between two migrations; ``None``
    allocates whole planes.  ``migrate_every``: env steps between two migrations.  ``yield_stress``: per-particle
    values for the WHOLE cloud (each rank keeps its part).  ``layout`` / ``comm`` override the balanced cut and the
    torch.distributed communicator (measurement tools: profiles/tools/slab_host_cost.py).  ``overlap``: run grid_op /
    grid_op.grad of the blocks outside the exchanged planes while the halos are in flight (``SlabEngine.overlap``).
    ``peer``: device-side halo exchange by peer writes + native substep loops (opt-in; None / False: off; the environment
    variable PLB_PEER_HALOS=0/1 overrides).  Returns (env, layout, owned_index)."""
    from .engine import taichi_env as te
    from .engine.losses import Loss
    from .engine.mpm_simulator import MPMSimulator
    from .engine.primitives import Primitives
    from .engine.shapes import Shapes

    if particles is not None:                       # caller-chosen cloud (e.g. a subsample), same on every rank
        x_all = np.ascontiguousarray(particles, np.float64)
        colors = np.zeros(len(x_all), np.int32)
    else:
        x_all, colors = Shapes(cfg.SHAPES).get()
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
    # PLMPM_PEER_FUSED=1 (opt-in): the fused exchange + grid kernels of the device-side exchange wait inside the launch for the
    # neighbours, so the grid workgroups of ALL ranks on a GPU must be resident at once -- 512 of them fit (k_grid_op_grad: two
    # 256-thread workgroups per CU); ranks that share a GPU (tests, emulations) take half of a rank's share, as a power of two.
    # (processes of a torch.distributed world only: a one-process emulation of a rank -- profiles/tools/slab_host_cost.py -- has
    # the GPU to itself)
    # The cap is part of the ENGINE (plmpm_config.grid_workgroups): the library takes the fused form only on engines created
    # with one (plmpm_peer_fused), so setting the variable after the build -- or building an Engine directly -- falls back to the
    # exchange kernels instead of timing out.  An explicit cfg.SIMULATOR.grid_workgroups is kept when it is the smaller one.
    per_gpu = ranks_sharing_my_gpu(group) if comm is None else 1
    cap = fused_grid_workgroups(os.environ.get("PLMPM_PEER_FUSED", "0") not in ("", "0"), world, per_gpu)
    if cap:
        asked = int(cfg.SIMULATOR.get("grid_workgroups", 0) or 0)
        cfg.SIMULATOR["grid_workgroups"] = min(asked, cap) if asked > 0 else cap
    env.n_particles = len(mine)
    z0, z1 = layout.slab(rank)
    window = None
    if xy_margin is not None:
        window = slab_window(x_all, n_grid, layout, rank, int(xy_margin))
    capacity = None
    if world > 1:
        capacity = int(len(mine) * capacity_factor) + 4096
    sim = MPMSimulator(cfg.SIMULATOR, env.primitives, compute_dtype=compute_dtype, device=device,
                       slab=(z0, z1), slab_halo=layout.halo if world > 1 else 0, grid_window=window,
                       particle_capacity=capacity)
    if world > 1:
        sim.engine.set_ids(mine)
        sim.engine = SlabEngine(sim.engine, layout, rank, group, comm, migrate_every=migrate_every, overlap=overlap, peer=peer)
        env.primitives._bind(sim.engine)
    if yield_stress is not None:
        sim._yield_stress = np.ascontiguousarray(np.asarray(yield_stress, np.float64)[mine])
    env.simulator = sim
    env.renderer = None
    env.loss = Loss(cfg.ENV.loss, sim)
    env._is_copy, env._tape = True, None
    env.initialize()
    if target_fn is not None:
        env.loss.load_target_density(grids=target_fn(x_all, sim))
    return env, layout, mine
