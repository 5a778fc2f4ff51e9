"""Thin object wrapper over the C ABI: one ``Engine`` = one ``plmpm_handle``.

Device memory comes from torch (ROCm caching allocator) and is bound into the
engine; kernels run on torch's current HIP stream so that they order with
``torch.distributed`` collectives.  All numerics live in libplmpm.so.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np
import torch

from .. import _lib as L


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


class Engine:
    def __init__(self, *, n_grid: int, n_particles: int, max_frames: int, substeps: int, dt: float, p_vol: float,
                 p_mass: float, gravity: Sequence[float], ground_friction: float, primitives: Sequence[dict] = (),
                 dtype: str = "float32", svd_grad_clamp: float = 1e-6, device: Optional[torch.device] = None,
                 slab: Optional[Sequence[int]] = None, store_grid="auto", slab_halo: int = 0, resort_steps: int = 2,
                 grid_window: Optional[Sequence[Sequence[int]]] = None, particle_capacity: Optional[int] = None,
                 allocate: bool = True, deterministic: bool = False, contact_min_adjoint: str = "add", minmax_tie: str = "second",
                 grid_workgroups: int = 0):
        self.lib = self._load_library()
        self.device = self._open_device(device)
        cfg = L.Config()
        cfg.dtype = L.F64 if dtype in ("float64", "f64") else L.F32
        cfg.n_grid, cfg.n_particles, cfg.max_frames, cfg.substeps = n_grid, n_particles, max_frames, substeps
        cfg.n_primitives = len(primitives)
        cfg.dt, cfg.p_vol, cfg.p_mass = dt, p_vol, p_mass
        cfg.gravity = (C.c_double * 3)(*[float(g) for g in gravity])
        cfg.ground_friction, cfg.svd_grad_clamp = float(ground_friction), float(svd_grad_clamp)
        cfg.slab_z0, cfg.slab_z1 = (0, n_grid) if slab is None else (int(slab[0]), int(slab[1]))
        # grid window (lo3, hi3) in nodes: only that box of the grid is allocated, stored per frame and swept
        win_nodes = n_grid ** 3
        if grid_window is not None:
            lo, hi = [int(v) for v in grid_window[0]], [int(v) for v in grid_window[1]]
            cfg.grid_lo, cfg.grid_hi = (C.c_int32 * 3)(*lo), (C.c_int32 * 3)(*hi)
            win_nodes = int(np.prod([(min(n_grid, (h + 3) // 4 * 4) - max(0, l) // 4 * 4) for l, h in zip(lo, hi)]))
        cap = int(particle_capacity) if particle_capacity else n_particles
        cfg.particle_capacity = cap
        if store_grid == "auto":
            # grid_m/grid_v_in + grid_v_out of every frame (8 scalars per node) stay resident -- no forward recompute in
            # substep_grad, fused g2p+p2g forward: 1.45x the substep rate -- while they take at most 40 % of the HBM
            # and, with the particle frames, at most 70 % (288 GB on an MI355X: 128^3 rollouts of up to ~1700 frames)
            tsz = 8 if cfg.dtype == L.F64 else 4
            grid_b = max_frames * 8 * tsz * win_nodes
            state_b = (max_frames + 1) * (cap + 255) // 256 * 256 * (24 + 21 * tsz)
            hbm = self._device_memory_bytes()
            store_grid = grid_b <= 0.40 * hbm and grid_b + state_b <= 0.70 * hbm
        cfg.store_grid = int(bool(store_grid))
        cfg.slab_halo = int(slab_halo)
        # re-sort the particles every so many env steps (single GPU); PLMPM_RESORT_STEPS overrides for experiments
        cfg.resort_steps = int(os.environ.get("PLMPM_RESORT_STEPS", resort_steps))
        self.store_grid = bool(store_grid)
        # bit-reproducible runs: integer-limb accumulation instead of floating-point atomics (include/plmpm.h)
        cfg.deterministic = int(bool(deterministic))
        # unverified Taichi autodiff semantics as switches (include/plmpm.h)
        cfg.contact_min_adjoint = {"add": 0, "argmin": 1}[contact_min_adjoint]
        cfg.minmax_tie = {"second": 0, "first": 1}[minmax_tie]
        self.grid_workgroups = int(grid_workgroups)         # as asked for (0 = the library's default)
        cfg.grid_workgroups = int(grid_workgroups)          # 0: 512; ranks that share a GPU pass 512 / ranks-per-GPU or less (include/plmpm.h)
        self.deterministic = bool(deterministic)
        parr = (L.Primitive * max(len(primitives), 1))()
        self.action_dims = []
        for i, p in enumerate(primitives):
            parr[i].shape = L.SHAPES[p["shape"]]
            parr[i].kinematics = L.KINEMATICS.get(p["shape"], 0)
            parr[i].action_dim = int(p.get("action_dim", 0))
            pr = list(p.get("params", ())) + [0.0, 0.0, 0.0]
            parr[i].params = (C.c_double * 3)(*pr[:3])
            parr[i].friction = float(p.get("friction", 0.9))
            sc = list(p.get("action_scale", ())) + [0.0] * L.MAX_ACTION_DIM
            parr[i].action_scale = (C.c_double * L.MAX_ACTION_DIM)(*sc[:L.MAX_ACTION_DIM])
            parr[i].lower_bound = (C.c_double * 3)(*p.get("lower_bound", (0.0, 0.0, 0.0)))
            parr[i].upper_bound = (C.c_double * 3)(*p.get("upper_bound", (1.0, 1.0, 1.0)))
            self.action_dims.append(parr[i].action_dim)
        self._view_cache = {}
        self.cfg, self.n_primitives = cfg, len(primitives)
        self.n_grid, self.n_particles, self.max_frames = n_grid, n_particles, max_frames
        self.dtype = "float64" if cfg.dtype == L.F64 else "float32"
        self.h = C.c_void_p()
        with self._device_guard():
            self._check(self.lib.plmpm_create(C.byref(cfg), parr, C.byref(self.h)))
            ws = L.Workspace()
            self._check(self.lib.plmpm_workspace_bytes(self.h, C.byref(ws)))
            self.workspace_bytes = {k: getattr(ws, k) for k, _ in L.Workspace._fields_}
            self._bufs = []
            if not allocate:                  # sizing only (how much HBM would this engine need?): nothing is bound
                return
            # torch owns the memory; keep the tensors alive as long as the handle
            self._bufs = [self._allocate(max(n, 256)) for n in (ws.state_bytes, ws.adjoint_bytes, ws.grid_bytes, ws.misc_bytes)]
            self.use_current_stream()
            self._check(self.lib.plmpm_bind_workspace(self.h, *[C.c_void_p(b.data_ptr()) for b in self._bufs]))

    def close(self):
        if getattr(self, "h", None):
            self.lib.plmpm_destroy(self.h)
            self.h = None
            self._bufs = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- plumbing: where the library, the device, its memory and its stream come from (one place each)
    def _load_library(self):
        return L.load()

    def _open_device(self, device):
        if not torch.cuda.is_available():
            raise L.EngineError("no ROCm device visible: the MPM engine has no CPU path")
        return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)

    def _device_memory_bytes(self):
        return torch.cuda.get_device_properties(self.device).total_memory

    def _device_guard(self):
        return torch.cuda.device(self.device)

    def _allocate(self, nbytes):
        return torch.empty(nbytes, dtype=torch.uint8, device=self.device)

    def _stream_handle(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _check(self, rc):
        if rc != 0:
            raise L.EngineError(self.lib.plmpm_last_error().decode())

    def use_current_stream(self):
        self._check(self.lib.plmpm_set_stream(self.h, C.c_void_p(self._stream_handle())))

    # ---- state
    def set_materials(self, mu, lam, ys):
        N = self.n_particles
        a = [np.ascontiguousarray(np.broadcast_to(np.asarray(v, np.float64), (N,))) for v in (mu, lam, ys)]
        self._check(self.lib.plmpm_set_materials(self.h, *[_ptr(v) for v in a]))

    def set_frame(self, f, x=None, v=None, F=None, C_=None, resort=False):
        # rows: the engine's own count at an episode reset (resort: a new epoch 0), else those of the frame's storage
        # epoch (a slab rank gains and loses rows by migration)
        N = self.n_particles if resort else self.frame_info(f)[0]
        x, v, F, C_ = _f64(x, (N, 3)), _f64(v, (N, 3)), _f64(F, (N, 3, 3)), _f64(C_, (N, 3, 3))
        self._check(self.lib.plmpm_set_frame(self.h, f, _ptr(x), _ptr(v), _ptr(F), _ptr(C_), int(resort)))

    def frame_info(self, f):
        """(rows, storage epoch, epoch of the resident adjoint or -1) of frame f."""
        n, e, a = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._check(self.lib.plmpm_frame_info(self.h, f, C.byref(n), C.byref(e), C.byref(a)))
        return n.value, e.value, a.value

    def get_frame(self, f, want=("x", "v", "F", "C")):
        N = self.frame_info(f)[0]
        out = {"x": np.empty((N, 3)) if "x" in want else None, "v": np.empty((N, 3)) if "v" in want else None,
               "F": np.empty((N, 3, 3)) if "F" in want else None, "C": np.empty((N, 3, 3)) if "C" in want else None}
        self._check(self.lib.plmpm_get_frame(self.h, f, _ptr(out["x"]), _ptr(out["v"]), _ptr(out["F"]), _ptr(out["C"])))
        return out

    def copy_frame(self, src, dst):
        self._check(self.lib.plmpm_copy_frame(self.h, src, dst))

    def set_primitive_state(self, prim, f, state8):
        """position(3) + rotation(4) + gap(1, Chopsticks; ignored by the other shapes)."""
        st = np.asarray(state8, np.float64).reshape(-1)
        s = np.zeros(8)
        s[:len(st)] = st                            # 7-long states (no gap) are accepted
        self._check(self.lib.plmpm_set_primitive_state(self.h, prim, f, _ptr(s)))

    def get_primitive_state(self, prim, f):
        s = np.empty(8)
        self._check(self.lib.plmpm_get_primitive_state(self.h, prim, f, _ptr(s)))
        return s

    def get_primitive_grad(self, prim, f):
        s = np.empty(8)
        self._check(self.lib.plmpm_get_primitive_grad(self.h, prim, f, _ptr(s)))
        return s

    def set_resort(self, on):
        """Switch the per-env-step re-sort off / on (optimizer/checkpoint.py keeps one order across segments).  Returns
        the previous setting, so that a caller can put it back."""
        prev = getattr(self, "_resort_on", True)
        self._check(self.lib.plmpm_set_resort(self.h, int(bool(on))))
        self._resort_on = bool(on)
        return prev

    def add_primitive_grad(self, prim, f, grad8):
        g = np.zeros(8)
        a = np.asarray(grad8, np.float64).reshape(-1)
        g[:len(a)] = a
        self._check(self.lib.plmpm_add_primitive_grad(self.h, prim, f, _ptr(g)))

    def set_softness(self, softness):
        self._check(self.lib.plmpm_set_softness(self.h, float(softness)))

    # ---- actions
    def set_action(self, step, n_substeps, action):
        a = _f64(action).reshape(-1) if action is not None else np.zeros(0)
        if a.size != sum(self.action_dims):
            raise ValueError(f"action has {a.size} entries, expected {sum(self.action_dims)}")
        self._check(self.lib.plmpm_set_action(self.h, step, n_substeps, _ptr(a) if a.size else None))

    def get_action_grad(self, n_steps):
        out = np.zeros((n_steps, sum(self.action_dims)))
        self._check(self.lib.plmpm_get_action_grad(self.h, n_steps, _ptr(out)))
        return out

    # ---- hot path
    def substep(self, f):
        self._check(self.lib.plmpm_substep(self.h, f))

    def step(self, first, n):
        self._check(self.lib.plmpm_step(self.h, first, n))

    def grad_begin(self, last_frame):
        self._check(self.lib.plmpm_grad_begin(self.h, last_frame))

    def substep_grad(self, f):
        self._check(self.lib.plmpm_substep_grad(self.h, f))

    def step_grad(self, first, n, step):
        self._check(self.lib.plmpm_step_grad(self.h, first, n, step))

    def segment_carry(self, from_frame, to_frame):
        self._check(self.lib.plmpm_segment_carry(self.h, from_frame, to_frame))

    def add_frame_grad(self, f, xa=None, va=None, Fa=None, Ca=None):
        N = self.frame_info(f)[0]
        xa, va, Fa, Ca = _f64(xa, (N, 3)), _f64(va, (N, 3)), _f64(Fa, (N, 3, 3)), _f64(Ca, (N, 3, 3))
        self._check(self.lib.plmpm_add_frame_grad(self.h, f, _ptr(xa), _ptr(va), _ptr(Fa), _ptr(Ca)))

    def get_frame_grad(self, f):
        n = C.c_int32(0)                                # rows of the epoch the adjoint is in (a slab frame that migrated: the epoch before)
        self._check(self.lib.plmpm_adjoint_rows(self.h, f, C.byref(n)))
        N = n.value
        out = {"x": np.empty((N, 3)), "v": np.empty((N, 3)), "F": np.empty((N, 3, 3)), "C": np.empty((N, 3, 3))}
        self._check(self.lib.plmpm_get_frame_grad(self.h, f, _ptr(out["x"]), _ptr(out["v"]), _ptr(out["F"]), _ptr(out["C"])))
        return out

    # ---- loss
    def loss_set_target(self, density):
        n = self.n_grid
        d = _f64(density, (n, n, n))
        self._check(self.lib.plmpm_loss_set_target(self.h, _ptr(d)))

    def loss_set_weights(self, sdf, density, contact, soft_contact):
        self._check(self.lib.plmpm_loss_set_weights(self.h, float(sdf), float(density), float(contact), int(bool(soft_contact))))

    def loss_forward(self, f):
        out = np.zeros(6)
        self._check(self.lib.plmpm_loss_forward(self.h, f, _ptr(out)))
        return dict(loss=out[0], sdf_loss=out[1], density_loss=out[2], contact_loss=out[3], iou=out[4])

    def loss_backward(self, f):
        self._check(self.lib.plmpm_loss_backward(self.h, f))

    def grid_mass(self, f):
        n = self.n_grid
        out = np.empty((n, n, n))
        self._check(self.lib.plmpm_get_grid_mass(self.h, f, _ptr(out)))
        return out

    def target_sdf(self):
        n = self.n_grid
        out = np.empty((n, n, n))
        self._check(self.lib.plmpm_loss_get_target_sdf(self.h, _ptr(out)))
        return out

    def grid_stats(self, f):
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.plmpm_grid_stats(self.h, f, C.byref(a), C.byref(b)))
        return a.value, b.value

    def tile_boxes(self, f):
        """(n_workgroups, 6) int32: stencil box origin + extent of each 256-particle workgroup of a scattered frame."""
        n = C.c_int(0)
        self._check(self.lib.plmpm_tile_boxes(self.h, f, None, 0, C.byref(n)))
        out = np.zeros((n.value, 6), np.int32)
        self._check(self.lib.plmpm_tile_boxes(self.h, f, _ptr(out), n.value, C.byref(n)))
        return out

    def order(self):
        p = np.empty(self.n_particles, np.int32)
        self._check(self.lib.plmpm_get_order(self.h, _ptr(p)))
        return p

    # ---- multi-GPU building blocks (plasticinelab_amd.distributed drives these)
    HALO_GRID_IN, HALO_GRID_OUT_ADJ, HALO_LOSS_MASS = 0, 1, 2

    def fk(self, first, n):
        self._check(self.lib.plmpm_fk(self.h, first, n))

    def p2g(self, f, chain=False):
        """p2g(f); ``chain``: fused with the g2p(f - 1) that ``grid_g2p(f - 1, chain=True)`` left pending."""
        self._check(self.lib.plmpm_p2g(self.h, f, int(chain)))

    def grid_g2p(self, f, chain=False):
        """grid_op(f) (adds the registered halo planes) + g2p(f); ``chain`` leaves the g2p to the next ``p2g``."""
        self._check(self.lib.plmpm_grid_g2p(self.h, f, int(chain)))

    def grid_interior(self, f):
        """grid_op(f) on the blocks outside the exchanged planes only (they need nothing from the neighbours): runs
        while the halos are in flight; the ``grid_g2p(f)`` that follows then does the exchanged planes and g2p."""
        self._check(self.lib.plmpm_grid_interior(self.h, f))

    def grad_gather_interior(self, f):
        """The same split of grid_op.grad: interior blocks now, the exchanged planes + p2g.grad in ``grad_gather(f)``."""
        self._check(self.lib.plmpm_grad_gather_interior(self.h, f))

    def grad_scatter(self, f):
        self._check(self.lib.plmpm_grad_scatter(self.h, f))

    def grad_gather(self, f):
        self._check(self.lib.plmpm_grad_gather(self.h, f))

    def chain_grad(self, first, n, step):
        self._check(self.lib.plmpm_chain_grad(self.h, first, n, step))

    @property
    def torch_dtype(self):
        return torch.float64 if self.dtype == "float64" else torch.float32

    def grid_window(self):
        """(origin node (3,), extent in 4^3 blocks (3,)) of the allocated grid window."""
        o, b = np.zeros(3, np.int32), np.zeros(3, np.int32)
        self._check(self.lib.plmpm_grid_window(self.h, _ptr(o), _ptr(b)))
        return o, b

    def halo_ncomp(self, field):
        return {self.HALO_GRID_IN: 4, self.HALO_GRID_OUT_ADJ: 3, self.HALO_LOSS_MASS: 1}[field]

    def halo_views(self, field, f, bz_a, bz_b):
        """One torch view per component of block planes [bz_a, bz_b) of a halo field -- the grid memory itself (zero
        copy): what a rank sends to the neighbour on that face.  Cached per (field, frame, planes)."""
        key = (field, f if field == self.HALO_GRID_IN else -1, bz_a, bz_b)
        v = self._view_cache.get(key)
        if v is None:
            v = []
            for c in range(self.halo_ncomp(field)):
                p, cnt = C.c_void_p(), C.c_size_t()
                self._check(self.lib.plmpm_halo_region(self.h, field, f, c, bz_a, bz_b, C.byref(p), C.byref(cnt)))
                v.append(self._view(p.value, cnt.value, self.torch_dtype))
            self._view_cache[key] = v
        return v

    def halo_set_recv(self, field, planes, bufs):
        """Register where the neighbours' copies of block planes ``planes = [(bz_a, bz_b), ...]`` arrive
        (``bufs[i]``: contiguous device tensor [ncomp, count]); grid_op / grid_op.grad add them on first touch."""
        nf = len(planes)
        za = (C.c_int * max(nf, 1))(*[p[0] for p in planes])
        zb = (C.c_int * max(nf, 1))(*[p[1] for p in planes])
        ptr = (C.c_void_p * max(nf, 1))(*[b.data_ptr() for b in bufs])
        self._check(self.lib.plmpm_halo_set_recv(self.h, field, nf, za, zb, ptr))
        self._recv_keepalive = getattr(self, "_recv_keepalive", {})
        self._recv_keepalive[field] = list(bufs)

    def halo_apply(self, field, f):
        self._check(self.lib.plmpm_halo_apply(self.h, field, f))

    # ---- device-side halo exchange (plmpm_peer.hip)
    def peer_alloc(self, field, bz_a, bz_b):
        """A receive area for block planes [bz_a, bz_b) of ``field`` in fine-grained device memory owned by the engine:
        (device pointer, 64-byte IPC handle a neighbouring process maps with ``peer_open``)."""
        nbytes, p = C.c_size_t(), C.c_void_p()
        self._check(self.lib.plmpm_peer_area_bytes(self.h, field, bz_a, bz_b, C.byref(nbytes)))
        handle = C.create_string_buffer(64)
        self._check(self.lib.plmpm_peer_alloc(self.h, nbytes, C.byref(p), handle))
        return p.value, handle.raw

    def peer_open(self, handle: bytes):
        p = C.c_void_p()
        self._check(self.lib.plmpm_peer_open(self.h, C.create_string_buffer(handle, 64), C.byref(p)))
        return p.value

    def halo_peer_setup(self, field, planes, local, remote):
        nf = len(planes)
        za = (C.c_int * max(nf, 1))(*[p[0] for p in planes])
        zb = (C.c_int * max(nf, 1))(*[p[1] for p in planes])
        lo = (C.c_void_p * max(nf, 1))(*local)
        re = (C.c_void_p * max(nf, 1))(*remote)
        self._check(self.lib.plmpm_halo_peer_setup(self.h, field, nf, za, zb, lo, re))

    def halo_peer_exchange(self, field, f):
        self._check(self.lib.plmpm_halo_peer_exchange(self.h, field, f))

    def peer_status(self):
        st = C.c_int()
        self._check(self.lib.plmpm_peer_status(self.h, C.byref(st)))
        return st.value

    def halo_peer_reset(self, phase):
        """Collective re-synchronisation of the device-side exchange in two phases, a host barrier over the ranks behind each:
        0 = drain (this rank's enqueued exchange kernels are finished), 1 = clear (counters, sequence numbers, status word)."""
        self._check(self.lib.plmpm_halo_peer_reset(self.h, int(phase)))

    def peer_fused(self):
        """True: the native slab loops fold each exchange into the grid kernel that consumes it (plmpm_peer_fused)."""
        k = C.c_int()
        self._check(self.lib.plmpm_peer_fused(self.h, C.byref(k)))
        return bool(k.value)

    def peer_memory_kind(self):
        k = C.c_int()
        self._check(self.lib.plmpm_peer_memory_kind(self.h, C.byref(k)))
        return "uncached" if k.value else "fine-grained"

    def peer_ping(self, field, token, timeout_s=5.0):
        """Collective first contact with the neighbours of ``field`` (plmpm_peer_ping): [(arrived, wait in us)] per face."""
        ok, us = (C.c_int * 2)(), (C.c_double * 2)()
        self._check(self.lib.plmpm_peer_ping(self.h, int(field), C.c_uint(int(token)), C.c_double(timeout_s), ok, us))
        return [(bool(ok[i]), float(us[i])) for i in range(2)]

    def debug_peer_spoil(self, factor):
        self._check(self.lib.plmpm_debug_peer_spoil(self.h, C.c_double(factor)))

    def debug_contact(self, seed=-1):
        """Entries in the list of blocks whose pose adjoints are still due; ``seed`` >= 0 first plants that many stale ones (test hook)."""
        k = C.c_int()
        self._check(self.lib.plmpm_debug_contact(self.h, int(seed), C.byref(k)))
        return k.value

    def slab_step(self, first, n):
        """fk + the forward substeps of one env step of a slab rank, exchanges included: enqueue only."""
        self._check(self.lib.plmpm_slab_step(self.h, first, n))

    def slab_step_grad(self, first, n):
        self._check(self.lib.plmpm_slab_step_grad(self.h, first, n))

    # ---- migration (slab engines)
    def set_ids(self, ids):
        ids = np.ascontiguousarray(ids, np.int32)
        assert len(ids) == self.n_particles
        self._check(self.lib.plmpm_set_ids(self.h, _ptr(ids)))

    def set_population(self, n_rows):
        """Slab engines: re-enter with a new set of ``n_rows`` rows (then ``set_ids``, ``set_frame(0, ..., resort=True)``,
        ``set_materials``): a segment checkpoint of a population that migration has changed."""
        self._check(self.lib.plmpm_set_population(self.h, int(n_rows)))
        self.n_particles = int(n_rows)

    def get_materials(self, f):
        """(mu, lam, yield_stress) of the rows of frame f, in ``get_frame``'s row order."""
        n = self.frame_info(f)[0]
        out = [np.empty(n) for _ in range(3)]
        self._check(self.lib.plmpm_get_materials(self.h, f, *[_ptr(a) for a in out]))
        return out

    def get_ids(self, f):
        ids = np.empty(self.frame_info(f)[0], np.int32)
        self._check(self.lib.plmpm_get_ids(self.h, f, _ptr(ids)))
        return ids

    MIG_ROW, MIG_ADJ_ROW = 28, 24

    def migrate_begin(self, f):
        """-> ((n_down, n_up), (rows_down, rows_up)): float64 device views of the packed rows that leave frame f."""
        cnt = np.zeros(2, np.int32)
        pd, pu = C.c_void_p(), C.c_void_p()
        self._check(self.lib.plmpm_migrate_begin(self.h, f, _ptr(cnt), C.byref(pd), C.byref(pu)))
        rows = [self._view(p.value, int(n) * self.MIG_ROW, torch.float64) if n > 0 else None for p, n in ((pd, cnt[0]), (pu, cnt[1]))]
        return (int(cnt[0]), int(cnt[1])), rows

    def migrate_finish(self, f, rows_down, rows_up):
        """Merge the arrivals (float64 device tensors of whole rows, or None), re-sort, new storage epoch -> rows."""
        n = C.c_int32(0)
        nd = 0 if rows_down is None else rows_down.numel() // self.MIG_ROW
        nu = 0 if rows_up is None else rows_up.numel() // self.MIG_ROW
        self._check(self.lib.plmpm_migrate_finish(self.h, f, nd, C.c_void_p(rows_down.data_ptr() if nd else 0),
                                              nu, C.c_void_p(rows_up.data_ptr() if nu else 0), C.byref(n)))
        return n.value

    def migrate_adjoint_begin(self, f):
        """-> ((send_down, send_up), (recv_down, recv_up), (rows_down, rows_up)) for the reverse exchange at frame f."""
        snd, rcv = np.zeros(2, np.int32), np.zeros(2, np.int32)
        pd, pu = C.c_void_p(), C.c_void_p()
        self._check(self.lib.plmpm_migrate_adjoint_begin(self.h, f, _ptr(snd), _ptr(rcv), C.byref(pd), C.byref(pu)))
        rows = [self._view(p.value, int(n) * self.MIG_ADJ_ROW, torch.float64) if n > 0 else None for p, n in ((pd, snd[0]), (pu, snd[1]))]
        return (int(snd[0]), int(snd[1])), (int(rcv[0]), int(rcv[1])), rows

    def migrate_adjoint_finish(self, f, rows_down, rows_up):
        self._check(self.lib.plmpm_migrate_adjoint_finish(self.h, f, C.c_void_p(rows_down.data_ptr() if rows_down is not None else 0),
                                                      C.c_void_p(rows_up.data_ptr() if rows_up is not None else 0)))

    # ---- per-primitive queries (Primitive.sdf / set_velocity, Loss.min_dist / dist_norm)
    def primitive_sdf(self, prim, f, points):
        pts = _f64(points).reshape(-1, 3)
        out = np.empty(len(pts))
        self._check(self.lib.plmpm_primitive_sdf(self.h, prim, f, _ptr(pts), len(pts), _ptr(out)))
        return out

    def set_velocity(self, prim, step, n_substeps):
        self._check(self.lib.plmpm_set_velocity(self.h, prim, step, n_substeps))

    def loss_contact_scalars(self):
        md, dn = np.zeros(L.MAX_PRIMITIVES), np.zeros(L.MAX_PRIMITIVES)
        self._check(self.lib.plmpm_loss_contact_scalars(self.h, _ptr(md), _ptr(dn)))
        return md[:self.n_primitives], dn[:self.n_primitives]

    def measure_hbm(self, nbytes=1 << 30, reps=5):
        """(copy GB/s, read GB/s) of this device, measured with the library's own 16-byte-per-lane kernels."""
        a = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        b = torch.empty_like(a)
        cg, rg = C.c_double(0), C.c_double(0)
        stream = self._stream_handle()
        self._check(self.lib.plmpm_measure_hbm(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), nbytes, reps, C.c_void_p(stream),
                                           C.byref(cg), C.byref(rg)))
        return cg.value, rg.value

    def _view(self, ptr, count, dtype):
        """torch view of engine-owned device memory (inside one of the bound workspaces)."""
        esz = torch.empty(0, dtype=dtype).element_size()
        for b in self._bufs:
            off = ptr - b.data_ptr()
            if 0 <= off and off + count * esz <= b.numel():
                return b[off:off + count * esz].view(dtype)
        raise L.EngineError("pointer outside the bound workspaces")

    def pose_grad_views(self, first, n_frames):
        pa, pc, ra, rc, ga, gc = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
        self._check(self.lib.plmpm_pose_grad_region(self.h, first, n_frames, C.byref(pa), C.byref(pc), C.byref(ra), C.byref(rc),
                                                C.byref(ga), C.byref(gc)))
        return (self._view(pa.value, pc.value, torch.float64), self._view(ra.value, rc.value, torch.float64),
                self._view(ga.value, gc.value, torch.float64))

    def loss_scatter(self, f):
        self._check(self.lib.plmpm_loss_scatter(self.h, f))

    def loss_partials(self, f, phase):
        out = np.zeros(32)
        self._check(self.lib.plmpm_loss_partials(self.h, f, phase, _ptr(out)))
        return out

    def loss_set_globals(self, g32):
        g = _f64(g32, (32,))
        self._check(self.lib.plmpm_loss_set_globals(self.h, _ptr(g)))

    def loss_finish(self, g32):
        g, out = _f64(g32, (32,)), np.zeros(6)
        self._check(self.lib.plmpm_loss_finish(self.h, _ptr(g), _ptr(out)))
        return dict(loss=out[0], sdf_loss=out[1], density_loss=out[2], contact_loss=out[3], iou=out[4])

    def loss_backward_local(self, f):
        self._check(self.lib.plmpm_loss_backward_local(self.h, f))

    def error_flags(self) -> int:
        """Device error word (bit 0: a particle left this rank's z-slab + halo or the halo window); cleared by the read."""
        e = C.c_int(0)
        self._check(self.lib.plmpm_check_error(self.h, C.byref(e)))
        return int(e.value)

    def check_error(self, flags=None):
        flags = self.error_flags() if flags is None else flags
        if flags & 1:
            raise L.EngineError("a particle's stencil left the allocated grid window, or (slab engines) this rank's slab + halo "
                                "between two migrations: widen grid_window / slab_halo or migrate more often")

    def profile_enable(self, on=True):
        self._check(self.lib.plmpm_profile_enable(self.h, int(on)))

    def profile_read(self):
        """{kernel name: (total ms, launches)} measured with HIP events on the launch stream."""
        k = self.lib.plmpm_profile_kernel_count()
        ms, cnt = np.zeros(k), np.zeros(k, np.int64)
        self._check(self.lib.plmpm_profile_read(self.h, _ptr(ms), _ptr(cnt)))
        return {self.lib.plmpm_profile_kernel_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(k)}

    def synchronize(self):
        torch.cuda.current_stream(self.device).synchronize()
