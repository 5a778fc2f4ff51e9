"""CPU tier, world_size 2 and 3 over gloo: the multi-GPU orchestration -- slab layout, zero-copy symmetric halo sum
exchange of block planes, particle migration at env-step boundaries (rows forward, adjoint rows back), loss-record
reduction, pose-adjoint sum + step ordering -- with a linear toy engine standing in for the HIP engine: the
distributed logic itself has no GPU dependency.  The real kernels behind the same SlabEngine are checked N-rank vs
1-rank in tests/test_gpu_distributed.py (-m gpu)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from plasticinelab_amd.distributed import HaloComm, SlabEngine, SlabLayout, slab_window

N = 32          # toy grid: 8 block planes
SUB = 2         # substeps per env step
STEPS = 3


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class ToyEngine:
    """Implements the phase API of Engine on CPU tensors with linear toy physics.  Every particle (global id, stencil
    base z, weight) deposits on the 3 node planes of its stencil; ``grid_g2p`` reads the summed planes back and, at
    the end of an env step, moves the particle by its per-id drift (so ownership changes and rows must migrate); the
    reverse pass scatters the particle's carried adjoint the same way and feeds what it reads back into that adjoint,
    so an adjoint row that is not sent home after a migration shows up in the result.  Linear, so N-rank results must
    equal the 1-rank ones to round-off."""
    HALO_GRID_IN, HALO_GRID_OUT_ADJ, HALO_LOSS_MASS = 0, 1, 2
    MIG_ROW, MIG_ADJ_ROW = 28, 24
    device = torch.device("cpu")
    torch_dtype = torch.float64

    def __init__(self, ids, base_z, weight, drift, layout, rank):
        self.layout, self.rank = layout, rank
        self.z0, self.z1 = layout.slab(rank)
        self.drift = drift                                     # global id -> layers per env step
        self.frames = {0: dict(ids=np.array(ids), bz=np.array(base_z), w=np.array(weight, float))}
        self.epoch_of, self.mig = {0: 0}, {}
        self.next_epoch = 3
        self.gin, self.goa, self.lm = {}, torch.zeros(3, N, N, N, dtype=torch.float64), None
        self.recv = {}
        self.read, self.adj_read = {}, {}
        self.adj, self.adj_epoch, self.adj_frame = None, -1, -1
        self.pos_l = torch.zeros(64, dtype=torch.float64)
        self.rot_l = torch.zeros(64, dtype=torch.float64)
        self.gap_l = torch.zeros(64, dtype=torch.float64)
        self.calls, self.pending = [], None

    # ---- plumbing the SlabEngine uses
    def halo_ncomp(self, field):
        return {0: 4, 1: 3, 2: 1}[field]

    def _field(self, field, f):
        return {0: self.gin.get(f), 1: self.goa, 2: self.lm}[field]

    def halo_views(self, field, f, a, b):
        g = self._field(field, f)
        return [g[c, 4 * a:4 * b].reshape(-1) for c in range(g.shape[0])]      # contiguous: views, like the engine's

    def halo_set_recv(self, field, planes, bufs):
        self.recv[field] = list(zip(planes, bufs))

    def _add_recv(self, field, f):
        g = self._field(field, f)
        for (a, b), buf in self.recv.get(field, []):
            g[:, 4 * a:4 * b] += buf.view(g.shape[0], 4 * (b - a), N, N)

    def halo_apply(self, field, f):
        assert field == 2
        self._add_recv(2, f)

    # ---- device-side exchange (csrc/plmpm_peer.hip), emulated: a receive area is a file in /dev/shm that the neighbour
    # maps by name -- what fine-grained device memory behind an IPC handle is on the GPUs.  Same layout (arrival counter,
    # two halves written alternately) and same protocol: write the neighbour's half, publish the counter, wait for ours.
    HEADER = 32

    def peer_alloc(self, field, a, b):
        cnt = self.halo_ncomp(field) * 4 * (b - a) * N * N
        self._areas = getattr(self, "_areas", [])
        name = f"/dev/shm/plb_toy_{os.getpid()}_{len(self._areas)}"
        arr = np.memmap(name, dtype=np.float64, mode="w+", shape=(self.HEADER + 2 * cnt,))
        arr[:] = 0
        self._areas.append(arr)
        self._owned = getattr(self, "_owned", []) + [name]
        return len(self._areas) - 1, name.encode().ljust(64, b"\0")

    def peer_open(self, handle):
        assert len(handle) == 64
        self._areas.append(np.memmap(handle.rstrip(b"\0").decode(), dtype=np.float64, mode="r+"))
        return len(self._areas) - 1

    def halo_peer_setup(self, field, planes, local, remote):
        self.peer = getattr(self, "peer", {})
        self.peer[field] = dict(planes=list(planes), local=[self._areas[i] for i in local], remote=[self._areas[i] for i in remote], seq=0)

    def halo_peer_exchange(self, field, f):
        import time
        P = self.peer[field]
        P["seq"] += 1
        seq, half, g = P["seq"], P["seq"] & 1, self._field(field, f)
        for (a, b), rem in zip(P["planes"], P["remote"]):
            data = g[:, 4 * a:4 * b].reshape(-1).numpy()
            rem[self.HEADER + half * data.size:self.HEADER + (half + 1) * data.size] = data
            rem[0] = seq                                        # published behind the data
        recv = []
        for (a, b), loc in zip(P["planes"], P["local"]):
            t0 = time.time()
            while loc[0] < seq:
                assert time.time() - t0 < 120, "the neighbour's arrival never came"
                time.sleep(1e-4)
            cnt = g.shape[0] * 4 * (b - a) * N * N
            recv.append(((a, b), torch.from_numpy(np.array(loc[self.HEADER + half * cnt:self.HEADER + (half + 1) * cnt]))))
        self.recv[field] = recv
        self.calls.append(("peer_exchange", field, f))

    def peer_status(self):
        return 0

    def slab_step(self, first, n):                              # plmpm_slab_step
        self.fk(first, n)
        pending = False
        for f in range(first, first + n):
            self.p2g(f, chain=pending)
            self.halo_peer_exchange(0, f)
            pending = f + 1 < first + n
            self.grid_g2p(f, chain=pending)

    def slab_step_grad(self, first, n):                         # plmpm_slab_step_grad
        for f in range(first + n - 1, first - 1, -1):
            self.grad_scatter(f)
            self.halo_peer_exchange(1, f)
            self.grad_gather(f)

    def close(self):
        for name in getattr(self, "_owned", []):
            try:
                os.unlink(name)
            except OSError:
                pass

    def frame_info(self, f):
        fr = self.frames[f]
        return len(fr["ids"]), self.epoch_of[f], (self.adj_epoch if self.adj_frame == f else -1)

    def error_flags(self):
        fr = self.frames[max(self.frames)]
        h = self.layout.halo
        bad = ((fr["bz"] < self.z0 - h) & (self.rank > 0)) | ((fr["bz"] + 2 >= self.z1 + h) & (self.rank < self.layout.world - 1))
        return int(bad.any())

    def check_error(self, flags=None):
        assert not (self.error_flags() if flags is None else flags), "a toy particle left slab + halo"

    # ---- forward
    def fk(self, first, n):
        self.calls.append(("fk", first, n))

    def p2g(self, f, chain=False):
        if chain:
            assert self.pending == f - 1
            self._g2p(f - 1)
            self.pending = None
        else:
            assert self.pending is None
        fr = self.frames[f]
        g = torch.zeros(4, N, N, N, dtype=torch.float64)
        for b, w in zip(fr["bz"], fr["w"]):
            for k in range(3):
                g[:, b + k, 1, 2] += w * (k + 1) * (f + 1)
        self.gin[f] = g
        self.calls.append(("p2g", f, bool(chain)))

    def grid_g2p(self, f, chain=False):
        self._add_recv(0, f)                                   # grid_op: partial sums + the neighbours'
        if chain:
            self.pending = f
        else:
            self._g2p(f)
        self.calls.append(("grid_g2p", f, bool(chain)))

    def grid_interior(self, f):
        # the blocks outside the exchanged planes: nothing of the neighbours' may have been added yet
        assert ("grid_g2p", f, True) not in self.calls and ("grid_g2p", f, False) not in self.calls
        self.calls.append(("grid_interior", f))

    def grad_gather_interior(self, f):
        assert self.adj_frame == f + 1
        self.calls.append(("grad_gather_interior", f))

    def _g2p(self, f):
        fr = self.frames[f]
        self.read[f] = {int(i): float(self.gin[f][0, b:b + 3, 1, 2].sum()) for i, b in zip(fr["ids"], fr["bz"])}
        bz = fr["bz"].copy()
        if (f + 1) % SUB == 0:                                 # end of an env step: the particles move
            bz = bz + np.array([self.drift[int(i)] for i in fr["ids"]], dtype=bz.dtype) if len(bz) else bz
        self.frames[f + 1] = dict(ids=fr["ids"].copy(), bz=bz, w=fr["w"].copy())
        self.epoch_of[f + 1] = self.epoch_of[f]

    # ---- reverse
    def grad_begin(self, last):
        self.adj = np.ones(len(self.frames[last]["ids"]))
        self.adj_epoch, self.adj_frame = self.epoch_of[last], last

    def grad_scatter(self, f):
        assert self.adj_frame == f + 1 and self.adj_epoch == self.epoch_of[f], (f, self.adj_frame, self.adj_epoch, self.epoch_of[f])
        fr = self.frames[f]
        self.goa = torch.zeros(3, N, N, N, dtype=torch.float64)
        for b, w, a in zip(fr["bz"], fr["w"], self.adj):
            self.goa[:, b:b + 3, 1, 2] += w * a

    def grad_gather(self, f):
        self._add_recv(1, f)
        fr = self.frames[f]
        got = np.array([float(self.goa[0, b:b + 3, 1, 2].sum()) for b in fr["bz"]])
        self.adj_read[f] = {int(i): g for i, g in zip(fr["ids"], got)}
        self.adj = self.adj + 0.125 * got
        self.adj_frame = f
        self.pos_l[f] += float(self.goa[0, self.z0:self.z1].sum())           # owned planes only, like k_grid_op_grad

    def chain_grad(self, first, n, step):
        # like plmpm_chain_grad: the (rank-summed) local pose adjoints of the step's frames are folded into the global
        # ones and cleared, so the frame two env steps share is not summed over the ranks twice
        self.calls.append(("chain_grad", first, n, step, float(self.pos_l[first:first + n + 1].sum())))
        self.pos_l[first:first + n + 1] = 0.0

    def pose_grad_views(self, first, nf):
        return self.pos_l[first:first + nf], self.rot_l[first:first + nf], self.gap_l[first:first + nf]

    # ---- migration
    def migrate_begin(self, f):
        fr = self.frames[f]
        cz = fr["bz"] + 1
        dest = np.where(cz < self.z0, 0, np.where(cz >= self.z1, 1, -1))
        rows = []
        for d in (0, 1):
            idx = np.nonzero(dest == d)[0]
            r = torch.zeros(len(idx), self.MIG_ROW, dtype=torch.float64)
            r[:, 0] = torch.as_tensor(fr["ids"][idx], dtype=torch.float64)
            r[:, 1] = torch.as_tensor(fr["bz"][idx], dtype=torch.float64)
            r[:, 2] = torch.as_tensor(fr["w"][idx])
            rows.append(r.reshape(-1) if len(idx) else None)
        self._pend = (f, dest)
        return (int((dest == 0).sum()), int((dest == 1).sum())), rows

    def migrate_finish(self, f, rows_d, rows_u):
        pf, dest = self._pend
        assert pf == f
        fr = self.frames[f]
        stay = np.nonzero(dest < 0)[0]
        arr = [r.view(-1, self.MIG_ROW) for r in (rows_d, rows_u) if r is not None]
        arr = torch.cat(arr) if arr else torch.zeros(0, self.MIG_ROW, dtype=torch.float64)
        n_in = [0 if r is None else r.numel() // self.MIG_ROW for r in (rows_d, rows_u)]
        ids = np.concatenate([fr["ids"][stay], arr[:, 0].numpy().astype(fr["ids"].dtype)])
        bz = np.concatenate([fr["bz"][stay], arr[:, 1].numpy().astype(fr["bz"].dtype)])
        w = np.concatenate([fr["w"][stay], arr[:, 2].numpy()])
        src = np.concatenate([stay, -1 - np.arange(len(arr))])
        o = np.argsort(ids, kind="stable")                      # the toy's "Hilbert re-sort"
        e = self.next_epoch
        self.next_epoch += 1
        self.mig[e] = dict(parent=self.epoch_of[f], src=src[o], leave=[np.nonzero(dest == 0)[0], np.nonzero(dest == 1)[0]],
                           n_in=n_in, n_old=len(fr["ids"]))
        self.frames[f] = dict(ids=ids[o], bz=bz[o], w=w[o])
        self.epoch_of[f] = e
        return len(ids)

    def migrate_adjoint_begin(self, f):
        assert self.adj_frame == f and self.adj_epoch == self.epoch_of[f]
        m = self.mig[self.adj_epoch]
        back = [torch.zeros(n, self.MIG_ADJ_ROW, dtype=torch.float64) for n in m["n_in"]]
        self._tmp = np.zeros(m["n_old"])
        for i, s in enumerate(m["src"]):
            if s >= 0:
                self._tmp[s] = self.adj[i]
            else:
                a = -1 - s
                (back[0] if a < m["n_in"][0] else back[1])[a if a < m["n_in"][0] else a - m["n_in"][0], 0] = self.adj[i]
        rows = [b.reshape(-1) if len(b) else None for b in back]
        return tuple(m["n_in"]), (len(m["leave"][0]), len(m["leave"][1])), rows

    def migrate_adjoint_finish(self, f, rows_d, rows_u):
        m = self.mig[self.adj_epoch]
        for lst, rows in zip(m["leave"], (rows_d, rows_u)):
            if len(lst):
                self._tmp[lst] = rows.view(-1, self.MIG_ADJ_ROW)[:, 0].numpy()
        self.adj, self.adj_epoch = self._tmp, m["parent"]

    # ---- loss
    def loss_set_weights(self, *a):
        pass

    def loss_scatter(self, f):
        fr = self.frames[f]
        self.lm = torch.zeros(1, N, N, N, dtype=torch.float64)
        for b, w in zip(fr["bz"], fr["w"]):
            self.lm[0, b:b + 3, 1, 2] += w
        self._loss_frame = f

    def loss_partials(self, f, phase):
        rec = np.zeros(32)
        own = self.lm[0, self.z0:self.z1]
        rec[0], rec[1], rec[2], rec[3], rec[4] = own.abs().sum(), 2 * own.sum(), own.max(), 3 * own.sum(), own.sum()
        rec[8:16] = 100000.0
        w = self.frames[f]["w"]
        rec[8] = w.min() if len(w) else 100000.0
        return rec

    def loss_set_globals(self, g):
        self.globals = g

    def loss_finish(self, g):
        return dict(loss=g[0] + g[1], iou=g[2], min_dist=g[8], sum_m=g[4])

    def loss_backward_local(self, f):
        self.calls.append(("loss_backward_local", f))


def _world(rank, world, port, ids, bz_all, w_all, drift, out, overlap=False, peer=False, halo=None, peer_broken_on=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = np.zeros((len(bz_all), 3)); x[:, 2] = (np.asarray(bz_all) + 0.7) / N
        layout = SlabLayout.balanced(x, N, world, halo)
        assert all(b - a >= 2 * layout.halo and a % 4 == 0 for a, b in zip(layout.bounds, layout.bounds[1:]))
        mine = np.nonzero(layout.owner_of(SlabLayout.stencil_base_z(x, N)) == rank)[0]
        toy = ToyEngine([ids[i] for i in mine], [bz_all[i] for i in mine], [w_all[i] for i in mine], drift, layout, rank)
        if peer_broken_on == rank:                              # this rank cannot map its neighbour's area (no IPC, say)
            def no_ipc(handle):
                raise RuntimeError("hipIpcOpenMemHandle failed: invalid argument")
            toy.peer_open = no_ipc
        eng = SlabEngine(toy, layout, rank, migrate_every=1, overlap=overlap, peer=peer)
        assert eng.native_loops == (peer and world > 1 and peer_broken_on is None), getattr(eng.comm, "peer_error", None)
        if peer_broken_on is not None:
            assert "hipIpcOpenMemHandle" in eng.comm.peer_error   # every rank knows why
        last = STEPS * SUB
        for k in range(STEPS):
            eng.step(k * SUB, SUB)
        assert toy.pending is None
        info = eng.loss_forward(last)
        toy.grad_begin(last)
        for k in reversed(range(STEPS)):
            eng.step_grad(k * SUB, SUB, k)
        out[rank] = dict(read=toy.read, adj=toy.adj_read, final_adj=dict(zip([int(i) for i in toy.frames[0]["ids"]], toy.adj)),
                         chain=[c for c in toy.calls if c[0] == "chain_grad"], info=info, bounds=layout.bounds,
                         order=[c[0] for c in toy.calls], moved=eng.rows_moved, migrations=eng.migrations,
                         counts=[len(toy.frames[k * SUB]["ids"]) for k in range(STEPS + 1)],
                         fwd=[c for c in toy.calls if c[0] in ("p2g", "grid_g2p")])
        dist.barrier()                                          # nobody unmaps an area a neighbour may still write
        toy.close()
    finally:
        dist.destroy_process_group()


def run(world, ids, bz, w, drift, overlap=False, peer=False, halo=None, peer_broken_on=None):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_world, args=(world, free_port(), ids, bz, w, drift, out, overlap, peer, halo, peer_broken_on), nprocs=world, join=True)
    return dict(out)


def test_layout_balanced_and_faces():
    rng = np.random.default_rng(0)
    x = np.zeros((1000, 3)); x[:, 2] = 0.2 + 0.6 * rng.random(1000)
    lay = SlabLayout.balanced(x, 64, 4)
    assert lay.bounds[0] == 0 and lay.bounds[-1] == 64
    assert all(b - a >= 8 and a % 4 == 0 for a, b in zip(lay.bounds, lay.bounds[1:]))      # faces on block planes
    own = lay.owner_of(SlabLayout.stencil_base_z(x, 64))
    counts = np.bincount(own, minlength=4)
    assert counts.min() > 120                                  # roughly balanced (faces snap to multiples of 4)
    f1 = lay.bounds[1] // 4
    assert lay.faces(0) == [(1, f1 - 1, f1 + 1)]              # one block plane either side of the face
    assert len(lay.faces(1)) == 2 and lay.faces(3)[0][0] == 2
    # the exchange planes of a slab's two faces never overlap
    for r in range(1, 3):
        (_, a0, b0), (_, a1, b1) = lay.faces(r)
        assert b0 <= a1
    with pytest.raises(ValueError):
        SlabLayout.balanced(x, 64, 9)                          # 64 layers cannot hold 9 slabs of >= 8
    with pytest.raises(ValueError):
        SlabLayout.balanced(x, 64, 2, halo=3)                  # the reach is 4 layers (3 of drift) or 2 (1 of drift)
    # thin slabs: a reach of 2 layers lets a slab be a single block plane, exchanged with both neighbours
    thin = SlabLayout.balanced(x, 64, 9, halo=2)
    assert thin.world == 9 and thin.halo == 2 and min(b - a for a, b in zip(thin.bounds, thin.bounds[1:])) == 4
    assert SlabLayout.balanced(x, 64, 9, None).bounds == thin.bounds and SlabLayout.balanced(x, 64, 4, None).bounds == lay.bounds
    one_plane = [r for r in range(1, 8) if thin.bounds[r + 1] - thin.bounds[r] == 4]
    (_, a0, b0), (_, a1, b1) = thin.faces(one_plane[0])
    assert (a1, b0) == (a0 + 1, a0 + 2)                        # the slab's own plane lies in both exchange ranges
    with pytest.raises(ValueError):
        SlabLayout.balanced(x, 64, 12, None)                   # the body spans ten block planes
    assert SlabLayout.balanced(x, 64, 1).bounds == (0, 64)
    # grid window of a middle rank: xy box of the cloud + margin, z = slab + one block plane either side
    x[:, 0] = 0.4 + 0.1 * rng.random(1000); x[:, 1] = 0.5
    lo, hi = slab_window(x, 64, lay, 1, xy_margin=3)
    assert (lo[2], hi[2]) == (lay.bounds[1] - 4, lay.bounds[2] + 4)
    assert lo[0] == int(0.4 * 64 - 0.5) - 3 and hi[1] == int(0.5 * 64 - 0.5) + 3 + 3
    lo0, hi0 = slab_window(x, 64, lay, 0, xy_margin=3)
    assert hi0[2] == lay.bounds[1] + 4 and lo0[2] < lay.bounds[1] - 8 and lo0[:2] == lo[:2]   # same xy window on every rank


@pytest.mark.parametrize("WORLD,OVERLAP,PEER", [(2, False, False), (3, False, False), (3, True, False), (2, False, True), (3, False, True)])
def test_ranks_equal_one_rank(WORLD, OVERLAP, PEER):
    """PEER: the device-side exchange protocol (receive areas mapped by name, two halves, arrival counters) and the native
    substep loops it enables, with shared-memory files standing in for IPC-mapped device memory."""
    rng = np.random.default_rng(1)
    n = 48
    ids = list(range(100, 100 + n))
    bz = [int(v) for v in rng.integers(3, 26, n)]              # stencils reach across the faces
    w = [float(v) for v in rng.random(n) + 0.5]
    drift = {i: int(d) for i, d in zip(ids, rng.integers(-1, 2, n))}        # -1 / 0 / +1 layers per env step
    one = run(1, ids, bz, w, drift)[0]
    many = run(WORLD, ids, bz, w, drift, OVERLAP, PEER)
    assert many[0]["bounds"] == many[1]["bounds"]
    if PEER:          # one exchange per grid phase, inside the native loops
        assert many[0]["order"].count("peer_exchange") == 2 * STEPS * SUB + 1              # + the loss mass grid
    if OVERLAP:       # every substep ran its interior pass between the scatter and the face pass, forward and reverse
        for r in range(WORLD):
            o = many[r]["order"]
            assert o.count("grid_interior") == STEPS * SUB and o.count("grad_gather_interior") == STEPS * SUB
            assert all(o[i + 1] == "grid_g2p" for i, c in enumerate(o) if c == "grid_interior")
    assert sum(many[r]["moved"] for r in range(WORLD)) > 0, "the toy rollout was meant to migrate rows"
    assert all(many[r]["migrations"] == STEPS - 1 for r in range(WORLD))    # before every env step but the first
    for k in range(STEPS + 1):
        assert sum(many[r]["counts"][k] for r in range(WORLD)) == n         # nobody lost, nobody duplicated
    for f in range(STEPS * SUB):
        got, adj = {}, {}
        for r in range(WORLD):
            assert not set(got) & set(many[r]["read"][f]), "a particle lives on two ranks"
            got.update(many[r]["read"][f]); adj.update(many[r]["adj"][f])
        assert set(got) == set(one["read"][f])
        for i in one["read"][f]:
            assert abs(got[i] - one["read"][f][i]) < 1e-12 * max(1.0, abs(got[i]))                  # forward halo sum exchange (+ migrated rows)
            assert abs(adj[i] - one["adj"][f][i]) < 1e-12 * max(1.0, abs(adj[i]))                   # reverse exchange; adjoint rows went back home
    fin = {}
    for r in range(WORLD):
        fin.update(many[r]["final_adj"])
    assert set(fin) == set(one["final_adj"]) and all(abs(fin[i] - one["final_adj"][i]) < 1e-12 * max(1.0, abs(fin[i])) for i in fin)
    # pose adjoints: owned-node contributions summed over ranks == single-rank total, on every rank
    for a, b in zip(many[0]["chain"], one["chain"]):
        assert abs(a[4] - b[4]) < 1e-12 * max(1.0, abs(b[4])) and a[:4] == b[:4]
    assert many[0]["chain"] == many[1]["chain"]
    # loss record: sums, max and min combine correctly
    for k in ("loss", "iou", "min_dist", "sum_m"):
        assert abs(many[0]["info"][k] - one["info"][k]) < 1e-12 * max(1.0, abs(one["info"][k])) and many[0]["info"][k] == many[1]["info"][k]
    assert many[0]["order"][0] == "fk" and many[0]["order"][-1] == "chain_grad"
    # g2p of every substep but an env step's last runs fused with the next p2g
    assert many[0]["fwd"][:4] == [("p2g", 0, False), ("grid_g2p", 0, True), ("p2g", 1, True), ("grid_g2p", 1, False)]


@pytest.mark.parametrize("PEER", [False, True])
def test_eight_thin_slabs_equal_one_rank(PEER):
    """World size 8 on a body of eight block planes: every slab is ONE block plane (reach 2 layers: one of stencil, one of
    drift), so a rank's plane is exchanged with both neighbours and both received copies are added -- the layout
    `bench.py --gpus 8` needs for the 40-layer cube of config 3."""
    rng = np.random.default_rng(2)
    n = 96
    ids = list(range(500, 500 + n))
    bz = [int(v) for v in rng.integers(0, 29, n)] + []
    bz[:8] = [4 * k + 1 for k in range(8)]                     # every plane holds a stencil centre
    w = [float(v) for v in rng.random(n) + 0.5]
    # one layer of drift per env step at most, away from the walls
    drift = {i: (int(rng.integers(0, 2)) if b < 15 else -int(rng.integers(0, 2))) for i, b in zip(ids, bz)}
    one = run(1, ids, bz, w, drift)[0]
    many = run(8, ids, bz, w, drift, peer=PEER)        # PEER: a one-plane slab's plane goes into BOTH neighbours' receive areas
    b = many[0]["bounds"]
    assert len(b) == 9 and all(hi - lo == 4 for lo, hi in zip(b, b[1:]))
    assert sum(many[r]["moved"] for r in range(8)) > 0
    for k in range(STEPS + 1):
        assert sum(many[r]["counts"][k] for r in range(8)) == n
    for f in range(STEPS * SUB):
        got, adj = {}, {}
        for r in range(8):
            assert not set(got) & set(many[r]["read"][f])
            got.update(many[r]["read"][f]); adj.update(many[r]["adj"][f])
        assert set(got) == set(one["read"][f])
        for i in one["read"][f]:
            assert abs(got[i] - one["read"][f][i]) < 1e-12 * max(1.0, abs(got[i]))
            assert abs(adj[i] - one["adj"][f][i]) < 1e-12 * max(1.0, abs(adj[i]))
    fin = {}
    for r in range(8):
        fin.update(many[r]["final_adj"])
    assert set(fin) == set(one["final_adj"]) and all(abs(fin[i] - one["final_adj"][i]) < 1e-12 * max(1.0, abs(fin[i])) for i in fin)
    for a, c in zip(many[0]["chain"], one["chain"]):
        assert abs(a[4] - c[4]) < 1e-12 * max(1.0, abs(c[4]))
    for k in ("loss", "iou", "min_dist", "sum_m"):
        assert abs(many[0]["info"][k] - one["info"][k]) < 1e-12 * max(1.0, abs(one["info"][k]))


def test_peer_setup_failure_on_one_rank_falls_back_everywhere():
    """One rank cannot map its neighbour's receive area: the peer path is off on EVERY rank (agreed collectively, with the
    reason), the torch.distributed point-to-point exchange takes over, and the results are still those of one rank."""
    rng = np.random.default_rng(3)
    n = 40
    ids = list(range(300, 300 + n))
    bz = [int(v) for v in rng.integers(3, 26, n)]
    w = [float(v) for v in rng.random(n) + 0.5]
    drift = {i: int(d) for i, d in zip(ids, rng.integers(-1, 2, n))}
    one = run(1, ids, bz, w, drift)[0]
    many = run(3, ids, bz, w, drift, peer=True, peer_broken_on=1)
    assert all("peer_exchange" not in many[r]["order"] for r in range(3))
    for f in range(STEPS * SUB):
        got = {}
        for r in range(3):
            got.update(many[r]["read"][f])
        assert set(got) == set(one["read"][f])
        assert all(abs(got[i] - one["read"][f][i]) < 1e-12 * max(1.0, abs(got[i])) for i in got)
    for k in ("loss", "iou", "min_dist", "sum_m"):
        assert abs(many[0]["info"][k] - one["info"][k]) < 1e-12 * max(1.0, abs(one["info"][k]))


def _world_fail(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bz_all = list(range(4, 24))
        x = np.zeros((len(bz_all), 3)); x[:, 2] = (np.asarray(bz_all) + 0.7) / N
        layout = SlabLayout.balanced(x, N, world)
        mine = np.nonzero(layout.owner_of(SlabLayout.stencil_base_z(x, N)) == rank)[0]
        toy = ToyEngine([100 + i for i in mine], [bz_all[i] for i in mine], [1.0] * len(mine), {100 + i: 0 for i in range(len(bz_all))}, layout, rank)
        if rank == 1:
            def boom(f):
                raise RuntimeError("migrate: 99999 rows leave at once, the row buffers hold 4096")
            toy.migrate_begin = boom
        eng = SlabEngine(toy, layout, rank, migrate_every=1)
        eng.step(0, SUB)
        try:
            eng.step(SUB, SUB)                     # migrates first: rank 1 fails inside migrate_begin
            out[rank] = "no exception"
        except RuntimeError as e:
            out[rank] = str(e)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_a_migration_failure_on_one_rank_raises_on_every_rank():
    """A host-side error inside plmpm_migrate_begin / _finish happens on one rank only; the others must not be left
    waiting in the row exchange: the ok flag is all-reduced and every rank raises."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_world_fail, args=(2, free_port(), out), nprocs=2, join=True)
    assert "rows leave at once" in out[1]
    assert "another rank failed in migrate_begin" in out[0]


def _world_gpu_share(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from plasticinelab_amd import distributed as D
        real = D.gpu_identity
        out[("plain", rank)] = D.ranks_sharing_my_gpu()
        # three ranks, two "devices" on one host: ranks 0 and 2 share one
        D.gpu_identity = lambda: ("host", f"gpu{rank % 2}")
        out[("mixed", rank)] = D.ranks_sharing_my_gpu()
        D.gpu_identity = real
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_ranks_per_gpu_comes_from_an_all_gather_of_device_identities():
    """`processes / visible devices` says one rank per GPU when several ranks are bound to ONE device while more are
    visible; the residency cap of the fused exchange + grid kernels must count the ranks that really share this process's GPU."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_world_gpu_share, args=(3, free_port(), out), nprocs=3, join=True)
    assert [out[("plain", r)] for r in range(3)] == [3, 3, 3]              # no GPU here: every rank reports (host, "cpu")
    assert [out[("mixed", r)] for r in range(3)] == [2, 1, 2]
