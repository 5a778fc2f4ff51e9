"""CPU tier, world_size 2 over gloo: the multi-GPU orchestration (slab layout, symmetric halo sum exchange,
flag merging, loss-record reduction, pose-adjoint sum + step ordering) with a linear toy engine standing in for
the HIP engine -- the distributed logic itself has no GPU dependency.  The real kernels behind the same
SlabEngine are checked 2-rank vs 1-rank in tests/test_gpu_distributed.py (-m gpu)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from plasticinelab_amd.distributed import HaloComm, SlabEngine, SlabLayout

N = 16          # toy grid


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class ToyEngine:
    """Implements the phase API of Engine on CPU tensors with linear toy physics: particles deposit their
    (id-dependent) weight on the 3 node planes of their stencil; grid_g2p reads it back; the reverse pass sends
    a cotangent the same way.  Linear, so 2-rank results must equal the 1-rank ones exactly."""
    HALO_GRID_IN, HALO_GRID_OUT_ADJ, HALO_LOSS_MASS = 0, 1, 2
    device = torch.device("cpu")

    def __init__(self, base_z, weight, layout, rank):
        self.bz, self.w, self.layout, self.rank = base_z, weight, layout, rank
        self.gin, self.goa, self.flags = {}, {}, {}
        self.read, self.adj_read = {}, {}
        self.pos_l = torch.zeros(64, dtype=torch.float64)
        self.rot_l = torch.zeros(64, dtype=torch.float64)
        self.calls = []
        self.pending = None

    def fk(self, first, n):
        self.calls.append(("fk", first, n))

    def p2g(self, f):
        g = torch.zeros(4, N, N, N, dtype=torch.float64)             # [comp, z, y, x]
        fl = torch.zeros(N // 4 * (N // 4) ** 2, dtype=torch.int32)
        for b, w in zip(self.bz, self.w):
            for k in range(3):
                g[:, b + k, 1, 2] += w * (k + 1) * (f + 1)
            fl[(b // 4) * (N // 4) ** 2] = 1
            fl[((b + 2) // 4) * (N // 4) ** 2] = 1
        self.gin[f], self.flags[f] = g, fl

    def grid_g2p(self, f):                                           # each particle reads its 3 planes
        self.read[f] = np.array([float(self.gin[f][0, b:b + 3, 1, 2].sum()) for b in self.bz])

    def grad_scatter(self, f):
        g = torch.zeros(3, N, N, N, dtype=torch.float64)
        for b, w in zip(self.bz, self.w):
            g[:, b:b + 3, 1, 2] += w
        self.goa = g

    def grad_gather(self, f):
        z0, z1 = self.layout.slab(self.rank)
        self.adj_read[f] = np.array([float(self.goa[0, b:b + 3, 1, 2].sum()) for b in self.bz])
        self.pos_l[f] += float(self.goa[0, z0:z1].sum())           # owned planes only, like k_grid_op_grad

    def chain_grad(self, first, n, step):
        self.calls.append(("chain_grad", first, n, step, float(self.pos_l[first:first + n + 1].sum())))

    def halo_pack(self, field, f, za, zb, out=None):
        src = {0: self.gin.get(f), 1: self.goa, 2: getattr(self, "lm", None)}[field]
        if out is not None:
            return out.copy_(src[:, za:zb])
        return src[:, za:zb].clone()

    def halo_unpack_add(self, field, f, za, zb, buf):
        src = {0: self.gin.get(f), 1: self.goa, 2: getattr(self, "lm", None)}[field]
        src[:, za:zb] += buf

    def slab_pre(self, field, f, faces, chain=False):
        if chain:                                       # g2p(f - 1) was deferred: it runs with this p2g, as in the library
            assert field == 0 and self.pending == f - 1, (field, f, self.pending)
            self.grid_g2p(f - 1)
            self.pending = None
        else:
            assert self.pending is None
        (self.p2g if field == 0 else self.grad_scatter)(f)
        for fc in faces:
            self.halo_pack(field, f, fc.za, fc.zb, out=fc.send)
        self.calls.append(("slab_pre", field, f, bool(chain)))

    def slab_post(self, field, f, faces, chain=False):
        for fc in faces:
            self.halo_unpack_add(field, f, fc.za, fc.zb, fc.recv)
        if field == 0 and chain:
            self.pending = f                            # plmpm_slab_post(chain=1): grid_op only, g2p left pending
        else:
            (self.grid_g2p if field == 0 else self.grad_gather)(f)
        self.calls.append(("slab_post", field, f, bool(chain)))

    def flags_view(self, f, bza, bzb):
        m = (N // 4) ** 2
        return self.flags[f][bza * m:bzb * m]

    def pose_grad_views(self, first, nf):
        return self.pos_l[first:first + nf], self.rot_l[first:first + nf]

    def loss_set_weights(self, *a):
        pass

    def loss_scatter(self, f):
        self.lm = torch.zeros(1, N, N, N, dtype=torch.float64)
        for b, w in zip(self.bz, self.w):
            self.lm[0, b:b + 3, 1, 2] += w

    def loss_partials(self, f, phase):
        z0, z1 = self.layout.slab(self.rank)
        rec = np.zeros(32)
        own = self.lm[0, z0:z1]
        rec[0], rec[1], rec[2], rec[3], rec[4] = own.abs().sum(), 2 * own.sum(), own.max(), 3 * own.sum(), own.sum()
        rec[8:16] = 100000.0
        rec[8] = min(self.w) if len(self.w) else 100000.0
        return rec

    def loss_set_globals(self, g):
        self.globals = g

    def loss_finish(self, g):
        return dict(loss=g[0] + g[1], iou=g[2], min_dist=g[8], sum_m=g[4])

    def loss_backward_local(self, f):
        self.calls.append(("loss_backward_local", f))

    def error_flags(self):
        return 0

    def check_error(self, flags=None):
        pass


def _world(rank, world, port, bz_all, w_all, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = np.zeros((len(bz_all), 3)); x[:, 2] = (np.asarray(bz_all) + 0.7) / N
        layout = SlabLayout.balanced(x, N, world, halo=2)
        assert all(b - a >= 4 for a, b in zip(layout.bounds, layout.bounds[1:]))
        mine = np.nonzero(layout.owner_of(SlabLayout.stencil_base_z(x, N)) == rank)[0]
        toy = ToyEngine([bz_all[i] for i in mine], [w_all[i] for i in mine], layout, rank)
        eng = SlabEngine(toy, layout, rank) if world > 1 else None
        if world == 1:
            # reference semantics without any exchange
            toy.fk(0, 3)
            for f in (0, 1, 2):
                toy.p2g(f); toy.grid_g2p(f)
            for f in (2, 1, 0):
                toy.grad_scatter(f); toy.grad_gather(f)
            toy.chain_grad(0, 3, 0)
            toy.loss_scatter(3); rec = toy.loss_partials(3, 0); info = toy.loss_finish(rec)
        else:
            eng.step(0, 3)
            assert toy.pending is None
            eng.step_grad(0, 3, 0)
            info = eng.loss_forward(3)
            # substep 0 built the plans through the unfused calls; after that one library call each side of the
            # exchange, with g2p(1) deferred into the p2g of substep 2
            fwd = [c for c in toy.calls if c[0].startswith("slab_") and c[1] == 0]
            assert fwd == [("slab_pre", 0, 1, False), ("slab_post", 0, 1, True), ("slab_pre", 0, 2, True), ("slab_post", 0, 2, False)], fwd
            bwd = [c for c in toy.calls if c[0].startswith("slab_") and c[1] == 1]
            assert [c[2] for c in bwd] == [1, 1, 0, 0] and not any(c[3] for c in bwd), bwd
        out[rank] = dict(mine=mine, read=toy.read, adj=toy.adj_read, flags={f: toy.flags[f].numpy().copy() for f in toy.flags},
                         chain=[c for c in toy.calls if c[0] == "chain_grad"], info=info,
                         bounds=layout.bounds, order=[c[0] for c in toy.calls])
    finally:
        dist.destroy_process_group()


def run(world, bz, w):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_world, args=(world, free_port(), bz, w, out), nprocs=world, join=True)
    return dict(out)


def test_layout_balanced_and_faces():
    rng = np.random.default_rng(0)
    x = np.zeros((1000, 3)); x[:, 2] = 0.3 + 0.4 * rng.random(1000)
    lay = SlabLayout.balanced(x, 64, 4, halo=2)
    assert lay.bounds[0] == 0 and lay.bounds[-1] == 64 and all(b - a >= 4 for a, b in zip(lay.bounds, lay.bounds[1:]))
    own = lay.owner_of(SlabLayout.stencil_base_z(x, 64))
    counts = np.bincount(own, minlength=4)
    assert counts.min() > 150                                  # roughly balanced
    assert lay.faces(0) == [(1, lay.bounds[1] - 2, lay.bounds[1] + 2)]
    assert len(lay.faces(1)) == 2 and lay.faces(3)[0][0] == 2
    with pytest.raises(ValueError):
        SlabLayout.balanced(x, 64, 40, halo=2)                 # slabs would be thinner than 2*halo
    assert SlabLayout.balanced(x, 64, 1).bounds == (0, 64)


@pytest.mark.parametrize("WORLD", [2, 3])
def test_ranks_equal_one_rank(WORLD):
    rng = np.random.default_rng(1)
    bz = [int(v) for v in rng.integers(1, 13, 40)]             # bases 1..12 -> stencils reach across the faces
    w = [float(v) for v in rng.random(40) + 0.5]
    one = run(1, bz, w)[0]
    two = run(WORLD, bz, w)
    assert two[0]["bounds"] == two[1]["bounds"] and sum(len(two[r]["mine"]) for r in range(WORLD)) == 40
    for f in (0, 1, 2):
        got = np.empty(40); adj = np.empty(40)
        for r in range(WORLD):
            got[two[r]["mine"]] = two[r]["read"][f]
            adj[two[r]["mine"]] = two[r]["adj"][f]
        assert np.allclose(got, one["read"][f], rtol=0, atol=1e-12)       # forward halo sum exchange
        assert np.allclose(adj, one["adj"][f], rtol=0, atol=1e-12)        # reverse halo sum exchange
        merged = np.maximum.reduce([two[r]["flags"][f] for r in range(WORLD)])
        assert np.array_equal(merged, one["flags"][f])                    # block flags OR-merged
    # pose adjoints: owned-node contributions summed over ranks == single-rank total, on every rank
    assert abs(two[0]["chain"][0][4] - one["chain"][0][4]) < 1e-12 and two[0]["chain"] == two[1]["chain"]
    # loss record: sums, max and min combine correctly
    for k in ("loss", "iou", "min_dist", "sum_m"):
        assert abs(two[0]["info"][k] - one["info"][k]) < 1e-12 and two[0]["info"][k] == two[1]["info"][k]
    assert two[0]["order"][0] == "fk" and two[0]["order"][-1] == "chain_grad"
