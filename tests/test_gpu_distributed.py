"""-m gpu: the real HIP kernels behind the z-slab SlabEngine -- windowed grids, zero-copy block-plane halos added
inside grid_op / grid_op.grad, particle migration with adjoint rows going back -- N ranks vs the single-rank engine
and the oracle-generated golden rollout (which the single-rank engine reproduces to 1e-10, test_gpu_rollout).  The
ranks share the box's one GPU and talk over gloo; on a multi-GPU node the same code runs over RCCL
(PLB_DIST_BACKEND=nccl, self-skipping below two GPUs)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests.util import GOLDEN, ROOT
from tests.gpu_util import relerr

pytestmark = pytest.mark.gpu


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(tmp_path, world, dtype, actions, xy_margin, migrate_every, backend="gloo", scene=None, overlap=False, deterministic=False, halo=None,
           segment=0, peer=False, fused=False):
    out = str(tmp_path / "r")
    act = str(tmp_path / "actions.npy")
    np.save(act, actions)
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", PLB_DIST_BACKEND=backend, PLB_TEST_OVERLAP="1" if overlap else "0", PLB_TEST_DETERMINISTIC="1" if deterministic else "0",
                   PLB_TEST_HALO="" if halo is None else str(halo), PLB_TEST_SEGMENT=str(segment) if segment else "", PLB_TEST_PEER="1" if peer else "0", PLMPM_PEER_TIMEOUT="60",
                   PLMPM_PEER_FUSED="1" if fused else "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), out, dtype, act,
                                       "none" if xy_margin is None else str(xy_margin), str(migrate_every)]
                                      + ([json.dumps(scene)] if scene else []),
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=900)[0].decode() for p in procs]
    if any(p.returncode != 0 for p in procs):
        raise AssertionError("\n".join(f"--- rank {r} (rc {p.returncode})\n{lg[-2500:]}" for r, (p, lg) in enumerate(zip(procs, logs))))
    return [np.load(f"{out}.{r}.npz") for r in range(world)]


def gather(res, n):
    x = np.full((n, 3), np.nan); v = np.full((n, 3), np.nan)
    seen = np.zeros(n, int)
    for r in res:
        x[r["ids"]] = r["x"]; v[r["ids"]] = r["v"]
        seen[r["ids"]] += 1
    assert (seen == 1).all(), "every particle lives on exactly one rank"
    return x, v


# xy_margin: whole planes allocated (None) or only the body's xy bounding box + that many node layers
@pytest.mark.parametrize("dtype,world,xy_margin", [("float64", 2, None), ("float32", 2, 8), ("float64", 3, 6)])
def test_slab_ranks_match_golden_rollout(tmp_path, dtype, world, xy_margin):
    g = np.load(os.path.join(GOLDEN, "rollout_small.npz"))
    res = launch(tmp_path, world, dtype, g["actions"], xy_margin, migrate_every=1)
    ltol, gtol, xtol = (1e-10, 1e-7, 1e-10) if dtype == "float64" else (1e-5, 1e-4, 2e-5)
    n = int(g["n_particles"])
    assert sum(int(r["count"]) for r in res) == n and min(int(r["count"]) for r in res) > 0
    for r in res:
        assert abs(float(r["loss"]) - float(g["loss"])) / abs(float(g["loss"])) < ltol      # every rank has the full loss
        assert relerr(r["grad"], g["grad"]) < gtol                                           # ... and the full gradient
        assert int(r["migrations"]) == len(g["actions"]) - 1
    x, v = gather(res, n)
    assert relerr(x, g["x_final"]) < xtol
    assert relerr(v, g["v_final"]) < (1e-8 if dtype == "float64" else 2e-3)
    if xy_margin is not None:                       # the window really is a strict part of the 64^3 grid
        o, b = res[0]["window"][:3], res[0]["window"][3:]
        assert (b[:2] * 4 < 64).all()


@pytest.mark.parametrize("dtype,world", [("float64", 3), ("float32", 2)])
def test_overlapped_exchange_matches_golden_rollout(tmp_path, dtype, world):
    """SlabEngine(overlap=True): grid_op / grid_op.grad of the blocks outside the exchanged planes are launched before the
    halos are waited for (plmpm_grid_interior / plmpm_grad_gather_interior), the planes themselves afterwards.  Same
    arithmetic per block, so the same results as the plain order -- here against the golden rollout, middle rank with
    two faces included."""
    g = np.load(os.path.join(GOLDEN, "rollout_small.npz"))
    res = launch(tmp_path, world, dtype, g["actions"], 6, migrate_every=1, overlap=True)
    ltol, gtol, xtol = (1e-10, 1e-7, 1e-10) if dtype == "float64" else (1e-5, 1e-4, 2e-5)
    for r in res:
        assert abs(float(r["loss"]) - float(g["loss"])) / abs(float(g["loss"])) < ltol
        assert relerr(r["grad"], g["grad"]) < gtol
    x, v = gather(res, int(g["n_particles"]))
    assert relerr(x, g["x_final"]) < xtol
    assert relerr(v, g["v_final"]) < (1e-8 if dtype == "float64" else 2e-3)


def single_rank(actions, dtype):
    from tests.test_gpu_rollout import make_env_sub, run_forward
    env = make_env_sub("Move", 2000, dtype)
    loss, grad = run_forward(env, actions)
    fr = env.simulator.engine.get_frame(env.simulator.cur, want=("x", "v"))
    return loss, grad, fr["x"], fr["v"]


@pytest.mark.parametrize("world,migrate_every", [(2, 1), (3, 2)])
def test_migration_over_a_long_rollout(tmp_path, world, migrate_every):
    """10 env steps with the manipulators shoving the body along z: rows cross the slab faces, ownership follows them
    (without migration the run raises, see below), and loss / gradient / trajectory stay those of one rank."""
    H = 10
    acts = np.zeros((H, 6))
    acts[:, 2] = 0.9; acts[:, 5] = 0.9              # both spheres push +z
    acts[:, 0] = 0.5; acts[:, 3] = -0.5             # ... and squeeze
    acts += np.random.default_rng(4).uniform(-0.1, 0.1, acts.shape)
    loss, grad, x1, v1 = single_rank(acts, "float64")
    res = launch(tmp_path, world, "float64", acts, 10, migrate_every)
    assert sum(int(r["rows_moved"]) for r in res) > 0, "the rollout was meant to move rows across the faces"
    for r in res:
        assert abs(float(r["loss"]) - loss) / abs(loss) < 1e-9
        assert relerr(r["grad"], grad) < 1e-7
    x, v = gather(res, 2000)
    assert relerr(x, x1) < 1e-9 and relerr(v, v1) < 1e-7


def test_config4_grid_in_four_slabs(tmp_path):
    """BASELINE configs[3] geometry -- the elastic block on a 256^3 grid, 79 substeps per env step -- cut into 4 z-slabs
    (all four ranks on the box's one GPU, halos over gloo) for 8 env steps with migration: every rank reproduces the
    single-rank loss and action gradient; the ranks' windowed per-frame grid stores are a fraction of a dense one.
    200k particles keep four processes + the single-rank run inside the test's time (the 2M-particle workload itself runs
    on one GPU in test_gpu_fullsize.py)."""
    import bench
    import torch
    from plasticinelab_amd.engine.taichi_env import TaichiEnv
    from plasticinelab_amd.optimizer.solver import Solver
    scene = dict(particles=200_000, quality=4, side=0.25, yield_stress=1e9)
    H, margin = 8, 24
    acts = bench.seeded_actions(H, 6)
    # single rank, on the same grid window the slab ranks' windows tile (a dense 256^3 store of 633 frames would not fit)
    cfg = bench.workload_cfg(scene["particles"], scene["quality"], max_steps=H * 79 + 1, yield_stress=1e9, side=0.25)
    from plasticinelab_amd.engine.shapes import Shapes
    x_all, _ = Shapes(cfg.SHAPES).get()
    b = (x_all * 256 - 0.5).astype(np.int64)
    cfg.SIMULATOR["grid_window"] = ([int(v) for v in np.maximum(b.min(0) - margin, 0)], [int(v) for v in np.minimum(b.max(0) + 3 + margin, 256)])
    env = TaichiEnv(cfg, compute_dtype="float64")
    env.initialize()
    env.loss.load_target_density(grids=bench._target(env.init_particles, env.simulator))
    env.loss.set_weights(10, 10, 1, False)
    loss, grad = Solver(env, None, None, softness=666.0, horizon=H).forward(env.get_state()["state"], acts)
    single_grid_bytes = env.simulator.engine.workspace_bytes["grid_bytes"]
    env.simulator.engine.close()
    del env
    torch.cuda.empty_cache()
    res = launch(tmp_path, 4, "float64", acts, margin, 1, scene=scene)
    assert [int(v) for v in res[0]["bounds"]][1:-1] == sorted(set(int(v) for v in res[0]["bounds"][1:-1]))
    for r in res:
        assert abs(float(r["loss"]) - loss) / abs(loss) < 1e-9
        assert relerr(r["grad"], grad) < 1e-7
        assert int(r["migrations"]) == H - 1
        assert int(r["grid_bytes"]) < 0.6 * single_grid_bytes            # a rank stores its slab + halo planes only
    assert sum(int(r["count"]) for r in res) == scene["particles"]
    print(f"\n[256^3 in 4 slabs] loss {loss:.9g}, rows moved {[int(r['rows_moved']) for r in res]}, grid store per rank "
          f"{[round(int(r['grid_bytes']) / 2**30, 2) for r in res]} GiB vs {single_grid_bytes / 2**30:.2f} GiB on one rank")


def test_config4_full_size_split_over_two_ranks(tmp_path):
    """BASELINE configs[3] at its own size -- 256^3 grid, 2M elastic particles, 79 substeps per env step -- as the
    config names it: split into z-slabs over 2 ranks (both on the box's one GPU, halos over gloo), 2 env steps fwd+bwd
    with a migration in between, against the single-rank run of the same workload (float64 engine: the comparison is
    then a statement about the decomposition, not about summation order)."""
    import bench
    import torch
    from plasticinelab_amd.engine.shapes import Shapes
    from plasticinelab_amd.engine.taichi_env import TaichiEnv
    from plasticinelab_amd.optimizer.solver import Solver
    scene = dict(particles=2_000_000, quality=4, side=0.25, yield_stress=1e9)
    H, margin = 2, 16
    acts = bench.seeded_actions(H, 6)
    cfg = bench.workload_cfg(scene["particles"], scene["quality"], max_steps=H * 79 + 1, yield_stress=1e9, side=0.25)
    x_all, _ = Shapes(cfg.SHAPES).get()
    b = (x_all * 256 - 0.5).astype(np.int64)
    cfg.SIMULATOR["grid_window"] = ([int(v) for v in np.maximum(b.min(0) - margin, 0)], [int(v) for v in np.minimum(b.max(0) + 3 + margin, 256)])
    env = TaichiEnv(cfg, compute_dtype="float64")
    env.initialize()
    env.loss.load_target_density(grids=bench._target(env.init_particles, env.simulator))
    env.loss.set_weights(10, 10, 1, False)
    loss, grad = Solver(env, None, None, softness=666.0, horizon=H).forward(env.get_state()["state"], acts)
    env.simulator.engine.close()
    del env
    torch.cuda.empty_cache()
    res = launch(tmp_path, 2, "float64", acts, margin, 1, scene=scene)
    for r in res:
        assert abs(float(r["loss"]) - loss) / abs(loss) < 1e-9
        assert relerr(r["grad"], grad) < 1e-7
        assert int(r["migrations"]) == H - 1
    assert sum(int(r["count"]) for r in res) == scene["particles"]
    print(f"\n[256^3 / 2M particles in 2 slabs] loss {loss:.9g}, particles per rank {[int(r['count']) for r in res]}, rows moved {[int(r['rows_moved']) for r in res]}")


@pytest.mark.parametrize("dtype,world,fused", [("float64", 2, True), ("float64", 3, True), ("float32", 3, True), ("float64", 3, False), ("float32", 2, False)])
def test_peer_write_halos_match_golden_rollout(tmp_path, dtype, world, fused):
    """Device-side halo exchange (csrc/plmpm_peer.hip): receive areas in uncached device memory mapped by the
    neighbours through IPC handles, the substep loops native (plmpm_slab_step / plmpm_slab_step_grad) -- no host-side
    communication per substep.  Not fused (the default): one kernel per exchange (copy, publish, wait) in front of the grid
    kernel; fused (PLMPM_PEER_FUSED=1): each exchange is part of the grid kernel that consumes it (send the owned blocks of the
    exchanged planes | interior blocks | wait for the neighbours | blocks of the exchanged planes).  Same planes and same
    node arithmetic as the torch.distributed transport, so the same results: the golden rollout with migration every env step,
    a middle rank with two faces included."""
    g = np.load(os.path.join(GOLDEN, "rollout_small.npz"))
    res = launch(tmp_path, world, dtype, g["actions"], 6, migrate_every=1, peer=True, fused=fused)
    assert all(int(r["fused"]) == int(fused) for r in res)
    ltol, gtol, xtol = (1e-10, 1e-7, 1e-10) if dtype == "float64" else (1e-5, 1e-4, 2e-5)
    assert all(int(r["native_loops"]) == 1 for r in res)
    for r in res:
        assert abs(float(r["loss"]) - float(g["loss"])) / abs(float(g["loss"])) < ltol
        assert relerr(r["grad"], g["grad"]) < gtol
        assert int(r["migrations"]) == len(g["actions"]) - 1
    x, v = gather(res, int(g["n_particles"]))
    assert relerr(x, g["x_final"]) < xtol
    assert relerr(v, g["v_final"]) < (1e-8 if dtype == "float64" else 2e-3)


@pytest.mark.parametrize("world,segment,peer", [(2, 2, False), (3, 1, False), (3, 2, True)])
def test_segment_checkpointed_backward_on_slab_ranks(tmp_path, world, segment, peer):
    """optimizer/checkpoint.py on z-slab ranks (long_term_gradient.ipynb cells 2-4 on a population that migration keeps
    changing): 6 env steps in segments of 2 (3 segments) / 1 (6 segments), migration before every env step; a checkpoint
    is the rank's rows at the boundary (ids, state, materials), a segment re-enters the engine with them as a new
    population, the carried adjoint is matched by global id.  Loss and action gradient equal the single-rank
    stored-trajectory run."""
    H = 6
    acts = np.zeros((H, 6))
    acts[:, 2] = 0.9; acts[:, 5] = 0.9              # both spheres push +z: rows cross the slab faces
    acts[:, 0] = 0.5; acts[:, 3] = -0.5
    acts += np.random.default_rng(5).uniform(-0.1, 0.1, acts.shape)
    loss, grad, x1, v1 = single_rank(acts, "float64")
    res = launch(tmp_path, world, "float64", acts, 10, 1, segment=segment, peer=peer)
    assert sum(int(r["rows_moved"]) for r in res) > 0
    for r in res:
        assert abs(float(r["loss"]) - loss) / abs(loss) < 1e-9
        assert relerr(r["grad"], grad) < 1e-7


@pytest.mark.parametrize("peer,fused", [(False, False), (True, False), (True, True)])
def test_thin_slabs_one_block_plane_per_rank(tmp_path, peer, fused):
    """The layout `bench.py --gpus 8` uses on config 3: a reach of 2 node layers (one of stencil, one of drift) lets a slab
    be ONE block plane, which then lies in the exchange range of both its faces -- it goes to both neighbours, and
    k_grid_op / k_grid_op_grad add both received copies.  Here: the benchmark's cube on a 64^3 grid (20 layers = 5-6 block
    planes) cut into 5 slabs, 4 env steps with migration every step, against the single-rank run."""
    import bench
    import torch
    from plasticinelab_amd.engine.taichi_env import TaichiEnv
    from plasticinelab_amd.optimizer.solver import Solver
    scene = dict(particles=20_000, quality=1, side=0.31, yield_stress=200.0)
    H = 4
    acts = bench.seeded_actions(H, 6)
    cfg = bench.workload_cfg(scene["particles"], scene["quality"], max_steps=H * 19 + 1, yield_stress=200.0, side=0.31)
    env = TaichiEnv(cfg, compute_dtype="float64")
    env.initialize()
    env.loss.load_target_density(grids=bench._target(env.init_particles, env.simulator))
    env.loss.set_weights(10, 10, 1, False)
    loss, grad = Solver(env, None, None, softness=666.0, horizon=H).forward(env.get_state()["state"], acts)
    env.simulator.engine.close()
    del env
    torch.cuda.empty_cache()
    res = launch(tmp_path, 5, "float64", acts, 10, 1, scene=scene, halo=2, peer=peer, fused=fused)
    assert all(int(r["native_loops"]) == int(peer) and int(r["fused"]) == int(fused) for r in res)
    b = [int(v) for v in res[0]["bounds"]]
    assert min(hi - lo for lo, hi in zip(b[1:-2], b[2:-1])) == 4, b          # the middle slabs are single block planes
    for r in res:
        assert abs(float(r["loss"]) - loss) / abs(loss) < 1e-9
        assert relerr(r["grad"], grad) < 1e-7
        assert int(r["migrations"]) == H - 1
    assert sum(int(r["count"]) for r in res) == scene["particles"]
    print(f"\n[thin slabs] bounds {b}, rows moved {[int(r['rows_moved']) for r in res]}")


def test_peer_exchange_wait_is_bounded(monkeypatch):
    """A neighbour that never publishes its arrival must not hang the GPU: the exchange kernel gives up after
    PLMPM_PEER_TIMEOUT seconds, leaves a status word, and SlabEngine._check (or the next exchange call) raises.  One
    process, middle rank of a 3-slab layout, "neighbours'" areas that nobody ever writes a counter into."""
    import sys
    import time
    import torch
    import bench
    sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))
    import slab_host_cost as shc
    from plasticinelab_amd import distributed as D

    class Silent(shc.LoopbackComm):
        def setup_peer(self, engine):
            faces = self.layout.faces(self.rank)
            for field in (engine.HALO_GRID_IN, engine.HALO_GRID_OUT_ADJ, engine.HALO_LOSS_MASS):
                local = [engine.peer_alloc(field, a, b)[0] for _n, a, b in faces]
                remote = [engine.peer_alloc(field, a, b)[0] for _n, a, b in faces]      # written to, never answered from
                engine.halo_peer_setup(field, [(a, b) for _n, a, b in faces], local, remote)
            self.peer_ready = True
            return True

    monkeypatch.setenv("PLMPM_PEER_TIMEOUT", "0.3")
    cfg = bench.workload_cfg(20_000, 1, max_steps=40)
    layout = D.SlabLayout(64, (0, 16, 36, 64))
    env, _, _ = D.make_slab_env(cfg, 1, 3, compute_dtype="float32", target_fn=bench._target, layout=layout,
                                comm=Silent(layout, 1, True), xy_margin=8, migrate_every=0)
    eng = env.simulator.engine
    assert eng.native_loops and eng.peer_status() == 0
    t0 = time.time()
    # the first exchange -- the loss mass grid, when set_state evaluates the initial loss -- already times out
    with pytest.raises(RuntimeError, match="timed out.*field 2"):
        env.set_state(env.get_state()["state"], 666.0, False)
    assert 0.25 < time.time() - t0 < 30
    if torch.cuda.is_available():
        torch.cuda.synchronize()                               # the GPU is still there
    st = eng.peer_status()
    assert st & 1 and (st >> 16) == eng.HALO_LOSS_MASS
    with pytest.raises(RuntimeError, match="earlier arrival timed out"):       # and the engine refuses to go on exchanging
        eng.halo_peer_exchange(eng.HALO_GRID_IN, 0)


def test_fused_exchange_wait_is_bounded_too(monkeypatch):
    """The same for the exchange FOLDED INTO the grid kernels (PLMPM_PEER_FUSED=1), where every workgroup of the launch waits: the
    loss-mass exchange is answered (loop-back), the substep fields' neighbours are silent -- the first forward substep's grid kernel
    gives up after PLMPM_PEER_TIMEOUT, every workgroup is released by the poller's tag, the rest of the env step drains without
    waiting again, and the step raises."""
    import sys
    import time
    import torch
    import bench
    sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))
    import slab_host_cost as shc
    from plasticinelab_amd import distributed as D

    class HalfSilent(shc.LoopbackComm):
        def setup_peer(self, engine):
            faces = self.layout.faces(self.rank)
            for field in (engine.HALO_GRID_IN, engine.HALO_GRID_OUT_ADJ, engine.HALO_LOSS_MASS):
                local = [engine.peer_alloc(field, a, b)[0] for _n, a, b in faces]
                remote = local if field == engine.HALO_LOSS_MASS else [engine.peer_alloc(field, a, b)[0] for _n, a, b in faces]
                engine.halo_peer_setup(field, [(a, b) for _n, a, b in faces], local, remote)
            self.peer_ready = True
            return True

    monkeypatch.setenv("PLMPM_PEER_TIMEOUT", "0.3")
    monkeypatch.setenv("PLMPM_PEER_FUSED", "1")
    cfg = bench.workload_cfg(20_000, 1, max_steps=40)
    layout = D.SlabLayout(64, (0, 16, 36, 64))
    env, _, _ = D.make_slab_env(cfg, 1, 3, compute_dtype="float32", target_fn=bench._target, layout=layout,
                                comm=HalfSilent(layout, 1, True), xy_margin=8, migrate_every=0)
    eng = env.simulator.engine
    assert eng.native_loops and eng.peer_fused() and eng.peer_status() == 0
    env.set_state(env.get_state()["state"], 666.0, False)            # the loss-mass exchange of the initial loss is answered
    assert eng.peer_status() == 0
    t0 = time.time()
    with pytest.raises(RuntimeError, match="timed out.*field 0"):
        env.step(bench.seeded_actions(1, 6)[0])
        eng._check()
    assert 0.25 < time.time() - t0 < 30                              # one timeout, not one per substep
    if torch.cuda.is_available():
        torch.cuda.synchronize()                                      # the GPU is still there
    st = eng.peer_status()
    assert st & 1 and (st >> 16) == eng.HALO_GRID_IN


@pytest.mark.parametrize("peer", [False, True])
def test_a_rank_without_particles_steps_and_differentiates(peer):
    """Migration can leave a rank without a single row (the body moved out of its slab): every particle launch is then of size zero
    and must be skipped, the grid kernels still run over the exchanged planes, state / gradient I/O returns empty arrays, and a
    whole env step forward + backward goes through.  One process, middle rank of three, loop-back exchange; re-entered with an
    EMPTY population (plmpm_set_population(0)), host-driven and native substep loops."""
    import sys
    import bench
    sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))
    import slab_host_cost as shc
    from plasticinelab_amd import distributed as D
    from plasticinelab_amd.engine.taichi_env import Tape
    cfg = bench.workload_cfg(20_000, 1, max_steps=40)
    layout = D.SlabLayout(64, (0, 16, 36, 64))
    env, _, _ = D.make_slab_env(cfg, 1, 3, compute_dtype="float32", target_fn=bench._target, layout=layout,
                                comm=shc.LoopbackComm(layout, 1, peer), xy_margin=8, migrate_every=0)
    eng = env.simulator.engine
    assert eng.native_loops == peer
    empty = dict(ids=np.zeros(0, np.int32), x=np.zeros((0, 3)), v=np.zeros((0, 3)), F=np.zeros((0, 3, 3)), C=np.zeros((0, 3, 3)),
                 mu=np.zeros(0), lam=np.zeros(0), ys=np.zeros(0), since=0)
    eng.reenter(empty)
    env.simulator.n_particles = env.n_particles = 0
    env.simulator.cur = 0
    env._is_copy = False                                               # tape mode (what set_state(..., is_copy=False) sets)
    env.loss.reset(); env.loss.clear()
    assert eng.frame_info(0)[0] == 0
    with Tape(env):
        env.step(bench.seeded_actions(1, 6)[0])
        env.compute_loss()
        assert eng.debug_contact(seed=3) == 3                          # stale entries in front of the reverse sweep (see the end)
    eng._check()
    f = env.simulator.cur
    assert f == env.simulator.substeps and eng.frame_info(f)[0] == 0
    fr = eng.get_frame(f)
    assert fr["x"].shape == (0, 3) and fr["F"].shape == (0, 3, 3)
    assert eng.get_frame_grad(0)["x"].shape == (0, 3)
    assert eng.grid_stats(0)[0] == 0                                   # nothing was scattered
    g = np.asarray(env.primitives.get_grad(1))
    assert g.shape == (1, 6) and np.isfinite(g).all() and np.isfinite(env.loss.loss)
    # no mass anywhere: this rank's density term is the target's own mass on the nodes it owns (z in [16, 36))
    want = float(np.abs(env.loss.target_density[:, :, 16:36]).sum())
    assert want > 0 and abs(env.loss.density_loss - want) < 1e-5 * want
    # The list of blocks whose pose adjoints are due is reset by the first particle workgroup of g2p.grad -- a launch this rank
    # skips.  Its grid kernel still lists blocks of the exchanged planes in which a neighbour's mass touches a manipulator, so the
    # reset must not depend on the particle launch: entries planted before a reverse substep are gone after it.
    # (three entries were planted before the reverse sweep, above)
    assert eng.debug_contact() == 0


def test_config5_rank_fits_in_hbm():
    """BASELINE configs[4]: 512^3 grid, 16M particles in a cube of side 0.25, 8 z-slabs.  What one rank has to allocate
    for a whole env step (159 substeps) in store mode, as plmpm_workspace_bytes reports it -- nothing is allocated
    here: < 200 GB (a dense per-frame store would need 4.3 GB x 159 per rank)."""
    from plasticinelab_amd.distributed import SlabLayout, slab_window
    from plasticinelab_amd.engine.core import Engine
    n, N, world, margin = 512, 16_000_000, 8, 24
    x = (np.random.default_rng(0).random((N, 3)) - 0.5) * 0.25 + np.array([0.5, 0.2, 0.5])
    lay = SlabLayout.balanced(x, n, world)
    owner = lay.owner_of(SlabLayout.stencil_base_z(x, n))
    worst = 0
    for rank in (0, 3, 7):
        mine = int((owner == rank).sum())
        lo, hi = slab_window(x, n, lay, rank, margin)
        eng = Engine(n_grid=n, n_particles=mine, max_frames=160, substeps=159, dt=0.5e-4 / 4, p_vol=(0.5 / n) ** 2, p_mass=(0.5 / n) ** 2,
                     gravity=(0, -1, 0), ground_friction=1.5, primitives=[], dtype="float32", slab=lay.slab(rank), slab_halo=4,
                     store_grid=True, grid_window=(lo, hi), particle_capacity=int(1.5 * mine), allocate=False)
        total = sum(eng.workspace_bytes.values())
        eng.close()
        worst = max(worst, total)
        assert 1.5e6 < mine < 2.6e6
    print(f"\n[512^3 / 8 slabs] largest rank workspace for a 159-substep env step: {worst / 2**30:.1f} GiB")
    assert worst < 200e9


def test_nccl_backend_two_gpus(tmp_path):
    """The same run over RCCL, one GPU per rank; needs two devices (skips on the single-GPU test boxes)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    g = np.load(os.path.join(GOLDEN, "rollout_small.npz"))
    res = launch(tmp_path, 2, "float64", g["actions"], 8, 1, backend="nccl")
    for r in res:
        assert abs(float(r["loss"]) - float(g["loss"])) / abs(float(g["loss"])) < 1e-10
        assert relerr(r["grad"], g["grad"]) < 1e-7


def test_nccl_backend_two_gpus_device_side_exchange(tmp_path):
    """... and with the halos written by the exchange kernel into the OTHER GPU's receive areas (xGMI peer writes, uncached
    areas, system-scope loads on the reading side): the configuration no single-GPU box can exercise."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    g = np.load(os.path.join(GOLDEN, "rollout_small.npz"))
    res = launch(tmp_path, 2, "float64", g["actions"], 8, 1, backend="nccl", peer=True)
    for r in res:
        assert abs(float(r["loss"]) - float(g["loss"])) / abs(float(g["loss"])) < 1e-10
        assert relerr(r["grad"], g["grad"]) < 1e-7


def test_leaving_the_grid_window_or_the_slab_raises():
    """A stencil outside the allocated grid window, or outside slab + halo in z, sets the error word
    (Engine.check_error raises) and is clamped into the window (no out-of-bounds access); inside both it does not."""
    from plasticinelab_amd._lib import EngineError
    from plasticinelab_amd.engine.core import Engine
    from tests import emul
    from tests.util import O, oracle_scene
    from tests.gpu_util import load_state
    _, sim, prims, x0 = oracle_scene("Move", 1, n_particles=1500)
    n = sim.n_grid
    b = (x0 * n - 0.5).astype(np.int64)
    lo, hi = b.min(0), b.max(0) + 3                     # node box of all stencils
    plist = [dict(shape=p.shape, action_dim=p.action_dim, params=emul.prim_par(p), friction=p.friction,
                  action_scale=p.action_scale, lower_bound=p.lower_bound, upper_bound=p.upper_bound) for p in prims]

    def engine(slab, window):
        eng = Engine(n_grid=n, n_particles=sim.n_particles, max_frames=sim.substeps + 1, substeps=sim.substeps, dt=sim.dt,
                     p_vol=sim.p_vol, p_mass=sim.p_mass, gravity=sim.gravity, ground_friction=sim.ground_friction,
                     primitives=plist, dtype="float32", slab=slab, slab_halo=4 if slab else 0, store_grid=True, grid_window=window)
        load_state(eng, 0, O.init_state(x0), O.materials(sim), O.init_poses(prims))
        eng.set_action(0, sim.substeps, np.zeros(sum(p.action_dim for p in prims)))
        eng.fk(0, 1)
        eng.p2g(0)
        return eng

    whole = ([0, 0, 0], [n, n, n])
    snug = ([int(v) - 2 for v in lo], [int(v) + 2 for v in hi])
    assert engine(None, None).error_flags() == 0
    assert engine(None, snug).error_flags() == 0
    o, blocks = engine(None, snug).grid_window()
    assert (o % 4 == 0).all() and (o <= np.array(snug[0])).all() and (o + 4 * blocks >= np.array(snug[1])).all()
    cut_x = ([int(lo[0]) + 8, 0, 0], [n, n, n])
    assert engine(None, cut_x).error_flags() & 1                         # the window cuts the body in x
    cut_y = ([0, 0, 0], [n, int(hi[1]) - 8, n])
    assert engine(None, cut_y).error_flags() & 1
    assert engine((int(lo[2]) // 4 * 4, n), whole).error_flags() == 0    # slab holds every stencil centre's reach
    with pytest.raises(EngineError, match="slab"):
        engine(((int(lo[2]) + 12) // 4 * 4, n), whole).check_error()     # slab + halo 4 misses the body's lower layers
