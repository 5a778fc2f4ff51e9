"""-m gpu: the real HIP kernels behind the z-slab SlabEngine, 2 ranks vs the oracle-generated golden rollout
(which the single-rank engine reproduces to 1e-10, test_gpu_rollout).  Both ranks share the box's one GPU and
exchange halos over gloo; on a multi-GPU node the same code runs over RCCL."""
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


# xy_margin: whole halo planes travel (None) or only the body's xy bounding box + that many node layers
@pytest.mark.parametrize("dtype,world,halo,xy_margin", [("float64", 2, 3, None), ("float32", 2, 3, 8), ("float64", 3, 2, 6)])
def test_slab_ranks_match_single_rank(tmp_path, dtype, world, halo, xy_margin):
    g = np.load(os.path.join(GOLDEN, "rollout_small.npz"))
    out = str(tmp_path / "r")
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), out, dtype, str(halo),
                                       "none" if xy_margin is None else str(xy_margin)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, lg in zip(procs, logs):
        assert p.returncode == 0, lg[-3000:]
    res = [np.load(f"{out}.{r}.npz") for r in range(world)]
    ltol, gtol, xtol = (1e-10, 1e-7, 1e-10) if dtype == "float64" else (1e-5, 1e-4, 2e-5)
    n = int(g["n_particles"])
    assert sum(len(r["mine"]) for r in res) == n and min(len(r["mine"]) for r in res) > 0
    x = np.empty((n, 3)); v = np.empty((n, 3))
    for r in res:
        assert abs(float(r["loss"]) - float(g["loss"])) / abs(float(g["loss"])) < ltol      # every rank has the full loss
        assert relerr(r["grad"], g["grad"]) < gtol                                           # ... and the full gradient
        x[r["mine"]] = r["x"]; v[r["mine"]] = r["v"]
    assert relerr(x, g["x_final"]) < xtol
    assert relerr(v, g["v_final"]) < (1e-8 if dtype == "float64" else 2e-3)


def test_leaving_the_slab_or_the_halo_window_raises():
    """Fixed ownership: a stencil outside slab + halo in z, or outside the exchanged xy window, sets the error word
    (Engine.check_error raises); inside both it does not."""
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

    def flags(slab, window):
        eng = Engine(n_grid=n, n_particles=sim.n_particles, max_frames=sim.substeps + 1, substeps=sim.substeps, dt=sim.dt,
                     p_vol=sim.p_vol, p_mass=sim.p_mass, gravity=sim.gravity, ground_friction=sim.ground_friction,
                     primitives=plist, dtype="float32", slab=slab, slab_halo=2, store_grid=True)
        if window is not None:
            eng.set_halo_window(*window)
        load_state(eng, 0, O.init_state(x0), O.materials(sim), O.init_poses(prims))
        eng.set_action(0, sim.substeps, np.zeros(sum(p.action_dim for p in prims)))
        eng.fk(0, 1)
        eng.p2g(0)
        return eng

    inside = (int(lo[2]), int(hi[2]))
    assert flags(inside, None).error_flags() == 0
    assert flags(inside, (max(int(lo[0]) - 2, 0), int(hi[0]) + 2, max(int(lo[1]) - 2, 0), int(hi[1]) + 2)).error_flags() == 0
    assert flags(inside, (int(lo[0]) + 3, int(hi[0]) + 2, 0, n)).error_flags() & 1         # x window cuts the body
    assert flags(inside, (0, n, max(int(lo[1]) - 2, 0), int(hi[1]) - 3)).error_flags() & 1         # y window cuts the body
    with pytest.raises(EngineError, match="z-slab"):
        flags((int(lo[2]) + 4, int(hi[2])), None).check_error()                            # slab + halo 2 misses 2 layers
