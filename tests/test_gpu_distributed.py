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
    ltol, gtol, xtol = (1e-10, 1e-7, 1e-10) if dtype == "float64" else (1e-5, 1e-3, 2e-5)
    n = int(g["n_particles"])
    assert sum(len(r["mine"]) for r in res) == n and min(len(r["mine"]) for r in res) > 0
    x = np.empty((n, 3)); v = np.empty((n, 3))
    for r in res:
        assert abs(float(r["loss"]) - float(g["loss"])) / abs(float(g["loss"])) < ltol      # every rank has the full loss
        assert relerr(r["grad"], g["grad"]) < gtol                                           # ... and the full gradient
        x[r["mine"]] = r["x"]; v[r["mine"]] = r["v"]
    assert relerr(x, g["x_final"]) < xtol
    assert relerr(v, g["v_final"]) < (1e-8 if dtype == "float64" else 2e-3)
