"""-m gpu: loss terms, their adjoint and the target-SDF preprocess against the oracle."""
import ctypes

import numpy as np
import pytest
import torch

from tests.util import O, oracle_scene, sparse_target
from tests.gpu_util import engine_for, load_state, preroll, relerr

pytestmark = pytest.mark.gpu


def c_sdf(lib, tgt, dx):
    n = tgt.shape[0]
    sdf, npn = np.empty((n, n, n)), np.empty((n, n, n, 3))
    lib.plb_oracle_target_sdf.restype = ctypes.c_int
    lib.plb_oracle_target_sdf(np.ascontiguousarray(tgt).ctypes.data_as(ctypes.c_void_p), ctypes.c_int(n),
                              ctypes.c_double(dx), ctypes.c_double(1000.0), ctypes.c_int(2 * n),
                              sdf.ctypes.data_as(ctypes.c_void_p), npn.ctypes.data_as(ctypes.c_void_p))
    return sdf


@pytest.mark.parametrize("soft", [False, True])
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_loss_and_grad(oracle_c, dtype, soft):
    cfg, sim, prims, x0 = oracle_scene("Move", 1, n_particles=2000)
    acts = np.zeros((1, 6)); acts[:, 0] = 0.5; acts[:, 3] = -0.2
    state, mats, poses = preroll(sim, prims, x0, acts)
    tgt = sparse_target("Move3D-v1")
    eng = engine_for(sim, prims, dtype=dtype)
    load_state(eng, 0, state, mats, poses)
    eng.loss_set_target(tgt)
    sdf = eng.target_sdf()
    ref_sdf = c_sdf(oracle_c, tgt, sim.dx)
    assert relerr(sdf, ref_sdf) < (1e-14 if dtype == "float64" else 1e-6)     # bit-exact in f64
    eng.loss_set_weights(10, 10, 1, soft)
    out = eng.loss_forward(0)

    x = state[0].clone().requires_grad_(True)
    pin = [(p.clone().requires_grad_(True), r.clone().requires_grad_(True)) for p, r in poses]
    L, parts = O.compute_loss(sim, O.LossCfg(soft_contact=soft), prims, x, pin,
                              torch.as_tensor(tgt.reshape(-1)), torch.as_tensor(ref_sdf.reshape(-1)))
    tol = 1e-10 if dtype == "float64" else 2e-5
    assert abs(out["loss"] - float(L)) / abs(float(L)) < tol
    for k in ("sdf_loss", "density_loss", "contact_loss"):
        ref = float(parts[k])
        assert abs(out[k] - ref) <= tol * max(abs(ref), 1e-12) + (1e-14 if dtype == "float64" else 1e-9), k
    gm = parts["grid_m"].detach()
    assert abs(out["iou"] - float(O.iou(gm, torch.as_tensor(tgt.reshape(-1))))) < (1e-10 if dtype == "float64" else 1e-5)

    gs = torch.autograd.grad(L, [x] + [p for p, _ in pin], allow_unused=True)
    eng.grad_begin(0)
    eng.loss_backward(0)
    ga = eng.get_frame_grad(0)
    assert relerr(ga["x"], gs[0].numpy()) < (1e-9 if dtype == "float64" else 5e-5)
    for k in range(len(prims)):
        ref = np.zeros(3) if gs[1 + k] is None else gs[1 + k].numpy()
        got = eng.get_primitive_grad(k, 0)[:3]
        assert np.abs(got - ref).max() <= (1e-9 if dtype == "float64" else 5e-5) * max(np.abs(ref).max(), 1e-12) + 1e-12
    eng.close()
