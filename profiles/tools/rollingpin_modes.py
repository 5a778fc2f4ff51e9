#!/usr/bin/env python
"""fp32 `rollingpin` case of tests/test_gpu_shapes.py: action-gradient error against the float64 oracle, run to run, for the
floating-point-atomics engine and for the deterministic engine (integer-limb sums: no arrival-order dependence)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.util import GOLDEN, sparse_target  # noqa: E402
from tests.gpu_util import relerr  # noqa: E402
from tests.shape_cases import TARGET, case_cfg, subsample  # noqa: E402


def run(name, dtype, deterministic):
    from plasticinelab_amd.engine import taichi_env as te
    from plasticinelab_amd.optimizer.solver import Solver

    class Sub(te.Shapes):
        def get(self):
            x, c = super().get()
            k = len(x) // len(subsample(x))
            return subsample(x), c[::k][:len(subsample(x))]

    cfg, soft, acts = case_cfg(name)
    if deterministic:
        cfg.SIMULATOR["deterministic"] = True
    orig, te.Shapes = te.Shapes, Sub
    try:
        env = te.TaichiEnv(cfg, compute_dtype=dtype)
    finally:
        te.Shapes = orig
    env.initialize()
    env.loss.load_target_density(grids=sparse_target(TARGET))
    env.loss.set_weights(10, 10, 1, soft)
    state0 = env.get_state()["state"]
    return Solver(env, None, None, softness=666.0, horizon=len(acts)).forward(state0, acts)


g = np.load(os.path.join(GOLDEN, "rollout_shapes.npz"))
name = sys.argv[1] if len(sys.argv) > 1 else "rollingpin"
ref = g[f"{name}_grad"]
l64, g64 = run(name, "float64", False)
print(f"f64 engine vs oracle: {relerr(g64, ref):.2e}")
for det in (False, True):
    for r in range(4):
        loss, grad = run(name, "float32", det)
        print(f"{name} fp32 deterministic={det} run {r}: vs oracle {relerr(grad, ref):.2e}  vs f64 engine {relerr(grad, g64):.2e}  per component vs oracle "
              f"{np.abs(grad - ref).max(0) / np.abs(ref).max(0)}", flush=True)
