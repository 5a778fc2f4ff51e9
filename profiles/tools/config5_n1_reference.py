#!/usr/bin/env python
"""Single-GPU reference record of BASELINE configs[4] as bench.py runs it at N = 8 (`secondary` point "mixed512_16000000p":
512^3 grid, 16M particles, sigma_y = 50 / 1e9 alternating, ONE env step of 159 substeps) -> profiles/n1_reference_points.json.

One GPU cannot hold the env step the way the 8 slab ranks do: 160 particle frames of 1.73 GB are 276 GB.  So
  * the LOSS (forward only) is computed in chunks of <= 40 substeps -- set_action(0, m, a * m / 159) gives the manipulators the
    same per-substep velocity as set_action(0, 159, a), the state of the last frame is copied to frame 0 (plmpm_copy_frame)
    and the next chunk starts there; the loss is evaluated on the final frame only, exactly what the one-step rollout of
    bench.py adds up;
  * the single-GPU RATE is substeps/s forward + backward over the first 40-substep window (median of 3), labelled as such:
    strong_scaling_eff of the 8-GPU point = its rate / (8 x this).

    python profiles/tools/config5_n1_reference.py [--scale 1.0] [--out profiles/n1_reference_points.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("PLMPM_RESORT_STEPS", "0")

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--chunk", type=int, default=40)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "n1_reference_points.json"))
    a = ap.parse_args()
    from plasticinelab_amd.engine.shapes import Shapes
    from plasticinelab_amd.engine.taichi_env import TaichiEnv
    pt = bench.point_config5(a.scale)
    n = int(128 * pt["quality"] * 0.5)
    sub = int(2e-3 // (0.5e-4 / (pt["quality"] * 0.5)))
    cfg = bench.workload_cfg(pt["particles"], pt["quality"], max_steps=a.chunk + 1, yield_stress=200.0, side=pt["side"])
    b = (Shapes(cfg.SHAPES).get()[0] * n - 0.5).astype(np.int64)
    cfg.SIMULATOR["grid_window"] = ([int(v) for v in np.maximum(b.min(0) - bench.XY_MARGIN, 0)],
                                    [int(v) for v in np.minimum(b.max(0) + 3 + bench.XY_MARGIN, n)])
    dev = torch.device("cuda", 0)
    env = TaichiEnv(cfg, compute_dtype="float32", device=dev)
    env.simulator._yield_stress = bench.mixed_yield(env.simulator.n_particles)
    env.initialize()
    env.loss.load_target_density(grids=bench._target(env.init_particles, env.simulator))
    env.loss.set_weights(10, 10, 1, False)
    sim, eng = env.simulator, env.simulator.engine
    assert sim.substeps == sub and sim.n_grid == n
    state0 = env.get_state()["state"]
    act = bench.seeded_actions(1, env.primitives.action_dim)[0]

    def chunk_fwd(m):
        env.primitives.set_action(0, m, act * m / sub)
        eng.step(0, m)

    # ---- rate: fwd + bwd over the first window
    m = min(a.chunk, sub)
    times = []
    for _ in range(4):
        env.set_state(state0, 666.0, False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        chunk_fwd(m)
        env.loss.clear_loss()
        env.loss.compute_loss_kernel(m)
        sim.grad_begin(m)
        env.loss.compute_loss_kernel_grad(m)
        eng.step_grad(0, m, 0)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    eng.check_error()
    rate = m / sorted(times[1:])[1]
    # ---- loss: the whole env step forward, in chunks
    env.set_state(state0, 666.0, False)
    done = 0
    while done < sub:
        m = min(a.chunk, sub - done)
        chunk_fwd(m)
        done += m
        if done < sub:
            eng.copy_frame(m, 0)
    eng.check_error()
    env.loss.clear_loss()
    loss = env.loss.compute_loss_kernel(m)["loss"]
    name = f"mixed{n}_{pt['particles']}p"
    rec = {"value": rate, "final_loss": float(loss),
           "source": f"profiles/tools/config5_n1_reference.py: one GPU, loss = forward of the {sub}-substep env step in chunks of {a.chunk} (frame copied "
                     f"back to 0 in between); value = substeps/s fwd + bwd over the first {min(a.chunk, sub)}-substep window, median of 3 "
                     "(160 frames of this workload are 276 GB: one GPU cannot hold the env step the way 8 slab ranks do)"}
    print(json.dumps({f"{name}|f32|1": rec}))
    try:
        with open(a.out) as f:
            d = json.load(f)
    except (OSError, ValueError):
        d = {}
    d[f"{name}|f32|1"] = rec
    with open(a.out, "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
