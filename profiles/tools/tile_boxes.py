"""Stencil-box sizes of the particle workgroups along the benchmark rollout (Engine.tile_boxes): how many exceed the
LDS tile (960 / 1024 nodes) and take the global-memory path.

    python profiles/tools/tile_boxes.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np, bench
class A: particles, quality, dtype, steps, warmup = 500000, 2, "float32", 4, 1
env,_ = bench.build_env(A, torch.device("cuda",0))
sim = env.simulator
st = env.get_state()["state"]
acts = bench.seeded_actions(4, env.primitives.action_dim)
env.set_state(st, 666.0, False)
for a in acts: env.step(a)
for f in (0, 1, 38, 39, 40, 100, 155):
    t = sim.engine.tile_boxes(f)
    nodes = t[:,3]*t[:,4]*t[:,5]
    print(f, len(t), "max", nodes.max(), "mean", nodes.mean().round(1), ">960:", (nodes>960).sum(), ">1024:", (nodes>1024).sum(), "pctl", np.percentile(nodes,[50,90,99]).round(0), "big at", np.nonzero(nodes>960)[0][:16])
