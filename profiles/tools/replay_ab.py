#!/usr/bin/env python
"""Kernel A/B on bit-identical inputs:  replay_ab.py [--step S] [--reps R] [--yield-stress Y] NAME ...

Runs the headline rollout with the in-tree library, stops the reverse sweep behind env step S + 1 (Tape hook), runs the
reverse substep of the last frame f of env step S properly, and then -- with the adjoint of frame f + 1, the grids of
frames f - 1 / f and the node adjoints of frame f all resident and real -- replays the three particle kernels through
every named build of the library (exp_libs/libplmpm_NAME.so, loaded into the same process, called on the SAME engine
handle through plmpm_replay): fused forward kernel of frame f, g2p.grad of frame f - 1, p2g.grad of frame f.
Prints mean microseconds per launch (HIP events, `reps` launches behind two untimed ones), arms alternating twice."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class Stop(Exception):
    pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--step", type=int, default=10)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--yield-stress", type=float, default=200.0)
    ap.add_argument("--particles", type=int, default=500_000)
    ap.add_argument("--quality", type=float, default=2)
    ap.add_argument("--window", type=int, default=-1)
    ap.add_argument("names", nargs="+")
    a = ap.parse_args()
    args = argparse.Namespace(steps=a.steps, warmup=0, quality=a.quality, particles=a.particles, dtype="float32", workload="config3_cube128",
                              yield_stress=a.yield_stress, side=0.31, window=a.window, deterministic=False)
    env, _ = bench.build_env(args, torch.device("cuda", 0))
    sim = env.simulator
    eng = sim.engine
    state0 = env.get_state()["state"]
    env.set_state(state0, 666.0, False)
    acts = bench.seeded_actions(a.steps, env.primitives.action_dim)
    libs = {}
    for n in a.names:
        path = os.path.join(ROOT, "plasticinelab_amd", "libplmpm.so") if n == "default" else os.path.join(ROOT, "exp_libs", f"libplmpm_{n}.so")
        lib = C.CDLL(path)
        lib.plmpm_replay.restype = C.c_int
        lib.plmpm_replay.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        lib.plmpm_last_error.restype = C.c_char_p
        libs[n] = lib
    sub = sim.substeps
    out = {}

    def replay(lib, kind, f):
        us = C.c_double(0)
        if lib.plmpm_replay(eng.h, kind, f, a.reps, C.byref(us)) != 0:
            raise RuntimeError(lib.plmpm_last_error().decode())
        return us.value

    def hook(step, first_frame):
        if step != a.step + 1:
            return
        f = first_frame - 1                       # last frame of env step a.step; adjoint of frame f + 1 is resident
        eng.substep_grad(f)                       # real g2p.grad / grid_op.grad / p2g.grad of frame f
        torch.cuda.synchronize()
        for rnd in range(2):
            for n, lib in libs.items():
                r = out.setdefault(n, {"g2p_p2g": [], "g2p_grad": [], "p2g_grad": [], "p2g": []})
                r["g2p_grad"].append(replay(lib, 1, f - 1))
                r["p2g_grad"].append(replay(lib, 2, f))
        for rnd in range(2):
            for n, lib in libs.items():
                out[n]["g2p_p2g"].append(replay(lib, 0, f))
                out[n]["p2g"].append(replay(lib, 3, f))
        raise Stop()

    from plasticinelab_amd.engine.taichi_env import Tape
    try:
        with Tape(env, after_step_grad=hook):
            for act in acts:
                env.step(act)
                env.compute_loss()
    except Stop:
        pass
    print(f"# replay at env step {a.step} (frame {(a.step + 1) * sub - 1}), sigma_y {a.yield_stress:g}, {a.reps} launches per number, us per launch")
    for n, r in out.items():
        print(f"{n:14s} " + "  ".join(f"{k} {' / '.join(f'{v:.2f}' for v in vs)}" for k, vs in r.items()), flush=True)


if __name__ == "__main__":
    main()
