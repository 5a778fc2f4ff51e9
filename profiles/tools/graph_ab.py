#!/usr/bin/env python
"""Launch boundaries on the stream against one hipGraph per env step, on identical work.

Runs the headline rollout to the reverse sweep of env step S (Tape hook: the adjoint of the step's last frame + 1 is
resident, all its grids are stored), then times the step's 39 forward substeps (77 launches) and its 39 reverse substeps
(117 launches) through plmpm_replay_step: `reps` eager repetitions against `reps` replays of one captured graph,
alternating three times.  Microseconds per fwd / bwd substep."""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class Stop(Exception):
    pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--step", type=int, default=10)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--particles", type=int, default=500_000)
    ap.add_argument("--side", type=float, default=0.31, help="62 500 particles at the headline's density: --particles 62500 --side 0.155")
    a = ap.parse_args()
    args = argparse.Namespace(steps=a.step + 2, warmup=0, quality=2, particles=a.particles, dtype="float32", workload="config3_cube128",
                              yield_stress=200.0, side=a.side, window=-1, deterministic=False)
    env, _ = bench.build_env(args, torch.device("cuda", 0))
    sim = env.simulator
    eng = sim.engine
    state0 = env.get_state()["state"]
    env.set_state(state0, 666.0, False)
    acts = bench.seeded_actions(args.steps, env.primitives.action_dim)
    sub = sim.substeps

    def replay(graph, d, first):
        us = C.c_double(0)
        rc = eng.lib.plmpm_replay_step(eng.h, graph, d, first, sub, a.reps, C.byref(us))
        if rc != 0:
            raise RuntimeError(eng.lib.plmpm_last_error().decode())
        return us.value / sub

    def hook(step, first_frame):
        if step != a.step + 1:
            return
        first = first_frame - sub
        out = {"bwd eager": [], "bwd graph": [], "fwd eager": [], "fwd graph": []}
        side = torch.cuda.Stream()                      # a stream of its own: the legacy default stream cannot be captured
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            eng.use_current_stream()
            for _ in range(3):
                out["bwd eager"].append(replay(0, 1, first))
                out["bwd graph"].append(replay(1, 1, first))
            for _ in range(3):
                out["fwd eager"].append(replay(0, 0, first))
                out["fwd graph"].append(replay(1, 0, first))
        print(f"# env step {a.step}: us per substep, {a.reps} repetitions of the step's {sub} substeps per number")
        for k, v in out.items():
            print(f"{k}: " + " / ".join(f"{x:.2f}" for x in v))
        e = sum(out["bwd eager"]) / 3 + sum(out["fwd eager"]) / 3
        g = sum(out["bwd graph"]) / 3 + sum(out["fwd graph"]) / 3
        print(f"fwd + bwd substep: eager {e:.1f} us, graph {g:.1f} us ({100 * (e - g) / e:.1f} % less)")
        raise Stop()

    from plasticinelab_amd.engine.taichi_env import Tape
    try:
        with Tape(env, after_step_grad=hook):
            for act in acts:
                env.step(act)
                env.compute_loss()
    except Stop:
        pass


if __name__ == "__main__":
    main()
