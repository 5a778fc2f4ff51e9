"""Per-kernel durations of the bench workload with the manipulators in contact ("near", the benchmark itself) or
moved away from the body ("far"): what the rigid-body contact code costs each kernel.

    python profiles/tools/contact_ablation.py near|far
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


class Args:
    particles, quality, dtype, steps, warmup = 500_000, 2, "float32", 4, 1


def main():
    far = len(sys.argv) > 1 and sys.argv[1] == "far"
    orig = bench.workload_cfg

    def workload_cfg(*a, **k):
        cfg = orig(*a, **k)
        if far:
            for p in cfg.PRIMITIVES:
                p["init_pos"] = (p["init_pos"][0], 0.8, p["init_pos"][2])
        return cfg

    bench.workload_cfg = workload_cfg
    env, _ = bench.build_env(Args, torch.device("cuda", 0))
    sim = env.simulator
    state0 = env.get_state()["state"]
    acts = bench.seeded_actions(Args.steps, env.primitives.action_dim)
    env.set_state(state0, 666.0, False)
    bench.rollout(env, acts)
    env.set_state(state0, 666.0, False)
    sim.engine.profile_enable(True)
    bench.rollout(env, acts)
    for k, (ms, cnt) in sim.engine.profile_read().items():
        if cnt:
            print(f"{'far ' if far else 'near'} {k:14s} {1e3 * ms / cnt:7.1f} us x {cnt}")


if __name__ == "__main__":
    main()
