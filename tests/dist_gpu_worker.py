"""Worker for tests/test_gpu_distributed.py: one rank of a z-slab run of the Move scene (2000-particle subsample)
on cuda:0 (all ranks share the single GPU of the test box; halos go through gloo, staged via host memory -- the
same SlabEngine code path that RCCL drives on a multi-GPU node)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, dtype, halo = sys.argv[1], sys.argv[2], int(sys.argv[3])
    xy_margin = None if len(sys.argv) < 5 or sys.argv[4] == "none" else int(sys.argv[4])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.util import GOLDEN, sparse_target
    from plasticinelab_amd.distributed import make_slab_env
    from plasticinelab_amd.engine.shapes import Shapes
    from plasticinelab_amd.envs.scenes import load_scene
    from plasticinelab_amd.optimizer.solver import Solver

    g = np.load(os.path.join(GOLDEN, "rollout_small.npz"))
    cfg = load_scene("Move", 1)
    cfg.ENV.loss.target_path = ""
    x_all, _ = Shapes(cfg.SHAPES).get()
    n = int(g["n_particles"])
    sub = np.ascontiguousarray(x_all[::len(x_all) // n][:n])
    env, layout, mine = make_slab_env(cfg, rank, world, halo=halo, compute_dtype=dtype, particles=sub, xy_margin=xy_margin,
                                      target_fn=lambda x, sim: sparse_target("Move3D-v1"))
    env.loss.set_weights(10, 10, 1, False)
    solver = Solver(env, None, None, softness=666.0, horizon=len(g["actions"]))
    state0 = env.get_state()["state"]
    loss, grad = solver.forward(state0, g["actions"])
    sim = env.simulator
    fr = sim.engine.get_frame(sim.cur)
    np.savez(f"{out_path}.{rank}.npz", loss=loss, grad=grad, mine=mine, x=fr["x"], v=fr["v"], bounds=np.array(layout.bounds))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
