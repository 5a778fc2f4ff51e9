"""Worker for tests/test_gpu_distributed.py: one rank of a z-slab run on cuda:0 (all ranks share the single GPU of the
test box; halos and migrating rows go through gloo, staged via host memory -- the same SlabEngine code path that RCCL
drives on a multi-GPU node; PLB_DIST_BACKEND=nccl uses one GPU per rank instead).

    dist_gpu_worker.py OUT DTYPE ACTIONS.npy XY_MARGIN|none MIGRATE_EVERY [SCENE_JSON]

Without SCENE_JSON the scene is Move-v1 with a 2000-particle subsample; with it the synthetic cube of bench.py
({"particles", "quality", "side", "yield_stress"}: the BASELINE config-3/4/5 workloads at a chosen size).
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, dtype, act_path = sys.argv[1], sys.argv[2], sys.argv[3]
    xy_margin = None if sys.argv[4] == "none" else int(sys.argv[4])
    migrate_every = int(sys.argv[5])
    scene = json.loads(sys.argv[6]) if len(sys.argv) > 6 else None
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("PLB_DIST_BACKEND", "gloo")
    overlap = os.environ.get("PLB_TEST_OVERLAP") == "1"      # interior grid blocks while the halos are in flight
    halo = int(os.environ["PLB_TEST_HALO"]) if os.environ.get("PLB_TEST_HALO") else None     # 2: thin slabs (one block plane)
    peer = os.environ.get("PLB_TEST_PEER") == "1"            # device-side halo exchange (peer writes through IPC-mapped areas)
    # PLMPM_TEST_INTERPRETER=1 (tests/test_emul_tier.py): the ranks run the device source on the CPU interpreter instead of a GPU --
    # same SlabEngine, same gloo exchange of host-staged halos, no ROCm device anywhere
    interpreter = os.environ.get("PLMPM_TEST_INTERPRETER") == "1"
    if interpreter:
        from tests import emul_engine
        import plasticinelab_amd.engine.mpm_simulator as ms
        ms.Engine = emul_engine.HostEngine
        # (peer writes work here too: the receive areas are POSIX shared memory behind the shim's hipIpc calls; the exchange folded into
        # the grid kernels needs its workgroups resident at once: PLMPM_EMUL_THREADS > 1, one OS thread per grid workgroup)
        assert backend == "gloo", "the interpreter: gloo control plane"
        assert os.environ.get("PLMPM_PEER_FUSED", "0") in ("", "0") or int(os.environ.get("PLMPM_EMUL_THREADS", "1")) > 1, "fused exchange: set PLMPM_EMUL_THREADS"
    dev = rank % torch.cuda.device_count() if backend == "nccl" else 0
    if not interpreter:
        torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.util import sparse_target
    from plasticinelab_amd.distributed import make_slab_env
    from plasticinelab_amd.engine.shapes import Shapes
    from plasticinelab_amd.envs.scenes import load_scene
    from plasticinelab_amd.optimizer.solver import Solver

    actions = np.load(act_path)
    deterministic = os.environ.get("PLB_TEST_DETERMINISTIC") == "1"
    if scene is None:
        cfg = load_scene("Move", 1)
        cfg.ENV.loss.target_path = ""
        cfg.SIMULATOR["deterministic"] = deterministic
        x_all, _ = Shapes(cfg.SHAPES).get()
        n = 2000
        sub = np.ascontiguousarray(x_all[::len(x_all) // n][:n])
        env, layout, mine = make_slab_env(cfg, rank, world, compute_dtype=dtype, particles=sub, xy_margin=xy_margin, halo=halo, peer=peer,
                                          migrate_every=migrate_every, target_fn=lambda x, sim: sparse_target("Move3D-v1"), overlap=overlap)
    else:
        import bench
        sub_per_step = int(2e-3 // (0.5e-4 / (scene["quality"] * 0.5)))
        cfg = bench.workload_cfg(scene["particles"], scene["quality"], max_steps=len(actions) * sub_per_step + 1,
                                 yield_stress=scene.get("yield_stress", 200.0), side=scene.get("side", 0.31))
        cfg.SIMULATOR["deterministic"] = deterministic
        ys = None
        if scene.get("mixed"):                       # config 5: half the particles yield (50), half do not (1e9)
            ys = np.where(np.arange(scene["particles"]) % 2 == 0, 50.0, 1e9)
        env, layout, mine = make_slab_env(cfg, rank, world, compute_dtype=dtype, xy_margin=xy_margin, migrate_every=migrate_every, halo=halo, peer=peer,
                                          target_fn=bench._target, yield_stress=ys, overlap=overlap)
    if peer:
        comm = env.simulator.engine.comm
        assert env.simulator.engine.native_loops, f"peer halos were asked for and could not be set up: {getattr(comm, 'peer_error', '?')}"
    env.loss.set_weights(10, 10, 1, False)
    solver = Solver(env, None, None, softness=666.0, horizon=len(actions))
    state0 = env.get_state()["state"]
    segment = int(os.environ.get("PLB_TEST_SEGMENT") or 0)
    if segment:                                         # segment-checkpointed backward on the slab ranks (optimizer/checkpoint.py)
        from plasticinelab_amd.optimizer.checkpoint import forward_checkpointed
        loss, grad = forward_checkpointed(env, state0, actions, segment)
        loss2, grad2 = forward_checkpointed(env, state0, actions, segment)      # the engine is left ready for another call
        assert abs(loss2 - loss) <= 1e-9 * abs(loss) and np.abs(grad2 - grad).max() <= 1e-7 * np.abs(grad).max()
    else:
        loss, grad = solver.forward(state0, actions)
    sim = env.simulator
    eng = sim.engine
    ids, fr = eng.get_frame_by_id(sim.cur, want=("x", "v"))
    ws = eng.workspace_bytes
    np.savez(f"{out_path}.{rank}.npz", loss=loss, grad=grad, mine=mine, ids=ids, x=fr["x"], v=fr["v"], bounds=np.array(layout.bounds),
             migrations=eng.migrations, rows_moved=eng.rows_moved, window=np.concatenate(eng.grid_window()),
             grid_bytes=ws["grid_bytes"], total_bytes=sum(ws.values()), count=eng.frame_info(sim.cur)[0],
             native_loops=int(getattr(eng, "native_loops", False)),
             fused=int(bool(getattr(eng, "native_loops", False)) and eng.peer_fused()))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
