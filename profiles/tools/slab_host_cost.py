"""Host issue cost of the z-slab path, measured on ONE GPU with a loop-back exchange.

The slab engine is built for a middle rank of a 3-rank layout (two faces, like every interior rank of an N-GPU run)
and its communicator is replaced by one that copies each face's send buffer into its receive buffer -- same kernel
sequence and Python / ctypes work per substep as a real run, no RCCL.  The numbers printed are
  issue us/substep : host time to enqueue K env steps forward + reverse, stream not waited on
  wall  us/substep : the same with the final synchronize
so that `issue` is the floor any N >= 2 run pays per substep before communication latency.

    python profiles/tools/slab_host_cost.py [--steps K] [--particles N]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from plasticinelab_amd import distributed as D  # noqa: E402


class LoopbackComm(D.HaloComm):
    """No neighbours: the registered receive buffers stay zero (one memset per face and exchange stands in for the
    arrival of a message), so the physics is that of a body with free faces; rows that leave are dropped."""

    def __init__(self, layout, rank, peer=False):
        self.layout, self.rank, self.group = layout, rank, None
        self.want_peer, self.peer_ready = bool(peer), False
        self.stage_host = False
        self.backend = "loopback"
        # (host tensors where there is no GPU: the tests run this communicator on the CPU interpreter of the device source too)
        self.scalar_device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self._recv, self._ops = {}, {}
        self.down = rank - 1 if rank > 0 else None
        self.up = rank + 1 if rank < layout.world - 1 else None

    def setup_peer(self, engine):
        """--peer: the device-side exchange with this rank's own receive areas as the "neighbours'" (the kernel copies the
        planes, publishes the counter it then waits for): same launches and host work per substep as a real run."""
        if not self.want_peer:
            return False
        faces = self.layout.faces(self.rank)
        for field in (engine.HALO_GRID_IN, engine.HALO_GRID_OUT_ADJ, engine.HALO_LOSS_MASS):
            local = [engine.peer_alloc(field, a, b)[0] for _n, a, b in faces]
            engine.halo_peer_setup(field, [(a, b) for _n, a, b in faces], local, local)
        self.peer_ready = True
        return True

    def exchange_start(self, engine, field, f):
        if not self.layout.faces(self.rank):
            return []
        if self.peer_ready:
            engine.halo_peer_exchange(field, f)
            return []
        if field not in self._recv:
            self.attach(engine, field, f)
        for rb in self._recv[field]:
            rb.zero_()
        return []

    def exchange_counts(self, n_down, n_up):
        return 0, 0

    def exchange_rows(self, send_down, send_up, n_recv_down, n_recv_up, width, device):
        return [None if n == 0 else torch.zeros(n * width, dtype=torch.float64, device=device) for n in (n_recv_down, n_recv_up)]

    def all_reduce_(self, t, op=None):
        return t

    def reduce_loss_record(self, rec, soft_contact, phase):
        return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--particles", type=int, default=500_000)
    ap.add_argument("--quality", type=float, default=2)
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--world", type=int, default=0, help="emulate the middle rank of a balanced N-slab cut (0: one rank owns all)")
    ap.add_argument("--xy-margin", type=int, default=24)
    ap.add_argument("--migrate-every", type=int, default=1)
    ap.add_argument("--overlap", action="store_true", help="interior grid blocks launched before the exchange is waited for")
    ap.add_argument("--profile", action="store_true", help="cProfile one rollout (top functions by own time)")
    ap.add_argument("--kernels", action="store_true", help="per-kernel HIP-event durations of one more rollout")
    ap.add_argument("--peer", action="store_true", help="device-side exchange (one kernel per exchange) + native substep loops")
    ap.add_argument("--grid-wg", type=int, default=0, help="persistent workgroups of the grid kernels (plmpm_config.grid_workgroups; 0 = the default)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    sub = int(2e-3 // (0.5e-4 / (args.quality * 0.5)))
    cfg = bench.workload_cfg(args.particles, args.quality, max_steps=max(args.steps, 1) * sub + 1)
    if args.grid_wg:
        cfg.SIMULATOR["grid_workgroups"] = args.grid_wg
    n = int(128 * args.quality * 0.5)
    # interior-rank geometry: rank 1 of 3 owns the whole body and has two faces just outside it
    layout = D.SlabLayout(n, (0, int(0.31 * n) // 4 * 4, (int(0.70 * n) + 3) // 4 * 4, n))
    rank, world = 1, 3
    if args.world > 1:                                   # the real cut bench.py would make, seen from a middle rank
        from plasticinelab_amd.engine.shapes import Shapes
        x_all, _ = Shapes(cfg.SHAPES).get()
        world, rank = args.world, args.world // 2
        layout = D.SlabLayout.balanced(x_all, n, world, None)      # reach 4, or 2 (one-plane slabs) when the body is too thin for that
    env, _, mine = D.make_slab_env(cfg, rank, world, compute_dtype=args.dtype, device=dev, target_fn=bench._target,
                                   layout=layout, comm=LoopbackComm(layout, rank, args.peer), xy_margin=args.xy_margin, migrate_every=args.migrate_every,
                                   overlap=args.overlap)
    print(f"native substep loops: {env.simulator.engine.native_loops}")
    print(f"rank {rank}/{world}: slab {layout.slab(rank)}, halo {layout.halo}, {len(mine)} particles, "
          f"grid window {[list(map(int, a)) for a in env.simulator.engine.grid_window()]}")
    env.loss.set_weights(10, 10, 1, False)
    sim = env.simulator
    state0 = env.get_state()["state"]
    acts = bench.seeded_actions(args.steps, env.primitives.action_dim)
    sub = sim.substeps
    for timed in (False, True, True):
        env.set_state(state0, 666.0, False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.rollout(env, acts)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if timed:
            k = args.steps * sub
            print(f"issue {1e6 * (t1 - t0) / k:7.1f} us/substep   wall {1e6 * (t2 - t0) / k:7.1f} us/substep   "
                  f"({k} fwd+bwd substeps)")
    if args.kernels:
        env.set_state(state0, 666.0, False)
        sim.engine.profile_enable(True)
        bench.rollout(env, acts)
        prof = sim.engine.profile_read()
        sim.engine.profile_enable(False)
        k = args.steps * sub
        print("kernels (HIP events), us per launch / us per fwd+bwd substep: " +
              ", ".join(f"{name} {1e3 * ms / cnt:.1f} / {1e3 * ms / k:.1f}" for name, (ms, cnt) in prof.items() if cnt))
        print(f"sum {sum(1e3 * ms / k for ms, cnt in prof.values()):.1f} us per fwd+bwd substep")
    if args.profile:
        import cProfile
        import pstats
        env.set_state(state0, 666.0, False)
        pr = cProfile.Profile()
        pr.enable()
        bench.rollout(env, acts)
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
