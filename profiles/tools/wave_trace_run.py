"""Per-wave phase trace of the three big particle kernels (needs a -DPLB_PHASE_TIMING build of libplmpm.so, passed
as EXP_LIB=...): runs the benchmark rollout and dumps the trace rows to gpurun_out/trace.npy."""
import os, sys, ctypes as C, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import plasticinelab_amd._lib as L
L.LIB_PATH = os.environ.get('EXP_LIB', L.LIB_PATH)      # a -DPLB_PHASE_TIMING build of libplmpm.so
import bench
class A: pass
args = A(); args.particles = int(os.environ.get('TRACE_PARTICLES', 500_000)); args.quality = 2; args.steps = 2; args.warmup = 1; args.dtype = 'float32'
args.side = float(os.environ.get('TRACE_SIDE', 0.31))        # 62 500 particles at the headline's density: TRACE_PARTICLES=62500 TRACE_SIDE=0.155
dev = torch.device('cuda:0')
env, _ = bench.build_env(args, dev)
sim = env.simulator
acts = bench.seeded_actions(2, env.primitives.action_dim)
state0 = env.get_state()["state"]
env.set_state(state0, 666.0, False)
bench.rollout(env, acts); torch.cuda.synchronize()
env.set_state(state0, 666.0, False)
lib = sim.engine.lib
bench.rollout(env, acts); torch.cuda.synchronize()
n = 3 * 16384 * 16
buf = (C.c_ulonglong * n)()
lib.plmpm_debug_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.plmpm_debug_trace(sim.engine.h, buf, n)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
np.save(os.path.join(ROOT, 'gpurun_out', 'trace.npy'), np.array(buf, dtype=np.uint64).reshape(3, 16384, 16))
