"""Soak of the device-side halo exchange (not collected by pytest; run by hand on a GPU box: python tests/soak_peer_exchange.py 50 [fused]):
long rollouts with 2, 3 and 4 ranks sharing the GPU against one rank -- a store lost once in a few thousand exchanges (what an
earlier version of the exchange kernel did) shows up as 1e-5 in the gradient.  Output of the final build: profiles/r03_peer_exchange_soak.txt
On a box with at least as many GPUs as ranks every rank takes its OWN GPU (backend nccl = RCCL for the per-env-step collectives, the
halos through xGMI peer writes): the case the single-GPU boxes cannot show."""
import sys, os, tempfile, pathlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_distributed import launch, single_rank, gather
from tests.gpu_util import relerr
H = int(sys.argv[1]) if len(sys.argv) > 1 else 40
FUSED = len(sys.argv) > 2 and sys.argv[2] == "fused"          # the exchange folded into the grid kernels (PLMPM_PEER_FUSED=1)
acts = np.zeros((H, 6))
acts[:, 2] = 0.4; acts[:, 5] = 0.4
acts[:, 0] = 0.3; acts[:, 3] = -0.3
acts += np.random.default_rng(7).uniform(-0.2, 0.2, acts.shape)
import torch
loss, grad, x1, v1 = single_rank(acts, "float64")
for world in (2, 3, 4):
    backend = "nccl" if torch.cuda.device_count() >= world else "gloo"      # one GPU per rank when the box has them
    for rep in range(2):
        with tempfile.TemporaryDirectory() as d:
            res = launch(pathlib.Path(d), world, "float64", acts, 10, 1, peer=True, backend=backend, fused=FUSED)
        worst_l = max(abs(float(r["loss"]) - loss) / abs(loss) for r in res)
        worst_g = max(relerr(r["grad"], grad) for r in res)
        x, v = gather(res, 2000)
        assert all(int(r["fused"]) == int(FUSED) for r in res)
        print(f"world {world} ({backend}, {'fused grid kernels' if FUSED else 'exchange kernels'}, {'one GPU per rank' if backend == 'nccl' else 'ranks share cuda:0'}) rep {rep}: {H} env steps = {H * 19 * 2} exchanges per face; loss rel {worst_l:.2e}, grad rel {worst_g:.2e}, x rel {relerr(x, x1):.2e}, rows moved {[int(r['rows_moved']) for r in res]}", flush=True)
        assert worst_l < 1e-9 and worst_g < 1e-7
print("soak ok")
