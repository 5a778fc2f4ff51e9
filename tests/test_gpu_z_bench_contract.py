"""-m gpu: bench.py as the driver runs it -- `python bench.py --gpus 1 ...` and `python -m torch.distributed.run
--nproc-per-node 2 ... bench.py --gpus 2 ...` -- prints ONE JSON line with the contract's keys, the N = 2 run is the SAME
workload cut into z-slabs ("strong", the loss equals the single-GPU run's), and its halos go through the device-side
exchange.  (Both ranks share the box's one GPU, so the per-env-step collectives use gloo: PLB_DIST_BACKEND.)  Reduced
particle count: this checks the contract, not the number."""
import json
import os
import socket
import subprocess
import sys

import pytest

from tests.util import ROOT

pytestmark = pytest.mark.gpu

KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline", "phases_s", "repeats", "value_min", "value_max", "repeat_ms_per_step", "secondary")
# a secondary point of the N >= 2 line: BASELINE configs[3] / configs[4] sizes cut into the same world's slabs
POINT_KEYS = ("label", "workload", "n_gpus", "n_grid", "n_particles", "steps", "value", "value_min", "value_max", "unit", "ms_per_step",
              "job_frac", "loss_check", "halo_transport", "strong_scaling_eff", "n1_value", "final_loss")
# the secondary points run at 3 % of their size here (PLB_SECONDARY_SCALE: same particles per cell) behind the reduced headline
SECONDARY = {"PLB_FORCE_SECONDARY": "1", "PLB_SECONDARY_SCALE": "0.03"}


def run(cmd, extra_env=None):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    return json.loads(lines[0])


def test_bench_lines_single_gpu_and_two_slabs():
    common = ["--steps", "2", "--warmup", "1", "--particles", "60000", "--no-cpu-baseline"]
    one = run([sys.executable, "bench.py", "--gpus", "1"] + common, SECONDARY)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    two = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "bench.py", "--gpus", "2"] + common,
              dict(SECONDARY, PLB_DIST_BACKEND="gloo", PLB_PEER_HALOS="1", PLB_SLAB_TIMEOUT="300"))
    for d, n in ((one, 1), (two, 2)):
        assert all(k in d for k in KEYS), [k for k in KEYS if k not in d]
        # the K-step rollout is timed `repeats` times; value / ms_per_step are the median repetition
        assert d["repeats"] == 5 and len(d["repeat_ms_per_step"]) == 5 and d["value_min"] <= d["value"] <= d["value_max"]
        assert abs(sorted(d["repeat_ms_per_step"])[2] - d["ms_per_step"]) < 1e-4
        assert d["n_gpus"] == n and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "substeps/s" and d["higher_is_better"] is True
        assert d["scaling"] == "strong" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
        assert "model" not in d["config"] and d["config"]["n_particles"] == 60000
        r = d["roofline"]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
        assert abs(d["value"] - d["steps"] * d["config"]["substeps_per_step"] / (1e-3 * d["ms_per_step"] * d["steps"])) < 1e-6 * d["value"]
    assert "FALLBACK" not in two["metric"] and "z-slabs" in two["config"]["parallelism"] and "IPC-mapped" in two["config"]["parallelism"]
    # the self-validation keys of the multi-GPU line: which transport carried the halos, that it was checked against the
    # library transport on one env step (fwd + bwd) before anything was timed, the communicator's size, the loss check
    tc = two["transport_check"]
    assert two["halo_transport"].startswith("peer-write") and tc["checked"] and tc["agree"] and tc["rel_loss"] < 1e-5 and tc["rel_grad"] < 1e-4
    assert two["dist_backend"] == "gloo" and two["rccl_ranks"] is None        # (RCCL reports its own rank count on a multi-GPU node)
    for d in (one, two):
        assert set(d["loss_check"]) >= {"n1_expected", "rel", "ok"}            # no committed N = 1 loss for this reduced workload: nulls
    assert "halo_exchange" in two["roofline"]["kernels"] and two["roofline"]["kernel"] != "halo_exchange"
    # the same workload: the two-slab run ends with the single-GPU run's loss (fp32 engines, different summation order)
    assert abs(two["final_loss"] - one["final_loss"]) < 1e-5 * abs(one["final_loss"])
    # N = 1: the config-4-size point and the float64 engine on the headline workload, in child processes behind the timed region
    s1, s64 = one["secondary"], one["secondary_f64"]
    assert "error" not in s1 and s1["n_grid"] == 256 and s1["steps"] == 2 and s1["value"] > 0 and 0 < s1["job_frac"] < 1, s1
    assert "error" not in s64 and s64["dtype"] == "f64" and s64["steps"] == 4 and s64["n_particles"] == 60000 and s64["value"] > 0, s64
    # N = 2: the SAME config-4-size workload cut into the world's two slabs, inside the same run, with its own self-checks
    assert isinstance(two["secondary"], list) and len(two["secondary"]) == 1
    p = two["secondary"][0]
    assert "error" not in p and all(k in p for k in POINT_KEYS), (p, [k for k in POINT_KEYS if k not in p])
    assert p["n_gpus"] == 2 and p["n_grid"] == 256 and p["workload"] == s1["workload"] and p["n_particles"] == s1["n_particles"]
    assert p["halo_transport"].startswith("peer-write") and p["value"] > 0 and 0 < p["job_frac"] < 1
    assert set(p["loss_check"]) >= {"n1_expected", "rel", "ok"}               # (no committed N = 1 record at this reduced size: nulls)
    # ... and it IS the same rollout as the single-GPU child process ran: same loss
    assert abs(p["final_loss"] - s1["final_loss"]) < 1e-5 * abs(s1["final_loss"]), (p["final_loss"], s1["final_loss"])
    # "halo-overlapped substeps" (BASELINE configs[4]): the same engine timed once more with the exchange folded into the grid
    # kernels -- the interior blocks' grid work hides the arrival -- and loss-checked like the default form
    ho = p["halo_overlapped"]
    assert "error" not in ho and ho["value"] > 0 and "folded into the grid kernels" in ho["halo_transport"], ho
    assert abs(ho["final_loss"] - s1["final_loss"]) < 1e-5 * abs(s1["final_loss"]) and set(ho["loss_check"]) >= {"n1_expected", "rel", "ok"}
    # the N = 1 points of the series measured in the SAME run on the same box: rank 0's child processes time the
    # headline and the configs[3]-size point on one GPU; the efficiencies against them stand beside the committed-reference ones
    nb = two["n1_same_box"]
    assert "error" not in nb["headline"] and nb["headline"]["workload"] == two["config"]["workload"] and nb["headline"]["steps"] == two["steps"]
    assert abs(nb["headline"]["final_loss"] - two["final_loss"]) < 1e-5 * abs(two["final_loss"])
    assert abs(two["strong_scaling_eff_same_box"] - two["value"] / (2 * nb["headline"]["value"])) < 1e-12
    assert "error" not in nb["configs[3]"] and nb["configs[3]"]["workload"] == p["workload"]
    assert abs(p["strong_scaling_eff_same_box"] - p["value"] / (2 * nb["configs[3]"]["value"])) < 1e-12 and p["n1_same_box_value"] == nb["configs[3]"]["value"]
    assert "n1_schedule" in p and p["grid_workgroups"] > 0                       # (built for the fused form too: capped, and the record says so)
    assert "n1_same_box" in two["phases_s"]
    # first contact: one record per rank -- device, peer access towards the neighbour, kind of receive-area
    # memory, IPC open, one hand-off of the exchange's own pattern across the face
    pf = tc["preflight"]
    assert len(pf) == 2 and [r["rank"] for r in pf] == [0, 1]
    for r in pf:
        nbr = str(1 - r["rank"])
        assert r["alloc"] in ("uncached", "fine-grained") and r["ipc_open"] == "ok", r
        assert r["neighbours"][nbr]["can_access_peer"] in ("same device", True, False), r
        assert r["ping"][nbr]["arrived"] is True and r["ping"][nbr]["wait_us"] >= 0, r


def test_a_spoiled_halo_flips_the_line():
    """The device-side exchange of rank 0 sends wrong values (test hook PLB_TEST_PEER_SPOIL): the transport check sees the
    loss / gradient of one env step disagree with the library transport, the run falls back to point-to-point halos, says so
    in `metric`, `halo_transport` and `config.parallelism` -- and still ends with the single-GPU loss."""
    common = ["--steps", "2", "--warmup", "1", "--particles", "60000", "--no-cpu-baseline", "--no-roofline"]
    one = run([sys.executable, "bench.py", "--gpus", "1"] + common)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    two = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "bench.py", "--gpus", "2"] + common,
              {"PLB_DIST_BACKEND": "gloo", "PLB_PEER_HALOS": "1", "PLB_SLAB_TIMEOUT": "300", "PLB_TEST_PEER_SPOIL": "0:1.5"})
    tc = two["transport_check"]
    assert tc["checked"] and not tc["agree"] and (tc["rel_loss"] > 1e-5 or tc["rel_grad"] > 1e-4)
    assert "FALLBACK" in two["metric"] and "FALLBACK" in two["config"]["parallelism"] and two["halo_transport"].startswith("gloo-p2p")
    assert two["scaling"] == "strong" and abs(two["final_loss"] - one["final_loss"]) < 1e-5 * abs(one["final_loss"])
