#!/usr/bin/env python
"""Benchmark: MPM substeps/sec (forward + backward) on the BASELINE.json workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--repeats R] [--dtype float32] [--no-cpu-baseline]

One "step" = one env step of the hot path: `substeps` x substep forward, the loss, and -- after the K
forward steps -- the K reverse steps (loss adjoint + `substeps` x substep_grad + primitive-chain adjoint),
exactly the call sequence of Solver.forward (plb/optimizer/solver.py:31-44).  The timed region starts with
the particle state already resident in HBM and covers K steps forward + K steps backward.  The K-step rollout is timed
R = 5 times (--repeats; the state is put back outside the timed spans, barrier + synchronize on both sides of each), `value`
and `ms_per_step` are the MEDIAN repetition, `value_min` / `value_max` / `repeat_ms_per_step` show the spread -- one 0.12 s
rollout on a fresh box varies by more than most kernel changes are worth.

Workload (config.workload "config3_cube128"): BASELINE.json configs[2] as synthesised in SURVEY.md 8(d):
128^3 grid (quality 2, 39 substeps / env step), one elastoplastic cube of side 0.31 at (0.5, 0.2, 0.5)
with 500 000 seed-0 uniform particles (~8 per cell), sigma_y = 200, two Sphere manipulators (r = 0.05)
touching opposite faces, seeded actions, own target grid (the cube's mass grid shifted by a few cells).

Prints ONE JSON line (rank 0).  N > 1: one process per GPU; the SAME total workload is cut into z-slabs
(plasticinelab_amd/distributed.py: windowed grids, zero-copy block-plane halo sum exchange over RCCL each substep
forward and adjoint, particle migration every env step) -> "scaling": "strong".  Slabs are whole 4-layer block planes:
two or more per rank while the body allows it (config 3's cube spans ten planes: N <= 5), one per rank beyond that
(N = 8: slabs of one or two planes, the plane of a one-plane slab exchanged with both neighbours).  If the slab path
cannot run at all (a particle leaving the grid window, an RCCL error, a hang caught by the watchdog) the bench falls
back to N independent replicas of the workload ("scaling": "weak") and says so in `metric` and in config.parallelism.

Extra points in the same line, measured AFTER the headline's timed region (they never touch it):
  N = 1  `secondary`      BASELINE configs[3]'s size on one GPU (256^3 / 2M elastic particles, 2 env steps), child process;
         `secondary_f64`  the float64 engine -- the reference's own precision (mpm_simulator.py:8) -- on the headline workload,
                          4 env steps, child process;
  N >= 2 `secondary`      a list, run inside the same world: configs[3] (256^3 / 2M, sigma_y = 1e9, 2 env steps) cut into N
                          z-slabs at N = 2, 4, 8, and configs[4] (512^3 / 16M, half sigma_y = 50 / half 1e9, ONE env step of 159
                          substeps with the per-frame grid store) at N = 8 -- the sizes SURVEY 8(e) says can scale -- each with
                          value, job_frac, loss_check against the committed single-GPU loss, halo_transport and
                          strong_scaling_eff against the committed single-GPU rate (profiles/n1_reference_points.json), and
                          `halo_overlapped`: the same engine with the exchange folded into the grid kernels (configs[4]'s
                          "halo-overlapped substeps"), timed and loss-checked the same way.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic scalars per particle (cN) and per active node (cA) of each kernel; the rows sum to the
# SURVEY.md 8(d) figure 150 N + 57 A per fwd+bwd substep (DESIGN.md "algorithmic bytes")
ALG = {
    "p2g": (36, 4), "grid_op": (0, 11), "g2p": (15, 3),
    "p2g_recompute": (27, 8), "grid_op_recompute": (0, 7), "g2p_grad": (18, 9),
    "grid_op_grad": (0, 11), "p2g_grad": (54, 4), "clear_active": (0, 0),
    "g2p_p2g": (51, 7),        # g2p(f-1) + p2g(f) fused
    # slab ranks with the device-side exchange folded into the grid kernels: the grid rows (the exchanged planes are not algorithmic bytes)
    "xchg+grid_op": (0, 11), "xchg+grid_op_grad": (0, 11),
}
HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); the copy / read rates this box reaches are measured live
PMC_FILE = os.path.join(ROOT, "profiles", "r05_pmc.json")


def pmc_traffic(kernel, workload, dtype, steps, warmup):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r05_pmc.json, written by
    profiles/tools/pmc_summary.py from separate --pmc FETCH_SIZE / WRITE_SIZE runs of this very command: 2 x
    FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md's HBM section).  None when the file is
    missing or was taken on another workload / dtype / --steps / --warmup -- a stale number is worse than none."""
    try:
        with open(PMC_FILE) as f:
            d = json.load(f)
        if d.get("workload") != workload or d.get("dtype") != dtype or d.get("steps") != steps or d.get("warmup") != warmup:
            return None, None
        k = d["kernels"].get(kernel)
        return (None, None) if k is None else (float(k["hbm_bytes_per_launch"]), d.get("source"))
    except (OSError, ValueError, KeyError):
        return None, None


VALU_FILE = os.path.join(ROOT, "profiles", "r06_valu_calibration.json")


def valu_roof(kernels, n_particles, workload, dtype, substeps_total):
    """SURVEY 8(d)'s secondary check: the vector-ALU issue time of a fwd+bwd substep, next to the HBM roof.  From the committed
    calibration (profiles/r06_valu_calibration.json, written by profiles/tools/valu_calibration.py): issue cycles per
    wave-instruction of every instruction class at the REAL shader clock (profiles/microbench/valu_calibration.hip: s_memtime
    against s_memrealtime) and the DYNAMIC instruction mix per wave of every hot kernel (rocprofv3 --pmc SQ_INSTS_VALU_* passes of
    this command).  issue_us of a kernel = waves per SIMD x sum(class count x class cycles) / clock -- the time the four SIMDs of
    every CU need to ISSUE the kernel's vector instructions if nothing else ever stalled; frac = that time over the measured
    kernel time: 1.0 would be a kernel bound by vector issue alone.  None when the file is missing or was taken on another
    workload / dtype."""
    try:
        with open(VALU_FILE) as f:
            cal = json.load(f)
    except (OSError, ValueError):
        return None
    if cal.get("workload") != workload or cal.get("dtype") != dtype:
        return None
    cyc, clock, simds = cal["cycles_per_wave_instruction"], float(cal["clock_ghz"]), int(cal.get("simds", 1024))
    waves = -(-int(n_particles) // 64)
    per, issue_total, time_total = {}, 0.0, 0.0
    for name, k in kernels.items():
        mix = cal["kernels"].get(name, {}).get("mix_per_wave")
        if not mix:
            continue
        w = cal["kernels"][name].get("waves") or waves              # grid kernels: the waves the PMC pass counted
        cycles = sum(float(n) * float(cyc.get(c, cyc["default"])) for c, n in mix.items())
        issue_us = (w / simds) * cycles / (clock * 1e3)
        per[name] = {"valu_per_wave": sum(mix.values()), "issue_cycles_per_wave": cycles, "issue_us": issue_us,
                     "frac_of_kernel_time": issue_us / k["avg_us"] if k["avg_us"] > 0 else None}
        issue_total += issue_us * k["launches"]
        time_total += k["avg_us"] * k["launches"]
    if not per:
        return None
    return {"issue_us_per_substep": issue_total / substeps_total, "frac": issue_total / time_total if time_total > 0 else None,
            "clock_ghz": clock, "cycles_per_instr_source": "profiles/" + os.path.basename(VALU_FILE) + ": " + str(cal.get("source", "")),
            "cycles_per_wave_instruction": cyc, "kernels": per}


def mixed_yield(n):
    """BASELINE configs[4]: half the particles plastic (sigma_y = 50), half elastic (1e9), alternating in caller order."""
    return np.where(np.arange(n) % 2 == 0, 50.0, 1e9)


def workload_cfg(n_particles=500_000, quality=2, max_steps=1024, yield_stress=200.0, side=0.31):
    from plasticinelab_amd.config import get_cfg_defaults
    cfg = get_cfg_defaults()
    r = 0.05
    cfg.merge({
        "SIMULATOR": {"quality": quality, "yield_stress": yield_stress, "E": 5000.0, "nu": 0.2, "max_steps": max_steps,
                      "n_particles": n_particles},
        "SHAPES": [{"shape": "box", "width": (side, side, side), "init_pos": (0.5, 0.2, 0.5), "n_particles": n_particles}],
        "PRIMITIVES": [
            {"shape": "Sphere", "radius": r, "init_pos": (0.5 - side / 2 - r, 0.2, 0.5), "friction": 0.9,
             "action": {"dim": 3, "scale": (0.01, 0.01, 0.01)}},
            {"shape": "Sphere", "radius": r, "init_pos": (0.5 + side / 2 + r, 0.2, 0.5), "friction": 0.9,
             "action": {"dim": 3, "scale": (0.01, 0.01, 0.01)}},
        ],
    }, strict=True)
    cfg["VARIANTS"] = None
    return cfg


def scene_cfg(name, n_particles=500_000, quality=2, max_steps=1024):
    """BASELINE configs[2] as it names the scenes: the reference's TripleMove-v1 (three boxes, six Sphere manipulators,
    plb/envs/triplemove.yml:3-64) or Rope-v1 (one bar, two Spheres + a static Cylinder, ground_friction 0.3,
    plb/envs/rope.yml:1-32) geometry on the 128^3 grid with the particle count raised to ~n_particles (split evenly over
    the scene's shapes; the reference samples 10k per shape)."""
    from plasticinelab_amd.envs.scenes import load_scene
    cfg = load_scene(name, 1)
    shapes = [dict(sh) for sh in cfg.SHAPES]
    per = -(-n_particles // len(shapes))
    for sh in shapes:
        sh["n_particles"] = per
    cfg.merge({"SIMULATOR": {"quality": quality, "max_steps": max_steps, "n_particles": per * len(shapes)}, "SHAPES": shapes}, strict=True)
    cfg.ENV.loss.target_path = ""
    cfg["VARIANTS"] = None
    return cfg


WORKLOADS = {"config3_cube128": None, "triplemove128": "TripleMove", "rope128": "Rope"}


def mass_grid(x, n, p_mass):
    """host copy of compute_grid_m_kernel (only used to synthesise the benchmark's own target grid): 27 weighted histograms
    over the bounding box of the stencils (np.bincount; the 16M-particle / 512^3 point would spend minutes in np.add.at)."""
    xs = x * n
    base = (xs - 0.5).astype(np.int64)
    fx = xs - base
    w = [0.5 * (1.5 - fx) ** 2, 0.75 - (fx - 1) ** 2, 0.5 * (fx - 0.5) ** 2]
    lo = base.min(0)
    ext = base.max(0) - lo + 3
    b = base - lo
    box = np.zeros(int(ext[0]) * int(ext[1]) * int(ext[2]))
    for i in range(3):
        for j in range(3):
            for k in range(3):
                idx = ((b[:, 0] + i) * ext[1] + b[:, 1] + j) * ext[2] + b[:, 2] + k
                box += np.bincount(idx, weights=w[i][:, 0] * w[j][:, 1] * w[k][:, 2] * p_mass, minlength=len(box))
    g = np.zeros((n, n, n))
    g[lo[0]:lo[0] + ext[0], lo[1]:lo[1] + ext[1], lo[2]:lo[2] + ext[2]] = box.reshape(ext)
    return g


def seeded_actions(K, A):
    a = np.random.default_rng(0).uniform(-1, 1, (K, A)) * 0.2
    a[:, 0] = 0.8          # left manipulator pushes +x into the cube
    a[:, 3] = -0.8         # right manipulator pushes -x
    return a


def _target(x_all, sim):
    """The workload's own target grid: the body's mass grid shifted by a few cells.  Scaled by (n / 128)^2 on finer grids: the
    reference marks a node as inside the target where target_density > 1e-4 (an absolute threshold, loss.py:86 "TODO: make it
    configurable"), and a node's mass at ~8 particles per cell is 1.2e-4 at 128^3 but 3e-5 at 256^3 -- unscaled, NO node of a
    256^3 / 512^3 target passes, the target SDF is `inf` (1000) everywhere and the loss is 1e4 x the body's mass plus noise,
    which would make the secondary points' loss_check a mass-conservation check only."""
    return mass_grid(np.clip(x_all + np.array([0.03, 0.0, 0.02]), 0.03, 0.97), sim.n_grid, sim.p_mass) * (sim.n_grid / 128.0) ** 2


XY_MARGIN = 24      # node layers around the body's bounding box that a rank's grid window covers (the rest is never touched)


def build_env(args, device, rank=0, world=1, slabs=False):
    from plasticinelab_amd.engine.taichi_env import TaichiEnv
    sub = int(2e-3 // (0.5e-4 / (args.quality * 0.5)))
    frames = max(args.steps, args.warmup, 1) * sub + 1
    scene = WORKLOADS.get(getattr(args, "workload", "config3_cube128"))
    if scene is not None:
        cfg = scene_cfg(scene, args.particles, args.quality, max_steps=frames)
    else:
        cfg = workload_cfg(args.particles, args.quality, max_steps=frames, yield_stress=getattr(args, "yield_stress", 200.0),
                           side=getattr(args, "side", 0.31))
    if getattr(args, "deterministic", False) and not slabs:
        cfg.SIMULATOR["deterministic"] = True
    if slabs:
        import torch.distributed as dist
        from plasticinelab_amd.distributed import make_slab_env
        # the device-side exchange is opt-in (distributed.HaloComm); the bench opts in and CHECKS it against the library
        # transport before timing anything (transport_check below) -- PLB_PEER_HALOS=0 keeps it off altogether
        env, layout, _ = make_slab_env(cfg, rank, world, compute_dtype=args.dtype, device=device, target_fn=_target,
                                       xy_margin=XY_MARGIN, migrate_every=1, peer=True,
                                       yield_stress=mixed_yield(args.particles) if getattr(args, "mixed_yield", False) else None)
        env.loss.set_weights(10, 10, 1, False)
        return env, (f"{world} z-slabs {list(layout.bounds)} (reach {layout.halo} layers), grid window = body + {XY_MARGIN} layers, zero-copy halo of "
                     f"one 4^3 block plane per face side, particle migration every env step")
    if getattr(args, "window", -1) >= 0:
        from plasticinelab_amd.engine.shapes import Shapes
        n = int(128 * args.quality * 0.5)
        b = (Shapes(cfg.SHAPES).get()[0] * n - 0.5).astype(np.int64)
        cfg.SIMULATOR["grid_window"] = ([int(v) for v in np.maximum(b.min(0) - args.window, 0)],
                                        [int(v) for v in np.minimum(b.max(0) + 3 + args.window, n)])
    env = TaichiEnv(cfg, compute_dtype=args.dtype, device=device)
    if getattr(args, "mixed_yield", False):
        env.simulator._yield_stress = mixed_yield(env.simulator.n_particles)
    env.initialize()
    env.loss.load_target_density(grids=_target(env.init_particles, env.simulator))
    env.loss.set_weights(10, 10, 1, False)
    note = "" if getattr(args, "window", -1) < 0 else f", grid window = body + {args.window} layers"
    if getattr(args, "deterministic", False):
        note += ", deterministic accumulation"
    if getattr(args, "mixed_yield", False):
        note += ", sigma_y = 50 / 1e9 alternating"
    return env, (("single GPU" if world == 1 else f"{world} independent replicas (no collective)") + note)


def rollout(env, actions):
    """K env steps forward + the reverse sweep (Tape): the timed body."""
    from plasticinelab_amd.engine.taichi_env import Tape
    with Tape(env):
        for a in actions:
            env.step(a)
            env.compute_loss()
    return env.loss.loss


def cpu_baseline(args, env):
    """The reference's CPU path is the Taichi CPU backend, which cannot be installed here; in
    its place: the C / OpenMP float64 restatement of one substep forward + reverse (oracle/mpm_substep_omp.c, kernel by
    kernel after mpm_simulator.py:60-278 in the reference's own layout -- AoS particles, dense n^3 grids, atomic
    scatters; checked against the torch oracle in tests/test_oracle_omp.py), timed on this box's host cores on the
    workload's particle cloud: median of 3 runs per thread count of a sweep (best count = the value) and of 3 runs on one
    core.  kind = "port"."""
    from oracle.omp_substep import OmpSubstep
    sim_g = env.simulator
    x0 = np.ascontiguousarray(getattr(env, "all_particles", None) if getattr(env, "all_particles", None) is not None else env.init_particles)
    N = len(x0)
    rng = np.random.default_rng(0)
    # a state with the features of a running rollout (velocities, velocity gradients, strained F; a third of it yields)
    v = rng.standard_normal((N, 3)) * 0.2
    Cm = rng.standard_normal((N, 3, 3)) * 1.0
    F = np.eye(3) + rng.standard_normal((N, 3, 3)) * 0.02
    E, nu = 5000.0, 0.2
    mu, lam = np.full(N, E / (2 * (1 + nu))), np.full(N, E * nu / ((1 + nu) * (1 - 2 * nu)))
    ys = np.full(N, float(getattr(args, "yield_stress", 200.0)))
    prims = list(env.primitives)
    pos = np.array([p.cfg.init_pos for p in prims], float)
    pos1 = pos + np.array([[0.8, 0, 0], [-0.8, 0, 0]])[:len(prims)] * 0.01 / sim_g.substeps
    omp = OmpSubstep(sim_g.n_grid, sim_g.dt, sim_g.p_vol, sim_g.p_mass, sim_g.default_gravity, sim_g.ground_friction, 666.0,
                     [p.params()[0] for p in prims], [p.friction for p in prims], N)
    cot = [rng.standard_normal((N, 3)), rng.standard_normal((N, 3)), rng.standard_normal((N, 3, 3)), rng.standard_normal((N, 3, 3))]

    def once():
        t0 = time.perf_counter()
        omp.forward(pos, pos1, x0, v, Cm, F, mu, lam, ys)
        omp.backward(pos, pos1, x0, v, Cm, F, mu, lam, ys, *cot)
        return time.perf_counter() - t0

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    omp.threads(avail)
    once()                                                     # first touch of the grids
    # "all cores": the atomic scatters stop scaling well before a 128-core box is full, so the thread count is swept
    # and the best one reported (all counts tried are in `thread_sweep`)
    sweep = {}
    for th in sorted({avail, max(avail // 2, 1), max(avail // 4, 1), min(32, avail), min(16, avail)}, reverse=True):
        omp.threads(th)
        sweep[th] = sorted(once() for _ in range(3))[1]
    cores = min(sweep, key=sweep.get)
    omp.threads(cores)
    t_all = sorted([sweep[cores]] + [once() for _ in range(2)])[1]
    omp.threads(1)
    t_one = sorted(once() for _ in range(3))[1]
    omp.threads(avail)
    return {"value": 1.0 / t_all, "unit": "substeps/s", "cores": cores, "kind": "port",
            "single_core_value": 1.0 / t_one, "all_core_seconds_per_substep": t_all, "single_core_seconds_per_substep": t_one,
            "cores_available": avail, "thread_sweep_seconds": {str(k): v for k, v in sweep.items()},
            "sample": f"one fwd+bwd substep of the {N}-particle / {sim_g.n_grid}^3 workload (seeded perturbed state: v, C, F != 0, I) through "
                      "oracle/mpm_substep_omp.c (C / OpenMP, float64, dense grids and the reference's recompute schedule): median of 3 runs per "
                      f"thread count, best count = {cores} of {avail} available; 1 thread: median of 3; Taichi (the reference's own CPU backend) is not installable here"}


def slab_how(env):
    """How the halos of a slab run travel, in words (config.parallelism)."""
    import torch.distributed as dist
    eng = env.simulator.engine
    backend = dist.get_backend()
    backend = "RCCL" if backend == "nccl" else backend
    if eng.native_loops:
        how = ("by the grid kernels themselves (exchange folded in: send | interior blocks | wait | face blocks, one launch)" if eng.peer_fused()
               else "by an exchange kernel")
        return (f"halos written {how} into the neighbours' IPC-mapped receive areas each substep (fwd + adjoint; device-side exchange, native "
                f"substep loops, no host-side communication per substep; {backend} for migration and the per-env-step reductions)")
    return f"halos summed over {backend} point-to-point each substep (fwd + adjoint)"


def one_step_loss_and_grad(env, state0, A):
    """One env step forward + its reverse on the slab engine's current transport: (loss, d loss / d action)."""
    env.set_state(state0, 666.0, False)
    loss = rollout(env, seeded_actions(1, A))
    return float(loss), np.array(env.primitives.get_grad(1), dtype=np.float64)


def transport_check(env, state0, A, agree, tol_loss=1e-5, tol_grad=1e-4, gather=None):
    """N > 1, device-side exchange set up: run ONE env step fwd + bwd through the peer-write exchange and again through the
    library's point-to-point transport (RCCL on the GPUs' backend) and compare loss and action gradient.  The first real
    multi-GPU run of this code happens without anybody watching: a visibility bug of the hand-written exchange across GPUs
    would give a wrong-but-plausible line, so the line says which transport it trusted and why.  Collective.  Returns the
    record that goes into the JSON line; the engine is left on the transport to time."""
    eng = env.simulator.engine
    rec = {"checked": False, "peer_available": bool(getattr(eng.comm, "peer_mapped", False))}
    # first-contact record of every rank (distributed.HaloComm.setup_peer): device, hipDeviceCanAccessPeer towards both neighbours,
    # kind of memory the receive areas got, whether the IPC handles opened, one hand-off of the exchange's pattern across each face
    rec["preflight"] = gather(getattr(eng.comm, "preflight", None)) if gather else None
    if not rec["peer_available"]:
        rec["reason"] = getattr(eng.comm, "peer_error", None) or "device-side exchange not set up (PLB_PEER_HALOS=0, or no IPC mapping)"
        rec["used"] = eng.use_transport("p2p")
        return rec
    old_to = os.environ.get("PLMPM_PEER_TIMEOUT")
    os.environ["PLMPM_PEER_TIMEOUT"] = os.environ.get("PLB_CHECK_PEER_TIMEOUT", "5")      # a dead hand-off costs seconds here, not minutes
    res, err = {}, {}
    for kind in ("peer", "p2p"):
        eng.use_transport(kind)
        try:
            res[kind] = one_step_loss_and_grad(env, state0, A)
            ok = True
        except Exception as e:                                    # noqa: BLE001
            ok, err[kind] = False, f"{type(e).__name__}: {str(e)[:160]}"
        if not agree(ok):
            res.pop(kind, None)
            err.setdefault(kind, "failed on another rank")
    if old_to is None:
        os.environ.pop("PLMPM_PEER_TIMEOUT", None)
    else:
        os.environ["PLMPM_PEER_TIMEOUT"] = old_to
    rec["checked"] = True
    rec["errors"] = err or None
    same = False
    if "peer" in res and "p2p" in res:
        (lp, gp), (lq, gq) = res["peer"], res["p2p"]
        rec["loss_peer"], rec["loss_p2p"] = lp, lq
        rec["rel_loss"] = abs(lp - lq) / max(abs(lq), 1e-300)
        rec["rel_grad"] = float(np.abs(gp - gq).max() / max(np.abs(gq).max(), 1e-300))
        same = rec["rel_loss"] < tol_loss and rec["rel_grad"] < tol_grad
    same = agree(same)
    rec["agree"] = same
    if same:
        rec["used"] = eng.use_transport("peer")
    elif "p2p" in res:
        rec["used"] = eng.use_transport("p2p")
        rec["reason"] = "device-side exchange disagrees with the library transport (or failed): FALLBACK to point-to-point halos"
    else:
        rec["used"] = None
        rec["reason"] = "neither transport completed one env step"
    return rec


N1_LOSS_FILE = os.path.join(ROOT, "profiles", "n1_final_loss.json")
N1_POINTS_FILE = os.path.join(ROOT, "profiles", "n1_reference_points.json")


def n1_reference(workload, dtype, steps):
    """The committed single-GPU record of a (workload, dtype, steps) rollout -- {"value", "final_loss", "source"} from
    profiles/n1_reference_points.json (written from N = 1 runs of this script / profiles/tools/config5_n1_reference.py), or the
    older loss-only file profiles/n1_final_loss.json -- or None."""
    key = f"{workload}|{dtype}|{steps}"
    try:
        with open(N1_POINTS_FILE) as f:
            rec = json.load(f).get(key)
        if rec is not None:
            return dict(rec, file="profiles/n1_reference_points.json")
    except (OSError, ValueError):
        pass
    try:
        with open(N1_LOSS_FILE) as f:
            v = json.load(f).get(key)
        if v is not None:
            return {"final_loss": v, "value": None, "file": "profiles/n1_final_loss.json"}
    except (OSError, ValueError):
        pass
    return None


def loss_check(workload, dtype, steps, final_loss, tol=1e-5):
    """final_loss against the committed single-GPU loss of the same (workload, dtype, steps).  The slab run computes the SAME
    rollout, so its loss must agree to the engines' round-off; `rel` beyond 1e-5 puts MISMATCH into `metric`."""
    ref = n1_reference(workload, dtype, steps)
    if ref is None or ref.get("final_loss") is None:
        return {"n1_expected": None, "rel": None, "ok": None, "source": "no committed single-GPU loss for this (workload, dtype, steps)"}
    exp = ref["final_loss"]
    rel = abs(final_loss - exp) / max(abs(exp), 1e-300)
    return {"n1_expected": exp, "rel": rel, "ok": bool(rel <= tol), "tol": tol, "source": ref["file"]}


def workload_name(args, n_grid):
    if getattr(args, "mixed_yield", False):
        return f"mixed{n_grid}_{args.particles}p"
    if args.workload != "config3_cube128":
        return args.workload if (args.particles, args.quality) == (500_000, 2) else f"{args.workload.rstrip('0123456789')}{n_grid}_{args.particles}p"
    return "config3_cube128" if (args.particles, args.quality, args.side) == (500_000, 2, 0.31) else f"cube{n_grid}_{args.particles}p"


def child_point(extra, timeout=600):
    """One more single-GPU point in a child process of its own (its frames do not meet the headline's): the bench line it
    prints, reduced to the keys a secondary point carries."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1"] + extra + ["--no-cpu-baseline", "--no-secondary"]
    # a child of one rank of an N > 1 run is a world of its own on that rank's GPU: the launcher's variables must not reach it
    cenv = {k: v for k, v in os.environ.items()
            if k not in ("RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE", "MASTER_ADDR",
                         "MASTER_PORT", "PLB_BENCH_NOTE", "PLB_FORCE_SECONDARY") and not k.startswith("TORCHELASTIC")}
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=cenv)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not line:
            return {"error": f"rc {p.returncode}: {p.stderr[-300:]}"}
        d = json.loads(line[-1])
        r = d["roofline"]
        return {"workload": d["config"]["workload"], "n_grid": d["config"]["n_grid"], "n_particles": d["config"]["n_particles"],
                "substeps_per_step": d["config"]["substeps_per_step"], "steps": d["steps"], "warmup": d["warmup"], "repeats": d["repeats"],
                "dtype": d["dtype"], "value": d["value"], "value_min": d["value_min"], "value_max": d["value_max"], "unit": d["unit"],
                "ms_per_step": d["ms_per_step"],
                "substep_kernel_sum_us": r["substep_kernel_sum_us"], "substep_alg_MB": r["substep_alg_MB"], "substep_frac": r["substep_frac"],
                "job_frac": r["job_frac"], "dominant_kernel": r["kernel"], "dominant_frac": r["frac"], "active_nodes": r["active_nodes"],
                "final_loss": d["final_loss"], "parallelism": d["config"]["parallelism"],
                "command": "python bench.py " + " ".join(cmd[2:])}
    except Exception as e:                                        # noqa: BLE001 -- a secondary point never takes the headline down
        return {"error": f"{type(e).__name__}: {e}"}


def secondary_scale():
    """Test knob: PLB_SECONDARY_SCALE = s runs the secondary points with s x the particles in a cube of s^(1/3) x the side (the
    same particles per cell); the contract tests use it, the driver's runs do not (1)."""
    return float(os.environ.get("PLB_SECONDARY_SCALE", "1"))


# the secondary points: BASELINE configs[3] and configs[4] as SURVEY 8(d) synthesises them (~8 particles per cell: cube side
# 0.25 of the domain at either grid)
def point_config4(scale=1.0):
    return dict(label="configs[3]: 256^3 / 2M elastic", particles=int(round(2_000_000 * scale)), quality=4.0, steps=2, warmup=1,
                yield_stress=1e9, side=0.25 * scale ** (1 / 3), mixed_yield=False)


def point_config5(scale=1.0):
    return dict(label="configs[4]: 512^3 / 16M, half sigma_y = 50 / half 1e9, one env step of 159 substeps", particles=int(round(16_000_000 * scale)),
                quality=8.0, steps=1, warmup=1, yield_stress=200.0, side=0.25 * scale ** (1 / 3), mixed_yield=True)


def secondary_point(args):
    """The config-4-size single-GPU point, so that it is the DRIVER that observes it: BASELINE configs[3]'s workload (256^3 grid,
    2M elastic particles: sigma_y = 1e9, cube side 0.25 = ~8 particles per cell, 79 substeps per env step) on one GPU, 2 env steps
    fwd + bwd, grid window = body + 24 layers.  Four times the particles amortise the workgroups' latency chains and the grid
    kernels: the whole substep reaches a higher fraction of the roofline than at 128^3 / 500k."""
    pt = point_config4(secondary_scale())
    return child_point(["--particles", str(pt["particles"]), "--quality", "4", "--side", repr(pt["side"]), "--window", "24", "--steps", "2", "--warmup", "1",
                        "--repeats", "3", "--yield-stress", "1e9", "--dtype", args.dtype])


def secondary_f64_point(args):
    """The float64 engine on the headline workload, 4 env steps fwd + bwd: the reference computes in float64 only
    (mpm_simulator.py:8), so this is its own precision stated beside the fp32 number (same kernels, T = double)."""
    d = child_point(["--steps", "4", "--warmup", "1", "--repeats", "3", "--dtype", "float64", "--particles", str(args.particles),
                     "--quality", repr(args.quality), "--side", repr(args.side), "--yield-stress", repr(args.yield_stress)])
    ref = n1_reference(d.get("workload"), "f32", 4) if "error" not in d else None
    if ref and ref.get("final_loss"):
        d["f32_loss_same_rollout"] = ref["final_loss"]
        d["rel_to_f32_loss"] = abs(d["final_loss"] - ref["final_loss"]) / abs(ref["final_loss"])
    return d


class World:
    """rank / world / device and the control-plane collectives of a run (a gloo group of its own when the data plane is RCCL)."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", 0))
        self.world = int(os.environ.get("WORLD_SIZE", 1))
        local = int(os.environ.get("LOCAL_RANK", 0)) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        self.device = torch.device("cuda", local)
        self.dist, self.ctl = None, None
        if self.world > 1:
            import datetime
            import torch.distributed as dist
            self.dist = dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = os.environ.get("PLB_DIST_BACKEND", "nccl")     # "gloo" lets several ranks share one GPU (testing only)
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=self.device)
                # control plane (agreement, barriers, the max over ranks) on its own gloo group: it keeps working when the
                # data plane -- RCCL point-to-point between slabs -- does not
                self.ctl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=1800))
            else:
                dist.init_process_group(backend)

    def barrier(self):
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier(group=self.ctl)
        torch.cuda.synchronize()

    def agree(self, ok):
        """True only if every rank succeeded."""
        if self.dist is None:
            return ok
        t = torch.tensor([1.0 if ok else 0.0])
        if self.ctl is None and self.dist.get_backend() == "nccl":
            t = t.to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.ctl)
        return bool(t.item() > 0.5)

    def max_over_ranks(self, x):
        if self.dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.ctl)
        return float(t.item())

    def gather_objects(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on every rank (control plane)."""
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj, group=self.ctl)
        return out

    def sum_over_ranks(self, values):
        t = torch.tensor([float(v) for v in values], dtype=torch.float64)
        if self.dist is not None:
            if self.ctl is None and self.dist.get_backend() == "nccl":
                t = t.to(self.device)
            self.dist.all_reduce(t, group=self.ctl)
            t = t.cpu()
        return [float(v) for v in t]


def timed_rollouts(Wd, env, state0, acts, repeats):
    """`repeats` x the K-step rollout, each bracketed by barrier + synchronize, the state put back outside the timed spans
    (inputs resident in HBM when a span starts): -> (seconds per repetition, max over ranks; loss of the last one)."""
    times, loss = [], None
    for _ in range(repeats):
        env.set_state(state0, 666.0, False)
        Wd.barrier()
        t0 = time.perf_counter()
        err = None
        try:
            loss = rollout(env, acts)
        except Exception as e:                                    # noqa: BLE001
            err = e
        Wd.barrier()
        dt = Wd.max_over_ranks(time.perf_counter() - t0)
        # a rank that failed alone still meets the others at the SAME collectives (barrier, max, agreement) and only then raises --
        # on every rank: nobody is left in a barrier while another rank has moved on to a different collective
        if not Wd.agree(err is None):
            raise err if err is not None else RuntimeError("the rollout failed on another rank")
        times.append(dt)
    return times, loss


def slab_point(Wd, base_args, pt, transport):
    """One secondary point at N >= 2, inside the running world: build the point's slab engines, warm up (= feasibility), time
    `repeats` rollouts, check the loss against the committed single-GPU loss.  Every failure is agreed on by all ranks before
    anyone moves on (a rank that bailed out alone would leave the others in their next exchange)."""
    import copy
    a = copy.copy(base_args)
    a.particles, a.quality, a.steps, a.warmup = pt["particles"], pt["quality"], pt["steps"], pt["warmup"]
    a.yield_stress, a.side, a.mixed_yield, a.workload, a.window, a.deterministic = pt["yield_stress"], pt["side"], pt["mixed_yield"], "config3_cube128", -1, False
    rec = {"label": pt["label"], "n_gpus": Wd.world, "steps": a.steps, "warmup": a.warmup, "dtype": "f32" if a.dtype == "float32" else "f64"}
    env, err = None, None
    t_build = time.perf_counter()
    fused_before = os.environ.get("PLMPM_PEER_FUSED")
    try:
        # built so that BOTH forms of the device-side exchange can run on it (the fused grid kernels need every grid workgroup
        # resident: make_slab_env caps plmpm_config.grid_workgroups when it sees PLMPM_PEER_FUSED=1); timed with the default form
        os.environ["PLMPM_PEER_FUSED"] = "1"
        try:
            env, parallelism = build_env(a, Wd.device, Wd.rank, Wd.world, slabs=True)
        finally:
            os.environ["PLMPM_PEER_FUSED"] = "0"
        eng = env.simulator.engine
        used = eng.use_transport(transport)               # what the headline's transport check settled on
        state0 = env.get_state()["state"]
        A = env.primitives.action_dim
        if a.warmup > 0:
            env.set_state(state0, 666.0, False)
            rollout(env, seeded_actions(a.warmup, A))
        ok = True
    except Exception as e:                                        # noqa: BLE001
        ok, err = False, f"{type(e).__name__}: {str(e)[:200]}"
    def restore_fused():
        if fused_before is None:
            os.environ.pop("PLMPM_PEER_FUSED", None)
        else:
            os.environ["PLMPM_PEER_FUSED"] = fused_before

    if not Wd.agree(ok):
        rec["error"] = err or "failed on another rank"
        restore_fused()
        return rec, env
    rec["build_and_warmup_s"] = round(time.perf_counter() - t_build, 2)
    # the persistent grid launches of THIS engine: capped for residency because the fused form is timed on it too (the library's
    # default is 512 where the window has that many blocks); `value` of the default transport is measured with this cap
    rec["grid_workgroups"] = int(getattr(env.simulator.engine, "grid_workgroups", 0) or 0)
    sim = env.simulator
    sub = sim.substeps
    try:
        times, loss = timed_rollouts(Wd, env, state0, seeded_actions(a.steps, A), max(1, min(3, base_args.repeats)))
        nodes, _ = sim.engine.grid_stats(0)
        ok = True
    except Exception as e:                                        # noqa: BLE001
        ok, err, times, loss, nodes = False, f"{type(e).__name__}: {str(e)[:200]}", [], None, 0
    if not Wd.agree(ok):
        rec["error"] = err or "failed on another rank"
        restore_fused()
        return rec, env
    n_all, a_all = Wd.sum_over_ranks([sim.n_particles, nodes])     # halo nodes are active on both neighbours: swept twice, counted twice
    med = sorted(times)[len(times) // 2]
    total = a.steps * sub
    name = workload_name(a, sim.n_grid)
    alg = 4.0 * (150 * n_all + 57 * a_all)
    rec.update({"workload": name, "n_grid": sim.n_grid, "n_particles": a.particles, "substeps_per_step": sub, "repeats": len(times),
                "value": total / med, "value_min": total / max(times), "value_max": total / min(times), "unit": "substeps/s",
                "ms_per_step": 1e3 * med / a.steps, "scaling": "strong", "final_loss": float(loss),
                "job_alg_MB_per_substep": alg * 1e-6, "job_frac": alg * (total / med) / (Wd.world * HBM_PEAK_GBS * 1e9),
                "active_nodes_all_ranks": int(a_all), "halo_transport": sim.engine.transport, "parallelism": parallelism + ", " + slab_how(env)})
    rec["loss_check"] = loss_check(name, rec["dtype"], a.steps, float(loss))
    ref = n1_reference(name, rec["dtype"], a.steps)
    if ref and ref.get("value"):
        rec["n1_value"] = ref["value"]
        rec["n1_source"] = ref.get("source")
        # what the single-GPU reference timed: "same rollout", or -- configs[4], whose 160 frames do not fit one GPU -- a window of it
        rec["n1_schedule"] = ref.get("schedule", "same rollout")
        rec["strong_scaling_eff"] = rec["value"] / (Wd.world * ref["value"])
    else:
        rec["n1_value"], rec["strong_scaling_eff"], rec["n1_schedule"] = None, None, None
    # BASELINE configs[4] asks for "halo-overlapped substeps": the same engine once more with the exchange FOLDED INTO the grid
    # kernels (PLMPM_PEER_FUSED=1: send | interior blocks | wait | blocks of the exchanged planes in one launch -- the interior hides
    # the arrival), timed and loss-checked like the default form.  Only on the device-side transport; its failure costs nothing else.
    if sim.engine.native_loops:
        try:
            os.environ["PLMPM_PEER_FUSED"] = "1"
            env.set_state(state0, 666.0, False)
            rollout(env, seeded_actions(1, A))                     # warm-up of the fused kernels (first use: code load)
            t2, l2 = timed_rollouts(Wd, env, state0, seeded_actions(a.steps, A), len(times))
            ok = True
        except Exception as e:                                    # noqa: BLE001
            ok, err = False, f"{type(e).__name__}: {str(e)[:200]}"
        finally:
            os.environ["PLMPM_PEER_FUSED"] = "0"
        if Wd.agree(ok):
            m2 = sorted(t2)[len(t2) // 2]
            rec["halo_overlapped"] = {"value": total / m2, "value_min": total / max(t2), "value_max": total / min(t2), "final_loss": float(l2),
                                      "loss_check": loss_check(name, rec["dtype"], a.steps, float(l2)),
                                      "halo_transport": "peer-write, exchange folded into the grid kernels (PLMPM_PEER_FUSED=1)",
                                      "speedup_vs_default": med / m2}
        else:
            rec["halo_overlapped"] = {"error": err or "failed on another rank"}
    restore_fused()
    return rec, env


def close_env(env):
    """Give an engine's HBM back (the secondary points are built after the headline's engine is gone)."""
    try:
        env.simulator.engine.close()
    except Exception:                                             # noqa: BLE001
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--repeats", type=int, default=5, help="the K-step rollout is timed this many times (state re-uploaded outside the timed "
                    "spans); value / ms_per_step are the median repetition")
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--particles", type=int, default=500_000)
    ap.add_argument("--workload", default="config3_cube128", choices=sorted(WORKLOADS),
                    help="config3_cube128: the headline (SURVEY 8d's synthetic cube, two spheres); triplemove128 / rope128: the reference's "
                         "TripleMove-v1 / Rope-v1 geometry (6 spheres / 2 spheres + static cylinder + ground friction) at 128^3 with ~500k "
                         "particles -- BASELINE configs[2] as it names the scenes")
    ap.add_argument("--quality", type=float, default=2)
    ap.add_argument("--side", type=float, default=0.31, help="side of the synthetic cube (0.31 at 128^3 / 500k = ~8 particles per cell)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--replicas", action="store_true", help="N > 1: independent replicas instead of z-slabs")
    ap.add_argument("--yield-stress", type=float, default=200.0, help="von Mises yield stress of the cube (200 = the workload; 1e9 = elastic: an "
                    "experiment knob, not the headline)")
    ap.add_argument("--mixed-yield", action="store_true", help="BASELINE configs[4]'s material: sigma_y = 50 / 1e9 alternating over the particles")
    ap.add_argument("--deterministic", action="store_true", help="single GPU: the bit-reproducible engine (integer-limb accumulation); "
                    "a cost measurement, not the headline")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary points (module docstring) that are measured AFTER the "
                    "headline's timed region and reported under `secondary` / `secondary_f64`")
    ap.add_argument("--window", type=int, default=-1, help="single GPU: allocate / sweep only the body's bounding box + this many node "
                    "layers of the grid (plmpm_config.grid_lo / grid_hi); -1 = the whole grid, as the reference lays it out")
    args = ap.parse_args()

    Wd = World()
    rank, world, device, dist = Wd.rank, Wd.world, Wd.device, Wd.dist
    barrier, agree, max_over_ranks = Wd.barrier, Wd.agree, Wd.max_over_ranks

    # A slab run that HANGS (a rank stuck in an RCCL call has no exception to catch) must not take the whole job with
    # it: if the slab build + warm-up has not been agreed on within PLB_SLAB_TIMEOUT seconds, every rank replaces its
    # own process image by the replica run of the same command (fresh GPU context, fresh rendezvous one port up).
    def arm_watchdog():
        import threading
        limit = float(os.environ.get("PLB_SLAB_TIMEOUT", 300))

        def fire():
            sys.stderr.write(f"[bench] rank {rank}: slab warm-up not finished after {limit:.0f} s -- re-running as replicas\n")
            sys.stderr.flush()
            env2 = dict(os.environ)
            env2["MASTER_PORT"] = str(int(env2.get("MASTER_PORT", "29500")) + 1)
            env2["TORCHELASTIC_USE_AGENT_STORE"] = "False"       # rank 0 hosts the new store itself
            env2["PLB_BENCH_NOTE"] = f"slab warm-up timed out after {limit:.0f} s"
            argv = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--replicas"] + ["--replicas"]
            os.execve(sys.executable, argv, env2)

        t = threading.Timer(limit, fire)
        t.daemon = True
        t.start()
        return t

    K, W, R = args.steps, args.warmup, max(1, args.repeats)
    phases = {}                                   # where this process' wall clock goes (seconds), reported as `phases_s`
    t_phase = time.perf_counter()

    def phase_done(name):
        nonlocal t_phase
        torch.cuda.synchronize()
        now = time.perf_counter()
        phases[name] = phases.get(name, 0.0) + now - t_phase
        t_phase = now

    slabs = world > 1 and not args.replicas
    env, state0, tcheck = None, None, None
    if slabs:
        note = ""
        watchdog = arm_watchdog()
        try:
            if os.environ.get("PLB_BENCH_FAKE_HANG") == str(rank):      # test hook for the watchdog
                time.sleep(1e6)
            env, parallelism = build_env(args, device, rank, world, slabs=True)
            ok = True
        except Exception as e:                                    # noqa: BLE001
            ok, note = False, f"{type(e).__name__}: {e}"
        if agree(ok):
            try:
                state0 = env.get_state()["state"]
                tcheck = transport_check(env, state0, env.primitives.action_dim, agree, gather=Wd.gather_objects)
                if tcheck.get("used") is None:
                    raise RuntimeError(tcheck.get("reason", "no usable halo transport"))
                parallelism += ", " + slab_how(env)
                if tcheck.get("checked") and not tcheck.get("agree"):
                    parallelism += " -- FALLBACK from the device-side exchange: " + str(tcheck.get("errors") or tcheck.get("reason"))
                # the warm-up doubles as the feasibility check
                env.set_state(state0, 666.0, False)
                rollout(env, seeded_actions(max(W, K), env.primitives.action_dim))
                ok = True
            except Exception as e:                                # noqa: BLE001
                ok, note = False, f"{type(e).__name__}: {e}"
            ok = agree(ok)
        else:
            ok = False
        watchdog.cancel()
        if not ok:
            if rank == 0:
                print(f"[bench] slab path unavailable ({note or 'another rank failed'}); falling back to replicas", file=sys.stderr)
            if env is not None:
                close_env(env)
            env, slabs, state0 = None, False, None
            os.environ["PLB_BENCH_NOTE"] = "slab path unavailable" + (f": {note[:120]}" if note else "")
            torch.cuda.empty_cache()
    if env is None:
        env, parallelism = build_env(args, device, rank, world, slabs=False)
        if os.environ.get("PLB_BENCH_NOTE"):
            parallelism += f" ({os.environ['PLB_BENCH_NOTE']})"
        state0 = env.get_state()["state"]
    sim = env.simulator
    sub = sim.substeps
    A = env.primitives.action_dim
    phase_done("build_and_upload")

    # the box's own HBM roofs (reported in `roofline`), measured before anything is timed: 1 GiB copy and read sweeps
    # with the library's 16 B / lane kernels -- which also brings a fresh box's clocks up before the warm-up steps
    copy_gbs, read_gbs = sim.engine.measure_hbm() if not args.no_roofline else (None, None)
    phase_done("hbm_roofs")
    if W > 0 and not slabs:
        env.set_state(state0, 666.0, False)
        rollout(env, seeded_actions(W, A))
    phase_done("warmup")
    acts = seeded_actions(K, A)
    # EXACTLY K steps per repetition, R repetitions, each bracketed by barrier + synchronize; inputs resident in HBM before each
    times, loss = timed_rollouts(Wd, env, state0, acts, R)
    phase_done("timed")
    elapsed = sorted(times)[len(times) // 2]                   # the median repetition (max over ranks each)
    total_substeps = K * sub * (1 if slabs else world)         # slabs: one shared workload; replicas: one each
    value = total_substeps / elapsed

    out = {
        # N = 1 is the first point of the strong-scaling series (the whole workload on one GPU); replicas are not the metric
        "metric": "MPM substeps/sec (fwd+bwd)" + ("" if slabs or world == 1 else f" -- FALLBACK: {world} independent replicas, not the sliced workload"),
        "value": value, "unit": "substeps/s", "n_gpus": world,
        "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "strong" if (slabs or world == 1) else "weak",
        "vs_baseline": None, "dtype": "f32" if args.dtype == "float32" else "f64", "data": "synthetic",
        "config": {"workload": workload_name(args, sim.n_grid),
                   "n_grid": sim.n_grid,
                   "n_particles": args.particles if slabs else sim.n_particles,
                   "substeps_per_step": sub, "primitives": len(env.primitives), "positions": "f64", "loss": "sdf+density+hard contact",
                   "parallelism": parallelism},
        # the K-step rollout was timed `repeats` times; value / ms_per_step are the median repetition
        "repeats": R, "value_min": total_substeps / max(times), "value_max": total_substeps / min(times),
        "repeat_ms_per_step": [round(1e3 * t / K, 5) for t in times], "timed_region_s": round(sum(times), 5),
        "final_loss": float(loss),
    }
    # self-validation of the line: the loss against the committed single-GPU loss of the same rollout,
    # which transport carried the halos and whether it was checked, and the size of the RCCL communicator
    out["loss_check"] = loss_check(out["config"]["workload"], out["dtype"], K, float(loss)) if (slabs or world == 1) else None
    if out["loss_check"] and out["loss_check"]["ok"] is False:
        out["metric"] += " -- MISMATCH: final loss differs from the single-GPU run of the same rollout"
    if slabs:
        out["halo_transport"] = env.simulator.engine.transport
        out["transport_check"] = tcheck
        if tcheck and tcheck.get("checked") and not tcheck.get("agree"):
            out["metric"] += " -- FALLBACK: device-side halo exchange failed its check, halos over point-to-point"
        ref = n1_reference(out["config"]["workload"], out["dtype"], K)
        out["n1_value"] = ref.get("value") if ref else None
        out["strong_scaling_eff"] = value / (world * ref["value"]) if ref and ref.get("value") else None
    if dist is not None:
        ranks = None
        if dist.get_backend() == "nccl":
            t = torch.ones(1, device=device)
            dist.all_reduce(t)                                    # what the RCCL communicator itself counts
            ranks = int(round(t.item()))
        out["rccl_ranks"] = ranks
        out["dist_backend"] = dist.get_backend()

    workload = out["config"]["workload"]
    if not args.no_roofline:
        # per-kernel durations, measured live with HIP events on the launch stream over the same K-step rollout.  N > 1:
        # every rank replays the rollout (it contains the exchanges); the reported kernels are rank 0's, their algorithmic
        # bytes those of rank 0's launches (its particles, its active nodes), the job-level fraction uses the whole
        # workload's bytes against N x the per-GPU peak.
        env.set_state(state0, 666.0, False)
        nodes, blocks = sim.engine.grid_stats(0)
        sim.engine.profile_enable(True)
        rollout(env, acts)
        prof = sim.engine.profile_read()
        sim.engine.profile_enable(False)
        phase_done("roofline_pass")
        N = sim.n_particles                                   # this rank's particles (at reset)
        tot = Wd.sum_over_ranks([N, nodes])                   # halo nodes are active on both neighbours: counted twice, as they are swept twice
        if slabs is False and world > 1:
            tot = [v / world for v in tot]                    # replicas: every rank holds the whole workload
    if rank == 0 and not args.no_roofline:
        kernels = {}
        for name, (ms, cnt) in prof.items():
            if cnt == 0:
                continue
            cN, cA = ALG.get(name, (0, 0))
            avg = ms / cnt
            kernels[name] = {"avg_us": 1e3 * avg, "launches": cnt, "alg_MB": 4e-6 * (cN * N + cA * nodes),
                             "GBps": 4e-9 * (cN * N + cA * nodes) / (1e-3 * avg) if avg > 0 else 0.0}
        # the dominant kernel among those that move the workload's bytes (the halo exchange of a slab run is a copy of
        # two block planes plus the wait for the neighbour: it counts in the per-substep sum, not as "the" kernel)
        dom = max((k for k in kernels if ALG.get(k, (0, 0)) != (0, 0) and not k.startswith("xchg+")), key=lambda k: kernels[k]["avg_us"] * kernels[k]["launches"])
        alg_substep = 4.0 * (150 * N + 57 * nodes)            # this rank's share
        alg_unit = 4.0 * (150 * float(tot[0]) + 57 * float(tot[1]))     # bytes of one substep as counted in `value`
        sum_us = sum(v["avg_us"] * v["launches"] for v in kernels.values()) / (K * sub)     # per fwd+bwd substep
        traffic, traffic_src = pmc_traffic(dom, workload, out["dtype"], K, W) if world == 1 else (None, None)
        for name in kernels:                      # per-kernel HBM bytes per launch from the same PMC passes (null when they do not apply)
            kernels[name]["traffic"] = pmc_traffic(name, workload, out["dtype"], K, W)[0] if world == 1 else None
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": kernels[dom]["GBps"] / HBM_PEAK_GBS,
                           "traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_src,
                           "algorithmic_bytes": kernels[dom]["alg_MB"] * 1e6,
                           "scope": "single GPU" if world == 1 else f"rank 0 of {world} ({N} particles, {nodes} active nodes)",
                           "active_nodes": nodes, "active_blocks": blocks,
                           "substep_alg_MB": alg_substep * 1e-6,
                           # the lenient dense-grid byte count (every node of the n^3 grid, not only the
                           # active ones) -- reported for reference, NOT what frac / substep_frac are computed from
                           "substep_alg_MB_dense_grid": 4.0 * (150 * float(tot[0]) + 57 * sim.n_grid ** 3) * 1e-6,
                           # the roofs this very box reaches, measured just now with the library's 16 B / lane kernels
                           "peak_measured_copy": copy_gbs, "peak_measured_read": read_gbs,
                           "substep_kernel_sum_us": sum_us,
                           "substep_frac": (alg_substep / (sum_us * 1e-6)) / (HBM_PEAK_GBS * 1e9),
                           "substep_frac_of_measured_read": (alg_substep / (sum_us * 1e-6)) / (read_gbs * 1e9),
                           # whole job, wall clock: the workload's algorithmic bytes per fwd+bwd substep x substeps/s
                           # against n_gpus x the spec peak
                           "job_alg_MB_per_substep": alg_unit * 1e-6,
                           "job_frac": alg_unit * value / (world * HBM_PEAK_GBS * 1e9),
                           "kernels": kernels}
        # the secondary (vector-ALU) roof of SURVEY 8(d): null until a calibration of this workload is committed
        out["roofline"]["valu"] = valu_roof(kernels, N, workload, out["dtype"], K * sub) if world == 1 else None
    headline = (args.workload, args.particles, args.quality, args.window, args.yield_stress, args.side, args.mixed_yield) == \
               ("config3_cube128", 500_000, 2, -1, 200.0, 0.31, False)
    # (PLB_FORCE_SECONDARY=1: the contract tests run the secondary points behind a reduced headline, with PLB_SECONDARY_SCALE)
    want_secondary = (headline or os.environ.get("PLB_FORCE_SECONDARY") == "1") and not args.no_secondary and not args.no_roofline and not args.deterministic and args.dtype == "float32"
    if world == 1 and want_secondary and rank == 0:
        out["secondary"] = secondary_point(args)
        phase_done("secondary")
        out["secondary_f64"] = secondary_f64_point(args)
        phase_done("secondary_f64")
    if world > 1 and want_secondary and slabs:
        # the sizes that can scale, inside the same world, after the headline's engine has given its HBM back.  A point
        # that hangs must not cost the headline: after PLB_SECONDARY_TIMEOUT seconds rank 0 prints the line as it stands and
        # every rank leaves.
        import threading
        transport = "peer" if (tcheck and tcheck.get("agree")) else "p2p"
        close_env(env)
        del env, sim
        torch.cuda.empty_cache()
        points = [point_config4(secondary_scale())] if world in (2, 4, 8) else []
        if world == 8:
            points.append(point_config5(secondary_scale()))
        out["secondary"] = []
        limit = float(os.environ.get("PLB_SECONDARY_TIMEOUT", 900))

        def give_up():
            if rank == 0:
                out["secondary"].append({"error": f"secondary points not finished after {limit:.0f} s: abandoned"})
                out["phases_s"] = {k: round(v, 4) for k, v in phases.items()}
                print(json.dumps(out), flush=True)
            os._exit(0)

        timer = threading.Timer(limit, give_up)
        timer.daemon = True
        timer.start()
        # The N = 1 points of the strong-scaling series measured on THIS box, in this run: rank 0 alone, in child
        # processes on its own GPU, times the headline and the configs[3]-size point on one GPU while the other ranks wait at the
        # control-plane barrier (~25 s).  `strong_scaling_eff` divides by a number committed from another box (boxes differ by 3 %
        # and more); `strong_scaling_eff_same_box` divides by these.
        same_box = {}
        if rank == 0 and os.environ.get("PLB_SAME_BOX_N1", "1") != "0":
            hl = child_point(["--steps", str(K), "--warmup", str(W), "--repeats", "3", "--dtype", args.dtype, "--particles", str(args.particles),
                              "--quality", repr(args.quality), "--side", repr(args.side), "--yield-stress", repr(args.yield_stress)])
            same_box["headline"] = hl
            same_box["configs[3]"] = secondary_point(args) if points else None
        barrier()
        phase_done("n1_same_box")
        if rank == 0 and same_box:
            hl = same_box["headline"]
            out["n1_same_box"] = {k: (v if v is None or "error" in v else {q: v[q] for q in ("workload", "value", "value_min", "value_max", "steps", "repeats", "final_loss", "command")})
                                  for k, v in same_box.items()}
            out["strong_scaling_eff_same_box"] = value / (world * hl["value"]) if hl and "error" not in hl and hl.get("value") else None
        for pt in points:
            rec, penv = slab_point(Wd, args, pt, transport)
            n1 = same_box.get("configs[3]") if pt is points[0] else None
            if rank == 0 and "error" not in rec:
                rec["strong_scaling_eff_same_box"] = (rec["value"] / (world * n1["value"])) if (n1 and "error" not in n1 and n1.get("value")) else None
                rec["n1_same_box_value"] = n1["value"] if (n1 and "error" not in n1) else None
            out["secondary"].append(rec)
            if penv is not None:
                close_env(penv)
            del penv
            torch.cuda.empty_cache()
            phase_done("secondary")
        timer.cancel()
        env = None
    # (rank 0 at N = 1 only; the C / OpenMP restatement knows Sphere manipulators only)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "config3_cube128" and env is not None:
        out["cpu_baseline"] = cpu_baseline(args, env)
        phase_done("cpu_baseline")
    if rank == 0:
        # rank 0's wall clock by phase: the GPU is busy in hbm_roofs, warmup, timed and roofline_pass only; build_and_upload
        # is host work (scene, 96 MB of float64 state over PCIe, first-touch of the frame store), cpu_baseline is host only
        out["phases_s"] = {k: round(v, 4) for k, v in phases.items()}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
