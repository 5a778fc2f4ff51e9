"""CPU tier: host-side mirrors of the reference interface that need no GPU (optimisers, primitive descriptions,
Tape event bookkeeping with a stub engine)."""
import json
import sys

import numpy as np
from tests.util import ROOT
import pytest

from plasticinelab_amd.engine.primitives import Primitive, Primitives
from plasticinelab_amd.envs.scenes import load_scene
from plasticinelab_amd.optimizer.optim import Adam, Momentum


def test_adam_matches_closed_form():
    p = np.zeros((2, 3))
    opt = Adam(p, lr=0.1)
    g = np.array([[1.0, -2.0, 0.5], [0.0, 3.0, -1.0]])
    out = opt.step(g)
    # first Adam step: m_hat = g, v_hat = g^2 -> step = lr * g/(|g| + eps)
    expect = -0.1 * g / (np.abs(g) + 1e-8)
    assert np.allclose(out, expect)
    out2 = opt.step(g)
    assert np.all(np.abs(out2) <= 1.0)
    big = Adam(np.full((1, 1), 0.99), lr=1.0)
    assert big.step(np.array([[-5.0]]))[0, 0] == 1.0          # clipped to bounds (optim.py:21)


def test_momentum():
    p = np.zeros(3)
    opt = Momentum(p, lr=0.5, momentum=0.9)
    g = np.array([1.0, 0.0, -1.0])
    assert np.allclose(opt.step(g), -0.5 * 0.1 * g)
    assert np.allclose(opt.step(g), -0.5 * 0.1 * g - 0.5 * (0.09 + 0.1) * g)


def test_primitives_container_mirrors_reference():
    prims = Primitives(load_scene("Rope", 1).PRIMITIVES)
    assert len(prims) == 3 and prims.action_dim == 6 and prims.state_dim == 21
    assert prims.action_dims == [0, 3, 6, 6]
    d = prims[2].describe()
    assert d["shape"] == "Cylinder" and d["params"] == (0.1, 0.2) and d["action_dim"] == 0
    assert prims[0].describe()["params"] == (0.03,) and prims[0].init_state[3:] == (1.0, 0.0, 0.0, 0.0)
    with pytest.raises(RuntimeError):
        prims[0].get_state(0)                       # not attached to a simulator yet
    with pytest.raises(NotImplementedError):
        Primitive({"shape": "Spatula"}, 0)
    with pytest.raises(AssertionError):             # primitives.py:92: Chopsticks need the 7-dim action
        Primitive({"shape": "Chopsticks"}, 0)
    c = Primitive({"shape": "Chopsticks", "h": 0.2, "r": 0.02, "init_gap": 0.08, "action": {"dim": 7, "scale": (0.02,) * 7}}, 0)
    assert c.state_dim == 8 and len(c.init_state) == 8 and c.init_state[7] == 0.08
    assert c.describe()["params"] == (0.2, 0.02, 0.06)
    with pytest.raises(KeyError):
        Primitive({"shape": "Sphere", "radiuss": 1.0}, 0)      # unknown keys are rejected like yacs does


class _StubEngine:
    def __init__(self):
        self.calls = []
        self.action_dims = [3, 3]

    def __getattr__(self, name):
        def f(*a, **k):
            self.calls.append((name,) + a)
            if name == "loss_forward":
                return dict(loss=1.0, sdf_loss=0.1, density_loss=0.2, contact_loss=0.3, iou=0.5)
            if name == "get_action_grad":
                return np.zeros((a[0], 6))
        return f


def test_tape_replays_in_reverse():
    """Tape records step/loss events and replays loss-grad, step-grad in reverse order (solver.py:36-44)."""
    from plasticinelab_amd.engine.taichi_env import Tape, TaichiEnv
    from plasticinelab_amd.engine.losses import Loss

    class Sim:
        substeps, cur, res, n_grid, dx, dim, n_particles = 19, 0, (64,) * 3, 64, 1 / 64, 3, 10
        primitives = ()

        def __init__(self):
            self.engine = _StubEngine()

        def step(self, is_copy, action=None):
            self.cur += self.substeps

        def grad_begin(self, f):
            self.engine.calls.append(("grad_begin", f))

        def step_grad(self, first, step):
            self.engine.calls.append(("step_grad", first, step))

    env = TaichiEnv.__new__(TaichiEnv)
    env.simulator = Sim()
    env.loss = Loss(type("C", (), {})(), env.simulator)
    env._is_copy, env._tape = False, None
    with Tape(env):
        for _ in range(3):
            env.step(np.zeros(6))
            env.compute_loss()
    calls = [c for c in env.simulator.engine.calls if c[0] in ("grad_begin", "loss_backward", "step_grad")]
    assert calls == [("grad_begin", 57), ("loss_backward", 57), ("step_grad", 38, 2), ("loss_backward", 38),
                     ("step_grad", 19, 1), ("loss_backward", 19), ("step_grad", 0, 0)]
    assert env.loss.loss == 3.0


def test_solver_nn_observation_layout_and_hook_order():
    """SolverNN host logic on a stub engine: observation layout of plb/engine/nn/mlp.py:63-84 (every obs_step-th
    particle: x then velocity_weight * v; then 7 pose numbers per manipulator), the clamp of mlp.py:98, and the order
    in which the reverse sweep pushes the observation adjoint back (right after the adjoint of each env step)."""
    import torch
    from plasticinelab_amd.engine.taichi_env import TaichiEnv
    from plasticinelab_amd.engine.losses import Loss
    from plasticinelab_amd.optimizer.solver_nn import SolverNN
    N, A = 40, 6
    log = []

    class Eng:
        action_dims = [3, 3]

        def get_frame(self, f, want=("x", "v")):
            x = np.arange(N * 3, dtype=float).reshape(N, 3) + 1000 * f
            return {"x": x, "v": -x}

        def get_action_grad(self, n):
            return np.arange(n * A, dtype=float).reshape(n, A) + 1.0

        def add_frame_grad(self, f, xa=None, va=None, Fa=None, Ca=None):
            log.append(("frame_grad", f, np.flatnonzero(np.abs(xa).sum(1) + np.abs(va).sum(1)).tolist()))

        def add_primitive_grad(self, i, f, g):
            log.append(("prim_grad", i, f, len(g)))

        def loss_forward(self, f):
            return dict(loss=1.0, sdf_loss=0.1, density_loss=0.2, contact_loss=0.3, iou=0.5)

        def __getattr__(self, name):
            return lambda *a, **k: log.append((name,) + a)

    class Prim:
        action_dim = 3

        def get_state(self, f):
            return np.full(7, 0.5)

    class Prims(list):
        action_dim = A

        def get_softness(self):
            return 666.0

        def set_softness(self, s):
            pass

    class Sim:
        substeps, cur, res, n_grid, dx, dim, n_particles = 19, 0, (64,) * 3, 64, 1 / 64, 3, N

        def __init__(self):
            self.engine = Eng()
            self.primitives = Prims([Prim(), Prim()])

        def set_state(self, f, state):
            self.cur = 0

        def step(self, is_copy, action=None):
            log.append(("step", self.cur, np.round(np.asarray(action), 6).tolist()))
            self.cur += self.substeps

        def grad_begin(self, f):
            log.append(("grad_begin", f))

        def step_grad(self, first, step):
            log.append(("step_grad", first, step))

    env = TaichiEnv.__new__(TaichiEnv)
    env.simulator = Sim()
    env.primitives = env.simulator.primitives
    env.loss = Loss(type("C", (), {})(), env.simulator)
    env._is_copy, env._tape = False, None
    policy = torch.nn.Linear(10 * 6 + 14, A)
    with torch.no_grad():
        policy.weight.zero_()
        policy.bias.copy_(torch.tensor([2.0, -3.0, 0.25, 0.0, 0.5, -0.5]))
    solver = SolverNN(env, policy, horizon=2, n_observed_particles=10, velocity_weight=0.5)
    assert solver.obs.obs_step == 4 and solver.obs.dim == 74
    o = solver.obs.read(19)
    assert o[:6].tolist() == [19000.0, 19001.0, 19002.0, -9500.0, -9500.5, -9501.0]          # particle 0: x, 0.5 v
    assert o[6:9].tolist() == [19012.0, 19013.0, 19014.0] and o[60:].tolist() == [0.5] * 14     # particle 4; two poses
    loss, grad = solver.forward(None)
    assert loss == 2.0 and grad.shape == (74 * A + A,)
    steps = [e for e in log if e[0] == "step"]
    assert steps[0][2] == [1.0, -1.0, 0.25, 0.0, 0.5, -0.5]                                     # clamped to [-1, 1]
    order = [e[:3] for e in log if e[0] in ("step_grad", "frame_grad", "prim_grad")]
    assert order == [("step_grad", 19, 1), ("frame_grad", 19, []), ("prim_grad", 0, 19), ("prim_grad", 1, 19),
                     ("step_grad", 0, 0), ("frame_grad", 0, []), ("prim_grad", 0, 0), ("prim_grad", 1, 0)]
    # bias gradient = sum over steps of d loss / d action, gated by the clamp (components 0 and 1 are saturated)
    assert np.allclose(grad[-A:], [0.0, 0.0, 3 + 9, 4 + 10, 5 + 11, 6 + 12])


def test_make_without_assets_says_what_to_pass():
    """plb.envs.make(name) needs the reference's target grids (plb/envs/assets/*.npy, not redistributed): the error
    names the two ways to supply one.  Raised before any GPU object is built, so it is checked on the CPU tier."""
    import pytest
    from plasticinelab_amd.envs import make
    with pytest.raises(FileNotFoundError, match="assets_dir"):
        make("Move-v1")


def test_optimizer_knobs_are_writable():
    """lr / bounds are plain attributes in the reference (optim.py:12-13): an lr schedule assigns to them."""
    import numpy as np
    from plasticinelab_amd.optimizer.optim import Adam, Momentum
    p = np.zeros(4)
    o = Adam(p, lr=0.1)
    o.lr = 0.01
    o.bounds = (-0.5, 0.5)
    o.step(np.ones(4))
    assert abs(p[0] + 0.01) < 1e-9 and o.iter == 1 and o.momentum_buffer.shape == (4,) and o.v_buffer.shape == (4,)
    m = Momentum(np.zeros(2))
    m.step(np.ones(2) * 1000)
    assert m.parameters[0] == -1.0 and abs(m.momentum_buffer[0] - 100.0) < 1e-9      # clipped to bounds


def test_scene_strings_cannot_run_code():
    from plasticinelab_amd.config import as_value
    assert as_value("(0.5, 0.25*2, 1/4)") == (0.5, 0.5, 0.25) and as_value("127<<16") == 127 << 16
    assert as_value("__import__('os').system('true')") == "__import__('os').system('true')"
    assert as_value("box") == "box"


def test_loss_check_flags_a_wrong_loss(tmp_path, monkeypatch):
    """loss_check compares with the committed single-GPU loss of the same (workload, dtype, steps) and a difference beyond 1e-5
    is a MISMATCH (pure host logic of bench.py, exercised here with a doctored table)."""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    table = tmp_path / "n1.json"
    table.write_text(json.dumps({"config3_cube128|f32|20": 730.97584}))
    monkeypatch.setattr(bench, "N1_LOSS_FILE", str(table))
    ok = bench.loss_check("config3_cube128", "f32", 20, 730.97583)
    bad = bench.loss_check("config3_cube128", "f32", 20, 731.2)
    none = bench.loss_check("config3_cube128", "f32", 7, 1.0)
    assert ok["ok"] is True and ok["rel"] < 1e-7 and bad["ok"] is False and bad["rel"] > 1e-4 and none["ok"] is None and none["n1_expected"] is None


def test_bench_secondary_points_and_reference_records(tmp_path, monkeypatch):
    """Host logic of bench.py's secondary points: the configs[3] / configs[4] recipes keep ~8 particles per cell at any test scale,
    the workload names they are filed under, and the committed single-GPU record (value + loss) a slab point is checked against."""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    for pt, n in ((bench.point_config4(), 256), (bench.point_config5(), 512), (bench.point_config4(0.03), 256), (bench.point_config5(0.01), 512)):
        assert int(128 * pt["quality"] * 0.5) == n
        ppc = pt["particles"] / (pt["side"] * n) ** 3
        assert 7.0 < ppc < 8.5, ppc                                              # the ~8 ppc cube recipe
    assert bench.point_config5()["mixed_yield"] and bench.point_config5()["steps"] == 1 and bench.point_config4()["yield_stress"] == 1e9
    ys = bench.mixed_yield(6)
    assert list(ys) == [50.0, 1e9, 50.0, 1e9, 50.0, 1e9]

    class A:
        workload, particles, quality, side, mixed_yield = "config3_cube128", 500_000, 2, 0.31, False
    assert bench.workload_name(A, 128) == "config3_cube128"
    A.particles, A.quality, A.side = 2_000_000, 4.0, 0.25
    assert bench.workload_name(A, 256) == "cube256_2000000p"
    A.particles, A.quality, A.mixed_yield = 16_000_000, 8.0, True
    assert bench.workload_name(A, 512) == "mixed512_16000000p"
    # the reference records: value + loss from n1_reference_points.json, loss-only entries from the older table
    pts, old = tmp_path / "pts.json", tmp_path / "old.json"
    pts.write_text(json.dumps({"cube256_2000000p|f32|2": {"value": 2000.0, "final_loss": 5.0, "source": "test"}}))
    old.write_text(json.dumps({"config3_cube128|f32|20": 730.97584}))
    monkeypatch.setattr(bench, "N1_POINTS_FILE", str(pts))
    monkeypatch.setattr(bench, "N1_LOSS_FILE", str(old))
    r = bench.n1_reference("cube256_2000000p", "f32", 2)
    assert r["value"] == 2000.0 and r["final_loss"] == 5.0
    assert bench.n1_reference("config3_cube128", "f32", 20) == {"final_loss": 730.97584, "value": None, "file": "profiles/n1_final_loss.json"}
    assert bench.n1_reference("cube256_2000000p", "f32", 3) is None
    assert bench.loss_check("cube256_2000000p", "f32", 2, 5.00001)["ok"] is True and bench.loss_check("cube256_2000000p", "f32", 2, 5.1)["ok"] is False
    # the target grid synthesised on the host: sums to N p_mass whatever the cloud
    x = np.random.default_rng(0).random((5000, 3)) * 0.3 + 0.3
    g = bench.mass_grid(x, 32, 1e-3)
    assert g.shape == (32, 32, 32) and abs(g.sum() - 5.0) < 1e-12 and (g >= 0).all()


def test_checkpointed_slab_cleanup_runs_collectives_only_after_agreed_failures(monkeypatch):
    """optimizer/checkpoint.py on slab ranks: the engine is always left with the population it started with, but the full
    restore re-synchronises the device-side exchange -- a collective -- and may only run where every rank runs it: after a normal
    return or a failure the ranks agreed on (SlabEngine._agree / _check mark those exceptions).  A failure of one rank alone
    restores local state only."""
    from plasticinelab_amd import distributed as D
    from plasticinelab_amd.optimizer import checkpoint as ck

    calls = []

    class Eng:
        def checkpoint(self, f):
            return {"ids": np.arange(5)}

        def reenter(self, c, collective=True):
            calls.append(collective)

    class Prim:
        def get_state(self, f):
            return np.zeros(7)

        def set_state(self, f, s):
            pass

    class Sim:
        engine, n_particles, cur, substeps = Eng(), 5, 0, 3

    class Env:
        simulator, loss, primitives, n_particles = Sim(), None, [Prim()], 5

        def set_state(self, *a):
            pass

    for exc, want in ((RuntimeError("NaN on this rank"), [False]), (D._collective(RuntimeError("agreed")), [True]), (None, [True])):
        calls.clear()

        def sweeps(*a, _exc=exc, **k):
            if _exc is not None:
                raise _exc
            return 1.5, np.zeros((2, 6))
        monkeypatch.setattr(ck, "_checkpointed_slab_sweeps", sweeps)
        if exc is None:
            assert ck._forward_checkpointed_slab(Env(), None, np.zeros((2, 6)), 1, 2, 3, 666.0)[0] == 1.5
        else:
            with pytest.raises(RuntimeError, match=str(exc)):
                ck._forward_checkpointed_slab(Env(), None, np.zeros((2, 6)), 1, 2, 3, 666.0)
        assert calls == want, (exc, calls)


def test_exchange_reset_is_two_phases_with_a_barrier_behind_each():
    """SlabEngine.reset_exchange: drain on every rank, barrier, clear on every rank, barrier -- a single clear +
    barrier could be overwritten by a neighbour whose exchange kernels were still draining.  Host logic only: the order of calls."""
    from plasticinelab_amd import distributed as D
    log = []

    class Eng:
        def halo_peer_reset(self, phase):
            log.append(("reset", phase))

    class Comm:
        peer_mapped = True
        scalar_device = "cpu"

        def all_reduce_(self, t, op=None):
            log.append(("barrier",))

    se = D.SlabEngine.__new__(D.SlabEngine)
    se._e, se.comm = Eng(), Comm()
    se.reset_exchange()
    assert log == [("reset", 0), ("barrier",), ("reset", 1), ("barrier",)]
    log.clear()
    se.comm.peer_mapped = False                     # point-to-point transport: nothing to re-synchronise
    se.reset_exchange()
    assert log == []


def test_grid_workgroups_of_engines_built_for_the_fused_exchange():
    """Every grid workgroup of every rank on a GPU must be resident while the fused exchange + grid kernels wait for the neighbours:
    512 fit a GPU; the cap leaves a margin and is a power of two (the flag layout of the grid kernels needs one)."""
    from plasticinelab_amd.distributed import fused_grid_workgroups as cap
    assert cap(False, 8, 1) == 0 and cap(False, 8, 8) == 0 and cap(True, 1, 1) == 0          # exchange kernels / one rank: the default
    assert cap(True, 8, 1) == 256 and cap(True, 2, 1) == 256                                 # one rank per GPU
    assert [cap(True, n, n) for n in (2, 3, 4, 5, 8, 16, 64)] == [128, 64, 64, 32, 32, 16, 8]
    for n in (2, 3, 4, 5, 8, 16):
        c = cap(True, n, n)
        assert c & (c - 1) == 0 and n * c <= 256


def test_bench_valu_roof_arithmetic(tmp_path, monkeypatch):
    """bench.py's secondary (vector-ALU) roof: issue time of a kernel = waves per SIMD x sum(class count x class cycles) / clock,
    from a calibration file of the same workload and dtype; null otherwise."""
    import json
    import bench
    cal = {"workload": "config3_cube128", "dtype": "f32", "clock_ghz": 2.0, "simds": 1024, "source": "synthetic",
           "cycles_per_wave_instruction": {"fma_f32": 4.0, "mul_add_f32": 2.0, "default": 3.0},
           "kernels": {"g2p_p2g": {"mix_per_wave": {"fma_f32": 1000, "mul_add_f32": 500, "other": 100}},
                       "grid_op": {"mix_per_wave": {"fma_f32": 10}, "waves": 2048}}}
    f = tmp_path / "cal.json"
    f.write_text(json.dumps(cal))
    monkeypatch.setattr(bench, "VALU_FILE", str(f))
    kernels = {"g2p_p2g": {"avg_us": 50.0, "launches": 10}, "grid_op": {"avg_us": 7.0, "launches": 10}, "p2g_grad": {"avg_us": 45.0, "launches": 10}}
    r = bench.valu_roof(kernels, 512_000, "config3_cube128", "f32", 10)
    waves = 512_000 // 64
    cyc = 1000 * 4.0 + 500 * 2.0 + 100 * 3.0
    want = (waves / 1024) * cyc / 2.0e3
    assert abs(r["kernels"]["g2p_p2g"]["issue_us"] - want) < 1e-9 and abs(r["kernels"]["g2p_p2g"]["frac_of_kernel_time"] - want / 50.0) < 1e-12
    want_g = (2048 / 1024) * 40.0 / 2.0e3
    assert abs(r["kernels"]["grid_op"]["issue_us"] - want_g) < 1e-12 and "p2g_grad" not in r["kernels"]
    assert abs(r["issue_us_per_substep"] - (want + want_g)) < 1e-9 and abs(r["frac"] - (want + want_g) / 57.0) < 1e-12
    assert bench.valu_roof(kernels, 512_000, "cube256_2000000p", "f32", 10) is None        # another workload: no stale constant
    monkeypatch.setattr(bench, "VALU_FILE", str(tmp_path / "missing.json"))
    assert bench.valu_roof(kernels, 512_000, "config3_cube128", "f32", 10) is None


def test_valu_calibration_tool_feeds_the_bench_line(tmp_path, monkeypatch):
    """profiles/tools/valu_calibration.py: the microbenchmark's table + the per-kernel SQ_INSTS_VALU_* counters -> the JSON bench.py's
    roofline.valu reads.  Synthetic inputs in the formats the GPU session writes (profiles/tools/r06_session.sh): the parse, the
    class -> stream pricing, the per-wave mix, the kernel-name mapping and the hand-off to bench.valu_roof."""
    import json
    import os
    import sys
    from tests.util import ROOT
    sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))
    import bench
    import valu_calibration as vc

    def row(name, *cyc):
        return f"{name:<58s}" + "".join(f"  W={w}: {c:5.2f} cyc @{1.93:4.2f} GHz" for w, c in zip((1, 2, 4, 8), cyc)) + "\n"

    names = [s for _c, streams in vc.CLASSES.values() for s in streams] + vc.DEFAULT_STREAMS
    text = "# gfx950, 256 CUs; header line\n" + "".join(row(n, 5.7, 4.2, 4.0 if "fma_f32" in n else 2.0, 4.0) for n in dict.fromkeys(names))
    table = vc.parse_table(text)
    assert len(table) == len(set(names)) and table["v_mul_f32 d,d,m"][4] == (2.0, 1.93) and table["v_rcp_f32"][1] == (5.7, 1.93)
    waves = 7813.0
    pmc = {"k_g2p_p2g<float, false>": {"calls": 10, "SQ_WAVES": waves, "SQ_INSTS_VALU": 3000 * waves, "SQ_INSTS_VALU_FMA_F32": 1000 * waves,
                                        "SQ_INSTS_VALU_MUL_F32": 400 * waves, "SQ_INSTS_VALU_INT32": 600 * waves, "SQ_INSTS_VALU_CVT": 70 * waves},
           "k_g2p_p2g<double, false>": {"calls": 10, "SQ_WAVES": waves, "SQ_INSTS_VALU": 1.0},
           "k_grid_op<float, false>": {"calls": 10, "SQ_WAVES": 2048.0, "SQ_INSTS_VALU": 2048.0 * 900, "SQ_INSTS_VALU_FMA_F64": 2048.0 * 500},
           "__amd_rocclr_copyBuffer": {"calls": 3, "SQ_WAVES": 10.0, "SQ_INSTS_VALU": 100.0}}
    cal = vc.build(table, pmc, "config3_cube128", "f32", 4, "synthetic")
    assert set(cal["kernels"]) == {"g2p_p2g", "grid_op"} and abs(cal["clock_ghz"] - 1.93) < 1e-12
    mix = cal["kernels"]["g2p_p2g"]["mix_per_wave"]
    assert abs(mix["fma_f32"] - 1000) < 1e-9 and abs(mix["other"] - (3000 - 1000 - 400 - 600 - 70)) < 1e-9
    assert cal["cycles_per_wave_instruction"]["fma_f32"] == 4.0 and cal["cycles_per_wave_instruction"]["mul_f32"] == 2.0
    f = tmp_path / "cal.json"
    f.write_text(json.dumps(cal))
    monkeypatch.setattr(bench, "VALU_FILE", str(f))
    r = bench.valu_roof({"g2p_p2g": {"avg_us": 50.0, "launches": 38}, "grid_op": {"avg_us": 7.0, "launches": 39}}, 500_000, "config3_cube128", "f32", 39)
    cycles = 1000 * 4.0 + (3000 - 1000) * 2.0                   # every other class and the default are 2.0 in the synthetic table
    want = (waves / 1024) * cycles / (1.93e3)
    assert abs(r["kernels"]["g2p_p2g"]["issue_us"] - want) < 1e-9 and 0 < r["frac"] < 1
