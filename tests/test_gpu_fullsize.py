"""-m gpu: the BASELINE workloads at FULL size -- config 3 (128^3 grid, 500k particles, two manipulators), config 4's
(256^3, 2M particles, elastic) and config 5's (512^3, 16M particles, half sigma_y = 50 / half 1e9, no grid store) on
one MI355X -- too big for the oracle, so parity goes through size-independent properties:

* mass: the loss's mass scatter sums to exactly N * p_mass;
* momentum: one substep in free fall with no manipulator in reach changes the total momentum by m g dt;
* the hand-derived adjoint against central finite differences of the loss along a random action direction
  (float64 engine, two env steps = 78 substeps fwd + bwd, across the storage re-sort cadence);
* fp32 engine against the fp64 engine on the same rollout: loss and action gradient within the north-star 1e-4.
"""
import numpy as np
import pytest

from tests.gpu_util import relerr

pytestmark = pytest.mark.gpu


def build(dtype, monkeypatch=None, resort=None):
    import torch
    import bench
    if monkeypatch is not None and resort is not None:
        monkeypatch.setenv("PLMPM_RESORT_STEPS", str(resort))

    class A:
        particles, quality, steps, warmup = 500_000, 2, 2, 0
    A.dtype = dtype
    env, _ = bench.build_env(A, torch.device("cuda", 0))
    return env, bench


def test_mass_and_momentum_full_size():
    env, bench = build("float32")
    sim = env.simulator
    eng = sim.engine
    gm = eng.grid_mass(0)
    assert abs(gm.sum() / (sim.n_particles * sim.p_mass) - 1.0) < 1e-5
    nodes, blocks = eng.grid_stats(0)
    assert nodes == int((gm > 0).sum()) and blocks > 0
    # one substep from rest: every particle (all clear of the floor and of the manipulators' reach is not required:
    # the spheres start outside the cube) gains dt * g * 30 in v (grid_op's gravity term)
    st = env.get_state()["state"]
    env.set_state(st, 666.0, False)
    eng.substep(0)
    v1 = eng.get_frame(1, want=("v",))["v"]
    g = np.asarray(sim.default_gravity, float) * 30.0 * sim.dt
    far = np.ones(len(v1), bool)
    for p in env.primitives:
        c = np.asarray(p.cfg.init_pos, float)
        far &= np.linalg.norm(st[0] - c, axis=1) > 0.12
    assert far.sum() > 100_000
    assert np.abs(v1[far] - g).max() < 1e-6 * max(np.abs(g).max(), 1e-9) + 2e-7


def test_adjoint_matches_finite_differences_full_size(monkeypatch):
    env, bench = build("float64", monkeypatch, resort=1)          # re-sort before the second env step
    A = env.primitives.action_dim
    acts = bench.seeded_actions(2, A)
    st = env.get_state()["state"]

    def run(a, grad):
        env.set_state(st, 666.0, False)
        if grad:
            loss = bench.rollout(env, a)
            return loss, env.primitives.get_grad(len(a))
        env.loss.clear_loss()
        for ai in a:
            env.step(ai)
            env.compute_loss()
        return env.loss.loss, None

    loss, g = run(acts, True)
    d = np.random.default_rng(1).standard_normal(acts.shape)
    d /= np.abs(d).max()
    eps = 1e-5                                                      # FD error 2e-4 at 1e-4, 3e-5 from 3e-5 down
    lp, _ = run(acts + eps * d, False)
    lm, _ = run(acts - eps * d, False)
    fd = (lp - lm) / (2 * eps)
    an = float((g * d).sum())
    assert abs(fd) > 0 and abs(an - fd) / abs(fd) < 1e-4, (an, fd)


def test_fp32_engine_tracks_fp64_engine_full_size():
    out = {}
    for dtype in ("float64", "float32"):
        env, bench = build(dtype)
        acts = bench.seeded_actions(2, env.primitives.action_dim)
        st = env.get_state()["state"]
        env.set_state(st, 666.0, False)
        loss = bench.rollout(env, acts)
        out[dtype] = (loss, env.primitives.get_grad(2).copy())
        del env
    l64, g64 = out["float64"]
    l32, g32 = out["float32"]
    assert abs(l32 - l64) / abs(l64) < 1e-4
    assert relerr(g32, g64) < 1e-4


def test_fp32_engine_tracks_fp64_engine_over_the_benchmark_horizon():
    """The rollout `bench.py --steps 20` times -- 20 env steps = 780 substeps forward + reverse, a re-sort every other step, the cube
    squeezed between the two manipulators until part of it yields -- in the fp32 engine against the fp64 engine: loss
    and action gradient within the north-star 1e-4 (measured: 4e-7 and 3e-5; horizons 2 / 5 / 10: 6e-6 / 5e-6 / 2e-5)."""
    import torch
    import bench

    class A:
        particles, quality, steps, warmup = 500_000, 2, 20, 0
    out = {}
    for dtype in ("float64", "float32"):
        A.dtype = dtype
        env, _ = bench.build_env(A, torch.device("cuda", 0))
        acts = bench.seeded_actions(A.steps, env.primitives.action_dim)
        env.set_state(env.get_state()["state"], 666.0, False)
        loss = bench.rollout(env, acts)
        out[dtype] = (float(loss), env.primitives.get_grad(A.steps).copy())
        env.simulator.engine.close()
        del env
        torch.cuda.empty_cache()
    (l64, g64), (l32, g32) = out["float64"], out["float32"]
    print(f"\n[config 3, 20 env steps] loss rel {abs(l32 - l64) / abs(l64):.2e}; action-gradient max-rel {relerr(g32, g64):.2e}")
    assert abs(l32 - l64) / abs(l64) < 1e-5
    assert relerr(g32, g64) < 1e-4


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3]: synthetic elastic block, 256^3 grid, 2M particles (here on one GPU; the z-slab cut of the same
# workload is checked against the single-rank engine in test_gpu_distributed.py at a size the box can run N ranks of)
def test_resort_on_a_256_grid_does_not_change_the_physics(monkeypatch):
    """The counting-sort re-sort at 256^3 (2^24 cells along the curve: its largest single-GPU case; finer grids take the
    radix sort): 200k particles, 3 env steps of 79 substeps with a re-sort before the 2nd and 3rd, float64 engine,
    against the same rollout without re-sorts."""
    import torch
    import bench

    class A:
        particles, quality, steps, warmup, yield_stress, side, dtype = 200_000, 4, 3, 0, 200.0, 0.2, "float64"
    out = {}
    for R in ("0", "1"):
        monkeypatch.setenv("PLMPM_RESORT_STEPS", R)
        env, _ = bench.build_env(A, torch.device("cuda", 0))
        acts = bench.seeded_actions(A.steps, env.primitives.action_dim)
        env.set_state(env.get_state()["state"], 666.0, False)
        loss = bench.rollout(env, acts)
        fr = env.simulator.engine.get_frame(env.simulator.cur, want=("x", "v"))
        out[R] = (np.array([float(loss)]), env.primitives.get_grad(A.steps).copy(), fr["x"], fr["v"])
        env.simulator.engine.close()
        del env
        torch.cuda.empty_cache()
    for a, b in zip(out["0"], out["1"]):
        assert relerr(a, b) < 1e-9


def build4(dtype):
    import torch
    import bench

    class A:
        particles, quality, steps, warmup, yield_stress, side = 2_000_000, 4, 1, 0, 1e9, 0.25
    A.dtype = dtype
    env, _ = bench.build_env(A, torch.device("cuda", 0))
    return env, bench


def test_config4_workload_full_size():
    """256^3 / 2M elastic particles, one whole env step (79 substeps) forward + backward: mass conservation, and the
    fp32 engine against the fp64 engine on loss and action gradient within the north-star 1e-4."""
    out = {}
    for dtype in ("float64", "float32"):
        env, bench = build4(dtype)
        sim = env.simulator
        assert sim.n_grid == 256 and sim.substeps == 79 and sim.n_particles == 2_000_000
        if dtype == "float32":
            gm = sim.engine.grid_mass(0)
            assert abs(gm.sum() / (sim.n_particles * sim.p_mass) - 1.0) < 1e-5
            nodes, blocks = sim.engine.grid_stats(0)
            assert nodes == int((gm > 0).sum())
            del gm
        acts = bench.seeded_actions(1, env.primitives.action_dim)
        st = env.get_state()["state"]
        env.set_state(st, 666.0, False)
        loss = bench.rollout(env, acts)
        out[dtype] = (loss, env.primitives.get_grad(1).copy())
        sim.engine.close()
        del env, sim
    l64, g64 = out["float64"]
    l32, g32 = out["float32"]
    print(f"\n[config 4 workload] loss {l64:.9g}; fp32 vs fp64: loss rel {abs(l32 - l64) / abs(l64):.2e}, grad max-rel {relerr(g32, g64):.2e}")
    assert np.abs(g64).max() > 0
    assert abs(l32 - l64) / abs(l64) < 1e-4
    assert relerr(g32, g64) < 1e-4


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4]: mixed elastic + plastic scene, 512^3 grid, 16M particles.  One rank cannot keep per-frame grids
# of 512^3 (4.3 GB per frame), so this is the recompute schedule (store_grid = 0) on a grid window around the body.
def build5(dtype, n_particles=16_000_000, frames=3):
    from plasticinelab_amd.engine.core import Engine
    n = 512
    dx = 1.0 / n
    side = 0.25
    rng = np.random.default_rng(0)
    x = (rng.random((n_particles, 3)) - 0.5) * side + np.array([0.5, 0.2, 0.5])
    r = 0.05
    # the test runs `frames` substeps as one short "env step": scale the action so that a substep moves a manipulator
    # about as far as in a 159-substep step of the real scene (0.01 / 159 per unit action)
    scale = (0.01 / 159 * (frames - 1),) * 3
    prims = [dict(shape="Sphere", action_dim=3, params=(r,), friction=0.9, action_scale=scale),
             dict(shape="Sphere", action_dim=3, params=(r,), friction=0.9, action_scale=scale)]
    lo = np.floor((x.min(0) - 0.05) * n).astype(int)
    hi = np.ceil((x.max(0) + 0.05) * n).astype(int)
    dt = 0.5e-4 / 4
    eng = Engine(n_grid=n, n_particles=n_particles, max_frames=frames, substeps=159, dt=dt, p_vol=(dx * 0.5) ** 2, p_mass=(dx * 0.5) ** 2,
                 gravity=(0, -1, 0), ground_friction=1.5, primitives=prims, dtype=dtype, store_grid=False,
                 grid_window=(lo, hi), resort_steps=0)
    E, nu = 5000.0, 0.2
    mu, lam = E / (2 * (1 + nu)), E * nu / ((1 + nu) * (1 - 2 * nu))
    ys = np.where(np.arange(n_particles) % 2 == 0, 50.0, 1e9)          # half plastic, half elastic
    F = np.broadcast_to(np.eye(3), (n_particles, 3, 3))
    eng.set_frame(0, x=x, v=np.zeros((n_particles, 3)), F=np.ascontiguousarray(F), C_=np.zeros((n_particles, 3, 3)), resort=True)
    eng.set_materials(mu, lam, ys)
    eng.set_softness(666.0)
    # the two spheres overlap the cube's x faces a little, so contact (and its pose adjoint) is active from the start
    eng.set_primitive_state(0, 0, [0.5 - side / 2 - 0.8 * r, 0.2, 0.5, 1, 0, 0, 0])
    eng.set_primitive_state(1, 0, [0.5 + side / 2 + 0.8 * r, 0.2, 0.5, 1, 0, 0, 0])
    return eng, x


def run5(eng, action, frames, cot):
    """`frames` substeps forward from frame 0 with `action`; L = sum(cot * x[frames]); -> (L, d L / d action or None)."""
    eng.set_action(0, frames, action)
    for f in range(frames):
        eng.substep(f)
    x = eng.get_frame(frames, want=("x",))["x"]
    return x, float((x * cot).sum())


def test_config5_workload_full_size():
    """512^3 grid, 16M particles, half yielding (sigma_y = 50) and half not (1e9), no per-frame grid store, grid window
    around the body: mass, free-fall momentum, the adjoint of two substeps against central finite differences along the
    action, and the fp32 engine against the fp64 one."""
    N, frames = 16_000_000, 2
    rng = np.random.default_rng(1)
    act = np.array([0.9, 0.1, -0.2, -0.9, 0.2, 0.1])
    d = np.array([1.0, 0.3, -0.5, -1.0, 0.2, 0.4])
    res, cot = {}, None
    for dtype in ("float64", "float32"):
        eng, x0 = build5(dtype, N, frames + 1)
        ws = eng.workspace_bytes
        assert ws["grid_bytes"] < 8e9, ws                       # the window, not 512^3 x 7 grids
        sub = slice(None, None, 997)
        if dtype == "float32":
            gm = eng.grid_mass(0)
            assert abs(gm.sum() / (N * (0.5 / 512) ** 2) - 1.0) < 1e-5
            del gm
        if cot is None:                                                  # the same cotangent for both engines
            cot = np.zeros((N, 3))
            touched = np.abs(x0[:, 0] - 0.5) > 0.25 / 2 - 0.02           # particles the spheres push: the loss looks at them
            cot[touched] = rng.standard_normal((int(touched.sum()), 3))
        x2, L = run5(eng, act, frames, cot)
        eng.grad_begin(frames)
        eng.add_frame_grad(frames, xa=cot)
        for f in range(frames - 1, -1, -1):
            eng.substep_grad(f)
        eng.chain_grad(0, frames, 0)                 # pose adjoints -> kinematics chain -> d L / d action
        g = eng.get_action_grad(1)[0]
        v1 = eng.get_frame(1, want=("v",))["v"]
        far = np.abs(x0[:, 0] - 0.5) < 0.05
        gdt = np.array([0.0, -1.0, 0.0]) * 30.0 * (0.5e-4 / 4)
        assert np.abs(v1[far] - gdt).max() < 1e-6 * np.abs(gdt).max() + 2e-7
        res[dtype] = (x2[sub].copy(), L, g.copy())
        if dtype == "float64":
            eps = 1e-2                                            # the action moves a sphere 6e-5 per unit and substep: still the linear regime
            _, Lp = run5(eng, act + eps * d, frames, cot)
            _, Lm = run5(eng, act - eps * d, frames, cot)
            fd = (Lp - Lm) / (2 * eps)
            an = float((g * d).sum())
            print(f"\n[config 5 workload] L {L:.9g}; adjoint along the action {an:.9g} vs central differences {fd:.9g}")
            assert abs(fd) > 0 and abs(an - fd) / abs(fd) < 1e-4, (an, fd)
        eng.close()
        del eng
    x64, L64, g64 = res["float64"]
    x32, L32, g32 = res["float32"]
    print(f"[config 5 workload] fp32 vs fp64: x max abs {np.abs(x32 - x64).max():.2e}, L rel {abs(L32 - L64) / abs(L64):.2e}, grad max-rel {relerr(g32, g64):.2e}")
    assert np.abs(x32 - x64).max() < 1e-7
    assert abs(L32 - L64) / abs(L64) < 1e-4
    assert relerr(g32, g64) < 1e-4


@pytest.mark.parametrize("workload", ["triplemove128", "rope128"])
def test_reference_scene_geometries_full_size(workload):
    """BASELINE configs[2] as it NAMES the scenes: TripleMove-v1 (three boxes, six Sphere manipulators,
    plb/envs/triplemove.yml:3-64) and Rope-v1 (a bar, two Spheres, a static Cylinder, ground_friction 0.3,
    plb/envs/rope.yml:1-32) on the 128^3 grid with ~500k particles (`bench.py --workload ...`).  Same size-independent
    properties as the synthetic cube: mass = N p_mass; the fp32 engine against the fp64 engine on a two-env-step rollout
    (78 substeps fwd + bwd): loss and action gradient within the north star's 1e-4; the adjoint against central finite
    differences along a random action direction.

    The finite differences need contact_min_adjoint = "argmin": in Rope-v1 the manipulators start clear of the bar
    (min_dist ~ 0.03), and the reference's gradient of the hard contact loss -- ti.atomic_min differentiated as an add
    (SURVEY Q10), the engine's default -- hands min_dist's adjoint to EVERY particle: d loss / d action ~ -90 where the
    derivative of the minimum itself is ~2e-5 (measured).  A finite difference sees the latter."""
    import torch
    import bench

    class A:
        particles, quality, steps, warmup = 500_000, 2, 2, 0
    A.workload = workload
    orig = bench.scene_cfg

    def argmin_cfg(*a, **k):
        c = orig(*a, **k)
        c.SIMULATOR["contact_min_adjoint"] = "argmin"
        return c
    bench.scene_cfg = argmin_cfg
    try:
        _reference_scene_body(A, workload, bench, torch)
    finally:
        bench.scene_cfg = orig


def _reference_scene_body(A, workload, bench, torch):
    out = {}
    for dtype in ("float64", "float32"):
        A.dtype = dtype
        env, _ = bench.build_env(A, torch.device("cuda", 0))
        sim = env.simulator
        assert len(env.primitives) == (6 if workload == "triplemove128" else 3) and abs(sim.n_particles - 500_000) <= 3
        if dtype == "float32":
            gm = sim.engine.grid_mass(0)
            assert abs(gm.sum() / (sim.n_particles * sim.p_mass) - 1.0) < 1e-5
        acts = bench.seeded_actions(2, env.primitives.action_dim)
        st = env.get_state()["state"]
        env.set_state(st, 666.0, False)
        loss = bench.rollout(env, acts)
        grad = env.primitives.get_grad(2).copy()
        out[dtype] = (loss, grad)
        if dtype == "float64":                                      # finite differences, forward only
            d = np.random.default_rng(1).standard_normal(acts.shape)
            d /= np.abs(d).max()
            eps, fl = 1e-5, []
            for sgn in (1, -1):
                env.set_state(st, 666.0, False)
                env.loss.clear_loss()
                for ai in acts + sgn * eps * d:
                    env.step(ai)
                    env.compute_loss()
                fl.append(env.loss.loss)
            fd, an = (fl[0] - fl[1]) / (2 * eps), float((grad * d).sum())
            assert abs(fd) > 0 and abs(an - fd) / abs(fd) < 2e-4, (an, fd)
        env.simulator.engine.close()
        del env
        torch.cuda.empty_cache()
    (l64, g64), (l32, g32) = out["float64"], out["float32"]
    print(f"\n[{workload}] loss {l64:.9g}; fp32 vs fp64: loss {abs(l32 - l64) / abs(l64):.2e}, action gradient {relerr(g32, g64):.2e}")
    assert abs(l32 - l64) / abs(l64) < 1e-4
    assert relerr(g32, g64) < 1e-4
