"""-m gpu: ragged and degenerate particle sets through one forward substep and its adjoint against the CPU oracle, through the
C ABI -- the sizes and arrangements the wave-level machinery has to get right without help from the workload: a single particle,
one short of / exactly / one past a wave and a workgroup (63 .. 65, 255 .. 257, 513 rows: padding lanes, a workgroup of one
particle), every particle in ONE cell (the longest possible runs of the segmented reduction, every lane adding to the same 27
nodes), every particle alone in its cell (no run longer than one lane), and particles on the walls of the domain (position
clamps of mpm_simulator.py:242 and the boundary branches of grid_op, :209-221).  float64 engine at 1e-9, float32 at 2e-5."""
import dataclasses

import numpy as np
import pytest
import torch

from tests.util import O, oracle_scene
from tests.gpu_util import engine_for, load_state, preroll, relerr

pytestmark = pytest.mark.gpu

TOL = {"float64": 1e-9, "float32": 2e-5}


@pytest.fixture(scope="module")
def rolled():
    cfg, sim, prims, x0 = oracle_scene("Move", 1, n_particles=2000)
    acts = np.zeros((2, 6)); acts[:, 0] = 0.9; acts[:, 3] = -0.9; acts[:, 1] = 0.3
    state, mats, poses = preroll(sim, prims, x0, acts)
    return sim, prims, state, mats, poses, acts


def check(sim, prims, state, mats, poses, action, dtype, ftol=1.0):
    """one substep forward + adjoint of seeded cotangents: engine vs oracle (torch autograd)"""
    n = len(state[0])
    sim = dataclasses.replace(sim, n_particles=n)
    eng = engine_for(sim, prims, dtype=dtype)
    eng.set_softness(666.0)
    load_state(eng, 0, state, mats, poses)
    eng.set_action(0, sim.substeps, action)
    eng.substep(0)
    got = eng.get_frame(1)
    a0 = torch.as_tensor(action, dtype=O.DT)
    vel = [O.set_velocity(p, a0[3 * k:3 * k + 3], sim.substeps) for k, p in enumerate(prims)]
    nxt = [O.forward_kinematics(p, pos, rot, v, w) for p, (pos, rot), (v, w) in zip(prims, poses, vel)]
    sin = tuple(t.clone().requires_grad_(True) for t in state)
    out = O.substep(sim, prims, 666.0, sin, mats, list(poses), nxt)
    tol = TOL[dtype] * ftol
    # C' = 4 inv_dx sum_o w_o v_o (x) dpos_o is a velocity GRADIENT: its natural scale is 4 |v|, whatever is left after the
    # cancellation.  Two particles pressed against a manipulator move almost rigidly (|v| = 4.7, |C| = 0.25): the fp32 engine's C is
    # then 3e-5 off in absolute terms = 1.6e-6 of 4 |v| -- ordinary round-off of a cancelling sum, but 1.3e-4 of max |C|; with more
    # particles somebody's C is O(4 |v|) and the max-norm hides it.  So C is compared on that scale.
    vmax = float(out[1].detach().abs().max())
    for key, ref in zip(("x", "v", "C", "F"), out):
        r = ref.detach().numpy()
        assert got[key].shape == r.shape, (key, n)
        scale = max(np.abs(r).max(), 4 * vmax) if key == "C" else max(np.abs(r).max(), 1e-300)
        assert np.abs(got[key] - r).max() / scale < tol, (key, n)
    g = torch.Generator().manual_seed(n)
    cot = [torch.randn(t.shape, generator=g, dtype=O.DT) for t in out]
    gs = torch.autograd.grad(sum((o * c).sum() for o, c in zip(out, cot)), list(sin))
    eng.grad_begin(1)
    eng.add_frame_grad(1, xa=cot[0].numpy(), va=cot[1].numpy(), Ca=cot[2].numpy(), Fa=cot[3].numpy())
    eng.substep_grad(0)
    ga = eng.get_frame_grad(0)
    for key, ref in zip(("x", "v", "C", "F"), gs):
        assert relerr(ga[key], ref.numpy()) < 5 * tol, (key, n)
    assert eng.error_flags() == 0
    eng.close()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 513])
def test_ragged_particle_counts(rolled, dtype, n):
    sim, prims, state, mats, poses, acts = rolled
    # the rows nearest the left manipulator first, so that even the single particle feels a contact
    order = torch.argsort(((state[0] - poses[0][0]) ** 2).sum(1))[:n]
    sub = tuple(t[order].clone() for t in state)
    m = tuple(t[order].clone() for t in mats)
    check(sim, prims, sub, m, poses, acts[0], dtype)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_every_particle_in_one_cell_and_every_particle_alone(rolled, dtype):
    sim, prims, state, mats, poses, acts = rolled
    n, N = sim.n_grid, 300
    rng = np.random.default_rng(5)
    base = [t[:N].clone() for t in state]
    m = tuple(t[:N].clone() for t in mats)
    # (a) all 300 in the cell whose stencil base is (20, 30, 25): x n - 0.5 in [base, base + 1)
    x = (np.array([20, 30, 25]) + 0.5 + rng.uniform(0.02, 0.98, (N, 3))) / n
    one_cell = (torch.as_tensor(x), base[1], base[2], base[3])
    check(sim, prims, one_cell, m, poses, acts[0], dtype)
    # (b) one particle per cell on a lattice with a stride of two cells: no two lanes share a stencil base
    ii = np.stack(np.meshgrid(np.arange(7), np.arange(7), np.arange(7), indexing="ij"), -1).reshape(-1, 3)[:N]
    x = (np.array([16, 20, 18]) + 2 * ii + 0.5 + rng.uniform(0.1, 0.9, (N, 3))) / n
    alone = (torch.as_tensor(x), base[1], base[2], base[3])
    check(sim, prims, alone, m, poses, acts[0], dtype)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_particles_at_the_walls(rolled, dtype):
    """Bodies resting against the floor, a side wall and the ceiling region with velocities pointing out of the domain: the floor's
    friction branches and the sticky side walls of grid_op, and the position clamp of g2p, on both sides of their conditions."""
    sim, prims, state, mats, poses, acts = rolled
    n, N = sim.n_grid, 192
    rng = np.random.default_rng(9)
    base = [t[:N].clone() for t in state]
    m = tuple(t[:N].clone() for t in mats)
    x = np.empty((N, 3))
    x[:64] = np.column_stack([rng.uniform(0.3, 0.5, 64), rng.uniform(3.0, 4.5, 64) / n, rng.uniform(0.3, 0.5, 64)])          # floor
    x[64:128] = np.column_stack([rng.uniform(3.0, 4.5, 64) / n, rng.uniform(0.2, 0.4, 64), rng.uniform(0.3, 0.5, 64)])       # x = 0 wall
    x[128:] = np.column_stack([rng.uniform(0.3, 0.5, 64), 1.0 - rng.uniform(3.5, 5.0, 64) / n, rng.uniform(0.3, 0.5, 64)])   # below the upper clamp
    v = rng.standard_normal((N, 3)) * 0.5
    v[:64, 1] -= 1.0; v[64:128, 0] -= 1.0; v[128:, 1] += 1.0
    st = (torch.as_tensor(x), torch.as_tensor(v), base[2], base[3])
    check(sim, prims, st, m, poses, acts[0], dtype, ftol=2.0)
