"""CPU tier: the C / OpenMP restatement of one substep and its reverse (oracle/mpm_substep_omp.c -- bench.py's CPU
baseline) against the torch oracle (oracle/plb_oracle.py, autograd): forward state and every adjoint, including the
manipulator position adjoints, with soft and hard contact, yielding and elastic particles, and the three floor-friction
branches of grid_op (mpm_simulator.py:200-219)."""
import numpy as np
import pytest
import torch

from oracle import plb_oracle as O
from oracle.omp_substep import OmpSubstep


def scene(softness, ground_friction, near_floor, seed, n=400):
    rng = np.random.default_rng(seed)
    sim = O.SimCfg(n_particles=n, quality=0.5, yield_stress=8.0, E=5000.0, nu=0.2, ground_friction=ground_friction)   # 32^3
    c = np.array([0.5, 0.08 if near_floor else 0.4, 0.5])
    x = c + (rng.random((n, 3)) - 0.5) * np.array([0.2, 0.1 if near_floor else 0.2, 0.2])
    v = rng.standard_normal((n, 3)) * 0.3
    if near_floor:
        v[:, 1] = -np.abs(v[:, 1]) - 0.5              # into the floor: the boundary branches fire
    Cm = rng.standard_normal((n, 3, 3)) * 2.0
    F = np.eye(3) + rng.standard_normal((n, 3, 3)) * 0.05      # some particles yield (sigma_y = 8), some do not
    ys = np.where(np.arange(n) % 3 == 0, 2e3, 8.0)              # (1e9 makes the torch oracle's unused yield branch overflow: 0 * inf in its autograd)
    prims = [O.PrimCfg(shape="Sphere", radius=0.06, init_pos=(0.42, float(c[1]), 0.5), friction=0.9, action_dim=3, action_scale=(0.01,) * 3),
             O.PrimCfg(shape="Sphere", radius=0.05, init_pos=(0.6, float(c[1]) + 0.02, 0.52), friction=0.4, action_dim=3, action_scale=(0.01,) * 3)]
    pos = np.array([p.init_pos for p in prims])
    pos1 = pos + rng.standard_normal((2, 3)) * 2e-5
    return sim, prims, x, v, Cm, F, ys, pos, pos1


@pytest.mark.parametrize("softness,gf,near_floor", [(666.0, 1.5, False), (0.0, 1.5, False), (666.0, 1.5, True), (0.0, 0.0, True), (666.0, 20.0, True)])
def test_omp_substep_matches_torch_oracle(softness, gf, near_floor):
    sim, prims, x, v, Cm, F, ys, pos, pos1 = scene(softness, gf, near_floor, seed=3)
    n = sim.n_particles
    mu, lam = np.full(n, sim.mu), np.full(n, sim.lam)
    T = lambda a: torch.tensor(a, dtype=O.DT, requires_grad=True)
    state = (T(x), T(v), T(Cm), T(F))
    rot = torch.tensor([1.0, 0.0, 0.0, 0.0], dtype=O.DT)
    pf, pf1 = [T(p) for p in pos], [T(p) for p in pos1]
    out = O.substep(sim, prims, softness, state, (torch.tensor(mu), torch.tensor(lam), torch.tensor(ys)),
                    [(p, rot) for p in pf], [(p, rot) for p in pf1])
    rng = np.random.default_rng(9)
    cot = [rng.standard_normal(tuple(o.shape)) for o in out]
    obj = sum((o * torch.tensor(cc)).sum() for o, cc in zip(out, cot))
    grads = torch.autograd.grad(obj, list(state) + pf + pf1)
    omp = OmpSubstep(sim.n_grid, sim.dt, sim.p_vol, sim.p_mass, sim.gravity, gf, softness, [p.radius for p in prims],
                     [p.friction for p in prims], n)
    fwd = omp.forward(pos, pos1, x, v, Cm, F, mu, lam, ys)
    for a, b, name in zip(fwd, out, "x v C F".split()):
        err = np.abs(a - b.detach().numpy()).max() / max(np.abs(b.detach().numpy()).max(), 1e-300)
        assert err < 1e-11, (name, err)
    (xa, va, Ca, Fa), pa, p1a = omp.backward(pos, pos1, x, v, Cm, F, mu, lam, ys, *cot)
    for a, b, name in zip((xa, va, Ca, Fa), grads[:4], "x.grad v.grad C.grad F.grad".split()):
        b = b.numpy()
        err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
        assert err < 1e-9, (name, err)
    ref_pa, ref_p1a = np.array([g.numpy() for g in grads[4:6]]), np.array([g.numpy() for g in grads[6:8]])
    scale = max(np.abs(ref_pa).max(), np.abs(ref_p1a).max())
    assert scale > 0, "the case was meant to have particles in contact with the manipulators"
    assert np.abs(pa - ref_pa).max() < 1e-9 * scale and np.abs(p1a - ref_p1a).max() < 1e-9 * scale
    # threads do not change the result beyond the summation order of the atomic scatters
    omp.threads(1)
    (xa1, _, _, _), _, _ = omp.backward(pos, pos1, x, v, Cm, F, mu, lam, ys, *cot)
    assert np.abs(xa1 - xa).max() < 1e-9 * max(np.abs(xa).max(), 1e-300)
