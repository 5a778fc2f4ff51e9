#!/usr/bin/env python
"""Generate Taichi-produced golden vectors for the hot path -- the missing pin of the oracle.

THIS SCRIPT DOES NOT RUN IN THE BUILD CONTAINER OR ON THE GPU BOX: it needs the reference checkout and the Taichi it was
written for.  Run it once, anywhere with a network, and commit the three .npz files it writes next to this script.

The environment that reproduces the reference's own run -- cell 1 of plb/optimizer/long_term_gradient.ipynb prints
"[Taichi] version 0.7.14, llvm 10.0.0, commit 58feee37, linux, python 3.7.3" -- in one go:

    conda create -n plb-pin python=3.7 -y && conda activate plb-pin        # taichi 0.7.14 ships wheels for CPython 3.6 - 3.8
    pip install taichi==0.7.14 "numpy<1.22" scipy pyyaml yacs "gym==0.17.3" opencv-python
    #   (setup.py:3 of the reference lists scipy, numpy, torch, opencv-python, tqdm, taichi, gym, tensorboard, yacs, baselines,
    #    all unpinned; `import plb` pulls in plb.envs only -- gym, yaml, yacs, numpy, and through plb.engine taichi, cv2 and
    #    scipy; torch, tqdm, tensorboard and baselines are imported by plb/algorithms and the logger alone, which this script
    #    never touches -- leave them out)
    TI_ARCH=x64 python tests/golden/make_taichi_golden.py /path/to/PlasticineLab

`TI_ARCH=x64` matters: plb/engine/taichi_env.py:6 runs `ti.init(arch=ti.gpu, debug=False, fast_math=True)` at import, and the
pin is BASELINE configs[0]'s backend, the Taichi CPU backend (the variable overrides the arch argument in 0.7.x; without it
and without a CUDA device Taichi falls back to x64 by itself and says so).  The simulator's fields are float64
(mpm_simulator.py:8 `dtype = ti.f64`) whatever the arch.  `fast_math=True` stays as the reference sets it: it is part of what
the reference computes (LLVM fast-math flags on the generated kernels; the oracle's tolerances of 1e-9 leave room for it).
The three files together are < 2 MB.  If they disagree with the oracle on the Q10 semantics, the fix is a flag flip
(`plmpm_config.contact_min_adjoint` / `minmax_tie`, `oracle/plb_oracle.py::SEMANTICS`) plus regenerated `rollout_*.npz`.

It drives the reference through its OWN public surface only (plb.envs.make, TaichiEnv.set_state / step / compute_loss,
ti.Tape, Primitives.get_grad, MPMSimulator.substep / substep_grad) -- nothing of the reference is copied here.

  taichi_move_v1.npz       BASELINE config 2: Move-v1, 50 env steps of np.random.default_rng(0).uniform(-1,1,(50,6))*0.01
                           with softness 666 (plb/optimizer/solver.py:31-44): loss, d loss / d actions, final x / v --
                           the same quantities as tests/golden/rollout_move_v1.npz (oracle-made).
  taichi_substep.npz       one substep from a pre-rolled Move-v1 state and its adjoint for seeded cotangents on
                           x, v, C, F of the next frame (hard AND soft contact loss do not enter here; the contact of
                           grid_op does): all inputs and outputs, so any engine can replay it.
  taichi_semantics.npz     the three autodiff rules the oracle assumes: adjoint routing of max / min on an
                           exact tie, atomic_min differentiated as an add, and the hard / soft contact-loss gradient of
                           loss.py:116-135 on Move-v1 after 3 env steps.

tests/test_taichi_golden.py consumes these files when they exist and reports "parity unpinned" when they do not.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main(ref_root):
    sys.path.insert(0, ref_root)
    import taichi as ti
    from plb.envs import make

    # ---------------------------------------------------------------- Move-v1 rollout (config 2)
    env = make("Move-v1")
    env.reset()
    te = env.taichi_env
    sim = te.simulator
    state = te.get_state()["state"]
    actions = np.random.default_rng(0).uniform(-1, 1, (50, te.primitives.action_dim)) * 0.01

    def forward(action, softness=666.0):
        te.set_state(state, softness, False)
        with ti.Tape(loss=te.loss.loss):
            for a in action:
                te.step(a)
                te.compute_loss()
        return te.loss.loss[None], te.primitives.get_grad(len(action))

    loss, grad = forward(actions)
    f = sim.cur
    np.savez_compressed(os.path.join(HERE, "taichi_move_v1.npz"), taichi_version=np.array(ti.__version__), actions=actions,
                        loss=np.array(loss), grad=grad, x_final=sim.get_x(f), v_final=sim.get_v(f),
                        n_particles=np.array(sim.n_particles), n_grid=np.array(sim.n_grid), substeps=np.array(sim.substeps))
    print("taichi_move_v1.npz: loss", loss, "|grad|max", np.abs(grad).max())

    # ---------------------------------------------------------------- one substep and its adjoint
    # pre-roll 3 env steps in copy mode with pushing actions so that C, F are non-trivial and contact is active
    te.set_state(state, 666.0, True)
    push = np.array([0.9, 0.2, -0.3, -0.9, 0.1, 0.4])
    for _ in range(3):
        te.step(push)
    pre = te.get_state()["state"]
    te.set_state(pre, 666.0, False)
    te.primitives.set_action(0, sim.substeps, push)             # fills v, w of the step's frames
    rng = np.random.default_rng(1)
    cot = [rng.standard_normal((sim.n_particles, 3)), rng.standard_normal((sim.n_particles, 3)),
           rng.standard_normal((sim.n_particles, 3, 3)), rng.standard_normal((sim.n_particles, 3, 3))]

    @ti.kernel
    def seed(xa: ti.ext_arr(), va: ti.ext_arr(), Ca: ti.ext_arr(), Fa: ti.ext_arr()):
        for p in range(sim.n_particles):
            for i in ti.static(range(3)):
                sim.x.grad[1, p][i] = xa[p, i]
                sim.v.grad[1, p][i] = va[p, i]
                for j in ti.static(range(3)):
                    sim.C.grad[1, p][i, j] = Ca[p, i, j]
                    sim.F.grad[1, p][i, j] = Fa[p, i, j]

    @ti.kernel
    def read(xa: ti.ext_arr(), va: ti.ext_arr(), Ca: ti.ext_arr(), Fa: ti.ext_arr()):
        for p in range(sim.n_particles):
            for i in ti.static(range(3)):
                xa[p, i] = sim.x.grad[0, p][i]
                va[p, i] = sim.v.grad[0, p][i]
                for j in ti.static(range(3)):
                    Ca[p, i, j] = sim.C.grad[0, p][i, j]
                    Fa[p, i, j] = sim.F.grad[0, p][i, j]

    with ti.Tape(loss=te.loss.loss):                            # clears every .grad field
        pass
    sim.substep(0)
    out = sim.get_state(1)
    seed(*cot)
    sim.substep_grad(0)
    got = [np.zeros_like(c) for c in cot]
    read(*got)
    pose_grad = [(np.array(p.position.grad[0].value if hasattr(p.position.grad[0], "value") else p.position.grad[0]),
                  np.array(p.position.grad[1].value if hasattr(p.position.grad[1], "value") else p.position.grad[1]))
                 for p in te.primitives]
    np.savez_compressed(os.path.join(HERE, "taichi_substep.npz"), taichi_version=np.array(ti.__version__), action=push,
                        x=pre[0], v=pre[1], F=pre[2], C=pre[3], prim=np.array(pre[4:]),
                        x1=out[0], v1=out[1], F1=out[2], C1=out[3], prim1=np.array(out[4:]),
                        cot_x=cot[0], cot_v=cot[1], cot_C=cot[2], cot_F=cot[3],
                        xa=got[0], va=got[1], Ca=got[2], Fa=got[3],
                        pos_grad0=np.array([g[0] for g in pose_grad]), pos_grad1=np.array([g[1] for g in pose_grad]))
    print("taichi_substep.npz written")

    # ---------------------------------------------------------------- autodiff semantics
    a = ti.field(ti.f64, shape=(), needs_grad=True)
    b = ti.field(ti.f64, shape=(), needs_grad=True)
    vals = ti.field(ti.f64, shape=4, needs_grad=True)
    mn = ti.field(ti.f64, shape=(), needs_grad=True)
    out_f = ti.field(ti.f64, shape=(), needs_grad=True)

    @ti.kernel
    def k_max():
        out_f[None] = ti.max(a[None], b[None])

    @ti.kernel
    def k_min():
        out_f[None] = ti.min(a[None], b[None])

    @ti.kernel
    def k_atomic_min():
        for i in range(4):
            ti.atomic_min(mn[None], vals[i])

    @ti.kernel
    def k_sq():
        out_f[None] = mn[None] ** 2

    sem = {}
    for name, kern in (("max", k_max), ("min", k_min)):
        a[None], b[None] = 0.25, 0.25                           # an exact tie
        with ti.Tape(loss=out_f):
            kern()
        sem[f"{name}_tie_grad"] = np.array([a.grad[None], b.grad[None]])
    vals.from_numpy(np.array([0.7, 0.2, 0.9, 0.2]))
    mn[None] = 100000.0
    with ti.Tape(loss=out_f):
        k_atomic_min()
        k_sq()
    sem["atomic_min_grad"] = vals.grad.to_numpy()
    sem["atomic_min_value"] = np.array(mn[None])
    # the contact loss gradient on the real scene: hard (default) and soft
    for soft in (False, True):
        te.loss.set_weights(sdf=0, density=0, contact=1, is_soft_contact=soft)
        l3, g3 = forward(np.tile(push, (3, 1)))
        sem[f"contact_{'soft' if soft else 'hard'}_loss"], sem[f"contact_{'soft' if soft else 'hard'}_grad"] = np.array(l3), g3
    sem["contact_actions"] = np.tile(push, (3, 1))
    np.savez_compressed(os.path.join(HERE, "taichi_semantics.npz"), taichi_version=np.array(ti.__version__), **sem)
    print("taichi_semantics.npz:", {k: v for k, v in sem.items() if v.size <= 4})


if __name__ == "__main__":
    if len(sys.argv) != 2:
        sys.exit(__doc__)
    main(sys.argv[1])
