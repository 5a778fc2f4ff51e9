"""-m gpu: error behaviour of the boundary on a live engine -- misuse returns a status and a message (the Python layer raises
EngineError with it), never a crash, and the engine stays usable afterwards.  The reference's counterparts are Python asserts and
index errors on Taichi fields (mpm_simulator.py:8, 33-38; primitives.py:291; taichi_env.py:61,84,92)."""
import numpy as np
import pytest

from tests.util import O, oracle_scene
from tests.gpu_util import engine_for, load_state

pytestmark = pytest.mark.gpu


def test_misuse_is_reported_and_the_engine_stays_usable():
    from plasticinelab_amd._lib import EngineError
    _, sim, prims, x0 = oracle_scene("Move", 1, n_particles=500)
    eng = engine_for(sim, prims, dtype="float32", max_frames=8)
    load_state(eng, 0, O.init_state(x0), O.materials(sim), O.init_poses(prims))
    F = 8
    with pytest.raises(EngineError, match="out of range"):
        eng.substep(F)                                                  # frames 0 .. F-1 can be stepped from
    with pytest.raises(EngineError, match="exceed max_frames"):
        eng.step(0, F + 1)
    with pytest.raises(EngineError, match="out of range"):
        eng.get_frame(F + 1)
    with pytest.raises(EngineError, match="not resident"):
        eng.substep_grad(0)                                             # no grad_begin: no adjoint frame to start from
    with pytest.raises(EngineError, match="not resident"):
        eng.get_frame_grad(3)
    eng.set_action(0, 4, np.zeros(6))
    eng.step(0, 4)
    eng.grad_begin(4)
    with pytest.raises(EngineError, match="not resident"):
        eng.substep_grad(1)                                             # out of order: frame 2's adjoint does not exist yet
    with pytest.raises(EngineError, match="full state"):
        eng.set_frame(0, x=np.asarray(x0), resort=True)                 # a re-sort needs x, v, F and C
    with pytest.raises(EngineError, match="slab engines only"):
        eng.set_population(10)
    # ... and after all that the reverse sweep of the four substeps still runs and gives finite numbers
    eng.add_frame_grad(4, xa=np.ones((500, 3)))
    for f in (3, 2, 1, 0):
        eng.substep_grad(f)
    g = eng.get_frame_grad(0)
    assert all(np.isfinite(g[k]).all() for k in ("x", "v", "C", "F")) and np.abs(g["x"]).max() > 0
    assert eng.error_flags() == 0
    eng.close()
