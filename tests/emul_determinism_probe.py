"""Helper of tests/test_emul_tier.py: the 3-step / 2000-particle golden rollout on the CPU interpreter of the device source, forward and
reverse, on the deterministic (integer-limb) or the floating-point-atomics engine; prints a digest of loss, action gradient and the final
x / v / F.  PLMPM_EMUL_SHUFFLE=<seed> (read by the interpreter) changes the order in which workgroups and lanes run: the deterministic
engine must print the SAME digest under every order, the floating-point-atomics engine need not.

    python -m tests.emul_determinism_probe float32 1|0
"""
import hashlib
import os
import sys

import numpy as np


def main():
    dtype, det = sys.argv[1], sys.argv[2] == "1"
    os.environ["PLMPM_TEST_INTERPRETER"] = "1"
    from tests import emul_engine
    import plasticinelab_amd.engine.mpm_simulator as ms
    ms.Engine = emul_engine.HostEngine
    from tests.test_gpu_deterministic import rollout
    from tests.test_gpu_rollout import make_env_sub
    from tests.util import GOLDEN
    g = np.load(os.path.join(GOLDEN, "rollout_small.npz"))
    env = make_env_sub("Move", int(g["n_particles"]), dtype, deterministic=det)
    out = rollout(env, g["actions"], env.get_state()["state"])
    h = hashlib.sha256()
    for a in out:
        h.update(np.ascontiguousarray(a).tobytes())
    rel = abs(float(out[0]) - float(g["loss"])) / abs(float(g["loss"]))
    print(f"DIGEST {h.hexdigest()} loss_rel {rel:.3e}")


if __name__ == "__main__":
    main()
