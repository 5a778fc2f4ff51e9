"""TEST INFRASTRUCTURE: bench.py's own control flow -- argument handling, build, warm-up, timed repetitions, roofline pass, JSON line --
on the CPU interpreter of the device source, at a size the interpreter finishes in seconds.  bench.py is what the driver runs unattended
at the end of a round; a NameError in a branch of it costs the round's number, and without a GPU nothing else executes it.  The timings
it prints here mean nothing (tests/test_emul_tier.py checks the keys and the loss, not the values).

    python tests/bench_on_interpreter.py <bench.py arguments>
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ["PLMPM_TEST_INTERPRETER"] = "1"
    import torch
    from tests import emul_engine
    import plasticinelab_amd.engine.core as core
    import plasticinelab_amd.engine.mpm_simulator as ms
    ms.Engine = core.Engine = emul_engine.HostEngine
    # the handful of torch.cuda calls of bench.py's World (device choice, synchronize, cache release)
    torch.cuda.device_count = lambda: 1
    torch.cuda.set_device = lambda *_a, **_k: None
    torch.cuda.synchronize = lambda *_a, **_k: None
    torch.cuda.empty_cache = lambda: None
    import bench
    bench.__file__ = os.path.abspath(__file__)          # the child processes bench.py starts (single-GPU points of an N > 1 line) come here too
    bench.main()


if __name__ == "__main__":
    main()
