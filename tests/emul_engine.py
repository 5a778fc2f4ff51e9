"""TEST INFRASTRUCTURE ONLY: the engine on the CPU interpreter of the device source.

``tests/host_emul/libplmpm_emul.so`` is plasticinelab_amd/csrc compiled by g++ against ``tests/host_emul/hipemu`` (fibers for
threads, lock-step 64-lane wave operations, the DPP lane maps of the ISA manual, plain-memory "HBM").  ``HostEngine`` is
``plasticinelab_amd.engine.core.Engine`` with that library and host memory behind it, so that the CPU-only test tier can run the
parity tests of the -m gpu tier on the kernels' own source -- tiling, sorting, segmented reductions, block flags, launch logic --
where there is no GPU.  It is a checker (minutes for what the GPU does in milliseconds); nothing in plasticinelab_amd imports this
module or that library, and the product Engine still refuses to start without a ROCm device.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import subprocess

import torch

from plasticinelab_amd import _lib as L
from plasticinelab_amd.engine.core import Engine

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_emul")
# Compile-time variants of the device source that are not the default build (they wait for a timing on the GPU, profiles/r06_notes.md):
# PLMPM_EMUL_VARIANT=<name> runs the tests on that variant's source, so that a variant is at least parity-green before it is timed.
VARIANTS = {
    "": "",
    "bufio": "-DPLB_BUFIO=1",                            # particle arrays behind buffer descriptors
    "pk": "-DPLB_PK_GATHER=3",                           # p2g.grad and g2p gathers on packed pairs
    "pkbuf": "-DPLB_PK_GATHER=3 -DPLB_BUFIO=1",
    # the default source under AddressSanitizer + UndefinedBehaviorSanitizer (make SAN=1); the process that loads it needs
    # LD_PRELOAD=$(gcc -print-file-name=libasan.so) and ASAN_OPTIONS=detect_leaks=0 (tests/test_emul_tier.py sets both)
    "asan": "",
    # ThreadSanitizer (make SAN=thread; LD_PRELOAD=$(gcc -print-file-name=libtsan.so), PLMPM_EMUL_THREADS > 1): races between workgroups
    "tsan": "",
}
VARIANT = os.environ.get("PLMPM_EMUL_VARIANT", "")
_lib = None


def lib_path(variant=VARIANT):
    return os.path.join(HERE, f"libplmpm_emul{'_' + variant if variant else ''}.so")


def build(variant=VARIANT):
    tag = "_" + variant if variant else ""
    subprocess.check_call(["make", "-s", "-j", "6", "-C", HERE, f"OBJDIR=build_emul{tag}", f"OUT=libplmpm_emul{tag}.so",
                           f"EXTRA={VARIANTS[variant]}", f"SAN={1 if variant == 'asan' else 'thread' if variant == 'tsan' else 0}", f"libplmpm_emul{tag}.so"])


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = L.bind(C.CDLL(lib_path()))
    return _lib


class HostEngine(Engine):
    def __init__(self, **kw):
        # a handful of persistent grid workgroups instead of the GPU's 512 (every workgroup of a launch is 256 fiber start-ups here), and
        # never more than the interpreter keeps resident at once (PLMPM_EMUL_THREADS: one OS thread per workgroup up to 16 -- the
        # exchange folded into the grid kernels waits inside the launch for workgroups of the same launch)
        cap = int(os.environ.get("PLMPM_EMUL_GRID_WG", 8))
        kw["grid_workgroups"] = min(int(kw.get("grid_workgroups") or cap), cap)
        super().__init__(**kw)

    def _load_library(self):
        return lib()

    def _open_device(self, device):
        return torch.device("cpu")

    def _device_memory_bytes(self):
        return 64 << 30

    def _device_guard(self):
        return contextlib.nullcontext()

    def _allocate(self, nbytes):
        # garbage, as torch.empty on a GPU is (NaN as a float, -1 as an int: nothing may rely on zeros) -- for workspaces up to 64 MiB;
        # the per-frame grid stores of a 64-frame engine are ~1 GiB, and touching that for every engine of every test is what the
        # tier's time would go into (untouched pages read as zero)
        raw = torch.empty(nbytes + 256, dtype=torch.uint8)
        if nbytes <= (64 << 20) or os.environ.get("PLMPM_EMUL_FILL") == "1":
            raw.fill_(0xff)
        off = (-raw.data_ptr()) % 256                                    # the library wants 256-byte aligned workspaces
        return raw[off:off + nbytes]

    def _stream_handle(self):
        return 0

    def synchronize(self):
        pass


def engine_for(sim, prims, dtype="float64", max_frames=64, svd_grad_clamp=1e-6, **engine_kw):
    """tests.gpu_util.engine_for on the interpreter."""
    from tests import emul
    plist = [dict(shape=p.shape, action_dim=p.action_dim, params=emul.prim_par(p), friction=p.friction,
                  action_scale=p.action_scale, lower_bound=p.lower_bound, upper_bound=p.upper_bound) for p in prims]
    return HostEngine(n_grid=sim.n_grid, n_particles=sim.n_particles, max_frames=max_frames, substeps=sim.substeps,
                      dt=sim.dt, p_vol=sim.p_vol, p_mass=sim.p_mass, gravity=sim.gravity,
                      ground_friction=sim.ground_friction, primitives=plist, dtype=dtype, svd_grad_clamp=svd_grad_clamp, **engine_kw)
