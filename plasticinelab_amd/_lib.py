"""ctypes binding of libplmpm.so (include/plmpm.h; the tools of include/plmpm_tools.h for bench.py / profiles / tests).

There is no fallback: if the HIP library is missing or fails to load, importing
the engine raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
(or ``make -C plasticinelab_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PLMPM_LIB: another build of the same library (profiling / A-B experiments under profiles/tools); never a fallback
LIB_PATH = os.environ.get("PLMPM_LIB") or os.path.join(_HERE, "libplmpm.so")

MAX_PRIMITIVES = 8
MAX_ACTION_DIM = 7
F32, F64 = 0, 1
SHAPES = {"Sphere": 0, "Capsule": 1, "Cylinder": 2, "Torus": 3, "Box": 4, "RollingPin": 1,   # RollingPin is a Capsule
          "Chopsticks": 5}
KINEMATICS = {"RollingPin": 1, "Chopsticks": 2}


class Config(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("n_grid", C.c_int32), ("n_particles", C.c_int32),
                ("max_frames", C.c_int32), ("substeps", C.c_int32), ("n_primitives", C.c_int32),
                ("dt", C.c_double), ("p_vol", C.c_double), ("p_mass", C.c_double),
                ("gravity", C.c_double * 3), ("ground_friction", C.c_double),
                ("svd_grad_clamp", C.c_double), ("slab_z0", C.c_int32), ("slab_z1", C.c_int32),
                ("store_grid", C.c_int32), ("slab_halo", C.c_int32), ("resort_steps", C.c_int32),
                ("grid_lo", C.c_int32 * 3), ("grid_hi", C.c_int32 * 3), ("particle_capacity", C.c_int32),
                ("deterministic", C.c_int32), ("contact_min_adjoint", C.c_int32), ("minmax_tie", C.c_int32),
                ("grid_workgroups", C.c_int32)]


class Primitive(C.Structure):
    _fields_ = [("shape", C.c_int32), ("action_dim", C.c_int32), ("params", C.c_double * 3),
                ("friction", C.c_double), ("action_scale", C.c_double * MAX_ACTION_DIM),
                ("lower_bound", C.c_double * 3), ("upper_bound", C.c_double * 3),
                ("kinematics", C.c_int32), ("reserved", C.c_int32)]


class Workspace(C.Structure):
    _fields_ = [("state_bytes", C.c_size_t), ("adjoint_bytes", C.c_size_t),
                ("grid_bytes", C.c_size_t), ("misc_bytes", C.c_size_t)]


# every symbol include/plmpm.h (the boundary) and include/plmpm_tools.h (measurement, diagnostics, test hooks) declare:
# name -> (restype, argtypes)
_P, _I, _D = C.c_void_p, C.c_int, C.c_double
SYMBOLS = {
    "plmpm_last_error": (C.c_char_p, []),
    "plmpm_version": (_I, []),
    "plmpm_build_flags": (_I, []),
    "plmpm_create": (_I, [C.POINTER(Config), C.POINTER(Primitive), C.POINTER(_P)]),
    "plmpm_destroy": (_I, [_P]),
    "plmpm_workspace_bytes": (_I, [_P, C.POINTER(Workspace)]),
    "plmpm_bind_workspace": (_I, [_P, _P, _P, _P, _P]),
    "plmpm_set_stream": (_I, [_P, _P]),
    "plmpm_set_materials": (_I, [_P, _P, _P, _P]),
    "plmpm_set_frame": (_I, [_P, _I, _P, _P, _P, _P, _I]),
    "plmpm_get_frame": (_I, [_P, _I, _P, _P, _P, _P]),
    "plmpm_copy_frame": (_I, [_P, _I, _I]),
    "plmpm_set_primitive_state": (_I, [_P, _I, _I, _P]),
    "plmpm_get_primitive_state": (_I, [_P, _I, _I, _P]),
    "plmpm_set_softness": (_I, [_P, _D]),
    "plmpm_set_action": (_I, [_P, _I, _I, _P]),
    "plmpm_get_action_grad": (_I, [_P, _I, _P]),
    "plmpm_substep": (_I, [_P, _I]),
    "plmpm_step": (_I, [_P, _I, _I]),
    "plmpm_grad_begin": (_I, [_P, _I]),
    "plmpm_substep_grad": (_I, [_P, _I]),
    "plmpm_step_grad": (_I, [_P, _I, _I, _I]),
    "plmpm_segment_carry": (_I, [_P, _I, _I]),
    "plmpm_add_frame_grad": (_I, [_P, _I, _P, _P, _P, _P]),
    "plmpm_get_frame_grad": (_I, [_P, _I, _P, _P, _P, _P]),
    "plmpm_set_resort": (_I, [_P, _I]),
    "plmpm_get_primitive_grad": (_I, [_P, _I, _I, _P]),
    "plmpm_add_primitive_grad": (_I, [_P, _I, _I, _P]),
    "plmpm_loss_set_target": (_I, [_P, _P]),
    "plmpm_loss_set_weights": (_I, [_P, _D, _D, _D, _I]),
    "plmpm_loss_forward": (_I, [_P, _I, _P]),
    "plmpm_loss_backward": (_I, [_P, _I]),
    "plmpm_get_grid_mass": (_I, [_P, _I, _P]),
    "plmpm_loss_get_target_sdf": (_I, [_P, _P]),
    "plmpm_grid_stats": (_I, [_P, _I, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "plmpm_tile_boxes": (_I, [_P, _I, _P, _I, C.POINTER(C.c_int)]),
    "plmpm_get_order": (_I, [_P, _P]),
    "plmpm_fk": (_I, [_P, _I, _I]),
    "plmpm_p2g": (_I, [_P, _I, _I]),
    "plmpm_grid_g2p": (_I, [_P, _I, _I]),
    "plmpm_grad_scatter": (_I, [_P, _I]),
    "plmpm_grad_gather": (_I, [_P, _I]),
    "plmpm_grid_interior": (_I, [_P, _I]),
    "plmpm_grad_gather_interior": (_I, [_P, _I]),
    "plmpm_chain_grad": (_I, [_P, _I, _I, _I]),
    "plmpm_grid_window": (_I, [_P, _P, _P]),
    "plmpm_halo_region": (_I, [_P, _I, _I, _I, _I, _I, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "plmpm_halo_set_recv": (_I, [_P, _I, _I, _P, _P, _P]),
    "plmpm_halo_apply": (_I, [_P, _I, _I]),
    "plmpm_peer_area_bytes": (_I, [_P, _I, _I, _I, C.POINTER(C.c_size_t)]),
    "plmpm_peer_alloc": (_I, [_P, C.c_size_t, C.POINTER(_P), _P]),
    "plmpm_peer_open": (_I, [_P, _P, C.POINTER(_P)]),
    "plmpm_halo_peer_setup": (_I, [_P, _I, _I, _P, _P, _P, _P]),
    "plmpm_halo_peer_exchange": (_I, [_P, _I, _I]),
    "plmpm_peer_status": (_I, [_P, C.POINTER(_I)]),
    "plmpm_halo_peer_reset": (_I, [_P, _I]),
    "plmpm_peer_fused": (_I, [_P, C.POINTER(_I)]),
    "plmpm_peer_memory_kind": (_I, [_P, C.POINTER(_I)]),
    "plmpm_peer_ping": (_I, [_P, _I, C.c_uint, _D, C.POINTER(_I), C.POINTER(_D)]),
    "plmpm_debug_peer_spoil": (_I, [_P, _D]),
    "plmpm_slab_step": (_I, [_P, _I, _I]),
    "plmpm_slab_step_grad": (_I, [_P, _I, _I]),
    "plmpm_set_ids": (_I, [_P, _P]),
    "plmpm_get_ids": (_I, [_P, _I, _P]),
    "plmpm_set_population": (_I, [_P, _I]),
    "plmpm_adjoint_rows": (_I, [_P, _I, C.POINTER(C.c_int32)]),
    "plmpm_get_materials": (_I, [_P, _I, _P, _P, _P]),
    "plmpm_frame_info": (_I, [_P, _I, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "plmpm_migrate_begin": (_I, [_P, _I, _P, C.POINTER(_P), C.POINTER(_P)]),
    "plmpm_migrate_finish": (_I, [_P, _I, _I, _P, _I, _P, C.POINTER(C.c_int32)]),
    "plmpm_migrate_adjoint_begin": (_I, [_P, _I, _P, _P, C.POINTER(_P), C.POINTER(_P)]),
    "plmpm_migrate_adjoint_finish": (_I, [_P, _I, _P, _P]),
    "plmpm_primitive_sdf": (_I, [_P, _I, _I, _P, _I, _P]),
    "plmpm_set_velocity": (_I, [_P, _I, _I, _I]),
    "plmpm_loss_contact_scalars": (_I, [_P, _P, _P]),
    "plmpm_measure_hbm": (_I, [_P, _P, C.c_size_t, _I, _P, C.POINTER(_D), C.POINTER(_D)]),
    "plmpm_pose_grad_region": (_I, [_P, _I, _I, C.POINTER(_P), C.POINTER(C.c_size_t), C.POINTER(_P), C.POINTER(C.c_size_t),
                                    C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "plmpm_action_grad_region": (_I, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "plmpm_loss_scatter": (_I, [_P, _I]),
    "plmpm_loss_partials": (_I, [_P, _I, _I, _P]),
    "plmpm_loss_set_globals": (_I, [_P, _P]),
    "plmpm_loss_finish": (_I, [_P, _P, _P]),
    "plmpm_loss_backward_local": (_I, [_P, _I]),
    "plmpm_check_error": (_I, [_P, C.POINTER(_I)]),
    "plmpm_debug_counters": (_I, [_P, _P]),
    "plmpm_debug_contact": (_I, [_P, _I, _P]),
    "plmpm_profile_enable": (_I, [_P, _I]),
    "plmpm_replay": (_I, [_P, _I, _I, _I, C.POINTER(C.c_double)]),
    "plmpm_replay_step": (_I, [_P, _I, _I, _I, _I, _I, C.POINTER(C.c_double)]),
    "plmpm_profile_kernel_count": (_I, []),
    "plmpm_profile_kernel_name": (C.c_char_p, [_I]),
    "plmpm_profile_read": (_I, [_P, _P, _P]),
}

_lib = None


class EngineError(RuntimeError):
    pass


def load():
    """Load libplmpm.so; raises if it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            f"{LIB_PATH} not found: the HIP engine is not built.  Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` at the repo root (needs hipcc).  There is no CPU implementation to fall back to.")
    _lib = bind(C.CDLL(LIB_PATH))
    return _lib


def bind(lib):
    """Give every entry point of a loaded build of the library its ctypes signature (load() for the product build; the tests'
    CPU interpreter of the device source, tests/emul_engine.py, binds its own build of the same sources)."""
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    return lib


def check(rc, lib=None):
    if rc != 0:
        raise EngineError((lib or load()).plmpm_last_error().decode())
