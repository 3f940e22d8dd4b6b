"""ctypes binding of libm4q_hip.so (the C ABI of include/m4q.h).

The HIP library is the product: nothing here falls back to a CPU path.  If the shared object is
missing, or a GPU entry point is called without a device, an exception is raised.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# M4Q_LIB: an experiment build of the same library (tools/build_variant.sh) - still the HIP library, never a CPU path
LIB_PATH = os.path.abspath(os.environ["M4Q_LIB"]) if os.environ.get("M4Q_LIB") else os.path.join(_HERE, "libm4q_hip.so")

QP_REF_LQR = 1
QP_DU_BAND = 2
QP_EXACT_BOX = 4
OPT_FORCE_COMPLEX = 1
OPT_NO_TRACELESS = 2
OPT_TILE = 4
OPT_NO_TILE = 8
OPT_NO_SG = 16
PLANT_NONE, PLANT_HAMILTONIAN, PLANT_GENERATOR = 0, 1, 2
E_UNSUPPORTED, E_BADARG, E_NODEVICE, E_TIMEOUT, E_COMM = -1001, -1002, -1003, -1004, -1005
UNIQUE_ID_BYTES = 128

(F_MODELS, F_X0, F_X_TARG, F_U_TARG, F_Q, F_R, F_QF, F_OP0, F_OPS, F_XS, F_US, F_CODES, F_STEPS_DONE,
 F_QP_SOLVES, F_X_GUESS, F_U_GUESS) = range(16)


class M4qError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libm4q_hip error %d: %s" % (code, msg))
        self.code = code


class Problem(C.Structure):
    """m4q_problem (include/m4q.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "dim_x", "dim_u", "order", "horizon", "n_steps", "max_iter", "warm_start", "qp_flags", "plant_kind",
        "model_per_instance", "plant_per_instance", "target_per_instance", "target_cols", "reserved", "measure_freq",
        "reserved2")] + [
        (n, C.c_double) for n in ("dt", "sat", "du", "ls_tol")]


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_vp = C.c_void_p
_i32 = C.c_int32

# name -> (restype, argtypes): every symbol include/m4q.h declares
PROTOTYPES = {
    "m4q_last_error": (C.c_char_p, []),
    "m4q_version": (C.c_char_p, []),
    "m4q_device_count": (C.c_int, []),
    "m4q_supported": (C.c_int, [_i32, _i32, _i32]),
    "m4q_library_size": (C.c_int, [_i32, _i32]),
    "m4q_power_list": (C.c_int, [_i32, _i32, _ip]),
    "m4q_linearize_batch": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _dp, _i32, _dp, _dp, _dp, _dp, _dp]),
    "m4q_quad_program_batch": (C.c_int, [_i32, _i32, _i32, _i32, _i32, C.c_double, C.c_double, _dp, _dp, _dp, _i32,
                                         _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "m4q_discretize_batch": (C.c_int, [_i32, _i32, _i32, _i32, C.c_double, _dp, _i32, _dp, _dp]),
    "m4q_session_build_models": (C.c_int, [_vp, C.c_double, _dp, _i32, _dp]),
    "m4q_plant_step_batch": (C.c_int, [_i32, _i32, _i32, _i32, C.c_double, _dp, _dp, _dp, _dp, _i32, _dp]),
    "m4q_mpc_batch": (C.c_int, [C.POINTER(Problem), _i32, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip,
                                _ip, _ip]),
    "m4q_session_create": (C.c_int, [C.POINTER(Problem), _i32, _i32, C.POINTER(_vp)]),
    "m4q_session_destroy": (None, [_vp]),
    "m4q_session_field_bytes": (C.c_size_t, [_vp, _i32]),
    "m4q_session_upload": (C.c_int, [_vp, _i32, _vp, C.c_size_t]),
    "m4q_session_download": (C.c_int, [_vp, _i32, _vp, C.c_size_t]),
    "m4q_session_put_state": (C.c_int, [_vp, _i32, _vp]),
    "m4q_session_get_state": (C.c_int, [_vp, _i32, _vp]),
    "m4q_session_device_ptr": (_vp, [_vp, _i32]),
    "m4q_session_bind_output": (C.c_int, [_vp, _i32, _vp, C.c_size_t]),
    "m4q_session_run": (C.c_int, [_vp, _i32, _i32]),
    "m4q_session_sync": (C.c_int, [_vp]),
    "m4q_session_set_codes": (C.c_int, [_vp, _ip]),
    "m4q_session_kernel_ms": (C.c_int, [_vp, C.POINTER(C.c_double), _ip]),
    "m4q_session_info": (C.c_int, [_vp, C.POINTER(C.c_int64), _ip, _ip]),
    "m4q_session_qp_stats": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "m4q_session_path": (C.c_int, [_vp]),
    "m4q_session_copy_final_state": (C.c_int, [_vp, _vp]),
    "m4q_session_copy_status": (C.c_int, [_vp, _vp]),
    "m4q_comm_unique_id": (C.c_int, [_vp]),
    "m4q_comm_create": (C.c_int, [_i32, _i32, _vp, _i32, C.POINTER(_vp)]),
    "m4q_comm_destroy": (None, [_vp]),
    "m4q_comm_gather": (C.c_int, [_vp, _vp, _vp, _vp, C.c_size_t, _i32, _i32]),
    "m4q_comm_wait": (C.c_int, [_vp, _i32]),
    "m4q_comm_allreduce_f64": (C.c_int, [_vp, _dp, _i32, _i32]),
    "m4q_device_alloc": (C.c_int, [C.c_size_t, _i32, C.POINTER(_vp)]),
    "m4q_device_free": (C.c_int, [_vp]),
    "m4q_device_read": (C.c_int, [_vp, _vp, C.c_size_t]),
    "m4q_device_write": (C.c_int, [_vp, _vp, C.c_size_t]),
}

_lib = None


def lib():
    """Load the shared library once; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python mpc4quantum_amd/csrc/build.py` "
                "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc < 0:
        raise M4qError(rc, lib().m4q_last_error().decode())
    return rc


def cbuf(a):
    """complex128 C-contiguous array -> (array kept alive, double*)."""
    a = np.ascontiguousarray(a, dtype=np.complex128)
    return a, a.ctypes.data_as(_dp)


def rbuf(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def ibuf(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def supported(dim_x, dim_u, order):
    return bool(lib().m4q_supported(dim_x, dim_u, order))


def device_count():
    return lib().m4q_device_count()
