"""ctypes binding of libmrbf.so (include/mrbf.h).  No CPU fallback: a missing library or GPU is an error."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MRBF_LIB") or os.path.join(_HERE, "libmrbf.so")  # MRBF_LIB: same override as HipRbf.jl (A/B runs of two builds)

c_dp = ctypes.POINTER(ctypes.c_double)
c_fp = ctypes.POINTER(ctypes.c_float)
c_ip = ctypes.POINTER(ctypes.c_int32)
c_vp = ctypes.c_void_p

# error codes (mrbf.h)
MRBF_OK, MRBF_ENOTPD, MRBF_ESINGULAR, MRBF_EHIP, MRBF_EBLAS, MRBF_ENOMEM, MRBF_ENODEVICE, MRBF_ENCCL = range(8)
PATH_CHOL, PATH_PROJ_CHOL, PATH_LU, PATH_MINNORM, PATH_ROUND4 = 1, 2, 3, 4, 5
OPT_GRAM_MODE, OPT_RESIDUAL, OPT_FORCE_PATH, OPT_CHOL_IMPL, OPT_EVAL_IMPL, OPT_TIMING, OPT_DIAG_IMPL, OPT_CHOL_WINDOW = 1, 2, 3, 4, 5, 6, 7, 8
OPT_SPIN_MS, OPT_DEBUG_FAULT, OPT_LAST_DEVICE_MS, OPT_SLOW_LAUNCHES = 9, 10, 11, 12
OPT_ARENA_BYTES, OPT_LIVE_HANDLES = 13, 14   # read only
FB_CHOL_HOST_DRIVEN, FB_BACKSOLVE_BLOCKED, FB_LU = 1, 2, 4


class FitInfo(ctypes.Structure):
    _fields_ = [("path", ctypes.c_int32), ("factor_info", ctypes.c_int32), ("n", ctypes.c_int32), ("q", ctypes.c_int32),
                ("rel_residual", ctypes.c_double), ("max_pitw", ctypes.c_double), ("mu", ctypes.c_double),
                ("ms_gram", ctypes.c_float), ("ms_project", ctypes.c_float), ("ms_factor", ctypes.c_float),
                ("ms_solve", ctypes.c_float), ("ms_check", ctypes.c_float), ("ms_total", ctypes.c_float),
                ("fallbacks", ctypes.c_int32), ("giveup_code", ctypes.c_int32), ("ms_factor_device", ctypes.c_float),
                ("slow_launches", ctypes.c_int32)]

    def asdict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


class EvalInfo(ctypes.Structure):
    _fields_ = [("ms_total", ctypes.c_float), ("ms_dist", ctypes.c_float), ("ms_kernel", ctypes.c_float),
                ("ms_contract", ctypes.c_float)]

    def asdict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


class PsOptions(ctypes.Structure):
    _fields_ = [("max_ideal_evals", ctypes.c_int32), ("max_ps_evals", ctypes.c_int32), ("max_polish_evals", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("seed", ctypes.c_uint64), ("t0", ctypes.c_double), ("xtol_rel", ctypes.c_double)]


class PsInfo(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int32), ("generations", ctypes.c_int32), ("evals_ideal", ctypes.c_int32),
                ("evals_ps", ctypes.c_int32), ("evals_polish", ctypes.c_int32), ("ms_total", ctypes.c_float), ("tau", ctypes.c_double)]

    def asdict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


PS_OK, PS_CRITICAL, PS_FAILURE = 0, 1, 2
ROLE_NONE, ROLE_EQ, ROLE_INEQ = -1, -2, -3
DISPATCH_REFERENCE, DISPATCH_DEVICE = 0, 1
FIT_FULL, FIT_FROM_ROUND4 = 0, 1
ENTRY_ROUND4, ENTRY_FIT_FROM_ROUND4, ENTRY_PS_STEP, ENTRY_BACKTRACK, ENTRY_AFFINE = 1, 2, 3, 4, 5


class PsProblem(ctypes.Structure):
    _fields_ = [("n_models", ctypes.c_int32), ("n_objectives", ctypes.c_int32), ("models", ctypes.POINTER(ctypes.c_void_p)),
                ("roles", ctypes.POINTER(ctypes.c_int32)), ("n_lin_eq", ctypes.c_int32), ("n_lin_ineq", ctypes.c_int32),
                ("A_eq", ctypes.c_void_p), ("b_eq", ctypes.c_void_p), ("A_ineq", ctypes.c_void_p), ("b_ineq", ctypes.c_void_p),
                ("eq_tol", ctypes.c_double)]


class Problem(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int64), ("m", ctypes.c_int64), ("d", ctypes.c_int32), ("k", ctypes.c_int32),
                ("kernel_id", ctypes.c_int32), ("poly_deg", ctypes.c_int32), ("a", ctypes.c_double), ("b", ctypes.c_double),
                ("centres", c_dp), ("values", c_dp), ("X", c_dp), ("weights_out", c_dp), ("poly_out", c_dp),
                ("vals_out", c_dp), ("jac_out", c_dp)]


class Result(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int32), ("device", ctypes.c_int32), ("fit", FitInfo), ("ms_eval", ctypes.c_float),
                ("checksum_w", ctypes.c_double), ("checksum_vals", ctypes.c_double)]


# every symbol include/mrbf.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "mrbf_init": (ctypes.c_int32, [ctypes.c_int32, ctypes.POINTER(c_vp)]),
    "mrbf_shutdown": (ctypes.c_int32, [c_vp]),
    "mrbf_last_error": (ctypes.c_char_p, [c_vp]),
    "mrbf_version": (ctypes.c_char_p, []),
    "mrbf_set_option": (ctypes.c_int32, [c_vp, ctypes.c_int32, ctypes.c_double]),
    "mrbf_get_option": (ctypes.c_int32, [c_vp, ctypes.c_int32, c_dp]),
    "mrbf_set_stream": (ctypes.c_int32, [c_vp, c_vp]),
    "mrbf_sync": (ctypes.c_int32, [c_vp]),
    "mrbf_gram": (ctypes.c_int32, [c_vp, ctypes.c_int64, ctypes.c_int32, c_vp, ctypes.c_int32, ctypes.c_double,
                                   ctypes.c_double, ctypes.c_int32, c_vp, c_vp, c_fp]),
    "mrbf_cross_gram": (ctypes.c_int32, [c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, c_vp, c_vp, ctypes.c_int32, ctypes.c_double,
                                         ctypes.c_double, c_vp]),
    "mrbf_fit": (ctypes.c_int32, [c_vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, c_vp, c_vp, ctypes.c_int32,
                                  ctypes.c_double, ctypes.c_double, ctypes.c_int32, ctypes.POINTER(c_vp), c_vp, c_vp,
                                  ctypes.POINTER(FitInfo)]),
    "mrbf_model_from_coeffs": (ctypes.c_int32, [c_vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, c_vp, c_vp, c_vp,
                                                ctypes.c_int32, ctypes.c_double, ctypes.c_double, ctypes.c_int32,
                                                ctypes.POINTER(c_vp)]),
    "mrbf_eval": (ctypes.c_int32, [c_vp, c_vp, ctypes.c_int64, c_vp, c_vp, c_vp, ctypes.POINTER(EvalInfo)]),
    "mrbf_backtrack": (ctypes.c_int32, [c_vp, c_vp, c_vp, c_vp, ctypes.c_double, ctypes.c_double, ctypes.c_int32,
                                        ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int32, c_vp, c_vp,
                                        c_vp, c_ip]),
    "mrbf_affine_scores": (ctypes.c_int32, [c_vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, c_vp, c_vp, ctypes.c_int32, c_vp,
                                            ctypes.POINTER(ctypes.c_int64), c_dp]),
    "mrbf_affine_select": (ctypes.c_int32, [c_vp, ctypes.c_int64, ctypes.c_int32, c_vp, ctypes.c_int32, c_vp, ctypes.c_int32, ctypes.c_double,
                                            ctypes.c_int32, ctypes.POINTER(ctypes.c_int64), c_ip, c_vp]),
    "mrbf_round4": (ctypes.c_int32, [c_vp, ctypes.c_int64, ctypes.c_int32, c_vp, ctypes.c_int64, c_vp, ctypes.c_int32, ctypes.c_double,
                                     ctypes.c_double, ctypes.c_int32, ctypes.c_int32, ctypes.c_double, c_vp, c_ip, ctypes.POINTER(c_vp)]),
    "mrbf_fit_from_round4": (ctypes.c_int32, [c_vp, c_vp, ctypes.c_int32, c_vp, ctypes.POINTER(c_vp), c_vp, c_vp, ctypes.POINTER(FitInfo)]),
    "mrbf_round4_sites": (ctypes.c_int32, [c_vp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), c_ip]),
    "mrbf_free_round4": (ctypes.c_int32, [c_vp, c_vp]),
    "mrbf_model_dims": (ctypes.c_int32, [c_vp, ctypes.POINTER(ctypes.c_int64), c_ip, c_ip, c_ip]),
    "mrbf_free_model": (ctypes.c_int32, [c_vp, c_vp]),
    "mrbf_batch_run": (ctypes.c_int32, [ctypes.c_int32, c_ip, ctypes.c_int64, ctypes.POINTER(Problem),
                                        ctypes.POINTER(Result)]),
    "mrbf_debug_mfma_layout": (ctypes.c_int32, [c_vp, c_vp, c_vp, c_vp]),
    "mrbf_debug_potrf": (ctypes.c_int32, [c_vp, ctypes.c_int64, c_vp, ctypes.c_int32, c_ip, c_fp]),
    "mrbf_debug_env": (ctypes.c_int32, [ctypes.c_char_p]),
    "mrbf_debug_diag": (ctypes.c_int32, [c_vp, c_vp, ctypes.c_int32, c_fp, c_dp, c_dp]),
    "mrbf_debug_ps_rank": (ctypes.c_int32, [c_vp, ctypes.c_int32, c_vp, c_vp, ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32, c_ip, c_ip]),
    "mrbf_debug_mfma_peak": (ctypes.c_int32, [c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_fp, c_dp]),
    "mrbf_debug_mfma_asm": (ctypes.c_int32, [c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_fp, c_dp, c_dp]),
    "mrbf_debug_dgemm": (ctypes.c_int32, [c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_fp, c_dp]),
    "mrbf_debug_mega_tables": (ctypes.c_int32, [ctypes.c_int32] * 8 + [c_vp]),
    "mrbf_debug_mega_tables2": (ctypes.c_int32, [ctypes.c_int32] * 8 + [c_vp, c_vp]),
    "mrbf_stochastic_rank": (ctypes.c_int32, [ctypes.c_int32, c_vp, c_vp, c_vp, ctypes.c_double, c_vp]),
    "mrbf_ps_step": (ctypes.c_int32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(PsOptions), c_vp, c_vp, c_vp,
                                      ctypes.POINTER(PsInfo)]),
    "mrbf_ps_step_problem": (ctypes.c_int32, [c_vp, ctypes.POINTER(PsProblem), c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(PsOptions),
                                              c_vp, c_vp, c_vp, ctypes.POINTER(PsInfo)]),
    "mrbf_dispatch_ps": (ctypes.c_int32, [ctypes.c_int32] * 6),
    "mrbf_dispatch_backtrack": (ctypes.c_int32, [ctypes.c_int32] * 3),
    "mrbf_dispatch_affine": (ctypes.c_int32, [ctypes.c_int64, ctypes.c_int32]),
    "mrbf_dispatch_round4": (ctypes.c_int32, [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64]),
    "mrbf_dispatch_fit": (ctypes.c_int32, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "mrbf_dispatch_after": (ctypes.c_int32, [ctypes.c_int32, ctypes.c_int32]),
}

_LIB = None


class MrbfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmrbf error %d: %s" % (code, msg))
        self.code = code


def load():
    """Load libmrbf.so and bind every exported symbol; raises if the extension has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libmrbf.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C morbit.jl_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def as_ptr(a):
    """void* of a NumPy array (host) or of anything exposing data_ptr() (a torch tensor, host or device)."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return ctypes.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):
        return ctypes.c_void_p(a.data_ptr())
    raise TypeError("expected numpy array or tensor, got %r" % type(a))


def host_f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


class Context:
    """One per host thread and GPU (mrbf_init / mrbf_shutdown)."""

    def __init__(self, device_id=-1):
        self.lib = load()
        h = c_vp()
        rc = self.lib.mrbf_init(device_id, ctypes.byref(h))
        if rc != 0:
            raise MrbfError(rc, (self.lib.mrbf_last_error(None) or b"").decode())
        self.h = h

    def check(self, rc):
        if rc != 0:
            raise MrbfError(rc, (self.lib.mrbf_last_error(self.h) or b"").decode())

    def set_option(self, key, value):
        self.check(self.lib.mrbf_set_option(self.h, key, float(value)))

    def get_option(self, key):
        v = ctypes.c_double()
        self.check(self.lib.mrbf_get_option(self.h, key, ctypes.byref(v)))
        return v.value

    def sync(self):
        self.check(self.lib.mrbf_sync(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.mrbf_shutdown(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DEFAULT_CTX = {}


def default_context(device_id=-1):
    import threading

    key = (threading.get_ident(), device_id)
    if key not in _DEFAULT_CTX:
        _DEFAULT_CTX[key] = Context(device_id)
    return _DEFAULT_CTX[key]
