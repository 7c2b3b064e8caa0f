"""ctypes binding of libjetship.so (the C ABI declared in include/jetship.h).

This is the Python twin of the `ccall` stubs a Julia maintainer would write: every
symbol of include/jetship.h is bound here with its exact C signature.  There is no CPU fallback:
if the shared library is missing the import fails loudly, and `init()` fails loudly when no
gfx950 device is visible.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

__all__ = ["lib", "check", "JetsHipError", "LIB_PATH", "BlockDesc", "SYMBOLS", "DTYPES", "KINDS"]

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JETSHIP_LIB", os.path.join(_HERE, "libjetship.so"))


class JetsHipError(RuntimeError):
    """Raised for every non-zero jh_status; carries jh_last_error() (reference: `error(...)`,
    src/Jets.jl:131,179,1116)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"libjetship status {status}: {message}")
        self.status = status
        self.message = message


class BlockDesc(C.Structure):
    """jh_block_desc (include/jetship.h)."""

    _fields_ = [
        ("kind", C.c_int32),
        ("adjoint", C.c_int32),
        ("coeff", C.c_void_p),
        ("scale_re", C.c_double),
        ("scale_im", C.c_double),
        ("nr", C.c_int64),
        ("nc", C.c_int64),
        ("scale_flags", C.c_int32),
        ("reserved", C.c_int32),
    ]


ABI_VERSION = 4                      # JETSHIP_ABI_VERSION of include/jetship.h
SCALAR_COMPLEX, SCALAR_WIDE = 1, 2   # JH_SCALAR_* of include/jetship.h


def scalar_flags(a) -> int:
    """The TYPE of a scalar as Julia's dispatch sees it (src/Jets.jl:1159 `a * m`; include/jetship.h JH_SCALAR_*), read off the Python object:
    a Python or numpy complex is a Complex scalar (full complex product even with a zero imaginary part); numpy's float64 / complex128 are
    Julia's Float64 / ComplexF64 (against 32-bit elements: promoted arithmetic, one rounding on the store); plain Python numbers and 32-bit
    numpy scalars are taken in the vectors' element type (Julia's `T(a)`, what an Int, a Float32 or an Irrational like pi gives)."""
    f = SCALAR_COMPLEX if isinstance(a, (complex, np.complexfloating)) else 0
    return f | (SCALAR_WIDE if isinstance(a, (np.float64, np.complex128)) else 0)


class ChainStage(C.Structure):
    """jh_chain_stage (include/jetship.h)."""

    _fields_ = [("kind", C.c_int32), ("flags", C.c_int32), ("a", C.c_double), ("coeff", C.POINTER(C.c_void_p)), ("row_flags", C.POINTER(C.c_uint8))]


STAGE_SCALE, STAGE_DIAG, STAGE_CONJ, STAGE_ROWSUM = 1, 2, 4, 8      # jh_stage_kind, JH_STAGE_CONJ, JH_STAGE_ROWSUM
CHAIN_FORWARD, CHAIN_ADJOINT, CHAIN_NORMAL = 0, 1, 2   # jh_chain_type


class LsqrResultC(C.Structure):
    """jh_lsqr_result of include/jetship.h."""

    _fields_ = [("istop", C.c_int32), ("itn", C.c_int32), ("r1norm", C.c_double), ("r2norm", C.c_double), ("anorm", C.c_double),
                ("acond", C.c_double), ("arnorm", C.c_double), ("xnorm", C.c_double)]


DTYPES = {"f32": 0, "f64": 1, "c32": 2, "c64": 3}
KINDS = {"zero": 0, "identity": 1, "scale": 2, "diag": 3, "dense": 4, "square": 5}

_vp = C.c_void_p
_i64 = C.c_int64
_i64p = C.POINTER(C.c_int64)
_dblp = C.POINTER(C.c_double)
_int = C.c_int
_intp = C.POINTER(C.c_int)
_vpp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); must list every function declared in include/jetship.h
SYMBOLS = {
    "jh_abi_version": (_int, []),
    "jh_last_error": (C.c_char_p, []),
    "jh_device_count": (_int, [_intp]),
    "jh_init": (_int, [_int]),
    "jh_shutdown": (_int, []),
    "jh_context_create": (_int, [_int, _intp]),
    "jh_context_use": (_int, [_int]),
    "jh_context_current": (_int, [_intp, _intp]),
    "jh_context_destroy": (_int, [_int]),
    "jh_set_device": (_int, [_int]),
    "jh_bvec_context": (_int, [_vp, _intp, _intp]),
    "jh_comm_init_all": (_int, [_int, _intp]),
    "jh_comm_group_begin": (_int, []),
    "jh_comm_group_end": (_int, []),
    "jh_device_info": (_int, [C.c_char_p, _int, _i64p, _i64p, _intp]),
    "jh_trim": (_int, []),
    "jh_get_stream": (_int, [_vpp]),
    "jh_set_stream": (_int, [_vp]),
    "jh_synchronize": (_int, []),
    "jh_event_create": (_int, [_vpp]),
    "jh_event_record": (_int, [_vp]),
    "jh_event_elapsed_ms": (_int, [_vp, _vp, C.POINTER(C.c_float)]),
    "jh_event_destroy": (_int, [_vp]),
    "jh_bvec_create": (_int, [_i64, _i64p, _int, _vpp]),
    "jh_bvec_create_uninit": (_int, [_i64, _i64p, _int, _vpp]),
    "jh_bvec_wrap": (_int, [_vp, _i64, _i64p, _int, _vpp]),
    "jh_bvec_view": (_int, [_vp, _i64, _i64, _vpp]),
    "jh_bvec_destroy": (_int, [_vp]),
    "jh_bvec_info": (_int, [_vp, _i64p, _i64p, _intp, _vpp]),
    "jh_bvec_block": (_int, [_vp, _i64, _i64p, _i64p, _vpp]),
    "jh_getblock_copy": (_int, [_vp, _i64, _vp, _int]),
    "jh_setblock_copy": (_int, [_vp, _i64, _vp, _int]),
    "jh_setblock_fill": (_int, [_vp, _i64, C.c_double, C.c_double]),
    "jh_fill": (_int, [_vp, C.c_double, C.c_double]),
    "jh_copy": (_int, [_vp, _vp]),
    "jh_download": (_int, [_vp, _i64, _i64, _vp]),
    "jh_upload": (_int, [_vp, _i64, _i64, _vp]),
    "jh_host_alloc": (_int, [C.c_size_t, _vpp]),
    "jh_host_free": (_int, [_vp]),
    "jh_host_register": (_int, [_vp, C.c_size_t]),
    "jh_host_unregister": (_int, [_vp]),
    "jh_fill_uniform": (_int, [_vp, C.c_uint64, C.c_uint64, _i64]),
    "jh_fill_normal": (_int, [_vp, C.c_uint64, C.c_uint64, _i64]),
    "jh_abs": (_int, [_vp, _vp]),
    "jh_lincomb": (_int, [_vp, _int, _dblp, _vpp]),
    "jh_lincomb_typed": (_int, [_vp, _int, _dblp, C.POINTER(C.c_int32), _vpp]),
    "jh_hadamard": (_int, [_vp, _vp, _vp, _int]),
    "jh_bcast_check": (_int, [C.c_char_p, _int, _int, _int]),
    "jh_bcast_compile": (_int, [C.c_char_p, _int, _int, _int, _vpp]),
    "jh_bcast_compile_mixed": (_int, [C.c_char_p, _int, _int, _int, _int, _vpp]),
    "jh_bcast_compile_typed": (_int, [C.c_char_p, _int, _int, _int, _int, _int, _vpp]),
    "jh_bcast_check_typed": (_int, [C.c_char_p, _int, _int, _int, _int, _int]),
    "jh_bcast_apply": (_int, [_vp, _vp, _vpp, _dblp]),
    "jh_bcast_apply_many": (_int, [_int, _vpp, _vpp, _vpp, _dblp]),
    "jh_bcast_destroy": (_int, [_vp]),
    "jh_dot": (_int, [_vp, _vp, _dblp, _dblp]),
    "jh_norm": (_int, [_vp, C.c_double, _dblp]),
    "jh_extrema": (_int, [_vp, _dblp, _dblp]),
    "jh_norm_blocks": (_int, [_vp, C.c_double, _dblp]),
    "jh_dot_blocks": (_int, [_vp, _vp, _dblp, _dblp]),
    "jh_gemv": (_int, [_vp, _i64, _i64, _int, _vp, _vp, _int]),
    "jh_blockop_create": (_int, [_i64, _i64, C.POINTER(BlockDesc), _i64p, _i64p, _int, _vpp]),
    "jh_blockop_destroy": (_int, [_vp]),
    "jh_blockop_mul": (_int, [_vp, _vp, _vp]),
    "jh_blockop_mul_adj": (_int, [_vp, _vp, _vp]),
    "jh_blockop_f": (_int, [_vp, _vp, _vp]),
    "jh_blockop_point": (_int, [_vp, _vp]),
    "jh_blockop_mul_adj_range": (_int, [_vp, _vp, _vp, _i64, _i64]),
    "jh_blockop_normal_mul_range": (_int, [_vp, _vp, _vp, _i64, _i64]),
    "jh_blockop_normal_mul": (_int, [_vp, _vp, _vp]),
    "jh_blocksum_mul": (_int, [_int, _vpp, _dblp, _dblp, _vp, _vp]),
    "jh_blocksum_mul_adj": (_int, [_int, _vpp, _dblp, _dblp, _vp, _vp]),
    "jh_blocksum_mul_typed": (_int, [_int, _vpp, _dblp, C.POINTER(C.c_int32), _dblp, _vp, _vp]),
    "jh_blocksum_mul_adj_typed": (_int, [_int, _vpp, _dblp, C.POINTER(C.c_int32), _dblp, _vp, _vp]),
    "jh_chain_create": (_int, [_vp, _int, _int, C.POINTER(ChainStage), _int, C.POINTER(ChainStage), _int, C.POINTER(ChainStage), _vpp]),
    "jh_chain_apply": (_int, [_vp, _vp, _vp, _int]),
    "jh_chain_destroy": (_int, [_vp]),
    "jh_blockop_mul_axpby": (_int, [_vp, _vp, _vp, C.c_double, C.c_double, _dblp]),
    "jh_blockop_mul_adj_axpby": (_int, [_vp, _vp, _vp, C.c_double, C.c_double, C.c_double, _dblp]),
    "jh_blockop_mul_scaled": (_int, [_vp, _vp, _vp, C.c_double, _int]),
    "jh_blockop_mul_adj_scaled": (_int, [_vp, _vp, _vp, C.c_double, _int]),
    "jh_blockop_bidiag_step": (_int, [_vp, _vp, _vp, _vp, C.c_double, C.c_double, _dblp]),
    "jh_blockop_bidiag_step_range": (_int, [_vp, _vp, _vp, _vp, C.c_double, C.c_double, _i64, _i64, _dblp]),
    "jh_normsq_reset": (_int, []),
    "jh_normsq_read": (_int, [_dblp]),
    "jh_lsqr_solve": (_int, [_vp, _vp, _vp, _int, C.c_double, C.c_double, C.c_double, C.c_double, _int, _int, C.POINTER(LsqrResultC), _dblp]),
    "jh_lsqr_solve_partitioned": (_int, [_vp, _vp, _vp, _int, C.c_double, C.c_double, C.c_double, C.c_double, _int, _int, C.POINTER(LsqrResultC), _dblp]),
    "jh_lsqr_solve_team": (_int, [_int, _vpp, _vpp, _vpp, _int, C.c_double, C.c_double, C.c_double, C.c_double, _int, _int, C.POINTER(LsqrResultC), _dblp]),
    "jh_cgls_solve": (_int, [_vp, _vp, _vp, _int, C.c_double, C.c_double, C.c_double, _int, _int, C.POINTER(LsqrResultC), _dblp]),
    "jh_cgls_solve_partitioned": (_int, [_vp, _vp, _vp, _int, C.c_double, C.c_double, C.c_double, _int, _int, C.POINTER(LsqrResultC), _dblp]),
    "jh_cgls_solve_team": (_int, [_int, _vpp, _vpp, _vpp, _int, C.c_double, C.c_double, C.c_double, _int, _int, C.POINTER(LsqrResultC), _dblp]),
    "jh_cgnr_solve": (_int, [_vp, _vp, _vp, _int, C.c_double, C.c_double, C.c_double, _int, _int, C.POINTER(LsqrResultC), _dblp]),
    "jh_cgnr_solve_partitioned": (_int, [_vp, _vp, _vp, _int, C.c_double, C.c_double, C.c_double, _int, _int, C.POINTER(LsqrResultC), _dblp]),
    "jh_cgnr_solve_team": (_int, [_int, _vpp, _vpp, _vpp, _int, C.c_double, C.c_double, C.c_double, _int, _int, C.POINTER(LsqrResultC), _dblp]),
    "jh_team_mul": (_int, [_int, _vpp, _vpp, _vpp]),
    "jh_team_mul_adj": (_int, [_int, _vpp, _vpp, _vpp, _int]),
    "jh_team_normal_mul": (_int, [_int, _vpp, _vpp, _vpp, _int]),
    "jh_comm_available": (_int, []),
    "jh_comm_unique_id": (_int, [_vp]),
    "jh_comm_init_rank": (_int, [_vp, _int, _int]),
    "jh_comm_destroy": (_int, []),
    "jh_comm_info": (_int, [_intp, _intp]),
    "jh_comm_allreduce_sum": (_int, [_vp]),
    "jh_comm_allreduce_scalars": (_int, [_dblp, _int, _int]),
    "jh_comm_allreduce_sum_range": (_int, [_vp, _i64, _i64]),
    "jh_comm_join": (_int, []),
    "jh_comm_allreduce_normsq": (_int, [_dblp]),
    "jh_tune_set": (_int, [C.c_char_p, _i64]),
    "jh_tune_get": (_int, [C.c_char_p, _i64p]),
    "jh_blockop_tune_get": (_int, [_vp, C.c_char_p, _i64p]),
    "jh_blockop_tune_set": (_int, [_vp, C.c_char_p, _i64]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"libjetship.so not found at {LIB_PATH}. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C jets.jl_amd/csrc`. There is no CPU fallback for the block-operator path."
        )
    handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(handle, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if handle.jh_abi_version() != ABI_VERSION:          # struct layouts (jh_block_desc) are part of the version: never mix
        raise ImportError(f"{LIB_PATH} speaks ABI version {handle.jh_abi_version()}, this binding {ABI_VERSION}: rebuild the library (make -C jets.jl_amd/csrc)")
    return handle


lib = _load()


def check(status: int) -> None:
    if status != 0:
        msg = lib.jh_last_error()
        raise JetsHipError(status, msg.decode("utf-8", "replace") if msg else "")
