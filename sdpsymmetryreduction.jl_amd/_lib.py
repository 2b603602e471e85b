"""ctypes binding of libsdpsr_hip.so (C ABI: include/sdpsr.h).  The library is the
product; this file only declares argument types.  There is no CPU fallback: if the
shared object is missing the import of the binding fails loudly."""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsdpsr_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "sdpsr.h")

MEM_HOST, MEM_DEVICE = 0, 1
SQUARE_AUTO, SQUARE_I8, SQUARE_F32, SQUARE_F64 = 0, 1, 2, 3
T_TOTAL, T_PROJECT, T_SQUARE, T_REFINE, T_EIGEN, T_ISO, T_IRRED, T_IMAGE, T_COUNT = range(9)

STATUS = {
    0: "OK", 1: "INVALID_DECOMPOSITION_FIELD", 2: "NUMERICAL_INCONSISTENCY", 3: "DIMENSION_MISMATCH",
    4: "LABEL_OVERFLOW", 5: "BAD_ARGUMENT", 6: "HIP_ERROR", 7: "SOLVER_ERROR", 8: "OUT_OF_MEMORY",
    9: "NOT_CONVERGED", 10: "BAD_STATE",
}


class Opts(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("square_mode", C.c_int32),
        ("channels", C.c_int32),
        ("max_iters", C.c_int32),
        ("confirm_rounds", C.c_int32),
        ("eig_driver", C.c_int32),
        ("reserved", C.c_int32 * 10),
    ]


def declared_symbols(header_path=HEADER_PATH):
    """Every function the header declares (used by the symbol-export test)."""
    txt = open(header_path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sdpsr_[a-z0-9_]+)\s*\(", txt)))


_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C sdpsymmetryreduction.jl_amd/csrc` "
            "(there is no CPU fallback for the HIP path)")
    lib = C.CDLL(LIB_PATH)
    vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_double
    pi64, pi32 = C.POINTER(C.c_int64), C.POINTER(C.c_int32)
    sigs = {
        "sdpsr_create": (C.c_int, [C.c_int, C.c_uint64, C.POINTER(Opts), C.POINTER(vp)]),
        "sdpsr_destroy": (None, [vp]),
        "sdpsr_last_error": (C.c_char_p, [vp]),
        "sdpsr_status_string": (C.c_char_p, [C.c_int]),
        "sdpsr_version": (C.c_int, []),
        "sdpsr_set_stream": (C.c_int, [vp, vp]),
        "sdpsr_synchronize": (C.c_int, [vp]),
        "sdpsr_wait_stream": (C.c_int, [vp, vp]),
        "sdpsr_set_seed": (C.c_int, [vp, C.c_uint64]),
        "sdpsr_partition_from_f64": (C.c_int, [vp, i64, vp, vp, pi64, C.c_int]),
        "sdpsr_partition_from_u32": (C.c_int, [vp, i64, vp, vp, pi64, C.c_int]),
        "sdpsr_refine": (C.c_int, [vp, i64, vp, pi64, vp, i64, C.c_int]),
        "sdpsr_partition_checksum": (C.c_int, [vp, i64, vp, vp, C.c_int]),
        "sdpsr_fill": (C.c_int, [vp, i64, vp, vp, i64, vp, C.c_int]),
        "sdpsr_randomize": (C.c_int, [vp, i64, vp, vp, C.c_int]),
        "sdpsr_clamp_round": (C.c_int, [vp, i64, vp, dbl, C.c_int]),
        "sdpsr_project_out": (C.c_int, [vp, i64, vp, vp, i64, C.c_int]),
        "sdpsr_square_f64": (C.c_int, [vp, i64, vp, vp, C.c_int]),
        "sdpsr_square_f32": (C.c_int, [vp, i64, vp, vp, C.c_int]),
        "sdpsr_square_i8": (C.c_int, [vp, i64, vp, vp, C.c_int]),
        "sdpsr_gemm_tn_f64": (C.c_int, [vp, i64, i64, i64, vp, i64, vp, i64, vp, i64, C.c_int]),
        "sdpsr_admissible_subspace": (C.c_int, [vp, i64, vp, vp, vp, i64, dbl, vp, pi64, pi32, vp, C.c_int]),
        "sdpsr_admissible_subspace_dense": (C.c_int, [vp, i64, i64, vp, vp, vp, dbl, vp, pi64, pi32, vp, C.c_int]),
        "sdpsr_reduce_constraints": (C.c_int, [vp, i64, vp, i64, i64, vp, vp, C.c_int]),
        "sdpsr_desymmetrize": (C.c_int, [vp, i64, vp, pi64, pi32, C.c_int]),
        "sdpsr_block_diagonalize": (C.c_int, [vp, i64, vp, i64, dbl, pi32, pi64, pi64, vp, C.c_int]),
        "sdpsr_block_sizes": (C.c_int, [vp, vp]),
        "sdpsr_q_hat": (C.c_int, [vp, vp, C.c_int]),
        "sdpsr_block_images": (C.c_int, [vp, vp, vp, vp, C.c_int]),
        "sdpsr_eigen_decomposition": (C.c_int, [vp, i64, vp, i64, dbl, pi32, pi32, C.c_int]),
        "sdpsr_block_diagonalize_complex": (C.c_int, [vp, i64, vp, i64, dbl, vp, pi64, pi32, pi64, pi64, C.c_int]),
        "sdpsr_block_sizes_complex": (C.c_int, [vp, vp]),
        "sdpsr_block_images_complex": (C.c_int, [vp, vp, vp, C.c_int]),
        "sdpsr_eigen_decomposition_batched": (C.c_int, [vp, i64, vp, i64, dbl, i64, vp, vp, vp, vp, C.c_int]),
        "sdpsr_syev_f64": (C.c_int, [vp, i64, vp, vp, vp, C.c_int]),
        "sdpsr_hint_symmetric_basis": (C.c_int, [vp, C.c_int]),
        "sdpsr_profile_kernel": (C.c_int, [vp, C.c_int, i64, i64, C.c_int, C.POINTER(C.c_double)]),
        "sdpsr_profile_clock": (C.c_int, [vp, C.c_int, i64, i64, C.c_int, C.POINTER(C.c_double)]),
    }
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    if missing:
        raise RuntimeError(f"libsdpsr_hip.so lacks symbols declared in include/sdpsr.h: {missing}")
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
