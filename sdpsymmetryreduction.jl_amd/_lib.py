"""ctypes binding of libsdpsr_hip.so (C ABI: include/sdpsr.h).  The library is the
product; this file only declares argument types.  There is no CPU fallback: if the
shared object is missing the import of the binding fails loudly."""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsdpsr_hip.so")
PROF_LIB_PATH = os.path.join(_HERE, "libsdpsr_prof.so")  # measurement entry points, not the product
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "sdpsr.h")
PROF_HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "sdpsr_prof.h")

MEM_HOST, MEM_DEVICE = 0, 1
SQUARE_AUTO, SQUARE_I8, SQUARE_F32, SQUARE_F64 = 0, 1, 2, 3
T_TOTAL, T_PROJECT, T_SQUARE, T_REFINE, T_EIGEN, T_ISO, T_IRRED, T_IMAGE, T_COUNT = range(9)
ROUND_NEAREST, ROUND_TRUNC = 0, 1
# sdpsr_opts.flags
FLAG_SEPARATE_REFINEMENTS = 1 << 0
FLAG_FRESH_IRREDUCIBLE_ELEMENT = 1 << 1
FLAG_ALWAYS_REORTHOGONALIZE = 1 << 2
FLAG_REFINE_NO_FUSE = 1 << 3
FLAG_UNPACK_EVERY_STEP = 1 << 4
FLAG_SPMM_ONE_BY_ONE = 1 << 5
FLAG_SINGLE_COUPLING_ELEMENT = 1 << 6
FLAG_SMALL_EIGEN_ON_DEVICE = 1 << 7
FLAG_NO_GRAPH = 1 << 8
FLAG_NO_VERIFY_SHORTCUT = 1 << 9
FLAG_FULL_BASIS_IMAGE = 1 << 10
FLAG_ALWAYS_PROJECT = 1 << 11
FLAG_SYTRD_PANELS = 1 << 12
FLAG_COUPLING_ON_HOST = 1 << 13
FLAG_SYTRD_ONE_LAUNCH = 1 << 14
FLAG_WAIT_FOR_EVERY_VERDICT = 1 << 15
BASIS_IMAGE_KERNELS = {"auto": 0, "two_stage": 1, "outer": 2, "chunk": 3}
REFINE_PATHS = {"auto": 0, "hash": 1, "sort": 2, "bucket": 3, "no_mid": 4, "mid_no_first": 6}

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
        ("flags", C.c_uint32),
        ("round_mode", C.c_int32),
        ("basis_image_kernel", C.c_int32),
        ("refine_path", C.c_int32),
        ("label_bits", C.c_int32),
        ("insert_wgs_per_cu", C.c_int32),
        ("square_kernel", C.c_int32),
        ("reserved", C.c_int32 * 3),
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
    # One HIP runtime per process.  The library asks for `libamdhip64.so.7`; torch's wheels carry their own copy with that
    # SONAME and load it under the file name `libamdhip64.so`.  torch first: the loader hands the library torch's copy (the
    # SONAMEs match) and device pointers, streams and events are shared.  The library first: /opt/rocm's runtime is loaded,
    # a later `import torch` loads its own copy beside it, and that second runtime finds no GPU ("No HIP GPUs are
    # available", tools/gpu/torch_after_lib.py).  So torch, where installed, is imported before the library is opened.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
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
        "sdpsr_dimension_trajectory": (C.c_int, [vp, vp, i32, pi32]),
        "sdpsr_partition_from_f64": (C.c_int, [vp, i64, vp, vp, pi64, C.c_int]),
        "sdpsr_partition_from_u32": (C.c_int, [vp, i64, vp, vp, pi64, C.c_int]),
        "sdpsr_partition_from_u64": (C.c_int, [vp, i64, vp, vp, pi64, C.c_int]),
        "sdpsr_refine": (C.c_int, [vp, i64, vp, pi64, vp, i64, C.c_int]),
        "sdpsr_partition_checksum": (C.c_int, [vp, i64, vp, vp, C.c_int]),
        "sdpsr_fill": (C.c_int, [vp, i64, vp, vp, i64, vp, C.c_int]),
        "sdpsr_randomize": (C.c_int, [vp, i64, vp, vp, C.c_int]),
        "sdpsr_clamp_round": (C.c_int, [vp, i64, vp, dbl, C.c_int]),
        "sdpsr_project_out": (C.c_int, [vp, i64, vp, vp, i64, C.c_int]),
        "sdpsr_square_f64": (C.c_int, [vp, i64, vp, vp, C.c_int]),
        "sdpsr_square_f32": (C.c_int, [vp, i64, vp, vp, C.c_int]),
        "sdpsr_square_i8": (C.c_int, [vp, i64, vp, vp, C.c_int]),
        "sdpsr_square_i8_symmetric": (C.c_int, [vp, i64, i64, vp, vp, C.c_int]),
        "sdpsr_gemm_tn_f64": (C.c_int, [vp, i64, i64, i64, vp, i64, vp, i64, vp, i64, C.c_int]),
        "sdpsr_admissible_subspace": (C.c_int, [vp, i64, vp, vp, vp, i64, dbl, vp, pi64, pi32, vp, C.c_int]),
        "sdpsr_admissible_subspace_dense": (C.c_int, [vp, i64, i64, vp, vp, vp, dbl, vp, pi64, pi32, vp, C.c_int]),
        "sdpsr_jordan_reduce": (C.c_int, [vp, i64, vp, vp, vp, i64, dbl, dbl, vp, pi64, pi32, pi32, pi64, pi64, vp, i64, vp, i64, vp, C.c_int]),
        "sdpsr_jordan_reduce_batch": (C.c_int, [vp, C.c_int32, vp, i64, vp, vp, vp, i64, dbl, dbl, vp, pi64, pi32, pi32, pi64, pi64, vp, vp, pi32,
                                                C.c_int]),
        "sdpsr_problem_create": (C.c_int, [vp, i64, vp, vp, vp, i64, C.c_int, C.c_int, C.POINTER(vp)]),
        "sdpsr_problem_destroy": (C.c_int, [vp]),
        "sdpsr_problem_reduce": (C.c_int, [vp, vp, dbl, dbl, vp, pi64, pi32, pi32, pi64, pi64, vp, i64, vp, i64, vp, C.c_int]),
        "sdpsr_problem_reduce_batch": (C.c_int, [vp, vp, C.c_int32, vp, dbl, dbl, vp, pi64, pi32, pi32, pi64, pi64, vp, vp, pi32, C.c_int]),
        "sdpsr_batch_block_sizes": (C.c_int, [vp, C.c_int32, vp]),
        "sdpsr_transfer_bytes": (C.c_int, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
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


_prof = None


def load_prof_library():
    """libsdpsr_prof.so (include/sdpsr_prof.h): per-kernel timing entry points for bench.py's roofline
    leg and tools/ -- built beside the product library, never needed by the product path."""
    global _prof
    if _prof is not None:
        return _prof
    load_library()  # libsdpsr_prof.so links against libsdpsr_hip.so ($ORIGIN rpath)
    if not os.path.exists(PROF_LIB_PATH):
        raise RuntimeError(f"{PROF_LIB_PATH} is missing: build it with `make -C sdpsymmetryreduction.jl_amd/csrc`")
    lib = C.CDLL(PROF_LIB_PATH)
    vp, i64 = C.c_void_p, C.c_int64
    for name in ("sdpsr_profile_kernel", "sdpsr_profile_clock"):
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = [vp, C.c_int, i64, i64, C.c_int, C.POINTER(C.c_double)]
    lib.sdpsr_profile_band_chase.restype = C.c_int
    lib.sdpsr_profile_band_chase.argtypes = [vp, i64, C.c_int, vp, vp, vp, C.POINTER(C.c_double)]
    lib.sdpsr_profile_band_reduce.restype = C.c_int
    lib.sdpsr_profile_band_reduce.argtypes = [vp, i64, C.c_int, vp, C.POINTER(C.c_double)]
    lib.sdpsr_profile_host_waits.restype = C.c_int
    lib.sdpsr_profile_host_waits.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.sdpsr_profile_sytrd_graphs.restype = C.c_int
    lib.sdpsr_profile_sytrd_graphs.argtypes = [vp, C.POINTER(C.c_double)]
    _prof = lib
    return lib
