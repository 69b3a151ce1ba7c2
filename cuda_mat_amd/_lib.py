"""ctypes binding of libcudamat_hip.so (the C ABI declared in include/cudamat.h).

The product has no CPU fallback: if the library is missing it is built with hipcc
(cross-compiles without a GPU); if that fails, or a compute entry point is called
without a HIP device, the call raises.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcudamat_hip.so")

OK = 0
PRECOND_NONE, PRECOND_ILU0, PRECOND_BLOCK_ILU0 = 0, 1, 2
LOOP_PBICGSTAB, LOOP_PBICGSTAB2, LOOP_PIPELINED = 0, 1, 2
FLAG_DEBUG, FLAG_PROFILE, FLAG_NO_EXIT, FLAG_X0_ONES = 1, 2, 4, 8


class CudamatError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("cudamat error %d: %s" % (code, msg))
        self.code = code


class Stats(C.Structure):
    """struct cudamat_stats (include/cudamat.h)"""
    _fields_ = [("iters", C.c_int), ("half_exit", C.c_int), ("converged", C.c_int),
                ("breakdown", C.c_int), ("nrm0", C.c_double), ("nrm", C.c_double),
                ("t_analysis", C.c_double), ("t_factor", C.c_double), ("t_solve", C.c_double),
                ("t_total", C.c_double), ("ms_spmv", C.c_double), ("n_spmv", C.c_int),
                ("ms_trsv", C.c_double), ("n_trsv", C.c_int), ("n_levels_l", C.c_int),
                ("n_levels_u", C.c_int), ("trsv_form", C.c_int), ("trsv_fallbacks", C.c_int),
                ("n_gather", C.c_int), ("n_allreduce", C.c_int), ("ms_gather", C.c_double),
                ("ms_gather_exposed", C.c_double), ("ms_allreduce", C.c_double), ("overlapped", C.c_int),
                ("loop_form", C.c_int), ("gather_fraction", C.c_double), ("ms_spmv_alone", C.c_double),
                ("loop_fallbacks", C.c_int), ("restarts", C.c_int), ("t_upload", C.c_double), ("t_setup", C.c_double),
                ("t_tune", C.c_double), ("spmv_mode", C.c_int), ("plan_reused", C.c_int),
                ("trsv_groups_l", C.c_int), ("trsv_groups_u", C.c_int)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)


GATHER_PART_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64)
_I64P = C.POINTER(C.c_int64)
GATHER_WINDOW_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, _I64P, _I64P, _I64P, _I64P)


class Comm(C.Structure):
    """struct cudamat_comm"""
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("user", C.c_void_p),
                ("allgather", ALLGATHER_FN), ("allreduce", ALLREDUCE_FN),
                ("gather_part", GATHER_PART_FN), ("comm_stream", C.c_void_p),
                ("allreduce_side", ALLREDUCE_FN), ("reduce_stream", C.c_void_p),
                ("gather_window", GATHER_WINDOW_FN)]


def build(force=False):
    """compile libcudamat_hip.so for gfx950 in-tree (make; hipcc --offload-arch=gfx950)"""
    if force:
        subprocess.run(["make", "-C", _HERE, "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", _HERE, "-j", "6"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libcudamat_hip.so failed:\n" + r.stdout + r.stderr)


_P = C.c_void_p
_SIGS = {
    "cudamat_version": (C.c_int, []),
    "cudamat_last_error": (C.c_char_p, []),
    "cudamat_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "cudamat_ctx_create": (C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    "cudamat_ctx_destroy": (C.c_int, [_P]),
    "cudamat_ctx_set_option": (C.c_int, [_P, C.c_char_p, C.c_char_p]),
    "cudamat_ctx_reset_options": (C.c_int, [_P]),
    "cudamat_options_help": (C.c_char_p, []),
    "cudamat_option_check": (C.c_int, [C.c_char_p, C.c_char_p]),
    "cudamat_ctx_sync": (C.c_int, [_P]),
    "cudamat_ctx_stream": (C.c_int, [_P, C.POINTER(_P)]),
    "cudamat_malloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "cudamat_free": (C.c_int, [_P, _P]),
    "cudamat_h2d": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "cudamat_d2h": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "cudamat_d2d": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "cudamat_memset": (C.c_int, [_P, _P, C.c_int, C.c_size_t]),
    "cudamat_timer_create": (C.c_int, [_P, C.POINTER(_P)]),
    "cudamat_timer_start": (C.c_int, [_P, _P]),
    "cudamat_timer_stop": (C.c_int, [_P, _P]),
    "cudamat_timer_elapsed_ms": (C.c_int, [_P, _P, C.POINTER(C.c_double)]),
    "cudamat_timer_destroy": (C.c_int, [_P, _P]),
    "cudamat_spmv": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, C.c_double, _P, _P, C.c_double, _P]),
    "cudamat_dot": (C.c_int, [_P, C.c_int64, _P, _P, _P]),
    "cudamat_nrm2": (C.c_int, [_P, C.c_int64, _P, _P]),
    "cudamat_axpy": (C.c_int, [_P, C.c_int64, C.c_double, _P, _P]),
    "cudamat_scal": (C.c_int, [_P, C.c_int64, C.c_double, _P]),
    "cudamat_solver_create": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int64, _P, _P, _P, C.c_int, C.POINTER(_P)]),
    "cudamat_solver_create_host": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int64, _P, _P, _P, C.c_int, C.POINTER(_P)]),
    "cudamat_solver_destroy": (C.c_int, [_P]),
    "cudamat_solver_set_shift": (C.c_int, [_P, _P]),
    "cudamat_solver_ilu0": (C.c_int, [_P]),
    "cudamat_solver_block_ilu0": (C.c_int, [_P]),
    "cudamat_solver_ilu0_values": (C.c_int, [_P, _P]),
    "cudamat_solver_ilu0_nnz": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "cudamat_solver_trsv_form": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "cudamat_solver_precond_apply": (C.c_int, [_P, _P, _P]),
    "cudamat_solver_set_comm": (C.c_int, [_P, C.POINTER(Comm)]),
    "cudamat_solve_sharded": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int,
                                        C.c_double, C.c_int, _P, C.POINTER(Stats)]),
    "cudamat_comm_dry_create": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(Comm)]),
    "cudamat_comm_dry_destroy": (C.c_int, [C.POINTER(Comm)]),
    "cudamat_rccl_available": (C.c_int, []),
    "cudamat_rccl_unique_id": (C.c_int, [_P]),
    "cudamat_rccl_comm_create": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(Comm)]),
    "cudamat_rccl_comm_destroy": (C.c_int, [C.POINTER(Comm)]),
    "cudamat_rccl_comm_abort": (C.c_int, [C.POINTER(Comm)]),
    "cudamat_solver_spmv_mode": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "cudamat_solver_spmv_kernel": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "cudamat_solver_value_dict": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "cudamat_solver_placement": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_char_p, C.c_int]),
    "cudamat_solver_spmv": (C.c_int, [_P, _P, _P]),
    "cudamat_solver_solve": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                       C.POINTER(Stats)]),
    "cudamat_solver_history": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int)]),
    "cudamat_solve": (C.c_int, [C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int,
                                C.c_double, C.c_int, _P, C.POINTER(Stats)]),
    "cudamat_plan_cache_clear": (C.c_int, []),
    "cudamat_pool_trim": (C.c_int, []),
    "cudamat_mem_info": (C.c_int, [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "cudamat_poisson5_nnz": (C.c_int64, [C.c_int, C.c_int]),
    "cudamat_gen_poisson5": (C.c_int, [_P, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, _P, _P, _P]),
    "cudamat_rand_row_nnz": (C.c_int, [C.c_int64, C.c_int]),
    "cudamat_gen_rand_rows": (C.c_int, [_P, C.c_int64, C.c_int, C.c_uint64, C.c_int64, C.c_int64,
                                        C.c_int, _P, _P, _P]),
    "cudamat_gen_xstar": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_uint64, _P]),
    "cudamat_load_mtx": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_double)),
                                   C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_int))]),
    "cudamat_host_free": (None, [_P]),
    "cudamat_to_dense_vector": (None, [C.c_int, C.c_int, _P, _P, _P]),
}

_lib = None


def lib():
    """load (building first if needed) the HIP library; raises when that is impossible"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)   # AttributeError here = ABI drift; fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        raise CudamatError(rc, lib().cudamat_last_error().decode(errors="replace"))


def device_count():
    n = C.c_int(0)
    rc = lib().cudamat_device_count(C.byref(n))
    return n.value if rc == OK else 0
