"""cuda_mat_amd -- MI355X-native BiCGSTAB (the hot path of Russoul/cuda-mat) behind the
reference's own entry points.  Compute = hand-written gfx950 HIP kernels in
libcudamat_hip.so, reached through the C ABI of include/cudamat.h; no CPU fallback."""
from ._lib import (FLAG_DEBUG, FLAG_NO_EXIT, FLAG_PROFILE, FLAG_X0_ONES, LOOP_PBICGSTAB,  # noqa: F401
                   LOOP_PBICGSTAB2, LOOP_PIPELINED, PRECOND_BLOCK_ILU0, PRECOND_ILU0, PRECOND_NONE, Comm, CudamatError, Stats, build,
                   device_count, lib)
from .api import (Context, DeviceArray, Solver, Timer, bicgstab, bicgstab_d,  # noqa: F401
                  bicgstab_lu_precond, loadMMSparseMatrix, toDenseVector, use_gpus)
