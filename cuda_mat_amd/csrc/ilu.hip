// ilu.hip -- ILU(0) factorisation and level-scheduled triangular solves (placeholder:
// filled in by the next milestone; every entry point fails loudly until then).
#include "solver.h"

using namespace cm;

namespace cm {
int ilu0_setup(cudamat_solver *) { set_error("ILU(0) not built yet"); return CUDAMAT_ERR_ARG; }
int ilu0_release(cudamat_solver *) { return CUDAMAT_OK; }
int trsv_apply(cudamat_solver *, const TriFactor &, bool, const double *, double *)
{
    set_error("ILU(0) not built yet");
    return CUDAMAT_ERR_ARG;
}
}  // namespace cm

extern "C" int cudamat_solver_ilu0(cudamat_solver *s)
{
    CM_ARG(s, "solver is NULL");
    return ilu0_setup(s);
}
extern "C" int cudamat_solver_ilu0_values(cudamat_solver *s, double *out_dev)
{
    CM_ARG(s && out_dev, "null pointer");
    set_error("ILU(0) not built yet");
    return CUDAMAT_ERR_ARG;
}
