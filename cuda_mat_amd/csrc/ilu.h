// ilu.h -- what ilu.hip (analysis, factorisation, storage of the factors) and trsv.hip (their application) share:
// the host-side launch plan of a triangular factor.  Internal.
#pragma once
#include <vector>

#include "solver.h"
#include "spmv_pb.h"

namespace cm {

constexpr int kSmallLevel = 2048;        // levels up to this many rows may share a single-block launch
constexpr int kSpinLimit = 1 << 21;      // polls of one dependency before a row gives up (option TRSV_SPIN_LIMIT)
constexpr int kLdsTrsvRows = 16384;     // the largest system whose triangular solves run in one workgroup with the vector in LDS

// lanes per row of the triangular-solve kernels from the mean row length
inline int pick_lanes(double mean)
{
    if (mean <= 3.0) return 2;
    if (mean <= 6.0) return 4;
    if (mean <= 12.0) return 8;
    if (mean <= 40.0) return 16;
    if (mean <= 96.0) return 32;
    return 64;
}

struct TriHost {   // host-side launch plan kept next to the TriFactor
    std::vector<int> seg_begin, seg_end;   // level ranges; a segment with end-begin > 1 is a small-level run
    std::vector<int> seg_group;            // group of every segment (segments never straddle groups)
    int *level_ptr_dev = nullptr;
    int lanes = 8;
    // hybrid solve (see split_factor): levels are cut into a few consecutive GROUPS; entries whose column
    // belongs to an EARLIER group ("far") are applied per group by one blocked two-phase SpMV, only the
    // entries inside the group ("near") stay in the gather-based level kernels
    bool hybrid = false;
    std::vector<int> grp_level;            // K+1 level boundaries of the groups
    std::vector<PbPlan> far;               // far[g]: rows of group g x columns of groups < g
    double *far_buf = nullptr;             // n doubles in level-major row order: far_g . out
    int *lev_dev = nullptr;                // device: level of every original row (kept until the split)
    bool want_hybrid = false;              // this factor alone would take the hybrid solve (the two factors decide together)
    bool syncfree = false;                 // one dependency-driven launch per group instead of one launch per level
    int spin_limit = kSpinLimit;
    int nap = 2;                           // s_sleep between polls (0 / 1 / 2 / 4 measured equal within noise)
    int occ = 8;                           // workgroups per CU the dependency-driven launch may keep resident
    bool lds = false;                      // n <= 16384, narrow levels: the whole solve in one workgroup, x in LDS
    unsigned *tickets = nullptr;           // device: one chunk-ticket counter per dependency-driven launch (group)
    int pb_strict = 0;                     // Config::pb_strict at set-up: passed to the far parts' phase 2
};

}  // namespace cm

// the launch plans hang off the solver as an opaque pointer (keeps solver.h light)
struct IluPlans {
    cm::TriHost L, U;
    int *err_host = nullptr, *err_dev = nullptr;   // pinned word a timed-out spin of k_trsv_syncfree sets
    // between the pattern-only analysis and the numeric part of ilu0_setup (the drop-in call runs the former beside its upload)
    int *d_flags = nullptr, *d_lev = nullptr;
    int maxrow_all = 0;
    bool analysed = false;
    std::vector<void *> deferred;                   // temporaries of an analysis that ran beside an upload, freed after it
    // level-major index spaces (both factors hybrid): U-position of every original row (the column map of the permuted
    // matrix, solver.hip ensure_perm_matrix), scratch vectors of the original-space wrapper (precond_apply_any)
    int *posU = nullptr;
    double *perm_a = nullptr, *perm_b = nullptr;
};

inline IluPlans *plans_of(cudamat_solver *s, bool create)
{
    if (!s->ilu_plans && create) s->ilu_plans = new IluPlans();
    return (IluPlans *)s->ilu_plans;
}
