// config.h -- every switch of libcudamat_hip.so in ONE place.
//
// A cudamat_ctx owns a Config.  It is filled from the environment (CUDAMAT_<NAME>) when the context is created --
// config.cpp holds the library's only getenv -- and changed afterwards only through cudamat_ctx_set_option(ctx,
// "<NAME>", "<value>").  Every consumer reads the context's Config at the moment it acts (solver creation, choice of
// the SpMV form, ILU(0) set-up, a solve), so an option set between two steps applies to the later one; nothing in
// the library caches a switch in a static.  Entry points without a caller-made context (cudamat_solve,
// cudamat_solve_sharded) read the environment per call.  The table of names, values and defaults is kOptions in
// config.cpp (printed by cudamat_options_help, listed in DESIGN.md "Switches").
//
// All switches are for testing and probing: the defaults are what ships, and no default depends on one.
#pragma once

namespace cm {

struct Config {
    // ---- diagnostics
    int verbose = 0;              // VERBOSE            set-up breakdowns and auto-tune timings on stderr
    int roctx = 0;                // ROCTX              roctx ranges around the phases of a solve
    // ---- SpMV form
    int spmv_mode = -1;           // SPMV_MODE          csr | pb | sell | pat: force the CSR forms / blocked two-phase / SELL-C-sigma / row patterns (-1: choose)
    int spmv_sell = 1;            // SPMV_SELL          0: keep SELL out of the candidates
    int spmv_form = 0;            // SPMV_FORM          lanes | tiles: inside the CSR forms, lanes per row / nnz-balanced tiles (0: choose)
    int spmv_lanes = 0;           // SPMV_LANES         2..64 lanes per row (and no stream tiles)
    int spmv_compress = 1;        // SPMV_COMPRESS      0: stream kernel on the plain 32-bit indices
    int spmv_align = 1;           // SPMV_ALIGN         0: compressed stream kernel without line-aligned copies of its streams
    int spmv_tune_full = 0;       // SPMV_TUNE          full: time every candidate even when the column span already decides
    int value_dict = 1;           // VALUE_DICT         0: keep fp64 values even when the matrix has <= 256 distinct ones
    int pb_min_waves = 0;         // PB_MIN_WAVES       blocked form: fewest phase-2 waves (0: 2048 / 4096 by shape)
    int pb_depth = 0;             // PB_DEPTH           blocked form: segment loads in flight per wave, 4 | 8 | 16 (0: by shape)
    int pool = 1;                 // POOL               0: every allocation goes to the runtime (read once per process)
    int skip_orig_copy = 1;       // SKIP_ORIG_COPY     0: the preconditioned drop-in call builds the original-space blocked copy although its loop runs on the permuted one
    int early_analysis = 1;       // EARLY_ANALYSIS     0: the drop-in call runs the ILU(0) level analysis after its upload, not beside it
    int pb_fill_occ = 0;          // PB_FILL_OCC        resident waves per CU of the two-pass fill's first pass (0: 8)
    int pb_fill2 = 1;             // PB_FILL2           0: the blocked copy is filled by the single-pass kernel of rounds 1-4
    int pb_place = 1;             // PB_PLACE           large blocked copies: product stream in a memory class of its own (1: resident solvers, 2: drop-in calls too, 0: off)
    int pb_place_max_ms = 300;    // PB_PLACE_MAX_MS    what the placement search may take (allocations included: some boxes allocate at 8 GB/s)
    int pb_probe_fail = 0;        // PB_PROBE_FAIL      1 (tests): the LDS-order probe reports "not lane order"
    int pb_strict = 0;            // PB_STRICT          1: phase 2 adds a row's products of one wave instruction rank by rank (architected order)
    // ---- loop forms
    long long fused = -1;         // FUSED              0: never fold the vector updates into the SpMVs; N: do it up to N rows (-1: short rows up to 3e5)
    int resident = 1;             // RESIDENT           0: never run the whole loop of a very small system in one launch
    int resident_spin_limit = -1; // RESIDENT_SPIN_LIMIT polls a grid-barrier wait of that loop may take (-1: 2^22)
    int pipe_rr = -1;             // PIPE_RR            residual replacement period of the pipelined loop (-1: 32; 0: never)
    // ---- ILU(0) and the triangular solves
    int trsv_syncfree = -1;       // TRSV_SYNCFREE      0: one launch per level; 1: dependency-driven launches also for narrow levels (-1: by level width)
    int trsv_lds = 1;             // TRSV_LDS           0: no single-workgroup LDS-resident solve for small systems
    int trsv_hybrid = -1;         // TRSV_HYBRID        0 | 1: far / near split of big factors (-1: by shape)
    int trsv_groups = 0;          // TRSV_GROUPS        groups of levels of a hybrid factor (0: levels / 17, 2..16)
    int trsv_lanes = 0;           // TRSV_LANES         lanes per row of the solve kernels (0: by row length)
    int trsv_spin_limit = 0;      // TRSV_SPIN_LIMIT    polls of one dependency before a row gives up (0: 2^21)
    int trsv_perm = 1;            // TRSV_PERM          0: permute around every M^-1 instead of running the loop in the level-major spaces
    int levels_sweep = 0;         // LEVELS_SWEEP       1: level analysis by relaxation sweeps
    int ilu0_simple = 0;          // ILU0_SIMPLE        1: numeric ILU(0) without LDS staging / prefetch
    // ---- row sharding
    int force_sharded = 0;        // FORCE_SHARDED      1: keep the collective path at world size 1
    int overlap = 1;              // OVERLAP            0: plain all-gather instead of pieces behind phase 1
    int overlap_chunks = 0;       // OVERLAP_CHUNKS     pieces per slice, 1..16 (0: 4)
    int windowed = 1;             // WINDOWED           0: never exchange windows (halo) only
    int sharded_one_device = 0;   // SHARDED_ONE_DEVICE 1: cudamat_solve_sharded with every rank on device 0, host-synchronised copies for RCCL
    // ---- drop-in entry point
    int plan_cache = 1;           // PLAN_CACHE         0: cudamat_solve does not keep the solver of its last call
    // ---- fault injection (tests)
    int fail_rank = -1, fail_call = -1;   // TEST_COMM_FAIL = rank:k   that rank's k-th all-reduce reports an error

    bool operator==(const Config &o) const;
    bool operator!=(const Config &o) const { return !(*this == o); }
};

Config config_from_env();
// "<NAME>" as in the table (without the CUDAMAT_ prefix; with it is accepted too).  Returns false (and sets the
// library's error string) for an unknown name or a value outside the option's range.
bool config_set(Config &cfg, const char *name, const char *value);
// one line per option: name, accepted values, default, meaning
const char *config_help();

}  // namespace cm
