// config.cpp -- the option table of libcudamat_hip.so and the library's only look at the environment.
#include "config.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <unistd.h>

#include <mutex>
#include <set>
#include <string>

extern char **environ;

#include "common.h"

namespace cm {

namespace {

enum Kind { K_FLAG, K_INT, K_LLONG, K_ENUM, K_FAIL };

struct Option {
    const char *name;
    Kind kind;
    int Config::*field;            // K_FLAG / K_INT / K_ENUM
    long long Config::*wide;       // K_LLONG
    long long lo, hi;              // accepted range (K_INT / K_LLONG)
    const char *words;             // K_ENUM: "word=value,word=value"
    const char *help;
};

#define OPT_FLAG(NAME, FIELD, HELP) {NAME, K_FLAG, &Config::FIELD, nullptr, 0, 1, nullptr, HELP}
#define OPT_INT(NAME, FIELD, LO, HI, HELP) {NAME, K_INT, &Config::FIELD, nullptr, LO, HI, nullptr, HELP}
#define OPT_ENUM(NAME, FIELD, WORDS, HELP) {NAME, K_ENUM, &Config::FIELD, nullptr, 0, 0, WORDS, HELP}

const Option kOptions[] = {
    OPT_INT("VERBOSE", verbose, 0, 2, "1: set-up breakdowns and auto-tune timings on stderr; 2: also the memory classes of a large blocked copy's arrays (timed probe, diagnosis)"),
    OPT_FLAG("ROCTX", roctx, "roctx ranges around solver creation, SpMV-form choice, ILU(0) set-up, the loop, SpMVs and all-reduces"),
    OPT_ENUM("SPMV_MODE", spmv_mode, "csr=0,pb=1,sell=2,pat=3", "force the CSR forms / the blocked two-phase form / SELL-C-sigma / the row-pattern dictionary"),
    OPT_FLAG("SPMV_SELL", spmv_sell, "0: keep SELL-C-sigma out of the candidates"),
    OPT_ENUM("SPMV_FORM", spmv_form, "lanes=1,tiles=2", "inside the CSR forms: lanes per row / nnz-balanced tiles"),
    OPT_INT("SPMV_LANES", spmv_lanes, 2, 64, "lanes per row, a power of two (and no stream tiles)"),
    OPT_FLAG("SPMV_COMPRESS", spmv_compress, "0: stream kernel on the plain 32-bit indices"),
    OPT_FLAG("SPMV_ALIGN", spmv_align, "0: compressed stream kernel without line-aligned copies of its value / offset streams"),
    OPT_ENUM("SPMV_TUNE", spmv_tune_full, "full=1", "time every SpMV candidate even when the column span already decides"),
    OPT_FLAG("VALUE_DICT", value_dict, "0: keep fp64 values even when the matrix has <= 256 distinct ones"),
    OPT_INT("PB_MIN_WAVES", pb_min_waves, 256, 1 << 20, "blocked form: fewest phase-2 waves"),
    OPT_INT("PB_DEPTH", pb_depth, 4, 16, "blocked form: segment loads in flight per phase-2 wave (4, 8 or 16)"),
    OPT_FLAG("POOL", pool, "0: device memory straight from hipMalloc / hipFree instead of the library's recycling pool (read once per process)"),
    OPT_FLAG("SKIP_ORIG_COPY", skip_orig_copy, "0: cudamat_solve with ILU(0) builds the blocked copy in the original index space even when its loop will run in the level-major spaces"),
    OPT_FLAG("EARLY_ANALYSIS", early_analysis, "0: cudamat_solve analyses the ILU(0) levels after the upload of the values instead of beside it"),
    OPT_INT("PB_FILL_OCC", pb_fill_occ, 0, 64, "resident waves per CU of the two-pass fill's first pass (0: 8; probing)"),
    OPT_FLAG("PB_FILL2", pb_fill2, "0: fill the blocked copy with the single-pass scatter kernel (rounds 1-4) instead of the two-pass partition"),
    OPT_INT("PB_PLACE", pb_place, 0, 2, "blocked copy of a large matrix: 1 (default): a resident solver's product stream is placed in a memory class of its own, values and indices in another (timed probe of the device's memory classes, 10-35 ms per copy: spmv_pb.hip); 2: the drop-in calls place theirs as well; 0: arrays wherever the allocator puts them"),
    OPT_INT("PB_PLACE_MAX_MS", pb_place_max_ms, 0, 60000, "PB_PLACE: what the search for an arrangement may take before the arrays are allocated as ever (ms; the slabs it allocates included -- 0.3 ms per 16 GB on most boxes, 2 s on some)"),
    OPT_FLAG("PB_PROBE_FAIL", pb_probe_fail, "1 (tests): treat the run-time LDS-order probe as failed, i.e. select the architected-order phase 2"),
    OPT_FLAG("PB_STRICT", pb_strict, "1: phase 2 adds a row's products of one wave instruction rank by rank (architected order, +10 %)"),
    {"FUSED", K_LLONG, nullptr, &Config::fused, 0, 1LL << 40, nullptr,
     "0: never fold the vector updates into the SpMVs (three launches per iteration); N: do it for every supported plan up to N rows"},
    OPT_FLAG("RESIDENT", resident, "0: never run the whole loop of a very small system in one launch"),
    OPT_INT("RESIDENT_SPIN_LIMIT", resident_spin_limit, 0, 1 << 30, "polls a grid-barrier wait of the single-launch loop may take"),
    OPT_INT("PIPE_RR", pipe_rr, 0, 1 << 20, "residual replacement period of the pipelined loop (0: never)"),
    OPT_FLAG("TRSV_SYNCFREE", trsv_syncfree, "0: one launch per level; 1: dependency-driven launches also for narrow levels"),
    OPT_FLAG("TRSV_LDS", trsv_lds, "0: no single-workgroup LDS-resident triangular solve for small systems"),
    OPT_FLAG("TRSV_HYBRID", trsv_hybrid, "0 | 1: never / always split big factors into far (blocked SpMV) and near (dependency-driven) parts"),
    OPT_INT("TRSV_GROUPS", trsv_groups, 2, 128, "groups of levels of a hybrid factor"),
    OPT_INT("TRSV_LANES", trsv_lanes, 2, 64, "lanes per row of the triangular-solve kernels, a power of two"),
    OPT_INT("TRSV_SPIN_LIMIT", trsv_spin_limit, 1, 1 << 30, "polls of one dependency before a row gives up"),
    OPT_FLAG("TRSV_PERM", trsv_perm, "0: permute around every M^-1 application instead of running the loop in the level-major spaces"),
    OPT_FLAG("LEVELS_SWEEP", levels_sweep, "1: level analysis by relaxation sweeps"),
    OPT_FLAG("ILU0_SIMPLE", ilu0_simple, "1: numeric ILU(0) without LDS staging / prefetch"),
    OPT_FLAG("FORCE_SHARDED", force_sharded, "1: keep the collective path at world size 1"),
    OPT_FLAG("OVERLAP", overlap, "0: plain all-gather instead of pieces behind phase 1 of the blocked SpMV"),
    OPT_INT("OVERLAP_CHUNKS", overlap_chunks, 1, 16, "pieces per slice of an overlapped gather"),
    OPT_FLAG("WINDOWED", windowed, "0: never exchange windows (halo) only"),
    OPT_FLAG("SHARDED_ONE_DEVICE", sharded_one_device,
             "1: cudamat_solve_sharded with every rank on device 0 and host-synchronised copies in place of RCCL (debugging aid)"),
    OPT_FLAG("PLAN_CACHE", plan_cache, "0: cudamat_solve does not keep the solver of its last call"),
    {"TEST_COMM_FAIL", K_FAIL, nullptr, nullptr, 0, 0, nullptr, "rank:k -- fault injection: that rank's k-th all-reduce reports an error (tests)"},
};

bool power_of_two(long long v) { return v > 0 && (v & (v - 1)) == 0; }

bool apply(Config &cfg, const Option &o, const char *value)
{
    switch (o.kind) {
    case K_FLAG:
        if (!strcmp(value, "0") || !strcmp(value, "1")) { cfg.*(o.field) = value[0] - '0'; return true; }
        return false;
    case K_INT: {
        char *end = nullptr;
        const long long v = strtoll(value, &end, 10);
        if (end == value || *end || v < o.lo || v > o.hi) return false;
        if ((!strcmp(o.name, "SPMV_LANES") || !strcmp(o.name, "TRSV_LANES")) && !power_of_two(v)) return false;
        if (!strcmp(o.name, "PB_DEPTH") && v != 4 && v != 8 && v != 16) return false;
        cfg.*(o.field) = (int)v;
        return true;
    }
    case K_LLONG: {
        char *end = nullptr;
        const long long v = strtoll(value, &end, 10);
        if (end == value || *end || v < o.lo || v > o.hi) return false;
        cfg.*(o.wide) = v;
        return true;
    }
    case K_ENUM: {
        const size_t len = strlen(value);
        for (const char *w = o.words; w && *w;) {
            const char *eq = strchr(w, '=');
            if ((size_t)(eq - w) == len && !strncmp(w, value, len)) { cfg.*(o.field) = atoi(eq + 1); return true; }
            const char *comma = strchr(eq, ',');
            w = comma ? comma + 1 : nullptr;
        }
        return false;
    }
    case K_FAIL: {
        int r = -1, k = -1;
        if (sscanf(value, "%d:%d", &r, &k) != 2 || r < 0 || k < 1) return false;
        cfg.fail_rank = r;
        cfg.fail_call = k;
        return true;
    }
    }
    return false;
}

}  // namespace

bool Config::operator==(const Config &o) const
{
    for (const Option &q : kOptions) {
        if (q.field && this->*(q.field) != o.*(q.field)) return false;
        if (q.wide && this->*(q.wide) != o.*(q.wide)) return false;
    }
    return fail_rank == o.fail_rank && fail_call == o.fail_call;
}

bool config_set(Config &cfg, const char *name, const char *value)
{
    if (!name || !value) { set_error("option name / value is NULL"); return false; }
    if (!strncmp(name, "CUDAMAT_", 8)) name += 8;
    for (const Option &o : kOptions) {
        if (strcmp(o.name, name)) continue;
        if (apply(cfg, o, value)) return true;
        set_error("option %s: value \"%s\" is not accepted (%s)", name, value, o.help);
        return false;
    }
    set_error("unknown option \"%s\" (cudamat_options_help lists them)", name);
    return false;
}

// one stderr line per distinct complaint and process: an A/B script that passes its switches through the environment must
// not measure the default configuration unnoticed because a value was mistyped or a switch no longer exists
static void warn_once(const std::string &text)
{
    static std::mutex mu;
    static std::set<std::string> seen;
    std::lock_guard<std::mutex> lock(mu);
    if (seen.insert(text).second) fprintf(stderr, "cudamat: %s\n", text.c_str());
}

Config config_from_env()
{
    Config cfg;
    for (const Option &o : kOptions) {
        const std::string var = std::string("CUDAMAT_") + o.name;
        const char *v = getenv(var.c_str());          // the library's only getenv
        if (!v || !*v) continue;
        if (!apply(cfg, o, v))
            warn_once(var + "=\"" + v + "\" is not an accepted value and was IGNORED (" + o.help + ")");
    }
    // CUDAMAT_* variables that name no switch of this library (bench.py's own CUDAMAT_BENCH_* aside)
    for (char **e = environ; e && *e; e++) {
        if (strncmp(*e, "CUDAMAT_", 8) || !strncmp(*e, "CUDAMAT_BENCH_", 14)) continue;
        const char *eq = strchr(*e, '=');
        if (!eq || !eq[1]) continue;
        const std::string name(*e + 8, (size_t)(eq - (*e + 8)));
        bool known = false;
        for (const Option &o : kOptions) known = known || name == o.name;
        if (!known) warn_once("CUDAMAT_" + name + " names no switch of this library and was IGNORED (cudamat_options_help lists them)");
    }
    return cfg;
}

const char *config_help()
{
    static const std::string text = [] {
        std::string t;
        for (const Option &o : kOptions) {
            t += "CUDAMAT_";
            t += o.name;
            t += o.kind == K_FLAG ? " = 0 | 1" : o.kind == K_ENUM ? " = " : o.kind == K_FAIL ? " = rank:k" : " = N";
            if (o.kind == K_ENUM) {
                for (const char *w = o.words; w && *w;) {
                    const char *eq = strchr(w, '=');
                    if (w != o.words) t += " | ";
                    t.append(w, (size_t)(eq - w));
                    const char *comma = strchr(eq, ',');
                    w = comma ? comma + 1 : nullptr;
                }
            }
            if (o.kind == K_INT || o.kind == K_LLONG) t += " in [" + std::to_string(o.lo) + ", " + std::to_string(o.hi) + "]";
            t += "   ";
            t += o.help;
            t += "\n";
        }
        return t;
    }();
    return text.c_str();
}

}  // namespace cm
