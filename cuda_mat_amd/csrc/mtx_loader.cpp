// mtx_loader.cpp -- Matrix Market coordinate reader behind cudamat_load_mtx and the
// drop-in loadMMSparseMatrix (include/mmio_wrapper.h).  Host only, no HIP.
//
// Behaviour follows the reference loader, mmio_wrapper.h:133-348 on top of NIST
// mmio.c (banner :103-186, size line :198-225, entries :277-311):
//   * real / integer coordinate matrices; complex (needs elem type 'z'/'c'),
//     pattern and dense/array files are rejected (:161-169);
//   * symmetric / hermitian / skew-symmetric storage is expanded to the full
//     pattern, the mirrored skew entry negated (:172-230);
//   * entries are sorted row-major (CSR) or column-major (CSC) (:240-264);
//   * the index base is auto-detected: an index 0 anywhere => base 0, an index
//     equal to the dimension anywhere => base 1, both => error, NEITHER => base 0
//     (:266-289 -- so a 1-based file whose last row and column are empty is
//     read as base 0, exactly like the reference);
//   * the result must pass the pattern check (:91-130): consistent nnz, base in
//     {0,1}, strictly increasing indices inside each row (duplicates => error).
// Outputs are malloc'd so that callers release them with free(), as example.cpp:370-374 does.
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "cudamat.h"

namespace {

struct Entry {
    int i, j;
    double v;
};

std::string lower(std::string s)
{
    for (char &c : s) c = (char)std::tolower((unsigned char)c);
    return s;
}

void close_file(std::FILE *&f)
{
    if (f) std::fclose(f);
    f = nullptr;
}

int fail(const char *msg, const char *file)
{
    std::fprintf(stderr, "!!!! %s: '%s'\n", msg, file);
    return CUDAMAT_ERR_IO;
}

}  // namespace

static int load_mtx(std::FILE *&f, const char *filename, int csr_format, int *m, int *n, int *nnz,
                    double **val, int **row, int **col);

extern "C" int cudamat_load_mtx(const char *filename, int csr_format, int *m, int *n, int *nnz,
                                double **val, int **row, int **col)
{
    if (!filename || !m || !n || !nnz || !val || !row || !col) return CUDAMAT_ERR_ARG;
    std::FILE *f = nullptr;
    int rc;
    try {                                   // nothing may unwind through the C boundary
        rc = load_mtx(f, filename, csr_format, m, n, nnz, val, row, col);
    } catch (const std::bad_alloc &) {
        std::fprintf(stderr, "!!!! out of memory while reading '%s'\n", filename);
        rc = CUDAMAT_ERR_NOMEM;
    }
    if (f) std::fclose(f);                  // (every path inside sets f to NULL after closing it)
    return rc;
}

static int load_mtx(std::FILE *&f, const char *filename, int csr_format, int *m, int *n, int *nnz,
                    double **val, int **row, int **col)
{
    f = std::fopen(filename, "r");
    if (!f) return fail("can not open file", filename);                 // mmio_wrapper.h:156
    char line[1100];
    char t0[80], t1[80], t2[80], t3[80], t4[80];
    if (!std::fgets(line, 1025, f) ||
        std::sscanf(line, "%64s %64s %64s %64s %64s", t0, t1, t2, t3, t4) != 5) {
        close_file(f);
        return fail("can not open file", filename);
    }
    const std::string object = lower(t1), format = lower(t2), field = lower(t3), symm = lower(t4);
    const bool field_ok = field == "real" || field == "integer" || field == "complex" || field == "pattern";
    const bool symm_ok = symm == "general" || symm == "symmetric" || symm == "hermitian" || symm == "skew-symmetric";
    if (std::strncmp(t0, "%%MatrixMarket", 14) != 0 || object != "matrix" || !field_ok || !symm_ok ||
        (format != "coordinate" && format != "array")) {
        close_file(f);
        return fail("can not open file", filename);
    }
    if (format != "coordinate" || (field == "real" && symm == "hermitian")) {   // mmio.c:92-101,352
        close_file(f);
        return fail("can not open file", filename);
    }
    if (field == "complex") {
        close_file(f);
        std::fprintf(stderr, "!!!! complex matrix requires type 'z' or 'c'\n");          // :162
        return CUDAMAT_ERR_IO;
    }
    if (field == "pattern") {
        close_file(f);
        std::fprintf(stderr, "!!!! dense, array, pattern and integer matrices are not supported\n");  // :167
        return CUDAMAT_ERR_IO;
    }
    int M = 0, N = 0, nz = 0;
    do {
        if (!std::fgets(line, 1025, f)) { close_file(f); return fail("can not open file", filename); }
    } while (line[0] == '%');
    if (std::sscanf(line, "%d %d %d", &M, &N, &nz) != 3) {
        int got;
        do {
            got = std::fscanf(f, "%d %d %d", &M, &N, &nz);
            if (got == EOF) { close_file(f); return fail("can not open file", filename); }
        } while (got != 3);
    }
    if (M < 0 || N < 0 || nz < 0) { close_file(f); return fail("can not open file", filename); }
    const bool mirror = symm != "general";
    // the interface counts entries in an int (mmio_wrapper.h:139): a header whose (mirrored) entry count
    // cannot be represented is refused before anything is allocated for it
    if ((long long)nz * (mirror ? 2 : 1) > (long long)INT_MAX) {
        close_file(f);
        std::fprintf(stderr, "!!!! '%s': %d stored entries%s do not fit the int interface\n", filename, nz,
                     mirror ? " (mirrored)" : "");
        return CUDAMAT_ERR_IO;
    }
    const bool skew = symm == "skew-symmetric";
    std::vector<Entry> e;
    e.reserve(std::min<size_t>((size_t)nz * (mirror ? 2 : 1), (size_t)1 << 26));   // a lying header must not reserve GBs
    for (int k = 0; k < nz; k++) {
        Entry t;
        if (std::fscanf(f, "%d %d %lg", &t.i, &t.j, &t.v) != 3) {
            close_file(f);
            return fail("can not open file", filename);
        }
        e.push_back(t);
        if (mirror && t.i != t.j) e.push_back(Entry{t.j, t.i, skew ? -t.v : t.v});
    }
    close_file(f);

    if (csr_format)
        std::stable_sort(e.begin(), e.end(), [](const Entry &a, const Entry &b) {
            return a.i != b.i ? a.i < b.i : a.j < b.j;
        });
    else
        std::stable_sort(e.begin(), e.end(), [](const Entry &a, const Entry &b) {
            return a.j != b.j ? a.j < b.j : a.i < b.i;
        });

    bool base0 = false, base1 = false;
    for (const Entry &t : e) {
        if (t.i == 0 || t.j == 0) base0 = true;
        if (t.i == M || t.j == N) base1 = true;
    }
    if (base0 && base1) {
        std::printf("Error: input matrix is base-0 and base-1 \n");                      // :282
        return CUDAMAT_ERR_IO;
    }
    const int base = base1 ? 1 : 0;
    const int total = (int)e.size();
    const int dim = csr_format ? M : N;
    // every major index must fall inside [base, base+dim): the reference would write
    // out of bounds here (mmio_wrapper.h:40); this loader reports it instead.
    const int odim = csr_format ? N : M;
    for (const Entry &t : e) {
        const int major = csr_format ? t.i : t.j, minor = csr_format ? t.j : t.i;
        if (major < base || major >= base + dim || minor >= base + odim) {
            std::fprintf(stderr, "!!!! verify_pattern failed\n");
            return CUDAMAT_ERR_IO;
        }
    }
    int *ptr = (int *)std::calloc((size_t)dim + 1, sizeof(int));
    int *idx = (int *)std::malloc(sizeof(int) * (size_t)(total > 0 ? total : 1));
    double *ov = (double *)std::malloc(sizeof(double) * (size_t)(total > 0 ? total : 1));
    if (!ptr || !idx || !ov) { std::free(ptr); std::free(idx); std::free(ov); return CUDAMAT_ERR_NOMEM; }
    ptr[0] = base;
    for (const Entry &t : e) ptr[(csr_format ? t.i : t.j) - base + 1]++;
    for (int k = 0; k < dim; k++) ptr[k + 1] += ptr[k];
    for (int k = 0; k < total; k++) {
        idx[k] = csr_format ? e[k].j : e[k].i;
        ov[k] = e[k].v;
    }
    bool bad = false;
    for (int r = 0; r < dim && !bad; r++)
        for (int c = ptr[r] - base; c < ptr[r + 1] - base && !bad; c++) {
            if (idx[c] < base) {
                std::fprintf(stderr, "Error (column vs. base index check failed): csrColInd[%d] < %d\n", c, base);
                bad = true;
            } else if (c + 1 < ptr[r + 1] - base && idx[c] >= idx[c + 1]) {
                std::fprintf(stderr, "Error (sorting of the column indecis check failed): (csrColInd[%d]=%d) >= (csrColInd[%d]=%d)\n",
                             c, idx[c], c + 1, idx[c + 1]);
                bad = true;
            }
        }
    if (bad) {
        std::fprintf(stderr, "!!!! verify_pattern failed\n");                            // :337
        std::free(ptr); std::free(idx); std::free(ov);
        return CUDAMAT_ERR_IO;
    }
    *m = M; *n = N; *nnz = total; *val = ov;
    if (csr_format) { *row = ptr; *col = idx; }
    else            { *col = ptr; *row = idx; }
    return CUDAMAT_OK;
}

extern "C" void cudamat_host_free(void *p) { std::free(p); }

// pbicgstab.cu:1101-1115: an n x 1 CSR "column vector" -> dense, missing rows = 0
extern "C" void cudamat_to_dense_vector(int n, int nnz, const double *A, const int *IA, double *out)
{
    (void)nnz;
    int seen = IA[0];
    int next = 0;
    for (int i = 0; i < n; ++i) {
        const bool has = IA[i + 1] - seen > 0;
        out[i] = has ? A[next++] : 0.0;
        if (has) seen = IA[i + 1];
    }
}
