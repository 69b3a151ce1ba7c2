// selftest.cpp -- host-only checks that run under AddressSanitizer (make -C cuda_mat_amd/host asan-check):
// the Matrix Market loader on the shipped fixtures and on hostile files, toDenseVector, and Matrix.h.
// No GPU is touched: sanitizers are for the CPU side (the pool has no GPU AddressSanitizer).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "Matrix.h"
#include "cudamat.h"

static int failures = 0;
#define EXPECT(cond)                                                       \
    do {                                                                   \
        if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } \
    } while (0)

static std::string write_file(const std::string &dir, const char *name, const char *text)
{
    const std::string path = dir + "/" + name;
    FILE *f = std::fopen(path.c_str(), "w");
    if (f) { std::fputs(text, f); std::fclose(f); }
    return path;
}

static int load(const std::string &path, bool csr, int *m, int *n, int *nnz, double **v, int **r, int **c)
{
    return cudamat_load_mtx(path.c_str(), csr ? 1 : 0, m, n, nnz, v, r, c);
}

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: selftest <golden dir> <scratch dir>\n"); return 2; }
    const std::string gold = argv[1], tmp = argv[2];
    int m, n, nnz;
    double *v;
    int *r, *c;
    // shipped fixtures: sizes the reference's loader reports
    struct { const char *file; int n, nnz; } fixtures[] = {{"mat3.mtx", 3, 8}, {"mat900.mtx", 900, 7744}, {"mat10000.mtx", 10000, 49600}};
    for (auto &fx : fixtures) {
        for (int csr = 0; csr < 2; csr++) {
            v = nullptr; r = c = nullptr;
            EXPECT(load(gold + "/" + fx.file, csr, &m, &n, &nnz, &v, &r, &c) == CUDAMAT_OK);
            EXPECT(m == fx.n && n == fx.n && nnz == fx.nnz);
            if (v) {
                const int *ptr = csr ? r : c;
                EXPECT(ptr[0] == 1 && ptr[fx.n] == 1 + fx.nnz);
            }
            cudamat_host_free(v); cudamat_host_free(r); cudamat_host_free(c);
        }
    }
    // vec3_d: rows 1 and 3 only -> dense [1, 0, 1]
    v = nullptr; r = c = nullptr;
    EXPECT(load(gold + "/vec3_d.mtx", true, &m, &n, &nnz, &v, &r, &c) == CUDAMAT_OK);
    if (v) {
        double dense[3] = {-1, -1, -1};
        cudamat_to_dense_vector(m, nnz, v, r, dense);
        EXPECT(dense[0] == 1.0 && dense[1] == 0.0 && dense[2] == 1.0);
    }
    cudamat_host_free(v); cudamat_host_free(r); cudamat_host_free(c);
    // hostile files: every one must come back as an error, with nothing leaked or overrun
    const char *bad[][2] = {
        {"huge.mtx", "%%MatrixMarket matrix coordinate real general\n3 3 2147483647\n1 1 1\n"},
        {"hugesym.mtx", "%%MatrixMarket matrix coordinate real symmetric\n3 3 1500000000\n1 1 1\n"},
        {"widecol.mtx", "%%MatrixMarket matrix coordinate real general\n3 3 2\n1 1 1\n2 7 2\n"},
        {"widerow.mtx", "%%MatrixMarket matrix coordinate real general\n3 3 2\n1 1 1\n9 2 2\n"},
        {"dup.mtx", "%%MatrixMarket matrix coordinate real general\n3 3 3\n1 1 1\n1 1 2\n3 3 1\n"},
        {"short.mtx", "%%MatrixMarket matrix coordinate real general\n3 3 3\n1 1 1\n2 2 2\n"},
        {"both.mtx", "%%MatrixMarket matrix coordinate real general\n3 3 2\n0 1 1\n3 3 2\n"},
        {"empty.mtx", ""},
    };
    for (auto &b : bad) {
        v = nullptr; r = c = nullptr;
        EXPECT(load(write_file(tmp, b[0], b[1]), true, &m, &n, &nnz, &v, &r, &c) != CUDAMAT_OK);
    }
    EXPECT(load(tmp + "/does_not_exist.mtx", true, &m, &n, &nnz, &v, &r, &c) != CUDAMAT_OK);
    // symmetric expansion incl. skew sign, on a file written here
    v = nullptr; r = c = nullptr;
    EXPECT(load(write_file(tmp, "skew.mtx", "%%MatrixMarket matrix coordinate real skew-symmetric\n3 3 2\n2 1 5\n3 3 7\n"), true,
                &m, &n, &nnz, &v, &r, &c) == CUDAMAT_OK);
    EXPECT(nnz == 3);
    if (v && nnz == 3) EXPECT(v[0] == -5.0 && v[1] == 5.0 && v[2] == 7.0);
    cudamat_host_free(v); cudamat_host_free(r); cudamat_host_free(c);

    // Matrix.h
    Matrix<double> A(2, 3, {1, 2, 3, 4, 5, 6});
    Matrix<double> At = Matrix<double>::transpose(A);
    EXPECT(At.n == 3 && At.m == 2 && At.get(2, 0) == 3 && At.get(0, 1) == 4);
    Matrix<double> G = mul(A, At);                 // [[14, 32], [32, 77]]
    EXPECT(G.get(0, 0) == 14 && G.get(0, 1) == 32 && G.get(1, 1) == 77);
    EXPECT(mul(Matrix<double>::identity(2), A).get(1, 2) == 6);
    EXPECT(Matrix<double>::without_column(A, 1).get(1, 1) == 6 && A.column(2).get(1, 0) == 6 && A.row(1).get(0, 0) == 4);
    EXPECT(Matrix<double>::is_zero(Matrix<double>(2, 2, 0.0), 1e-12) && !Matrix<double>::is_zero(A, 1e-12));
    std::vector<double> val;
    std::vector<int> rp, ci;
    EXPECT(Matrix<double>(2, 2, {4, 0, -1, 4}).to_csr(1, &val, &rp, &ci) == 3 && rp[2] == 4 && ci[1] == 1);
    bool threw = false;
    try { A.get(2, 0); } catch (const std::out_of_range &) { threw = true; }
    EXPECT(threw);
    if (failures == 0) std::printf("SELFTEST_OK\n");
    return failures ? 1 : 0;
}
