// example.cpp -- command-line driver on top of include/pbicgstab.h.
//
// Accepts the switches of the reference's driver (example.cpp:193-223 there) and prints the same lines
// a user of it greps for ("nnz=", "success", "result:", "algorithm delta time = ", "total delta time = ",
// "method failed"):
//     -M<matrix.mtx>  -V<vector.mtx>  -D (trace)  -P (print x)  -N<dim> -R<P(zero)> (random system)
//     device=<num> is accepted and ignored (one GPU per process)
// and adds  -G<ngpu> (row-shard the solve over that many GPUs of the node: RCCL all-gather of the SpMV input and
// all-reduce of the dot products, one host thread per device; with -C2 every GPU factors its diagonal block)
// and  -T<tol>  -I<maxit>  -S<seed>  -C<0|1|2>:
//     2 = ILU(0)-preconditioned (the reference's only mode; default)
//     0 = no preconditioner
//     1 = the (A0 + I*d) entry point, with A's diagonal split off and x0 = 1
// Unlike the reference, the exit status reflects the outcome (upstream always returns EXIT_FAILURE) and a
// preconditioned run that exhausts maxit reports "method failed" (upstream cannot tell).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "mmio_wrapper.h"
#include "pbicgstab.h"

namespace {

struct Options {
    char *matrix_file = nullptr;
    char *vector_file = nullptr;
    bool trace = false;
    bool print_result = false;
    double p_zero_matrix = 0.99;
    double p_zero_vector = 0.2;
    int dim = 10000;
    int maxit = 2000;
    double tol = 1e-6;
    int method = 2;
    long seed = -1;
    int gpus = 1;
};

struct System {
    int n = 0, nnz = 0;
    double *A = nullptr;
    int *iA = nullptr, *jA = nullptr;
    double *b = nullptr;
    ~System() { free(A); free(iA); free(jA); free(b); }
};

void usage()
{
    printf("WARNING: it is assumed that the matrices are stored in Matrix Market format with double as element type\n"
           " Usage: ./BiCGStab -M[matrix.mtx] -V[vector.mtx] [-D] -R[prob of zero] -N[dim] [-P] [device=<num>]\n"
           "By default matrix will be random, N = 10000, P(X = 0)=0.99, vector will be random, P(X = 0)=0.1\n"
           "example usage:\n"
           "./example.exe -M\"mat10000.mtx\"\n"
           "./example.exe -M\"mat3.mtx\" -V\"vec3.mtx\" -D -P\n"
           "./example.exe -N\"40\" -R\"0.5\" -D\n");
}

bool parse(int argc, char **argv, Options &o)
{
    for (int i = 1; i < argc; ++i) {
        char *arg = argv[i];
        if (arg[0] != '-') continue;                 // e.g. device=0
        char *value = arg + 2;
        switch (arg[1]) {
        case 'M': o.matrix_file = value; break;
        case 'V': o.vector_file = value; break;
        case 'D': o.trace = true; break;
        case 'P': o.print_result = true; break;
        case 'R': o.p_zero_matrix = std::stod(value); break;
        case 'N': o.dim = std::stoi(value); break;
        case 'T': o.tol = std::stod(value); break;
        case 'I': o.maxit = std::stoi(value); break;
        case 'C': o.method = std::stoi(value); break;
        case 'S': o.seed = std::stol(value); break;
        case 'G': o.gpus = std::stoi(value); break;
        default:
            fprintf(stderr, "Unknown switch '-%s'\n", arg + 1);
            return false;
        }
    }
    return true;
}

template <typename T>
T *to_malloc(const std::vector<T> &v)
{
    T *p = static_cast<T *>(malloc(sizeof(T) * (v.empty() ? 1 : v.size())));
    if (!v.empty()) memcpy(p, v.data(), sizeof(T) * v.size());
    return p;
}

// matrix from a file, or the reference's random recipe: diagonal in [1,10], an off-diagonal entry in
// [1,10] with probability 1 - P(zero), entries below 1e-3 dropped
bool load_matrix(const Options &o, System &s)
{
    if (o.matrix_file) {
        int rows = 0, cols = 0;
        if (loadMMSparseMatrix(o.matrix_file, 'd', true, &rows, &cols, &s.nnz, &s.A, &s.iA, &s.jA)) {
            fprintf(stderr, "!!!! cusparseLoadMMSparseMatrix FAILED\n");
            return false;
        }
        if (rows != cols) {
            fprintf(stderr, "!!!! square matrix is expected\n");
            return false;
        }
        s.n = rows;
        return true;
    }
    std::vector<double> vals;
    std::vector<int> rowp, coli;
    const double pz = o.p_zero_matrix;
    s.nnz = fill_csr_matrix<Base1>(o.dim, o.dim, &vals, &rowp, &coli, [pz](int i, int j) -> double {
        if (i == j) return rand_float(1, 10);
        return rand_float_0_1() >= pz ? rand_float(1, 10) : 0.0;
    }, 1e-3);
    if (vals.empty()) {
        fprintf(stderr, "!!!! all random elements of the random matrix are zeros !\n");
        return false;
    }
    s.n = o.dim;
    s.A = to_malloc(vals);
    s.iA = to_malloc(rowp);
    s.jA = to_malloc(coli);
    return true;
}

bool load_rhs(const Options &o, System &s)
{
    s.b = static_cast<double *>(malloc(sizeof(double) * s.n));
    if (!o.vector_file) {
        gen_rand_vector(s.n, s.b, o.p_zero_vector, 1, 5.0);
        return true;
    }
    int rows = 0, cols = 0, entries = 0;
    double *v = nullptr;
    int *vi = nullptr, *vj = nullptr;
    if (loadMMSparseMatrix(o.vector_file, 'd', true, &rows, &cols, &entries, &v, &vi, &vj)) {
        fprintf(stderr, "!!!! cusparseLoadMMSparseMatrix FAILED\n");
        return false;
    }
    bool ok = true;
    if (cols != 1) { fprintf(stderr, "b must be a vector !\n"); ok = false; }
    else if (rows != s.n) { fprintf(stderr, "incorrect dim\n"); ok = false; }
    else toDenseVector(rows, entries, v, vi, s.b);
    free(v); free(vi); free(vj);
    return ok;
}

// A = A0 + I*d with the stored diagonal moved into d
bool solve_split(const System &s, const Options &o, double *x, double *dt)
{
    const int base = s.iA[0];
    std::vector<double> a0, d(s.n, 0.0), ones(s.n, 1.0);
    std::vector<int> rp(1, base), ci;
    for (int i = 0; i < s.n; ++i) {
        for (int k = s.iA[i] - base; k < s.iA[i + 1] - base; ++k) {
            if (s.jA[k] - base == i) { d[i] = s.A[k]; continue; }
            a0.push_back(s.A[k]);
            ci.push_back(s.jA[k]);
        }
        rp.push_back(base + static_cast<int>(a0.size()));
    }
    const int nnz0 = static_cast<int>(a0.size());
    if (a0.empty()) { a0.push_back(0.0); ci.push_back(base); }   // keep the pointers valid
    return bicgstab(s.n, nnz0, a0.data(), rp.data(), ci.data(), d.data(), ones.data(), s.b, o.maxit, o.tol, o.trace, x, dt);
}

}  // namespace

int main(int argc, char *argv[])
{
    Options opt;
    usage();
    if (!parse(argc, argv, opt)) return EXIT_FAILURE;
    if (opt.seed >= 0) srand(static_cast<unsigned>(opt.seed));
    if (opt.matrix_file) printf("Using matrix input file [%s]\n", opt.matrix_file);
    if (opt.vector_file) printf("Using vector input file [%s]\n", opt.vector_file);

    int devices = 0;
    if (cudamat_device_count(&devices) != CUDAMAT_OK || devices < 1) {
        fprintf(stderr, "!!!! no HIP device: %s\n", cudamat_last_error());
        return EXIT_FAILURE;
    }

    if (opt.gpus > 1) {
        if (opt.gpus > devices && !getenv("CUDAMAT_SHARDED_ONE_DEVICE")) {
            fprintf(stderr, "!!!! -G%d: only %d HIP device(s) visible\n", opt.gpus, devices);
            return EXIT_FAILURE;
        }
        cudamat_use_gpus(opt.gpus);
        printf("Using %d GPUs (row blocks of %d rows)\n", opt.gpus, (opt.matrix_file ? 0 : opt.dim + opt.gpus - 1) / opt.gpus);
    }

    System sys;
    if (!load_matrix(opt, sys) || !load_rhs(opt, sys)) return EXIT_FAILURE;
    std::vector<double> x(sys.n, 0.0);
    std::cout << "nnz=" << sys.nnz << std::endl;

    double dtAlg = 0.0;
    const double t_begin = second();
    bool solved;
    if (opt.method == 0) {
        solved = bicgstab(sys.n, sys.nnz, sys.A, sys.iA, sys.jA, sys.b, opt.maxit, opt.tol, opt.trace, x.data(), &dtAlg);
    } else if (opt.method == 1) {
        solved = solve_split(sys, opt, x.data(), &dtAlg);
    } else {
        solved = bicgstab_lu_precond(sys.n, sys.nnz, sys.A, sys.iA, sys.jA, sys.b, opt.maxit, opt.tol, opt.trace, x.data(),
                                     &dtAlg) && cudamat_last_stats()->converged;
    }
    const double t_end = second();

    if (!solved) {
        std::cerr << "method failed" << std::endl;
        return EXIT_FAILURE;
    }
    std::cout << "success" << std::endl;
    if (opt.print_result) {
        std::ostringstream text;
        dump_vector(text, sys.n, x.data());
        std::cout << "result:" << std::endl << text.str() << std::endl;
    }
    std::cout << "algorithm delta time = " << dtAlg << " s" << std::endl;
    std::cout << "total delta time = " << t_end - t_begin << " s" << std::endl;
    const cudamat_stats *st = cudamat_last_stats();
    std::cout << "iterations = " << st->iters << (st->half_exit ? " (+half step)" : "")
              << ", ||r||/||r0|| = " << (st->nrm0 > 0 ? st->nrm / st->nrm0 : 0.0) << std::endl;
    return EXIT_SUCCESS;
}
