// example.cpp -- the command-line driver, same switches and output lines as the reference's
// example.cpp:168-378, on top of include/pbicgstab.h.
//
//   ./example -M<matrix.mtx> -V<vector.mtx> [-D] -R<prob of zero> -N<dim> [-P] [device=<num>]
// additions: -T<tol> -I<maxit> -C<0|1|2> (0 = no preconditioner, 1 = the (A0 + I d) form with the
// diagonal split off, 2 = ILU(0), the reference's only choice and the default) -S<seed>.
// Differences from the reference: the exit status is 0 on success (the reference always returns
// EXIT_FAILURE, example.cpp:169,377); "method failed" is also printed when the preconditioned loop
// runs out of iterations (the reference cannot tell, pbicgstab.cu:408).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "mmio_wrapper.h"
#include "pbicgstab.h"

int main(int argc, char *argv[])
{
    char *matrix_filename = nullptr;
    char *vector_filename = nullptr;
    bool debug = false, print = false;
    double prob_of_zero_mat = 0.99;
    const double prob_of_zero_vec = 0.2;
    int dim = 10000;
    int maxit = 2000;
    double tol = 1e-6;
    int method = 2;
    long seed = -1;

    printf("WARNING: it is assumed that the matrices are stored in Matrix Market format with double as element type\n"
           " Usage: ./BiCGStab -M[matrix.mtx] -V[vector.mtx] [-D] -R[prob of zero] -N[dim] [-P] [device=<num>]\n"
           "By default matrix will be random, N = 10000, P(X = 0)=0.99, vector will be random, P(X = 0)=0.1\n"
           "example usage:\n"
           "./example.exe -M\"mat10000.mtx\"\n"
           "./example.exe -M\"mat3.mtx\" -V\"vec3.mtx\" -D -P\n"
           "./example.exe -N\"40\" -R\"0.5\" -D\n");

    for (int i = 0; i < argc; ++i) {
        const char *a = argv[i];
        if (a[0] != '-') continue;          // argv[0], device=<n>
        switch (a[1]) {
        case 'M': matrix_filename = argv[i] + 2; break;
        case 'V': vector_filename = argv[i] + 2; break;
        case 'D': debug = true; break;
        case 'R': prob_of_zero_mat = std::stod(a + 2); break;
        case 'P': print = true; break;
        case 'N': dim = std::stoi(a + 2); break;
        case 'T': tol = std::stod(a + 2); break;
        case 'I': maxit = std::stoi(a + 2); break;
        case 'C': method = std::stoi(a + 2); break;
        case 'S': seed = std::stol(a + 2); break;
        default:
            fprintf(stderr, "Unknown switch '-%s'\n", a + 1);
            return EXIT_FAILURE;
        }
    }
    if (seed >= 0) srand((unsigned)seed);
    if (matrix_filename) printf("Using matrix input file [%s]\n", matrix_filename);
    if (vector_filename) printf("Using vector input file [%s]\n", vector_filename);

    int ndev = 0;
    if (cudamat_device_count(&ndev) != CUDAMAT_OK || ndev < 1) {
        fprintf(stderr, "!!!! no HIP device: %s\n", cudamat_last_error());
        return EXIT_FAILURE;
    }

    int n = 0, nnz = 0;
    double *A = nullptr, *b = nullptr, *x = nullptr;
    int *iA = nullptr, *jA = nullptr;

    if (matrix_filename) {
        int matrixN, matrixM;
        if (loadMMSparseMatrix(matrix_filename, 'd', true, &matrixM, &matrixN, &nnz, &A, &iA, &jA)) {
            fprintf(stderr, "!!!! cusparseLoadMMSparseMatrix FAILED\n");
            return EXIT_FAILURE;
        }
        if (matrixN != matrixM) {
            fprintf(stderr, "!!!! square matrix is expected\n");
            return EXIT_FAILURE;
        }
        n = matrixN;
    } else {
        std::vector<double> vA;
        std::vector<int> vIA, vJA;
        nnz = fill_csr_matrix<Base::Base1>(dim, dim, &vA, &vIA, &vJA, [&](int i, int j) {
            if (i == j) return rand_float(1, 10);            // A[i,i] is never zero
            return rand_float_0_1() >= prob_of_zero_mat ? rand_float(1, 10) : 0.0;
        }, 1e-3);
        n = dim;
        if (vA.empty()) {
            fprintf(stderr, "!!!! all random elements of the random matrix are zeros !\n");
            return EXIT_FAILURE;
        }
        A = static_cast<double *>(malloc(sizeof(double) * nnz));
        iA = static_cast<int *>(malloc(sizeof(int) * (n + 1)));
        jA = static_cast<int *>(malloc(sizeof(int) * nnz));
        memcpy(A, vA.data(), sizeof(double) * nnz);
        memcpy(iA, vIA.data(), sizeof(int) * (n + 1));
        memcpy(jA, vJA.data(), sizeof(int) * nnz);
    }

    b = static_cast<double *>(malloc(sizeof(double) * n));
    if (vector_filename) {
        int vN, vM, vnnz;
        double *vA = nullptr;
        int *vIA = nullptr, *vJA = nullptr;
        if (loadMMSparseMatrix(vector_filename, 'd', true, &vM, &vN, &vnnz, &vA, &vIA, &vJA)) {
            fprintf(stderr, "!!!! cusparseLoadMMSparseMatrix FAILED\n");
            return EXIT_FAILURE;
        }
        if (vN != 1) { fprintf(stderr, "b must be a vector !\n"); return EXIT_FAILURE; }
        if (vM != n) { fprintf(stderr, "incorrect dim\n"); return EXIT_FAILURE; }
        toDenseVector(vM, vnnz, vA, vIA, b);
        free(vA); free(vIA); free(vJA);
    } else {
        gen_rand_vector(n, b, prob_of_zero_vec, 1, 5.0);
    }
    x = static_cast<double *>(malloc(sizeof(double) * n));

    std::cout << "nnz=" << nnz << std::endl;

    double dtAlg = 0.0;
    const double t1 = second();
    bool solved = false;
    if (method == 2) {
        solved = bicgstab_lu_precond(n, nnz, A, iA, jA, b, maxit, tol, debug, x, &dtAlg);
        solved = solved && cudamat_last_stats()->converged;
    } else if (method == 0) {
        solved = bicgstab(n, nnz, A, iA, jA, b, maxit, tol, debug, x, &dtAlg);
    } else {
        // split A = A0 + I*d and start from x0 = 1: the path example.cpp:33-106 exercises
        const int base = iA[0];
        std::vector<double> A0, d(n, 0.0), x0(n, 1.0);
        std::vector<int> iA0(1, base), jA0;
        for (int i = 0; i < n; ++i) {
            for (int k = iA[i] - base; k < iA[i + 1] - base; ++k) {
                if (jA[k] - base == i) d[i] = A[k];
                else { A0.push_back(A[k]); jA0.push_back(jA[k]); }
            }
            iA0.push_back(base + (int)A0.size());
        }
        if (A0.empty()) { A0.push_back(0.0); jA0.push_back(base); }
        solved = bicgstab(n, iA0[n] - base, A0.data(), iA0.data(), jA0.data(), d.data(), x0.data(), b, maxit, tol, debug,
                          x, &dtAlg);
    }
    const double t2 = second();

    if (solved) {
        std::cout << "success" << std::endl;
        if (print) {
            std::cout << "result:" << std::endl;
            std::ostringstream s;
            dump_vector(s, n, x);
            std::cout << s.str() << std::endl;
        }
        std::cout << "algorithm delta time = " << dtAlg << " s" << std::endl;
        std::cout << "total delta time = " << t2 - t1 << " s" << std::endl;
        const cudamat_stats *st = cudamat_last_stats();
        std::cout << "iterations = " << st->iters << (st->half_exit ? " (+half step)" : "")
                  << ", ||r||/||r0|| = " << (st->nrm0 > 0 ? st->nrm / st->nrm0 : 0.0) << std::endl;
    } else {
        std::cerr << "method failed" << std::endl;
    }
    free(x); free(b); free(A); free(iA); free(jA);
    return solved ? EXIT_SUCCESS : EXIT_FAILURE;
}
