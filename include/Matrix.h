/* Matrix.h -- small dense row-major matrix, kept so that code written against the reference's headers keeps
 * compiling (BASELINE.json names Matrix.h next to pbicgstab.h).  NOT on the solver path: the reference's
 * Matrix.h is a dense toy that its own build excludes (CMakeLists.txt:17; it needs -fconcepts).  This header is an
 * independent C++17 implementation of the same public names -- get/set/row/column, mul, transpose, identity,
 * is_zero, without_column (Matrix.h:57-100 there) -- with the INTENDED maths where upstream is off (its mul sums
 * k < a.n instead of a.m, column() walks i < m, transpose() indexes the result the wrong way round), plus
 * to_csr(), the bridge to the solvers of pbicgstab.h. */
#ifndef CUDAMAT_MATRIX_H
#define CUDAMAT_MATRIX_H

#include <cmath>
#include <initializer_list>
#include <stdexcept>
#include <vector>

template <class T>
class Matrix {
    std::vector<T> cells_;   // row-major, n x m

public:
    const int n, m;

    Matrix(int rows, int cols) : cells_(checked(rows, cols)), n(rows), m(cols) {}
    Matrix(int rows, int cols, const T &fill) : cells_(checked(rows, cols), fill), n(rows), m(cols) {}
    Matrix(int rows, int cols, std::initializer_list<T> row_major) : cells_(row_major), n(rows), m(cols)
    {
        if (cells_.size() != checked(rows, cols)) throw std::runtime_error("Matrix: initializer list has the wrong length");
    }
    Matrix(const Matrix &) = default;
    Matrix &operator=(const Matrix &other)
    {
        if (n != other.n || m != other.m) throw std::runtime_error("A = B called on matrices of different sizes");
        cells_ = other.cells_;
        return *this;
    }

    const T &get(int i, int j) const { return cells_[at(i, j)]; }
    Matrix &set(int i, int j, const T &e)
    {
        cells_[at(i, j)] = e;
        return *this;
    }

    Matrix row(int i) const
    {
        Matrix r(1, m);
        for (int j = 0; j < m; ++j) r.set(0, j, get(i, j));
        return r;
    }
    Matrix column(int j) const
    {
        Matrix c(n, 1);
        for (int i = 0; i < n; ++i) c.set(i, 0, get(i, j));
        return c;
    }

    static Matrix transpose(const Matrix &a)
    {
        Matrix t(a.m, a.n);
        for (int i = 0; i < a.n; ++i)
            for (int j = 0; j < a.m; ++j) t.set(j, i, a.get(i, j));
        return t;
    }
    static Matrix identity(int size)
    {
        Matrix e(size, size, T(0));
        for (int i = 0; i < size; ++i) e.set(i, i, T(1));
        return e;
    }
    static bool is_zero(const Matrix &a, const T eps)
    {
        for (const T &v : a.cells_)
            if (std::abs(v) >= eps) return false;
        return true;
    }
    static Matrix without_column(const Matrix &a, int drop)
    {
        if (a.m < 1 || drop < 0 || drop >= a.m) throw std::runtime_error("without_column: no such column");
        Matrix r(a.n, a.m - 1);
        for (int i = 0; i < a.n; ++i)
            for (int j = 0, k = 0; j < a.m; ++j)
                if (j != drop) r.set(i, k++, a.get(i, j));
        return r;
    }

    /* CSR arrays (index base `base`, entries with |a_ij| > drop_below kept) for bicgstab()/bicgstab_lu_precond() */
    int to_csr(int base, std::vector<T> *val, std::vector<int> *row_ptr, std::vector<int> *col_idx, T drop_below = T(0)) const
    {
        val->clear();
        col_idx->clear();
        row_ptr->assign(1, base);
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < m; ++j)
                if (std::abs(get(i, j)) > drop_below) {
                    val->push_back(get(i, j));
                    col_idx->push_back(j + base);
                }
            row_ptr->push_back(base + static_cast<int>(val->size()));
        }
        return static_cast<int>(val->size());
    }

private:
    static size_t checked(int rows, int cols)
    {
        if (rows < 0 || cols < 0) throw std::runtime_error("Matrix: negative dimension");
        return static_cast<size_t>(rows) * static_cast<size_t>(cols);
    }
    size_t at(int i, int j) const
    {
        if (i < 0 || i >= n || j < 0 || j >= m) throw std::out_of_range("Matrix: index out of range");
        return static_cast<size_t>(i) * static_cast<size_t>(m) + static_cast<size_t>(j);
    }
};

template <class T>
Matrix<T> mul(const Matrix<T> &a, const Matrix<T> &b)
{
    if (a.m != b.n) throw std::runtime_error("mul(A,B) called on incompatible matrices");
    Matrix<T> c(a.n, b.m);
    for (int i = 0; i < a.n; ++i)
        for (int j = 0; j < b.m; ++j) {
            T sum = T(0);
            for (int k = 0; k < a.m; ++k) sum += a.get(i, k) * b.get(k, j);
            c.set(i, j, sum);
        }
    return c;
}

#endif /* CUDAMAT_MATRIX_H */
