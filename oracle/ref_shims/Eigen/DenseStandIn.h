// Test infrastructure only (oracle/).  Minimal stand-in for the part of Eigen that the reference's SparseBlock
// (/root/reference/src/droid_kernels.cu:1126-1228) uses.  Eigen itself is absent here (it is vendored inside the
// un-vendored lietorch submodule), so the reference's sparse fp64 LLT is replaced by a DENSE fp64 LLT with the same
// contract: duplicates in setFromTriplets are summed, the factorisation reads the LOWER triangle (Eigen's default
// UpLo for SimplicialLLT), info() != Success when a pivot is not positive.  AMD ordering only changes the order of
// the fp64 operations, not the mathematical result.
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>

namespace Eigen {

enum ComputationInfo { Success = 0, NumericalIssue = 1 };

template <typename S> struct Triplet {
  int r, c; S v;
  Triplet() : r(0), c(0), v(0) {}
  Triplet(int r_, int c_, S v_) : r(r_), c(c_), v(v_) {}
  int row() const { return r; }
  int col() const { return c; }
  S value() const { return v; }
};

template <typename S> class VectorX {
 public:
  std::vector<S> d;
  VectorX() {}
  explicit VectorX(std::size_t n) : d(n, S(0)) {}
  static VectorX Zero(std::size_t n) { return VectorX(n); }
  S& operator()(std::size_t i) { return d[i]; }
  const S& operator()(std::size_t i) const { return d[i]; }
  S* data() { return d.data(); }
  const S* data() const { return d.data(); }
  std::size_t size() const { return d.size(); }
  VectorX operator-(const VectorX& o) const { VectorX r(d.size()); for (std::size_t i = 0; i < d.size(); ++i) r.d[i] = d[i] - o.d[i]; return r; }
};
typedef VectorX<double> VectorXd;

// expression pieces for   L.diagonal().array() += ep + lm * L.diagonal().array();
struct ArrayValues { std::vector<double> v; };
inline ArrayValues operator*(double s, const ArrayValues& a) { ArrayValues r(a); for (auto& x : r.v) x *= s; return r; }
inline ArrayValues operator+(double s, const ArrayValues& a) { ArrayValues r(a); for (auto& x : r.v) x += s; return r; }

template <typename S> class SparseMatrix;

template <typename S> struct DiagArrayRef {
  SparseMatrix<S>* m;
  operator ArrayValues() const;
  DiagArrayRef& operator+=(const ArrayValues& a);
};
template <typename S> inline ArrayValues operator*(double s, const DiagArrayRef<S>& a) { return s * ArrayValues(a); }
template <typename S> struct DiagRef { SparseMatrix<S>* m; DiagArrayRef<S> array() { return DiagArrayRef<S>{m}; } };

template <typename S> class SparseMatrix {        // dense column-major storage behind the sparse interface
 public:
  int nr, nc; std::vector<S> d;
  SparseMatrix() : nr(0), nc(0) {}
  SparseMatrix(int r, int c) : nr(r), nc(c), d(std::size_t(r) * c, S(0)) {}
  int rows() const { return nr; }
  int cols() const { return nc; }
  S& at(int r, int c) { return d[std::size_t(c) * nr + r]; }
  const S& at(int r, int c) const { return d[std::size_t(c) * nr + r]; }
  template <typename It> void setFromTriplets(It b, It e) {
    std::fill(d.begin(), d.end(), S(0));
    for (; b != e; ++b) at(b->row(), b->col()) += b->value();       // duplicates are summed, as Eigen does
  }
  SparseMatrix operator-(const SparseMatrix& o) const { SparseMatrix r(nr, nc); for (std::size_t i = 0; i < d.size(); ++i) r.d[i] = d[i] - o.d[i]; return r; }
  DiagRef<S> diagonal() { return DiagRef<S>{this}; }
};
template <typename S> DiagArrayRef<S>::operator ArrayValues() const { ArrayValues r; r.v.resize(m->nr); for (int i = 0; i < m->nr; ++i) r.v[i] = m->at(i, i); return r; }
template <typename S> DiagArrayRef<S>& DiagArrayRef<S>::operator+=(const ArrayValues& a) { for (int i = 0; i < m->nr; ++i) m->at(i, i) += a.v[i]; return *this; }

class MatrixXd {                                  // only  MatrixXd(A).data()  is used (get_dense)
 public:
  int nr, nc; std::vector<double> d;
  MatrixXd() : nr(0), nc(0) {}
  explicit MatrixXd(const SparseMatrix<double>& a) : nr(a.nr), nc(a.nc), d(a.d) {}
  double* data() { return d.data(); }
};

template <typename M> class SimplicialLLT {       // dense lower Cholesky, column by column
 public:
  int n; std::vector<double> L; ComputationInfo st;
  SimplicialLLT() : n(0), st(NumericalIssue) {}
  void compute(const M& a) {
    n = a.rows(); L = a.d; st = Success;
    for (int j = 0; j < n; ++j) {
      double* cj = &L[std::size_t(j) * n];
      for (int k = 0; k < j; ++k) {
        const double* ck = &L[std::size_t(k) * n];
        const double f = ck[j];
        if (f != 0.0) for (int i = j; i < n; ++i) cj[i] -= f * ck[i];
      }
      const double p = cj[j];
      if (!(p > 0.0)) { st = NumericalIssue; return; }
      const double s = std::sqrt(p);
      cj[j] = s;
      for (int i = j + 1; i < n; ++i) cj[i] /= s;
    }
  }
  ComputationInfo info() const { return st; }
  VectorXd solve(const VectorXd& b) const {
    VectorXd x(b);
    for (int j = 0; j < n; ++j) {                 // L y = b
      x.d[j] /= L[std::size_t(j) * n + j];
      const double xj = x.d[j];
      for (int i = j + 1; i < n; ++i) x.d[i] -= L[std::size_t(j) * n + i] * xj;
    }
    for (int j = n - 1; j >= 0; --j) {            // L^T x = y
      double s = x.d[j];
      for (int i = j + 1; i < n; ++i) s -= L[std::size_t(j) * n + i] * x.d[i];
      x.d[j] = s / L[std::size_t(j) * n + j];
    }
    return x;
  }
};

}  // namespace Eigen
