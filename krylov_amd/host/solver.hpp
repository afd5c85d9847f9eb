// Solver classes with the interface of the reference's include/solver.hpp.
// GMRESSolver<m> / GMRESTDPSolver<m> run on the device through the C ABI;
// LUSolver / TriDiagSolver / SparseTriDiagSolver are the small host-side direct
// solvers the Python module also exposes.
#ifndef KRYLOV_HOST_SOLVER_HPP_
#define KRYLOV_HOST_SOLVER_HPP_

#include <cassert>
#include <memory>
#include <utility>

#include "constants.hpp"
#include "device.hpp"
#include "matrix.hpp"

class Solver {
public:
  explicit Solver(unsigned long sys_size) : vars(vector::shape_type({sys_size})) {
    for (real &v : vars) v = s_nan;  // include/solver.hpp:16-18
  }
  const vector &prev_result() const noexcept { return vars; }

protected:
  vector vars;
};

// Doolittle LU without pivoting (include/solver.hpp:28-113), host only.
class LUSolver : public Solver {
public:
  explicit LUSolver(unsigned long n)
      : Solver(n), l(matrix::shape_type{n, n}), u(matrix::shape_type{n, n}), partial(vector::shape_type{n}) {}

  std::pair<const matrix &, const matrix &> lu_decomp(const matrix &a) {
    const unsigned long n = a.shape()[0];
    for (real &v : l) v = 0.0;
    for (real &v : u) v = 0.0;
    for (unsigned long k = 0; k < n; k++) {
      l(k, k) = 1.0;
      for (unsigned long j = k; j < n; j++) {  // row k of u, diagonal first
        real acc = a(k, j);
        for (unsigned long i = 0; i < k; i++) acc -= l(k, i) * u(i, j);
        u(k, j) = acc / l(k, k);
      }
      for (unsigned long j = k + 1; j < n; j++) {  // column k of l
        real acc = a(j, k);
        for (unsigned long i = 0; i < k; i++) acc -= l(j, i) * u(i, k);
        l(j, k) = acc / u(k, k);
      }
    }
    return {l, u};
  }

  const vector &solve(const matrix &a, const vector &b) {
    const unsigned long n = a.shape()[0];
    lu_decomp(a);
    for (unsigned long i = 0; i < n; i++) {
      real acc = b(i);
      for (unsigned long j = 0; j < i; j++) acc -= l(i, j) * partial(j);
      partial(i) = acc / l(i, i);
    }
    for (unsigned long i = n; i-- > 0;) {
      real acc = partial(i);
      for (unsigned long j = i + 1; j < n; j++) acc -= u(i, j) * vars(j);
      vars(i) = acc / u(i, i);
    }
    return vars;
  }

  unsigned long size() { return l.shape()[0]; }

protected:
  matrix l, u;
  vector partial;
};

// Thomas algorithm on an n x 3 band array (include/solver.hpp:241-277), host only.
class SparseTriDiagSolver : public Solver {
public:
  explicit SparseTriDiagSolver(unsigned long n) : Solver(n), u_diag(vector::shape_type{n > 0 ? n : 1}) {}

  const vector &solve(const matrix &d, const vector &b) {
    const unsigned long n = vars.shape()[0];
    vars(0) = b(0) / d(0, 1);
    u_diag(0) = d(0, 2) / d(0, 1);
    for (unsigned long i = 1; i < n; i++) {
      const real diag = d(i, 1) - u_diag(i - 1) * d(i, 0);
      vars(i) = (b(i) - vars(i - 1) * d(i, 0)) / diag;
      if (i + 1 < n) u_diag(i) = d(i, 2) / diag;
    }
    for (unsigned long i = n - 1; i-- > 0;) vars(i) -= vars(i + 1) * u_diag(i);
    return vars;
  }

protected:
  vector u_diag;
};

// Thomas algorithm on a dense tridiagonal matrix (include/solver.hpp:335-368), host only.
class TriDiagSolver : public Solver {
public:
  explicit TriDiagSolver(unsigned long n) : Solver(n), u_diag(vector::shape_type{n > 0 ? n : 1}) {}

  const vector &solve(const matrix &a, const vector &b) {
    const unsigned long n = vars.shape()[0];
    vars(0) = b(0) / a(0, 0);
    if (n > 1) u_diag(0) = a(0, 1) / a(0, 0);
    for (unsigned long i = 1; i < n; i++) {
      const real diag = a(i, i) - u_diag(i - 1) * a(i, i - 1);
      vars(i) = (b(i) - vars(i - 1) * a(i, i - 1)) / diag;
      if (i + 1 < n) u_diag(i) = a(i, i + 1) / diag;
    }
    for (unsigned long i = n - 1; i-- > 0;) vars(i) -= vars(i + 1) * u_diag(i);
    return vars;
  }

protected:
  vector u_diag;
};

namespace krylov {

// Owns a device solver handle; shared by the plain and the preconditioned class.
class DeviceGmres {
public:
  DeviceGmres(unsigned long n, int m, int flags) : n_(n), m_(m) {
    check(kry_gmres_create(default_context(), (int64_t)n, m, flags, &g_), "kry_gmres_create");
  }
  ~DeviceGmres() {
    kry_gmres_destroy(g_);
    if (op_) kry_op_destroy(op_);
  }
  DeviceGmres(const DeviceGmres &) = delete;
  DeviceGmres &operator=(const DeviceGmres &) = delete;

  // the dense `matrix` argument becomes a device operator; it is rebuilt on
  // every call because the caller may have changed the matrix in between
  kry_op *upload(const matrix &a) {
    if (op_) kry_op_destroy(op_);
    op_ = nullptr;
    check(kry_op_create_dense(default_context(), (int64_t)a.shape()[0], a.data(), &op_), "kry_op_create_dense");
    return op_;
  }
  kry_gmres *handle() noexcept { return g_; }
  unsigned long n_;
  int m_;

private:
  kry_gmres *g_ = nullptr;
  kry_op *op_ = nullptr;
};

}  // namespace krylov

// include/solver.hpp:115-239, evaluated on the device.
template <unsigned int num_iters> class GMRESSolver : public Solver {
public:
  static constexpr unsigned int iterations = num_iters;
  static constexpr int device_flags = KRY_GMRES_PLAIN;

  explicit GMRESSolver(unsigned long sys_size) : GMRESSolver(sys_size, KRY_GMRES_PLAIN) {}
  virtual ~GMRESSolver() = default;

  const vector &solve(const matrix &system, const vector &sol) {
    krylov::check(kry_gmres_solve(dev_->handle(), dev_->upload(system), sol.data(), vars.data()), "kry_gmres_solve");
    return vars;
  }

  real initial_subspace(const vector &sol) {
    real beta = 0.0;
    krylov::check(kry_gmres_initial_subspace(dev_->handle(), sol.data(), &beta), "kry_gmres_initial_subspace");
    return beta;
  }

  virtual real add_arnoldi(const matrix &system) {
    real h = 0.0;
    krylov::check(kry_gmres_add_arnoldi(dev_->handle(), dev_->upload(system), &h), "kry_gmres_add_arnoldi");
    return h;
  }

  const vector &subspace_vector(int i) {
    scratch_ = vector(vector::shape_type{dev_->n_});
    krylov::check(kry_gmres_subspace_vector(dev_->handle(), i, scratch_.data()), "kry_gmres_subspace_vector");
    return scratch_;
  }

  real hessenberg(int i, int j) const {
    real h = 0.0;
    krylov::check(kry_gmres_hessenberg(dev_->handle(), i, j, &h), "kry_gmres_hessenberg");
    return h;
  }

protected:
  GMRESSolver(unsigned long sys_size, int flags)
      : Solver(sys_size), dev_(std::make_shared<krylov::DeviceGmres>(sys_size, (int)num_iters, flags)) {}
  std::shared_ptr<krylov::DeviceGmres> dev_;
  vector scratch_;
};

// include/solver.hpp:279-333, evaluated on the device.  As in the reference the
// returned vector is the combination of the basis vectors (M^-1 is not applied).
template <unsigned int num_iters> class GMRESTDPSolver : public GMRESSolver<num_iters> {
public:
  using Parent = GMRESSolver<num_iters>;
  static constexpr int device_flags = KRY_GMRES_TDP;

  explicit GMRESTDPSolver(unsigned long sys_size)
      : Parent(sys_size, KRY_GMRES_TDP), sparse_tdiags(matrix::shape_type{sys_size, 3}) {}
  virtual ~GMRESTDPSolver() = default;

  const matrix &assemble_preconditioner(const matrix &system) {
    krylov::check(kry_gmres_assemble_preconditioner(this->dev_->handle(), this->dev_->upload(system),
                                                    sparse_tdiags.data()),
                  "kry_gmres_assemble_preconditioner");
    return sparse_tdiags;
  }

protected:
  matrix sparse_tdiags;
};

#endif  // KRYLOV_HOST_SOLVER_HPP_
