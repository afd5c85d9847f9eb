// Host-side counterparts of the reference's free functions in
// include/matrix.hpp.  The operator apply runs on the device.
#ifndef KRYLOV_HOST_MATRIX_HPP_
#define KRYLOV_HOST_MATRIX_HPP_

#include <cassert>
#include <cmath>

#include "constants.hpp"
#include "device.hpp"

// include/matrix.hpp:8-21 (small host matrices only: the reduced systems)
inline matrix &dot(matrix &result, const matrix &lhs, const matrix &rhs) {
  assert(lhs.shape()[1] == rhs.shape()[0]);
  for (unsigned long i = 0; i < lhs.shape()[0]; i++)
    for (unsigned long j = 0; j < rhs.shape()[1]; j++) {
      real acc = 0.0;
      for (unsigned long k = 0; k < lhs.shape()[1]; k++) acc += lhs(i, k) * rhs(k, j);
      result(i, j) = acc;
    }
  return result;
}

// include/matrix.hpp:23-33: y = A x, evaluated on the device (kry_op_apply).
inline vector &dot(vector &result, const matrix &lhs, const vector &rhs) {
  assert(lhs.shape()[0] == lhs.shape()[1] && lhs.shape()[1] == rhs.shape()[0]);
  kry_op *op = nullptr;
  krylov::check(kry_op_create_dense(krylov::default_context(), (int64_t)lhs.shape()[0], lhs.data(), &op),
                "kry_op_create_dense");
  const int st = kry_op_apply(op, rhs.data(), result.data());
  kry_op_destroy(op);
  krylov::check(st, "kry_op_apply");
  return result;
}

// include/matrix.hpp:35-44
inline real dot(real &result, const vector &lhs, const vector &rhs) noexcept {
  result = 0.0;
  for (unsigned long i = 0; i < lhs.shape()[0]; i++) result += lhs(i) * rhs(i);
  return result;
}

// include/matrix.hpp:46-60
inline real l2_norm(const vector &v) noexcept {
  real l2 = 0.0;
  for (real c : v) l2 += c * c;
  return std::sqrt(l2);
}

inline real normalize(vector &v) noexcept {
  const real l2 = l2_norm(v);
  for (real &r : v) r /= l2;
  return l2;
}

#endif  // KRYLOV_HOST_MATRIX_HPP_
