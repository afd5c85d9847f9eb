// Host-side counterparts of the reference's free functions in
// include/matrix.hpp.  The operator apply runs on the device.
#ifndef KRYLOV_HOST_MATRIX_HPP_
#define KRYLOV_HOST_MATRIX_HPP_

#include <algorithm>
#include <cassert>
#include <cmath>
#include <numeric>

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

// The three vector helpers of include/matrix.hpp:35-60, stated with the standard
// algorithms.  Their semantics are what the reference's loops define: a strictly
// sequential left-to-right sum starting from 0.0 (std::inner_product evaluates
// acc = acc + a * b in order), and normalisation by true division.
inline real dot(real &result, const vector &lhs, const vector &rhs) noexcept {
  // the reference bounds its loop by the left operand's length
  return result = std::inner_product(lhs.begin(), lhs.end(), rhs.begin(), real(0.0));
}

inline real l2_norm(const vector &v) noexcept { return std::sqrt(std::inner_product(v.begin(), v.end(), v.begin(), real(0.0))); }

inline real normalize(vector &v) noexcept {
  const real length = l2_norm(v);
  std::transform(v.begin(), v.end(), v.begin(), [length](real entry) { return entry / length; });
  return length;
}

#endif  // KRYLOV_HOST_MATRIX_HPP_
