// Basic types of the host mirror: the reference builds these on xtensor
// (include/constants.hpp:9-18); here they are small self-contained containers
// with the subset of the interface the solver / time-discretisation classes use.
#ifndef KRYLOV_HOST_CONSTANTS_HPP_
#define KRYLOV_HOST_CONSTANTS_HPP_

#include <array>
#include <cstddef>
#include <initializer_list>
#include <limits>
#include <tuple>
#include <type_traits>
#include <vector>

using real = double;
using triple = std::tuple<real, real, real>;

constexpr real pi = 3.1415926535897932384626433832795;
constexpr real q_nan = std::numeric_limits<real>::quiet_NaN();
constexpr real s_nan = std::numeric_limits<real>::signaling_NaN();

// Dense rank-R array of doubles, row-major, uninitialised on construction.
template <std::size_t R> class dense_array {
public:
  using shape_type = std::array<unsigned long, R>;

  dense_array() : shape_{} {}
  explicit dense_array(const shape_type &shape) : shape_(shape), data_(count(shape)) {}
  // vector{1.0, 2.0, ...}
  template <std::size_t Q = R, std::enable_if_t<Q == 1, int> = 0>
  dense_array(std::initializer_list<real> values) : data_(values) {
    shape_[0] = data_.size();
  }
  // matrix{{1.0, 2.0}, {3.0, 4.0}}
  template <std::size_t Q = R, std::enable_if_t<Q == 2, int> = 0>
  dense_array(std::initializer_list<std::initializer_list<real>> rows) {
    shape_[0] = rows.size();
    shape_[1] = rows.size() ? rows.begin()->size() : 0;
    for (const auto &r : rows) data_.insert(data_.end(), r.begin(), r.end());
  }

  const shape_type &shape() const noexcept { return shape_; }
  std::size_t size() const noexcept { return data_.size(); }
  real *data() noexcept { return data_.data(); }
  const real *data() const noexcept { return data_.data(); }
  auto begin() noexcept { return data_.begin(); }
  auto end() noexcept { return data_.end(); }
  auto begin() const noexcept { return data_.begin(); }
  auto end() const noexcept { return data_.end(); }

  real &operator()(unsigned long i) noexcept { return data_[i]; }
  real operator()(unsigned long i) const noexcept { return data_[i]; }
  real &operator()(unsigned long i, unsigned long j) noexcept { return data_[i * shape_[1] + j]; }
  real operator()(unsigned long i, unsigned long j) const noexcept { return data_[i * shape_[1] + j]; }

  dense_array &operator*=(real s) noexcept {
    for (real &v : data_) v *= s;
    return *this;
  }

private:
  static std::size_t count(const shape_type &s) {
    std::size_t c = 1;
    for (auto d : s) c *= d;
    return c;
  }
  shape_type shape_;
  std::vector<real> data_;
};

using matrix = dense_array<2>;
using vector = dense_array<1>;

#endif  // KRYLOV_HOST_CONSTANTS_HPP_
