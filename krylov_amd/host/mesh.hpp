// Host-side Mesh with the interface of the reference's include/mesh.hpp.  It
// is the description handed to the device (geometry + initial fields) and
// the container results are copied back into; the time stepping itself never
// reads it.
#ifndef KRYLOV_HOST_MESH_HPP_
#define KRYLOV_HOST_MESH_HPP_

#include <algorithm>
#include <cassert>
#include <functional>
#include <utility>

#include "constants.hpp"

// Strided 1-D window into one field of a Mesh: a ghost or boundary strip.
// Dir only distinguishes the two view types of the reference (vertical strips
// run along j, horizontal ones along i).
template <int Dir> class FieldStrip {
public:
  FieldStrip(real *first, std::size_t stride, std::size_t count) : first_(first), stride_(stride), count_(count) {}
  std::array<std::size_t, 1> shape() const noexcept { return {count_}; }
  real &operator()(std::size_t k) const noexcept { return first_[k * stride_]; }

private:
  real *first_;
  std::size_t stride_, count_;
};

class Mesh {
public:
  static constexpr int ghost_cells = 1;
  using vert_view = FieldStrip<0>;   // fixed i, running j
  using horiz_view = FieldStrip<1>;  // fixed j, running i

  Mesh(const std::pair<real, real> &corner_1, const std::pair<real, real> &corner_2, const size_t x_cells,
       const size_t y_cells)
      : min_x_(std::min(corner_1.first, corner_2.first)), max_x_(std::max(corner_1.first, corner_2.first)),
        min_y_(std::min(corner_1.second, corner_2.second)), max_y_(std::max(corner_1.second, corner_2.second)),
        dx_((max_x_ - min_x_) / x_cells), dy_((max_y_ - min_y_) / y_cells), cx_((int)x_cells), cy_((int)y_cells),
        temp_(matrix::shape_type{x_cells + 2, y_cells + 2}), vel_u_(matrix::shape_type{x_cells + 2, y_cells + 2}),
        vel_v_(matrix::shape_type{x_cells + 2, y_cells + 2}) {
    // ghosts and corners: signalling NaN; interior: 0 (src/mesh.cpp:26-51)
    for (matrix *f : {&temp_, &vel_u_, &vel_v_}) {
      for (real &v : *f) v = s_nan;
      for (int i = 0; i < cx_; i++)
        for (int j = 0; j < cy_; j++) (*f)(i + 1, j + 1) = 0.0;
    }
  }

  Mesh(const std::pair<real, real> &corner_1, const std::pair<real, real> &corner_2, const size_t x_cells,
       const size_t y_cells, std::function<triple(real, real)> f)
      : Mesh(corner_1, corner_2, x_cells, y_cells) {
    for (int i = 0; i < cx_; i++)
      for (int j = 0; j < cy_; j++) {
        auto [t, u, v] = f(median_x(i), median_y(j));
        Temp(i, j) = t;
        vel_u(i, j) = u;
        vel_v(i, j) = v;
      }
  }

  // include/mesh.hpp:20-54, same operation order
  real left_x(int i) const noexcept { return min_x_ + i * dx_; }
  real right_x(int i) const noexcept { return min_x_ + (i + 1) * dx_; }
  real bottom_y(int j) const noexcept { return min_y_ + j * dy_; }
  real top_y(int j) const noexcept { return min_y_ + (j + 1) * dy_; }
  real median_x(int i) const noexcept { return min_x_ + dx_ / 2 + i * dx_; }
  real median_y(int j) const noexcept { return min_y_ + dy_ / 2 + j * dy_; }
  int x_idx(real x) const noexcept { return static_cast<int>(x / dx_ - min_x_ / dx_); }
  int y_idx(real y) const noexcept { return static_cast<int>(y / dy_ - min_y_ / dy_); }
  real dx() const noexcept { return dx_; }
  real dy() const noexcept { return dy_; }
  real min_x() const noexcept { return min_x_; }
  real max_x() const noexcept { return max_x_; }
  real min_y() const noexcept { return min_y_; }
  real max_y() const noexcept { return max_y_; }
  int cells_x() const noexcept { return cx_; }
  int cells_y() const noexcept { return cy_; }

  real Temp(int i, int j) const noexcept { return temp_(i + 1, j + 1); }
  real &Temp(int i, int j) noexcept { return temp_(i + 1, j + 1); }
  real vel_u(int i, int j) const noexcept { return vel_u_(i + 1, j + 1); }
  real &vel_u(int i, int j) noexcept { return vel_u_(i + 1, j + 1); }
  real vel_v(int i, int j) const noexcept { return vel_v_(i + 1, j + 1); }
  real &vel_v(int i, int j) noexcept { return vel_v_(i + 1, j + 1); }

  const real *data_Temp() const noexcept { return temp_.data(); }
  const real *data_vel_u() const noexcept { return vel_u_.data(); }
  const real *data_vel_v() const noexcept { return vel_v_.data(); }
  real *data(int field) noexcept { return field == 0 ? temp_.data() : field == 1 ? vel_u_.data() : vel_v_.data(); }
  const matrix &array_Temp() const noexcept { return temp_; }
  const matrix &array_vel_u() const noexcept { return vel_u_; }
  const matrix &array_vel_v() const noexcept { return vel_v_; }

  // Ghost / boundary strips (include/mesh.hpp:134-235); every strip covers the
  // cy (vertical) or cx (horizontal) cells next to the interior.
  enum Side { Bottom, Top, Left, Right };

  vert_view ghostcells_left_Temp() noexcept { return vstrip(0, Left, true); }
  vert_view ghostcells_right_Temp() noexcept { return vstrip(0, Right, true); }
  horiz_view ghostcells_top_Temp() noexcept { return hstrip(0, Top, true); }
  horiz_view ghostcells_bottom_Temp() noexcept { return hstrip(0, Bottom, true); }
  vert_view ghostcells_left_vel_u() noexcept { return vstrip(1, Left, true); }
  vert_view ghostcells_right_vel_u() noexcept { return vstrip(1, Right, true); }
  horiz_view ghostcells_top_vel_u() noexcept { return hstrip(1, Top, true); }
  horiz_view ghostcells_bottom_vel_u() noexcept { return hstrip(1, Bottom, true); }
  vert_view ghostcells_left_vel_v() noexcept { return vstrip(2, Left, true); }
  vert_view ghostcells_right_vel_v() noexcept { return vstrip(2, Right, true); }
  horiz_view ghostcells_top_vel_v() noexcept { return hstrip(2, Top, true); }
  horiz_view ghostcells_bottom_vel_v() noexcept { return hstrip(2, Bottom, true); }
  vert_view bndrycells_left_Temp() noexcept { return vstrip(0, Left, false); }
  vert_view bndrycells_right_Temp() noexcept { return vstrip(0, Right, false); }
  horiz_view bndrycells_top_Temp() noexcept { return hstrip(0, Top, false); }
  horiz_view bndrycells_bottom_Temp() noexcept { return hstrip(0, Bottom, false); }
  vert_view bndrycells_left_vel_u() noexcept { return vstrip(1, Left, false); }
  vert_view bndrycells_right_vel_u() noexcept { return vstrip(1, Right, false); }
  horiz_view bndrycells_top_vel_u() noexcept { return hstrip(1, Top, false); }
  horiz_view bndrycells_bottom_vel_u() noexcept { return hstrip(1, Bottom, false); }
  vert_view bndrycells_left_vel_v() noexcept { return vstrip(2, Left, false); }
  vert_view bndrycells_right_vel_v() noexcept { return vstrip(2, Right, false); }
  horiz_view bndrycells_top_vel_v() noexcept { return hstrip(2, Top, false); }
  horiz_view bndrycells_bottom_vel_v() noexcept { return hstrip(2, Bottom, false); }

private:
  horiz_view hstrip(int field, Side s, bool ghost) noexcept {
    const std::size_t P = (std::size_t)cy_ + 2;
    const std::size_t col = s == Bottom ? (ghost ? 0 : 1) : (ghost ? cy_ + 1 : cy_);
    return horiz_view(data(field) + P + col, P, (std::size_t)cx_);
  }
  vert_view vstrip(int field, Side s, bool ghost) noexcept {
    const std::size_t P = (std::size_t)cy_ + 2;
    const std::size_t row = s == Left ? (ghost ? 0 : 1) : (ghost ? cx_ + 1 : cx_);
    return vert_view(data(field) + row * P + 1, 1, (std::size_t)cy_);
  }

  real min_x_, max_x_, min_y_, max_y_;
  real dx_, dy_;
  int cx_, cy_;
  matrix temp_, vel_u_, vel_v_;
};

#endif  // KRYLOV_HOST_MESH_HPP_
