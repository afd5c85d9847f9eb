// Boundary-condition description with the interface of the reference's
// include/boundary_conds.hpp.  Inside a time-stepping object only the
// function, type and size are used (the ghost fill runs on the device); the
// host-side apply() serves stand-alone Mesh objects.
#ifndef KRYLOV_HOST_BOUNDARY_CONDS_HPP_
#define KRYLOV_HOST_BOUNDARY_CONDS_HPP_

#include <functional>
#include <memory>

#include "constants.hpp"

inline real homogeneous(real, real, real) { return 0.0; }
inline real null_coord(int) { return 0.0; }

enum class BCType { Dirichlet, Neumann };

template <typename viewt> class BoundaryCondition : public std::enable_shared_from_this<BoundaryCondition<viewt>> {
public:
  using bc_function = std::function<real(real, real, real)>;
  using cell_coord_function = std::function<real(int)>;

  BoundaryCondition(const viewt &ghost_cells, const viewt &bndry_cells, const cell_coord_function &bndry_x,
                    const cell_coord_function &bndry_y, const bc_function &bc, const real size, const real weight)
      : ghost_cells_(ghost_cells), bndry_cells_(bndry_cells), bc_(bc), bndry_x_(bndry_x), bndry_y_(bndry_y),
        size_(size), weight_(weight) {}
  BoundaryCondition(const viewt &ghost_cells, const viewt &bndry_cells, const cell_coord_function &bndry_x,
                    const cell_coord_function &bndry_y, const bc_function &bc, const real size, const BCType type)
      : BoundaryCondition(ghost_cells, bndry_cells, bndry_x, bndry_y, bc, size,
                          type == BCType::Dirichlet ? 1.0 : 0.0) {}
  BoundaryCondition(const viewt &ghost_cells, const viewt &bndry_cells, const real size, const real weight)
      : BoundaryCondition(ghost_cells, bndry_cells, null_coord, null_coord, homogeneous, size, weight) {}
  BoundaryCondition(const viewt &ghost_cells, const viewt &bndry_cells, const real size, const BCType type)
      : BoundaryCondition(ghost_cells, bndry_cells, null_coord, null_coord, homogeneous, size, type) {}

  // Robin-weighted ghost value, include/boundary_conds.hpp:62-66
  real ghost_cell(const real bndry_val, const real x, const real y, const real time) const {
    const real bc = bc_(x, y, time);
    return (2.0 * bc - bndry_val) * weight_ + (size_ * bc + bndry_val) * (1.0 - weight_);
  }

  void apply(const real time) {
    for (unsigned long idx = 0; idx < bndry_cells_.shape()[0]; idx++)
      ghost_cells_(idx) = ghost_cell(bndry_cells_(idx), bndry_x_(idx), bndry_y_(idx), time);
  }

  real implicit_euler_term() const noexcept { return 2.0 * weight_ - 1.0; }
  real weight() const noexcept { return weight_; }

protected:
  viewt ghost_cells_, bndry_cells_;
  bc_function bc_;
  cell_coord_function bndry_x_, bndry_y_;
  real size_, weight_;
};

template <typename viewt> class DirichletBC : public BoundaryCondition<viewt> {
public:
  using Base = BoundaryCondition<viewt>;
  DirichletBC(const viewt &ghost_cells, const viewt &bndry_cells, const typename Base::cell_coord_function &bndry_x,
              const typename Base::cell_coord_function &bndry_y, const typename Base::bc_function &bc)
      : Base(ghost_cells, bndry_cells, bndry_x, bndry_y, bc, 0.0, 1.0) {}
  DirichletBC(const viewt &ghost_cells, const viewt &bndry_cells) : Base(ghost_cells, bndry_cells, 0.0, 1.0) {}
};

template <typename viewt> class NeumannBC : public BoundaryCondition<viewt> {
public:
  using Base = BoundaryCondition<viewt>;
  NeumannBC(const viewt &ghost_cells, const viewt &bndry_cells, const typename Base::cell_coord_function &bndry_x,
            const typename Base::cell_coord_function &bndry_y, const typename Base::bc_function &bc, const real size)
      : Base(ghost_cells, bndry_cells, bndry_x, bndry_y, bc, size, 0.0) {}
  NeumannBC(const viewt &ghost_cells, const viewt &bndry_cells, const real = 0.0)
      : Base(ghost_cells, bndry_cells, 0.0, 0.0) {}
};

#endif  // KRYLOV_HOST_BOUNDARY_CONDS_HPP_
