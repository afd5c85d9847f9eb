// Boundary conditions for stand-alone Mesh objects on the host, under the
// reference's public names (include/boundary_conds.hpp): BCType,
// BoundaryCondition<strip>::apply / ghost_cell / implicit_euler_term, and the
// DirichletBC / NeumannBC shorthands.  Inside a time-stepping object the ghost
// fill runs on the device (kry_ie_apply_bcs) and nothing of this file is used
// except BCType.
//
// One condition is a pair (weight w, size s): the ghost value for an interior
// neighbour b and a prescribed value c is
//     w (2 c - b) + (1 - w) (s c + b),
// so w = 1 mirrors about c (Dirichlet) and w = 0 extrapolates a gradient
// (Neumann).  The arithmetic lives in krylov::GhostRule; the class template
// only walks a strip of cells.
#ifndef KRYLOV_HOST_BOUNDARY_CONDS_HPP_
#define KRYLOV_HOST_BOUNDARY_CONDS_HPP_

#include <functional>
#include <memory>
#include <utility>

#include "constants.hpp"

enum class BCType { Dirichlet, Neumann };

inline real homogeneous(real, real, real) { return 0.0; }
inline real null_coord(int) { return 0.0; }

namespace krylov {

struct GhostRule {
  real size = 0.0;
  real weight = 1.0;

  GhostRule() = default;
  GhostRule(real s, real w) : size(s), weight(w) {}
  GhostRule(real s, BCType t) : size(s), weight(t == BCType::Dirichlet ? 1.0 : 0.0) {}

  // same operation order as include/boundary_conds.hpp:62-66
  real ghost(real inner, real prescribed) const {
    return (2.0 * prescribed - inner) * weight + (size * prescribed + inner) * (1.0 - weight);
  }
  // coefficient of the ghost unknown in an implicit system row (the interior
  // neighbour's coefficient is 1), include/boundary_conds.hpp:77-81
  real system_term() const { return 2.0 * weight - 1.0; }
};

}  // namespace krylov

template <typename Strip> class BoundaryCondition : public std::enable_shared_from_this<BoundaryCondition<Strip>> {
public:
  using bc_function = std::function<real(real, real, real)>;
  using cell_coord_function = std::function<real(int)>;

  // Argument order of the reference's constructors; the last argument is
  // either the weight itself or a BCType.
  template <typename WeightOrType>
  BoundaryCondition(const Strip &ghost, const Strip &inner, cell_coord_function x_of, cell_coord_function y_of,
                    bc_function value, const real size, const WeightOrType w)
      : ghost_(ghost), inner_(inner), x_of_(std::move(x_of)), y_of_(std::move(y_of)), value_(std::move(value)),
        rule_(size, w) {}

  // homogeneous condition
  template <typename WeightOrType>
  BoundaryCondition(const Strip &ghost, const Strip &inner, const real size, const WeightOrType w)
      : BoundaryCondition(ghost, inner, null_coord, null_coord, homogeneous, size, w) {}

  real ghost_cell(const real inner_val, const real x, const real y, const real time) const {
    return rule_.ghost(inner_val, value_(x, y, time));
  }

  // The strip length is taken from the interior strip (the ghost strip of the
  // right temperature column is declared one short in the reference, Mesh
  // hands out equally long ones here).
  void apply(const real time) {
    const unsigned long cells = inner_.shape()[0];
    for (unsigned long k = 0; k < cells; k++) ghost_(k) = ghost_cell(inner_(k), x_of_((int)k), y_of_((int)k), time);
  }

  real implicit_euler_term() const noexcept { return rule_.system_term(); }
  real weight() const noexcept { return rule_.weight; }

private:
  Strip ghost_, inner_;
  cell_coord_function x_of_, y_of_;
  bc_function value_;
  krylov::GhostRule rule_;
};

// Shorthands with fixed weights.
template <typename Strip> struct DirichletBC : BoundaryCondition<Strip> {
  using B = BoundaryCondition<Strip>;
  DirichletBC(const Strip &ghost, const Strip &inner, typename B::cell_coord_function x_of,
              typename B::cell_coord_function y_of, typename B::bc_function value)
      : B(ghost, inner, std::move(x_of), std::move(y_of), std::move(value), 0.0, 1.0) {}
  DirichletBC(const Strip &ghost, const Strip &inner) : B(ghost, inner, 0.0, 1.0) {}
};

template <typename Strip> struct NeumannBC : BoundaryCondition<Strip> {
  using B = BoundaryCondition<Strip>;
  NeumannBC(const Strip &ghost, const Strip &inner, typename B::cell_coord_function x_of,
            typename B::cell_coord_function y_of, typename B::bc_function value, const real size)
      : B(ghost, inner, std::move(x_of), std::move(y_of), std::move(value), size, 0.0) {}
  NeumannBC(const Strip &ghost, const Strip &inner, const real /*size has no effect on a zero gradient*/ = 0.0)
      : B(ghost, inner, 0.0, 0.0) {}
};

#endif  // KRYLOV_HOST_BOUNDARY_CONDS_HPP_
