// TimeDisc / ImplicitEuler with the interface of the reference's
// include/time_disc.hpp; all state lives on the device (kry_ie_*), the host
// object keeps the configuration, the boundary functions and a mirror Mesh
// that is refreshed on demand.
#ifndef KRYLOV_HOST_TIME_DISC_HPP_
#define KRYLOV_HOST_TIME_DISC_HPP_

#include <exception>
#include <functional>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "boundary_conds.hpp"
#include "mesh.hpp"
#include "solver.hpp"
#include "space_disc.hpp"

using BCSpec = std::pair<BCType, std::function<real(real, real, real)>>;

inline BCSpec default_bc() {
  return {BCType::Dirichlet, [](real, real, real) { return 0.0; }};
}

constexpr unsigned long num_eqns(int num_x_cells, int num_y_cells, int num_ghost_cells) {
  return (num_x_cells + 2 * num_ghost_cells) * (num_y_cells + 2 * num_ghost_cells) - 4;
}

template <typename SpaceDisc_> class TimeDisc {
public:
  using SpaceDisc = SpaceDisc_;

  // BC order as in the reference: temperature bottom, top, left, right, then
  // the same for vel_u and vel_v (include/time_disc.hpp:23-50).
  TimeDisc(const std::pair<real, real> corner_1, const std::pair<real, real> corner_2, const size_t x_cells,
           const size_t y_cells, std::function<triple(real, real)> initial_t_u_v, const SpaceDisc &space_,
           BCSpec bottom_temp = default_bc(), BCSpec top_temp = default_bc(), BCSpec left_temp = default_bc(),
           BCSpec right_temp = default_bc(), BCSpec bottom_u = default_bc(), BCSpec top_u = default_bc(),
           BCSpec left_u = default_bc(), BCSpec right_u = default_bc(), BCSpec bottom_v = default_bc(),
           BCSpec top_v = default_bc(), BCSpec left_v = default_bc(), BCSpec right_v = default_bc())
      : cur_state(corner_1, corner_2, x_cells, y_cells, initial_t_u_v), space(space_), corner_1_(corner_1),
        corner_2_(corner_2),
        bcs_{bottom_temp, top_temp, left_temp, right_temp, bottom_u, top_u, left_u, right_u, bottom_v, top_v, left_v,
             right_v} {}

  virtual ~TimeDisc() {
    if (ie_) kry_ie_destroy(ie_);
  }
  TimeDisc(const TimeDisc &) = delete;
  TimeDisc &operator=(const TimeDisc &) = delete;

  // Host copy of the current state (fields are fetched from the device).
  const Mesh &mesh() {
    if (ie_)
      for (int f = 0; f < 3; f++) krylov::check(kry_ie_get_field(ie_, f, cur_state.data(f)), "kry_ie_get_field");
    return cur_state;
  }
  const SpaceDisc &space_disc() noexcept { return space; }
  real cur_time() const noexcept { return ie_ ? kry_ie_time(ie_) : 0.0; }
  real diffusion() const noexcept { return space.diffusion(); }
  real reynolds() const noexcept { return space.reynolds(); }
  real prandtl() const noexcept { return space.prandtl(); }
  real eckert() const noexcept { return space.eckert(); }

  // timestep() is noexcept as in the reference (include/time_disc.hpp:277, :443), where
  // failure only ever shows as NaN.  A step that could not run (a device or library
  // failure, an exception out of a boundary function) likewise returns NaN; what happened
  // is kept here.  No counterpart in the reference.
  bool failed() const noexcept { return static_cast<bool>(failure_); }
  const std::string &last_error() const noexcept { return last_error_; }
  // Rethrows the failure of the last timestep(), if any (for callers that prefer
  // exceptions, e.g. the Python module for exceptions raised by Python callables).
  void rethrow_failure() const {
    if (failure_) std::rethrow_exception(failure_);
  }

protected:
  // Runs one step's host side; any exception becomes the NaN result.
  template <typename F> real guarded_step(F &&step) noexcept {
    failure_ = nullptr;
    last_error_.clear();
    try {
      return step();
    } catch (const std::exception &e) {
      failure_ = std::current_exception();
      try { last_error_ = e.what(); } catch (...) {}
    } catch (...) {
      failure_ = std::current_exception();
    }
    return q_nan;
  }
  std::exception_ptr failure_;
  std::string last_error_;

  // Creates the device object for a solver with m iterations.
  void create_device(int m, int flags) {
    const int cx = cur_state.cells_x(), cy = cur_state.cells_y();
    std::vector<real> T((size_t)cx * cy), u((size_t)cx * cy), v((size_t)cx * cy);
    for (int i = 0; i < cx; i++)
      for (int j = 0; j < cy; j++) {
        T[(size_t)i * cy + j] = cur_state.Temp(i, j);
        u[(size_t)i * cy + j] = cur_state.vel_u(i, j);
        v[(size_t)i * cy + j] = cur_state.vel_v(i, j);
      }
    int dirichlet[12];
    for (int k = 0; k < 12; k++) dirichlet[k] = bcs_[k].first == BCType::Dirichlet;
    krylov::check(kry_ie_create(krylov::default_context(), corner_1_.first, corner_1_.second, corner_2_.first,
                                corner_2_.second, cx, cy, T.data(), u.data(), v.data(), space.diffusion(),
                                space.reynolds(), space.prandtl(), space.eckert(), dirichlet, m, flags, &ie_),
                  "kry_ie_create");
  }

  // Samples the twelve boundary functions at the reference's coordinates
  // (include/time_disc.hpp:55-76) for time t; out[f * 4 + side] holds one
  // value per boundary cell of that side.
  void sample_bc_values(const real t, std::vector<real> (&out)[12]) const {
    const int cx = cur_state.cells_x(), cy = cur_state.cells_y();
    for (int f = 0; f < 3; f++)
      for (int side = 0; side < 4; side++) {
        const auto &fn = bcs_[f * 4 + side].second;
        const bool horiz = side == KRY_SIDE_BOTTOM || side == KRY_SIDE_TOP;
        std::vector<real> &vals = out[f * 4 + side];
        vals.resize(horiz ? cx : cy);
        for (int k = 0; k < (int)vals.size(); k++) {
          real x, y;
          if (side == KRY_SIDE_BOTTOM) { x = cur_state.median_x(k); y = cur_state.bottom_y(0); }
          else if (side == KRY_SIDE_TOP) { x = cur_state.median_x(k); y = cur_state.top_y(cy - 1); }
          else if (side == KRY_SIDE_LEFT) { x = cur_state.left_x(0); y = cur_state.median_y(k); }
          else { x = cur_state.right_x(cx - 1); y = cur_state.median_y(k); }
          vals[k] = fn(x, y, t);
        }
      }
  }

  // Hands sampled values to the device object (waits for queued device work
  // that may still read the previous values).
  void stage_bc_values(const std::vector<real> (&vals)[12], const real t) {
    for (int f = 0; f < 3; f++)
      for (int side = 0; side < 4; side++)
        krylov::check(kry_ie_set_bc_values(ie_, f, side, vals[f * 4 + side].data(), (int)vals[f * 4 + side].size()),
                      "kry_ie_set_bc_values");
    staged_ = true;
    staged_time_ = t;
  }

  // Boundary values of the current time on the device, sampling only if the
  // staged ones belong to another time.
  void push_bc_values() {
    const real t = cur_time();
    if (staged_ && staged_time_ == t) return;
    sample_bc_values(t, bc_scratch_);
    stage_bc_values(bc_scratch_, t);
  }

  void apply_bcs() {
    push_bc_values();
    krylov::check(kry_ie_apply_bcs(ie_), "kry_ie_apply_bcs");
  }

  Mesh cur_state;
  SpaceDisc space;
  std::pair<real, real> corner_1_, corner_2_;
  BCSpec bcs_[12];
  kry_ie *ie_ = nullptr;
  std::vector<real> bc_scratch_[12];
  bool staged_ = false;
  real staged_time_ = 0.0;
};

namespace krylov {
// Whether a solver type carries what the device step needs to know about it.
template <typename S, typename = void> struct is_device_solver : std::false_type {};
template <typename S>
struct is_device_solver<S, std::void_t<decltype(S::iterations), decltype(S::device_flags)>> : std::true_type {};
}  // namespace krylov

// The reference composes ImplicitEuler<SpaceDisc_, Solver_> from ANY pair of types with the right members
// (include/time_disc.hpp:190-191: the solver through Solver_(n), solve(), prev_result(), :234, :280, :380; the
// discretisation through Dx_*/Dy_*/flux_integral/source_fd, :268-302).  The device step is narrower, and says so
// at compile time instead of silently computing something else: the physics on the device is that of
// SecondOrderCentered (only its four parameters travel to the device), and the solver is one of the device
// solvers (GMRESSolver<m>, GMRESTDPSolver<m>), which tell the device object their cycle length and kind.
template <typename SpaceDisc_, typename Solver_> class ImplicitEuler : public TimeDisc<SpaceDisc_> {
  static_assert(std::is_same<SpaceDisc_, SecondOrderCentered>::value,
                "ImplicitEuler<SpaceDisc_, Solver_> (reference seam include/time_disc.hpp:190-191): the device step "
                "implements the assembly of SecondOrderCentered only; another SpaceDisc_ would be ignored");
  static_assert(krylov::is_device_solver<Solver_>::value,
                "ImplicitEuler<SpaceDisc_, Solver_> (reference seam include/time_disc.hpp:190-191): Solver_ must be a "
                "device solver providing `iterations` and `device_flags` (GMRESSolver<m> or GMRESTDPSolver<m>)");

public:
  using Solver = Solver_;
  using SpaceDisc = SpaceDisc_;
  using Base = TimeDisc<SpaceDisc_>;

  template <typename... BCs>
  ImplicitEuler(const std::pair<real, real> corner_1, const std::pair<real, real> corner_2, const size_t cells_x,
                const size_t cells_y, std::function<triple(real, real)> initial_t_u_v, const SpaceDisc &space,
                BCs... bcs)
      : Base(corner_1, corner_2, cells_x, cells_y, initial_t_u_v, space, bcs...),
        n_(num_eqns((int)cells_x, (int)cells_y, Mesh::ghost_cells)) {
    this->create_device((int)Solver::iterations, Solver::device_flags);
    // second half of the reference constructor (include/time_disc.hpp:246-274)
    this->push_bc_values();
    krylov::check(kry_ie_init_offdiag(this->ie_), "kry_ie_init_offdiag");
  }

  // The step is queued on the device and, while it runs, the boundary
  // functions are sampled for the time the NEXT step will start from (they
  // depend on position and time only); for Python callables that is the
  // expensive host part of a step, and it is hidden behind the device work.
  // If a boundary function throws while the queued step runs, the step itself is still waited for (its
  // result is dropped, the returned value is NaN), and nothing stays staged for the new time: the next
  // timestep() samples the boundary values again before it queues anything.
  real timestep(const real dt) noexcept {
    return this->guarded_step([&]() -> real {
      this->push_bc_values();
      krylov::check(kry_ie_timestep_async(this->ie_, dt), "kry_ie_timestep_async");
      const real t_next = this->cur_time();
      real max_delta = 0.0;
      try {
        this->sample_bc_values(t_next, this->bc_scratch_);
      } catch (...) {
        (void)kry_ie_last_max_delta(this->ie_, &max_delta);  // drain the step that is in flight
        this->staged_ = false;
        throw;
      }
      krylov::check(kry_ie_last_max_delta(this->ie_, &max_delta), "kry_ie_last_max_delta");
      this->stage_bc_values(this->bc_scratch_, t_next);
      return max_delta;
    });
  }

  std::pair<const matrix &, const vector &> assemble_system(const real dt) {
    this->push_bc_values();
    krylov::check(kry_ie_assemble_system(this->ie_, dt), "kry_ie_assemble_system");
    return {sys(), sol()};
  }

  // include/time_disc.hpp:341-376
  unsigned long cell_idx(const int i, const int j) const noexcept {
    const int cy = this->cur_state.cells_y(), gc = Mesh::ghost_cells;
    if (i == -gc) return j;
    if (i == this->cur_state.cells_x()) return cy + i * (cy + 2 * gc) + j;
    return cy + i * (cy + 2 * gc) + gc + j;
  }

  // Dense copies in the reference's numbering (small meshes only).
  const matrix &sys() {
    system_ = matrix(matrix::shape_type{n_, n_});
    krylov::check(kry_ie_get_dense_system(this->ie_, system_.data()), "kry_ie_get_dense_system");
    return system_;
  }
  const vector &sol() {
    sol_vec_ = vector(vector::shape_type{n_});
    krylov::check(kry_ie_get_rhs(this->ie_, sol_vec_.data()), "kry_ie_get_rhs");
    return sol_vec_;
  }
  const vector &delta() {
    delta_ = vector(vector::shape_type{n_});
    krylov::check(kry_ie_get_delta(this->ie_, delta_.data()), "kry_ie_get_delta");
    return delta_;
  }
  void use_graph(bool on) { krylov::check(kry_ie_use_graph(this->ie_, on ? 1 : 0), "kry_ie_use_graph"); }

protected:
  unsigned long n_;
  matrix system_;
  vector sol_vec_, delta_;
};

// include/time_disc.hpp:388-540: per step one tridiagonal sweep along y and one
// along x, both on the device.
template <typename SpaceDisc_> class ImplicitEulerApproxFact : public TimeDisc<SpaceDisc_> {
public:
  using SpaceDisc = SpaceDisc_;
  using Base = TimeDisc<SpaceDisc_>;

  template <typename... BCs>
  ImplicitEulerApproxFact(const std::pair<real, real> corner_1, const std::pair<real, real> corner_2,
                          const size_t cells_x, const size_t cells_y,
                          std::function<triple(real, real)> initial_t_u_v, const SpaceDisc &space, BCs... bcs)
      : Base(corner_1, corner_2, cells_x, cells_y, initial_t_u_v, space, bcs...) {
    this->create_device(0, 0);  // no Krylov solver: its storage holds the line systems
    this->apply_bcs();          // include/time_disc.hpp:440
  }

  real timestep(const real dt) noexcept {
    return this->guarded_step([&]() -> real {
      this->push_bc_values();
      real max_delta = 0.0;
      krylov::check(kry_ie_timestep_approx_fact(this->ie_, dt, &max_delta), "kry_ie_timestep_approx_fact");
      return max_delta;
    });
  }

  const matrix &assemble_system_dy(const real dt, const int i) { return line_system(system_dy_, true, i, dt); }
  const matrix &assemble_system_dx(const real dt, const int j) { return line_system(system_dx_, false, j, dt); }

protected:
  const matrix &line_system(matrix &out, const bool along_y, const int index, const real dt) {
    this->apply_bcs();
    const size_t rows = (along_y ? this->cur_state.cells_y() : this->cur_state.cells_x()) + 2 * Mesh::ghost_cells;
    out = matrix(matrix::shape_type{rows, 3});
    krylov::check(kry_ie_get_af_system(this->ie_, along_y ? 1 : 0, index, dt, out.data()), "kry_ie_get_af_system");
    return out;
  }
  matrix system_dx_, system_dy_;
};

#endif  // KRYLOV_HOST_TIME_DISC_HPP_
