// SecondOrderCentered: the physical parameters handed to the device, plus the
// reference's coefficient accessors evaluated on a host Mesh
// (include/space_disc.hpp:83-113) for inspection from Python.
#ifndef KRYLOV_HOST_SPACE_DISC_HPP_
#define KRYLOV_HOST_SPACE_DISC_HPP_

#include "constants.hpp"
#include "mesh.hpp"

class SecondOrderCentered {
public:
  constexpr SecondOrderCentered(const real diffusion, const real reynolds, const real prandtl,
                                const real eckert) noexcept
      : diffusion_(diffusion), reynolds_(reynolds), prandtl_(prandtl), eckert_(eckert),
        inv_rp_(1.0 / (reynolds * prandtl)) {}

  real diffusion() const noexcept { return diffusion_; }
  real reynolds() const noexcept { return reynolds_; }
  real prandtl() const noexcept { return prandtl_; }
  real eckert() const noexcept { return eckert_; }

  real Dx_p1(const Mesh &m, int i, int j) const noexcept {
    return (m.vel_u(i + 1, j) / 2.0 - diffusion_ * inv_rp_ / m.dx()) / m.dx();
  }
  real Dx_0(const Mesh &m, int, int) const noexcept { return 2.0 * diffusion_ * inv_rp_ / (m.dx() * m.dx()); }
  real Dx_m1(const Mesh &m, int i, int j) const noexcept {
    return (-m.vel_u(i - 1, j) / 2.0 - diffusion_ * inv_rp_ / m.dx()) / m.dx();
  }
  real Dy_p1(const Mesh &m, int i, int j) const noexcept {
    return (m.vel_v(i, j + 1) / 2.0 - diffusion_ * inv_rp_ / m.dy()) / m.dy();
  }
  real Dy_0(const Mesh &m, int, int) const noexcept { return 2.0 * diffusion_ * inv_rp_ / (m.dy() * m.dy()); }
  real Dy_m1(const Mesh &m, int i, int j) const noexcept {
    return (-m.vel_v(i, j - 1) / 2.0 - diffusion_ * inv_rp_ / m.dy()) / m.dy();
  }

private:
  real diffusion_, reynolds_, prandtl_, eckert_, inv_rp_;
};

#endif  // KRYLOV_HOST_SPACE_DISC_HPP_
