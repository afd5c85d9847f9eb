// Python module `krylov`: the class and method names of the reference's
// src/bindpy.cpp on top of the device implementation, so that scripts written
// against the reference (scripts/prob1.py:4-9, :65-98) import and run.
#include <pybind11/functional.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <stdexcept>
#include <string>

#include "boundary_conds.hpp"
#include "constants.hpp"
#include "mesh.hpp"
#include "solver.hpp"
#include "space_disc.hpp"
#include "time_disc.hpp"

namespace py = pybind11;
using BCFn = std::function<real(real, real, real)>;
using Base = TimeDisc<SecondOrderCentered>;

namespace {

py::object field_array(const Mesh &m, const real *data) {
  const int rows = m.cells_x() + 2, cols = m.cells_y() + 2;
  return py::array((py::ssize_t)rows * cols, data).attr("reshape")(rows, cols);
}

// Names that exist in the reference module but whose classes are outside the
// device path (dense LU of the whole system, approximate factorisation).
struct OutOfScopeLU {};

template <typename T> void bind_out_of_scope(py::module_ &m, const char *name, const char *why) {
  const std::string msg = std::string(name) + " is not part of this build: " + why;
  py::class_<T>(m, name).def(py::init([msg](py::args, py::kwargs) -> T * { throw std::runtime_error(msg); }));
}

// timestep() is noexcept and reports a step that could not run as NaN (the reference's
// only failure signal).  An exception raised by one of the Python callables is raised
// again here, so that a typo in a lambda does not silently end the caller's time loop;
// device and library failures stay NaN, with failed() / last_error() for the reason.
template <typename TD> real step_from_python(TD &sim, const real dt) {
  const real d = sim.timestep(dt);
  if (sim.failed()) {
    try {
      sim.rethrow_failure();
    } catch (const py::error_already_set &) {
      throw;
    } catch (...) {
    }
  }
  return d;
}

template <typename SolverT> void bind_implicit_euler(py::module_ &m, const char *name) {
  using IE = ImplicitEuler<SecondOrderCentered, SolverT>;
  using P = std::pair<real, real>;
  using BC = std::pair<BCType, BCFn>;
  py::class_<IE, std::shared_ptr<IE>, Base>(m, name)
      .def(py::init<const P, const P, const size_t, const size_t, std::function<triple(real, real)>,
                    const SecondOrderCentered &, BC, BC, BC, BC, BC, BC, BC, BC, BC, BC, BC, BC>())
      .def("timestep", &step_from_python<IE>)
      .def("system", &IE::sys)
      .def("solution", &IE::sol)
      .def("delta", &IE::delta)
      .def("use_graph", &IE::use_graph);  // extension: replay the step through a hipGraph
}

template <typename SolverT> void bind_gmres(py::module_ &m, const char *name) {
  py::class_<SolverT>(m, name).def(py::init<unsigned long>()).def("solve", &SolverT::solve);
}

}  // namespace

PYBIND11_MODULE(krylov, module) {
  module.doc() = "Solvers for the incompressible energy equation (MI355X device implementation)";

  py::class_<Mesh> mesh(module, "Mesh");
  mesh.def(py::init<std::pair<real, real>, std::pair<real, real>, size_t, size_t,
                    std::function<triple(real, real)>>())
      .def(py::init<std::pair<real, real>, std::pair<real, real>, size_t, size_t>())
      .def("__getitem__", [](const Mesh &m, std::pair<int, int> p) { return m.Temp(p.first, p.second); })
      .def("__setitem__",
           [](Mesh &m, std::pair<int, int> p, real val) {
             m.Temp(p.first, p.second) = val;
             return m;
           })
      .def("cells_x", &Mesh::cells_x)
      .def("cells_y", &Mesh::cells_y)
      .def("median_x", &Mesh::median_x)
      .def("median_y", &Mesh::median_y)
      .def("dx", &Mesh::dx)
      .def("dy", &Mesh::dy)
      .def("Temp_array", [](Mesh &m) { return field_array(m, m.data_Temp()); })
      .def("vel_u_array", [](Mesh &m) { return field_array(m, m.data_vel_u()); })
      .def("vel_v_array", [](Mesh &m) { return field_array(m, m.data_vel_v()); })
      .def("grid_x",
           [](Mesh &m) {
             matrix g(matrix::shape_type{(unsigned long)m.cells_x() + 2, (unsigned long)m.cells_y() + 2});
             for (int i = 0; i < m.cells_x() + 2; i++)
               for (int j = 0; j < m.cells_y() + 2; j++) g(i, j) = m.median_x(i - Mesh::ghost_cells);
             return field_array(m, g.data());
           })
      .def("grid_y", [](Mesh &m) {
        matrix g(matrix::shape_type{(unsigned long)m.cells_x() + 2, (unsigned long)m.cells_y() + 2});
        for (int i = 0; i < m.cells_x() + 2; i++)
          for (int j = 0; j < m.cells_y() + 2; j++) g(i, j) = m.median_y(j - Mesh::ghost_cells);
        return field_array(m, g.data());
      });

  // Boundary conditions on a stand-alone Mesh (host side).  As in the reference
  // the factories take the Mesh as their first argument, the vertical ones use
  // size = dy and the horizontal ones size = dx (src/bindpy.cpp:113, :175).
  enum class BCSideVert { Left, Right };
  using BCV = BoundaryCondition<Mesh::vert_view>;
  py::class_<BCV, std::shared_ptr<BCV>> bc_vert(module, "BoundaryConditionVert");
  auto vert_factory = [](int field) {
    return [field](Mesh &m, const BCSideVert side, const BCType type, const BCFn val) {
      const bool left = side == BCSideVert::Left;
      Mesh::vert_view ghost = field == 0 ? (left ? m.ghostcells_left_Temp() : m.ghostcells_right_Temp())
                              : field == 1 ? (left ? m.ghostcells_left_vel_u() : m.ghostcells_right_vel_u())
                                           : (left ? m.ghostcells_left_vel_v() : m.ghostcells_right_vel_v());
      Mesh::vert_view bndry = field == 0 ? (left ? m.bndrycells_left_Temp() : m.bndrycells_right_Temp())
                              : field == 1 ? (left ? m.bndrycells_left_vel_u() : m.bndrycells_right_vel_u())
                                           : (left ? m.bndrycells_left_vel_v() : m.bndrycells_right_vel_v());
      const Mesh geom = m;  // coordinates are taken from a copy, the strips alias the caller's mesh
      return std::make_shared<BCV>(
          ghost, bndry, [geom, left](int) { return left ? geom.left_x(0) : geom.left_x(geom.cells_x()); },
          [geom](int j) { return geom.median_y(j); }, val, geom.dy(), type);
    };
  };
  bc_vert.def("Temp_bc", vert_factory(0), py::keep_alive<0, 1>())
      .def("vel_u_bc", vert_factory(1), py::keep_alive<0, 1>())
      .def("vel_v_bc", vert_factory(2), py::keep_alive<0, 1>())
      .def("apply", &BCV::apply);
  py::enum_<BCSideVert>(bc_vert, "BCSide").value("left", BCSideVert::Left).value("right", BCSideVert::Right).export_values();

  enum class BCSideHoriz { Top, Bottom };
  using BCH = BoundaryCondition<Mesh::horiz_view>;
  py::class_<BCH, std::shared_ptr<BCH>> bc_horiz(module, "BoundaryConditionHoriz");
  auto horiz_factory = [](int field) {
    return [field](Mesh &m, const BCSideHoriz side, const BCType type, const BCFn val) {
      const bool top = side == BCSideHoriz::Top;
      Mesh::horiz_view ghost = field == 0 ? (top ? m.ghostcells_top_Temp() : m.ghostcells_bottom_Temp())
                               : field == 1 ? (top ? m.ghostcells_top_vel_u() : m.ghostcells_bottom_vel_u())
                                            : (top ? m.ghostcells_top_vel_v() : m.ghostcells_bottom_vel_v());
      Mesh::horiz_view bndry = field == 0 ? (top ? m.bndrycells_top_Temp() : m.bndrycells_bottom_Temp())
                               : field == 1 ? (top ? m.bndrycells_top_vel_u() : m.bndrycells_bottom_vel_u())
                                            : (top ? m.bndrycells_top_vel_v() : m.bndrycells_bottom_vel_v());
      const Mesh geom = m;
      return std::make_shared<BCH>(
          ghost, bndry, [geom](int i) { return geom.median_x(i); },
          [geom, top](int) { return top ? geom.bottom_y(geom.cells_y()) : geom.bottom_y(0); }, val, geom.dx(), type);
    };
  };
  bc_horiz.def("Temp_bc", horiz_factory(0), py::keep_alive<0, 1>())
      .def("vel_u_bc", horiz_factory(1), py::keep_alive<0, 1>())
      .def("vel_v_bc", horiz_factory(2), py::keep_alive<0, 1>())
      .def("apply", &BCH::apply);
  py::enum_<BCSideHoriz>(bc_horiz, "BCSide").value("top", BCSideHoriz::Top).value("bottom", BCSideHoriz::Bottom).export_values();

  py::enum_<BCType>(module, "BCType").value("dirichlet", BCType::Dirichlet).value("neumann", BCType::Neumann).export_values();

  py::class_<SecondOrderCentered>(module, "SecondOrderCentered")
      .def(py::init<const real, const real, const real, const real>())
      .def("Dx_m1", &SecondOrderCentered::Dx_m1)
      .def("Dx_0", &SecondOrderCentered::Dx_0)
      .def("Dx_p1", &SecondOrderCentered::Dx_p1)
      .def("Dy_m1", &SecondOrderCentered::Dy_m1)
      .def("Dy_0", &SecondOrderCentered::Dy_0)
      .def("Dy_p1", &SecondOrderCentered::Dy_p1);

  py::class_<Base, std::shared_ptr<Base>>(module, "_TimeDiscBase")
      .def("mesh", &Base::mesh)
      .def("space_disc", &Base::space_disc)
      .def("cur_time", &Base::cur_time)
      .def("failed", &Base::failed)           // extensions: why the last timestep() returned NaN
      .def("last_error", &Base::last_error);

  bind_out_of_scope<OutOfScopeLU>(module, "ImplicitEulerLUSolver",
                                  "the dense LU solve of the whole N x N system is outside the device path");
  {
    // src/bindpy.cpp:270-295
    using AF = ImplicitEulerApproxFact<SecondOrderCentered>;
    using P = std::pair<real, real>;
    using BC = std::pair<BCType, BCFn>;
    py::class_<AF, std::shared_ptr<AF>, Base>(module, "ImplicitEulerAFSolver")
        .def(py::init<const P, const P, const size_t, const size_t, std::function<triple(real, real)>,
                      const SecondOrderCentered &, BC, BC, BC, BC, BC, BC, BC, BC, BC, BC, BC, BC>())
        .def("timestep", &step_from_python<AF>)
        .def("assemble_system_dx", &AF::assemble_system_dx)
        .def("assemble_system_dy", &AF::assemble_system_dy);
  }
  bind_implicit_euler<GMRESSolver<10>>(module, "ImplicitEulerGMRESSolver10");
  bind_implicit_euler<GMRESSolver<20>>(module, "ImplicitEulerGMRESSolver20");
  bind_implicit_euler<GMRESSolver<30>>(module, "ImplicitEulerGMRESSolver30");
  bind_implicit_euler<GMRESSolver<40>>(module, "ImplicitEulerGMRESSolver40");
  bind_implicit_euler<GMRESSolver<50>>(module, "ImplicitEulerGMRESSolver50");
  bind_implicit_euler<GMRESTDPSolver<10>>(module, "ImplicitEulerGMRESTDPSolver10");
  bind_implicit_euler<GMRESTDPSolver<20>>(module, "ImplicitEulerGMRESTDPSolver20");
  bind_implicit_euler<GMRESTDPSolver<30>>(module, "ImplicitEulerGMRESTDPSolver30");
  bind_implicit_euler<GMRESTDPSolver<40>>(module, "ImplicitEulerGMRESTDPSolver40");
  bind_implicit_euler<GMRESTDPSolver<50>>(module, "ImplicitEulerGMRESTDPSolver50");

  py::class_<matrix>(module, "Matrix")
      .def(py::init<matrix::shape_type>())
      .def("shape", [](const matrix &m) { return m.shape(); })
      .def("__getitem__", [](const matrix &m, std::pair<int, int> idx) { return m(idx.first, idx.second); })
      .def("__setitem__",
           [](matrix &dest, std::pair<int, int> idx, real val) {
             dest(idx.first, idx.second) = val;
             return val;
           })
      .def("to_array",
           [](const matrix &src) {
             return py::array((py::ssize_t)(src.shape()[0] * src.shape()[1]), src.data())
                 .attr("reshape")(src.shape()[0], src.shape()[1]);
           })
      .def_static("from_array", [](const py::array_t<real> &np_arr) {
        matrix m(matrix::shape_type{(unsigned long)np_arr.shape(0), (unsigned long)np_arr.shape(1)});
        auto src = np_arr.unchecked<2>();
        for (py::ssize_t i = 0; i < src.shape(0); i++)
          for (py::ssize_t j = 0; j < src.shape(1); j++) m(i, j) = src(i, j);
        return m;
      });

  py::class_<vector>(module, "Vector")
      .def(py::init<vector::shape_type>())
      .def("shape", [](const vector &v) { return v.shape(); })
      .def("__getitem__", [](const vector &v, int idx) { return v(idx); })
      .def("__setitem__",
           [](vector &dest, int idx, real val) {
             dest(idx) = val;
             return val;
           })
      .def("to_array", [](const vector &src) { return py::array((py::ssize_t)src.shape()[0], src.data()); })
      .def_static("from_array", [](const py::array_t<real> &np_arr) {
        vector v(vector::shape_type{(unsigned long)np_arr.shape(0)});
        auto src = np_arr.unchecked<1>();
        for (py::ssize_t i = 0; i < src.shape(0); i++) v(i) = src(i);
        return v;
      });

  py::class_<LUSolver>(module, "LUSolver")
      .def(py::init<unsigned long>())
      .def("solve", &LUSolver::solve)
      .def("lu_decomp", [](LUSolver &s, const matrix &a) {
        auto lu = s.lu_decomp(a);
        return std::make_pair(matrix(lu.first), matrix(lu.second));
      });
  py::class_<TriDiagSolver>(module, "TriDiagSolver").def(py::init<unsigned long>()).def("solve", &TriDiagSolver::solve);

  bind_gmres<GMRESSolver<10>>(module, "GMRESSolver10");
  bind_gmres<GMRESSolver<20>>(module, "GMRESSolver20");
  bind_gmres<GMRESSolver<30>>(module, "GMRESSolver30");
  bind_gmres<GMRESSolver<40>>(module, "GMRESSolver40");
  bind_gmres<GMRESSolver<50>>(module, "GMRESSolver50");
  bind_gmres<GMRESTDPSolver<10>>(module, "GMRESTDPSolver10");
  bind_gmres<GMRESTDPSolver<20>>(module, "GMRESTDPSolver20");
  bind_gmres<GMRESTDPSolver<30>>(module, "GMRESTDPSolver30");
  bind_gmres<GMRESTDPSolver<40>>(module, "GMRESTDPSolver40");
  bind_gmres<GMRESTDPSolver<50>>(module, "GMRESTDPSolver50");
}
