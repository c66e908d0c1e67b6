// pybind_l2f.cpp — the binding a maintainer of l2f's C++ Python module would write over libraptor_quad.so
// (INTEGRATION.md section 3): the names, argument order and array conventions of the reference's example
// (/root/reference/README.md:48-61,94-99) as a pybind11 module.  Everything forwards to the C ABI of
// include/raptor_quad.h; failures become Python exceptions carrying rq_last_error().
//
//   g++ -O2 -std=c++17 -shared -fPIC $(python -m pybind11 --includes) -Iinclude examples/pybind_l2f.cpp \
//       -Lraptor_amd -lraptor_quad -Wl,-rpath,$PWD/raptor_amd -o l2f_mi355x$(python3-config --extension-suffix)
//
// tests/test_gpu_boundary.py::test_pybind11_binding_runs_the_readme_loop builds it and runs the README loop through
// it on the GPU, against the ctypes binding.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <stdexcept>
#include <vector>

#include "raptor_quad.h"

namespace py = pybind11;

static void check(int rc) {
    if (rc != RQ_OK) throw std::runtime_error(std::string(rq_status_string(rc)) + ": " + rq_last_error());
}

struct Device {
    rq_device* h = nullptr;
    explicit Device(int ordinal) { check(rq_device_create(ordinal, &h)); }
    ~Device() { rq_device_destroy(h); }
    Device(const Device&) = delete;
};
struct Rng {
    rq_rng* h = nullptr;
    ~Rng() { rq_rng_destroy(h); }
};
struct Environment {
    rq_env* h = nullptr;
    uint32_t n;
    explicit Environment(uint32_t n_envs) : n(n_envs) {}
    ~Environment() { rq_env_destroy(h); }
};
struct Parameters {
    rq_params* h = nullptr;
    ~Parameters() { rq_params_destroy(h); }
};
struct State {
    rq_state* h = nullptr;
    ~State() { rq_state_destroy(h); }
};
struct Raptor {
    rq_policy* h = nullptr;
    Raptor(Device& d, py::array_t<float, py::array::c_style | py::array::forcecast> weights) {
        check(rq_policy_create(d.h, weights.data(), (size_t)weights.size(), &h));
    }
    ~Raptor() { rq_policy_destroy(h); }
};

using Array = py::array_t<float, py::array::c_style>;

PYBIND11_MODULE(l2f_mi355x, m) {
    m.attr("OBSERVATION_DIM") = RQ_OBSERVATION_DIM;
    m.attr("ACTION_DIM") = RQ_ACTION_DIM;
    py::class_<Device>(m, "Device").def(py::init<int>(), py::arg("ordinal") = 0);
    py::class_<Rng>(m, "VectorRng").def(py::init<>());
    py::class_<Environment>(m, "VectorEnvironment")
        .def(py::init<uint32_t>(), py::arg("n_environments"))
        .def_property_readonly("N_ENVIRONMENTS", [](const Environment& e) {
            uint32_t n = e.n;
            if (e.h) check(rq_env_num_envs(e.h, &n));                                            // README.md:55
            return n;
        });
    py::class_<Parameters>(m, "VectorParameters").def(py::init<>());
    py::class_<State>(m, "VectorState").def(py::init<>()).def("assign", [](State& s, const State& other) {
        check(rq_state_assign(s.h, other.h));                                                // README.md:99
    });
    py::class_<Raptor>(m, "Raptor")
        .def(py::init<Device&, py::array_t<float, py::array::c_style | py::array::forcecast>>())
        .def("reset", [](Raptor& p) { check(rq_policy_reset(p.h)); })                        // README.md:94
        .def("evaluate_step", [](Raptor& p, py::array_t<float, py::array::c_style | py::array::forcecast> obs) {
            if (obs.ndim() != 2 || obs.shape(1) < RQ_POLICY_INPUT_DIM) throw std::invalid_argument("observation must be [B, >= 22]");
            Array act({(py::ssize_t)obs.shape(0), (py::ssize_t)RQ_ACTION_DIM});
            check(rq_policy_evaluate_step(p.h, nullptr, obs.data(), (uint32_t)obs.shape(0), (uint32_t)obs.shape(1),
                                          act.mutable_data()));                              // README.md:97
            return act;
        });

    m.def("initialize_rng", [](Device& d, Rng& r, uint64_t seed) {                            // README.md:58
        if (!r.h) check(rq_rng_create(d.h, &r.h));
        check(rq_initialize_rng(d.h, r.h, seed));
    });
    m.def("initialize_environment", [](Device& d, Environment& e) {                           // README.md:59
        if (!e.h) check(rq_env_create(d.h, e.n, 0, &e.h));
        check(rq_initialize_environment(d.h, e.h));
    });
    m.def("sample_initial_parameters", [](Device& d, Environment& e, Parameters& p, Rng& r) {  // README.md:60
        if (!p.h) check(rq_params_create(e.h, &p.h));
        check(rq_sample_initial_parameters(d.h, e.h, p.h, r.h));
    });
    m.def("sample_initial_state", [](Device& d, Environment& e, Parameters& p, State& s, Rng& r) {  // README.md:61
        if (!s.h) check(rq_state_create(e.h, &s.h));
        check(rq_sample_initial_state(d.h, e.h, p.h, s.h, r.h));
    });
    m.def("observe", [](Device& d, Environment& e, Parameters& p, State& s, Array obs, Rng& r) {   // README.md:96
        if (obs.ndim() != 2 || obs.shape(0) != e.n || obs.shape(1) != RQ_OBSERVATION_DIM)
            throw std::invalid_argument("observation must be [N_ENVIRONMENTS, OBSERVATION_DIM] float32");
        check(rq_observe(d.h, e.h, p.h, s.h, obs.mutable_data(), r.h));
    });
    m.def("step", [](Device& d, Environment& e, Parameters& p, State& s,
                     py::array_t<float, py::array::c_style | py::array::forcecast> action, State& next, Rng& r) {
        if (action.ndim() != 2 || action.shape(0) != e.n || action.shape(1) != RQ_ACTION_DIM)
            throw std::invalid_argument("action must be [N_ENVIRONMENTS, 4]");
        if (!next.h) check(rq_state_create(e.h, &next.h));
        std::vector<float> dts(e.n);
        check(rq_step(d.h, e.h, p.h, s.h, action.data(), next.h, r.h, dts.data()));           // README.md:98
        return dts;                                                                           // the per-env dt list
    });
    m.def("state_array", [](State& s, Environment& e) {                                       // .states[i].position: [:, 0:3]
        Array out({(py::ssize_t)e.n, (py::ssize_t)RQ_STATE_DIM});
        check(rq_state_get(s.h, out.mutable_data()));
        return out;
    });
}
