// raptor_quad.hpp — header-only C++17 convenience layer over the C ABI (raptor_quad.h).
//
// rl-tools / l2f expose the environment as free functions dispatched on a device object
// (`rlt::step(device, env, parameters, state, action, next_state, rng)`, call sites in
// /root/reference/README.md:58-61,96-99).  This header gives C++ callers the same shape over
// libraptor_quad.so: RAII owners for the opaque handles and free functions with the reference's
// names and argument order.  Failures throw raptor_quad::Error (status + rq_last_error()).
// It adds no functionality of its own; everything forwards to the extern "C" entry points.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "raptor_quad.h"

namespace raptor_quad {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string& what) : std::runtime_error(what), status(s) {}
};

inline void check(int status) {
    if (status != RQ_OK) throw Error(status, std::string(rq_status_string(status)) + ": " + rq_last_error());
}

namespace detail {
template <typename T, int (*Destroy)(T*)>
struct Handle {
    T* h = nullptr;
    Handle() = default;
    Handle(const Handle&) = delete;
    Handle& operator=(const Handle&) = delete;
    Handle(Handle&& o) noexcept : h(o.h) { o.h = nullptr; }
    Handle& operator=(Handle&& o) noexcept { if (this != &o) { reset(); h = o.h; o.h = nullptr; } return *this; }
    ~Handle() { reset(); }
    void reset() { if (h) { Destroy(h); h = nullptr; } }
};
}  // namespace detail

// l2f.Device (README.md:49)
struct Device : detail::Handle<rq_device, rq_device_destroy> {
    explicit Device(int hip_device_ordinal = 0) { check(rq_device_create(hip_device_ordinal, &h)); }
    void synchronize() { check(rq_device_synchronize(h)); }
};

// vector.VectorRng (README.md:50)
struct Rng : detail::Handle<rq_rng, rq_rng_destroy> {
    explicit Rng(Device& d) { check(rq_rng_create(d.h, &h)); }
};

// vector.VectorEnvironment (README.md:51); the batch size is a runtime value
struct Environment : detail::Handle<rq_env, rq_env_destroy> {
    std::uint32_t N_ENVIRONMENTS;
    static constexpr std::uint32_t OBSERVATION_DIM = RQ_OBSERVATION_DIM;
    static constexpr std::uint32_t ACTION_DIM = RQ_ACTION_DIM;
    Environment(Device& d, std::uint32_t n_envs, std::uint64_t global_env_offset = 0) : N_ENVIRONMENTS(n_envs) {
        check(rq_env_create(d.h, n_envs, global_env_offset, &h));
    }
    rq_env_config config() const { rq_env_config c; check(rq_env_get_config(h, &c)); return c; }
    void set_config(const rq_env_config& c) { check(rq_env_set_config(h, &c)); }
    std::vector<float> returns() const { std::vector<float> v(N_ENVIRONMENTS); check(rq_env_get_returns(h, v.data(), 0)); return v; }
    std::vector<float> finished_returns() const { std::vector<float> v(N_ENVIRONMENTS); check(rq_env_get_finished_returns(h, v.data(), 0)); return v; }
};

// vector.VectorParameters / VectorState (README.md:53-56)
struct Parameters : detail::Handle<rq_params, rq_params_destroy> {
    explicit Parameters(Environment& e) { check(rq_params_create(e.h, &h)); }
};
struct State : detail::Handle<rq_state, rq_state_destroy> {
    std::uint32_t n;
    explicit State(Environment& e) : n(e.N_ENVIRONMENTS) { check(rq_state_create(e.h, &h)); }
    void assign(const State& other) { check(rq_state_assign(h, other.h)); }          // README.md:99
    std::vector<float> host() const { std::vector<float> v(std::size_t(n) * RQ_STATE_DIM); check(rq_state_get(h, v.data())); return v; }
};

// foundation_policy.Raptor (README.md:19-24)
struct Raptor : detail::Handle<rq_policy, rq_policy_destroy> {
    Raptor(Device& d, const float* weights, std::size_t n_weights = RQ_POLICY_NUM_WEIGHTS) {
        check(rq_policy_create(d.h, weights, n_weights, &h));
    }
    void reset() { check(rq_policy_reset(h)); }
    // observation [batch, obs_stride] (first 22 columns used) -> action [batch, 4]
    void evaluate_step(const float* observation, std::uint32_t batch, std::uint32_t obs_stride, float* action) {
        check(rq_policy_evaluate_step(h, nullptr, observation, batch, obs_stride, action));
    }
    void evaluate_step(Environment& env) {   // device-resident: env observation buffer -> env action buffer
        check(rq_policy_evaluate_step(h, env.h, nullptr, env.N_ENVIRONMENTS, 0, nullptr));
    }
    // observation [steps, batch, obs_stride] -> action [steps, batch, 4] (host arrays), one launch
    void evaluate_sequence(const float* observation, std::uint32_t steps, std::uint32_t batch, std::uint32_t obs_stride,
                           float* action) {
        check(rq_policy_evaluate_sequence(h, observation, steps, batch, obs_stride, action, RQ_DST_HOST));
    }
    void set_precision(rq_policy_precision p) { check(rq_policy_set_precision(h, p)); }
    // optional layers named by rl-tools (README.md:114,116), identity in the shipped checkpoint
    void set_standardize(const float* mean, const float* std) { check(rq_policy_set_standardize(h, mean, std)); }
    void set_sample_and_squash(rq_sample_and_squash_mode mode, const float* log_std_weights = nullptr,
                               const float* log_std_bias = nullptr, std::uint64_t seed = 0) {
        check(rq_policy_set_sample_and_squash(h, mode, log_std_weights, log_std_bias, seed));
    }
};

// rollout recording buffer: what a learner's data collection consumes (README.md:208)
struct Trajectory : detail::Handle<rq_trajectory, rq_trajectory_destroy> {
    Trajectory(Environment& e, std::uint32_t capacity_steps) { check(rq_trajectory_create(e.h, capacity_steps, &h)); }
    std::uint32_t length() const { std::uint32_t n = 0; check(rq_trajectory_length(h, &n, nullptr)); return n; }
    void reset() { check(rq_trajectory_reset(h)); }
};

// a bank of MLP teachers (README.md:208-216): input (in_dim <= 22) -> h1 -> h2 -> 4
struct TeacherBank : detail::Handle<rq_teacher_bank, rq_teacher_bank_destroy> {
    TeacherBank(Device& d, const float* weights, std::uint32_t n_teachers, std::uint32_t in_dim, std::uint32_t h1,
                std::uint32_t h2, rq_activation hidden = RQ_ACT_RELU, rq_activation output = RQ_ACT_IDENTITY) {
        check(rq_teacher_bank_create(d.h, weights, n_teachers, in_dim, h1, h2, hidden, output, &h));
    }
    void set_precision(rq_policy_precision p) { check(rq_teacher_bank_set_precision(h, p)); }
};

// one rank of the multi-GPU job: RCCL all-gather of episode returns issued by the library itself
struct Communicator : detail::Handle<rq_comm, rq_comm_destroy> {
    static std::vector<char> unique_id() { std::vector<char> id(RQ_COMM_ID_BYTES); check(rq_comm_unique_id(id.data(), id.size())); return id; }
    Communicator(Device& d, std::uint32_t n_ranks, std::uint32_t rank, const std::vector<char>& id) {
        check(rq_comm_create(d.h, n_ranks, rank, id.data(), id.size(), &h));
    }
    void allgather_returns(Environment& env) { check(rq_allgather_returns(env.h, h)); }      // enqueued, overlaps the next rollout
    std::vector<float> gathered() {
        std::uint32_t count = 0;
        check(rq_comm_gathered(h, nullptr, &count, nullptr));
        std::vector<float> v(count);
        check(rq_comm_gathered(h, nullptr, nullptr, v.data()));
        return v;
    }
};

// ---- the l2f vector:: free functions, reference argument order ------------------------------------
inline void initialize_rng(Device& d, Rng& rng, std::uint64_t seed) { check(rq_initialize_rng(d.h, rng.h, seed)); }
inline void initialize_environment(Device& d, Environment& env) { check(rq_initialize_environment(d.h, env.h)); }
inline void sample_initial_parameters(Device& d, Environment& env, Parameters& p, Rng& rng) {
    check(rq_sample_initial_parameters(d.h, env.h, p.h, rng.h));
}
inline void sample_initial_state(Device& d, Environment& env, Parameters& p, State& s, Rng& rng) {
    check(rq_sample_initial_state(d.h, env.h, p.h, s.h, rng.h));
}
// observation: host [N, OBSERVATION_DIM] or nullptr (keep it on the device)
inline void observe(Device& d, Environment& env, Parameters& p, State& s, float* observation, Rng& rng) {
    check(rq_observe(d.h, env.h, p.h, s.h, observation, rng.h));
}
// action: host [N, 4] or nullptr (the env's device action buffer); returns dt in seconds
inline float step(Device& d, Environment& env, Parameters& p, State& s, const float* action, State& next, Rng& rng) {
    check(rq_step(d.h, env.h, p.h, s.h, action, next.h, rng.h, nullptr));
    return env.config().dt;
}
// the loop body README.md:95-99, n_steps times, on the device
inline void rollout(Device& d, Environment& env, Parameters& p, State& s, Raptor& policy, Rng& rng,
                    std::uint32_t n_steps, rq_rollout_mode mode = RQ_ROLLOUT_FUSED, bool autoreset = false) {
    check(rq_rollout(d.h, env.h, p.h, s.h, policy.h, rng.h, n_steps, mode, autoreset ? std::uint32_t(RQ_ROLLOUT_AUTORESET) : 0u));
}
// the same, appending every transition to a trajectory buffer
inline void rollout(Device& d, Environment& env, Parameters& p, State& s, Raptor& policy, Rng& rng,
                    std::uint32_t n_steps, Trajectory& traj, rq_rollout_mode mode = RQ_ROLLOUT_FUSED, bool autoreset = false) {
    check(rq_rollout_record(d.h, env.h, p.h, s.h, policy.h, rng.h, n_steps, mode,
                            autoreset ? std::uint32_t(RQ_ROLLOUT_AUTORESET) : 0u, traj.h));
}
// teacher teacher_id[i] labels every recorded step of env i; action_out: host [length, N, 4] or nullptr
inline void relabel_teachers(Trajectory& traj, TeacherBank& bank, const std::uint32_t* teacher_id, float* action_out,
                             bool overwrite = false) {
    check(rq_trajectory_relabel_teachers(traj.h, bank.h, teacher_id, action_out, overwrite ? 1 : 0));
}

}  // namespace raptor_quad
