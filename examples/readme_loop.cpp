// examples/readme_loop.cpp — the reference's rollout loop (/root/reference/README.md:94-99) in C++17 through
// include/raptor_quad.hpp (free functions with the reference's names and argument order), first with host
// arrays crossing the boundary every call, then as one fused device rollout.
//
//   g++ -std=c++17 -O2 -Iinclude examples/readme_loop.cpp -Lraptor_amd -lraptor_quad -Wl,-rpath,$PWD/raptor_amd
//   ./a.out raptor_amd/data/raptor_policy.bin
#include <cmath>
#include <cstdio>
#include <vector>

#include "raptor_quad.hpp"

namespace rq = raptor_quad;

int main(int argc, char** argv) {
    std::vector<float> weights(RQ_POLICY_NUM_WEIGHTS);
    std::FILE* f = std::fopen(argc > 1 ? argv[1] : "raptor_amd/data/raptor_policy.bin", "rb");
    if (!f || std::fread(weights.data(), sizeof(float), weights.size(), f) != weights.size()) {
        std::fprintf(stderr, "cannot read the policy weights\n");
        return 1;
    }
    std::fclose(f);
    try {
        constexpr std::uint32_t N = 8;
        rq::Device device;
        rq::Rng rng(device);
        rq::Environment env(device, N);
        rq::initialize_rng(device, rng, 0);
        rq::initialize_environment(device, env);
        rq::Parameters params(env);
        rq::State state(env), next_state(env);
        rq::sample_initial_parameters(device, env, params, rng);
        rq::sample_initial_state(device, env, params, state, rng);
        rq::Raptor policy(device, weights.data());

        std::vector<float> observation(N * env.OBSERVATION_DIM), action(N * env.ACTION_DIM);
        policy.reset();
        for (int i = 0; i < 250; ++i) {                                             // README.md:95-99
            rq::observe(device, env, params, state, observation.data(), rng);
            policy.evaluate_step(observation.data(), N, env.OBSERVATION_DIM, action.data());
            rq::step(device, env, params, state, action.data(), next_state, rng);
            state.assign(next_state);
        }
        rq::rollout(device, env, params, state, policy, rng, 250);                  // the same loop, fused on the device
        const std::vector<float> s = state.host();
        double worst = 0;
        for (std::uint32_t e = 0; e < N; ++e) {
            const float* p = &s[e * RQ_STATE_DIM];
            worst = std::fmax(worst, std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]));
        }
        std::printf("8 quadrotors after 500 steps: max |position| = %.3f m\n", worst);

        // around the loop: record 50 steps, label them with a bank of two 22-16-16-4 MLP teachers (README.md:208-216),
        // and all-gather the episode returns through a one-rank RCCL communicator created by the library itself
        rq::Trajectory traj(env, 50);
        rq::rollout(device, env, params, state, policy, rng, 50, traj, RQ_ROLLOUT_FUSED, /*autoreset=*/true);
        const std::size_t per_teacher = 16 * 22 + 16 + 16 * 16 + 16 + 4 * 16 + 4;
        std::vector<float> tw(2 * per_teacher);
        for (std::size_t i = 0; i < tw.size(); ++i) tw[i] = 0.05f * float(int(i % 13) - 6);
        rq::TeacherBank bank(device, tw.data(), 2, 22, 16, 16, RQ_ACT_TANH, RQ_ACT_TANH);
        std::vector<std::uint32_t> teacher(N);
        for (std::uint32_t e = 0; e < N; ++e) teacher[e] = e % 2;
        std::vector<float> labels(std::size_t(traj.length()) * N * 4);
        rq::relabel_teachers(traj, bank, teacher.data(), labels.data());
        float amax = 0;
        for (float v : labels) amax = std::fmax(amax, std::fabs(v));
        rq::Communicator comm(device, 1, 0, rq::Communicator::unique_id());
        comm.allgather_returns(env);
        const std::vector<float> gathered = comm.gathered();
        std::printf("recorded %u steps, teacher labels max |a| = %.3f (tanh output), gathered %zu returns\n",
                    traj.length(), amax, gathered.size());
        if (!(amax <= 1.0f) || gathered.size() != N) return 2;
    } catch (const rq::Error& e) {
        std::fprintf(stderr, "raptor_quad error %d: %s\n", e.status, e.what());
        return 1;
    }
    return 0;
}
