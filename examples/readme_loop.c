/*
 * examples/readme_loop.c — the reference's rollout loop (/root/reference/README.md:94-99) written
 * against the C ABI alone (plain C, no Python, no torch):
 *
 *   gcc -std=c11 -O2 -Iinclude examples/readme_loop.c -Lraptor_amd -lraptor_quad \
 *       -Wl,-rpath,$PWD/raptor_amd -o readme_loop && ./readme_loop raptor_amd/data/raptor_policy.bin
 */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include "raptor_quad.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != RQ_OK) {                                                          \
            fprintf(stderr, "%s -> %s: %s\n", #call, rq_status_string(rc_), rq_last_error()); \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main(int argc, char** argv) {
    enum { N = 8, STEPS = 500 };
    float weights[RQ_POLICY_NUM_WEIGHTS];
    FILE* f = fopen(argc > 1 ? argv[1] : "raptor_amd/data/raptor_policy.bin", "rb");
    if (!f || fread(weights, sizeof(float), RQ_POLICY_NUM_WEIGHTS, f) != RQ_POLICY_NUM_WEIGHTS) {
        fprintf(stderr, "cannot read the policy weights\n");
        return 1;
    }
    fclose(f);

    rq_device* device; rq_rng* rng; rq_env* env; rq_params* params; rq_state *state, *next_state; rq_policy* policy;
    CHECK(rq_device_create(0, &device));                                   /* l2f.Device()            README.md:49 */
    CHECK(rq_rng_create(device, &rng));                                    /* vector.VectorRng()               :50 */
    CHECK(rq_env_create(device, N, 0, &env));                              /* vector.VectorEnvironment()       :51 */
    CHECK(rq_initialize_rng(device, rng, 0));                              /*                                  :58 */
    CHECK(rq_initialize_environment(device, env));                         /*                                  :59 */
    CHECK(rq_params_create(env, &params));
    CHECK(rq_state_create(env, &state));
    CHECK(rq_state_create(env, &next_state));
    CHECK(rq_sample_initial_parameters(device, env, params, rng));         /*                                  :60 */
    CHECK(rq_sample_initial_state(device, env, params, state, rng));       /*                                  :61 */
    CHECK(rq_policy_create(device, weights, RQ_POLICY_NUM_WEIGHTS, &policy));  /* Raptor()                     :48 */

    static float observation[N * RQ_OBSERVATION_DIM], action[N * RQ_ACTION_DIM], dts[N], s[N * RQ_STATE_DIM];
    CHECK(rq_policy_reset(policy));                                        /* policy.reset()                   :94 */
    struct timespec t0, t1;
    enum { WARM = 100 };      /* the first calls pay for what happens once: code objects, pinned buffers, the resident executor's stream */
    for (int step = 0; step < STEPS; ++step) {
        if (step == WARM) clock_gettime(CLOCK_MONOTONIC, &t0);
        CHECK(rq_observe(device, env, params, state, observation, rng));   /*                                  :96 */
        CHECK(rq_policy_evaluate_step(policy, NULL, observation, N, RQ_OBSERVATION_DIM, action));  /* [:, :22]   :97 */
        CHECK(rq_step(device, env, params, state, action, next_state, rng, dts));                  /*            :98 */
        CHECK(rq_state_assign(state, next_state));                         /*                                  :99 */
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    CHECK(rq_state_get(state, s));
    printf("%d iterations of observe -> evaluate_step -> step -> assign with host arrays: %.1f us per iteration (the last %d)\n", STEPS,
           ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / 1e3 / (STEPS - WARM), STEPS - WARM);
    for (int i = 0; i < N; ++i)
        printf("env %d: position (%+.3f %+.3f %+.3f) after %d steps of %.0f ms\n", i, s[i * RQ_STATE_DIM],
               s[i * RQ_STATE_DIM + 1], s[i * RQ_STATE_DIM + 2], STEPS, dts[i] * 1e3f);

    rq_policy_destroy(policy); rq_state_destroy(next_state); rq_state_destroy(state); rq_params_destroy(params);
    rq_env_destroy(env); rq_rng_destroy(rng); rq_device_destroy(device);
    return 0;
}
