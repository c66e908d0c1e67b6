/* sanitize_driver.c - test infrastructure: drives every entry point of raptor_oracle.c with exactly sized heap
 * buffers so that a build with -fsanitize=address,undefined (tests/test_oracle_env.py::test_oracle_under_sanitizers)
 * reports any out-of-bounds access, use of uninitialised stack arrays' neighbours, signed overflow or misaligned
 * access in the restatement the HIP kernels are checked against.  Prints one line of checksums; exit code 0. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/raptor_quad.h"

#define ORC_EXPORT
void orc_default_config(rq_env_config* c);
void orc_sample_initial_parameters(const rq_env_config* c, uint64_t seed, uint32_t epoch, uint64_t env_offset, uint32_t n, float* params);
void orc_sample_initial_state(const rq_env_config* c, uint64_t seed, uint32_t* episode, uint64_t env_offset, uint32_t n, const float* params, float* state);
void orc_observe(const rq_env_config* c, uint64_t seed, uint32_t epoch, uint64_t env_offset, uint32_t n, const float* params, const float* state, float* obs);
void orc_step(const rq_env_config* c, uint32_t n, const float* params, const float* state, const float* action, float* next_state, float* reward, uint8_t* terminated);
void orc_actor_sequence(const float* w, const float* in, float* out, uint32_t T, uint32_t B);
void orc_actor_batch_step(const float* w, const float* obs, uint32_t obs_stride, float* h, float* act, uint32_t n);
void orc_rollout_record(const rq_env_config* c, const float* w, uint64_t seed, uint32_t epoch0, uint64_t env_offset, uint32_t n,
                        const float* params, float* state, float* hidden, uint32_t K, uint32_t flags, float* returns,
                        uint32_t* steps, float* fin_returns, uint32_t* fin_lengths, uint32_t* fin_counts,
                        uint32_t* fin_terminated, uint8_t* frozen, uint32_t* episode, float* last_reward,
                        uint8_t* last_terminated, int nthreads, float* traj_obs, float* traj_act, float* traj_rew,
                        uint8_t* traj_done);
void orc_teacher_relabel(const float* w, uint32_t in, uint32_t h1, uint32_t h2, int act, int out_act, const float* obs,
                         const uint32_t* teacher_id, uint32_t T, uint32_t n, float* out, int nthreads);

static void* buf(size_t bytes) {          /* exactly sized: one byte past the end is a sanitizer report */
    void* p = calloc(bytes ? bytes : 1, 1);
    if (!p) { fprintf(stderr, "out of memory\n"); exit(2); }
    return p;
}
static double sum(const float* x, size_t n) { double s = 0; for (size_t i = 0; i < n; ++i) s += x[i]; return s; }

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s raptor_policy.bin\n", argv[0]); return 2; }
    float* w = buf(2084 * sizeof(float));
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(w, sizeof(float), 2084, f) != 2084) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(f);
    const uint32_t n = 37, K = 23, T = 5;             /* odd sizes: no accidental alignment */
    rq_env_config c;
    orc_default_config(&c);
    c.episode_step_limit = 9;                          /* episode ends and auto-resets inside the rollout */
    c.noise_position = 0.01f; c.noise_angular_velocity = 0.02f;
    c.disturbance_force_std = 0.05f; c.disturbance_torque_std = 0.01f;
    float* params = buf((size_t)n * RQ_PARAM_DIM * sizeof(float));
    float* state = buf((size_t)n * RQ_STATE_DIM * sizeof(float));
    float* next = buf((size_t)n * RQ_STATE_DIM * sizeof(float));
    float* obs = buf((size_t)n * RQ_OBSERVATION_DIM * sizeof(float));
    float* act = buf((size_t)n * 4 * sizeof(float));
    float* hidden = buf((size_t)n * 16 * sizeof(float));
    float* rew = buf(n * sizeof(float));
    uint8_t* term = buf(n);
    uint32_t* episode = buf(n * sizeof(uint32_t));
    orc_sample_initial_parameters(&c, 7, 0, 123456789012ull, n, params);
    orc_sample_initial_state(&c, 7, episode, 123456789012ull, n, params, state);
    orc_observe(&c, 7, 3, 123456789012ull, n, params, state, obs);
    orc_actor_batch_step(w, obs, RQ_OBSERVATION_DIM, hidden, act, n);
    orc_step(&c, n, params, state, act, next, rew, term);
    double chk = sum(next, (size_t)n * RQ_STATE_DIM) + sum(rew, n);
    /* closed loop with auto-reset, recorded */
    float *returns = buf(n * 4), *fin_ret = buf(n * 4), *last_r = buf(n * 4);
    uint32_t *steps = buf(n * 4), *fin_len = buf(n * 4), *fin_cnt = buf(n * 4), *fin_term = buf(n * 4);
    uint8_t *frozen = buf(n), *last_t = buf(n);
    float* t_obs = buf((size_t)K * n * 22 * 4); float* t_act = buf((size_t)K * n * 4 * 4); float* t_rew = buf((size_t)K * n * 4);
    uint8_t* t_done = buf((size_t)K * n);
    for (int flags = 0; flags < 2; ++flags) {
        memset(hidden, 0, (size_t)n * 16 * sizeof(float));
        orc_rollout_record(&c, w, 7, 0, 123456789012ull, n, params, state, hidden, K, (uint32_t)flags, returns, steps, fin_ret,
                           fin_len, fin_cnt, fin_term, frozen, episode, last_r, last_t, 2, t_obs, t_act, t_rew, t_done);
        chk += sum(t_act, (size_t)K * n * 4) + sum(t_rew, (size_t)K * n);
    }
    orc_rollout_record(&c, w, 7, 0, 0, n, params, state, hidden, 3, 1, returns, steps, fin_ret, fin_len, fin_cnt, fin_term,
                       frozen, episode, last_r, last_t, 1, NULL, NULL, NULL, NULL);        /* no trajectory */
    /* the sequence form on the first two recorded envs' observations */
    float* seq_in = buf((size_t)T * 2 * 22 * 4); float* seq_out = buf((size_t)T * 2 * 4 * 4);
    for (uint32_t t = 0; t < T; ++t) memcpy(seq_in + (size_t)t * 2 * 22, t_obs + (size_t)t * n * 22, 2 * 22 * sizeof(float));
    orc_actor_sequence(w, seq_in, seq_out, T, 2);
    chk += sum(seq_out, (size_t)T * 2 * 4);
    /* teachers: 3 MLPs 22-16-64-4 (tanh, tanh) and input width 13 */
    for (uint32_t in = 13; in <= 22; in += 9) {
        const uint32_t h1 = 16, h2 = 64, nt = 3;
        const size_t per = (size_t)h1 * in + h1 + (size_t)h2 * h1 + h2 + 4 * (size_t)h2 + 4;
        float* tw = buf(per * nt * sizeof(float));
        for (size_t i = 0; i < per * nt; ++i) tw[i] = 0.01f * (float)((int)(i * 2654435761u % 201u) - 100);
        uint32_t* ids = buf(n * sizeof(uint32_t));
        for (uint32_t i = 0; i < n; ++i) ids[i] = i % nt;
        float* lab = buf((size_t)K * n * 4 * sizeof(float));
        orc_teacher_relabel(tw, in, h1, h2, 2, 2, t_obs, ids, K, n, lab, 2);
        chk += sum(lab, (size_t)K * n * 4);
        free(tw); free(ids); free(lab);
    }
    printf("oracle under sanitizers: checksum %.6f\n", chk);
    free(w); free(params); free(state); free(next); free(obs); free(act); free(hidden); free(rew); free(term); free(episode);
    free(returns); free(fin_ret); free(last_r); free(steps); free(fin_len); free(fin_cnt); free(fin_term); free(frozen); free(last_t);
    free(t_obs); free(t_act); free(t_rew); free(t_done); free(seq_in); free(seq_out);
    return 0;
}
