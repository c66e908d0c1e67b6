// rq_rollout.hpp - the fused rollout kernel (the loop body of README.md:95-99 x K in one launch) as a template over the
// actor build, shared by the two translation units that instantiate it:
//   rq_kernels.hip        the exact-fp32 actors (ActorF32 / ActorF32Lean): their instruction order is pinned by hand
//                         (sched_barrier / sched_group_barrier around the MFMA batches);
//   rq_kernels_16bit.hip  the bf16 and split-f16 actors, whose MFMAs co-execute with the vector unit and whose loop is therefore
//                         a scheduling problem of its own.  Round 4 compiled this unit with -mllvm
//                         -amdgpu-sched-strategy=max-ilp (a lone wave issues one vector instruction per 5.06 cycles but stalls to
//                         8.25 when it consumes the result of the one right in front of it, tools/lonewave.hip: 24 -> 5 such pairs
//                         per step) - and took it back: the two-waves-per-SIMD bf16 build then differed from run to run
//                         (raptor_amd/build.py SOURCE_FLAGS, test_fused_rollout_is_deterministic).  The unit stays the place
//                         for such per-build flags.
#pragma once
#include <hip/hip_ext.h>

#include <cstdlib>
#include <type_traits>

#include "rq_device_math.hpp"

namespace rq {

static constexpr int kBlock = 256;      // 4 waves; streaming kernels
static constexpr int kFusedBlock = 64;  // 1 wave per workgroup: spreads 65 536 envs as 1024 WGs over 256 CUs

// Stores of write-once streams leave as NON-TEMPORAL stores (round 5, same-box A/B, profiles/r05_ab_nt_stores.txt): the observation
// k_observe / k_step write (k_observe at 2 097 152 envs 84 -> 63 us = 0.61 -> 0.82 of 8 TB/s, at 262 144 envs 8.2 -> 6.2 us, at
// 65 536 envs inside the noise) and the trajectory recorder's (2.5 - 4 % on the recorded rollout).  NOT the next state, the policy
// state or the actions (k_step, k_actor_step / k_actor_stream, the fused kernel's epilogue: measured, nothing), and not loads
// (round 5's first experiment: non-temporal LOADS cost k_actor_stream 10 - 25 %).
template <bool NT, class T>
__device__ __forceinline__ void put(T* p, T v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
static constexpr bool kNtObs = true, kNtTraj = true;
static constexpr int kNtTrajAux = 2;           // the `nt` cache-policy bit of a raw buffer store

__device__ __forceinline__ uint32_t env_index() { return blockIdx.x * blockDim.x + threadIdx.x; }

// Field f of a field-major SoA buffer: env i of the batch is element i of this row.
template <typename T>
__device__ __forceinline__ T* field(T* base, uint32_t f, uint32_t ld) { return base + (size_t)f * ld; }

// register budget of a kernel built around an actor type: waves per SIMD it is compiled for
template <typename A> struct WavesPerSimd { static constexpr int value = 1; static constexpr bool bf16 = false; };
template <> struct WavesPerSimd<ActorF32Lean> { static constexpr int value = 2; static constexpr bool bf16 = false; };
template <> struct WavesPerSimd<ActorBF16> { static constexpr int value = 1; static constexpr bool bf16 = true; };
// the split-f16 actor: its MFMAs co-execute with the VALU like the bf16 ones; one build, the 512-register budget
template <> struct WavesPerSimd<ActorF16X2> { static constexpr int value = 1; static constexpr bool bf16 = true; };

// ------------------------------------------------------------------ fused rollout ------
// K iterations of observe -> evaluate_step -> step -> assign with the env state, the GRU
// hidden state, the per-env constants, the policy weights and the episode statistics resident
// in VGPRs; HBM is touched once before and once after the K steps.
// Control flow is wave-uniform around the MFMAs (see k_actor_step): lanes past the end of the
// batch shadow env n-1, frozen envs keep stepping a scratch copy that is never committed; only
// the rare auto-reset branch (no MFMA inside) diverges.

template <bool NOISE, bool AUTORESET, bool RECORD, bool SAS, typename ACTOR>
__global__ __launch_bounds__(kFusedBlock, WavesPerSimd<ACTOR>::value) void k_rollout_fused(Batch b, StepCfg c, NoiseCfg nc, SampleCfg sc,
                                                               uint64_t seed, uint32_t epoch0, uint32_t n_steps,
                                                               const float* __restrict__ params,
                                                               float* __restrict__ state,
                                                               float* __restrict__ hidden,
                                                               const float* __restrict__ w,
                                                               const float* __restrict__ packed, StatsPtrs st,
                                                               TrajPtrs traj, SasArgs sas,
                                                               unsigned long long* __restrict__ span) {
    // kernel-level timing (rq_device_set_rollout_timing): every wave leaves the wall-clock ticks (constant rate) at which it
    // came in and went out, and its XCD: the eight dies' counters are offset against one another by microseconds, one die's
    // are consistent - the host takes first-in / last-out per die
    // (one record per wave, no atomics: 128 waves of a die updating one word cost the launch 8 us)
    unsigned long long t_in = 0;
    if (span != nullptr) t_in = (unsigned long long)wall_clock64();
    const uint32_t i0 = env_index();
    const uint32_t wave_base = i0 & ~63u;
    const uint32_t i = i0 < b.n ? i0 : b.n - 1;
    const bool valid = i0 < b.n;
    const size_t ld = b.ld;
    const uint64_t genv = b.env_offset + i;
    // what the ahead-of-time sampler needs is asked for first: it runs while the rest of the prologue's loads are in flight
    uint32_t ep = AUTORESET ? st.episode[i] : 0u;
    float hover_rpm = 0.0f;
    if (AUTORESET) hover_rpm = field(params, RQ_P_HOVER_RPM, ld)[i];
    const EnvConsts k = make_consts([&](int f) { return field(params, f, ld)[i]; });
    QuadState y;
    f32x2 LA01, LA23;
    float f6[6], hQ[4][4];
    y.load([&](int j) { return field(state, j, ld)[i]; });
    LA01 = f32x2{field(state, (RQ_S_LAST_ACTION + 0), ld)[i], field(state, (RQ_S_LAST_ACTION + 1), ld)[i]};
    LA23 = f32x2{field(state, (RQ_S_LAST_ACTION + 2), ld)[i], field(state, (RQ_S_LAST_ACTION + 3), ld)[i]};
#pragma unroll
    for (int j = 0; j < 6; ++j) f6[j] = field(state, (RQ_S_FORCE + j), ld)[i];
    load_hidden_q(hidden, ld, wave_base, b.n, hQ);
    // the running episode's return and length ride in registers; a FINISHED episode's record goes straight to memory when
    // it ends (below): four values less to carry through the loop, five instructions less per step
    float ep_ret = st.returns[i];
    uint32_t ep_steps = st.steps[i];
    float last_r = st.last_reward[i];
    const uint8_t last_t_raw = st.last_terminated[i];
    uint8_t last_d = AUTORESET ? (uint8_t)0 : st.last_done[i];       // auto-reset: rebuilt in the epilogue
    const uint8_t frozen_raw = st.frozen[i];
    // the operand image (L2-resident after a die's first wave) is asked for AFTER the env's own fields: those come from
    // HBM / the memory-side cache and their latency is the long one
    ACTOR actor;
    actor.template load_issue<kFusedBlock / 64>(packed);
    // (kAhead, pre, pre_mask: see "sampled AHEAD" below)
    constexpr bool kAhead = AUTORESET && (WavesPerSimd<ACTOR>::value == 1 || WavesPerSimd<ACTOR>::bf16);
    constexpr int kPre = 19;
    float pre[kPre];
    uint64_t pre_mask = 0;                       // wave-uniform
    if (kAhead) {
        // Every launch STARTS with valid parked values (round 4, second half): the sampler runs here, inline, for all 64
        // lanes, while the prologue's ~90 load instructions are in flight - its inputs were asked for first, and it sits
        // between the request for the operand image and the image's move into its registers (a CALL would wait for every
        // outstanding load; so would anything placed behind actor.park()).  Before, a launch began with nothing parked and
        // the wave that met the launch's first episode end sampled there and then: 1.9 us that the other 1 023 waves of a
        // 65 536-env launch waited for at its end (tools/wave_timeline.py: the slowest wave's steps 2.4 us longer than the
        // median wave's in a 20-step launch; now 0.9, for 1.2 us more prologue in every wave: 20-step regions 77.24 ->
        // 76.66 us on one box, three alternations).
        float s0[17], la0[4], f0[6];
        sample_state(sc, seed, ep, genv, field(params, RQ_P_MASS, ld)[i], hover_rpm, field(params, RQ_P_ROTOR_POS, ld)[i],
                     field(params, (RQ_P_ROTOR_POS + 1), ld)[i], s0, la0, f0);
#pragma unroll
        for (int j = 0; j < 13; ++j) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(pre[j]) : "v"(s0[j]));
#pragma unroll
        for (int j = 0; j < 6; ++j) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(pre[13 + j]) : "v"(f0[j]));
        pre_mask = ~0ull;
    }
    actor.park();
    float h0Q[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h0Q[t][r] = actor.h0(r);
    // looked at only now (the empty asm keeps the compares from drifting up between the image's loads, where they
    // made those wait for every load before them)
    uint32_t last_t_bits = last_t_raw, frozen_bits = frozen_raw;
    asm volatile("" : "+v"(last_t_bits), "+v"(frozen_bits));
    bool last_t = last_t_bits != 0;
    const bool was_frozen = frozen_bits != 0;
    // The next episode's initial state, sampled AHEAD of the episode end (round 3).  sample_initial_state of an env depends
    // on (seed, episode counter, global env id, a few parameters) and on nothing the running episode computes, so it need
    // not wait for the end: the 19 values that are not constants (position .. angular velocity, the disturbance) are kept
    // in ACCUMULATION registers for every lane, and an env whose episode ends takes them with 19 register reads.  The
    // sampler itself (six Philox blocks, sin / cos, Box-Muller: ~1 000 instructions, and at an episode end it used to run
    // for the one or two lanes concerned while the other 62 waited - the slowest wave of a 20-step launch paid it three
    // times, tools/wave_timeline.py) runs for ALL 64 lanes at once: in the prologue of every launch (above, under the
    // prologue's loads) and again only when an ending env finds its values used up - its second end since the last
    // sampling: `pre_mask` has a bit per lane whose parked values are for its current episode counter.  Lanes that still
    // hold valid ones get the same values again (same counter, same function), so the refill is unconditional.
    // Only the builds with one wave per SIMD do this: the two-waves-per-SIMD builds have 256 registers per wave in all,
    // every one of them an architected register; asking for accumulation registers splits that budget 128 + 128 and the
    // hot loop spills (262 144 envs: 0.70 -> 0.55 of the peak).  There the second wave fills the time one spends sampling.
    // (The two-wave bf16 build has the room: 6.45 -> 5.7 us per step of 262 144 envs with it.)
    // the env index as the rare paths see it: opaque, so that the addresses they form are computed there and then instead of
    // being kept through the loop (see the epilogue)
    auto rare_index = [&]() { uint32_t r = i; asm volatile("" : "+v"(r)); return r; };
    auto refill = [&]() {                        // every lane: sample_initial_state for its episode counter ep
        const uint32_t ir = rare_index();
        const PreSample fresh = sample_state_ahead(sc, seed, ep, genv, field(params, RQ_P_MASS, ld)[ir], hover_rpm,
                                                   field(params, RQ_P_ROTOR_POS, ld)[ir], field(params, (RQ_P_ROTOR_POS + 1), ld)[ir]);
#pragma unroll
        for (int j = 0; j < kPre; ++j) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(pre[j]) : "v"(fresh[j]));
        pre_mask = ~0ull;
    };
    auto take_presampled = [&]() {               // this lane's env starts its next episode (its parked values are valid)
        float fr[kPre];
        if constexpr (kAhead) {
#pragma unroll
            for (int j = 0; j < kPre; ++j) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(fr[j]) : "a"(pre[j]));
        } else {                                 // sampled here and now, for the lanes whose episode ended
            const PreSample fresh = sample_state_ahead(sc, seed, ep, genv, field(params, RQ_P_MASS, ld)[i], hover_rpm,
                                                       field(params, RQ_P_ROTOR_POS, ld)[i],
                                                       field(params, (RQ_P_ROTOR_POS + 1), ld)[i]);
#pragma unroll
            for (int j = 0; j < kPre; ++j) fr[j] = fresh[j];
        }
        y.load([&](int j) { return j < 13 ? fr[j] : hover_rpm; });
        LA01 = f32x2{0.0f, 0.0f}; LA23 = f32x2{0.0f, 0.0f};       // sample_state: last action 0, rotors at hover
#pragma unroll
        for (int j = 0; j < 6; ++j) f6[j] = fr[13 + j];
        if (valid) {                                 // the new episode's disturbance: written now, not carried to the end
            const uint32_t ir = rare_index();
#pragma unroll
            for (int j = 0; j < 6; ++j) field(state, (RQ_S_FORCE + j), ld)[ir] = fr[13 + j];
        }
        ep += 1;
    };
    if (AUTORESET) {
        // An env left frozen by an earlier rollout WITHOUT auto-reset (its episode is over) starts its next
        // episode here, as every episode end under auto-reset does: re-sampled, policy state reset.  The
        // chained mode does the same before its first step (k_thaw_frozen).
        const uint64_t thaw = __builtin_amdgcn_ballot_w64(was_frozen);
        if (thaw != 0) {                             // (kAhead: the prologue parked the values they take)
            if (was_frozen) take_presampled();
            pre_mask &= ~thaw;
            select_hidden_q(thaw, h0Q, hQ);
        }
    }
    typename ACTOR::Carry carry;          // what the actor carries from one step into the next (ActorF32T::Carry)
    actor.prime(hQ, carry);
    Disturbance ds = make_disturbance(k, c.gravity, f6);
    bool frozen = AUTORESET ? false : was_frozen;
    uint32_t last_step = n_steps;                      // without auto-reset: the last step of this launch the env took
    // wave-uniform: no env of this wave distinguishes rotor spin-up from spin-down (see dynamics<SYM_TAU>)
    const bool sym_tau = __builtin_amdgcn_ballot_w64(k.itr != k.itf) == 0;

    // The loop exists twice, once per dynamics variant (round 3): with the wave-uniform choice inside the loop the two
    // variants met in a join that cost the state's registers a copy per step (~8 moves) plus the branch itself.
    auto rollout_loop = [&](auto sym_choice) {
    constexpr bool SYM = decltype(sym_choice)::value;
    for (uint32_t t = 0; t < n_steps; ++t) {
        const uint64_t live = AUTORESET ? ~0ull : __builtin_amdgcn_ballot_w64(!frozen);
        if (!AUTORESET && live == 0) break;   // wave-uniform exit: every env of the wave is frozen
        float o[22], a[4];
        observe_head<NOISE>(y, LA01, LA23, nc, seed, epoch0 + t, genv, o);
        float hn[4][4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) hn[tt][r] = hQ[tt][r];
        // Trajectory stores (RECORD): one coalesced 256-byte store per field per wave; buffer stores: resource = this
        // step's block of the trajectory (base moved on the SALU), scalar offset = field row, vector offset = the
        // lane's env - no per-lane 64-bit address arithmetic, no per-lane pointers kept alive across the loop; lanes
        // past the batch are sent out of range (the hardware drops out-of-range buffer stores).  The 22 observation
        // stores are handed to the actor, which places them between the MFMAs of its first GRU pass; the action
        // follows the actor, reward and done code the env step.
        const uint32_t row = (uint32_t)ld * 4u;                     // bytes per field row (ld < 2^30)
        const uint32_t lane_off = valid ? i * 4u : 0xFFFFFFFFu;
        if (RECORD) {
            const size_t tt = traj.t0 + t;
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(traj.obs + tt * 22 * ld, 0, 22u * row, 0x00020000);
            actor.template step_fused<22>(o, hn, a, carry, [&] {
#pragma unroll
                for (int j = 0; j < 22; ++j)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, o[j]), ro, lane_off, (uint32_t)j * row, kNtTrajAux);
            });
        } else {
            actor.template step_fused<0>(o, hn, a, carry, [] {});
        }
        if (SAS) sample_and_squash(sas, epoch0 + t, genv, hn, a);        // SampleAndSquash output stage (rare)
        if (RECORD) {
            const size_t tt = traj.t0 + t;
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(traj.act + tt * 4 * ld, 0, 4u * row, 0x00020000);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, a[j]), ra, lane_off, (uint32_t)j * row, kNtTrajAux);
        }
        if (AUTORESET) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) hQ[tt][r] = hn[tt][r];
        } else {
            select_hidden_q(live, hn, hQ);    // frozen envs keep their hidden state
        }
        // everything above belongs to the actor (MFMA results consumed, transposes done); the env step below
        // contains hand-placed packed instructions the compiler's hazard tracking does not see through
        // (round 4 measured the 16-bit builds without this barrier - the env step free to mix with their co-executing MFMAs:
        // no gain)
        __builtin_amdgcn_sched_barrier(0);
        QuadState yn = y;
        f32x2 A01, A23;
        bool term;
        const float r = step_inplace<SYM>(c, k, ds, yn, a, A01, A23, term);
        // the reward is wanted HERE: left to itself its arithmetic sinks below the episode-end block, which overwrites the
        // state it reads - and the state then lives twice, nine copies per step
        asm volatile("" :: "v"(r));
        bool ended = false;
        uint8_t done_code = 4;          // frozen: computed on a scratch copy, not committed
        if constexpr (AUTORESET) {
            // every env steps (auto-reset never freezes).  What an episode END needs - the finished episode's record, the
            // counters' reset, the next initial state - sits behind ONE wave-uniform test of the ballot further down
            // (round 4: as lane-wise selects and an exec-masked block it cost every step ~6 vector and ~8 scalar
            // instructions, the ballot itself was rebuilt from a 0 / 1 select); last_terminated / last_done are not
            // carried either: the epilogue reads them off the last step's termination mask and the step counter.
            y = yn;
            LA01 = A01; LA23 = A23;
            if (__builtin_expect(c.action_history_raw != 0, 0)) {      // wave-uniform (kernel argument)
                LA01 = f32x2{a[0], a[1]}; LA23 = f32x2{a[2], a[3]};
            }
            last_r = r; last_t = term;
            ep_ret += r;
            ep_steps += 1;
            // ONE compare whose result is the ballot (a termination counts as the limit reached): the ballot of an OR of
            // two lane masks is rebuilt by the compiler from a 0 / 1 select and a compare
            // (opaque, or the compiler turns the select + compare back into the OR)
            uint32_t reached = term ? 0xFFFFFFFFu : ep_steps;
            asm volatile("" : "+v"(reached));
            ended = reached >= c.episode_step_limit;
            if (RECORD) done_code = term ? 1 : (ended ? 2 : 0);
        } else if (!frozen) {           // commit
            y = yn;
            LA01 = A01; LA23 = A23;
            if (__builtin_expect(c.action_history_raw != 0, 0)) {      // wave-uniform (kernel argument)
                LA01 = f32x2{a[0], a[1]}; LA23 = f32x2{a[2], a[3]};
            }
            last_r = r; last_t = term;
            ep_ret += r;
            ep_steps += 1;
            ended = term || ep_steps >= c.episode_step_limit;
            done_code = term ? 1 : (ended ? 2 : 0);
            last_d = done_code;
            last_step = t;                             // (an env that froze earlier in the launch reports 4: see below)
            if (ended) {
                if (valid) {                           // lanes past the batch shadow env n - 1: they must not count twice
                    const uint32_t ir = rare_index();
                    st.fin_returns[ir] = ep_ret;
                    st.fin_lengths[ir] = ep_steps;
                    (void)__hip_atomic_fetch_add(&st.fin_counts[ir], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (term) (void)__hip_atomic_fetch_add(&st.fin_terminated[ir], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                ep_ret = 0.0f;
                ep_steps = 0;
                frozen = true;
            }
        }
        if (RECORD) {   // reward and done code of this transition (the observation and action went out above)
            const size_t tt = traj.t0 + t;
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(traj.rew + tt * ld, 0, row, 0x00020000);
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(traj.done + tt * ld, 0, (uint32_t)ld, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, r), rr, valid ? i * 4u : 0xFFFFFFFFu, 0, kNtTrajAux);
            __builtin_amdgcn_raw_buffer_store_b8(done_code, rd, valid ? i : 0xFFFFFFFFu, 0, kNtTrajAux);
        }
        if (AUTORESET) {   // the envs whose episode ended: record, next initial state, h <- initial_hidden_state
            const uint64_t ended_mask = __builtin_amdgcn_ballot_w64(ended);
            if (ended_mask != 0) {
                if (ended) {
                    if (valid) {                           // lanes past the batch shadow env n - 1: they must not count twice
                        const uint32_t ir = rare_index();
                        st.fin_returns[ir] = ep_ret;
                        st.fin_lengths[ir] = ep_steps;
                        (void)__hip_atomic_fetch_add(&st.fin_counts[ir], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (term) (void)__hip_atomic_fetch_add(&st.fin_terminated[ir], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    ep_ret = 0.0f;
                    ep_steps = 0;
                }
                if constexpr (kAhead) {
                    if ((ended_mask & ~pre_mask) != 0) refill();      // wave-uniform; rare (see above)
                }
                if (ended) {
                    take_presampled();
                    ds = make_disturbance(k, c.gravity, f6);
                }
                pre_mask &= ~ended_mask;
                select_hidden_q(ended_mask, h0Q, hQ);
                actor.reset_carry(ended_mask, hQ, carry);
            }
        }
    }
    };
    // (the core-clock counter beside the constant-rate one: cycles over ticks is the clock the wave's steps really ran at)
    unsigned long long t_loop = 0, t_done = 0, c_loop = 0, c_done = 0;
    if (span != nullptr) { t_loop = (unsigned long long)wall_clock64(); c_loop = (unsigned long long)__builtin_readcyclecounter(); }
    if (sym_tau) rollout_loop(std::true_type{});
    else         rollout_loop(std::false_type{});
    if (span != nullptr) { c_done = (unsigned long long)__builtin_readcyclecounter(); t_done = (unsigned long long)wall_clock64(); }

    const bool commit = AUTORESET || !was_frozen;     // under auto-reset a frozen env was thawed above
    // an env whose episode ended BEFORE the launch's last step sat out the rest of it: its last transition of this rollout is
    // "not stepped" (4), as the chain of k_step launches reports it (found by the random-settings test, round 3)
    if (!AUTORESET && n_steps > 0 && last_step + 1 != n_steps) last_d = 4;
    // auto-reset: the last transition's done code from what the loop left behind - terminated, or ended by the step limit
    // (an episode end zeroes the step counter; n_steps > 0 in every launch), or neither
    if (AUTORESET) last_d = last_t ? 1 : (ep_steps == 0 ? 2 : 0);
    // The stores go to the addresses the prologue loaded from, and left alone the compiler keeps those ~55 64-bit
    // addresses alive through the whole loop - parked in accumulation registers: ~110 moves in, ~110 out, per launch.
    // An env index it cannot see through makes it compute them again here (55 adds).
    uint32_t ie = i;
    asm volatile("" : "+v"(ie));
    size_t lde = ld;                                  // likewise the field rows' scalar bases (they were kept in VGPR lanes)
    asm volatile("" : "+s"(lde));
    if (valid && commit) {
        y.store([&](int j, float v) { field(state, j, lde)[ie] = v; });
        field(state, (RQ_S_LAST_ACTION + 0), lde)[ie] = LA01[0]; field(state, (RQ_S_LAST_ACTION + 1), lde)[ie] = LA01[1];
        field(state, (RQ_S_LAST_ACTION + 2), lde)[ie] = LA23[0]; field(state, (RQ_S_LAST_ACTION + 3), lde)[ie] = LA23[1];
        st.returns[ie] = ep_ret;
        st.steps[ie] = ep_steps;
        st.last_reward[ie] = last_r;
        st.last_terminated[ie] = last_t ? 1 : 0;
        st.last_done[ie] = last_d;
        if (AUTORESET) {
            st.episode[ie] = ep;
            if (was_frozen) st.frozen[ie] = 0;
        }
        if (frozen) st.frozen[ie] = 1;
    }
    if (valid && !commit && n_steps > 0) st.last_done[ie] = 4;   // not stepped by this rollout (as k_step reports it)
    store_hidden_q(hidden, lde, wave_base, __builtin_amdgcn_ballot_w64(valid && commit), hQ);
    if (span != nullptr) {
        __builtin_amdgcn_s_waitcnt(0);                // the wave's stores have left
        if (threadIdx.x == 0) {
            const unsigned long long xcd = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;        // HW_REG_XCC_ID[3:0]
            unsigned long long* rec = span + 4 * (size_t)blockIdx.x;
            rec[0] = t_in;
            rec[1] = ((unsigned long long)wall_clock64() & 0x0FFFFFFFFFFFFFFFull) | (xcd << 60);
            rec[2] = t_loop;          // prologue issued (its loads may still be in flight), first step about to start
            rec[3] = t_done;          // last step done, the epilogue's stores not yet issued
            span[4 * (size_t)gridDim.x + blockIdx.x] = c_done - c_loop;      // core-clock cycles between rec[2] and rec[3]
        }
    }
}

// ------------------------------------------------------------------ launch -------------
// span != nullptr: every wave leaves (in, out | xcd << 60, loop begin, loop end) wall-clock ticks at span[4 * workgroup]
// and, behind all of those, the core-clock cycles its steps took at span[4 * workgroups + workgroup]
// (rq_device_last_rollout_ms).  Round 2 took the kernel's begin / end from hipExtLaunchKernel events; calibrated under
// rocprofv3 in one process, an event-carrying launch itself runs ~4 us longer than a plain one and the events read
// ~4 us more on top.
struct FusedArgs {
    Batch b; StepCfg c; NoiseCfg nc; SampleCfg sc; uint64_t seed; uint32_t epoch0, n_steps;
    const float* params; float* state; float* hidden; const float* weights; const float* packed;
    StatsPtrs st; TrajPtrs traj; SasArgs sas; unsigned long long* span;
};

// the 16-bit actors' instantiations (rq_kernels_16bit.hip)
hipError_t launch_rollout_fused_16bit(hipStream_t s, const FusedArgs& a, bool noise, bool ar, int precision);


template <bool NZ, bool AR, bool RC, bool SAS, typename ACTOR>
inline void launch_fused_instance(hipStream_t s, const FusedArgs& a) {
    const unsigned g = (a.b.n + kFusedBlock - 1) / kFusedBlock;
    hipLaunchKernelGGL((k_rollout_fused<NZ, AR, RC, SAS, ACTOR>), dim3(g), dim3(kFusedBlock), 0, s,
                       a.b, a.c, a.nc, a.sc, a.seed, a.epoch0, a.n_steps, a.params, a.state, a.hidden, a.weights, a.packed,
                       a.st, a.traj, a.sas, a.span);
}

// the instantiation for (noise, auto-reset, recording) of one actor build; SAS = with the SampleAndSquash output stage
template <bool SAS, typename ACTOR>
inline void launch_fused_actor(hipStream_t s, const FusedArgs& a, bool noise, bool ar) {
    const bool rec = a.traj.obs != nullptr;
#define RQ_FUSED_RC(NZ, AR) do { if (rec) launch_fused_instance<NZ, AR, true, SAS, ACTOR>(s, a); \
                                 else     launch_fused_instance<NZ, AR, false, SAS, ACTOR>(s, a); } while (0)
    if (noise) { if (ar) RQ_FUSED_RC(true, true); else RQ_FUSED_RC(true, false); }
    else       { if (ar) RQ_FUSED_RC(false, true); else RQ_FUSED_RC(false, false); }
#undef RQ_FUSED_RC
}

}  // namespace rq
