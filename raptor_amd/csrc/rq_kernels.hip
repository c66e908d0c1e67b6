// rq_kernels.hip — CDNA4 (gfx950) kernels of the quadrotor rollout path.
//
// Mapping: one wavefront lane = one environment; env i of a batch reads field f at
// base[f*ld + i], so every vector memory instruction of a wave touches 256 contiguous bytes.
// No inter-block data reuse exists (envs are independent), so block->XCD placement is
// irrelevant for these kernels; the grid is simply ceil(n / block).
//
//   kernel               replaces (reference call site)                          bound
//   k_sample_params      vector.sample_initial_parameters   README.md:60         HBM (write 104 B/env)
//   k_sample_state       vector.sample_initial_state        README.md:61         HBM
//   k_observe            vector.observe                     README.md:96         HBM (read 92 B, write 104 B /env)
//   k_actor_step         Raptor.evaluate_step               README.md:97         fp32 FMA rate (232 B/env, 3.9 kFLOP on MFMA)
//   k_step               vector.step + state.assign         README.md:98-99      HBM (297 B/env)
//   k_rollout_fused      the loop body README.md:95-99 x K                       fp32 FMA rate (MFMA + VALU share it;
//                                                                                state, hidden, weights in registers)
//   k_record             chained-mode trajectory append                          HBM
//   k_soa_to_rows /      the NumPy-array side of a call at >= 1024 envs          HBM (+ PCIe copy)
//   k_rows_to_soa        (row-major [n][dim] <-> field-major [dim][ld], LDS tile)
// Below 1024 envs k_observe / k_actor_step / k_step exchange host rows through a pinned mailbox (Mailbox).
#include <hip/hip_ext.h>

#include <cstdlib>
#include <type_traits>

#include "rq_rollout.hpp"      // k_rollout_fused and what it shares with the kernels below (field(), block sizes, ...)

namespace rq {

// ---- launch or append a graph node (rq_kernels.hpp GraphSink) -----------------------------------------------------------------
static thread_local GraphSink* tl_graph_sink = nullptr;
void set_graph_sink(GraphSink* sink) { tl_graph_sink = sink; }

template <typename T> struct arg_of { using type = T; };
// the kernel's own parameter types decide how each argument is converted and copied (the node stores a copy of every argument)
template <typename... KArgs>
static hipError_t graph_add(GraphSink* g, void (*kernel)(KArgs...), dim3 grid, dim3 block, typename arg_of<KArgs>::type... args) {
    if (g->status != hipSuccess) return g->status;
    void* ptrs[sizeof...(KArgs)] = {const_cast<void*>(static_cast<const void*>(&args))...};
    hipKernelNodeParams p{};
    p.func = reinterpret_cast<void*>(kernel);
    p.gridDim = grid; p.blockDim = block; p.sharedMemBytes = 0; p.kernelParams = ptrs; p.extra = nullptr;
    hipGraphNode_t node = nullptr;
    g->status = hipGraphAddKernelNode(&node, g->graph, g->last ? &g->last : nullptr, g->last ? 1 : 0, &p);
    if (g->status == hipSuccess) { g->last = node; ++g->nodes; }
    return g->status;
}
// KERNEL<<<GRID, BLOCK, 0, STREAM>>>(args) - or, under a GraphSink, the same launch as a node behind the previous one
#define RQ_KLAUNCH(KERNEL, GRID, BLOCK, STREAM, ...)                                                         \
    do {                                                                                                     \
        if (tl_graph_sink) (void)graph_add(tl_graph_sink, KERNEL, dim3(GRID), dim3(BLOCK), __VA_ARGS__);     \
        else KERNEL<<<(GRID), (BLOCK), 0, (STREAM)>>>(__VA_ARGS__);                                          \
    } while (0)
#define RQ_KLAUNCH_STATUS() (tl_graph_sink ? tl_graph_sink->status : hipGetLastError())

// Publish completion of this launch in the host mailbox: every workgroup makes its stores visible at system
// scope and counts itself; the last one resets the counter and writes seq to the pinned flag.
__device__ __forceinline__ void mailbox_signal(const Mailbox& mb) {
    if (mb.flag == nullptr) return;          // wave-uniform (kernel argument)
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (gridDim.x == 1) {                    // the only workgroup: nothing to count (small batches: one atomic less)
            __hip_atomic_store(mb.flag, mb.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        const uint32_t done = __hip_atomic_fetch_add(mb.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            __hip_atomic_store(mb.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mb.flag, mb.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ------------------------------------------------------------------ sampling -----------
__global__ __launch_bounds__(kBlock) void k_sample_params(Batch b, SampleCfg c, uint64_t seed, uint32_t epoch,
                                                          float* __restrict__ params) {
    const uint32_t i = env_index();
    if (i >= b.n) return;
    float p[RQ_PARAM_DIM];
    sample_params(c, seed, epoch, b.env_offset + i, p);
#pragma unroll
    for (int f = 0; f < RQ_PARAM_DIM; ++f) field(params, f, b.ld)[i] = p[f];
}

__global__ __launch_bounds__(kBlock) void k_sample_state(Batch b, SampleCfg c, uint64_t seed,
                                                         const float* __restrict__ params, float* __restrict__ state,
                                                         StatsPtrs st) {
    const uint32_t i = env_index();
    if (i >= b.n) return;
    const uint32_t ep = st.episode[i];
    float s[17], la[4], f[6];
    sample_state(c, seed, ep, b.env_offset + i, field(params, RQ_P_MASS, b.ld)[i],
                 field(params, RQ_P_HOVER_RPM, b.ld)[i], field(params, RQ_P_ROTOR_POS, b.ld)[i],
                 field(params, (RQ_P_ROTOR_POS + 1), b.ld)[i], s, la, f);
#pragma unroll
    for (int k = 0; k < 17; ++k) field(state, k, b.ld)[i] = s[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) field(state, (RQ_S_LAST_ACTION + k), b.ld)[i] = la[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) field(state, (RQ_S_FORCE + k), b.ld)[i] = f[k];
    st.episode[i] = ep + 1;
    st.frozen[i] = 0;
    st.returns[i] = 0.0f;      // a new episode begins: no return / step count carried over from an abandoned one
    st.steps[i] = 0;
}

// ------------------------------------------------------------------ observe ------------
template <bool NOISE>
__global__ __launch_bounds__(kBlock) void k_observe(Batch b, NoiseCfg nc, uint64_t seed, uint32_t epoch_offset,
                                                    const uint32_t* __restrict__ epoch_base,
                                                    const float* __restrict__ params,
                                                    const float* __restrict__ state, float* __restrict__ obs,
                                                    Mailbox mb) {
    const uint32_t i = env_index();
    if (i < b.n) {
        // inside a replayed hipGraph the noise epoch cannot be a baked-in argument: it is read from a
        // device counter the graph itself advances (k_advance_u32); eager launches pass nullptr
        const uint32_t epoch = epoch_offset + (epoch_base != nullptr ? *epoch_base : 0u);
        QuadState y;
        y.load([&](int k) { return field(state, k, b.ld)[i]; });
        const f32x2 LA01 = {field(state, (RQ_S_LAST_ACTION + 0), b.ld)[i], field(state, (RQ_S_LAST_ACTION + 1), b.ld)[i]};
        const f32x2 LA23 = {field(state, (RQ_S_LAST_ACTION + 2), b.ld)[i], field(state, (RQ_S_LAST_ACTION + 3), b.ld)[i]};
        const float rmin = field(params, RQ_P_RPM_MIN, b.ld)[i];
        const float rmax = field(params, RQ_P_RPM_MAX, b.ld)[i];
        float head[22], o[RQ_OBSERVATION_DIM];
        observe_head<NOISE>(y, LA01, LA23, nc, seed, epoch, b.env_offset + i, head);
#pragma unroll
        for (int k = 0; k < 22; ++k) o[k] = head[k];
        // privileged tail: normalised rotor speeds
        const float inv = 2.0f / (rmax - rmin);
        o[22] = fmaf(y.R01[0] - rmin, inv, -1.0f); o[23] = fmaf(y.R01[1] - rmin, inv, -1.0f);
        o[24] = fmaf(y.R23[0] - rmin, inv, -1.0f); o[25] = fmaf(y.R23[1] - rmin, inv, -1.0f);
#pragma unroll
        for (int k = 0; k < RQ_OBSERVATION_DIM; ++k) put<kNtObs>(&field(obs, k, b.ld)[i], o[k]);
        if (mb.rows_out != nullptr) {            // wave-uniform (kernel argument)
#pragma unroll
            for (int k = 0; k < RQ_OBSERVATION_DIM; ++k) mb.rows_out[(size_t)i * RQ_OBSERVATION_DIM + k] = o[k];
        }
    }
    mailbox_signal(mb);
}

// ------------------------------------------------------------------ actor --------------
// MFMA ignores the EXEC mask and mixes the lanes of a wave, so every MFMA must execute in
// wave-uniform control flow with all 64 lanes holding valid data: lanes past the end of the
// batch (and frozen envs) compute on a clamped index and only their STORES are predicated —
// no lane leaves early.
// One 64-env group per wave (every batch up to 262 144 envs, and the host-row mailbox path of the small ones); the large
// batches take k_actor_stream below.
template <typename ACTOR>
__global__ __launch_bounds__(kBlock, 2) void k_actor_step(uint32_t n, const float* __restrict__ packed,
                                                       const float* __restrict__ obs, uint32_t ld_obs,
                                                       const float* hidden_in, float* hidden,   // the same buffer unless speculative
                                                       uint32_t ld_h, float* __restrict__ act, uint32_t ld_act,
                                                       const uint8_t* __restrict__ frozen, SasArgs sas,
                                                       Mailbox mb) {
    ACTOR actor;
    actor.template load<kBlock / 64>(packed);     // 18 KB of operand image per wave
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_base = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 64;
    if (wave_base >= n) { mailbox_signal(mb); return; }          // wave-uniform
    const uint32_t i0 = wave_base + lane;
    const uint32_t i = i0 < n ? i0 : n - 1;
    float x[22], hQ[4][4], a[4];
    if (mb.rows_in != nullptr) {         // wave-uniform (kernel argument)
#pragma unroll
        for (int k = 0; k < 22; ++k) x[k] = mb.rows_in[(size_t)i * mb.in_stride + k];
    } else {
#pragma unroll
        for (int k = 0; k < 22; ++k) x[k] = field(obs, k, ld_obs)[i];
    }
    load_hidden_q(hidden_in, ld_h, wave_base, n, hQ);      // == hidden unless the new state goes elsewhere (speculation)
    const uint32_t fz = frozen != nullptr ? (uint32_t)frozen[i] : 0u;
    const bool commit = (i0 < n) && fz == 0;
    const uint64_t commit_mask = __builtin_amdgcn_ballot_w64(commit);
    actor.step(x, hQ, a);
    if (sas.mode)                        // wave-uniform (kernel argument)
        sample_and_squash(sas, sas.epoch + (sas.epoch_base != nullptr ? *sas.epoch_base : 0u), sas.env_offset + i, hQ, a);
    store_hidden_q(hidden, ld_h, wave_base, commit_mask, hQ);
    if (commit) {
#pragma unroll
        for (int k = 0; k < 4; ++k) field(act, k, ld_act)[i] = a[k];
        if (mb.rows_out != nullptr) {
#pragma unroll
            for (int k = 0; k < 4; ++k) mb.rows_out[(size_t)i * 4 + k] = a[k];
        }
    }
    mailbox_signal(mb);
}

// The streaming form for the large batches (from 262 144 envs): a wave works through groups_per_wave consecutive 64-env groups,
// the next group's inputs in flight while this one is on the matrix cores - the 18 KB operand image per wave is amortised and
// memory and matrix phases overlap (the waves of a launch move in lock-step, so without it they alternate).  Round 4 took the
// per-access overhead out of the loop.  Written with plain pointers it spent, per group, 91 64-bit address additions (one per load
// and store: field row base + lane), 53 register copies (the prefetched group moved into the working registers) and five
// exec-masked store blocks: ~0.5 us of a lone wave's issue slots per 2.5 us group.  Here every access is a buffer
// instruction - resource = the buffer, scalar offset = the field row, vector offset = the lane's env (one shift per group) -
// stores of lanes that must not commit are sent out of range (the hardware drops them), and the loop is unrolled by two over
// two register sets that swap roles.  Same loads, same arithmetic (ACTOR::step), same stores: bit-identical results.
template <typename ACTOR>
__global__ __launch_bounds__(kBlock, 2) void k_actor_stream(uint32_t n, uint32_t groups_per_wave, const float* __restrict__ packed,
                                                            const float* __restrict__ obs, uint32_t ld_obs,
                                                            const float* hidden_in, float* hidden, uint32_t ld_h,
                                                            float* __restrict__ act, uint32_t ld_act,
                                                            const uint8_t* __restrict__ frozen, SasArgs sas) {
    ACTOR actor;
    actor.template load<kBlock / 64>(packed);
    const uint32_t lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t first = wave * groups_per_wave * 64;
    if (first >= n) return;                                  // wave-uniform
    const uint32_t n_groups = min(groups_per_wave, (n - first + 63u) / 64u);
    const uint32_t row_o = ld_obs * 4u, row_h = ld_h * 4u, row_a = ld_act * 4u;          // bytes per field row
    const __amdgpu_buffer_rsrc_t r_obs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(obs), 0, 22u * row_o, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_hin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hidden_in), 0, 16u * row_h, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_hout = __builtin_amdgcn_make_buffer_rsrc(hidden, 0, 16u * row_h, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_act = __builtin_amdgcn_make_buffer_rsrc(act, 0, 4u * row_a, 0x00020000);
    const uint32_t q_rows = q * 4u * row_h;                  // this lane group's first hidden row (Q layout: rows 4q .. 4q + 3)
    struct Group { float x[22]; float hQ[4][4]; uint32_t fz; };
    auto load_group = [&](uint32_t wave_base, Group& G) {
        const uint32_t i0 = wave_base + lane;
        const uint32_t i = i0 < n ? i0 : n - 1;              // lanes past the batch shadow env n - 1 (finite inputs for the MFMAs)
#pragma unroll
        for (int k = 0; k < 22; ++k)
            G.x[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_obs, i * 4u, (uint32_t)k * row_o, 0));
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            uint32_t e = wave_base + 16 * t + j;
            e = e < n ? e : n - 1;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                G.hQ[t][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_hin, e * 4u + q_rows, (uint32_t)r * row_h, 0));
        }
        G.fz = frozen != nullptr ? (uint32_t)frozen[i] : 0u;
    };
    auto step_and_store = [&](uint32_t wave_base, Group& G) {
        const uint32_t i0 = wave_base + lane;
        const uint32_t i = i0 < n ? i0 : n - 1;
        const bool commit = (i0 < n) && G.fz == 0;
        const uint64_t commit_mask = __builtin_amdgcn_ballot_w64(commit);
        float a[4];
        actor.step(G.x, G.hQ, a);
        if (sas.mode)                        // wave-uniform (kernel argument)
            sample_and_squash(sas, sas.epoch + (sas.epoch_base != nullptr ? *sas.epoch_base : 0u), sas.env_offset + i, G.hQ, a);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool ok = (commit_mask >> (16 * t + j)) & 1ull;
            const uint32_t vo = ok ? (wave_base + 16 * t + j) * 4u + q_rows : 0xFFFFFFFFu;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, G.hQ[t][r]), r_hout, vo, (uint32_t)r * row_h, 0);
        }
        const uint32_t va = commit ? i * 4u : 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, a[k]), r_act, va, (uint32_t)k * row_a, 0);
    };
    // Software pipeline over two register sets: while one group is on the matrix cores the other's loads are in flight.
    // Vector-memory operations complete in order and the compiler's wait for a group's inputs counts what was issued
    // behind them on EVERY path: the prefetch is therefore unconditional (past the wave's last group it re-reads that group).
    Group A, B;
    load_group(first, A);
#pragma unroll 1
    for (uint32_t g = 0; g < n_groups; g += 2) {
        const uint32_t base = first + g * 64;
        const bool has_b = g + 1 < n_groups;                 // wave-uniform
        load_group(has_b ? base + 64 : base, B);
        step_and_store(base, A);
        if (!has_b) break;
        load_group(g + 2 < n_groups ? base + 128 : base + 64, A);
        step_and_store(base + 64, B);
    }
}

// Raptor evaluated over a whole observation SEQUENCE in one launch (rl-tools evaluates [seq, batch, feature]
// tensors: the known-answer example of the checkpoint is one, checkpoint.h:197-215): obs [T][n][stride]
// row-major -> act [T][n][4] row-major, the tensors' own layout, no field-major detour.  A wave owns 64 batch
// elements for all T steps: operand image and GRU state stay in registers, each lane streams its own row
// (11 x 8-byte loads when the stride is even) and writes its action as one 16-byte store; the next step's
// row is in flight while the current one is on the matrix cores.  Bound: f32 MFMA rate (3.9 kFLOP per 104 B).
template <typename ACTOR>
__global__ __launch_bounds__(kFusedBlock, WavesPerSimd<ACTOR>::value) void k_actor_sequence(
        uint32_t n, uint32_t steps, const float* __restrict__ packed, const float* __restrict__ obs, uint32_t stride,
        float* __restrict__ hidden, uint32_t ld_h, float* __restrict__ act, uint32_t squash) {
    ACTOR actor;
    actor.template load<kFusedBlock / 64>(packed);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_base = blockIdx.x * kFusedBlock;
    const uint32_t i0 = wave_base + lane;
    const uint32_t i = i0 < n ? i0 : n - 1;
    const bool valid = i0 < n;
    float hQ[4][4];
    load_hidden_q(hidden, ld_h, wave_base, n, hQ);
    const bool wide = (stride & 1u) == 0;                 // wave-uniform: rows are 8-byte aligned
    auto load_row = [&](uint32_t t, float (&x)[22]) {
        const float* row = obs + ((size_t)t * n + i) * stride;
        if (wide) {
            const float2* r2 = reinterpret_cast<const float2*>(row);
#pragma unroll
            for (int k = 0; k < 11; ++k) { const float2 v = r2[k]; x[2 * k] = v.x; x[2 * k + 1] = v.y; }
        } else {
#pragma unroll
            for (int k = 0; k < 22; ++k) x[k] = row[k];
        }
    };
    // rows in flight ahead of the matrix work: one step for the f32 actor (a step is ~3 us of MFMA, longer than a
    // cold row's latency; a second row in flight measured 3 % slower), two for the bf16 actor (1.5 us steps: +12 %)
    constexpr bool kTwoAhead = WavesPerSimd<ACTOR>::bf16;
    float x[22], x1[22];
    load_row(0, x);
    if (kTwoAhead) load_row(steps > 1 ? 1 : 0, x1);
    typename ACTOR::Carry carry;          // the f32 actor pipelines across the step boundary (ActorF32T::Carry)
    actor.prime(hQ, carry);
    for (uint32_t t = 0; t < steps; ++t) {
        float xn[22];
        const bool more = t + (kTwoAhead ? 2 : 1) < steps;                  // wave-uniform
        if (more) load_row(t + (kTwoAhead ? 2 : 1), xn);
        float a[4];
        actor.template step_fused<0>(x, hQ, a, carry, [] {});
        if (squash) squash_action(a);
        if (valid) *reinterpret_cast<float4*>(act + ((size_t)t * n + i0) * 4) = make_float4(a[0], a[1], a[2], a[3]);
        if (kTwoAhead) {
#pragma unroll
            for (int k = 0; k < 22; ++k) { x[k] = x1[k]; x1[k] = more ? xn[k] : x1[k]; }
        } else if (more) {
#pragma unroll
            for (int k = 0; k < 22; ++k) x[k] = xn[k];
        }
    }
    store_hidden_q(hidden, ld_h, wave_base, __builtin_amdgcn_ballot_w64(valid), hQ);
}

// Relabel a recorded trajectory: evaluate `packed` - any policy of the
// supported topology, e.g. a teacher or a newer student - on the observations a rollout stored, in the
// trajectory's own field-major layout obs [T][22][ld] -> act [T][4][ld], following the recorded episode
// structure: after a step whose done code is 1 or 2 (episode ended) the GRU state returns to the initial
// hidden state, a step with code 4 (env frozen, not stepped) does not advance it.  With the policy that
// recorded the trajectory this reproduces the recorded actions bit for bit.
template <typename ACTOR>
__global__ __launch_bounds__(kFusedBlock, WavesPerSimd<ACTOR>::value) void k_actor_relabel(
        uint32_t n, uint32_t ld, uint32_t steps, const float* __restrict__ packed, const float* __restrict__ obs,
        const uint8_t* __restrict__ done, float* __restrict__ hidden, uint32_t ld_h, float* __restrict__ act,
        uint32_t squash) {
    ACTOR actor;
    actor.template load<kFusedBlock / 64>(packed);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_base = blockIdx.x * kFusedBlock;
    const uint32_t i0 = wave_base + lane;
    const uint32_t i = i0 < n ? i0 : n - 1;
    const bool valid = i0 < n;
    float hQ[4][4], h0Q[4][4];
    load_hidden_q(hidden, ld_h, wave_base, n, hQ);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h0Q[t][r] = actor.h0(r);
    typename ACTOR::Carry carry;          // see k_actor_sequence
    actor.prime(hQ, carry);
    for (uint32_t t = 0; t < steps; ++t) {
        float x[22], a[4], hn[4][4];
#pragma unroll
        for (int k = 0; k < 22; ++k) x[k] = field(obs, t * 22 + k, ld)[i];
        const uint8_t d = done[(size_t)t * ld + i];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) hn[tt][r] = hQ[tt][r];
        const typename ACTOR::Saved before = actor.carry_of(carry);
        actor.template step_fused<0>(x, hn, a, carry, [] {});
        if (squash) squash_action(a);
        select_hidden_q(__builtin_amdgcn_ballot_w64(d != 4), hn, hQ);          // frozen: state not advanced
        const uint64_t held = __builtin_amdgcn_ballot_w64(d == 4);
        if (held != 0) actor.hold_carry(held, before, carry);                   // ... nor what is carried for its next step
        const uint64_t ended = __builtin_amdgcn_ballot_w64(d == 1 || d == 2);
        if (ended != 0) {                                                       // episode end: policy reset
            select_hidden_q(ended, h0Q, hQ);
            actor.reset_carry(ended, hQ, carry);
        }
        if (valid) {
#pragma unroll
            for (int k = 0; k < 4; ++k) field(act, t * 4 + k, ld)[i] = a[k];
        }
    }
    store_hidden_q(hidden, ld_h, wave_base, __builtin_amdgcn_ballot_w64(valid), hQ);
}

// ------------------------------------------------------------------ step ---------------
__device__ __forceinline__ Stats load_stats(const StatsPtrs& st, uint32_t i) {
    return {st.returns[i], st.steps[i], st.fin_returns[i], st.fin_lengths[i], st.fin_counts[i], st.fin_terminated[i]};
}
__device__ __forceinline__ void store_stats(const StatsPtrs& st, uint32_t i, const Stats& s, bool ended) {
    st.returns[i] = s.ret;
    st.steps[i] = s.steps;
    if (ended) {
        st.fin_returns[i] = s.fin_ret; st.fin_lengths[i] = s.fin_len;
        st.fin_counts[i] = s.fin_cnt; st.fin_terminated[i] = s.fin_term;
    }
}

// the observation of the state a step writes, produced by the step itself: obs [RQ_OBSERVATION_DIM][ld] (nullptr = off).
// rq_step below 1 024 envs (the small-batch loop's cache, no noise) and the chained rollout's two-kernel step (round 3:
// observe of step t + 1 folded into step t; with noise: the draw of epoch `epoch` + *epoch_base, as k_observe makes it).
struct ObsNext {
    float* obs;
    NoiseCfg nc;
    uint32_t noise;               // 0 / 1
    uint32_t epoch;
    const uint32_t* epoch_base;
};

template <bool ROLLOUT>
__device__ __forceinline__ void step_env(uint32_t i, const Batch& b, const StepCfg& c, const float* __restrict__ params,
                                         const float* state, float* __restrict__ action, float* next_state,
                                         const StatsPtrs& st, uint32_t flags, const SampleCfg& sc, uint64_t seed,
                                         float* __restrict__ hidden, const float* __restrict__ weights,
                                         const Mailbox& mb, const ObsNext& on) {
    if (ROLLOUT && st.frozen[i]) { st.last_done[i] = 4; return; }
    const size_t ld = b.ld;
    const EnvConsts k = make_consts([&](int f) { return field(params, f, ld)[i]; });
    QuadState y;
    float f6[6], a[4];
    f32x2 AC01, AC23;
    y.load([&](int j) { return field(state, j, ld)[i]; });
#pragma unroll
    for (int j = 0; j < 6; ++j) f6[j] = field(state, (RQ_S_FORCE + j), ld)[i];
    if (mb.rows_in != nullptr) {      // actions handed over in the host mailbox (kernel argument)
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = mb.rows_in[(size_t)i * mb.in_stride + j]; field(action, j, ld)[i] = a[j]; }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = field(action, j, ld)[i];
    }
    Stats s = load_stats(st, i);
    const Disturbance ds = make_disturbance(k, c.gravity, f6);
    bool term;
    const float r = step_inplace<false>(c, k, ds, y, a, AC01, AC23, term);
    if (c.action_history_raw) { AC01 = f32x2{a[0], a[1]}; AC23 = f32x2{a[2], a[3]}; }   // what ActionHistory(1) keeps
    const bool ended = stats_update(c.episode_step_limit, r, term, s);
    st.last_reward[i] = r;
    st.last_terminated[i] = term ? 1 : 0;
    st.last_done[i] = term ? 1 : (ended ? 2 : 0);
    store_stats(st, i, s, ended);
    bool write_dist = (next_state != state);
    if (ROLLOUT && ended) {
        if (flags & RQ_ROLLOUT_AUTORESET) {
            const uint32_t ep = st.episode[i];
            float fresh[17], la0[4];
            sample_state(sc, seed, ep, b.env_offset + i, field(params, RQ_P_MASS, ld)[i],
                         field(params, RQ_P_HOVER_RPM, ld)[i], field(params, RQ_P_ROTOR_POS, ld)[i],
                         field(params, (RQ_P_ROTOR_POS + 1), ld)[i], fresh, la0, f6);
            y.load([&](int j) { return fresh[j]; });
            AC01 = f32x2{la0[0], la0[1]}; AC23 = f32x2{la0[2], la0[3]};
            st.episode[i] = ep + 1;
            write_dist = true;
#pragma unroll
            for (int j = 0; j < 16; ++j) field(hidden, j, ld)[i] = weights[OFF_H0 + j];
        } else {
            st.frozen[i] = 1;
        }
    }
    y.store([&](int j, float v) { field(next_state, j, ld)[i] = v; });
    field(next_state, (RQ_S_LAST_ACTION + 0), ld)[i] = AC01[0]; field(next_state, (RQ_S_LAST_ACTION + 1), ld)[i] = AC01[1];
    field(next_state, (RQ_S_LAST_ACTION + 2), ld)[i] = AC23[0]; field(next_state, (RQ_S_LAST_ACTION + 3), ld)[i] = AC23[1];
    if (on.obs != nullptr) {   // wave-uniform (kernel argument): what k_observe would assemble for the state just written
        float head[22], o[RQ_OBSERVATION_DIM];
        if (on.noise) {
            const uint32_t epoch = on.epoch + (on.epoch_base != nullptr ? *on.epoch_base : 0u);
            observe_head<true>(y, AC01, AC23, on.nc, seed, epoch, b.env_offset + i, head);
        } else {
            observe_head<false>(y, AC01, AC23, on.nc, seed, 0u, b.env_offset + i, head);
        }
#pragma unroll
        for (int j = 0; j < 22; ++j) o[j] = head[j];
        const float inv = 2.0f / (k.rmax - k.rmin);
        o[22] = fmaf(y.R01[0] - k.rmin, inv, -1.0f); o[23] = fmaf(y.R01[1] - k.rmin, inv, -1.0f);
        o[24] = fmaf(y.R23[0] - k.rmin, inv, -1.0f); o[25] = fmaf(y.R23[1] - k.rmin, inv, -1.0f);
#pragma unroll
        for (int j = 0; j < RQ_OBSERVATION_DIM; ++j) put<kNtObs>(&field(on.obs, j, ld)[i], o[j]);
        if (mb.rows_out != nullptr) {
#pragma unroll
            for (int j = 0; j < RQ_OBSERVATION_DIM; ++j) mb.rows_out[(size_t)i * RQ_OBSERVATION_DIM + j] = o[j];
        }
    }
    if (write_dist) {
#pragma unroll
        for (int j = 0; j < 6; ++j) field(next_state, (RQ_S_FORCE + j), ld)[i] = f6[j];
    }
}


template <bool ROLLOUT>
__global__ __launch_bounds__(kBlock) void k_step(Batch b, StepCfg c, const float* __restrict__ params,
                                                 const float* state, float* __restrict__ action,
                                                 float* next_state, StatsPtrs st, uint32_t flags, SampleCfg sc,
                                                 uint64_t seed, float* __restrict__ hidden,
                                                 const float* __restrict__ weights, Mailbox mb, ObsNext on) {
    const uint32_t i = env_index();
    if (i < b.n)
        step_env<ROLLOUT>(i, b, c, params, state, action, next_state, st, flags, sc, seed, hidden, weights, mb, on);
    mailbox_signal(mb);
}
// ------------------------------------------------------------------ resident executor ---
// The reference's loop at its own batch (README.md:96-99, `vector8`): observe -> evaluate_step -> step -> assign on a handful of envs
// with NumPy arrays at every call.  Rounds 3-5 served an iteration with ONE kernel round trip (k_step assembles the next observation,
// a speculative k_actor_step evaluates the policy on it) - two launches, 19.4 us per iteration against 6.9 us for a native CPU loop.
// What is left of that is launch + completion.  This kernel removes it: ONE workgroup stays on the device for as long as the host
// keeps calling step() on the same objects, polls a 64-byte command line in pinned host memory, and for every command does exactly what
// the two launches did - what step_env<false> does (resident_env_step below: actions in from pinned rows, next state, statistics, the
// next observation to the env's buffer and to pinned rows), then ACTOR::step on that observation (new hidden state to the policy's spare buffer, actions to pinned
// rows) - publishing the same two sequence numbers in the same pinned flag.  The policy's operand image is loaded once for the
// kernel's lifetime.  The kernel leaves on a QUIT command, or by itself after `idle_ticks` without one (a host that died or went
// away must not leave a wave spinning), and says so in `exited`; a command it never consumed is replayed by the host as launches
// (rq_capi_vector.cpp resident_*).  It is never the device stream's business: the host retires it before anything else is enqueued.
// Kernels: k_resident_loop (13 .. 256 envs, 1 - 4 waves), k_resident_small (at most 12 envs), k_resident_policy (the policy alone).

// A single-wave kernel publishes a sequence number: every lane's stores of this command are released to the system by ONE fence the
// whole wave executes (write-back + wait), then lane 0 stores the number.  (__threadfence_system() + a release store, as the launches'
// mailbox_signal writes it, is that fence, an invalidate nobody needs here, and the write-back a second time: 0.2 us per publication.)
__device__ __forceinline__ void publish(uint32_t* flag, uint32_t seq, uint32_t lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (lane == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// What a resident wave keeps of its env from command to command: the constants (the parameters cannot change under a running kernel),
// the state, the disturbance, the statistics.  A command whose input buffer is the one the previous command wrote (the loop's own shape:
// state.assign(next_state)) loads nothing; any other buffer is loaded as step_env loads it.  Everything is still STORED every step:
// the buffers are what the API shows.
struct ResidentEnv {
    EnvConsts k;
    QuadState y;
    float f6[6];
    Stats st;
    const float* have_state;         // the buffer whose contents y / f6 hold (wave-uniform)
};

// step_env<false> for one command of a resident wave, its loads skipped where the registers hold the values: same functions in the
// same order, same stores (+ the observation rows to pinned memory first: they are what the host waits for).  x: the 22 policy inputs
// of the observation this step assembles (zeros for lanes past the batch: they feed the matrix cores).
__device__ __forceinline__ void resident_env_step(const ResidentArgs& ra, ResidentEnv& e, uint32_t i, bool valid, const float* state_in,
                                                  float* state_out, float* obs_out, const float (&a_in)[4], float (&x)[22]) {
    const size_t ld = ra.b.ld;
    if (state_in != e.have_state) {          // wave-uniform: not the buffer the previous command wrote
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        e.y.load([&](int f) { return field(state_in, f, ld)[i]; });
#pragma unroll
        for (int f = 0; f < 6; ++f) e.f6[f] = field(state_in, (RQ_S_FORCE + f), ld)[i];
    }
#pragma unroll
    for (int c = 0; c < 22; ++c) x[c] = 0.0f;
    f32x2 AC01, AC23;
    const Disturbance ds = make_disturbance(e.k, ra.c.gravity, e.f6);
    bool term;
    const float r = step_inplace<false>(ra.c, e.k, ds, e.y, a_in, AC01, AC23, term);
    if (ra.c.action_history_raw) { AC01 = f32x2{a_in[0], a_in[1]}; AC23 = f32x2{a_in[2], a_in[3]}; }
    const bool ended = stats_update(ra.c.episode_step_limit, r, term, e.st);
    float o[RQ_OBSERVATION_DIM];
    observe_head<false>(e.y, AC01, AC23, NoiseCfg{}, ra.seed, 0u, ra.b.env_offset + i, x);
#pragma unroll
    for (int c = 0; c < 22; ++c) o[c] = x[c];
    const float inv = 2.0f / (e.k.rmax - e.k.rmin);
    o[22] = fmaf(e.y.R01[0] - e.k.rmin, inv, -1.0f); o[23] = fmaf(e.y.R01[1] - e.k.rmin, inv, -1.0f);
    o[24] = fmaf(e.y.R23[0] - e.k.rmin, inv, -1.0f); o[25] = fmaf(e.y.R23[1] - e.k.rmin, inv, -1.0f);
    if (valid) {
#pragma unroll
        for (int c = 0; c < RQ_OBSERVATION_DIM; ++c) ra.rows_obs[(size_t)i * RQ_OBSERVATION_DIM + c] = o[c];       // what the host waits for first
#pragma unroll
        for (int c = 0; c < 4; ++c) field(ra.act, c, ld)[i] = a_in[c];
        ra.st.last_reward[i] = r;
        ra.st.last_terminated[i] = term ? 1 : 0;
        ra.st.last_done[i] = term ? 1 : (ended ? 2 : 0);
        store_stats(ra.st, i, e.st, ended);
        e.y.store([&](int f, float v) { field(state_out, f, ld)[i] = v; });
        field(state_out, (RQ_S_LAST_ACTION + 0), ld)[i] = AC01[0]; field(state_out, (RQ_S_LAST_ACTION + 1), ld)[i] = AC01[1];
        field(state_out, (RQ_S_LAST_ACTION + 2), ld)[i] = AC23[0]; field(state_out, (RQ_S_LAST_ACTION + 3), ld)[i] = AC23[1];
#pragma unroll
        for (int c = 0; c < RQ_OBSERVATION_DIM; ++c) put<kNtObs>(&field(obs_out, c, ld)[i], o[c]);
#pragma unroll
        for (int f = 0; f < 6; ++f) field(state_out, (RQ_S_FORCE + f), ld)[i] = e.f6[f];
    } else {
#pragma unroll
        for (int c = 0; c < 22; ++c) x[c] = 0.0f;                 // lanes past the batch feed the matrix cores zeros
    }
    e.have_state = state_out;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES <= 4 ? 1 : 2) void k_resident_loop(ResidentArgs ra) {
    typedef ActorF32Lean ACTOR;                 // the build launch_actor_step takes for fp32 policies: the same bits
    __shared__ uint32_t sh_pkt[16];
    __shared__ uint32_t sh_sum[WAVES];
    ACTOR actor;
    actor.template load<WAVES>(ra.packed);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t wave_base = wave * 64u;
    const uint32_t n = ra.b.n;
    const uint32_t i0 = wave_base + lane;
    const bool valid = i0 < n;
    const uint32_t i = valid ? i0 : n - 1;
    // (round 6, second half) env constants, state, statistics and the policy's hidden state stay in registers from command to command,
    // as in k_resident_small below: a command that continues where the previous one stopped loads nothing (README loop at 100 envs:
    // two dependent trips to memory less per iteration)
    ResidentEnv e{make_consts([&](int f) { return field(ra.params, f, ra.b.ld)[i]; }), QuadState{}, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
                  load_stats(ra.st, i), nullptr};
    float hQ[4][4] = {};
    const float* have_hidden = nullptr;
    uint32_t expect = ra.first_packet;
    uint32_t left_bits = 0;                     // why the kernel left (workgroup-uniform)
    unsigned long long idle_since = (unsigned long long)wall_clock64();
    const unsigned long long born = idle_since;
    for (;;) {
        if (wave == 0) {
            uint32_t w = 0;
            for (;;) {
                if (lane < 16) w = __hip_atomic_load(const_cast<uint32_t*>(ra.packet) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const uint32_t head = __builtin_amdgcn_readlane(w, kRpHead), tail = __builtin_amdgcn_readlane(w, kRpTail);
                if (head == expect && tail == expect) break;
                const unsigned long long now = (unsigned long long)wall_clock64();       // idle for too long, or old enough (a device-wide
                if (now - idle_since > ra.idle_ticks || now - born > ra.life_ticks) {                                                 // synchronize waits for this kernel)
                    w = lane == kRpBits ? (kRbQuit | (now - born > ra.life_ticks ? kRbLeftOld : kRbLeftIdle)) : w;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (lane < 16) sh_pkt[lane] = w;
        }
        __syncthreads();
        const uint32_t bits = __builtin_amdgcn_readfirstlane(sh_pkt[kRpBits]);
        if (bits & kRbQuit) { left_bits = bits; break; }                 // workgroup-uniform
        const unsigned long long t_seen = (unsigned long long)wall_clock64();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                    // what the host wrote before the packet (the action rows)
        // (the builtin returns int: without the casts a low half with its top bit set sign-extends over the high half)
        const float* state_in = reinterpret_cast<const float*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(sh_pkt[kRpStateInHi]) << 32) |
                                                               (uint32_t)__builtin_amdgcn_readfirstlane(sh_pkt[kRpStateInLo]));
        float* state_out = reinterpret_cast<float*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(sh_pkt[kRpStateOutHi]) << 32) |
                                                    (uint32_t)__builtin_amdgcn_readfirstlane(sh_pkt[kRpStateOutLo]));
        const uint32_t seq_step = __builtin_amdgcn_readfirstlane(sh_pkt[kRpSeqStep]);
        const uint32_t seq_spec = __builtin_amdgcn_readfirstlane(sh_pkt[kRpSeqSpec]);
        const uint32_t want_sum = __builtin_amdgcn_readfirstlane(sh_pkt[kRpChecksum]);
        float* obs_out = ra.obs_buf[bits & kRbObsSel ? 1 : 0];
        const float* hidden_in = ra.hidden[bits & kRbHiddenSel ? 1 : 0];
        float* hidden_out = ra.hidden[bits & kRbHiddenSel ? 0 : 1];
        // the action rows travel beside the packet, not inside it: their sum must be the packet's (a line of rows read before the
        // host wrote it would otherwise go unnoticed); re-read until it is - the host wrote them before the packet
        float a_in[4];
        bool rows_bad = false;
        for (uint32_t tries = 0;; ++tries) {
            uint32_t sum = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t u = valid ? __hip_atomic_load(reinterpret_cast<const uint32_t*>(ra.rows_action) + (size_t)i * 4 + k,
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
                a_in[k] = __builtin_bit_cast(float, u);
                sum += u;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
            if (WAVES > 1) {
                if (lane == 0) sh_sum[wave] = sum;
                __syncthreads();
                sum = 0;
#pragma unroll
                for (int v = 0; v < WAVES; ++v) sum += sh_sum[v];
                __syncthreads();
            }
            if (sum == want_sum) break;
            // rows that never add up (the host writes them before the line: this does not happen) are not stepped with: the kernel
            // leaves, the command counts as never taken and the host replays it as launches (workgroup-uniform: sum and tries are)
            if (tries > 4096u) { rows_bad = true; break; }
        }
        if (rows_bad) { left_bits = kRbLeftIdle; break; }
        const unsigned long long t_rows = (unsigned long long)wall_clock64();
        float x[22];
        resident_env_step(ra, e, i, valid, state_in, state_out, obs_out, a_in, x);
        const unsigned long long t_stepped = (unsigned long long)wall_clock64();
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(ra.flag, seq_step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t_flag1 = (unsigned long long)wall_clock64();
        // the policy on the observation just assembled: what the speculative k_actor_step launch computed
        float a[4];
        if (hidden_in != have_hidden) load_hidden_q(hidden_in, ra.ld_h, wave_base, n, hQ);      // workgroup-uniform
        actor.step(x, hQ, a);
        store_hidden_q(hidden_out, ra.ld_h, wave_base, __builtin_amdgcn_ballot_w64(valid), hQ);
        have_hidden = hidden_out;
        if (valid) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { field(ra.pol_act, k, ra.ld_h)[i] = a[k]; ra.rows_act[(size_t)i * 4 + k] = a[k]; }
        }
        const unsigned long long t_acted = (unsigned long long)wall_clock64();
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(ra.flag, seq_spec, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        expect += 1;
        idle_since = (unsigned long long)wall_clock64();
        if (threadIdx.x == 0 && ra.timing != nullptr) {                  // a diagnostic (rq_device_get_resident_timing): 100 MHz ticks of this command
            ra.timing[0] = t_seen; ra.timing[1] = t_rows; ra.timing[2] = t_stepped; ra.timing[3] = t_flag1; ra.timing[4] = t_acted;
            ra.timing[5] = idle_since;
        }
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        ra.exited[1] = left_bits & (kRbLeftIdle | kRbLeftOld);      // why: 0 = told to
        __hip_atomic_store(ra.exited, ra.launch_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The same executor for the reference's own batch (`vector8`: at most 12 envs, one wave), with everything that is latency taken out of
// a command's path (device-side timeline of a command at 8 envs: 8.2 us with k_resident_loop<1>, of which 1.7 us fetching the action
// rows, 2.5 us the step, 2.9 us the policy, 1 us the two publications):
//  - the action rows ride in the same vector load as the command line (lanes 16..63 of the poll read 48 dwords beside the 16 of the
//    packet): no second trip over PCIe;
//  - the env's constants, its state, its statistics and the policy's hidden state STAY IN REGISTERS from command to command - a
//    command whose input state / hidden buffer is the one the previous command wrote (the loop's own shape: assign, speculation hit)
//    loads nothing; any other buffer is loaded as before.  Everything is still stored every step: the buffers are what the API shows;
//  - the policy runs on ONE 16-env tile (ActorF32T::step_tile0: 30 MFMAs instead of 120; same bits for the envs that exist).
// Nothing else of the protocol differs: same packet, same flags, same rows out.
__global__ __launch_bounds__(64, 1) void k_resident_small(ResidentArgs ra) {
    typedef ActorF32Lean ACTOR;
    ACTOR actor;
    actor.template load<1>(ra.packed);
    const uint32_t lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
    const uint32_t n = ra.b.n;                 // <= 12
    const bool valid = lane < n;
    const uint32_t i = valid ? lane : n - 1;
    const size_t ld = ra.b.ld;
    ResidentEnv e{make_consts([&](int f) { return field(ra.params, f, ld)[i]; }), QuadState{}, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
                  load_stats(ra.st, i), nullptr};
    float hq[4] = {0.f, 0.f, 0.f, 0.f};
    const float* have_hidden = nullptr;
    const uint32_t hj = j < n ? j : n - 1;     // tile 0 of the Q layout: lane (q, j) = env j, hidden features 4 q .. 4 q + 3
    uint32_t bits_at_exit = 0;
    const uint32_t* poll_at = lane < 16 ? const_cast<const uint32_t*>(ra.packet) + lane : ra.small_rows + (lane - 16);
    uint32_t expect = ra.first_packet;
    unsigned long long idle_since = (unsigned long long)wall_clock64();
    const unsigned long long born = idle_since;
    for (;;) {
        uint32_t w, bits;
        for (;;) {
            w = __hip_atomic_load(poll_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const uint32_t head = __builtin_amdgcn_readlane(w, kRpHead), tail = __builtin_amdgcn_readlane(w, kRpTail);
            bits = __builtin_amdgcn_readlane(w, kRpBits);
            if (head == expect && tail == expect) {
                if (bits & kRbQuit) break;
                uint32_t sum = (lane >= 16 && lane < 16 + 4 * n) ? w : 0u;       // the rows travelled beside the packet: their sum is in it
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
                if (sum == (uint32_t)__builtin_amdgcn_readlane(w, kRpChecksum)) break;
            }
            const unsigned long long now = (unsigned long long)wall_clock64();           // idle for too long, or old enough (a device-wide
            if (now - idle_since > ra.idle_ticks || now - born > ra.life_ticks) {                                  // synchronize waits for this kernel)
                bits = kRbQuit | (now - born > ra.life_ticks ? kRbLeftOld : kRbLeftIdle);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (bits & kRbQuit) { bits_at_exit = bits; break; }
        const bool timed = ra.timing != nullptr;       // a diagnostic (RQ_RESIDENT_TIMING): reading the clock five times costs a command ~0.5 us
        const unsigned long long t_seen = timed ? (unsigned long long)wall_clock64() : 0ull;
        const float* state_in = reinterpret_cast<const float*>(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(w, kRpStateInHi) << 32) |
                                                               (uint32_t)__builtin_amdgcn_readlane(w, kRpStateInLo));
        float* state_out = reinterpret_cast<float*>(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(w, kRpStateOutHi) << 32) |
                                                    (uint32_t)__builtin_amdgcn_readlane(w, kRpStateOutLo));
        const uint32_t seq_step = __builtin_amdgcn_readlane(w, kRpSeqStep), seq_spec = __builtin_amdgcn_readlane(w, kRpSeqSpec);
        float* obs_out = ra.obs_buf[bits & kRbObsSel ? 1 : 0];
        const float* hidden_in = ra.hidden[bits & kRbHiddenSel ? 1 : 0];
        float* hidden_out = ra.hidden[bits & kRbHiddenSel ? 0 : 1];
        float a_in[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) a_in[c] = __builtin_bit_cast(float, (uint32_t)__shfl((int)w, (int)(16 + 4 * i + c)));
        const unsigned long long t_rows = timed ? (unsigned long long)wall_clock64() : 0ull;
        float x[22];
        if (hidden_in != have_hidden) {
#pragma unroll
            for (int c = 0; c < 4; ++c) hq[c] = hidden_in[(size_t)(4 * q + c) * ra.ld_h + hj];
        }
        resident_env_step(ra, e, i, valid, state_in, state_out, obs_out, a_in, x);
        const unsigned long long t_stepped = timed ? (unsigned long long)wall_clock64() : 0ull;
        // ---- the policy on that observation; the step's rows are published from inside it, behind the MFMAs that need neither the
        // observation nor the wait for the rows' stores (0.4 us of a command's 3.5 stood there) ----
        unsigned long long t_flag1 = 0ull;
        float a[4];
        actor.step_tile0(x, hq, a, [&] {
            publish(ra.flag, seq_step, lane);
            if (timed) t_flag1 = (unsigned long long)wall_clock64();
        });
        if (j < n) {
#pragma unroll
            for (int c = 0; c < 4; ++c) hidden_out[(size_t)(4 * q + c) * ra.ld_h + j] = hq[c];
        }
        have_hidden = hidden_out;
        if (valid) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { ra.rows_act[(size_t)i * 4 + c] = a[c]; field(ra.pol_act, c, ra.ld_h)[i] = a[c]; }
        }
        const unsigned long long t_acted = timed ? (unsigned long long)wall_clock64() : 0ull;
        publish(ra.flag, seq_spec, lane);
        expect += 1;
        idle_since = (unsigned long long)wall_clock64();
        if (lane == 0 && timed) {
            ra.timing[0] = t_seen; ra.timing[1] = t_rows; ra.timing[2] = t_stepped; ra.timing[3] = t_flag1; ra.timing[4] = t_acted;
            ra.timing[5] = idle_since;
        }
    }
    __threadfence_system();
    if (lane == 0) {
        ra.exited[1] = bits_at_exit & (kRbLeftIdle | kRbLeftOld);      // why: 0 = told to
        __hip_atomic_store(ra.exited, ra.launch_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The policy alone, for a caller with a simulator of its own (README.md:17-25: `policy.evaluate_step(observation)[0]` a thousand times
// at batch 1): one wave stays on the device, takes the observation rows of at most 16 envs as commands and answers with the action rows -
// a PCIe round trip around ~1.5 us of work instead of a launch and its completion (13.9 us per call at batch 1 -> see DESIGN.md section 5).
// Same protocol as above (command line, checksum over the rows, `exited`), same arithmetic as the launch it replaces
// (ActorF32T::step_tile0: the bits k_actor_step gives envs 0..15), hidden state resident in registers and stored every step to the
// policy's buffer (that buffer is what the API shows).  Rows in command memory: [batch][24] floats (22 + 2 of padding: the host writes
// whole 16-byte words); up to two rows ride in the poll itself (lanes 16..63), more are fetched once the line has been seen.
__global__ __launch_bounds__(64, 1) void k_resident_policy(ResidentArgs ra) {
    typedef ActorF32Lean ACTOR;
    ACTOR actor;
    actor.template load<1>(ra.packed);
    const uint32_t lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
    const uint32_t n = ra.b.n;                 // <= kResidentPolicyBatch
    const bool valid = lane < n;
    const uint32_t i = valid ? lane : n - 1;
    const uint32_t hj = j < n ? j : n - 1;     // tile 0 of the Q layout: lane (q, j) = env j, hidden features 4 q .. 4 q + 3
    float* hidden = ra.hidden[0];
    float hq[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) hq[c] = hidden[(size_t)(4 * q + c) * ra.ld_h + hj];
    const bool in_poll = n <= 2;               // wave-uniform
    const uint32_t* poll_at = lane < 16 ? const_cast<const uint32_t*>(ra.packet) + lane : ra.small_rows + (lane - 16);
    uint32_t expect = ra.first_packet, left_bits = 0;
    unsigned long long idle_since = (unsigned long long)wall_clock64();
    const unsigned long long born = idle_since;
    for (;;) {
        uint32_t w, bits;
        float x[22];
        for (;;) {
            w = __hip_atomic_load(poll_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const uint32_t head = __builtin_amdgcn_readlane(w, kRpHead), tail = __builtin_amdgcn_readlane(w, kRpTail);
            bits = __builtin_amdgcn_readlane(w, kRpBits);
            if (head == expect && tail == expect) {
                if (bits & kRbQuit) break;
                uint32_t sum = 0;
                if (in_poll) {
#pragma unroll
                    for (int c = 0; c < 22; ++c) {
                        const uint32_t u = (uint32_t)__shfl((int)w, (int)(16 + 24 * i + c));
                        x[c] = __builtin_bit_cast(float, u);
                        sum += valid ? u : 0u;
                    }
                } else {
                    uint32_t u[22];
#pragma unroll
                    for (int c = 0; c < 22; ++c)
                        u[c] = __hip_atomic_load(ra.small_rows + 24 * i + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
                    for (int c = 0; c < 22; ++c) { x[c] = __builtin_bit_cast(float, u[c]); sum += valid ? u[c] : 0u; }
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
                if (sum == (uint32_t)__builtin_amdgcn_readlane(w, kRpChecksum)) break;      // else: rows not all there yet - look again
            }
            const unsigned long long now = (unsigned long long)wall_clock64();
            if (now - idle_since > ra.idle_ticks || now - born > ra.life_ticks) {
                bits = kRbQuit | (now - born > ra.life_ticks ? kRbLeftOld : kRbLeftIdle);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (bits & kRbQuit) { left_bits = bits; break; }
        const uint32_t seq = __builtin_amdgcn_readlane(w, kRpSeqSpec);
        if (!valid) {
#pragma unroll
            for (int c = 0; c < 22; ++c) x[c] = 0.0f;                 // lanes past the batch feed the matrix cores zeros
        }
        float a[4];
        actor.step_tile0(x, hq, a, [] {});
        if (j < n) {
#pragma unroll
            for (int c = 0; c < 4; ++c) hidden[(size_t)(4 * q + c) * ra.ld_h + j] = hq[c];
        }
        if (valid) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { ra.rows_act[(size_t)i * 4 + c] = a[c]; field(ra.pol_act, c, ra.ld_h)[i] = a[c]; }
        }
        publish(ra.flag, seq, lane);
        expect += 1;
        idle_since = (unsigned long long)wall_clock64();
    }
    __threadfence_system();
    if (lane == 0) {
        ra.exited[1] = left_bits & (kRbLeftIdle | kRbLeftOld);      // why: 0 = told to
        __hip_atomic_store(ra.exited, ra.launch_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t launch_resident_policy(hipStream_t s, const ResidentArgs& ra) {
    if (ra.b.n == 0 || ra.b.n > kResidentPolicyBatch || ra.small_rows == nullptr) return hipErrorInvalidValue;
    k_resident_policy<<<1, 64, 0, s>>>(ra);
    return hipGetLastError();
}

hipError_t launch_resident(hipStream_t s, const ResidentArgs& ra) {
    const uint32_t waves = (ra.b.n + 63u) / 64u;
    if (ra.b.n == 0 || waves > 4) return hipErrorInvalidValue;
    if (ra.b.n <= kResidentSmallEnvs && ra.small_rows != nullptr) k_resident_small<<<1, 64, 0, s>>>(ra);
    else if (waves <= 1) k_resident_loop<1><<<1, 64, 0, s>>>(ra);
    else if (waves <= 2) k_resident_loop<2><<<1, 128, 0, s>>>(ra);
    else                 k_resident_loop<4><<<1, 256, 0, s>>>(ra);
    return hipGetLastError();
}

// Chained-mode counterpart of the fused kernel's prologue under auto-reset: envs left frozen by an earlier
// rollout start their next episode (sample_initial_state with the env's episode counter, policy state reset).
__global__ __launch_bounds__(kBlock) void k_thaw_frozen(Batch b, SampleCfg c, uint64_t seed,
                                                        const float* __restrict__ params, float* __restrict__ state,
                                                        StatsPtrs st, float* __restrict__ hidden,
                                                        const float* __restrict__ weights) {
    const uint32_t i = env_index();
    if (i >= b.n || !st.frozen[i]) return;
    const size_t ld = b.ld;
    const uint32_t ep = st.episode[i];
    float s[17], la[4], f[6];
    sample_state(c, seed, ep, b.env_offset + i, field(params, RQ_P_MASS, ld)[i], field(params, RQ_P_HOVER_RPM, ld)[i],
                 field(params, RQ_P_ROTOR_POS, ld)[i], field(params, (RQ_P_ROTOR_POS + 1), ld)[i], s, la, f);
#pragma unroll
    for (int k = 0; k < 17; ++k) field(state, k, ld)[i] = s[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) field(state, (RQ_S_LAST_ACTION + k), ld)[i] = la[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) field(state, (RQ_S_FORCE + k), ld)[i] = f[k];
#pragma unroll
    for (int j = 0; j < 16; ++j) field(hidden, j, ld)[i] = weights[OFF_H0 + j];
    st.episode[i] = ep + 1;
    st.frozen[i] = 0;
}

__global__ __launch_bounds__(kBlock) void k_fill_f32(float* p, float v, uint32_t count) {
    const uint32_t i = env_index();
    if (i < count) p[i] = v;
}

// ------------------------------------------------------------------ launchers ----------
static inline unsigned grid_for(uint32_t n, int block) { return (n + block - 1) / block; }

hipError_t launch_sample_params(hipStream_t s, Batch b, SampleCfg c, uint64_t seed, uint32_t epoch, float* params) {
    if (b.n == 0) return hipSuccess;
    k_sample_params<<<grid_for(b.n, kBlock), kBlock, 0, s>>>(b, c, seed, epoch, params);
    return hipGetLastError();
}

hipError_t launch_sample_state(hipStream_t s, Batch b, SampleCfg c, uint64_t seed, const float* params,
                               float* state, StatsPtrs st) {
    if (b.n == 0) return hipSuccess;
    k_sample_state<<<grid_for(b.n, kBlock), kBlock, 0, s>>>(b, c, seed, params, state, st);
    return hipGetLastError();
}

hipError_t launch_observe(hipStream_t s, Batch b, NoiseCfg nc, bool noise, uint64_t seed, uint32_t epoch,
                          const uint32_t* epoch_base, const float* params, const float* state, float* obs,
                          Mailbox mb) {
    if (b.n == 0) return hipSuccess;
    if (noise) k_observe<true><<<grid_for(b.n, kBlock), kBlock, 0, s>>>(b, nc, seed, epoch, epoch_base, params, state, obs, mb);
    else       k_observe<false><<<grid_for(b.n, kBlock), kBlock, 0, s>>>(b, nc, seed, epoch, epoch_base, params, state, obs, mb);
    return hipGetLastError();
}

__global__ void k_advance_u32(uint32_t* p, uint32_t add, uint32_t set, int do_set) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *p = do_set ? set : *p + add;
}
hipError_t launch_set_u32(hipStream_t s, uint32_t* p, uint32_t value) {
    k_advance_u32<<<1, 64, 0, s>>>(p, 0u, value, 1);
    return hipGetLastError();
}
hipError_t launch_add_u32(hipStream_t s, uint32_t* p, uint32_t add) {
    RQ_KLAUNCH(k_advance_u32, 1, 64, s, p, add, 0u, 0);
    return RQ_KLAUNCH_STATUS();
}

hipError_t launch_actor_step(hipStream_t s, uint32_t n, const float* packed, const float* obs, uint32_t ld_obs,
                             float* hidden, uint32_t ld_h, float* act, uint32_t ld_act, const uint8_t* frozen,
                             int precision, SasArgs sas, Mailbox mb, const float* hidden_in) {
    if (hidden_in == nullptr) hidden_in = hidden;
    if (n == 0) return hipSuccess;
    // enough waves to fill the 1024 SIMDs first, then several 64-env groups per wave so that the
    // per-wave operand-image load (18 KB, more than a group's own 14.8 KB of data) is amortised
    // (2 097 152 envs, groups per wave 2 / 4 / 8 / 11 / 16 / 32: 125.9 / 119.6 / 116.6 / 116.2 / 122.2 / 110.8 us)
    const uint32_t groups = (n + 63) / 64;
    static const uint32_t forced = [] { const char* e = std::getenv("RQ_ACTOR_GROUPS_PER_WAVE"); return e ? (uint32_t)std::atoi(e) : 0u; }();
    const uint32_t gpw = forced ? forced : actor_groups_per_wave(n);       // the override: launch-shape sweeps (tools/)
    const unsigned grid = grid_for((groups + gpw - 1) / gpw * 64, kBlock);
    // the streaming kernel: no host rows (they exist below 1 024 envs only) and every byte offset in 32 bits; otherwise one
    // group per wave
    const bool stream = gpw > 1 && mb.rows_in == nullptr && mb.rows_out == nullptr && mb.flag == nullptr &&
                        (uint64_t)(ld_h > ld_obs ? ld_h : ld_obs) * 4u * 26u < 0x7FFFFFFFull;
    const unsigned grid1 = grid_for(groups * 64, kBlock);
#define RQ_LAUNCH_ACTOR(ACT)                                                                                                            \
    do {                                                                                                                                \
        if (stream) RQ_KLAUNCH(k_actor_stream<ACT>, grid, kBlock, s, n, gpw, packed, obs, ld_obs, hidden_in, hidden, ld_h, act, ld_act, frozen, sas);   \
        else        RQ_KLAUNCH(k_actor_step<ACT>, grid1, kBlock, s, n, packed, obs, ld_obs, hidden_in, hidden, ld_h, act, ld_act, frozen, sas, mb);    \
    } while (0)
    if (precision == RQ_POLICY_F16X2_MFMA)     RQ_LAUNCH_ACTOR(ActorF16X2);
    else if (precision == RQ_POLICY_BF16_MFMA) RQ_LAUNCH_ACTOR(ActorBF16);
    else                                       RQ_LAUNCH_ACTOR(ActorF32Lean);   // the two-tiles-per-pass build: ~30 registers fewer live, 2-3 % faster at every size
#undef RQ_LAUNCH_ACTOR
    return RQ_KLAUNCH_STATUS();
}

hipError_t launch_actor_sequence(hipStream_t s, uint32_t n, uint32_t steps, const float* packed, const float* obs,
                                 uint32_t stride, float* hidden, uint32_t ld_h, float* act, int precision) {
    if (n == 0 || steps == 0) return hipSuccess;
    const uint32_t squash = ((uint32_t)precision >> 8) & 1u;
    precision &= 0xff;
    const unsigned g = grid_for(n, kFusedBlock);
    const bool lean = n > 65536u;          // two waves per SIMD only pay when there are that many
#define RQ_LAUNCH_SEQ(ACT) k_actor_sequence<ACT><<<g, kFusedBlock, 0, s>>>(n, steps, packed, obs, stride, hidden, ld_h, act, squash)
    if (precision == RQ_POLICY_F16X2_MFMA) RQ_LAUNCH_SEQ(ActorF16X2);
    else if (precision == RQ_POLICY_BF16_MFMA) RQ_LAUNCH_SEQ(ActorBF16);          // one build at every size (round 5: see ActorBF16Lean)
    else                                  { if (lean) RQ_LAUNCH_SEQ(ActorF32Lean); else RQ_LAUNCH_SEQ(ActorF32); }
#undef RQ_LAUNCH_SEQ
    return hipGetLastError();
}

hipError_t launch_actor_relabel(hipStream_t s, uint32_t n, uint32_t ld, uint32_t steps, const float* packed,
                                const float* obs, const uint8_t* done, float* hidden, uint32_t ld_h, float* act,
                                int precision) {
    if (n == 0 || steps == 0) return hipSuccess;
    const uint32_t squash = ((uint32_t)precision >> 8) & 1u;
    precision &= 0xff;
    const unsigned g = grid_for(n, kFusedBlock);
    const bool lean = n > 65536u;
#define RQ_LAUNCH_RELABEL(ACT) k_actor_relabel<ACT><<<g, kFusedBlock, 0, s>>>(n, ld, steps, packed, obs, done, hidden, ld_h, act, squash)
    if (precision == RQ_POLICY_F16X2_MFMA) RQ_LAUNCH_RELABEL(ActorF16X2);
    else if (precision == RQ_POLICY_BF16_MFMA) RQ_LAUNCH_RELABEL(ActorBF16);      // one build at every size (round 5: see ActorBF16Lean)
    else                                  { if (lean) RQ_LAUNCH_RELABEL(ActorF32Lean); else RQ_LAUNCH_RELABEL(ActorF32); }
#undef RQ_LAUNCH_RELABEL
    return hipGetLastError();
}

hipError_t launch_step(hipStream_t s, Batch b, StepCfg c, const float* params, const float* state,
                       float* action, float* next_state, StatsPtrs st, int rollout, uint32_t flags,
                       SampleCfg sc, uint64_t seed, float* hidden, const float* weights, Mailbox mb, float* obs_of_next,
                       NoiseCfg nc, bool noise, uint32_t obs_epoch, const uint32_t* obs_epoch_base) {
    if (b.n == 0) return hipSuccess;
    const ObsNext on{obs_of_next, nc, noise ? 1u : 0u, obs_epoch, obs_epoch_base};
    if (rollout)
        RQ_KLAUNCH(k_step<true>, grid_for(b.n, kBlock), kBlock, s, b, c, params, state, action, next_state, st, flags,
                   sc, seed, hidden, weights, mb, on);
    else
        RQ_KLAUNCH(k_step<false>, grid_for(b.n, kBlock), kBlock, s, b, c, params, state, action, next_state, st, flags,
                   sc, seed, hidden, weights, mb, on);
    return RQ_KLAUNCH_STATUS();
}

hipError_t launch_thaw_frozen(hipStream_t s, Batch b, SampleCfg c, uint64_t seed, const float* params, float* state,
                              StatsPtrs st, float* hidden, const float* weights) {
    if (b.n == 0) return hipSuccess;
    k_thaw_frozen<<<grid_for(b.n, kBlock), kBlock, 0, s>>>(b, c, seed, params, state, st, hidden, weights);
    return hipGetLastError();
}

hipError_t launch_rollout_fused(hipStream_t s, Batch b, StepCfg c, NoiseCfg nc, bool noise, SampleCfg sc,
                                uint64_t seed, uint32_t epoch0, uint32_t n_steps, uint32_t flags,
                                const float* params, float* state, float* hidden, const float* weights,
                                const float* packed, StatsPtrs st, int precision, SasArgs sas, TrajPtrs traj,
                                unsigned long long* span) {
    if (b.n == 0 || n_steps == 0) return hipSuccess;
    const bool ar = (flags & RQ_ROLLOUT_AUTORESET) != 0;
    const FusedArgs a{b, c, nc, sc, seed, epoch0, n_steps, params, state, hidden, weights, packed, st, traj, sas, span};
    // the 16-bit actors live in their own translation unit (rq_kernels_16bit.hip: another instruction scheduler)
    if (precision == RQ_POLICY_F16X2_MFMA || precision == RQ_POLICY_BF16_MFMA)
        return launch_rollout_fused_16bit(s, a, noise, ar, precision);
    // Two builds of the same loop (same arithmetic, GRU two tiles at a time): a 512-register one for one wave per
    // SIMD - every batch up to 65 536 envs (1024 SIMDs x 64 lanes) - and a 256-register one, two waves per SIMD,
    // beyond.  The 256-register build parks loop invariants in scratch before the loop (~7 us per launch); at one
    // wave per SIMD both run the loop at the same speed (3.21 vs 3.23 us/step), so the small batches take the
    // build with the cheaper prologue.  With the SampleAndSquash stage: only the 256-register builds carry it.
    const bool lean = b.n > 65536u;
    if (sas.mode != RQ_SAS_OFF) launch_fused_actor<true, ActorF32Lean>(s, a, noise, ar);
    else if (lean)              launch_fused_actor<false, ActorF32Lean>(s, a, noise, ar);
    else                        launch_fused_actor<false, ActorF32>(s, a, noise, ar);
    return hipGetLastError();
}

__global__ __launch_bounds__(kBlock) void k_record(Batch b, const float* __restrict__ obs,
                                                   const float* __restrict__ act, StatsPtrs st, TrajPtrs traj) {
    const uint32_t i = env_index();
    if (i >= b.n) return;
    const size_t ld = b.ld, tt = traj.t0;
#pragma unroll
    for (int j = 0; j < 22; ++j) put<kNtTraj>(&traj.obs[(tt * 22 + j) * ld + i], field(obs, j, ld)[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) put<kNtTraj>(&traj.act[(tt * 4 + j) * ld + i], field(act, j, ld)[i]);
    traj.rew[tt * ld + i] = st.last_reward[i];
    traj.done[tt * ld + i] = st.last_done[i];
}

hipError_t launch_record(hipStream_t s, Batch b, const float* obs, const float* act, StatsPtrs st, TrajPtrs traj) {
    if (b.n == 0) return hipSuccess;
    k_record<<<grid_for(b.n, kBlock), kBlock, 0, s>>>(b, obs, act, st, traj);
    return hipGetLastError();
}

// ------------------------------------------------------------------ layout changes -----
// The boundary speaks row-major [n][dim] (NumPy), the engine field-major [dim][ld].  One wave moves a
// 64-env tile through LDS so that BOTH sides are touched in whole 256-byte lines: SoA side one field
// row per instruction, row-major side the tile's 64*dim contiguous floats.  kMaxDim bounds the LDS tile.
static constexpr int kMaxDim = 32;

// blockIdx.y = slab: slab s reads soa + s*dim*ld and writes rows + s*n*dim (the steps of a trajectory)
__global__ __launch_bounds__(64) void k_soa_to_rows(const float* __restrict__ soa, uint32_t ld, uint32_t dim,
                                                      uint32_t n, float* __restrict__ rows) {
    __shared__ float tile[64 * (kMaxDim + 1)];
    const uint32_t lane = threadIdx.x, base = blockIdx.x * 64u;
    soa += (size_t)blockIdx.y * dim * ld;
    rows += (size_t)blockIdx.y * n * dim;
    const uint32_t pitch = dim + 1;                       // odd pitch for dim = 4, 16, 26: no bank conflicts
    for (uint32_t f = 0; f < dim; ++f) tile[lane * pitch + f] = soa[(size_t)f * ld + base + lane];   // ld >= base + 64
    __syncthreads();
    const uint32_t count = (n - base < 64u ? n - base : 64u) * dim;
    float* out = rows + (size_t)base * dim;
    for (uint32_t k = lane; k < count; k += 64u) out[k] = tile[(k / dim) * pitch + (k % dim)];
}

// rows [n][stride] (first dim columns) -> SoA [dim][ld]; lanes n..ld-1 are zeroed
__global__ __launch_bounds__(64) void k_rows_to_soa(const float* __restrict__ rows, uint32_t stride, uint32_t dim,
                                                      uint32_t n, uint32_t ld, float* __restrict__ soa) {
    __shared__ float tile[64 * (kMaxDim + 1)];
    const uint32_t lane = threadIdx.x, base = blockIdx.x * 64u;
    const uint32_t pitch = dim + 1;
    const uint32_t envs = base < n ? (n - base < 64u ? n - base : 64u) : 0u;
    const float* in = rows + (size_t)base * stride;
    for (uint32_t k = lane; k < envs * stride; k += 64u) {
        const uint32_t e = k / stride, c = k % stride;
        if (c < dim) tile[e * pitch + c] = in[k];
    }
    __syncthreads();
    for (uint32_t f = 0; f < dim; ++f) soa[(size_t)f * ld + base + lane] = lane < envs ? tile[lane * pitch + f] : 0.0f;
}

hipError_t launch_soa_to_rows(hipStream_t s, const float* soa, uint32_t ld, uint32_t dim, uint32_t n, float* rows,
                              uint32_t slabs) {
    if (n == 0 || dim == 0 || slabs == 0) return hipSuccess;
    if (dim > (uint32_t)kMaxDim || slabs > 65535u) return hipErrorInvalidValue;
    k_soa_to_rows<<<dim3((n + 63u) / 64u, slabs), 64, 0, s>>>(soa, ld, dim, n, rows);
    return hipGetLastError();
}

hipError_t launch_rows_to_soa(hipStream_t s, const float* rows, uint32_t stride, uint32_t dim, uint32_t n, uint32_t ld,
                              float* soa) {
    if (ld == 0 || dim == 0) return hipSuccess;
    if (dim > (uint32_t)kMaxDim || stride < dim) return hipErrorInvalidValue;
    k_rows_to_soa<<<ld / 64u, 64, 0, s>>>(rows, stride, dim, n, ld, soa);
    return hipGetLastError();
}

hipError_t launch_fill_f32(hipStream_t s, float* p, float v, uint32_t count) {
    if (count == 0) return hipSuccess;
    k_fill_f32<<<grid_for(count, kBlock), kBlock, 0, s>>>(p, v, count);
    return hipGetLastError();
}

}  // namespace rq
