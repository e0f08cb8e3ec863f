// Registration::align's optimiser loop on the device, for every OptimizationMethod, with the robust-scale annealing of
// pipeline::RobustAligner around it (replaces registration.hpp:201-276, 803-828, 830-895, 897-964, dogleg_step.hpp:35-101 and
// pipeline/robust.hpp:78-111 for the prepared GICP / point-to-distribution path): ONE launch, ONE read-back per alignment.
//
// The reference drives LM and dog-leg from the host: per outer iteration one search + K11 (submit, wait, toCPU) and per trial
// step a solve on the host, a K12 launch with two freshly allocated USM vectors, another wait (registration.hpp:674-675,
// 685-686, 854). Its own example — LM, Geman-McClure, three annealing levels on a 1000-point random sample
// (example_registration.cpp:29-55) — is about sixty such round trips for work that one compute unit finishes in microseconds.
//
// gicp_optimize_kernel is a persistent launch of up to 256 workgroups (256 lanes each up to 64 K source points, 1024 beyond:
// see the note above the kernel). A STEP is a pass over the source:
//   linearise  fused_point of the per-iteration Gauss-Newton kernel (certificate -> cached correspondence, else exact NN on
//              the grid; 28 sums + inlier count)                                                -> one partial row per workgroup
//   trial      error_prepared_point: K12 at the trial pose over the cache rows (frozen correspondences) -> one partial row
// Between steps every workgroup waits for all rows (arrival counter sharded over 8 lines, one lane polls with sc1 loads and
// s_sleep, bounded by wall_clock64 — the hand-off of gicp_align_persistent_kernel), sums them in reduce_rows_1024's fixed
// order and lets ONE lane run the optimiser's state machine (opt_after_*) on LDS: every workgroup takes the same decisions from
// the same bits, so nothing is broadcast. A launch of ONE workgroup (up to 1024 points: the reference pipeline's default
// random sample) keeps its totals in LDS and touches no counter.
// Rows ping-pong by step parity: a workgroup can only write the row of step s + 1 after every workgroup has stored the row
// of step s, i.e. after every workgroup has finished reading the rows of step s - 1 that it overwrites.
#include <algorithm>

#include "registration_device.h"

namespace sp {
namespace {

// PHASE_FUSED (wave-per-point launches of 256-lane workgroups): a trial step that ALSO linearises at the trial pose, into the
// other set of cache rows — accepted trials are the rule, and the linearisation of the next outer iteration is then already
// there: an outer iteration costs one step (one hand-off between the workgroups) instead of two.
enum { PHASE_LIN = 0, PHASE_TRIAL = 1, PHASE_FUSED = 2 };

struct OptCtl {  // the optimiser's state between steps (LDS, identical in every workgroup)
    int phase;
    int done;             // 0 go on, 1 finished, 2 a wait ran out
    int level, iter, inner;
    int cache_valid;
    float lambda, radius;
    float cur_error, last_error;
    float predicted, step_norm;  // dog-leg step in flight
    int conv_ok;          // trial step: success ? is_converged(delta) : false   (registration.hpp:843-847)
    int conv_any;         // trial step: is_converged(delta)                      (:867, :878, :951)
    int converged;
    float res_error;
    unsigned res_inlier, res_iterations;
    unsigned n_lin, n_trial, searched, log_n;
    int took;             // the latest trial's pose became the pose
    int spec_level;       // PHASE_FUSED: the level whose robust scale the speculative linearisation uses
    int cur;              // which set of cache rows holds the correspondences of the latest linearisation (0: the source's own)
};

struct OptArgs {
    float* part[2];
    unsigned* tickets;
    unsigned long long* trows[2];  // WAVEQ: tagged partial rows, by step parity
    unsigned tag_base;             // WAVEQ: this launch's epoch << 20 (tags of earlier launches never match: no zeroing per launch)
    float4* rows[2];               // cache rows: [0] the source's own (what the launch leaves), [1] the other set of fused steps
    float4* qcert[2];              // WAVEQ: the margin certificates beside them (null: none)
    int qcert_trusted;             // ... and whether those of set 0 belong to its rows as the launch finds them
    int fuse;                      // trial steps also linearise (PHASE_FUSED)
    const float* T_init;          // device: initial guess
    float* T_out;                 // device: final pose (may alias T_init: it is read before anything is written)
    sp_opt_params opt;
    float scales[SP_OPT_MAX_LEVELS];
    int n_levels;
    int reuse;                    // later linearisations may trust the correspondence cache
    sp_align_result* result;
    unsigned long long budget;    // wall_clock64 ticks a wait may take
};

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return v < lo ? lo : (hi < v ? hi : v); }  // std::clamp

__device__ __forceinline__ bool is_converged6(const float* d, float crit_rot, float crit_trans) {  // registration.hpp:407-410
    const float nr = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const float nt = sqrtf(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    return nr < crit_rot && nt < crit_trans;
}

struct OptShared {
    float sT[16];      // current pose (result.T)
    float sTt[16];     // trial pose
    float sTlin[16];   // pose of the latest linearisation
    sp_linearized slin;  // system of the latest linearisation
    float sdelta[8];
    LdltScratch ldlt_ws;
    OptCtl ctl;
    // what the state machine reads of the launch's arguments (it is a real function: arguments passed by reference would have
    // to live in scratch memory for the whole kernel)
    sp_opt_params opt;
    int n_levels, reuse;
    sp_align_result* result;
};

// One LM trial: delta = LDLT(H + lambda I).solve(-b), T_trial = T exp(delta)   (registration.hpp:841-848)
__device__ __forceinline__ void lm_try(OptShared& S) {
    const sp_opt_params& o = S.opt;
#pragma unroll
    for (int i = 0; i < 16; ++i) S.sTt[i] = S.sT[i];
    gn_update_impl(&S.slin, S.sTt, S.ctl.lambda, o.crit_rotation, o.crit_translation, S.sdelta, false, S.ldlt_ws);
    S.ctl.conv_ok = S.sdelta[6] > 0.5f ? 1 : 0;
    S.ctl.conv_any = is_converged6(S.sdelta, o.crit_rotation, o.crit_translation) ? 1 : 0;
    S.ctl.phase = PHASE_TRIAL;
}

// An outer iteration has ended (`accepted`: sp_opt_log_entry::accepted): log it; converged or the last iteration ends the
// level (registration.hpp:266-268); after the last level the launch is done.
__device__ __forceinline__ void end_outer(OptShared& S, int accepted, unsigned trials, bool publish) {
    OptCtl& c = S.ctl;
    const sp_opt_params& o = S.opt;
    c.res_iterations = (unsigned)c.iter;
    if (publish && c.log_n < (unsigned)SP_OPT_LOG_ENTRIES) {
        sp_opt_log_entry e;
        e.level = (uint16_t)c.level; e.iteration = (uint16_t)c.iter; e.trials = (uint16_t)trials; e.accepted = (uint16_t)accepted;
        e.damping = o.method == SP_OPT_POWELL_DOGLEG ? c.radius : c.lambda;
        e.error = c.res_error;
        S.result->log[c.log_n] = e;
    }
    ++c.log_n;
    if (!c.converged && c.iter + 1 < o.max_iterations) {
        ++c.iter;
        c.phase = PHASE_LIN;
        c.cache_valid = S.reuse;
        return;
    }
    if (c.level + 1 < S.n_levels) {  // the next robust scale: a fresh align() from this pose (pipeline/robust.hpp:100-111)
        ++c.level;
        c.iter = 0;
        c.lambda = o.lm_init_lambda;
        c.radius = o.dl_initial_radius;
        c.converged = 0;
        c.res_error = FLT_MAX;  // RegistrationResult's defaults (result.hpp:12-28)
        c.res_inlier = 0;
        c.res_iterations = 0;
        c.phase = PHASE_LIN;
        c.cache_valid = S.reuse;  // certificates prove every reused correspondence: the same neighbours as a fresh search
        return;
    }
    c.done = 1;
}

// After a linearisation step: tot = 28 sums, the uint32 count, the searched count as a float value.
__device__ __forceinline__ void opt_after_linearize_impl(OptShared& S, const float* tot, bool publish) {
    OptCtl& c = S.ctl;
    unpack_totals(tot, kAcc - 1, &S.slin);
    ++c.n_lin;
    c.searched += (unsigned)tot[kAcc];
#pragma unroll
    for (int i = 0; i < 16; ++i) S.sTlin[i] = S.sT[i];
    const sp_opt_params& o = S.opt;
    if (o.method == SP_OPT_GAUSS_NEWTON) {  // registration.hpp:803-828
        gn_update_impl(&S.slin, S.sT, o.gn_lambda, o.crit_rotation, o.crit_translation, S.sdelta, false, S.ldlt_ws);
        c.converged = S.sdelta[6] > 0.5f ? 1 : 0;
        c.res_error = S.slin.error;
        c.res_inlier = S.slin.inlier;
        end_outer(S, 1, 0, publish);
    } else if (o.method == SP_OPT_LEVENBERG_MARQUARDT) {  // :830-895
        c.cur_error = S.slin.error;
        c.last_error = FLT_MAX;
        c.inner = 0;
        if (o.lm_max_inner_iterations <= 0) end_outer(S, 0, 0, publish);  // (no trial: result.converged stays false)
        else lm_try(S);
    } else {  // :897-964
        c.res_error = S.slin.error;
        c.res_inlier = S.slin.inlier;
        c.cur_error = S.slin.error;
        c.radius = clampf(c.radius, o.dl_min_radius, o.dl_max_radius);
        const DoglegStep6 dl = dogleg_step6(S.slin.H, S.slin.b, c.radius, S.ldlt_ws);
        if (dl.predicted_reduction <= 0.0f) {
            c.radius = clampf(c.radius * o.dl_gamma_decrease, o.dl_min_radius, o.dl_max_radius);
            end_outer(S, 0, 0, publish);
        } else {
            const Rigid upd = rigid_mul(load_rigid_colmajor(S.sT), se3_exp(dl.p));
            store_rigid_colmajor(upd, S.sTt);
            c.conv_any = is_converged6(dl.p, o.crit_rotation, o.crit_translation) ? 1 : 0;
            c.predicted = dl.predicted_reduction;
            c.step_norm = dl.step_norm;
            c.phase = PHASE_TRIAL;
        }
    }
}

// After a trial step: tot[0] = the robust error at the trial pose, tot[1] = the uint32 inlier count.
__device__ __forceinline__ void opt_after_trial_impl(OptShared& S, const float* tot, bool publish) {
    OptCtl& c = S.ctl;
    const sp_opt_params& o = S.opt;
    const float new_error = tot[0];
    const unsigned inl = __float_as_uint(tot[1]);
    ++c.n_trial;
    c.took = 0;
    auto take = [&] {
#pragma unroll
        for (int i = 0; i < 16; ++i) S.sT[i] = S.sTt[i];
        c.res_error = new_error;
        c.res_inlier = inl;
        c.took = 1;
    };
    if (o.method == SP_OPT_LEVENBERG_MARQUARDT) {
        const unsigned tries = (unsigned)c.inner + 1u;
        if (new_error <= c.cur_error) {  // :866-876
            c.converged = c.conv_any;
            take();
            c.lambda = clampf(c.lambda / o.lm_lambda_factor, o.lm_min_lambda, o.lm_max_lambda);
            end_outer(S, 1, tries, publish);
        } else if (fabsf(new_error - c.last_error) <= 1e-6f) {  // :877-884
            c.converged = c.conv_any;
            take();
            end_outer(S, 2, tries, publish);
        } else {  // :885-889
            c.lambda = clampf(c.lambda * o.lm_lambda_factor, o.lm_min_lambda, o.lm_max_lambda);
            c.last_error = new_error;
            ++c.inner;
            if (c.inner < o.lm_max_inner_iterations) {
                lm_try(S);
            } else {
                c.converged = c.conv_ok;  // what the last trial left in result.converged (:843-847)
                end_outer(S, 0, tries, publish);
            }
        }
    } else {  // dog-leg (:936-962)
        const float rho = (c.cur_error - new_error) / c.predicted;
        if (rho < o.dl_eta1) {
            c.radius = clampf(c.radius * o.dl_gamma_decrease, o.dl_min_radius, o.dl_max_radius);
            end_outer(S, 0, 1, publish);
        } else {
            c.converged = c.conv_any;
            take();
            if (rho > o.dl_eta2 && c.step_norm >= c.radius * 0.99f)
                c.radius = clampf(c.radius * o.dl_gamma_increase, o.dl_min_radius, o.dl_max_radius);
            end_outer(S, 1, 1, publish);
        }
    }
}

// The state machine as real functions (the 1024-lane instantiations: inlined it takes the point loops' registers with it, 90
// spilled VGPRs) and inlined (the 256-lane instantiations have 256 registers: 5.5 -> 1.5 us per linearisation step).
// After a fused step: tot[0..29] the linearisation at the trial pose (sums | count | searched), tot[30] / tot[31] the trial's
// robust error / inlier count. The trial is decided exactly as a plain trial step decides it; when its pose was taken and what
// follows is the linearisation that was speculated on (same pose, the level whose scale it used), that linearisation is adopted
// as if its step had just run: the same sequence of state-machine calls as the unfused loop, one hand-off less.
__device__ __forceinline__ void opt_after_fused_impl(OptShared& S, const float* tot, bool publish) {
    OptCtl& c = S.ctl;
    const float trial[2] = {tot[30], tot[31]};
    opt_after_trial_impl(S, trial, publish);
    if (!c.done && c.took && c.phase == PHASE_LIN && c.level == c.spec_level) {
        c.cur ^= 1;  // the rows the speculative linearisation wrote are the correspondences now
        opt_after_linearize_impl(S, tot, publish);
    }
}
// A trial is due: make it a fused step when there is a linearisation to speculate on (an accepted trial that ends the last level
// is followed by nothing).
__device__ __forceinline__ void opt_upgrade_trial(OptShared& S, int fuse) {
    OptCtl& c = S.ctl;
    if (!fuse || c.done || c.phase != PHASE_TRIAL) return;
    const int next = (c.conv_any || c.iter + 1 >= S.opt.max_iterations) ? c.level + 1 : c.level;
    if (next < S.n_levels) { c.phase = PHASE_FUSED; c.spec_level = next; }
}
__device__ __noinline__ void opt_after_linearize_call(OptShared& S, const float* tot, bool publish) { opt_after_linearize_impl(S, tot, publish); }
__device__ __noinline__ void opt_after_trial_call(OptShared& S, const float* tot, bool publish) { opt_after_trial_impl(S, tot, publish); }

// The results (workgroup 0): RegistrationResult of the last level + the linearisation pose + counters.
__device__ __forceinline__ void opt_publish(float* T_out, const OptShared& S) {
    sp_align_result* const r = S.result;
    const OptCtl& c = S.ctl;
    const unsigned t = threadIdx.x;
    if (t < 16) { r->T[t] = S.sT[t]; T_out[t] = S.sT[t]; }
    else if (t < 32) r->T_lin[t - 16] = S.sTlin[t - 16];
    else if (t >= 64 && t < 100) r->H[t - 64] = S.slin.H[t - 64];
    else if (t >= 128 && t < 134) r->b[t - 128] = S.slin.b[t - 128];
    else if (t == 192) {
        r->error = c.res_error;
        r->error_raw = S.slin.error;
        r->inlier = c.res_inlier;
        r->iterations = c.res_iterations;
        r->converged = (unsigned)c.converged;
        r->status = 0u;
        r->linearizations = c.n_lin;
        r->trials = c.n_trial;
        r->searched = c.searched;
        r->damping = S.opt.method == SP_OPT_POWELL_DOGLEG ? c.radius : c.lambda;
        r->log_entries = c.log_n < (unsigned)SP_OPT_LOG_ENTRIES ? c.log_n : (unsigned)SP_OPT_LOG_ENTRIES;
        r->pad[1] = r->pad[2] = 0u;  // (pad[0]: the done flag, stored last by the caller of this function)
    }
}

// BLOCK: lanes per workgroup. The search of a linearisation is bound by vector issue (thousands of wave instructions per 64
// points), so a SMALL source wants its waves on many compute units, each wave alone on its SIMD — workgroups of 256 lanes —
// and pays for it with the arrival counter between steps; a large source fills every compute unit with 1024 lanes anyway.
// The hand-off between two steps of a wave-per-point launch (up to 256 small workgroups, a step of a few microseconds: the
// counter hand-off — drain the row's stores, count in, poll the counter, read all rows: three dependent trips to memory —
// costs more than the step). Here a row is 32 granules {value, tag = step + 1} and each granule travels in ONE 8-byte store
// (the exchange slots' trick, comm.hip): a reader polls the rows themselves, every lane its share of granules, all loads of a
// round in flight together; a round in which every tag matches IS the data. Rows ping-pong by step parity as the counter form's
// do. A tag is (epoch of the launch) << 20 | (step + 1) mod 2^20: the buffer is zeroed when the source is created and when the
// 12-bit epoch wraps, never per launch (a row is rewritten by its owner every second step, so a stale tag is at most two steps old).
// Sums in a fixed order (lane (group, slot): rows group, group + G, ... in order; then the groups in order): the same bits in
// every workgroup. On return red[0][e] holds the totals (slot nv the uint32 count); false: the wait ran out.
template <int BLOCK>
__device__ __forceinline__ bool tagged_rows_exchange(unsigned long long* rows, unsigned tag, float mine, unsigned grid, int nv,
                                                     float (*red)[kPartial], unsigned long long budget, unsigned* s_wait,
                                                     unsigned more_counts = 0u) {
    constexpr unsigned kGroups = BLOCK / 32;
    constexpr int kPer = kAlignMaxBlocks / (int)kGroups;
    const unsigned e = threadIdx.x & 31u, grp = threadIdx.x >> 5;
    if (threadIdx.x < 32)
        __hip_atomic_store(rows + (size_t)blockIdx.x * 32 + e, ((unsigned long long)tag << 32) | __float_as_uint(mine),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool is_count = (int)e == nv || ((more_counts >> e) & 1u) != 0u;  // (slots summed as uint32 counts)
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        unsigned long long g[kPer];
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const unsigned r = grp + (unsigned)j * kGroups;
            g[j] = r < grid ? __hip_atomic_load(rows + (size_t)r * 32 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                            : ((unsigned long long)tag << 32);
        }
        bool ok = true;
        float sum = 0.0f;
        unsigned c = 0;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            ok = ok && (unsigned)(g[j] >> 32) == tag;
            const unsigned bits = (unsigned)g[j];
            if (is_count) c += bits;
            else sum += __uint_as_float(bits);
        }
        if (__syncthreads_and(ok ? 1 : 0)) {
            red[grp][e] = is_count ? __uint_as_float(c) : sum;
            break;
        }
        if (threadIdx.x == 0) *s_wait = (wall_clock64() - t0 > budget) ? 2u : 0u;
        __syncthreads();
        if (*s_wait == 2u) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        float t = 0.0f;
        unsigned ct = 0;
#pragma unroll
        for (unsigned p = 0; p < kGroups; ++p) {
            const float v = red[p][e];
            if (is_count) ct += __float_as_uint(v);
            else t += v;
        }
        red[0][e] = is_count ? __uint_as_float(ct) : t;
    }
    __syncthreads();
    return true;
}
// A workgroup's totals of a wave-per-point step: every lane of a wave holds the wave's sums, so lane 0 of each wave hands them
// over and lane e < 32 returns slot e of the workgroup's row (sums | uint32 count bits | searched as a float value | zeros).
// A fused step (NV = 28) also hands over its trial: slot 30 the robust error (a float sum), slot 31 the inlier count (uint32 bits).
template <int NV, int BLOCK>
__device__ __forceinline__ float waveq_row_slot(const float (&acc)[NV], unsigned cnt, unsigned extra, float (*red)[kPartial],
                                                float trial_error = 0.0f, unsigned trial_cnt = 0u) {
    const unsigned lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) {
#pragma unroll
        for (int e = 0; e < NV; ++e) red[wave][e] = acc[e];
        red[wave][NV] = __uint_as_float(cnt);
        red[wave][NV + 1] = __uint_as_float(extra);
        if (NV + 3 < kPartial) {
            red[wave][kPartial - 2] = trial_error;
            red[wave][kPartial - 1] = __uint_as_float(trial_cnt);
        }
    }
    __syncthreads();
    float v = 0.0f;
    if ((int)threadIdx.x < NV || (NV + 3 < kPartial && (int)threadIdx.x == kPartial - 2)) {
#pragma unroll
        for (int w = 0; w < BLOCK / kWave; ++w) v += red[w][threadIdx.x];
    } else if ((int)threadIdx.x == NV || (int)threadIdx.x == NV + 1 || (NV + 3 < kPartial && (int)threadIdx.x == kPartial - 1)) {
        unsigned c = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / kWave; ++w) c += __float_as_uint(red[w][threadIdx.x]);
        v = (int)threadIdx.x == NV + 1 ? (float)c : __uint_as_float(c);
    }
    return v;
}

// WAVEQ: one wave per source point in the linearisation steps (fused_query_wave) — sources of up to kWaveQueryMax points, spread
// over up to 256 workgroups of four waves: the step then costs a handful of dependent round trips instead of the longest
// per-lane walk through a 2x2x2 block of a surface cloud (the reference's example, 1000 points against 6 k: 16-26 us per
// linearisation per lane, a few us per wave).
// By itself (sp_gicp_source_set_wave_per_point = 1, the default) the launch takes this form up to kWaveQueryMax points, where
// every point has a wave to itself. Told to (= 2: the caller knows the target's cells are crowded — sp_grid_max_cell_points — a
// raw LiDAR scan with thousands of returns in one cell) up to kWaveQueryForcedMax: a wave then walks through n / 4096 points one
// after the other, which loses to a lane per point on a cloud of even density (20 k points in a filled box: 67 against 21 us
// per iteration) and wins where a lane's walk through its 2x2x2 block is thousands of candidates long (the reference's bundled
// scans at full resolution, 5032 returns in the sensor's own cell: 212 against 686 us; scratch/waveq_crossover.py).
constexpr size_t kWaveQueryMax = 2048, kWaveQueryForcedMax = 131072;
template <int LOSS, bool FAST_NN, bool P2D, int BLOCK, bool WAVEQ = false>
__global__ __launch_bounds__(BLOCK) void gicp_optimize_kernel(FusedParams P, OptArgs A) {
    __shared__ OptShared S;
    __shared__ float red[BLOCK / 32][kPartial];
    __shared__ unsigned s_wait;
    const unsigned stride = gridDim.x * BLOCK;
    unsigned tile = blockIdx.x;
    if ((gridDim.x & 7u) == 0u) tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const bool single = gridDim.x == 1;
    const bool publish = blockIdx.x == 0;
    if (threadIdx.x < 16) {
        S.sT[threadIdx.x] = A.T_init[threadIdx.x];
        S.sTlin[threadIdx.x] = A.T_init[threadIdx.x];
    } else if (threadIdx.x == 32) {
        OptCtl c;
        c.phase = PHASE_LIN; c.done = 0; c.level = 0; c.iter = 0; c.inner = 0;
        c.cache_valid = P.cache_valid;
        c.lambda = A.opt.lm_init_lambda; c.radius = A.opt.dl_initial_radius;
        c.cur_error = 0.0f; c.last_error = FLT_MAX; c.predicted = 0.0f; c.step_norm = 0.0f;
        c.conv_ok = 0; c.conv_any = 0; c.converged = 0;
        c.res_error = FLT_MAX; c.res_inlier = 0; c.res_iterations = 0;
        c.n_lin = 0; c.n_trial = 0; c.searched = 0; c.log_n = 0;
        c.took = 0; c.spec_level = 0; c.cur = 0;
        S.ctl = c;
        S.opt = A.opt;
        S.n_levels = A.n_levels;
        S.reuse = A.reuse;
        S.result = A.result;
    } else if (threadIdx.x >= 64 && threadIdx.x < 64 + 48) {
        reinterpret_cast<float*>(&S.slin)[threadIdx.x - 64] = 0.0f;
    }
    __syncthreads();
#ifdef SP_OPT_TIMING
    unsigned long long wg_t0 = 0;
#define SP_WG_BEGIN() if (threadIdx.x == 0) wg_t0 = wall_clock64()
#define SP_WG_END() if (threadIdx.x == 0 && step < 24 && blockIdx.x < 8) g_sp_wg[step * 8 + blockIdx.x] = wall_clock64() - wg_t0
#else
#define SP_WG_BEGIN()
#define SP_WG_END()
#endif
#ifdef SP_OPT_TIMING  // development builds: wall_clock64 stamps of the first steps into the tail of the result's log
    unsigned long long* const stamps = reinterpret_cast<unsigned long long*>(&A.result->log[8]);
#define SP_STAMP(k) if (publish && threadIdx.x == 0 && step < 20) stamps[step * 5 + (k)] = wall_clock64()
#else
#define SP_STAMP(k)
#endif
    for (unsigned step = 0;; ++step) {
        const int phase = S.ctl.phase;  // uniform over the grid
#ifdef SP_OPT_TIMING
        if (publish && threadIdx.x == 0) g_sp_step = step;
        __syncthreads();
#endif
        SP_STAMP(0);
        SP_WG_BEGIN();
        float* const row = A.part[step & 1] + (size_t)blockIdx.x * kPartial;
        [[maybe_unused]] float mine = 0.0f;  // WAVEQ: slot threadIdx.x < 32 of this workgroup's row
        if constexpr (WAVEQ) P.ccache = A.rows[S.ctl.cur];  // (uniform; the other kinds of launch never leave set 0)
        if (phase == PHASE_LIN) {
            P.scale = A.scales[S.ctl.level];
            P.cache_valid = S.ctl.cache_valid;
            const Rigid T = uniform_pose(S.sT);
            float acc[kAcc - 1];
            unsigned cnt = 0, searched = 0;
#pragma unroll
            for (int e = 0; e < kAcc - 1; ++e) acc[e] = 0.0f;
            if constexpr (WAVEQ) {
                const unsigned nw = BLOCK / kWave;
                const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
                // (rows that another kind of launch refreshed carry no margin certificates of their own: the launch's first
                // linearisation only writes them)
                const bool qc_trust = A.qcert_trusted || S.ctl.n_lin > 0;
                for (unsigned i = blockIdx.x * nw + wave; i < P.n; i += gridDim.x * nw)
                    fused_query_wave<LOSS, P2D>(P, T, i, acc, cnt, searched, nullptr, qc_trust ? A.qcert[S.ctl.cur] : nullptr, A.qcert[S.ctl.cur]);
                SP_STAMP(4);
                mine = waveq_row_slot<kAcc - 1, BLOCK>(acc, cnt, searched, red);
            } else if constexpr (FAST_NN && BLOCK <= 256) {  // (the staged search: open queries are finished by the whole wave)
                for (unsigned b = tile * BLOCK; b < P.n; b += stride)
                    fused_point_wave<LOSS, P2D>(P, T, b + threadIdx.x, b + threadIdx.x < P.n, acc, cnt, searched);
            } else {  // (1024 lanes: 128 registers — the per-lane stages of the per-iteration kernel fit them, the wave's tail does not)
                for (unsigned i = tile * BLOCK + threadIdx.x; i < P.n; i += stride)
                    fused_point<LOSS, FAST_NN, P2D, kSeedSearches, kNegCert>(P, T, i, acc, cnt, searched);
            }
            if constexpr (!WAVEQ) {
                SP_STAMP(4);
                if (single) block_reduce_lds<kAcc - 1, BLOCK>(acc, cnt, searched, red[0]);
                else block_reduce_store<kAcc - 1, BLOCK, true>(acc, cnt, row, false, searched);
            }
        } else if (WAVEQ && phase == PHASE_FUSED) {
            if constexpr (WAVEQ) {
                // the trial (frozen correspondences of the current rows, this level's scale) and the linearisation at its pose
                // (fresh correspondences into the OTHER rows, the scale of the level that would follow) in one pass
                const Rigid T = uniform_pose(S.sTt);
                const Rigid TL = uniform_pose(S.sTlin);
                float acc[kAcc - 1], acc_t[1] = {0.0f};
                unsigned cnt = 0, searched = 0, cnt_t = 0;
#pragma unroll
                for (int e = 0; e < kAcc - 1; ++e) acc[e] = 0.0f;
                float4* const rows_out = A.rows[S.ctl.cur ^ 1];
                const float scale_trial = A.scales[S.ctl.level], scale_lin = A.scales[S.ctl.spec_level];
                const int reuse = S.reuse;
                const unsigned nw = BLOCK / kWave;
                const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
                for (unsigned i = blockIdx.x * nw + wave; i < P.n; i += gridDim.x * nw) {
                    P.scale = scale_trial;
                    error_prepared_point<LOSS, P2D>(P, T, TL, i, acc_t, cnt_t);
                    P.scale = scale_lin;
                    P.cache_valid = reuse;
                    fused_query_wave<LOSS, P2D>(P, T, i, acc, cnt, searched, rows_out, A.qcert[S.ctl.cur], A.qcert[S.ctl.cur ^ 1]);
                }
                SP_STAMP(4);
                mine = waveq_row_slot<kAcc - 1, BLOCK>(acc, cnt, searched, red, acc_t[0], cnt_t);
            }
        } else {
            const Rigid T = uniform_pose(S.sTt);
            const Rigid TL = uniform_pose(S.sTlin);
            P.scale = A.scales[S.ctl.level];
            float acc[1] = {0.0f};
            unsigned cnt = 0;
            if constexpr (WAVEQ) {
                // the wave that searched for a point also evaluates it: a cache row is only ever touched by one wave of the launch
                // (rows written on one XCD are not visible through another XCD's L2 before the launch ends)
                const unsigned nw = BLOCK / kWave;
                const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
                for (unsigned i = blockIdx.x * nw + wave; i < P.n; i += gridDim.x * nw)
                    error_prepared_point<LOSS, P2D>(P, T, TL, i, acc, cnt);
                mine = waveq_row_slot<1, BLOCK>(acc, cnt, 0u, red);
            } else {
                for (unsigned i = tile * BLOCK + threadIdx.x; i < P.n; i += stride)
                    error_prepared_point<LOSS, P2D>(P, T, TL, i, acc, cnt);
                if (single) block_reduce_lds<1, BLOCK>(acc, cnt, 0u, red[0]);
                else block_reduce_store<1, BLOCK, true>(acc, cnt, row, false, 0u);
            }
        }
        SP_STAMP(1);
        SP_WG_END();
        auto wait_ran_out = [&] {  // (uniform per workgroup; every workgroup runs into the same bound) — loud: NaN pose, status 2
            if (publish) {
                if (threadIdx.x < 16) { A.T_out[threadIdx.x] = __int_as_float(0x7fc00000); A.result->T[threadIdx.x] = __int_as_float(0x7fc00000); }
                if (threadIdx.x == 16) { A.result->status = 2u; A.result->converged = 0u; A.result->log_entries = 0u; }
                __syncthreads();
                if (threadIdx.x == 0) {
                    __threadfence_system();
                    __hip_atomic_store(&A.result->pad[0], (unsigned)SP_ALIGN_RESULT_DONE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        };
        if constexpr (WAVEQ) {
            if (single) {
                __syncthreads();
                if (threadIdx.x < 32) red[0][threadIdx.x] = mine;
                __syncthreads();
            } else if (!tagged_rows_exchange<BLOCK>(A.trows[step & 1], A.tag_base | ((step + 1u) & 0xfffffu), mine, gridDim.x,
                                                    phase == PHASE_TRIAL ? 1 : kAcc - 1, red, A.budget, &s_wait,
                                                    phase == PHASE_FUSED ? 1u << (kPartial - 1) : 0u)) {
                wait_ran_out();
                return;
            }
        } else if (!single) {
            // the hand-off of gicp_align_persistent_kernel: wave 0 (the storing lanes) drains its sc1 stores, lane 0 signals
            if (threadIdx.x < kWave) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (threadIdx.x == 0) {
                    __hip_atomic_fetch_add(A.tickets + (blockIdx.x & (kTicketShards - 1)) * kTicketStride, 1u, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned want = gridDim.x * (step + 1u);
                    const unsigned long long t0 = wall_clock64();
                    unsigned flag = 0;
                    for (;;) {
                        unsigned have = 0;
#pragma unroll
                        for (int sh = 0; sh < kTicketShards; ++sh)
                            have += __hip_atomic_load(A.tickets + sh * kTicketStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (have >= want) break;
                        if (wall_clock64() - t0 > A.budget) { flag = 2; break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    s_wait = flag;
                }
            }
            __syncthreads();
            if (s_wait == 2) {
                wait_ran_out();
                return;
            }
            reduce_rows_block<BLOCK, true>(A.part[step & 1], gridDim.x, phase == PHASE_LIN ? kAcc - 1 : 1, red);
        }
        SP_STAMP(2);
        if (threadIdx.x == 0) {
            if constexpr (BLOCK <= 256) {
                if (phase == PHASE_LIN) opt_after_linearize_impl(S, red[0], publish);
                else if (WAVEQ && phase == PHASE_FUSED) opt_after_fused_impl(S, red[0], publish);
                else opt_after_trial_impl(S, red[0], publish);
                if constexpr (WAVEQ) opt_upgrade_trial(S, A.fuse);
            } else {
                if (phase == PHASE_LIN) opt_after_linearize_call(S, red[0], publish);
                else opt_after_trial_call(S, red[0], publish);
            }
        }
        __syncthreads();
        SP_STAMP(3);
        if (S.ctl.done) {
#ifdef SP_OPT_TIMING
            if (publish) g_sp_step = 0;
#endif
            if (publish) {
                opt_publish(A.T_out, S);
                // the block is complete: the flag goes last, behind a system-scope release — result_device may be host-mapped memory
                // whose pad[0] the caller cleared and spins on (no read-back copy, no synchronisation)
                __syncthreads();
                if (threadIdx.x == 0) {
                    __threadfence_system();
                    __hip_atomic_store(&A.result->pad[0], (unsigned)SP_ALIGN_RESULT_DONE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            if constexpr (WAVEQ) {
                if (S.ctl.cur != 0) {  // the launch leaves its correspondences in the source's own rows: every wave copies its points'
                    const unsigned nw = BLOCK / kWave;
                    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
                    const unsigned lane = threadIdx.x & (kWave - 1);
                    for (unsigned i = blockIdx.x * nw + wave; i < P.n; i += gridDim.x * nw)
                        if (lane < 3) A.rows[0][3 * (size_t)i + lane] = A.rows[1][3 * (size_t)i + lane];
                        else if (lane == 3 && A.qcert[0]) A.qcert[0][i] = A.qcert[1][i];
                }
            }
            return;
        }
    }
}

}  // namespace
}  // namespace sp

#ifdef SP_OPT_TIMING
extern "C" int sp_internal_opt_debug(unsigned long long* out384) {
    return hipMemcpyFromSymbol(out384, HIP_SYMBOL(sp::g_sp_dbg), 24 * 16 * 8) == hipSuccess ? 0 : 3;
}
extern "C" int sp_internal_opt_debug_wg(unsigned long long* out192) {
    return hipMemcpyFromSymbol(out192, HIP_SYMBOL(sp::g_sp_wg), 24 * 8 * 8) == hipSuccess ? 0 : 3;
}
#endif
extern "C" int sp_gicp_align_optimize(const sp_gicp_target* target, const sp_gicp_source* source, float* transT_device,
                                     const sp_factor_params* params, const sp_opt_params* opt, const float* robust_scales,
                                     int n_levels, sp_align_result* result_device, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    using namespace sp;
    hipStream_t st = as_stream(stream);
    if (!target || !source || !params || !opt || !transT_device || !result_device || !robust_scales) return SP_ERR_INVALID_ARGUMENT;
    if (n_levels < 1 || n_levels > SP_OPT_MAX_LEVELS) {
        sp_set_error("[sp_gicp_align_optimize] n_levels must be in 1..SP_OPT_MAX_LEVELS");
        return SP_ERR_INVALID_ARGUMENT;
    }
    if (opt->method != SP_OPT_GAUSS_NEWTON && opt->method != SP_OPT_LEVENBERG_MARQUARDT && opt->method != SP_OPT_POWELL_DOGLEG) {
        sp_set_error("[sp_gicp_align_optimize] unknown optimization method");
        return SP_ERR_INVALID_ARGUMENT;
    }
    if (opt->max_iterations < 1 || opt->max_iterations > 65535) {
        sp_set_error("[sp_gicp_align_optimize] max_iterations must be in 1..65535 (0 iterations: the result is the initial guess)");
        return SP_ERR_INVALID_ARGUMENT;
    }
    if (const int rc = check_prepared_reg("align_optimize", target, params); rc != SP_OK) return rc;
    const size_t n = source->n;
    if (n == 0) {
        sp_set_error("[sp_gicp_align_optimize] empty source (Registration::align returns the initial guess)");
        return SP_ERR_INVALID_ARGUMENT;
    }
    if (!workspace || workspace_bytes < sp_gicp_workspace_bytes(n)) {
        sp_set_error("[Registration] workspace too small (sp_gicp_workspace_bytes)");
        return SP_ERR_INVALID_ARGUMENT;
    }
    if (source->opt_persistent == 0) {
        sp_set_error("[sp_gicp_align_optimize] not available: persistent launches are switched off for this source "
                     "(sp_gicp_source_set_persistent)");
        return SP_ERR_RUNTIME;
    }
    // workgroups of 256 lanes while they all fit the device one per compute unit (64 K points), of 1024 beyond
    const bool small = n <= (size_t)256 * 256;
    const unsigned block = small ? 256u : (unsigned)kAlignBlock;
    const bool fast = source->opt_fast_nn < 0 ? source->sorted : (source->opt_fast_nn != 0);
    const bool waveq = fast && ((source->opt_wave_query == 1 && n <= kWaveQueryMax) || (source->opt_wave_query == 2 && n <= kWaveQueryForcedMax));
    // (wave per point: four waves a workgroup, each alone on its SIMD with 256 registers, up to two points a wave; beyond, sixteen
    // waves a workgroup: a wave then walks through its points one after the other and the other waves of its SIMD hide its round
    // trips — 212 against 404 us per iteration on the reference's bundled scans)
    const unsigned wq_block = n <= kWaveQueryMax ? 256u : (unsigned)kAlignBlock;
    const unsigned wq_waves = wq_block / kWave;
    const unsigned grid = std::min<unsigned>((unsigned)(waveq ? (n + wq_waves - 1) / wq_waves : (n + block - 1) / block), (unsigned)kAlignMaxBlocks);
    PersistGuard* const guard = persist_acquire(st, grid);
    if (!guard) {
        sp_set_error("[sp_gicp_align_optimize] not available now: the launch cannot be resident (grid larger than the device, stream "
                     "capturing, or another stream's persistent launch still running)");
        return SP_ERR_RUNTIME;
    }
    struct Release {
        PersistGuard* g; hipStream_t st;
        ~Release() { persist_release(g, st); }
    } release{guard, st};
    OptArgs A;
    float* const rows = static_cast<float*>(workspace);
    A.part[0] = rows;
    A.part[1] = rows + (size_t)kAlignMaxBlocks * kPartial;
    A.tickets = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + kTicketOffsetBytes);
    A.trows[0] = source->opt_rows;
    A.trows[1] = source->opt_rows + (size_t)kAlignMaxBlocks * 32;
    A.rows[0] = source->ccache;
    A.rows[1] = source->ccache2;
    // (the margin certificates: with the reuse of correspondences on, for sources the wave-per-point form serves)
    const bool margin_certs = waveq && source->opt_reuse != 0 && source->qcert != nullptr && n <= source->qcert_points;
    A.qcert[0] = margin_certs ? source->qcert : nullptr;
    A.qcert[1] = margin_certs ? source->qcert2 : nullptr;
    A.qcert_trusted = (margin_certs && source->qcert_valid && source->cache_valid) ? 1 : 0;
    // (trial steps that also linearise: the wave-per-point launches of 256-lane workgroups, where a step is a hand-off between
    // hundreds of workgroups around two microseconds of work)
    A.fuse = (waveq && wq_block == 256u && source->ccache2 != nullptr && source->opt_fuse_trials && (!margin_certs || source->qcert2 != nullptr)) ? 1 : 0;
    A.tag_base = 0;
    if (grid > 1 && waveq) {
        // a row's tag is (epoch of the launch, step): rows left by earlier launches never match, so nothing is zeroed per launch —
        // only when the 12-bit epoch wraps (the buffer is zero when the source is created)
        if (++source->opt_epoch >= 4096u) {
            if (zero_async(source->opt_rows, kOptRowsBytes, st) != SP_OK) return SP_ERR_HIP;
            source->opt_epoch = 1;
        }
        A.tag_base = source->opt_epoch << 20;
    } else if (grid > 1 && zero_async(A.tickets, kTicketShards * kTicketStride * sizeof(unsigned), st) != SP_OK) return SP_ERR_HIP;
    target->note(st);
    FusedParams P = make_fused_params(target, source, params, transT_device, 1, nullptr, nullptr);
    A.T_init = transT_device;
    A.T_out = transT_device;
    A.opt = *opt;
    for (int l = 0; l < SP_OPT_MAX_LEVELS; ++l) A.scales[l] = robust_scales[l < n_levels ? l : n_levels - 1];
    A.n_levels = n_levels;
    A.reuse = (source->opt_reuse && P.ccache != nullptr) ? 1 : 0;
    A.result = result_device;
    A.budget = 50ull * 100000ull;  // 50 ms of wall_clock64 (100 MHz) per wait
    if (P.ccache == nullptr) {  // (a target without certificates has no cache rows: the trial steps read them)
        sp_set_error("[sp_gicp_align_optimize] the prepared target has no reuse certificates");
        return SP_ERR_INVALID_ARGUMENT;
    }
    const bool p2d = params->reg_type == SP_REG_POINT_TO_DISTRIBUTION;
#define SP_LAUNCH_OPT2(L, B)                                                                        \
    if (fast && p2d) gicp_optimize_kernel<L, true, true, B><<<grid, B, 0, st>>>(P, A);              \
    else if (fast) gicp_optimize_kernel<L, true, false, B><<<grid, B, 0, st>>>(P, A);               \
    else if (p2d) gicp_optimize_kernel<L, false, true, B><<<grid, B, 0, st>>>(P, A);                \
    else gicp_optimize_kernel<L, false, false, B><<<grid, B, 0, st>>>(P, A)
#define SP_LAUNCH_OPT(L)                                                                           \
    if (waveq && wq_block == 256u && p2d) gicp_optimize_kernel<L, true, true, 256, true><<<grid, 256, 0, st>>>(P, A);  \
    else if (waveq && wq_block == 256u) gicp_optimize_kernel<L, true, false, 256, true><<<grid, 256, 0, st>>>(P, A);   \
    else if (waveq && p2d) gicp_optimize_kernel<L, true, true, kAlignBlock, true><<<grid, kAlignBlock, 0, st>>>(P, A);  \
    else if (waveq) gicp_optimize_kernel<L, true, false, kAlignBlock, true><<<grid, kAlignBlock, 0, st>>>(P, A);       \
    else if (small) { SP_LAUNCH_OPT2(L, 256); }                                                    \
    else { SP_LAUNCH_OPT2(L, kAlignBlock); }
    switch (params->robust_type) {
        case SP_LOSS_NONE: SP_LAUNCH_OPT(LOSS_NONE); break;
        case SP_LOSS_HUBER: SP_LAUNCH_OPT(LOSS_HUBER); break;
        case SP_LOSS_TUKEY: SP_LAUNCH_OPT(LOSS_TUKEY); break;
        case SP_LOSS_CAUCHY: SP_LAUNCH_OPT(LOSS_CAUCHY); break;
        case SP_LOSS_GEMAN_MCCLURE: SP_LAUNCH_OPT(LOSS_GEMAN_MCCLURE); break;
        default: sp_set_error("[Registration::dispatch] Combination not found in tags!"); return SP_ERR_RUNTIME;
    }
#undef SP_LAUNCH_OPT2
#undef SP_LAUNCH_OPT
    source->cache_valid = true;
    source->qcert_valid = margin_certs;
    return launch_status();
}
