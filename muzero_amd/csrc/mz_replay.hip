// Replay-side training targets for a batch of finished games, straight from the padded history block the
// self-play loop records: the step immediately after the hot path.
//
// What this replaces in the reference (file:line relative to /root/reference):
//   ReplayBuffer.compute_target_value     replay_buffer.py:66-73   n-step value target, fp64, left-to-right sum
//   ReplayBuffer.preprocess_history       replay_buffer.py:75-99   padding past the terminal state + inv_to_scalar
//   ScalarConfig.inv_to_scalar            config.py:15-36          trailing axis of 1, or the two-hot support
//   scalar_to_support                     utils.py:138-143         float32, sqrt transform
//
// Layout: histories are move-major [move][game] (selfplay.BatchedHistory), so the lanes of a wavefront (consecutive
// games) read and write consecutive addresses.  One lane per (position, game).  HBM-bound: every input element is
// read once by the position that owns it plus up to td_steps rewards per value target (L2-resident re-reads).
// Numerics: Python float arithmetic restated in fp64 with -ffp-contract=off; disc**i comes from a host table
// (Python's pow need not equal the device's); float32 casts round to nearest like tf.constant(dtype=float32).
#include "mz_common.h"
#include <cmath>

namespace {

constexpr int kReplayThreads = 256;

struct ReplayDev {
    int G, M, A, K, td, value_support, reward_support;
    float value_lo, value_rng, reward_lo, reward_rng;   // (tensor - a) / (b - a): a and (b - a) rounded to float32
    const int* lengths;
    const int* actions;            // [M][G]
    const float* rewards;          // [M][G]
    const double* root_values;     // [M][G]
    const float* policies;         // [M][G][A]
    const int* random_actions;     // [G][K]
    const double* powers;          // disc**i, i = 0 .. max(td, M)
    int* ext_actions;              // [M+K][G]
    float* target_rewards;         // [M+K][G][Dr]
    float* target_values;          // [M+K+1][G][Dv]
    float* target_policies;        // [M+K+1][G][A]
    double* total_value;           // [G]
};

// inv_to_scalar of one value into `out` (D floats): D == 1 -> the value; else the two-hot support (utils.py:138-143)
__device__ void store_scalar(float v, int support, float lo, float rng, float* out) {
    if (support <= 0) {
        out[0] = v;
        return;
    }
    const float x = (v - lo) / rng;                                              // config.py:36
    const float y = (float)support * (sqrtf(x + 1.f) - 1.f) / (sqrtf(2.f) - 1.f);
    const float low = floorf(y);
    const int hi_i = (int)ceilf(y), lo_i = (int)low;
    const float p = y - low;
    for (int i = 0; i <= support; ++i) out[i] = p * (i == hi_i ? 1.f : 0.f) + (1.f - p) * (i == lo_i ? 1.f : 0.f);
}

__global__ void replay_targets_kernel(ReplayDev d) {
    const int g = blockIdx.x * kReplayThreads + threadIdx.x;
    const int idx = blockIdx.y;                       // position in the padded game, 0 .. M+K
    if (g >= d.G) return;
    const int L = d.lengths[g];
    const size_t G = (size_t)d.G;
    const int Dr = d.reward_support > 0 ? d.reward_support + 1 : 1;
    const int Dv = d.value_support > 0 ? d.value_support + 1 : 1;

    if (idx < d.M + d.K) {
        // actions extended by random ones, rewards by zeros (replay_buffer.py:76-84); beyond L+K: array padding
        int a = 0;
        if (idx < L) a = d.actions[(size_t)idx * G + g];
        else if (idx < L + d.K) a = d.random_actions[(size_t)g * d.K + (idx - L)];
        d.ext_actions[(size_t)idx * G + g] = a;
        float* tr = d.target_rewards + ((size_t)idx * G + g) * Dr;
        if (idx < L + d.K) {
            store_scalar(idx < L ? d.rewards[(size_t)idx * G + g] : 0.f, d.reward_support, d.reward_lo, d.reward_rng, tr);
        } else {
            for (int i = 0; i < Dr; ++i) tr[i] = 0.f;
        }
    }
    float* tv = d.target_values + ((size_t)idx * G + g) * Dv;
    float* tp = d.target_policies + ((size_t)idx * G + g) * d.A;
    if (idx < L) {
        // compute_target_value (replay_buffer.py:66-73)
        const int b = idx + d.td;
        double s = 0.0;
        const int end = b < L ? b : L;
        for (int i = idx; i < end; ++i) s = s + (double)d.rewards[(size_t)i * G + g] * d.powers[i - idx];
        double value = s;                              // `0 + sum` when the bootstrap index is past the end
        if (b < L) value = d.root_values[(size_t)b * G + g] * d.powers[d.td] + s;
        store_scalar((float)value, d.value_support, d.value_lo, d.value_rng, tv);
        for (int k = 0; k < d.A; ++k) tp[k] = d.policies[((size_t)idx * G + g) * d.A + k];
    } else if (idx < L + d.K + 1) {
        store_scalar(0.f, d.value_support, d.value_lo, d.value_rng, tv);          // :88-89
        const float uni = 1.f / (float)d.A;                                       // :92-96
        for (int k = 0; k < d.A; ++k) tp[k] = uni;
    } else {
        for (int i = 0; i < Dv; ++i) tv[i] = 0.f;
        for (int k = 0; k < d.A; ++k) tp[k] = 0.f;
    }
    if (idx == 0 && d.total_value) {                                              // :98
        double s = 0.0;
        for (int i = 0; i < L; ++i) s = s + (double)d.rewards[(size_t)i * G + g] * d.powers[i];
        d.total_value[g] = s;
    }
}

}  // namespace

extern "C" int mz_replay_targets(void* stream, const mz_replay_config_t* c, const int32_t* lengths,
                                 const int32_t* actions, const float* rewards, const double* root_values,
                                 const float* policies, const int32_t* random_actions, const double* discount_powers,
                                 int32_t* out_extended_actions, float* out_target_rewards, float* out_target_values,
                                 float* out_target_policies, double* out_total_value) {
    MZ_REQUIRE(c != nullptr, "config is NULL");
    MZ_REQUIRE(c->num_games >= 0 && c->max_moves >= 1 && c->num_actions >= 1 && c->num_unroll_steps >= 0 &&
                   c->td_steps >= 0,
               "bad sizes in mz_replay_config_t");
    MZ_REQUIRE(c->value_support >= 0 && c->reward_support >= 0, "support sizes must be >= 0 (0 = scalar)");
    MZ_REQUIRE(c->value_support == 0 || c->value_max > c->value_min, "value support needs value_max > value_min");
    MZ_REQUIRE(c->reward_support == 0 || c->reward_max > c->reward_min, "reward support needs reward_max > reward_min");
    MZ_REQUIRE(c->max_moves + c->num_unroll_steps + 1 <= 65535, "max_moves + num_unroll_steps + 1 must be <= 65535");
    if (c->num_games == 0) return MZ_OK;
    MZ_REQUIRE(lengths && actions && rewards && root_values && policies && discount_powers,
               "lengths/actions/rewards/root_values/policies/discount_powers is NULL");
    MZ_REQUIRE(c->num_unroll_steps == 0 || random_actions, "random_actions is NULL");
    MZ_REQUIRE(out_extended_actions && out_target_rewards && out_target_values && out_target_policies,
               "an output pointer is NULL (only out_total_value may be)");
    ReplayDev d;
    d.G = c->num_games;
    d.M = c->max_moves;
    d.A = c->num_actions;
    d.K = c->num_unroll_steps;
    d.td = c->td_steps;
    d.value_support = c->value_support;
    d.reward_support = c->reward_support;
    d.value_lo = (float)c->value_min;
    d.value_rng = (float)(c->value_max - c->value_min);
    d.reward_lo = (float)c->reward_min;
    d.reward_rng = (float)(c->reward_max - c->reward_min);
    d.lengths = lengths;
    d.actions = actions;
    d.rewards = rewards;
    d.root_values = root_values;
    d.policies = policies;
    d.random_actions = random_actions;
    d.powers = discount_powers;
    d.ext_actions = out_extended_actions;
    d.target_rewards = out_target_rewards;
    d.target_values = out_target_values;
    d.target_policies = out_target_policies;
    d.total_value = out_total_value;
    const dim3 grid((d.G + kReplayThreads - 1) / kReplayThreads, d.M + d.K + 1);
    replay_targets_kernel<<<grid, kReplayThreads, 0, static_cast<hipStream_t>(stream)>>>(d);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mz::fail(MZ_ERR_HIP, "mz_replay_targets launch failed: %s", hipGetErrorString(e));
    return MZ_OK;
}
