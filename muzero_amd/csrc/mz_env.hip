// Batched environments on the device: T games advance by one move per launch, so that a self-play move never
// returns to the host between the search and the next observation.
//
// What this replaces in the reference (file:line relative to /root/reference):
//   TicTacToeEnvironment.reset/step/_get_state/_legal_actions/_compute_reward    games/tictactoe.py:71-148
//   OpenAIEnvironment.reset/step for gym CartPole-v1                               environment.py:44-63
//   (gym is a third-party package, absent and unpinned: the CartPole-v1 equations are restated here and in
//    muzero_amd/environment.py -- "parity unpinned" against gym itself, exact against that restatement up to
//    the device libm's sin/cos.)
//
// State is caller-owned SoA (torch tensors); one lane per game; every array is touched by one coalesced access.
// Finished games are frozen: step() leaves them untouched, observe() reports n_legal = 0, which the tree pool
// and the inference kernels take as "inactive" (include/mzhip.h, mz_expand_roots).
#include "mz_common.h"
#include <cmath>

namespace {

constexpr int kEnvThreads = 256;

__global__ void ttt_reset_kernel(int T, signed char* board, int* to_play, int* steps, int* ended) {
    const int t = blockIdx.x * kEnvThreads + threadIdx.x;
    if (t >= T) return;
#pragma unroll
    for (int c = 0; c < 9; ++c) board[(size_t)t * 9 + c] = -1;       // games/tictactoe.py:144
    to_play[t] = 0;
    steps[t] = 0;
    ended[t] = 0;
}

// reset of the games with where[t] != 0 only (a finished slot is restarted while the others play on)
__global__ void ttt_reset_where_kernel(int T, signed char* board, int* to_play, int* steps, int* ended, const int* where) {
    const int t = blockIdx.x * kEnvThreads + threadIdx.x;
    if (t >= T || where[t] == 0) return;
#pragma unroll
    for (int c = 0; c < 9; ++c) board[(size_t)t * 9 + c] = -1;
    to_play[t] = 0;
    steps[t] = 0;
    ended[t] = 0;
}

// _get_state (games/tictactoe.py:107-111): planes [board == 0, board == 1, to_play] stacked on the last axis.
__global__ void ttt_observe_kernel(int T, const signed char* board, const int* to_play, const int* ended,
                                   float* obs, int* legal, int* n_legal) {
    const int t = blockIdx.x * kEnvThreads + threadIdx.x;
    if (t >= T) return;
    const float tp = (float)to_play[t];
    const bool over = ended[t] != 0;
    int n = 0;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const int v = board[(size_t)t * 9 + c];
        if (obs) {
            float* o = obs + (size_t)t * 27 + c * 3;
            o[0] = v == 0 ? 1.f : 0.f;
            o[1] = v == 1 ? 1.f : 0.f;
            o[2] = tp;
        }
        if (!over && v == -1) {                                    // _legal_actions: ascending order (:104-105)
            if (legal) legal[(size_t)t * 9 + n] = c;
            ++n;
        }
    }
    if (legal)
        for (int k = n; k < 9; ++k) legal[(size_t)t * 9 + k] = -1;
    if (n_legal) n_legal[t] = n;
}

// step (games/tictactoe.py:121-141).  status: 0 = moved, 1 = game already over (untouched), 2 = illegal action
// (untouched; the reference raises MuZeroEnvironmentError).
__global__ void ttt_step_kernel(int T, signed char* board, int* to_play, int* steps, int* ended, const int* action,
                                float* reward, int* status, int* any_illegal) {
    const int t = blockIdx.x * kEnvThreads + threadIdx.x;
    if (t >= T) return;
    if (ended[t]) {
        if (reward) reward[t] = 0.f;
        if (status) status[t] = 1;
        return;
    }
    const int a = action[t];
    signed char b[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) b[c] = board[(size_t)t * 9 + c];
    bool legal = a >= 0 && a < 9;
    if (legal) {
        bool empty = false;
#pragma unroll
        for (int c = 0; c < 9; ++c) empty |= (c == a && b[c] == -1);
        legal = empty;
    }
    if (!legal) {
        if (reward) reward[t] = 0.f;
        if (status) status[t] = 2;
        if (any_illegal) atomicOr(any_illegal, 1);
        return;
    }
    const int me = to_play[t];
#pragma unroll
    for (int c = 0; c < 9; ++c)
        if (c == a) b[c] = (signed char)me;
    board[(size_t)t * 9 + a] = (signed char)me;
    auto line = [&](int i, int j, int k) { return b[i] == me && b[j] == me && b[k] == me; };
    const bool won = line(0, 1, 2) || line(3, 4, 5) || line(6, 7, 8) || line(0, 3, 6) || line(1, 4, 7) ||
                     line(2, 5, 8) || line(0, 4, 8) || line(2, 4, 6);                      // _compute_reward (:113-119)
    const int s = steps[t] + 1;
    steps[t] = s;
    to_play[t] = 1 - me;
    ended[t] = (won || s == 9) ? 1 : 0;                             // draw after nine moves (:137-139)
    if (reward) reward[t] = won ? 1.f : 0.f;
    if (status) status[t] = 0;
}

// ---- CartPole-v1 (gym's classic-control equations, Euler integration, fp64 state like gym's Python floats)
__global__ void cartpole_reset_kernel(int T, double* state, int* steps, int* ended, const double* init) {
    const int t = blockIdx.x * kEnvThreads + threadIdx.x;
    if (t >= T) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) state[(size_t)t * 4 + i] = init[(size_t)t * 4 + i];
    steps[t] = 0;
    ended[t] = 0;
}

__global__ void cartpole_reset_where_kernel(int T, double* state, int* steps, int* ended, const double* init,
                                            const int* where) {
    const int t = blockIdx.x * kEnvThreads + threadIdx.x;
    if (t >= T || where[t] == 0) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) state[(size_t)t * 4 + i] = init[(size_t)t * 4 + i];
    steps[t] = 0;
    ended[t] = 0;
}

__global__ void cartpole_observe_kernel(int T, const double* state, const int* ended, float* obs, int* legal,
                                        int* n_legal) {
    const int t = blockIdx.x * kEnvThreads + threadIdx.x;
    if (t >= T) return;
    if (obs) {
#pragma unroll
        for (int i = 0; i < 4; ++i) obs[(size_t)t * 4 + i] = (float)state[(size_t)t * 4 + i];   // environment.py:59
    }
    if (legal) {
        legal[(size_t)t * 2] = 0;
        legal[(size_t)t * 2 + 1] = 1;
    }
    if (n_legal) n_legal[t] = ended[t] ? 0 : 2;                                          // environment.py:54-55
}

__global__ void cartpole_step_kernel(int T, double* state, int* steps, int* ended, const int* action, float* reward,
                                     int* status, int* any_illegal) {
    const int t = blockIdx.x * kEnvThreads + threadIdx.x;
    if (t >= T) return;
    if (ended[t]) {
        if (reward) reward[t] = 0.f;
        if (status) status[t] = 1;
        return;
    }
    const int a = action[t];
    if (a != 0 && a != 1) {
        if (reward) reward[t] = 0.f;
        if (status) status[t] = 2;
        if (any_illegal) atomicOr(any_illegal, 1);
        return;
    }
    constexpr double kGravity = 9.8, kMassCart = 1.0, kMassPole = 0.1, kLength = 0.5, kForce = 10.0, kTau = 0.02;
    constexpr double kTotalMass = kMassPole + kMassCart, kPoleMassLength = kMassPole * kLength;
    const double kThetaThreshold = 12 * 2 * 3.141592653589793 / 360, kXThreshold = 2.4;
    double x = state[(size_t)t * 4], x_dot = state[(size_t)t * 4 + 1];
    double theta = state[(size_t)t * 4 + 2], theta_dot = state[(size_t)t * 4 + 3];
    const double force = a == 1 ? kForce : -kForce;
    const double costheta = cos(theta), sintheta = sin(theta);
    const double temp = (force + kPoleMassLength * (theta_dot * theta_dot) * sintheta) / kTotalMass;
    const double thetaacc = (kGravity * sintheta - costheta * temp) /
                            (kLength * (4.0 / 3.0 - kMassPole * (costheta * costheta) / kTotalMass));
    const double xacc = temp - kPoleMassLength * thetaacc * costheta / kTotalMass;
    x = x + kTau * x_dot;
    x_dot = x_dot + kTau * xacc;
    theta = theta + kTau * theta_dot;
    theta_dot = theta_dot + kTau * thetaacc;
    state[(size_t)t * 4] = x;
    state[(size_t)t * 4 + 1] = x_dot;
    state[(size_t)t * 4 + 2] = theta;
    state[(size_t)t * 4 + 3] = theta_dot;
    const int s = steps[t] + 1;
    steps[t] = s;
    ended[t] = (fabs(x) > kXThreshold || fabs(theta) > kThetaThreshold || s >= 500) ? 1 : 0;
    if (reward) reward[t] = 1.f;
    if (status) status[t] = 0;
}

int env_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mz::fail(MZ_ERR_HIP, "%s launch failed: %s", what, hipGetErrorString(e));
    return MZ_OK;
}

inline int env_grid(int T) { return (T + kEnvThreads - 1) / kEnvThreads; }

}  // namespace

extern "C" {

int mz_env_ttt_reset(void* stream, int32_t num_games, int8_t* board, int32_t* to_play, int32_t* steps, int32_t* ended) {
    MZ_REQUIRE(num_games >= 0, "num_games must be >= 0 (got %d)", num_games);
    if (num_games == 0) return MZ_OK;
    MZ_REQUIRE(board && to_play && steps && ended, "board/to_play/steps/ended is NULL");
    ttt_reset_kernel<<<env_grid(num_games), kEnvThreads, 0, static_cast<hipStream_t>(stream)>>>(
        num_games, reinterpret_cast<signed char*>(board), to_play, steps, ended);
    return env_launch_status("mz_env_ttt_reset");
}

int mz_env_ttt_reset_where(void* stream, int32_t num_games, int8_t* board, int32_t* to_play, int32_t* steps, int32_t* ended,
                           const int32_t* where) {
    MZ_REQUIRE(num_games >= 0, "num_games must be >= 0 (got %d)", num_games);
    if (num_games == 0) return MZ_OK;
    MZ_REQUIRE(board && to_play && steps && ended && where, "board/to_play/steps/ended/where is NULL");
    ttt_reset_where_kernel<<<env_grid(num_games), kEnvThreads, 0, static_cast<hipStream_t>(stream)>>>(
        num_games, reinterpret_cast<signed char*>(board), to_play, steps, ended, where);
    return env_launch_status("mz_env_ttt_reset_where");
}

int mz_env_ttt_observe(void* stream, int32_t num_games, const int8_t* board, const int32_t* to_play,
                       const int32_t* ended, float* out_observation, int32_t* out_legal, int32_t* out_n_legal) {
    MZ_REQUIRE(num_games >= 0, "num_games must be >= 0 (got %d)", num_games);
    if (num_games == 0) return MZ_OK;
    MZ_REQUIRE(board && to_play && ended, "board/to_play/ended is NULL");
    ttt_observe_kernel<<<env_grid(num_games), kEnvThreads, 0, static_cast<hipStream_t>(stream)>>>(
        num_games, reinterpret_cast<const signed char*>(board), to_play, ended, out_observation, out_legal, out_n_legal);
    return env_launch_status("mz_env_ttt_observe");
}

int mz_env_ttt_step(void* stream, int32_t num_games, int8_t* board, int32_t* to_play, int32_t* steps, int32_t* ended,
                    const int32_t* action, float* out_reward, int32_t* out_status, int32_t* any_illegal) {
    MZ_REQUIRE(num_games >= 0, "num_games must be >= 0 (got %d)", num_games);
    if (num_games == 0) return MZ_OK;
    MZ_REQUIRE(board && to_play && steps && ended && action, "board/to_play/steps/ended/action is NULL");
    ttt_step_kernel<<<env_grid(num_games), kEnvThreads, 0, static_cast<hipStream_t>(stream)>>>(
        num_games, reinterpret_cast<signed char*>(board), to_play, steps, ended, action, out_reward, out_status,
        any_illegal);
    return env_launch_status("mz_env_ttt_step");
}

int mz_env_cartpole_reset(void* stream, int32_t num_games, double* state, int32_t* steps, int32_t* ended,
                          const double* initial_state) {
    MZ_REQUIRE(num_games >= 0, "num_games must be >= 0 (got %d)", num_games);
    if (num_games == 0) return MZ_OK;
    MZ_REQUIRE(state && steps && ended && initial_state, "state/steps/ended/initial_state is NULL");
    cartpole_reset_kernel<<<env_grid(num_games), kEnvThreads, 0, static_cast<hipStream_t>(stream)>>>(
        num_games, state, steps, ended, initial_state);
    return env_launch_status("mz_env_cartpole_reset");
}

int mz_env_cartpole_reset_where(void* stream, int32_t num_games, double* state, int32_t* steps, int32_t* ended,
                                const double* initial_state, const int32_t* where) {
    MZ_REQUIRE(num_games >= 0, "num_games must be >= 0 (got %d)", num_games);
    if (num_games == 0) return MZ_OK;
    MZ_REQUIRE(state && steps && ended && initial_state && where, "state/steps/ended/initial_state/where is NULL");
    cartpole_reset_where_kernel<<<env_grid(num_games), kEnvThreads, 0, static_cast<hipStream_t>(stream)>>>(
        num_games, state, steps, ended, initial_state, where);
    return env_launch_status("mz_env_cartpole_reset_where");
}

int mz_env_cartpole_observe(void* stream, int32_t num_games, const double* state, const int32_t* ended,
                            float* out_observation, int32_t* out_legal, int32_t* out_n_legal) {
    MZ_REQUIRE(num_games >= 0, "num_games must be >= 0 (got %d)", num_games);
    if (num_games == 0) return MZ_OK;
    MZ_REQUIRE(state && ended, "state/ended is NULL");
    cartpole_observe_kernel<<<env_grid(num_games), kEnvThreads, 0, static_cast<hipStream_t>(stream)>>>(
        num_games, state, ended, out_observation, out_legal, out_n_legal);
    return env_launch_status("mz_env_cartpole_observe");
}

int mz_env_cartpole_step(void* stream, int32_t num_games, double* state, int32_t* steps, int32_t* ended,
                         const int32_t* action, float* out_reward, int32_t* out_status, int32_t* any_illegal) {
    MZ_REQUIRE(num_games >= 0, "num_games must be >= 0 (got %d)", num_games);
    if (num_games == 0) return MZ_OK;
    MZ_REQUIRE(state && steps && ended && action, "state/steps/ended/action is NULL");
    cartpole_step_kernel<<<env_grid(num_games), kEnvThreads, 0, static_cast<hipStream_t>(stream)>>>(
        num_games, state, steps, ended, action, out_reward, out_status, any_illegal);
    return env_launch_status("mz_env_cartpole_step");
}

}  // extern "C"
