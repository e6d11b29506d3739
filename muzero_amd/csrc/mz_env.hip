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
#include "mz_env_dev.h"
#include <cmath>

namespace {

constexpr int kEnvThreads = 256;

__global__ void ttt_reset_kernel(int T, signed char* board, int* to_play, int* steps, int* ended) {
    const int t = blockIdx.x * kEnvThreads + threadIdx.x;
    if (t >= T) return;
    ttt_reset_one(TttEnv{board, to_play, steps, ended}, t);
}

// reset of the games with where[t] != 0 only (a finished slot is restarted while the others play on)
__global__ void ttt_reset_where_kernel(int T, signed char* board, int* to_play, int* steps, int* ended, const int* where) {
    const int t = blockIdx.x * kEnvThreads + threadIdx.x;
    if (t >= T || where[t] == 0) return;
    ttt_reset_one(TttEnv{board, to_play, steps, ended}, t);
}

__global__ void ttt_observe_kernel(int T, const signed char* board, const int* to_play, const int* ended,
                                   float* obs, int* legal, int* n_legal) {
    const int t = blockIdx.x * kEnvThreads + threadIdx.x;
    if (t >= T) return;
    const TttEnv e{const_cast<signed char*>(board), const_cast<int*>(to_play), nullptr, const_cast<int*>(ended)};
    ttt_observe_one(e, t, obs ? obs + (size_t)t * 27 : nullptr, legal ? legal + (size_t)t * 9 : nullptr,
                    n_legal ? n_legal + t : nullptr);
}

__global__ void ttt_step_kernel(int T, signed char* board, int* to_play, int* steps, int* ended, const int* action,
                                float* reward, int* status, int* any_illegal) {
    const int t = blockIdx.x * kEnvThreads + threadIdx.x;
    if (t >= T) return;
    float r;
    const int st = ttt_step_one(TttEnv{board, to_play, steps, ended}, t, action[t], r, any_illegal);
    if (reward) reward[t] = r;
    if (status) status[t] = st;
}

__global__ void cartpole_reset_kernel(int T, double* state, int* steps, int* ended, const double* init) {
    const int t = blockIdx.x * kEnvThreads + threadIdx.x;
    if (t >= T) return;
    cartpole_reset_one(CartEnv{state, steps, ended, init}, t);
}

__global__ void cartpole_reset_where_kernel(int T, double* state, int* steps, int* ended, const double* init,
                                            const int* where) {
    const int t = blockIdx.x * kEnvThreads + threadIdx.x;
    if (t >= T || where[t] == 0) return;
    cartpole_reset_one(CartEnv{state, steps, ended, init}, t);
}

__global__ void cartpole_observe_kernel(int T, const double* state, const int* ended, float* obs, int* legal,
                                        int* n_legal) {
    const int t = blockIdx.x * kEnvThreads + threadIdx.x;
    if (t >= T) return;
    const CartEnv e{const_cast<double*>(state), nullptr, const_cast<int*>(ended), nullptr};
    cartpole_observe_one(e, t, obs ? obs + (size_t)t * 4 : nullptr, legal ? legal + (size_t)t * 2 : nullptr,
                         n_legal ? n_legal + t : nullptr);
}

__global__ void cartpole_step_kernel(int T, double* state, int* steps, int* ended, const int* action, float* reward,
                                     int* status, int* any_illegal) {
    const int t = blockIdx.x * kEnvThreads + threadIdx.x;
    if (t >= T) return;
    float r;
    const int st = cartpole_step_one(CartEnv{state, steps, ended, nullptr}, t, action[t], r, any_illegal);
    if (reward) reward[t] = r;
    if (status) status[t] = st;
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
