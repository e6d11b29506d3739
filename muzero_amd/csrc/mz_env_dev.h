// The two batched environments as per-game device functions, shared by the one-call-per-step kernels (mz_env.hip) and the
// fused self-play bookkeeping kernels (mz_history.hip).  Reference (file:line relative to /root/reference):
//   TicTacToeEnvironment.reset/step/_get_state/_legal_actions/_compute_reward    games/tictactoe.py:71-148
//   OpenAIEnvironment.reset/step around gym CartPole-v1 (equations restated; parity unpinned against gym)  environment.py:44-63
#pragma once
#include <hip/hip_runtime.h>

namespace {

struct TttEnv {            // caller-owned SoA state of G boards
    signed char* board;    // [G, 9]: -1 empty, 0 / 1 the players' marks
    int* to_play;
    int* steps;
    int* ended;
    static constexpr int kObs = 27, kActions = 9;
};

__device__ __forceinline__ void ttt_reset_one(const TttEnv& e, int t) {
#pragma unroll
    for (int c = 0; c < 9; ++c) e.board[(size_t)t * 9 + c] = -1;       // games/tictactoe.py:144
    e.to_play[t] = 0;
    e.steps[t] = 0;
    e.ended[t] = 0;
}

// _get_state (games/tictactoe.py:107-111): planes [board == 0, board == 1, to_play] stacked on the last axis; the legal
// actions in ascending order (:104-105), -1 past them; a finished game has none.  obs / legal / n_legal: this game's rows.
__device__ __forceinline__ void ttt_observe_one(const TttEnv& e, int t, float* obs, int* legal, int* n_legal) {
    const float tp = (float)e.to_play[t];
    const bool over = e.ended[t] != 0;
    int n = 0;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const int v = e.board[(size_t)t * 9 + c];
        if (obs) {
            float* o = obs + c * 3;
            o[0] = v == 0 ? 1.f : 0.f;
            o[1] = v == 1 ? 1.f : 0.f;
            o[2] = tp;
        }
        if (!over && v == -1) {
            if (legal) legal[n] = c;
            ++n;
        }
    }
    if (legal)
        for (int k = n; k < 9; ++k) legal[k] = -1;
    if (n_legal) *n_legal = n;
}

// step (games/tictactoe.py:121-141) -> status: 0 = moved, 1 = game already over (untouched), 2 = illegal action
// (untouched; the reference raises MuZeroEnvironmentError).
__device__ __forceinline__ int ttt_step_one(const TttEnv& e, int t, int a, float& reward, int* any_illegal) {
    reward = 0.f;
    if (e.ended[t]) return 1;
    signed char b[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) b[c] = e.board[(size_t)t * 9 + c];
    bool legal = a >= 0 && a < 9;
    if (legal) {
        bool empty = false;
#pragma unroll
        for (int c = 0; c < 9; ++c) empty |= (c == a && b[c] == -1);
        legal = empty;
    }
    if (!legal) {
        if (any_illegal) atomicOr(any_illegal, 1);
        return 2;
    }
    const int me = e.to_play[t];
#pragma unroll
    for (int c = 0; c < 9; ++c)
        if (c == a) b[c] = (signed char)me;
    e.board[(size_t)t * 9 + a] = (signed char)me;
    auto line = [&](int i, int j, int k) { return b[i] == me && b[j] == me && b[k] == me; };
    const bool won = line(0, 1, 2) || line(3, 4, 5) || line(6, 7, 8) || line(0, 3, 6) || line(1, 4, 7) ||
                     line(2, 5, 8) || line(0, 4, 8) || line(2, 4, 6);                      // _compute_reward (:113-119)
    const int s = e.steps[t] + 1;
    e.steps[t] = s;
    e.to_play[t] = 1 - me;
    e.ended[t] = (won || s == 9) ? 1 : 0;                             // draw after nine moves (:137-139)
    reward = won ? 1.f : 0.f;
    return 0;
}

// ---- CartPole-v1 (gym's classic-control equations, Euler integration, fp64 state like gym's Python floats)
struct CartEnv {
    double* state;         // [G, 4]: x, x_dot, theta, theta_dot
    int* steps;
    int* ended;
    const double* init;    // [G, 4]: the state a (re)started game begins in
    static constexpr int kObs = 4, kActions = 2;
};

__device__ __forceinline__ void cartpole_reset_one(const CartEnv& e, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) e.state[(size_t)t * 4 + i] = e.init[(size_t)t * 4 + i];
    e.steps[t] = 0;
    e.ended[t] = 0;
}

__device__ __forceinline__ void cartpole_observe_one(const CartEnv& e, int t, float* obs, int* legal, int* n_legal) {
    if (obs) {
#pragma unroll
        for (int i = 0; i < 4; ++i) obs[i] = (float)e.state[(size_t)t * 4 + i];             // environment.py:59
    }
    if (legal) {
        legal[0] = 0;
        legal[1] = 1;
    }
    if (n_legal) *n_legal = e.ended[t] ? 0 : 2;                                           // environment.py:54-55
}

__device__ __forceinline__ int cartpole_step_one(const CartEnv& e, int t, int a, float& reward, int* any_illegal) {
    reward = 0.f;
    if (e.ended[t]) return 1;
    if (a != 0 && a != 1) {
        if (any_illegal) atomicOr(any_illegal, 1);
        return 2;
    }
    constexpr double kGravity = 9.8, kMassCart = 1.0, kMassPole = 0.1, kLength = 0.5, kForce = 10.0, kTau = 0.02;
    constexpr double kTotalMass = kMassPole + kMassCart, kPoleMassLength = kMassPole * kLength;
    const double kThetaThreshold = 12 * 2 * 3.141592653589793 / 360, kXThreshold = 2.4;
    double x = e.state[(size_t)t * 4], x_dot = e.state[(size_t)t * 4 + 1];
    double theta = e.state[(size_t)t * 4 + 2], theta_dot = e.state[(size_t)t * 4 + 3];
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
    e.state[(size_t)t * 4] = x;
    e.state[(size_t)t * 4 + 1] = x_dot;
    e.state[(size_t)t * 4 + 2] = theta;
    e.state[(size_t)t * 4 + 3] = theta_dot;
    const int s = e.steps[t] + 1;
    e.steps[t] = s;
    e.ended[t] = (fabs(x) > kXThreshold || fabs(theta) > kThetaThreshold || s >= 500) ? 1 : 0;
    reward = 1.f;
    return 0;
}

// the player to move (history rows): TicTacToe keeps it, CartPole has one player
__device__ __forceinline__ int env_to_play(const TttEnv& e, int t) { return e.to_play[t]; }
__device__ __forceinline__ int env_to_play(const CartEnv&, int) { return 0; }
__device__ __forceinline__ void env_reset_one(const TttEnv& e, int t) { ttt_reset_one(e, t); }
__device__ __forceinline__ void env_reset_one(const CartEnv& e, int t) { cartpole_reset_one(e, t); }
__device__ __forceinline__ void env_observe_one(const TttEnv& e, int t, float* o, int* l, int* n) { ttt_observe_one(e, t, o, l, n); }
__device__ __forceinline__ void env_observe_one(const CartEnv& e, int t, float* o, int* l, int* n) { cartpole_observe_one(e, t, o, l, n); }
__device__ __forceinline__ int env_step_one(const TttEnv& e, int t, int a, float& r, int* ill) { return ttt_step_one(e, t, a, r, ill); }
__device__ __forceinline__ int env_step_one(const CartEnv& e, int t, int a, float& r, int* ill) { return cartpole_step_one(e, t, a, r, ill); }

}  // namespace
