// Game histories of a batch of self-play slots on the device, and the hand-over of finished games.
//
// What this replaces in the reference (file:line relative to /root/reference):
//   GameHistory's per-move appends                              game.py:8-41, 66-75 (Game.apply, store_search_statistics)
//   the agent's outer loop: play a game, hand its history to the replay buffer, start the next one at once
//                                                               agent_services.py:47-54
// A "slot" is one of G concurrent games.  Slots advance independently: slot g is at move steps[g] of ITS game, so records
// are scattered to row steps[g] of the slot's padded history (move-major arrays, muzero_amd.selfplay.BatchedHistory).
// When a slot's game ends, mz_history_flush copies the finished history into the `finished` store at the game's serial
// number and hands the slot a new serial; the environment's reset_where then restarts exactly those slots, so every move
// searches a full batch (the reference's agent never idles between games either).
// HBM-bound byte moving: one lane per element, coalesced over the game axis.
#include "mz_common.h"
#include "mz_env_dev.h"

namespace {

constexpr int kThreads = 256;

struct Hist {          // mz_history_t by value
    int G, M, O, A;
    float* obs;        // [M+1, G, O]
    int* to_play;      // [M+1, G]
    int* action;       // [M, G]
    float* reward;     // [M, G]
    double* root_value;  // [M, G]
    float* policy;     // [M, G, A]
    int* length;       // [G]
};

Hist unpack(const mz_history_t* h) {
    return Hist{h->num_games, h->max_moves, h->observation_size, h->num_actions, h->observations, h->to_plays, h->actions,
                h->rewards, h->root_values, h->policies, h->lengths};
}

// observation + to_play of every slot at row steps[g] (game.py:57-61: the observation list is one longer than the actions)
__global__ void record_observation_kernel(Hist h, const int* steps, const float* obs, const int* to_play) {
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= (size_t)h.G * h.O) return;
    const int g = (int)(i / h.O), o = (int)(i % h.O);
    const int m = steps[g];
    if (m < 0 || m > h.M) return;
    h.obs[((size_t)m * h.G + g) * h.O + o] = obs[i];
    if (o == 0) h.to_play[(size_t)m * h.G + g] = to_play[g];
}

// the move just played by every slot whose step succeeded (status == 0): row steps[g] - 1 (steps is already advanced)
__global__ void record_move_kernel(Hist h, const int* steps, const int* status, const int* action, const float* reward,
                                   const double* root_value, const float* policy) {
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= (size_t)h.G * h.A) return;
    const int g = (int)(i / h.A), a = (int)(i % h.A);
    if (status[g] != 0) return;
    const int m = steps[g] - 1;
    if (m < 0 || m >= h.M) return;
    h.policy[((size_t)m * h.G + g) * h.A + a] = policy[i];              // game.py:73-75
    if (a == 0) {
        h.action[(size_t)m * h.G + g] = action[g];                      // game.py:66-71
        h.reward[(size_t)m * h.G + g] = reward[g];
        h.root_value[(size_t)m * h.G + g] = root_value[g];
    }
}

// Slots whose game is over (ended != 0, or steps == M: the move capacity, agent.py:34 `len(history) < max_moves`): their
// destination in the finished store = their serial; new serials are handed out in slot order (deterministic): one
// workgroup scans the G flags.  counters[0] = next serial, counters[1] = games stored so far.
constexpr int kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void flush_plan_kernel(int G, int M, int capacity, const int* steps,
                                                                  const int* ended, int* serial, long long* counters,
                                                                  int* flush_dst, int* reset_mask) {
    __shared__ int s_sum[kScanThreads / 64], s_stored[kScanThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (G + kScanThreads - 1) / kScanThreads;
    const int lo = tid * per, hi = min(G, lo + per);
    int n = 0, stored = 0;
    for (int g = lo; g < hi; ++g) {
        const bool over = ended[g] != 0 || steps[g] >= M;
        n += over ? 1 : 0;
        stored += (over && serial[g] < capacity) ? 1 : 0;
    }
    int in = n, is = stored;                                   // inclusive scan inside the wavefront ...
    for (int off = 1; off < 64; off <<= 1) {
        const int a = __shfl_up(in, off), b = __shfl_up(is, off);
        if (lane >= off) {
            in += a;
            is += b;
        }
    }
    if (lane == 63) {
        s_sum[wave] = in;
        s_stored[wave] = is;
    }
    __syncthreads();
    int total = 0, total_stored = 0;                           // ... plus the wavefronts before this one
    for (int w = 0; w < kScanThreads / 64; ++w) {
        if (w < wave) in += s_sum[w];
        total += s_sum[w];
        total_stored += s_stored[w];
    }
    const long long next0 = counters[0];
    int rank = in - n;                                         // exclusive
    for (int g = lo; g < hi; ++g) {
        const bool over = ended[g] != 0 || steps[g] >= M;
        flush_dst[g] = (over && serial[g] < capacity) ? serial[g] : -1;
        reset_mask[g] = over ? 1 : 0;
        if (over) serial[g] = (int)(next0 + rank++);
    }
    __syncthreads();                                           // every thread has read counters[0]
    if (tid == 0) {
        counters[0] = next0 + total;
        counters[1] += total_stored;
    }
}

// the history of slot g (rows 0 .. len) to column d of the finished store, by the 64 lanes of one wavefront
__device__ __forceinline__ void copy_slot(const Hist& live, const Hist& fin, int g, int d, int len, int lane) {
    const int O = live.O, A = live.A;
    for (int i = lane; i < (len + 1) * O; i += 64) {
        const int m = i / O, o = i % O;
        fin.obs[((size_t)m * fin.G + d) * O + o] = live.obs[((size_t)m * live.G + g) * O + o];
    }
    for (int i = lane; i < len * A; i += 64) {
        const int m = i / A, a = i % A;
        fin.policy[((size_t)m * fin.G + d) * A + a] = live.policy[((size_t)m * live.G + g) * A + a];
    }
    for (int m = lane; m <= len; m += 64) {
        fin.to_play[(size_t)m * fin.G + d] = live.to_play[(size_t)m * live.G + g];
        if (m < len) {
            fin.action[(size_t)m * fin.G + d] = live.action[(size_t)m * live.G + g];
            fin.reward[(size_t)m * fin.G + d] = live.reward[(size_t)m * live.G + g];
            fin.root_value[(size_t)m * fin.G + d] = live.root_value[(size_t)m * live.G + g];
        }
    }
    if (lane == 0) fin.length[d] = len;
}

// one workgroup per slot: copies the slot's history (rows 0 .. len) to column flush_dst[g] of the finished store
__global__ __launch_bounds__(64) void flush_copy_kernel(Hist live, Hist fin, const int* steps, const int* flush_dst) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const int d = flush_dst[g];
    if (d < 0) return;
    copy_slot(live, fin, g, d, min(steps[g], live.M), lane);
}

// ---- the bookkeeping between two searches of a continuously refilled batch, fused (mz_selfplay_turnover)
// One wavefront per slot, after flush_plan_kernel has numbered the finished games: what the separate calls
//   observe -> record_observation -> flush (copy) -> reset_where -> observe -> record_observation -> dirichlet_from_gamma
// do in seven launches, plus the per-slot temperature (config.py:153-157).  Lane 0 runs the environment's per-game
// functions into LDS; the 64 lanes move the rows.
template <class Env>
__global__ __launch_bounds__(64) void turnover_kernel(Env env, Hist live, Hist fin, const int* flush_dst,
                                                      const int* reset_mask, const double* gamma, int freezing_moves,
                                                      double temperature, float* out_obs, int* out_legal,
                                                      int* out_n_legal, double* out_noise, double* out_temperature) {
    constexpr int O = Env::kObs, A = Env::kActions;
    __shared__ float sObs[O];
    __shared__ int sLegal[A];
    __shared__ int sN, sToPlay;
    const int g = blockIdx.x, lane = threadIdx.x, G = live.G;
    int m = env.steps[g];
    const bool restart = reset_mask[g] != 0;
    const int d = flush_dst[g];
    if (lane == 0) {
        env_observe_one(env, g, sObs, sLegal, &sN);
        sToPlay = env_to_play(env, g);
    }
    __syncthreads();
    if (m >= 0 && m <= live.M) {                 // the observation the game is in now (for a finished game: its last)
        for (int o = lane; o < O; o += 64) live.obs[((size_t)m * G + g) * O + o] = sObs[o];
        if (lane == 0) live.to_play[(size_t)m * G + g] = sToPlay;
    }
    if (restart) {
        __syncthreads();                         // the row above is read back by other lanes
        if (d >= 0) copy_slot(live, fin, g, d, min(m, live.M), lane);
        __syncthreads();
        if (lane == 0) {
            env_reset_one(env, g);
            env_observe_one(env, g, sObs, sLegal, &sN);
            sToPlay = env_to_play(env, g);
        }
        __syncthreads();
        m = 0;
        for (int o = lane; o < O; o += 64) live.obs[((size_t)g) * O + o] = sObs[o];
        if (lane == 0) live.to_play[g] = sToPlay;
    }
    for (int o = lane; o < O; o += 64) out_obs[(size_t)g * O + o] = sObs[o];
    if (lane < A) out_legal[(size_t)g * A + lane] = sLegal[lane];
    if (lane == 0) {
        const int n = sN;
        out_n_legal[g] = n;
        out_temperature[g] = m < freezing_moves ? temperature : 0.0;
        if (gamma) {                             // dirichlet_kernel's arithmetic
            double total = 0.0;
            for (int k = 0; k < n; ++k) total += gamma[(size_t)g * A + k];
            for (int k = 0; k < A; ++k)
                out_noise[(size_t)g * A + k] = (k < n && total > 0.0) ? gamma[(size_t)g * A + k] / total : 0.0;
        }
    }
}

// environment step + record_move of every slot in one launch (one lane per slot)
template <class Env>
__global__ void step_record_kernel(Env env, Hist h, const int* action, const double* root_value, const float* policy,
                                   float* out_reward, int* out_status, int* any_illegal) {
    constexpr int A = Env::kActions;
    const int g = blockIdx.x * kThreads + threadIdx.x;
    if (g >= h.G) return;
    float r;
    const int a = action[g];
    const int st = env_step_one(env, g, a, r, any_illegal);
    out_reward[g] = r;
    out_status[g] = st;
    if (st != 0) return;
    const int m = env.steps[g] - 1;              // the step above advanced it
    if (m < 0 || m >= h.M) return;
    for (int k = 0; k < A; ++k) h.policy[((size_t)m * h.G + g) * A + k] = policy[(size_t)g * A + k];
    h.action[(size_t)m * h.G + g] = a;
    h.reward[(size_t)m * h.G + g] = r;
    h.root_value[(size_t)m * h.G + g] = root_value[g];
}

// np.random.dirichlet([alpha] * n) (agent.py:172) from independent Gamma(alpha, 1) draws: the first n_legal[g] entries of
// row g divided by their sum (summed in order, fp64), the rest zero.  One lane per game: A is small.
__global__ void dirichlet_kernel(int G, int A, const double* gamma, const int* n_legal, double* noise) {
    const int g = blockIdx.x * kThreads + threadIdx.x;
    if (g >= G) return;
    const int n = min(max(n_legal[g], 0), A);
    double total = 0.0;
    for (int k = 0; k < n; ++k) total += gamma[(size_t)g * A + k];
    for (int k = 0; k < A; ++k) noise[(size_t)g * A + k] = (k < n && total > 0.0) ? gamma[(size_t)g * A + k] / total : 0.0;
}

int hist_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mz::fail(MZ_ERR_HIP, "%s launch failed: %s", what, hipGetErrorString(e));
    return MZ_OK;
}

int check_hist(const mz_history_t* h, const char* name) {
    MZ_REQUIRE(h != nullptr, "%s is NULL", name);
    MZ_REQUIRE(h->num_games >= 1 && h->max_moves >= 1 && h->observation_size >= 1 && h->num_actions >= 1,
               "%s: num_games/max_moves/observation_size/num_actions must be >= 1", name);
    MZ_REQUIRE(h->observations && h->to_plays && h->actions && h->rewards && h->root_values && h->policies && h->lengths,
               "%s: every array of mz_history_t must be non-NULL", name);
    return MZ_OK;
}

int check_env(const mz_env_t* env) {
    MZ_REQUIRE(env != nullptr, "env is NULL");
    MZ_REQUIRE(env->kind == MZ_ENV_TICTACTOE || env->kind == MZ_ENV_CARTPOLE, "unknown environment kind %d", env->kind);
    MZ_REQUIRE(env->num_games >= 1, "env->num_games must be >= 1");
    MZ_REQUIRE(env->state && env->steps && env->ended, "env->state/steps/ended is NULL");
    MZ_REQUIRE(env->kind != MZ_ENV_TICTACTOE || env->to_play, "env->to_play is NULL (TicTacToe)");
    MZ_REQUIRE(env->kind != MZ_ENV_CARTPOLE || env->restart_state, "env->restart_state is NULL (CartPole)");
    return MZ_OK;
}
TttEnv as_ttt(const mz_env_t* e) { return TttEnv{static_cast<signed char*>(e->state), e->to_play, e->steps, e->ended}; }
CartEnv as_cart(const mz_env_t* e) { return CartEnv{static_cast<double*>(e->state), e->steps, e->ended, e->restart_state}; }

}  // namespace

extern "C" {

int mz_history_record_observation(void* stream, const mz_history_t* hist, const int32_t* steps, const float* observation,
                                  const int32_t* to_play) {
    int s = check_hist(hist, "hist");
    if (s != MZ_OK) return s;
    MZ_REQUIRE(steps && observation && to_play, "steps/observation/to_play is NULL");
    const Hist h = unpack(hist);
    const size_t n = (size_t)h.G * h.O;
    record_observation_kernel<<<(unsigned)((n + kThreads - 1) / kThreads), kThreads, 0, static_cast<hipStream_t>(stream)>>>(
        h, steps, observation, to_play);
    return hist_status("mz_history_record_observation");
}

int mz_history_record_move(void* stream, const mz_history_t* hist, const int32_t* steps, const int32_t* status,
                           const int32_t* action, const float* reward, const double* root_value, const float* policy) {
    int s = check_hist(hist, "hist");
    if (s != MZ_OK) return s;
    MZ_REQUIRE(steps && status && action && reward && root_value && policy,
               "steps/status/action/reward/root_value/policy is NULL");
    const Hist h = unpack(hist);
    const size_t n = (size_t)h.G * h.A;
    record_move_kernel<<<(unsigned)((n + kThreads - 1) / kThreads), kThreads, 0, static_cast<hipStream_t>(stream)>>>(
        h, steps, status, action, reward, root_value, policy);
    return hist_status("mz_history_record_move");
}

int mz_dirichlet_from_gamma(void* stream, int32_t num_games, int32_t num_actions, const double* gamma,
                            const int32_t* n_legal, double* out_noise) {
    MZ_REQUIRE(num_games >= 1 && num_actions >= 1, "num_games/num_actions must be >= 1");
    MZ_REQUIRE(gamma && n_legal && out_noise, "gamma/n_legal/out_noise is NULL");
    dirichlet_kernel<<<(num_games + kThreads - 1) / kThreads, kThreads, 0, static_cast<hipStream_t>(stream)>>>(
        num_games, num_actions, gamma, n_legal, out_noise);
    return hist_status("mz_dirichlet_from_gamma");
}

int mz_history_flush(void* stream, const mz_history_t* live, const mz_history_t* finished, const int32_t* steps,
                     const int32_t* ended, int32_t* serial, int64_t* counters, int32_t* flush_dst, int32_t* reset_mask) {
    int s = check_hist(live, "live");
    if (s != MZ_OK) return s;
    s = check_hist(finished, "finished");
    if (s != MZ_OK) return s;
    MZ_REQUIRE(live->max_moves == finished->max_moves && live->observation_size == finished->observation_size &&
                   live->num_actions == finished->num_actions,
               "live and finished histories must agree on max_moves, observation_size and num_actions");
    MZ_REQUIRE(steps && ended && serial && counters && flush_dst && reset_mask,
               "steps/ended/serial/counters/flush_dst/reset_mask is NULL");
    const Hist l = unpack(live), f = unpack(finished);
    hipStream_t st = static_cast<hipStream_t>(stream);
    flush_plan_kernel<<<1, kScanThreads, 0, st>>>(l.G, l.M, f.G, steps, ended, serial,
                                                  reinterpret_cast<long long*>(counters), flush_dst, reset_mask);
    flush_copy_kernel<<<l.G, 64, 0, st>>>(l, f, steps, flush_dst);
    return hist_status("mz_history_flush");
}

int mz_selfplay_turnover(void* stream, const mz_env_t* env, const mz_history_t* live, const mz_history_t* finished,
                         int32_t* serial, int64_t* counters, int32_t* flush_dst, int32_t* reset_mask, const double* gamma,
                         int32_t freezing_moves, double temperature, float* out_observation, int32_t* out_legal,
                         int32_t* out_n_legal, double* out_noise, double* out_temperature) {
    int s = check_env(env);
    if (s != MZ_OK) return s;
    s = check_hist(live, "live");
    if (s != MZ_OK) return s;
    s = check_hist(finished, "finished");
    if (s != MZ_OK) return s;
    MZ_REQUIRE(live->max_moves == finished->max_moves && live->observation_size == finished->observation_size &&
                   live->num_actions == finished->num_actions,
               "live and finished histories must agree on max_moves, observation_size and num_actions");
    MZ_REQUIRE(live->num_games == env->num_games, "live history holds %d slots, the environment %d", live->num_games,
               env->num_games);
    const int O = env->kind == MZ_ENV_TICTACTOE ? TttEnv::kObs : CartEnv::kObs;
    const int A = env->kind == MZ_ENV_TICTACTOE ? TttEnv::kActions : CartEnv::kActions;
    MZ_REQUIRE(live->observation_size == O && live->num_actions == A,
               "history rows (%d observation floats, %d actions) do not match the environment (%d, %d)",
               live->observation_size, live->num_actions, O, A);
    MZ_REQUIRE(serial && counters && flush_dst && reset_mask, "serial/counters/flush_dst/reset_mask is NULL");
    MZ_REQUIRE(out_observation && out_legal && out_n_legal && out_temperature,
               "out_observation/out_legal/out_n_legal/out_temperature is NULL");
    MZ_REQUIRE(gamma == nullptr || out_noise != nullptr, "gamma given without out_noise");
    const Hist l = unpack(live), f = unpack(finished);
    hipStream_t st = static_cast<hipStream_t>(stream);
    flush_plan_kernel<<<1, kScanThreads, 0, st>>>(l.G, l.M, f.G, env->steps, env->ended, serial,
                                                  reinterpret_cast<long long*>(counters), flush_dst, reset_mask);
    if (env->kind == MZ_ENV_TICTACTOE)
        turnover_kernel<TttEnv><<<l.G, 64, 0, st>>>(as_ttt(env), l, f, flush_dst, reset_mask, gamma, freezing_moves,
                                                   temperature, out_observation, out_legal, out_n_legal, out_noise,
                                                   out_temperature);
    else
        turnover_kernel<CartEnv><<<l.G, 64, 0, st>>>(as_cart(env), l, f, flush_dst, reset_mask, gamma, freezing_moves,
                                                    temperature, out_observation, out_legal, out_n_legal, out_noise,
                                                    out_temperature);
    return hist_status("mz_selfplay_turnover");
}

int mz_selfplay_step_record(void* stream, const mz_env_t* env, const mz_history_t* live, const int32_t* action,
                            const double* root_value, const float* policy, float* out_reward, int32_t* out_status,
                            int32_t* any_illegal) {
    int s = check_env(env);
    if (s != MZ_OK) return s;
    s = check_hist(live, "live");
    if (s != MZ_OK) return s;
    MZ_REQUIRE(live->num_games == env->num_games, "live history holds %d slots, the environment %d", live->num_games,
               env->num_games);
    const int A = env->kind == MZ_ENV_TICTACTOE ? TttEnv::kActions : CartEnv::kActions;
    MZ_REQUIRE(live->num_actions == A, "history rows hold %d actions, the environment has %d", live->num_actions, A);
    MZ_REQUIRE(action && root_value && policy && out_reward && out_status,
               "action/root_value/policy/out_reward/out_status is NULL");
    const Hist l = unpack(live);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = (l.G + kThreads - 1) / kThreads;
    if (env->kind == MZ_ENV_TICTACTOE)
        step_record_kernel<TttEnv><<<grid, kThreads, 0, st>>>(as_ttt(env), l, action, root_value, policy, out_reward,
                                                             out_status, any_illegal);
    else
        step_record_kernel<CartEnv><<<grid, kThreads, 0, st>>>(as_cart(env), l, action, root_value, policy, out_reward,
                                                              out_status, any_illegal);
    return hist_status("mz_selfplay_step_record");
}

}  // extern "C"
