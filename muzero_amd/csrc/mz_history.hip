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
    __shared__ int s_sum[kScanThreads];
    __shared__ int s_stored[kScanThreads];
    const int tid = threadIdx.x;
    const int per = (G + kScanThreads - 1) / kScanThreads;
    const int lo = tid * per, hi = min(G, lo + per);
    int n = 0, stored = 0;
    for (int g = lo; g < hi; ++g) {
        const bool over = ended[g] != 0 || steps[g] >= M;
        n += over ? 1 : 0;
        stored += (over && serial[g] < capacity) ? 1 : 0;
    }
    s_sum[tid] = n;
    s_stored[tid] = stored;
    __syncthreads();
    for (int off = 1; off < kScanThreads; off <<= 1) {         // inclusive scan of the per-thread counts
        const int v = tid >= off ? s_sum[tid - off] : 0;
        const int w = tid >= off ? s_stored[tid - off] : 0;
        __syncthreads();
        s_sum[tid] += v;
        s_stored[tid] += w;
        __syncthreads();
    }
    const long long next0 = counters[0];
    int rank = s_sum[tid] - n;                                 // exclusive
    for (int g = lo; g < hi; ++g) {
        const bool over = ended[g] != 0 || steps[g] >= M;
        flush_dst[g] = (over && serial[g] < capacity) ? serial[g] : -1;
        reset_mask[g] = over ? 1 : 0;
        if (over) serial[g] = (int)(next0 + rank++);
    }
    __syncthreads();
    if (tid == kScanThreads - 1) {
        counters[0] = next0 + s_sum[tid];
        counters[1] += s_stored[tid];
    }
}

// one workgroup per slot: copies the slot's history (rows 0 .. len) to column flush_dst[g] of the finished store
__global__ __launch_bounds__(64) void flush_copy_kernel(Hist live, Hist fin, const int* steps, const int* flush_dst) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const int d = flush_dst[g];
    if (d < 0) return;
    const int len = min(steps[g], live.M);
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

}  // extern "C"
