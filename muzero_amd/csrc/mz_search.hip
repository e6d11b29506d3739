// One move of T games enqueued from native code: the batched MCTSAgent.make_move (reference agent.py:219-232)
//   initial_inference -> expand roots -> Dirichlet noise -> N x (select -> recurrent_inference -> expand -> backprop)
//   -> root statistics -> action sampling
// as a handful of HIP launches (initial inference, root expansion + noise + first selection, the N simulations -- one launch
// where the network has a run kernel, else one or two per simulation --, root statistics + sampling) on one stream with no
// host synchronisation.  Driving the same sequence from Python costs
// ~100 us of interpreter time per simulation, more than the kernels themselves at 4096 games.
#include "mz_common.h"
#include "mz_pool_internal.h"

extern "C" int mz_search_move(mz_pool_t* pool, mz_net_t* net, const mz_move_io_t* io) {
    MZ_REQUIRE(pool && net && io, "pool/net/io is NULL");
    MZ_REQUIRE(io->observation && io->noise && io->u && io->temperature, "observation/noise/u/temperature is NULL");
    MZ_REQUIRE(io->hidden && io->logits && io->reward && io->value && io->parent_block && io->action,
               "hidden/logits/reward/value/parent_block/action workspaces must be non-NULL");
    MZ_REQUIRE(io->num_trees >= 1 && io->num_simulations >= 1 && io->hidden_size >= 1, "bad sizes in mz_move_io_t");
    // every launch below sizes its grid by the pool and addresses the workspaces by io: they must describe the same move
    MZ_REQUIRE(io->num_trees == pool->dev.T, "mz_move_io_t.num_trees = %d, the pool holds %d trees", io->num_trees,
               pool->dev.T);
    MZ_REQUIRE(io->num_simulations <= pool->dev.N, "mz_move_io_t.num_simulations = %d exceeds the pool's max_simulations %d",
               io->num_simulations, pool->dev.N);
    MZ_REQUIRE(io->hidden_size == mz_net_hidden_size_of(net), "mz_move_io_t.hidden_size = %d, the network's hidden state has %d floats",
               io->hidden_size, mz_net_hidden_size_of(net));
    MZ_REQUIRE(mz_net_num_actions_of(net) == pool->dev.A, "the network has %d actions, the pool %d", mz_net_num_actions_of(net),
               pool->dev.A);
    MZ_REQUIRE(static_cast<hipStream_t>(io->stream) == pool->stream, "mz_move_io_t.stream is not the pool's stream");
    MZ_REQUIRE(mz_net_stream_of(net) == pool->stream, "net and pool must launch on the same stream");
    const int ntrace = (io->trace_reward != nullptr) + (io->trace_value != nullptr) + (io->trace_logits != nullptr) +
                       (io->trace_parent != nullptr) + (io->trace_action != nullptr);
    MZ_REQUIRE(ntrace == 0 || ntrace == 5, "the five trace_* pointers of mz_move_io_t must be all NULL or all set");
    const bool trace = ntrace == 5;
    const int T = io->num_trees, N = io->num_simulations;
    const int64_t H = io->hidden_size;
    const size_t A = (size_t)pool->dev.A;
    hipStream_t st = pool->stream;
    void** ev = io->events;
#define MZ_TRY_(expr)               \
    do {                            \
        int s_ = (expr);            \
        if (s_ != MZ_OK) return s_; \
    } while (0)
    MZ_TRY_(mz_pool_reset(pool));
    // initial inference; from here to the end (every exit, errors included) the hidden workspace is the network's
    struct EndMove {
        mz_net_t* net;
        ~EndMove() { mz_net_end_move(net); }
    } end_move{net};
    // ... then the root expansion, the noise and the first selection
    MZ_TRY_(mz_net_begin_move_roots(net, pool, T, io->observation, io->hidden, io->value, io->logits, io->legal, io->n_legal,
                                    io->noise, io->noise_frac, io->parent_block, io->action));
    if (!ev || mz_net_run_is_one_launch(net, pool)) {
        // run_mcts as one call (one launch where the network and pool allow it).  Timing events then bracket the whole
        // run: those of the first simulation that has any are recorded, [0] before it, [1] and [2] after it.
        int s0 = -1;
        for (int s = 0; ev && s < N && s0 < 0; ++s)
            if (ev[4 * s]) s0 = s;
        if (ev) MZ_TRY_(mz_net_set_mid_event(net, nullptr));
        if (s0 >= 0) MZ_HIP(hipEventRecord(static_cast<hipEvent_t>(ev[4 * s0]), st));
        MZ_TRY_(mz_infer_recurrent_run(net, pool, N, 0, io->hidden, io->parent_block, (int64_t)T * H, io->action,
                                       trace ? io->trace_reward : io->reward, trace ? io->trace_value : io->value,
                                       trace ? io->trace_logits : io->logits, trace ? (int64_t)T : 0,
                                       trace ? io->trace_parent : nullptr, trace ? io->trace_action : nullptr));
        if (s0 >= 0) {
            MZ_HIP(hipEventRecord(static_cast<hipEvent_t>(ev[4 * s0 + 1]), st));
            MZ_HIP(hipEventRecord(static_cast<hipEvent_t>(ev[4 * s0 + 2]), st));
        }
    } else {
        for (int s = 0; s < N; ++s) {
            const bool timed = ev && ev[4 * s];           // a simulation whose events are NULL is not timed
            if (timed) {
                MZ_HIP(hipEventRecord(static_cast<hipEvent_t>(ev[4 * s]), static_cast<hipStream_t>(io->stream)));
                MZ_TRY_(mz_net_set_mid_event(net, ev[4 * s + 1]));
            } else if (ev) {
                MZ_TRY_(mz_net_set_mid_event(net, nullptr));
            }
            if (trace) {      // this simulation's selection, before the fused step overwrites it with the next one
                MZ_HIP(hipMemcpyAsync(io->trace_parent + (size_t)s * T, io->parent_block, (size_t)T * sizeof(int32_t),
                                      hipMemcpyDeviceToDevice, st));
                MZ_HIP(hipMemcpyAsync(io->trace_action + (size_t)s * T, io->action, (size_t)T * sizeof(int32_t),
                                      hipMemcpyDeviceToDevice, st));
            }
            MZ_TRY_(mz_infer_recurrent_step(net, pool, s + 1 < N, io->hidden, io->parent_block, (int64_t)T * H, io->action,
                                            io->hidden + (size_t)(s + 1) * T * H,
                                            trace ? io->trace_reward + (size_t)s * T : io->reward,
                                            trace ? io->trace_value + (size_t)s * T : io->value,
                                            trace ? io->trace_logits + (size_t)s * T * A : io->logits));
            if (timed) MZ_HIP(hipEventRecord(static_cast<hipEvent_t>(ev[4 * s + 2]), static_cast<hipStream_t>(io->stream)));
        }
    }
    if (ev) MZ_TRY_(mz_net_set_mid_event(net, nullptr));
    if (io->out_action)
        MZ_TRY_(mz_root_finish(pool, io->out_visits, io->out_root_value, io->out_minmax, io->out_policy, io->u,
                               io->temperature, io->out_action));
    else
        MZ_TRY_(mz_root_stats(pool, io->out_visits, io->out_root_value, io->out_minmax, io->out_policy));
#undef MZ_TRY_
    return MZ_OK;
}

extern "C" int mz_search_move_injected(mz_pool_t* pool, int32_t num_simulations, const float* policy, int32_t is_logits,
                                       const float* reward, const float* value, const int32_t* legal,
                                       const int32_t* n_legal, const double* noise, double noise_frac,
                                       int32_t* parent_block, int32_t* action, void** events) {
    MZ_REQUIRE(pool && policy && reward && value && parent_block && action, "pool/policy/reward/value/workspaces is NULL");
    const int N = num_simulations;
    MZ_REQUIRE(N >= 1 && N <= pool->dev.N, "num_simulations must be in [1, %d] (got %d)", pool->dev.N, N);
    const size_t T = (size_t)pool->dev.T, TA = T * (size_t)pool->dev.A;
#define MZ_TRY_(expr)               \
    do {                            \
        int s_ = (expr);            \
        if (s_ != MZ_OK) return s_; \
    } while (0)
    MZ_TRY_(mz_pool_reset(pool));
    MZ_TRY_(mz_expand_roots_noise(pool, policy, is_logits, legal, n_legal, noise, noise_frac, parent_block, action));
    // Timing events around single fused steps ask for the launch per simulation; otherwise the N simulations are ONE
    // launch (run_injected_kernel), bracketed by the first 'before' and the last 'after' event when they are given.
    bool per_step = false;
    for (int i = 1; events && i < 2 * (N - 1) - 1; ++i) per_step |= events[i] != nullptr;
    if (!per_step) {
        if (events && N > 1 && events[0]) MZ_HIP(hipEventRecord(static_cast<hipEvent_t>(events[0]), pool->stream));
        MZ_TRY_(mz_pool_run_injected(pool, N, false, policy + TA, is_logits != 0, reward + T, value + T, TA, T, parent_block,
                                     action));
        if (events && N > 1 && events[2 * (N - 2) + 1])
            MZ_HIP(hipEventRecord(static_cast<hipEvent_t>(events[2 * (N - 2) + 1]), pool->stream));
        return MZ_OK;
    }
    for (int s = 1; s <= N; ++s) {
        const float* pol = policy + (size_t)s * TA;
        const float* rw = reward + (size_t)s * T;
        const float* vl = value + (size_t)s * T;
        if (s == N) {
            MZ_TRY_(is_logits ? mz_expand_backprop_logits(pool, pol, rw, vl) : mz_expand_backprop(pool, pol, rw, vl));
            break;
        }
        // events[2(s-1)] is recorded before fused step s, events[2(s-1)+1] after it; either may be NULL (a pair around one
        // step times that launch, a 'before' of the first step and an 'after' of the last one time the run of them)
        if (events && events[2 * (s - 1)])
            MZ_HIP(hipEventRecord(static_cast<hipEvent_t>(events[2 * (s - 1)]), pool->stream));
        MZ_TRY_(is_logits ? mz_step_logits(pool, pol, rw, vl, parent_block, action)
                          : mz_step(pool, pol, rw, vl, parent_block, action));
        if (events && events[2 * (s - 1) + 1])
            MZ_HIP(hipEventRecord(static_cast<hipEvent_t>(events[2 * (s - 1) + 1]), pool->stream));
    }
#undef MZ_TRY_
    return MZ_OK;
}
