// One move of T games enqueued from native code: the batched MCTSAgent.make_move (reference agent.py:219-232)
//   initial_inference -> expand roots -> Dirichlet noise -> N x (select -> recurrent_inference -> expand -> backprop)
//   -> root statistics -> action sampling
// as ~2N+6 HIP launches on one stream with no host synchronisation.  Driving the same sequence from Python costs
// ~100 us of interpreter time per simulation, more than the kernels themselves at 4096 games.
#include "mz_common.h"

extern "C" int mz_search_move(mz_pool_t* pool, mz_net_t* net, const mz_move_io_t* io) {
    MZ_REQUIRE(pool && net && io, "pool/net/io is NULL");
    MZ_REQUIRE(io->observation && io->noise && io->u && io->temperature, "observation/noise/u/temperature is NULL");
    MZ_REQUIRE(io->hidden && io->logits && io->reward && io->value && io->parent_block && io->action,
               "hidden/logits/reward/value/parent_block/action workspaces must be non-NULL");
    MZ_REQUIRE(io->num_trees >= 1 && io->num_simulations >= 1 && io->hidden_size >= 1, "bad sizes in mz_move_io_t");
    const int T = io->num_trees, N = io->num_simulations;
    const int64_t H = io->hidden_size;
    void** ev = io->events;
#define MZ_TRY_(expr)               \
    do {                            \
        int s_ = (expr);            \
        if (s_ != MZ_OK) return s_; \
    } while (0)
    MZ_TRY_(mz_pool_reset(pool));
    MZ_TRY_(mz_infer_initial(net, T, io->observation, io->hidden, io->value, io->logits, nullptr));
    MZ_TRY_(mz_expand_roots_logits(pool, io->logits, io->legal, io->n_legal));
    MZ_TRY_(mz_add_root_noise(pool, io->noise, io->noise_frac));
    MZ_TRY_(mz_select(pool, io->parent_block, io->action));
    for (int s = 0; s < N; ++s) {
        if (ev) {
            MZ_HIP(hipEventRecord(static_cast<hipEvent_t>(ev[4 * s]), static_cast<hipStream_t>(io->stream)));
            MZ_TRY_(mz_net_set_mid_event(net, ev[4 * s + 1]));
        }
        MZ_TRY_(mz_infer_recurrent_step(net, pool, s + 1 < N, io->hidden, io->parent_block, (int64_t)T * H, io->action,
                                        io->hidden + (size_t)(s + 1) * T * H, io->reward, io->value, io->logits));
        if (ev) MZ_HIP(hipEventRecord(static_cast<hipEvent_t>(ev[4 * s + 2]), static_cast<hipStream_t>(io->stream)));
        if (ev) MZ_HIP(hipEventRecord(static_cast<hipEvent_t>(ev[4 * s + 3]), static_cast<hipStream_t>(io->stream)));
    }
    if (ev) MZ_TRY_(mz_net_set_mid_event(net, nullptr));
    MZ_TRY_(mz_root_stats(pool, io->out_visits, io->out_root_value, io->out_minmax, io->out_policy));
    if (io->out_action) MZ_TRY_(mz_select_action(pool, io->u, io->temperature, io->out_action));
#undef MZ_TRY_
    return MZ_OK;
}
