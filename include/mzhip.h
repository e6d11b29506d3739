/*
 * mzhip.h -- C ABI of libmzhip.so, the MI355X (gfx950) MuZero self-play hot path.
 *
 * Drop-in boundary for the reference's MCTS self-play path (fidel-schaposnik/muzero):
 *   MCTSAgent.run_mcts  and everything it calls            reference agent.py:154-232
 *   MinMaxStats / KnownBounds                               reference utils.py:37-76
 *   MuZeroConfig.exploration_function                       reference config.py:159-160
 *   Network.initial_inference / recurrent_inference         reference network.py:209-254
 *   BatchNetworkOutput.split_batch / to_scalar              reference network.py:165-174, config.py:25-40
 *
 * Conventions
 *   - plain C types only; every array argument is a DEVICE pointer unless it says "host";
 *   - all work is enqueued on the HIP stream the pool / net was created with (or set_stream); calls
 *     return after enqueueing unless they say "synchronises";
 *   - every function returns MZ_OK (0) or a negative mz_status; mz_last_error() gives the message of the
 *     last failure on the calling thread; nothing throws across this boundary;
 *   - a pool is driven by one host thread at a time (the reference agent is single-threaded);
 *   - T = trees (concurrent games), A = action_space_size, N = num_simulations.
 *
 * Tree layout: every expanded node owns a "block" of A child slots.  Block 0 is the root; the leaf expanded
 * by simulation s (0-based) becomes block s+1 in EVERY tree (the search runs in lockstep), so a
 * hidden-state tensor shaped [N+1][T][H] is addressed by (block, tree) with no per-tree allocator.
 */
#ifndef MZHIP_H
#define MZHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MZHIP_VERSION 1

typedef enum mz_status {
    MZ_OK = 0,
    MZ_ERR_INVALID_ARG = -1,
    MZ_ERR_OOM = -2,
    MZ_ERR_HIP = -3,
    MZ_ERR_STATE = -4,      /* call sequence violated, e.g. more simulations than max_simulations */
    MZ_ERR_NO_DEVICE = -5
} mz_status;

typedef struct mz_pool mz_pool_t;
typedef struct mz_net mz_net_t;

/* Search configuration: MCTSConfig + GameConfig + value KnownBounds (reference config.py:43-53,91-102). */
typedef struct mz_pool_config {
    int32_t num_trees;         /* T >= 1 */
    int32_t num_actions;       /* A in [1, 4096] */
    int32_t max_simulations;   /* N >= 1; the pool holds N+1 blocks per tree */
    int32_t num_players;       /* 1 or 2; 2 negates the discount (agent.py:150-152) */
    int32_t has_known_bounds;  /* 0: live MinMaxStats (utils.py:63-66); 1: use known_min/known_max */
    int32_t lanes_per_tree;    /* 0: the library chooses; else 2,4,...,64: GPU lanes that share one tree (a tuning
                                  knob: results do not depend on it) */
    double discount;           /* GameConfig.discount */
    double default_value;      /* MCTSConfig.default_value: exploitation score of unvisited children */
    double known_min;          /* KnownBounds.endpoints() (utils.py:45-49) */
    double known_max;
    const double* log_table;   /* host, N+2 entries, log_table[n] = 2*log(n+1) (config.py:159-160);
                                  NULL: the library tabulates it with the host libm */
} mz_pool_config_t;

const char* mz_last_error(void);
int mz_version(void);
/* Number of visible HIP devices (0 when there is none); never initialises a device context. */
int mz_device_count(void);

/* ---- tree pool ------------------------------------------------------------------------------------ */

/* Allocates the SoA tree pool on the current HIP device. stream: a hipStream_t (NULL = default stream). */
int mz_pool_create(const mz_pool_config_t* config, void* stream, mz_pool_t** out_pool);
int mz_pool_destroy(mz_pool_t* pool);
int mz_pool_set_stream(mz_pool_t* pool, void* stream);

/* New root + MinMaxStats for every tree, no tree reuse (MCTSAgent.make_move, agent.py:220-221). */
int mz_pool_reset(mz_pool_t* pool);

/* MCTSAgent.expand_node on the roots (agent.py:154-160,223-225).
 *   priors  [T,A] f32: NetworkOutput.masked_policy(legal) as the reference computes it -- FULL length A,
 *           zero at illegal actions.  Root child k (action legal[t,k]) receives priors[t,k]: the
 *           reference zips the legal list with the full-length policy, i.e. positionally.
 *   legal   [T,A] i32: legal actions of tree t in legal[t, 0..n_legal[t]); NULL = every action legal.
 *   n_legal [T]   i32: NULL = A.  n_legal[t] == 0 marks tree t inactive: every later call skips it and
 *           reports parent = action = -1 (the reference would raise on an un-expandable root). */
int mz_expand_roots(mz_pool_t* pool, const float* priors, const int32_t* legal, const int32_t* n_legal);
/* Same, computing masked_policy on the device from raw logits [T,A] f32 (network.py:142-146: exp(logit),
 * mask, NumPy-pairwise fp32 sum, IEEE divide, no max-subtraction).  expf differs from NumPy's SIMD exp
 * by <= 2 ulp, so this entry is tolerance-tested, not bit-exact. */
int mz_expand_roots_logits(mz_pool_t* pool, const float* logits, const int32_t* legal, const int32_t* n_legal);

/* MCTSAgent.add_exploration_noise (agent.py:170-175): prior = prior*(1-frac) + noise*frac in fp64.
 *   noise [T,A] f64: Dirichlet draw of tree t in noise[t, 0..n_legal[t]) (injected: the reference draws it
 *   with np.random.dirichlet on the host). */
int mz_add_root_noise(mz_pool_t* pool, const double* noise, double frac);

/* MCTSAgent.select_leaf for every tree (agent.py:177-190), one sub-wave group per tree.
 *   out_parent_block [T] i32: block of leaf.parent -- the row of the [N+1][T][H] hidden tensor to feed
 *                             recurrent_inference with (agent.py:204);
 *   out_action       [T] i32: the last action on the path (agent.py:205).
 * The selection path is kept in the pool for the following mz_expand_backprop. */
int mz_select(mz_pool_t* pool, int32_t* out_parent_block, int32_t* out_action);
/* The selection of ONE tree given by the caller instead of computed by mz_select: select_leaf started at a node below
 * the root (reference agent.py:177-181 takes any node; its backpropagate then still walks up to the root, agent.py:192-196).
 * blocks_host / actions_host (HOST, `length` entries): the chain from the root to the leaf -- entry i is the record
 * (expanded node) and the action of the child taken in it; blocks[0] = 0, every child but the last must be expanded with
 * blocks[i+1] as its record, the last one unvisited.  Writes that tree's path as mz_select would have, and
 * out_parent_block[tree] / out_action[tree].  The pool must hold one tree, or a pending mz_select for all trees (whose
 * selection of `tree` this call replaces).  MZ_ERR_INVALID_ARG if the chain is not one.  Synchronises the stream. */
int mz_pool_set_selection(mz_pool_t* pool, int32_t tree, const int32_t* blocks_host, const int32_t* actions_host,
                          int32_t length, int32_t* out_parent_block, int32_t* out_action);

/* expand_node(leaf, all actions) + backpropagate(leaf, value) (agent.py:208-211, 192-196) for the leaf of
 * the last mz_select.  priors [T,A] f32 (masked_policy over all A actions), reward [T] f32, value [T] f32
 * are the already-scalarised network outputs (split_batch, network.py:165-174).  The new node becomes
 * block (number of simulations done so far + 1) in every tree. */
int mz_expand_backprop(mz_pool_t* pool, const float* priors, const float* reward, const float* value);
int mz_expand_backprop_logits(mz_pool_t* pool, const float* logits, const float* reward, const float* value);

/* mz_expand_backprop immediately followed by the next mz_select in ONE launch (halves the tree launches
 * of run_mcts; results are identical to the two separate calls). */
int mz_step(mz_pool_t* pool, const float* priors, const float* reward, const float* value,
            int32_t* out_parent_block, int32_t* out_action);
int mz_step_logits(mz_pool_t* pool, const float* logits, const float* reward, const float* value,
                   int32_t* out_parent_block, int32_t* out_action);

/* Search statistics of the roots (MCTSAgent.make_move, agent.py:229-231). Any output may be NULL.
 *   visits     [T,A] i32: visit count of the root child for each ACTION (0 where illegal);
 *   root_value [T]   f64: root.value = value_sum / visit_count (nan before the first simulation);
 *   minmax     [T,2] f64: MinMaxStats.minimum, .maximum (+inf, -inf while untouched / with known bounds);
 *   policy     [T,A] f32: visits / root visit count, computed in fp64 and rounded to fp32 like
 *                         tf.constant(python floats). */
int mz_root_stats(mz_pool_t* pool, int32_t* visits, double* root_value, double* minmax, float* policy);

/* MCTSAgent.select_action / softmax_sample (agent.py:162-168,213-217).
 *   u           [T] f64: the uniform draw random.random() that random.choices would consume (injected);
 *   temperature [T] f64: visit_softmax_temperature_fn per tree (config.py:153-157); 0 = arg-max with ties
 *                        to the largest action; otherwise weights count**(1/t), cumulative fp64 sums and
 *                        bisect_right(cum, u*total, 0, n-1) like CPython's random.choices.
 *   out_action  [T] i32. */
int mz_select_action(mz_pool_t* pool, const double* u, const double* temperature, int32_t* out_action);

/* Exact sampling weights for a temperature t.  softmax_sample raises the visit counts to 1/t with the HOST libm's pow
 * (`count ** (1 / temperature)`, agent.py:167; t comes from visit_softmax_temperature_fn, config.py:153-157, or the
 * --temperature override, agent_services.py:21-24), which a device pow does not reproduce bit for bit -- so the host
 * tabulates it, like the exploration table of mz_pool_config_t:
 *   weights_host: HOST pointer, max_simulations+1 doubles, weights_host[c] = (double)c ** (1/t) evaluated by the caller.
 * mz_select_action and mz_search_move use the table for every tree whose temperature equals t (exact f64 match).
 * A pool holds 8 tables; a ninth temperature replaces the oldest, registering the same t again overwrites its table.
 * t = 0 (arg-max) and 1/t in {1, 2, 4} (exact products) need no table; any other unregistered temperature falls back
 * to the device's pow (within 1 ulp of libm, not bit-exact).  Synchronises the pool's stream. */
int mz_pool_set_temperature_table(mz_pool_t* pool, double temperature, const double* weights_host);

/* Measurement counters, host output, synchronises the stream.
 *   out[0] simulations backed up (summed over trees), out[1] sum over those simulations of d = number of
 *   expanded nodes on the selection path (root included), out[2] select launches, out[3] backprop launches. */
int mz_pool_counters(mz_pool_t* pool, int64_t* out4_host);

/* Inspection (Node views, tests): copies one block of one tree to HOST arrays of A entries each
 * (any may be NULL); synchronises.  block 0 = root (its priors are the fp64 noise-mixed ones).
 *   actions i32, count i32, value_sum f64, prior f64, reward f32 (valid where count>0), child i32 (block
 *   of the child where count>0, else -1).  *out_num_children receives the number of valid entries. */
int mz_pool_read_block(mz_pool_t* pool, int32_t tree, int32_t block, int32_t* out_num_children,
                       int32_t* actions, int32_t* count, double* value_sum, double* prior, float* reward,
                       int32_t* child);
/* Inspection: the roots' fp32 priors as expand_node stored them, BEFORE the noise mix (NetworkOutput.masked_policy's
 * k-th entry for the k-th listed child, agent.py:158-160; mz_pool_read_block reports the fp64 priors after it).
 *   priors: HOST, A floats, the first *out_num_children valid.  Synchronises. */
int mz_pool_read_root_priors(mz_pool_t* pool, int32_t tree, int32_t* out_num_children, float* priors);
/* Number of blocks currently in use per tree (= simulations done + 1 after mz_expand_roots, else 0). */
int mz_pool_num_blocks(mz_pool_t* pool);
/* Device bytes held by the pool. */
int64_t mz_pool_bytes(mz_pool_t* pool);

/* ---- networks (Network.initial_inference / recurrent_inference, reference network.py:209-254) ------- */

typedef enum mz_net_kind {
    MZ_NET_TICTACTOE = 1,   /* reference games/tictactoe.py:151-220 */
    MZ_NET_CARTPOLE = 2,    /* reference games/cartpole.py:63-139  */
    MZ_NET_ATARI_SHAPED = 3 /* no reference class: the synthetic network of BASELINE.json configs[3] (18 actions, hidden
                               state (2,128) = 256 floats, conv-residual dynamics), assembled from the reference's blocks
                               (one_hot_tensor_encoder network.py:14-40; conv + dense stacks as games/tictactoe.py:188-213);
                               definition: DESIGN.md and oracle/net_oracle.py AtariShapedOracle */
} mz_net_kind;

/* Number of f32 weights the flat weight buffer of a network kind holds (layout: DESIGN.md "weights"). */
int64_t mz_net_num_weights(int32_t kind);
/* Hidden-state floats per game for a network kind (TicTacToe 3*3*128, CartPole 128, Atari-shaped 2*128). */
int32_t mz_net_hidden_size(int32_t kind);
/* Action-space size (9, 2, 18) and observation floats per game (27, 4, 256) of a network kind; -1 for an unknown kind. */
int32_t mz_net_num_actions(int32_t kind);
int32_t mz_net_observation_size(int32_t kind);

/* Uploads (and re-lays-out for the kernels) a flat f32 weight buffer in Keras order.
 *   weights_host: HOST pointer, mz_net_num_weights(kind) floats;  training_steps: Network.training_steps().
 * Numeric domain.  The kernels multiply on f16 matrix cores with every f32 operand carried as two halfs,
 * v = h + l/2048 (h = f16(v), l = f16((v - h) * 2048), both rounded toward zero): products and sums are exact to f32
 * accumulation, results within 1e-5 of an fp64 evaluation (relative to each output tensor's scale) -- provided every
 * weight and every activation (hidden states, layer outputs, observations fed to a matrix stage) satisfies
 * |v| <= 65504, f16's largest finite value; below 6.1e-5 a value is carried with absolute precision 2.9e-11 instead of
 * relative precision 2^-21.  A weight outside the range (or not finite) is rejected here with MZ_ERR_INVALID_ARG; an
 * activation outside it saturates at 65504 and raises the handle's range flag (mz_net_status).  (CartPole batches above
 * 8192 run on f32 matrix instructions and have no such limit.) */
int mz_net_create(int32_t kind, const float* weights_host, int64_t num_weights, int64_t training_steps,
                  void* stream, mz_net_t** out_net);
int mz_net_destroy(mz_net_t* net);
int mz_net_set_stream(mz_net_t* net, void* stream);
int64_t mz_net_training_steps(mz_net_t* net);
/* Health of the inferences run on this handle since the last call: *out_flags_host (HOST) receives bit 0 = some
 * activation left the kernels' numeric domain (|v| > 65504, see mz_net_create) and was saturated -- outputs computed
 * from it are not to the stated tolerance.  Clears the flag.  Synchronises the handle's stream. */
int mz_net_status(mz_net_t* net, int32_t* out_flags_host);
/* Measurement hook (no counterpart in the reference): when hip_event (a hipEvent_t) is non-NULL, every later
 * inference call that consists of two launches (TicTacToe: per-cell kernel, then head kernel) records it on the
 * stream between them, so a caller bracketing the call with its own events can time each kernel. NULL disables. */
int mz_net_set_mid_event(mz_net_t* net, void* hip_event);
/* Test / measurement switches of mz_infer_recurrent_run for the TicTacToe network (no counterpart in the reference; the
 * library itself never changes them): mode 0 (default) = all simulations in one launch where the pool allows it; 1 = the
 * same launch with full agent-scope fences at every meeting of a row block's workgroups (the path taken by itself when
 * they are found on different XCDs); 2 = a launch per simulation (the kernels of mz_infer_recurrent_step).  All three
 * give bit-identical trees and outputs.
 * mz_debug_run_info: out4_host[0] = row-block runs that took the fence path since the handle was created, [1] = row-block
 * runs in all, [2] = 100 MHz wall-clock ticks the workgroups of the one-launch move spent inside their jobs, [3] = the
 * number of jobs (workgroups x simulations) those ticks belong to.  Synchronises. */
int mz_debug_set_run_mode(mz_net_t* net, int32_t mode);
/* Test / measurement switch of the TicTacToe job kernel's shape: rows = 0 (default) chooses by batch size -- 16 games x one
 * board cell per workgroup up to 768 games, 32 games x one cell up to 2560, 64 games x a group of cells beyond --, 16, 32
 * or 64 force that shape.  The single-cell shapes give bit-identical results; the 64-game shape sums the prediction
 * conv's per-cell terms in a different order: it agrees with them to float rounding (all are held to 1e-5 of the fp64
 * oracle), not bit for bit.  The initial inference follows the same choice: batches of the 64-game shape take a 64-game
 * kernel whose 3 -> 128 conv runs on the matrix cores (agreement with the 32-game kernel of the smaller batches: float
 * rounding, both within 1e-5 of the oracle). */
int mz_debug_set_job_rows(mz_net_t* net, int32_t rows);
int mz_debug_run_info(mz_net_t* net, int64_t* out4_host);

/* Network.initial_inference (network.py:221-229) fused with split_batch's to_scalar (network.py:165-174).
 *   observation [B, obs] f32 (TicTacToe 27 = 3x3x3 NHWC, CartPole 4, Atari-shaped 256 = 2x128);
 *   out_hidden  [B, H] f32;  out_value [B] f32 (scalar: identity or support_to_scalar, utils.py:146-148);
 *   out_logits  [B, A] f32;  out_value_raw: NULL or the un-scalarised head ([B,1] or [B,101]). */
int mz_infer_initial(mz_net_t* net, int32_t batch, const float* observation, float* out_hidden,
                     float* out_value, float* out_logits, float* out_value_raw);

/* Network.recurrent_inference (network.py:246-254) fused with the state-action encoder
 * (network.py:14-40,88-104), the parent-hidden-state gather of run_mcts (agent.py:204) and to_scalar.
 *   hidden_base f32, parent_block [B] i32 (NULL = all zero), block_stride (floats between consecutive
 *       blocks): entry b reads its H input floats at hidden_base + parent_block[b]*block_stride + b*H.
 *       Pass the [N+1][T][H] hidden tensor, mz_select's out_parent_block and block_stride = T*H to gather
 *       leaf.parent.hidden_state in the kernel's operand load; pass NULL / 0 for a plain [B,H] batch.
 *       A negative parent_block skips that entry (inactive tree): its outputs are left untouched;
 *   action [B] i32 in [0, A);
 *   out_hidden [B,H] (e.g. block s+1 of the hidden tensor); out_reward [B]; out_value [B] (scalarised);
 *   out_logits [B,A]; out_value_raw: NULL or the un-scalarised value head ([B,1] or [B,101]). */
int mz_infer_recurrent(mz_net_t* net, int32_t batch, const float* hidden_base, const int32_t* parent_block,
                       int64_t block_stride, const int32_t* action, float* out_hidden, float* out_reward,
                       float* out_value, float* out_logits, float* out_value_raw);

/* mz_infer_recurrent for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for callers that
 * cannot call hipMalloc -- code under HIP graph capture, tight serving loops, harnesses that forbid allocation.
 * Same arguments, same kernels and bit-identical results as mz_infer_recurrent, but the kernels' scratch (TicTacToe:
 * the prediction-conv partials and reward partial dots that the job kernel hands to the head kernel) is a static device
 * array inside the library (MZ_NOALLOC_MAX_BATCH games, 4.76 MB per GPU, resolved once per device by mz_net_create)
 * instead of hipMalloc'ed memory: the call reaches no hipMalloc / hipFree or allocating library routine.
 *   - TicTacToe: batch > MZ_NOALLOC_MAX_BATCH is rejected with MZ_ERR_INVALID_ARG (split the batch, or use
 *     mz_infer_recurrent); the CartPole kernels use no scratch and take any batch;
 *   - every call on one GPU shares that GPU's array: two calls that overlap on one GPU (different streams or host
 *     threads) race -- serialise them.  Nothing in the array has to be cleared between calls. */
#define MZ_NOALLOC_MAX_BATCH 1024
int mz_infer_recurrent_noalloc(mz_net_t* net, int32_t batch, const float* hidden_base, const int32_t* parent_block,
                               int64_t block_stride, const int32_t* action, float* out_hidden, float* out_reward,
                               float* out_value, float* out_logits, float* out_value_raw);

/* mz_infer_recurrent over all T trees of `pool` immediately followed by the tree update with its outputs:
 * mz_step_logits (with_select != 0: results land in parent_block / action, which are read first as this
 * simulation's leaf parents and actions) or mz_expand_backprop_logits (with_select == 0, the last simulation).
 * Same results as the separate calls; for the TicTacToe network the head of the inference and the tree step run as
 * one launch (each workgroup owns 16 games end to end).  net and pool must use the same device and stream. */
int mz_infer_recurrent_step(mz_net_t* net, mz_pool_t* pool, int32_t with_select, const float* hidden_base,
                            int32_t* parent_block, int64_t block_stride, int32_t* action, float* out_hidden,
                            float* out_reward, float* out_value, float* out_logits);

/* run_mcts (reference agent.py:198-211) for all T trees of `pool`: `num_simulations` times { recurrent inference of the
 * selected leaves' parents -> expand_node -> backpropagate -> select_leaf of the next simulation }, i.e. that many
 * mz_infer_recurrent_step calls with with_select = 1, the last one with with_select = final_select.  The pool must hold a
 * pending selection in parent_block / action (mz_select or a step with select).  Simulation i writes the hidden rows of
 * the record it creates, block b = (simulations done so far) + 1 + i, to hidden_base + b * block_stride, and its outputs to
 * out_reward / out_value + i * out_stride and out_logits + i * out_stride * A (out_stride = 0: every simulation overwrites
 * the same T rows; else >= T).  trace_parent / trace_action (both or neither, [num_simulations][T]) receive the selection
 * each simulation started from.  Same results as the separate calls.  For the CartPole network (pools of <= 8192 trees
 * with the default lanes per tree and max_simulations <= 62, so that 16 trees fit a workgroup's LDS) the whole run is ONE
 * launch: a workgroup keeps its 16 games from the first simulation to
 * the last, so no launch boundary, no host call and no other workgroup's data separates two simulations; so it is for
 * the Atari-shaped network (64 games a workgroup, pools of 18 actions), and for the TicTacToe network with the default
 * 16 lanes per tree and more than 2560 trees (the four workgroups of 64 games stay resident and hand each other their
 * results through their XCD's L2 twice per simulation; if that ever fails to complete within a second the launch gives
 * up and mz_pool_counters reports MZ_ERR_HIP); other networks and pools -- smaller TicTacToe batches among them -- run it
 * as the sequence of mz_infer_recurrent_step launches.  net and pool must use the same device and stream. */
int mz_infer_recurrent_run(mz_net_t* net, mz_pool_t* pool, int32_t num_simulations, int32_t final_select,
                           float* hidden_base, int32_t* parent_block, int64_t block_stride, int32_t* action,
                           float* out_reward, float* out_value, float* out_logits, int64_t out_stride,
                           int32_t* trace_parent, int32_t* trace_action);

/* ---- one move of T games, enqueued natively (batched MCTSAgent.make_move, reference agent.py:219-232) ------- */

typedef struct mz_move_io {
    int32_t num_trees;             /* T: must equal the pool's */
    int32_t num_simulations;       /* N <= the pool's max_simulations */
    int32_t hidden_size;           /* H of the network */
    int32_t reserved;
    void* stream;                  /* the hipStream_t the pool and the net launch on (for the timing events) */
    const float* observation;      /* [T, obs] f32 */
    const int32_t* legal;          /* [T, A] i32 or NULL (all legal) */
    const int32_t* n_legal;        /* [T] i32 or NULL */
    const double* noise;           /* [T, A] f64 Dirichlet draws (agent.py:172) */
    double noise_frac;             /* MCTSConfig.root_exploration_fraction */
    const double* u;               /* [T] f64 uniforms for the action sampling */
    const double* temperature;     /* [T] f64 */
    float* hidden;                 /* workspace of [N+1][T][H] floats: block s+1 = hidden states of simulation s.
                                      Scratch: its contents after the call are in the network's internal format
                                      (TicTacToe: split-f16 images of the f32 rows), not f32 hidden states */
    float* logits;                 /* workspace [T, A] f32 */
    float* reward;                 /* workspace [T] f32 */
    float* value;                  /* workspace [T] f32 */
    int32_t* parent_block;         /* workspace [T] i32 */
    int32_t* action;               /* workspace [T] i32 */
    int32_t* out_visits;           /* [T, A] i32 or NULL */
    double* out_root_value;        /* [T] f64 or NULL */
    double* out_minmax;            /* [T, 2] f64 or NULL */
    float* out_policy;             /* [T, A] f32 or NULL */
    int32_t* out_action;           /* [T] i32 or NULL */
    void** events;                 /* NULL, or HOST array of 4*N hipEvent_t (measurement only); for every simulation s
                                      whose events[4s] is not NULL: [4s] is recorded before the inference, [4s+1] between
                                      its two launches (if it has two), [4s+2] after the tree step -- which shares the
                                      second launch for the TicTacToe network, so [4s+1]..[4s+2] spans head + tree step
                                      there; [4s+3] is unused.  A record costs ~13 us of stream time: sample sparsely.
                                      Where mz_infer_recurrent_run is ONE launch (CartPole, Atari-shaped), only the events of the first
                                      simulation that has any are recorded: [4s] before and [4s+1], [4s+2] after that launch,
                                      which holds all N simulations */
    /* Optional search trace (tests, debugging; every pointer NULL = off, all five non-NULL = on): what simulation s
     * fed to and got from the network is kept in row s instead of being overwritten in the one-row workspaces above.
     * The launches are the same, only their output pointers differ. */
    float* trace_reward;           /* [N, T] f32 */
    float* trace_value;            /* [N, T] f32 */
    float* trace_logits;           /* [N, T, A] f32 */
    int32_t* trace_parent;         /* [N, T] i32: block of leaf.parent of simulation s (-1: inactive tree) */
    int32_t* trace_action;         /* [N, T] i32: the action leading to the leaf */
} mz_move_io_t;

/* make_move for every tree of the pool: initial_inference, root expansion (+noise), N simulations with the network in
 * the loop (mz_infer_recurrent_run), root statistics and action sampling -- 2N+5 (TicTacToe) or 5 (CartPole,
 * Atari-shaped: any N) launches on the stream, no host synchronisation.
 * io is checked against the handles (MZ_ERR_INVALID_ARG): num_trees == the pool's, num_simulations <= its max_simulations,
 * hidden_size and the action count == the network's, and pool, net and io->stream name one stream. */
int mz_search_move(mz_pool_t* pool, mz_net_t* net, const mz_move_io_t* io);

/* The tree half of one move against a network whose outputs are given: run_mcts (agent.py:198-211) with
 * recurrent_inference replaced by table look-ups, the way the reference's MCTS is exercised with a stub network and the
 * way the tree kernels are measured alone.  Root expansion (+noise when noise != NULL), then N simulations as N+1
 * launches on the pool's stream, no host synchronisation; root statistics / action sampling are the caller's next calls.
 *   policy [N+1, T, A] f32 device: row 0 the root's masked policy, row s the policy of simulation s's new node
 *                                  (logits when is_logits != 0, see mz_expand_backprop_logits)
 *   reward, value [N+1, T] f32 device (row 0 unused); legal/n_legal as mz_expand_roots; noise as mz_add_root_noise
 *   parent_block, action [T] i32 device workspaces (the last select's outputs)
 *   The N simulations are ONE launch (a wavefront keeps its trees from the first simulation to the last: no launch
 *   boundary and no other tree's depth separates two of its simulations) unless single steps are to be timed:
 *   events: NULL, or HOST array of 2*(N-1) hipEvent_t (measurement only; a record costs stream time).  With only
 *           events[0] and events[2(N-2)+1] set, they are recorded before and after that one launch.  Any other non-NULL
 *           entry asks for the launch per simulation: events[2(s-1)] is then recorded before fused step s (1-based),
 *           events[2(s-1)+1] after it. */
int mz_search_move_injected(mz_pool_t* pool, int32_t num_simulations, const float* policy, int32_t is_logits,
                            const float* reward, const float* value, const int32_t* legal, const int32_t* n_legal,
                            const double* noise, double noise_frac, int32_t* parent_block, int32_t* action,
                            void** events);

/* ---- batched environments on the device (the callers either side of a move) --------------- *
 * State is caller-owned SoA over num_games games; every call enqueues one launch on `stream` (a hipStream_t).
 * A finished game is frozen: step leaves it untouched and observe reports n_legal = 0, which mz_expand_roots
 * and the inference entry points treat as "inactive".
 *
 * TicTacToe (reference games/tictactoe.py:71-148):
 *   board [G,9] i8 (-1 empty, else the player id), to_play [G] i32, steps [G] i32, ended [G] i32 (0/1).
 *   observe: out_observation [G,27] f32 = planes [board==0, board==1, to_play] stacked last (NHWC 3x3x3,
 *            games/tictactoe.py:107-111); out_legal [G,9] i32 = empty cells ascending, padded with -1;
 *            out_n_legal [G] i32 (0 once ended).  Any output may be NULL.
 *   step:    plays action[g] for to_play[g]; out_reward [G] f32 = 1 on a winning move else 0; the game ends on a
 *            win or after nine moves.  out_status [G] i32: 0 moved, 1 already ended, 2 illegal action (game left
 *            untouched; the reference raises MuZeroEnvironmentError, games/tictactoe.py:122-123); any_illegal [1]
 *            i32 is OR-ed with 1 when some game saw an illegal action.  out_reward/out_status/any_illegal may be NULL. */
int mz_env_ttt_reset(void* stream, int32_t num_games, int8_t* board, int32_t* to_play, int32_t* steps, int32_t* ended);
int mz_env_ttt_observe(void* stream, int32_t num_games, const int8_t* board, const int32_t* to_play,
                       const int32_t* ended, float* out_observation, int32_t* out_legal, int32_t* out_n_legal);
int mz_env_ttt_step(void* stream, int32_t num_games, int8_t* board, int32_t* to_play, int32_t* steps, int32_t* ended,
                    const int32_t* action, float* out_reward, int32_t* out_status, int32_t* any_illegal);

/* CartPole-v1 (reference environment.py:44-63 wraps gym.make('CartPole-v1'); gym is third-party and absent, so
 * its classic-control equations are restated: Euler step tau = 0.02, force +-10, episode ends at |x| > 2.4,
 * |theta| > 12 degrees or 500 steps; reward 1 per step).
 *   state [G,4] f64 (x, x_dot, theta, theta_dot), steps [G] i32, ended [G] i32;
 *   reset copies initial_state [G,4] f64 (the caller draws U(-0.05, 0.05) like gym's reset);
 *   observe: out_observation [G,4] f32 (environment.py:59), out_legal [G,2] = {0,1}, out_n_legal = 2 or 0;
 *   step: as above with actions {0,1}. */
int mz_env_cartpole_reset(void* stream, int32_t num_games, double* state, int32_t* steps, int32_t* ended,
                          const double* initial_state);
int mz_env_cartpole_observe(void* stream, int32_t num_games, const double* state, const int32_t* ended,
                            float* out_observation, int32_t* out_legal, int32_t* out_n_legal);
int mz_env_cartpole_step(void* stream, int32_t num_games, double* state, int32_t* steps, int32_t* ended,
                         const int32_t* action, float* out_reward, int32_t* out_status, int32_t* any_illegal);

/* Restart of the games with where[g] != 0 only (every other game is left as it is): a finished self-play slot starts
 * its next game while the others play on, like the reference's agent loop (agent_services.py:47-48). */
int mz_env_ttt_reset_where(void* stream, int32_t num_games, int8_t* board, int32_t* to_play, int32_t* steps, int32_t* ended,
                           const int32_t* where);
int mz_env_cartpole_reset_where(void* stream, int32_t num_games, double* state, int32_t* steps, int32_t* ended,
                                const double* initial_state, const int32_t* where);

/* ---- game histories of G self-play slots and the hand-over of finished games -------- *
 * GameHistory (reference game.py:8-41) for G concurrent games as padded move-major device arrays, M = move capacity:
 *   observations [M+1,G,obs] f32 | to_plays [M+1,G] i32 | actions [M,G] i32 | rewards [M,G] f32 | root_values [M,G] f64 |
 *   policies [M,G,A] f32 | lengths [G] i32.
 * Slots advance independently: slot g is at move steps[g] of ITS game (the environment's step counter).
 *   record_observation: observation/to_play of every slot -> row steps[g] (Game.__init__/apply, game.py:52-61,66-71);
 *   record_move: for slots whose step succeeded (status[g] == 0), with steps already advanced by the step:
 *                action, reward, root value, policy -> row steps[g]-1 (game.py:66-75);
 *   flush: slots whose game is over (ended[g] != 0, or steps[g] == M: agent.py:34's max_moves) hand their history to the
 *          `finished` store, column = the game's serial number (serial [G] i32; games with serial >= the store's
 *          num_games are dropped), and receive the next free serials in slot order; counters [2] i64 = {next serial,
 *          games stored}; flush_dst [G] i32 (workspace) and reset_mask [G] i32 (1 where the slot must be restarted:
 *          pass it to the environment's reset_where).  What the reference's agent does between two games:
 *          remote_replay_buffer.save_history(history), then the next play_game (agent_services.py:47-54). */
typedef struct mz_history {
    int32_t num_games, max_moves, observation_size, num_actions;
    float* observations;
    int32_t* to_plays;
    int32_t* actions;
    float* rewards;
    double* root_values;
    float* policies;
    int32_t* lengths;
} mz_history_t;
int mz_history_record_observation(void* stream, const mz_history_t* hist, const int32_t* steps, const float* observation,
                                  const int32_t* to_play);
int mz_history_record_move(void* stream, const mz_history_t* hist, const int32_t* steps, const int32_t* status,
                           const int32_t* action, const float* reward, const double* root_value, const float* policy);
int mz_history_flush(void* stream, const mz_history_t* live, const mz_history_t* finished, const int32_t* steps,
                     const int32_t* ended, int32_t* serial, int64_t* counters, int32_t* flush_dst, int32_t* reset_mask);

/* The same bookkeeping with the environment inside, for a batch that is refilled every move (SelfPlayEngine.
 * play_continuous): fourteen small launches a move become four.  No counterpart of their own in the reference: they are
 * the calls above and the mz_env_* calls, fused.
 *   mz_env_t: one of the two batched environments as caller-owned state (the arrays of mz_env_ttt_* / mz_env_cartpole_*);
 *       restart_state [G,4] f64 = the states restarted CartPole games begin in (NULL for TicTacToe).
 *   mz_selfplay_turnover = observe -> mz_history_record_observation -> mz_history_flush -> reset of the finished slots ->
 *       observe -> record_observation -> mz_dirichlet_from_gamma (when gamma != NULL), and out_temperature[g] =
 *       temperature while the slot's game has played fewer than freezing_moves moves, else 0 (config.py:153-157).
 *       Outputs: the search's inputs for the next move.
 *   mz_selfplay_step_record = mz_env_*_step -> mz_history_record_move. */
enum { MZ_ENV_TICTACTOE = 1, MZ_ENV_CARTPOLE = 2 };
typedef struct mz_env {
    int32_t kind, num_games;
    void* state;                 /* TicTacToe: int8 board [G,9]; CartPole: f64 [G,4] */
    int32_t* to_play;            /* [G] (TicTacToe; ignored for CartPole) */
    int32_t* steps;              /* [G] */
    int32_t* ended;              /* [G] */
    const double* restart_state; /* CartPole: [G,4] */
} mz_env_t;
int mz_selfplay_turnover(void* stream, const mz_env_t* env, const mz_history_t* live, const mz_history_t* finished,
                         int32_t* serial, int64_t* counters, int32_t* flush_dst, int32_t* reset_mask, const double* gamma,
                         int32_t freezing_moves, double temperature, float* out_observation, int32_t* out_legal,
                         int32_t* out_n_legal, double* out_noise, double* out_temperature);
int mz_selfplay_step_record(void* stream, const mz_env_t* env, const mz_history_t* live, const int32_t* action,
                            const double* root_value, const float* policy, float* out_reward, int32_t* out_status,
                            int32_t* any_illegal);

/* The root noise of a batch of games from independent Gamma(alpha, 1) draws (the caller's RNG): out_noise[g, k] =
 * gamma[g, k] / sum(gamma[g, 0..n_legal[g])) for k < n_legal[g], else 0 -- np.random.dirichlet([alpha] * n_children)
 * (agent.py:172) for every game in one launch, in the layout mz_add_root_noise / mz_search_move take.  [G, A] f64. */
int mz_dirichlet_from_gamma(void* stream, int32_t num_games, int32_t num_actions, const double* gamma,
                            const int32_t* n_legal, double* out_noise);

/* ---- replay-side training targets (the step after the path) -------------------------------- *
 * ReplayBuffer.preprocess_history + compute_target_value (reference replay_buffer.py:66-99) for G finished games
 * held as padded move-major arrays (muzero_amd.selfplay.BatchedHistory), M = move capacity, K = num_unroll_steps:
 *   lengths [G] i32; actions [M,G] i32; rewards [M,G] f32; root_values [M,G] f64; policies [M,G,A] f32;
 *   random_actions [G,K] i32: the random actions appended past the terminal state (random.randrange in the
 *       reference, replay_buffer.py:78; injected);
 *   discount_powers f64, max(td_steps, M)+1 entries: effective_discount**i evaluated by the host (Python pow);
 *   out_extended_actions [M+K,G] i32; out_target_rewards [M+K,G,Dr] f32; out_target_values [M+K+1,G,Dv] f32;
 *   out_target_policies [M+K+1,G,A] f32 (uniform rows past the end); out_total_value [G] f64 or NULL.
 *   Dr/Dv = 1 for a scalar head (support 0) else support+1 (two-hot, utils.py:138-143 with config.py:34-36 bounds).
 * Entries of game g beyond its own length (+K, +K+1) are zero padding. */
typedef struct mz_replay_config {
    int32_t num_games, max_moves, num_actions, num_unroll_steps, td_steps;
    int32_t value_support;      /* 0 = ScalarConfig.support_size None */
    int32_t reward_support;
    int32_t reserved;
    double value_min, value_max, reward_min, reward_max;   /* KnownBounds.endpoints(); used only with a support */
} mz_replay_config_t;
int mz_replay_targets(void* stream, const mz_replay_config_t* config, const int32_t* lengths, const int32_t* actions,
                      const float* rewards, const double* root_values, const float* policies,
                      const int32_t* random_actions, const double* discount_powers, int32_t* out_extended_actions,
                      float* out_target_rewards, float* out_target_values, float* out_target_policies,
                      double* out_total_value);

#ifdef __cplusplus
}
#endif
#endif /* MZHIP_H */
