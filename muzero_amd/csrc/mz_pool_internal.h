// Host-side state of a tree pool, shared by mz_tree.hip and the fused inference + tree-step launch in mz_net.hip.
#pragma once
#include "mz_tree_dev.h"
#include <vector>

struct mz_pool {
    PoolDev dev{};
    hipStream_t stream = nullptr;
    int device = 0;
    int G = 2;                 // lanes per tree
    int P = 1;                 // children per lane loaded together by the descending kernels
    int blocks_in_use = 0;     // 0 = roots not expanded; else simulations done + 1
    bool pending = false;      // a select is waiting for its expand_backprop
    long long select_launches = 0, backprop_launches = 0;
    size_t bytes = 0;
    std::vector<void*> allocations;
    long long* counter_scratch = nullptr;
    double* pow_dev = nullptr;  // PoolDev::pow_table
    int pow_next = 0;           // next temperature table to replace once all kPowTables are taken
    PoolDev* dev_image = nullptr;   // `dev` in device memory, for kernels that read it phase by phase instead of holding
    PoolDev image_host{};           //   all of it in scalar registers for their whole run (mz_pool_device_image)
    bool image_valid = false;
};

// `pool->dev` as the device holds it (re-uploaded on the pool's stream when it has changed since the last call), or
// nullptr with mz_last_error set.
const void* mz_pool_device_image(mz_pool* pool);      // (a PoolDev: the type has no linkage across translation units)

// Call-sequence bookkeeping of one expand+backprop (fused with the next select when `with_select`): checks what
// mz_step / mz_expand_backprop check and returns the block the new nodes take; commit after a successful launch.
int mz_pool_begin_backprop(mz_pool* pool, bool with_select, int* out_new_block);
void mz_pool_commit_backprop(mz_pool* pool, bool with_select);

// The same for `num_simulations` expand+backprops in one launch (mz_infer_recurrent_run), each but the last fused with
// the next select, the last one too when `final_select`; returns the block of the first.
int mz_pool_begin_run(mz_pool* pool, int num_simulations, bool final_select, int* out_first_block);
void mz_pool_commit_run(mz_pool* pool, int num_simulations, bool final_select);
// run_mcts against given outputs in one launch (mz_tree.hip, run_injected_kernel)
int mz_pool_run_injected(mz_pool* pool, int num_simulations, bool final_select, const float* pol, bool logits,
                         const float* reward, const float* value, size_t pol_stride, size_t rv_stride, int32_t* out_parent,
                         int32_t* out_action);

// A native move (mz_search_move) brackets its inferences with these: begin = the initial inference (mz_infer_initial's
// arguments; hidden_block0 receives the root hidden states), after which and until end the network keeps the move's
// hidden workspace in an internal format (mz_net.hip).
struct mz_net;
extern "C" int mz_net_begin_move(mz_net* net, int batch, const float* observation, float* hidden_block0, float* out_value,
                                 float* out_logits);
extern "C" void mz_net_end_move(mz_net* net);
// mz_net_begin_move, then mz_expand_roots_noise (below) on the logits it produced, with the first selection: as few launches
// as the network allows
extern "C" int mz_net_begin_move_roots(mz_net* net, mz_pool* pool, int batch, const float* observation, float* hidden_block0,
                                       float* out_value, float* out_logits, const int32_t* legal, const int32_t* n_legal,
                                       const double* noise, double frac, int32_t* out_parent_block, int32_t* out_action);

// what mz_search_move checks a network handle against (mz_net.hip)
extern "C" int mz_net_hidden_size_of(const mz_net* net);
extern "C" int mz_net_num_actions_of(const mz_net* net);
extern "C" hipStream_t mz_net_stream_of(const mz_net* net);
// mz_infer_recurrent_run is one launch for this network and pool (else it loops over mz_infer_recurrent_step)
extern "C" int mz_net_run_is_one_launch(const mz_net* net, const mz_pool* pool);

// mz_expand_roots (is_logits = 0) or mz_expand_roots_logits followed by mz_add_root_noise (noise may be NULL) and, when
// out_parent_block is not NULL, by the first mz_select: one launch
extern "C" int mz_expand_roots_noise(mz_pool_t* pool, const float* pol, int32_t is_logits, const int32_t* legal,
                                     const int32_t* n_legal, const double* noise, double frac, int32_t* out_parent_block,
                                     int32_t* out_action);

// mz_root_stats followed by mz_select_action, one launch
extern "C" int mz_root_finish(mz_pool_t* pool, int32_t* visits, double* root_value, double* minmax, float* policy,
                              const double* u, const double* temperature, int32_t* out_action);
