// GPU-resident SoA tree pool and the lockstep MCTS kernels (gfx950, wave64).
//
// What this replaces in the reference (fidel-schaposnik/muzero, file:line relative to /root/reference):
//   Node                       agent.py:103-123   -> child slots of a per-node block in HBM
//   MinMaxStats                utils.py:52-76     -> one (min,max) fp64 pair per tree
//   expand_node/masked_policy  agent.py:154-160, network.py:142-146
//   add_exploration_noise      agent.py:170-175
//   select_leaf/child/ucb      agent.py:177-190
//   backpropagate              agent.py:192-196
//   run_mcts loop body         agent.py:198-211
//   select_action/softmax      agent.py:162-168,213-217
//
// Layout.  Tree t owns N+1 records ("blocks") of RS bytes (a multiple of 128), block e at blocks + (t*(N+1)+e)*RS:
//     select half:  Slot[A]     { q f64, visit_count i32, prior f32, child_hi i32, act i32 }   24 B, array of structures
//     backup half:  BackSlot[A] { value_sum f64, reward f32, pad }                             16 B, array of structures
// (mz_tree_dev.h).  One dependent step of a descent reads whole slots, one per lane; a backup touches ONE 16-byte backup
// entry per path node (its visit count travels in the path entry) and writes count and q of that slot.
// q = reward + discount*value_sum/visit_count, the exploitation term of ucb_score (agent.py:188), already normalised when
// the value bounds are known: computed once per backup with the same IEEE operations the reference performs at every
// look, so select divides only for visited children.
// Block e is the node expanded by simulation e-1 in every tree (lockstep), block 0 is the root.  Only visit_count, prior
// and act are initialised at expansion; q / child_hi / value_sum / reward of a slot are written by the first backup
// through it (visited <=> expanded) and never read before.  Below the root a node keeps its children by descending prior
// (slot 0 = action 0), so that select scores a prefix of the node (select_tree, mz_tree_dev.h).
// Root priors are fp64 after the noise mix (agent.py:175) and live in root_prior[T][A]; the root's action list is
// the act field of its slots.
//
// Execution.  One group of G lanes per tree (2..64, group_width below): 64/G trees per wavefront, lanes stride over the
// slots, (score, action) arg-max by DPP / permlane-swap steps inside the group (no LDS).  Every tree keeps the same
// group -> workgroup -> XCD assignment in every launch, so its records stay in that XCD's L2.
//
// Numerics (SURVEY.md traps T1-T8): fp64 scores, fp32 priors/rewards widened, no FMA contraction
// (compiled with -ffp-contract=off), 2*log(n+1) tabulated by the host libm, ties to the largest action,
// all-NaN scores pick the first child, IEEE fp64 division.
#include "mz_common.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <type_traits>
#include <vector>

#include "mz_tree_dev.h"

namespace {

// one child per lane: all 8 wavefronts per SIMD resident (64 VGPRs), so a 16384-tree launch needs one round of workgroups
#define MZ_WAVES(P)

template <int G, int P>
__global__ __launch_bounds__(kBlockThreads) MZ_WAVES(P) void select_kernel(PoolDev p, int* out_parent, int* out_action) {
    __shared__ double s_log[kLogLds];
    const int gid = blockIdx.x * kBlockThreads + threadIdx.x;
    const int tree = gid / G, gl = gid % G;
    const bool active = tree < p.T && p.n_root[tree] > 0;
    const LdsLog lds_log = stage_log_table(p, s_log);
    select_tree<G, P>(p, tree, active, gl, out_parent, out_action, lds_log);
}

// the LDS rows of this workgroup's groups (mz_tree_dev.h, TreeLds) and the one of the calling lane's group
#define MZ_TREE_LDS(G)                                                                                         \
    __shared__ __attribute__((aligned(16))) float s_rows[TreeLds<G>::kGroups * TreeLds<G>::kRow];              \
    float* lds_row = s_rows + (threadIdx.x / G) * TreeLds<G>::kRow

// What a backup needs of its tree, all requested together: whether it is active, the prefetch, the network's outputs.
template <int G, int PB>
struct BackupInputs {
    bool active;
    int n_root;
    BackupPrefetchT<PB> pf;
    float reward, value;
    const float* pol_row;
};
template <int G, int PB>
__device__ __forceinline__ BackupInputs<G, PB> backup_inputs(const PoolDev& p, int tree, int gl, const float* pol,
                                                             const float* reward, const float* value) {
    BackupInputs<G, PB> b;
    const bool in = tree < p.T;
    const int tc = in ? tree : 0;
    const int n_root = p.n_root[tc];
    b.pf = backprop_prefetch<G, PB>(p, tc, in, gl);
    b.reward = reward[tc];
    b.value = value[tc];
    b.pol_row = pol + (size_t)tc * p.A;
    b.active = in && n_root > 0;
    b.n_root = n_root;
    if (!b.active) {
        b.pf.len = 0;
        b.reward = 0.f;
        b.value = 0.f;
    }
    return b;
}

// The network outputs' policy row of every tree of the workgroup and the log table, requested FIRST, before the backup's
// inputs: they land in the round trip that brings the path entries, instead of costing one of their own later (the log
// table before the barrier, the logits at the start of the expansion).  stage_requests() issues the loads;
// stage_commit() parks them in LDS and ends with the workgroup barrier.  Rows wider than the LDS row stay in global memory.
template <int G>
struct EarlyStage {
    static constexpr int KL = TreeLds<G>::kMaxA / G;      // policy entries per lane
    float pv[KL];
    double lt[kLogLds / kBlockThreads];
};
template <int G>
__device__ __forceinline__ EarlyStage<G> stage_requests(const PoolDev& p, int tree, int gl, const float* pol) {
    EarlyStage<G> e;
    const bool row_ok = tree < p.T && p.A <= TreeLds<G>::kMaxA;
#pragma unroll
    for (int j = 0; j < EarlyStage<G>::KL; ++j) {
        const int k = gl + j * G;
        e.pv[j] = (row_ok && k < p.A) ? pol[(size_t)tree * p.A + k] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < kLogLds / kBlockThreads; ++j) {
        const int i = (int)threadIdx.x + j * kBlockThreads;
        e.lt[j] = (p.N + 2 <= kLogLds && i < p.N + 2) ? p.log_table[i] : 0.0;
    }
    return e;
}
// returns the LDS log table (or nullptr); *pol_row is redirected to the staged row when it was staged
template <int G>
__device__ __forceinline__ LdsLog stage_commit(const PoolDev& p, const EarlyStage<G>& e, int gl, float* s_pol_row,
                                               double* s_log, const float** pol_row) {
    if (p.A <= TreeLds<G>::kMaxA) {
#pragma unroll
        for (int j = 0; j < EarlyStage<G>::KL; ++j) {
            const int k = gl + j * G;
            if (k < p.A) s_pol_row[k] = e.pv[j];
        }
        *pol_row = s_pol_row;
    }
    const bool log_ok = p.N + 2 <= kLogLds;
    if (log_ok) {
#pragma unroll
        for (int j = 0; j < kLogLds / kBlockThreads; ++j) {
            const int i = (int)threadIdx.x + j * kBlockThreads;
            if (i < p.N + 2) s_log[i] = e.lt[j];
        }
    }
    __syncthreads();
    return log_ok ? (LdsLog)s_log : nullptr;
}
#define MZ_POL_LDS(G)                                                                                          \
    __shared__ float s_pol[TreeLds<G>::kGroups * TreeLds<G>::kMaxA];                                            \
    float* s_pol_row = s_pol + (threadIdx.x / G) * TreeLds<G>::kMaxA

template <int G, bool LOGITS>
__global__ __launch_bounds__(kBlockThreads) void backprop_kernel(PoolDev p, int nb, const float* pol,
                                                                 const float* reward, const float* value) {
    const int gid = blockIdx.x * kBlockThreads + threadIdx.x;
    const int tree = gid / G, gl = gid % G;
    MZ_TREE_LDS(G);
    constexpr int PB = TreeLds<G>::kPB;
    const BackupInputs<G, PB> b = backup_inputs<G, PB>(p, tree, gl, pol, reward, value);
    backprop_tree<G, LOGITS, PB>(p, tree, b.active, gl, nb, b.pol_row, b.reward, b.value, b.pf, lds_row);
}

// expand+backprop of simulation s fused with select of simulation s+1.  A tree's group lies inside one wavefront, whose
// vector memory operations are performed in program order, so the records it has just written are the ones it reads
// back: a wavefront-scope fence (compiler ordering only) is enough, and the wavefronts of a workgroup stay decoupled.
template <int G, int P, bool LOGITS>
__global__ __launch_bounds__(kBlockThreads) MZ_WAVES(P) void step_kernel(PoolDev p, int nb, const float* pol,
                                                             const float* reward, const float* value,
                                                             int* out_parent, int* out_action) {
    __shared__ double s_log[kLogLds];
    const int gid = blockIdx.x * kBlockThreads + threadIdx.x;
    const int tree = gid / G, gl = gid % G;
    MZ_TREE_LDS(G);
    MZ_POL_LDS(G);
    constexpr int PB = TreeLds<G>::kPB;
    const EarlyStage<G> early = stage_requests<G>(p, tree, gl, pol);
    const BackupInputs<G, PB> b = backup_inputs<G, PB>(p, tree, gl, pol, reward, value);
    const float* pol_row = b.pol_row;
    const LdsLog lds_log = stage_commit<G>(p, early, gl, s_pol_row, s_log, &pol_row);
    backprop_tree<G, LOGITS, PB>(p, tree, b.active, gl, nb, pol_row, b.reward, b.value, b.pf, lds_row);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    select_tree<G, P>(p, tree, b.active, gl, out_parent, out_action, lds_log);
}

// run_mcts against GIVEN network outputs (agent.py:198-211 with the network replaced by arrays) in ONE launch: the body of
// step_kernel looped over the simulations.  Nothing in a search crosses trees and a tree's group lies inside one
// wavefront, so a wavefront keeps its trees from the first simulation to the last and no barrier, launch boundary or
// other wavefront's deepest tree separates two of its simulations: where a launch per simulation ends with the deepest
// of ALL trees every time, a wavefront here only ever waits for its own.  Simulation s (0-based) reads its outputs at
// pol + s * pol_stride, reward / value + s * rv_stride and creates record nb0 + s; the last one selects only when
// final_select is set.  Same records, statistics and selections as the launches it replaces.
template <int G, int P, bool LOGITS>
__global__ __launch_bounds__(kBlockThreads) void run_injected_kernel(PoolDev p, int nb0, int nsim, int final_select,
                                                                     const float* pol, const float* reward,
                                                                     const float* value, size_t pol_stride, size_t rv_stride,
                                                                     int* out_parent, int* out_action) {
    __shared__ double s_log[kLogLds];
    MZ_TREE_LDS(G);
    MZ_POL_LDS(G);
    constexpr int PB = TreeLds<G>::kPB;
    const LdsLog lds_log = stage_log_table(p, s_log);            // once for the run (a workgroup barrier: the only one)
#pragma unroll 1
    for (int s = 0; s < nsim; ++s) {
        asm volatile("" ::: "memory");
        int tx = threadIdx.x;                                    // lane geometry from an opaque copy: nothing per-lane is
        asm volatile("" : "+v"(tx));                             // hoisted out of the loop and kept across it
        const int gid = blockIdx.x * kBlockThreads + tx;
        const int tree = gid / G, gl = gid % G;
        const float* pol_s = pol + (size_t)s * pol_stride;
        // the policy row first, with the backup's first round trip (see stage_requests)
        constexpr int KL = EarlyStage<G>::KL;
        const bool row_ok = tree < p.T && p.A <= TreeLds<G>::kMaxA;
        float pv[KL];
#pragma unroll
        for (int j = 0; j < KL; ++j) {
            const int k = gl + j * G;
            pv[j] = (row_ok && k < p.A) ? pol_s[(size_t)tree * p.A + k] : 0.f;
        }
        const BackupInputs<G, PB> b = backup_inputs<G, PB>(p, tree, gl, pol_s, reward + (size_t)s * rv_stride,
                                                           value + (size_t)s * rv_stride);
        // the root's slots as they are before this backup, for the select that follows it (mz_tree_dev.h, RootPre)
        const bool with_select = s + 1 < nsim || final_select != 0;
        RootPre<P> pre = root_prefetch<G, P>(p, tree, tree < p.T && with_select, gl);
        pre.n_root = b.n_root;
        RootFwd fw;
        const float* pol_row = b.pol_row;
        if (p.A <= TreeLds<G>::kMaxA) {
#pragma unroll
            for (int j = 0; j < KL; ++j) {
                const int k = gl + j * G;
                if (k < p.A) s_pol_row[k] = pv[j];
            }
            pol_row = s_pol_row;
            __builtin_amdgcn_wave_barrier();                     // the row is this wavefront's own: DS order is enough
        }
        backprop_tree<G, LOGITS, PB, true>(p, tree, b.active, gl, nb0 + s, pol_row, b.reward, b.value, b.pf, lds_row, &fw);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (with_select) select_tree<G, P, true>(p, tree, b.active, gl, out_parent, out_action, lds_log, &pre, &fw);
        __builtin_amdgcn_wave_barrier();
    }
}

// mz_pool_set_selection: the selection path of ONE tree given from outside (select_leaf started below the root by the
// drop-in agent, agent.py:177-181), written as select_tree would have written it: path entries with the slots' visit
// counts, path length, leaf parent / action, and the leaf's slot covered by its node's prefix.  One lane; the chain is
// validated link by link (error flag bit 4 otherwise).
__global__ void set_selection_kernel(PoolDev p, int tree, const int* blocks, const int* actions, int len, int* out_parent,
                                     int* out_action) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    bool bad = len < 1 || len > p.N + 1 || blocks[0] != 0;
    int* hi_home = nullptr;
    int hi_word = 0;
    for (int i = 0; i < len && !bad; ++i) {
        const int e = blocks[i];
        if (e < 0 || e > p.N) {
            bad = true;
            break;
        }
        const int n = e == 0 ? p.n_root[tree] : p.A;
        Slot* slots = f_slots(p, rec(p, tree, e));
        int k = -1;
        for (int j = 0; j < n; ++j)
            if (slots[j].act == actions[i]) {
                k = j;
                break;
            }
        if (k < 0) {
            bad = true;
            break;
        }
        const Slot sl = slots[k];
        p.path[(size_t)tree * (p.N + 1) + i] = make_int2((e << kPathShift) | k, sl.count);
        if (i + 1 < len) {                           // an expanded child: the next record must be its own
            if (sl.count <= 0 || (sl.child_hi & kChildMask) != blocks[i + 1]) bad = true;
            hi_home = &slots[k].child_hi;
            hi_word = sl.child_hi;
        } else if (sl.count != 0) {                  // the leaf must be unvisited
            bad = true;
        } else {
            out_parent[tree] = e;
            out_action[tree] = actions[i];
            p.path_len[tree] = len;
            const int hi = (int)((unsigned)hi_word >> kChildBits);
            if (i > 0 && k + 1 > hi) *hi_home = (min(k + 1, kMaxActions - 1) << kChildBits) | (hi_word & kChildMask);
        }
    }
    if (bad) {
        out_parent[tree] = -1;
        out_action[tree] = -1;
        p.path_len[tree] = 0;
        atomicOr(p.err_flag, 16);
    }
}

// Root expansion (agent.py:154-160 on the root; positional priors, trap T5).  noise != NULL: add_exploration_noise
// (agent.py:170-175) in the same pass, the arithmetic of root_noise_kernel on the prior just computed.
// P > 0: followed by the first selection (select_kernel<G, P>'s, agent.py:177-190) of the same trees -- a tree's group
// lies inside one wavefront, which reads back what it has just written (step_kernel) -- one launch less per move.
template <int G, bool LOGITS, int P = 0>
__global__ __launch_bounds__(kBlockThreads) void expand_roots_kernel(PoolDev p, const float* pol, const int* legal,
                                                                     const int* n_legal, const double* noise,
                                                                     double frac, int* out_parent = nullptr,
                                                                     int* out_action = nullptr) {
    const int gid = blockIdx.x * kBlockThreads + threadIdx.x;
    expand_roots_tree<G, LOGITS, P>(p, gid / G, gid % G, pol, legal, n_legal, noise, frac, out_parent, out_action);
}

__global__ void root_noise_kernel(PoolDev p, const double* noise, double frac) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)p.T * p.A) return;
    const int tree = (int)(i / p.A), k = (int)(i % p.A);
    if (k >= p.n_root[tree]) return;
    p.root_prior[i] = p.root_prior[i] * (1 - frac) + noise[i] * frac;     // agent.py:175 (fp64, uncontracted)
}

// One wavefront per tree: the rows of visits / policy are cleared and the root's children scattered into them by all
// 64 lanes (one lane per tree took 236 us at A = 362).
__device__ __forceinline__ void root_stats_tree(const PoolDev& p, int tree, int lane, int* visits, double* root_value,
                                                double* minmax, float* policy) {
    const int A = p.A, n = p.n_root[tree];
    const int rc = p.root_count[tree];
    for (int a = lane; a < A; a += 64) {
        if (visits) visits[(size_t)tree * A + a] = 0;
        if (policy) policy[(size_t)tree * A + a] = 0.f;
    }
    // the zeros have reached memory before any lane overwrites one of them with a child's count
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const Slot* slots = f_slots(p, rec(p, tree, 0));
    for (int k = lane; k < n; k += 64) {
        const int a = slots[k].act;
        const int c = slots[k].count;
        if (visits) visits[(size_t)tree * A + a] = c;
        if (policy) policy[(size_t)tree * A + a] = (float)((double)c / (double)rc);    // agent.py:230
    }
    if (lane == 0) {
        if (root_value)
            root_value[tree] = rc > 0 ? p.root_vsum[tree] / (double)rc : std::numeric_limits<double>::quiet_NaN();
        if (minmax) {
            minmax[2 * tree] = p.mm_min[tree];
            minmax[2 * tree + 1] = p.mm_max[tree];
        }
    }
}
__global__ __launch_bounds__(64) void root_stats_kernel(PoolDev p, int* visits, double* root_value, double* minmax,
                                                        float* policy) {
    if ((int)blockIdx.x < p.T) root_stats_tree(p, blockIdx.x, threadIdx.x, visits, root_value, minmax, policy);
}

// softmax_sample (agent.py:162-168) with CPython's random.choices arithmetic.  One wavefront per tree: the lanes read the
// root's children and raise their counts to 1/t side by side (element-wise, so the values are those of a sequential
// pass); the cumulative sums, whose order is part of the result, are lane 0's.
// MAXA: the LDS staging's capacity (>= the pool's action count); the small one keeps the workgroups of a narrow game
// from being limited to three per CU by 48 KB of LDS each
template <int MAXA>
__device__ __forceinline__ void select_action_tree(const PoolDev& p, int tree, int lane, const double* u,
                                                   const double* temperature, int* out_action) {
    __shared__ double s_w[MAXA];
    __shared__ int s_act[MAXA];
    const int n = p.n_root[tree];
    if (n <= 0) {
        if (lane == 0) out_action[tree] = -1;
        return;
    }
    const Slot* slots = f_slots(p, rec(p, tree, 0));
    const double t = temperature[tree];
    if (t == 0.0) {                               // max((count, action)): ties to the largest action
        long long best = -1;
        for (int k = lane; k < n; k += 64) {
            const long long key = ((long long)slots[k].count << 32) | (unsigned)slots[k].act;
            best = key > best ? key : best;
        }
        for (int o = 32; o >= 1; o >>= 1) {
            const long long other = __shfl_xor(best, o);
            best = other > best ? other : best;
        }
        if (lane == 0) out_action[tree] = (int)(best & 0xffffffffll);
        return;
    }
    const double inv_t = 1 / t;
    // count ** (1/t) is the host libm's pow in the reference (agent.py:167), which no device pow reproduces bit for bit:
    // a temperature registered with mz_pool_set_temperature_table is looked up in the host's own table
    const double* tab = nullptr;
#pragma unroll
    for (int i = 0; i < kPowTables; ++i)
        if (i < p.n_pow && p.pow_t[i] == t) tab = p.pow_table + (size_t)i * (p.N + 1);
    auto weight = [&](int c) -> double {
        if (tab != nullptr) return tab[min(max(c, 0), p.N)];
        if (inv_t == 1.0) return (double)c;       // pow(c, 1.0) == c exactly
        if (inv_t == 2.0) return (double)c * (double)c;
        if (inv_t == 4.0) {
            const double c2 = (double)c * (double)c;
            return c2 * c2;
        }
        return pow((double)c, inv_t);             // unregistered temperature: the device's pow (not bit-exact)
    };
    for (int k = lane; k < n; k += 64) {
        s_w[k] = weight(slots[k].count);
        s_act[k] = slots[k].act;
    }
    __syncthreads();
    if (lane != 0) return;
    double total = 0.0;                           // itertools.accumulate: sequential fp64 sums
    for (int k = 0; k < n; ++k) total = (k == 0) ? s_w[0] : total + s_w[k];
    const double x = u[tree] * (total + 0.0);
    // bisect_right(cum, x, 0, n-1) == number of cum[i] <= x for i < n-1 (cum is non-decreasing)
    double cum = 0.0;
    int idx = 0;
    for (int k = 0; k < n - 1; ++k) {
        cum = (k == 0) ? s_w[0] : cum + s_w[k];
        if (!(x < cum)) idx = k + 1;
        else break;
    }
    out_action[tree] = s_act[idx];
}
constexpr int kSmallA = 64;
template <int MAXA>
__global__ __launch_bounds__(64) void select_action_kernel(PoolDev p, const double* u, const double* temperature,
                                                           int* out_action) {
    if ((int)blockIdx.x < p.T) select_action_tree<MAXA>(p, blockIdx.x, threadIdx.x, u, temperature, out_action);
}
// root_stats_tree + select_action_tree of one tree whose action axis fits a wavefront with a lane to spare (A < 64), as one
// chain of two memory round trips: everything that does not depend on a count is requested first (child k of the root in
// lane k); the counts reach their action's lane through a forward permute (lanes without a child send theirs to the spare
// lane 63) instead of a row of zeros that must land before the scatter; the weights' sequential fp64 sums run over
// v_readlane in lane order -- the arithmetic of select_action_tree, without its LDS staging and barrier.
__device__ __forceinline__ void root_finish_small(const PoolDev& p, int tree, int lane, int* visits, double* root_value,
                                                  double* minmax, float* policy, const double* u,
                                                  const double* temperature, int* out_action) {
    const int A = p.A;
    const Slot* slots = f_slots(p, rec(p, tree, 0));
    const int n = p.n_root[tree];
    const int rc = p.root_count[tree];
    int act = 0, cnt = 0;
    if (lane < A) {                                   // (the record holds A slots; those beyond n are not used below)
        act = slots[lane].act;
        cnt = slots[lane].count;
    }
    const double t = temperature[tree], uu = u[tree];
    double vsum = 0.0, mn = 0.0, mx = 0.0;
    if (lane == 0) {
        vsum = p.root_vsum[tree];
        mn = p.mm_min[tree];
        mx = p.mm_max[tree];
    }
    const bool live = lane < n;
    // mz_root_stats: visits and policy over the full action axis (agent.py:230)
    const int got = __builtin_amdgcn_ds_permute((live ? act : 63) << 2, live ? (cnt << 1) | 1 : 0);
    if (lane < A) {
        const int c = got >> 1;
        if (visits) visits[(size_t)tree * A + lane] = c;
        if (policy) policy[(size_t)tree * A + lane] = (got & 1) ? (float)((double)c / (double)rc) : 0.f;
    }
    if (lane == 0) {
        if (root_value) root_value[tree] = rc > 0 ? vsum / (double)rc : std::numeric_limits<double>::quiet_NaN();
        if (minmax) {
            minmax[2 * tree] = mn;
            minmax[2 * tree + 1] = mx;
        }
    }
    // mz_select_action (agent.py:162-168)
    if (n <= 0) {
        if (lane == 0) out_action[tree] = -1;
        return;
    }
    if (t == 0.0) {                               // max((count, action)): ties to the largest action
        long long best = live ? ((long long)cnt << 32) | (unsigned)act : -1;
        for (int o = 32; o >= 1; o >>= 1) {
            const long long other = __shfl_xor(best, o);
            best = other > best ? other : best;
        }
        if (lane == 0) out_action[tree] = (int)(best & 0xffffffffll);
        return;
    }
    const double inv_t = 1 / t;
    const double* tab = nullptr;                  // the host's own pow table of a registered temperature
#pragma unroll
    for (int i = 0; i < kPowTables; ++i)
        if (i < p.n_pow && p.pow_t[i] == t) tab = p.pow_table + (size_t)i * (p.N + 1);
    double wgt = 0.0;
    if (live) {
        if (tab != nullptr) wgt = tab[min(max(cnt, 0), p.N)];
        else if (inv_t == 1.0) wgt = (double)cnt;       // pow(c, 1.0) == c exactly
        else if (inv_t == 2.0) wgt = (double)cnt * (double)cnt;
        else if (inv_t == 4.0) {
            const double c2 = (double)cnt * (double)cnt;
            wgt = c2 * c2;
        } else wgt = pow((double)cnt, inv_t);           // unregistered temperature: the device's pow (not bit-exact)
    }
    const int wlo = (int)(__double_as_longlong(wgt) & 0xffffffffll), whi = (int)(__double_as_longlong(wgt) >> 32);
    auto weight_of = [&](int k) -> double {       // (k is uniform)
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane(wlo, k), hi = (unsigned)__builtin_amdgcn_readlane(whi, k);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    double total = 0.0;                           // itertools.accumulate: sequential fp64 sums
    for (int k = 0; k < n; ++k) total = (k == 0) ? weight_of(0) : total + weight_of(k);
    const double x = uu * (total + 0.0);
    // bisect_right(cum, x, 0, n-1) == number of cum[i] <= x for i < n-1 (cum is non-decreasing)
    double cum = 0.0;
    int idx = 0;
    for (int k = 0; k < n - 1; ++k) {
        cum = (k == 0) ? weight_of(0) : cum + weight_of(k);
        if (!(x < cum)) idx = k + 1;
        else break;
    }
    const int chosen = __builtin_amdgcn_readlane(act, __builtin_amdgcn_readfirstlane(idx));
    if (lane == 0) out_action[tree] = chosen;
}

// the end of a move in one launch: mz_root_stats, then mz_select_action
template <int MAXA>
__global__ __launch_bounds__(64) void root_finish_kernel(PoolDev p, int* visits, double* root_value, double* minmax,
                                                         float* policy, const double* u, const double* temperature,
                                                         int* out_action) {
    if ((int)blockIdx.x >= p.T) return;
    if (MAXA == kSmallA && p.A < 64) {
        root_finish_small(p, blockIdx.x, threadIdx.x, visits, root_value, minmax, policy, u, temperature, out_action);
        return;
    }
    root_stats_tree(p, blockIdx.x, threadIdx.x, visits, root_value, minmax, policy);
    select_action_tree<MAXA>(p, blockIdx.x, threadIdx.x, u, temperature, out_action);
}

__global__ void sum_counters_kernel(PoolDev p, long long* out2) {
    __shared__ long long s_len[kBlockThreads], s_sims[kBlockThreads];
    long long a = 0, b = 0;
    for (int t = threadIdx.x; t < p.T; t += kBlockThreads) {
        a += p.sims[t];
        b += p.sum_len[t];
    }
    s_sims[threadIdx.x] = a;
    s_len[threadIdx.x] = b;
    __syncthreads();
    for (int off = kBlockThreads / 2; off >= 1; off >>= 1) {
        if (threadIdx.x < off) {
            s_sims[threadIdx.x] += s_sims[threadIdx.x + off];
            s_len[threadIdx.x] += s_len[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out2[0] = s_sims[0];
        out2[1] = s_len[0];
    }
}

// Lanes per tree when the caller leaves the choice to the library: the power of two covering the action axis, halved
// when that would idle 40 % of the lanes and the launch has wavefronts to spare (measured at 16384 x 18 actions: 38 us
// a step with 16 lanes, 48 us with 32; at 4096 x 9 the 16-lane group stays ahead of 8).  Results do not depend on it:
// every reduction is over a total order, every sum in a fixed order.
int group_width(int A, int T) {
    int g = 2;
    while (g < A && g < 64) g <<= 1;
    if (g > 2 && 10 * A <= 6 * g && (long long)T * g / 64 > 4096) g >>= 1;
    return g;
}

// NumPy pairwise_sum recursion for n elements -> leaves (lo,n) in order + merges (dst,src) in evaluation order.
int build_sum_program(int lo, int n, std::vector<int>& leaves, std::vector<int>& merges) {
    if (n <= 128) {
        leaves.push_back(lo);
        leaves.push_back(n);
        return (int)leaves.size() / 2 - 1;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    const int left = build_sum_program(lo, n2, leaves, merges);
    const int right = build_sum_program(lo + n2, n - n2, leaves, merges);
    merges.push_back(left);
    merges.push_back(right);
    return left;
}

}  // namespace

#include "mz_pool_internal.h"

namespace {

template <typename T>
int dev_alloc(mz_pool* pool, T** out, size_t count) {
    void* ptr = nullptr;
    const size_t bytes = count * sizeof(T);
    hipError_t e = hipMalloc(&ptr, bytes ? bytes : 16);
    if (e != hipSuccess)
        return mz::fail(e == hipErrorOutOfMemory ? MZ_ERR_OOM : MZ_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", bytes,
                        hipGetErrorString(e));
    pool->allocations.push_back(ptr);
    pool->bytes += bytes;
    *out = static_cast<T*>(ptr);
    return MZ_OK;
}

#define MZ_TRY(expr)              \
    do {                          \
        int _s = (expr);          \
        if (_s != MZ_OK) return _s; \
    } while (0)

int grid_for(const mz_pool* pool) {
    const int trees_per_block = kBlockThreads / pool->G;
    return (pool->dev.T + trees_per_block - 1) / trees_per_block;
}

template <typename F>
void dispatch_group(int G, F&& f) {
    switch (G) {
        case 2: f(std::integral_constant<int, 2>{}); break;
        case 4: f(std::integral_constant<int, 4>{}); break;
        case 8: f(std::integral_constant<int, 8>{}); break;
        case 16: f(std::integral_constant<int, 16>{}); break;
        case 32: f(std::integral_constant<int, 32>{}); break;
        default: f(std::integral_constant<int, 64>{}); break;
    }
}

// (G, P) for the kernels that descend: P children per lane are loaded together (the power of two covering
// ceil(A / G), at most 8; wider nodes take several rounds)
template <typename F>
void dispatch_group_passes(int G, int P, F&& f) {
    dispatch_group(G, [&](auto g) {
        switch (P) {
            case 1: f(g, std::integral_constant<int, 1>{}); break;
            case 2: f(g, std::integral_constant<int, 2>{}); break;
            case 4: f(g, std::integral_constant<int, 4>{}); break;
            default: f(g, std::integral_constant<int, 8>{}); break;
        }
    });
}

int passes_for(int A, int G) {
    const int need = (A + G - 1) / G;
    int P = 1;
    while (P < need && P < 8) P <<= 1;
    return P;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mz::fail(MZ_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return MZ_OK;
}

int expand_roots_impl(mz_pool* pool, const float* pol, const int32_t* legal, const int32_t* n_legal, bool logits,
                      const double* noise = nullptr, double frac = 0.0, int32_t* out_parent = nullptr,
                      int32_t* out_action = nullptr) {
    MZ_REQUIRE(pool != nullptr, "pool is NULL");
    MZ_REQUIRE(pol != nullptr, "priors/logits is NULL");
    MZ_HIP(hipSetDevice(pool->device));
    const int grid = grid_for(pool);
    if (out_parent != nullptr) {             // with the first selection
        MZ_REQUIRE(out_action != nullptr, "out_action is NULL");
        if (pool->dev.N < 1) return mz::fail(MZ_ERR_STATE, "pool is full: it holds no simulation");
        dispatch_group_passes(pool->G, pool->P, [&](auto g, auto pp) {
            constexpr int G = decltype(g)::value;
            constexpr int P = decltype(pp)::value;
            if (logits)
                expand_roots_kernel<G, true, P><<<grid, kBlockThreads, 0, pool->stream>>>(pool->dev, pol, legal, n_legal, noise,
                                                                                          frac, out_parent, out_action);
            else
                expand_roots_kernel<G, false, P><<<grid, kBlockThreads, 0, pool->stream>>>(pool->dev, pol, legal, n_legal, noise,
                                                                                           frac, out_parent, out_action);
        });
        MZ_TRY(check_launch("expand_roots_kernel + select"));
        pool->blocks_in_use = 1;
        pool->select_launches++;
        pool->pending = true;
        return MZ_OK;
    }
    dispatch_group(pool->G, [&](auto g) {
        constexpr int G = decltype(g)::value;
        if (logits)
            expand_roots_kernel<G, true><<<grid, kBlockThreads, 0, pool->stream>>>(pool->dev, pol, legal, n_legal, noise,
                                                                                   frac);
        else
            expand_roots_kernel<G, false><<<grid, kBlockThreads, 0, pool->stream>>>(pool->dev, pol, legal, n_legal, noise,
                                                                                    frac);
    });
    MZ_TRY(check_launch("expand_roots_kernel"));
    pool->blocks_in_use = 1;
    pool->pending = false;
    return MZ_OK;
}

int backprop_impl(mz_pool* pool, const float* pol, const float* reward, const float* value, bool logits,
                  int32_t* out_parent, int32_t* out_action, bool fused) {
    MZ_REQUIRE(pool != nullptr, "pool is NULL");
    MZ_REQUIRE(pol && reward && value, "priors/logits, reward and value must be non-NULL");
    if (fused) MZ_REQUIRE(out_parent && out_action, "out_parent_block and out_action must be non-NULL");
    int nb = 0;
    MZ_TRY(mz_pool_begin_backprop(pool, fused, &nb));
    MZ_HIP(hipSetDevice(pool->device));
    const int grid = grid_for(pool);
    dispatch_group_passes(pool->G, pool->P, [&](auto g, auto pp) {
        constexpr int G = decltype(g)::value;
        constexpr int P = decltype(pp)::value;
        hipStream_t st = pool->stream;
        if (fused && logits)
            step_kernel<G, P, true><<<grid, kBlockThreads, 0, st>>>(pool->dev, nb, pol, reward, value, out_parent, out_action);
        else if (fused)
            step_kernel<G, P, false><<<grid, kBlockThreads, 0, st>>>(pool->dev, nb, pol, reward, value, out_parent, out_action);
        else if (logits)
            backprop_kernel<G, true><<<grid, kBlockThreads, 0, st>>>(pool->dev, nb, pol, reward, value);
        else
            backprop_kernel<G, false><<<grid, kBlockThreads, 0, st>>>(pool->dev, nb, pol, reward, value);
    });
    MZ_TRY(check_launch(fused ? "step_kernel" : "backprop_kernel"));
    mz_pool_commit_backprop(pool, fused);
    return MZ_OK;
}

}  // namespace

// Internal (mz_pool_internal.h): `num_simulations` x { expand + backup with the given outputs, next select } in one
// launch; the pool must hold a pending selection.
int mz_pool_run_injected(mz_pool* pool, int num_simulations, bool final_select, const float* pol, bool logits,
                         const float* reward, const float* value, size_t pol_stride, size_t rv_stride, int32_t* out_parent,
                         int32_t* out_action) {
    int nb0 = 0;
    MZ_TRY(mz_pool_begin_run(pool, num_simulations, final_select, &nb0));
    MZ_HIP(hipSetDevice(pool->device));
    // Lanes per tree of THIS launch (results do not depend on them).  Where the pool's width fills all four wavefront
    // slots of every SIMD, half as many wavefronts with twice the trees each run their chains faster in the loop
    // (16384 x 18 actions: 23.7 us a simulation with 8 lanes, 25.5 with 16; a launch per simulation ties at 33.7 / 34.0).
    int G = pool->G, P = pool->P;
    if (G >= 16 && G < 64 && pool->dev.A <= 2 * G && (long long)pool->dev.T * G / 64 >= 4096) {
        G >>= 1;
        P = passes_for(pool->dev.A, G);
    }
    const int grid = (int)(((long long)pool->dev.T * G + kBlockThreads - 1) / kBlockThreads);
    dispatch_group_passes(G, P, [&](auto g, auto pp) {
        constexpr int G = decltype(g)::value;
        constexpr int P = decltype(pp)::value;
        if (logits)
            run_injected_kernel<G, P, true><<<grid, kBlockThreads, 0, pool->stream>>>(
                pool->dev, nb0, num_simulations, final_select ? 1 : 0, pol, reward, value, pol_stride, rv_stride, out_parent,
                out_action);
        else
            run_injected_kernel<G, P, false><<<grid, kBlockThreads, 0, pool->stream>>>(
                pool->dev, nb0, num_simulations, final_select ? 1 : 0, pol, reward, value, pol_stride, rv_stride, out_parent,
                out_action);
    });
    MZ_TRY(check_launch("run_injected_kernel"));
    mz_pool_commit_run(pool, num_simulations, final_select);
    return MZ_OK;
}

const void* mz_pool_device_image(mz_pool* pool) {
    if (hipSetDevice(pool->device) != hipSuccess) return nullptr;
    if (!pool->dev_image) {
        void* ptr = nullptr;
        if (hipMalloc(&ptr, sizeof(PoolDev)) != hipSuccess) {
            mz::fail(MZ_ERR_OOM, "device image of the pool descriptor");
            return nullptr;
        }
        pool->allocations.push_back(ptr);
        pool->dev_image = static_cast<PoolDev*>(ptr);
        pool->image_valid = false;
    }
    if (!pool->image_valid || memcmp(&pool->image_host, &pool->dev, sizeof(PoolDev)) != 0) {
        // the previous image may still be read by launches on the stream: the copy is ordered behind them, and its
        // source is copied to a staging buffer before the call returns (pageable memory)
        pool->image_host = pool->dev;
        if (hipMemcpyAsync(pool->dev_image, &pool->image_host, sizeof(PoolDev), hipMemcpyHostToDevice, pool->stream) !=
            hipSuccess) {
            mz::fail(MZ_ERR_HIP, "upload of the pool descriptor failed");
            return nullptr;
        }
        pool->image_valid = true;
    }
    return pool->dev_image;
}

int mz_pool_begin_backprop(mz_pool* pool, bool with_select, int* out_new_block) {
    if (!pool->pending) return mz::fail(MZ_ERR_STATE, "expand_backprop without a preceding select");
    const int nb = pool->blocks_in_use;
    if (nb > pool->dev.N) return mz::fail(MZ_ERR_STATE, "pool is full: %d simulations done", nb - 1);
    if (with_select && nb + 1 > pool->dev.N)
        return mz::fail(MZ_ERR_STATE, "mz_step on the last simulation (%d of %d): use mz_expand_backprop", nb,
                        pool->dev.N);
    *out_new_block = nb;
    return MZ_OK;
}

int mz_pool_begin_run(mz_pool* pool, int num_simulations, bool final_select, int* out_first_block) {
    if (!pool->pending) return mz::fail(MZ_ERR_STATE, "expand_backprop without a preceding select");
    const int nb = pool->blocks_in_use;
    if (num_simulations < 1) return mz::fail(MZ_ERR_INVALID_ARG, "num_simulations must be >= 1 (got %d)", num_simulations);
    if (nb + num_simulations - 1 + (final_select ? 1 : 0) > pool->dev.N)
        return mz::fail(MZ_ERR_STATE, "%d more simulations%s on a pool with %d of %d done", num_simulations,
                        final_select ? " and a select" : "", nb - 1, pool->dev.N);
    *out_first_block = nb;
    return MZ_OK;
}

void mz_pool_commit_run(mz_pool* pool, int num_simulations, bool final_select) {
    pool->select_launches += num_simulations - (final_select ? 0 : 1);
    pool->backprop_launches += num_simulations;
    pool->blocks_in_use += num_simulations;
    pool->pending = final_select;
}

void mz_pool_commit_backprop(mz_pool* pool, bool with_select) {
    if (with_select) pool->select_launches++;
    pool->backprop_launches++;
    pool->blocks_in_use += 1;
    pool->pending = with_select;
}

extern "C" {

int mz_pool_create(const mz_pool_config_t* cfg, void* stream, mz_pool_t** out_pool) {
    MZ_REQUIRE(cfg != nullptr && out_pool != nullptr, "config/out_pool is NULL");
    *out_pool = nullptr;
    MZ_REQUIRE(cfg->num_trees >= 1, "num_trees must be >= 1 (got %d)", cfg->num_trees);
    MZ_REQUIRE(cfg->num_actions >= 1 && cfg->num_actions <= kMaxActions, "num_actions must be in [1, %d] (got %d)",
               kMaxActions, cfg->num_actions);
    MZ_REQUIRE(cfg->max_simulations >= 1 && cfg->max_simulations <= kMaxSimulations,
               "max_simulations must be in [1, %d] (got %d)", kMaxSimulations, cfg->max_simulations);
    MZ_REQUIRE(cfg->num_players == 1 || cfg->num_players == 2, "num_players must be 1 or 2 (got %d)",
               cfg->num_players);
    MZ_REQUIRE(cfg->lanes_per_tree == 0 || (cfg->lanes_per_tree >= 2 && cfg->lanes_per_tree <= 64 &&
                                            (cfg->lanes_per_tree & (cfg->lanes_per_tree - 1)) == 0),
               "lanes_per_tree must be 0 or a power of two in [2, 64] (got %d)", cfg->lanes_per_tree);
    if (cfg->has_known_bounds)
        MZ_REQUIRE(cfg->known_max == cfg->known_max && cfg->known_min == cfg->known_min, "known bounds are NaN");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return mz::fail(MZ_ERR_NO_DEVICE, "no HIP device visible: libmzhip has no CPU fallback");
    }
    mz_pool* pool = new (std::nothrow) mz_pool();
    if (!pool) return mz::fail(MZ_ERR_OOM, "host allocation failed");
    int status = MZ_OK;
    do {
        hipError_t e = hipGetDevice(&pool->device);
        if (e != hipSuccess) {
            status = mz::fail(MZ_ERR_HIP, "hipGetDevice failed: %s", hipGetErrorString(e));
            break;
        }
        pool->stream = static_cast<hipStream_t>(stream);
        PoolDev& d = pool->dev;
        d.T = cfg->num_trees;
        d.A = cfg->num_actions;
        d.N = cfg->max_simulations;
        d.has_bounds = cfg->has_known_bounds ? 1 : 0;
        d.off_back = ((int)sizeof(Slot) * d.A + 15) / 16 * 16;
        d.RS = (d.off_back + (int)sizeof(BackSlot) * d.A + 127) / 128 * 128;
        d.hi_unsorted = d.A - 1 < kMaxActions - 1 ? (d.A > 1 ? d.A - 1 : 1) : kMaxActions - 1;
        d.tree_stride = (size_t)(d.N + 1) * (size_t)d.RS;
        d.discount = cfg->num_players == 2 ? -cfg->discount : cfg->discount;      // agent.py:150-152
        d.default_value = cfg->default_value;
        d.kmin = cfg->known_min;
        d.kmax = cfg->known_max;
        pool->G = cfg->lanes_per_tree ? cfg->lanes_per_tree : group_width(d.A, d.T);
        pool->P = passes_for(d.A, pool->G);
        const size_t TA = (size_t)d.T * d.A;
        double* log_dev = nullptr;
        int* leaf_dev = nullptr;
        int* merge_dev = nullptr;
        if ((status = dev_alloc(pool, &d.blocks, (size_t)d.T * d.tree_stride)) != MZ_OK) break;
        if ((status = dev_alloc(pool, &d.root_prior, TA)) != MZ_OK) break;
        if ((status = dev_alloc(pool, &d.n_root, (size_t)d.T)) != MZ_OK) break;
        if ((status = dev_alloc(pool, &d.root_count, (size_t)d.T)) != MZ_OK) break;
        if ((status = dev_alloc(pool, &d.root_vsum, (size_t)d.T)) != MZ_OK) break;
        if ((status = dev_alloc(pool, &d.mm_min, (size_t)d.T)) != MZ_OK) break;
        if ((status = dev_alloc(pool, &d.mm_max, (size_t)d.T)) != MZ_OK) break;
        if ((status = dev_alloc(pool, &d.path, (size_t)d.T * (d.N + 1))) != MZ_OK) break;
        if ((status = dev_alloc(pool, &d.path_len, (size_t)d.T)) != MZ_OK) break;
        if ((status = dev_alloc(pool, &d.sum_len, (size_t)d.T)) != MZ_OK) break;
        if ((status = dev_alloc(pool, &d.sims, (size_t)d.T)) != MZ_OK) break;
        if ((status = dev_alloc(pool, &d.err_flag, (size_t)1)) != MZ_OK) break;
        if ((status = dev_alloc(pool, &log_dev, (size_t)d.N + 2)) != MZ_OK) break;
        if ((status = dev_alloc(pool, &pool->counter_scratch, (size_t)2)) != MZ_OK) break;
        if ((status = dev_alloc(pool, &pool->pow_dev, (size_t)kPowTables * (d.N + 1))) != MZ_OK) break;
        d.pow_table = pool->pow_dev;
        d.n_pow = 0;
        std::vector<double> table(d.N + 2);
        for (int n = 0; n < d.N + 2; ++n)
            table[n] = cfg->log_table ? cfg->log_table[n] : 2 * std::log((double)(n + 1));   // config.py:160
        // Children kept by descending prior, so that select reads a prefix of a node (mz_tree_dev.h, select_tree): sound
        // when the exploration factor is finite and >= 0 (it is 2 log(n+1) unless the caller's table says otherwise) and
        // worth it when ranking a new node's children is cheap.
        bool table_ok = true;
        for (double x : table) table_ok = table_ok && x >= 0.0 && std::isfinite(x);
        const int per_lane = (d.A + pool->G - 1) / pool->G;
        d.sort_nodes = (table_ok && d.A >= 4 && d.A * per_lane <= 512) ? 1 : 0;
        std::vector<int> leaves, merges;
        build_sum_program(0, d.A, leaves, merges);
        d.n_sum_leaf = (int)leaves.size() / 2;
        d.n_sum_merge = (int)merges.size() / 2;
        if ((status = dev_alloc(pool, &leaf_dev, leaves.size())) != MZ_OK) break;
        if ((status = dev_alloc(pool, &merge_dev, merges.size() + 2)) != MZ_OK) break;
        hipError_t err = hipSuccess;
#define MZ_CREATE_HIP(expr)                                                                      \
    if ((err = (expr)) != hipSuccess) {                                                          \
        status = mz::fail(MZ_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(err));           \
        break;                                                                                   \
    }
        MZ_CREATE_HIP(hipMemcpy(log_dev, table.data(), table.size() * sizeof(double), hipMemcpyHostToDevice));
        MZ_CREATE_HIP(hipMemcpy(leaf_dev, leaves.data(), leaves.size() * sizeof(int), hipMemcpyHostToDevice));
        if (!merges.empty())
            MZ_CREATE_HIP(hipMemcpy(merge_dev, merges.data(), merges.size() * sizeof(int), hipMemcpyHostToDevice));
        MZ_CREATE_HIP(hipMemset(d.n_root, 0, (size_t)d.T * sizeof(int)));
        MZ_CREATE_HIP(hipMemset(d.sum_len, 0, (size_t)d.T * sizeof(long long)));
        MZ_CREATE_HIP(hipMemset(d.sims, 0, (size_t)d.T * sizeof(long long)));
        MZ_CREATE_HIP(hipMemset(d.err_flag, 0, sizeof(int)));
        MZ_CREATE_HIP(hipMemset(d.path_len, 0, (size_t)d.T * sizeof(int)));
        // zeroed: the backup prefetches path slots before it knows the path length (mz_tree_dev.h, BackupPrefetch)
        MZ_CREATE_HIP(hipMemset(d.path, 0, (size_t)d.T * (d.N + 1) * sizeof(int2)));
#undef MZ_CREATE_HIP
        d.log_table = log_dev;
        d.sum_leaf = leaf_dev;
        d.sum_merge = merge_dev;
    } while (0);
    if (status != MZ_OK) {
        for (void* ptr : pool->allocations) (void)hipFree(ptr);
        delete pool;
        return status;
    }
    *out_pool = pool;
    return MZ_OK;
}

int mz_pool_destroy(mz_pool_t* pool) {
    if (!pool) return MZ_OK;
    (void)hipSetDevice(pool->device);
    (void)hipStreamSynchronize(pool->stream);
    for (void* ptr : pool->allocations) (void)hipFree(ptr);
    delete pool;
    return MZ_OK;
}

int mz_pool_set_stream(mz_pool_t* pool, void* stream) {
    MZ_REQUIRE(pool != nullptr, "pool is NULL");
    pool->stream = static_cast<hipStream_t>(stream);
    return MZ_OK;
}

int mz_pool_set_temperature_table(mz_pool_t* pool, double temperature, const double* weights_host) {
    MZ_REQUIRE(pool != nullptr && weights_host != nullptr, "pool/weights_host is NULL");
    MZ_REQUIRE(temperature == temperature && temperature != 0.0, "temperature must be a non-zero number (0 is the arg-max)");
    PoolDev& d = pool->dev;
    int slot = -1;
    for (int i = 0; i < d.n_pow; ++i)
        if (d.pow_t[i] == temperature) slot = i;
    if (slot < 0) {
        slot = d.n_pow < kPowTables ? d.n_pow : pool->pow_next;      // full: replace the oldest registration
        if (d.n_pow == kPowTables) pool->pow_next = (pool->pow_next + 1) % kPowTables;
    }
    MZ_HIP(hipSetDevice(pool->device));
    MZ_HIP(hipStreamSynchronize(pool->stream));      // launches in flight may still read the slot being replaced
    MZ_HIP(hipMemcpy(pool->pow_dev + (size_t)slot * (d.N + 1), weights_host, (size_t)(d.N + 1) * sizeof(double),
                     hipMemcpyHostToDevice));
    d.pow_t[slot] = temperature;
    if (slot == d.n_pow) d.n_pow++;
    return MZ_OK;
}

int mz_pool_reset(mz_pool_t* pool) {
    MZ_REQUIRE(pool != nullptr, "pool is NULL");
    pool->blocks_in_use = 0;     // per-tree state is re-initialised by the next mz_expand_roots
    pool->pending = false;
    return MZ_OK;
}

int mz_expand_roots(mz_pool_t* pool, const float* priors, const int32_t* legal, const int32_t* n_legal) {
    return expand_roots_impl(pool, priors, legal, n_legal, false);
}

int mz_expand_roots_logits(mz_pool_t* pool, const float* logits, const int32_t* legal, const int32_t* n_legal) {
    return expand_roots_impl(pool, logits, legal, n_legal, true);
}

// Internal to the library (mz_pool_internal.h): mz_expand_roots[_logits] followed by mz_add_root_noise -- and, when
// out_parent_block is given, by the first mz_select -- as one launch.
__attribute__((visibility("hidden"))) int mz_expand_roots_noise(mz_pool_t* pool, const float* pol, int32_t is_logits,
                                                                const int32_t* legal, const int32_t* n_legal,
                                                                const double* noise, double frac,
                                                                int32_t* out_parent_block, int32_t* out_action) {
    return expand_roots_impl(pool, pol, legal, n_legal, is_logits != 0, noise, frac, out_parent_block, out_action);
}

int mz_add_root_noise(mz_pool_t* pool, const double* noise, double frac) {
    MZ_REQUIRE(pool != nullptr && noise != nullptr, "pool/noise is NULL");
    if (pool->blocks_in_use != 1) return mz::fail(MZ_ERR_STATE, "add_root_noise must directly follow expand_roots");
    MZ_HIP(hipSetDevice(pool->device));
    const size_t n = (size_t)pool->dev.T * pool->dev.A;
    hipLaunchKernelGGL(root_noise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, pool->stream, pool->dev,
                       noise, frac);
    return check_launch("root_noise_kernel");
}

int mz_select(mz_pool_t* pool, int32_t* out_parent_block, int32_t* out_action) {
    MZ_REQUIRE(pool != nullptr, "pool is NULL");
    MZ_REQUIRE(out_parent_block && out_action, "out_parent_block and out_action must be non-NULL");
    if (pool->blocks_in_use < 1) return mz::fail(MZ_ERR_STATE, "select before expand_roots");
    if (pool->pending) return mz::fail(MZ_ERR_STATE, "select while a previous select awaits expand_backprop");
    if (pool->blocks_in_use > pool->dev.N)
        return mz::fail(MZ_ERR_STATE, "pool is full: %d simulations done", pool->blocks_in_use - 1);
    MZ_HIP(hipSetDevice(pool->device));
    const int grid = grid_for(pool);
    dispatch_group_passes(pool->G, pool->P, [&](auto g, auto pp) {
        constexpr int G = decltype(g)::value;
        constexpr int P = decltype(pp)::value;
        select_kernel<G, P><<<grid, kBlockThreads, 0, pool->stream>>>(pool->dev, out_parent_block, out_action);
    });
    MZ_TRY(check_launch("select_kernel"));
    pool->select_launches++;
    pool->pending = true;
    return MZ_OK;
}

int mz_pool_set_selection(mz_pool_t* pool, int32_t tree, const int32_t* blocks_host, const int32_t* actions_host,
                          int32_t length, int32_t* out_parent_block, int32_t* out_action) {
    MZ_REQUIRE(pool != nullptr && blocks_host && actions_host && out_parent_block && out_action,
               "pool/blocks/actions/out_parent_block/out_action is NULL");
    MZ_REQUIRE(tree >= 0 && tree < pool->dev.T, "tree %d out of range [0,%d)", tree, pool->dev.T);
    MZ_REQUIRE(length >= 1 && length <= pool->dev.N + 1, "path length must be in [1, %d] (got %d)", pool->dev.N + 1, length);
    if (pool->blocks_in_use < 1) return mz::fail(MZ_ERR_STATE, "set_selection before expand_roots");
    if (pool->blocks_in_use > pool->dev.N)
        return mz::fail(MZ_ERR_STATE, "pool is full: %d simulations done", pool->blocks_in_use - 1);
    // every tree needs a selection before the next backup: either one is pending for all of them already (this call
    // replaces the one of `tree`), or the pool holds this one tree only
    if (!pool->pending && pool->dev.T != 1)
        return mz::fail(MZ_ERR_STATE, "set_selection on a pool of %d trees needs a pending mz_select", pool->dev.T);
    MZ_HIP(hipSetDevice(pool->device));
    int* dev = nullptr;
    MZ_HIP(hipMalloc(reinterpret_cast<void**>(&dev), 2 * (size_t)length * sizeof(int)));
    hipError_t e = hipMemcpyAsync(dev, blocks_host, (size_t)length * sizeof(int), hipMemcpyHostToDevice, pool->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(dev + length, actions_host, (size_t)length * sizeof(int), hipMemcpyHostToDevice, pool->stream);
    if (e == hipSuccess) {
        set_selection_kernel<<<1, 64, 0, pool->stream>>>(pool->dev, tree, dev, dev + length, length, out_parent_block,
                                                        out_action);
        e = hipGetLastError();
    }
    int err = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&err, pool->dev.err_flag, sizeof(int), hipMemcpyDeviceToHost, pool->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(pool->stream);
    (void)hipFree(dev);
    if (e != hipSuccess) return mz::fail(MZ_ERR_HIP, "set_selection failed: %s", hipGetErrorString(e));
    if (err & 16) {
        MZ_HIP(hipMemset(pool->dev.err_flag, 0, sizeof(int)));
        return mz::fail(MZ_ERR_INVALID_ARG, "the given path is not a root-to-leaf chain of tree %d (an action that is not a "
                                            "child, a link that does not lead to the next record, or a visited leaf)", tree);
    }
    pool->pending = true;
    return MZ_OK;
}

int mz_expand_backprop(mz_pool_t* pool, const float* priors, const float* reward, const float* value) {
    return backprop_impl(pool, priors, reward, value, false, nullptr, nullptr, false);
}

int mz_expand_backprop_logits(mz_pool_t* pool, const float* logits, const float* reward, const float* value) {
    return backprop_impl(pool, logits, reward, value, true, nullptr, nullptr, false);
}

int mz_step(mz_pool_t* pool, const float* priors, const float* reward, const float* value, int32_t* out_parent_block,
            int32_t* out_action) {
    return backprop_impl(pool, priors, reward, value, false, out_parent_block, out_action, true);
}

int mz_step_logits(mz_pool_t* pool, const float* logits, const float* reward, const float* value,
                   int32_t* out_parent_block, int32_t* out_action) {
    return backprop_impl(pool, logits, reward, value, true, out_parent_block, out_action, true);
}

int mz_root_stats(mz_pool_t* pool, int32_t* visits, double* root_value, double* minmax, float* policy) {
    MZ_REQUIRE(pool != nullptr, "pool is NULL");
    if (pool->blocks_in_use < 1) return mz::fail(MZ_ERR_STATE, "root_stats before expand_roots");
    MZ_HIP(hipSetDevice(pool->device));
    hipLaunchKernelGGL(root_stats_kernel, dim3(pool->dev.T), dim3(64), 0, pool->stream, pool->dev, visits, root_value,
                       minmax, policy);
    return check_launch("root_stats_kernel");
}

int mz_select_action(mz_pool_t* pool, const double* u, const double* temperature, int32_t* out_action) {
    MZ_REQUIRE(pool != nullptr, "pool is NULL");
    MZ_REQUIRE(u && temperature && out_action, "u, temperature and out_action must be non-NULL");
    if (pool->blocks_in_use < 1) return mz::fail(MZ_ERR_STATE, "select_action before expand_roots");
    MZ_HIP(hipSetDevice(pool->device));
    if (pool->dev.A <= kSmallA)
        select_action_kernel<kSmallA><<<pool->dev.T, 64, 0, pool->stream>>>(pool->dev, u, temperature, out_action);
    else
        select_action_kernel<kMaxActions><<<pool->dev.T, 64, 0, pool->stream>>>(pool->dev, u, temperature, out_action);
    return check_launch("select_action_kernel");
}

// Internal to the library (mz_pool_internal.h): mz_root_stats followed by mz_select_action, as one launch.
__attribute__((visibility("hidden"))) int mz_root_finish(mz_pool_t* pool, int32_t* visits, double* root_value,
                                                         double* minmax, float* policy, const double* u,
                                                         const double* temperature, int32_t* out_action) {
    MZ_REQUIRE(pool != nullptr, "pool is NULL");
    MZ_REQUIRE(u && temperature && out_action, "u, temperature and out_action must be non-NULL");
    if (pool->blocks_in_use < 1) return mz::fail(MZ_ERR_STATE, "root statistics before expand_roots");
    MZ_HIP(hipSetDevice(pool->device));
    if (pool->dev.A <= kSmallA)
        root_finish_kernel<kSmallA><<<pool->dev.T, 64, 0, pool->stream>>>(pool->dev, visits, root_value, minmax, policy, u,
                                                                          temperature, out_action);
    else
        root_finish_kernel<kMaxActions><<<pool->dev.T, 64, 0, pool->stream>>>(pool->dev, visits, root_value, minmax, policy,
                                                                              u, temperature, out_action);
    return check_launch("root_finish_kernel");
}

int mz_pool_counters(mz_pool_t* pool, int64_t* out4_host) {
    MZ_REQUIRE(pool != nullptr && out4_host != nullptr, "pool/out is NULL");
    MZ_HIP(hipSetDevice(pool->device));
    hipLaunchKernelGGL(sum_counters_kernel, dim3(1), dim3(kBlockThreads), 0, pool->stream, pool->dev,
                       pool->counter_scratch);
    MZ_TRY(check_launch("sum_counters_kernel"));
    long long host[2] = {0, 0};
    MZ_HIP(hipMemcpyAsync(host, pool->counter_scratch, sizeof(host), hipMemcpyDeviceToHost, pool->stream));
    int err = 0;
    MZ_HIP(hipMemcpyAsync(&err, pool->dev.err_flag, sizeof(int), hipMemcpyDeviceToHost, pool->stream));
    MZ_HIP(hipStreamSynchronize(pool->stream));
    if (err & 8)
        return mz::fail(MZ_ERR_HIP, "a one-launch run of simulations gave up waiting for the workgroups of its row block "
                                    "(flags 0x%x): results of that move are invalid", err);
    if (err != 0) return mz::fail(MZ_ERR_INVALID_ARG, "device-side validation failed (flags 0x%x): bad legal/n_legal", err);
    out4_host[0] = host[0];
    out4_host[1] = host[1];
    out4_host[2] = pool->select_launches;
    out4_host[3] = pool->backprop_launches;
    return MZ_OK;
}

int mz_pool_read_block(mz_pool_t* pool, int32_t tree, int32_t block, int32_t* out_num_children, int32_t* actions,
                       int32_t* count, double* value_sum, double* prior, float* reward, int32_t* child) {
    MZ_REQUIRE(pool != nullptr, "pool is NULL");
    const PoolDev& d = pool->dev;
    MZ_REQUIRE(tree >= 0 && tree < d.T, "tree %d out of range [0,%d)", tree, d.T);
    MZ_REQUIRE(block >= 0 && block < pool->blocks_in_use, "block %d out of range [0,%d)", block, pool->blocks_in_use);
    MZ_HIP(hipSetDevice(pool->device));
    MZ_HIP(hipStreamSynchronize(pool->stream));
    const int A = d.A;
    int n = A;
    if (block == 0) MZ_HIP(hipMemcpy(&n, d.n_root + tree, sizeof(int), hipMemcpyDeviceToHost));
    if (out_num_children) *out_num_children = n;
    std::vector<char> recbuf(d.RS);
    MZ_HIP(hipMemcpy(recbuf.data(), d.blocks + (size_t)tree * d.tree_stride + (size_t)block * d.RS, d.RS,
                     hipMemcpyDeviceToHost));
    const Slot* slots = reinterpret_cast<const Slot*>(recbuf.data());
    const BackSlot* back = reinterpret_cast<const BackSlot*>(recbuf.data() + d.off_back);
    std::vector<double> rprior(A);
    if (block == 0)
        MZ_HIP(hipMemcpy(rprior.data(), d.root_prior + (size_t)tree * A, A * sizeof(double), hipMemcpyDeviceToHost));
    for (int s = 0; s < n; ++s) {
        // children in insertion order: the root's slots are in it; below the root slot s holds action slots[s].act
        const int k = block == 0 ? s : slots[s].act;
        if (k < 0 || k >= n) return mz::fail(MZ_ERR_STATE, "corrupt record: slot %d holds action %d", s, k);
        const bool visited = slots[s].count > 0;
        if (actions) actions[k] = block == 0 ? slots[s].act : k;
        if (count) count[k] = slots[s].count;
        if (value_sum) value_sum[k] = visited ? back[s].vsum : 0.0;
        if (prior) prior[k] = block == 0 ? rprior[s] : (double)slots[s].prior;
        if (reward) reward[k] = visited ? back[s].reward : std::numeric_limits<float>::quiet_NaN();
        if (child) child[k] = visited ? (slots[s].child_hi & kChildMask) : -1;
    }
    return MZ_OK;
}

int mz_pool_read_root_priors(mz_pool_t* pool, int32_t tree, int32_t* out_num_children, float* priors) {
    MZ_REQUIRE(pool != nullptr && priors != nullptr, "pool/priors is NULL");
    const PoolDev& d = pool->dev;
    MZ_REQUIRE(tree >= 0 && tree < d.T, "tree %d out of range [0,%d)", tree, d.T);
    MZ_REQUIRE(pool->blocks_in_use >= 1, "roots are not expanded");
    MZ_HIP(hipSetDevice(pool->device));
    MZ_HIP(hipStreamSynchronize(pool->stream));
    int n = 0;
    MZ_HIP(hipMemcpy(&n, d.n_root + tree, sizeof(int), hipMemcpyDeviceToHost));
    if (out_num_children) *out_num_children = n;
    std::vector<char> recbuf(d.RS);
    MZ_HIP(hipMemcpy(recbuf.data(), d.blocks + (size_t)tree * d.tree_stride, d.RS, hipMemcpyDeviceToHost));
    const Slot* slots = reinterpret_cast<const Slot*>(recbuf.data());
    for (int k = 0; k < n; ++k) priors[k] = slots[k].prior;
    return MZ_OK;
}

#ifdef MZ_TREE_TRACE
// experiment builds only: copies out and clears the per-wavefront phase counters of mz_tree_dev.h
int mz_debug_tree_trace(uint32_t* out, int64_t out_bytes) {
    MZ_HIP(hipDeviceSynchronize());
    MZ_REQUIRE(out_bytes == (int64_t)sizeof(unsigned) * kTraceWaves * 2 * kTracePhases, "trace buffer size");
    MZ_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tree_trace), (size_t)out_bytes));
    void* sym = nullptr;
    MZ_HIP(hipGetSymbolAddress(&sym, HIP_SYMBOL(g_tree_trace)));
    MZ_HIP(hipMemset(sym, 0, (size_t)out_bytes));
    return MZ_OK;
}
#endif

int mz_pool_num_blocks(mz_pool_t* pool) { return pool ? pool->blocks_in_use : 0; }

int64_t mz_pool_bytes(mz_pool_t* pool) { return pool ? (int64_t)pool->bytes : 0; }

}  // extern "C"
