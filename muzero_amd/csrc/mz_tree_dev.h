// Device-side tree code shared by the tree kernels (mz_tree.hip) and the fused head + tree-step kernel of the
// TicTacToe network (mz_net.hip).  See mz_tree.hip for the layout and the numerics rules.
#pragma once
#include "mz_common.h"
#include <limits>

namespace {

constexpr int kBlockThreads = 256;
constexpr int kPathShift = 12;                 // path entry = (block << 12) | child index
constexpr int kMaxActions = 1 << kPathShift;   // 4096
constexpr int kMaxSimulations = (1 << 19) - 2;

struct PoolDev {
    int T, A, N;
    int has_bounds;
    int RS;                    // record stride in bytes
    size_t tree_stride;        // (N+1)*RS
    char* blocks;
    double* root_prior;        // [T][A]
    int* root_action;          // [T][A]
    int* n_root;               // [T]
    int* root_count;           // [T]
    double* root_vsum;         // [T]
    double* mm_min;            // [T]
    double* mm_max;            // [T]
    int* path;                 // [T][N+1]
    int* path_len;             // [T]
    long long* sum_len;        // [T] cumulative sum of path lengths backed up (measurement)
    long long* sims;           // [T] cumulative simulations backed up
    int* err_flag;             // [1]
    const double* log_table;   // [N+2]
    const int* sum_leaf;       // NumPy pairwise-sum program for n = A: leaves (lo, n) ...
    const int* sum_merge;      // ... and merges (dst, src)
    int n_sum_leaf, n_sum_merge;
    double discount, default_value, kmin, kmax;
};

__device__ __forceinline__ char* rec(const PoolDev& p, int tree, int e) {
    return p.blocks + (size_t)tree * p.tree_stride + (size_t)e * (size_t)p.RS;
}
__device__ __forceinline__ double* f_vsum(const PoolDev& p, char* r) { return reinterpret_cast<double*>(r); }
__device__ __forceinline__ int* f_count(const PoolDev& p, char* r) { return reinterpret_cast<int*>(r + 8 * (size_t)p.A); }
__device__ __forceinline__ float* f_prior(const PoolDev& p, char* r) { return reinterpret_cast<float*>(r + 12 * (size_t)p.A); }
__device__ __forceinline__ float* f_reward(const PoolDev& p, char* r) { return reinterpret_cast<float*>(r + 16 * (size_t)p.A); }
__device__ __forceinline__ int* f_child(const PoolDev& p, char* r) { return reinterpret_cast<int*>(r + 20 * (size_t)p.A); }

// ---------------------------------------------------------------------------------------------------
// select_leaf (agent.py:177-190) for one tree by a group of G lanes.  Every lane of the wavefront calls
// this (inactive lanes with active=false) because the reduction shuffles must be wave-uniform.
// ---------------------------------------------------------------------------------------------------
template <int G>
__device__ void select_tree(const PoolDev& p, int tree, bool active, int gl, int* out_parent, int* out_action) {
    int e = 0, np_ = 0, nchild = 0, depth = 0;
    bool done = !active;
    bool ident = false;            // MinMaxStats.normalize returns the value unchanged (utils.py:71-72)
    double na = p.kmin, nd = p.kmax - p.kmin;
    if (active) {
        np_ = p.root_count[tree];
        nchild = p.n_root[tree];
        if (!p.has_bounds) {
            const double mn = p.mm_min[tree], mx = p.mm_max[tree];
            ident = (mx == mn);
            na = mn;
            nd = mx - mn;
        }
    }
    for (int it = 0; it <= p.N; ++it) {          // a path holds at most N+1 nodes: every wave reaches the exit
        if (!__any(!done)) break;
        double bs = 0.0;
        int bkey = -1, bc = 0, bchild = 0;        // local best: key, its visit count and child block
        int s0nan = 0, c0 = 0, child0 = 0, act0 = 0;   // child 0 (lane 0): the winner when its score is NaN
        if (!done) {
            // One dependent round trip per level: every field of every child this lane owns is requested before
            // anything is computed (value_sum / reward / child of an unvisited slot are garbage and unused), and
            // the winner's count and child block come back through a shuffle, not through a second load.
            char* r = rec(p, tree, e);
            const double* vs = f_vsum(p, r);
            const int* cnt = f_count(p, r);
            const float* pr = f_prior(p, r);
            const float* rw = f_reward(p, r);
            const int* chd = f_child(p, r);
            const double lt = p.log_table[np_];                       // 2*log(Np+1), config.py:159-160
            const bool root = (e == 0);
            for (int k = gl; k < nchild; k += G) {
                const int c = cnt[k];
                const double vsum = vs[k];
                const float rwd = rw[k];
                const int ch = chd[k];
                const double prior = root ? p.root_prior[(size_t)tree * p.A + k] : (double)pr[k];
                const int act = root ? p.root_action[(size_t)tree * p.A + k] : k;
                const double val = vsum / (double)(c > 0 ? c : 1);    // Node.value, agent.py:122
                const bool visited = c > 0 && val == val;
                // isnan(node.value) -> default value, agent.py:188
                const double exploit = visited ? (double)rwd + p.discount * val : p.default_value;
                const double nrm = ident ? exploit : (exploit - na) / nd;          // utils.py:68-76
                const double sc = nrm + prior * (lt / (double)(c + 1));            // agent.py:189-190
                const int key = (act << 16) | k;
                if (k == 0) {
                    s0nan = (sc != sc);
                    c0 = c;
                    child0 = ch;
                    act0 = act;
                }
                if (sc == sc && (bkey < 0 || sc > bs || (sc == bs && key > bkey))) {
                    bs = sc;
                    bkey = key;
                    bc = c;
                    bchild = ch;
                }
            }
        }
        const int lkey = bkey;                    // this lane's own best, kept to recognise the owner of the winner
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) {
            const double os = __shfl_xor(bs, off, G);
            const int ok = __shfl_xor(bkey, off, G);
            if (ok >= 0 && (bkey < 0 || os > bs || (os == bs && ok > bkey))) {
                bs = os;
                bkey = ok;
            }
        }
        s0nan = __shfl(s0nan, 0, G);
        // Python's max() over (score, action, child) tuples never replaces a NaN-scored first element and never
        // accepts a NaN-scored later one (agent.py:184): NaN first child wins, else the non-NaN arg-max.
        const bool first_wins = s0nan || bkey < 0;
        const int k = first_wins ? 0 : (bkey & 0xffff);
        const int owner = first_wins ? 0 : (k & (G - 1));            // child k is scored by lane k % G
        (void)lkey;
        const int wc = __shfl(first_wins ? c0 : bc, owner, G);
        const int wchild = __shfl(first_wins ? child0 : bchild, owner, G);
        const int wact = first_wins ? __shfl(act0, 0, G) : (bkey >> 16);
        if (!done) {
            if (gl == 0) p.path[(size_t)tree * (p.N + 1) + depth] = (e << kPathShift) | k;
            if (wc > 0) {                    // visited <=> expanded: descend (agent.py:179-180)
                np_ = wc;
                e = wchild;
                nchild = p.A;
                ++depth;
            } else {
                done = true;
                if (gl == 0) {
                    out_parent[tree] = e;
                    out_action[tree] = wact;
                    p.path_len[tree] = depth + 1;
                }
            }
        }
    }
    if (active && !done && gl == 0) {        // unreachable for a consistent pool; never leave outputs unset
        out_parent[tree] = -1;
        out_action[tree] = -1;
        p.path_len[tree] = 0;
        atomicOr(p.err_flag, 1);
    }
    if (!active && gl == 0 && tree < p.T) {
        out_parent[tree] = -1;
        out_action[tree] = -1;
    }
}

// NumPy float32 pairwise summation (trap T4) of a[0..A) by the lanes of one group.  The host compiled the
// recursion for n = A into leaves (lo, n<=128) and merges (dst += src); lanes 0..7 are the eight strided
// accumulators of a leaf.  `scratch` (>= n_sum_leaf floats) receives the leaf sums.  Returns the sum in
// every lane.  Requires the group to have made a[] visible (workgroup fence + barrier by the caller).
template <int G>
__device__ float numpy_sum_f32(const PoolDev& p, const float* a, float* scratch, int gl) {
    constexpr int W = G < 8 ? G : 8;
    for (int s = 0; s < p.n_sum_leaf; ++s) {
        const int lo = p.sum_leaf[2 * s], n = p.sum_leaf[2 * s + 1];
        float res;
        if (n < 8) {
            res = 0.f;
            if (gl == 0)
                for (int i = 0; i < n; ++i) res += a[lo + i];
        } else {
            const int nfull = n - (n % 8);
            float r = 0.f;
            if (W == 8) {
                if (gl < 8) {
                    r = a[lo + gl];
                    for (int i = 8; i < nfull; i += 8) r += a[lo + i + gl];
                }
                // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
                r += __shfl_xor(r, 1, W);
                r += __shfl_xor(r, 2, W);
                r += __shfl_xor(r, 4, W);
                res = r;
            } else {   // groups narrower than 8 lanes only exist for A <= 4 < 8: not reached
                res = 0.f;
            }
            if (gl == 0)
                for (int i = nfull; i < n; ++i) res += a[lo + i];
        }
        if (gl == 0) scratch[s] = res;
    }
    if (gl == 0) {
        for (int m = 0; m < p.n_sum_merge; ++m) {
            const int dst = p.sum_merge[2 * m], src = p.sum_merge[2 * m + 1];
            scratch[dst] = scratch[dst] + scratch[src];
        }
    }
    float total = (gl == 0) ? scratch[0] : 0.f;
    return __shfl(total, 0, G);
}

// NumPy float32 pairwise summation of one value per lane (lane k < n holds a[k], the others 0), n <= G <= 64: the
// single-leaf case of the recursion above, entirely in registers.  Returns the sum in every lane of the group.
template <int G>
__device__ __forceinline__ float numpy_sum_lanes(float x, int n, int gl) {
    float res = 0.f;
    if (n < 8) {
        for (int i = 0; i < n; ++i) res += __shfl(x, i, G);
    } else {
        constexpr int W = G < 8 ? G : 8;
        const int nfull = n - (n % 8);
        float r = x;                                                   // lanes 0..7: the eight strided accumulators
        for (int i = 8; i < nfull; i += 8) r += __shfl(x, (gl + i) & (G - 1), G);
        r += __shfl_xor(r, 1, W);                                      // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
        r += __shfl_xor(r, 2, W);
        r += __shfl_xor(r, 4, W);
        res = r;
        for (int i = nfull; i < n; ++i) res += __shfl(x, i, G);
    }
    return __shfl(res, 0, G);
}

// ---------------------------------------------------------------------------------------------------
// expand_node(leaf, all actions) + backpropagate(leaf, value)  (agent.py:208-211, 154-160, 192-196)
// ---------------------------------------------------------------------------------------------------
// What the backup reads before it can do anything, requested as early as the caller can (the fused inference +
// tree kernel asks for it before the network head runs): the path length, this lane's path slot (position gl,
// counted from the root; entries past the length are stale but valid: the path array starts zeroed), the slot's
// statistics, and the root's.
struct BackupPrefetch {
    int len, slot, c, root_count;
    double vs, root_vsum;
    float rw;
};

template <int G>
__device__ __forceinline__ BackupPrefetch backprop_prefetch(const PoolDev& p, int tree, bool active, int gl) {
    BackupPrefetch f;
    f.len = 0;
    f.slot = 0;
    f.c = 0;
    f.root_count = 0;
    f.vs = 0.0;
    f.root_vsum = 0.0;
    f.rw = 0.f;
    if (active) {
        f.len = p.path_len[tree];
        f.slot = p.path[(size_t)tree * (p.N + 1) + min(gl, p.N)];
        f.root_count = p.root_count[tree];
        f.root_vsum = p.root_vsum[tree];
        const int k = f.slot & (kMaxActions - 1);
        char* r = rec(p, tree, f.slot >> kPathShift);
        f.c = f_count(p, r)[k];
        f.vs = f_vsum(p, r)[k];
        f.rw = f_reward(p, r)[k];
    }
    return f;
}

// pol_row: this tree's A priors or logits (global or LDS); rnew, value_f32: its scalarised reward and value.
template <int G, bool LOGITS>
__device__ void backprop_tree(const PoolDev& p, int tree, bool active, int gl, int nb, const float* pol_row,
                              float rnew, float value_f32, const BackupPrefetch& pf) {
    const int A = p.A;
    const int len = pf.len;
    double v = active ? (double)value_f32 : 0.0;
    if (LOGITS && A <= G) {
        // masked_policy over all actions (network.py:142-146): exp, NumPy-ordered fp32 sum, IEEE divide -- one
        // prior per lane, no trip through memory
        const float x = (active && gl < A) ? expf(pol_row[gl]) : 0.f;     // no max-subtraction (network.py:145)
        const float total = numpy_sum_lanes<G>(x, A, gl);
        if (active && gl < A) {
            char* r = rec(p, tree, nb);
            f_count(p, r)[gl] = 0;
            f_prior(p, r)[gl] = x / total;
        }
    } else {
        if (active) {
            char* r = rec(p, tree, nb);
            int* cnt = f_count(p, r);
            float* pr = f_prior(p, r);
            for (int k = gl; k < A; k += G) {
                cnt[k] = 0;
                pr[k] = LOGITS ? expf(pol_row[k]) : pol_row[k];
            }
        }
        if (LOGITS) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            char* r = rec(p, active ? tree : 0, active ? nb : 0);
            float total = numpy_sum_f32<G>(p, f_prior(p, r), f_reward(p, r), active ? gl : 1 << 30);
            if (active) {
                float* pr = f_prior(p, r);
                for (int k = gl; k < A; k += G) pr[k] = pr[k] / total;   // network.py:146
            }
        }
    }
    // Backup chain leaf -> root: all lanes replay the fp64 recurrence value = reward + discount*value
    // (agent.py:195) through shuffles; the lane that owns a path position keeps the value that reached it.
    const int top = len - 1;
    double mn_l = std::numeric_limits<double>::infinity();
    double mx_l = -std::numeric_limits<double>::infinity();
    if (!__any(len > G)) {
        // every path of this wavefront fits one group: lane gl owns position gl, its slot came with the prefetch
        const bool has = active && gl < len;
        const float rw = (gl == top) ? rnew : pf.rw;
        double vin = v;
        for (int j = G - 1; j >= 0; --j) {
            if (!__any(j < len)) continue;
            const float rj = __shfl(rw, j, G);
            if (gl == j) vin = v;
            if (j < len) v = (double)rj + p.discount * v;
        }
        if (has) {
            const int k = pf.slot & (kMaxActions - 1);
            char* r = rec(p, tree, pf.slot >> kPathShift);
            const int c = pf.c + 1;                   // Node.update_value, agent.py:119-123
            const double vs = (pf.c > 0 ? pf.vs : 0.0) + vin;
            f_count(p, r)[k] = c;
            f_vsum(p, r)[k] = vs;
            const double mean = vs / (double)c;
            if (mean > mx_l) mx_l = mean;             // MinMaxStats.update, utils.py:63-66
            if (mean < mn_l) mn_l = mean;
            if (gl == top) {                          // the leaf: now expanded (agent.py:156-157)
                f_child(p, r)[k] = nb;
                f_reward(p, r)[k] = rnew;
            }
        }
    } else {
        // long paths: groups of G positions from the leaf down, lane j of a round owns position top-base-j
        for (int base = 0;; base += G) {
            const bool more = base < len;
            if (!__any(more)) break;
            const int pos = top - base - gl;
            const bool has = more && pos >= 0;
            int c = 0, k = 0;
            double vs = 0.0;
            float rw = 0.f;
            char* r = nullptr;
            if (has) {
                const int slot = p.path[(size_t)tree * (p.N + 1) + pos];
                k = slot & (kMaxActions - 1);
                r = rec(p, tree, slot >> kPathShift);
                c = f_count(p, r)[k];
                if (c > 0) vs = f_vsum(p, r)[k];
                rw = (pos == top) ? rnew : f_reward(p, r)[k];
            }
            const int n_here = more ? min(G, len - base) : 0;
            double vin = v;
            for (int j = 0; j < G; ++j) {
                if (!__any(j < n_here)) break;
                const float rj = __shfl(rw, j, G);
                if (gl == j) vin = v;
                if (j < n_here) v = (double)rj + p.discount * v;
            }
            if (has) {
                c += 1;
                vs = vs + vin;
                f_count(p, r)[k] = c;
                f_vsum(p, r)[k] = vs;
                const double mean = vs / (double)c;
                if (mean > mx_l) mx_l = mean;
                if (mean < mn_l) mn_l = mean;
                if (pos == top) {
                    f_child(p, r)[k] = nb;
                    f_reward(p, r)[k] = rnew;
                }
            }
        }
    }
    if (active && gl == 0) {                          // the root (reward None -> chain ends, agent.py:195)
        const int rc = pf.root_count + 1;
        const double rvs = pf.root_vsum + v;
        p.root_count[tree] = rc;
        p.root_vsum[tree] = rvs;
        const double mean = rvs / (double)rc;
        if (mean > mx_l) mx_l = mean;
        if (mean < mn_l) mn_l = mean;
        p.sum_len[tree] += len;
        p.sims[tree] += 1;
    }
    if (!p.has_bounds) {
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) {
            const double omx = __shfl_xor(mx_l, off, G);
            const double omn = __shfl_xor(mn_l, off, G);
            if (omx > mx_l) mx_l = omx;
            if (omn < mn_l) mn_l = omn;
        }
        if (active && gl == 0) {
            if (mx_l > p.mm_max[tree]) p.mm_max[tree] = mx_l;
            if (mn_l < p.mm_min[tree]) p.mm_min[tree] = mn_l;
        }
    }
}

}  // namespace
