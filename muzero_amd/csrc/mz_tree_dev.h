// Device-side tree code shared by the tree kernels (mz_tree.hip) and the fused head + tree-step kernel of the
// TicTacToe network (mz_net.hip).  See mz_tree.hip for the layout and the numerics rules.
#pragma once
#include "mz_common.h"
#include <limits>

namespace {

constexpr int kBlockThreads = 256;
constexpr int kPathShift = 12;                 // path entry .x = (block << 12) | child index
constexpr int kMaxActions = 1 << kPathShift;   // 4096
constexpr int kMaxSimulations = (1 << 19) - 2; // (count << 12 | action) and (block << 12 | slot) stay below 2^31

// Backup half of a child slot (array of structures: one 16-byte access per path node).
struct __attribute__((aligned(16))) BackSlot {
    double vsum;      // Node.value_sum (agent.py:112)
    float reward;     // Node.reward (agent.py:156)
    int pad;
};

struct PoolDev {
    int T, A, N;
    int has_bounds;
    int RS;                    // record stride in bytes (multiple of 128)
    int off_back;              // byte offset of the BackSlot array inside a record
    size_t tree_stride;        // (N+1)*RS
    char* blocks;
    double* root_prior;        // [T][A]
    int* root_action;          // [T][A]
    int* n_root;               // [T]
    int* root_count;           // [T]
    double* root_vsum;         // [T]
    double* mm_min;            // [T]
    double* mm_max;            // [T]
    int2* path;                // [T][N+1]: {(block << 12) | slot, the slot's visit count when it was selected}
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

// Record of one expanded node (block e of tree t), A child slots:
//   select half, structure of arrays:  q f64[A] | visit_count i32[A] | prior f32[A] | child i32[A]      (20*A bytes)
//   backup half, array of structures:  { value_sum f64, reward f32, pad } [A] at off_back               (16*A bytes)
// q is what ucb_score needs of a visited child, computed once per backup instead of once per look:
//   reward + discount*value (agent.py:188), and with known bounds already MinMaxStats.normalize'd (utils.py:68-76).
__device__ __forceinline__ char* rec(const PoolDev& p, int tree, int e) {
    return p.blocks + (size_t)tree * p.tree_stride + (size_t)e * (size_t)p.RS;
}
__device__ __forceinline__ double* f_q(const PoolDev& p, char* r) { return reinterpret_cast<double*>(r); }
__device__ __forceinline__ int* f_count(const PoolDev& p, char* r) { return reinterpret_cast<int*>(r + 8 * (size_t)p.A); }
__device__ __forceinline__ float* f_prior(const PoolDev& p, char* r) { return reinterpret_cast<float*>(r + 12 * (size_t)p.A); }
__device__ __forceinline__ int* f_child(const PoolDev& p, char* r) { return reinterpret_cast<int*>(r + 16 * (size_t)p.A); }
__device__ __forceinline__ BackSlot* f_back(const PoolDev& p, char* r) { return reinterpret_cast<BackSlot*>(r + p.off_back); }

// ---------------------------------------------------------------------------------------------------
// All-lanes reductions inside groups of G consecutive lanes without touching LDS: butterfly steps 1 and 2 are DPP
// quad permutes, 4 and 8 the DPP half-row / row mirrors (valid because the operator is commutative and the previous
// steps made quads / half rows uniform), 16 and 32 the gfx950 row / half-wave swaps.
// ---------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int x) {
    return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
    return __int_as_float(dpp_i32<CTRL>(__float_as_int(x)));
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    return __hiloint2double(dpp_i32<CTRL>(__double2hiint(x)), dpp_i32<CTRL>(__double2loint(x)));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;

__device__ __forceinline__ double max_f64(double a, double b) { return b > a ? b : a; }   // operands are never NaN
__device__ __forceinline__ double min_f64(double a, double b) { return b < a ? b : a; }
__device__ __forceinline__ int max_i32(int a, int b) { return b > a ? b : a; }

template <int G, bool MAX>
__device__ __forceinline__ double group_minmax_f64(double x) {
    auto op = [](double a, double b) { return MAX ? max_f64(a, b) : min_f64(a, b); };
    if (G >= 2) x = op(x, dpp_f64<kDppXor1>(x));
    if (G >= 4) x = op(x, dpp_f64<kDppXor2>(x));
    if (G >= 8) x = op(x, dpp_f64<kDppHalfMirror>(x));
    if (G >= 16) x = op(x, dpp_f64<kDppMirror>(x));
    if (G >= 32) {
        const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
        const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        x = op(__hiloint2double((int)h[0], (int)l[0]), __hiloint2double((int)h[1], (int)l[1]));
    }
    if (G >= 64) {
        const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
        const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        x = op(__hiloint2double((int)h[0], (int)l[0]), __hiloint2double((int)h[1], (int)l[1]));
    }
    return x;
}

template <int G>
__device__ __forceinline__ int group_max_i32(int x) {
    if (G >= 2) x = max_i32(x, dpp_i32<kDppXor1>(x));
    if (G >= 4) x = max_i32(x, dpp_i32<kDppXor2>(x));
    if (G >= 8) x = max_i32(x, dpp_i32<kDppHalfMirror>(x));
    if (G >= 16) x = max_i32(x, dpp_i32<kDppMirror>(x));
    if (G >= 32) {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
        x = max_i32((int)r[0], (int)r[1]);
    }
    if (G >= 64) {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
        x = max_i32((int)r[0], (int)r[1]);
    }
    return x;
}

// keeps a loaded value in a register at this point of the program: the compiler can neither sink its load into a
// later conditional block nor wait for it earlier than here
__device__ __forceinline__ void pin(int& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(double& x) { asm volatile("" : "+v"(x)); }

// ---------------------------------------------------------------------------------------------------
// select_leaf (agent.py:177-190) for one tree by a group of G lanes; lane gl scores children gl, gl+G, ...  Every
// lane of the wavefront calls this (inactive lanes with active=false): the reductions are wave-wide instructions.
// P = children per lane whose loads are in flight together (P*G >= A for one round trip per level).
// ---------------------------------------------------------------------------------------------------
template <int G, int P = 1>
__device__ void select_tree(const PoolDev& p, int tree, bool active, int gl, int* out_parent, int* out_action) {
    const int A = p.A;
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
    // isnan(child.value) -> the default value, normalised like any other (agent.py:188, utils.py:68-76): the
    // exploitation term of every unvisited child
    const double dflt = ident ? p.default_value : (p.default_value - na) / nd;
    const bool renorm = !p.has_bounds && !ident;             // q holds raw reward + discount*value
    const int gbase = (int)(threadIdx.x & 63u) & ~(G - 1);   // first lane of this group in the wavefront
    const double ninf = -std::numeric_limits<double>::infinity();
    for (int it = 0; it <= p.N; ++it) {          // a path holds at most N+1 nodes: every wave reaches the exit
        if (!__any(!done)) break;
        const bool root = (it == 0);             // block 0 is only ever the first node of a path
        double bs = ninf;
        int bkey = -1, bcc = 0, bch = 0;         // local best: key, (count << 12 | action), child block
        int s0nan = 0, cc0 = 0, ch0 = 0;         // child 0 (lane 0 of the group): the winner when its score is NaN
        if (!done) {
            char* r = rec(p, tree, e);
            const double* qv = f_q(p, r);
            const int* cnt = f_count(p, r);
            const float* prf = f_prior(p, r);
            const int* chd = f_child(p, r);
            const double lt = p.log_table[np_];                       // 2*log(Np+1), config.py:159-160
            for (int base = 0; base < nchild; base += P * G) {
                // One dependent round trip per level: every field of every child this lane owns is requested
                // before anything is computed (q / child of an unvisited slot are garbage and unused), and the
                // winner's count and child block come back through a shuffle, not through a second load.
                int c[P], ch[P], act[P];
                double q[P], pr[P];
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    const int k = min(base + j * G + gl, nchild - 1);
                    c[j] = cnt[k];
                    q[j] = qv[k];
                    ch[j] = chd[k];
                    if (root) {
                        pr[j] = p.root_prior[(size_t)tree * A + k];
                        act[j] = p.root_action[(size_t)tree * A + k];
                    } else {
                        pr[j] = (double)prf[k];
                        act[j] = k;
                    }
                }
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    pin(c[j]);
                    pin(q[j]);
                    pin(ch[j]);
                    pin(pr[j]);
                    pin(act[j]);
                }
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    const int k = base + j * G + gl;
                    if (k < nchild) {
                        double x = dflt, ex = lt;                             // unvisited: lt / (0 + 1) == lt
                        if (c[j] > 0) {
                            x = renorm ? (q[j] - na) / nd : q[j];             // utils.py:68-76
                            ex = lt / (double)(c[j] + 1);                     // config.py:160
                        }
                        const double sc = x + pr[j] * ex;                     // agent.py:189-190
                        const int key = (act[j] << 16) | k;
                        const int cc = (c[j] << kPathShift) | act[j];
                        if (k == 0) {
                            s0nan = (sc != sc);
                            cc0 = cc;
                            ch0 = ch[j];
                        }
                        if (sc == sc && (bkey < 0 || sc > bs || (sc == bs && key > bkey))) {
                            bs = sc;
                            bkey = key;
                            bcc = cc;
                            bch = ch[j];
                        }
                    }
                }
            }
        }
        // arg-max over the group: the largest non-NaN score, ties to the largest (action, index) key (trap T1)
        const double gmax = group_minmax_f64<G, true>(bs);
        const int gkey = group_max_i32<G>((bkey >= 0 && bs == gmax) ? bkey : -1);
        // Python's max() over (score, action, child) tuples never replaces a NaN-scored first element and never
        // accepts a NaN-scored later one (agent.py:184): NaN first child wins, else the non-NaN arg-max.
        const bool first_wins = ((__ballot(s0nan != 0) >> gbase) & 1ull) != 0 || gkey < 0;
        const int k = first_wins ? 0 : (gkey & 0xffff);
        const int owner = first_wins ? 0 : (k & (G - 1));            // child k is scored by lane k % G
        const int wcc = __shfl(first_wins ? cc0 : bcc, owner, G);
        const int wchild = __shfl(first_wins ? ch0 : bch, owner, G);
        const int wc = wcc >> kPathShift;
        if (!done) {
            if (gl == 0) p.path[(size_t)tree * (p.N + 1) + depth] = make_int2((e << kPathShift) | k, wc);
            if (wc > 0) {                    // visited <=> expanded: descend (agent.py:179-180)
                np_ = wc;
                e = wchild;
                nchild = A;
                ++depth;
            } else {
                done = true;
                if (gl == 0) {
                    out_parent[tree] = e;
                    out_action[tree] = wcc & (kMaxActions - 1);
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

// NumPy float32 pairwise summation (trap T4) of a[0..A) by the lanes of one group, through global memory.  The
// host compiled the recursion for n = A into leaves (lo, n<=128) and merges (dst += src); lanes 0..7 are the eight
// strided accumulators of a leaf.  `scratch` (>= n_sum_leaf floats) receives the leaf sums.  Returns the sum in
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
            } else {   // a group narrower than the eight accumulators: its lane 0 keeps all of them
                res = 0.f;
                if (gl == 0) {
                    float acc[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = a[lo + i];
                    for (int m = 8; m < nfull; m += 8) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc[i] += a[lo + m + i];
                    }
                    res = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
                }
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

// LDS row of one group of G lanes (one tree), owned by the wavefront the group lies in -- DS operations of a wavefront
// execute in program order, so the lanes of a group hand values to each other through it without a barrier.
// Holds, one after the other in time: the softmax terms of the new node + the leaf sums of the NumPy recursion;
// then the rewards along the path (floats) and the values reaching each path position (doubles).
template <int G>
struct TreeLds {
    static constexpr int kPB = 4;                            // path positions per lane held in registers
    static constexpr int kPathCap = G * kPB;                 // longest path the LDS chain replays
    static constexpr int kMaxA = G == 64 ? 1024 : 8 * G;     // widest action axis staged in the row
    static constexpr int kLeafScratch = G == 64 ? 32 : 16;   // leaves of the pairwise recursion (n <= 128 each)
    static constexpr int kSoftmax = kMaxA + kLeafScratch;
    static constexpr int kChain = 3 * kPathCap + 2;          // floats + doubles (8-byte aligned)
    static constexpr int kRow = ((kSoftmax > kChain ? kSoftmax : kChain) + 3) / 4 * 4;   // floats, 16-byte multiple
    static constexpr int kGroups = kBlockThreads / G;
};

// NumPy float32 pairwise summation (trap T4) over a row a[0..A) staged in LDS by one group: eight lanes (an octet)
// per leaf, G/8 leaves at a time, the 8-accumulator combine by DPP; groups narrower than an octet leave it to
// their lane 0.  Returns the sum in every lane of the group.
template <int G>
__device__ __forceinline__ float numpy_sum_lds(const PoolDev& p, const float* a, float* scratch, int gl) {
    if (G >= 8) {
        constexpr int OCT = G >= 8 ? G / 8 : 1;
        const int oct = gl >> 3, i = gl & 7;
        for (int s0 = 0; s0 < p.n_sum_leaf; s0 += OCT) {
            const int s = s0 + oct;
            const bool has = s < p.n_sum_leaf;
            const int lo = has ? p.sum_leaf[2 * s] : 0, n = has ? p.sum_leaf[2 * s + 1] : 0;
            const int nfull = n - (n % 8);
            float r = 0.f;
            if (nfull >= 8) {
                r = a[lo + i];
                for (int m = 8; m < nfull; m += 8) r += a[lo + m + i];
            }
            r += dpp_f32<kDppXor1>(r);             // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)): fp32 + is commutative,
            r += dpp_f32<kDppXor2>(r);             // so every lane of the octet ends with lane 0's bits
            r += dpp_f32<kDppHalfMirror>(r);
            if (has && i == 0) {
                float res = n < 8 ? 0.f : r;
                for (int t = nfull; t < n; ++t) res += a[lo + t];
                scratch[s] = res;
            }
        }
    } else if (gl == 0) {
        for (int s = 0; s < p.n_sum_leaf; ++s) {
            const int lo = p.sum_leaf[2 * s], n = p.sum_leaf[2 * s + 1];
            const int nfull = n - (n % 8);
            float res = 0.f;
            if (nfull >= 8) {
                float r[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) r[i] = a[lo + i];
                for (int m = 8; m < nfull; m += 8) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) r[i] += a[lo + m + i];
                }
                res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
            }
            for (int t = nfull; t < n; ++t) res += a[lo + t];
            scratch[s] = res;
        }
    }
    if (gl == 0) {
        for (int m = 0; m < p.n_sum_merge; ++m) {
            const int dst = p.sum_merge[2 * m], src = p.sum_merge[2 * m + 1];
            scratch[dst] = scratch[dst] + scratch[src];
        }
    }
    __builtin_amdgcn_wave_barrier();
    return scratch[0];
}

// NumPy float32 pairwise summation of one value per lane (lane k < n holds a[k], the others 0), n <= G <= 64: the
// single-leaf case of the recursion above, entirely in registers.  Returns the sum in every lane of the group.
template <int G>
__device__ __forceinline__ float numpy_sum_lanes(float x, int n, int gl) {
    float res = 0.f;
    if (n < 8) {
        for (int i = 0; i < n; ++i) res += __shfl(x, i, G);
    } else {
        const int nfull = n - (n % 8);
        float r = x;                                                   // lanes 0..7: the eight strided accumulators
        for (int i = 8; i < nfull; i += 8) r += __shfl(x, (gl + i) & (G - 1), G);
        r += dpp_f32<kDppXor1>(r);                                     // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
        r += dpp_f32<kDppXor2>(r);
        r += dpp_f32<kDppHalfMirror>(r);
        res = r;
        for (int i = nfull; i < n; ++i) res += __shfl(x, i, G);
    }
    return __shfl(res, 0, G);
}

// ---------------------------------------------------------------------------------------------------
// expand_node(leaf, all actions) + backpropagate(leaf, value)  (agent.py:208-211, 154-160, 192-196)
// ---------------------------------------------------------------------------------------------------
// What the backup reads before it can do anything, requested as early as the caller can (the fused inference +
// tree kernel asks for it before the network head runs): the path length, this lane's PB path entries (positions
// gl, gl+G, ... counted from the root; entries past the length are stale but valid: the path array starts zeroed)
// with the slots' visit counts as select saw them, the slots' backup halves, and the root's statistics.
template <int PB>
struct BackupPrefetchT {
    int len, root_count;
    double root_vsum;
    int slot[PB], c[PB];
    double vs[PB];
    float rw[PB];
};
using BackupPrefetch = BackupPrefetchT<1>;

template <int G, int PB = 1>
__device__ __forceinline__ BackupPrefetchT<PB> backprop_prefetch(const PoolDev& p, int tree, bool active, int gl) {
    BackupPrefetchT<PB> f;
    f.len = 0;
    f.root_count = 0;
    f.root_vsum = 0.0;
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        f.slot[j] = 0;
        f.c[j] = 0;
        f.vs[j] = 0.0;
        f.rw[j] = 0.f;
    }
    if (active) {
        f.len = p.path_len[tree];
        int2 pe[PB];
#pragma unroll
        for (int j = 0; j < PB; ++j) pe[j] = p.path[(size_t)tree * (p.N + 1) + min(gl + G * j, p.N)];
        f.root_count = p.root_count[tree];
        f.root_vsum = p.root_vsum[tree];
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            f.slot[j] = pe[j].x;
            f.c[j] = pe[j].y;
            const BackSlot b = f_back(p, rec(p, tree, pe[j].x >> kPathShift))[pe[j].x & (kMaxActions - 1)];
            f.vs[j] = b.vsum;
            f.rw[j] = b.reward;
        }
    }
    return f;
}

// Node.update_value (agent.py:119-123) on slot k of record r, plus what select will want of it next time.
// c_old/vs_old: the slot before (vs_old ignored for an unvisited slot); rw: the node's reward.
__device__ __forceinline__ void backup_slot(const PoolDev& p, char* r, int k, int c_old, double vs_old, double vin,
                                            float rw, bool leaf, int nb, double& mn_l, double& mx_l) {
    const int c = c_old + 1;
    const double vs = (c_old > 0 ? vs_old : 0.0) + vin;
    f_count(p, r)[k] = c;
    BackSlot* b = f_back(p, r) + k;
    if (leaf) {                                   // the leaf: now expanded (agent.py:156-157)
        BackSlot nbk;
        nbk.vsum = vs;
        nbk.reward = rw;
        nbk.pad = 0;
        *b = nbk;
        f_child(p, r)[k] = nb;
    } else {
        b->vsum = vs;
    }
    const double mean = vs / (double)c;
    if (mean > mx_l) mx_l = mean;                 // MinMaxStats.update, utils.py:63-66
    if (mean < mn_l) mn_l = mean;
    // ucb_score's exploitation term of this child (agent.py:188), normalised here when the bounds are known
    const double ex = (mean == mean) ? (double)rw + p.discount * mean : p.default_value;
    f_q(p, r)[k] = p.has_bounds ? (ex - p.kmin) / (p.kmax - p.kmin) : ex;
}

// pol_row: this tree's A priors or logits (global or LDS); rnew, value_f32: its scalarised reward and value.
// lds_row: this group's TreeLds<G>::kRow floats of LDS (16-byte aligned), or nullptr (then PB must be 1, and the
// softmax of an action axis wider than the group goes through global memory and a workgroup barrier).
template <int G, bool LOGITS, int PB = 1>
__device__ void backprop_tree(const PoolDev& p, int tree, bool active, int gl, int nb, const float* pol_row,
                              float rnew, float value_f32, const BackupPrefetchT<PB>& pf, float* lds_row = nullptr) {
    using L = TreeLds<G>;
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
    } else if (LOGITS && lds_row != nullptr && A <= L::kMaxA) {
        // exp terms staged in the group's LDS row, summed there in NumPy's order
        for (int k = gl; k < A; k += G) lds_row[k] = active ? expf(pol_row[k]) : 0.f;
        __builtin_amdgcn_wave_barrier();
        const float total = numpy_sum_lds<G>(p, lds_row, lds_row + L::kMaxA, gl);
        if (active) {
            char* r = rec(p, tree, nb);
            int* cnt = f_count(p, r);
            float* pr = f_prior(p, r);
            for (int k = gl; k < A; k += G) {
                cnt[k] = 0;
                pr[k] = lds_row[k] / total;                               // network.py:146
            }
        }
        __builtin_amdgcn_wave_barrier();
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
            // the child-block field of a fresh record is unused until the first backup through a slot: scratch
            float total = numpy_sum_f32<G>(p, f_prior(p, r), reinterpret_cast<float*>(f_child(p, r)),
                                           active ? gl : 1 << 30);
            if (active) {
                float* pr = f_prior(p, r);
                for (int k = gl; k < A; k += G) pr[k] = pr[k] / total;   // network.py:146
            }
        }
    }
    // Backup chain leaf -> root: the fp64 recurrence value = reward + discount*value (agent.py:195) is replayed in
    // order; the lane that owns a path position keeps the value that reached it.
    const int top = len - 1;
    double mn_l = std::numeric_limits<double>::infinity();
    double mx_l = -std::numeric_limits<double>::infinity();
    if (lds_row != nullptr && !__any(len > L::kPathCap)) {
        // every lane of the group replays the chain from the rewards staged in the group's LDS row and leaves the
        // value entering each position there
        float* rws = lds_row;
        double* vins = reinterpret_cast<double*>(lds_row + L::kPathCap);
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int pos = gl + G * j;
            if (pos < len) rws[pos] = (pos == top) ? rnew : pf.rw[j];
        }
        __builtin_amdgcn_wave_barrier();
        for (int pos = top; pos >= 0; --pos) {
            vins[pos] = v;
            v = (double)rws[pos] + p.discount * v;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int pos = gl + G * j;
            if (active && pos < len)
                backup_slot(p, rec(p, tree, pf.slot[j] >> kPathShift), pf.slot[j] & (kMaxActions - 1), pf.c[j], pf.vs[j],
                            vins[pos], (pos == top) ? rnew : pf.rw[j], pos == top, nb, mn_l, mx_l);
        }
    } else if (!__any(len > G)) {
        // every path of this wavefront fits one group: lane gl owns position gl, its slot came with the prefetch;
        // the rewards travel by shuffles
        const bool has = active && gl < len;
        const float rw = (gl == top) ? rnew : pf.rw[0];
        double vin = v;
        for (int j = G - 1; j >= 0; --j) {
            if (!__any(j < len)) continue;
            const float rj = __shfl(rw, j, G);
            if (gl == j) vin = v;
            if (j < len) v = (double)rj + p.discount * v;
        }
        if (has)
            backup_slot(p, rec(p, tree, pf.slot[0] >> kPathShift), pf.slot[0] & (kMaxActions - 1), pf.c[0], pf.vs[0], vin,
                        rw, gl == top, nb, mn_l, mx_l);
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
                const int2 pe = p.path[(size_t)tree * (p.N + 1) + pos];
                k = pe.x & (kMaxActions - 1);
                r = rec(p, tree, pe.x >> kPathShift);
                c = pe.y;
                const BackSlot b = f_back(p, r)[k];
                vs = b.vsum;
                rw = (pos == top) ? rnew : b.reward;
            }
            const int n_here = more ? min(G, len - base) : 0;
            double vin = v;
            for (int j = 0; j < G; ++j) {
                if (!__any(j < n_here)) break;
                const float rj = __shfl(rw, j, G);
                if (gl == j) vin = v;
                if (j < n_here) v = (double)rj + p.discount * v;
            }
            if (has) backup_slot(p, r, k, c, vs, vin, rw, pos == top, nb, mn_l, mx_l);
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
        // NaN means never enter mx_l / mn_l (the comparisons above are false), so these are plain reductions
        mx_l = group_minmax_f64<G, true>(mx_l);
        mn_l = group_minmax_f64<G, false>(mn_l);
        if (active && gl == 0) {
            if (mx_l > p.mm_max[tree]) p.mm_max[tree] = mx_l;
            if (mn_l < p.mm_min[tree]) p.mm_min[tree] = mn_l;
        }
    }
}

}  // namespace
