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

// Select half of a child slot.  A record is Slot[A] followed by BackSlot[A].
//   Root record (block 0): slot k is the k-th listed legal action (insertion order; prior unused: the root's priors
//   are fp64 after the noise mix and live in root_prior).
//   Other records: slot 0 is action 0 (the first child in insertion order, whose NaN score decides Python's max);
//   slots 1..A-1 hold the other actions by descending prior when the node is `sorted`, else in action order.
struct __attribute__((aligned(8))) Slot {
    double q;         // exploitation term of ucb_score for a visited child (see backup_slot); garbage when count == 0
    int count;        // Node.visit_count (agent.py:113)
    float prior;      // Node.prior (agent.py:110)
    int child_hi;     // (hi << 20) | block of the child node, written when the child is expanded;
                      //   hi = 1 + the largest visited slot of THAT node (>= 1), kept up to date by select
    int act;          // the action of this slot
};
static_assert(sizeof(Slot) == 24, "Slot layout");
constexpr int kChildBits = 20;                 // block index < 2^20 (kMaxSimulations), hi <= 4095
constexpr int kChildMask = (1 << kChildBits) - 1;

// Backup half of a child slot (one 16-byte access per path node).
struct __attribute__((aligned(16))) BackSlot {
    double vsum;      // Node.value_sum (agent.py:112)
    float reward;     // Node.reward (agent.py:156)
    int pad;
};

constexpr int kPowTables = 8;                  // temperature tables a pool holds (mz_pool_set_temperature_table)

struct PoolDev {
    int T, A, N;
    int has_bounds;
    int RS;                    // record stride in bytes (multiple of 128)
    int off_back;              // byte offset of the BackSlot array inside a record
    int sort_nodes;            // new nodes keep their children by descending prior (select reads a prefix)
    int hi_unsorted;           // hi of an unsorted node: min(A-1, 4095), i.e. always scan everything
    size_t tree_stride;        // (N+1)*RS
    char* blocks;
    double* root_prior;        // [T][A]
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
    int n_pow;                 // registered temperature tables
    double pow_t[kPowTables];  // their temperatures ...
    const double* pow_table;   // ... and weights: [kPowTables][N+1], entry c = float(c) ** (1/t) by the host's libm
};

__device__ __forceinline__ char* rec(const PoolDev& p, int tree, int e) {
    return p.blocks + (size_t)tree * p.tree_stride + (size_t)e * (size_t)p.RS;
}
__device__ __forceinline__ Slot* f_slots(const PoolDev& p, char* r) { return reinterpret_cast<Slot*>(r); }
__device__ __forceinline__ BackSlot* f_back(const PoolDev& p, char* r) { return reinterpret_cast<BackSlot*>(r + p.off_back); }

// ---------------------------------------------------------------------------------------------------
// All-lanes reductions inside groups of G consecutive lanes without touching LDS: butterfly steps 1 and 2 are DPP
// quad permutes, 4 and 8 the DPP half-row / row mirrors (valid because the operator is commutative and the previous
// steps made quads / half rows uniform), 16 and 32 the gfx950 row / half-wave swaps.
// ---------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int x) {
    // every lane is written (full row / bank masks, permutations inside a row): the destination's old value is
    // irrelevant, and left undefined the compiler neither copies x first nor keeps the move apart from its consumer
    return __builtin_amdgcn_mov_dpp(x, CTRL, 0xf, 0xf, true);
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

// operands are never NaN: exactly one v_max_f64 / v_min_f64 each (through the builtins the compiler first canonicalises
// every operand, an extra v_max_f64 x, x per operand and reduction step)
__device__ __forceinline__ double max_f64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double min_f64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
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

// Phase timing for experiment builds (-DMZ_TREE_TRACE, tools/tree_trace.py): lane 0 of every wavefront adds the shader
// clocks since its previous stamp to g_tree_trace[phase] after draining its memory operations.  Compiles to nothing in
// the library.
#ifdef MZ_TREE_TRACE
constexpr int kTraceWaves = 1 << 15, kTracePhases = 12;
__device__ unsigned g_tree_trace[kTraceWaves][2 * kTracePhases];    // per wavefront: clocks, then stamp counts
struct TreeTrace {
    long long t;
    unsigned clk[kTracePhases], cnt[kTracePhases];
    __device__ TreeTrace() : t(clock64()) {
        for (int i = 0; i < kTracePhases; ++i) clk[i] = cnt[i] = 0;
    }
    __device__ __forceinline__ void stamp(int phase) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const long long now = clock64();
#pragma unroll
        for (int i = 0; i < kTracePhases; ++i)
            if (i == phase) {
                clk[i] += (unsigned)(now - t);
                cnt[i] += 1;
            }
        t = clock64();
    }
    __device__ __forceinline__ void stamp_nowait(int phase) {       // without draining memory operations
        const long long now = clock64();
#pragma unroll
        for (int i = 0; i < kTracePhases; ++i)
            if (i == phase) {
                clk[i] += (unsigned)(now - t);
                cnt[i] += 1;
            }
        t = now;
    }
    __device__ ~TreeTrace() {
        const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
        if ((threadIdx.x & 63) == 0 && wave < kTraceWaves)
            for (int i = 0; i < kTracePhases; ++i)
                if (cnt[i]) {
                    g_tree_trace[wave][i] += clk[i];
                    g_tree_trace[wave][kTracePhases + i] += cnt[i];
                }
    }
};
#define MZ_TRACE_DECL TreeTrace mz_trace_
#define MZ_TRACE(phase) mz_trace_.stamp(phase)
#ifdef MZ_JOB_TRACE
#define MZ_TRACE_NW(phase) mz_trace_.stamp_nowait(phase)
#else
#define MZ_TRACE_NW(phase)
#endif
#else
#define MZ_TRACE_NW(phase)
#define MZ_TRACE_DECL
#define MZ_TRACE(phase)
#endif

// ---------------------------------------------------------------------------------------------------
// select_leaf (agent.py:177-190) for one tree by a group of G lanes; lane gl scores slots gl, gl+G, ...  Every
// lane of the wavefront calls this (inactive lanes with active=false): the reductions are wave-wide instructions.
// P = slots per lane whose loads are in flight together.
//
// At the root every listed child is scored.  Below it a sorted node is scored over a prefix: with hi = 1 + its
// largest visited slot, slots [0, hi+2) hold every visited child, slot 0, and the two unvisited children of largest
// prior.  An unvisited child scores normalize(default) + prior*(2 log(Np+1)), non-decreasing in the prior (IEEE
// rounding is monotone, the factor is >= 0: checked when the pool is created), so no slot past the prefix can beat
// the last slot of the prefix; it can only TIE with it, and ties go to the largest action, so the prefix grows
// while its last slot's score is not strictly below the best one.  Bit-identical to scoring every child.
// ---------------------------------------------------------------------------------------------------
// The exploration table 2*log(n+1), n <= N+1, staged in LDS by the whole workgroup when it fits (a descent then reads
// nothing but slots from global memory).  Returns the LDS table or nullptr; contains a workgroup barrier.
constexpr int kLogLds = 1024;
typedef const __attribute__((address_space(3))) double* LdsLog;   // typed as LDS: never merged with a global pointer
__device__ __forceinline__ LdsLog stage_log_table(const PoolDev& p, double* s_log) {
    if (p.N + 2 > kLogLds) return nullptr;
    for (int i = threadIdx.x; i < p.N + 2; i += blockDim.x) s_log[i] = p.log_table[i];
    __syncthreads();
    return (LdsLog)s_log;
}

// ---- The root level without a round trip (run_injected_kernel).  A simulation's backup and the select that follows it
// belong to one group: what the select reads of the root -- its children's slots, their priors, the root's count and
// the tree's value bounds -- is what was there BEFORE the backup, except for the one child on the path and the
// statistics the backup itself computes.  So the root's slots are requested together with the backup's inputs
// (root_prefetch: they land while the new node is being expanded), the backup hands over what it changed (RootFwd),
// and the select's first level scores from registers.  Same values, same order, same result.
template <int PP>
struct RootPre {                   // this lane's slots j * G + gl, j < PP, of the root as they were before the backup
    int c[PP], ch[PP], act[PP];
    double q[PP], pr[PP];
    int n_root;
};
struct RootFwd {                   // what the backup changed (every lane of the group holds the same values)
    bool ok;                       // false: the backup took the long-path route, the select reads the root itself
    int k0, c0, ch0;               // the root's child on the path: slot, new count, new child word (ch_set)
    bool ch_set;
    double q0;                     // its new exploitation term as stored in the slot
    int root_count;                // the root's new visit count
    double mn, mx;                 // the tree's value bounds after the backup (pools without known bounds)
};
// (slots past the root's listed children exist in the record and are never scored: no need to know n_root to ask)
template <int G, int PP>
__device__ __forceinline__ RootPre<PP> root_prefetch(const PoolDev& p, int tree, bool in, int gl) {
    RootPre<PP> r;
    r.n_root = 0;                                // the caller's, once it has arrived
    const int upto = min(p.A, PP * G);
    const Slot* slots = f_slots(p, rec(p, in ? tree : 0, 0));
#pragma unroll
    for (int j = 0; j < PP; ++j) {
        const int k = max(min(j * G + gl, upto - 1), 0);
        const Slot s = slots[k];
        r.c[j] = s.count;
        r.q[j] = s.q;
        r.ch[j] = s.child_hi;
        r.act[j] = s.act;
        r.pr[j] = p.root_prior[(size_t)(in ? tree : 0) * p.A + k];
    }
    return r;
}

struct SelectNorm {                // MinMaxStats.normalize as this select call sees it (utils.py:68-76)
    double na, nd, dflt;           // (x - na) / nd; dflt: the normalised default value of an unvisited child
    bool renorm;                   // q holds raw reward + discount*value (bounds unknown and max != min)
};
struct LevelWinner {
    int k;                         // winning slot
    int cc;                        // (its visit count << 12) | its action
    int ch;                        // its child word
};

// One node: scores slots [0, target) -- growing the prefix while its last slot ties with the best -- with PP slots per
// lane in flight, and returns the winner in every lane of the group.  Wave-wide: lanes of groups that are `done`
// come along idle.
template <int G, int PP, bool ROOT, bool FWD = false>
__device__ __forceinline__ LevelWinner scan_level(const PoolDev& p, int tree, int gl, bool done, const Slot* slots,
                                                  int np_, LdsLog lds_log, const SelectNorm& nm, int target,
                                                  const RootPre<PP>* pre = nullptr, const RootFwd* fw = nullptr) {
    const int A = p.A;
    const int gbase = (int)(threadIdx.x & 63u) & ~(G - 1);   // first lane of this group in the wavefront
    const unsigned long long gbits = G == 64 ? ~0ull : ((1ull << G) - 1ull);
    const double ninf = -std::numeric_limits<double>::infinity();
    double bs = ninf;
    int bkey = -1, bcc = 0, bch = 0;         // local best: key, (count << 12 | action), child word
    int s0nan = 0, cc0 = 0, ch0 = 0;         // slot 0 (lane 0 of the group): the winner when its score is NaN
    double fsc = 0.0;                        // score of the last slot of the prefix, in the lane that owns it
    double lt = 0.0;                         // 2*log(Np+1), config.py:159-160
    int scanned = 0;
    bool scanning = !done;
    int gkey = -1;
    for (int round = 0; round <= kMaxActions; ++round) {          // bounded: the prefix grows by >= 1 a round
        if (!__any(scanning)) break;
        if (scanning) {
            // One dependent round trip per level: every slot this lane owns is requested before anything is
            // computed (q / child of an unvisited slot are garbage and unused), and the winner's count and
            // child come back through a shuffle, not through a second load.
            const int base = scanned & ~(G - 1);                  // slot k is always scored by lane k % G
            const int upto = min(target, base + PP * G);
            int c[PP], ch[PP], act[PP];
            double q[PP], pr[PP];
            if (round == 0) {                                     // in flight with the slots
                if (lds_log != nullptr) lt = lds_log[np_];
                else lt = p.log_table[np_];
            }
            if (FWD && round == 0 && fw->ok) {
                // the root as prefetched before the backup, with the backup's one change
#pragma unroll
                for (int j = 0; j < PP; ++j) {
                    const bool hit = j * G + gl == fw->k0;
                    c[j] = hit ? fw->c0 : pre->c[j];
                    q[j] = hit ? fw->q0 : pre->q[j];
                    ch[j] = hit && fw->ch_set ? fw->ch0 : pre->ch[j];
                    act[j] = pre->act[j];
                    pr[j] = pre->pr[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < PP; ++j) {
                    const int k = min(base + j * G + gl, upto - 1);
                    const Slot s = slots[k];
                    c[j] = s.count;
                    q[j] = s.q;
                    ch[j] = s.child_hi;
                    act[j] = s.act;
                    pr[j] = ROOT ? p.root_prior[(size_t)tree * A + k] : (double)s.prior;
                }
            }
            pin(lt);
#pragma unroll
            for (int j = 0; j < PP; ++j) {
                pin(c[j]);
                pin(q[j]);
                pin(ch[j]);
                pin(pr[j]);
                pin(act[j]);
            }
#pragma unroll
            for (int j = 0; j < PP; ++j) {
                const int k = base + j * G + gl;
                if (k >= scanned && k < upto) {
                    double x = nm.dflt, ex = lt;                          // unvisited: lt / (0 + 1) == lt
                    if (c[j] > 0) {
                        x = nm.renorm ? (q[j] - nm.na) / nm.nd : q[j];    // utils.py:68-76
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
                    if (k == target - 1) fsc = sc;
                    if (sc == sc && (bkey < 0 || sc > bs || (sc == bs && key > bkey))) {
                        bs = sc;
                        bkey = key;
                        bcc = cc;
                        bch = ch[j];
                    }
                }
            }
            scanned = upto;
        }
        // arg-max over the group: the largest non-NaN score, ties to the largest (action, index) key (trap T1)
        const double gmax = group_minmax_f64<G, true>(bs);
        gkey = group_max_i32<G>((bkey >= 0 && bs == gmax) ? bkey : -1);
        if (ROOT) {
            if (scanned >= target) scanning = false;
        } else {
            // the lane that scored the last slot of the prefix says whether it is strictly below the best
            const bool tie = gl == ((target - 1) & (G - 1)) && !(fsc < gmax);
            const bool grow = ((__ballot(tie) >> gbase) & gbits) != 0;
            if (scanning && scanned >= target) {
                if (target < A && grow) target = min(A, target + G);      // a tie (or NaN): look further
                else scanning = false;
            }
        }
    }
    // Python's max() over (score, action, child) tuples never replaces a NaN-scored first element and never
    // accepts a NaN-scored later one (agent.py:184): NaN first child wins, else the non-NaN arg-max.
    const bool first_wins = ((__ballot(s0nan != 0) >> gbase) & 1ull) != 0 || gkey < 0;
    LevelWinner w;
    w.k = first_wins ? 0 : (gkey & 0xffff);
    const int owner = first_wins ? 0 : (w.k & (G - 1));
    w.cc = __shfl(first_wins ? cc0 : bcc, owner, G);
    w.ch = __shfl(first_wins ? ch0 : bch, owner, G);
    return w;
}

// lds_log: the pool's log table staged in LDS (see stage_log_table), or nullptr to read it from global memory.
template <int G, int P = 1, bool FWD = false>
__device__ void select_tree(const PoolDev& p, int tree, bool active, int gl, int* out_parent, int* out_action,
                            LdsLog lds_log = nullptr, const RootPre<P>* pre = nullptr, const RootFwd* fw = nullptr) {
    const int A = p.A;
    int e = 0, np_ = 0, target = 0, depth = 0, hi = 0;
    int* hi_home = nullptr;        // where the current node's (hi, block) word lives: its parent's slot
    int hi_word = 0;
    bool done = !active;
    bool ident = false;            // MinMaxStats.normalize returns the value unchanged (utils.py:71-72)
    SelectNorm nm;
    nm.na = p.kmin;
    nm.nd = p.kmax - p.kmin;
    if (FWD && active && fw->ok) {               // (fw->ok is the same in every lane of the wavefront)
        np_ = fw->root_count;
        target = pre->n_root;
        if (!p.has_bounds) {
            ident = (fw->mx == fw->mn);
            nm.na = fw->mn;
            nm.nd = fw->mx - fw->mn;
        }
    } else if (active) {
        np_ = p.root_count[tree];
        target = p.n_root[tree];
        if (!p.has_bounds) {
            const double mn = p.mm_min[tree], mx = p.mm_max[tree];
            ident = (mx == mn);
            nm.na = mn;
            nm.nd = mx - mn;
        }
    }
    // isnan(child.value) -> the default value, normalised like any other (agent.py:188, utils.py:68-76): the
    // exploitation term of every unvisited child
    nm.dflt = ident ? p.default_value : (p.default_value - nm.na) / nm.nd;
    nm.renorm = !p.has_bounds && !ident;
    // the loads above are waited for HERE, not inside the descent (whose waits would otherwise cover its own stores)
    pin(np_);
    pin(target);
    pin(nm.dflt);
    MZ_TRACE_DECL;
    MZ_TRACE(8);                                 // select: root statistics loaded
    for (int it = 0; it <= p.N; ++it) {          // a path holds at most N+1 nodes: every wave reaches the exit
        if (!__any(!done)) break;
        const bool root = (it == 0);             // block 0 is only ever the first node of a path
        Slot* slots = nullptr;
        if (!done) slots = f_slots(p, rec(p, tree, e));
        // the root: every listed child, P per lane; below it a prefix (see above), one slot per lane and round --
        // unless the pool keeps its nodes unsorted, whose every child is read, again P per lane
        const LevelWinner w = root           ? scan_level<G, P, true, FWD>(p, tree, gl, done, slots, np_, lds_log, nm, target, pre, fw)
                              : p.sort_nodes ? scan_level<G, 1, false>(p, tree, gl, done, slots, np_, lds_log, nm, target)
                                             : scan_level<G, P, false>(p, tree, gl, done, slots, np_, lds_log, nm, target);
        const int k = w.k, wc = w.cc >> kPathShift;
        MZ_TRACE(root ? 9 : 10);                 // select: one level
        if (!done) {
            if (gl == 0) p.path[(size_t)tree * (p.N + 1) + depth] = make_int2((e << kPathShift) | k, wc);
            if (wc > 0) {                    // visited <=> expanded: descend (agent.py:179-180)
                hi_home = &slots[k].child_hi;
                hi_word = w.ch;
                np_ = wc;
                e = w.ch & kChildMask;
                hi = (int)((unsigned)w.ch >> kChildBits);
                target = min(hi + 2, A);
                ++depth;
            } else {
                done = true;
                if (gl == 0) {
                    out_parent[tree] = e;
                    out_action[tree] = w.cc & (kMaxActions - 1);
                    p.path_len[tree] = depth + 1;
                    // this slot is visited once the backup has run: the node's prefix must cover it from now on
                    if (!root && k + 1 > hi)
                        *hi_home = (min(k + 1, kMaxActions - 1) << kChildBits) | (hi_word & kChildMask);
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
    const bool one_leaf = p.n_sum_leaf == 1;      // n <= 128: the recursion is the leaf (0, A); nothing to read
    for (int s = 0; s < p.n_sum_leaf; ++s) {
        const int lo = one_leaf ? 0 : p.sum_leaf[2 * s], n = one_leaf ? p.A : p.sum_leaf[2 * s + 1];
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
// The sum program of a wide action axis as one lane needs it, requested BEFORE the row it will sum is staged: the leaf
// this lane's octet takes in the first pass and the first merges.  Read inside numpy_sum_lds they were dependent global
// round trips in the middle of the expansion (Go shape: four leaves, three merges -- a third of the expansion's time).
constexpr int kSumProgMerges = 4;
struct SumProg {
    bool ok;                       // the program was requested (more than one leaf)
    int lo, n;                     // leaf `octet` (0, 0 beyond the last)
    int dst[kSumProgMerges], src[kSumProgMerges];
};
template <int G>
__device__ __forceinline__ SumProg sum_prog_load(const PoolDev& p, int gl) {
    SumProg sp;
    sp.ok = G >= 8 && p.n_sum_leaf > 1;
    sp.lo = sp.n = 0;
#pragma unroll
    for (int m = 0; m < kSumProgMerges; ++m) sp.dst[m] = sp.src[m] = 0;
    if (sp.ok) {
        const int s = gl >> 3;
        if (s < p.n_sum_leaf) {
            sp.lo = p.sum_leaf[2 * s];
            sp.n = p.sum_leaf[2 * s + 1];
        }
#pragma unroll
        for (int m = 0; m < kSumProgMerges; ++m)
            if (m < p.n_sum_merge) {
                sp.dst[m] = p.sum_merge[2 * m];
                sp.src[m] = p.sum_merge[2 * m + 1];
            }
    }
    return sp;
}

template <int G>
__device__ __forceinline__ float numpy_sum_lds(const PoolDev& p, const float* a, float* scratch, int gl,
                                               const SumProg* sp = nullptr) {
    // n <= 128: the recursion is the single leaf (0, A) -- known without the dependent global reads of the program
    const bool one_leaf = p.n_sum_leaf == 1;
    const bool pre = sp != nullptr && sp->ok;
    if (G >= 8) {
        constexpr int OCT = G >= 8 ? G / 8 : 1;
        const int oct = gl >> 3, i = gl & 7;
        for (int s0 = 0; s0 < p.n_sum_leaf; s0 += OCT) {
            const int s = s0 + oct;
            const bool has = s < p.n_sum_leaf;
            const bool first = pre && s0 == 0;                   // this pass's leaves came with the prefetch
            const int lo = first ? sp->lo : has && !one_leaf ? p.sum_leaf[2 * s] : 0,
                      n = first ? sp->n : !has ? 0 : one_leaf ? p.A : p.sum_leaf[2 * s + 1];
            const int nfull = n - (n % 8);
            float r = 0.f;
            if (nfull >= 8) {
                // a leaf has at most 128 elements (NumPy's block size): this lane's up to sixteen are all requested before
                // the first is added (the loop of dependent LDS reads was a sixteen-deep latency chain), then added in order
                float v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = a[lo + min(8 * j, nfull - 8) + i];
                r = v[0];
#pragma unroll
                for (int j = 1; j < 16; ++j)
                    if (8 * j < nfull) r += v[j];
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
            const int lo = one_leaf ? 0 : p.sum_leaf[2 * s], n = one_leaf ? p.A : p.sum_leaf[2 * s + 1];
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
            int dst, src;
            if (pre && m < kSumProgMerges) {
                dst = sp->dst[0];
                src = sp->src[0];
#pragma unroll
                for (int j = 1; j < kSumProgMerges; ++j)
                    if (m == j) {
                        dst = sp->dst[j];
                        src = sp->src[j];
                    }
            } else {
                dst = p.sum_merge[2 * m];
                src = p.sum_merge[2 * m + 1];
            }
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
    double mm_min, mm_max;       // the tree's MinMaxStats before this backup (pools without known bounds)
    long long sum_len, sims;     // the tree's measurement counters (lane 0 of the group adds to them)
    int slot[PB], c[PB];
    double vs[PB];
    float rw[PB];
};
using BackupPrefetch = BackupPrefetchT<1>;

// `in`: the tree exists (tree < T); an inactive tree's entries are read all the same (valid memory, unused), so that
// nothing here waits for the load that says whether the tree is active.
template <int G, int PB = 1>
__device__ __forceinline__ BackupPrefetchT<PB> backprop_prefetch(const PoolDev& p, int tree, bool in, int gl) {
    BackupPrefetchT<PB> f;
    f.len = 0;
    f.root_count = 0;
    f.root_vsum = 0.0;
    f.mm_min = f.mm_max = 0.0;
    f.sum_len = f.sims = 0;
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        f.slot[j] = 0;
        f.c[j] = 0;
        f.vs[j] = 0.0;
        f.rw[j] = 0.f;
    }
    if (in) {
        f.len = p.path_len[tree];
        int2 pe[PB];
#pragma unroll
        for (int j = 0; j < PB; ++j) pe[j] = p.path[(size_t)tree * (p.N + 1) + min(gl + G * j, p.N)];
        f.root_count = p.root_count[tree];
        f.root_vsum = p.root_vsum[tree];
        if (!p.has_bounds) {
            f.mm_min = p.mm_min[tree];
            f.mm_max = p.mm_max[tree];
        }
        if (gl == 0) {
            f.sum_len = p.sum_len[tree];
            f.sims = p.sims[tree];
        }
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
__device__ __forceinline__ double backup_slot(const PoolDev& p, char* r, int k, int c_old, double vs_old, double vin,
                                              float rw, bool leaf, int nb, int hi0, double& mn_l, double& mx_l) {
    const int c = c_old + 1;
    const double vs = (c_old > 0 ? vs_old : 0.0) + vin;
    Slot* s = f_slots(p, r) + k;
    BackSlot* b = f_back(p, r) + k;
    s->count = c;
    if (leaf) {                                   // the leaf: now expanded (agent.py:156-157)
        BackSlot nbk;
        nbk.vsum = vs;
        nbk.reward = rw;
        nbk.pad = 0;
        *b = nbk;
        s->child_hi = (hi0 << kChildBits) | nb;
    } else {
        b->vsum = vs;
    }
    const double mean = vs / (double)c;
    if (mean > mx_l) mx_l = mean;                 // MinMaxStats.update, utils.py:63-66
    if (mean < mn_l) mn_l = mean;
    // ucb_score's exploitation term of this child (agent.py:188), normalised here when the bounds are known
    const double ex = (mean == mean) ? (double)rw + p.discount * mean : p.default_value;
    const double q = p.has_bounds ? (ex - p.kmin) / (p.kmax - p.kmin) : ex;
    s->q = q;
    return q;
}

// A child slot at expansion: count 0, prior, action -- written as the WHOLE 24-byte slot (q and the child word as zeros:
// nobody reads them before the first backup through the slot writes them), so that a record's select half is
// overwritten line by line and the L2 never has to fetch a line from HBM just to merge twelve new bytes into it.
__device__ __forceinline__ void store_new_slot(Slot* s, float prior, int action) {
    Slot v;
    v.q = 0.0;
    v.count = 0;
    v.prior = prior;
    v.child_hi = 0;
    v.act = action;
    *s = v;
}

// A prior the prefix argument of select_tree can rely on: finite and >= 0.
__device__ __forceinline__ bool regular_prior(float x) { return x >= 0.f && x <= 3.4028234663852886e38f; }

// Slot (1 + rank) of action i >= 1 among actions 1..A-1 by descending (prior, action); pj(j) returns prior j.
template <typename F>
__device__ __forceinline__ int sorted_slot(int A, int i, float pi, F&& pj) {
    int rank = 0;
    for (int j = 1; j < A; ++j) {
        const float x = pj(j);
        rank += (x > pi || (x == pi && j > i)) ? 1 : 0;
    }
    return 1 + rank;
}

// The same rank over a row of priors staged in LDS (16-byte aligned, readable up to the next multiple of four past A):
// four priors per LDS read, where the generic loop pays an LDS round trip per prior.
__device__ __forceinline__ int sorted_slot_lds(int A, int i, float pi, const float* row) {
    int rank = 0;
#pragma unroll 2
    for (int j0 = 0; j0 < A; j0 += 4) {
        const float4 x4 = *reinterpret_cast<const float4*>(row + j0);
        const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u;
            rank += (j >= 1 && j < A && (xs[u] > pi || (xs[u] == pi && j > i))) ? 1 : 0;
        }
    }
    return 1 + rank;
}

// pol_row: this tree's A priors or logits (global or LDS); rnew, value_f32: its scalarised reward and value.
// lds_row: this group's TreeLds<G>::kRow floats of LDS (16-byte aligned), or nullptr (then PB must be 1, and the
// softmax of an action axis wider than the group goes through global memory and a workgroup barrier).
template <int G, bool LOGITS, int PB = 1, bool FWD = false>
__device__ void backprop_tree(const PoolDev& p, int tree, bool active, int gl, int nb, const float* pol_row,
                              float rnew, float value_f32, const BackupPrefetchT<PB>& pf, float* lds_row = nullptr,
                              RootFwd* fw = nullptr) {
    using L = TreeLds<G>;
    const int A = p.A;
    const int len = pf.len;
    double v = active ? (double)value_f32 : 0.0;
    MZ_TRACE_DECL;
    MZ_TRACE(1);                                 // backup: prefetch landed
    // ---- expand_node: the new record's slots (count 0, prior, action), sorted by prior when the pool sorts
    int hi0 = p.hi_unsorted;                                   // hi of the new node as its parent's slot will hold it
    const unsigned long long gmask = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << ((threadIdx.x & 63u) & ~(G - 1));
    if (A <= G) {
        // one child per lane, entirely in registers
        float pr = 0.f;
        if (LOGITS) {
            // masked_policy over all actions (network.py:142-146): exp, NumPy-ordered fp32 sum, IEEE divide
            const float x = (active && gl < A) ? expf(pol_row[gl]) : 0.f; // no max-subtraction (network.py:145)
            const float total = numpy_sum_lanes<G>(x, A, gl);
            pr = x / total;
        } else if (active && gl < A) {
            pr = pol_row[gl];
        }
        const bool sorted = p.sort_nodes && (__ballot(gl < A && !regular_prior(pr)) & gmask) == 0;
        int slot = gl;
        if (sorted) {
            hi0 = 1;
            const int s1 = sorted_slot(A, gl, pr, [&](int j) { return __shfl(pr, j, G); });
            if (gl >= 1) slot = s1;
        }
        if (active && gl < A) store_new_slot(f_slots(p, rec(p, tree, nb)) + slot, pr, gl);
    } else if (lds_row != nullptr && A <= L::kMaxA) {
        // priors staged in the group's LDS row (softmax terms summed there in NumPy's order)
        constexpr bool kWide = L::kMaxA > 128;               // rows that can take more than one leaf of the sum
        SumProg sp;
        sp.ok = false;
        if (LOGITS && kWide) sp = sum_prog_load<G>(p, gl);
        constexpr int kHeld = 8;                             // softmax terms a lane keeps in registers
        if (!p.sort_nodes && A <= kHeld * G) {
            // a pool that does not rank its nodes (wide action axes): nothing to check and nothing to hand to other lanes, a
            // lane's priors go from its softmax terms straight into its slots -- the terms stay in registers (the row is
            // staged for the sum only), all requested before the first exponential
            float x[kHeld];
#pragma unroll
            for (int j = 0; j < kHeld; ++j) {
                const int k = gl + G * j;
                x[j] = k < A ? pol_row[k] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < kHeld; ++j) {
                const int k = gl + G * j;
                if (k < A) {
                    x[j] = !active ? 0.f : LOGITS ? expf(x[j]) : x[j];
                    if (LOGITS) lds_row[k] = x[j];
                }
            }
            __builtin_amdgcn_wave_barrier();
            float total = 1.f;
            if (LOGITS) total = numpy_sum_lds<G>(p, lds_row, lds_row + L::kMaxA, gl, kWide ? &sp : nullptr);
            Slot* slots = f_slots(p, rec(p, active ? tree : 0, active ? nb : 0));
#pragma unroll
            for (int j = 0; j < kHeld; ++j) {
                const int k = gl + G * j;
                if (active && k < A) store_new_slot(slots + k, LOGITS ? x[j] / total : x[j], k);     // network.py:146
            }
            __builtin_amdgcn_wave_barrier();
        } else {
        for (int k = gl; k < A; k += G) lds_row[k] = !active ? 0.f : LOGITS ? expf(pol_row[k]) : pol_row[k];
        __builtin_amdgcn_wave_barrier();
        if (LOGITS) {
            const float total = numpy_sum_lds<G>(p, lds_row, lds_row + L::kMaxA, gl, kWide ? &sp : nullptr);
            for (int k = gl; k < A; k += G) lds_row[k] = lds_row[k] / total;          // network.py:146
            __builtin_amdgcn_wave_barrier();
        }
        bool bad = false;
        for (int k = gl; k < A; k += G) bad |= !regular_prior(lds_row[k]);
        const bool sorted = p.sort_nodes && (__ballot(bad) & gmask) == 0;
        if (sorted) hi0 = 1;
        Slot* slots = f_slots(p, rec(p, active ? tree : 0, active ? nb : 0));
        for (int k = gl; k < A; k += G) {
            const float pr = lds_row[k];
            int slot = k;
            if (sorted && k >= 1) slot = sorted_slot_lds(A, k, pr, lds_row);
            if (active) store_new_slot(slots + slot, pr, k);
        }
        __builtin_amdgcn_wave_barrier();
        }
    } else {
        // no LDS row, or an action axis wider than it: unsorted slots; softmax terms staged in the new record's
        // backup half (unused until the first backup through a slot) and summed through global memory
        char* r = rec(p, active ? tree : 0, active ? nb : 0);
        float* stage = reinterpret_cast<float*>(f_back(p, r));
        if (LOGITS) {
            if (active)
                for (int k = gl; k < A; k += G) stage[k] = expf(pol_row[k]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const float total = numpy_sum_f32<G>(p, stage, stage + A, active ? gl : 1 << 30);
            if (active) {
                Slot* slots = f_slots(p, r);
                for (int k = gl; k < A; k += G) {
                    slots[k].count = 0;
                    slots[k].prior = stage[k] / total;                    // network.py:146
                    slots[k].act = k;
                }
            }
        } else if (active) {
            Slot* slots = f_slots(p, r);
            for (int k = gl; k < A; k += G) {
                slots[k].count = 0;
                slots[k].prior = pol_row[k];
                slots[k].act = k;
            }
        }
    }
    MZ_TRACE(2);                                 // backup: new node written
    // Backup chain leaf -> root: the fp64 recurrence value = reward + discount*value (agent.py:195) is replayed in
    // order; the lane that owns a path position keeps the value that reached it.
    const int top = len - 1;
    double mn_l = std::numeric_limits<double>::infinity();
    double mx_l = -std::numeric_limits<double>::infinity();
    double q_pos0 = 0.0;                         // FWD: the new exploitation term of the path's first slot (lane 0 of the group)
    bool fwd_ok = true;
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
            if (active && pos < len) {
                const double qn = backup_slot(p, rec(p, tree, pf.slot[j] >> kPathShift), pf.slot[j] & (kMaxActions - 1),
                                              pf.c[j], pf.vs[j], vins[pos], (pos == top) ? rnew : pf.rw[j], pos == top, nb,
                                              hi0, mn_l, mx_l);
                if (FWD && j == 0) q_pos0 = qn;  // (position 0 is lane 0's first)
            }
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
            q_pos0 = backup_slot(p, rec(p, tree, pf.slot[0] >> kPathShift), pf.slot[0] & (kMaxActions - 1), pf.c[0],
                                 pf.vs[0], vin, rw, gl == top, nb, hi0, mn_l, mx_l);
    } else {
        fwd_ok = false;
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
            if (has) backup_slot(p, r, k, c, vs, vin, rw, pos == top, nb, hi0, mn_l, mx_l);
        }
    }
    MZ_TRACE(3);                                 // backup: chain replayed, slots updated
    if (active && gl == 0) {                          // the root (reward None -> chain ends, agent.py:195)
        const int rc = pf.root_count + 1;
        const double rvs = pf.root_vsum + v;
        p.root_count[tree] = rc;
        p.root_vsum[tree] = rvs;
        if (!p.has_bounds) {                          // MinMaxStats.update is a no-op under known bounds (utils.py:63-66)
            const double mean = rvs / (double)rc;
            if (mean > mx_l) mx_l = mean;
            if (mean < mn_l) mn_l = mean;
        }
        p.sum_len[tree] = pf.sum_len + len;           // came with the prefetch: no dependent load at the end of the backup
        p.sims[tree] = pf.sims + 1;
    }
    if (!p.has_bounds) {
        // NaN means never enter mx_l / mn_l (the comparisons above are false), so these are plain reductions
        mx_l = group_minmax_f64<G, true>(mx_l);
        mn_l = group_minmax_f64<G, false>(mn_l);
        if (active && gl == 0) {                      // nothing but this backup writes them: the prefetched values are current
            if (mx_l > pf.mm_max) p.mm_max[tree] = mx_l;
            if (mn_l < pf.mm_min) p.mm_min[tree] = mn_l;
        }
    }
    if (FWD) {
        // what the select that follows must know of this backup: lane 0 of the group owned the path's first slot
        fw->ok = fwd_ok;
        fw->k0 = __shfl(pf.slot[0] & (kMaxActions - 1), 0, G);
        fw->c0 = __shfl(pf.c[0], 0, G) + 1;
        fw->ch_set = len == 1;                        // the root's child is the new node's parent slot
        fw->ch0 = (hi0 << kChildBits) | nb;
        fw->q0 = __shfl(q_pos0, 0, G);
        fw->root_count = pf.root_count + 1;
        fw->mx = mx_l > pf.mm_max ? mx_l : pf.mm_max;
        fw->mn = mn_l < pf.mm_min ? mn_l : pf.mm_min;
    }
    MZ_TRACE(4);                                 // backup: root and min/max updated
}

// Root expansion of tree `tree` by its G lanes (lane gl) -- the body of expand_roots_kernel (mz_tree.hip, where the comment
// is) and of the TicTacToe initial inference's head kernel (mz_net.hip).  Every thread of the workgroup must call it
// (barriers inside); nothing of the workgroup may follow that needs the threads of trees beyond the pool.
template <int G, bool LOGITS, int P>
__device__ __forceinline__ void expand_roots_tree(const PoolDev& p, int tree, int gl, const float* pol, const int* legal,
                                                  const int* n_legal, const double* noise, double frac, int* out_parent,
                                                  int* out_action) {
    const int A = p.A;
    const bool in = tree < p.T;
    int n = 0;
    if (in) {
        n = n_legal ? n_legal[tree] : A;
        if (n < 0 || n > A) {
            n = 0;
            if (gl == 0) atomicOr(p.err_flag, 2);
        }
    }
    char* r = rec(p, in ? tree : 0, 0);
    // staging for the exp terms over the FULL action axis + the leaf sums: the root's backup half, unused until the
    // first backup
    float* stage = reinterpret_cast<float*>(f_back(p, r));
    float total = 1.f;
    if (LOGITS) {
        // masked_policy over the FULL action axis (network.py:143-146): exp(logit) at legal actions, 0 elsewhere
        if (in && n > 0)
            for (int k = gl; k < A; k += G) stage[k] = 0.f;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (in && n > 0) {
            for (int k = gl; k < n; k += G) {
                const int a = legal ? legal[(size_t)tree * A + k] : k;
                if (a >= 0 && a < A) stage[a] = expf(pol[(size_t)tree * A + a]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        total = numpy_sum_f32<G>(p, stage, stage + A, (in && n > 0) ? gl : 1 << 30);
    }
    if (!in) return;                       // (whole groups: a tree's lanes leave together)
    Slot* slots = f_slots(p, r);
    for (int k = gl; k < n; k += G) {
        int a = legal ? legal[(size_t)tree * A + k] : k;
        if (a < 0 || a >= A) {
            a = 0;
            atomicOr(p.err_flag, 4);
        }
        const float pk = LOGITS ? stage[k] / total : pol[(size_t)tree * A + k];   // child k <- policy[k] (trap T5)
        slots[k].count = 0;
        slots[k].prior = pk;
        slots[k].act = a;
        const double rp = (double)pk;
        p.root_prior[(size_t)tree * A + k] =
            noise ? rp * (1 - frac) + noise[(size_t)tree * A + k] * frac : rp;     // agent.py:175 (fp64, uncontracted)
    }
    if (gl == 0) {
        p.n_root[tree] = n;
        p.root_count[tree] = 0;
        p.root_vsum[tree] = 0.0;
        p.mm_min[tree] = std::numeric_limits<double>::infinity();    // utils.py:60-61
        p.mm_max[tree] = -std::numeric_limits<double>::infinity();
        p.path_len[tree] = 0;
    }
    if constexpr (P > 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        select_tree<G, P>(p, tree, n > 0, gl, out_parent, out_action, nullptr);
    }
}

}  // namespace
