// Fused inference kernels for the reference's two networks (gfx950).
//
// Replaces, in the reference (file:line relative to /root/reference):
//   Network.initial_inference / recurrent_inference     network.py:209-254
//   one_hot_tensor_encoder / binary_plane_encoder        network.py:14-40, 88-104
//   scalar_to_support_model (CartPole preprocessing)     network.py:107-116, utils.py:138-143
//   split_batch + ScalarConfig.to_scalar                 network.py:165-174, config.py:25-40, utils.py:146-148
//   TicTacToeNetwork layer stacks                        games/tictactoe.py:179-213
//   CartPoleNetwork layer stacks                         games/cartpole.py:92-132
//   the leaf.parent.hidden_state gather of run_mcts      agent.py:204
//
// Arithmetic: every 128 x 128 contraction runs on f16 MFMA with both f32 operands split in two halfs (v = h + l/2048,
// "split-f16", see below): three exact f16 products per f32 product, f32 accumulation, results within 1e-5 of the fp64
// oracle.  Only CartPole batches beyond kCart16MaxBatch take the older f32-MFMA row-tile kernel (cartpole_kernel).
//
// TicTacToe (hidden = 3x3x128):
//   job kernel (ttt_ring_kernel, 64 games x a group of board cells of nearly equal cost): the 3x3 `same` conv as chained
//     128x128 stages over the neighbour cells' hidden rows, gathered straight from the [N+1][T][H] hidden tensor by
//     parent block (no gather copy); the one-hot action planes collapse into a per-(action, cell) bias table; Dense
//     128->128; writes hidden', the reward-head partial dots and the prediction conv's partial product of its cells.
//   head (ttt_head16, 16 games): sums the partials, Dense 128->128, value/policy/reward heads; inside a search it is
//     one launch with the tree step of its games (ttt_head_step_kernel).
// CartPole (hidden = 128): one kernel per call (cartpole16_kernel, 16 games per workgroup), six chained stages (binary
//   action plane -> bias vector), softmax over the 101-bin value support and support_to_scalar in the epilogue; inside a
//   search the same launch ends with the tree step of its games.
#include "mz_common.h"
#include "mz_pool_internal.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <utility>
#include <mutex>
#include <vector>

// ---- scratch of the allocation-free entry point (mz_infer_recurrent_noalloc) ------------------------------------------
// The TicTacToe kernels hand the prediction-conv partials ([game][1152] floats, one 128-float slice per job) and the
// reward-head partial dots ([game][9] floats) from the job kernel to the head kernel through device memory.  The ordinary
// entry points keep that buffer in hipMalloc'ed memory sized to the largest batch seen (ensure_scratch); callers that must
// not allocate get it from this array instead: kNoAllocMaxBatch games, 4.76 MB, held by every GPU the library runs on
// until the process ends.  In host code `scratch` is only a handle for hipGetSymbolAddress -- the device address is
// looked up once per device (scratch_for_device) and THAT pointer is what kernels and the head launch receive.
constexpr int kNoAllocMaxBatch = 1024;
constexpr size_t kScratchPpart = (size_t)kNoAllocMaxBatch * 1152;      // floats
constexpr size_t kScratchRpart = (size_t)kNoAllocMaxBatch * 16;        // floats ([game][9], padded)
static __device__ __align__(256) float scratch[kScratchPpart + kScratchRpart];

namespace {

constexpr int kMaxDevices = 64;
float* g_scratch_ptr[kMaxDevices] = {};
std::mutex g_scratch_mutex;

// Device address of `scratch` on the CURRENT device `dev` (each GPU holds its own copy at its own address); looked up at
// set-up (mz_net_create), never in the allocation-free entry point itself.
int scratch_for_device(int dev, float** out) {
    if (dev < 0 || dev >= kMaxDevices) return mz::fail(MZ_ERR_INVALID_ARG, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        float* ptr = nullptr;
        MZ_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch)));
        if (!ptr) return mz::fail(MZ_ERR_HIP, "hipGetSymbolAddress(scratch) returned NULL");
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return MZ_OK;
}

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kThreads = 256;
constexpr int kLd = 132;           // LDS row stride in floats for K = 128 (33 16-B slots: odd -> conflict-free)

// C/D element (m-tile m, register r) of a 32x32x2 accumulator held by lane `lane`: row within the tile
__device__ __forceinline__ int acc_row(int m, int r, int lane) { return 32 * m + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// The whole [128][32-column] weight slice of one wave for one GEMM stage, in registers (64 VGPRs).  It is loaded
// one stage AHEAD (during the previous stage's MFMAs), so no MFMA ever waits on a weight load, and -- vmcnt
// being in-order -- the activation-tile gather issued after it gets a whole stage to land as well.
struct BFrag {
    float4 v[16];
};

// W packed as Wp[K/8 = 16][128][8]: lane (n = 32*wave + lane%32, kh = lane/32) takes k = 8*kb + 4*kh .. +3
__device__ __forceinline__ void load_b(BFrag& b, const float* __restrict__ Wp, int lane, int wave) {
    const float4* wp = reinterpret_cast<const float4*>(Wp) + (wave * 32 + (lane & 31)) * 2 + (lane >> 5);
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) b.v[kb] = wp[kb * 256];
}

// acc[MT] += sA[32*MT][128] (LDS, stride kLd) * W[128][this wave's 32 columns] held in b.
template <int MT>
__device__ __forceinline__ void mma_tile(const float* sA, const BFrag& b, f32x16 (&acc)[MT], int lane) {
    const float* ap = sA + (lane & 31) * kLd + 4 * (lane >> 5);
    float4 a[MT], an[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const float4*>(ap + m * 32 * kLd);
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
        if (kb + 1 < 16) {
#pragma unroll
            for (int m = 0; m < MT; ++m) an[m] = *reinterpret_cast<const float4*>(ap + m * 32 * kLd + (kb + 1) * 8);
        }
        const float4 bv = b.v[kb];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].x, bv.x, acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].y, bv.y, acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].z, bv.z, acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].w, bv.w, acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = an[m];
    }
}

template <int MT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[MT]) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
}

// out[row][n] = act(acc + bias[n]) into an LDS tile (stride kLd)
template <int MT, bool RELU>
__device__ __forceinline__ void store_acc_lds(const f32x16 (&acc)[MT], const float* __restrict__ bias, float* sOut,
                                              int lane, int wave) {
    const int n = wave * 32 + (lane & 31);
    const float bn = bias ? bias[n] : 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[m][r] + bn;
            if (RELU) v = fmaxf(v, 0.f);
            sOut[acc_row(m, r, lane) * kLd + n] = v;
        }
}

// dot of a 32-float slice of an LDS row with a 32-float slice of weights, then summed over the 4 threads of a row
__device__ __forceinline__ float dot32x4(const float* h, const float* wv) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 32; j += 4) {
        const float4 a = *reinterpret_cast<const float4*>(h + j);
        const float4 b = *reinterpret_cast<const float4*>(wv + j);
        s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    s += __shfl_xor(s, 1, 4);
    s += __shfl_xor(s, 2, 4);
    return s;
}

// ----------------------------------------------------------------------------------------------- TicTacToe
struct TttDev {
    const float* rep_conv;      // [9 taps][3][128] (Keras order)
    const float* rep_conv_b;    // [128]
    const float* rep_dense_b;
    const float* dyn_actbias;   // [9 actions][9 cells][128]: conv bias + sum of the action plane's valid taps
    const float* dyn_dense_b;
    const float* dyn_reward_w;  // [1152]
    const float* pred_conv_b;
    const float* pred_dense_p;  // split-f16 fragments (Packer::add_split16)
    const float* pred_dense_b;
    const float* pred_head_p;   // split-f16 fragments (Packer::add_split16): column 0 = value head, 1..9 = policy heads
    const float* pred_head_b;   // [16]: value bias, 9 policy biases, zeros
    const float* pred_conv_h;   // [9 cells] x split-f16 fragments for v_mfma_f32_32x32x16_f16 (Packer::add_split_f16):
    const float* rep_dense_h;   //   the initial-inference kernel's operands
    const float* dyn_conv_s;    // [9 taps] x split-f16 fragments for v_mfma_f32_16x16x32_f16 (Packer::add_split16):
    const float* dyn_dense_s;   //   the job kernel's operands (hidden channels 0..127 of the conv)
    const float* pred_conv_s;   // [9 cells]
    int job_cells[4][3];        // the cells of job j, -1 = none
    const int4* ring_desc;      // the jobs as stages of ttt_ring_kernel: (kind, cell, weight matrix, source cell)
    int ring_first[20];         // job j owns descriptors [ring_first[j], ring_first[j+1]): j = 0..3 the four cell groups of
                                // the 64-game workgroups, j = 5 + c the single cell c of the 16-game workgroups,
                                // j = 15..18 the four cell groups inside ttt_run_kernel
    float dyn_reward_b, pred_value_b;
    int* range_flag;            // raised when a kernel splits a value outside the split-f16 domain (mz_net_status)
};

// ---- split-f16 MFMA --------------------------------------------------------------------------------------------------
// An f32 value v is carried as two halfs, h = f16(v) and l = f16((v - h) * 2048): v = h + l/2048 to 22 mantissa bits.
// A product a*b is then ah*bh + (ah*bl + al*bh)/2048 (+ al*bl/2^22, dropped: 2.4e-7 relative), every term an exact
// f16 x f16 product accumulated in f32 by v_mfma_f32_32x32x16_f16.  Three of those cover 16 k's in 3 x 32 cycles; the
// f32 MFMA takes 8 x 64 for the same 16 k's.  Measured against the fp64 oracle the outputs differ by ~1e-7, inside
// the 1e-5 an f32 kernel is held to (tests/test_net_gpu.py) and no worse than f32 accumulation itself.
// Operand images: weights pre-split on the host (Packer::add_split16 / add_split_f16); activation tiles in LDS as
// [row][128 h | 128 l] halfs.
// Domain: |v| <= 65504 for every weight and activation (f16's largest finite value; the conversions saturate there),
// checked for weights by mz_net_create and for activations by the kernels (mz_net_status).
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
constexpr int kLdH = 2 * kLd;        // tile row stride in halfs
constexpr float kSplit = 2048.f, kUnsplit = 1.f / 2048.f;

// Round-toward-zero conversions (v - h is exact whatever the rounding of h; the residual is then < 1 ulp of h instead of
// half of one, still 21 bits in all) -- the packed conversion of the gather commit, so that a value splits into the same
// (h, l) wherever it is split: by the stage that produces it (split workspace) or by the stages that gather it.
__device__ __forceinline__ void split_f16(float v, _Float16& h, _Float16& l) {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
    h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v, 0.f))[0];
    l = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz((v - (float)h) * kSplit, 0.f))[0];
}
// The split image of a row of 128 values is [128 h | 128 l] halfs.  Four consecutive values of a row -> their four h and
// four l halfs, 8 bytes each (two packed conversions and packed f32 arithmetic per pair).
// `vmax` keeps the largest magnitude this lane has split (two v_max3 with |.| source modifiers): a value beyond f16's
// range saturates in the conversion, which the kernels report through range_flag (note_range) instead of hiding it.
__device__ __forceinline__ void split4(const float (&v)[4], uint2& h, uint2& l, float& vmax) {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    vmax = fmaxf(fmaxf(vmax, fabsf(v[0])), fabsf(v[1]));
    vmax = fmaxf(fmaxf(vmax, fabsf(v[2])), fabsf(v[3]));
    unsigned hw[2], lw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const f16x2 hh = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v[2 * i], v[2 * i + 1]));
        const f32x2 vv = {v[2 * i], v[2 * i + 1]}, hf = {(float)hh[0], (float)hh[1]};
        const f32x2 d = (vv - hf) * kSplit;
        hw[i] = __builtin_bit_cast(unsigned, hh);
        lw[i] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(d[0], d[1]));
    }
    h = make_uint2(hw[0], hw[1]);
    l = make_uint2(lw[0], lw[1]);
}

constexpr float kSplitMax = 65504.f;          // f16's largest finite value
// end of a kernel: raise the network's range flag if this lane split a value outside the split-f16 domain
__device__ __forceinline__ void note_range(float vmax, int* range_flag) {
    if (!(vmax <= kSplitMax)) atomicOr(range_flag, 1);
}

// acc += acx / 2048: the cross terms of a split-f16 accumulation join the h x h terms once, at its end
template <int MB>
__device__ __forceinline__ void fold4(f32x4 (&acc)[MB], const f32x4 (&acx)[MB]) {
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] += acx[m] * kUnsplit;
}

template <int MB>
__device__ __forceinline__ void zero4(f32x4 (&acc)[MB]) {
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// A load that is never served from this CU's L1 (agent scope: `sc1`; tools/micro/inv_probe2.hip: it returns what another
// CU of the XCD has stored, at the latency of an L2 hit, where a plain load returns the L1's stale line and
// `buffer_inv sc0` drops nothing).  Everything one workgroup of a row block reads of what another one wrote DURING the
// launch and may have cached before -- partials, reward partials, selections -- is read this way; hidden rows are written
// once per launch and read only afterwards, so no CU can hold an older copy of them.
template <bool COH>
__device__ __forceinline__ float4 load4(const float4* p) {
    if (!COH) return *p;
    // 16 bytes in one instruction: the `nt` load misses the L1 like the `sc1` one (same probe), and the fence path's
    // `buffer_inv sc1` covers it where the workgroups are on different XCDs
    typedef float __attribute__((ext_vector_type(4))) v4;
    const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
template <bool COH>
__device__ __forceinline__ float load1(const float* p) {
    if (!COH) return *p;
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
template <bool COH>
__device__ __forceinline__ int load1(const int* p) {
    if (!COH) return *p;
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the TicTacToe job kernel
// Recurrent inference, uniform-cost jobs.  A job = 64 games x a group of board cells chosen so that every workgroup runs
// 16-18 GEMM stages: {centre, corner 0} = 11+6, {edges 1,3} = 8+8, {edges 5,7} = 8+8, {corners 2,6,8} = 6+6+6 (a cell
// costs its valid conv taps + Dense + prediction-conv term).  4096 games -> 256 jobs, one 512-thread workgroup per CU.
// Stage s multiplies with the weights requested during stage s-1 (in-place register ring) and a tile gathered during
// stages s-2 / s-1.  A workgroup loads every weight fragment once for its 64 rows (a 32-row tile was bound by the bytes
// a CU pulls through its vector-memory path): wave v owns output columns 16v .. 16v+15
// of all 64 rows, as four 16-row blocks on v_mfma_f32_16x16x32_f16 (weights as the row operand, so a lane holds four
// consecutive columns of one game per block).  A lane's operand for k-step s is the 8 halfs k = 32s + 8 (lane/16) .. +7
// of column / game lane % 16: weights packed by Packer::add_split16, activations as [row][128 h | 128 l] images in LDS.
constexpr int kThreadsW = 512;
constexpr int kLdW = 272;          // tile row stride in halfs: 136 dwords, the stride at which the sixteen-lane groups of a
                                   // ds_read_b128 (row = lane % 16, 16-byte chunk = lane / 16) fall on 64 distinct banks
struct BFragW {
    float4 v[8];                   // [k-step][h | l] of this wave's 16 columns
};
__device__ __forceinline__ void load_bw(BFragW& b, const float* __restrict__ Ws, int lane, int wave) {
    const float4* wp = reinterpret_cast<const float4*>(Ws) + wave * 64 + lane;       // Packer::add_split16
#pragma unroll
    for (int j = 0; j < 8; ++j) b.v[j] = wp[j * 512];
}

// How a workgroup's stages follow each other (ttt_ring_kernel): every stage reads one 64 x 128 split tile; the tiles live in a ring of three: stage d reads tile d % 3, whatever wrote it
// (the gather of a neighbour cell's hidden rows for a conv tap, the last tap's epilogue for the Dense stage, the Dense
// epilogue for the prediction-conv stage).  With three tiles the gather for stage d+1 can be committed in the MIDDLE of
// stage d (after k-step 1) and made visible by a barrier there: the waves that are slowest past the previous barrier read
// tiles (d-1) % 3 or d % 3, never the one being written.  A stage whose successor is a gathered tap therefore ends by
// reading the successor's first fragments (k-step 3) and the successor starts with its MFMAs, no barrier or LDS latency
// in between; only the two stages whose tile comes from an epilogue (Dense, prediction conv) start cold behind an
// end-of-stage barrier.  acx (the cross terms) is folded into acc once per accumulation, not once per stage.
// Stage kinds of the descriptors (TttDev::ring_desc: .x kind, .y cell, .z weight matrix, .w source cell):
constexpr int kRingTapFirst = 0, kRingTap = 1, kRingTapLast = 2, kRingDense = 3, kRingPred = 4, kRingPredRemote = 5;
// More of the descriptors:
//   * a prediction-conv stage's .w = (slice << 8) | flags: kPredFresh = the partial starts with this cell (else it joins the
//     job's running sum in sAccP), kPredStore = the sum so far is this job's partial of `slice` (one 128-float slice of a
//     game's partials row; the head adds the slices in index order, so WHO computes a slice does not change a bit);
//   * kRingPredRemote (ttt_run_kernel only): the prediction-conv term of a cell whose hidden' rows ANOTHER workgroup of the
//     row block produced in this simulation -- the three-corner job hands two of its three prediction-conv stages to the
//     two-edge jobs, which would otherwise wait for it (job balance).  Its tile is gathered from the
//     rows just written (the producer raises a per-cell flag once its stores are complete, the consumer looks at the flag
//     before it requests the rows), everything else is a prediction-conv stage;
//   * a tap's .x carries in bits 8..15 (1 + the cell whose hidden' rows are complete by then and whose flag this stage
//     raises: the second tap of the producer's next cell), 0 = none.  (A first look at the flag some stages before the
//     consumer needs the answer was measured and dropped: an agent-scope load at the head of a stage's in-order queue
//     of weight loads cost 1 % of the move; the one look where the answer is needed almost always finds the flag up.)
constexpr int kPredFresh = 1, kPredStore = 2;
constexpr int kSelTagShift = 20;                     // a selection word: tag << 20 | (parent block + 1) << 8 | action & 0xff
constexpr long long kRunSpinTicks = 100000000;       // bounded waits of ttt_run_kernel: 1 s of the 100 MHz wall clock
__device__ __forceinline__ bool ring_gathered(int kind) { return kind <= kRingTapLast || kind == kRingPredRemote; }

// a gathered tile: the staged 16 bytes of every thread (four per thread: rows idx / 32, 16-byte column idx % 32) into LDS
template <bool SPLIT_IO, int MB>
__device__ __forceinline__ void ring_commit(_Float16* buf, const float4 (&stg)[MB], unsigned svalid, int tid, float& vmax) {
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int idx = tid + kThreadsW * i, r = idx >> 5, c4 = idx & 31;
        const bool ok = (svalid >> i) & 1u;
        if (SPLIT_IO) {                      // the row is a split image already: 16 bytes of it, as they are
            *reinterpret_cast<float4*>(buf + r * kLdW + c4 * 8) = ok ? stg[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            const float v[4] = {ok ? stg[i].x : 0.f, ok ? stg[i].y : 0.f, ok ? stg[i].z : 0.f, ok ? stg[i].w : 0.f};
            uint2 h, l;
            split4(v, h, l, vmax);
            *reinterpret_cast<uint2*>(buf + r * kLdW + c4 * 4) = h;
            *reinterpret_cast<uint2*>(buf + r * kLdW + 128 + c4 * 4) = l;
        }
    }
}

// The job of workgroup (row block rb, job) -- the whole of ttt_ring_kernel, and one phase of a simulation in
// ttt_run_kernel (whose pointers change meaning between two calls: no __restrict__ here).  Ends with every wave past
// its last LDS access of the job but NOT with a barrier.
// MB = 16-row blocks per workgroup: 4 (64 games x a group of cells, the shape for large batches), or 1 / 2 (16 / 32 games
// x ONE cell: nine workgroups per 16 / 32 games, for batches that would leave most of the chip idle at 64 games a
// workgroup -- every workgroup still streams every weight fragment of its stages, so they are the shapes of choice only
// while whole CUs would otherwise have nothing to do; ttt_mb).  `pslice`: which 128-float slice of a game's partials this job's sum goes to.
template <bool SPLIT_IO, int MB = 4>
__device__ __forceinline__ bool ttt_ring_body(const TttDev& w, int B, const float* in_base, const int* parent_block,
                                              long block_stride, const int* action, float* out_hidden, float* ppart,
                                              float* rpart, int rb, int job, int* trace_parent, int* trace_action,
                                              bool coh = false, unsigned long long* cell_flags = nullptr,
                                              unsigned long long flag_value = 0, int* err_flag = nullptr,
                                              const unsigned long long* sel_words = nullptr, unsigned long long sel_tag = 0,
                                              long long* waited = nullptr) {
    constexpr int ROWS = 16 * MB, TILE = ROWS * kLdW;
    // descriptor list: the four cell groups (as launches of their own), the nine cells, the four cell groups inside
    // ttt_run_kernel (cell_flags != nullptr: two prediction-conv stages change workgroups)
    const int djob = MB != 4 ? 5 + job : cell_flags != nullptr ? 15 + job : job;
    __shared__ __attribute__((aligned(16))) _Float16 sRing[3 * TILE];
    __shared__ long sRowOff[ROWS];
    __shared__ int sAct[ROWS];
    __shared__ __attribute__((aligned(16))) float sBias[11 * 128];   // conv bias + action-plane bias of the current cell,
                                                     // per action; [9*128..]: the cell's 128 reward-head weights;
                                                     // [10*128..]: the dense layer's bias
    __shared__ float sRew[8 * ROWS];                 // reward-head partials of the eight waves' column ranges
    __shared__ int sSelOk;                           // ttt_run_kernel: the tagged selections arrived (0: gave up)
    __shared__ float4 sAccP[MB * kThreadsW];          // prediction-conv partial, summed over the job's cells: slot [m][tid]
                                                     // belongs to thread tid alone (registers are the scarce resource)
    // the thread index through a copy the compiler cannot see through: inside ttt_run_kernel's simulation loop nothing
    // derived from it can then be hoisted out of the loop and kept in registers across the other phases
    int tid_opaque = threadIdx.x;
    asm volatile("" : "+v"(tid_opaque));
    const int tid = tid_opaque, lane = tid & 63, wave = tid >> 6;
    const int row0 = rb * ROWS;
    const int dbase = w.ring_first[djob], d1 = w.ring_first[djob + 1] - dbase;
    const int n16 = lane & 15;
    const float4* wlane = reinterpret_cast<const float4*>(w.dyn_conv_s) + wave * 64 + lane;

    if (tid < 128) sBias[10 * 128 + tid] = w.dyn_dense_b[tid];
    BFragW bw;
    typedef int __attribute__((ext_vector_type(4))) Desc;
    typedef const __attribute__((address_space(4))) Desc* DescPtr;
    const DescPtr desc = (DescPtr)w.ring_desc + dbase;
    const Desc D0 = desc[0];
    load_bw(bw, w.dyn_conv_s + (size_t)D0.z * 16384, lane, wave);
    if (tid < ROWS) {
        const int b = row0 + tid;
        long off = -1;
        int a = 0;
        unsigned long long word = 0;
        if (MB == 4 && sel_words != nullptr) {
            // ttt_run_kernel, from a launch's second simulation on: the selections of the row block's 64 games arrive as
            // words that carry the simulation they are for (kSelTagShift) -- the wavefront looks at them until every one
            // is this simulation's: the meeting after the tree steps and the load of the selection in one round trip
            bool ok = b >= B, gave_up = false;
            const long long t0 = wall_clock64();
            for (;;) {
                if (!ok) {
                    word = __hip_atomic_load(sel_words + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = (word >> kSelTagShift) == sel_tag;
                }
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > kRunSpinTicks) {
                    gave_up = true;
                    break;
                }
            }
            if (tid == 0) {
                sSelOk = gave_up ? 0 : 1;
                if (gave_up) atomicOr(err_flag, 8);
                *waited += wall_clock64() - t0;      // (thread 0's: not part of the job's own time, mz_debug_run_info)
            }
        }
        if (b < B) {
            // coh: the selection was stored by another workgroup of this launch (ttt_run_kernel)
            int pb;
            if (MB == 4 && sel_words != nullptr) {
                pb = (int)((word >> 8) & 0xfffu) - 1;
                a = (int)(signed char)(word & 0xffu);
            } else {
                pb = !parent_block ? 0 : coh ? load1<true>(parent_block + b) : parent_block[b];
                a = coh ? load1<true>(action + b) : action[b];
            }
            if (pb >= 0) off = (long)pb * block_stride + (long)b * 1152;
            if (trace_parent != nullptr && job == 0) {      // this simulation's selection as the search made it
                trace_parent[b] = pb;
                trace_action[b] = a;
            }
            if (a < 0 || a > 8) a = 0;
        }
        sRowOff[tid] = off;
        sAct[tid] = a;
    }
    __syncthreads();
    if (MB == 4 && sel_words != nullptr && sSelOk == 0) return false;         // (the whole workgroup)
    const float* sptr[MB];
    unsigned svalid = 0;
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int idx = tid + kThreadsW * i, r = idx >> 5, c4 = idx & 31;
        const long off = sRowOff[r];
        sptr[i] = in_base + (off >= 0 ? off : 0) + c4 * 4;
        if (off >= 0) svalid |= 1u << i;
    }
    float4 stg[MB];
    float vmax = 0.f;                                // largest magnitude split by this lane (note_range)
#pragma unroll
    for (int i = 0; i < MB; ++i) stg[i] = *reinterpret_cast<const float4*>(sptr[i] + D0.w * 128);
    ring_commit<SPLIT_IO, MB>(sRing, stg, svalid, tid, vmax);
    {   // stage 1's tile (a tap: every cell has at least four) is on its way while stage 0 starts
        const Desc D1 = desc[1];
#pragma unroll
        for (int i = 0; i < MB; ++i) stg[i] = *reinterpret_cast<const float4*>(sptr[i] + D1.w * 128);
    }
    __syncthreads();

    // rows with an output (game < B and a parent): one bit per row; this lane's rows are n16 + 16 m, m = 0..3
    const unsigned long long rowmask = __ballot(lane < ROWS && sRowOff[lane < ROWS ? lane : 0] >= 0);
    int lact[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) lact[m] = sAct[16 * m + n16];
    f32x4 acc[MB], acx[MB];
    float4 af[2][MB][2];                              // activation fragments: ring over k-steps, carried across stages
    int cur = 0, nxt = 1;                            // ring positions of this stage's and the next stage's tile
    Desc Dc = D0, Dn = desc[1], Dn2 = desc[2];
    MZ_TRACE_DECL;
    MZ_TRACE_NW(0);                                  // prologue

#pragma unroll 1
    for (int d = 0; d < d1; ++d) {
        // Everything below derives its lane geometry from a copy of the thread index the compiler cannot see through:
        // hoisted out of the loop, the dozens of per-lane addresses and masks of the epilogues would not fit the 256
        // registers a wavefront has here (they spilled); recomputed where they are used they cost a few VALU ops.
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6, n16 = lane & 15, g = lane >> 4, lcol = wave * 16 + 4 * g;
        const int kind = Dc.x & 0xff, cell = Dc.y;
        const bool has_next = d + 1 < d1;
        // the descriptor three stages ahead: requested under the LAST k-step's MFMAs (below).  Requested here it was in
        // flight at the stage's first LDS wait, and a wait with scalar loads outstanding can only be for everything
        // (lgkmcnt counts both, scalar loads return out of order): the first MFMAs then waited for the fragment reads of the
        // SECOND k-step, issued just before them, instead of for their own only.  (Taken from a per-lane table by
        // v_readlane -- no memory operation at all -- the stage loop was slower: 92.9 against 94.2 M simulations/s.)
        Desc Dn3;
        // next stage's weights (a redundant reload of this stage's on the last stage keeps the loop branch-free)
        const float4* wnext = wlane + (size_t)(has_next ? Dn.z : Dc.z) * 4096;
        // af[0] holds this stage's first fragments already: requested under the previous stage's last MFMAs (a gathered
        // tile) or, in the 16-game jobs, right behind the barrier that ended its epilogue (a tile written by that epilogue:
        // +0.5 % at 512 games; in the 64-game jobs the 32 registers held across the epilogue spill: 87.3 vs 90.7 M)
        constexpr bool kPrimeCold = MB == 1;
        const bool primed = d > 0 && (kPrimeCold || ring_gathered(kind));
        const bool commit = has_next && ring_gathered(Dn.x & 0xff);     // the next stage's tile is gathered: it is committed
                                                                 // after k-step 1, a barrier follows, k-step 3 reads its
                                                                 // first fragments
        const bool load2 = d + 2 < d1 && ring_gathered(Dn2.x & 0xff);   // request the tile of the stage after the next
        const bool remote2 = (Dn2.x & 0xff) == kRingPredRemote;  // ... from this simulation's hidden' rows of cell Dn2.y
        const int src2 = remote2 ? Dn2.y * 128 : Dn2.w * 128;
        if (MB == 4 && remote2) {
            // the remote prediction-conv stage is a job's last gathered stage: from here on the gather pointers are the
            // rows row0 + r of THIS simulation's block, as their producer stored them (outside the MFMA section, which
            // stays the scheduling region it was)
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const int idx = tid + kThreadsW * i, r = idx >> 5, c4 = idx & 31;
                sptr[i] = out_hidden + (size_t)(row0 + r) * 1152 + c4 * 4;
            }
        }
        float bt[3];
        if (kind == kRingTapFirst) {
            // first tap of a cell: its bias table for all nine actions and its reward weights, requested ahead of this
            // stage's weight loads and parked in LDS at the end of the stage (vmcnt is in-order)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int idx = tid + kThreadsW * i;
                bt[i] = idx < 9 * 128    ? w.dyn_actbias[((idx >> 7) * 9 + cell) * 128 + (idx & 127)]
                        : idx < 10 * 128 ? w.dyn_reward_w[cell * 128 + (idx & 127)]
                                         : 0.f;
            }
        }
        if (kind == kRingTapFirst || kind >= kRingDense) {       // a new accumulation
            asm volatile("");
            zero4(acc);
            zero4(acx);
        }
        MZ_TRACE_NW(1);                              // stage set-up
        {   // ---- the stage's MFMAs: 4 k-steps x 4 row blocks x 3
            const _Float16* ap = sRing + cur * TILE + n16 * kLdW + 8 * g;
            const _Float16* np = sRing + nxt * TILE + n16 * kLdW + 8 * g;
            if (!primed) {
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    af[0][m][0] = *reinterpret_cast<const float4*>(ap + m * 16 * kLdW);
                    af[0][m][1] = *reinterpret_cast<const float4*>(ap + m * 16 * kLdW + 128);
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s + 1 < 4) {
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        af[(s + 1) & 1][m][0] = *reinterpret_cast<const float4*>(ap + m * 16 * kLdW + 32 * (s + 1));
                        af[(s + 1) & 1][m][1] = *reinterpret_cast<const float4*>(ap + m * 16 * kLdW + 128 + 32 * (s + 1));
                    }
                } else if (commit) {                 // the next stage's first fragments (its tile is behind the barrier)
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        af[0][m][0] = *reinterpret_cast<const float4*>(np + m * 16 * kLdW);
                        af[0][m][1] = *reinterpret_cast<const float4*>(np + m * 16 * kLdW + 128);
                    }
                }
                if (s == 3) Dn3 = desc[d + 3 < d1 ? d + 3 : 0];
                const f16x8 bh = __builtin_bit_cast(f16x8, bw.v[2 * s]), bl = __builtin_bit_cast(f16x8, bw.v[2 * s + 1]);
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const f16x8 ah = __builtin_bit_cast(f16x8, af[s & 1][m][0]), al = __builtin_bit_cast(f16x8, af[s & 1][m][1]);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah, acc[m], 0, 0, 0);
                    acx[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah, acx[m], 0, 0, 0);
                    acx[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al, acx[m], 0, 0, 0);
                }
                bw.v[2 * s] = wnext[(2 * s) * 512];  // in-place ring: the next stage's weight fragments
                bw.v[2 * s + 1] = wnext[(2 * s + 1) * 512];
                // pin the interleave: the next k-step's eight fragment reads, this k-step's twelve MFMAs, its VMEM reads
                if (s + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 2 * MB, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * MB, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                if (s == 1 && commit) {              // the tile requested during the previous stage, then the barrier
                    ring_commit<SPLIT_IO, MB>(sRing + nxt * TILE, stg, svalid, tid, vmax);
                    __syncthreads();
                    // A cell's SECOND tap may carry the cell whose hidden' rows this workgroup stored two stages ago (in
                    // the epilogue of that cell's Dense stage).  The weight fragments of this stage were requested during
                    // the stage in between, i.e. AFTER those stores; every wave that is past this barrier has waited for
                    // them, and vector memory operations retire in order (one vmcnt for loads and stores on this part):
                    // the rows are in the XCD's L2 -- raise the cell's flag.
                    if (MB == 4 && cell_flags != nullptr && ((Dc.x >> 8) & 0xff) != 0 && tid == 0)
                        __hip_atomic_fetch_max(cell_flags + (((Dc.x >> 8) & 0xff) - 1 == 2 ? 0 : 1), flag_value,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (s == 2 && load2) {               // the tile two stages ahead
#pragma unroll
                    for (int i = 0; i < MB; ++i) stg[i] = *reinterpret_cast<const float4*>(sptr[i] + src2);
                }
            }
        }
        MZ_TRACE_NW(2);                              // MFMA section
        __builtin_amdgcn_sched_barrier(0);
        if (kind == kRingTapFirst) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int idx = tid + kThreadsW * i;
                if (idx < 10 * 128) sBias[idx] = bt[i];
            }
        } else if (kind == kRingTapLast) {   // last tap: + bias, relu -> the Dense stage's tile (games/tictactoe.py:189)
            fold4(acc, acx);
            _Float16* sX = sRing + nxt * TILE;
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const float4 bq = *reinterpret_cast<const float4*>(sBias + lact[m] * 128 + lcol);
                const float v[4] = {fmaxf(acc[m][0] + bq.x, 0.f), fmaxf(acc[m][1] + bq.y, 0.f),
                                    fmaxf(acc[m][2] + bq.z, 0.f), fmaxf(acc[m][3] + bq.w, 0.f)};
                uint2 h, l;
                split4(v, h, l, vmax);
                _Float16* row = sX + (16 * m + n16) * kLdW + lcol;
                *reinterpret_cast<uint2*>(row) = h;
                *reinterpret_cast<uint2*>(row + 128) = l;
            }
            MZ_TRACE_NW(3);                          // epilogue
            if (MB == 4 && cell_flags != nullptr && d + 3 < d1 && (Dn3.x & 0xff) == kRingPredRemote && tid == 0) {
                // the rows of the remote prediction-conv stage are requested during the next stage: their producer must
                // have raised the cell's flag (it did so a dozen stages ago unless something is badly out of step)
                const unsigned long long* flag = cell_flags + (Dn3.y == 2 ? 0 : 1);
                const long long t0 = wall_clock64();
                for (;;) {
                    const unsigned long long flag_seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((long long)(flag_seen - flag_value) >= 0) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (wall_clock64() - t0 > kRunSpinTicks) {
                        atomicOr(err_flag, 8);
                        break;
                    }
                }
            }
            __syncthreads();
            MZ_TRACE_NW(4);                          // barrier
            if (kPrimeCold) {   // the Dense stage's first fragments, requested before the loop's bookkeeping and set-up
                const _Float16* np = sRing + nxt * TILE + n16 * kLdW + 8 * g;
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    af[0][m][0] = *reinterpret_cast<const float4*>(np + m * 16 * kLdW);
                    af[0][m][1] = *reinterpret_cast<const float4*>(np + m * 16 * kLdW + 128);
                }
            }
        } else if (kind == kRingDense) {     // Dense 128 -> 128 relu = hidden'[cell] (games/tictactoe.py:191)
            fold4(acc, acx);
            float* const ocell = out_hidden + (size_t)row0 * 1152 + cell * 128;       // uniform
            // The Dense output tile is the next stage's (the cell's prediction-conv term) -- unless that term is computed
            // elsewhere and the next stage is the first tap of another cell, whose gathered tile already sits in `nxt`:
            // then the rows are staged for the copy-out in THIS stage's tile, once every wave is done reading it.
            if (commit) __syncthreads();
            _Float16* sH = sRing + (commit ? cur : nxt) * TILE;
            {
                const float4 bq = *reinterpret_cast<const float4*>(sBias + 10 * 128 + lcol);
                const float4 rw = *reinterpret_cast<const float4*>(sBias + 9 * 128 + lcol);
                float rdot[MB];
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const float v[4] = {fmaxf(acc[m][0] + bq.x, 0.f), fmaxf(acc[m][1] + bq.y, 0.f),
                                        fmaxf(acc[m][2] + bq.z, 0.f), fmaxf(acc[m][3] + bq.w, 0.f)};
                    // reward head (games/tictactoe.py:193-195): this lane's four columns
                    rdot[m] = __builtin_fmaf(v[3], rw.w, __builtin_fmaf(v[2], rw.z, __builtin_fmaf(v[1], rw.y, v[0] * rw.x)));
                    uint2 h, l;
                    split4(v, h, l, vmax);
                    _Float16* row = sH + (16 * m + n16) * kLdW + lcol;
                    *reinterpret_cast<uint2*>(row) = h;
                    *reinterpret_cast<uint2*>(row + 128) = l;
                    if (!SPLIT_IO && ((rowmask >> (16 * m + n16)) & 1ull))                      // f32, unsplit
                        *reinterpret_cast<float4*>(ocell + (unsigned)(16 * m + n16) * 1152u + lcol) =
                            make_float4(v[0], v[1], v[2], v[3]);
                }
                // ... then the wave's sixteen columns: lanes n16 + 16 g, g = 0..3, hold the four parts of game n16
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const unsigned x = __float_as_uint(rdot[m]);
                    const auto r16 = __builtin_amdgcn_permlane16_swap(x, x, false, false);
                    const unsigned y = __float_as_uint(__uint_as_float(r16[0]) + __uint_as_float(r16[1]));
                    const auto r32 = __builtin_amdgcn_permlane32_swap(y, y, false, false);
                    if (g == 0) sRew[wave * ROWS + 16 * m + n16] = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
                }
            }
            MZ_TRACE_NW(3);
            __syncthreads();
            MZ_TRACE_NW(4);
            if (kPrimeCold && !commit) {   // the prediction-conv stage's first fragments: requested ahead of the copy-out
                                           // below, whose LDS reads and stores then issue inside their latency
                const _Float16* np = sH + n16 * kLdW + 8 * g;
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    af[0][m][0] = *reinterpret_cast<const float4*>(np + m * 16 * kLdW);
                    af[0][m][1] = *reinterpret_cast<const float4*>(np + m * 16 * kLdW + 128);
                }
            }
            if (SPLIT_IO) {                        // the tile is the row image the workspace wants: 16 bytes a thread
#pragma unroll
                for (int i = 0; i < MB; ++i) {
                    const int idx = tid + kThreadsW * i, r = idx >> 5, c4 = idx & 31;
                    if ((svalid >> i) & 1u)
                        __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(sH + r * kLdW + c4 * 8),
                                                    reinterpret_cast<f32x4*>(ocell + (unsigned)r * 1152u + c4 * 4));
                }
            }
            if (tid < ROWS) {
                if ((rowmask >> tid) & 1ull) {
                    float rs = 0.f;
#pragma unroll
                    for (int v8 = 0; v8 < 8; ++v8) rs += sRew[v8 * ROWS + tid];
                    rpart[(size_t)(row0 + tid) * 9 + cell] = rs;
                }
            }
            MZ_TRACE_NW(5);                          // after the barrier
        } else if (kind >= kRingPred) {      // this cell's term of the prediction conv (games/tictactoe.py:201)
            fold4(acc, acx);
            const int pflags = Dc.w & 0xff;
            if (!(pflags & kPredFresh)) {            // joins the job's running sum
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const float4 q = sAccP[m * kThreadsW + tid];
                    acc[m] += f32x4{q.x, q.y, q.z, q.w};
                }
            }
            if (!(pflags & kPredStore)) {
#pragma unroll
                for (int m = 0; m < MB; ++m) sAccP[m * kThreadsW + tid] = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
            } else {              // one 128-float slice of the games' partials rows is complete
                float* const pjob = ppart + (size_t)row0 * 1152 + (Dc.w >> 8) * 128;
#pragma unroll
                for (int m = 0; m < MB; ++m)
                    if ((rowmask >> (16 * m + n16)) & 1ull)
                        *reinterpret_cast<f32x4*>(pjob + (unsigned)(16 * m + n16) * 1152u + lcol) = acc[m];
            }
            MZ_TRACE_NW(3);
        }
        cur = nxt;
        nxt = nxt == 2 ? 0 : nxt + 1;
        Dc = Dn;
        Dn = Dn2;
        Dn2 = Dn3;
    }
    note_range(vmax, w.range_flag);
    return true;
}

// blockIdx -> (job, row block): the jobs of a row block (MB = 4: four cell groups x 64 games; MB = 1: nine cells x 16
// games) run on one XCD (blockIdx % 8: workgroups are dealt to the XCDs round-robin) and share its L2 -- they gather the
// same parent rows, and in ttt_run_kernel they hand their results to each other through it.  The cells of the 16-game
// jobs go out heaviest first (centre, edges, corners: 11 / 8 / 6 stages).
template <int MB = 4>
__device__ __forceinline__ void ttt_job_of_block(int& job, int& rb) {
    const int xl = blockIdx.x & 7, bi = blockIdx.x >> 3;
    if (MB == 4) {
        job = bi & 3;
        rb = (bi >> 2) * 8 + xl;
    } else {
        const int order = bi % 9;                    // 4 | 1 3 5 7 | 0 2 6 8
        job = order == 0 ? 4 : order <= 4 ? 2 * order - 1 : order == 5 ? 0 : order == 6 ? 2 : order == 7 ? 6 : 8;
        rb = (bi / 9) * 8 + xl;
    }
}
// workgroups of the job kernel for a batch
static int ttt_job_grid(int batch, int mb) {
    const int nrb = (batch + 16 * mb - 1) / (16 * mb);
    return 8 * (mb == 4 ? 4 : 9) * ((nrb + 7) / 8);
}

template <bool SPLIT_IO, int MB>
__global__ __launch_bounds__(kThreadsW, MB == 4 ? 1 : 2) void ttt_ring_kernel(TttDev w, int B, const float* __restrict__ in_base,
                                                                const int* __restrict__ parent_block, long block_stride,
                                                                const int* __restrict__ action,
                                                                float* __restrict__ out_hidden, float* __restrict__ ppart,
                                                                float* __restrict__ rpart) {
    int job, rb;
    ttt_job_of_block<MB>(job, rb);
    if (rb >= (B + 16 * MB - 1) / (16 * MB)) return;  // grid padding (uniform per workgroup)
    ttt_ring_body<SPLIT_IO, MB>(w, B, in_base, parent_block, block_stride, action, out_hidden, ppart, rpart, rb, job, nullptr,
                                nullptr);
}

// One 128 x 128 GEMM stage on split operands without the job kernel's pipelining: acc += sA[32][128] * W, weights in b
// (split fragments), product transposed like mma_block_h.  Returns with the cross terms folded in.
__device__ __forceinline__ void mma_tile_h_plain(const _Float16* sA, const BFrag& b, f32x16& acc, int lane) {
    const _Float16* ap = sA + (lane & 31) * kLdH + 8 * (lane >> 5);
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    float4 ah = *reinterpret_cast<const float4*>(ap), al = *reinterpret_cast<const float4*>(ap + 128);
#pragma unroll
    for (int T = 0; T < 8; ++T) {
        float4 nh = ah, nl = al;
        if (T + 1 < 8) {
            nh = *reinterpret_cast<const float4*>(ap + (T + 1) * 16);
            nl = *reinterpret_cast<const float4*>(ap + 128 + (T + 1) * 16);
        }
        const f16x8 xh = __builtin_bit_cast(f16x8, ah), xl = __builtin_bit_cast(f16x8, al);
        const f16x8 bh = __builtin_bit_cast(f16x8, b.v[2 * T]), bl = __builtin_bit_cast(f16x8, b.v[2 * T + 1]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, xh, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, xh, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, xl, acc2, 0, 0, 0);
        ah = nh;
        al = nl;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += acc2[r] * kUnsplit;
}

// Initial inference, split-f16: the 32-row job kernel's grid and jobs (32 games x a group of cells, TttDev::job_cells).  Per cell: representation Conv2D 3x3 same 3 -> 128
// relu on the VALU (K = 27, games/tictactoe.py:182), Dense 128 -> 128 relu = hidden[cell] (:184), and the cell's term
// of the prediction conv (:201) accumulated over the job's cells: one partial per (game, job), so the head is the
// recurrent one's.  SPLIT_IO: hidden leaves as split images (the workspace of a native move), else as f32.
template <bool SPLIT_IO>
__global__ __launch_bounds__(kThreads, 2) void ttt_initial_kernel_h(TttDev w, int B, const float* __restrict__ observation,
                                                                    float* __restrict__ out_hidden,
                                                                    float* __restrict__ ppart) {
    constexpr int ROWS = 32;
    __shared__ __attribute__((aligned(16))) float sXF[ROWS * kLd];
    __shared__ __attribute__((aligned(16))) float sHF[ROWS * kLd];
    __shared__ float sObs[ROWS * 27];
    __shared__ __attribute__((aligned(16))) float sConvW[27 * 128];      // the 3 -> 128 conv's weights, all nine taps
    _Float16* sX = reinterpret_cast<_Float16*>(sXF);
    _Float16* sH = reinterpret_cast<_Float16*>(sHF);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nrb = (B + ROWS - 1) / ROWS;
    const int xl = blockIdx.x & 7, bi = blockIdx.x >> 3;
    const int job = ((bi & 3) + ((bi >> 5) & 1) * 2) & 3, rb = (bi >> 2) * 8 + xl;
    if (rb >= nrb) return;                           // grid padding (uniform per workgroup)
    const int row0 = rb * ROWS;
    float vmax = 0.f;
    for (int i = tid; i < ROWS * 27; i += kThreads) {
        const int b = row0 + i / 27;
        sObs[i] = b < B ? observation[(size_t)b * 27 + i % 27] : 0.f;
    }
    for (int i = tid; i < 27 * 32; i += kThreads)
        reinterpret_cast<float4*>(sConvW)[i] = reinterpret_cast<const float4*>(w.rep_conv)[i];
    BFrag bd;
    load_b(bd, w.rep_dense_h, lane, wave);           // the same Dense for every cell
    const int lrow = lane & 31, lcol = wave * 32 + 4 * (lane >> 5);     // transposed accumulators, see mma_block_h
    const bool lrow_ok = row0 + lrow < B;
    const int c4 = tid & 31, r0 = tid >> 5;          // conv: columns 4 c4 .. +3 of rows r0 + 8 i
    const float4 cb = *reinterpret_cast<const float4*>(w.rep_conv_b + 4 * c4);
    float4 db[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) db[q] = *reinterpret_cast<const float4*>(w.rep_dense_b + lcol + 8 * q);
    f32x16 accp;
#pragma unroll
    for (int r = 0; r < 16; ++r) accp[r] = 0.f;
    __syncthreads();
#pragma unroll 1
    for (int ci3 = 0; ci3 < 3; ++ci3) {
        const int cell = w.job_cells[job][ci3];      // kernel argument: a scalar load
        if (cell < 0) break;
        BFrag bp;
        load_b(bp, w.pred_conv_h + (size_t)cell * 16384, lane, wave);
        const int ci = cell / 3, cj = cell % 3;
        float4 s[4] = {cb, cb, cb, cb};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ni = ci + t / 3 - 1, nj = cj + t % 3 - 1;
            if (ni < 0 || ni >= 3 || nj < 0 || nj >= 3) continue;
            const float4* wt = reinterpret_cast<const float4*>(sConvW + t * 3 * 128) + c4;
            const float4 w0 = wt[0], w1 = wt[32], w2 = wt[64];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* o = sObs + (r0 + 8 * i) * 27 + (ni * 3 + nj) * 3;
                const float o0 = o[0], o1 = o[1], o2 = o[2];
                s[i].x = __builtin_fmaf(o2, w2.x, __builtin_fmaf(o1, w1.x, __builtin_fmaf(o0, w0.x, s[i].x)));
                s[i].y = __builtin_fmaf(o2, w2.y, __builtin_fmaf(o1, w1.y, __builtin_fmaf(o0, w0.y, s[i].y)));
                s[i].z = __builtin_fmaf(o2, w2.z, __builtin_fmaf(o1, w1.z, __builtin_fmaf(o0, w0.z, s[i].z)));
                s[i].w = __builtin_fmaf(o2, w2.w, __builtin_fmaf(o1, w1.w, __builtin_fmaf(o0, w0.w, s[i].w)));
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x[4] = {fmaxf(s[i].x, 0.f), fmaxf(s[i].y, 0.f), fmaxf(s[i].z, 0.f), fmaxf(s[i].w, 0.f)};
            uint2 h, l;
            split4(x, h, l, vmax);
            _Float16* row = sX + (r0 + 8 * i) * kLdH + 4 * c4;
            *reinterpret_cast<uint2*>(row) = h;
            *reinterpret_cast<uint2*>(row + 128) = l;
        }
        __syncthreads();                             // sX complete; the previous cell's readers of sH are done
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        mma_tile_h_plain(sX, bd, acc, lane);
        float* const ocell = out_hidden + (size_t)row0 * 1152 + cell * 128;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float x[4] = {fmaxf(acc[4 * q] + db[q].x, 0.f), fmaxf(acc[4 * q + 1] + db[q].y, 0.f),
                                fmaxf(acc[4 * q + 2] + db[q].z, 0.f), fmaxf(acc[4 * q + 3] + db[q].w, 0.f)};
            uint2 h, l;
            split4(x, h, l, vmax);
            *reinterpret_cast<uint2*>(sH + lrow * kLdH + lcol + 8 * q) = h;
            *reinterpret_cast<uint2*>(sH + lrow * kLdH + 128 + lcol + 8 * q) = l;
            if (!SPLIT_IO && lrow_ok)
                *reinterpret_cast<float4*>(ocell + (unsigned)lrow * 1152u + lcol + 8 * q) = make_float4(x[0], x[1], x[2], x[3]);
        }
        __syncthreads();
        if (SPLIT_IO) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = r0 + 8 * i;
                if (row0 + r < B)
                    *reinterpret_cast<float4*>(ocell + (unsigned)r * 1152u + c4 * 4) =
                        *reinterpret_cast<const float4*>(sH + r * kLdH + c4 * 8);
            }
        }
        mma_tile_h_plain(sH, bp, accp, lane);
    }
    if (lrow_ok) {
        float* const pjob = ppart + (size_t)row0 * 1152 + job * 128;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(pjob + (unsigned)lrow * 1152u + lcol + 8 * q) =
                make_float4(accp[4 * q], accp[4 * q + 1], accp[4 * q + 2], accp[4 * q + 3]);
    }
    note_range(vmax, w.range_flag);
}

// One 64 x 128 split tile times one 128 x 128 split matrix on the job kernel's geometry: wave v owns columns 16 v .. + 15 of
// all 64 rows as four 16-row blocks (v_mfma_f32_16x16x32_f16, weights as the row operand), h x h terms into acc, the
// cross terms into acx (fold4 at the end of the accumulation).  `next`: the matrix that replaces b in place, fragment by
// fragment as the k-steps consume it (nullptr: b stays).
template <bool NEXT = false>
__device__ __forceinline__ void mma_tile_w(const _Float16* tile, BFragW& b, f32x4 (&acc)[4], f32x4 (&acx)[4], int n16, int g,
                                           const float4* next = nullptr) {
    const _Float16* ap = tile + n16 * kLdW + 8 * g;
    float4 af[2][4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        af[0][m][0] = *reinterpret_cast<const float4*>(ap + m * 16 * kLdW);
        af[0][m][1] = *reinterpret_cast<const float4*>(ap + m * 16 * kLdW + 128);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (s + 1 < 4) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                af[(s + 1) & 1][m][0] = *reinterpret_cast<const float4*>(ap + m * 16 * kLdW + 32 * (s + 1));
                af[(s + 1) & 1][m][1] = *reinterpret_cast<const float4*>(ap + m * 16 * kLdW + 128 + 32 * (s + 1));
            }
        }
        const f16x8 bh = __builtin_bit_cast(f16x8, b.v[2 * s]), bl = __builtin_bit_cast(f16x8, b.v[2 * s + 1]);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f16x8 ah = __builtin_bit_cast(f16x8, af[s & 1][m][0]), al = __builtin_bit_cast(f16x8, af[s & 1][m][1]);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah, acc[m], 0, 0, 0);
            acx[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah, acx[m], 0, 0, 0);
            acx[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al, acx[m], 0, 0, 0);
        }
        if (NEXT) {
            b.v[2 * s] = next[(2 * s) * 512];
            b.v[2 * s + 1] = next[(2 * s + 1) * 512];
        }
        // pin the interleave (left alone the scheduler sinks every fragment read to its first use and each group of
        // MFMAs waits out an LDS latency): the next k-step's eight fragment reads, this k-step's twelve MFMAs, the loads
        if (s + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        if (NEXT) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
    }
}

// Initial inference for the batches of the 64-game jobs: 64 games x a group of cells per 512-thread workgroup, the job
// kernel's grid (ttt_job_of_block), tile layout and MFMA geometry.  The 32-game kernel above takes 21 us for 4096 games
// -- the time of two thirds of a simulation for a quarter of its arithmetic: a K = 27 conv on the VALU with 135 LDS
// reads a thread and cell, and every MFMA stage cold behind a barrier.  Here
//   * the conv of cell c is ONE split-f16 k-step over the game's observation row [cell'][plane] (27 values, padded to 32)
//     with a per-cell 32 x 128 matrix that holds the tap reaching cell' from c and zeros elsewhere (rep_conv_c, packed by
//     mz_net_create): twelve MFMAs a wavefront;
//   * the cells are software-pipelined over two conv tiles and two hidden tiles: between two barriers a workgroup runs the
//     prediction-conv term of cell j-1, the Dense of cell j, the conv of cell j+1 and the Dense epilogue of cell j, then
//     copies cell j's hidden rows out under the next section -- one barrier a cell, 96 MFMAs a wavefront between them;
//   * a workgroup loads every weight fragment once for its 64 games (the prediction-conv fragments in place, under the
//     MFMAs that consume their predecessors).
// Same layers per cell as above (games/tictactoe.py:182-184, :201); the partial of a job is accumulated over its cells in
// the MFMA accumulators and folded once.  rep_dense_s: the representation Dense as Packer::add_split16 fragments.
template <bool SPLIT_IO>
__global__ __launch_bounds__(kThreadsW, 1) void ttt_initial_kernel_w(TttDev w, const float* __restrict__ rep_dense_s,
                                                                     const float* __restrict__ rep_conv_c, int B,
                                                                     const float* __restrict__ observation,
                                                                     float* __restrict__ out_hidden,
                                                                     float* __restrict__ ppart) {
    constexpr int ROWS = 64, TILE = ROWS * kLdW;
    constexpr int kLdO = 72;       // observation row stride in halfs: [32 h | 32 l] + 8
    __shared__ __attribute__((aligned(16))) _Float16 sX[2 * TILE];
    __shared__ __attribute__((aligned(16))) _Float16 sH[2 * TILE];
    __shared__ __attribute__((aligned(16))) _Float16 sObs[ROWS * kLdO];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n16 = lane & 15, g = lane >> 4, lcol = wave * 16 + 4 * g;
    int job, rb;
    ttt_job_of_block<4>(job, rb);
    if (rb >= (B + ROWS - 1) / ROWS) return;         // grid padding (uniform per workgroup)
    const int row0 = rb * ROWS;
    const int ncell = w.job_cells[job][2] >= 0 ? 3 : 2;
    float vmax = 0.f;
    float obs[4];                                    // 64 rows x 32 k's: the split image of the observation rows
#pragma unroll
    for (int i = 0; i < 4; ++i) {                    // (unconditional loads from clamped addresses, all four requested
        const int idx = tid + kThreadsW * i, r = idx >> 5, k = idx & 31;      // before the first is used)
        obs[i] = observation[(size_t)min(row0 + r, B - 1) * 27 + min(k, 26)];
    }
    asm volatile("" : "+v"(obs[0]), "+v"(obs[1]), "+v"(obs[2]), "+v"(obs[3]));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + kThreadsW * i, r = idx >> 5, k = idx & 31;
        const float v = k < 27 && row0 + r < B ? obs[i] : 0.f;
        _Float16 h, l;
        split_f16(v, h, l);
        vmax = fmaxf(vmax, fabsf(v));
        sObs[r * kLdO + k] = h;
        sObs[r * kLdO + 32 + k] = l;
    }
    const float4* const cw = reinterpret_cast<const float4*>(rep_conv_c) + wave * 64 + lane;
    const float4* const pw = reinterpret_cast<const float4*>(w.pred_conv_s) + wave * 64 + lane;
    const float4 cb = *reinterpret_cast<const float4*>(w.rep_conv_b + lcol);
    const float4 db = *reinterpret_cast<const float4*>(w.rep_dense_b + lcol);
    // relu(conv + bias) of one cell into a tile as split images; ch / cl: the cell's matrix fragments (h | l)
    auto conv_cell = [&](const float4 ch, const float4 cl, _Float16* tile) {
        const f16x8 bh = __builtin_bit_cast(f16x8, ch), bl = __builtin_bit_cast(f16x8, cl);
        const _Float16* ap = sObs + n16 * kLdO + 8 * g;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f16x8 ah = __builtin_bit_cast(f16x8, *reinterpret_cast<const float4*>(ap + m * 16 * kLdO));
            const f16x8 al = __builtin_bit_cast(f16x8, *reinterpret_cast<const float4*>(ap + m * 16 * kLdO + 32));
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, x = {0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah, a, 0, 0, 0);
            x = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah, x, 0, 0, 0);
            x = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al, x, 0, 0, 0);
            a += x * kUnsplit;
            const float v[4] = {fmaxf(a[0] + cb.x, 0.f), fmaxf(a[1] + cb.y, 0.f), fmaxf(a[2] + cb.z, 0.f),
                                fmaxf(a[3] + cb.w, 0.f)};
            uint2 h, l;
            split4(v, h, l, vmax);
            _Float16* row = tile + (16 * m + n16) * kLdW + lcol;
            *reinterpret_cast<uint2*>(row) = h;
            *reinterpret_cast<uint2*>(row + 128) = l;
        }
    };
    // relu(Dense + bias) = hidden[cell] into a tile as split images (and, unsplit, to the f32 rows)
    auto dense_out = [&](f32x4 (&acc)[4], const f32x4 (&acx)[4], _Float16* tile, int cell) {
        fold4(acc, acx);
        float* const ocell = out_hidden + (size_t)row0 * 1152 + cell * 128;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float x[4] = {fmaxf(acc[m][0] + db.x, 0.f), fmaxf(acc[m][1] + db.y, 0.f), fmaxf(acc[m][2] + db.z, 0.f),
                                fmaxf(acc[m][3] + db.w, 0.f)};
            uint2 h, l;
            split4(x, h, l, vmax);
            _Float16* row = tile + (16 * m + n16) * kLdW + lcol;
            *reinterpret_cast<uint2*>(row) = h;
            *reinterpret_cast<uint2*>(row + 128) = l;
            if (!SPLIT_IO && row0 + 16 * m + n16 < B)
                *reinterpret_cast<float4*>(ocell + (unsigned)(16 * m + n16) * 1152u + lcol) = make_float4(x[0], x[1], x[2], x[3]);
        }
    };
    // a complete hidden tile is the row image the split workspace wants: 16 bytes a thread
    auto copy_out = [&](const _Float16* tile, int cell) {
        if (!SPLIT_IO) return;
        float* const ocell = out_hidden + (size_t)row0 * 1152 + cell * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + kThreadsW * i, r = idx >> 5, q = idx & 31;
            if (row0 + r < B)
                *reinterpret_cast<float4*>(ocell + (unsigned)r * 1152u + q * 4) =
                    *reinterpret_cast<const float4*>(tile + r * kLdW + q * 8);
        }
    };
    const int cell0 = w.job_cells[job][0], cell1 = w.job_cells[job][1];
    float4 ch = cw[(size_t)cell0 * 1024], cl = cw[(size_t)cell0 * 1024 + 512];
    BFragW bd, bp;
    load_bw(bd, rep_dense_s, lane, wave);            // the same Dense for every cell
    __syncthreads();                                 // sObs
    conv_cell(ch, cl, sX);
    ch = cw[(size_t)cell1 * 1024];
    cl = cw[(size_t)cell1 * 1024 + 512];
    load_bw(bp, w.pred_conv_s + (size_t)cell0 * 16384, lane, wave);
    f32x4 accp[4], accpx[4], acc[4], acx[4];
    zero4(accp);
    zero4(accpx);
    zero4(acc);
    zero4(acx);
    __syncthreads();
    // section 0: Dense of the first cell, conv of the second
    mma_tile_w(sX, bd, acc, acx, n16, g);
    conv_cell(ch, cl, sX + TILE);
    dense_out(acc, acx, sH, cell0);
    __syncthreads();
    copy_out(sH, cell0);
#pragma unroll 1
    for (int j = 1; j < ncell; ++j) {
        const int cell = w.job_cells[job][j];        // kernel argument: a scalar load
        const int nc = j + 1 < ncell ? w.job_cells[job][j + 1] : cell;
        ch = cw[(size_t)nc * 1024];
        cl = cw[(size_t)nc * 1024 + 512];
        zero4(acc);
        zero4(acx);
        // the previous cell's term of the prediction conv (its fragments make way for this cell's), this cell's Dense
        mma_tile_w<true>(sH + ((j - 1) & 1) * TILE, bp, accp, accpx, n16, g, pw + (size_t)cell * 4096);
        mma_tile_w(sX + (j & 1) * TILE, bd, acc, acx, n16, g);
        if (j + 1 < ncell) conv_cell(ch, cl, sX + ((j + 1) & 1) * TILE);
        dense_out(acc, acx, sH + (j & 1) * TILE, cell);
        __syncthreads();
        copy_out(sH + (j & 1) * TILE, cell);
    }
    mma_tile_w(sH + ((ncell - 1) & 1) * TILE, bp, accp, accpx, n16, g);
    fold4(accp, accpx);
    float* const pjob = ppart + (size_t)row0 * 1152 + job * 128;
#pragma unroll
    for (int m = 0; m < 4; ++m)
        if (row0 + 16 * m + n16 < B)
            *reinterpret_cast<f32x4*>(pjob + (unsigned)(16 * m + n16) * 1152u + lcol) = accp[m];
    note_range(vmax, w.range_flag);
}

// Prediction head of 16 games by one 256-thread workgroup: relu(sum of the NP partial products + bias)
// (games/tictactoe.py:201) -> Dense 128 -> 128 relu (:204) -> value (tanh, :206) and the 9 policy logits (:208), and
// the reward head's tanh over the 9 per-cell partial dots (:193-195).  Split-f16 arithmetic like the job kernel
// (v = h + l/2048, three exact f16 products per f32 product) on v_mfma_f32_16x16x32_f16: a Dense layer is 4 k-steps x 3
// MFMAs of 16 cycles per 16-column block instead of 32 f32 MFMAs of 32.  A lane's operand for k-step s is the 8 halfs
// k = 32 s + 8 (lane / 16) .. + 7 of row / column lane % 16 (weights packed that way by Packer::add_split16, activations
// as [game][128 h | 128 l] images in LDS).  Wave v owns the Dense layer's columns 32v .. 32v+31 as two 16-column blocks,
// with the weights as the row operand: a lane then holds four consecutive columns of one game, split and stored 8 bytes
// at a time.  The ten head outputs are one more 16-column block (wave 0).
// Results: sOut[game][0..8] logits, [9] value, [10] reward.  Ends with a barrier.
// The head's weights as a lane holds them: what ttt_head16 needs that does not depend on any game (ttt_run_kernel asks for
// them while it waits for the row block's partials).
// CB = 16-column blocks of the Dense layer per wavefront: 2 in a 256-thread workgroup (four wavefronts), 1 in the 512-thread
// workgroup of ttt_run_kernel (eight wavefronts: half the fragments to hold, half the MFMAs a wavefront; the same values)
template <int CB = 2>
struct HeadW {
    float4 db[CB];                 // Dense bias of this lane's columns
    float hb;                      // head bias of output lane % 16
    float4 bF[CB][4][2];           // Dense fragments [column block][k-step][h | l]
    float4 hF[4][2];               // head fragments (wavefront 0)
};
// wavefront 0's fragments of the ten outputs.  CB = 1 (ttt_run_kernel) asks for them inside ttt_head16, behind the loads
// of the partials: held across the meeting they cost 32 registers the job loop does not have.
template <int CB>
__device__ __forceinline__ void ttt_head_weights_out(const TttDev& w, HeadW<CB>& hw, int lane) {
    const float4* hp = reinterpret_cast<const float4*>(w.pred_head_p) + lane;                 // column block 0
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        hw.hF[s][0] = hp[(2 * s) * 512];
        hw.hF[s][1] = hp[(2 * s + 1) * 512];
    }
}
template <int CB>
__device__ __forceinline__ void ttt_head_weights(const TttDev& w, HeadW<CB>& hw) {
    int tid_opaque = threadIdx.x;                    // (opaque to the compiler: see ttt_ring_body)
    asm volatile("" : "+v"(tid_opaque));
    const int tid = tid_opaque & (CB == 2 ? kThreads - 1 : kThreadsW - 1), lane = tid & 63, wave = tid >> 6;
    const int n16 = lane & 15, g = lane >> 4;
    // biases first: a load issued after the MFMAs would show its whole latency
#pragma unroll
    for (int c = 0; c < CB; ++c) hw.db[c] = *reinterpret_cast<const float4*>(w.pred_dense_b + (CB * wave + c) * 16 + 4 * g);
    hw.hb = w.pred_head_b[n16];
    const float4* wp = reinterpret_cast<const float4*>(w.pred_dense_p) + wave * (64 * CB) + lane;   // column block CB wave + c
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            hw.bF[c][s][0] = wp[(2 * s) * 512 + c * 64];
            hw.bF[c][s][1] = wp[(2 * s + 1) * 512 + c * 64];
        }
    if (CB == 2 && wave == 0) ttt_head_weights_out(w, hw, lane);
}

template <int NP, bool REWARD, bool COH = false, int CB = 2>
__device__ __forceinline__ void ttt_head16(const TttDev& w, int B, int row0, const float* __restrict__ ppart,
                                           const float* __restrict__ rpart, float* sXF, float* sHF, float (*sOut)[12],
                                           const HeadW<CB>& hw) {
    _Float16* sX = reinterpret_cast<_Float16*>(sXF);
    _Float16* sH = reinterpret_cast<_Float16*>(sHF);
    float vmax = 0.f;
    // CB = 2: four wavefronts (a 512-thread workgroup's upper four would repeat the lower four's work: the same values
    // into the same LDS words); CB = 1: eight wavefronts, one row of the tile a thread and one column block a wavefront.
    // Every barrier below is reached by the whole workgroup.
    int tid_opaque = threadIdx.x;                    // (opaque to the compiler: see ttt_ring_body)
    asm volatile("" : "+v"(tid_opaque));
    const int tid = tid_opaque & (CB == 2 ? kThreads - 1 : kThreadsW - 1), lane = tid & 63, wave = tid >> 6;
    const int n16 = lane & 15, g = lane >> 4;
    constexpr int RT = CB;                           // rows of the tile per thread
    const float4 (&db)[CB] = hw.db;
    const float hb = hw.hb;
    const float4 (&bF)[CB][4][2] = hw.bF;
    HeadW<CB> hw_out;                                // CB = 1: the output fragments, requested below
    // the reward partials too (wave 1): requested now, summed under the first barrier instead of at the very end
    float rp[9];
    if (REWARD && wave == 1 && lane < 16) {
        const int b = min(row0 + lane, B - 1);
#pragma unroll
        for (int c = 0; c < 9; ++c) rp[c] = load1<COH>(rpart + (size_t)b * 9 + c);
    }
    {   // every load of the tile is issued before the first use
        const int c4 = tid & 31;
        const float4 bn = *reinterpret_cast<const float4*>(w.pred_conv_b + 4 * c4);
        float4 v[RT][NP];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int b = min(row0 + (tid >> 5) + 8 * i, B - 1);
            const float4* pp = reinterpret_cast<const float4*>(ppart + (size_t)b * 1152) + c4;
#pragma unroll
            for (int c = 0; c < NP; ++c) v[i][c] = load4<COH>(pp + c * 32);
        }
        if (CB == 1 && wave == 0) ttt_head_weights_out(w, hw_out, lane);
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            float4 t = bn;
            if (NP == 6) {
                // the 64-game jobs as launches of their own keep the corners' terms in slices of their own (4: cell 2,
                // 5: cell 6); ttt_run_kernel computes them in the workgroups of slices 1 and 2 and adds them there --
                // the same two additions, made here first, give the same bits
#pragma unroll
                for (int c = 1; c <= 2; ++c) {
                    v[i][c].x += v[i][c + 3].x;
                    v[i][c].y += v[i][c + 3].y;
                    v[i][c].z += v[i][c + 3].z;
                    v[i][c].w += v[i][c + 3].w;
                }
            }
#pragma unroll
            for (int c = 0; c < (NP == 6 ? 4 : NP); ++c) {
                t.x += v[i][c].x;
                t.y += v[i][c].y;
                t.z += v[i][c].z;
                t.w += v[i][c].w;
            }
            const float x[4] = {fmaxf(t.x, 0.f), fmaxf(t.y, 0.f), fmaxf(t.z, 0.f), fmaxf(t.w, 0.f)};
            uint2 h, l;
            split4(x, h, l, vmax);
            _Float16* row = sX + ((tid >> 5) + 8 * i) * kLdH + 4 * c4;
            *reinterpret_cast<uint2*>(row) = h;
            *reinterpret_cast<uint2*>(row + 128) = l;
        }
    }
    if (REWARD && wave == 1 && lane < 16) {
        float rs = 0.f;
#pragma unroll
        for (int c = 0; c < 9; ++c) rs += rp[c];
        sOut[lane][10] = tanhf(rs + w.dyn_reward_b);
    }
    __syncthreads();
    {
        const _Float16* ap = sX + n16 * kLdH + 8 * g;
        float4 aF[4][2];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            aF[s][0] = *reinterpret_cast<const float4*>(ap + 32 * s);
            aF[s][1] = *reinterpret_cast<const float4*>(ap + 128 + 32 * s);
        }
        f32x4 acc[CB], acx[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c) acc[c] = acx[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f16x8 ah = __builtin_bit_cast(f16x8, aF[s][0]), al = __builtin_bit_cast(f16x8, aF[s][1]);
#pragma unroll
            for (int c = 0; c < CB; ++c) {           // weights as the row operand: acc[c][i] = column 16c + 4g + i, game n16
                const f16x8 bh = __builtin_bit_cast(f16x8, bF[c][s][0]), bl = __builtin_bit_cast(f16x8, bF[c][s][1]);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah, acc[c], 0, 0, 0);
                acx[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah, acx[c], 0, 0, 0);
                acx[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al, acx[c], 0, 0, 0);
            }
        }
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const float bb[4] = {db[c].x, db[c].y, db[c].z, db[c].w};
            float x[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = fmaxf(acc[c][i] + acx[c][i] * kUnsplit + bb[i], 0.f);
            uint2 h, l;
            split4(x, h, l, vmax);
            _Float16* row = sH + n16 * kLdH + (CB * wave + c) * 16 + 4 * g;
            *reinterpret_cast<uint2*>(row) = h;
            *reinterpret_cast<uint2*>(row + 128) = l;
        }
    }
    __syncthreads();
    if (wave == 0) {
        const _Float16* ap = sH + n16 * kLdH + 8 * g;
        float4 aF[4][2];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            aF[s][0] = *reinterpret_cast<const float4*>(ap + 32 * s);
            aF[s][1] = *reinterpret_cast<const float4*>(ap + 128 + 32 * s);
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acx0 = {0.f, 0.f, 0.f, 0.f}, acx1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {                // games as the row operand: acc[i] = game 4g + i, output n16
            const f16x8 ah = __builtin_bit_cast(f16x8, aF[s][0]), al = __builtin_bit_cast(f16x8, aF[s][1]);
            const float4 (&hF)[4][2] = CB == 1 ? hw_out.hF : hw.hF;
            const f16x8 bh = __builtin_bit_cast(f16x8, hF[s][0]), bl = __builtin_bit_cast(f16x8, hF[s][1]);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
            acx0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acx0, 0, 0, 0);
            acx1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acx1, 0, 0, 0);
        }
        if (n16 < 10) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = acc[i] + (acx0[i] + acx1[i]) * kUnsplit + hb;
                if (n16 == 0) sOut[4 * g + i][9] = tanhf(v);        // value; to_scalar is the identity (config.py:28)
                else sOut[4 * g + i][n16 - 1] = v;
            }
        }
    }
    note_range(vmax, w.range_flag);
    __syncthreads();
}

// Kernel B.  grid = ceil(B / 16).  parent_block only tells which rows are skipped.
template <bool INITIAL, int NP>
__global__ __launch_bounds__(kThreads) void ttt_head_kernel(TttDev w, int B, const int* __restrict__ parent_block,
                                                            const float* __restrict__ ppart, const float* __restrict__ rpart,
                                                            float* __restrict__ out_reward, float* __restrict__ out_value,
                                                            float* __restrict__ out_logits, float* __restrict__ out_value_raw) {
    __shared__ __attribute__((aligned(16))) float sBuf[2][16 * kLd];
    __shared__ float sOut[16][12];
    const int row0 = blockIdx.x * 16;
    HeadW<2> hw;
    ttt_head_weights(w, hw);
    ttt_head16<NP, !INITIAL>(w, B, row0, ppart, rpart, sBuf[0], sBuf[1], sOut, hw);
    const int tid = threadIdx.x;
    if (tid < 16) {
        const int b = row0 + tid;
        if (b < B && (INITIAL || !parent_block || parent_block[b] >= 0)) {
            out_value[b] = sOut[tid][9];
            if (out_value_raw) out_value_raw[b] = sOut[tid][9];
#pragma unroll
            for (int o = 0; o < 9; ++o) out_logits[(size_t)b * 9 + o] = sOut[tid][o];
            if (!INITIAL) out_reward[b] = sOut[tid][10];
        }
    }
}

// Kernel B of a native move's initial inference fused with what follows it: the root expansion over the logits just
// computed, the Dirichlet noise and the first selection (expand_roots_kernel<16, true, 1>'s code, mz_tree_dev.h, on the same
// 16 lanes per game: bit-identical trees).  One launch boundary less per move.
__global__ __launch_bounds__(kThreads) void ttt_head_roots_kernel(TttDev w, PoolDev p, const float* __restrict__ ppart,
                                                                  float* out_value, float* out_logits, const int* legal,
                                                                  const int* n_legal, const double* noise, double frac,
                                                                  int* out_parent, int* out_action) {
    constexpr int G = 16;
    __shared__ __attribute__((aligned(16))) float sBuf[2][16 * kLd];
    __shared__ float sOut[16][12];
    const int row0 = blockIdx.x * 16, tid = threadIdx.x;
    HeadW<2> hw;
    ttt_head_weights(w, hw);
    ttt_head16<4, false>(w, p.T, row0, ppart, nullptr, sBuf[0], sBuf[1], sOut, hw);
    if (tid < 16 * 10) {
        const int b = row0 + tid / 10, o = tid % 10;
        if (b < p.T) {
            if (o == 9) out_value[b] = sOut[tid / 10][9];
            else out_logits[(size_t)b * 9 + o] = sOut[tid / 10][o];
        }
    }
    // the tree code reads the logits where the caller gets them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    expand_roots_tree<G, true, 1>(p, row0 + tid / G, tid % G, out_logits, legal, n_legal, noise, frac, out_parent, out_action);
}

// Kernel B fused with the tree step (expand + backprop of this simulation, select of the next: mz_step_logits).
// One workgroup = 16 games: the head needs nothing but these games' partials and the tree step nothing but their
// logits / value / reward, which go from the head to the tree code through LDS.  One launch boundary less per
// simulation.  Tree side: 16 lanes per game (A = 9), exactly step_kernel<16, true>'s code (mz_tree_dev.h), so the trees
// are bit-identical to the separate launches.
// 512 threads: the head on eight wavefronts (one Dense column block each, ttt_head16's CB = 1), the tree step on the lower
// four.
template <bool SELECT, int NP>
__global__ __launch_bounds__(kThreadsW) void ttt_head_step_kernel(TttDev w, PoolDev p, int nb, const float* __restrict__ ppart,
                                                                 const float* __restrict__ rpart, float* __restrict__ out_reward,
                                                                 float* __restrict__ out_value, float* __restrict__ out_logits,
                                                                 int* out_parent, int* out_action) {
    constexpr int G = 16;
    __shared__ __attribute__((aligned(16))) float sBuf[2][16 * kLd];
    __shared__ float sOut[16][12];
    // the 16 games of a block lie in the row block the job kernel ran on this block's XCD (blockIdx % 8): their partials
    // are in this XCD's L2
    // (NP = 9, the 16-game jobs: row block b of the job kernel ran on XCD b % 8, and so does block b here)
    const int xk = blockIdx.x >> 3;
    const int tid = threadIdx.x,
              row0 = NP == 9 ? blockIdx.x * 16 : (((xk >> 2) * 8 + (blockIdx.x & 7)) * 4 + (xk & 3)) * 16;
    const int hgame = (tid & (kThreads - 1)) / G;
    const int tree = tid < kThreads ? row0 + hgame : p.T, gl = tid % G;
    const bool active = tree < p.T && p.n_root[tree] > 0;
    const BackupPrefetch pf = backprop_prefetch<G>(p, tree, active, gl);      // lands while the head computes
    MZ_TRACE_DECL;
    HeadW<1> hw;
    ttt_head_weights(w, hw);
    ttt_head16<NP, true, false, 1>(w, p.T, row0, ppart, rpart, sBuf[0], sBuf[1], sOut, hw);
    MZ_TRACE(0);                                                              // head
    if (active && gl < 11) {                                         // the plain outputs of mz_infer_recurrent
        const float x = sOut[hgame][gl];
        if (gl < 9) out_logits[(size_t)tree * 9 + gl] = x;
        else if (gl == 9) out_value[tree] = x;
        else out_reward[tree] = x;
    }
    // ---- tree step of these 16 games (agent.py:208-211, then 177-190 for the next simulation)
    const float* mine = sOut[hgame];
    backprop_tree<G, true>(p, tree, active, gl, nb, mine, mine[10], mine[9], pf);
    if (SELECT) {
        // a tree's 16 lanes lie inside one wavefront, whose vector memory operations are performed in program order:
        // compiler ordering is enough (step_kernel, mz_tree.hip)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        select_tree<G>(p, tree, active, gl, out_parent, out_action);
    }
}

// ---- run_mcts of a TicTacToe batch in ONE launch (ttt_run_kernel) ---------------------------------------------------
// The four workgroups of a row block (64 games x four cell groups) stay resident for all N simulations of a move and
// meet twice per simulation: after the jobs (the prediction-conv partials, reward partials and hidden' rows of the row
// block are complete -> each workgroup runs the head and the tree step of 16 of the 64 games) and after the tree steps
// (the next selection of all 64 games is known -> the next jobs gather their parent rows).  What a launch boundary did
// for these two hand-overs -- make one workgroup's stores visible to the others -- the XCD's L2 does: the four
// workgroups run on ONE XCD (ttt_job_of_block), a store is in that L2 once `s_waitcnt vmcnt(0)` has returned, and a
// reader only has to read past its CU's L1 (`sc1` / `nt` loads: load1 / load4) what it may have cached before.  The
// expensive half of an agent-scope release (`buffer_wbl2`: write the whole L2 back) is never needed on that path.
// On that path the second hand-over is not even a meeting: the selections go out as words tagged with the simulation
// they are for (RunSync::sel_words) and the next job's first wavefront looks at its row block's 64 words until all carry
// its simulation -- one round trip where the counter, the look at it and the load of the selection were three.
//   * Where the workgroups run is checked, not assumed: each ORs the bit of its hardware XCC id into a word of its row
//     block before a first meeting made of agent-scope atomics only; four equal ids -> the L2 path; anything else -> the
//     same protocol with full agent-scope release / acquire fences around every meeting (correct wherever the workgroups
//     are, slower).
//   * No wave can wait for ever: the workgroups of a launch are dispatched in blockIdx order XCD by XCD, a row block's
//     four are consecutive in their XCD's order, so whenever a workgroup is resident every workgroup it can ever wait for
//     is resident too or will be as soon as some complete row block ahead of it retires; and a meeting that is not
//     complete after kRunSpinTicks of the 100 MHz wall clock gives up, raises bit 3 of the pool's error flag and ends
//     the workgroup's run (every later meeting of the others then times out the same way).
// The arrival counters are 64-bit words that carry the launch's serial number in their upper bits (host: mz_net::run_epoch),
// set by an atomic max at the first meeting: nothing has to be cleared between launches, and a launch that gave up leaves
// nothing behind that the next one could take for its own arrivals.
constexpr int kRunEpochShift = 20;

constexpr int kRunMaxRowBlocks = 4096;               // 262144 games per launch
struct RunSync {
    unsigned long long* arrive;     // [row block]: (epoch << 20) + arrivals of this launch
    unsigned* xcc_mask;             // [row block]: bits of the XCC ids its workgroups have run on
    unsigned long long* cell_flags; // [row block][2]: (epoch << 20) + 1 + the last simulation whose hidden' rows of the
                                    // corners 2 / 6 are complete (the prediction-conv stages that change workgroups)
    unsigned long long* stats;      // [0] row-block runs that took the fence path, [1] row-block runs, [2] 100 MHz wall-clock
                                    // ticks workgroups spent in their jobs, [3] the jobs those ticks are of (measurement)
    unsigned long long* sel_words;  // [row block][64]: a game's selection for the next simulation, tagged with the
                                    // launch and that simulation: (epoch << 12 | simulation) << kSelTagShift |
                                    // (parent block + 1) << 8 | action & 0xff
    unsigned long long epoch_base;  // epoch << kRunEpochShift of this launch
    int force_fences;               // tests: take the agent-scope fence path whatever the placement
    int tagged_ok;                  // the simulations and blocks of this launch fit a selection word's fields
};

// A meeting of the four workgroups of a row block, in two halves so that loads which do not depend on the others can be
// requested in between.  All 512 threads call both.  `fenced`: full agent-scope release / acquire (workgroups on
// different XCDs).  meet_arrive: this workgroup's stores are visible, its arrival counted; the workgroup that arrives
// last learns so from the counter's old value and does not have to look again.  meet_wait -> false: gave up.
__device__ __forceinline__ void meet_arrive(unsigned long long* arrive, unsigned long long target, bool fenced, bool first,
                                            unsigned long long epoch_base, int* s_flag) {
    if (fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");        // buffer_wbl2 sc1 + wait
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");           // this wave's stores are in the XCD's L2
    __syncthreads();
    if (threadIdx.x == 0) {
        if (first) {
            const unsigned long long was = __hip_atomic_fetch_max(arrive, epoch_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" ::"v"(was));                                   // performed before the arrival below is issued
        }
        const unsigned long long before = __hip_atomic_fetch_add(arrive, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_flag[0] = (long long)(before + 1 - target) >= 0 ? 1 : 0;         // 1: everybody is here already
    }
}
__device__ __forceinline__ bool meet_wait(unsigned long long* arrive, unsigned long long target, bool fenced, int* s_flag,
                                          int* err_flag) {
    if (threadIdx.x == 0 && s_flag[0] == 0) {
        const long long t0 = wall_clock64();
        int ok = 1;
        for (;;) {
            const unsigned long long seen = __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((long long)(seen - target) >= 0) break;
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > kRunSpinTicks) {
                ok = 0;
                atomicOr(err_flag, 8);
                break;
            }
        }
        s_flag[0] = ok;
    }
    __syncthreads();
    if (fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");        // buffer_inv sc1
    return s_flag[0] != 0;                  // (the caller alternates between two flag words from meeting to meeting)
}

// Experiment builds (-DMZ_RUN_TRACE, tools/ttt_run_trace.py): thread 0 of every workgroup adds the 100 MHz wall-clock
// ticks of each phase of a simulation to g_run_trace[blockIdx][phase].  Compiles to nothing in the library.
#ifdef MZ_RUN_TRACE
__device__ unsigned long long g_run_trace[2048][8];
#define MZ_RUN_STAMP(phase)                                                       \
    do {                                                                          \
        if (threadIdx.x == 0 && blockIdx.x < 2048) {                              \
            const long long now_ = wall_clock64();                                \
            g_run_trace[blockIdx.x][phase] += (unsigned long long)(now_ - rt_t_); \
            rt_t_ = now_;                                                         \
        }                                                                         \
    } while (0)
#define MZ_RUN_STAMP_DECL long long rt_t_ = wall_clock64()
#else
#define MZ_RUN_STAMP(phase)
#define MZ_RUN_STAMP_DECL
#endif

template <bool SPLIT_IO>
__global__ __launch_bounds__(kThreadsW, 1) void ttt_run_kernel(TttDev w, const PoolDev* pool_image, int T, int* err_flag,
                                                               RunSync sync, int nb0, int nsim,
                                                               int final_select, float* hidden_base, int* parent_block,
                                                               long block_stride, int* action, float* ppart, float* rpart,
                                                               float* out_reward, float* out_value, float* out_logits,
                                                               long out_stride, int* trace_parent, int* trace_action) {
    constexpr int G = 16;
    __shared__ __attribute__((aligned(16))) float sBuf[2][16 * kLd];
    __shared__ float sOut[16][12];
    __shared__ int s_flags[3];                       // [0], [1]: the meetings' flag words, alternating; [2]: fence path
    __shared__ int s_sel[2][16];                     // the tree steps' selections (parent block, action) on their way out
    int job, rb;
    ttt_job_of_block<4>(job, rb);
    if (rb >= (T + 63) / 64) return;               // grid padding: the same for the four workgroups of a row block
    // ---- first meeting: agent-scope atomics only (no data crosses it); tells where the four workgroups run
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(6164) & 15u;        // hwreg(HW_REG_XCC_ID, 0, 4)
        const unsigned was = __hip_atomic_fetch_or(sync.xcc_mask + rb, 1u << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(was) : "memory");
    }
    unsigned long long target = sync.epoch_base + 4;
    meet_arrive(sync.arrive + rb, target, false, true, sync.epoch_base, s_flags);
    bool alive = meet_wait(sync.arrive + rb, target, true, s_flags, err_flag);
    if (threadIdx.x == 0) {
        const unsigned m = __hip_atomic_load(sync.xcc_mask + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_flags[2] = (sync.force_fences != 0 || __builtin_popcount(m) != 1) ? 1 : 0;
    }
    __syncthreads();
    const bool fenced = s_flags[2] != 0;
    if (threadIdx.x == 0 && job == 0) {
        atomicAdd(sync.stats + 1, 1ull);
        if (fenced) atomicAdd(sync.stats, 1ull);
    }
    const int hrow0 = rb * 64 + job * 16;
    // On the L2 path the second meeting of a simulation is no counter: the 16 selections a workgroup has made go out as
    // tagged words and the next job's first wavefront looks at the 64 of its row block until all are there (ttt_ring_body).
    // A workgroup that sees them knows that all four are past their heads (a selection is made after its game's partials
    // were read), so its next job may overwrite the partials; hidden rows were published by the meeting after the jobs.
    const bool tagged = !fenced && sync.tagged_ok != 0;
    const unsigned long long sel_epoch = ((sync.epoch_base >> kRunEpochShift) & 0xffffffffull) << 12;
    MZ_RUN_STAMP_DECL;
    long long job_ticks = 0;                         // thread 0: wall-clock ticks inside the jobs (mz_debug_run_info)
    int jobs_done = 0;
#pragma unroll 1
    for (int i = 0; i < nsim && alive; ++i) {
        asm volatile("" ::: "memory");
        const int nb = nb0 + i;
        const bool traced = trace_parent != nullptr;
        const long long job_t0 = wall_clock64();
        long long sel_wait = 0;
        alive = ttt_ring_body<SPLIT_IO>(w, T, hidden_base, parent_block, block_stride, action,
                                        hidden_base + (size_t)nb * block_stride, ppart, rpart, rb, job,
                                        traced ? trace_parent + (size_t)i * T : nullptr,
                                        traced ? trace_action + (size_t)i * T : nullptr, true, sync.cell_flags + 2 * rb,
                                        sync.epoch_base + (unsigned long long)i + 1, err_flag,
                                        tagged && i > 0 ? sync.sel_words + (size_t)rb * 64 : nullptr,
                                        sel_epoch | (unsigned long long)i, &sel_wait);
        if (!alive) break;
        job_ticks += wall_clock64() - job_t0 - sel_wait;
        ++jobs_done;
        MZ_RUN_STAMP(0);
        target += 4;
        meet_arrive(sync.arrive + rb, target, fenced, false, 0, s_flags + 1);
        HeadW<1> hw;                                 // requested while the arrival is on its way
        ttt_head_weights(w, hw);
        alive = meet_wait(sync.arrive + rb, target, fenced, s_flags + 1, err_flag);
        MZ_RUN_STAMP(1);
        if (!alive) break;
        // ---- head and tree step of 16 games (ttt_head_step_kernel's body): 16 of the row block's 64 games, by the
        // workgroup's lower four wavefronts (lane geometry from an opaque copy of the thread index: not hoisted)
        int tx = threadIdx.x;
        asm volatile("" : "+v"(tx));
        const int ht = tx & (kThreads - 1);
        // The pool descriptor (65 scalars) is read HERE, every simulation, from its image in device memory (scalar loads
        // through a pointer the compiler cannot see through, so that it cannot keep them across the jobs): held for the
        // whole run it pushed the job loop's own scalars out of the register file -- 66 v_readlane a stage.
        struct PoolWords {
            int w[sizeof(PoolDev) / sizeof(int)];
        };
        typedef const __attribute__((address_space(4))) int* PoolImage;
        PoolImage image = (PoolImage)reinterpret_cast<const int*>(pool_image);
        asm volatile("" : "+s"(image));
        PoolWords words;
#pragma unroll
        for (int j = 0; j < (int)(sizeof(PoolDev) / sizeof(int)); ++j) words.w[j] = image[j];
        const PoolDev p = __builtin_bit_cast(PoolDev, words);
        const int tree = tx < kThreads ? hrow0 + ht / G : p.T, gl = ht % G;
        const bool active = tree < p.T && p.n_root[tree] > 0;
        const BackupPrefetch pf = backprop_prefetch<G>(p, tree, active, gl);      // lands while the head computes
        ttt_head16<4, true, true, 1>(w, p.T, hrow0, ppart, rpart, sBuf[0], sBuf[1], sOut, hw);
        const bool store_out = out_stride != 0 || i + 1 == nsim;                  // nobody overwrites these
        if (active && gl < 11 && store_out) {
            const float x = sOut[ht / G][gl];
            if (gl < 9) out_logits[((size_t)i * out_stride + tree) * 9 + gl] = x;
            else if (gl == 9) out_value[(size_t)i * out_stride + tree] = x;
            else out_reward[(size_t)i * out_stride + tree] = x;
        }
        MZ_RUN_STAMP(2);
        const float* mine = sOut[ht / G];
        backprop_tree<G, true>(p, tree, active, gl, nb, mine, mine[10], mine[9], pf);
        if (i + 1 < nsim || final_select != 0) {
            // (a tree's 16 lanes lie inside one wavefront: compiler ordering is enough, as in step_kernel)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            MZ_RUN_STAMP(3);
            if (!tagged) {
                select_tree<G>(p, tree, active, gl, parent_block, action);
                MZ_RUN_STAMP(4);
                if (i + 1 < nsim) {                  // the selection of all 64 games, for the next jobs
                    target += 4;
                    meet_arrive(sync.arrive + rb, target, fenced, false, 0, s_flags);
                    alive = meet_wait(sync.arrive + rb, target, fenced, s_flags, err_flag);
                    MZ_RUN_STAMP(5);
                }
            } else {
                select_tree<G>(p, tree, active, gl, &s_sel[0][0] - hrow0, &s_sel[1][0] - hrow0);
                __syncthreads();
                MZ_RUN_STAMP(4);
                if (tx < 16 && hrow0 + tx < p.T) {
                    const int pa = s_sel[0][tx], ac = s_sel[1][tx];
                    if (i + 1 < nsim) {
                        const unsigned long long word = ((sel_epoch | (unsigned long long)(i + 1)) << kSelTagShift) |
                                                        ((unsigned long long)(unsigned)(pa + 1) << 8) | (unsigned)(ac & 0xff);
                        __hip_atomic_store(sync.sel_words + hrow0 + tx, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {                         // the launch's last: for the host, or the next launch of the move
                        parent_block[hrow0 + tx] = pa;
                        action[hrow0 + tx] = ac;
                    }
                }
                MZ_RUN_STAMP(5);
            }
        }
    }
    if (threadIdx.x == 0 && jobs_done > 0) {
        atomicAdd(sync.stats + 2, (unsigned long long)job_ticks);
        atomicAdd(sync.stats + 3, (unsigned long long)jobs_done);
    }
}

// ----------------------------------------------------------------------------------------------- CartPole
struct CartDev {
    const float* rep_conv;      // [4 positions][101 bins][128] (Keras order)
    const float* rep_conv_b;
    const float* rep_dense_p;
    const float* rep_dense_b;
    const float* dyn_conv_p;    // packed [16][128][8]: taps of position 0 (the hidden row)
    const float* dyn_conv_b;    // [128]
    const float* dyn_plane;     // [128]: sum_k W[1][k][:], the all-ones action plane (network.py:88-104)
    const float* dyn_dense_p;
    const float* dyn_dense_b;
    const float* dyn_rdense_p;
    const float* dyn_rdense_b;
    const float* dyn_reward_w;  // [128]
    const float* pred_dense1_p;
    const float* pred_dense1_b;
    const float* pred_dense2_p;
    const float* pred_dense2_b;
    const float* pred_value_p;  // packed [16][128][8], columns >= 101 zero
    const float* pred_value_b;  // [128], entries >= 101 zero
    const float* pred_policy_w; // [2][128]
    const float* pred_policy_b; // [2]
    // the seven 128 x 128 operands again as split-f16 fragments (Packer::add_split16): cartpole16_kernel
    const float* rep_dense_s;
    const float* dyn_conv_s;
    const float* dyn_dense_s;
    const float* dyn_rdense_s;
    const float* pred_dense1_s;
    const float* pred_dense2_s;
    const float* pred_value_s;
    float dyn_reward_b;
    int* range_flag;            // see TttDev
};

// One launch per call; grid = ceil(B / (32*MT)).  Six chained GEMM stages; each stage's weights are loaded into
// registers while the previous stage multiplies.
template <bool INITIAL, int MT>
__global__ __launch_bounds__(kThreads) void cartpole_kernel(CartDev w, int B, const float* __restrict__ in_base,
                                                            const int* __restrict__ parent_block, long block_stride,
                                                            const int* __restrict__ action, float* __restrict__ out_hidden,
                                                            float* __restrict__ out_reward, float* __restrict__ out_value,
                                                            float* __restrict__ out_logits, float* __restrict__ out_value_raw) {
    constexpr int ROWS = 32 * MT;
    constexpr int NSTG = MT * 4;
    __shared__ __attribute__((aligned(16))) float sBuf[2][ROWS * kLd];
    __shared__ long sRowOff[ROWS];
    __shared__ float sAct[ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * ROWS;
    float* sX = sBuf[0];
    float* sH = sBuf[1];
    BFrag b0, b1;
    load_b(b0, INITIAL ? w.rep_dense_p : w.dyn_conv_p, lane, wave);
    if (tid < ROWS) {
        const int b = row0 + tid;
        long off = -1;
        float a = 0.f;
        if (b < B) {
            if (INITIAL) {
                off = (long)b * 4;
            } else {
                const int pb = parent_block ? parent_block[b] : 0;
                if (pb >= 0) off = (long)pb * block_stride + (long)b * 128;
                a = action[b] == 1 ? 1.f : 0.f;
            }
        }
        sRowOff[tid] = off;
        sAct[tid] = a;
    }
    __syncthreads();
    f32x16 acc[MT];
    const int n = wave * 32 + (lane & 31);
    if (INITIAL) {
        // scalar_to_support two-hot (utils.py:138-143, network.py:107-116) -> Conv1D k=4 valid 101 -> 128 relu
        // (games/cartpole.py:98): only <= 8 of the 404 inputs are non-zero, gathered on the VALU.
        const int nn = tid & 127;
        const float lo[4] = {-4.8f, -5.f, -0.418f, -5.f}, hi[4] = {4.8f, 5.f, 0.418f, 5.f};
        const float bn = w.rep_conv_b[nn];
        for (int r = tid >> 7; r < ROWS; r += 2) {
            float s = bn;
            if (sRowOff[r] >= 0) {
#pragma unroll
                for (int pos = 0; pos < 4; ++pos) {
                    const float x = (in_base[sRowOff[r] + pos] - lo[pos]) / (hi[pos] - lo[pos]);
                    const float y = 100.f * (sqrtf(x + 1.f) - 1.f) / (sqrtf(2.f) - 1.f);
                    const float fl = floorf(y);
                    const float p = y - fl;
                    const int il = (int)fl, ih = (int)ceilf(y);
                    // tf.one_hot yields an all-zero row for an out-of-range index
                    if (ih >= 0 && ih <= 100) s += p * w.rep_conv[(pos * 101 + ih) * 128 + nn];
                    if (il >= 0 && il <= 100) s += (1.f - p) * w.rep_conv[(pos * 101 + il) * 128 + nn];
                }
            }
            sX[r * kLd + nn] = fmaxf(s, 0.f);
        }
        load_b(b1, w.pred_dense1_p, lane, wave);
        __syncthreads();
        zero_acc(acc);
        mma_tile<MT>(sX, b0, acc, lane);                                      // Dense relu = hidden (:101)
    } else {
        // gather leaf.parent.hidden_state rows (agent.py:204) into LDS
#pragma unroll
        for (int i = 0; i < NSTG; ++i) {
            const int idx = tid + kThreads * i, r = idx >> 5, c4 = idx & 31;
            const long off = sRowOff[r];
            const float4 v = off >= 0 ? *reinterpret_cast<const float4*>(in_base + off + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(sH + r * kLd + c4 * 4) = v;
        }
        load_b(b1, w.dyn_dense_p, lane, wave);
        __syncthreads();
        // Conv1D k=2 valid over [hidden ; action plane] (games/cartpole.py:106): hidden @ W[0] + a * sum_k W[1][k]
        zero_acc(acc);
        mma_tile<MT>(sH, b0, acc, lane);
        const float bn = w.dyn_conv_b[n], pn = w.dyn_plane[n];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = acc_row(m, r, lane);
                sX[row * kLd + n] = fmaxf(acc[m][r] + bn + sAct[row] * pn, 0.f);
            }
        load_b(b0, w.dyn_rdense_p, lane, wave);
        __syncthreads();
        zero_acc(acc);
        mma_tile<MT>(sX, b1, acc, lane);                                      // Dense relu = hidden' (:108)
        load_b(b1, w.pred_dense1_p, lane, wave);
    }
    {   // hidden state out (both modes); sH's previous contents (the gathered input) are no longer read
        const float bn = (INITIAL ? w.rep_dense_b : w.dyn_dense_b)[n];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = acc_row(m, r, lane);
                const float v = fmaxf(acc[m][r] + bn, 0.f);
                sH[row * kLd + n] = v;
                if (sRowOff[row] >= 0) out_hidden[(size_t)(row0 + row) * 128 + n] = v;
            }
    }
    __syncthreads();
    const int row = tid >> 2, q = tid & 3, b = row0 + row;
    const bool rowthread = tid < ROWS * 4;
    const bool live = rowthread && sRowOff[rowthread ? row : 0] >= 0;
    if (!INITIAL) {
        // reward branch: Dense 128 relu, Dense 1 sigmoid (games/cartpole.py:111-114)
        zero_acc(acc);
        mma_tile<MT>(sH, b0, acc, lane);
        store_acc_lds<MT, true>(acc, w.dyn_rdense_b, sX, lane, wave);
        __syncthreads();
        if (rowthread) {
            const float s = dot32x4(sX + row * kLd + q * 32, w.dyn_reward_w + q * 32);
            if (q == 0 && live) out_reward[b] = 1.f / (1.f + expf(-(s + w.dyn_reward_b)));
        }
        __syncthreads();
    }
    // prediction: Dense relu, Dense relu (games/cartpole.py:121-124); b1 = pred_dense1 in both modes
    load_b(b0, w.pred_dense2_p, lane, wave);
    zero_acc(acc);
    mma_tile<MT>(sH, b1, acc, lane);
    store_acc_lds<MT, true>(acc, w.pred_dense1_b, sX, lane, wave);
    load_b(b1, w.pred_value_p, lane, wave);
    __syncthreads();
    zero_acc(acc);
    mma_tile<MT>(sX, b0, acc, lane);
    __syncthreads();
    store_acc_lds<MT, true>(acc, w.pred_dense2_b, sH, lane, wave);
    __syncthreads();
    // value head: Dense 101 (columns padded to 128) -> softmax -> support_to_scalar (:125, utils.py:146-148)
    zero_acc(acc);
    mma_tile<MT>(sH, b1, acc, lane);
    store_acc_lds<MT, false>(acc, w.pred_value_b, sX, lane, wave);
    __syncthreads();
    if (rowthread) {
        float* z = sX + row * kLd;
        float mx = -INFINITY;
        for (int j = q; j < 101; j += 4) mx = fmaxf(mx, z[j]);
        mx = fmaxf(mx, __shfl_xor(mx, 1, 4));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 4));
        float den = 0.f;
        for (int j = q; j < 101; j += 4) {
            const float e = expf(z[j] - mx);
            z[j] = e;
            den += e;
        }
        den += __shfl_xor(den, 1, 4);
        den += __shfl_xor(den, 2, 4);
        float y = 0.f;
        for (int j = q; j < 101; j += 4) {
            const float pj = z[j] / den;
            if (out_value_raw && live) out_value_raw[(size_t)b * 101 + j] = pj;
            y += pj * (float)j;
        }
        y += __shfl_xor(y, 1, 4);
        y += __shfl_xor(y, 2, 4);
        if (q == 0 && live) {
            y = y / 100.f;
            const float t = (sqrtf(2.f) - 1.f) * y + 1.f;
            out_value[b] = 0.f + (100.f - 0.f) * (t * t - 1.f);        // config.py:38-40 with bounds (0,100)
        }
        // policy logits: Dense 2 (:127)
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const float s = dot32x4(sH + row * kLd + q * 32, w.pred_policy_w + o * 128 + q * 32);
            if (q == 0 && live) out_logits[(size_t)b * 2 + o] = s + w.pred_policy_b[o];
        }
    }
}

// CartPole on 16-game tiles (batches <= kCart16MaxBatch).  The 32-row kernel above is a latency chain of six stages per
// wavefront (1024 games = 32 workgroups on 256 CUs, 25 us a call); here a workgroup is 16 games x 512 threads, eight
// wavefronts own 16 output columns each, and the arithmetic is the split-f16 scheme of the TicTacToe kernels
// (v = h + l/2048, three exact f16 products per f32 product, f32 accumulation) on v_mfma_f32_16x16x32_f16: a stage is
// 4 k-steps x 3 MFMAs of 16 cycles, where the f32 version ran 2 x 16 dependent 32-cycle MFMAs.  Weights as the row
// operand: a lane holds four consecutive columns of game lane % 16.  Same stages and outputs as cartpole_kernel.
struct BFrag16 {
    float4 v[8];                   // [k-step][h | l] of this wave's 16 columns
};
__device__ __forceinline__ void load_b16(BFrag16& b, const float* __restrict__ Ws, int lane, int wave) {
    const float4* wp = reinterpret_cast<const float4*>(Ws) + wave * 64 + lane;       // Packer::add_split16
#pragma unroll
    for (int j = 0; j < 8; ++j) b.v[j] = wp[j * 512];
}
// sA[16][128 h | 128 l] (LDS, stride kLdW halfs) * W[128][this wave's 16 columns]; returns [i] = column 4*(lane/16) + i of
// the wave's 16, game lane % 16
__device__ __forceinline__ f32x4 mma16(const _Float16* sA, const BFrag16& b, int lane) {
    const _Float16* ap = sA + (lane & 15) * kLdW + 8 * (lane >> 4);
    float4 a[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        a[s][0] = *reinterpret_cast<const float4*>(ap + 32 * s);
        a[s][1] = *reinterpret_cast<const float4*>(ap + 128 + 32 * s);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acx0 = {0.f, 0.f, 0.f, 0.f}, acx1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const f16x8 ah = __builtin_bit_cast(f16x8, a[s][0]), al = __builtin_bit_cast(f16x8, a[s][1]);
        const f16x8 bh = __builtin_bit_cast(f16x8, b.v[2 * s]), bl = __builtin_bit_cast(f16x8, b.v[2 * s + 1]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah, acc, 0, 0, 0);
        acx0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah, acx0, 0, 0, 0);
        acx1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al, acx1, 0, 0, 0);
    }
    return acc + (acx0 + acx1) * kUnsplit;
}
// sum over the four lanes n16 + 16 g of a game (the wave's sixteen columns); every lane gets it
__device__ __forceinline__ float sum_over_g(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const unsigned y = __float_as_uint(__uint_as_float(r16[0]) + __uint_as_float(r16[1]));
    const auto r32 = __builtin_amdgcn_permlane32_swap(y, y, false, false);
    return __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
}

constexpr int kThreads16 = 512;
constexpr int kCart16MaxBatch = 8192;      // up to two 16-game workgroups per CU; larger batches take the 32/64-row tiles

// FUSE: 0 = the network alone; 1 = + expand and backprop of this simulation for the 16 games (mz_expand_backprop_logits),
// 2 = + select of the next one (mz_step_logits), by wave 0 with two lanes per game -- exactly step_kernel<2, 1, true>'s
// code (mz_tree_dev.h), fed through LDS: one launch per simulation.  parent_block / action are then read at the start and
// rewritten at the end, each workgroup its own 16 games.
template <bool INITIAL, int FUSE>
__device__ __forceinline__ void cartpole16_body(const CartDev& w, int B, const float* __restrict__ in_base,
                                                int* parent_block, long block_stride, int* action,
                                                float* __restrict__ out_hidden, float* __restrict__ out_reward,
                                                float* __restrict__ out_value, float* __restrict__ out_logits,
                                                float* __restrict__ out_value_raw, const PoolDev& p, int nb,
                                                bool do_select = true) {
    constexpr int ROWS = 16;
    constexpr int G = 2, PB = TreeLds<G>::kPB;
    __shared__ __attribute__((aligned(16))) float sTreeRows[FUSE ? ROWS * TreeLds<G>::kRow : 4];
    __shared__ float sOut[ROWS][4];                  // reward, value, the two logits: the tree step's inputs
    __shared__ __attribute__((aligned(16))) _Float16 sBuf[2][ROWS * kLdW];      // split activation tiles
    __shared__ __attribute__((aligned(16))) float sV[ROWS * kLd];                // the 101 value logits, f32
    __shared__ float sPart[3][8][ROWS];              // reward / policy 0 / policy 1 partial dots of the eight waves
    __shared__ long sRowOff[ROWS];
    __shared__ float sAct[ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n16 = lane & 15, g = lane >> 4;
    const int lcol = wave * 16 + 4 * g;                  // this lane's four output columns, of game n16
    const int row0 = blockIdx.x * ROWS;
    _Float16* sX = sBuf[0];
    _Float16* sH = sBuf[1];
    // the tree step's lanes: wave 0, lanes 0..31 = 16 games x 2; the others come along inactive, outside every tree
    const int tree = (FUSE && tid < ROWS * G) ? row0 + tid / G : p.T, gl = tid % G;
    bool tree_active = false;
    BackupPrefetchT<PB> pf;
    if (FUSE && wave == 0) {
        tree_active = tree < p.T && p.n_root[tree] > 0;
        pf = backprop_prefetch<G, PB>(p, tree, tree_active, gl);              // lands while the network computes
    }
    // Every bias and head weight this thread will need, requested first: vmcnt is in-order, so a load issued in an
    // epilogue would wait for the weight fragments of the stage after next that are requested ahead of it.
    auto f4 = [&](const float* ptr) { return *reinterpret_cast<const float4*>(ptr + lcol); };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 bias_hidden = f4(INITIAL ? w.rep_dense_b : w.dyn_dense_b);
    const float4 bias_conv = INITIAL ? zero4 : f4(w.dyn_conv_b), plane = INITIAL ? zero4 : f4(w.dyn_plane);
    const float4 bias_rdense = INITIAL ? zero4 : f4(w.dyn_rdense_b), rw = INITIAL ? zero4 : f4(w.dyn_reward_w);
    const float4 bias_p1 = f4(w.pred_dense1_b), bias_p2 = f4(w.pred_dense2_b), bias_value = f4(w.pred_value_b);
    const float4 pw0 = f4(w.pred_policy_w), pw1 = f4(w.pred_policy_w + 128);
    const float policy_b0 = w.pred_policy_b[0], policy_b1 = w.pred_policy_b[1];
    BFrag16 b0, b1;
    load_b16(b0, INITIAL ? w.rep_dense_s : w.dyn_conv_s, lane, wave);
    if (tid < ROWS) {
        const int b = row0 + tid;
        long off = -1;
        float a = 0.f;
        if (b < B) {
            if (INITIAL) {
                off = (long)b * 4;
            } else {
                const int pb = parent_block ? parent_block[b] : 0;
                if (pb >= 0) off = (long)pb * block_stride + (long)b * 128;
                a = action[b] == 1 ? 1.f : 0.f;
            }
        }
        sRowOff[tid] = off;
        sAct[tid] = a;
    }
    __syncthreads();
    const bool lrow_ok = sRowOff[n16] >= 0;
    // relu(acc + bias) of this lane's four columns (v), stored as a split row piece
    auto act4 = [&](const f32x4& acc, const float4& bias, float (&v)[4]) {
        v[0] = fmaxf(acc[0] + bias.x, 0.f);
        v[1] = fmaxf(acc[1] + bias.y, 0.f);
        v[2] = fmaxf(acc[2] + bias.z, 0.f);
        v[3] = fmaxf(acc[3] + bias.w, 0.f);
    };
    float vmax = 0.f;
    auto store4 = [&](const float (&v)[4], _Float16* sOut) {
        uint2 h, l;
        split4(v, h, l, vmax);
        *reinterpret_cast<uint2*>(sOut + n16 * kLdW + lcol) = h;
        *reinterpret_cast<uint2*>(sOut + n16 * kLdW + 128 + lcol) = l;
    };
    auto dot4 = [&](const float (&v)[4], const float4& wv) {
        return __builtin_fmaf(v[3], wv.w, __builtin_fmaf(v[2], wv.z, __builtin_fmaf(v[1], wv.y, v[0] * wv.x)));
    };
    f32x4 acc;
    float v[4];
    if (INITIAL) {
        // scalar_to_support two-hot (utils.py:138-143, network.py:107-116) -> Conv1D k=4 valid 101 -> 128 relu
        // (games/cartpole.py:98): only <= 8 of the 404 inputs are non-zero, gathered on the VALU.
        const int nn = tid & 127;
        const float lo[4] = {-4.8f, -5.f, -0.418f, -5.f}, hi[4] = {4.8f, 5.f, 0.418f, 5.f};
        const float bn = w.rep_conv_b[nn];
        for (int r = tid >> 7; r < ROWS; r += 4) {
            float s = bn;
            if (sRowOff[r] >= 0) {
#pragma unroll
                for (int pos = 0; pos < 4; ++pos) {
                    const float x = (in_base[sRowOff[r] + pos] - lo[pos]) / (hi[pos] - lo[pos]);
                    const float y = 100.f * (sqrtf(x + 1.f) - 1.f) / (sqrtf(2.f) - 1.f);
                    const float fl = floorf(y);
                    const float pf = y - fl;
                    const int il = (int)fl, ih = (int)ceilf(y);
                    // tf.one_hot yields an all-zero row for an out-of-range index
                    if (ih >= 0 && ih <= 100) s += pf * w.rep_conv[(pos * 101 + ih) * 128 + nn];
                    if (il >= 0 && il <= 100) s += (1.f - pf) * w.rep_conv[(pos * 101 + il) * 128 + nn];
                }
            }
            _Float16 h, l;
            vmax = fmaxf(vmax, s);
            split_f16(fmaxf(s, 0.f), h, l);
            sX[r * kLdW + nn] = h;
            sX[r * kLdW + 128 + nn] = l;
        }
        load_b16(b1, w.pred_dense1_s, lane, wave);
        __syncthreads();
        acc = mma16(sX, b0, lane);                                            // Dense relu = hidden (:101)
    } else {
        {   // gather leaf.parent.hidden_state rows (agent.py:204) into LDS: 16 rows x 32 pieces of four floats
            const int r = tid >> 5, c4 = tid & 31;
            const long off = sRowOff[r];
            const float4 x = off >= 0 ? *reinterpret_cast<const float4*>(in_base + off + c4 * 4) : zero4;
            const float xv[4] = {x.x, x.y, x.z, x.w};
            uint2 h, l;
            split4(xv, h, l, vmax);
            *reinterpret_cast<uint2*>(sH + r * kLdW + c4 * 4) = h;
            *reinterpret_cast<uint2*>(sH + r * kLdW + 128 + c4 * 4) = l;
        }
        load_b16(b1, w.dyn_dense_s, lane, wave);
        __syncthreads();
        // Conv1D k=2 valid over [hidden ; action plane] (games/cartpole.py:106): hidden @ W[0] + a * sum_k W[1][k]
        acc = mma16(sH, b0, lane);
        {
            const float a = sAct[n16];
            const float4 bc = make_float4(bias_conv.x + a * plane.x, bias_conv.y + a * plane.y, bias_conv.z + a * plane.z,
                                          bias_conv.w + a * plane.w);
            act4(acc, bc, v);
            store4(v, sX);
        }
        load_b16(b0, w.dyn_rdense_s, lane, wave);
        __syncthreads();
        acc = mma16(sX, b1, lane);                                            // Dense relu = hidden' (:108)
        load_b16(b1, w.pred_dense1_s, lane, wave);
    }
    // hidden state out (both modes); sH's previous contents (the gathered input) are no longer read
    act4(acc, bias_hidden, v);
    store4(v, sH);
    if (lrow_ok) *reinterpret_cast<float4*>(out_hidden + (size_t)(row0 + n16) * 128 + lcol) = make_float4(v[0], v[1], v[2], v[3]);
    __syncthreads();
    if (!INITIAL) {
        // reward branch: Dense 128 relu, Dense 1 sigmoid (games/cartpole.py:111-114); the dot from the accumulators
        acc = mma16(sH, b0, lane);
        act4(acc, bias_rdense, v);
        const float part = sum_over_g(dot4(v, rw));
        if (g == 0) sPart[0][wave][n16] = part;
    }
    // prediction: Dense relu, Dense relu (games/cartpole.py:121-124); b1 = pred_dense1 in both modes
    load_b16(b0, w.pred_dense2_s, lane, wave);
    acc = mma16(sH, b1, lane);
    act4(acc, bias_p1, v);
    store4(v, sX);
    load_b16(b1, w.pred_value_s, lane, wave);
    __syncthreads();                                     // sX complete, sH no longer read, sPart[0] complete
    if (!INITIAL && tid < ROWS && sRowOff[tid] >= 0) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += sPart[0][k][tid];
        const float r = 1.f / (1.f + expf(-(s + w.dyn_reward_b)));
        out_reward[row0 + tid] = r;
        sOut[tid][0] = r;
    }
    acc = mma16(sX, b0, lane);
    act4(acc, bias_p2, v);
    store4(v, sH);
    {   // policy logits: Dense 2 (:127), from the accumulators
        const float p0 = sum_over_g(dot4(v, pw0)), p1 = sum_over_g(dot4(v, pw1));
        if (g == 0) {
            sPart[1][wave][n16] = p0;
            sPart[2][wave][n16] = p1;
        }
    }
    __syncthreads();
    // value head: Dense 101 (columns padded to 128) -> softmax -> support_to_scalar (:125, utils.py:146-148)
    acc = mma16(sH, b1, lane);
    *reinterpret_cast<float4*>(sV + n16 * kLd + lcol) =
        make_float4(acc[0] + bias_value.x, acc[1] + bias_value.y, acc[2] + bias_value.z, acc[3] + bias_value.w);
    __syncthreads();
    const int row = (tid >> 4) & 15, q = tid & 15, b = row0 + row;        // 16 threads per row (threads 0..255)
    const bool rowthread = tid < ROWS * 16;
    const bool live = rowthread && sRowOff[row] >= 0;
    if (rowthread) {
        float* z = sV + row * kLd;
        float mx = -INFINITY;
        for (int j = q; j < 101; j += 16) mx = fmaxf(mx, z[j]);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
        float den = 0.f;
        for (int j = q; j < 101; j += 16) {
            const float e = expf(z[j] - mx);
            z[j] = e;
            den += e;
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) den += __shfl_xor(den, o, 16);
        float y = 0.f;
        for (int j = q; j < 101; j += 16) {
            const float pj = z[j] / den;
            if (out_value_raw && live) out_value_raw[(size_t)b * 101 + j] = pj;
            y += pj * (float)j;
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) y += __shfl_xor(y, o, 16);
        if (q == 0 && live) {
            y = y / 100.f;
            const float t = (sqrtf(2.f) - 1.f) * y + 1.f;
            const float val = 0.f + (100.f - 0.f) * (t * t - 1.f);     // config.py:38-40 with bounds (0,100)
            out_value[b] = val;
            sOut[row][1] = val;
        }
        if (q < 2 && live) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += sPart[1 + q][k][row];
            const float logit = s + (q == 0 ? policy_b0 : policy_b1);
            out_logits[(size_t)b * 2 + q] = logit;
            sOut[row][2 + q] = logit;
        }
    }
    note_range(vmax, w.range_flag);
    if (FUSE) {
        __syncthreads();
        if (wave == 0) {      // tree step of these 16 games (agent.py:208-211, then 177-190 for the next simulation)
            const float* mine = sOut[tid < ROWS * G ? tid / G : 0];
            float* lds_row = sTreeRows + (tid < ROWS * G ? tid / G : 0) * TreeLds<G>::kRow;
            backprop_tree<G, true, PB>(p, tree, tree_active, gl, nb, mine + 2, mine[0], mine[1], pf, lds_row);
            if (FUSE == 2 && do_select) {
                // a tree's two lanes lie inside this wavefront, whose vector memory operations are performed in program
                // order: compiler ordering is enough (step_kernel, mz_tree.hip)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                select_tree<G>(p, tree, tree_active, gl, parent_block, action);
            }
        }
    }
}

template <bool INITIAL>
__global__ __launch_bounds__(kThreads16) void cartpole16_kernel(CartDev w, int B, const float* __restrict__ in_base,
                                                                const int* __restrict__ parent_block, long block_stride,
                                                                const int* __restrict__ action,
                                                                float* __restrict__ out_hidden, float* __restrict__ out_reward,
                                                                float* __restrict__ out_value, float* __restrict__ out_logits,
                                                                float* __restrict__ out_value_raw) {
    cartpole16_body<INITIAL, 0>(w, B, in_base, const_cast<int*>(parent_block), block_stride, const_cast<int*>(action),
                                out_hidden, out_reward, out_value, out_logits, out_value_raw, PoolDev{}, 0);
}

template <bool SELECT>
__global__ __launch_bounds__(kThreads16) void cartpole16_step_kernel(CartDev w, PoolDev p, int nb,
                                                                     const float* __restrict__ in_base, int* parent_block,
                                                                     long block_stride, int* action,
                                                                     float* __restrict__ out_hidden,
                                                                     float* __restrict__ out_reward,
                                                                     float* __restrict__ out_value,
                                                                     float* __restrict__ out_logits) {
    cartpole16_body<false, SELECT ? 2 : 1>(w, p.T, in_base, parent_block, block_stride, action, out_hidden, out_reward,
                                           out_value, out_logits, nullptr, p, nb);
}

// run_mcts (agent.py:198-211) of 16 games in ONE launch: `nsim` simulations -- recurrent inference, expand_node,
// backpropagate, select_leaf of the next one -- looped inside the kernel.  A workgroup owns its 16 games from the first
// simulation to the last, so nothing crosses workgroups and nothing a simulation needs waits for a launch:
//   * the 16 trees live in LDS for the whole run (records packed to 80 bytes: Slot[2] + BackSlot[2]; per-tree root
//     statistics, min/max, path; the log table): copied in at the start, copied back to the pool at the end.  The tree
//     code is mz_tree_dev.h's, unchanged, run on a PoolDev whose pointers address the LDS copy -- a descent level costs
//     an LDS round trip instead of an L2 one;
//   * the weights of the first four stages (conv, dense, reward dense, prediction dense 1: 128 registers a lane) stay
//     in registers across all simulations; the last two stages' stream in while the first four compute;
//   * biases and head weights sit in LDS; the hidden rows a simulation gathers were written by this workgroup
//     (workgroup-scope release / acquire of __syncthreads).
// Simulation i takes record nb0 + i and writes hidden block nb0 + i; its outputs go to out_* + i * out_stride rows
// (out_stride = 0: every simulation overwrites the same T rows), its selection to trace_parent / trace_action [i][T] when
// those are set.  The last simulation selects only if `final_select`.  Same arithmetic, stage by stage, as
// cartpole16_body: same results.
// All-lanes reduction over the 16 lanes of a DPP row with a commutative operator: the xor-butterfly's operand pairs
// (steps 1 and 2 quad permutes; 4 and 8 the half-row / row mirrors, which pair the same values once quads / half rows are
// uniform), hence the bits of the __shfl_xor loop, without its four LDS-crossbar round trips.
template <typename Op>
__device__ __forceinline__ float row16_allreduce(float x, Op op) {
    x = op(x, dpp_f32<kDppXor1>(x));
    x = op(x, dpp_f32<kDppXor2>(x));
    x = op(x, dpp_f32<kDppHalfMirror>(x));
    x = op(x, dpp_f32<kDppMirror>(x));
    return x;
}
// workgroup barrier that orders LDS only: a wave's outstanding global stores are not waited for (__syncthreads does)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
constexpr int kCartRunRS = 80, kCartRunOffBack = 48;
struct CartRunLayout {          // byte offsets inside the dynamic LDS block
    int recs, root_prior, root_vsum, mm_min, mm_max, sum_len, sims, path, log, xtab, ints, total;
};
__host__ __device__ inline CartRunLayout cart_run_layout(int nrec, int N) {
    CartRunLayout L;
    int o = 0;
    L.recs = o;        o += 16 * nrec * kCartRunRS;
    L.root_prior = o;  o += 16 * 2 * 8;
    L.root_vsum = o;   o += 16 * 8;
    L.mm_min = o;      o += 16 * 8;
    L.mm_max = o;      o += 16 * 8;
    L.sum_len = o;     o += 16 * 8;
    L.sims = o;        o += 16 * 8;
    L.path = o;        o += 16 * (N + 1) * 8;
    L.log = o;         o += (N + 2) * 8;
    L.xtab = o;        o += (N + 2) * (N + 1) * 8;     // exploration_function(Np, Nc) for every pair of counts
    L.ints = o;        o += 5 * 16 * 4;          // n_root, root_count, path_len, parent, action
    L.total = (o + 15) & ~15;
    return L;
}
// select_leaf (agent.py:177-190) for a tree with TWO actions held in LDS, by the four lanes of a DPP quad: select_tree /
// scan_level of mz_tree_dev.h with what two children leave of them.  A node's prefix is always both children (hi + 2 >= 2),
// lanes 0 and 1 score one child each, every lane of the quad learns both scores through quad broadcasts and takes the
// same decision: first child if either score is NaN (trap T2: a NaN first child wins, a NaN second child never does), else
// the larger score, ties to the larger (action, slot) key (trap T1).  The exploration factor 2 log(Np + 1) / (Nc + 1)
// (config.py:159-160) comes from `xtab`, filled in the kernel's prologue by the same IEEE division, so a level is an LDS
// read of the slot, one of the table, a multiply-add and the exchange.  Same path entries, hi words and outputs as
// select_tree, bit for bit (tests/test_selfplay_gpu.py, tests/test_fullsize_gpu.py).
template <int LANE>
__device__ __forceinline__ int quad_bcast(int x) {
    return __builtin_amdgcn_mov_dpp(x, LANE * 0x55, 0xf, 0xf, true);      // quad_perm:[LANE, LANE, LANE, LANE]
}
template <int LANE>
__device__ __forceinline__ double quad_bcast(double x) {
    return __hiloint2double(quad_bcast<LANE>(__double2hiint(x)), quad_bcast<LANE>(__double2loint(x)));
}
__device__ __forceinline__ void cart2_select(const PoolDev& q, const double* xtab, int lt, bool active, int gl,
                                             int* out_parent, int* out_action) {
    const int N1 = q.N + 1;
    int e = 0, np_ = 0, target = 0, depth = 0, hi = 0;
    int* hi_home = nullptr;
    int hi_word = 0;
    bool done = !active;
    bool ident = false;
    double na = q.kmin, nd = q.kmax - q.kmin;
    if (active) {
        np_ = q.root_count[lt];
        target = q.n_root[lt];
        if (!q.has_bounds) {
            const double mn = q.mm_min[lt], mx = q.mm_max[lt];
            ident = (mx == mn);
            na = mn;
            nd = mx - mn;
        }
    }
    const double dflt = ident ? q.default_value : (q.default_value - na) / nd;      // utils.py:68-76 on the default value
    const bool renorm = !q.has_bounds && !ident;
    for (int it = 0; it <= q.N; ++it) {          // a path holds at most N+1 nodes: the wave reaches the exit
        if (!__any(!done)) break;
        const bool root = (it == 0);
        Slot* slots = f_slots(q, rec(q, lt, done ? 0 : e));
        double sc = 0.0;
        int c = 0, ch = 0, act = 0;
        if (!done && gl < target) {
            const Slot s = slots[gl];
            c = s.count;
            ch = s.child_hi;
            act = s.act;
            const double pr = root ? q.root_prior[lt * 2 + gl] : (double)s.prior;
            double x = dflt;
            if (c > 0) x = renorm ? (s.q - na) / nd : s.q;
            sc = x + pr * xtab[np_ * N1 + c];                           // agent.py:189-190
        }
        const double sc0 = quad_bcast<0>(sc), sc1 = quad_bcast<1>(sc);
        const int c0 = quad_bcast<0>(c), c1 = quad_bcast<1>(c);
        const int ch0 = quad_bcast<0>(ch), ch1 = quad_bcast<1>(ch);
        const int act0 = quad_bcast<0>(act), act1 = quad_bcast<1>(act);
        // (action << 16 | slot) keys: slot 1's is larger iff act1 >= act0
        const bool both = (target >= 2) & (sc0 == sc0) & (sc1 == sc1);
        const int k = (both & ((sc1 > sc0) | ((sc1 == sc0) & (act1 >= act0)))) ? 1 : 0;
        const int wc = k ? c1 : c0, wch = k ? ch1 : ch0, wact = k ? act1 : act0;
        if (!done) {
            if (gl == 0) q.path[lt * N1 + depth] = make_int2((e << kPathShift) | k, wc);
            if (wc > 0) {                    // visited <=> expanded: descend (agent.py:179-180)
                hi_home = &slots[k].child_hi;
                hi_word = wch;
                np_ = wc;
                e = wch & kChildMask;
                hi = (int)((unsigned)wch >> kChildBits);
                target = min(hi + 2, 2);
                ++depth;
            } else {
                done = true;
                if (gl == 0) {
                    out_parent[lt] = e;
                    out_action[lt] = wact;
                    q.path_len[lt] = depth + 1;
                    if (!root && k + 1 > hi) *hi_home = (min(k + 1, kMaxActions - 1) << kChildBits) | (hi_word & kChildMask);
                }
            }
        }
    }
    if (active && !done && gl == 0) {        // unreachable for a consistent pool; never leave outputs unset
        out_parent[lt] = -1;
        out_action[lt] = -1;
        q.path_len[lt] = 0;
        atomicOr(q.err_flag, 1);
    }
    if (!active && gl == 0) {
        out_parent[lt] = -1;
        out_action[lt] = -1;
    }
}

template <int LANE>
__device__ __forceinline__ float quad_bcast(float x) {
    return __int_as_float(quad_bcast<LANE>(__float_as_int(x)));
}
// expand_node + backpropagate (agent.py:154-160, 192-196) for the same trees and lanes: backprop_tree<G, true> of
// mz_tree_dev.h for two actions.  The new record's two slots are (count 0, prior, action) in action order -- with two
// actions the ranking by prior leaves slot 1 where it is and hi is 1 either way; masked_policy's fp32 sum of two terms is
// (0 + e0) + e1.  The path is backed up four positions a round from the leaf down: each lane fetches the entry, slot
// and reward of one position, the four rewards go round the quad by DPP, every lane replays the fp64 recurrence
// value = reward + discount * value (agent.py:195, separately rounded) and keeps the value that reached its position,
// then updates its slot (backup_slot: the same code).  Same records, root statistics, min/max and counters, bit for bit.
__device__ __forceinline__ void cart2_backprop(const PoolDev& q, int lt, bool active, int gl, int nb, float logit0,
                                               float logit1, float rnew, float value_f32) {
    const int N1 = q.N + 1;
    const int len = active ? q.path_len[lt] : 0;
    {   // masked_policy over both actions (network.py:142-146): exp, NumPy-ordered fp32 sum, IEEE divide
        const float e0 = expf(logit0), e1 = expf(logit1);
        const float total = (0.f + e0) + e1;
        const float pr = (gl == 0 ? e0 : e1) / total;
        if (active && gl < 2) {
            Slot* s = f_slots(q, rec(q, lt, nb)) + gl;
            s->count = 0;
            s->prior = pr;
            s->act = gl;
        }
    }
    double v = active ? (double)value_f32 : 0.0;
    const int top = len - 1;
    double mn_l = std::numeric_limits<double>::infinity();
    double mx_l = -std::numeric_limits<double>::infinity();
    for (int base = 0; base <= q.N; base += 4) {          // bounded: a path holds at most N+1 nodes
        if (!__any(base < len)) break;
        const int pos = top - base - gl;
        const bool has = active && base < len && pos >= 0;
        int c = 0, k = 0;
        double vs = 0.0;
        float rw = 0.f;
        char* r = rec(q, lt, 0);
        if (has) {
            const int2 pe = q.path[lt * N1 + pos];
            k = pe.x & (kMaxActions - 1);
            r = rec(q, lt, pe.x >> kPathShift);
            c = pe.y;
            const BackSlot b = f_back(q, r)[k];
            vs = b.vsum;
            rw = (pos == top) ? rnew : b.reward;
        }
        const float r0 = quad_bcast<0>(rw), r1 = quad_bcast<1>(rw), r2 = quad_bcast<2>(rw), r3 = quad_bcast<3>(rw);
        const int n_here = base < len ? min(4, len - base) : 0;
        double vin = v;
        // lane j of the round owns position top - base - j: it keeps the value that reaches it, then the value moves on
        if (0 < n_here) v = (double)r0 + q.discount * v;
        if (gl == 1) vin = v;
        if (1 < n_here) v = (double)r1 + q.discount * v;
        if (gl == 2) vin = v;
        if (2 < n_here) v = (double)r2 + q.discount * v;
        if (gl == 3) vin = v;
        if (3 < n_here) v = (double)r3 + q.discount * v;
        if (has) backup_slot(q, r, k, c, vs, vin, rw, pos == top, nb, 1, mn_l, mx_l);
    }
    if (active && gl == 0) {                          // the root (reward None -> chain ends, agent.py:195)
        const int rc = q.root_count[lt] + 1;
        const double rvs = q.root_vsum[lt] + v;
        q.root_count[lt] = rc;
        q.root_vsum[lt] = rvs;
        if (!q.has_bounds) {                          // MinMaxStats.update is a no-op under known bounds (utils.py:63-66)
            const double mean = rvs / (double)rc;
            if (mean > mx_l) mx_l = mean;
            if (mean < mn_l) mn_l = mean;
        }
        q.sum_len[lt] += len;
        q.sims[lt] += 1;
    }
    if (!q.has_bounds) {
        mx_l = group_minmax_f64<4, true>(mx_l);
        mn_l = group_minmax_f64<4, false>(mn_l);
        if (active && gl == 0) {
            if (mx_l > q.mm_max[lt]) q.mm_max[lt] = mx_l;
            if (mn_l < q.mm_min[lt]) q.mm_min[lt] = mn_l;
        }
    }
}

constexpr int kCartRunMaxDynLds = 120 * 1024;    // beside ~36 KB of static LDS

constexpr int kRunThreads = 512;      // eight wavefronts, 16 output columns each: 256 registers a lane, 192 of them weights
__global__ __launch_bounds__(kRunThreads) void cartpole16_run_kernel(CartDev w, PoolDev p, int nb0, int nsim, int final_select,
                                                                    float* hidden_base, int* parent_block, long block_stride,
                                                                    int* action, float* out_reward, float* out_value,
                                                                    float* out_logits, long out_stride, int* trace_parent,
                                                                    int* trace_action) {
    constexpr int ROWS = 16;
    constexpr int G = 4;                             // lanes per tree HERE (results do not depend on it): wave 0 = 16 DPP quads
    extern __shared__ __attribute__((aligned(16))) char sTree[];
    __shared__ float sOut[ROWS][4];                  // reward, value, the two logits: the tree step's inputs
    __shared__ __attribute__((aligned(16))) _Float16 sBuf[2][ROWS * kLdW];      // split activation tiles
    __shared__ __attribute__((aligned(16))) float sV[ROWS * kLd];                // the 101 value logits, f32
    __shared__ float sPart[3][8][ROWS];              // reward / policy 0 / policy 1 partial dots of the eight column blocks
    __shared__ long sRowOff[ROWS];
    __shared__ float sAct[ROWS];
    // rows: 0 conv bias, 1 action plane, 2 hidden dense bias, 3 reward dense bias, 4 reward weights, 5 / 6 prediction dense
    // biases, 7 value bias, 8 / 9 policy weights
    __shared__ __attribute__((aligned(16))) float sBias[10 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * ROWS;
    const int nrec = nb0 + nsim;
    const CartRunLayout L = cart_run_layout(nrec, p.N);
    _Float16* sX = sBuf[0];
    _Float16* sH = sBuf[1];

    // ---- the pool as this workgroup sees it: 16 trees in LDS
    PoolDev q = p;
    q.T = ROWS;
    q.RS = kCartRunRS;
    q.off_back = kCartRunOffBack;
    q.tree_stride = (size_t)nrec * kCartRunRS;
    q.blocks = sTree + L.recs;
    q.root_prior = reinterpret_cast<double*>(sTree + L.root_prior);
    q.root_vsum = reinterpret_cast<double*>(sTree + L.root_vsum);
    q.mm_min = reinterpret_cast<double*>(sTree + L.mm_min);
    q.mm_max = reinterpret_cast<double*>(sTree + L.mm_max);
    q.sum_len = reinterpret_cast<long long*>(sTree + L.sum_len);
    q.sims = reinterpret_cast<long long*>(sTree + L.sims);
    q.path = reinterpret_cast<int2*>(sTree + L.path);
    int* sInts = reinterpret_cast<int*>(sTree + L.ints);
    q.n_root = sInts;
    q.root_count = sInts + 16;
    q.path_len = sInts + 32;
    int* sParent = sInts + 48;
    int* sAction = sInts + 64;
    double* sLog = reinterpret_cast<double*>(sTree + L.log);
    double* sXtab = reinterpret_cast<double*>(sTree + L.xtab);

    // the weights, resident for the whole run: six stages x this wavefront's 16-column block x 32 registers
    BFrag16 wc, wd, wr, w1, w2, wv;
    load_b16(wc, w.dyn_conv_s, lane, wave);
    load_b16(wd, w.dyn_dense_s, lane, wave);
    load_b16(wr, w.dyn_rdense_s, lane, wave);
    load_b16(w1, w.pred_dense1_s, lane, wave);
    load_b16(w2, w.pred_dense2_s, lane, wave);
    load_b16(wv, w.pred_value_s, lane, wave);
    {   // records [0, nb0) of the 16 trees, 16 bytes a thread and step; the records to come start zeroed
        const int per_tree = nrec * 5;
        for (int idx = tid; idx < ROWS * per_tree; idx += kRunThreads) {
            const int t = idx / per_tree, rem = idx - t * per_tree, e = rem / 5, c = rem - e * 5;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row0 + t < p.T && e < nb0)
                v = *reinterpret_cast<const float4*>(rec(p, row0 + t, e) + (c < 3 ? c * 16 : p.off_back + (c - 3) * 16));
            *reinterpret_cast<float4*>(sTree + L.recs + (size_t)t * q.tree_stride + e * kCartRunRS + c * 16) = v;
        }
        for (int idx = tid; idx < ROWS * (p.N + 1); idx += kRunThreads) {
            const int t = idx / (p.N + 1), d = idx - t * (p.N + 1);
            q.path[idx] = row0 + t < p.T ? p.path[(size_t)(row0 + t) * (p.N + 1) + d] : make_int2(0, 0);
        }
        for (int idx = tid; idx < p.N + 2; idx += kRunThreads) sLog[idx] = p.log_table[idx];
        for (int idx = tid; idx < (p.N + 2) * (p.N + 1); idx += kRunThreads) {      // config.py:159-160, as scan_level divides
            const int np_ = idx / (p.N + 1), c = idx - np_ * (p.N + 1);
            sXtab[idx] = p.log_table[np_] / (double)(c + 1);
        }
        if (tid < ROWS) {
            const int t = row0 + tid;
            const bool in = t < p.T;
            q.n_root[tid] = in ? p.n_root[t] : 0;
            q.root_count[tid] = in ? p.root_count[t] : 0;
            q.path_len[tid] = in ? p.path_len[t] : 0;
            sParent[tid] = in ? parent_block[t] : -1;
            sAction[tid] = in ? action[t] : -1;
            q.root_vsum[tid] = in ? p.root_vsum[t] : 0.0;
            q.mm_min[tid] = in ? p.mm_min[t] : 0.0;
            q.mm_max[tid] = in ? p.mm_max[t] : 0.0;
            q.root_prior[2 * tid] = in ? p.root_prior[(size_t)t * 2] : 0.0;
            q.root_prior[2 * tid + 1] = in ? p.root_prior[(size_t)t * 2 + 1] : 0.0;
            q.sum_len[tid] = 0;
            q.sims[tid] = 0;
        }
        for (int idx = tid; idx < 10 * 128; idx += kRunThreads) {
            const int r = idx >> 7, c = idx & 127;
            const float* src = r == 0   ? w.dyn_conv_b
                               : r == 1 ? w.dyn_plane
                               : r == 2 ? w.dyn_dense_b
                               : r == 3 ? w.dyn_rdense_b
                               : r == 4 ? w.dyn_reward_w
                               : r == 5 ? w.pred_dense1_b
                               : r == 6 ? w.pred_dense2_b
                               : r == 7 ? w.pred_value_b
                                        : w.pred_policy_w + (r - 8) * 128;
            sBias[idx] = src[c];
        }
    }
    const float policy_b0 = w.pred_policy_b[0], policy_b1 = w.pred_policy_b[1];
    __syncthreads();
    MZ_TRACE_DECL;
    MZ_TRACE(0);                                         // prologue: weights, trees and tables in place

    float vmax = 0.f;

#pragma unroll 1
    for (int i = 0; i < nsim; ++i) {
        asm volatile("" ::: "memory");                   // nothing below is hoisted out of the loop (registers)
        // Every per-lane address and mask below derives from a copy of the thread index the compiler cannot see through:
        // hoisted out of the loop they would have to live beside 384 registers of weights (they spilled).
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6, n16 = lane & 15, g = lane >> 4;
        const int lcol0 = wave * 16 + 4 * g;             // this lane's four output columns of game n16
        // the tree step's lanes: wavefront 0 = 16 DPP quads = the 16 trees (spread over four wavefronts, four trees each,
        // the step was slower: 104 against 110 M simulations/s, measured on round 2's four-wavefront workgroup)
        const bool tree_lane = wave == 0;
        const int ltree = (lane >> 2) & 15, gl = lane & (G - 1);
        const bool tree_active = tree_lane && row0 + ltree < p.T && q.n_root[ltree] > 0;
        auto bias4 = [&](int r) { return *reinterpret_cast<const float4*>(sBias + r * 128 + lcol0); };
        auto act4 = [&](const f32x4& acc, const float4& bias, float (&v)[4]) {
            v[0] = fmaxf(acc[0] + bias.x, 0.f);
            v[1] = fmaxf(acc[1] + bias.y, 0.f);
            v[2] = fmaxf(acc[2] + bias.z, 0.f);
            v[3] = fmaxf(acc[3] + bias.w, 0.f);
        };
        auto store4 = [&](const float (&v)[4], _Float16* sOutT) {
            uint2 h, l;
            split4(v, h, l, vmax);
            *reinterpret_cast<uint2*>(sOutT + n16 * kLdW + lcol0) = h;
            *reinterpret_cast<uint2*>(sOutT + n16 * kLdW + 128 + lcol0) = l;
        };
        auto dot4 = [&](const float (&v)[4], const float4& wv) {
            return __builtin_fmaf(v[3], wv.w, __builtin_fmaf(v[2], wv.z, __builtin_fmaf(v[1], wv.y, v[0] * wv.x)));
        };
        const int nb = nb0 + i;
        const bool keep = out_stride != 0 || i + 1 == nsim;       // outputs no later simulation overwrites
        if (tid < ROWS) {
            const int b = row0 + tid;
            const int pb = sParent[tid], a = sAction[tid];
            long off = -1;
            if (b < p.T && pb >= 0) off = (long)pb * block_stride + (long)b * 128;
            sRowOff[tid] = off;
            sAct[tid] = a == 1 ? 1.f : 0.f;
            if (trace_parent != nullptr && b < p.T) {
                trace_parent[(size_t)i * p.T + b] = pb;
                trace_action[(size_t)i * p.T + b] = a;
            }
        }
        lds_barrier();
        const bool lrow_ok = sRowOff[n16] >= 0;
        f32x4 acc;
        float v[4];
        // gather leaf.parent.hidden_state rows (agent.py:204) into LDS: 16 rows x 32 pieces of four floats
        {
            const int idx = tid, r = idx >> 5, c4 = idx & 31;
            const long off = sRowOff[r];
            const float4 x = off >= 0 ? *reinterpret_cast<const float4*>(hidden_base + off + c4 * 4)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
            const float xv[4] = {x.x, x.y, x.z, x.w};
            uint2 h, l;
            split4(xv, h, l, vmax);
            *reinterpret_cast<uint2*>(sH + r * kLdW + c4 * 4) = h;
            *reinterpret_cast<uint2*>(sH + r * kLdW + 128 + c4 * 4) = l;
        }
        lds_barrier();
        MZ_TRACE(5);                                     // parent rows gathered
        // Conv1D k=2 valid over [hidden ; action plane] (games/cartpole.py:106): hidden @ W[0] + a * sum_k W[1][k]
        acc = mma16(sH, wc, lane);
        {
            const float a = sAct[n16];
            const float4 bc = bias4(0), pl = bias4(1);
            const float4 bca = make_float4(bc.x + a * pl.x, bc.y + a * pl.y, bc.z + a * pl.z, bc.w + a * pl.w);
            act4(acc, bca, v);
            store4(v, sX);
        }
        lds_barrier();
        acc = mma16(sX, wd, lane);                                            // Dense relu = hidden' (:108)
        {
            act4(acc, bias4(2), v);
            store4(v, sH);                               // sH's previous contents (the gathered input) are no longer read
            if (lrow_ok)
                *reinterpret_cast<float4*>(hidden_base + (size_t)nb * block_stride + (size_t)(row0 + n16) * 128 + lcol0) =
                    make_float4(v[0], v[1], v[2], v[3]);
        }
        lds_barrier();
        // reward branch: Dense 128 relu, Dense 1 sigmoid (games/cartpole.py:111-114); the dot from the accumulators
        acc = mma16(sH, wr, lane);
        {
            act4(acc, bias4(3), v);
            const float part = sum_over_g(dot4(v, bias4(4)));
            if (g == 0) sPart[0][wave][n16] = part;
        }
        // prediction: Dense relu, Dense relu (games/cartpole.py:121-124)
        acc = mma16(sH, w1, lane);
        {
            act4(acc, bias4(5), v);
            store4(v, sX);
        }
        lds_barrier();                                     // sX complete, sH no longer read, sPart[0] complete
        MZ_TRACE(6);                                     // four stages
        if (tid < ROWS && sRowOff[tid] >= 0) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += sPart[0][k][tid];
            const float r = 1.f / (1.f + expf(-(s + w.dyn_reward_b)));
            if (keep) out_reward[(size_t)i * out_stride + row0 + tid] = r;
            sOut[tid][0] = r;
        }
        acc = mma16(sX, w2, lane);
        {
            act4(acc, bias4(6), v);
            store4(v, sH);
            // policy logits: Dense 2 (:127), from the accumulators
            const float p0 = sum_over_g(dot4(v, bias4(8))), p1 = sum_over_g(dot4(v, bias4(9)));
            if (g == 0) {
                sPart[1][wave][n16] = p0;
                sPart[2][wave][n16] = p1;
            }
        }
        lds_barrier();
        // value head: Dense 101 (columns padded to 128) -> softmax -> support_to_scalar (:125, utils.py:146-148)
        acc = mma16(sH, wv, lane);
        {
            const float4 bv = bias4(7);
            *reinterpret_cast<float4*>(sV + n16 * kLd + lcol0) =
                make_float4(acc[0] + bv.x, acc[1] + bv.y, acc[2] + bv.z, acc[3] + bv.w);
        }
        lds_barrier();
        MZ_TRACE(7);                                     // two stages
        const int row = (tid >> 4) & 15, qq = tid & 15, b = row0 + row;        // 16 threads per row (threads 0..255)
        const bool rowthread = tid < ROWS * 16;
        const bool live = rowthread && sRowOff[row] >= 0;
        if (rowthread) {
            // this thread's bins qq, qq + 16, ... (seven, the last one past the 101 for qq >= 5) in registers; the same
            // operations in the same order as cartpole16_body's three passes over LDS
            const float* z = sV + row * kLd;
            float zz[7];
#pragma unroll
            for (int t = 0; t < 7; ++t) zz[t] = z[min(qq + 16 * t, 100)];
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 7; ++t)
                if (qq + 16 * t < 101) mx = fmaxf(mx, zz[t]);
            mx = row16_allreduce(mx, [](float x, float y) { return fmaxf(x, y); });
            float den = 0.f;
#pragma unroll
            for (int t = 0; t < 7; ++t)
                if (qq + 16 * t < 101) {
                    zz[t] = expf(zz[t] - mx);
                    den += zz[t];
                }
            den = row16_allreduce(den, [](float x, float y) { return x + y; });
            float y = 0.f;
#pragma unroll
            for (int t = 0; t < 7; ++t)
                if (qq + 16 * t < 101) {
                    const float pj = zz[t] / den;
                    y += pj * (float)(qq + 16 * t);
                }
            y = row16_allreduce(y, [](float x, float y2) { return x + y2; });
            if (qq == 0 && live) {
                y = y / 100.f;
                const float t = (sqrtf(2.f) - 1.f) * y + 1.f;
                const float val = 0.f + (100.f - 0.f) * (t * t - 1.f);     // config.py:38-40 with bounds (0,100)
                if (keep) out_value[(size_t)i * out_stride + b] = val;
                sOut[row][1] = val;
            }
            if (qq < 2 && live) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) s += sPart[1 + qq][k][row];
                const float logit = s + (qq == 0 ? policy_b0 : policy_b1);
                if (keep) out_logits[((size_t)i * out_stride + b) * 2 + qq] = logit;
                sOut[row][2 + qq] = logit;
            }
        }
        lds_barrier();
        MZ_TRACE(11);                                    // softmax, outputs
        if (tree_lane) {      // tree step of these 16 games (agent.py:208-211, then 177-190 for the next simulation)
            const int lt = ltree;
            const float* mine = sOut[lt];
            cart2_backprop(q, lt, tree_active, gl, nb, mine[2], mine[3], mine[0], mine[1]);
            MZ_TRACE(1);                                 // expand + backup
            if (i + 1 < nsim || final_select != 0) {
                // a tree's four lanes lie inside this wavefront, whose LDS operations are performed in program order
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                cart2_select(q, sXtab, lt, tree_active, gl, sParent, sAction);
            }
            MZ_TRACE(9);                                 // select
        }
        __syncthreads();                                 // also: this simulation's hidden rows are visible to the gathers to come
    }
    note_range(vmax, w.range_flag);

    // ---- the trees go back to the pool
    {
        const int per_tree = nrec * 5;
        for (int idx = tid; idx < ROWS * per_tree; idx += kRunThreads) {
            const int t = idx / per_tree, rem = idx - t * per_tree, e = rem / 5, c = rem - e * 5;
            if (row0 + t < p.T)
                *reinterpret_cast<float4*>(rec(p, row0 + t, e) + (c < 3 ? c * 16 : p.off_back + (c - 3) * 16)) =
                    *reinterpret_cast<const float4*>(sTree + L.recs + (size_t)t * q.tree_stride + e * kCartRunRS + c * 16);
        }
        for (int idx = tid; idx < ROWS * (p.N + 1); idx += kRunThreads) {
            const int t = idx / (p.N + 1), d = idx - t * (p.N + 1);
            if (row0 + t < p.T) p.path[(size_t)(row0 + t) * (p.N + 1) + d] = q.path[idx];
        }
        if (tid < ROWS && row0 + tid < p.T) {
            const int t = row0 + tid;
            p.root_count[t] = q.root_count[tid];
            p.path_len[t] = q.path_len[tid];
            p.root_vsum[t] = q.root_vsum[tid];
            p.mm_min[t] = q.mm_min[tid];
            p.mm_max[t] = q.mm_max[tid];
            p.sum_len[t] += q.sum_len[tid];
            p.sims[t] += q.sims[tid];
            if (final_select != 0) {
                parent_block[t] = sParent[tid];
                action[t] = sAction[tid];
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------- Atari-shaped
// The synthetic network of BASELINE.json configs[3] (18 actions, 256-d hidden state, conv-residual dynamics; defined in
// oracle/net_oracle.py AtariShapedOracle from the reference's blocks: one_hot_tensor_encoder network.py:14-40, the conv +
// dense stacks of games/tictactoe.py:188-213).  Hidden state = (2 positions, 128 channels).  A k=3 `same` Conv1D over two
// positions is, per output position, two 128 x 128 products (position p: taps 1, 2 of inputs 0, 1 for p = 0; taps 0, 1 for
// p = 1); the 18 one-hot action planes are constant per position and collapse into a [action][position][128] bias table.
//
// One 512-thread workgroup runs the WHOLE network for 64 games: nothing but the hidden rows in and the outputs out touches
// HBM, the 24 weight matrices (1.5 MB, the same for every workgroup, L2-resident) stream through an in-place register
// ring one stage ahead, and the activations ping-pong between two pairs of LDS tiles:
//   recurrent: gather -> X;  input conv X -> Y;  2 x { conv a: Y -> X;  conv b (+ residual): X -> Y };  hidden' = Y;
//              prediction conv Y -> X0;  dense X0 -> X1;  heads X1 -> value, logits;   reward = tanh(<hidden', w> + b)
//   initial:   observation -> X;  conv X -> Y;  dense (per position) Y -> X = hidden;  prediction from X.
// 16384 games = 256 workgroups, one per CU.  A stage is the job kernel's (ttt_ring_kernel): wave v owns output columns
// 16v .. 16v+15 of all 64 rows as four 16-row blocks on v_mfma_f32_16x16x32_f16, split-f16 operands; one barrier per
// LAYER (the tiles a layer writes are not the ones it reads).  The residual stream stays in f32 registers (a lane holds
// four columns of four rows per position), so a residual add is exact f32, never a re-read of a split image.
struct AtariDev {
    const float* rec_s;         // recurrent pass: 24 stage matrices (Packer::add_split16) in consumption order:
                                //   5 conv layers x taps [1, 2, 0, 1] | prediction conv taps 0, 1 | dense | heads (32 columns)
    const float* ini_s;         // initial pass: conv taps [1, 2, 0, 1] | dense | dense | prediction conv 0, 1 | dense | heads
    const float* dyn_actbias;   // [18 actions][2 positions][128]: input-conv bias + the action plane's valid taps
    const float* dyn_bias;      // [4][128]: biases of res1a, res1b, res2a, res2b
    const float* dyn_reward_w;  // [2][128]
    const float* rep_conv_b;    // [128]
    const float* rep_dense_b;   // [128]
    const float* pred_conv_b;   // [128]
    const float* pred_dense_b;  // [128]
    const float* pred_head_b;   // [32]: value bias, 18 policy biases, zeros
    float dyn_reward_b;
    int* range_flag;            // see TttDev
};

// One 128 x 128 stage: acc[m] += sA[rows 16m .. 16m+15][128] * W[128][this wave's 16 columns] (weights as the row operand:
// acc[m][i] = column 4 (lane / 16) + i of the wave's 16, row 16 m + lane % 16); acx[m] collects the two cross terms
// (bl*ah + bh*al), which the caller folds in once per accumulation: acc += acx / 2048.  bw holds this stage's fragments
// and is re-loaded in place with the next stage's (wnext, this lane's slice) as the k-steps retire.  The activation
// fragments run through a two-deep ring `af` that CONTINUES across stages: PRIMED says the caller's previous stage already
// requested this stage's first k-step, and next_sA (a tile that is complete NOW -- not one this layer is still writing) is
// the tile whose first k-step this stage requests under its last MFMAs, so that a stage never starts by waiting for LDS.
template <bool LOAD_NEXT, bool PRIMED>
__device__ __forceinline__ void stage64(const _Float16* sA, const _Float16* next_sA, BFragW& bw,
                                        const float4* __restrict__ wnext, f32x4 (&acc)[4], f32x4 (&acx)[4],
                                        float4 (&af)[2][4][2], int n16, int g) {
    const _Float16* ap = sA + n16 * kLdW + 8 * g;
    if (!PRIMED) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            af[0][m][0] = *reinterpret_cast<const float4*>(ap + m * 16 * kLdW);
            af[0][m][1] = *reinterpret_cast<const float4*>(ap + m * 16 * kLdW + 128);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);      // the priming reads come first
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const _Float16* np = s + 1 < 4 ? ap + 32 * (s + 1) : next_sA != nullptr ? next_sA + n16 * kLdW + 8 * g : nullptr;
        if (s + 1 < 4 || next_sA != nullptr) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                af[(s + 1) & 1][m][0] = *reinterpret_cast<const float4*>(np + m * 16 * kLdW);
                af[(s + 1) & 1][m][1] = *reinterpret_cast<const float4*>(np + m * 16 * kLdW + 128);
            }
        }
        const f16x8 bh = __builtin_bit_cast(f16x8, bw.v[2 * s]), bl = __builtin_bit_cast(f16x8, bw.v[2 * s + 1]);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f16x8 ah = __builtin_bit_cast(f16x8, af[s & 1][m][0]), al = __builtin_bit_cast(f16x8, af[s & 1][m][1]);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah, acc[m], 0, 0, 0);
            acx[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah, acx[m], 0, 0, 0);
            acx[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al, acx[m], 0, 0, 0);
        }
        if (LOAD_NEXT) {
            bw.v[2 * s] = wnext[(2 * s) * 512];
            bw.v[2 * s + 1] = wnext[(2 * s + 1) * 512];
        }
        // pin the interleave: the next k-step's eight fragment reads, this k-step's twelve MFMAs, its VMEM reads
        if (s + 1 < 4 || next_sA != nullptr) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        if (LOAD_NEXT) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
    }
}

// acc += acx / 2048: the cross terms of a finished accumulation
// v[m] = relu(x[m] + bias) of this lane's four columns; split and stored into rows 16 m + n16 of an LDS tile
__device__ __forceinline__ void store_tile4(const f32x4 (&v)[4], _Float16* tile, int n16, int lcol, float& vmax) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float x[4] = {v[m][0], v[m][1], v[m][2], v[m][3]};
        uint2 h, l;
        split4(x, h, l, vmax);
        _Float16* row = tile + (16 * m + n16) * kLdW + lcol;
        *reinterpret_cast<uint2*>(row) = h;
        *reinterpret_cast<uint2*>(row + 128) = l;
    }
}

constexpr int kAtariStageFloats = 16384;         // one split-f16 128 x 128 matrix = the bytes of 128 x 128 floats

// FUSE: 0 = the network alone; 1 = + expand and backprop of this simulation for the 64 games (mz_expand_backprop_logits),
// 2 = + select of the next one (mz_step_logits): eight lanes per game, one pass of the 512 threads -- step_kernel's code
// (mz_tree_dev.h; results do not depend on the lanes per tree), its LDS rows and log table laid over the activation
// tiles, which are dead by then.  The backup's inputs are requested at the start and land while the network computes.
// parent_block / action are then read at the start and rewritten at the end, each workgroup its own 64 games.
constexpr int kAtariTreeG = 8, kAtariTreeP = 4;     // 18 children at the root: one round of 4 slots per lane
// The pool descriptor (65 scalars) read from its image in device memory through a pointer the compiler cannot see through:
// what a run kernel holds by value for a whole move crowds the scalar registers of everything in between.
__device__ __forceinline__ PoolDev load_pool_image(const PoolDev* pool_image) {
    struct PoolWords {
        int w[sizeof(PoolDev) / sizeof(int)];
    };
    typedef const __attribute__((address_space(4))) int* PoolImage;
    PoolImage image = (PoolImage) reinterpret_cast<const int*>(pool_image);
    asm volatile("" : "+s"(image));
    PoolWords words;
#pragma unroll
    for (int j = 0; j < (int)(sizeof(PoolDev) / sizeof(int)); ++j) words.w[j] = image[j];
    return __builtin_bit_cast(PoolDev, words);
}
template <bool INITIAL, int FUSE>
__device__ __forceinline__ void atari_body(const AtariDev& w, int B, const float* __restrict__ in_base, int* parent_block,
                                           long block_stride, int* action, float* __restrict__ out_hidden,
                                           float* __restrict__ out_reward, float* __restrict__ out_value,
                                           float* __restrict__ out_logits, float* __restrict__ out_value_raw,
                                           const PoolDev& p_arg, int nb, bool do_select = true, bool keep_out = true,
                                           LdsLog staged_log = nullptr, const PoolDev* p_image = nullptr) {
    constexpr int ROWS = 64;
    constexpr int TG = kAtariTreeG, TPB = TreeLds<TG>::kPB;
    // p_image: read the descriptor where it is used (the start and the tree step), not across the network's stages
    const PoolDev p = p_image != nullptr ? load_pool_image(p_image) : p_arg;
    static_assert(ROWS * TG == kThreadsW, "one pass of the workgroup covers its 64 trees");
    __shared__ __attribute__((aligned(16))) _Float16 sT[4][ROWS * kLdW];      // X0, X1, Y0, Y1
    __shared__ float sRew[8 * ROWS];                 // reward-head partials of the eight waves' column ranges
    __shared__ float sOut[ROWS][20];                 // value, 18 logits, reward
    __shared__ long sRowOff[ROWS];
    __shared__ int sAct[ROWS];
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                    // opaque: inside atari_run_kernel's loop nothing derived from it is hoisted
    const int lane = tid & 63, wave = tid >> 6;
    const int n16 = lane & 15, g = lane >> 4;
    const int lcol = wave * 16 + 4 * g;              // this lane's four columns: lcol .. lcol + 3
    const int row0 = blockIdx.x * ROWS;
    const float4* wlane = reinterpret_cast<const float4*>(INITIAL ? w.ini_s : w.rec_s) + wave * 64 + lane;
    const int tree = FUSE ? row0 + tid / TG : 0, tgl = tid % TG;
    bool tree_active = false;
    BackupPrefetchT<TPB> pf;
    if (FUSE) tree_active = tree < p.T && p.n_root[tree] > 0;
    BFragW bw;
    load_bw(bw, INITIAL ? w.ini_s : w.rec_s, lane, wave);
    float vmax = 0.f;                                // largest magnitude split by this lane (note_range)
    if (tid < ROWS) {
        const int b = row0 + tid;
        long off = -1;
        int a = 0;
        if (b < B) {
            if (INITIAL) {
                off = (long)b * 256;
            } else {
                const int pb = parent_block ? parent_block[b] : 0;
                if (pb >= 0) off = (long)pb * block_stride + (long)b * 256;
                a = action[b];
                if (a < 0 || a > 17) a = 0;
            }
        }
        sRowOff[tid] = off;
        sAct[tid] = a;
    }
    __syncthreads();
    {   // the 64 input rows (leaf.parent.hidden_state, agent.py:204, or the observations), 256 floats each -> X0 | X1
        float4 in[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + kThreadsW * i, r = idx >> 6, c4 = idx & 63;
            const long off = sRowOff[r];
            in[i] = off >= 0 ? *reinterpret_cast<const float4*>(in_base + off + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + kThreadsW * i, r = idx >> 6, c4 = idx & 63;
            const float x[4] = {in[i].x, in[i].y, in[i].z, in[i].w};
            uint2 h, l;
            split4(x, h, l, vmax);
            _Float16* row = sT[c4 >> 5] + r * kLdW + (c4 & 31) * 4;
            *reinterpret_cast<uint2*>(row) = h;
            *reinterpret_cast<uint2*>(row + 128) = l;
        }
    }
    MZ_TRACE_DECL;
    const unsigned long long rowmask = __ballot(sRowOff[lane] >= 0);     // rows with an output; this lane's: n16 + 16 m
    // the residual stream x[position][row block]: first the input conv's per-row bias (conv bias + action planes, the
    // one-hot encoder's planes being constant per position: network.py:14-40), then the layer outputs
    f32x4 res[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float4 bq = INITIAL ? *reinterpret_cast<const float4*>(w.rep_conv_b + lcol)
                                      : *reinterpret_cast<const float4*>(w.dyn_actbias + (sAct[16 * m + n16] * 2 + p) * 128 + lcol);
            res[p][m] = f32x4{bq.x, bq.y, bq.z, bq.w};
        }
    __syncthreads();
    MZ_TRACE_NW(0);                                  // prologue: gather, split, barrier

    int stage = 0;                                   // index of the matrix bw holds; the ring loads stage + 1
    f32x4 acc[4], acx[4];
    float4 af[2][4][2];                              // the activation-fragment ring (stage64)
    constexpr int kSF = kAtariStageFloats / 4;       // float4 per stage matrix
    constexpr int NL = INITIAL ? 1 : 5;
    float4 rwq[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};     // reward-head weights of this lane's
    if (!INITIAL) {                                                                          // columns, requested early
        rwq[0] = *reinterpret_cast<const float4*>(w.dyn_reward_w + lcol);
        rwq[1] = *reinterpret_cast<const float4*>(w.dyn_reward_w + 128 + lcol);
    }
#pragma unroll 1
    for (int l = 0; l < NL; ++l) {
        const _Float16* src0 = sT[(l & 1) ? 2 : 0];
        const _Float16* src1 = sT[(l & 1) ? 3 : 1];
        _Float16* dst[2] = {sT[(l & 1) ? 0 : 2], sT[(l & 1) ? 1 : 3]};
        float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!INITIAL && l > 0) bq = *reinterpret_cast<const float4*>(w.dyn_bias + (l - 1) * 128 + lcol);
        const f32x4 bl = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            zero4(acc);
            zero4(acx);
            // the first stage of a layer starts cold (its tile was being written until the barrier); every later stage
            // of the layer finds its first fragments requested by the one before
            if (p == 0) stage64<true, false>(src0, src1, bw, wlane + (size_t)(stage + 1) * kSF, acc, acx, af, n16, g);
            else stage64<true, true>(src0, src1, bw, wlane + (size_t)(stage + 1) * kSF, acc, acx, af, n16, g);
            stage64<true, true>(src1, p == 0 ? src0 : nullptr, bw, wlane + (size_t)(stage + 2) * kSF, acc, acx, af, n16, g);
            stage += 2;
            fold4(acc, acx);
            MZ_TRACE_NW(1);                          // two stages
            f32x4 v[4];
            if (l == 0) {                            // input conv: + bias (+ action planes), relu = x
#pragma unroll
                for (int m = 0; m < 4; ++m) v[m] = __builtin_elementwise_max(acc[m] + res[p][m], f32x4{0.f, 0.f, 0.f, 0.f});
            } else if (l & 1) {                      // conv a of a residual block: relu
#pragma unroll
                for (int m = 0; m < 4; ++m) v[m] = __builtin_elementwise_max(acc[m] + bl, f32x4{0.f, 0.f, 0.f, 0.f});
            } else {                                 // conv b: x = relu(x + conv)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    v[m] = __builtin_elementwise_max(res[p][m] + (acc[m] + bl), f32x4{0.f, 0.f, 0.f, 0.f});
            }
            if (!(l & 1)) {
#pragma unroll
                for (int m = 0; m < 4; ++m) res[p][m] = v[m];
            }
            store_tile4(v, dst[p], n16, lcol, vmax);
            MZ_TRACE_NW(2);                          // epilogue
        }
        __syncthreads();
        MZ_TRACE_NW(3);                              // barrier
    }
    const _Float16* fin0 = INITIAL ? sT[0] : sT[2];      // the tiles holding the hidden state the prediction reads
    const _Float16* fin1 = INITIAL ? sT[1] : sT[3];
    _Float16* tP = INITIAL ? sT[2] : sT[0];
    _Float16* tD = INITIAL ? sT[3] : sT[1];
    if (INITIAL) {
        // Dense 128 -> 128 relu per position = the hidden state: Y -> X
        const float4 bq = *reinterpret_cast<const float4*>(w.rep_dense_b + lcol);
        const f32x4 bd = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            zero4(acc);
            zero4(acx);
            if (p == 0) stage64<true, false>(sT[2], sT[3], bw, wlane + (size_t)(stage + 1) * kSF, acc, acx, af, n16, g);
            else stage64<true, true>(sT[3], nullptr, bw, wlane + (size_t)(stage + 1) * kSF, acc, acx, af, n16, g);
            stage += 1;
            fold4(acc, acx);
#pragma unroll
            for (int m = 0; m < 4; ++m) res[p][m] = __builtin_elementwise_max(acc[m] + bd, f32x4{0.f, 0.f, 0.f, 0.f});
            store_tile4(res[p], sT[p], n16, lcol, vmax);
        }
        __syncthreads();
    }
    {   // hidden state out, f32, from the residual registers (streaming stores: nothing here reads them back); the reward
        // head's partial dot from the same registers
        float rdot[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float4 rw = rwq[p];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 x = res[p][m];
                if ((rowmask >> (16 * m + n16)) & 1ull)
                    __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(out_hidden + (size_t)(row0 + 16 * m + n16) * 256 +
                                                                            p * 128 + lcol));
                if (!INITIAL)
                    rdot[m] = __builtin_fmaf(x[3], rw.w, __builtin_fmaf(x[2], rw.z, __builtin_fmaf(x[1], rw.y, __builtin_fmaf(x[0], rw.x, rdot[m]))));
            }
        }
        if (!INITIAL) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float s = sum_over_g(rdot[m]);
                if (g == 0) sRew[wave * ROWS + 16 * m + n16] = s;
            }
        }
    }
    MZ_TRACE_NW(4);                                  // hidden state out, reward partials
    // the backup's inputs: requested here, where the residual registers have just died, not at the start (32 registers held
    // across the dynamics stages); they land during the prediction's three stages and heads
    if (FUSE) pf = backprop_prefetch<TG, TPB>(p, tree, tree < p.T, tgl);
    // prediction: Conv1D k=2 valid over the two positions, relu -> P
    {
        const float4 bq = *reinterpret_cast<const float4*>(w.pred_conv_b + lcol);
        const f32x4 bp = {bq.x, bq.y, bq.z, bq.w};
        zero4(acc);
        zero4(acx);
        stage64<true, false>(fin0, fin1, bw, wlane + (size_t)(stage + 1) * kSF, acc, acx, af, n16, g);
        stage64<true, true>(fin1, nullptr, bw, wlane + (size_t)(stage + 2) * kSF, acc, acx, af, n16, g);
        stage += 2;
        fold4(acc, acx);
        f32x4 v[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) v[m] = __builtin_elementwise_max(acc[m] + bp, f32x4{0.f, 0.f, 0.f, 0.f});
        store_tile4(v, tP, n16, lcol, vmax);
    }
    __syncthreads();
    if (!INITIAL && tid < ROWS) {                    // reward = tanh(<hidden', w> + b)
        float rs = 0.f;
#pragma unroll
        for (int v8 = 0; v8 < 8; ++v8) rs += sRew[v8 * ROWS + tid];
        sOut[tid][19] = tanhf(rs + w.dyn_reward_b);
    }
    {   // Dense 128 -> 128 relu -> D
        const float4 bq = *reinterpret_cast<const float4*>(w.pred_dense_b + lcol);
        const f32x4 bd = {bq.x, bq.y, bq.z, bq.w};
        zero4(acc);
        zero4(acx);
        stage64<true, false>(tP, nullptr, bw, wlane + (size_t)(stage + 1) * kSF, acc, acx, af, n16, g);
        stage += 1;
        fold4(acc, acx);
        f32x4 v[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) v[m] = __builtin_elementwise_max(acc[m] + bd, f32x4{0.f, 0.f, 0.f, 0.f});
        store_tile4(v, tD, n16, lcol, vmax);
    }
    __syncthreads();
    MZ_TRACE_NW(5);                                  // prediction conv + dense (3 stages, 2 barriers)
    if (wave < 2) {                                  // heads: column 0 = value (tanh), 1 .. 18 = policy logits
        const float4 bq = *reinterpret_cast<const float4*>(w.pred_head_b + lcol);
        zero4(acc);
        zero4(acx);
        stage64<false, false>(tD, nullptr, bw, wlane, acc, acx, af, n16, g);
        fold4(acc, acx);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float x[4] = {acc[m][0] + bq.x, acc[m][1] + bq.y, acc[m][2] + bq.z, acc[m][3] + bq.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = lcol + i;
                if (c < 19) sOut[16 * m + n16][c] = c == 0 ? tanhf(x[i]) : x[i];
            }
        }
    }
    __syncthreads();
    // keep_out = false: a later simulation of the same launch overwrites these rows (atari_run_kernel)
    for (int idx = tid; keep_out && idx < ROWS * 18; idx += kThreadsW) {
        const int r = idx / 18, o = idx % 18;
        if (sRowOff[r] >= 0) out_logits[(size_t)(row0 + r) * 18 + o] = sOut[r][1 + o];
    }
    if (keep_out && tid < ROWS && sRowOff[tid] >= 0) {
        out_value[row0 + tid] = sOut[tid][0];         // to_scalar is the identity (no support)
        if (out_value_raw) out_value_raw[row0 + tid] = sOut[tid][0];
        if (!INITIAL) out_reward[row0 + tid] = sOut[tid][19];
    }
    note_range(vmax, w.range_flag);
    MZ_TRACE_NW(6);                                  // heads, outputs
    if (FUSE) {
        // tree step of these 64 games (agent.py:208-211, then 177-190 for the next simulation): the tiles are free
        float* rows = reinterpret_cast<float*>(&sT[0][0]);
        double* s_log = reinterpret_cast<double*>(rows + ROWS * TreeLds<TG>::kRow);
        // the log table: staged by the caller once for all its simulations (atari_run_kernel), or here, over the tiles
        // (the barrier before the output stores has ended every read of them)
        const PoolDev p2 = p_image != nullptr ? load_pool_image(p_image) : p_arg;
        const PoolDev& p = p2;
        const LdsLog lds_log = staged_log != nullptr ? staged_log : stage_log_table(p, s_log);
        const float* mine = sOut[tid / TG];
        backprop_tree<TG, true, TPB>(p, tree, tree_active, tgl, nb, mine + 1, mine[19], mine[0], pf,
                                     rows + (tid / TG) * TreeLds<TG>::kRow);
        if (FUSE == 2 && do_select) {
            // a tree's eight lanes lie inside one wavefront, whose vector memory operations are performed in program
            // order: compiler ordering is enough (step_kernel, mz_tree.hip)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            select_tree<TG, kAtariTreeP>(p, tree, tree_active, tgl, parent_block, action, lds_log);
        }
    }
}

template <bool INITIAL>
__global__ __launch_bounds__(kThreadsW, 1) void atari_kernel(AtariDev w, int B, const float* __restrict__ in_base,
                                                             const int* __restrict__ parent_block, long block_stride,
                                                             const int* __restrict__ action, float* __restrict__ out_hidden,
                                                             float* __restrict__ out_reward, float* __restrict__ out_value,
                                                             float* __restrict__ out_logits,
                                                             float* __restrict__ out_value_raw) {
    atari_body<INITIAL, 0>(w, B, in_base, const_cast<int*>(parent_block), block_stride, const_cast<int*>(action), out_hidden,
                           out_reward, out_value, out_logits, out_value_raw, PoolDev{}, 0);
}

// run_mcts (agent.py:198-211) of 64 games in ONE launch: the body above, looped.  A workgroup's games, trees and hidden
// rows are its own from the first simulation to the last (the rows a simulation gathers were written by this workgroup:
// the barrier at the end of a simulation publishes them), so no launch boundary and no other workgroup's deepest tree
// separates two simulations.  Outputs and traces as in cartpole16_run_kernel.
__global__ __launch_bounds__(kThreadsW, 1) void atari_run_kernel(AtariDev w, const PoolDev* pool_image, int T, int nb0, int nsim, int final_select,
                                                                 float* hidden_base, int* parent_block, long block_stride,
                                                                 int* action, float* out_reward, float* out_value,
                                                                 float* out_logits, long out_stride, int* trace_parent,
                                                                 int* trace_action) {
    __shared__ double s_log_run[kLogLds];             // 2 log(n + 1): staged once for the whole run
    __shared__ int s_sel[2][64];                      // the selection between two simulations: leaf parents, actions
    const int row0 = blockIdx.x * 64;
    if (threadIdx.x < 64) {
        const int b = row0 + threadIdx.x;
        s_sel[0][threadIdx.x] = b < T ? parent_block[b] : -1;
        s_sel[1][threadIdx.x] = b < T ? action[b] : -1;
    }
    LdsLog run_log;                                   // never nullptr here: the host checked N + 2 <= kLogLds
    {
        const PoolDev p0 = load_pool_image(pool_image);         // (the descriptor is re-read where the body uses it)
        run_log = stage_log_table(p0, s_log_run);
    }
    // the body indexes both by game: game b of this workgroup is entry b - row0 of the LDS copies
    int* const sel_parent = &s_sel[0][0] - row0;
    int* const sel_action = &s_sel[1][0] - row0;
#pragma unroll 1
    for (int i = 0; i < nsim; ++i) {
        asm volatile("" ::: "memory");
        if (trace_parent != nullptr && threadIdx.x < 64) {
            const int b = row0 + threadIdx.x;
            if (b < T) {
                trace_parent[(size_t)i * T + b] = s_sel[0][threadIdx.x];
                trace_action[(size_t)i * T + b] = s_sel[1][threadIdx.x];
            }
        }
        float* oh = hidden_base + (size_t)(nb0 + i) * block_stride;
        float* orw = out_reward + (size_t)i * out_stride;
        float* ov = out_value + (size_t)i * out_stride;
        float* ol = out_logits + (size_t)i * out_stride * 18;
        atari_body<false, 2>(w, T, hidden_base, sel_parent, block_stride, sel_action, oh, orw, ov, ol, nullptr, PoolDev{},
                             nb0 + i, i + 1 < nsim || final_select != 0, out_stride != 0 || i + 1 == nsim, run_log,
                             pool_image);
        __syncthreads();
    }
    if (final_select != 0 && threadIdx.x < 64 && row0 + threadIdx.x < T) {
        parent_block[row0 + threadIdx.x] = s_sel[0][threadIdx.x];
        action[row0 + threadIdx.x] = s_sel[1][threadIdx.x];
    }
}

template <bool SELECT>
__global__ __launch_bounds__(kThreadsW, 1) void atari_step_kernel(AtariDev w, PoolDev p, int nb,
                                                                  const float* __restrict__ in_base, int* parent_block,
                                                                  long block_stride, int* action,
                                                                  float* __restrict__ out_hidden, float* __restrict__ out_reward,
                                                                  float* __restrict__ out_value, float* __restrict__ out_logits) {
    atari_body<false, SELECT ? 2 : 1>(w, p.T, in_base, parent_block, block_stride, action, out_hidden, out_reward, out_value,
                                      out_logits, nullptr, p, nb);
}

struct Packer {
    std::vector<float> buf;
    size_t add(const float* src, size_t n) {
        const size_t off = buf.size();
        buf.insert(buf.end(), src, src + n);
        while (buf.size() % 4) buf.push_back(0.f);       // keep every table 16-B aligned
        return off;
    }
    size_t add(const std::vector<float>& v) { return add(v.data(), v.size()); }
    // W[K][N] row-major (N <= 128) -> Wp[K/8][128][8], missing columns zero
    size_t add_packed(const float* W, int K, int N, int ldw) {
        std::vector<float> p((size_t)K * 128, 0.f);
        for (int k = 0; k < K; ++k)
            for (int n = 0; n < N; ++n) p[((size_t)(k / 8) * 128 + n) * 8 + (k % 8)] = W[(size_t)k * ldw + n];
        return add(p);
    }
    // W[128][N] row-major (N <= 128) -> split-f16 fragments for v_mfma_f32_16x16x32_f16 in the order ttt_head16 loads
    // them: 16-byte entry [s][plane][block = n / 16][lane = 16 g + n % 16] holds the 8 halfs k = 32s + 8g .. +7 of column
    // n (plane 0 = h, 1 = l; missing columns zero): one load instruction of a wavefront reads 1 KiB of consecutive bytes.
    size_t add_split16(const float* W, int N, int ldw) {
        std::vector<_Float16> p((size_t)128 * 128 * 2, (_Float16)0.f);
        for (int k = 0; k < 128; ++k)
            for (int n = 0; n < N; ++n) {
                const float w = W[(size_t)k * ldw + n];
                const _Float16 h = (_Float16)w;
                const _Float16 l = (_Float16)((w - (float)h) * 2048.f);
                const int st = k / 32, g = (k % 32) / 8, i = k % 8, lane = 16 * g + n % 16, blk = n / 16;
                p[((((size_t)st * 2 + 0) * 8 + blk) * 64 + lane) * 8 + i] = h;
                p[((((size_t)st * 2 + 1) * 8 + blk) * 64 + lane) * 8 + i] = l;
            }
        std::vector<float> f((size_t)128 * 128);
        memcpy(f.data(), p.data(), f.size() * sizeof(float));
        return add(f);
    }
    // W[32][128] row-major -> the one k-step of add_split16's layout: 16-byte entry [plane][block = n / 16][lane = 16 g +
    // n % 16] holds the 8 halfs k = 8g .. +7 of column n (4096 floats)
    size_t add_split16_k32(const float* W) {
        std::vector<_Float16> p((size_t)32 * 128 * 2, (_Float16)0.f);
        for (int k = 0; k < 32; ++k)
            for (int n = 0; n < 128; ++n) {
                const float w = W[(size_t)k * 128 + n];
                const _Float16 h = (_Float16)w;
                const _Float16 l = (_Float16)((w - (float)h) * 2048.f);
                const int g = k / 8, i = k % 8, lane = 16 * g + n % 16, blk = n / 16;
                p[(((size_t)0 * 8 + blk) * 64 + lane) * 8 + i] = h;
                p[(((size_t)1 * 8 + blk) * 64 + lane) * 8 + i] = l;
            }
        std::vector<float> f((size_t)32 * 128);
        memcpy(f.data(), p.data(), f.size() * sizeof(float));
        return add(f);
    }
    // W[128][128] row-major -> split-f16 MFMA fragments for v_mfma_f32_32x32x16_f16, the size of the f32 matrix:
    // 16-byte entry [j = 2t + plane][n][g] holds the 8 halfs k = 16t + 8g .. +7 of column n; plane 0 = h = f16(w),
    // plane 1 = l = f16((w - h) * 2048).  w ~= h + l/2048 to 22 mantissa bits.
    size_t add_split_f16(const float* W, int ldw) {
        std::vector<_Float16> p((size_t)128 * 128 * 2);
        for (int k = 0; k < 128; ++k)
            for (int n = 0; n < 128; ++n) {
                const float w = W[(size_t)k * ldw + n];
                const _Float16 h = (_Float16)w;
                const _Float16 l = (_Float16)((w - (float)h) * 2048.f);
                const int t = k / 16, g = (k % 16) / 8, i = k % 8;
                p[(((size_t)(2 * t) * 128 + n) * 2 + g) * 8 + i] = h;
                p[(((size_t)(2 * t + 1) * 128 + n) * 2 + g) * 8 + i] = l;
            }
        std::vector<float> f((size_t)128 * 128);
        memcpy(f.data(), p.data(), f.size() * sizeof(float));
        return add(f);
    }
};

constexpr int64_t kTttWeights = 3456 + 128 + 16384 + 128 + 157824 + 128 + 16384 + 128 + 1152 + 1 + 147456 + 128 + 16384 + 128 + 128 + 1 + 1152 + 9;
constexpr int64_t kAtariWeights = (3 * 128 * 128 + 128) + (128 * 128 + 128) + (3 * 146 * 128 + 128) + 4 * (3 * 128 * 128 + 128) +
                                  (256 + 1) + (2 * 128 * 128 + 128) + (128 * 128 + 128) + (128 + 1) + (128 * 18 + 18);
constexpr int64_t kCartWeights = 51712 + 128 + 16384 + 128 + 32768 + 128 + 16384 + 128 + 16384 + 128 + 128 + 1 + 16384 + 128 + 16384 + 128 + 12928 + 101 + 256 + 2;

}  // namespace

struct mz_net {
    int kind = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t training_steps = 0;
    float* weights = nullptr;      // one device allocation holding every packed table
    TttDev ttt{};
    const float* ttt_rep_dense_s = nullptr;   // representation Dense as add_split16 fragments (ttt_initial_kernel_w)
    const float* ttt_rep_conv_c = nullptr;    // [9 cells] x the 3 -> 128 conv as a 32 x 128 matrix over the observation row
    CartDev cart{};
    AtariDev atari{};
    float* ppart = nullptr;        // TicTacToe scratch: [cap][9][128] prediction-conv partials
    float* rpart = nullptr;        //                    [cap][9] reward partials
    int scratch_cap = 0;
    int* range_flag = nullptr;     // device word raised by the kernels when a value leaves the split-f16 domain
    float* static_scratch = nullptr;  // this device's copy of the file-scope `scratch` array (mz_infer_recurrent_noalloc)
    // ttt_run_kernel's meeting words, zeroed once: [kRunMaxRowBlocks] arrival counters, [kRunMaxRowBlocks][2] cell flags,
    // XCC masks, four counters (mz_debug_run_info), then [kRunMaxRowBlocks][64] tagged selection words (2 MB)
    unsigned long long* run_sync = nullptr;
    unsigned long long run_epoch = 0; // serial number of the last ttt_run_kernel launch
    int job_rows = 0;                 // mz_debug_set_job_rows: 0 = by batch size (ttt_mb), 16 or 64 = that shape always
    int run_mode = 0;                 // mz_debug_set_run_mode: 0 = one launch where possible, 1 = ... with agent-scope
                                      // fences at every meeting, 2 = a launch per simulation
    bool hidden_split = false;        // inside a native move: the hidden workspace holds split images (mz_net_begin_move)
    hipEvent_t mid_event = nullptr;   // measurement hook, see mz_net_set_mid_event
};

namespace {

int ensure_scratch(mz_net* net, int batch) {
    if (net->kind != MZ_NET_TICTACTOE || batch <= net->scratch_cap) return MZ_OK;
    MZ_HIP(hipStreamSynchronize(net->stream));
    if (net->ppart) (void)hipFree(net->ppart);
    if (net->rpart) (void)hipFree(net->rpart);
    net->ppart = net->rpart = nullptr;
    net->scratch_cap = 0;
    const int cap = (batch + 63) / 64 * 64;
    MZ_HIP(hipMalloc(reinterpret_cast<void**>(&net->ppart), (size_t)cap * 9 * 128 * sizeof(float)));
    MZ_HIP(hipMalloc(reinterpret_cast<void**>(&net->rpart), (size_t)cap * 9 * sizeof(float)));
    net->scratch_cap = cap;
    return MZ_OK;
}

int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mz::fail(MZ_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return MZ_OK;
}

}  // namespace

extern "C" {

int64_t mz_net_num_weights(int32_t kind) {
    return kind == MZ_NET_TICTACTOE ? kTttWeights : kind == MZ_NET_CARTPOLE ? kCartWeights
           : kind == MZ_NET_ATARI_SHAPED ? kAtariWeights : -1;
}

int32_t mz_net_hidden_size(int32_t kind) {
    return kind == MZ_NET_TICTACTOE ? 1152 : kind == MZ_NET_CARTPOLE ? 128 : kind == MZ_NET_ATARI_SHAPED ? 256 : -1;
}

int32_t mz_net_num_actions(int32_t kind) {
    return kind == MZ_NET_TICTACTOE ? 9 : kind == MZ_NET_CARTPOLE ? 2 : kind == MZ_NET_ATARI_SHAPED ? 18 : -1;
}

int32_t mz_net_observation_size(int32_t kind) {
    return kind == MZ_NET_TICTACTOE ? 27 : kind == MZ_NET_CARTPOLE ? 4 : kind == MZ_NET_ATARI_SHAPED ? 256 : -1;
}

int mz_net_create(int32_t kind, const float* wh, int64_t num_weights, int64_t training_steps, void* stream,
                  mz_net_t** out_net) {
    MZ_REQUIRE(out_net != nullptr, "out_net is NULL");
    *out_net = nullptr;
    MZ_REQUIRE(kind == MZ_NET_TICTACTOE || kind == MZ_NET_CARTPOLE || kind == MZ_NET_ATARI_SHAPED, "unknown network kind %d",
               kind);
    MZ_REQUIRE(wh != nullptr, "weights_host is NULL");
    MZ_REQUIRE(num_weights == mz_net_num_weights(kind), "network kind %d takes %lld weights, got %lld", kind,
               (long long)mz_net_num_weights(kind), (long long)num_weights);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return mz::fail(MZ_ERR_NO_DEVICE, "no HIP device visible: libmzhip has no CPU fallback");
    }
    mz_net* net = new (std::nothrow) mz_net();
    if (!net) return mz::fail(MZ_ERR_OOM, "host allocation failed");
    net->kind = kind;
    net->stream = static_cast<hipStream_t>(stream);
    net->training_steps = training_steps;
    // split-f16 domain (see "split-f16 MFMA" above): every weight must be finite and within f16's range
    for (int64_t i = 0; i < num_weights; ++i) {
        if (!(std::fabs(wh[i]) <= 65504.f)) {
            delete net;
            return mz::fail(MZ_ERR_INVALID_ARG, "weight %lld = %g is outside the kernels' operand range (finite, |w| <= 65504)",
                            (long long)i, (double)wh[i]);
        }
    }
    Packer pk;
    std::vector<std::pair<const float**, size_t>> fix;      // device pointer fields and their offsets in pk.buf
    auto put = [&](const float*& field, size_t off) { fix.push_back({&field, off}); };
    const float* p = wh;
    auto take = [&](size_t n) {
        const float* q = p;
        p += n;
        return q;
    };
    if (kind == MZ_NET_TICTACTOE) {
        TttDev& t = net->ttt;
        const float* rep_conv = take(3456);
        const float* rep_conv_b = take(128);
        const float* rep_dense = take(16384);
        const float* rep_dense_b = take(128);
        const float* dyn_conv = take(157824);          // (3,3,137,128)
        const float* dyn_conv_b = take(128);
        const float* dyn_dense = take(16384);
        const float* dyn_dense_b = take(128);
        const float* dyn_reward = take(1152);
        const float* dyn_reward_b = take(1);
        const float* pred_conv = take(147456);         // (3,3,128,128)
        const float* pred_conv_b = take(128);
        const float* pred_dense = take(16384);
        const float* pred_dense_b = take(128);
        const float* pred_value = take(128);
        const float* pred_value_b = take(1);
        const float* pred_policy = take(1152);         // (128,9)
        const float* pred_policy_b = take(9);
        put(t.rep_conv, pk.add(rep_conv, 3456));
        put(t.rep_conv_b, pk.add(rep_conv_b, 128));
        put(t.rep_dense_b, pk.add(rep_dense_b, 128));
        std::vector<float> actbias(9 * 9 * 128);
        for (int a = 0; a < 9; ++a)
            for (int c = 0; c < 9; ++c)
                for (int n = 0; n < 128; ++n) {
                    double s = dyn_conv_b[n];
                    for (int tp = 0; tp < 9; ++tp) {
                        const int ni = c / 3 + tp / 3 - 1, nj = c % 3 + tp % 3 - 1;
                        if (ni >= 0 && ni < 3 && nj >= 0 && nj < 3) s += dyn_conv[((size_t)tp * 137 + 128 + a) * 128 + n];
                    }
                    actbias[((size_t)a * 9 + c) * 128 + n] = (float)s;
                }
        put(t.dyn_actbias, pk.add(actbias));
        put(t.dyn_dense_b, pk.add(dyn_dense_b, 128));
        put(t.dyn_reward_w, pk.add(dyn_reward, 1152));
        put(t.pred_conv_b, pk.add(pred_conv_b, 128));
        put(t.pred_dense_p, pk.add_split16(pred_dense, 128, 128));
        put(t.pred_dense_b, pk.add(pred_dense_b, 128));
        std::vector<float> headW(128 * 16, 0.f), headB(16, 0.f);      // [k][16]: value | 9 policy | zeros
        for (int k = 0; k < 128; ++k) {
            headW[(size_t)k * 16] = pred_value[k];
            for (int o = 0; o < 9; ++o) headW[(size_t)k * 16 + 1 + o] = pred_policy[(size_t)k * 9 + o];
        }
        headB[0] = pred_value_b[0];
        for (int o = 0; o < 9; ++o) headB[1 + o] = pred_policy_b[o];
        t.dyn_reward_b = dyn_reward_b[0];
        t.pred_value_b = pred_value_b[0];
        // the four uniform-cost jobs
        static const int kJobCells[4][3] = {{4, 0, -1}, {1, 3, -1}, {5, 7, -1}, {2, 6, 8}};
        std::vector<float> desc;      // int4 bit patterns stored in the float buffer
        auto push = [&](int a, int b, int c, int d) {
            const int v[4] = {a, b, c, d};
            for (int i = 0; i < 4; ++i) {
                float f;
                memcpy(&f, &v[i], 4);
                desc.push_back(f);
            }
        };
        for (int j = 0; j < 4; ++j)
            for (int ci = 0; ci < 3; ++ci) t.job_cells[j][ci] = kJobCells[j][ci];
        // their stages for the ring kernel: (kind, cell, weight matrix, source cell); matrices 0..8 = conv taps,
        // 9 = Dense, 10 + c = prediction conv of cell c (consecutive in the packed buffer, checked below)
        // one cell: its valid taps, Dense, and (pred_w >= 0) its prediction-conv stage with .w = pred_w; `signal` = the cell
        // whose flag the second tap raises (-1: none)
        auto emit_cell = [&](int c, int pred_w, int signal) {
            int valid[9], nv = 0;
            for (int tp = 0; tp < 9; ++tp) {
                const int ni = c / 3 + tp / 3 - 1, nj = c % 3 + tp % 3 - 1;
                if (ni >= 0 && ni < 3 && nj >= 0 && nj < 3) valid[nv++] = tp;
            }
            for (int i = 0; i < nv; ++i) {
                const int tp = valid[i];
                push(i == 0 ? kRingTapFirst : i == nv - 1 ? kRingTapLast : (kRingTap | (i == 1 ? (signal + 1) << 8 : 0)), c, tp,
                     (c / 3 + tp / 3 - 1) * 3 + (c % 3 + tp % 3 - 1));
            }
            push(kRingDense, c, 9, 0);
            if (pred_w >= 0) push(kRingPred, c, 10 + c, pred_w);
        };
        auto pw = [](int slice, int flags) { return (slice << 8) | flags; };
        // Slices of a game's partials row (the head adds them in index order): 0 = cells {4, 0}, 1 = {1, 3}, 2 = {5, 7},
        // 3 = {8}, 4 = {2}, 5 = {6} for the 64-game jobs -- the corners' terms stay apart so that it does not matter, to
        // the last bit, which workgroup computes them; slice c = cell c for the 16-game jobs.
        // (a) the 64-game jobs as launches of their own
        t.ring_first[0] = (int)desc.size() / 4;
        emit_cell(4, pw(0, kPredFresh), -1);
        emit_cell(0, pw(0, kPredStore), -1);
        t.ring_first[1] = (int)desc.size() / 4;
        emit_cell(1, pw(1, kPredFresh), -1);
        emit_cell(3, pw(1, kPredStore), -1);
        t.ring_first[2] = (int)desc.size() / 4;
        emit_cell(5, pw(2, kPredFresh), -1);
        emit_cell(7, pw(2, kPredStore), -1);
        t.ring_first[3] = (int)desc.size() / 4;
        emit_cell(2, pw(4, kPredFresh | kPredStore), -1);
        emit_cell(6, pw(5, kPredFresh | kPredStore), -1);
        emit_cell(8, pw(3, kPredFresh | kPredStore), -1);
        t.ring_first[4] = (int)desc.size() / 4;
        // (b) the nine single-cell jobs of the 16-game workgroups
        for (int c = 0; c < 9; ++c) {
            t.ring_first[5 + c] = (int)desc.size() / 4;
            emit_cell(c, pw(c, kPredFresh | kPredStore), -1);
        }
        t.ring_first[14] = (int)desc.size() / 4;
        // (c) the 64-game jobs inside ttt_run_kernel: the corners 2 and 6 keep their taps and Dense in job 3, which raises
        // their flags at the first tap of its next cell; their prediction-conv terms end jobs 1 and 2
        t.ring_first[15] = (int)desc.size() / 4;
        emit_cell(4, pw(0, kPredFresh), -1);
        emit_cell(0, pw(0, kPredStore), -1);
        t.ring_first[16] = (int)desc.size() / 4;
        emit_cell(1, pw(1, kPredFresh), -1);
        emit_cell(3, pw(1, 0), -1);
        push(kRingPredRemote, 2, 10 + 2, pw(1, kPredStore));        // (p1 + p3) + p2: what the head adds for the launches
        t.ring_first[17] = (int)desc.size() / 4;
        emit_cell(5, pw(2, kPredFresh), -1);
        emit_cell(7, pw(2, 0), -1);
        push(kRingPredRemote, 6, 10 + 6, pw(2, kPredStore));
        t.ring_first[18] = (int)desc.size() / 4;
        emit_cell(2, -1, -1);
        emit_cell(6, -1, 2);
        emit_cell(8, pw(3, kPredFresh | kPredStore), 6);
        t.ring_first[19] = (int)desc.size() / 4;
        put(reinterpret_cast<const float*&>(t.ring_desc), pk.add(desc));
        put(t.pred_head_p, pk.add_split16(headW.data(), 16, 16));
        put(t.pred_head_b, pk.add(headB));
        auto add_series = [&](const float*& field, int count, auto&& one) {      // `count` matrices, consecutive
            size_t first = 0;
            for (int i = 0; i < count; ++i) {
                const size_t o = one(i);
                if (i == 0) put(field, first = o);
            }
            return first;
        };
        add_series(t.pred_conv_h, 9, [&](int c) { return pk.add_split_f16(pred_conv + (size_t)c * 128 * 128, 128); });
        put(t.rep_dense_h, pk.add_split_f16(rep_dense, 128));
        const size_t o_conv =
            add_series(t.dyn_conv_s, 9, [&](int tp) { return pk.add_split16(dyn_conv + (size_t)tp * 137 * 128, 128, 128); });
        const size_t o_dense = pk.add_split16(dyn_dense, 128, 128);
        put(t.dyn_dense_s, o_dense);
        const size_t o_pred =
            add_series(t.pred_conv_s, 9, [&](int c) { return pk.add_split16(pred_conv + (size_t)c * 128 * 128, 128, 128); });
        // the ring kernel addresses the nineteen matrices as dyn_conv_s + 16384 * index
        if (o_dense != o_conv + 9 * 16384 || o_pred != o_dense + 16384) {
            delete net;
            return mz::fail(MZ_ERR_STATE, "packed TicTacToe stage matrices are not consecutive");
        }
        put(net->ttt_rep_dense_s, pk.add_split16(rep_dense, 128, 128));
        // the 3 x 3 same conv of cell c as ONE matrix product over the game's observation row [cell'][plane] (K = 27, padded
        // to 32): row cell' * 3 + p = the tap that reaches cell' from c, zero where cell' is no neighbour of c
        for (int c = 0; c < 9; ++c) {
            std::vector<float> wc((size_t)32 * 128, 0.f);
            for (int tp = 0; tp < 9; ++tp) {
                const int ni = c / 3 + tp / 3 - 1, nj = c % 3 + tp % 3 - 1;
                if (ni < 0 || ni >= 3 || nj < 0 || nj >= 3) continue;
                for (int pl = 0; pl < 3; ++pl)
                    memcpy(&wc[(size_t)((ni * 3 + nj) * 3 + pl) * 128], rep_conv + (size_t)(tp * 3 + pl) * 128, 128 * sizeof(float));
            }
            const size_t o = pk.add_split16_k32(wc.data());
            if (c == 0) put(net->ttt_rep_conv_c, o);
        }
    } else if (kind == MZ_NET_ATARI_SHAPED) {
        AtariDev& a = net->atari;
        const float* rep_conv = take(3 * 128 * 128);   // (3,128,128)
        const float* rep_conv_b = take(128);
        const float* rep_dense = take(16384);
        const float* rep_dense_b = take(128);
        const float* dyn_conv = take(3 * 146 * 128);   // (3,146,128)
        const float* dyn_conv_b = take(128);
        const float* res_k[4];
        const float* res_b[4];
        for (int i = 0; i < 4; ++i) {                  // res1a, res1b, res2a, res2b: (3,128,128) + (128)
            res_k[i] = take(3 * 128 * 128);
            res_b[i] = take(128);
        }
        const float* dyn_reward = take(256);
        const float* dyn_reward_b = take(1);
        const float* pred_conv = take(2 * 128 * 128);  // (2,128,128)
        const float* pred_conv_b = take(128);
        const float* pred_dense = take(16384);
        const float* pred_dense_b = take(128);
        const float* pred_value = take(128);
        const float* pred_value_b = take(1);
        const float* pred_policy = take(128 * 18);     // (128,18)
        const float* pred_policy_b = take(18);
        std::vector<float> headW(128 * 32, 0.f), headB(32, 0.f);      // [k][32]: value | 18 policy | zeros
        for (int k = 0; k < 128; ++k) {
            headW[(size_t)k * 32] = pred_value[k];
            for (int o = 0; o < 18; ++o) headW[(size_t)k * 32 + 1 + o] = pred_policy[(size_t)k * 18 + o];
        }
        headB[0] = pred_value_b[0];
        for (int o = 0; o < 18; ++o) headB[1 + o] = pred_policy_b[o];
        // a k=3 `same` conv over two positions: output 0 = in0 @ tap 1 + in1 @ tap 2, output 1 = in0 @ tap 0 + in1 @ tap 1
        static const int kTapOrder[4] = {1, 2, 0, 1};
        auto add_conv = [&](const float* k3, int cin) {           // the hidden channels (rows 0..127) of each tap
            size_t first = 0;
            for (int i = 0; i < 4; ++i) {
                const size_t o = pk.add_split16(k3 + (size_t)kTapOrder[i] * cin * 128, 128, 128);
                if (i == 0) first = o;
            }
            return first;
        };
        auto add_tail = [&]() {                                    // prediction conv taps 0, 1 | dense | heads
            pk.add_split16(pred_conv, 128, 128);
            pk.add_split16(pred_conv + 128 * 128, 128, 128);
            pk.add_split16(pred_dense, 128, 128);
            pk.add_split16(headW.data(), 32, 32);
        };
        put(a.rec_s, add_conv(dyn_conv, 146));
        for (int i = 0; i < 4; ++i) add_conv(res_k[i], 128);
        add_tail();
        put(a.ini_s, add_conv(rep_conv, 128));
        pk.add_split16(rep_dense, 128, 128);
        pk.add_split16(rep_dense, 128, 128);
        add_tail();
        std::vector<float> actbias(18 * 2 * 128);
        for (int act = 0; act < 18; ++act)
            for (int pos = 0; pos < 2; ++pos)
                for (int n = 0; n < 128; ++n) {
                    double sum = dyn_conv_b[n];
                    for (int j = 0; j < 3; ++j) {
                        const int in_pos = pos + j - 1;                       // the plane is 1 at every real position
                        if (in_pos >= 0 && in_pos < 2) sum += dyn_conv[((size_t)j * 146 + 128 + act) * 128 + n];
                    }
                    actbias[((size_t)act * 2 + pos) * 128 + n] = (float)sum;
                }
        put(a.dyn_actbias, pk.add(actbias));
        std::vector<float> rb(4 * 128);
        for (int i = 0; i < 4; ++i) memcpy(rb.data() + i * 128, res_b[i], 128 * sizeof(float));
        put(a.dyn_bias, pk.add(rb));
        put(a.dyn_reward_w, pk.add(dyn_reward, 256));
        put(a.rep_conv_b, pk.add(rep_conv_b, 128));
        put(a.rep_dense_b, pk.add(rep_dense_b, 128));
        put(a.pred_conv_b, pk.add(pred_conv_b, 128));
        put(a.pred_dense_b, pk.add(pred_dense_b, 128));
        put(a.pred_head_b, pk.add(headB));
        a.dyn_reward_b = dyn_reward_b[0];
    } else {
        CartDev& c = net->cart;
        const float* rep_conv = take(51712);           // (4,101,128)
        const float* rep_conv_b = take(128);
        const float* rep_dense = take(16384);
        const float* rep_dense_b = take(128);
        const float* dyn_conv = take(32768);           // (2,128,128)
        const float* dyn_conv_b = take(128);
        const float* dyn_dense = take(16384);
        const float* dyn_dense_b = take(128);
        const float* dyn_rdense = take(16384);
        const float* dyn_rdense_b = take(128);
        const float* dyn_reward = take(128);
        const float* dyn_reward_b = take(1);
        const float* pred_dense1 = take(16384);
        const float* pred_dense1_b = take(128);
        const float* pred_dense2 = take(16384);
        const float* pred_dense2_b = take(128);
        const float* pred_value = take(12928);         // (128,101)
        const float* pred_value_b = take(101);
        const float* pred_policy = take(256);          // (128,2)
        const float* pred_policy_b = take(2);
        put(c.rep_conv, pk.add(rep_conv, 51712));
        put(c.rep_conv_b, pk.add(rep_conv_b, 128));
        put(c.rep_dense_p, pk.add_packed(rep_dense, 128, 128, 128));
        put(c.rep_dense_b, pk.add(rep_dense_b, 128));
        put(c.dyn_conv_p, pk.add_packed(dyn_conv, 128, 128, 128));
        put(c.dyn_conv_b, pk.add(dyn_conv_b, 128));
        std::vector<float> plane(128);
        for (int n = 0; n < 128; ++n) {
            double s = 0.0;
            for (int k = 0; k < 128; ++k) s += dyn_conv[((size_t)128 + k) * 128 + n];
            plane[n] = (float)s;
        }
        put(c.dyn_plane, pk.add(plane));
        put(c.dyn_dense_p, pk.add_packed(dyn_dense, 128, 128, 128));
        put(c.dyn_dense_b, pk.add(dyn_dense_b, 128));
        put(c.dyn_rdense_p, pk.add_packed(dyn_rdense, 128, 128, 128));
        put(c.dyn_rdense_b, pk.add(dyn_rdense_b, 128));
        put(c.dyn_reward_w, pk.add(dyn_reward, 128));
        put(c.pred_dense1_p, pk.add_packed(pred_dense1, 128, 128, 128));
        put(c.pred_dense1_b, pk.add(pred_dense1_b, 128));
        put(c.pred_dense2_p, pk.add_packed(pred_dense2, 128, 128, 128));
        put(c.pred_dense2_b, pk.add(pred_dense2_b, 128));
        put(c.pred_value_p, pk.add_packed(pred_value, 128, 101, 101));
        std::vector<float> vb(128, 0.f);
        for (int n = 0; n < 101; ++n) vb[n] = pred_value_b[n];
        put(c.pred_value_b, pk.add(vb));
        std::vector<float> polT(2 * 128);
        for (int k = 0; k < 128; ++k)
            for (int o = 0; o < 2; ++o) polT[(size_t)o * 128 + k] = pred_policy[(size_t)k * 2 + o];
        put(c.pred_policy_w, pk.add(polT));
        put(c.pred_policy_b, pk.add(pred_policy_b, 2));
        put(c.rep_dense_s, pk.add_split16(rep_dense, 128, 128));
        put(c.dyn_conv_s, pk.add_split16(dyn_conv, 128, 128));
        put(c.dyn_dense_s, pk.add_split16(dyn_dense, 128, 128));
        put(c.dyn_rdense_s, pk.add_split16(dyn_rdense, 128, 128));
        put(c.pred_dense1_s, pk.add_split16(pred_dense1, 128, 128));
        put(c.pred_dense2_s, pk.add_split16(pred_dense2, 128, 128));
        put(c.pred_value_s, pk.add_split16(pred_value, 101, 101));
        c.dyn_reward_b = dyn_reward_b[0];
    }
    const size_t range_off = pk.add(std::vector<float>(4, 0.f));       // the range flag (an int) lives with the tables
    hipError_t e = hipGetDevice(&net->device);
    if (e == hipSuccess) {
        const int st = scratch_for_device(net->device, &net->static_scratch);      // once per device, that device current
        if (st != MZ_OK) {
            delete net;
            return st;
        }
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&net->weights), pk.buf.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(net->weights, pk.buf.data(), pk.buf.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (net->weights) (void)hipFree(net->weights);
        delete net;
        return mz::fail(e == hipErrorOutOfMemory ? MZ_ERR_OOM : MZ_ERR_HIP, "weight upload failed: %s", hipGetErrorString(e));
    }
    if (kind == MZ_NET_TICTACTOE) {
        const size_t bytes = (size_t)kRunMaxRowBlocks * (3 * sizeof(unsigned long long) + sizeof(unsigned)) +
                             4 * sizeof(unsigned long long) + (size_t)kRunMaxRowBlocks * 64 * sizeof(unsigned long long);
        e = hipMalloc(reinterpret_cast<void**>(&net->run_sync), bytes);
        if (e == hipSuccess) e = hipMemset(net->run_sync, 0, bytes);
        if (e != hipSuccess) {
            if (net->run_sync) (void)hipFree(net->run_sync);
            (void)hipFree(net->weights);
            delete net;
            return mz::fail(e == hipErrorOutOfMemory ? MZ_ERR_OOM : MZ_ERR_HIP, "allocation failed: %s", hipGetErrorString(e));
        }
    }
    for (auto& f : fix) *f.first = net->weights + f.second;
    net->range_flag = reinterpret_cast<int*>(net->weights + range_off);
    net->ttt.range_flag = net->cart.range_flag = net->atari.range_flag = net->range_flag;
    *out_net = net;
    return MZ_OK;
}

int mz_net_destroy(mz_net_t* net) {
    if (!net) return MZ_OK;
    (void)hipSetDevice(net->device);
    (void)hipStreamSynchronize(net->stream);
    if (net->weights) (void)hipFree(net->weights);
    if (net->ppart) (void)hipFree(net->ppart);
    if (net->rpart) (void)hipFree(net->rpart);
    if (net->run_sync) (void)hipFree(net->run_sync);
    delete net;
    return MZ_OK;
}

int mz_net_set_stream(mz_net_t* net, void* stream) {
    MZ_REQUIRE(net != nullptr, "net is NULL");
    if (net->stream != static_cast<hipStream_t>(stream)) {
        // work enqueued on the old stream may still be using the handle's scratch, which a larger batch on the new
        // stream would free and reallocate (ensure_scratch synchronises only the stream it runs on)
        MZ_HIP(hipSetDevice(net->device));
        MZ_HIP(hipStreamSynchronize(net->stream));
    }
    net->stream = static_cast<hipStream_t>(stream);
    return MZ_OK;
}

int64_t mz_net_training_steps(mz_net_t* net) { return net ? net->training_steps : -1; }

int mz_net_status(mz_net_t* net, int32_t* out_flags_host) {
    MZ_REQUIRE(net != nullptr && out_flags_host != nullptr, "net/out_flags_host is NULL");
    MZ_HIP(hipSetDevice(net->device));
    int flags = 0;
    MZ_HIP(hipMemcpyAsync(&flags, net->range_flag, sizeof(int), hipMemcpyDeviceToHost, net->stream));
    MZ_HIP(hipMemsetAsync(net->range_flag, 0, sizeof(int), net->stream));
    MZ_HIP(hipStreamSynchronize(net->stream));
    *out_flags_host = flags;
    return MZ_OK;
}

#ifdef MZ_RUN_TRACE
int mz_debug_run_trace(uint64_t* out, int64_t out_bytes) {
    MZ_REQUIRE(out != nullptr && out_bytes == (int64_t)sizeof(g_run_trace), "out must hold %zu bytes", sizeof(g_run_trace));
    void* sym = nullptr;
    MZ_HIP(hipGetSymbolAddress(&sym, HIP_SYMBOL(g_run_trace)));
    MZ_HIP(hipDeviceSynchronize());
    MZ_HIP(hipMemcpy(out, sym, (size_t)out_bytes, hipMemcpyDeviceToHost));
    MZ_HIP(hipMemset(sym, 0, (size_t)out_bytes));
    return MZ_OK;
}
#endif

#ifdef MZ_TREE_TRACE
// experiment builds only: the per-wavefront phase counters of THIS translation unit (the job kernel's stamps)
int mz_debug_net_trace(uint32_t* out, int64_t out_bytes) {
    MZ_HIP(hipDeviceSynchronize());
    MZ_REQUIRE(out_bytes == (int64_t)sizeof(unsigned) * kTraceWaves * 2 * kTracePhases, "trace buffer size");
    MZ_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tree_trace), (size_t)out_bytes));
    void* sym = nullptr;
    MZ_HIP(hipGetSymbolAddress(&sym, HIP_SYMBOL(g_tree_trace)));
    MZ_HIP(hipMemset(sym, 0, (size_t)out_bytes));
    return MZ_OK;
}
#endif

int mz_net_set_mid_event(mz_net_t* net, void* hip_event) {
    MZ_REQUIRE(net != nullptr, "net is NULL");
    net->mid_event = static_cast<hipEvent_t>(hip_event);
    return MZ_OK;
}

// Rows per workgroup tile (32 or 64) of the large-batch CartPole kernel.  64 halves the weight traffic per game; 32
// doubles the number of workgroups, which wins while a 64-row grid cannot fill the 256 CUs.
static int tile_rows(int batch) { return (batch + 63) / 64 >= 512 ? 64 : 32; }

static int ttt_mb(const mz_net* net, int batch);

// what follows the initial inference of a native move on the trees (mz_net_begin_move_roots)
struct RootsArgs {
    mz_pool* pool;
    const int32_t *legal, *n_legal;
    const double* noise;
    double frac;
    int32_t *out_parent, *out_action;
};

static int infer_initial(mz_net_t* net, int32_t batch, const float* observation, float* out_hidden, float* out_value,
                         float* out_logits, float* out_value_raw, bool split_hidden, const RootsArgs* roots = nullptr) {
    MZ_REQUIRE(net != nullptr, "net is NULL");
    MZ_REQUIRE(batch >= 0, "batch must be >= 0 (got %d)", batch);
    net->hidden_split = false;   // a move starts here; a native move that failed half way may have left it set
    if (batch == 0) return MZ_OK;
    MZ_REQUIRE(observation && out_hidden && out_value && out_logits, "observation/out_hidden/out_value/out_logits is NULL");
    MZ_HIP(hipSetDevice(net->device));
    const int rows = tile_rows(batch);
    const int nrb = (batch + rows - 1) / rows;
    hipStream_t st = net->stream;
    if (net->kind == MZ_NET_TICTACTOE) {
        int s = ensure_scratch(net, batch);
        if (s != MZ_OK) return s;
        const int nhb = (batch + 15) / 16;
        const int nrb32 = (batch + 31) / 32;
        if (ttt_mb(net, batch) == 4) {               // the batches of the 64-game jobs
            if (split_hidden)
                ttt_initial_kernel_w<true><<<ttt_job_grid(batch, 4), kThreadsW, 0, st>>>(
                    net->ttt, net->ttt_rep_dense_s, net->ttt_rep_conv_c, batch, observation, out_hidden, net->ppart);
            else
                ttt_initial_kernel_w<false><<<ttt_job_grid(batch, 4), kThreadsW, 0, st>>>(
                    net->ttt, net->ttt_rep_dense_s, net->ttt_rep_conv_c, batch, observation, out_hidden, net->ppart);
        } else if (split_hidden)
            ttt_initial_kernel_h<true><<<32 * ((nrb32 + 7) / 8), kThreads, 0, st>>>(net->ttt, batch, observation,
                                                                                   out_hidden, net->ppart);
        else
            ttt_initial_kernel_h<false><<<32 * ((nrb32 + 7) / 8), kThreads, 0, st>>>(net->ttt, batch, observation,
                                                                                    out_hidden, net->ppart);
        if (net->mid_event) MZ_HIP(hipEventRecord(net->mid_event, st));
        if (roots != nullptr)
            ttt_head_roots_kernel<<<nhb, kThreads, 0, st>>>(net->ttt, roots->pool->dev, net->ppart, out_value, out_logits,
                                                           roots->legal, roots->n_legal, roots->noise, roots->frac,
                                                           roots->out_parent, roots->out_action);
        else
            ttt_head_kernel<true, 4><<<nhb, kThreads, 0, st>>>(net->ttt, batch, nullptr, net->ppart, net->rpart, nullptr,
                                                              out_value, out_logits, out_value_raw);
    } else if (net->kind == MZ_NET_ATARI_SHAPED) {
        atari_kernel<true><<<(batch + 63) / 64, kThreadsW, 0, st>>>(net->atari, batch, observation, nullptr, 0, nullptr,
                                                                   out_hidden, nullptr, out_value, out_logits, out_value_raw);
    } else if (batch <= kCart16MaxBatch) {
        cartpole16_kernel<true><<<(batch + 15) / 16, kThreads16, 0, st>>>(net->cart, batch, observation, nullptr, 0, nullptr,
                                                                         out_hidden, nullptr, out_value, out_logits,
                                                                         out_value_raw);
    } else {
        if (rows == 64)
            cartpole_kernel<true, 2><<<nrb, kThreads, 0, st>>>(net->cart, batch, observation, nullptr, 0, nullptr,
                                                              out_hidden, nullptr, out_value, out_logits, out_value_raw);
        else
            cartpole_kernel<true, 1><<<nrb, kThreads, 0, st>>>(net->cart, batch, observation, nullptr, 0, nullptr,
                                                              out_hidden, nullptr, out_value, out_logits, out_value_raw);
    }
    return launch_status("initial inference");
}

int mz_infer_initial(mz_net_t* net, int32_t batch, const float* observation, float* out_hidden, float* out_value,
                     float* out_logits, float* out_value_raw) {
    return infer_initial(net, batch, observation, out_hidden, out_value, out_logits, out_value_raw, false);
}

// The TicTacToe job kernel, reading and writing f32 hidden rows or, inside a native move, their split images.
// Three shapes by batch size (16-row blocks per workgroup): up to kTttSmallBatch games 16 games x ONE cell (nine
// workgroups per 16 games, a head that sums nine partials), up to kTttMidBatch games 32 games x one cell (half the
// weight bytes per game through the L2, which is what bounds the single-cell jobs once most CUs have one), beyond
// that 64 games x a group of cells -- the shape of the one-launch move.  Measured (simulations/s at 25 simulations a move,
// 16 / 32 / 64 games): 512 games 22.0 / 20.0 / 11.4 M, 768: 29.5 / 29.2 / 18.1, 1024: 34.9 / 36.7 / 22.5, 1536: 41.4 /
// 46.9 / 35.9, 2048: 47.1 / 54.4 / 44.0, 2560: 50.8 / 59.2 / 58.6, 3072: 53.2 / 64.5 / 69.8, 4096: 56.7 / 70.5 / 90.4.
constexpr int kTttSmallBatch = 768, kTttMidBatch = 2560;
static int ttt_mb(const mz_net* net, int batch) {
    return net->job_rows == 16   ? 1
           : net->job_rows == 32 ? 2
           : net->job_rows == 64 ? 4
           : batch <= kTttSmallBatch ? 1
           : batch <= kTttMidBatch   ? 2
                                     : 4;
}

static void launch_job_kernel(mz_net* net, int batch, const float* hidden_base, const int* parent_block, long bs,
                              const int* action, float* out_hidden, bool split_hidden, float* ppart, float* rpart) {
    hipStream_t st = net->stream;
    const int mb = ttt_mb(net, batch), grid = ttt_job_grid(batch, mb);
    if (mb == 1) {
        if (split_hidden)
            ttt_ring_kernel<true, 1><<<grid, kThreadsW, 0, st>>>(net->ttt, batch, hidden_base, parent_block, bs, action,
                                                                 out_hidden, ppart, rpart);
        else
            ttt_ring_kernel<false, 1><<<grid, kThreadsW, 0, st>>>(net->ttt, batch, hidden_base, parent_block, bs, action,
                                                                  out_hidden, ppart, rpart);
    } else if (mb == 2) {
        if (split_hidden)
            ttt_ring_kernel<true, 2><<<grid, kThreadsW, 0, st>>>(net->ttt, batch, hidden_base, parent_block, bs, action,
                                                                 out_hidden, ppart, rpart);
        else
            ttt_ring_kernel<false, 2><<<grid, kThreadsW, 0, st>>>(net->ttt, batch, hidden_base, parent_block, bs, action,
                                                                  out_hidden, ppart, rpart);
    } else if (split_hidden) {
        ttt_ring_kernel<true, 4><<<grid, kThreadsW, 0, st>>>(net->ttt, batch, hidden_base, parent_block, bs, action,
                                                             out_hidden, ppart, rpart);
    } else {
        ttt_ring_kernel<false, 4><<<grid, kThreadsW, 0, st>>>(net->ttt, batch, hidden_base, parent_block, bs, action,
                                                              out_hidden, ppart, rpart);
    }
}

// split_hidden: the hidden rows read and written are the split images of a native move's workspace (TicTacToe)
static int infer_recurrent(mz_net_t* net, int32_t batch, const float* hidden_base, const int32_t* parent_block,
                           int64_t block_stride, const int32_t* action, float* out_hidden, float* out_reward,
                           float* out_value, float* out_logits, float* out_value_raw, bool split_hidden,
                           float* ppart = nullptr, float* rpart = nullptr) {
    // ppart / rpart: the TicTacToe kernels' scratch; NULL = the handle's own, (re)allocated here to fit the batch
    MZ_REQUIRE(net != nullptr, "net is NULL");
    MZ_REQUIRE(batch >= 0, "batch must be >= 0 (got %d)", batch);
    if (batch == 0) return MZ_OK;
    MZ_REQUIRE(hidden_base && action && out_hidden && out_reward && out_value && out_logits,
               "hidden_base/action/out_hidden/out_reward/out_value/out_logits is NULL");
    MZ_REQUIRE(block_stride >= 0, "block_stride must be >= 0");
    MZ_HIP(hipSetDevice(net->device));
    const int rows = tile_rows(batch);
    const int nrb = (batch + rows - 1) / rows;
    hipStream_t st = net->stream;
    const long bs = (long)block_stride;
    if (net->kind == MZ_NET_TICTACTOE) {
        if (!ppart) {
            int s = ensure_scratch(net, batch);
            if (s != MZ_OK) return s;
            ppart = net->ppart;
            rpart = net->rpart;
        }
        launch_job_kernel(net, batch, hidden_base, parent_block, bs, action, out_hidden, split_hidden, ppart, rpart);
        if (net->mid_event) MZ_HIP(hipEventRecord(net->mid_event, st));
        const int nhb = (batch + 15) / 16;
        if (ttt_mb(net, batch) != 4)
            ttt_head_kernel<false, 9><<<nhb, kThreads, 0, st>>>(net->ttt, batch, parent_block, ppart, rpart,
                                                               out_reward, out_value, out_logits, out_value_raw);
        else
            ttt_head_kernel<false, 6><<<nhb, kThreads, 0, st>>>(net->ttt, batch, parent_block, ppart, rpart,
                                                               out_reward, out_value, out_logits, out_value_raw);
    } else if (net->kind == MZ_NET_ATARI_SHAPED) {
        atari_kernel<false><<<(batch + 63) / 64, kThreadsW, 0, st>>>(net->atari, batch, hidden_base, parent_block, bs, action,
                                                                    out_hidden, out_reward, out_value, out_logits,
                                                                    out_value_raw);
    } else if (batch <= kCart16MaxBatch) {
        cartpole16_kernel<false><<<(batch + 15) / 16, kThreads16, 0, st>>>(net->cart, batch, hidden_base, parent_block, bs,
                                                                          action, out_hidden, out_reward, out_value,
                                                                          out_logits, out_value_raw);
    } else {
        if (rows == 64)
            cartpole_kernel<false, 2><<<nrb, kThreads, 0, st>>>(net->cart, batch, hidden_base, parent_block, bs, action,
                                                               out_hidden, out_reward, out_value, out_logits, out_value_raw);
        else
            cartpole_kernel<false, 1><<<nrb, kThreads, 0, st>>>(net->cart, batch, hidden_base, parent_block, bs, action,
                                                               out_hidden, out_reward, out_value, out_logits, out_value_raw);
    }
    return launch_status("recurrent inference");
}

int mz_infer_recurrent(mz_net_t* net, int32_t batch, const float* hidden_base, const int32_t* parent_block,
                       int64_t block_stride, const int32_t* action, float* out_hidden, float* out_reward,
                       float* out_value, float* out_logits, float* out_value_raw) {
    return infer_recurrent(net, batch, hidden_base, parent_block, block_stride, action, out_hidden, out_reward, out_value,
                           out_logits, out_value_raw, false);
}

// mz_infer_recurrent for callers that must not allocate (HIP graph capture, serving loops): the TicTacToe kernels' scratch
// is this device's copy of the file-scope `scratch` array, whose address mz_net_create looked up; no hipMalloc / hipFree
// or allocating library routine is reached from here.  Nothing in the scratch has to start at zero: the job kernel
// writes every partial the head kernel uses of an active row, and the head writes no output for the others.
int mz_infer_recurrent_noalloc(mz_net_t* net, int32_t batch, const float* hidden_base, const int32_t* parent_block,
                               int64_t block_stride, const int32_t* action, float* out_hidden, float* out_reward,
                               float* out_value, float* out_logits, float* out_value_raw) {
    MZ_REQUIRE(net != nullptr, "net is NULL");
    float* ppart = nullptr;
    float* rpart = nullptr;
    if (net->kind == MZ_NET_TICTACTOE) {
        MZ_REQUIRE(batch <= kNoAllocMaxBatch,
                   "mz_infer_recurrent_noalloc: batch %d exceeds the static scratch (at most %d games per call); use "
                   "mz_infer_recurrent or split the batch", batch, kNoAllocMaxBatch);
        MZ_REQUIRE(net->static_scratch != nullptr, "static scratch was not resolved for device %d", net->device);
        ppart = net->static_scratch;
        rpart = net->static_scratch + kScratchPpart;
    }
    return infer_recurrent(net, batch, hidden_base, parent_block, block_stride, action, out_hidden, out_reward, out_value,
                           out_logits, out_value_raw, false, ppart, rpart);
}

// Internal to the library (mz_pool_internal.h): the initial inference of a native move.  The move owns its hidden
// workspace, so the TicTacToe kernels keep it as split images from the root rows on (until mz_net_end_move).
__attribute__((visibility("hidden"))) int mz_net_begin_move(mz_net* net, int batch, const float* observation,
                                                            float* hidden_block0, float* out_value, float* out_logits) {
    const bool split = net != nullptr && net->kind == MZ_NET_TICTACTOE;
    const int s = infer_initial(net, batch, observation, hidden_block0, out_value, out_logits, nullptr, split);
    if (s == MZ_OK && batch > 0) net->hidden_split = split;
    return s;
}
// mz_net_begin_move followed by mz_expand_roots_noise with the first selection (logits = the inference's).  TicTacToe on a
// 16-lane pool: the head kernel of the initial inference goes on into the root expansion (ttt_head_roots_kernel).
__attribute__((visibility("hidden"))) int mz_net_begin_move_roots(mz_net* net, mz_pool* pool, int batch,
                                                                  const float* observation, float* hidden_block0,
                                                                  float* out_value, float* out_logits, const int32_t* legal,
                                                                  const int32_t* n_legal, const double* noise, double frac,
                                                                  int32_t* out_parent, int32_t* out_action) {
    MZ_REQUIRE(net != nullptr && pool != nullptr, "net/pool is NULL");
    MZ_REQUIRE(out_parent != nullptr && out_action != nullptr, "out_parent_block/out_action is NULL");
    const bool fused = net->kind == MZ_NET_TICTACTOE && pool->G == 16 && pool->P == 1 && pool->dev.A == 9 &&
                       pool->dev.T == batch && batch > 0 && pool->device == net->device && pool->stream == net->stream;
    if (!fused) {
        const int s = mz_net_begin_move(net, batch, observation, hidden_block0, out_value, out_logits);
        if (s != MZ_OK) return s;
        return mz_expand_roots_noise(pool, out_logits, 1, legal, n_legal, noise, frac, out_parent, out_action);
    }
    if (pool->dev.N < 1) return mz::fail(MZ_ERR_STATE, "pool is full: it holds no simulation");
    const RootsArgs roots{pool, legal, n_legal, noise, frac, out_parent, out_action};
    const int s = infer_initial(net, batch, observation, hidden_block0, out_value, out_logits, nullptr, true, &roots);
    if (s != MZ_OK) return s;
    net->hidden_split = true;
    pool->blocks_in_use = 1;     // (what expand_roots_impl leaves behind)
    pool->select_launches++;
    pool->pending = true;
    return MZ_OK;
}
__attribute__((visibility("hidden"))) void mz_net_end_move(mz_net* net) { net->hidden_split = false; }
__attribute__((visibility("hidden"))) int mz_net_hidden_size_of(const mz_net* net) { return mz_net_hidden_size(net->kind); }
__attribute__((visibility("hidden"))) int mz_net_num_actions_of(const mz_net* net) { return mz_net_num_actions(net->kind); }
__attribute__((visibility("hidden"))) hipStream_t mz_net_stream_of(const mz_net* net) { return net->stream; }

static bool run_fusable_cart(const mz_net* net, const mz_pool* pool) {
    // the trees of a workgroup's 16 games must fit its LDS for a whole move (the pool's max_simulations + 1 records)
    return net->kind == MZ_NET_CARTPOLE && pool->G == 2 && pool->dev.A == 2 && pool->dev.T <= kCart16MaxBatch &&
           pool->dev.off_back % 16 == 0 && cart_run_layout(pool->dev.N + 1, pool->dev.N).total <= kCartRunMaxDynLds;
}
static bool run_fusable_atari(const mz_net* net, const mz_pool* pool) {
    return net->kind == MZ_NET_ATARI_SHAPED && pool->dev.A == 18 && pool->dev.N + 2 <= kLogLds;
}
static bool run_fusable_ttt(const mz_net* net, const mz_pool* pool) {
    // the fused head + tree step is built for 16 lanes per tree; a row block's four workgroups must be co-resident, which
    // one 512-thread workgroup per CU always allows
    return net->kind == MZ_NET_TICTACTOE && net->run_mode != 2 && pool->G == 16 && pool->dev.A == 9 &&
           ttt_mb(net, pool->dev.T) == 4 &&
           (pool->dev.T + 63) / 64 <= kRunMaxRowBlocks;
}
__attribute__((visibility("hidden"))) int mz_net_run_is_one_launch(const mz_net* net, const mz_pool* pool) {
    return run_fusable_cart(net, pool) || run_fusable_atari(net, pool) || run_fusable_ttt(net, pool) ? 1 : 0;
}

int mz_debug_set_run_mode(mz_net_t* net, int32_t mode) {
    MZ_REQUIRE(net != nullptr, "net is NULL");
    MZ_REQUIRE(mode >= 0 && mode <= 2, "run mode must be 0, 1 or 2 (got %d)", mode);
    net->run_mode = mode;
    return MZ_OK;
}

int mz_debug_set_job_rows(mz_net_t* net, int32_t rows) {
    MZ_REQUIRE(net != nullptr, "net is NULL");
    MZ_REQUIRE(rows == 0 || rows == 16 || rows == 32 || rows == 64, "job rows must be 0, 16, 32 or 64 (got %d)", rows);
    net->job_rows = rows;
    return MZ_OK;
}

int mz_debug_run_info(mz_net_t* net, int64_t* out4_host) {
    MZ_REQUIRE(net != nullptr && out4_host != nullptr, "net/out4_host is NULL");
    for (int i = 0; i < 4; ++i) out4_host[i] = 0;
    if (!net->run_sync) return MZ_OK;
    MZ_HIP(hipSetDevice(net->device));
    MZ_HIP(hipStreamSynchronize(net->stream));
    unsigned long long host[4] = {0, 0, 0, 0};
    const char* base = reinterpret_cast<const char*>(net->run_sync) +
                       (size_t)kRunMaxRowBlocks * (3 * sizeof(unsigned long long) + sizeof(unsigned));
    MZ_HIP(hipMemcpy(host, base, sizeof(host), hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i) out4_host[i] = (int64_t)host[i];
    return MZ_OK;
}

int mz_infer_recurrent_run(mz_net_t* net, mz_pool_t* pool, int32_t num_simulations, int32_t final_select,
                           float* hidden_base, int32_t* parent_block, int64_t block_stride, int32_t* action,
                           float* out_reward, float* out_value, float* out_logits, int64_t out_stride,
                           int32_t* trace_parent, int32_t* trace_action) {
    MZ_REQUIRE(net != nullptr && pool != nullptr, "net/pool is NULL");
    MZ_REQUIRE(hidden_base && parent_block && action && out_reward && out_value && out_logits,
               "hidden_base/parent_block/action/out_reward/out_value/out_logits is NULL");
    MZ_REQUIRE(num_simulations >= 1, "num_simulations must be >= 1 (got %d)", num_simulations);
    MZ_REQUIRE(block_stride >= (int64_t)pool->dev.T * mz_net_hidden_size(net->kind),
               "block_stride must hold the hidden rows of all %d trees", pool->dev.T);
    MZ_REQUIRE(out_stride == 0 || out_stride >= pool->dev.T, "out_stride must be 0 or >= the number of trees");
    MZ_REQUIRE((trace_parent == nullptr) == (trace_action == nullptr), "trace_parent and trace_action go together");
    MZ_REQUIRE(net->device == pool->device && net->stream == pool->stream, "net and pool must share device and stream");
    const int T = pool->dev.T, A = pool->dev.A;
    if (run_fusable_cart(net, pool)) {
        int nb0 = 0;
        int s = mz_pool_begin_run(pool, num_simulations, final_select != 0, &nb0);
        if (s != MZ_OK) return s;
        MZ_HIP(hipSetDevice(net->device));
        const int dyn_lds = cart_run_layout(nb0 + num_simulations, pool->dev.N).total;
        MZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cartpole16_run_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kCartRunMaxDynLds));
        cartpole16_run_kernel<<<(T + 15) / 16, kRunThreads, dyn_lds, net->stream>>>(
            net->cart, pool->dev, nb0, num_simulations, final_select, hidden_base, parent_block, (long)block_stride, action,
            out_reward, out_value, out_logits, (long)out_stride, trace_parent, trace_action);
        s = launch_status("recurrent inference + tree step, all simulations");
        if (s != MZ_OK) return s;
        mz_pool_commit_run(pool, num_simulations, final_select != 0);
        return MZ_OK;
    }
    if (run_fusable_ttt(net, pool)) {
        int nb0 = 0;
        int s = mz_pool_begin_run(pool, num_simulations, final_select != 0, &nb0);
        if (s != MZ_OK) return s;
        MZ_HIP(hipSetDevice(net->device));
        s = ensure_scratch(net, T);
        if (s != MZ_OK) return s;
        const PoolDev* image = static_cast<const PoolDev*>(mz_pool_device_image(pool));
        if (!image) return MZ_ERR_HIP;
        RunSync sync;
        sync.arrive = net->run_sync;
        sync.cell_flags = net->run_sync + kRunMaxRowBlocks;                          // [row block][2]
        sync.xcc_mask = reinterpret_cast<unsigned*>(net->run_sync + 3 * kRunMaxRowBlocks);
        sync.stats = reinterpret_cast<unsigned long long*>(sync.xcc_mask + kRunMaxRowBlocks);
        sync.sel_words = sync.stats + 4;                                             // [row block][64]
        sync.tagged_ok = num_simulations < 4096 && nb0 + num_simulations < 4095 ? 1 : 0;
        sync.epoch_base = (++net->run_epoch) << kRunEpochShift;
        sync.force_fences = net->run_mode == 1;
        const int grid = 32 * (((T + 63) / 64 + 7) / 8);
        if (net->hidden_split)
            ttt_run_kernel<true><<<grid, kThreadsW, 0, net->stream>>>(
                net->ttt, image, T, pool->dev.err_flag, sync, nb0, num_simulations, final_select, hidden_base, parent_block, (long)block_stride,
                action, net->ppart, net->rpart, out_reward, out_value, out_logits, (long)out_stride, trace_parent,
                trace_action);
        else
            ttt_run_kernel<false><<<grid, kThreadsW, 0, net->stream>>>(
                net->ttt, image, T, pool->dev.err_flag, sync, nb0, num_simulations, final_select, hidden_base, parent_block, (long)block_stride,
                action, net->ppart, net->rpart, out_reward, out_value, out_logits, (long)out_stride, trace_parent,
                trace_action);
        s = launch_status("recurrent inference + tree step, all simulations");
        if (s != MZ_OK) return s;
        mz_pool_commit_run(pool, num_simulations, final_select != 0);
        return MZ_OK;
    }
    if (run_fusable_atari(net, pool)) {
        int nb0 = 0;
        int s = mz_pool_begin_run(pool, num_simulations, final_select != 0, &nb0);
        if (s != MZ_OK) return s;
        MZ_HIP(hipSetDevice(net->device));
        const PoolDev* image = static_cast<const PoolDev*>(mz_pool_device_image(pool));
        if (!image) return MZ_ERR_HIP;
        atari_run_kernel<<<(T + 63) / 64, kThreadsW, 0, net->stream>>>(
            net->atari, image, T, nb0, num_simulations, final_select, hidden_base, parent_block, (long)block_stride, action,
            out_reward, out_value, out_logits, (long)out_stride, trace_parent, trace_action);
        s = launch_status("recurrent inference + tree step, all simulations");
        if (s != MZ_OK) return s;
        mz_pool_commit_run(pool, num_simulations, final_select != 0);
        return MZ_OK;
    }
    for (int i = 0; i < num_simulations; ++i) {       // the same work, a simulation at a time
        if (trace_parent) {
            MZ_HIP(hipMemcpyAsync(trace_parent + (size_t)i * T, parent_block, (size_t)T * sizeof(int32_t),
                                  hipMemcpyDeviceToDevice, net->stream));
            MZ_HIP(hipMemcpyAsync(trace_action + (size_t)i * T, action, (size_t)T * sizeof(int32_t),
                                  hipMemcpyDeviceToDevice, net->stream));
        }
        const int nb = pool->blocks_in_use;
        const int s = mz_infer_recurrent_step(net, pool, (i + 1 < num_simulations || final_select) ? 1 : 0, hidden_base,
                                              parent_block, block_stride, action, hidden_base + (size_t)nb * block_stride,
                                              out_reward + (size_t)i * out_stride, out_value + (size_t)i * out_stride,
                                              out_logits + (size_t)i * out_stride * A);
        if (s != MZ_OK) return s;
    }
    return MZ_OK;
}

int mz_infer_recurrent_step(mz_net_t* net, mz_pool_t* pool, int32_t with_select, const float* hidden_base,
                            int32_t* parent_block, int64_t block_stride, int32_t* action, float* out_hidden,
                            float* out_reward, float* out_value, float* out_logits) {
    MZ_REQUIRE(net != nullptr && pool != nullptr, "net/pool is NULL");
    MZ_REQUIRE(hidden_base && parent_block && action && out_hidden && out_reward && out_value && out_logits,
               "hidden_base/parent_block/action/out_hidden/out_reward/out_value/out_logits is NULL");
    MZ_REQUIRE(block_stride >= 0, "block_stride must be >= 0");
    MZ_REQUIRE(net->device == pool->device && net->stream == pool->stream, "net and pool must share device and stream");
    const int batch = pool->dev.T;
    // the fused launches are built for the default lane counts; other pools run the same work as separate launches
    const bool fusable = net->kind == MZ_NET_TICTACTOE && pool->G == 16 && pool->dev.A == 9;
    const bool fusable_cart = net->kind == MZ_NET_CARTPOLE && pool->G == 2 && pool->dev.A == 2 && batch <= kCart16MaxBatch;
    if (fusable_cart) {                               // CartPole: network and tree step of 16 games in one workgroup
        int nb = 0;
        int s = mz_pool_begin_backprop(pool, with_select != 0, &nb);
        if (s != MZ_OK) return s;
        MZ_HIP(hipSetDevice(net->device));
        const int grid = (batch + 15) / 16;
        if (with_select)
            cartpole16_step_kernel<true><<<grid, kThreads16, 0, net->stream>>>(net->cart, pool->dev, nb, hidden_base,
                                                                              parent_block, (long)block_stride, action,
                                                                              out_hidden, out_reward, out_value, out_logits);
        else
            cartpole16_step_kernel<false><<<grid, kThreads16, 0, net->stream>>>(net->cart, pool->dev, nb, hidden_base,
                                                                               parent_block, (long)block_stride, action,
                                                                               out_hidden, out_reward, out_value, out_logits);
        if (net->mid_event) MZ_HIP(hipEventRecord(net->mid_event, net->stream));
        s = launch_status("recurrent inference + tree step");
        if (s != MZ_OK) return s;
        mz_pool_commit_backprop(pool, with_select != 0);
        return MZ_OK;
    }
    // the Atari-shaped network: any lanes per tree in the pool (the fused kernel brings its own: results do not depend on it)
    const bool fusable_atari = net->kind == MZ_NET_ATARI_SHAPED && pool->dev.A == 18 && pool->dev.N + 2 <= kLogLds;
    if (fusable_atari) {                              // network and tree step of 64 games in one workgroup
        int nb = 0;
        int s = mz_pool_begin_backprop(pool, with_select != 0, &nb);
        if (s != MZ_OK) return s;
        MZ_HIP(hipSetDevice(net->device));
        const int grid = (batch + 63) / 64;
        if (with_select)
            atari_step_kernel<true><<<grid, kThreadsW, 0, net->stream>>>(net->atari, pool->dev, nb, hidden_base, parent_block,
                                                                        (long)block_stride, action, out_hidden, out_reward,
                                                                        out_value, out_logits);
        else
            atari_step_kernel<false><<<grid, kThreadsW, 0, net->stream>>>(net->atari, pool->dev, nb, hidden_base, parent_block,
                                                                         (long)block_stride, action, out_hidden, out_reward,
                                                                         out_value, out_logits);
        if (net->mid_event) MZ_HIP(hipEventRecord(net->mid_event, net->stream));
        s = launch_status("recurrent inference + tree step");
        if (s != MZ_OK) return s;
        mz_pool_commit_backprop(pool, with_select != 0);
        return MZ_OK;
    }
    if (!fusable) {                                   // the same work as separate launches
        int s = infer_recurrent(net, batch, hidden_base, parent_block, block_stride, action, out_hidden, out_reward,
                                out_value, out_logits, nullptr, net->hidden_split);
        if (s != MZ_OK) return s;
        if (net->mid_event && net->kind != MZ_NET_TICTACTOE) MZ_HIP(hipEventRecord(net->mid_event, net->stream));
        return with_select ? mz_step_logits(pool, out_logits, out_reward, out_value, parent_block, action)
                           : mz_expand_backprop_logits(pool, out_logits, out_reward, out_value);
    }
    int nb = 0;
    int s = mz_pool_begin_backprop(pool, with_select != 0, &nb);
    if (s != MZ_OK) return s;
    MZ_HIP(hipSetDevice(net->device));
    s = ensure_scratch(net, batch);
    if (s != MZ_OK) return s;
    hipStream_t st = net->stream;
    launch_job_kernel(net, batch, hidden_base, parent_block, (long)block_stride, action, out_hidden, net->hidden_split,
                      net->ppart, net->rpart);
    if (net->mid_event) MZ_HIP(hipEventRecord(net->mid_event, st));
    if (ttt_mb(net, batch) != 4) {
        const int grid = (batch + 15) / 16;
        if (with_select)
            ttt_head_step_kernel<true, 9><<<grid, kThreadsW, 0, st>>>(net->ttt, pool->dev, nb, net->ppart, net->rpart,
                                                                     out_reward, out_value, out_logits, parent_block, action);
        else
            ttt_head_step_kernel<false, 9><<<grid, kThreadsW, 0, st>>>(net->ttt, pool->dev, nb, net->ppart, net->rpart,
                                                                      out_reward, out_value, out_logits, parent_block, action);
    } else {
        const int grid = 32 * (((batch + 63) / 64 + 7) / 8);
        if (with_select)
            ttt_head_step_kernel<true, 6><<<grid, kThreadsW, 0, st>>>(net->ttt, pool->dev, nb, net->ppart, net->rpart,
                                                                     out_reward, out_value, out_logits, parent_block, action);
        else
            ttt_head_step_kernel<false, 6><<<grid, kThreadsW, 0, st>>>(net->ttt, pool->dev, nb, net->ppart, net->rpart,
                                                                      out_reward, out_value, out_logits, parent_block, action);
    }
    s = launch_status("recurrent inference + tree step");
    if (s != MZ_OK) return s;
    mz_pool_commit_backprop(pool, with_select != 0);
    return MZ_OK;
}

}  // extern "C"
