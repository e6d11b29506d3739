// Profiling aid (not part of the library): what one CU sustains on the inner loop of the split-f16 network kernels.
// A 512-thread workgroup per CU runs "stages" of 4 k-steps x (8 ds_read_b128 + 12 v_mfma_f32_16x16x32_f16 + 2 global
// 16-byte loads), the instruction mix of stage64 / ttt_job_kernel_w (mz_net.hip), with parts of the mix switched off:
//   mode bit 0 = MFMAs, bit 1 = LDS fragment reads, bit 2 = weight loads from global memory, bit 3 = the gather of the
//   next activation tile (4 global 16-byte loads per lane and stage, rows of 512 bytes 4608 bytes apart, then 4
//   ds_write_b128).
// Prints shader clocks per stage (s_memtime), the shader clock rate seen (s_memtime / s_memrealtime at 100 MHz) and the
// f16 MFMA rate, for one and two wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr int kLdW = 272;

template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void probe(const float4* __restrict__ wts, const float4* __restrict__ rows,
                                                    int stages, float* out, unsigned long long* clocks) {
    __shared__ __attribute__((aligned(16))) _Float16 tile[2][64 * kLdW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n16 = lane & 15, g = lane >> 4;
    for (int i = tid; i < 2 * 64 * kLdW; i += THREADS) (&tile[0][0])[i] = (_Float16)(0.001f * (i & 63));
    __syncthreads();
    f32x4 acc[4], acx[4];
    for (int m = 0; m < 4; ++m) acc[m] = acx[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 bw[8];
    const float4* wl = wts + wave * 64 + lane;
    for (int j = 0; j < 8; ++j) bw[j] = wl[j * 512];
    float4 af[2][4][2];
    for (int m = 0; m < 4; ++m) af[0][m][0] = af[0][m][1] = af[1][m][0] = af[1][m][1] = make_float4(1e-3f, 2e-3f, 1e-3f, 0.f);
    float4 stg[4];
    const float4* gp[4];
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + THREADS * i, r = (idx >> 5) & 63, c4 = idx & 31;
        gp[i] = rows + ((size_t)blockIdx.x * 64 + r) * 288 + c4;           // 1152 floats per game
        stg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
#pragma unroll 1
    for (int st = 0; st < stages; ++st) {
        const _Float16* ap = tile[st & 1] + n16 * kLdW + 8 * g;
        const float4* wnext = wl + (size_t)((st + 1) & 15) * 4096;
        if (MODE & 2) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                af[0][m][0] = *reinterpret_cast<const float4*>(ap + m * 16 * kLdW);
                af[0][m][1] = *reinterpret_cast<const float4*>(ap + m * 16 * kLdW + 128);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if ((MODE & 2) && s + 1 < 4) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    af[(s + 1) & 1][m][0] = *reinterpret_cast<const float4*>(ap + m * 16 * kLdW + 32 * (s + 1));
                    af[(s + 1) & 1][m][1] = *reinterpret_cast<const float4*>(ap + m * 16 * kLdW + 128 + 32 * (s + 1));
                }
            }
            const f16x8 bh = __builtin_bit_cast(f16x8, bw[2 * s]), bl = __builtin_bit_cast(f16x8, bw[2 * s + 1]);
            if (MODE & 1) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const f16x8 ah = __builtin_bit_cast(f16x8, af[s & 1][m][0]), al = __builtin_bit_cast(f16x8, af[s & 1][m][1]);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah, acc[m], 0, 0, 0);
                    acx[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah, acx[m], 0, 0, 0);
                    acx[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al, acx[m], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) {       // keep the operands alive
                    acc[m][0] += af[s & 1][m][0].x + bw[2 * s].x;
                    acx[m][0] += af[s & 1][m][1].y + bw[2 * s + 1].y;
                }
            }
            if (MODE & 4) {
                bw[2 * s] = wnext[(2 * s) * 512];
                bw[2 * s + 1] = wnext[(2 * s + 1) * 512];
            }
            if ((MODE & 2) && s + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            if (MODE & 1) __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
            if (MODE & 4) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            if ((MODE & 8) && s == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int idx = tid + THREADS * i, r = (idx >> 5) & 63, c4 = idx & 31;
                    *reinterpret_cast<float4*>(tile[(st + 1) & 1] + r * kLdW + c4 * 8) = stg[i];
                }
            }
            if ((MODE & 8) && s == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) stg[i] = gp[i][(st % 9) * 32];
                __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
            }
        }
        if (MODE & 8) __syncthreads();
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0.f;
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3] + acx[m][0] + acx[m][1] + acx[m][2] + acx[m][3];
    out[(size_t)blockIdx.x * THREADS + tid] = s;
    if (lane == 0) {
        clocks[((size_t)blockIdx.x * (THREADS / 64) + wave) * 2] = t1 - t0;
        clocks[((size_t)blockIdx.x * (THREADS / 64) + wave) * 2 + 1] = r1 - r0;
    }
}

template <int MODE, int THREADS>
void run(const char* what, const float4* wts, const float4* rows, float* out, unsigned long long* clocks, int grid) {
    const int stages = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<MODE, THREADS><<<grid, THREADS>>>(wts, rows, stages, out, clocks);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE, THREADS><<<grid, THREADS>>>(wts, rows, stages, out, clocks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2];
    hipMemcpy(c, clocks, sizeof(c), hipMemcpyDeviceToHost);
    const double clk = (double)c[0] / stages, ghz = (double)c[0] / ((double)c[1] * 10.0);
    const double flops = (double)grid * (THREADS / 64) * stages * 48.0 * 16384.0;
    printf("%-34s %d waves/SIMD: %7.0f clk/stage  %.2f GHz  %8.3f ms", what, THREADS / 256, clk, ghz, ms);
    if (MODE & 1) printf("  %7.1f TFLOP/s f16 executed (%.1f clk per MFMA and SIMD)", flops / (ms * 1e-3) / 1e12,
                         clk / (48.0 * (THREADS / 256)));
    printf("\n");
    fflush(stdout);
}


// The same stage with four wavefronts per workgroup, each owning 32 output columns of all 64 rows (one wavefront per
// SIMD, twice the accumulators and weight fragments per wavefront): every wavefront still reads the whole tile, so the
// LDS bytes per stage halve (4 x 32 KB) while the MFMA work per SIMD stays what it was.
template <int MODE>
__global__ __launch_bounds__(256, 1) void probe_wide(const float4* __restrict__ wts, int stages, float* out,
                                                     unsigned long long* clocks) {
    __shared__ __attribute__((aligned(16))) _Float16 tile[2][64 * kLdW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n16 = lane & 15, g = lane >> 4;
    for (int i = tid; i < 2 * 64 * kLdW; i += 256) (&tile[0][0])[i] = (_Float16)(0.001f * (i & 63));
    __syncthreads();
    f32x4 acc[2][4], acx[2][4];
    for (int c = 0; c < 2; ++c)
        for (int m = 0; m < 4; ++m) acc[c][m] = acx[c][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 bw[2][8];
    const float4* wl = wts + wave * 128 + lane;
    for (int c = 0; c < 2; ++c)
        for (int j = 0; j < 8; ++j) bw[c][j] = wl[j * 512 + c * 64];
    float4 af[2][4][2];
    for (int m = 0; m < 4; ++m) af[0][m][0] = af[0][m][1] = af[1][m][0] = af[1][m][1] = make_float4(1e-3f, 2e-3f, 1e-3f, 0.f);
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
#pragma unroll 1
    for (int st = 0; st < stages; ++st) {
        const _Float16* ap = tile[st & 1] + n16 * kLdW + 8 * g;
        const float4* wnext = wl + (size_t)((st + 1) & 15) * 4096;
        if (MODE & 2) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                af[0][m][0] = *reinterpret_cast<const float4*>(ap + m * 16 * kLdW);
                af[0][m][1] = *reinterpret_cast<const float4*>(ap + m * 16 * kLdW + 128);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if ((MODE & 2) && s + 1 < 4) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    af[(s + 1) & 1][m][0] = *reinterpret_cast<const float4*>(ap + m * 16 * kLdW + 32 * (s + 1));
                    af[(s + 1) & 1][m][1] = *reinterpret_cast<const float4*>(ap + m * 16 * kLdW + 128 + 32 * (s + 1));
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f16x8 bh = __builtin_bit_cast(f16x8, bw[c][2 * s]), bl = __builtin_bit_cast(f16x8, bw[c][2 * s + 1]);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const f16x8 ah = __builtin_bit_cast(f16x8, af[s & 1][m][0]), al = __builtin_bit_cast(f16x8, af[s & 1][m][1]);
                    acc[c][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah, acc[c][m], 0, 0, 0);
                    acx[c][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah, acx[c][m], 0, 0, 0);
                    acx[c][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al, acx[c][m], 0, 0, 0);
                }
            }
            if (MODE & 4) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    bw[c][2 * s] = wnext[(2 * s) * 512 + c * 64];
                    bw[c][2 * s + 1] = wnext[(2 * s + 1) * 512 + c * 64];
                }
            }
            if ((MODE & 2) && s + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
            if (MODE & 4) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float sum = 0.f;
    for (int c = 0; c < 2; ++c)
        for (int m = 0; m < 4; ++m) sum += acc[c][m][0] + acc[c][m][1] + acc[c][m][2] + acc[c][m][3] + acx[c][m][0] + acx[c][m][1] + acx[c][m][2] + acx[c][m][3];
    out[(size_t)blockIdx.x * 256 + tid] = sum;
    if (lane == 0) {
        clocks[((size_t)blockIdx.x * 4 + wave) * 2] = t1 - t0;
        clocks[((size_t)blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0;
    }
}

template <int MODE>
void run_wide(const char* what, const float4* wts, float* out, unsigned long long* clocks, int grid) {
    const int stages = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe_wide<MODE><<<grid, 256>>>(wts, stages, out, clocks);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe_wide<MODE><<<grid, 256>>>(wts, stages, out, clocks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * stages * 96.0 * 16384.0;
    printf("%-34s 4 waves x 32 columns: %8.3f ms = %6.0f ns per stage  %7.1f TFLOP/s f16 executed\n", what, ms,
           ms * 1e6 / stages, flops / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

int main() {
    float4* wts;
    float* out;
    unsigned long long* clocks;
    hipMalloc(&wts, (size_t)17 * 4096 * sizeof(float4));
    hipMemset(wts, 0, (size_t)17 * 4096 * sizeof(float4));
    hipMalloc(&out, (size_t)256 * 512 * sizeof(float));
    hipMalloc(&clocks, (size_t)256 * 8 * 2 * sizeof(unsigned long long));
    const int grid = 256;
    float4* rows;
    hipMalloc(&rows, (size_t)grid * 64 * 1152 * sizeof(float));
    hipMemset(rows, 0, (size_t)grid * 64 * 1152 * sizeof(float));
    run<1, 256>("MFMA only", wts, rows, out, clocks, grid);
    run<1, 512>("MFMA only", wts, rows, out, clocks, grid);
    run<2, 512>("LDS reads only", wts, rows, out, clocks, grid);
    run<4, 512>("weight loads only", wts, rows, out, clocks, grid);
    run<3, 256>("MFMA + LDS reads", wts, rows, out, clocks, grid);
    run<3, 512>("MFMA + LDS reads", wts, rows, out, clocks, grid);
    run<5, 512>("MFMA + weight loads", wts, rows, out, clocks, grid);
    run<7, 256>("MFMA + LDS reads + weight loads", wts, rows, out, clocks, grid);
    run<7, 512>("MFMA + LDS reads + weight loads", wts, rows, out, clocks, grid);
    run_wide<1>("MFMA only", wts, out, clocks, grid);
    run_wide<3>("MFMA + LDS reads", wts, out, clocks, grid);
    run_wide<7>("MFMA + LDS reads + weight loads", wts, out, clocks, grid);
    run<8, 512>("gather + barrier only", wts, rows, out, clocks, grid);
    run<12, 512>("weights + gather", wts, rows, out, clocks, grid);
    run<15, 512>("everything + gather + barrier", wts, rows, out, clocks, grid);
    return 0;
}
