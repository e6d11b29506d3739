// Profiling aid (not part of the library): latency of a dependent chain of scattered record reads, the access
// pattern of a tree descent.  Each wavefront reads a `rec_bytes` record (lanes read consecutive 8-byte words), takes
// the next record index from it, and repeats `depth` times.  Reports ns per dependent step for several wave counts
// and pool sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

__global__ void chase(const unsigned long long* buf, size_t nrec, int words, int depth, unsigned long long* out) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
    const int lane = threadIdx.x & 63;
    size_t idx = (size_t)wave * 7919u % nrec;
    unsigned long long acc = 0;
    for (int d = 0; d < depth; ++d) {
        const unsigned long long v = buf[idx * words + (lane % words)];
        acc += v;
        // next index comes from word 0 of the record: dependent
        const unsigned long long w0 = __shfl(v, 0, 64);
        idx = (size_t)(w0 % nrec);
    }
    if (acc == 0x123456789ull) out[0] = acc;
}

int main() {
    const int words_list[] = {8, 48};          // 64-byte and 384-byte records
    const size_t pool_mb[] = {64, 640, 8192};
    const int waves_list[] = {1024, 4096, 8192};
    for (int words : words_list)
        for (size_t mb : pool_mb) {
            const size_t nrec = mb * 1024 * 1024 / (words * 8);
            unsigned long long* buf;
            if (hipMalloc(&buf, nrec * words * 8) != hipSuccess) { printf("alloc fail\n"); return 1; }
            std::vector<unsigned long long> h(nrec * words);
            std::mt19937_64 rng(1);
            for (size_t i = 0; i < nrec * words; ++i) h[i] = rng();
            hipMemcpy(buf, h.data(), h.size() * 8, hipMemcpyHostToDevice);
            unsigned long long* out;
            hipMalloc(&out, 8);
            for (int waves : waves_list) {
                const int depth = 64;
                hipEvent_t e0, e1;
                hipEventCreate(&e0);
                hipEventCreate(&e1);
                chase<<<waves / 4, 256>>>(buf, nrec, words, depth, out);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                chase<<<waves / 4, 256>>>(buf, nrec, words, depth, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                printf("record %4d B  pool %5zu MB  waves %5d: %7.1f ns per dependent step  (%.2f TB/s)\n", words * 8, mb,
                       waves, ms * 1e6 / depth, (double)waves * words * 8 * depth / (ms * 1e-3) / 1e12);
                fflush(stdout);
            }
            hipFree(buf);
            hipFree(out);
        }
    return 0;
}
