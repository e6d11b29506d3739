// Follow-up to inv_probe.hip: which load flavours see another CU's store (same XCD / other XCD) without invalidating
// anything, and what they and buffer_inv sc1 cost.  One lane per workgroup; times in shader clocks (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int F>
__device__ __forceinline__ int load_f(const int* p) {
    int v;
    if (F == 0) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (F == 1) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (F == 2) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (F == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (F == 4) asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int F>
__global__ void visible(int* data, unsigned long long* flag, int* out, int writer) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    if (blockIdx.x == 0) {
        const int v0 = load_f<0>(data);                  // the line is in this CU's L1 now
        __hip_atomic_fetch_add(flag, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 100000000) { out[2] = -1; return; }
        }
        out[0] = v0;
        out[1] = load_f<F>(data);
        out[5] = load_f<0>(data);                        // and a plain load after it: did the flavoured load refresh L1?
    } else if ((int)blockIdx.x == writer) {
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 100000000) { out[2] = -2; return; }
        }
        asm volatile("global_store_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" ::"v"(data), "v"(42) : "memory");
        __hip_atomic_fetch_add(flag + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// latency of a dependent chain of loads over `n` lines (stride 4 KiB apart: no two in one L1 line), flavour F, second pass
template <int F>
__global__ void latency(int* chain, int n, int passes, long long* out, int inv_between) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int idx = 0;
    for (int i = 0; i < n; ++i) idx = load_f<0>(chain + idx);      // warm: lines in L1 and L2
    long long total = 0;
    for (int p = 0; p < passes; ++p) {
        if (inv_between == 1) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        const long long a = clock64();
        idx = 0;
        for (int i = 0; i < n; ++i) idx = load_f<F>(chain + idx);
        total += clock64() - a;
    }
    out[0] = total;
    out[1] = idx;
}

__global__ void inv_cost(long long* out, int n) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const long long a = clock64();
    for (int i = 0; i < n; ++i) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
    out[0] = clock64() - a;
}

int main() {
    int *data, *out, *chain;
    unsigned long long* flag;
    long long* lat;
    hipMalloc(&data, 4096);
    hipMalloc(&out, 64);
    hipMalloc(&flag, 64);
    hipMalloc(&lat, 64);
    const int n = 64, stride = 1024;                     // ints: 4 KiB apart
    hipMalloc(&chain, n * stride * 4);
    {
        int* h = new int[n * stride]();
        for (int i = 0; i < n; ++i) h[i * stride] = ((i + 1) % n) * stride;
        hipMemcpy(chain, h, n * stride * 4, hipMemcpyHostToDevice);
    }
    const char* names[5] = {"plain", "sc0", "sc1", "sc0 sc1", "nt"};
    for (int writer : {8, 1}) {
        for (int f = 0; f < 5; ++f) {
            hipMemset(data, 0, 4096);
            hipMemset(out, 0, 64);
            hipMemset(flag, 0, 64);
            if (f == 0) visible<0><<<16, 64>>>(data, flag, out, writer);
            if (f == 1) visible<1><<<16, 64>>>(data, flag, out, writer);
            if (f == 2) visible<2><<<16, 64>>>(data, flag, out, writer);
            if (f == 3) visible<3><<<16, 64>>>(data, flag, out, writer);
            if (f == 4) visible<4><<<16, 64>>>(data, flag, out, writer);
            int h[6];
            hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            printf("writer on %s XCD, re-read with %-8s: %d %s; plain load after it: %d%s\n", writer == 8 ? "the same" : "another",
                   names[f], h[1], h[1] == 42 ? "fresh" : "STALE", h[5], h[2] ? "  TIMEOUT" : "");
        }
    }
    long long h[2];
    for (int f = 0; f < 5; ++f) {
        for (int inv = 0; inv < 2; ++inv) {
            if (f == 0) latency<0><<<1, 64>>>(chain, n, 20, lat, inv);
            if (f == 1) latency<1><<<1, 64>>>(chain, n, 20, lat, inv);
            if (f == 2) latency<2><<<1, 64>>>(chain, n, 20, lat, inv);
            if (f == 3) latency<3><<<1, 64>>>(chain, n, 20, lat, inv);
            if (f == 4) latency<4><<<1, 64>>>(chain, n, 20, lat, inv);
            hipMemcpy(h, lat, sizeof(h), hipMemcpyDeviceToHost);
            printf("dependent load, %-8s%s: %6.0f clocks\n", names[f], inv ? ", buffer_inv sc1 before each pass" : "",
                   (double)h[0] / (20.0 * n));
        }
    }
    inv_cost<<<1, 64>>>(lat, 100);
    hipMemcpy(h, lat, sizeof(h), hipMemcpyDeviceToHost);
    printf("buffer_inv sc1 + wait: %6.0f clocks each\n", (double)h[0] / 100.0);
    return 0;
}
