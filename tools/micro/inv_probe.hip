// Does `buffer_inv sc0` drop a CU's L1 lines (no threadgroup-split mode)?  Workgroup 0 (reader) and workgroup 8 (writer)
// run on one XCD (blockIdx % 8) but on different CUs.  The reader caches a line, the writer overwrites it with a plain
// store and signals through atomics, the reader re-reads it with a plain load after (mode 0) nothing, (mode 1)
// buffer_inv sc0, (mode 2) buffer_inv sc1.  Expect: mode 0 stale (the hazard is real), modes 1 and 2 fresh.
//   hipcc -O2 --offload-arch=gfx950 -o tools/micro/inv_probe tools/micro/inv_probe.hip && tools/micro/inv_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ int plain_load(const int* p) {
    int v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

__global__ void probe(int* data, unsigned long long* flag, int* out, int mode, int reader, int writer) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    if ((int)blockIdx.x == reader) {
        out[3] = __builtin_amdgcn_s_getreg(6164) & 15;
        const int v0 = plain_load(data);                 // the line is in this CU's L1 now
        __hip_atomic_fetch_add(flag, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 100000000) { out[2] = -1; return; }
        }
        if (mode == 1) asm volatile("buffer_inv sc0" ::: "memory");
        if (mode == 2) asm volatile("buffer_inv sc1" ::: "memory");
        const int v1 = plain_load(data);
        out[0] = v0;
        out[1] = v1;
    } else if ((int)blockIdx.x == writer) {
        out[4] = __builtin_amdgcn_s_getreg(6164) & 15;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 100000000) { out[2] = -2; return; }
        }
        asm volatile("global_store_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" ::"v"(data), "v"(42) : "memory");
        __hip_atomic_fetch_add(flag + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main() {
    int *data, *out;
    unsigned long long* flag;
    hipMalloc(&data, 4096);
    hipMalloc(&out, 64);
    hipMalloc(&flag, 64);
    for (int writer : {8, 1}) {                            // same XCD as the reader; the next XCD
        for (int mode = 0; mode < 3; ++mode) {
            hipMemset(data, 0, 4096);
            hipMemset(out, 0, 64);
            hipMemset(flag, 0, 64);
            probe<<<16, 64>>>(data, flag, out, mode, 0, writer);
            int h[5];
            hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            printf("writer block %d (xcc %d, reader xcc %d) mode %d (%s): first read %d, re-read %d%s\n", writer, h[4], h[3], mode,
                   mode == 0 ? "no invalidate" : mode == 1 ? "buffer_inv sc0" : "buffer_inv sc1", h[0], h[1],
                   h[2] ? "  TIMEOUT" : h[1] == 42 ? "  fresh" : "  STALE");
        }
    }
    return 0;
}
