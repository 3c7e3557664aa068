// Micro-benchmark: what is the READ-ONLY streaming ceiling of one MI355X?  The forward-statistics pass of K7, the
// cross-entropy forward and the gradient-norm pass read a tensor and write (almost) nothing; they measure 3.3-3.8 TB/s
// while read+write passes reach 5-6 TB/s.  This program reads a 411 MB bf16 tensor (three rotating copies, larger
// than the 256 MB Infinity Cache) with the cheapest possible consumer (an OR of the words; one 4-byte store per wave).
// Build: hipcc -O3 --offload-arch=gfx950 -o read_only read_only.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }

__device__ __forceinline__ void finish(u32x4 acc, unsigned* out) {
    unsigned v = acc[0] | acc[1] | acc[2] | acc[3];
    if (v == 0xdeadbeefu) out[blockIdx.x] = v;      // never true for the fill pattern: the loads cannot be dropped, nothing is written
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_flat(const u32x4* x, unsigned* out, int64_t nvec) {
    const int64_t v = (int64_t)blockIdx.x * (256 * U) + threadIdx.x;
    u32x4 a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = v + u * 256 < nvec ? ld<NT>(x + v + u * 256) : u32x4{0, 0, 0, 0};
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u) acc |= a[u];
    finish(acc, out);
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_stride(const u32x4* x, unsigned* out, int64_t nvec) {
    u32x4 acc = {0, 0, 0, 0};
    const int64_t step = (int64_t)gridDim.x * 256 * U;
    for (int64_t v = (int64_t)blockIdx.x * 256 * U + threadIdx.x; v < nvec; v += step) {
        u32x4 a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = v + u * 256 < nvec ? ld<NT>(x + v + u * 256) : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u) acc |= a[u];
    }
    finish(acc, out);
}

// block-private contiguous ranges (the K7 "ranges" shape)
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_ranges(const u32x4* x, unsigned* out, int64_t nvec) {
    u32x4 acc = {0, 0, 0, 0};
    const int64_t per_block = (nvec + gridDim.x - 1) / gridDim.x / 256 * 256 + 256;
    const int64_t v0 = blockIdx.x * per_block;
    int64_t v1 = v0 + per_block; if (v1 > nvec) v1 = nvec;
    for (int64_t v = v0 + threadIdx.x; v < v1; v += 256 * U) {
        u32x4 a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = v + u * 256 < v1 ? ld<NT>(x + v + u * 256) : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u) acc |= a[u];
    }
    finish(acc, out);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main() {
    const int64_t n = 802816LL * 256, nvec = n / 8;
    const int NS = 3;
    std::vector<u32x4*> X(NS);
    for (int i = 0; i < NS; ++i) { CK(hipMalloc(&X[i], n * 2)); CK(hipMemset(X[i], 0x3c, n * 2)); }
    unsigned* out; CK(hipMalloc(&out, 1 << 22));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch(i % NS);
        CK(hipDeviceSynchronize());
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 9; ++i) launch(i % NS);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 9; if (ms < best) best = ms;
        }
        printf("%-40s %8.1f us  %5.2f TB/s\n", name, best * 1e3, n * 2 / (best * 1e-3) / 1e12);
        fflush(stdout);
    };
#define FLAT(U, NT) run("flat   U=" #U " nt=" #NT, [&](int i) { \
        hipLaunchKernelGGL((k_flat<U, NT>), dim3((unsigned)((nvec + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, X[i], out, nvec); })
#define STRIDE(U, NT, G) run("stride U=" #U " nt=" #NT " grid=" #G, [&](int i) { \
        hipLaunchKernelGGL((k_stride<U, NT>), dim3(G), dim3(256), 0, 0, X[i], out, nvec); })
#define RANGES(U, NT, G) run("ranges U=" #U " nt=" #NT " grid=" #G, [&](int i) { \
        hipLaunchKernelGGL((k_ranges<U, NT>), dim3(G), dim3(256), 0, 0, X[i], out, nvec); })
    FLAT(1, false); FLAT(2, false); FLAT(4, false); FLAT(8, false); FLAT(4, true); FLAT(8, true);
    STRIDE(4, false, 768); STRIDE(4, false, 2048); STRIDE(4, false, 8192); STRIDE(8, false, 2048); STRIDE(4, true, 2048); STRIDE(8, true, 8192);
    RANGES(4, false, 768); RANGES(4, false, 2048); RANGES(4, false, 8192); RANGES(8, true, 2048);
    return 0;
}
