// Micro-benchmark: how should a bf16 per-channel affine (+ residual) pass over an [M, C] tensor far larger than the
// Infinity Cache be shaped to reach the HBM streaming rate?  Build: hipcc -O3 --offload-arch=gfx950 -o stream_variants
// Variants: grid-stride loop with 768 blocks (the K7 shape) vs one small block per contiguous piece (what the
// framework's elementwise kernels do), rows in flight per thread, non-temporal loads / stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ void cvt_in(u32x4 r, float (&v)[8]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[2 * k] = __uint_as_float(r[k] << 16); v[2 * k + 1] = __uint_as_float(r[k] & 0xffff0000u); }
}
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 b2; typedef __attribute__((ext_vector_type(2))) float f2;
    f2 v = {lo, hi}; b2 b = __builtin_convertvector(v, b2); return __builtin_bit_cast(unsigned, b);
}
__device__ __forceinline__ u32x4 cvt_out(const float (&v)[8]) {
    u32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = pack2(v[2 * k], v[2 * k + 1]);
    return r;
}

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// A: grid-stride, block walks contiguous pieces of 256 x 16 B, U pieces in flight
template <int U, bool RES, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_stride(const u32x4* x, const u32x4* res, u32x4* y, const float* sc, const float* sh,
                                                int64_t nvec, int C) {
    const int c0 = (threadIdx.x * 8) & (C - 1);
    float s[8], h[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s[k] = sc[c0 + k]; h[k] = sh[c0 + k]; }
    const int64_t per_block = (nvec + gridDim.x - 1) / gridDim.x / 256 * 256 + 256;
    const int64_t v0 = blockIdx.x * per_block;
    int64_t v1 = v0 + per_block; if (v1 > nvec) v1 = nvec;
    for (int64_t v = v0 + threadIdx.x; v < v1; v += 256 * U) {
        u32x4 a[U], r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (v + u * 256 < v1) { a[u] = ld<NTL>(x + v + u * 256); if (RES) r[u] = ld<NTL>(res + v + u * 256); }
#pragma unroll
        for (int u = 0; u < U; ++u) if (v + u * 256 < v1) {
            float f[8], g[8];
            cvt_in(a[u], f);
            if (RES) cvt_in(r[u], g);
#pragma unroll
            for (int k = 0; k < 8; ++k) { f[k] = fmaf(f[k], s[k], h[k]); if (RES) f[k] += g[k]; f[k] = fmaxf(f[k], 0.f); }
            st<NTS>(y + v + u * 256, cvt_out(f));
        }
    }
}

// B: one block per U contiguous pieces, no loop (huge grid)
template <int U, bool RES, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_flat(const u32x4* x, const u32x4* res, u32x4* y, const float* sc, const float* sh,
                                              int64_t nvec, int C) {
    const int c0 = (threadIdx.x * 8) & (C - 1);
    float s[8], h[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s[k] = sc[c0 + k]; h[k] = sh[c0 + k]; }
    const int64_t v = (int64_t)blockIdx.x * (256 * U) + threadIdx.x;
    u32x4 a[U], r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (v + u * 256 < nvec) { a[u] = ld<NTL>(x + v + u * 256); if (RES) r[u] = ld<NTL>(res + v + u * 256); }
#pragma unroll
    for (int u = 0; u < U; ++u) if (v + u * 256 < nvec) {
        float f[8], g[8];
        cvt_in(a[u], f);
        if (RES) cvt_in(r[u], g);
#pragma unroll
        for (int k = 0; k < 8; ++k) { f[k] = fmaf(f[k], s[k], h[k]); if (RES) f[k] += g[k]; f[k] = fmaxf(f[k], 0.f); }
        st<NTS>(y + v + u * 256, cvt_out(f));
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int64_t M = 802816; const int C = 256; const int64_t n = M * C, nvec = n / 8;
    const int NS = 3;
    std::vector<u32x4*> X(NS), R(NS), Y(NS);
    for (int i = 0; i < NS; ++i) { CK(hipMalloc(&X[i], n * 2)); CK(hipMalloc(&R[i], n * 2)); CK(hipMalloc(&Y[i], n * 2));
        CK(hipMemset(X[i], 0x3c, n * 2)); CK(hipMemset(R[i], 0x3d, n * 2)); }
    float *sc, *sh; CK(hipMalloc(&sc, C * 4)); CK(hipMalloc(&sh, C * 4)); CK(hipMemset(sc, 0, C * 4)); CK(hipMemset(sh, 0, C * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, int streams, auto launch) {
        for (int i = 0; i < 3; ++i) launch(i % NS);
        CK(hipDeviceSynchronize());
        float best = 1e9, tot = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 9; ++i) launch(i % NS);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 9; tot += ms; if (ms < best) best = ms;
        }
        printf("%-44s %8.1f us  %5.2f TB/s\n", name, best * 1e3, streams * n * 2 / (best * 1e-3) / 1e12);
        fflush(stdout);
    };
#define STRIDE(U, RES, NTL, NTS, G) run("stride U=" #U " res=" #RES " ntl=" #NTL " nts=" #NTS " grid=" #G, (RES) ? 3 : 2, [&](int i) { \
        hipLaunchKernelGGL((k_stride<U, RES, NTL, NTS>), dim3(G), dim3(256), 0, 0, X[i], R[i], Y[i], sc, sh, nvec, C); })
#define FLAT(U, RES, NTL, NTS) run("flat   U=" #U " res=" #RES " ntl=" #NTL " nts=" #NTS, (RES) ? 3 : 2, [&](int i) { \
        hipLaunchKernelGGL((k_flat<U, RES, NTL, NTS>), dim3((unsigned)((nvec + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, X[i], R[i], Y[i], sc, sh, nvec, C); })
    run("hipMemcpyDtoD", 2, [&](int i) { CK(hipMemcpyAsync(Y[i], X[i], n * 2, hipMemcpyDeviceToDevice, 0)); });
    STRIDE(4, false, false, false, 768); STRIDE(4, true, false, false, 768);
    STRIDE(2, true, false, false, 768); STRIDE(8, true, false, false, 768);
    STRIDE(4, true, false, false, 2048); STRIDE(4, true, false, false, 8192); STRIDE(2, true, false, false, 4096);
    STRIDE(4, true, true, true, 768); STRIDE(4, true, false, true, 768); STRIDE(4, true, true, false, 768);
    FLAT(1, true, false, false); FLAT(2, true, false, false); FLAT(4, true, false, false); FLAT(8, true, false, false);
    FLAT(4, true, true, true); FLAT(4, true, false, true); FLAT(2, true, false, true); FLAT(4, false, false, false); FLAT(4, false, false, true);
    return 0;
}
