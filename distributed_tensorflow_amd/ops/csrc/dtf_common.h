// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels of this package.
// Wave = 64 lanes; every memory-bound kernel moves 16 B per lane per access.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#define DTF_WAVE 64

// Debug build (`python -m distributed_tensorflow_amd.ops.build --debug-bounds` -> libdtf_hip_debug.so, selected with
// DTF_HIP_LIB): DTF_IDX(i, limit, what) checks 0 <= i < limit at the places where an index is COMPUTED FROM DATA or from
// a launch-geometry assumption (gathered rows, sorted-run boundaries, routed slots, tile edges, pooled windows).  A
// violation prints one greppable line and the access is redirected to element 0, so the run finishes and reports
// instead of faulting the device (no sanitizer / debugger runs on the GPU pool).  Release builds: the plain index.
#ifdef DTF_DEBUG_BOUNDS
#include <stdio.h>
__device__ __forceinline__ long long dtf_idx_check(long long i, long long limit, const char* what, int line) {
    if (i < 0 || i >= limit) {
        printf("DTF_BOUNDS_VIOLATION %s line %d: index %lld not in [0, %lld) block (%u,%u) thread %u\n", what, line, i,
               limit, blockIdx.x, blockIdx.y, threadIdx.x);
        return 0;
    }
    return i;
}
#define DTF_IDX(i, limit, what) dtf_idx_check((long long)(i), (long long)(limit), what, __LINE__)
#else
#define DTF_IDX(i, limit, what) (i)
#endif

enum DtfDtype : int { DTF_F32 = 0, DTF_BF16 = 1 };

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __forceinline__ float bf16_to_f32(unsigned short u) {
    return __uint_as_float(((unsigned int)u) << 16);
}

// round-to-nearest-even, NaN stays NaN (quiet).  hipcc folds a pair of these
// into one v_cvt_pk_bf16_f32 on gfx950 when written as a plain __bf16 cast.
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}

__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    f32x2_t v = {lo, hi};
    bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(unsigned int, b);
}

// 8 consecutive elements -> 8 floats, from either dtype (16-B aligned for bf16,
// 32-B span for f32 read as two 16-B loads).
template <int DT>
__device__ __forceinline__ void load8(const void* base, int64_t i, float (&v)[8]) {
    if constexpr (DT == DTF_BF16) {
        u32x4 r = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(base) + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = __uint_as_float(r[k] << 16);
            v[2 * k + 1] = __uint_as_float(r[k] & 0xffff0000u);
        }
    } else {
        const f32x4* p = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + i);
        f32x4 a = p[0], b = p[1];
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = a[k]; v[4 + k] = b[k]; }
    }
}

template <int DT>
__device__ __forceinline__ void store8(void* base, int64_t i, const float (&v)[8]) {
    if constexpr (DT == DTF_BF16) {
        u32x4 r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = pack_bf16x2(v[2 * k], v[2 * k + 1]);
        *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(base) + i) = r;
    } else {
        f32x4* p = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + i);
        f32x4 a, b;
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k] = v[k]; b[k] = v[4 + k]; }
        p[0] = a; p[1] = b;
    }
}

template <int DT>
__device__ __forceinline__ float load1(const void* base, int64_t i) {
    if constexpr (DT == DTF_BF16) return bf16_to_f32(reinterpret_cast<const unsigned short*>(base)[i]);
    else return reinterpret_cast<const float*>(base)[i];
}

template <int DT>
__device__ __forceinline__ void store1(void* base, int64_t i, float v) {
    if constexpr (DT == DTF_BF16) reinterpret_cast<unsigned short*>(base)[i] = f32_to_bf16(v);
    else reinterpret_cast<float*>(base)[i] = v;
}

// ReLU that PROPAGATES NaN like the framework's (fmaxf(NaN, 0) is 0: a non-finite activation would silently become a
// dead unit -- that is how a library GEMM solution that returned NaN went unnoticed: the loss stayed finite)
__device__ __forceinline__ float relu_nan(float x) { return x < 0.f ? 0.f : x; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// grid sizing for streaming kernels: enough blocks to fill 256 CUs x 8, then
// grid-stride (cdna_hip_programming.md Guideline 11)
static inline int dtf_stream_grid(int64_t work_items, int block) {
    int64_t b = (work_items + block - 1) / block;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

// Elementwise passes without a per-block result (K1, K2, casts): one block per `block` work items, no cap -- every
// lane runs its loop body once.  Measured on the shard-apply kernels against the 2048-block grid-strided shape
// (scripts/ps_apply_bench.py, profiles/r2_straggler_kernels.txt): Adam 4.86 -> 5.46 TB/s on 110 M elements,
// 4.59 -> 5.20 on 25.6 M; SGD-momentum 5.00 -> 5.38.  DTF_APPLY_GRID_CAP restores a cap for A/B runs.
static inline int dtf_flat_grid(int64_t work_items, int block) {
    static int64_t cap = 0;
    if (cap == 0) {
        const char* e = getenv("DTF_APPLY_GRID_CAP");
        cap = e ? atoll(e) : (1ll << 30);
        if (cap < 1) cap = 1ll << 30;
    }
    int64_t b = (work_items + block - 1) / block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

// hipGetLastError() reports the last error of ANY runtime call of this thread
// (e.g. a benign probe inside MIOpen), so entry points clear it first.
#define DTF_CLEAR_ERRORS() (void)hipGetLastError()

#define DTF_CHECK_LAUNCH()                                     \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) return (int)e__;                \
    } while (0)
