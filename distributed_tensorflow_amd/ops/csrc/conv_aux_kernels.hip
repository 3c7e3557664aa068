// Filter transform for "data gradient as a forward convolution" (ops/conv.py): the NHWC filter
// W[Cout][KH*KW][Cin] becomes W'[Cin][KH*KW][Cout] with the taps in reverse order (a 180-degree rotation),
// i.e. one [Cout x Cin] -> [Cin x Cout] transpose per tap.  The framework's flip + permuted copy are two
// launches of ~5 and ~9 us on these <= 5 MB tensors, 49 times per Inception-v3 step; this is one.
#include "dtf_common.h"

#define FT_TILE 32      // small tiles: these tensors are tiny, the kernel is latency-bound, parallelism is what helps

__global__ __launch_bounds__(256) void filter_flip_transpose_kernel(const unsigned short* __restrict__ w,
                                                                    unsigned short* __restrict__ out, int Cout, int Cin,
                                                                    int T) {
    __shared__ unsigned short tile[FT_TILE][FT_TILE + 2];
    const int t = blockIdx.z;
    const int ci0 = blockIdx.x * FT_TILE, co0 = blockIdx.y * FT_TILE;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll 4
    for (int r = ty; r < FT_TILE; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        if (co < Cout && ci < Cin) tile[r][tx] = w[((int64_t)co * T + t) * Cin + ci];
    }
    __syncthreads();
    const int to = T - 1 - t;
#pragma unroll 4
    for (int r = ty; r < FT_TILE; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < Cin && co < Cout) out[((int64_t)ci * T + to) * Cout + co] = tile[tx][r];
    }
}

// Spatial subsampling of an NHWC tensor for strided 1x1 convolutions (ops/conv.py subsample_fork): out[n, oh, ow, :] =
// x[n, s oh, s ow, :], and its adjoint dst[n, s oh, s ow, :] += g[n, oh, ow, :] in place.  One lane per 16 B; a pixel's C
// channels are contiguous on both sides, so every wave instruction moves whole 128-byte lines (C >= 64).  The
// framework's generic strided copy / add took 76 / 60 us on ResNet-50's [256, 256, 56 -> 28, 56 -> 28] (103 MB moved).
template <bool ADD>
__global__ __launch_bounds__(256) void subsample_kernel(unsigned short* __restrict__ big, unsigned short* __restrict__ small_,
                                                        int H, int W, int OH, int OW, int cv, int s, int64_t total) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;      // 16-byte vector of the small tensor
    if (v >= total) return;
    const int c = (int)(v % cv);
    const int64_t pix = v / cv;
    const int ow = (int)(pix % OW);
    const int64_t t = pix / OW;
    const int oh = (int)(t % OH);
    const int64_t n = t / OH;
    const int64_t src = (((n * H + (int64_t)oh * s) * W + (int64_t)ow * s) * cv + c);
    u32x4* bp = reinterpret_cast<u32x4*>(big) + DTF_IDX(src, (int64_t)(total / ((int64_t)OH * OW)) * H * W, "subsample big index");
    u32x4* sp = reinterpret_cast<u32x4*>(small_) + v;
    if (!ADD) {
        *sp = __builtin_nontemporal_load(bp);
    } else {
        const u32x4 a = *bp, g = __builtin_nontemporal_load(sp);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float lo = __uint_as_float(a[k] << 16) + __uint_as_float(g[k] << 16);
            const float hi = __uint_as_float(a[k] & 0xffff0000u) + __uint_as_float(g[k] & 0xffff0000u);
            o[k] = pack_bf16x2(lo, hi);
        }
        *bp = o;
    }
}

extern "C" {

// w: bf16 [Cout][T][Cin] -> out: bf16 [Cin][T][Cout], taps reversed
int dtf_filter_flip_transpose(const void* w, void* out, int Cout, int Cin, int T, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (Cout <= 0 || Cin <= 0 || T <= 0 || T > 65535) return -70;
    hipLaunchKernelGGL(filter_flip_transpose_kernel, dim3((Cin + FT_TILE - 1) / FT_TILE, (Cout + FT_TILE - 1) / FT_TILE, T),
                       dim3(256), 0, stream, (const unsigned short*)w, (unsigned short*)out, Cout, Cin, T);
    DTF_CHECK_LAUNCH();
    return 0;
}

// add == 0: small[N, OH, OW, C] = big[N, ::s, ::s, C];  add == 1: big[N, ::s, ::s, C] += small (bf16, C % 8 == 0)
int dtf_subsample_nhwc(void* big, void* small_, int N, int H, int W, int C, int s, int add, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || s < 1 || (C & 7)) return -71;
    if ((((uintptr_t)big) | ((uintptr_t)small_)) & 15) return -16;
    const int OH = (H + s - 1) / s, OW = (W + s - 1) / s, cv = C / 8;
    const int64_t total = (int64_t)N * OH * OW * cv;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (add)
        hipLaunchKernelGGL((subsample_kernel<true>), dim3(grid), dim3(256), 0, stream, (unsigned short*)big, (unsigned short*)small_,
                           H, W, OH, OW, cv, s, total);
    else
        hipLaunchKernelGGL((subsample_kernel<false>), dim3(grid), dim3(256), 0, stream, (unsigned short*)big, (unsigned short*)small_,
                           H, W, OH, OW, cv, s, total);
    DTF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
