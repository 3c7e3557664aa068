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

}  // extern "C"
