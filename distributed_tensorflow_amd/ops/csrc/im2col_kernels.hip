// K21: patch matrix of a FEW-CHANNEL convolution input (an image: 3 channels; a grey-scale digit: 1), for the weight
// gradient of image-input layers.
//
// dW[o][r][s][c] = sum over pixels of dY[pix][o] * X[n, oh*stride - pad + r, ow*stride - pad + s, c] is a
// (pixels x Cout)^T (pixels x T) product with T = KH*KW*Cin <= 32 columns (27 for a 3x3 layer on RGB).  None of the
// im2col-free pipelines fits it (K6T-conv gathers 8-channel chunks; a pixel here is 6 bytes), MIOpen's kernel for these
// layers accumulates with atomics (Inception-v3's 3 -> 32 stem layer was the one tensor of the model whose gradient
// differed run to run), and padding the image to 8 channels makes K6T-conv chew a 72-wide K in a 256-wide tile (1 ms).
// With T <= 32 the patch matrix is small enough to simply write down: P[pixel][32] bf16 (64 B per pixel, as much as
// dY of a 32-channel layer), columns t = (r*KW + s)*Cin + c, zero for t >= T and for taps outside the image; K6T
// (fixed-order slabs, bitwise reproducible) then computes dY^T P, whose first T columns are the gradient in the
// channels-last parameter layout [Cout][KH][KW][Cin].
//
// Mapping: one lane per 16-B chunk of P (8 taps); 4 consecutive lanes = one pixel, so a wave writes 1 KB contiguous and
// its 2-byte gathers fall into a few cache lines of 1-3 image rows.  Rows >= M (the caller pads M to a multiple of 32
// for K6T) are written as zeros.
#include "dtf_common.h"

__global__ __launch_bounds__(256) void im2col_small_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ P,
                                                           int64_t M, int64_t Mpad, int H, int W, int Cin, int OH, int OW,
                                                           int KW, int T, int stride, int pad_h, int pad_w) {
    const int64_t nchunk = Mpad * 4;
    const int kwc = KW * Cin;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < nchunk; id += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = id >> 2;
        const int t0 = (int)(id & 3) * 8;
        unsigned short v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0;
        if (pix < M && t0 < T) {
            const int ow = (int)(pix % OW);
            const int64_t q = pix / OW;
            const int oh = (int)(q % OH);
            const int64_t n = q / OH;
            const int ih0 = oh * stride - pad_h, iw0 = ow * stride - pad_w;
            const unsigned short* img = x + n * (int64_t)H * W * Cin;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = t0 + j;
                if (t < T) {
                    const int r = t / kwc, rem = t - r * kwc;
                    const int s = rem / Cin, c = rem - s * Cin;
                    const int ih = ih0 + r, iw = iw0 + s;
                    if (ih >= 0 && ih < H && iw >= 0 && iw < W)
                        v[j] = img[DTF_IDX(((int64_t)ih * W + iw) * Cin + c, (int64_t)H * W * Cin, "im2col_small tap")];
                }
            }
        }
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (unsigned)v[2 * k] | ((unsigned)v[2 * k + 1] << 16);
        *reinterpret_cast<u32x4*>(P + id * 8) = o;
    }
}

extern "C" {

// 0 when the geometry is one the kernel handles
int dtf_im2col_small_ok(int64_t NB, int H, int W, int Cin, int KH, int KW, int stride, int pad_h, int pad_w) {
    if (NB <= 0 || H <= 0 || W <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad_h < 0 || pad_w < 0) return -70;
    if (Cin * KH * KW > 32) return -71;
    if (H + 2 * pad_h < KH || W + 2 * pad_w < KW) return -72;
    if (NB * (int64_t)H * W * Cin >= (1ll << 40)) return -73;
    return 0;
}

// x: [NB, H, W, Cin] bf16 (NHWC-dense); P: [Mpad, 32] bf16 with Mpad >= M = NB*OH*OW (rows >= M zero)
int dtf_im2col_small(const void* x, void* P, int64_t NB, int H, int W, int Cin, int KH, int KW, int stride, int pad_h,
                     int pad_w, int64_t Mpad, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    const int rc = dtf_im2col_small_ok(NB, H, W, Cin, KH, KW, stride, pad_h, pad_w);
    if (rc) return rc;
    const int OH = (H + 2 * pad_h - KH) / stride + 1, OW = (W + 2 * pad_w - KW) / stride + 1;
    const int64_t M = NB * (int64_t)OH * OW;
    if (Mpad < M) return -74;
    if (((uintptr_t)P) & 15) return -16;
    int64_t blocks = (Mpad * 4 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(im2col_small_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const unsigned short*)x,
                       (unsigned short*)P, M, Mpad, H, W, Cin, OH, OW, KW, Cin * KH * KW, stride, pad_h, pad_w);
    DTF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
