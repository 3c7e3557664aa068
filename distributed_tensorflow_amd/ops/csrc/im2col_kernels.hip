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
// Mapping: one lane per pixel = one 64-byte row of P, transposed through LDS for the stores.  The tap -> (filter row, column,
// address delta) table is built on the host and passed by value, so after unrolling every entry is a scalar register;
// the pixel decomposition is 32-bit (the caller keeps M below 2^31).  The first version -- one lane per 16-B chunk with
// 64-bit div / mod per lane and per tap -- took 320 us for Inception-v3's stem layer at batch 256 (364 MB of P: ~100 us
// of HBM time).  Rows >= M (the caller pads M to a multiple of 32 for K6T) are written as zeros.
#include "dtf_common.h"

struct Im2colTab {
    int delta[32];              // element offset of tap t from the window origin: (r * W + s) * Cin + c
    signed char r[32], s[32];   // filter row / column of tap t (taps >= T: unused)
};

__global__ __launch_bounds__(256) void im2col_small_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ P,
                                                           int M, int Mpad, int H, int W, int Cin, int OH, int OW, int KH,
                                                           int KW, int T, int stride, int pad_h, int pad_w, Im2colTab tab) {
    // a lane's 64-byte row goes through LDS (80-byte pitch: 16 lanes x 4 dwords cover the 64 banks once) so that the
    // global stores are 16 B per lane at consecutive addresses -- four partial 16-B writes into 64 different lines per
    // store instruction ran at half the rate (222 us vs the chunk-per-lane version's coalesced stores)
    __shared__ __attribute__((aligned(16))) unsigned char stage[256 * 80];
    const int tid = threadIdx.x;
    for (int base = blockIdx.x * 256; base < Mpad; base += gridDim.x * 256) {
        const int pix = base + tid;
        unsigned v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = 0u;
        if (pix < M) {
            const int ow = pix % OW, q = pix / OW;
            const int oh = q % OH, n = q / OH;
            const int ih0 = oh * stride - pad_h, iw0 = ow * stride - pad_w;
            const int64_t img = (int64_t)n * H * W * Cin;
            const int64_t org = ((int64_t)ih0 * W + iw0) * Cin;          // window origin inside the image (may be negative)
            const bool interior = ih0 >= 0 && iw0 >= 0 && ih0 + KH <= H && iw0 + KW <= W;
#pragma unroll
            for (int t = 0; t < 32; ++t) {
                if (t < T) {
                    const bool ok = interior || ((unsigned)(ih0 + tab.r[t]) < (unsigned)H && (unsigned)(iw0 + tab.s[t]) < (unsigned)W);
                    unsigned e = 0u;
                    if (ok) e = x[img + DTF_IDX(org + tab.delta[t], (int64_t)H * W * Cin, "im2col_small tap")];
                    v[t >> 1] |= (t & 1) ? (e << 16) : e;
                }
            }
        }
        u32x4* mine = reinterpret_cast<u32x4*>(stage + tid * 80);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            u32x4 o = {v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
            mine[k] = o;
        }
        __syncthreads();
        const int npix = min(256, Mpad - base);                      // Mpad is a multiple of 32
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = k * 256 + tid;                              // chunk j of this block's 256 rows
            if ((j >> 2) < npix)
                *reinterpret_cast<u32x4*>(P + (int64_t)base * 32 + (int64_t)j * 8) =
                    *reinterpret_cast<const u32x4*>(stage + (j >> 2) * 80 + (j & 3) * 16);
        }
        __syncthreads();
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
    if (Mpad >= (1ll << 31) - (1ll << 25)) return -75;          // 32-bit pixel index plus one grid stride
    Im2colTab tab{};
    const int T = Cin * KH * KW;
    for (int t = 0; t < T; ++t) {
        const int r = t / (KW * Cin), rem = t - r * (KW * Cin);
        const int sc = rem / Cin, c = rem - sc * Cin;
        tab.delta[t] = (r * W + sc) * Cin + c;
        tab.r[t] = (signed char)r;
        tab.s[t] = (signed char)sc;
    }
    int64_t blocks = (Mpad + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(im2col_small_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const unsigned short*)x,
                       (unsigned short*)P, (int)M, (int)Mpad, H, W, Cin, OH, OW, KH, KW, T, stride, pad_h, pad_w, tab);
    DTF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
