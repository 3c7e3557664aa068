// K9: max pooling for NHWC bf16 activations on gfx950 (the ResNet stem pool and Inception's grid
// reductions).  Both directions are pure streaming kernels, 16 B (8 channels) per lane per access:
//
//   forward : read x once (window overlap served by L2), write y and a 1-byte window position per
//             element (which of the k*k taps won; first maximum in scan order, NaN propagates --
//             the same rule as the framework's reference pooling)
//   backward: a *gather* over the <= ceil(k/s)^2 windows that cover an input pixel: reads dy and
//             the tap bytes, writes every dx element exactly once.  No atomics, no zero-fill pass,
//             and x is not read at all (the scatter formulation needs fill + atomics or a re-read
//             of x: 656 us vs ~150 us for the 112x112x64 stem pool at batch 256).
#include "dtf_common.h"

// AFFINE: the input is a convolution output whose training-mode BatchNorm + ReLU has not been applied yet: every tap
// is mapped through relu(x * scale[c] + shift[c]) (rounded to bf16 like the K7 apply pass would store it) on the
// way in, so the normalised full-resolution tensor -- 2.8x the pool output at 3x3/2 -- is never written or read.
template <int K, bool AFFINE>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const unsigned short* __restrict__ x,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          unsigned short* __restrict__ y,
                                                          unsigned char* __restrict__ tap, int N, int H, int W,
                                                          int C, int OH, int OW, int stride, int pad) {
    const int C8 = C >> 3;
    const int64_t total = (int64_t)N * OH * OW * C8;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(t % C8);
        int64_t pix = t / C8;
        const int ow = (int)(pix % OW);
        pix /= OW;
        const int oh = (int)(pix % OH);
        const int n = (int)(pix / OH);
        const int h0 = oh * stride - pad, w0 = ow * stride - pad;
        float best[8];
        unsigned bi[8];
        const int kh0 = h0 < 0 ? -h0 : 0, kw0 = w0 < 0 ? -w0 : 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; bi[k] = kh0 * K + kw0; }
        const unsigned short* xn = x + (int64_t)n * H * W * C + c8 * 8;
        float sc[8], sh[8];
        if (AFFINE) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { sc[k] = scale[c8 * 8 + k]; sh[k] = shift[c8 * 8 + k]; }
        }
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            const int h = h0 + kh;
            if (h < 0 || h >= H) continue;
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const int w = w0 + kw;
                if (w < 0 || w >= W) continue;
                float v[8];
                load8<DTF_BF16>(xn, DTF_IDX(((int64_t)h * W + w) * C, (int64_t)H * W * C, "maxpool_fwd window"), v);
                if (AFFINE) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = bf16_to_f32(f32_to_bf16(relu_nan(fmaf(v[k], sc[k], sh[k]))));
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const bool take = (v[k] > best[k]) || (v[k] != v[k]);
                    best[k] = take ? v[k] : best[k];
                    bi[k] = take ? (unsigned)(kh * K + kw) : bi[k];
                }
            }
        }
        const int64_t o = (((int64_t)n * OH + oh) * OW + ow) * C + c8 * 8;
        store8<DTF_BF16>(y, o, best);
        u32x2 pk;
        pk[0] = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
        pk[1] = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
        *reinterpret_cast<u32x2*>(tap + o) = pk;
    }
}

template <int K>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const unsigned short* __restrict__ dy,
                                                          const unsigned char* __restrict__ tap,
                                                          unsigned short* __restrict__ dx, int N, int H, int W,
                                                          int C, int OH, int OW, int stride, int pad) {
    const int C8 = C >> 3;
    const int64_t total = (int64_t)N * H * W * C8;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(t % C8);
        int64_t pix = t / C8;
        const int w = (int)(pix % W);
        pix /= W;
        const int h = (int)(pix % H);
        const int n = (int)(pix / H);
        // windows oh with oh*stride - pad <= h <= oh*stride - pad + K-1
        int oh_lo = h + pad - (K - 1);
        oh_lo = oh_lo <= 0 ? 0 : (oh_lo + stride - 1) / stride;
        int oh_hi = (h + pad) / stride;
        if (oh_hi > OH - 1) oh_hi = OH - 1;
        int ow_lo = w + pad - (K - 1);
        ow_lo = ow_lo <= 0 ? 0 : (ow_lo + stride - 1) / stride;
        int ow_hi = (w + pad) / stride;
        if (ow_hi > OW - 1) ow_hi = OW - 1;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int oh = oh_lo; oh <= oh_hi; ++oh) {
            const unsigned kh = (unsigned)(h + pad - oh * stride);
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const unsigned me = kh * K + (unsigned)(w + pad - ow * stride);
                const int64_t o = DTF_IDX((((int64_t)n * OH + oh) * OW + ow) * C + c8 * 8, (int64_t)N * OH * OW * C,
                                          "maxpool_bwd window");
                const u32x2 pk = *reinterpret_cast<const u32x2*>(tap + o);
                float g[8];
                load8<DTF_BF16>(dy, o, g);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned b = (pk[k >> 2] >> ((k & 3) * 8)) & 0xffu;
                    acc[k] += (b == me) ? g[k] : 0.f;
                }
            }
        }
        store8<DTF_BF16>(dx, ((((int64_t)n * H + h) * W + w) * C) + c8 * 8, acc);
    }
}

// 3x3 / stride 2 (the ResNet stem pool, Inception's grid reductions): one lane owns a 2x2 patch of input pixels at
// even coordinates.  The four pixels are covered by the same <= 2x2 pooling windows, so a patch costs 4 window loads
// (dy + tap bytes) instead of the 1 + 2 + 2 + 4 = 9 the per-pixel gather issues, and writes 4 x 16 B.  Windows are
// visited in the same (oh, ow) order per pixel as maxpool_bwd_kernel: the sums are bitwise identical.
__global__ __launch_bounds__(256) void maxpool_bwd_k3s2_kernel(const unsigned short* __restrict__ dy,
                                                               const unsigned char* __restrict__ tap,
                                                               unsigned short* __restrict__ dx, int N, int H, int W,
                                                               int C, int OH, int OW, int pad) {
    const int C8 = C >> 3, PH = (H + 1) >> 1, PW = (W + 1) >> 1;
    const int64_t total = (int64_t)N * PH * PW * C8;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(t % C8);
        int64_t pix = t / C8;
        const int w0 = 2 * (int)(pix % PW);
        pix /= PW;
        const int h0 = 2 * (int)(pix % PH);
        const int n = (int)(pix / PH);
        int oh_lo = h0 + pad - 2;
        oh_lo = oh_lo <= 0 ? 0 : (oh_lo + 1) >> 1;
        int oh_hi = (h0 + 1 + pad) >> 1;
        if (oh_hi > OH - 1) oh_hi = OH - 1;
        int ow_lo = w0 + pad - 2;
        ow_lo = ow_lo <= 0 ? 0 : (ow_lo + 1) >> 1;
        int ow_hi = (w0 + 1 + pad) >> 1;
        if (ow_hi > OW - 1) ow_hi = OW - 1;
        float acc[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[q][k] = 0.f;
        }
        for (int oh = oh_lo; oh <= oh_hi; ++oh) {
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const int64_t o = DTF_IDX((((int64_t)n * OH + oh) * OW + ow) * C + c8 * 8, (int64_t)N * OH * OW * C,
                                          "maxpool_bwd_k3s2 window");
                const u32x2 pk = *reinterpret_cast<const u32x2*>(tap + o);
                float g[8];
                load8<DTF_BF16>(dy, o, g);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned kh = (unsigned)(h0 + (q >> 1) + pad - 2 * oh), kw = (unsigned)(w0 + (q & 1) + pad - 2 * ow);
                    if (kh < 3u && kw < 3u) {
                        const unsigned me = kh * 3 + kw;
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const unsigned b = (pk[k >> 2] >> ((k & 3) * 8)) & 0xffu;
                            acc[q][k] += (b == me) ? g[k] : 0.f;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int h = h0 + (q >> 1), w = w0 + (q & 1);
            if (h < H && w < W) store8<DTF_BF16>(dx, ((((int64_t)n * H + h) * W + w) * C) + c8 * 8, acc[q]);
        }
    }
}

// The same gather for max_pool(relu(BN(z))) (ops/stem.py) with the BatchNorm backward REDUCTION folded in: the lane that
// has just produced the gradient of four pixels also loads their z, applies the ReLU mask (recomputed from z, scale,
// shift like the K7 backward) and accumulates s1 = sum g, s2 = sum g (z - mean) per channel -- the separate K7 reduction
// pass would read the 411 MB gradient and z once more (167 us at batch 256; the extra z read here costs ~70).  g is
// the bf16-ROUNDED gradient, i.e. exactly what the stored tensor holds.  A lane's channel group is fixed (C / 8 divides
// the block and the grid stride), partial rows [block][2][C] in the K7 finalize layout, fixed summation order.
__global__ __launch_bounds__(256) void maxpool_bwd_k3s2_bnred_kernel(const unsigned short* __restrict__ dy,
                                                                     const unsigned char* __restrict__ tap,
                                                                     const unsigned short* __restrict__ z,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ scale,
                                                                     const float* __restrict__ shift,
                                                                     unsigned short* __restrict__ dx, float* __restrict__ partial,
                                                                     int N, int H, int W, int C, int OH, int OW, int pad) {
    const int C8 = C >> 3, PH = (H + 1) >> 1, PW = (W + 1) >> 1;
    const int64_t total = (int64_t)N * PH * PW * C8;
    const int c8 = threadIdx.x % C8;
    float mu[8], sc[8], sh[8], s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mu[k] = mean[c8 * 8 + k]; sc[k] = scale[c8 * 8 + k]; sh[k] = shift[c8 * 8 + k];
        s1[k] = 0.f; s2[k] = 0.f;
    }
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t pix = t / C8;
        const int w0 = 2 * (int)(pix % PW);
        pix /= PW;
        const int h0 = 2 * (int)(pix % PH);
        const int n = (int)(pix / PH);
        int oh_lo = h0 + pad - 2;
        oh_lo = oh_lo <= 0 ? 0 : (oh_lo + 1) >> 1;
        int oh_hi = (h0 + 1 + pad) >> 1;
        if (oh_hi > OH - 1) oh_hi = OH - 1;
        int ow_lo = w0 + pad - 2;
        ow_lo = ow_lo <= 0 ? 0 : (ow_lo + 1) >> 1;
        int ow_hi = (w0 + 1 + pad) >> 1;
        if (ow_hi > OW - 1) ow_hi = OW - 1;
        // z of the four pixels first: independent of the window loads below
        u32x4 zr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int h = h0 + (q >> 1), w = w0 + (q & 1);
            zr[q] = (u32x4){0u, 0u, 0u, 0u};
            if (h < H && w < W) zr[q] = *reinterpret_cast<const u32x4*>(z + ((((int64_t)n * H + h) * W + w) * C) + c8 * 8);
        }
        float acc[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[q][k] = 0.f;
        }
        for (int oh = oh_lo; oh <= oh_hi; ++oh) {
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const int64_t o = DTF_IDX((((int64_t)n * OH + oh) * OW + ow) * C + c8 * 8, (int64_t)N * OH * OW * C,
                                          "maxpool_bwd_k3s2_bnred window");
                const u32x2 pk = *reinterpret_cast<const u32x2*>(tap + o);
                float g[8];
                load8<DTF_BF16>(dy, o, g);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned kh = (unsigned)(h0 + (q >> 1) + pad - 2 * oh), kw = (unsigned)(w0 + (q & 1) + pad - 2 * ow);
                    if (kh < 3u && kw < 3u) {
                        const unsigned me = kh * 3 + kw;
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const unsigned b = (pk[k >> 2] >> ((k & 3) * 8)) & 0xffu;
                            acc[q][k] += (b == me) ? g[k] : 0.f;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int h = h0 + (q >> 1), w = w0 + (q & 1);
            if (h < H && w < W) {
                u32x4 r;
#pragma unroll
                for (int k = 0; k < 4; ++k) r[k] = pack_bf16x2(acc[q][2 * k], acc[q][2 * k + 1]);
                *reinterpret_cast<u32x4*>(dx + ((((int64_t)n * H + h) * W + w) * C) + c8 * 8) = r;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float z0 = __uint_as_float(zr[q][k] << 16), z1 = __uint_as_float(zr[q][k] & 0xffff0000u);
                    float g0 = __uint_as_float(r[k] << 16), g1 = __uint_as_float(r[k] & 0xffff0000u);
                    g0 = fmaf(z0, sc[2 * k], sh[2 * k]) > 0.f ? g0 : 0.f;
                    g1 = fmaf(z1, sc[2 * k + 1], sh[2 * k + 1]) > 0.f ? g1 : 0.f;
                    s1[2 * k] += g0; s2[2 * k] = fmaf(g0, z0 - mu[2 * k], s2[2 * k]);
                    s1[2 * k + 1] += g1; s2[2 * k + 1] = fmaf(g1, z1 - mu[2 * k + 1], s2[2 * k + 1]);
                }
            }
        }
    }
    __shared__ float red[2][8 * 257];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        red[0][k * 257 + threadIdx.x] = s1[k];
        red[1][k * 257 + threadIdx.x] = s2[k];
    }
    __syncthreads();
    const int lanes = 256 / C8;                    // lanes that share a channel group: tid = c8 + C8 * j
    for (int idx = threadIdx.x; idx < 2 * C; idx += 256) {
        const int stat = idx >= C, c = idx - stat * C;
        const int k = c & 7, g8 = c >> 3;
        float sum = 0.f;
        for (int j = 0; j < lanes; ++j) sum += red[stat][k * 257 + g8 + C8 * j];
        partial[((int64_t)blockIdx.x * 2 + stat) * C + c] = sum;
    }
}

// ---------------------------------------------------------------------------
// Average pooling (Inception's 3x3/1 "pool + 1x1" branches, the aux head's 5x5/3).  Same two
// gather shapes as above; the divisor of a window is recomputed from its coordinates.
// include_pad: divide by the window clipped to the *padded* image (the framework default);
// otherwise by the number of real pixels (TF 'SAME' average pooling).
// ---------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ float avg_divisor(int oh, int ow, int H, int W, int stride, int pad, int include_pad) {
    int h0 = oh * stride - pad, w0 = ow * stride - pad;
    int h1 = h0 + K, w1 = w0 + K;
    if (h1 > H + pad) h1 = H + pad;
    if (w1 > W + pad) w1 = W + pad;
    if (include_pad) return (float)((h1 - h0) * (w1 - w0));
    if (h0 < 0) h0 = 0;
    if (w0 < 0) w0 = 0;
    if (h1 > H) h1 = H;
    if (w1 > W) w1 = W;
    return (float)((h1 - h0) * (w1 - w0));
}

template <int K>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const unsigned short* __restrict__ x,
                                                          unsigned short* __restrict__ y, int N, int H, int W, int C,
                                                          int OH, int OW, int stride, int pad, int include_pad, int64_t ld) {
    // ld: pixel stride (elements) of x -- C, or the parent's channel count for a channel slice read in place
    const int C8 = C >> 3;
    const int64_t total = (int64_t)N * OH * OW * C8;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(t % C8);
        int64_t pix = t / C8;
        const int ow = (int)(pix % OW);
        pix /= OW;
        const int oh = (int)(pix % OH);
        const int n = (int)(pix / OH);
        const int h0 = oh * stride - pad, w0 = ow * stride - pad;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        const unsigned short* xn = x + (int64_t)n * H * W * ld + c8 * 8;
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            const int h = h0 + kh;
            if (h < 0 || h >= H) continue;
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const int w = w0 + kw;
                if (w < 0 || w >= W) continue;
                float v[8];
                load8<DTF_BF16>(xn, ((int64_t)h * W + w) * ld, v);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += v[k];
            }
        }
        const float inv = 1.f / avg_divisor<K>(oh, ow, H, W, stride, pad, include_pad);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] *= inv;
        store8<DTF_BF16>(y, (((int64_t)n * OH + oh) * OW + ow) * C + c8 * 8, acc);
    }
}

template <int K>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const unsigned short* __restrict__ dy,
                                                          unsigned short* __restrict__ dx, int N, int H, int W, int C,
                                                          int OH, int OW, int stride, int pad, int include_pad, int64_t ld) {
    // ld: pixel stride of dx (written into a channel slice of a wider gradient tensor when > C)
    const int C8 = C >> 3;
    const int64_t total = (int64_t)N * H * W * C8;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(t % C8);
        int64_t pix = t / C8;
        const int w = (int)(pix % W);
        pix /= W;
        const int h = (int)(pix % H);
        const int n = (int)(pix / H);
        int oh_lo = h + pad - (K - 1);
        oh_lo = oh_lo <= 0 ? 0 : (oh_lo + stride - 1) / stride;
        int oh_hi = (h + pad) / stride;
        if (oh_hi > OH - 1) oh_hi = OH - 1;
        int ow_lo = w + pad - (K - 1);
        ow_lo = ow_lo <= 0 ? 0 : (ow_lo + stride - 1) / stride;
        int ow_hi = (w + pad) / stride;
        if (ow_hi > OW - 1) ow_hi = OW - 1;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int oh = oh_lo; oh <= oh_hi; ++oh)
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const float inv = 1.f / avg_divisor<K>(oh, ow, H, W, stride, pad, include_pad);
                float g[8];
                load8<DTF_BF16>(dy, (((int64_t)n * OH + oh) * OW + ow) * C + c8 * 8, g);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = fmaf(g[k], inv, acc[k]);
            }
        store8<DTF_BF16>(dx, ((((int64_t)n * H + h) * W + w) * ld) + c8 * 8, acc);
    }
}

extern "C" {

int dtf_maxpool_fwd(const void* x, void* y, void* tap, int N, int H, int W, int C, int OH, int OW, int k,
                    int stride, int pad, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (C % 8 != 0 || N <= 0 || OH <= 0 || OW <= 0 || stride < 1 || pad < 0 || pad >= k) return -20;
    if (OH != (H + 2 * pad - k) / stride + 1 || OW != (W + 2 * pad - k) / stride + 1) return -21;
    const int64_t items = (int64_t)N * OH * OW * (C / 8);
    int64_t blocks = (items + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    const unsigned short* xs = (const unsigned short*)x;
    unsigned short* ys = (unsigned short*)y;
    unsigned char* ts = (unsigned char*)tap;
    if (k == 3)
        hipLaunchKernelGGL((maxpool_fwd_kernel<3, false>), dim3((unsigned)blocks), dim3(256), 0, stream, xs, nullptr, nullptr, ys, ts, N, H, W, C, OH, OW, stride, pad);
    else if (k == 2)
        hipLaunchKernelGGL((maxpool_fwd_kernel<2, false>), dim3((unsigned)blocks), dim3(256), 0, stream, xs, nullptr, nullptr, ys, ts, N, H, W, C, OH, OW, stride, pad);
    else
        return -23;
    DTF_CHECK_LAUNCH();
    return 0;
}

// max_pool( relu(x * scale[c] + shift[c]) ): x is the raw convolution output, scale / shift the fp32 [C] vectors of
// its BatchNorm (see maxpool_fwd_kernel AFFINE)
int dtf_maxpool_fwd_affine(const void* x, const float* scale, const float* shift, void* y, void* tap, int N, int H, int W,
                           int C, int OH, int OW, int k, int stride, int pad, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (C % 8 != 0 || N <= 0 || OH <= 0 || OW <= 0 || stride < 1 || pad < 0 || pad >= k || !scale || !shift) return -20;
    if (OH != (H + 2 * pad - k) / stride + 1 || OW != (W + 2 * pad - k) / stride + 1) return -21;
    const int64_t items = (int64_t)N * OH * OW * (C / 8);
    int64_t blocks = (items + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    const unsigned short* xs = (const unsigned short*)x;
    unsigned short* ys = (unsigned short*)y;
    unsigned char* ts = (unsigned char*)tap;
    if (k == 3)
        hipLaunchKernelGGL((maxpool_fwd_kernel<3, true>), dim3((unsigned)blocks), dim3(256), 0, stream, xs, scale, shift, ys, ts, N, H, W, C, OH, OW, stride, pad);
    else if (k == 2)
        hipLaunchKernelGGL((maxpool_fwd_kernel<2, true>), dim3((unsigned)blocks), dim3(256), 0, stream, xs, scale, shift, ys, ts, N, H, W, C, OH, OW, stride, pad);
    else
        return -23;
    DTF_CHECK_LAUNCH();
    return 0;
}

int dtf_maxpool_bwd(const void* dy, const void* tap, void* dx, int N, int H, int W, int C, int OH, int OW, int k,
                    int stride, int pad, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (C % 8 != 0 || N <= 0 || OH <= 0 || OW <= 0 || stride < 1 || pad < 0 || pad >= k) return -20;
    if (OH != (H + 2 * pad - k) / stride + 1 || OW != (W + 2 * pad - k) / stride + 1) return -21;
    const int64_t items = (int64_t)N * H * W * (C / 8);
    int64_t blocks = (items + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    const unsigned short* dys = (const unsigned short*)dy;
    const unsigned char* ts = (const unsigned char*)tap;
    unsigned short* dxs = (unsigned short*)dx;
    if (k == 3 && stride == 2) {
        const int64_t patches = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
        int64_t pb = (patches + 255) / 256;
        if (pb > 65536) pb = 65536;
        hipLaunchKernelGGL(maxpool_bwd_k3s2_kernel, dim3((unsigned)pb), dim3(256), 0, stream, dys, ts, dxs, N, H, W, C, OH, OW, pad);
    } else if (k == 3)
        hipLaunchKernelGGL((maxpool_bwd_kernel<3>), dim3((unsigned)blocks), dim3(256), 0, stream, dys, ts, dxs, N, H, W, C, OH, OW, stride, pad);
    else if (k == 2)
        hipLaunchKernelGGL((maxpool_bwd_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, stream, dys, ts, dxs, N, H, W, C, OH, OW, stride, pad);
    else
        return -23;
    DTF_CHECK_LAUNCH();
    return 0;
}

// dtf_maxpool_bwd for the 3x3 / 2 pool of relu(BN(z)) with the BatchNorm backward reduction folded in; partial: fp32
// [dtf_maxpool_bwd_bnred_blocks][2][C] (the K7 backward finalize's input); C a divisor of 2048 with C / 8 dividing 256
int64_t dtf_maxpool_bwd_bnred_blocks(int N, int H, int W, int C) {
    const int64_t patches = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
    int64_t pb = (patches + 255) / 256;
    if (pb > 2048) pb = 2048;
    return pb < 1 ? 1 : pb;
}

int dtf_maxpool_bwd_bnred(const void* dy, const void* tap, const void* z, const float* mean, const float* scale,
                          const float* shift, void* dx, float* partial, int N, int H, int W, int C, int OH, int OW, int k,
                          int stride, int pad, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (k != 3 || stride != 2 || C % 8 != 0 || C > 2048 || 256 % (C / 8) != 0 || N <= 0 || OH <= 0 || OW <= 0 || pad < 0 || pad >= k)
        return -20;
    if (OH != (H + 2 * pad - k) / stride + 1 || OW != (W + 2 * pad - k) / stride + 1) return -21;
    const unsigned pb = (unsigned)dtf_maxpool_bwd_bnred_blocks(N, H, W, C);
    hipLaunchKernelGGL(maxpool_bwd_k3s2_bnred_kernel, dim3(pb), dim3(256), 0, stream, (const unsigned short*)dy,
                       (const unsigned char*)tap, (const unsigned short*)z, mean, scale, shift, (unsigned short*)dx, partial, N, H,
                       W, C, OH, OW, pad);
    DTF_CHECK_LAUNCH();
    return 0;
}

// ld: pixel stride of the full-resolution side (x forward, dx backward); 0 = C
int dtf_avgpool(const void* in, void* out, int backward, int N, int H, int W, int C, int OH, int OW, int k, int stride,
                int pad, int include_pad, int64_t ld, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (ld == 0) ld = C;
    if (ld < C || (ld & 7)) return -24;
    if (C % 8 != 0 || N <= 0 || OH <= 0 || OW <= 0 || stride < 1 || pad < 0 || pad >= k) return -20;
    if (OH != (H + 2 * pad - k) / stride + 1 || OW != (W + 2 * pad - k) / stride + 1) return -21;
    const int64_t items = (int64_t)N * (backward ? (int64_t)H * W : (int64_t)OH * OW) * (C / 8);
    int64_t blocks = (items + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    const unsigned short* is = (const unsigned short*)in;
    unsigned short* os = (unsigned short*)out;
#define AVG(KK)                                                                                                   \
    do {                                                                                                          \
        if (backward)                                                                                             \
            hipLaunchKernelGGL((avgpool_bwd_kernel<KK>), dim3((unsigned)blocks), dim3(256), 0, stream, is, os, N, \
                               H, W, C, OH, OW, stride, pad, include_pad, ld);                                    \
        else                                                                                                      \
            hipLaunchKernelGGL((avgpool_fwd_kernel<KK>), dim3((unsigned)blocks), dim3(256), 0, stream, is, os, N, \
                               H, W, C, OH, OW, stride, pad, include_pad, ld);                                    \
    } while (0)
    if (k == 3) AVG(3);
    else if (k == 2) AVG(2);
    else if (k == 5) AVG(5);
    else return -23;
#undef AVG
    DTF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
