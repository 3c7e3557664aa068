// K12: softmax cross-entropy over bf16 logits for gfx950 -- the MLM / classifier head loss without the fp32
// copy of the logits.  The framework path (logits.float() -> log_softmax -> nll) moves a [4864 x 30522] BERT
// head through HBM ~10 times in fp32; here the forward reads the bf16 row once (online max / sum-exp, fp32 in
// registers) and keeps one log-sum-exp per row, the backward reads it once more and writes bf16 dlogits:
//   loss[r]      = lse[r] - x[r, label[r]]                       (0 for ignored rows)
//   dlogits[r,j] = (exp(x[r,j] - lse[r]) - [j == label[r]]) * row_scale[r]
// One 256-thread block per row.  Rows are only 2-byte aligned in general (V = 30522): the row is cut into an
// unaligned head, a 16-byte-aligned body read 8 elements per lane, and a tail.
#include "dtf_common.h"
#include <math.h>

#define CE_T 256

__device__ __forceinline__ void ce_online(float x, float& m, float& s) {
    if (x > m) { s = s * __expf(m - x) + 1.f; m = x; }
    else s += __expf(x - m);
}

__device__ __forceinline__ void ce_combine(float& m, float& s, float m2, float s2) {
    const float mm = fmaxf(m, m2);
    s = (m == -INFINITY ? 0.f : s * __expf(m - mm)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mm));
    m = mm;
}

__global__ __launch_bounds__(CE_T) void ce_fwd_kernel(const unsigned short* __restrict__ logits,
                                                      const int64_t* __restrict__ labels, float* __restrict__ lse_out,
                                                      float* __restrict__ loss_out, int V, int64_t ignore_index) {
    const int64_t r = blockIdx.x;
    const unsigned short* row = logits + r * V;
    const int head = (int)(((16 - ((uintptr_t)row & 15)) & 15) >> 1);      // elements before the first 16-B boundary
    const int h = head < V ? head : V;
    const int nbody = (V - h) >> 3;
    float m = -INFINITY, s = 0.f;
    if ((int)threadIdx.x < h) ce_online(bf16_to_f32(row[threadIdx.x]), m, s);
    const u32x4* body = reinterpret_cast<const u32x4*>(row + h);
    // a read-only pass: four independent 16-byte loads per lane in flight before any of them is consumed
    // (one load per lane left the kernel latency-bound at 3.6 TB/s, profiles/r1_14)
    int i = threadIdx.x;
    for (; i + 3 * CE_T < nbody; i += 4 * CE_T) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = body[i + u * CE_T];
        float x[32];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { x[8 * u + 2 * k] = __uint_as_float(v[u][k] << 16); x[8 * u + 2 * k + 1] = __uint_as_float(v[u][k] & 0xffff0000u); }
        }
        float mx = x[0];
#pragma unroll
        for (int k = 1; k < 32; ++k) mx = fmaxf(mx, x[k]);
        if (mx > m) { s *= __expf(m - mx); m = mx; }
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            s0 += __expf(x[k] - m); s1 += __expf(x[8 + k] - m); s2 += __expf(x[16 + k] - m); s3 += __expf(x[24 + k] - m);
        }
        s += (s0 + s1) + (s2 + s3);
    }
    for (; i < nbody; i += CE_T) {
        const u32x4 v = body[i];
        float x[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { x[2 * k] = __uint_as_float(v[k] << 16); x[2 * k + 1] = __uint_as_float(v[k] & 0xffff0000u); }
        float mx = x[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) mx = fmaxf(mx, x[k]);
        if (mx > m) { s *= __expf(m - mx); m = mx; }
#pragma unroll
        for (int k = 0; k < 8; ++k) s += __expf(x[k] - m);
    }
    for (int j = h + nbody * 8 + threadIdx.x; j < V; j += CE_T) ce_online(bf16_to_f32(row[j]), m, s);
    // wave then block combine
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        ce_combine(m, s, m2, s2);
    }
    __shared__ float sm[CE_T / 64], ss[CE_T / 64];
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = m; ss[threadIdx.x >> 6] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float M = sm[0], S = ss[0];
        for (int w = 1; w < CE_T / 64; ++w) ce_combine(M, S, sm[w], ss[w]);
        const float lse = M + __logf(S);
        lse_out[r] = lse;
        const int64_t y = labels[r];
        loss_out[r] = (y == ignore_index || y < 0 || y >= V) ? 0.f : lse - bf16_to_f32(row[y]);
    }
}

__global__ __launch_bounds__(CE_T) void ce_bwd_kernel(const unsigned short* __restrict__ logits,
                                                      const int64_t* __restrict__ labels, const float* __restrict__ lse,
                                                      const float* __restrict__ row_scale, unsigned short* __restrict__ dlogits,
                                                      int V, int64_t ignore_index) {
    const int64_t r = blockIdx.x;
    const unsigned short* row = logits + r * V;
    unsigned short* drow = dlogits + r * V;
    const float l = lse[r], g = row_scale[r];
    int64_t y = labels[r];
    if (y == ignore_index || y < 0 || y >= V) y = -1;
    const int head = (int)(((16 - ((uintptr_t)row & 15)) & 15) >> 1);
    const int h = head < V ? head : V;
    const int nbody = (V - h) >> 3;
    if ((int)threadIdx.x < h) {
        const int j = threadIdx.x;
        drow[j] = f32_to_bf16((__expf(bf16_to_f32(row[j]) - l) - (j == y ? 1.f : 0.f)) * g);
    }
    const u32x4* body = reinterpret_cast<const u32x4*>(row + h);
    u32x4* dbody = reinterpret_cast<u32x4*>(drow + h);          // same alignment as the logits row (same offset r * V)
    for (int i = threadIdx.x; i < nbody; i += CE_T) {
        const u32x4 v = body[i];
        const int j0 = h + i * 8;
        float x[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { x[2 * k] = __uint_as_float(v[k] << 16); x[2 * k + 1] = __uint_as_float(v[k] & 0xffff0000u); }
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = (__expf(x[k] - l) - ((int64_t)(j0 + k) == y ? 1.f : 0.f)) * g;
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pack_bf16x2(x[2 * k], x[2 * k + 1]);
        dbody[i] = o;
    }
    for (int j = h + nbody * 8 + threadIdx.x; j < V; j += CE_T)
        drow[j] = f32_to_bf16((__expf(bf16_to_f32(row[j]) - l) - (j == y ? 1.f : 0.f)) * g);
}

extern "C" {

// logits: bf16 [M, V] contiguous; labels: int64 [M]; lse, loss: fp32 [M]
int dtf_ce_fwd(const void* logits, const int64_t* labels, float* lse, float* loss, int64_t M, int V, int64_t ignore_index,
               hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (M <= 0 || V <= 0) return 0;
    if (M > 0x7fffffff) return -80;
    hipLaunchKernelGGL(ce_fwd_kernel, dim3((unsigned)M), dim3(CE_T), 0, stream, (const unsigned short*)logits, labels, lse, loss, V,
                       ignore_index);
    DTF_CHECK_LAUNCH();
    return 0;
}

// dlogits: bf16 [M, V], allocated like logits (both bases 16-byte aligned); row_scale: fp32 [M] = dLoss/dloss[r]
int dtf_ce_bwd(const void* logits, const int64_t* labels, const float* lse, const float* row_scale, void* dlogits, int64_t M,
               int V, int64_t ignore_index, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (M <= 0 || V <= 0) return 0;
    if (M > 0x7fffffff) return -80;
    if ((((uintptr_t)logits) | ((uintptr_t)dlogits)) & 15) return -16;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)M), dim3(CE_T), 0, stream, (const unsigned short*)logits, labels, lse,
                       row_scale, (unsigned short*)dlogits, V, ignore_index);
    DTF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
