// K6: bf16 MFMA GEMM with fused epilogue for gfx950 (replaces TF's
// MatMul + BiasAdd + Relu/Gelu chain on the worker):
//
//     Y[M,N] = act( X[M,K] . W[N,K]^T + bias[N] ) (+ residual[M,N])     bf16 in, fp32 acc, bf16 out
//
// (torch.nn.Linear layout: both operands are K-contiguous, which is exactly the
// fragment shape of v_mfma_f32_16x16x32_bf16: lane l holds 8 consecutive k of row l&15.)
//
// Structure (CDNA4, wave64):
//   * 128 x 128 output tile per 256-thread block, 4 waves as 2 x 2, each wave a 64 x 64
//     sub-tile = 4 x 4 MFMA 16x16 accumulators (64 fp32 regs per lane);
//   * BK = 64: A and B tiles staged global -> registers -> LDS as 16-B chunks
//     (global_load_dwordx4 / ds_write_b128), two LDS buffers; the loads of tile k+1 are issued
//     BEFORE the MFMAs of tile k and written to LDS AFTER them (issue-early / write-late),
//     one __syncthreads() per K step;
//   * LDS rows are 128 B; chunk' = chunk ^ (row & 7) XOR swizzle makes every ds_read_b128
//     fragment read conflict-free (rows r and r+2 would otherwise share banks: 8-way);
//   * epilogue: accumulators -> fp32 LDS tile (the two staging buffers are exactly 64 KB) ->
//     each thread takes 8 consecutive columns: + bias, activation, + residual, one rounding to
//     bf16, 16-B coalesced stores; optionally also stores the pre-activation (for GELU backward);
//   * 1-D grid with the bijective XCD remap: the 8 XCDs each get a contiguous range of tiles so
//     neighbouring tiles (same X row panel) hit the same 4 MiB L2.
// Requirements: K % 8 == 0, 16-B aligned operands; M, N arbitrary (edges predicated).
#include "dtf_common.h"

#define GM_BM 128
#define GM_BN 128
#define GM_BK 64

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7: far below bf16 resolution): one v_rcp, one
// v_exp and a degree-5 Horner chain instead of libm's branchy erff -- the epilogue must stay
// hidden behind the next tile's MFMAs of the co-resident block.
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.f - p * t * __expf(-ax * ax);
    return copysignf(e, x);
}

__device__ __forceinline__ float gelu_tanh_free(float x) {
    // exact (erf) GELU like torch.nn.functional.gelu default
    return 0.5f * x * (1.f + fast_erf(x * 0.70710678118654752f));
}

struct GemmArgs {
    const unsigned short* X; const unsigned short* W; const void* bias; const unsigned short* residual;
    unsigned short* Y; unsigned short* preact;
    int M, N, K; int bias_dtype; int act; int ntm, ntn;
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, pos = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
}

__global__ __launch_bounds__(256, 2) void gemm_bias_act_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // staging: buf b, operand o (0 = A, 1 = B): offset (b*2 + o) * 16 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = tile / a.ntn, tn = tile - tm * a.ntn;
    const int m0 = tm * GM_BM, n0 = tn * GM_BN;
    const int K = a.K;

    // per-thread staging chunks: 4 of A and 4 of B; chunk q = tid + 256*i -> row q>>3, kchunk q&7
    u32x4 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i, row = q >> 3, kc = q & 7;
            const int k = k0 + kc * 8;
            const int gm = m0 + row, gn = n0 + row;
            u32x4 z = {0u, 0u, 0u, 0u};
            ra[i] = (gm < a.M && k < K) ? *reinterpret_cast<const u32x4*>(a.X + (int64_t)gm * K + k) : z;
            rb[i] = (gn < a.N && k < K) ? *reinterpret_cast<const u32x4*>(a.W + (int64_t)gn * K + k) : z;
        }
    };
    auto lstore = [&](int buf) {
        unsigned char* A = smem + (buf * 2 + 0) * 16384;
        unsigned char* B = smem + (buf * 2 + 1) * 16384;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i, row = q >> 3, kc = q & 7;
            const int off = row * 128 + ((kc ^ (row & 7)) << 4);
            *reinterpret_cast<u32x4*>(A + off) = ra[i];
            *reinterpret_cast<u32x4*>(B + off) = rb[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (K + GM_BK - 1) / GM_BK;
    gload(0);
    lstore(0);
    __syncthreads();
    const int frow = lane & 15, fq = lane >> 4;          // fragment row / k-quarter of this lane
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * GM_BK);
        const unsigned char* A = smem + (buf * 2 + 0) * 16384;
        const unsigned char* B = smem + (buf * 2 + 1) * 16384;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t fa[4], fb[4];
            const int kc = kk * 4 + fq;                  // 16-B chunk index along k
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ar = wr * 64 + i * 16 + frow;
                fa[i] = *reinterpret_cast<const bf16x8_t*>(A + ar * 128 + ((kc ^ (ar & 7)) << 4));
                const int br = wc * 64 + i * 16 + frow;
                fb[i] = *reinterpret_cast<const bf16x8_t*>(B + br * 128 + ((kc ^ (br & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: accumulators -> fp32 LDS tile [128][128] (row stride 132 floats: 16-B aligned,
    // breaks the power-of-two stride) -> fused bias/act/residual -> coalesced bf16 stores
    float* ct = reinterpret_cast<float*>(smem);
    const int LDC = GM_BN + 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wr * 64 + i * 16 + fq * 4 + r;    // C/D map: row = (lane>>4)*4 + reg
                const int col = wc * 64 + j * 16 + frow;          //          col = lane & 15
                ct[row * LDC + col] = acc[i][j][r];
            }
    __syncthreads();
    const bool nvec = (a.N & 7) == 0;
    // 128 rows x 16 column-octets = 2048 items, 8 per thread
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int item = tid + 256 * it;
        const int row = item >> 4, c8 = (item & 15) << 3;
        const int gm = m0 + row, gn = n0 + c8;
        if (gm >= a.M || gn >= a.N) continue;
        float v[8];
        const f32x4 lo = *reinterpret_cast<const f32x4*>(ct + row * LDC + c8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(ct + row * LDC + c8 + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = lo[k]; v[4 + k] = hi[k]; }
        const int64_t o = (int64_t)gm * a.N + gn;
        if (nvec) {
            if (a.bias) {
                float b[8];
                if (a.bias_dtype == DTF_BF16) load8<DTF_BF16>(a.bias, gn, b); else load8<DTF_F32>(a.bias, gn, b);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += b[k];
            }
            if (a.preact) store8<DTF_BF16>(a.preact, o, v);
            if (a.act == ACT_RELU) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
            } else if (a.act == ACT_GELU) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = gelu_tanh_free(v[k]);
            }
            if (a.residual) {
                float r[8];
                load8<DTF_BF16>(a.residual, o, r);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += r[k];
            }
            store8<DTF_BF16>(a.Y, o, v);
        } else {
            for (int k = 0; k < 8 && gn + k < a.N; ++k) {
                float x = v[k];
                if (a.bias) x += (a.bias_dtype == DTF_BF16) ? load1<DTF_BF16>(a.bias, gn + k) : load1<DTF_F32>(a.bias, gn + k);
                if (a.preact) a.preact[o + k] = f32_to_bf16(x);
                if (a.act == ACT_RELU) x = fmaxf(x, 0.f);
                else if (a.act == ACT_GELU) x = gelu_tanh_free(x);
                if (a.residual) x += bf16_to_f32(a.residual[o + k]);
                a.Y[o + k] = f32_to_bf16(x);
            }
        }
    }
}

// Backward helper of the fused epilogue: dZ = dY * act'(Z) (bf16) and dbias[n] = sum_m dZ[m,n]
// in one pass (two-stage deterministic column reduction like the BN kernels).
//   ACT_RELU: Z argument is the forward OUTPUT (mask = y > 0); ACT_GELU: Z is the pre-activation.
__global__ __launch_bounds__(256) void act_bwd_bias_kernel(const unsigned short* __restrict__ dy,
                                                           const unsigned short* __restrict__ z,
                                                           unsigned short* __restrict__ dz,
                                                           float* __restrict__ partial, int64_t M, int N, int act,
                                                           int rows_per_block) {
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    const int c0 = blockIdx.y * 64 + tx * 8;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    if (c0 < N) {
        for (int64_t r = r0 + ty; r < r1; r += 32) {
            float d[8], zz[8];
            load8<DTF_BF16>(dy, r * N + c0, d);
            if (act != ACT_NONE) load8<DTF_BF16>(z, r * N + c0, zz);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float g = d[k];
                if (act == ACT_RELU) g = zz[k] > 0.f ? g : 0.f;
                else if (act == ACT_GELU) {
                    const float x = zz[k];
                    const float cdf = 0.5f * (1.f + fast_erf(x * 0.70710678118654752f));
                    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
                    g *= cdf + x * pdf;
                }
                d[k] = g;
                s[k] += g;
            }
            if (dz) store8<DTF_BF16>(dz, r * N + c0, d);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) s[k] += __shfl_xor(s[k], o, 64);
    __shared__ float red[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < 8)
#pragma unroll
        for (int k = 0; k < 8; ++k) red[wave][lane * 8 + k] = s[k];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = blockIdx.y * 64 + threadIdx.x;
        if (c < N) partial[(int64_t)blockIdx.x * N + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
}

template <int PDT>
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ partial, int nblk, int N,
                                                              void* out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    float s = 0.f;
    for (int j = 0; j < nblk; ++j) s += partial[(int64_t)j * N + c];
    store1<PDT>(out, c, s);
}

extern "C" {

int dtf_gemm_bias_act(const void* X, const void* W, const void* bias, int bias_dtype, const void* residual, void* Y,
                      void* preact, int M, int N, int K, int act, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || (K & 7)) return -40;
    if ((((uintptr_t)X) | ((uintptr_t)W) | ((uintptr_t)Y) | ((uintptr_t)residual) | ((uintptr_t)preact)) & 15) return -16;
    GemmArgs a{(const unsigned short*)X, (const unsigned short*)W, bias, (const unsigned short*)residual,
               (unsigned short*)Y, (unsigned short*)preact, M, N, K, bias_dtype, act,
               (M + GM_BM - 1) / GM_BM, (N + GM_BN - 1) / GM_BN};
    static bool attr_set = false;
    const int lds = GM_BM * (GM_BN + 4) * 4;     // 67584 B: epilogue tile (>= 64 KB staging)
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm_bias_act_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_bias_act_kernel, dim3(a.ntm * a.ntn), dim3(256), lds, stream, a);
    DTF_CHECK_LAUNCH();
    return 0;
}

int64_t dtf_act_bwd_partial_floats(int64_t M, int N) {
    int64_t nb = (M + 255) / 256;
    if (nb > 512) nb = 512;
    return nb * N;
}

// dz may alias dy (in place) or be null (bias grad only, act == NONE)
int dtf_act_bwd_bias(const void* dy, const void* z, void* dz, void* dbias, int bias_dtype, float* partial, int64_t M,
                     int N, int act, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (M <= 0 || N <= 0) return 0;
    if (N & 7) return -41;
    int64_t nb = (M + 255) / 256;
    if (nb > 512) nb = 512;
    int64_t rpb = (M + nb - 1) / nb;
    rpb = (rpb + 31) / 32 * 32;
    nb = (M + rpb - 1) / rpb;
    hipLaunchKernelGGL(act_bwd_bias_kernel, dim3((int)nb, (N + 63) / 64), dim3(256), 0, stream,
                       (const unsigned short*)dy, (const unsigned short*)z, (unsigned short*)dz, partial, M, N, act,
                       (int)rpb);
    if (dbias) {
        if (bias_dtype == DTF_BF16)
            hipLaunchKernelGGL((colsum_finalize_kernel<DTF_BF16>), dim3((N + 255) / 256), dim3(256), 0, stream,
                               partial, (int)nb, N, dbias);
        else
            hipLaunchKernelGGL((colsum_finalize_kernel<DTF_F32>), dim3((N + 255) / 256), dim3(256), 0, stream,
                               partial, (int)nb, N, dbias);
    }
    DTF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
