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
#include <stdlib.h>

#define GM_BM 128
#define GM_BN 128
#define GM_BK 64

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_GELU_TANH = 3 };

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

// tanh-form GELU (the original BERT / GPT-2 definition): x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3);
// one v_exp + one v_rcp: about a third cheaper than the erf form in an epilogue that is pure VALU.
__device__ __forceinline__ float gelu_tanh_fwd(float x) {
    const float t = x * fmaf(-0.1029432f, x * x, -2.3022082f);        // -2u * log2(e)
    return x * __frcp_rn(1.f + exp2f(t));
}

// d/dx [x * s(x)], s = sigmoid(2u):  s + x * s * (1 - s) * 2u',  2u' = 2 sqrt(2/pi) (1 + 3*0.044715 x^2)
__device__ __forceinline__ float gelu_tanh_grad(float x) {
    const float x2 = x * x;
    const float t = x * fmaf(-0.1029432f, x2, -2.3022082f);
    const float s = __frcp_rn(1.f + exp2f(t));
    return fmaf(x * s * (1.f - s), fmaf(0.2140585f, x2, 1.5957691f), s);
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
            } else if (a.act == ACT_GELU_TANH) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = gelu_tanh_fwd(v[k]);
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
                else if (a.act == ACT_GELU_TANH) x = gelu_tanh_fwd(x);
                if (a.residual) x += bf16_to_f32(a.residual[o + k]);
                a.Y[o + k] = f32_to_bf16(x);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// 256 x 256 tile variant for large problems (>= ~2 tiles per CU): 8 waves as 2(M) x 4(N), each a
// 128 x 64 sub-tile (8 x 4 accumulators = 128 fp32 regs), BK = 64, ~1 block per CU.
// Staging is LDS-DMA (global_load_lds, 16 B per lane): no staging registers, no ds_write pass.  A
// wave instruction fills 1 KB linearly (see the staging comment in the kernel).  Four 32 KB LDS buffers:
// three K tiles stream in while one feeds the MFMAs; one raw barrier per K step.  Rows past M / N are clamped to the last valid row (their results are never
// stored); K must be a multiple of 32 here (no zero fill with LDS-DMA) -- other K use the 128 kernel.
// The fused epilogue runs in two 128-column halves through an fp32 LDS tile.
// ---------------------------------------------------------------------------
#define G2_BM 256
#define G2_BN 256
#define G2_LDC 132

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__global__ __launch_bounds__(512) void gemm256_bias_act_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = tile / a.ntn, tn = tile - tm * a.ntn;
    const int m0 = tm * G2_BM, n0 = tn * G2_BN;
    const int K = a.K;

    // staging: K tiles of 32 (64-B LDS rows), FOUR 32 KB buffers.  One wave instruction fills 1 KB =
    // 16 rows x 64 B, which is exactly the image of one 16-row MFMA fragment group: lane (frow, fq)
    // reads byte frow*64 + fq*16 of it, i.e. the 64 lanes of a ds_read_b128 cover the 1 KB linearly --
    // conflict-free with no swizzle.  Wave w stages row groups 2w, 2w+1 of A and of B.
    const unsigned short* pa[2];
    const unsigned short* pb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 16 + (lane >> 2);
        int gm = m0 + row, gn = n0 + row;
        if (gm > a.M - 1) gm = a.M - 1;
        if (gn > a.N - 1) gn = a.N - 1;
        pa[i] = a.X + (int64_t)gm * K + (lane & 3) * 8;
        pb[i] = a.W + (int64_t)gn * K + (lane & 3) * 8;
    }
    auto stage = [&](int t) {
        unsigned char* base = smem + (t & 3) * 32768 + wave * 2048;
        const int k0 = t * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((glb_void_t*)(pa[i] + k0), (lds_void_t*)(base + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void_t*)(pb[i] + k0), (lds_void_t*)(base + 16384 + i * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Pipeline.  (1) Tiles t+1 .. t+3 are in flight while tile t feeds the MFMAs (~3 K tiles of
    // look-ahead; one tile ahead is shorter than the loaded HBM/L2 latency and ran at 50 % MFMA rate).
    // The LDS-DMA stays in flight ACROSS barriers: counted vmcnt (4 glds per wave per tile) and raw
    // s_barrier -- __syncthreads() would drain vmcnt(0).
    // (2) Ping-pong: the two waves that share a SIMD (w and w+4, i.e. wr = 0 / 1) run one barrier apart,
    // so while one half of the block issues its 32 MFMAs the other half reads its fragments from LDS:
    //
    //   wr=0:  R0 | M0 w1 | R1 | M1 w2 | R2 ...            R = 12 ds_read_b128 (+ restage), M = 32 MFMAs,
    //   wr=1:     | R0 w1 | M0 | R1 w2 | M1 ...            wN = wait for own pieces of tile N, | = s_barrier
    //
    // Tile N is first read (by wr=0) after the barrier that follows both groups' wN.  Buffers (t+2)&3 and
    // (t+3)&3 == (t-1)&3 are restaged in phase R(t), which for both groups starts after the barrier that
    // ended the last read of tile t-1 (wr=1's R(t-1)).
    const int nk = K / 32;
    const int frow = lane & 15, fq = lane >> 4;
    const int lane_off = frow * 64 + fq * 16;
    // Tiles are staged in PAIRS (at even t: tiles t+2 and t+3) so that the two 64-B halves of every
    // 128-B line of X / W are requested back to back -- one K tile at a time re-fetched half lines from
    // HBM once the operand panels outgrew L2 (8192^3: 931 vs 1205 TF).
    // tiles that may still be in flight when tile t+1 is needed: staged so far minus (t+1)
    // K <= 4096 keeps the deeper single-tile schedule (stage t+3 every step): 4-8 % faster there.
    const bool pair = K > 4096;
#define G2_WAIT_NEXT(T)                                                              \
    do {                                                                             \
        int upto = (pair && ((T) & 1)) ? (T) + 2 : (T) + 3;                          \
        if (upto > nk - 1) upto = nk - 1;                                            \
        const int n_out = upto - ((T) + 1);                                          \
        if (n_out >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             \
        else if (n_out == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        \
    } while (0)
    stage(0);
    if (nk > 1) stage(1);
    if (!pair && nk > 2) stage(2);
    {
        int staged = pair ? 2 : 3;
        if (staged > nk) staged = nk;
        if (staged == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (staged == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // stagger the second wave of every SIMD
    for (int t = 0; t < nk; ++t) {
        if (!pair) {
            if (t + 3 < nk) stage(t + 3);
        } else if ((t & 1) == 0) {
            if (t + 2 < nk) stage(t + 2);
            if (t + 3 < nk) stage(t + 3);
        }
        const unsigned char* A = smem + (t & 3) * 32768 + wr * 8192 + lane_off;
        const unsigned char* B = smem + (t & 3) * 32768 + 16384 + wc * 4096 + lane_off;
        bf16x8_t fa[8], fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(B + j * 1024);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(A + i * 1024);
        if (wr == 1) G2_WAIT_NEXT(t);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (wr == 0) G2_WAIT_NEXT(t);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
#undef G2_WAIT_NEXT
    __syncthreads();

    float* ct = reinterpret_cast<float*>(smem);
    const bool nvec = (a.N & 7) == 0;
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        if ((wc >> 1) == h) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = wr * 128 + i * 16 + fq * 4 + r;
                        const int col = (wc & 1) * 64 + j * 16 + frow;
                        ct[row * G2_LDC + col] = acc[i][j][r];
                    }
        }
        __syncthreads();
#pragma unroll 2
        for (int it = 0; it < 8; ++it) {
            const int item = tid + 512 * it;
            const int row = item >> 4, c8 = (item & 15) << 3;
            const int gm = m0 + row, gn = n0 + h * 128 + c8;
            if (gm >= a.M || gn >= a.N) continue;
            float v[8];
            const f32x4 lo = *reinterpret_cast<const f32x4*>(ct + row * G2_LDC + c8);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(ct + row * G2_LDC + c8 + 4);
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
                } else if (a.act == ACT_GELU_TANH) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = gelu_tanh_fwd(v[k]);
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
                else if (a.act == ACT_GELU_TANH) x = gelu_tanh_fwd(x);
                    else if (a.act == ACT_GELU_TANH) x = gelu_tanh_fwd(x);
                    if (a.residual) x += bf16_to_f32(a.residual[o + k]);
                    a.Y[o + k] = f32_to_bf16(x);
                }
            }
        }
        __syncthreads();
    }
}

// Backward helper of the fused epilogue: dZ = dY * act'(Z) (bf16) and dbias[n] = sum_m dZ[m,n]
// in one pass (two-stage deterministic column reduction like the BN kernels).
//   ACT_RELU: Z argument is the forward OUTPUT (mask = y > 0); ACT_GELU / ACT_GELU_TANH: Z is the pre-activation.
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
                } else if (act == ACT_GELU_TANH) {
                    g *= gelu_tanh_grad(zz[k]);
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

// One block per 8 columns: 32 row lanes x 8 columns, every lane keeps up to 16 independent loads in
// flight (the first version summed all <= 512 partial rows serially per thread: 34 us per call).
// Fixed summation order -> bitwise reproducible bias gradients.
template <int PDT>
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ partial, int nblk, int N,
                                                              void* out) {
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    float s = 0.f;
    if (c < N) {
#pragma unroll 4
        for (int j = rl; j < nblk; j += 32) s += partial[(int64_t)j * N + c];
    }
    // lanes with equal (lane & 7) hold the same column: fold lane bits 3..5, then the 4 waves in order
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
    __shared__ float red[4][8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < 8) red[wave][lane] = s;
    __syncthreads();
    if (threadIdx.x < 8 && c < N)
        store1<PDT>(out, c, ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x]);
}

static int force_tile = 0;       // 128 | 256 pins the tile variant (DTF_GEMM_TILE or dtf_gemm_set_tile; A/B runs)

extern "C" {

void dtf_gemm_set_tile(int t) { force_tile = t; }

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
    const int lds256 = G2_BM * G2_LDC * 4;       // 135168 B: epilogue half tile (>= 128 KB staging)
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm_bias_act_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipFuncSetAttribute((const void*)gemm256_bias_act_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds256);
        const char* e = getenv("DTF_GEMM_TILE");
        if (e && force_tile == 0) force_tile = atoi(e);
        attr_set = true;
    }
    const int ntm2 = (M + G2_BM - 1) / G2_BM, ntn2 = (N + G2_BN - 1) / G2_BN;
    bool big = (K % 32) == 0 && (int64_t)ntm2 * ntn2 >= 384;     // >= 1.5 tiles per CU
    if (force_tile == 128) big = false;
    if (force_tile == 256 && (K % 32) == 0) big = true;
    if (big) {
        a.ntm = ntm2;
        a.ntn = ntn2;
        hipLaunchKernelGGL(gemm256_bias_act_kernel, dim3(ntm2 * ntn2), dim3(512), lds256, stream, a);
    } else {
        hipLaunchKernelGGL(gemm_bias_act_kernel, dim3(a.ntm * a.ntn), dim3(256), lds, stream, a);
    }
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
            hipLaunchKernelGGL((colsum_finalize_kernel<DTF_BF16>), dim3((N + 7) / 8), dim3(256), 0, stream,
                               partial, (int)nb, N, dbias);
        else
            hipLaunchKernelGGL((colsum_finalize_kernel<DTF_F32>), dim3((N + 7) / 8), dim3(256), 0, stream,
                               partial, (int)nb, N, dbias);
    }
    DTF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
