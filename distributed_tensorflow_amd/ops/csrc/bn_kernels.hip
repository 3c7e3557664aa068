// K7: conv epilogues for gfx950 -- training-mode BatchNorm fused with the
// residual add and the ReLU that follow a convolution, NHWC bf16 (SURVEY 2.4;
// replaces TF's FusedBatchNorm + Add + Relu on the worker).  The convolution
// itself stays in MIOpen; everything between two convolutions is these kernels.
//
// The tensor is viewed as [M = N*H*W rows][C channels], channels innermost.
// All kernels are HBM-bound; the point of the fusion is the pass count:
//
//   forward   stats   : read x                      (2 B/elem)
//             apply   : read x (+res), write y      (4-6 B/elem)
//   backward  reduce  : read dy, x (+1 bit mask)     (4 B/elem)
//             apply   : read dy, x (+mask), write dx (+dres)   (6-8 B/elem)
//
// against ~26-38 B/elem for separate BN / add / ReLU ops.
//
// Reductions are two-stage and deterministic: a [row-chunk][channel-tile] grid
// writes per-block partial sums, a small finalize kernel adds them in fixed
// order (no float atomics -> bitwise reproducible statistics).
//
// Thread mapping of the streaming kernels: a 256-thread block covers a tile of
// 64 channels (8 lanes x 8 channels = one 128-B line per row) x 32 rows per
// iteration, 4 iterations unrolled so every lane keeps 4-8 16-B loads in flight.
#include "dtf_common.h"
#include <stdlib.h>

#define BN_CT 64          // channels per block tile
#define BN_ROWS 32        // rows per block iteration
#define BN_UNROLL 4

__device__ __forceinline__ int64_t dtf_mask_bytes_dev(int64_t M, int C) { return (int64_t)((C + BN_CT - 1) / BN_CT) * M * 8; }

enum { BN_FWD_STATS = 0, BN_BWD_RELU_X = 1, BN_BWD_PLAIN = 2, BN_BWD_RELU_MASK = 3 };

// Row-chunk schedule.  rows_per_block > 0: block b owns the contiguous rows [b * rows_per_block, ...) (round 1).
// rows_per_block == 0 ("sweep", the default): the grid walks the tensor together -- block b takes chunks b,
// b + gridDim.x, ... of BN_ROWS * U rows, so at any moment the resident blocks read ONE window of adjacent rows.
// With 768 private ranges the chip streamed from 768 scattered places at once and lost DRAM page locality: measured
// on a cold [802816 x 256] bf16 pass (x, residual -> y): 249 us vs 194-216 us (scripts/ubench/stream_variants.hip,
// profiles/r2_bn_sweep_ab.txt).  Reduction order per block stays fixed -> statistics remain bitwise reproducible.
#define BN_SCHED(U)                                                                                          \
    const bool sweep__ = rows_per_block == 0;                                                                \
    const int64_t r0 = sweep__ ? (int64_t)blockIdx.x * (BN_ROWS * (U)) : (int64_t)blockIdx.x * rows_per_block; \
    const int64_t rstep = sweep__ ? (int64_t)gridDim.x * (BN_ROWS * (U)) : (int64_t)(BN_ROWS * (U));          \
    int64_t r1 = sweep__ ? M : r0 + rows_per_block;                                                          \
    if (r1 > M) r1 = M;
// how the backward kernels learn the ReLU mask:
//   RELU_X    recompute  fma(x, scale, shift) > 0  -- bitwise the forward's own decision,
//             no extra tensor read (BN+ReLU without residual)
//   RELU_MASK 1 bit per element written by the forward apply (BN+add+ReLU): 0.125 B/elem
//             instead of re-reading the 2 B/elem output
// mask layout: [channel tile][row][8 bytes], byte tx holds the 8 channels of lane tx

__device__ __forceinline__ void ld8bf(const unsigned short* p, float (&v)[8]) {
    u32x4 r = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[2 * k] = __uint_as_float(r[k] << 16);
        v[2 * k + 1] = __uint_as_float(r[k] & 0xffff0000u);
    }
}

__device__ __forceinline__ void st8bf(unsigned short* p, const float (&v)[8]) {
    u32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = pack_bf16x2(v[2 * k], v[2 * k + 1]);
    *reinterpret_cast<u32x4*>(p) = r;
}

// POOL variants of the backward kernels: the layer's output went straight into a max pool (ops/pool.py
// bn_relu_max_pool2d), so its gradient is not a tensor but the pool's gather over the windows covering each pixel
// (K9's backward formulation) of the POOLED gradient + winning-tap bytes -- the full-resolution gradient is never
// written or read.  Same arithmetic as maxpool_bwd_kernel, including the bf16 rounding of the per-pixel sum.
struct PoolGeom {
    const unsigned short* dyp; const unsigned char* tap;
    int H, W, OH, OW, K, stride, pad;
};

__device__ __forceinline__ void pool_gather_dy(const PoolGeom& g, int64_t row, int C, int c0, float (&d)[8]) {
    const unsigned r = (unsigned)row;
    const int w = (int)(r % (unsigned)g.W);
    const unsigned q = r / (unsigned)g.W;
    const int h = (int)(q % (unsigned)g.H);
    const int n = (int)(q / (unsigned)g.H);
    int oh_lo = h + g.pad - (g.K - 1);
    oh_lo = oh_lo <= 0 ? 0 : (oh_lo + g.stride - 1) / g.stride;
    int oh_hi = (h + g.pad) / g.stride;
    if (oh_hi > g.OH - 1) oh_hi = g.OH - 1;
    int ow_lo = w + g.pad - (g.K - 1);
    ow_lo = ow_lo <= 0 ? 0 : (ow_lo + g.stride - 1) / g.stride;
    int ow_hi = (w + g.pad) / g.stride;
    if (ow_hi > g.OW - 1) ow_hi = g.OW - 1;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
        const unsigned kh = (unsigned)(h + g.pad - oh * g.stride);
        for (int ow = ow_lo; ow <= ow_hi; ++ow) {
            const unsigned me = kh * (unsigned)g.K + (unsigned)(w + g.pad - ow * g.stride);
            const int64_t o = DTF_IDX((((int64_t)n * g.OH + oh) * g.OW + ow) * C + c0,
                                      (int64_t)(n + 1) * g.OH * g.OW * C, "bn pool-gather window");
            const u32x2 pk = *reinterpret_cast<const u32x2*>(g.tap + o);
            float v[8];
            ld8bf(g.dyp + o, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned b = (pk[k >> 2] >> ((k & 3) * 8)) & 0xffu;
                acc[k] += (b == me) ? v[k] : 0.f;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k] = bf16_to_f32(f32_to_bf16(acc[k]));
}

// Backward finalize folded into the reduction (round 3).  The stand-alone finalize kernel sits on the critical path
// between the reduce and the apply pass of every layer, and once the weight gradients run on their side stream its few
// small blocks share the chip with a GEMM: 5 us alone, 11-17 us measured in the step, 53 times.  Here the blocks of one
// channel tile (blockIdx.y) take a ticket after their partial row is visible device-wide; the block that draws the last
// ticket folds the tile's gridDim.x rows in a fixed order (so the result does not depend on WHICH block is last --
// bitwise reproducible) and writes dgamma, dbeta and the apply coefficients for its 64 channels.  The tickets reset
// themselves, so the counters stay zero between launches (and across hipGraph replays).
struct BnFin {
    unsigned* cnt;            // [channel tiles] tickets, zero on entry; null = no fused finalize
    const void* gamma; const float* invstd;
    void* dgamma; void* dbeta; float* coef;
    float inv_count; int pdt;
};

// partial[(blockIdx.x * 2 + stat) * C + c]
// U = rows in flight per lane: the single-operand statistics pass needs 8 (64 B per lane is too
// little to cover the HBM latency: 3.3 TB/s measured), the two-operand backward pass 4.
template <int MODE, int U, bool POOL = false>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const unsigned short* __restrict__ x,
                                                        const unsigned short* __restrict__ dy,
                                                        const unsigned char* __restrict__ mask,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift,
                                                        float* __restrict__ partial, int64_t M, int C,
                                                        int rows_per_block, int64_t lddy, int64_t ldx,
                                                        PoolGeom pg = PoolGeom{}, BnFin fin = BnFin{}) {
    const int tx = threadIdx.x & 7;
    const int ty = threadIdx.x >> 3;
    const int c0 = blockIdx.y * BN_CT + tx * 8;
    const bool cvalid = c0 < C;
    const int bx = blockIdx.x;
    BN_SCHED(U)
    float s1[8], s2[8], mu[8], sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; mu[k] = 0.f; sc[k] = 0.f; sh[k] = 0.f; }
    if (MODE == BN_FWD_STATS && cvalid) {
        // shifted accumulation: sums of (x - s) and (x - s)^2 with s = the channel's value in row 0 (a sample of
        // the distribution, identical for every block).  var = E[d^2] - E[d]^2 then cancels only digits of the
        // SPREAD, not of the mean: E[x^2] - E[x]^2 in fp32 loses the variance once |mean| > ~300 std.
        ld8bf(x + c0, mu);
    }
    if (MODE != BN_FWD_STATS && cvalid) {
#pragma unroll
        for (int k = 0; k < 8; ++k) mu[k] = mean[c0 + k];
        if (MODE == BN_BWD_RELU_X) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { sc[k] = scale[c0 + k]; sh[k] = shift[c0 + k]; }
        }
    }
    const unsigned char* mrow = mask + ((int64_t)blockIdx.y * M) * 8 + tx;
    if (cvalid) {
        for (int64_t r = r0 + ty; r < r1; r += rstep) {
            float xv[U][8], dv[U][8];
            unsigned mb[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t row = r + (int64_t)u * BN_ROWS;
                ok[u] = row < r1;
                if (ok[u]) {
                    (void)DTF_IDX(row, M, "bn_reduce row");
                    (void)DTF_IDX(c0 + 7, C, "bn_reduce channel");
                    ld8bf(x + row * ldx + c0, xv[u]);
                    if (MODE != BN_FWD_STATS) {
                        if (POOL) pool_gather_dy(pg, row, C, c0, dv[u]);
                        else ld8bf(dy + row * lddy + c0, dv[u]);
                    }
                    if (MODE == BN_BWD_RELU_MASK) mb[u] = mrow[row * 8];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!ok[u]) continue;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (MODE == BN_FWD_STATS) {
                        const float d = xv[u][k] - mu[k];
                        s1[k] += d;
                        s2[k] = fmaf(d, d, s2[k]);
                    } else {
                        float d = dv[u][k];
                        if (MODE == BN_BWD_RELU_X) d = fmaf(xv[u][k], sc[k], sh[k]) > 0.f ? d : 0.f;
                        if (MODE == BN_BWD_RELU_MASK) d = ((mb[u] >> k) & 1u) ? d : 0.f;
                        s1[k] += d;
                        s2[k] = fmaf(d, xv[u][k] - mu[k], s2[k]);
                    }
                }
            }
        }
    }
    // lanes with equal (lane & 7) hold the same channels: fold lane bits 3..5
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
            s1[k] += __shfl_xor(s1[k], o, 64);
            s2[k] += __shfl_xor(s2[k], o, 64);
        }
    }
    __shared__ float red[4][2][BN_CT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            red[wave][0][lane * 8 + k] = s1[k];
            red[wave][1][lane * 8 + k] = s2[k];
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * BN_CT) {
        const int stat = threadIdx.x / BN_CT, cc = threadIdx.x % BN_CT;
        const int c = blockIdx.y * BN_CT + cc;
        if (c < C) {
            float s = red[0][stat][cc] + red[1][stat][cc] + red[2][stat][cc] + red[3][stat][cc];
            float* dst = partial + ((int64_t)bx * 2 + stat) * C + c;
            // fused finalize: write-through (sc1) stores, so the hand-off needs no L2 write-back in 768 blocks (a
            // __threadfence() per block measured +4 ms on the ResNet-50 step: every fence walks the L2's dirty lines,
            // the previous kernel's output included)
            if (MODE != BN_FWD_STATS && fin.cnt) __hip_atomic_store(dst, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *dst = s;
        }
    }
    if (MODE == BN_FWD_STATS || fin.cnt == nullptr) return;
    // ---- fused finalize: the last block of this channel tile folds the tile's partial rows ----
    __shared__ int last_s;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave: its rows have left the CU ...
    __syncthreads();                                   // ... before the one lane that signals for all of them adds
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(fin.cnt + blockIdx.y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == gridDim.x - 1;
        last_s = last;
        if (last) {                                    // ONE agent acquire, in the reducer only
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!last_s) return;
    const int nblk = (int)gridDim.x;
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;          // 16 channel quads x 16 row lanes
    const int cb = blockIdx.y * BN_CT + cq * 4;
    f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
    if (cb < C) {
        const float* pp = partial + cb;
#pragma unroll 8
        for (int j = rl; j < nblk; j += 16) {
            a1 += *reinterpret_cast<const f32x4*>(pp + ((int64_t)j * 2 + 0) * C);
            a2 += *reinterpret_cast<const f32x4*>(pp + ((int64_t)j * 2 + 1) * C);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {         // row lanes of one wave: lane bits 4, 5
        a1[k] += __shfl_xor(a1[k], 16, 64); a2[k] += __shfl_xor(a2[k], 16, 64);
        a1[k] += __shfl_xor(a1[k], 32, 64); a2[k] += __shfl_xor(a2[k], 32, 64);
    }
    if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[wave][0][lane * 4 + k] = a1[k]; red[wave][1][lane * 4 + k] = a2[k]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) fin.cnt[blockIdx.y] = 0;                   // next launch starts from zero tickets
    if (threadIdx.x < BN_CT) {
        const int cc = threadIdx.x, c = blockIdx.y * BN_CT + cc;
        if (c < C) {
            const float t1 = ((red[0][0][cc] + red[1][0][cc]) + red[2][0][cc]) + red[3][0][cc];   // sum dy'
            const float t2 = ((red[0][1][cc] + red[1][1][cc]) + red[2][1][cc]) + red[3][1][cc];   // sum dy' (x - mean)
            const float is = fin.invstd[c], mu_c = mean[c];
            const float g = fin.pdt == DTF_BF16 ? load1<DTF_BF16>(fin.gamma, c) : load1<DTF_F32>(fin.gamma, c);
            const float dg = t2 * is;
            if (fin.pdt == DTF_BF16) { store1<DTF_BF16>(fin.dgamma, c, dg); store1<DTF_BF16>(fin.dbeta, c, t1); }
            else { store1<DTF_F32>(fin.dgamma, c, dg); store1<DTF_F32>(fin.dbeta, c, t1); }
            const float k1 = g * is;
            const float k2 = -k1 * is * dg * fin.inv_count;
            const float k3 = -k1 * t1 * fin.inv_count - k2 * mu_c;
            fin.coef[c] = k1;
            fin.coef[C + c] = k2;
            fin.coef[2 * C + c] = k3;
        }
    }
}

// Forward statistics, flat variant: the tensor is read as ONE contiguous stream, 4 KB (256 lanes x 16 B) per block
// iteration, U iterations in flight, grid-strided -- the access shape that reads a cold tensor at 6.3-6.9 TB/s
// (scripts/ubench/read_only.hip, profiles/r2_straggler_kernels.txt), where the [rows x 64-channel tile] mapping above
// touches one 128-B line per row and tile and stays at 3.3-3.6 TB/s.  Lane mapping as in the flat apply kernels
// below: cv = C / 8 vectors per row, rpi = 256 / cv rows per block iteration, lane tid owns channels 8 * (tid % cv)
// of row tid / cv (C <= 2048; for a dense tensor an iteration is one contiguous piece, a row stride only changes
// the addresses).  The fold over the rpi lanes that share a channel and the partial layout ([block][2][C]) match
// bn_reduce_kernel, so the finalize kernels are shared; summation order is fixed by the code (bitwise reproducible).
template <int U, bool NT>
__global__ __launch_bounds__(256) void bn_stats_flat_kernel(const unsigned short* __restrict__ x,
                                                            float* __restrict__ partial, int64_t M, int C, int cv,
                                                            int rpi, int64_t ldx) {
    const int tid = threadIdx.x;
    const int tr = tid / cv, tc = tid - tr * cv, c0 = tc * 8;
    const bool active = tr < rpi;
    float mu[8], s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; mu[k] = 0.f; }
    if (active) {
        ld8bf(x + c0, mu);                  // shifted accumulation against row 0 (see bn_reduce_kernel)
        const int64_t step = (int64_t)gridDim.x * (U * rpi);
        for (int64_t r = (int64_t)blockIdx.x * (U * rpi) + tr; r < M; r += step) {
            u32x4 a[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t row = r + (int64_t)u * rpi;
                ok[u] = row < M;
                if (ok[u]) {
                    const u32x4* p = reinterpret_cast<const u32x4*>(x + DTF_IDX(row, M, "bn_stats_flat row") * ldx + c0);
                    a[u] = NT ? __builtin_nontemporal_load(p) : *p;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!ok[u]) continue;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d0 = __uint_as_float(a[u][k] << 16) - mu[2 * k];
                    const float d1 = __uint_as_float(a[u][k] & 0xffff0000u) - mu[2 * k + 1];
                    s1[2 * k] += d0;
                    s2[2 * k] = fmaf(d0, d0, s2[2 * k]);
                    s1[2 * k + 1] += d1;
                    s2[2 * k + 1] = fmaf(d1, d1, s2[2 * k + 1]);
                }
            }
        }
    }
    __shared__ float red[2][8 * 257];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        // [k][tid] layout with a 257-float pitch: conflict-free stores AND fold reads (adjacent channels are
        // adjacent k rows: a 256 pitch put all eight on one bank, 78 % conflict cycles in the PMC run)
        red[0][k * 257 + tid] = s1[k];
        red[1][k * 257 + tid] = s2[k];
    }
    __syncthreads();
    // fold the rpi row lanes that share a channel vector, in fixed order
    for (int idx = tid; idx < 2 * C; idx += 256) {
        const int stat = idx >= C, c = idx - stat * C;
        const int k = c & 7, t0 = c >> 3;
        float sum = 0.f;
        for (int g = 0; g < rpi; ++g) sum += red[stat][k * 257 + t0 + g * cv];
        partial[((int64_t)blockIdx.x * 2 + stat) * C + c] = sum;
    }
}

// Finalize: one block per 8 channels.  The kernel is pure latency (a few hundred KB of partials out
// of L2), so the partial rows are spread over 128 row lanes x 2 four-channel groups with 16-B loads:
// at most 8 independent load pairs per lane, all in flight at once (the first version walked 64 rows
// per lane one dependent iteration at a time: 8-9 us per call, 106-188 calls per step).
// Summation order is fixed by the code (lane partials -> xor-shuffle tree -> 4 waves in order), so the
// statistics stay bitwise reproducible.
#define BN_FIN_CH 8
__device__ __forceinline__ void reduce_partials8(const float* __restrict__ partial, int nblk, int C, float& o1,
                                                 float& o2) {
    const int cg = threadIdx.x & 1, rl = threadIdx.x >> 1;
    const int c4 = blockIdx.x * BN_FIN_CH + cg * 4;
    f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
    if (c4 < C) {
#pragma unroll 4
        for (int j = rl; j < nblk; j += 128) {
            a1 += *reinterpret_cast<const f32x4*>(partial + ((int64_t)j * 2 + 0) * C + c4);
            a2 += *reinterpret_cast<const f32x4*>(partial + ((int64_t)j * 2 + 1) * C + c4);
        }
    }
    // fold the 32 row lanes of this wave (lane bits 1..5)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 2; o < 64; o <<= 1) {
            a1[k] += __shfl_xor(a1[k], o, 64);
            a2[k] += __shfl_xor(a2[k], o, 64);
        }
    }
    __shared__ float red[4][2][8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[wave][lane][k] = a1[k]; red[wave][lane][4 + k] = a2[k]; }
    }
    __syncthreads();
    o1 = 0.f; o2 = 0.f;
    if (threadIdx.x < BN_FIN_CH) {
        const int g = threadIdx.x >> 2, k = threadIdx.x & 3;
        o1 = ((red[0][g][k] + red[1][g][k]) + red[2][g][k]) + red[3][g][k];
        o2 = ((red[0][g][4 + k] + red[1][g][4 + k]) + red[2][g][4 + k]) + red[3][g][4 + k];
    }
}

template <int PDT>
__global__ __launch_bounds__(256) void bn_fwd_finalize_kernel(const float* __restrict__ partial, int nblk, int C,
                                                              float inv_count, float unbias, const void* gamma,
                                                              const void* beta, float* running_mean,
                                                              float* running_var, float momentum, float eps,
                                                              float* mean_out, float* invstd_out,
                                                              float* scale_out, float* shift_out,
                                                              const unsigned short* __restrict__ x0) {
    // x0: row 0 of the tensor the partials were accumulated against (bn_reduce_kernel's shift), or null when the
    // partials are plain sums (K6 STATS epilogue)
    const int c = blockIdx.x * BN_FIN_CH + threadIdx.x;
    // the per-channel operands are fetched BEFORE the reduction: after it they would be a second dependent memory round
    // trip in a kernel that is nothing but latency (106 calls per ResNet-50 step)
    const bool mine = threadIdx.x < BN_FIN_CH && c < C;
    float g = 0.f, b = 0.f, rm = 0.f, rv = 0.f, xs = 0.f;
    if (mine) {
        g = load1<PDT>(gamma, c); b = load1<PDT>(beta, c);
        if (running_mean) { rm = running_mean[c]; rv = running_var[c]; }
        if (x0) xs = bf16_to_f32(x0[c]);
    }
    float s1, s2;
    reduce_partials8(partial, nblk, C, s1, s2);
    if (mine) {
        const double md = (double)s1 * inv_count;
        double var = (double)s2 * inv_count - md * md;
        if (var < 0.0) var = 0.0;
        const double m = md + (double)xs;
        const float mean = (float)m;
        const float invstd = rsqrtf((float)var + eps);
        mean_out[c] = mean;
        invstd_out[c] = invstd;
        const float sc = g * invstd;
        scale_out[c] = sc;
        shift_out[c] = fmaf(-mean, sc, b);
        if (running_mean) {
            running_mean[c] = fmaf(momentum, mean - rm, rm);
            running_var[c] = fmaf(momentum, (float)var * unbias - rv, rv);
        }
    }
}

// dgamma, dbeta + the three per-channel coefficients of the elementwise pass:
//   dx = k1*dy' + k2*x + k3   (see below)
template <int PDT>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int C,
                                                              float inv_count, const void* gamma,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, void* dgamma,
                                                              void* dbeta, float* coef /* [3][C] */) {
    const int c = blockIdx.x * BN_FIN_CH + threadIdx.x;
    const bool mine = threadIdx.x < BN_FIN_CH && c < C;
    float is = 0.f, mu = 0.f, g = 0.f;          // fetched before the reduction (see the forward finalize)
    if (mine) { is = invstd[c]; mu = mean[c]; g = load1<PDT>(gamma, c); }
    float s1, s2;   // s1 = sum dy', s2 = sum dy' * (x - mean)
    reduce_partials8(partial, nblk, C, s1, s2);
    if (mine) {
        const float dg = s2 * is;            // sum dy' * xhat
        store1<PDT>(dgamma, c, dg);
        store1<PDT>(dbeta, c, s1);
        // dx = g*is * (dy' - s1/M - xhat * dg/M),  xhat = (x - mu) * is
        const float k1 = g * is;
        const float k2 = -k1 * is * dg * inv_count;
        const float k3 = -k1 * s1 * inv_count - k2 * mu;
        coef[c] = k1;
        coef[C + c] = k2;
        coef[2 * C + c] = k3;
    }
}

// y = [relu]( x*scale[c] + shift[c] [+ res] ).  Same [row chunk][channel tile] mapping as the
// reductions: a lane owns 8 fixed channels, so scale/shift live in registers.
template <bool RELU, bool RES>
__global__ __launch_bounds__(256) void bn_fwd_apply_kernel(const unsigned short* __restrict__ x,
                                                           const unsigned short* __restrict__ res,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           unsigned short* __restrict__ y,
                                                           unsigned char* __restrict__ mask, int64_t M, int C,
                                                           int rows_per_block, int64_t ldy, int64_t ldx) {
    const int tx = threadIdx.x & 7;
    const int ty = threadIdx.x >> 3;
    const int c0 = blockIdx.y * BN_CT + tx * 8;
    if (c0 >= C) return;
    BN_SCHED(BN_UNROLL)
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc[k] = scale[c0 + k]; sh[k] = shift[c0 + k]; }
    unsigned char* mrow = mask + ((int64_t)blockIdx.y * M) * 8 + tx;
    for (int64_t r = r0 + ty; r < r1; r += rstep) {
        float xv[BN_UNROLL][8], rv[BN_UNROLL][8];
        bool ok[BN_UNROLL];
#pragma unroll
        for (int u = 0; u < BN_UNROLL; ++u) {
            const int64_t row = r + (int64_t)u * BN_ROWS;
            ok[u] = row < r1;
            if (ok[u]) {
                ld8bf(x + row * ldx + c0, xv[u]);
                if (RES) ld8bf(res + row * C + c0, rv[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < BN_UNROLL; ++u) {
            if (!ok[u]) continue;
            const int64_t row = r + (int64_t)u * BN_ROWS;
            unsigned m = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float o = fmaf(xv[u][k], sc[k], sh[k]);
                if (RES) o += rv[u][k];
                if (RELU) {
                    m |= (o > 0.f ? 1u : 0u) << k;
                    o = relu_nan(o);
                }
                xv[u][k] = o;
            }
            st8bf(y + row * ldy + c0, xv[u]);
            if (RELU && RES) mrow[row * 8] = (unsigned char)m;
        }
    }
}

// dx = k1*dy' + k2*x + k3 ; dres = dy'   (dy' = dy under the ReLU mask)
template <int MODE, bool RES, bool POOL = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const unsigned short* __restrict__ dy,
                                                           const unsigned short* __restrict__ x,
                                                           const unsigned char* __restrict__ mask,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ coef,
                                                           unsigned short* __restrict__ dx,
                                                           unsigned short* __restrict__ dres, int64_t M, int C,
                                                           int rows_per_block, int64_t lddy, int64_t ldx, int64_t lddx,
                                                           PoolGeom pg = PoolGeom{}) {
    const int tx = threadIdx.x & 7;
    const int ty = threadIdx.x >> 3;
    const int c0 = blockIdx.y * BN_CT + tx * 8;
    if (c0 >= C) return;
    BN_SCHED(BN_UNROLL)
    float k1[8], k2[8], k3[8], sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        k1[k] = coef[c0 + k]; k2[k] = coef[C + c0 + k]; k3[k] = coef[2 * C + c0 + k];
        sc[k] = 0.f; sh[k] = 0.f;
    }
    if (MODE == BN_BWD_RELU_X) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { sc[k] = scale[c0 + k]; sh[k] = shift[c0 + k]; }
    }
    const unsigned char* mrow = mask + ((int64_t)blockIdx.y * M) * 8 + tx;
    for (int64_t r = r0 + ty; r < r1; r += rstep) {
        float xv[BN_UNROLL][8], dv[BN_UNROLL][8];
        unsigned mb[BN_UNROLL];
        bool ok[BN_UNROLL];
#pragma unroll
        for (int u = 0; u < BN_UNROLL; ++u) {
            const int64_t row = r + (int64_t)u * BN_ROWS;
            ok[u] = row < r1;
            if (ok[u]) {
                if (POOL) pool_gather_dy(pg, row, C, c0, dv[u]);
                else ld8bf(dy + row * lddy + c0, dv[u]);
                ld8bf(x + row * ldx + c0, xv[u]);
                if (MODE == BN_BWD_RELU_MASK) mb[u] = mrow[row * 8];
            }
        }
#pragma unroll
        for (int u = 0; u < BN_UNROLL; ++u) {
            if (!ok[u]) continue;
            const int64_t row = r + (int64_t)u * BN_ROWS;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float d = dv[u][k];
                if (MODE == BN_BWD_RELU_X) d = fmaf(xv[u][k], sc[k], sh[k]) > 0.f ? d : 0.f;
                if (MODE == BN_BWD_RELU_MASK) d = ((mb[u] >> k) & 1u) ? d : 0.f;
                dv[u][k] = d;
                xv[u][k] = fmaf(k1[k], d, fmaf(k2[k], xv[u][k], k3[k]));
            }
            st8bf(dx + row * lddx + c0, xv[u]);
            if (RES) st8bf(dres + row * C + c0, dv[u]);
        }
    }
}

// ---------------------------------------------------------------------------
// Flat variants of the elementwise passes (C <= 2048).  One block per U contiguous pieces of ~4 KB, no loop: the
// shape in which a cold x (+ residual) -> y pass measured 6.05-6.37 TB/s against 4.8-5.0 for the [rows x 64-channel
// tile] kernels above (scripts/ubench/stream_variants.hip, profiles/r2_bn_schedule_ab.txt).  The ReLU bit-mask
// layout is unchanged.
// ---------------------------------------------------------------------------
template <bool NT> __device__ __forceinline__ u32x4 ldv(const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void stv(u32x4* p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
__device__ __forceinline__ void cvt8(u32x4 r, float (&v)[8]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[2 * k] = __uint_as_float(r[k] << 16); v[2 * k + 1] = __uint_as_float(r[k] & 0xffff0000u); }
}
__device__ __forceinline__ u32x4 pack8(const float (&v)[8]) {
    u32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = pack_bf16x2(v[2 * k], v[2 * k + 1]);
    return r;
}

// Lane mapping: cv = C / 8 vectors per row, rpi = 256 / cv rows per block iteration; lane tid owns the channels
// 8 * (tid % cv) of row tid / cv (lanes >= rpi * cv idle: none when C is a power of two, < 10 % otherwise).  For a
// dense tensor a block iteration is ONE contiguous piece of rpi * C * 2 bytes; row strides (channel slices of a wider
// NHWC tensor, read or written in place) only change the address arithmetic.
template <bool RELU, bool RES, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void bn_fwd_apply_flat_kernel(const unsigned short* __restrict__ x,
                                                                const unsigned short* __restrict__ res,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                unsigned short* __restrict__ y,
                                                                unsigned char* __restrict__ mask, int64_t M, int C,
                                                                int cv, int rpi, int64_t ldy, int64_t ldx) {
    const int tr = threadIdx.x / cv, c0 = (threadIdx.x - tr * cv) * 8;
    if (tr >= rpi) return;
    const int64_t row0 = (int64_t)blockIdx.x * (U * rpi) + tr;
    u32x4 a[U], r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t row = row0 + u * rpi;
        if (row < M) {
            a[u] = ldv<NTL>(reinterpret_cast<const u32x4*>(x + row * ldx + c0));
            if (RES) r[u] = ldv<NTL>(reinterpret_cast<const u32x4*>(res + row * C + c0));
        }
    }
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc[k] = scale[c0 + k]; sh[k] = shift[c0 + k]; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t row = row0 + u * rpi;
        if (row >= M) continue;
        float f[8], g[8];
        cvt8(a[u], f);
        if (RES) cvt8(r[u], g);
        unsigned m = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float o = fmaf(f[k], sc[k], sh[k]);
            if (RES) o += g[k];
            if (RELU) {
                m |= (o > 0.f ? 1u : 0u) << k;
                o = relu_nan(o);
            }
            f[k] = o;
        }
        stv<NTS>(reinterpret_cast<u32x4*>(y + row * ldy + c0), pack8(f));
        if (RELU && RES)
            mask[DTF_IDX(((int64_t)(c0 >> 6) * M + row) * 8 + ((c0 & 63) >> 3), dtf_mask_bytes_dev(M, C), "bn flat mask")] = (unsigned char)m;
    }
}

template <int MODE, bool RES, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void bn_bwd_apply_flat_kernel(const unsigned short* __restrict__ dy,
                                                                const unsigned short* __restrict__ x,
                                                                const unsigned char* __restrict__ mask,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ coef,
                                                                unsigned short* __restrict__ dx,
                                                                unsigned short* __restrict__ dres, int64_t M, int C,
                                                                int cv, int rpi, int64_t lddy, int64_t ldx, int64_t lddx) {
    const int tr = threadIdx.x / cv, c0 = (threadIdx.x - tr * cv) * 8;
    if (tr >= rpi) return;
    const int64_t row0 = (int64_t)blockIdx.x * (U * rpi) + tr;
    u32x4 a[U], d[U];
    unsigned mb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t row = row0 + u * rpi;
        if (row < M) {
            d[u] = ldv<NTL>(reinterpret_cast<const u32x4*>(dy + row * lddy + c0));
            a[u] = ldv<NTL>(reinterpret_cast<const u32x4*>(x + row * ldx + c0));
            if (MODE == BN_BWD_RELU_MASK) mb[u] = mask[((int64_t)(c0 >> 6) * M + row) * 8 + ((c0 & 63) >> 3)];
        }
    }
    float k1[8], k2[8], k3[8], sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        k1[k] = coef[c0 + k]; k2[k] = coef[C + c0 + k]; k3[k] = coef[2 * C + c0 + k];
        sc[k] = 0.f; sh[k] = 0.f;
    }
    if (MODE == BN_BWD_RELU_X) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { sc[k] = scale[c0 + k]; sh[k] = shift[c0 + k]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t row = row0 + u * rpi;
        if (row >= M) continue;
        float xf[8], df[8];
        cvt8(a[u], xf);
        cvt8(d[u], df);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float g = df[k];
            if (MODE == BN_BWD_RELU_X) g = fmaf(xf[k], sc[k], sh[k]) > 0.f ? g : 0.f;
            if (MODE == BN_BWD_RELU_MASK) g = ((mb[u] >> k) & 1u) ? g : 0.f;
            df[k] = g;
            xf[k] = fmaf(k1[k], g, fmaf(k2[k], xf[k], k3[k]));
        }
        stv<NTS>(reinterpret_cast<u32x4*>(dx + row * lddx + c0), pack8(xf));
        if (RES) stv<NTS>(reinterpret_cast<u32x4*>(dres + row * C + c0), pack8(df));
    }
}

// Backward reduction, flat variant (see bn_stats_flat_kernel): s1 = sum dy', s2 = sum dy' * (x - mean) per channel,
// dy' = dy under the ReLU mask.  Dense tensors, C a power of two; partial layout [block][2][C].
template <int MODE, int U, bool NT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_flat_kernel(const unsigned short* __restrict__ x,
                                                                 const unsigned short* __restrict__ dy,
                                                                 const unsigned char* __restrict__ mask,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 float* __restrict__ partial, int64_t M, int C, int log2c) {
    const int tid = threadIdx.x;
    const int64_t nvec = (M * C) >> 3;
    const int c0 = (tid * 8) & (C - 1);
    float mu[8], sc[8], sh[8], s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; mu[k] = mean[c0 + k]; sc[k] = 0.f; sh[k] = 0.f; }
    if (MODE == BN_BWD_RELU_X) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { sc[k] = scale[c0 + k]; sh[k] = shift[c0 + k]; }
    }
    const u32x4* xv = reinterpret_cast<const u32x4*>(x);
    const u32x4* dv = reinterpret_cast<const u32x4*>(dy);
    const unsigned char* mbase = mask + (int64_t)(c0 >> 6) * M * 8 + ((c0 & 63) >> 3);
    const int64_t step = (int64_t)gridDim.x * (256 * U);
    for (int64_t v = (int64_t)blockIdx.x * (256 * U) + tid; v < nvec; v += step) {
        u32x4 a[U], d[U];
        unsigned mb[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t w = v + u * 256;
            ok[u] = w < nvec;
            if (ok[u]) {
                d[u] = ldv<NT>(dv + w);
                a[u] = ldv<NT>(xv + w);
                if (MODE == BN_BWD_RELU_MASK) mb[u] = mbase[((w << 3) >> log2c) * 8];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            float xf[8], df[8];
            cvt8(a[u], xf);
            cvt8(d[u], df);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float g = df[k];
                if (MODE == BN_BWD_RELU_X) g = fmaf(xf[k], sc[k], sh[k]) > 0.f ? g : 0.f;
                if (MODE == BN_BWD_RELU_MASK) g = ((mb[u] >> k) & 1u) ? g : 0.f;
                s1[k] += g;
                s2[k] = fmaf(g, xf[k] - mu[k], s2[k]);
            }
        }
    }
    __shared__ float red[2][2048];
#pragma unroll
    for (int k = 0; k < 8; ++k) { red[0][k * 256 + tid] = s1[k]; red[1][k * 256 + tid] = s2[k]; }
    __syncthreads();
    const int groups = 2048 / C;
    for (int idx = tid; idx < 2 * C; idx += 256) {
        const int stat = idx >= C, c = idx - stat * C;
        const int k = c & 7, t0 = c >> 3;
        float sum = 0.f;
        for (int g = 0; g < groups; ++g) sum += red[stat][k * 256 + t0 + g * (C >> 3)];
        partial[((int64_t)blockIdx.x * 2 + stat) * C + c] = sum;
    }
}

// ---------------------------------------------------------------------------
// Two BatchNorms feeding one add + ReLU: the first block of every ResNet stage,
//   out = relu( BN_a(conv3 output) + BN_b(shortcut conv output) ).
// As two layers that is: BN_b apply (write sc), BN_a apply with residual (read sc); backward: BN_a writes the
// masked gradient for the shortcut (dres), BN_b's reduce and apply passes read it back.  One node instead moves
// every tensor once per pass: forward apply reads xa, xb and writes out (+ mask); backward reduce reads dy, xa, xb;
// backward apply reads dy, xa, xb and writes dxa, dxb -- 6 activation-sized passes fewer per block.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn2_fwd_apply_kernel(const unsigned short* __restrict__ xa,
                                                            const unsigned short* __restrict__ xb,
                                                            const float* __restrict__ scale_a, const float* __restrict__ shift_a,
                                                            const float* __restrict__ scale_b, const float* __restrict__ shift_b,
                                                            unsigned short* __restrict__ y, unsigned char* __restrict__ mask,
                                                            int64_t M, int C, int rows_per_block) {
    const int tx = threadIdx.x & 7;
    const int ty = threadIdx.x >> 3;
    const int c0 = blockIdx.y * BN_CT + tx * 8;
    if (c0 >= C) return;
    BN_SCHED(BN_UNROLL)
    float sa[8], ha[8], sb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sa[k] = scale_a[c0 + k];
        sb[k] = scale_b[c0 + k];
        ha[k] = shift_a[c0 + k] + shift_b[c0 + k];
    }
    unsigned char* mrow = mask + ((int64_t)blockIdx.y * M) * 8 + tx;
    for (int64_t r = r0 + ty; r < r1; r += rstep) {
        float av[BN_UNROLL][8], bv[BN_UNROLL][8];
        bool ok[BN_UNROLL];
#pragma unroll
        for (int u = 0; u < BN_UNROLL; ++u) {
            const int64_t row = r + (int64_t)u * BN_ROWS;
            ok[u] = row < r1;
            if (ok[u]) {
                ld8bf(xa + row * C + c0, av[u]);
                ld8bf(xb + row * C + c0, bv[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < BN_UNROLL; ++u) {
            if (!ok[u]) continue;
            const int64_t row = r + (int64_t)u * BN_ROWS;
            unsigned m = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float o = fmaf(av[u][k], sa[k], fmaf(bv[u][k], sb[k], ha[k]));
                m |= (o > 0.f ? 1u : 0u) << k;
                av[u][k] = relu_nan(o);
            }
            st8bf(y + row * C + c0, av[u]);
            mrow[row * 8] = (unsigned char)m;
        }
    }
}

// partial_a / partial_b [(blockIdx.x * 2 + stat) * C + c]: stat 0 = sum dy', stat 1 = sum dy' * (x - mean)
__global__ __launch_bounds__(256) void bn2_bwd_reduce_kernel(const unsigned short* __restrict__ dy,
                                                             const unsigned short* __restrict__ xa,
                                                             const unsigned short* __restrict__ xb,
                                                             const unsigned char* __restrict__ mask,
                                                             const float* __restrict__ mean_a, const float* __restrict__ mean_b,
                                                             float* __restrict__ partial_a, float* __restrict__ partial_b,
                                                             int64_t M, int C, int rows_per_block) {
    const int tx = threadIdx.x & 7;
    const int ty = threadIdx.x >> 3;
    const int c0 = blockIdx.y * BN_CT + tx * 8;
    const bool cvalid = c0 < C;
    const int bx = blockIdx.x;
    BN_SCHED(BN_UNROLL)
    float s1[8], s2a[8], s2b[8], ma[8], mb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2a[k] = 0.f; s2b[k] = 0.f; ma[k] = 0.f; mb[k] = 0.f; }
    if (cvalid) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { ma[k] = mean_a[c0 + k]; mb[k] = mean_b[c0 + k]; }
        const unsigned char* mrow = mask + ((int64_t)blockIdx.y * M) * 8 + tx;
        for (int64_t r = r0 + ty; r < r1; r += rstep) {
            float av[BN_UNROLL][8], bv[BN_UNROLL][8], dv[BN_UNROLL][8];
            unsigned mk[BN_UNROLL];
            bool ok[BN_UNROLL];
#pragma unroll
            for (int u = 0; u < BN_UNROLL; ++u) {
                const int64_t row = r + (int64_t)u * BN_ROWS;
                ok[u] = row < r1;
                if (ok[u]) {
                    ld8bf(dy + row * C + c0, dv[u]);
                    ld8bf(xa + row * C + c0, av[u]);
                    ld8bf(xb + row * C + c0, bv[u]);
                    mk[u] = mrow[row * 8];
                }
            }
#pragma unroll
            for (int u = 0; u < BN_UNROLL; ++u) {
                if (!ok[u]) continue;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float d = ((mk[u] >> k) & 1u) ? dv[u][k] : 0.f;
                    s1[k] += d;
                    s2a[k] = fmaf(d, av[u][k] - ma[k], s2a[k]);
                    s2b[k] = fmaf(d, bv[u][k] - mb[k], s2b[k]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
            s1[k] += __shfl_xor(s1[k], o, 64);
            s2a[k] += __shfl_xor(s2a[k], o, 64);
            s2b[k] += __shfl_xor(s2b[k], o, 64);
        }
    }
    __shared__ float red[4][3][BN_CT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            red[wave][0][lane * 8 + k] = s1[k];
            red[wave][1][lane * 8 + k] = s2a[k];
            red[wave][2][lane * 8 + k] = s2b[k];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3 * BN_CT) {
        const int stat = threadIdx.x / BN_CT, cc = threadIdx.x % BN_CT;
        const int c = blockIdx.y * BN_CT + cc;
        if (c < C) {
            const float v = red[0][stat][cc] + red[1][stat][cc] + red[2][stat][cc] + red[3][stat][cc];
            if (stat == 0) {
                partial_a[((int64_t)bx * 2 + 0) * C + c] = v;
                partial_b[((int64_t)bx * 2 + 0) * C + c] = v;
            } else if (stat == 1) {
                partial_a[((int64_t)bx * 2 + 1) * C + c] = v;
            } else {
                partial_b[((int64_t)bx * 2 + 1) * C + c] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void bn2_bwd_apply_kernel(const unsigned short* __restrict__ dy,
                                                            const unsigned short* __restrict__ xa,
                                                            const unsigned short* __restrict__ xb,
                                                            const unsigned char* __restrict__ mask,
                                                            const float* __restrict__ coef_a, const float* __restrict__ coef_b,
                                                            unsigned short* __restrict__ dxa, unsigned short* __restrict__ dxb,
                                                            int64_t M, int C, int rows_per_block) {
    const int tx = threadIdx.x & 7;
    const int ty = threadIdx.x >> 3;
    const int c0 = blockIdx.y * BN_CT + tx * 8;
    if (c0 >= C) return;
    BN_SCHED(2)
    float a1[8], a2[8], a3[8], b1[8], b2[8], b3[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a1[k] = coef_a[c0 + k]; a2[k] = coef_a[C + c0 + k]; a3[k] = coef_a[2 * C + c0 + k];
        b1[k] = coef_b[c0 + k]; b2[k] = coef_b[C + c0 + k]; b3[k] = coef_b[2 * C + c0 + k];
    }
    const unsigned char* mrow = mask + ((int64_t)blockIdx.y * M) * 8 + tx;
    for (int64_t r = r0 + ty; r < r1; r += rstep) {
        float av[2][8], bv[2][8], dv[2][8];
        unsigned mk[2];
        bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t row = r + (int64_t)u * BN_ROWS;
            ok[u] = row < r1;
            if (ok[u]) {
                ld8bf(dy + row * C + c0, dv[u]);
                ld8bf(xa + row * C + c0, av[u]);
                ld8bf(xb + row * C + c0, bv[u]);
                mk[u] = mrow[row * 8];
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!ok[u]) continue;
            const int64_t row = r + (int64_t)u * BN_ROWS;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float d = ((mk[u] >> k) & 1u) ? dv[u][k] : 0.f;
                av[u][k] = fmaf(a1[k], d, fmaf(a2[k], av[u][k], a3[k]));
                bv[u][k] = fmaf(b1[k], d, fmaf(b2[k], bv[u][k], b3[k]));
            }
            st8bf(dxa + row * C + c0, av[u]);
            st8bf(dxb + row * C + c0, bv[u]);
        }
    }
}

// flat variants of the dual-BatchNorm elementwise passes (same lane mapping as bn_fwd_apply_flat_kernel; all
// operands of this node are dense)
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void bn2_fwd_apply_flat_kernel(const unsigned short* __restrict__ xa,
                                                                 const unsigned short* __restrict__ xb,
                                                                 const float* __restrict__ scale_a, const float* __restrict__ shift_a,
                                                                 const float* __restrict__ scale_b, const float* __restrict__ shift_b,
                                                                 unsigned short* __restrict__ y, unsigned char* __restrict__ mask,
                                                                 int64_t M, int C, int cv, int rpi) {
    const int tr = threadIdx.x / cv, c0 = (threadIdx.x - tr * cv) * 8;
    if (tr >= rpi) return;
    const int64_t row0 = (int64_t)blockIdx.x * (U * rpi) + tr;
    u32x4 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t row = row0 + u * rpi;
        if (row < M) {
            a[u] = ldv<NTL>(reinterpret_cast<const u32x4*>(xa + row * C + c0));
            b[u] = ldv<NTL>(reinterpret_cast<const u32x4*>(xb + row * C + c0));
        }
    }
    float sa[8], ha[8], sb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sa[k] = scale_a[c0 + k];
        sb[k] = scale_b[c0 + k];
        ha[k] = shift_a[c0 + k] + shift_b[c0 + k];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t row = row0 + u * rpi;
        if (row >= M) continue;
        float av[8], bv[8];
        cvt8(a[u], av);
        cvt8(b[u], bv);
        unsigned m = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float o = fmaf(av[k], sa[k], fmaf(bv[k], sb[k], ha[k]));
            m |= (o > 0.f ? 1u : 0u) << k;
            av[k] = relu_nan(o);
        }
        stv<NTS>(reinterpret_cast<u32x4*>(y + row * C + c0), pack8(av));
        mask[DTF_IDX(((int64_t)(c0 >> 6) * M + row) * 8 + ((c0 & 63) >> 3), dtf_mask_bytes_dev(M, C), "bn2 flat mask")] = (unsigned char)m;
    }
}

template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void bn2_bwd_apply_flat_kernel(const unsigned short* __restrict__ dy,
                                                                 const unsigned short* __restrict__ xa,
                                                                 const unsigned short* __restrict__ xb,
                                                                 const unsigned char* __restrict__ mask,
                                                                 const float* __restrict__ coef_a, const float* __restrict__ coef_b,
                                                                 unsigned short* __restrict__ dxa, unsigned short* __restrict__ dxb,
                                                                 int64_t M, int C, int cv, int rpi) {
    const int tr = threadIdx.x / cv, c0 = (threadIdx.x - tr * cv) * 8;
    if (tr >= rpi) return;
    const int64_t row0 = (int64_t)blockIdx.x * (U * rpi) + tr;
    u32x4 a[U], b[U], d[U];
    unsigned mk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t row = row0 + u * rpi;
        if (row < M) {
            d[u] = ldv<NTL>(reinterpret_cast<const u32x4*>(dy + row * C + c0));
            a[u] = ldv<NTL>(reinterpret_cast<const u32x4*>(xa + row * C + c0));
            b[u] = ldv<NTL>(reinterpret_cast<const u32x4*>(xb + row * C + c0));
            mk[u] = mask[((int64_t)(c0 >> 6) * M + row) * 8 + ((c0 & 63) >> 3)];
        }
    }
    float a1[8], a2[8], a3[8], b1[8], b2[8], b3[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a1[k] = coef_a[c0 + k]; a2[k] = coef_a[C + c0 + k]; a3[k] = coef_a[2 * C + c0 + k];
        b1[k] = coef_b[c0 + k]; b2[k] = coef_b[C + c0 + k]; b3[k] = coef_b[2 * C + c0 + k];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t row = row0 + u * rpi;
        if (row >= M) continue;
        float av[8], bv[8], dv[8];
        cvt8(a[u], av);
        cvt8(b[u], bv);
        cvt8(d[u], dv);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float g = ((mk[u] >> k) & 1u) ? dv[k] : 0.f;
            av[k] = fmaf(a1[k], g, fmaf(a2[k], av[k], a3[k]));
            bv[k] = fmaf(b1[k], g, fmaf(b2[k], bv[k], b3[k]));
        }
        stv<NTS>(reinterpret_cast<u32x4*>(dxa + row * C + c0), pack8(av));
        stv<NTS>(reinterpret_cast<u32x4*>(dxb + row * C + c0), pack8(bv));
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static int bn_sweep = -1;          // DTF_BN_SWEEP: 1 (default) grid-strided sweep, 0 private contiguous ranges
static inline void bn_grid(int64_t M, int C, int* nblk_x, int* rows_per_block, int* ctiles) {
    const int ct = (C + BN_CT - 1) / BN_CT;
    static int target = 0;
    if (target == 0) {
        const char* e = getenv("DTF_BN_BLOCKS");
        // 3 blocks per CU in total.  Measured through the whole ResNet-50 / Inception-v3 step (DTF_BN_BLOCKS A/B):
        // 512: 10 135 / 8 102 images/s, 640: 10 209, 768: 10 391 / 8 228, 896: 10 178, 1024: 10 258 / 8 159,
        // 1536: 10 221, 2048: 10 271 -- whole multiples of the 256 CUs, and few enough partial rows for the finalize
        target = e ? atoi(e) : 768;
        if (target < 64) target = 768;
    }
    int64_t want = target / ct;
    if (want < 1) want = 1;
    const int64_t quantum = BN_ROWS * BN_UNROLL;
    int64_t max_blocks = (M + quantum - 1) / quantum;
    if (want > max_blocks) want = max_blocks;
    int64_t rpb = (M + want - 1) / want;
    rpb = (rpb + BN_ROWS - 1) / BN_ROWS * BN_ROWS;
    *nblk_x = (int)((M + rpb - 1) / rpb);
    // the sweep pays on the big few-channel tensors (stem, stages 1-2); with >= 16 channel tiles a block row is a
    // handful of blocks and private ranges measured 5-10 % faster
    *rows_per_block = (bn_sweep == 0 || ct > 8) ? (int)rpb : 0;
    *ctiles = ct;
}

// Forward statistics pass: flat stream kernel where the layout allows it (see bn_stats_flat_kernel), the
// [rows x channel tile] kernel otherwise.  Returns the number of partial rows written.
static int bn_flat_stats = -1;     // DTF_BN_FLAT_STATS: 0 off, 1 (default) on, 2 on with non-temporal loads (better alone, worse
                                   // once the apply passes are flat too: the apply re-reads x out of the Infinity Cache)
static inline int bn_flat_grid(int64_t M, int C) {
    if (C < 8 || C > 2048 || (C & 7)) return 0;
    int64_t g = 262144 / C;                 // partial rows: <= 2 MB of fp32 partials
    if (g > 1024) g = 1024;
    if (g < 256) g = 256;
    const int64_t rows = 8 * (256 / (C / 8));       // rows per block iteration x U
    const int64_t most = (M + rows - 1) / rows;
    if (g > most) g = most;
    return (int)g;
}
static int launch_fwd_stats(const unsigned short* x, float* partial, int64_t M, int C, int64_t ldx, hipStream_t stream) {
    if (bn_flat_stats < 0) {
        const char* e = getenv("DTF_BN_FLAT_STATS");
        bn_flat_stats = e ? atoi(e) : 1;
    }
    const int g = bn_flat_stats ? bn_flat_grid(M, C) : 0;
    if (g > 0) {
        const int cv = C / 8, rpi = 256 / cv;
        if (bn_flat_stats == 2)
            hipLaunchKernelGGL((bn_stats_flat_kernel<8, true>), dim3(g), dim3(256), 0, stream, x, partial, M, C, cv, rpi, ldx);
        else
            hipLaunchKernelGGL((bn_stats_flat_kernel<8, false>), dim3(g), dim3(256), 0, stream, x, partial, M, C, cv, rpi, ldx);
        return g;
    }
    int nb, rpb, ct;
    bn_grid(M, C, &nb, &rpb, &ct);
    hipLaunchKernelGGL((bn_reduce_kernel<BN_FWD_STATS, 8>), dim3(nb, ct), dim3(256), 0, stream, x, nullptr, nullptr, nullptr,
                       nullptr, nullptr, partial, M, C, rpb, (int64_t)C, ldx);
    return nb;
}

// DTF_BN_FUSED_FINALIZE: 1 the backward reduce kernels finalize their own channel tile (BnFin), 0 (default) the
// stand-alone finalize kernel.  Measured through the ResNet-50 step (scripts/gpu_r3_m.sh, A/B/A/B on one box):
// fused 20.32 / 20.33 ms, stand-alone 20.18 / 20.15 ms -- the reducer's serial fold of rows that the write-through
// stores dropped from L2 (62-70 GB/s per block, 98-393 KB per tile) plus its acquire cost as much as the launch they
// replace.  Kept as a switch with its tests; the form with a __threadfence() per block measured 24.5 ms.
#define BN_MAX_TICKETS 64
static int bn_fused_on = -1;
static inline bool bn_fused_finalize(const unsigned* tickets, int ctiles) {
    if (bn_fused_on < 0) { const char* e = getenv("DTF_BN_FUSED_FINALIZE"); bn_fused_on = e ? (atoi(e) != 0) : 0; }
    return bn_fused_on && tickets != nullptr && ctiles <= BN_MAX_TICKETS;
}

static void bn_init_order() {
    if (bn_sweep < 0) {
        const char* e = getenv("DTF_BN_SWEEP");
        bn_sweep = e ? (atoi(e) != 0) : 1;
    }
}

extern "C" {

// run-time form of DTF_BN_FUSED_FINALIZE (tests); returns the previous setting (-1: environment not read yet)
int dtf_bn_set_fused_finalize(int on) { const int was = bn_fused_on; bn_fused_on = on ? 1 : 0; return was; }

// A/B switch of the row-chunk schedule (see BN_SCHED): 1 sweep, 0 private ranges
void dtf_bn_set_sweep(int on) { bn_sweep = on ? 1 : 0; }

// workspace query: floats needed for the partial sums of an [M, C] reduction
int64_t dtf_bn_partial_floats(int64_t M, int C) {
    int nb, rpb, ct;
    bn_grid(M, C, &nb, &rpb, &ct);
    const int flat = bn_flat_grid(M, C);
    return (int64_t)(nb > flat ? nb : flat) * 2 * C;
}

void dtf_bn_set_flat_stats(int mode) { bn_flat_stats = mode; }

// bytes of the ReLU bit mask of an [M, C] tensor (BN + residual + ReLU only)
int64_t dtf_bn_mask_bytes(int64_t M, int C) { return (int64_t)((C + BN_CT - 1) / BN_CT) * M * 8; }

// DTF_BN_FLAT_APPLY selects the flat elementwise kernels for dense power-of-two-channel tensors:
//   0 off, 1 U=1, 2 U=2, 5 U=4 (plain loads/stores), 3 U=4 non-temporal loads+stores, 4 U=4 nt stores, 6 U=4 nt loads,
//   7 U=2 nt loads, 8 U=8 nt loads, 9 U=2 nt loads+stores
static int bn_flat_apply = -1;
static inline int bn_flat_apply_mode(int64_t M, int C) {
    if (bn_flat_apply < 0) {
        const char* e = getenv("DTF_BN_FLAT_APPLY");
        bn_flat_apply = e ? atoi(e) : 9;
    }
    if (!bn_flat_apply || C < 8 || C > 2048) return 0;
    return bn_flat_apply;
}
static inline int ilog2(int c) { int l = 0; while ((1 << l) < c) ++l; return l; }
#define BN_FLAT_DISPATCH(MODE_, LAUNCH)                      \
    switch (MODE_) {                                         \
        case 1: LAUNCH(1, false, false); break;              \
        case 2: LAUNCH(2, false, false); break;              \
        case 3: LAUNCH(4, true, true); break;                \
        case 4: LAUNCH(4, false, true); break;               \
        case 6: LAUNCH(4, true, false); break;               \
        case 7: LAUNCH(2, true, false); break;               \
        case 8: LAUNCH(8, true, false); break;               \
        case 9: LAUNCH(2, true, true); break;                \
        default: LAUNCH(4, false, false); break;             \
    }
static inline unsigned bn_flat_blocks(int64_t M, int C, int mode) {
    const int U = mode == 1 ? 1 : ((mode == 2 || mode == 7 || mode == 9) ? 2 : (mode == 8 ? 8 : 4));
    const int64_t rows = (int64_t)U * (256 / (C / 8));
    return (unsigned)((M + rows - 1) / rows);
}

static void launch_fwd_apply(const void* x, const void* res, void* y, void* mask, const float* scale,
                             const float* shift, int64_t M, int C, int relu, int64_t ldy, int64_t ldx, hipStream_t stream) {
    int nb, rpb, ct;
    bn_grid(M, C, &nb, &rpb, &ct);
    const unsigned short *xs = (const unsigned short*)x, *rs = (const unsigned short*)res;
    unsigned short* ys = (unsigned short*)y;
    unsigned char* ms = (unsigned char*)mask;
    const int flat = bn_flat_apply_mode(M, C);
    if (flat) {
        const unsigned fb = bn_flat_blocks(M, C, flat);
        const int cv = C / 8, rpi = 256 / cv;
#define FLATF(R, S, U_, NL, NS) hipLaunchKernelGGL((bn_fwd_apply_flat_kernel<R, S, U_, NL, NS>), dim3(fb), dim3(256), 0, stream, xs, rs, scale, shift, ys, ms, M, C, cv, rpi, ldy, ldx)
#define FL_TT(U_, NL, NS) FLATF(true, true, U_, NL, NS)
#define FL_TF(U_, NL, NS) FLATF(true, false, U_, NL, NS)
#define FL_FT(U_, NL, NS) FLATF(false, true, U_, NL, NS)
#define FL_FF(U_, NL, NS) FLATF(false, false, U_, NL, NS)
        if (relu) { if (res) { BN_FLAT_DISPATCH(flat, FL_TT) } else { BN_FLAT_DISPATCH(flat, FL_TF) } }
        else { if (res) { BN_FLAT_DISPATCH(flat, FL_FT) } else { BN_FLAT_DISPATCH(flat, FL_FF) } }
#undef FL_TT
#undef FL_TF
#undef FL_FT
#undef FL_FF
#undef FLATF
        return;
    }
#define FWD(R, S) hipLaunchKernelGGL((bn_fwd_apply_kernel<R, S>), dim3(nb, ct), dim3(256), 0, stream, xs, rs, scale, shift, ys, ms, M, C, rpb, ldy, ldx)
    if (relu) { if (res) FWD(true, true); else FWD(true, false); }
    else { if (res) FWD(false, true); else FWD(false, false); }
#undef FWD
}

// Training forward.  x,y,res: bf16 [M,C]; gamma/beta: param dtype; running_*: fp32 or null;
// stats outputs mean/invstd/scale/shift: fp32 [C]; partial: workspace; mask: dtf_bn_mask_bytes
// (only written when relu && res).
int dtf_bn_fwd_train(const void* x, const void* res, void* y, void* mask, const void* gamma, const void* beta,
                     int param_dtype, float* running_mean, float* running_var, float momentum, float eps,
                     float* mean, float* invstd, float* scale, float* shift, float* partial, int64_t M, int C,
                     int relu, int64_t ldy, int64_t ldx, hipStream_t stream) {
    // ldx: row stride (elements) of x -- C for a dense tensor, the total channel count when x is a channel slice
    // of a wider NHWC tensor (a branch of a fused multi-head convolution output read in place)
    // ldy: row stride (elements) of y -- C for a dense tensor, the total channel count when y is a
    // channel slice of a wider NHWC tensor (branch outputs written straight into a concat buffer)
    DTF_CLEAR_ERRORS();
    bn_init_order();
    if (C % 8 != 0 || M <= 0) return -20;
    if (ldy == 0) ldy = C;
    if (ldx == 0) ldx = C;
    if (ldy < C || (ldy & 7) || ldx < C || (ldx & 7)) return -24;
    if (relu && res && !mask) return -22;
    const int nb = launch_fwd_stats((const unsigned short*)x, partial, M, C, ldx, stream);
    const float inv_count = 1.0f / (float)M;
    const float unbias = M > 1 ? (float)M / (float)(M - 1) : 1.f;
    const int fgrid = (C + BN_FIN_CH - 1) / BN_FIN_CH;
    if (param_dtype == DTF_BF16)
        hipLaunchKernelGGL((bn_fwd_finalize_kernel<DTF_BF16>), dim3(fgrid), dim3(256), 0, stream, partial, nb, C,
                           inv_count, unbias, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd,
                           scale, shift, (const unsigned short*)x);
    else
        hipLaunchKernelGGL((bn_fwd_finalize_kernel<DTF_F32>), dim3(fgrid), dim3(256), 0, stream, partial, nb, C,
                           inv_count, unbias, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd,
                           scale, shift, (const unsigned short*)x);
    launch_fwd_apply(x, res, y, mask, scale, shift, M, C, relu, ldy, ldx, stream);
    DTF_CHECK_LAUNCH();
    return 0;
}

// Statistics + finalize only (the apply pass is fused into the consumer: ops/pool.py bn_relu_max_pool2d)
int dtf_bn_stats_train(const void* x, const void* gamma, const void* beta, int param_dtype, float* running_mean,
                       float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                       float* partial, int64_t M, int C, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    bn_init_order();
    if (C % 8 != 0 || M <= 0) return -20;
    const int nb = launch_fwd_stats((const unsigned short*)x, partial, M, C, (int64_t)C, stream);
    const float inv_count = 1.0f / (float)M;
    const float unbias = M > 1 ? (float)M / (float)(M - 1) : 1.f;
    const int fgrid = (C + BN_FIN_CH - 1) / BN_FIN_CH;
    if (param_dtype == DTF_BF16)
        hipLaunchKernelGGL((bn_fwd_finalize_kernel<DTF_BF16>), dim3(fgrid), dim3(256), 0, stream, partial, nb, C, inv_count,
                           unbias, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift,
                           (const unsigned short*)x);
    else
        hipLaunchKernelGGL((bn_fwd_finalize_kernel<DTF_F32>), dim3(fgrid), dim3(256), 0, stream, partial, nb, C, inv_count,
                           unbias, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift,
                           (const unsigned short*)x);
    DTF_CHECK_LAUNCH();
    return 0;
}

// Training forward with the statistics partials already computed by the producer of x (K6 STATS epilogue):
// partial = fp32 [nblk][2][C] column sums / sums of squares; finalize + apply only.
int dtf_bn_fwd_train_pre(const void* x, const void* res, void* y, void* mask, const void* gamma, const void* beta,
                         int param_dtype, float* running_mean, float* running_var, float momentum, float eps,
                         float* mean, float* invstd, float* scale, float* shift, const float* partial, int nblk,
                         int64_t M, int C, int relu, int64_t ldy, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    bn_init_order();
    if (C % 8 != 0 || M <= 0 || nblk <= 0) return -20;
    if (ldy == 0) ldy = C;
    if (ldy < C || (ldy & 7)) return -24;
    if (relu && res && !mask) return -22;
    const float inv_count = 1.0f / (float)M;
    const float unbias = M > 1 ? (float)M / (float)(M - 1) : 1.f;
    const int fgrid = (C + BN_FIN_CH - 1) / BN_FIN_CH;
    if (param_dtype == DTF_BF16)
        hipLaunchKernelGGL((bn_fwd_finalize_kernel<DTF_BF16>), dim3(fgrid), dim3(256), 0, stream, partial, nblk, C,
                           inv_count, unbias, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd,
                           scale, shift, (const unsigned short*)nullptr);
    else
        hipLaunchKernelGGL((bn_fwd_finalize_kernel<DTF_F32>), dim3(fgrid), dim3(256), 0, stream, partial, nblk, C,
                           inv_count, unbias, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd,
                           scale, shift, (const unsigned short*)nullptr);
    launch_fwd_apply(x, res, y, mask, scale, shift, M, C, relu, ldy, (int64_t)C, stream);
    DTF_CHECK_LAUNCH();
    return 0;
}

// Finalize only, from statistics partials of the producer (plain sums: the K15 stem forward, whose consumer applies the
// normalisation itself -- ops/stem.py).
int dtf_bn_finalize_pre(const void* gamma, const void* beta, int param_dtype, float* running_mean, float* running_var,
                        float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                        const float* partial, int nblk, int64_t M, int C, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (C % 8 != 0 || M <= 0 || nblk <= 0) return -20;
    const float inv_count = 1.0f / (float)M;
    const float unbias = M > 1 ? (float)M / (float)(M - 1) : 1.f;
    const int fgrid = (C + BN_FIN_CH - 1) / BN_FIN_CH;
    if (param_dtype == DTF_BF16)
        hipLaunchKernelGGL((bn_fwd_finalize_kernel<DTF_BF16>), dim3(fgrid), dim3(256), 0, stream, partial, nblk, C, inv_count,
                           unbias, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift,
                           (const unsigned short*)nullptr);
    else
        hipLaunchKernelGGL((bn_fwd_finalize_kernel<DTF_F32>), dim3(fgrid), dim3(256), 0, stream, partial, nblk, C, inv_count,
                           unbias, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift,
                           (const unsigned short*)nullptr);
    DTF_CHECK_LAUNCH();
    return 0;
}

// Inference / frozen-statistics forward: scale/shift precomputed by the caller.
int dtf_bn_fwd_apply(const void* x, const void* res, void* y, void* mask, const float* scale, const float* shift,
                     int64_t M, int C, int relu, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    bn_init_order();
    if (C % 8 != 0 || M <= 0) return -20;
    if (relu && res && !mask) return -22;
    launch_fwd_apply(x, res, y, mask, scale, shift, M, C, relu, (int64_t)C, (int64_t)C, stream);
    DTF_CHECK_LAUNCH();
    return 0;
}

// Training backward.  relu_mode: 0 none, 1 ReLU without residual (mask recomputed from x,
// scale, shift), 3 ReLU after residual add (bit mask from the forward).  dres may be null.
// coef: fp32 [3*C] workspace.
int dtf_bn_bwd_train(const void* dy, const void* x, const void* mask, const void* gamma, int param_dtype,
                     const float* mean, const float* invstd, const float* scale, const float* shift, void* dx,
                     void* dres, void* dgamma, void* dbeta, float* coef, float* partial, int64_t M, int C,
                     int relu_mode, int64_t lddy, int64_t ldx, int64_t lddx, unsigned* tickets, hipStream_t stream) {
    // tickets: BN_MAX_TICKETS zeroed uint32 owned by the calling stream (fused finalize, see BnFin), or null
    // lddx: row stride of dx (written into a channel slice of a wider gradient tensor when > C)
    // lddy: row stride (elements) of dy (a channel slice of the concat gradient is read in place)
    DTF_CLEAR_ERRORS();
    bn_init_order();
    if (C % 8 != 0 || M <= 0) return -20;
    if (lddy == 0) lddy = C;
    if (ldx == 0) ldx = C;
    if (lddx == 0) lddx = C;
    if (lddy < C || (lddy & 7) || ldx < C || (ldx & 7) || lddx < C || (lddx & 7)) return -24;
    if (relu_mode == BN_BWD_RELU_MASK && !mask) return -22;
    int nb, rpb, ct;
    bn_grid(M, C, &nb, &rpb, &ct);
    const unsigned short *dys = (const unsigned short*)dy, *xs = (const unsigned short*)x;
    const unsigned char* ms = (const unsigned char*)mask;
    int nred = nb;
    bool fused = false;
    const float inv_count = 1.0f / (float)M;
    // DTF_BN_FLAT_REDUCE: 0 off, 1 flat U=4, 2 flat U=4 non-temporal loads, 3 flat U=2, 4 flat U=2 non-temporal loads
    static int flat_reduce = -1;
    if (flat_reduce < 0) { const char* e = getenv("DTF_BN_FLAT_REDUCE"); flat_reduce = e ? atoi(e) : 0; }
    const int fg = (flat_reduce && lddy == C && ldx == C && !(C & (C - 1))) ? bn_flat_grid(M, C) : 0;
    if (fg > 0) {
        const int l2 = ilog2(C);
        nred = fg;
#define FRED(MODE, U_, NT_) hipLaunchKernelGGL((bn_bwd_reduce_flat_kernel<MODE, U_, NT_>), dim3(fg), dim3(256), 0, stream, xs, dys, ms, mean, scale, shift, partial, M, C, l2)
#define FRED_M(MODE) do { if (flat_reduce == 1) FRED(MODE, 4, false); else if (flat_reduce == 2) FRED(MODE, 4, true); else if (flat_reduce == 3) FRED(MODE, 2, false); else FRED(MODE, 2, true); } while (0)
        if (relu_mode == BN_BWD_RELU_X) FRED_M(BN_BWD_RELU_X);
        else if (relu_mode == BN_BWD_RELU_MASK) FRED_M(BN_BWD_RELU_MASK);
        else FRED_M(BN_BWD_PLAIN);
#undef FRED_M
#undef FRED
    } else {
        fused = bn_fused_finalize(tickets, ct);
        BnFin fin{};
        if (fused) fin = BnFin{tickets, gamma, invstd, dgamma, dbeta, coef, inv_count, param_dtype};
#define RED(MODE) hipLaunchKernelGGL((bn_reduce_kernel<MODE, BN_UNROLL>), dim3(nb, ct), dim3(256), 0, stream, xs, dys, ms, mean, scale, shift, partial, M, C, rpb, lddy, ldx, PoolGeom{}, fin)
    if (relu_mode == BN_BWD_RELU_X) RED(BN_BWD_RELU_X);
    else if (relu_mode == BN_BWD_RELU_MASK) RED(BN_BWD_RELU_MASK);
    else RED(BN_BWD_PLAIN);
#undef RED
    }
    const int fgrid = (C + BN_FIN_CH - 1) / BN_FIN_CH;
    if (!fused && param_dtype == DTF_BF16)
        hipLaunchKernelGGL((bn_bwd_finalize_kernel<DTF_BF16>), dim3(fgrid), dim3(256), 0, stream, partial, nred, C,
                           inv_count, gamma, mean, invstd, dgamma, dbeta, coef);
    else if (!fused)
        hipLaunchKernelGGL((bn_bwd_finalize_kernel<DTF_F32>), dim3(fgrid), dim3(256), 0, stream, partial, nred, C,
                           inv_count, gamma, mean, invstd, dgamma, dbeta, coef);
    if (!dx) {          // reduce + finalize only: the consumer applies dx = k1 dy' + k2 x + k3 itself (K15 stem weight gradient)
        DTF_CHECK_LAUNCH();
        return 0;
    }
    unsigned short *dxs = (unsigned short*)dx, *drs = (unsigned short*)dres;
    const int flat = bn_flat_apply_mode(M, C);
    if (flat) {
        const unsigned fb = bn_flat_blocks(M, C, flat);
        const int cv = C / 8, rpi = 256 / cv;
#define FLATB(MODE, S, U_, NL, NS) hipLaunchKernelGGL((bn_bwd_apply_flat_kernel<MODE, S, U_, NL, NS>), dim3(fb), dim3(256), 0, stream, dys, xs, ms, scale, shift, coef, dxs, drs, M, C, cv, rpi, lddy, ldx, lddx)
#define FB_XT(U_, NL, NS) FLATB(BN_BWD_RELU_X, true, U_, NL, NS)
#define FB_XF(U_, NL, NS) FLATB(BN_BWD_RELU_X, false, U_, NL, NS)
#define FB_MT(U_, NL, NS) FLATB(BN_BWD_RELU_MASK, true, U_, NL, NS)
#define FB_MF(U_, NL, NS) FLATB(BN_BWD_RELU_MASK, false, U_, NL, NS)
#define FB_PT(U_, NL, NS) FLATB(BN_BWD_PLAIN, true, U_, NL, NS)
#define FB_PF(U_, NL, NS) FLATB(BN_BWD_PLAIN, false, U_, NL, NS)
        if (relu_mode == BN_BWD_RELU_X) { if (dres) { BN_FLAT_DISPATCH(flat, FB_XT) } else { BN_FLAT_DISPATCH(flat, FB_XF) } }
        else if (relu_mode == BN_BWD_RELU_MASK) { if (dres) { BN_FLAT_DISPATCH(flat, FB_MT) } else { BN_FLAT_DISPATCH(flat, FB_MF) } }
        else { if (dres) { BN_FLAT_DISPATCH(flat, FB_PT) } else { BN_FLAT_DISPATCH(flat, FB_PF) } }
#undef FB_XT
#undef FB_XF
#undef FB_MT
#undef FB_MF
#undef FB_PT
#undef FB_PF
#undef FLATB
        DTF_CHECK_LAUNCH();
        return 0;
    }
#define BWD(MODE, S) hipLaunchKernelGGL((bn_bwd_apply_kernel<MODE, S>), dim3(nb, ct), dim3(256), 0, stream, dys, xs, ms, scale, shift, coef, dxs, drs, M, C, rpb, lddy, ldx, lddx)
    if (relu_mode == BN_BWD_RELU_X) { if (dres) BWD(BN_BWD_RELU_X, true); else BWD(BN_BWD_RELU_X, false); }
    else if (relu_mode == BN_BWD_RELU_MASK) { if (dres) BWD(BN_BWD_RELU_MASK, true); else BWD(BN_BWD_RELU_MASK, false); }
    else { if (dres) BWD(BN_BWD_PLAIN, true); else BWD(BN_BWD_PLAIN, false); }
#undef BWD
    DTF_CHECK_LAUNCH();
    return 0;
}

// Backward finalize only, from reduction partials the producer of dy already wrote (the K9 pool gather of ops/stem.py):
// dgamma, dbeta and the coefficients of dx = k1 dy' + k2 x + k3.
int dtf_bn_bwd_finalize_pre(const float* partial, int nblk, const void* gamma, int param_dtype, const float* mean,
                            const float* invstd, void* dgamma, void* dbeta, float* coef, int64_t M, int C, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (C % 8 != 0 || M <= 0 || nblk <= 0) return -20;
    const float inv_count = 1.0f / (float)M;
    const int fgrid = (C + BN_FIN_CH - 1) / BN_FIN_CH;
    if (param_dtype == DTF_BF16)
        hipLaunchKernelGGL((bn_bwd_finalize_kernel<DTF_BF16>), dim3(fgrid), dim3(256), 0, stream, partial, nblk, C, inv_count, gamma,
                           mean, invstd, dgamma, dbeta, coef);
    else
        hipLaunchKernelGGL((bn_bwd_finalize_kernel<DTF_F32>), dim3(fgrid), dim3(256), 0, stream, partial, nblk, C, inv_count, gamma,
                           mean, invstd, dgamma, dbeta, coef);
    DTF_CHECK_LAUNCH();
    return 0;
}

// Backward of max_pool(relu(BN(x))) (ops/pool.py bn_relu_max_pool2d): dyp / tap are the POOLED gradient
// [N, OH, OW, C] and the pool's winning-tap bytes; see PoolGeom.
int dtf_bn_bwd_train_pool(const void* dyp, const void* tap, const void* x, const void* gamma, int param_dtype,
                          const float* mean, const float* invstd, const float* scale, const float* shift, void* dx,
                          void* dgamma, void* dbeta, float* coef, float* partial, int N, int H, int W, int C, int OH, int OW,
                          int k, int stride, int pad, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    bn_init_order();
    const int64_t M = (int64_t)N * H * W;
    if (C % 8 != 0 || M <= 0 || M >= (1ll << 32) || stride < 1 || pad < 0 || pad >= k) return -20;
    if (OH != (H + 2 * pad - k) / stride + 1 || OW != (W + 2 * pad - k) / stride + 1) return -21;
    int nb, rpb, ct;
    bn_grid(M, C, &nb, &rpb, &ct);
    PoolGeom pg{(const unsigned short*)dyp, (const unsigned char*)tap, H, W, OH, OW, k, stride, pad};
    const unsigned short* xs = (const unsigned short*)x;
    hipLaunchKernelGGL((bn_reduce_kernel<BN_BWD_RELU_X, BN_UNROLL, true>), dim3(nb, ct), dim3(256), 0, stream, xs, nullptr, nullptr,
                       mean, scale, shift, partial, M, C, rpb, (int64_t)C, (int64_t)C, pg);
    const float inv_count = 1.0f / (float)M;
    const int fgrid = (C + BN_FIN_CH - 1) / BN_FIN_CH;
    if (param_dtype == DTF_BF16)
        hipLaunchKernelGGL((bn_bwd_finalize_kernel<DTF_BF16>), dim3(fgrid), dim3(256), 0, stream, partial, nb, C, inv_count, gamma,
                           mean, invstd, dgamma, dbeta, coef);
    else
        hipLaunchKernelGGL((bn_bwd_finalize_kernel<DTF_F32>), dim3(fgrid), dim3(256), 0, stream, partial, nb, C, inv_count, gamma,
                           mean, invstd, dgamma, dbeta, coef);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<BN_BWD_RELU_X, false, true>), dim3(nb, ct), dim3(256), 0, stream, nullptr, xs, nullptr,
                       scale, shift, coef, (unsigned short*)dx, nullptr, M, C, rpb, (int64_t)C, (int64_t)C, (int64_t)C, pg);
    DTF_CHECK_LAUNCH();
    return 0;
}

// relu( BN_a(xa) + BN_b(xb) ), training mode.  stats_a / stats_b: fp32 [4][C] = mean, invstd, scale, shift (outputs);
// partial: workspace of 2 * dtf_bn_partial_floats(M, C) floats; mask: dtf_bn_mask_bytes(M, C)
static int bn2_fwd_train_impl(const void* xa, const void* xb, void* y, void* mask, const void* gamma_a, const void* beta_a,
                      const void* gamma_b, const void* beta_b, int param_dtype, float* rm_a, float* rv_a, float* rm_b,
                      float* rv_b, float momentum_a, float eps_a, float momentum_b, float eps_b, float* stats_a,
                      float* stats_b, float* partial, int64_t M, int C, const float* pre_a, int nblk_a, const float* pre_b,
                      int nblk_b, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    bn_init_order();
    if (C % 8 != 0 || M <= 0 || !mask) return -20;
    int nb, rpb, ct;
    bn_grid(M, C, &nb, &rpb, &ct);
    const float* pa = partial;
    const float* pb = partial + dtf_bn_partial_floats(M, C);
    const unsigned short *xas = (const unsigned short*)xa, *xbs = (const unsigned short*)xb;
    // pre_a / pre_b: statistics partials (plain sums) the producers of xa / xb already wrote (K17 epilogue): no pass over them
    int nsa, nsb;
    const unsigned short *x0a = xas, *x0b = xbs;          // row 0 of a tensor whose partials were accumulated shifted
    if (pre_a) { pa = pre_a; nsa = nblk_a; x0a = nullptr; }
    else nsa = launch_fwd_stats(xas, partial, M, C, (int64_t)C, stream);
    if (pre_b) { pb = pre_b; nsb = nblk_b; x0b = nullptr; }
    else nsb = launch_fwd_stats(xbs, partial + dtf_bn_partial_floats(M, C), M, C, (int64_t)C, stream);
    const float inv_count = 1.0f / (float)M;
    const float unbias = M > 1 ? (float)M / (float)(M - 1) : 1.f;
    const int fgrid = (C + BN_FIN_CH - 1) / BN_FIN_CH;
#define FIN(PDT)                                                                                                              \
    do {                                                                                                                      \
        hipLaunchKernelGGL((bn_fwd_finalize_kernel<PDT>), dim3(fgrid), dim3(256), 0, stream, pa, nsa, C, inv_count, unbias,   \
                           gamma_a, beta_a, rm_a, rv_a, momentum_a, eps_a, stats_a, stats_a + C, stats_a + 2 * C,             \
                           stats_a + 3 * C, x0a);                                                                             \
        hipLaunchKernelGGL((bn_fwd_finalize_kernel<PDT>), dim3(fgrid), dim3(256), 0, stream, pb, nsb, C, inv_count, unbias,   \
                           gamma_b, beta_b, rm_b, rv_b, momentum_b, eps_b, stats_b, stats_b + C, stats_b + 2 * C,             \
                           stats_b + 3 * C, x0b);                                                                             \
    } while (0)
    if (param_dtype == DTF_BF16) FIN(DTF_BF16); else FIN(DTF_F32);
#undef FIN
    if (bn_flat_apply_mode(M, C)) {
        // U = 2, non-temporal loads + stores: the variant that measured best for the single-BN passes
        const int cv = C / 8, rpi = 256 / cv;
        const unsigned fb = (unsigned)((M + 2 * rpi - 1) / (2 * rpi));
        hipLaunchKernelGGL((bn2_fwd_apply_flat_kernel<2, true, true>), dim3(fb), dim3(256), 0, stream, xas, xbs, stats_a + 2 * C,
                           stats_a + 3 * C, stats_b + 2 * C, stats_b + 3 * C, (unsigned short*)y, (unsigned char*)mask, M, C, cv, rpi);
    } else
    hipLaunchKernelGGL(bn2_fwd_apply_kernel, dim3(nb, ct), dim3(256), 0, stream, xas, xbs, stats_a + 2 * C, stats_a + 3 * C,
                       stats_b + 2 * C, stats_b + 3 * C, (unsigned short*)y, (unsigned char*)mask, M, C, rpb);
    DTF_CHECK_LAUNCH();
    return 0;
}

int dtf_bn2_fwd_train(const void* xa, const void* xb, void* y, void* mask, const void* gamma_a, const void* beta_a,
                      const void* gamma_b, const void* beta_b, int param_dtype, float* rm_a, float* rv_a, float* rm_b,
                      float* rv_b, float momentum_a, float eps_a, float momentum_b, float eps_b, float* stats_a,
                      float* stats_b, float* partial, int64_t M, int C, hipStream_t stream) {
    return bn2_fwd_train_impl(xa, xb, y, mask, gamma_a, beta_a, gamma_b, beta_b, param_dtype, rm_a, rv_a, rm_b, rv_b, momentum_a,
                              eps_a, momentum_b, eps_b, stats_a, stats_b, partial, M, C, nullptr, 0, nullptr, 0, stream);
}

// the same with statistics partials of xa and / or xb from their producers (fp32 [nblk][2][C] plain sums; null: own pass)
int dtf_bn2_fwd_train_pre(const void* xa, const void* xb, void* y, void* mask, const void* gamma_a, const void* beta_a,
                          const void* gamma_b, const void* beta_b, int param_dtype, float* rm_a, float* rv_a, float* rm_b,
                          float* rv_b, float momentum_a, float eps_a, float momentum_b, float eps_b, float* stats_a,
                          float* stats_b, float* partial, int64_t M, int C, const float* pre_a, int nblk_a, const float* pre_b,
                          int nblk_b, hipStream_t stream) {
    return bn2_fwd_train_impl(xa, xb, y, mask, gamma_a, beta_a, gamma_b, beta_b, param_dtype, rm_a, rv_a, rm_b, rv_b, momentum_a,
                              eps_a, momentum_b, eps_b, stats_a, stats_b, partial, M, C, pre_a, nblk_a, pre_b, nblk_b, stream);
}

// backward of dtf_bn2_fwd_train.  dgb_a / dgb_b: param dtype [2][C] = dgamma, dbeta; coef: fp32 [2][3][C] workspace
int dtf_bn2_bwd_train(const void* dy, const void* xa, const void* xb, const void* mask, const void* gamma_a,
                      const void* gamma_b, int param_dtype, const float* stats_a, const float* stats_b, void* dxa, void* dxb,
                      void* dgb_a, void* dgb_b, float* coef, float* partial, int64_t M, int C, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    bn_init_order();
    if (C % 8 != 0 || M <= 0 || !mask) return -20;
    int nb, rpb, ct;
    bn_grid(M, C, &nb, &rpb, &ct);
    float* pa = partial;
    float* pb = partial + (int64_t)nb * 2 * C;
    float* coef_a = coef;
    float* coef_b = coef + 3 * (int64_t)C;
    const unsigned short *dys = (const unsigned short*)dy, *xas = (const unsigned short*)xa, *xbs = (const unsigned short*)xb;
    hipLaunchKernelGGL(bn2_bwd_reduce_kernel, dim3(nb, ct), dim3(256), 0, stream, dys, xas, xbs, (const unsigned char*)mask,
                       stats_a, stats_b, pa, pb, M, C, rpb);
    const float inv_count = 1.0f / (float)M;
    const int fgrid = (C + BN_FIN_CH - 1) / BN_FIN_CH;
    const int esz = param_dtype == DTF_BF16 ? 2 : 4;
#define FIN(PDT)                                                                                                              \
    do {                                                                                                                      \
        hipLaunchKernelGGL((bn_bwd_finalize_kernel<PDT>), dim3(fgrid), dim3(256), 0, stream, pa, nb, C, inv_count, gamma_a,    \
                           stats_a, stats_a + C, dgb_a, (void*)((char*)dgb_a + (int64_t)C * esz), coef_a);                    \
        hipLaunchKernelGGL((bn_bwd_finalize_kernel<PDT>), dim3(fgrid), dim3(256), 0, stream, pb, nb, C, inv_count, gamma_b,    \
                           stats_b, stats_b + C, dgb_b, (void*)((char*)dgb_b + (int64_t)C * esz), coef_b);                    \
    } while (0)
    if (param_dtype == DTF_BF16) FIN(DTF_BF16); else FIN(DTF_F32);
#undef FIN
    if (bn_flat_apply_mode(M, C)) {
        const int cv = C / 8, rpi = 256 / cv;
        const unsigned fb = (unsigned)((M + 2 * rpi - 1) / (2 * rpi));
        hipLaunchKernelGGL((bn2_bwd_apply_flat_kernel<2, true, true>), dim3(fb), dim3(256), 0, stream, dys, xas, xbs,
                           (const unsigned char*)mask, coef_a, coef_b, (unsigned short*)dxa, (unsigned short*)dxb, M, C, cv, rpi);
    } else
    hipLaunchKernelGGL(bn2_bwd_apply_kernel, dim3(nb, ct), dim3(256), 0, stream, dys, xas, xbs, (const unsigned char*)mask, coef_a,
                       coef_b, (unsigned short*)dxa, (unsigned short*)dxb, M, C, rpb);
    DTF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
