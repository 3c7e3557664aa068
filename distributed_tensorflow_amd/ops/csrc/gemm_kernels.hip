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

__device__ __forceinline__ float gelu_erf(float x) {
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
    float* stats;        // gemm256p_kernel<.., STATS = true>: per-row-block column sums / sums of squares of Y
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
                for (int k = 0; k < 8; ++k) v[k] = gelu_erf(v[k]);
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
            store8<DTF_BF16>(a.Y, DTF_IDX(o, (int64_t)a.M * a.N - 7, "gemm128 epilogue store"), v);
        } else {
            for (int k = 0; k < 8 && gn + k < a.N; ++k) {
                float x = v[k];
                if (a.bias) x += (a.bias_dtype == DTF_BF16) ? load1<DTF_BF16>(a.bias, gn + k) : load1<DTF_F32>(a.bias, gn + k);
                if (a.preact) a.preact[o + k] = f32_to_bf16(x);
                if (a.act == ACT_RELU) x = fmaxf(x, 0.f);
                else if (a.act == ACT_GELU) x = gelu_erf(x);
                else if (a.act == ACT_GELU_TANH) x = gelu_tanh_fwd(x);
                if (a.residual) x += bf16_to_f32(a.residual[o + k]);
                a.Y[o + k] = f32_to_bf16(x);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// 256 x 256 tile variant for large problems: persistent blocks (one per CU) that walk the tile list,
// 8 waves as 2(M) x 4(N), each a 128 x 64 sub-tile (8 x 4 accumulators = 128 fp32 regs), ~1 block/CU.
//  * Staging is LDS-DMA (global_load_lds, 16 B per lane): no staging registers, no ds_write pass.
//    K tiles of 32, FOUR 32 KB buffers.  One wave instruction fills 1 KB = the image of one 16-row MFMA
//    fragment group ([16 rows][64 B], chunk-XOR-swizzled on the source side: conflict-free ds_read_b128).
//  * Pipeline: three K tiles in flight while one feeds the MFMAs (one tile ahead is shorter than the
//    loaded HBM/L2 latency: 50 % MFMA rate).  The LDS-DMA stays in flight ACROSS barriers: counted
//    vmcnt (4 glds per wave per tile) and raw s_barrier -- __syncthreads() would drain vmcnt(0).
//  * Ping-pong: the two waves that share a SIMD (w and w+4, wr = 0 / 1) run one barrier apart, so while
//    one half of the block issues its 32 MFMAs the other half reads its fragments from LDS:
//        wr=0:  R0 | M0 w1 | R1 | M1 w2 | ...     R = 12 ds_read_b128 (+ restage), M = 32 MFMAs,
//        wr=1:     | R0 w1 | M0 | R1 w2 | ...     wN = wait for own pieces of tile N, | = s_barrier
//    Tile N is first read (by wr=0) after the barrier that follows both groups' wN; buffers are restaged
//    in phase R(t), which for both groups starts after the barrier that ended the last read of tile t-1.
//  * K > 4096 stages tiles in PAIRS so the two 64-B halves of every 128-B line are requested back to back
//    (8192^3: 931 -> 1155 TF); smaller K keeps the deeper one-tile-per-step schedule.
//  * Epilogue straight from the accumulators: the MFMA operands are swapped (D = W.X^T fragment), so a
//    lane holds 4 CONSECUTIVE output columns of one row -- bias / activation / residual in registers and
//    8-byte stores (the four j fragments of a wave complete each 128-B line), no LDS round trip and no
//    barriers.  LDS is therefore free during the epilogue: the first K tiles of the block's NEXT output
//    tile are already streaming in while the current one is written out (with one block per CU nothing
//    else would hide the prologue / epilogue: they were 60 % of a K = 768 tile's time).
// Rows past M / N are clamped to the last valid row (their results are never stored); K must be a
// multiple of 32 here (no zero fill with LDS-DMA) -- other K use the 128 kernel.
// ---------------------------------------------------------------------------
#define G2_BM 256
#define G2_BN 256

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ float k6_act(float x, int act) {
    if (act == ACT_RELU) return fmaxf(x, 0.f);
    if (act == ACT_GELU) return gelu_erf(x);
    if (act == ACT_GELU_TANH) return gelu_tanh_fwd(x);
    return x;
}

// sum over the 16 lanes of a DPP row (lane & 15): row_ror 8, row_ror 4, then the two quad permutes
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    return v;
}

// STATS: the epilogue also reduces the (bf16-rounded) output tile over its rows -- per 128-row half tile the
// column sums and sums of squares, written as partial rows [(2 tm + wr) * 2 + stat][N] in the layout the K7
// BatchNorm finalize kernel reads: a 1x1 convolution feeding a training-mode BatchNorm needs no separate
// statistics pass over its output (2 B/element of HBM reads saved).
// NJ: 16-column B fragments per wave -- the tile is 256 x (64 * NJ): NJ = 3 (256 x 192) where that fills the CUs in
// fewer tile-rounds (N = 768: 384 tiles of 256^2 are 1.5 rounds on 256 CUs, 512 tiles of 256 x 192 are exactly 2).
template <int ACT, bool STATS = false, int NJ = 4>
__global__ __launch_bounds__(512) void gemm256p_kernel(GemmArgs a, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int K = a.K;
    const int nk = K / 32;
    const bool pair = K > 4096;
    const int frow = lane & 15, fq = lane >> 4;
    const int lane_off = frow * 64 + ((fq ^ ((4 - (frow >> 2)) & 3)) << 4);

    const unsigned short* pa[2];
    const unsigned short* pb[2];
    int m0 = 0, n0 = 0;
    auto set_tile = [&](int tile) {        // wave w stages row groups 2w, 2w+1 of A and of B
        const int tm = tile / a.ntn, tn = tile - tm * a.ntn;
        m0 = tm * G2_BM;
        n0 = tn * (64 * NJ);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // LDS-DMA lane l lands at byte 16 l of the 1 KB piece = [16 rows][4 chunks of 16 B] with 4
            // consecutive lanes on one 64-B row segment (what the texture addresser coalesces; one lane per
            // row measured 12-20 % slower).  Read plainly (frow*64 + fq*16) that image is 2-way conflicted:
            // ds_read_b128 serves lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... and within a group
            // the lanes of one fq all hit the same chunk column (PMC: 4 conflict cycles per read).  XOR the
            // chunk with s(row >> 2), s = [0,3,2,1]: every group then covers all 16 chunk slots once.  The
            // swizzle goes on the per-lane SOURCE address and on the read address.
            const int row = (wave * 2 + i) * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);        // (row >> 2) & 3 == (lane >> 4) & 3
            int gm = m0 + row, gn = n0 + row;
            if (gm > a.M - 1) gm = a.M - 1;
            if (gn > a.N - 1) gn = a.N - 1;
            pa[i] = a.X + (int64_t)gm * K + chunk * 8;
            pb[i] = a.W + (int64_t)gn * K + chunk * 8;
        }
    };
    auto stage = [&](int t) {
        unsigned char* base = smem + (t & 3) * 32768 + wave * 2048;
        const int k0 = t * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((glb_void_t*)(pa[i] + k0), (lds_void_t*)(base + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void_t*)(pb[i] + k0), (lds_void_t*)(base + 16384 + i * 1024), 16, 0, 0);
        }
    };
    auto prologue = [&]() {
        stage(0);
        if (nk > 1) stage(1);
        if (!pair && nk > 2) stage(2);
    };
    // tiles that may still be in flight when tile t+1 is needed: staged so far (UPTO) minus (t+1)
#define G2_WAIT_NEXT(T, UPTO)                                                        \
    do {                                                                             \
        int upto = (UPTO);                                                           \
        if (upto > nk - 1) upto = nk - 1;                                            \
        const int n_out = upto - ((T) + 1);                                          \
        if (n_out >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             \
        else if (n_out == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        \
    } while (0)

    int tile = xcd_remap(blockIdx.x, gridDim.x);        // neighbouring tiles on one XCD (shared L2)
    if (tile < ntiles) {
        set_tile(tile);
        prologue();
    }
    for (; tile < ntiles; tile += gridDim.x) {
        const int cm0 = m0, cn0 = n0;
        f32x4 acc[8][NJ];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // the prefetched K tiles (and the previous epilogue's stores) have landed.  This wait must be the
        // BUILTIN (a real S_WAITCNT the compiler's wait-count pass sees, vmcnt(0) = 0x0F70 on gfx9): with an
        // asm string it still believes the epilogue's stores are pending at the K loop and drains
        // vmcnt(0) before every LDS-DMA and every ds_read of the loop (measured: 1265 -> 800 TF).
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();          // stagger the second wave of every SIMD
        for (int t = 0; t < nk; ++t) {
            const unsigned char* A = smem + (t & 3) * 32768 + wr * 8192 + lane_off;
            const unsigned char* B = smem + (t & 3) * 32768 + 16384 + wc * (NJ * 1024) + lane_off;
            bf16x8_t fa[8], fb[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(B + j * 1024);
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(A + i * 1024);
            // wr=1 waits for tile t+1 here, BEFORE it restages in its M phase: staged so far = through t+2
            // (pair mode: through t+1 at even t, t+2 at odd t)
            if (wr == 1) G2_WAIT_NEXT(t, pair ? ((t & 1) ? t + 2 : t + 1) : t + 2);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < NJ; ++j)      // operands swapped: D[n][m], 4 consecutive n per lane
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[0], acc[0][j], 0, 0, 0);
            // the restage rides in the MFMA phase: an LDS-DMA piece costs 60-180 issue cycles, which made the
            // READ phase (12 ds_read_b128 + 4 pieces) twice as long as the 512-cycle MFMA phase it overlaps
            if (!pair) {
                if (t + 3 < nk) stage(t + 3);
            } else if ((t & 1) == 0) {
                if (t + 2 < nk) stage(t + 2);
                if (t + 3 < nk) stage(t + 3);
            }
#pragma unroll
            for (int i = 1; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            if (wr == 0) G2_WAIT_NEXT(t, (pair && (t & 1)) ? t + 2 : t + 3);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();          // re-align the two groups: every LDS read is done
        __builtin_amdgcn_sched_barrier(0);
        // bias of this tile first: an ordinary load's first use drains every LDS-DMA issued before it
        float bv[NJ][4];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = cn0 + wc * (16 * NJ) + j * 16 + fq * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[j][r] = 0.f;
            if ((a.N & 7) == 0 && a.bias && n < a.N) {
                if (a.bias_dtype == DTF_BF16) {
                    const u32x2 t2 = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(a.bias) + n);
                    bv[j][0] = __uint_as_float(t2[0] << 16); bv[j][1] = __uint_as_float(t2[0] & 0xffff0000u);
                    bv[j][2] = __uint_as_float(t2[1] << 16); bv[j][3] = __uint_as_float(t2[1] & 0xffff0000u);
                } else {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.bias) + n);
                    bv[j][0] = t4[0]; bv[j][1] = t4[1]; bv[j][2] = t4[2]; bv[j][3] = t4[3];
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_sched_barrier(0);
        if (tile + (int)gridDim.x < ntiles) {               // next tile's first K tiles stream in under the epilogue
            set_tile(tile + gridDim.x);
            prologue();
        }

        // ---- epilogue.  Lane (frow, fq) of fragment (i, j) holds row m = frow, columns 16j + 4fq .. +3:
        // bias + activation in registers, then each wave transposes its 16 x 64 fragment row through a
        // PRIVATE 4 KB piece of K-tile buffer 3 (free until the next tile's first restage; wave-private, so
        // no block barrier: LDS operations of one wave complete in order) and writes full 128-B lines:
        // 4 lanes x 32 B per row, residual added in fp32 before the single rounding.  The LDS accesses are
        // inline asm so that the compiler does not drain the in-flight LDS-DMA of the next tile first.
        if ((a.N & 7) == 0) {
            const unsigned ebase = (unsigned)(uintptr_t)(lds_void_t*)smem + 98304u + (unsigned)wave * 4096u;
            const int rr = lane >> 3, rc = lane & 7;            // read-back: row (and row + 8), 8-column group
            float st1[NJ][4], st2[NJ][4];
            if (STATS) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { st1[j][r] = 0.f; st2[j][r] = 0.f; }
            }
            // One trip of a 16 x (16 NJ) fragment row through the wave's LDS piece: fragment layout in, line layout
            // out (lane = row rr [+8], column octet rc: 8 lanes x 16 B cover one 128-B output line, so every store
            // instruction writes whole lines; 4 lanes x 2 half-filled 32-B sectors per row measured 2.5 TB/s).
            auto transpose16 = [&](const f32x4 (&v)[NJ], f32x4 (&o4)[4]) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const unsigned wa = ebase + (unsigned)(frow * 256 + (((j * 4 + fq) ^ frow) << 4));
                    asm volatile("ds_write_b128 %0, %1" ::"v"(wa), "v"(v[j]) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = rr + 8 * (q >> 1);
                    const unsigned ra = ebase + (unsigned)(row * 256 + (((rc * 2 + (q & 1)) ^ row) << 4));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(o4[q]) : "v"(ra) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            };
            const int on = cn0 + wc * (16 * NJ) + rc * 8;
            const bool col_ok = on < a.N && rc < 2 * NJ;          // N % 8 == 0: an octet is all-or-nothing
            // residual octets of fragment row i: loaded one iteration ahead, BEFORE the stores of the iteration in
            // between -- vmcnt counts loads and stores in order, so a load issued after the stores would make its
            // consumer wait for their write acknowledgements too (measured: +20 us on a 50 MB residual)
            u32x4 rres[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
            auto load_res = [&](int i, u32x4 (&dst)[2]) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int om = cm0 + wr * 128 + i * 16 + rr + 8 * h;
                    if (om < a.M && col_ok) dst[h] = *reinterpret_cast<const u32x4*>(a.residual + (int64_t)om * a.N + on);
                }
            };
            if (a.residual) load_res(0, rres);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = cm0 + wr * 128 + i * 16 + frow;
                f32x4 v[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[j][r] = acc[i][j][r] + bv[j][r];
                    if (STATS && m < a.M) {          // statistics of what the consumer will read: the rounded values
                        const unsigned p0 = pack_bf16x2(v[j][0], v[j][1]), p1 = pack_bf16x2(v[j][2], v[j][3]);
                        const float q[4] = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u),
                                            __uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
#pragma unroll
                        for (int r = 0; r < 4; ++r) { st1[j][r] += q[r]; st2[j][r] = fmaf(q[r], q[r], st2[j][r]); }
                    }
                }
                f32x4 o4[4];
                if (a.preact && m < a.M) {          // pre-activation (saved for the backward), straight from the fragments
                    // (a second trip through LDS for whole-line stores measured 1-7 us SLOWER per GEMM)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int n = cn0 + wc * (16 * NJ) + j * 16 + fq * 4;
                        if (n < a.N) {
                            u32x2 pk;
                            pk[0] = pack_bf16x2(v[j][0], v[j][1]);
                            pk[1] = pack_bf16x2(v[j][2], v[j][3]);
                            *reinterpret_cast<u32x2*>(a.preact + (int64_t)m * a.N + n) = pk;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[j][r] = k6_act(v[j][r], ACT);
                transpose16(v, o4);
                u32x4 nres[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
                if (a.residual && i + 1 < 8) load_res(i + 1, nres);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int om = cm0 + wr * 128 + i * 16 + rr + 8 * h;
                    if (om < a.M && col_ok) {
                        float v8[8];
#pragma unroll
                        for (int k = 0; k < 4; ++k) { v8[k] = o4[2 * h][k]; v8[4 + k] = o4[2 * h + 1][k]; }
                        if (a.residual) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                v8[2 * k] += __uint_as_float(rres[h][k] << 16);
                                v8[2 * k + 1] += __uint_as_float(rres[h][k] & 0xffff0000u);
                            }
                        }
                        store8<DTF_BF16>(a.Y, DTF_IDX((int64_t)om * a.N + on, (int64_t)a.M * a.N - 7, "gemm256p epilogue store"), v8);
                    }
                }
                rres[0] = nres[0];
                rres[1] = nres[1];
            }
            if (STATS) {
                const int blk = (cm0 / G2_BM) * 2 + wr;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    f32x4 o1, o2;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { o1[r] = row16_sum(st1[j][r]); o2[r] = row16_sum(st2[j][r]); }
                    const int n = cn0 + wc * (16 * NJ) + j * 16 + fq * 4;
                    if (frow == 0 && n < a.N) {
                        *reinterpret_cast<f32x4*>(a.stats + ((int64_t)blk * 2 + 0) * a.N + n) = o1;
                        *reinterpret_cast<f32x4*>(a.stats + ((int64_t)blk * 2 + 1) * a.N + n) = o2;
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = cm0 + wr * 128 + i * 16 + frow;
                if (m >= a.M) continue;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = cn0 + wc * (16 * NJ) + j * 16 + fq * 4 + r;
                        if (n >= a.N) continue;
                        const int64_t o = (int64_t)m * a.N + n;
                        float x = acc[i][j][r];
                        if (a.bias) x += (a.bias_dtype == DTF_BF16) ? load1<DTF_BF16>(a.bias, n) : load1<DTF_F32>(a.bias, n);
                        if (a.preact) a.preact[o] = f32_to_bf16(x);
                        x = k6_act(x, ACT);
                        if (a.residual) x += bf16_to_f32(a.residual[o]);
                        a.Y[o] = f32_to_bf16(x);
                    }
            }
        }
    }
#undef G2_WAIT_NEXT
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

static int force_nj = 0;         // 3 | 4 pins the 256p tile width (DTF_GEMM_NJ)
static int force_tile = 0;       // 128 | 256 pins the tile variant (DTF_GEMM_TILE or dtf_gemm_set_tile; A/B runs)

// y = act(z) on a dense bf16 [n] stream: the activation pass behind a LIBRARY GEMM whose epilogue added the bias and
// wrote the pre-activation z (kept for the backward).  With per-shape tuned hipBLASLt solutions the pair
// addmm + this pass beats the fused K6 epilogue on BERT's FFN1 shape (131 + ~70 us vs 252 us,
// scripts/k6_epilogue_bench.py): K6's K = 768 tiles are epilogue-bound, see docs/STATUS.md.
// One block per two 4 KB pieces, non-temporal loads (z is not read again before the backward).
__global__ __launch_bounds__(256) void act_fwd_kernel(const unsigned short* __restrict__ z, unsigned short* __restrict__ y,
                                                      int64_t nvec, int act) {
    const int64_t v0 = (int64_t)blockIdx.x * 512 + threadIdx.x;
    const u32x4* zv = reinterpret_cast<const u32x4*>(z);
    u32x4* yv = reinterpret_cast<u32x4*>(y);
    u32x4 a[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (v0 + u * 256 < nvec) a[u] = __builtin_nontemporal_load(zv + v0 + u * 256);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (v0 + u * 256 >= nvec) continue;
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float lo = __uint_as_float(a[u][k] << 16), hi = __uint_as_float(a[u][k] & 0xffff0000u);
            if (act == ACT_RELU) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
            else if (act == ACT_GELU) { lo = gelu_erf(lo); hi = gelu_erf(hi); }
            else if (act == ACT_GELU_TANH) { lo = gelu_tanh_fwd(lo); hi = gelu_tanh_fwd(hi); }
            o[k] = pack_bf16x2(lo, hi);
        }
        yv[v0 + u * 256] = o;
    }
}

extern "C" {

// y = act(z), n % 8 == 0, 16-byte aligned
int dtf_act_fwd(const void* z, void* y, int64_t n, int act, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (n <= 0) return 0;
    if ((n & 7) || ((((uintptr_t)z) | ((uintptr_t)y)) & 15)) return -16;
    const int64_t nvec = n >> 3;
    hipLaunchKernelGGL(act_fwd_kernel, dim3((unsigned)((nvec + 511) / 512)), dim3(256), 0, stream, (const unsigned short*)z,
                       (unsigned short*)y, nvec, act);
    DTF_CHECK_LAUNCH();
    return 0;
}

void dtf_gemm_set_tile(int t) { force_tile = t; }
void dtf_gemm_set_nj(int nj) { force_nj = nj; }

int dtf_gemm_bias_act(const void* X, const void* W, const void* bias, int bias_dtype, const void* residual, void* Y,
                      void* preact, int M, int N, int K, int act, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || (K & 7)) return -40;
    if ((((uintptr_t)X) | ((uintptr_t)W) | ((uintptr_t)Y) | ((uintptr_t)residual) | ((uintptr_t)preact)) & 15) return -16;
    GemmArgs a{(const unsigned short*)X, (const unsigned short*)W, bias, (const unsigned short*)residual,
               (unsigned short*)Y, (unsigned short*)preact, M, N, K, bias_dtype, act,
               (M + GM_BM - 1) / GM_BM, (N + GM_BN - 1) / GM_BN, nullptr};
    static bool attr_set = false;
    const int lds = GM_BM * (GM_BN + 4) * 4;     // 67584 B: epilogue tile (>= 64 KB staging)
    const int lds256 = 131072;                   // four 32 KB K-tile buffers
    static int num_cu = 0;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm_bias_act_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)gemm256p_kernel<ACT_NONE, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds256);
        (void)hipFuncSetAttribute((const void*)gemm256p_kernel<ACT_RELU, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds256);
        (void)hipFuncSetAttribute((const void*)gemm256p_kernel<ACT_GELU, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds256);
        (void)hipFuncSetAttribute((const void*)gemm256p_kernel<ACT_GELU_TANH, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds256);
        (void)hipFuncSetAttribute((const void*)gemm256p_kernel<ACT_NONE, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds256);
        (void)hipFuncSetAttribute((const void*)gemm256p_kernel<ACT_RELU, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds256);
        (void)hipFuncSetAttribute((const void*)gemm256p_kernel<ACT_GELU, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds256);
        (void)hipFuncSetAttribute((const void*)gemm256p_kernel<ACT_GELU_TANH, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds256);
        const char* ej = getenv("DTF_GEMM_NJ");
        if (ej) force_nj = atoi(ej);
        int dev = 0;
        hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || num_cu <= 0) num_cu = 256;
        const char* e = getenv("DTF_GEMM_TILE");
        if (e && force_tile == 0) force_tile = atoi(e);
        attr_set = true;
    }
    const int ntm2 = (M + G2_BM - 1) / G2_BM, ntn2 = (N + G2_BN - 1) / G2_BN;
    bool big = (K % 32) == 0 && (int64_t)ntm2 * ntn2 >= 384;     // >= 1.5 tiles per CU
    if (force_tile == 128) big = false;
    if (force_tile == 256 && (K % 32) == 0) big = true;
    if (big) {
        // 256 x 192 tiles where they need fewer (size-weighted) rounds over the CUs.  DTF_GEMM_NJ=3|4 pins the
        // choice (A/B runs).
        const int ntn3 = (N + 191) / 192;
        const int64_t rounds4 = ((int64_t)ntm2 * ntn2 + num_cu - 1) / num_cu, rounds3 = ((int64_t)ntm2 * ntn3 + num_cu - 1) / num_cu;
        // measured: a 256 x 192 round costs 0.84 of a 256 x 256 round (not 0.75: fewer MFMAs per A fragment read)
        bool nj3 = (N & 7) == 0 && (double)rounds3 * 0.84 < 0.97 * (double)rounds4;
        if (force_nj == 3) nj3 = (N & 7) == 0;
        if (force_nj == 4) nj3 = false;
        a.ntm = ntm2;
        a.ntn = nj3 ? ntn3 : ntn2;
        const int ntiles = a.ntm * a.ntn;
        const int grid = ntiles < num_cu ? ntiles : num_cu;
#define G2_LAUNCH(A_)                                                                                                        \
    do {                                                                                                                     \
        if (nj3) hipLaunchKernelGGL((gemm256p_kernel<A_, false, 3>), dim3(grid), dim3(512), lds256, stream, a, ntiles);      \
        else hipLaunchKernelGGL((gemm256p_kernel<A_, false, 4>), dim3(grid), dim3(512), lds256, stream, a, ntiles);          \
    } while (0)
        if (act == ACT_RELU) G2_LAUNCH(ACT_RELU);
        else if (act == ACT_GELU) G2_LAUNCH(ACT_GELU);
        else if (act == ACT_GELU_TANH) G2_LAUNCH(ACT_GELU_TANH);
        else G2_LAUNCH(ACT_NONE);
#undef G2_LAUNCH
    } else {
        hipLaunchKernelGGL(gemm_bias_act_kernel, dim3(a.ntm * a.ntn), dim3(256), lds, stream, a);
    }
    DTF_CHECK_LAUNCH();
    return 0;
}

// partial rows of dtf_gemm_stats: two 128-row halves per 256-row tile
int64_t dtf_gemm_stats_blocks(int64_t M) { return 2 * ((M + G2_BM - 1) / G2_BM); }

// Y[M,N] = X[M,K] . W[N,K]^T (bf16) plus the BatchNorm statistics partials of Y (see gemm256p_kernel STATS):
// stats: fp32 [dtf_gemm_stats_blocks(M)][2][N]
int dtf_gemm_stats(const void* X, const void* W, void* Y, float* stats, int M, int N, int K, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || (K & 31) || (N & 7)) return -42;
    if ((((uintptr_t)X) | ((uintptr_t)W) | ((uintptr_t)Y) | ((uintptr_t)stats)) & 15) return -16;
    GemmArgs a{(const unsigned short*)X, (const unsigned short*)W, nullptr, nullptr, (unsigned short*)Y, nullptr, M, N, K, 0,
               ACT_NONE, (M + G2_BM - 1) / G2_BM, (N + G2_BN - 1) / G2_BN, stats};
    static bool attr_set = false;
    static int num_cu = 0;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm256p_kernel<ACT_NONE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || num_cu <= 0) num_cu = 256;
        attr_set = true;
    }
    const int ntiles = a.ntm * a.ntn;
    hipLaunchKernelGGL((gemm256p_kernel<ACT_NONE, true>), dim3(ntiles < num_cu ? ntiles : num_cu), dim3(512), 131072, stream, a,
                       ntiles);
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
