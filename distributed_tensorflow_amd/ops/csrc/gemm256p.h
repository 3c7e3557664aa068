// K6 persistent 256-row LDS-DMA GEMM pipeline, shared by the plain GEMM entry points (gemm_kernels.hip) and the
// implicit-GEMM convolution (conv_kernels.hip).  See the block comment above gemm256p_kernel.
#pragma once
#include "dtf_common.h"

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
    // CONV = true (implicit GEMM over an NHWC image, conv_kernels.hip): X is the image [NB, cH, cW, cC], W the filter
    // [N = Cout][cKH][cKW][cC] (K = cKH * cKW * cC), Y the NHWC output [NB, cOH, cOW, N] (M = NB * cOH * cOW)
    const unsigned short* zero;      // >= 64 B of zeros: the source of filter taps that fall into the padding
    int cH, cW, cC, cOH, cOW, cKH, cKW, cS, cPH, cPW;
    int group_m;                     // tile order: row panels per group (>= 1; 1 = row-major)
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, pos = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
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
//  * Tile order: groups of 4 row panels x all column panels, row panel fastest, on top of the XCD remap: the ~32
//    tiles that run on one XCD at a time form a 4 x 8 block and pull 4 + 8 operand panels per K step through its
//    L2 instead of 1 + 32 (8192^3: 1074 -> 1337 TF together with dropping the earlier "stage K tiles in pairs" mode,
//    which only papered over that traffic).
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
#ifndef G2_RH
#define G2_RH -1
#endif
#define G2_BN 256

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ float k6_act(float x, int act) {
    if (act == ACT_RELU) return relu_nan(x);
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
// MI: 16-row A fragments per wave -- the tile is (32 * MI) x (64 * NJ).  MI = 3..8 lets the host pick the row-tile
// height that fills the last round over the CUs (ResNet-50 stage 3 at batch 256: 196 tiles of 256 rows leave 60 CUs
// idle, 224 tiles of 224 rows leave 32 and are 7/8 the work each).
// CONV: the A operand is gathered from an NHWC image (implicit im2col): row m = output pixel (n, oh, ow), K tile t =
// 32 input channels of ONE filter tap (cC % 32 == 0).  LDS-DMA takes a per-lane SOURCE address, so the gather costs
// one add + one select per piece: lane pointer of the patch origin + wave-uniform tap offset, or the zero page when
// the tap falls into the padding (a per-row bit mask computed once per tile).  Any stride / padding, cKH * cKW <= 32.
template <int ACT, bool STATS = false, int NJ = 4, int MI = 8, bool CONV = false>
__global__ __launch_bounds__(512) void gemm256p_kernel(GemmArgs a, int ntiles) {
    constexpr int NA = (MI + 3) / 4;           // 16-row pieces of A / of B that one wave stages per K tile (odd MI /
    constexpr int NB = (NJ + 1) / 2;           // NJ: the last pieces are over-staged rows of the neighbouring tile)
    constexpr int GL = NA + NB;                // LDS-DMA instructions per wave per K tile
    constexpr int BM = 32 * MI;
    // pieces restaged in the READ phase (the rest in the MFMA phase): all of them when the partner's MFMA phase is
    // long enough to hide them, one less for the short MFMA phases of the small tiles (measured, G2_RH = 0..4 builds:
    // 4096^3 1175 / 1175 / 1207 / 1256 / 1276 TF; 256-channel 3x3 convolution 65.4 / 62.8 / 58.3 / 59.1 / 56.2 us;
    // 160 x 192 tile 61.1 / 57.8 / 56.8 / 55.8 / 61.7 us)
    constexpr int RH = G2_RH >= 0 ? (G2_RH < GL ? G2_RH : GL) : (MI * NJ >= 24 ? GL : GL - 1);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int K = a.K;
    const int nk = K / 32;
    const int frow = lane & 15, fq = lane >> 4;
    const int lane_off = frow * 64 + ((fq ^ ((4 - (frow >> 2)) & 3)) << 4);

    const unsigned short* pa[NA];
    const unsigned short* pb[NB];
    unsigned amask[NA];                    // CONV: bit (r * cKW + s) = tap (r, s) of this lane's row is inside the image
    int s_tap = 0, s_cc = 0, s_s = 0, s_koff = 0;      // CONV: staging cursor (wave-uniform): tap, channel tile, s, offset
    const int cpt = CONV ? a.cC / 32 : 1;
    const unsigned short* zp = nullptr;
    if constexpr (CONV) zp = a.zero + ((lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3)) * 8;
    int m0 = 0, n0 = 0;
    auto set_tile = [&](int tile) {        // wave w stages row groups NA w .. of A and NB w .. of B
        // grouped order: group_m row panels x all column panels at a time, row panel fastest -- the ~32 consecutive
        // tiles that run on one XCD at a time then cover a (group_m x 32 / group_m) block and fetch group_m A panels +
        // 32 / group_m B panels per K step into its L2 instead of 1 + 32 (row-major order)
        const int width = a.group_m * a.ntn, g = tile / width, first = g * a.group_m;
        const int gsz = a.ntm - first < a.group_m ? a.ntm - first : a.group_m;
        const int in_g = tile - g * width;
        const int tn = in_g / gsz, tm = first + (in_g - tn * gsz);
        m0 = tm * BM;
        n0 = tn * (64 * NJ);
        if constexpr (CONV) {
            s_tap = 0; s_cc = 0; s_s = 0; s_koff = 0;
            const int ohw = a.cOH * a.cOW;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int row = (wave * NA + i) * 16 + (lane >> 2);
                const int chunk = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);
                int gm = m0 + row;
                if (gm > a.M - 1) gm = a.M - 1;
                const int n = gm / ohw, rem = gm - n * ohw;
                const int oh = rem / a.cOW, ow = rem - oh * a.cOW;
                const int ih0 = oh * a.cS - a.cPH, iw0 = ow * a.cS - a.cPW;
                pa[i] = a.X + (((int64_t)n * a.cH + ih0) * a.cW + iw0) * a.cC + chunk * 8;
                unsigned mk = 0;
                for (int r = 0; r < a.cKH; ++r)
                    for (int s = 0; s < a.cKW; ++s)
                        if ((unsigned)(ih0 + r) < (unsigned)a.cH && (unsigned)(iw0 + s) < (unsigned)a.cW)
                            mk |= 1u << (r * a.cKW + s);
                amask[i] = mk;
            }
        }
#pragma unroll
        for (int i = 0; i < (NA > NB ? NA : NB); ++i) {
            // LDS-DMA lane l lands at byte 16 l of the 1 KB piece = [16 rows][4 chunks of 16 B] with 4
            // consecutive lanes on one 64-B row segment (what the texture addresser coalesces; one lane per
            // row measured 12-20 % slower).  Read plainly (frow*64 + fq*16) that image is 2-way conflicted:
            // ds_read_b128 serves lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... and within a group
            // the lanes of one fq all hit the same chunk column (PMC: 4 conflict cycles per read).  XOR the
            // chunk with s(row >> 2), s = [0,3,2,1]: every group then covers all 16 chunk slots once.  The
            // swizzle goes on the per-lane SOURCE address and on the read address.
            const int chunk = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);        // (row >> 2) & 3 == (lane >> 4) & 3
            if (!CONV && i < NA) {
                int gm = m0 + (wave * NA + i) * 16 + (lane >> 2);
                if (gm > a.M - 1) gm = a.M - 1;
                pa[i] = a.X + (int64_t)gm * K + chunk * 8;
            }
            if (i < NB) {
                int gn = n0 + (wave * NB + i) * 16 + (lane >> 2);
                if (gn > a.N - 1) gn = a.N - 1;
                pb[i] = a.W + (int64_t)gn * K + chunk * 8;
            }
        }
    };
    // piece q of K tile t: the wave's A pieces first (q < NA), then its B pieces
    auto stage_piece = [&](int t, int q) {
        unsigned char* base = smem + (t & 3) * 32768;
        const int k0 = t * 32;
        if (q < NA) {
            const unsigned short* src;
            if constexpr (CONV) src = ((amask[q] >> s_tap) & 1u) ? pa[q] + s_koff : zp;
            else src = pa[q] + k0;
            __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(base + (wave * NA + q) * 1024), 16, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds((glb_void_t*)(pb[q - NA] + k0), (lds_void_t*)(base + 16384 + (wave * NB + q - NA) * 1024), 16, 0, 0);
        }
    };
    auto stage_advance = [&]() {
        if constexpr (CONV) {          // next K tile: 32 channels on; past the last channel tile the next tap IS +32
            s_koff += 32;              // elements on (tap offset = (r cW + s) cC), except at the end of a filter row
            if (++s_cc == cpt) {
                s_cc = 0;
                ++s_tap;
                if (++s_s == a.cKW) { s_s = 0; s_koff += (a.cW - a.cKW) * a.cC; }
            }
        }
    };
    auto stage = [&](int t) {
#pragma unroll
        for (int q = 0; q < GL; ++q) stage_piece(t, q);
        stage_advance();
    };
    auto prologue = [&]() {
        stage(0);
        if (nk > 1) stage(1);
        if (nk > 2) stage(2);
    };
    // tiles that may still be in flight when tile t+1 is needed: staged so far (UPTO) minus (t+1)
#define G2_WAIT_NEXT(T, UPTO)                                                        \
    do {                                                                             \
        int upto = (UPTO);                                                           \
        if (upto > nk - 1) upto = nk - 1;                                            \
        const int n_out = upto - ((T) + 1);                                          \
        if (n_out >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GL) : "memory");   \
        else if (n_out == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GL) : "memory");  \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        \
    } while (0)

    int tile = xcd_remap(blockIdx.x, gridDim.x);        // neighbouring tiles on one XCD (shared L2)
    if (tile < ntiles) {
        set_tile(tile);
        prologue();
    }
    for (; tile < ntiles; tile += gridDim.x) {
        const int cm0 = m0, cn0 = n0;
        f32x4 acc[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
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
            const unsigned char* A = smem + (t & 3) * 32768 + wr * (MI * 1024) + lane_off;
            const unsigned char* B = smem + (t & 3) * 32768 + 16384 + wc * (NJ * 1024) + lane_off;
            bf16x8_t fa[MI], fb[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(B + j * 1024);
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(A + i * 1024);
            // RH of the GL pieces of tile t+3 are issued HERE, in the READ phase, where their issue stalls overlap the
            // partner wave's MFMAs (all GL here made this phase longer than the MFMA phase it hides behind; all GL in
            // the MFMA phase leave the matrix pipe idle while this wave stalls on them).  The buffer was last read in
            // R(t-1): retired by the lgkmcnt(0) both groups execute before the barrier that ends their READ phase.
            const bool rh_now = RH > 0 && t + 3 < nk;
            if (rh_now) {
#pragma unroll
                for (int q = 0; q < RH; ++q) stage_piece(t + 3, q);
            }
            // wr=1 waits for tile t+1 here, BEFORE it restages in its M phase: staged so far = through t+2
            // plus the RH early pieces of t+3
            if (wr == 1) {
                if (rh_now) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GL + RH) : "memory");
                else G2_WAIT_NEXT(t, t + 2);
            }
            if (RH > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < NJ; ++j)      // operands swapped: D[n][m], 4 consecutive n per lane
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[0], acc[0][j], 0, 0, 0);
            // the restage rides in the MFMA phase: an LDS-DMA piece costs 60-180 issue cycles, which made the
            // READ phase (12 ds_read_b128 + 4 pieces) twice as long as the 512-cycle MFMA phase it overlaps.  All pieces
            // in ONE block: one piece between every two fragment rows measured 24 % SLOWER (4096^3: 1172 -> 896 TF).
            if (t + 3 < nk) {
#pragma unroll
                for (int q = RH; q < GL; ++q) stage_piece(t + 3, q);
                stage_advance();
            }
#pragma unroll
            for (int i = 1; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            if (wr == 0) G2_WAIT_NEXT(t, t + 3);
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
                    const int om = cm0 + wr * (16 * MI) + i * 16 + rr + 8 * h;
                    if (om < a.M && col_ok) dst[h] = *reinterpret_cast<const u32x4*>(a.residual + (int64_t)om * a.N + on);
                }
            };
            if (a.residual) load_res(0, rres);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = cm0 + wr * (16 * MI) + i * 16 + frow;
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
                if (a.residual && i + 1 < MI) load_res(i + 1, nres);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int om = cm0 + wr * (16 * MI) + i * 16 + rr + 8 * h;
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
                const int blk = (cm0 / BM) * 2 + wr;
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
            for (int i = 0; i < MI; ++i) {
                const int m = cm0 + wr * (16 * MI) + i * 16 + frow;
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
