// K6T: weight-gradient GEMM for gfx950 -- C[N,K] = A[M,N]^T . B[M,K] with M (tokens / pixels) the
// reduction dimension: dW = dZ^T . X of a Linear layer (and of a 1x1 convolution in NHWC).
// Both operands are row-major with the REDUCTION index on the rows, i.e. transposed w.r.t. what an
// MFMA fragment wants (8 consecutive k per lane).  The library GEMM reaches ~515 TFLOP/s on the BERT
// shapes (M = 32768) and is 5x slower than MIOpen on the conv shapes; this kernel feeds the MFMAs
// through the CDNA4 transposing LDS read instead:
//
//   * 256 (n) x 256 (k) fp32 output tile per 512-thread block, 8 waves as 2 x 4, 128 x 64 per wave;
//   * K tiles of 32 reduction rows: [32][256] bf16 images of A and of B (512-B rows) staged by LDS-DMA
//     (global_load_lds, 16 B/lane, full 512-B row segments from HBM), four 32 KB buffers, three tiles
//     in flight, counted vmcnt + raw s_barrier, the two waves of every SIMD ping-ponged one barrier
//     apart -- the pipeline of gemm256_bias_act_kernel (gemm_kernels.hip), see the comments there;
//   * fragments by ds_read_b64_tr_b16: a 16-lane group reads a 4-row x 16-column block and gets it
//     column-major, so two reads give a lane its 8 consecutive reduction rows of one column -- the
//     MFMA A (and B) fragment in natural k order.  The 8 rows a 32-lane half touches per read would
//     all hit the same 32-B bank slot (row stride 512 B); the image is XOR-swizzled at 32-B
//     granularity, chunk' = chunk ^ ((row & 3) | ((row >> 3) & 1) << 2), applied on the per-lane
//     SOURCE address of the LDS-DMA and on the read address;
//   * split over M: grid = tiles x splits (~one block per CU); every block writes its fp32 tile to a
//     slab, a second kernel adds the slabs in fixed order (deterministic) and rounds once to bf16.
// Requirements: M % 32 == 0, N % 8 == 0, K % 8 == 0.
#include "dtf_common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) short tn_s16x8;
typedef __attribute__((ext_vector_type(4))) short tn_s16x4;
typedef __attribute__((address_space(3))) tn_s16x4 tn_lds_s16x4;
typedef __attribute__((address_space(3))) void tn_lds_void;
typedef const __attribute__((address_space(1))) void tn_glb_void;

struct TnArgs {
    const unsigned short* A; const unsigned short* B; float* slab; float* bias_slab;
    int M, N, K; int ntn, ntk, splits, m_per_split;
    // CONV variant (weight gradient of a KH x KW convolution, NHWC): B is the input image [NB, H, W, Cin] and
    // the B operand row of output pixel m = (n, oh, ow) is its im2col patch, gathered by the per-lane LDS-DMA
    // source addresses; K = KH * KW * Cin.  Out-of-image taps and rows >= M read a 16-byte zero page.
    // The m -> (n, oh, ow) decomposition is a per-geometry table (conv_tab_kernel): entry m = { pixel index of the
    // patch origin (n*H + oh*stride - pad_h)*W + ow*stride - pad_w, origin row | origin column << 16 }; a wave
    // fetches the four entries of its four rows per K step with one scalar load.
    const unsigned short* zero;
    const int* tab;
    int Cin, H, W, KW;
};

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef const __attribute__((address_space(4))) i32x4 tn_const_i32x4;

__device__ __forceinline__ int tn_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, pos = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
}

// The transposing reads are inline asm on purpose: through the builtin, hipcc treats every LDS read as a
// possible consumer of the in-flight LDS-DMA and drains vmcnt(0) before the first read of each K step,
// which serialises the whole pipeline (measured: 3.5x slower).  With asm the ordering is ours: counted
// vmcnt + barrier before a tile is read (below), lgkmcnt(0) + sched_barrier before its MFMAs.
__device__ __forceinline__ tn_s16x4 tn_tr_read(unsigned addr) {
    tn_s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ tn_s16x4 tn_tr_read_2k(unsigned addr) {
    tn_s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(v) : "v"(addr));
    return v;
}

// BIAS: also produce the column sums of A (the bias gradient sum_m dZ[m, n]) -- the waves of the first
// column of k tiles multiply their A fragments by an all-ones B fragment (8 extra MFMAs per K step for
// one wave in four of one block in ntk), which saves the separate pass over dZ.
// NI: 16-row A fragments per wave -- the block's n extent is 32 * NI (256, 192, 128 or 64 output rows), chosen on
// the host so that narrow N (Cout of a convolution) does not pay for a 256-row tile.
template <bool BIAS, bool CONV, int NI>
__global__ __launch_bounds__(512) void gemm_tn_kernel(TnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int bid = tn_xcd_remap(blockIdx.x, gridDim.x);
    const int ntiles = a.ntn * a.ntk;
    const int split = bid / ntiles, tile = bid - split * ntiles;
    const int tn = tile / a.ntk, tk = tile - tn * a.ntk;
    const int n0 = tn * (32 * NI), k0 = tk * 256;
    const int m_begin = split * a.m_per_split;
    int m_end = m_begin + a.m_per_split;
    if (m_end > a.M) m_end = a.M;
    const int nk = m_end > m_begin ? (m_end - m_begin + 31) >> 5 : 0;      // plain variant: M % 32 == 0

    // staging: wave w fills row pairs 2w, 2w+1 (1 KB = 2 rows x 512 B per instruction) of both images
    const unsigned short* pa[2];
    const unsigned short* pb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 2 + (lane >> 5);
        const int sx = (row & 3) | (((row >> 3) & 1) << 2);
        const int col = ((((lane & 31) >> 1) ^ sx) << 4) + ((lane & 1) << 3);      // source column (elements)
        int ca = n0 + col, cb = k0 + col;
        if (ca > a.N - 8) ca = a.N - 8;          // clamped columns only feed outputs that are never stored
        if (cb > a.K - 8) cb = a.K - 8;
        pa[i] = a.A + (int64_t)(m_begin + row) * a.N + ca;
        pb[i] = a.B + (int64_t)(m_begin + row) * a.K + cb;
    }
    const int64_t stepA = (int64_t)32 * a.N, stepB = (int64_t)32 * a.K;
    // CONV: per-lane constants of the im2col gather (the lane's k column never changes: k0 is per block)
    int cv_dh[2], cv_dw[2], cv_off[2];
    bool cv_kvalid[2];
    const int half = lane >> 5;
    if (CONV) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 2 + half;
            const int sx = (row & 3) | (((row >> 3) & 1) << 2);
            const int col = ((((lane & 31) >> 1) ^ sx) << 4) + ((lane & 1) << 3);
            const int k = k0 + col;
            cv_kvalid[i] = k < a.K;
            const int tap = k / a.Cin;
            cv_dh[i] = tap / a.KW;
            cv_dw[i] = tap - cv_dh[i] * a.KW;
            cv_off[i] = (cv_dh[i] * a.W + cv_dw[i]) * a.Cin + (k - tap * a.Cin);
        }
    }
    // table entries of the wave's rows 4 w .. 4 w + 3 of reduction tile t: 32 B, one scalar load pair
    auto load_tab = [&](int t, i32x4& e01, i32x4& e23) {
        tn_const_i32x4* tp = (tn_const_i32x4*)(uintptr_t)(a.tab + (int64_t)(m_begin + t * 32 + wave * 4) * 2);
        e01 = tp[0];
        e23 = tp[1];
    };
    auto stage = [&](int t, const i32x4& e01, const i32x4& e23) {
        unsigned char* base = smem + (t & 3) * 32768 + wave * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!CONV) {
                __builtin_amdgcn_global_load_lds((tn_glb_void*)(pa[i] + t * stepA), (tn_lds_void*)(base + i * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((tn_glb_void*)(pb[i] + t * stepB), (tn_lds_void*)(base + 16384 + i * 1024), 16, 0, 0);
            } else {
                const i32x4 e = i == 0 ? e01 : e23;
                const int pix = half ? e[2] : e[0];
                const int org = half ? e[3] : e[1];
                const int ih = (int)(short)(org & 0xffff) + cv_dh[i], iw = (org >> 16) + cv_dw[i];
                const unsigned short* sa = pa[i] + t * stepA;
                const unsigned short* sb = a.B + ((int64_t)pix * a.Cin + cv_off[i]);
                if (!(cv_kvalid[i] && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W)) sb = a.zero;
                if (m_begin + t * 32 + 32 > a.M) {          // uniform: only the last tile of the last split
                    if (m_begin + t * 32 + (wave * 2 + i) * 2 + half >= a.M) { sa = a.zero; sb = a.zero; }
                }
                __builtin_amdgcn_global_load_lds((tn_glb_void*)sa, (tn_lds_void*)(base + i * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((tn_glb_void*)sb, (tn_lds_void*)(base + 16384 + i * 1024), 16, 0, 0);
            }
        }
    };

    f32x4 acc[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 bacc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) bacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = BIAS && wc == 0 && tk == 0;
    tn_s16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;          // bf16 1.0

    // transposed-read addressing: lane = 16 g + 4 q + p supplies row (8 g + 4 h + q), columns 4p .. 4p+3 of
    // the fragment's 16-column chunk c; byte = row * 512 + ((c ^ sx) << 5) + 8 p, sx = q | (g & 1) << 2
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int sx = q | ((g & 1) << 2);
    const unsigned lds0 = (unsigned)(uintptr_t)(tn_lds_void*)smem;
    const int row_off = (int)lds0 + (g * 8 + q) * 512 + p * 8;        // h = 0; h = 1 adds 4 rows = 2048 B
    int offA[NI], offB[4];
#pragma unroll
    for (int i = 0; i < NI; ++i) offA[i] = row_off + (((wr * NI + i) ^ sx) << 5);
#pragma unroll
    for (int j = 0; j < 4; ++j) offB[j] = row_off + 16384 + (((wc * 4 + j) ^ sx) << 5);

#define TN_WAIT_NEXT(T)                                                              \
    do {                                                                             \
        int upto = (T) + 3;                                                          \
        if (upto > nk - 1) upto = nk - 1;                                            \
        const int n_out = upto - ((T) + 1);                                          \
        if (n_out >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             \
        else if (n_out == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        \
    } while (0)
    i32x4 e01 = {0, 0, 0, 0}, e23 = {0, 0, 0, 0};
    if (nk > 0) {
        if (CONV) load_tab(0, e01, e23);
        stage(0, e01, e23);
        if (nk > 1) { if (CONV) load_tab(1, e01, e23); stage(1, e01, e23); }
        if (nk > 2) { if (CONV) load_tab(2, e01, e23); stage(2, e01, e23); }
        if (CONV && nk > 3) load_tab(3, e01, e23);
        if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // stagger the second wave of every SIMD
    for (int t = 0; t < nk; ++t) {
        if (t + 3 < nk) {
            stage(t + 3, e01, e23);
            if (CONV && t + 4 < nk) load_tab(t + 4, e01, e23);       // consumed one iteration later
        }
        const unsigned img = (unsigned)((t & 3) * 32768);
        tn_s16x4 la[NI], ha[NI], lb[4], hb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lb[j] = tn_tr_read(img + offB[j]);
            hb[j] = tn_tr_read_2k(img + offB[j]);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            la[i] = tn_tr_read(img + offA[i]);
            ha[i] = tn_tr_read_2k(img + offA[i]);
        }
        if (wr == 1) TN_WAIT_NEXT(t);
        // the reads are RETIRED before the barrier, not only issued: the other wave group restages this buffer right
        // after it (stage(t + 4) at the top of its next iteration).  With the wait behind the barrier the kernel was
        // correct alone and wrong under load: next to the compute stream's kernels (ops/streams.py) 10-28 of ResNet-50's
        // 1x1 weight gradients differed from run to run by 1e-3 .. 2e-2 (scripts/determinism_check.py)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        tn_s16x8 fa[NI], fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = __builtin_shufflevector(lb[j], hb[j], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int i = 0; i < NI; ++i) fa[i] = __builtin_shufflevector(la[i], ha[i], 0, 1, 2, 3, 4, 5, 6, 7);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        if (BIAS && do_bias) {
#pragma unroll
            for (int i = 0; i < NI; ++i) bacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, bacc[i], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (wr == 0) TN_WAIT_NEXT(t);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
#undef TN_WAIT_NEXT

    // fp32 tile -> slab[split][N][K]; C map: row = 4 g + r (of the 16-row fragment), col = lane & 15
    float* out = a.slab + (int64_t)split * a.N * a.K;
    const int fr = lane & 15;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wr * (16 * NI) + i * 16 + g * 4 + r;
            if (n >= a.N) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kc = k0 + wc * 64 + j * 16 + fr;
                if (kc < a.K) out[(int64_t)n * a.K + kc] = acc[i][j][r];
            }
        }
    if (BIAS && do_bias && fr == 0) {        // every column of bacc holds the row sums; column 0 stores them
        float* bo = a.bias_slab + (int64_t)split * a.N;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wr * (16 * NI) + i * 16 + g * 4 + r;
                if (n < a.N) bo[n] = bacc[i][r];
            }
    }
}

// out[i] = sum_s slab[s][i] in a fixed order, one rounding.  A block covers 64 float4 columns x 4 slab lanes: a lane sums
// the slabs s = lane (mod 4) with eight loads in flight, the four lanes are folded through LDS in lane order (the K14
// reduce).  The first version gave every column ONE thread that walked all the slabs: with few outputs and many splits
// (a 1x1 layer's 64 x 256 filter: 64 blocks, 64 slabs each) that is a chain of dependent-latency loads on a quarter of the
// CUs -- 17-32 us per call, 2.3 % of a ResNet-50 step over its 37 calls.
template <int ODT>
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ slab, int splits, int64_t n4,
                                                        int64_t stride, void* __restrict__ out) {
    const int64_t col = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int sl = threadIdx.x >> 6;
    const int64_t stride4 = stride >> 2;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (col < n4) {
        const f32x4* p = reinterpret_cast<const f32x4*>(slab) + col;
        int b = sl;
        for (; b + 28 < splits; b += 32) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(b + 4 * u) * stride4];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < splits; b += 4) s += p[(int64_t)b * stride4];
    }
    __shared__ f32x4 red[4][64];
    red[sl][threadIdx.x & 63] = s;
    __syncthreads();
    if (sl != 0 || col >= n4) return;
    s = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    if (ODT == DTF_BF16) {
        u32x2 r;
        r[0] = pack_bf16x2(s[0], s[1]);
        r[1] = pack_bf16x2(s[2], s[3]);
        *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(out) + col * 4) = r;
    } else {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + col * 4) = s;
    }
}

// n extent of the block tile: 32 * NI rows, NI in {8, 6, 4, 2}, from a two-term cost model (microseconds):
//   main loop : K steps per block x rounds over the CUs x (NI + 1.5) x 0.082 -- NI fragments of MFMA work plus a fixed
//               ~1.5 fragments for the B reads / staging that do not shrink with NI (0.78 us per step at NI = 8);
//   slabs     : every block writes its fp32 tile and the reduce kernel reads it back: blocks x 32 NI x 256 x 4 B x 2 at
//               ~4.5 TB/s.  With few output tiles (a 1x1 convolution's 1024 x 256 filter is 4 tiles -> 64 splits) the
//               slabs, not the MFMAs, are most of the time, and a narrower tile halves them (measured 48 -> 37 us).
static inline void tn_splits(int mt, int ntiles, int* splits, int* per) {
    int s = 256 / ntiles;
    if (s < 1) s = 1;
    if (s > mt / 4) s = mt / 4 > 0 ? mt / 4 : 1;          // at least 4 reduction tiles per split
    *per = (mt + s - 1) / s;
    *splits = (mt + *per - 1) / *per;
}

// CONV: a K step costs ~1.05 us whatever NI (measured 256ch 3x3 at 14x14: NI = 8 and NI = 4 both 1.04-1.09 us per
// step -- the im2col addressing and gather, not the MFMAs, set it), so more, narrower tiles only add steps: there NI
// is picked to waste the fewest output rows.
static inline int tn_pick_ni(int M, int N, int K, bool conv) {
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("DTF_TN_NI");
        forced = e ? atoi(e) : 0;
    }
    if (forced == 2 || forced == 4 || forced == 6 || forced == 8) return forced;
    int best = 8;
    float best_cost = 1e30f;
    if (conv) {
        for (int ni = 8; ni >= 2; ni -= 2) {
            const float cost = (float)((N + 32 * ni - 1) / (32 * ni)) * ((float)ni + 1.5f);
            if (cost < best_cost - 1e-3f) { best_cost = cost; best = ni; }
        }
        return best;
    }
    const int mt = (M + 31) / 32, ntk = (K + 255) / 256;
    for (int ni = 8; ni >= 2; ni -= 2) {
        const int ntiles = ((N + 32 * ni - 1) / (32 * ni)) * ntk;
        int splits, per;
        tn_splits(mt, ntiles, &splits, &per);
        const int64_t blocks = (int64_t)ntiles * splits;
        const float rounds = (float)((blocks + 255) / 256);
        const float frag = (float)ni + 1.5f < 5.0f ? 5.0f : (float)ni + 1.5f;      // NI = 2 is LDS-read-bound: no cheaper than NI = 4
        const float loop = (float)per * rounds * frag * 0.082f;
        const float slab = (float)blocks * (float)ni * (32.f * 256.f * 4.f * 2.f / 4.5e6f);
        const float cost = loop + slab;
        if (cost < best_cost * 0.93f) { best_cost = cost; best = ni; }
    }
    return best;
}

static inline void tn_plan(int M, int N, int K, int* ni, int* ntn, int* ntk, int* splits, int* mps, bool conv = false) {
    *ni = tn_pick_ni(M, N, K, conv);
    *ntn = (N + 32 * *ni - 1) / (32 * *ni);
    *ntk = (K + 255) / 256;
    int per;
    tn_splits((M + 31) / 32, *ntn * *ntk, splits, &per);
    *mps = per * 32;
}

// table of the CONV variant (see TnArgs); entries M .. roundup(M, 32) - 1 never pass the bounds test
__global__ __launch_bounds__(256) void conv_tab_kernel(int* __restrict__ tab, int M, int Mpad, int H, int W, int OH, int OW,
                                                       int stride, int pad_h, int pad_w) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= Mpad) return;
    int pix = 0, org = (int)0x80008000u;
    if (m < M) {
        const int n = m / (OH * OW), r = m - n * (OH * OW);
        const int oh = r / OW, ow = r - oh * OW;
        const int ih0 = oh * stride - pad_h, iw0 = ow * stride - pad_w;
        pix = (n * H + ih0) * W + iw0;
        org = (ih0 & 0xffff) | (iw0 << 16);
    }
    tab[2 * m] = pix;
    tab[2 * m + 1] = org;
}

template <bool BIAS, bool CONV>
static void tn_launch(const TnArgs& a, int ni, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<BIAS, CONV, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<BIAS, CONV, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<BIAS, CONV, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<BIAS, CONV, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        attr_set = true;
    }
    const dim3 grid(a.ntn * a.ntk * a.splits), block(512);
    switch (ni) {
    case 8: hipLaunchKernelGGL((gemm_tn_kernel<BIAS, CONV, 8>), grid, block, 131072, stream, a); break;
    case 6: hipLaunchKernelGGL((gemm_tn_kernel<BIAS, CONV, 6>), grid, block, 131072, stream, a); break;
    case 4: hipLaunchKernelGGL((gemm_tn_kernel<BIAS, CONV, 4>), grid, block, 131072, stream, a); break;
    default: hipLaunchKernelGGL((gemm_tn_kernel<BIAS, CONV, 2>), grid, block, 131072, stream, a); break;
    }
}

static void tn_reduce_launch(const float* slab, int splits, int64_t n, void* out, int out_dtype, hipStream_t stream) {
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 63) / 64;
    if (blocks < 1) blocks = 1;
    if (out_dtype == DTF_BF16)
        hipLaunchKernelGGL((tn_reduce_kernel<DTF_BF16>), dim3((unsigned)blocks), dim3(256), 0, stream, slab, splits, n4, n, out);
    else
        hipLaunchKernelGGL((tn_reduce_kernel<DTF_F32>), dim3((unsigned)blocks), dim3(256), 0, stream, slab, splits, n4, n, out);
}

extern "C" {

// fp32 workspace (floats) for dtf_gemm_tn: the dW slabs, then the bias slabs
int64_t dtf_gemm_tn_workspace(int M, int N, int K) {
    int ni, ntn, ntk, splits, mps;
    tn_plan(M, N, K, &ni, &ntn, &ntk, &splits, &mps);
    return (int64_t)splits * N * K + (int64_t)splits * N;
}

// C[N,K] = A[M,N]^T . B[M,K]; A, B bf16 row-major; C bf16 (out_dtype 1) or fp32 (0)
// dbias (optional, [N], bias_dtype): column sums of A
int dtf_gemm_tn(const void* A, const void* B, void* C, int out_dtype, void* dbias, int bias_dtype, float* workspace,
                int M, int N, int K, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    if ((M & 31) || (N & 7) || (K & 7)) return -60;
    if ((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C) | ((uintptr_t)workspace)) & 15) return -16;
    TnArgs a{};
    int ni;
    a.A = (const unsigned short*)A; a.B = (const unsigned short*)B; a.slab = workspace; a.M = M; a.N = N; a.K = K;
    tn_plan(M, N, K, &ni, &a.ntn, &a.ntk, &a.splits, &a.m_per_split);
    a.bias_slab = workspace + (int64_t)a.splits * N * K;
    if (dbias) tn_launch<true, false>(a, ni, stream);
    else tn_launch<false, false>(a, ni, stream);
    tn_reduce_launch(workspace, a.splits, (int64_t)N * K, C, out_dtype, stream);
    if (dbias) tn_reduce_launch(a.bias_slab, a.splits, (int64_t)N, dbias, bias_dtype, stream);
    DTF_CHECK_LAUNCH();
    return 0;
}

// fp32 workspace (floats) for dtf_conv_wrw
int64_t dtf_conv_wrw_workspace(int64_t M, int Cout, int Ktot) {
    int ni, ntn, ntk, splits, mps;
    tn_plan((int)M, Cout, Ktot, &ni, &ntn, &ntk, &splits, &mps, true);
    return (int64_t)splits * Cout * Ktot;
}

// ints of the geometry table of dtf_conv_wrw (2 per output pixel, padded to whole reduction tiles)
int64_t dtf_conv_wrw_table_ints(int64_t M) { return ((M + 31) / 32 * 32) * 2; }

// fills the table; it depends on the geometry only, so the caller builds it once per layer shape
int dtf_conv_wrw_table(void* tab, int NB, int H, int W, int OH, int OW, int stride, int pad_h, int pad_w, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    const int64_t M64 = (int64_t)NB * OH * OW;
    if (M64 <= 0 || M64 >= (1 << 24) || (int64_t)NB * H * W >= (1ll << 31) || H >= 32768 || W >= 32768 || stride < 1) return -61;
    if (((uintptr_t)tab) & 31) return -16;
    const int M = (int)M64, Mpad = (M + 31) / 32 * 32;
    hipLaunchKernelGGL(conv_tab_kernel, dim3((Mpad + 255) / 256), dim3(256), 0, stream, (int*)tab, M, Mpad, H, W, OH, OW, stride,
                       pad_h, pad_w);
    DTF_CHECK_LAUNCH();
    return 0;
}

// Weight gradient of a 2-D convolution, NHWC bf16, groups = 1, dilation = 1:
//   dW[Cout][KH][KW][Cin] = sum_{n, oh, ow} dY[n, oh, ow, Cout] * X[n, oh*stride - pad_h + kh, ow*stride - pad_w + kw, Cin]
// dy: [NB, OH, OW, Cout]; x: [NB, H, W, Cin]; zero: >= 16 bytes of zeros in device memory; tab: dtf_conv_wrw_table
// of the same geometry.
int dtf_conv_wrw(const void* dy, const void* x, void* dw, int out_dtype, float* workspace, const void* zero, const void* tab,
                 int NB, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    const int64_t M64 = (int64_t)NB * OH * OW;
    const int Ktot = KH * KW * Cin;
    if (M64 <= 0 || M64 >= (1 << 24) || (Cin & 7) || (Cout & 7)) return -61;
    if ((((uintptr_t)dy) | ((uintptr_t)x) | ((uintptr_t)dw) | ((uintptr_t)workspace) | ((uintptr_t)zero) | ((uintptr_t)tab)) & 15)
        return -16;
    const int M = (int)M64;
    TnArgs a{};
    int ni;
    a.A = (const unsigned short*)dy; a.B = (const unsigned short*)x; a.slab = workspace; a.M = M; a.N = Cout; a.K = Ktot;
    tn_plan(M, Cout, Ktot, &ni, &a.ntn, &a.ntk, &a.splits, &a.m_per_split, true);
    a.zero = (const unsigned short*)zero;
    a.tab = (const int*)tab;
    a.Cin = Cin; a.H = H; a.W = W; a.KW = KW;
    tn_launch<false, true>(a, ni, stream);
    tn_reduce_launch(workspace, a.splits, (int64_t)Cout * Ktot, dw, out_dtype, stream);
    DTF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
