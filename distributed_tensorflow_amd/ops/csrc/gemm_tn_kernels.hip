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
    const unsigned short* zero;
    int Cin, H, W, OH, OW, KW, stride, pad_h, pad_w;
    float inv_ohw, inv_ow;
};

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
template <bool BIAS, bool CONV>
__global__ __launch_bounds__(512) void gemm_tn_kernel(TnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int bid = tn_xcd_remap(blockIdx.x, gridDim.x);
    const int ntiles = a.ntn * a.ntk;
    const int split = bid / ntiles, tile = bid - split * ntiles;
    const int tn = tile / a.ntk, tk = tile - tn * a.ntk;
    const int n0 = tn * 256, k0 = tk * 256;
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
    int cv_row[2], cv_c[2], cv_dh[2], cv_dw[2], cv_ca[2];
    bool cv_kvalid[2];
    if (CONV) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 2 + (lane >> 5);
            const int sx = (row & 3) | (((row >> 3) & 1) << 2);
            const int col = ((((lane & 31) >> 1) ^ sx) << 4) + ((lane & 1) << 3);
            const int k = k0 + col;
            cv_kvalid[i] = k < a.K;
            const int tap = k / a.Cin;
            cv_c[i] = k - tap * a.Cin;
            cv_dh[i] = tap / a.KW;
            cv_dw[i] = tap - cv_dh[i] * a.KW;
            cv_row[i] = row;
            int ca = n0 + col;
            if (ca > a.N - 8) ca = a.N - 8;
            cv_ca[i] = ca;
        }
    }
    auto stage = [&](int t) {
        unsigned char* base = smem + (t & 3) * 32768 + wave * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!CONV) {
                __builtin_amdgcn_global_load_lds((tn_glb_void*)(pa[i] + t * stepA), (tn_lds_void*)(base + i * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((tn_glb_void*)(pb[i] + t * stepB), (tn_lds_void*)(base + 16384 + i * 1024), 16, 0, 0);
            } else {
                const int m = m_begin + t * 32 + cv_row[i];
                const unsigned short* sa = a.zero;
                const unsigned short* sb = a.zero;
                if (m < a.M) {
                    sa = a.A + (int64_t)m * a.N + cv_ca[i];
                    // m -> (n, oh, ow): float reciprocal + one correction step (m < 2^24 checked on the host)
                    const int ohw = a.OH * a.OW;
                    int n = (int)((float)m * a.inv_ohw);
                    int r = m - n * ohw;
                    if (r < 0) { --n; r += ohw; } else if (r >= ohw) { ++n; r -= ohw; }
                    int oh = (int)((float)r * a.inv_ow);
                    int ow = r - oh * a.OW;
                    if (ow < 0) { --oh; ow += a.OW; } else if (ow >= a.OW) { ++oh; ow -= a.OW; }
                    const int ih = oh * a.stride - a.pad_h + cv_dh[i], iw = ow * a.stride - a.pad_w + cv_dw[i];
                    if (cv_kvalid[i] && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W)
                        sb = a.B + (((int64_t)n * a.H + ih) * a.W + iw) * a.Cin + cv_c[i];
                }
                __builtin_amdgcn_global_load_lds((tn_glb_void*)sa, (tn_lds_void*)(base + i * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((tn_glb_void*)sb, (tn_lds_void*)(base + 16384 + i * 1024), 16, 0, 0);
            }
        }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 bacc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
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
    int offA[8], offB[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) offA[i] = row_off + (((wr * 8 + i) ^ sx) << 5);
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
    if (nk > 0) {
        stage(0);
        if (nk > 1) stage(1);
        if (nk > 2) stage(2);
        if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // stagger the second wave of every SIMD
    for (int t = 0; t < nk; ++t) {
        if (t + 3 < nk) stage(t + 3);
        const unsigned img = (unsigned)((t & 3) * 32768);
        tn_s16x4 la[8], ha[8], lb[4], hb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lb[j] = tn_tr_read(img + offB[j]);
            hb[j] = tn_tr_read_2k(img + offB[j]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            la[i] = tn_tr_read(img + offA[i]);
            ha[i] = tn_tr_read_2k(img + offA[i]);
        }
        if (wr == 1) TN_WAIT_NEXT(t);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        tn_s16x8 fa[8], fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = __builtin_shufflevector(lb[j], hb[j], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = __builtin_shufflevector(la[i], ha[i], 0, 1, 2, 3, 4, 5, 6, 7);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        if (BIAS && do_bias) {
#pragma unroll
            for (int i = 0; i < 8; ++i) bacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, bacc[i], 0, 0, 0);
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
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wr * 128 + i * 16 + g * 4 + r;
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
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wr * 128 + i * 16 + g * 4 + r;
                if (n < a.N) bo[n] = bacc[i][r];
            }
    }
}

// out[i] = sum_s slab[s][i] (fixed order), one rounding; 4 elements per thread
template <int ODT>
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ slab, int splits, int64_t n4,
                                                        int64_t stride, void* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 s = *reinterpret_cast<const f32x4*>(slab + i * 4);
        for (int k = 1; k < splits; ++k) s += *reinterpret_cast<const f32x4*>(slab + k * stride + i * 4);
        if (ODT == DTF_BF16) {
            u32x2 r;
            r[0] = pack_bf16x2(s[0], s[1]);
            r[1] = pack_bf16x2(s[2], s[3]);
            *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(out) + i * 4) = r;
        } else {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + i * 4) = s;
        }
    }
}

static inline void tn_plan(int M, int N, int K, int* ntn, int* ntk, int* splits, int* mps) {
    *ntn = (N + 255) / 256;
    *ntk = (K + 255) / 256;
    const int ntiles = *ntn * *ntk;
    int s = 256 / ntiles;
    if (s < 1) s = 1;
    const int mt = (M + 31) / 32;                // reduction tiles
    if (s > mt / 4) s = mt / 4 > 0 ? mt / 4 : 1; // at least 4 tiles per split
    const int per = (mt + s - 1) / s;
    *mps = per * 32;
    *splits = (mt + per - 1) / per;
}

extern "C" {

// fp32 workspace (floats) for dtf_gemm_tn: the dW slabs, then the bias slabs
int64_t dtf_gemm_tn_workspace(int M, int N, int K) {
    int ntn, ntk, splits, mps;
    tn_plan(M, N, K, &ntn, &ntk, &splits, &mps);
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
    a.A = (const unsigned short*)A; a.B = (const unsigned short*)B; a.slab = workspace; a.M = M; a.N = N; a.K = K;
    tn_plan(M, N, K, &a.ntn, &a.ntk, &a.splits, &a.m_per_split);
    a.bias_slab = workspace + (int64_t)a.splits * N * K;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm_tn_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        hipFuncSetAttribute((const void*)gemm_tn_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        attr_set = true;
    }
    if (dbias)
        hipLaunchKernelGGL((gemm_tn_kernel<true, false>), dim3(a.ntn * a.ntk * a.splits), dim3(512), 131072, stream, a);
    else
        hipLaunchKernelGGL((gemm_tn_kernel<false, false>), dim3(a.ntn * a.ntk * a.splits), dim3(512), 131072, stream, a);
    const int64_t n4 = (int64_t)N * K / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (out_dtype == DTF_BF16)
        hipLaunchKernelGGL((tn_reduce_kernel<DTF_BF16>), dim3((unsigned)blocks), dim3(256), 0, stream, workspace, a.splits,
                           n4, (int64_t)N * K, C);
    else
        hipLaunchKernelGGL((tn_reduce_kernel<DTF_F32>), dim3((unsigned)blocks), dim3(256), 0, stream, workspace, a.splits,
                           n4, (int64_t)N * K, C);
    if (dbias) {
        const int64_t b4 = N / 4;
        if (bias_dtype == DTF_BF16)
            hipLaunchKernelGGL((tn_reduce_kernel<DTF_BF16>), dim3((unsigned)((b4 + 255) / 256)), dim3(256), 0, stream,
                               a.bias_slab, a.splits, b4, (int64_t)N, dbias);
        else
            hipLaunchKernelGGL((tn_reduce_kernel<DTF_F32>), dim3((unsigned)((b4 + 255) / 256)), dim3(256), 0, stream,
                               a.bias_slab, a.splits, b4, (int64_t)N, dbias);
    }
    DTF_CHECK_LAUNCH();
    return 0;
}

// fp32 workspace (floats) for dtf_conv_wrw
int64_t dtf_conv_wrw_workspace(int64_t M, int Cout, int Ktot) {
    int ntn, ntk, splits, mps;
    tn_plan((int)M, Cout, Ktot, &ntn, &ntk, &splits, &mps);
    return (int64_t)splits * Cout * Ktot;
}

// Weight gradient of a 2-D convolution, NHWC bf16, groups = 1, dilation = 1:
//   dW[Cout][KH][KW][Cin] = sum_{n, oh, ow} dY[n, oh, ow, Cout] * X[n, oh*stride - pad_h + kh, ow*stride - pad_w + kw, Cin]
// dy: [NB, OH, OW, Cout]; x: [NB, H, W, Cin]; zero: >= 16 bytes of zeros in device memory.
int dtf_conv_wrw(const void* dy, const void* x, void* dw, int out_dtype, float* workspace, const void* zero, int NB, int H,
                 int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int pad_h, int pad_w,
                 hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    const int64_t M64 = (int64_t)NB * OH * OW;
    const int Ktot = KH * KW * Cin;
    if (M64 <= 0 || M64 >= (1 << 24) || (Cin & 7) || (Cout & 7) || stride < 1) return -61;
    if ((((uintptr_t)dy) | ((uintptr_t)x) | ((uintptr_t)dw) | ((uintptr_t)workspace) | ((uintptr_t)zero)) & 15) return -16;
    const int M = (int)M64;
    TnArgs a{};
    a.A = (const unsigned short*)dy; a.B = (const unsigned short*)x; a.slab = workspace; a.M = M; a.N = Cout; a.K = Ktot;
    tn_plan(M, Cout, Ktot, &a.ntn, &a.ntk, &a.splits, &a.m_per_split);
    a.zero = (const unsigned short*)zero;
    a.Cin = Cin; a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.KW = KW; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w;
    a.inv_ohw = 1.0f / (float)(OH * OW);
    a.inv_ow = 1.0f / (float)OW;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm_tn_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_tn_kernel<false, true>), dim3(a.ntn * a.ntk * a.splits), dim3(512), 131072, stream, a);
    const int64_t n4 = (int64_t)Cout * Ktot / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (out_dtype == DTF_BF16)
        hipLaunchKernelGGL((tn_reduce_kernel<DTF_BF16>), dim3((unsigned)blocks), dim3(256), 0, stream, workspace, a.splits,
                           n4, (int64_t)Cout * Ktot, dw);
    else
        hipLaunchKernelGGL((tn_reduce_kernel<DTF_F32>), dim3((unsigned)blocks), dim3(256), 0, stream, workspace, a.splits,
                           n4, (int64_t)Cout * Ktot, dw);
    DTF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
