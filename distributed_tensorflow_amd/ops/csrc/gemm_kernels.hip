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

#include "gemm256p.h"

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
                for (int k = 0; k < 8; ++k) v[k] = relu_nan(v[k]);
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
                if (a.act == ACT_RELU) x = relu_nan(x);
                else if (a.act == ACT_GELU) x = gelu_erf(x);
                else if (a.act == ACT_GELU_TANH) x = gelu_tanh_fwd(x);
                if (a.residual) x += bf16_to_f32(a.residual[o + k]);
                a.Y[o + k] = f32_to_bf16(x);
            }
        }
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
            if (act == ACT_RELU) { lo = relu_nan(lo); hi = relu_nan(hi); }
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
    { static int gm = 0; if (gm <= 0) { const char* e = getenv("DTF_GEMM_GROUP_M"); gm = e ? atoi(e) : 4; if (gm < 1) gm = 1; } a.group_m = gm; }
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
    { static int gm = 0; if (gm <= 0) { const char* e = getenv("DTF_GEMM_GROUP_M"); gm = e ? atoi(e) : 4; if (gm < 1) gm = 1; } a.group_m = gm; }
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
