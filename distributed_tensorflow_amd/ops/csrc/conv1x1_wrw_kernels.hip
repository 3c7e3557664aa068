// K19: streaming weight gradient of the narrow 1x1 convolutions of ResNet's 56 x 56 stage (gfx950).
//
//   C[256][64] = A[M][256]^T . B[M][64]      M = N*H*W pixels (802 816 at batch 256), bf16 rows, fp32 result
//
// i.e. dW of a 64 -> 256-channel layer (A = dY, B = X: C = dW) and of a 256 -> 64-channel layer (A = X, B = dY:
// C = dW^T, stored transposed by the slab reduction).  514 MB in, 32 KB out, 26 GFLOP: a read stream, not a GEMM.
// MIOpen runs these through a CK batched GEMM + workspace fill at 106-157 us per call (3.3-4.8 TB/s; six calls per
// ResNet-50 step); K6T (gemm_tn_kernels.hip) loses there because its 256 x 256 output tile stages and multiplies a
// 256-column B image of which 64 columns exist.  Same ingredients, shaped for this problem:
//
//   * persistent 512-thread blocks, one per CU; block b streams pixels [b * mpb, (b + 1) * mpb) in 64-pixel tiles;
//   * a tile = A image [64][256] (512-B rows, 32 KB) + B image [64][64] (128-B rows, 8 KB), brought in by LDS-DMA
//     (global_load_lds, 16 B per lane: every wave issues 4 + 1 instructions per tile), three 40 KB buffers, two tiles
//     in flight, counted vmcnt + ONE raw s_barrier per tile;
//   * fragments by ds_read_b64_tr_b16 (the pixels are the MFMA k index, the images are pixel-major): both images are
//     XOR-swizzled at 32-B granularity on the LDS-DMA SOURCE address so that the eight rows a 32-lane half reads land
//     in eight different bank slots -- A: chunk ^ ((row & 3) | ((row >> 3) & 1) << 2) (K6T's), B (four chunks per
//     row, two rows per bank row): chunk ^ (((row >> 1) & 1) | ((row >> 3) & 1) << 1);
//   * wave w owns output rows 32 w .. 32 w + 31 and all 64 columns: 2 x 4 accumulators of 16 x 16 (32 VGPRs), 16 MFMAs
//     per tile -- the matrix pipe idles, as it should;
//   * every block writes its fp32 tile to a slab; wrw1x1_reduce_kernel adds the slabs in a fixed order (deterministic)
//     with 256 blocks x 16 slab lanes, rounds once to bf16 and stores C or C^T.
// Requirements: M % 64 == 0, 16-byte aligned operands.
#include "dtf_common.h"

typedef __attribute__((ext_vector_type(8))) short w1_s16x8;
typedef __attribute__((ext_vector_type(4))) short w1_s16x4;
typedef __attribute__((address_space(3))) void w1_lds_void;
typedef const __attribute__((address_space(1))) void w1_glb_void;

struct Wrw1Args {
    const unsigned short* A;   // [M][256]
    const unsigned short* B;   // [M][64]
    float* slab;               // [blocks][256][64]
    int64_t M;
    int tiles_per_block;       // 64-pixel tiles per block (the last blocks may get fewer / none)
};

#define W1_TILE_BYTES 40960
#define W1_B_OFF 32768
// LDS buffers: NBUF - 1 tiles in flight.  3 buffers (80 KB in flight at most, 40 KB right after a tile lands) measured
// 3.25 TB/s; 4 buffers = the whole 160 KB LDS of the CU: see profiles/r3_k19_wrw1x1.txt
#ifndef W1_NBUF
#define W1_NBUF 4
#endif

// inline asm on purpose (see gemm_tn_kernels.hip): through the builtin hipcc drains vmcnt(0) before every LDS read
// while an LDS-DMA is in flight
__device__ __forceinline__ w1_s16x4 w1_tr(unsigned addr) {
    w1_s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

__global__ __launch_bounds__(512) void wrw1x1_kernel(Wrw1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t ntiles_all = a.M >> 6;
    const int64_t t_begin = (int64_t)blockIdx.x * a.tiles_per_block;
    int64_t t_end = t_begin + a.tiles_per_block;
    if (t_end > ntiles_all) t_end = ntiles_all;
    const int nk = t_end > t_begin ? (int)(t_end - t_begin) : 0;

    // ---- staging sources.  A: instruction i (0..3) of wave w fills rows (4w + i) * 2 + (lane >> 5), 512 B each
    const unsigned short* pa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 2 + (lane >> 5);
        const int sx = (row & 3) | (((row >> 3) & 1) << 2);
        const int col = ((((lane & 31) >> 1) ^ sx) << 4) + ((lane & 1) << 3);
        pa[i] = a.A + ((t_begin << 6) + row) * 256 + col;
    }
    // B: the one instruction of wave w fills rows 8 w + (lane >> 3), 128 B each
    const unsigned short* pb;
    {
        const int row = wave * 8 + (lane >> 3);
        const int f = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
        const int col = ((((lane & 7) >> 1) ^ f) << 4) + ((lane & 1) << 3);
        pb = a.B + ((t_begin << 6) + row) * 64 + col;
    }
    auto stage = [&](int t) {
        unsigned char* base = smem + (t % W1_NBUF) * W1_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((w1_glb_void*)(pa[i] + (int64_t)t * (64 * 256)),
                                             (w1_lds_void*)(base + (wave * 4 + i) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((w1_glb_void*)(pb + (int64_t)t * (64 * 64)),
                                         (w1_lds_void*)(base + W1_B_OFF + wave * 1024), 16, 0, 0);
    };

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- transposed-read addresses: lane = 16 g + 4 q + p supplies row 8 g + 4 h + q (+ 32 for the second k step),
    // columns 4 p .. 4 p + 3 of a 16-column chunk c
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const unsigned lds0 = (unsigned)(uintptr_t)(w1_lds_void*)smem;
    const int sxa = q | ((g & 1) << 2);
    const int fb = (q >> 1) | ((g & 1) << 1);
    unsigned offA[2], offB[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) offA[i] = lds0 + (g * 8 + q) * 512 + p * 8 + (((wave * 2 + i) ^ sxa) << 5);
#pragma unroll
    for (int j = 0; j < 4; ++j) offB[j] = lds0 + W1_B_OFF + (g * 8 + q) * 128 + p * 8 + ((j ^ fb) << 5);

#pragma unroll
    for (int t0 = 0; t0 < W1_NBUF - 1; ++t0)
        if (t0 < nk) stage(t0);
    for (int t = 0; t < nk; ++t) {
        // this wave's five loads of tile t have landed (the tiles behind it -- up to NBUF - 2 -- may still fly) ...
        const int behind = nk - 1 - t < W1_NBUF - 2 ? nk - 1 - t : W1_NBUF - 2;
        if (behind >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (behind == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ... and after the barrier everybody's have.  Every wave has also finished READING tile t - 1 (lgkmcnt(0)
        // below precedes its arrival here), whose buffer tile t + NBUF - 1 is staged into next.
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (t + W1_NBUF - 1 < nk) stage(t + W1_NBUF - 1);
        const unsigned img = (unsigned)((t % W1_NBUF) * W1_TILE_BYTES);
        w1_s16x4 la[2][2], ha[2][2], lb[2][4], hb[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lb[ks][j] = w1_tr(img + offB[j] + ks * (32 * 128));
                hb[ks][j] = w1_tr(img + offB[j] + ks * (32 * 128) + 4 * 128);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                la[ks][i] = w1_tr(img + offA[i] + ks * (32 * 512));
                ha[ks][i] = w1_tr(img + offA[i] + ks * (32 * 512) + 4 * 512);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            w1_s16x8 fa[2], fbv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) fbv[j] = __builtin_shufflevector(lb[ks][j], hb[ks][j], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = __builtin_shufflevector(la[ks][i], ha[ks][i], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fbv[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // fp32 tile -> slab[block][256][64]; C map of the 16 x 16 accumulator: row = 4 g + r, column = lane & 15
    float* out = a.slab + (int64_t)blockIdx.x * (256 * 64);
    const int fr = lane & 15;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = wave * 32 + i * 16 + g * 4 + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) out[n * 64 + j * 16 + fr] = acc[i][j][r];
        }
}

// out = sum over the slabs in a fixed order, one rounding to bf16.  256 blocks; a block owns 16 float4 columns of the
// 4096 (= 256 x 64 / 4), its 256 threads are 16 columns x 16 slab lanes: lane l sums slabs l, l + 16, ... (eight loads in
// flight), the 16 lanes are folded through LDS in lane order.  TRANS: C is [256][64] = dW^T of a 256 -> 64-channel layer;
// store out[c][r] = C[r][c].
template <bool TRANS>
__global__ __launch_bounds__(256) void wrw1x1_reduce_kernel(const float* __restrict__ slab, int nslab,
                                                            unsigned short* __restrict__ out) {
    const int c16 = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int col4 = blockIdx.x * 16 + c16;                       // float4 index into the [256][64] tile
    const f32x4* p = reinterpret_cast<const f32x4*>(slab) + col4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int b = sl;
    for (; b + 112 < nslab; b += 128) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(b + 16 * u) * 4096];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < nslab; b += 16) s += p[(int64_t)b * 4096];
    __shared__ f32x4 red[16][16];
    red[sl][c16] = s;
    __syncthreads();
    if (sl != 0) return;
    f32x4 tot = red[0][c16];
#pragma unroll
    for (int k = 1; k < 16; ++k) tot += red[k][c16];
    const int e = col4 * 4, r = e >> 6, c = e & 63;               // element (r, c .. c + 3) of C
    if (!TRANS) {
        u32x2 w;
        w[0] = pack_bf16x2(tot[0], tot[1]);
        w[1] = pack_bf16x2(tot[2], tot[3]);
        *reinterpret_cast<u32x2*>(out + e) = w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) out[(c + k) * 256 + r] = f32_to_bf16(tot[k]);
    }
}

extern "C" {

int dtf_wrw1x1_ok(int64_t M) { return M >= 64 * 256 && (M & 63) == 0; }

int64_t dtf_wrw1x1_workspace_floats(int64_t M) { (void)M; return (int64_t)256 * 256 * 64; }

// A [M][256], B [M][64] bf16; out bf16: [256][64] (trans = 0) or [64][256] (trans = 1); ws: dtf_wrw1x1_workspace_floats
int dtf_wrw1x1(const void* A, const void* B, void* out, void* ws, int64_t M, int trans, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (!dtf_wrw1x1_ok(M)) return -2;
    if ((((uintptr_t)A | (uintptr_t)B | (uintptr_t)out | (uintptr_t)ws) & 15) != 0) return -16;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)wrw1x1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           W1_NBUF * W1_TILE_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int64_t ntiles = M >> 6;
    const int blocks = 256;
    Wrw1Args a;
    a.A = (const unsigned short*)A;
    a.B = (const unsigned short*)B;
    a.slab = (float*)ws;
    a.M = M;
    a.tiles_per_block = (int)((ntiles + blocks - 1) / blocks);
    hipLaunchKernelGGL(wrw1x1_kernel, dim3(blocks), dim3(512), W1_NBUF * W1_TILE_BYTES, stream, a);
    DTF_CHECK_LAUNCH();
    if (trans)
        hipLaunchKernelGGL(wrw1x1_reduce_kernel<true>, dim3(256), dim3(256), 0, stream, (const float*)ws, blocks,
                           (unsigned short*)out);
    else
        hipLaunchKernelGGL(wrw1x1_reduce_kernel<false>, dim3(256), dim3(256), 0, stream, (const float*)ws, blocks,
                           (unsigned short*)out);
    DTF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
