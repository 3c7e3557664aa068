// K17: 1x1 convolution with FEW input channels as a streaming kernel with the BatchNorm statistics in its epilogue:
//     y[M, Cout] = x[M, Cin] . w[Cout, Cin]^T,   partial[block][2][Cout] = column sums / sums of squares of (bf16) y.
// ResNet's stage-1 expand layers (64 -> 256 channels at 56 x 56, batch 256: 103 MB in, 411 MB out) are pure HBM traffic:
// the validated library GEMM takes 107 us and the K7 statistics pass over the output another 75 us.  K6 / K13 with the
// statistics epilogue lose here (their 256-wide tiles re-stage the operand per column panel).  This kernel (the K16
// recipe without taps): the whole filter in registers (a wave owns 16 NJ output channels: KH x NJ fragments), persistent
// blocks, 128-pixel tiles LDS-DMA double-buffered as swizzled pixel slots that ALL waves of the block read, operands
// swapped with permuted filter rows so a lane stores 4 NJ consecutive channels of one pixel, statistics per lane ->
// DPP row sum -> one partial row per block (a channel belongs to one wave: no cross-wave reduce).
#include "dtf_common.h"

typedef __attribute__((ext_vector_type(8))) short p1_s16x8;
typedef __attribute__((address_space(3))) void p1_lds_void;
typedef const __attribute__((address_space(1))) void p1_glb_void;

#define P1_T 128          // pixels per tile

struct P1Args {
    const unsigned short* x; const unsigned short* w; const unsigned short* zero; unsigned short* y; float* partial;
    int64_t M; int Cin, Cout, ntiles, npw, npieces, bufbytes;
};

__device__ __forceinline__ float p1_row16_sum(float v) {      // sum over the 16 lanes of a DPP row (lane & 15)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    return v;
}

extern __shared__ __attribute__((aligned(16))) unsigned p1_smem[];

// KH: K halves of 32 input channels (Cin = 32 KH); NJ: 16-channel output fragments per wave; NW: waves (Cout = 16 NJ NW)
template <int KH, int NJ, int NW>
__global__ __launch_bounds__(NW * 64) void conv1x1_stats_kernel(P1Args a) {
    constexpr int SLOT = KH * 64;                 // bytes of one pixel
    constexpr int CPS = SLOT / 16;                // 16-byte chunks per pixel
    constexpr int PPP = 1024 / SLOT;              // pixels per 1 KB LDS-DMA piece
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, qd = lane >> 4;

    // piece j = pixels PPP j .. of the tile; lane -> pixel PPP j + lane / CPS, physical chunk lane % CPS = logical ^ (pixel % CPS)
    auto stage = [&](int tile, int buf) {
        unsigned char* base = reinterpret_cast<unsigned char*>(p1_smem) + buf * a.bufbytes;
#pragma unroll 1
        for (int k = 0; k < a.npw; ++k) {
            int j = wave + NW * k;
            if (j >= a.npieces) j = a.npieces - 1;        // surplus slots repeat the last piece (identical data)
            const int pos = PPP * j + lane / CPS;
            const int c = (lane % CPS) ^ (pos & (CPS - 1));
            const int64_t px = (int64_t)tile * P1_T + pos;
            const unsigned short* src = px < a.M ? a.x + DTF_IDX(px * a.Cin + c * 8, a.M * a.Cin - 7, "K17 input tile edge") : a.zero;
            __builtin_amdgcn_global_load_lds((p1_glb_void*)src, (p1_lds_void*)(base + j * 1024), 16, 0, 0);
        }
    };

    // the filter: fragment (kh, j) row i = output channel 16 NJ wave + (i >> 2) * 4 NJ + 4 j + (i & 3)
    p1_s16x8 wf[KH][NJ];
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int och = 16 * NJ * wave + (i16 >> 2) * (4 * NJ) + 4 * j + (i16 & 3);
            wf[kh][j] = *reinterpret_cast<const p1_s16x8*>(a.w + (int64_t)och * a.Cin + kh * 32 + qd * 8);
        }
    float s1[4 * NJ], s2[4 * NJ];
#pragma unroll
    for (int k = 0; k < 4 * NJ; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
    __builtin_amdgcn_s_waitcnt(0x0F70);           // the filter has landed before the counted waits start counting

    const unsigned lds0 = (unsigned)(uintptr_t)(p1_lds_void*)p1_smem;
    const int ch0 = 16 * NJ * wave + qd * (4 * NJ);        // this lane's 4 NJ consecutive output channels
    int tile = blockIdx.x;
    if (tile < a.ntiles) stage(tile, 0);
    int it = 0;
    for (; tile < a.ntiles; tile += gridDim.x, ++it) {
        const int buf = it & 1;
        const int next = tile + gridDim.x;
        if (next < a.ntiles) {
            stage(next, buf ^ 1);                  // streams in under this tile's work
            switch (a.npw) {                       // this tile's pieces were issued before the npw of the next one
#define P1_CASE(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
            P1_CASE(1) P1_CASE(2) P1_CASE(3) P1_CASE(4) P1_CASE(5) P1_CASE(6) P1_CASE(7) P1_CASE(8)
#undef P1_CASE
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned bufa = lds0 + (unsigned)(buf * a.bufbytes);
#pragma unroll
        for (int g = 0; g < P1_T / 64; ++g) {                 // four 16-pixel fragments at a time
            p1_s16x8 fx[4][KH];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pos = 64 * g + 16 * u + i16;
#pragma unroll
                for (int kh = 0; kh < KH; ++kh) {
                    const unsigned ad = bufa + (unsigned)(pos * SLOT + (((kh * 4 + qd) ^ (pos & (CPS - 1))) << 4));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(fx[u][kh]) : "v"(ad));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc[4][NJ];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[u][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kh = 0; kh < KH; ++kh)
                        acc[u][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kh][j], fx[u][kh], acc[u][j], 0, 0, 0);
                }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t px = (int64_t)tile * P1_T + 64 * g + 16 * u + i16;
                if (px < a.M) {
                    unsigned o[2 * NJ];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        o[2 * j] = pack_bf16x2(acc[u][j][0], acc[u][j][1]);
                        o[2 * j + 1] = pack_bf16x2(acc[u][j][2], acc[u][j][3]);
                    }
                    unsigned short* yp = a.y + DTF_IDX(px * a.Cout + ch0, a.M * a.Cout - 4 * NJ + 1, "K17 output tile edge");
#pragma unroll
                    for (int v = 0; v < NJ / 2; ++v)
                        *reinterpret_cast<u32x4*>(yp + 8 * v) = (u32x4){o[4 * v], o[4 * v + 1], o[4 * v + 2], o[4 * v + 3]};
                    // statistics of what the consumer reads: the rounded values
#pragma unroll
                    for (int k = 0; k < 2 * NJ; ++k) {
                        const float a0 = __uint_as_float(o[k] << 16), a1 = __uint_as_float(o[k] & 0xffff0000u);
                        s1[2 * k] += a0; s2[2 * k] = fmaf(a0, a0, s2[2 * k]);
                        s1[2 * k + 1] += a1; s2[2 * k + 1] = fmaf(a1, a1, s2[2 * k + 1]);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();              // every wave is done with this buffer before it is restaged
    }
    // one partial row per block: a channel belongs to one wave and one lane quarter
#pragma unroll
    for (int k = 0; k < 4 * NJ; ++k) {
        const float t1 = p1_row16_sum(s1[k]), t2 = p1_row16_sum(s2[k]);
        if (i16 == 0) {
            a.partial[((int64_t)blockIdx.x * 2 + 0) * a.Cout + ch0 + k] = t1;
            a.partial[((int64_t)blockIdx.x * 2 + 1) * a.Cout + ch0 + k] = t2;
        }
    }
}

static int p1_plan(int64_t M, int Cin, int Cout, P1Args* a, int* grid) {
    if (M < 1 || M >= (1ll << 31) * 8 || !((Cin == 64 && Cout == 256))) return -1;       // the instantiated shape
    a->M = M; a->Cin = Cin; a->Cout = Cout;
    a->ntiles = (int)((M + P1_T - 1) / P1_T);
    a->npieces = P1_T * Cin * 2 / 1024;
    a->bufbytes = a->npieces * 1024;
    a->npw = (a->npieces + 7) / 8;
    if (a->npw > 8) return -1;
    int cu = 0, dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
    static int per_cu = 0;
    if (per_cu == 0) { const char* e = getenv("DTF_K17_BLOCKS_PER_CU"); per_cu = e ? atoi(e) : 3; if (per_cu < 1) per_cu = 3; }
    *grid = cu * per_cu < a->ntiles ? cu * per_cu : a->ntiles;
    return 0;
}

extern "C" {

int dtf_conv1x1_stats_ok(int64_t M, int Cin, int Cout) {
    P1Args a{};
    int grid;
    return p1_plan(M, Cin, Cout, &a, &grid) == 0;
}

// partial rows ([blocks][2][Cout] fp32) the kernel writes
int64_t dtf_conv1x1_stats_blocks(int64_t M, int Cin, int Cout) {
    P1Args a{};
    int grid = 0;
    if (p1_plan(M, Cin, Cout, &a, &grid) != 0) return 0;
    return grid;
}

// y[M, Cout] = x[M, Cin] . w[Cout, Cin]^T + the statistics partials of y; zero: >= 16 bytes of zeros
int dtf_conv1x1_stats(const void* x, const void* w, const void* zero, void* y, float* partial, int64_t M, int Cin, int Cout,
                      hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    P1Args a{};
    int grid = 0;
    if (p1_plan(M, Cin, Cout, &a, &grid) != 0) return -80;
    if ((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)y) | ((uintptr_t)partial) | ((uintptr_t)zero)) & 15) return -16;
    a.x = (const unsigned short*)x; a.w = (const unsigned short*)w; a.zero = (const unsigned short*)zero;
    a.y = (unsigned short*)y; a.partial = partial;
    hipLaunchKernelGGL((conv1x1_stats_kernel<2, 2, 8>), dim3(grid), dim3(512), 2 * a.bufbytes, stream, a);
    DTF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
