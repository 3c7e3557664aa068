// K16: direct 3x3, stride 1, padding 1 convolution of 64 -> 64 channels on NHWC bf16 (ResNet's stage-1 3x3 layers,
// forward and -- with the flipped, transposed filter -- data gradient), BatchNorm statistics in the epilogue.
//
// The 8-wave GEMM pipeline of K13 reads too many LDS bytes per MFMA when the output tile is 64 columns wide, and an
// implicit GEMM stages every input pixel once per filter tap; MIOpen's tuned kernel runs these layers at 107-112 us
// (batch 256, 56 x 56: 540 TFLOP/s).  Here (the K15 forward recipe):
//   * the whole filter lives in REGISTERS: a wave owns 32 output channels = 9 taps x 2 K halves x 2 fragments (144
//     registers), read straight from the [Cout][3][3][Cin] parameter, once per (persistent) block;
//   * a block owns strips of R output rows of one image; the R + 2 input rows (with one zero pixel left and right) go
//     into LDS by LDS-DMA, double-buffered, 16-byte chunks XOR-swizzled by the pixel position on the source side, so a
//     filter tap is a SHIFT of the pixel position: every pixel is staged once and read 9 times out of LDS
//     (ds_read_b128, 18 reads for 36 MFMAs per 16-pixel fragment and wave);
//   * operands swapped (D = W . X^T) with the filter rows permuted: a lane ends up with 8 consecutive output channels
//     of one pixel -- one 16-byte store -- and reduces them into the per-block statistics row.
#include "dtf_common.h"

typedef __attribute__((ext_vector_type(8))) short c3_s16x8;
typedef __attribute__((address_space(3))) void c3_lds_void;
typedef const __attribute__((address_space(1))) void c3_glb_void;

#define C3_C 64
#define C3_MAXP 16        // LDS-DMA pieces per wave and strip (upper bound of the unrolled staging loop)

struct C3Args {
    const unsigned short* x; const unsigned short* w; const unsigned short* zero; unsigned short* y; float* partial;
    int NB, H, W, R, P, strips_per_img, nstrips, bufbytes, npieces, npw, nfr;
};

__device__ __forceinline__ float c3_row16_sum(float v) {      // sum over the 16 lanes of a DPP row (lane & 15)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    return v;
}

extern __shared__ __attribute__((aligned(16))) unsigned c3_smem[];

// PB: row pitch in bytes (P * 128, P a multiple of 8 slots): the filter row's immediate read offset must be a constant
template <int PB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_c64_kernel(C3Args a) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fg = wave & 1, chalf = wave >> 1;
    const int i16 = lane & 15, qd = lane >> 4;
    constexpr int P = PB / 128;
    const int W = a.W;

    // ---- staging: piece j = LDS positions 8 j .. 8 j + 7 (a position = one pixel slot of 128 bytes; row i of the strip
    // window holds slots 0 .. W + 1, slot 0 and W + 1 are the zero padding); lane -> position 8 j + (lane >> 3), physical
    // 16-byte chunk lane & 7 = logical chunk ^ (position & 7)
    auto stage = [&](int strip, int buf) {
        const int n = strip / a.strips_per_img, oh0 = (strip - n * a.strips_per_img) * a.R;
        const unsigned short* ximg = a.x + (int64_t)n * a.H * W * C3_C;
        unsigned char* base = reinterpret_cast<unsigned char*>(c3_smem) + buf * a.bufbytes;
#pragma unroll 1
        for (int k = 0; k < a.npw; ++k) {
            {
                int j = wave + 4 * k;
                if (j >= a.npieces) j = a.npieces - 1;        // surplus slots repeat the last piece (identical data)
                const int pos = 8 * j + (lane >> 3);
                const int i = pos / P, p = pos - i * P;
                const int ih = oh0 - 1 + i, iw = p - 1;
                const int c = (lane & 7) ^ (pos & 7);
                const unsigned short* src = ximg + ((int64_t)ih * W + iw) * C3_C + c * 8;
                if (i >= a.R + 2 || (unsigned)ih >= (unsigned)a.H || (unsigned)iw >= (unsigned)W) src = a.zero;
                __builtin_amdgcn_global_load_lds((c3_glb_void*)src, (c3_lds_void*)(base + j * 1024), 16, 0, 0);
            }
        }
    };

    // ---- the filter: fragment (tap, kh, j) row i = output channel chalf * 32 + (i >> 2) * 8 + j * 4 + (i & 3), K index
    // kh * 32 + qd * 8 .. of the tap's 64 input channels
    c3_s16x8 wf[9][2][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int och = chalf * 32 + (i16 >> 2) * 8 + j * 4 + (i16 & 3);
                wf[t][kh][j] = *reinterpret_cast<const c3_s16x8*>(a.w + (och * 9 + t) * C3_C + kh * 32 + qd * 8);
            }
    float s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
    __builtin_amdgcn_s_waitcnt(0x0F70);           // the filter has landed before the counted waits start counting

    const unsigned lds0 = (unsigned)(uintptr_t)(c3_lds_void*)c3_smem;
    int strip = blockIdx.x;
    if (strip < a.nstrips) stage(strip, 0);
    int it = 0;
    for (; strip < a.nstrips; strip += gridDim.x, ++it) {
        const int buf = it & 1;
        const int next = strip + gridDim.x;
        if (next < a.nstrips) {
            stage(next, buf ^ 1);                  // streams in under this strip's MFMAs
            switch (a.npw) {                       // this strip's pieces were issued before the npw of the next one
#define C3_CASE(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
            C3_CASE(1) C3_CASE(2) C3_CASE(3) C3_CASE(4) C3_CASE(5) C3_CASE(6) C3_CASE(7) C3_CASE(8) C3_CASE(9) C3_CASE(10)
            C3_CASE(11) C3_CASE(12) C3_CASE(13) C3_CASE(14) C3_CASE(15) C3_CASE(16)
#undef C3_CASE
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int n = strip / a.strips_per_img, oh0 = (strip - n * a.strips_per_img) * a.R;
        const unsigned bufa = lds0 + (unsigned)(buf * a.bufbytes);
        // Two fragments (f, f + 2) per iteration, one register set each: while one fragment's 12 MFMAs of a filter row
        // run, the other fragment's reads of that row and this fragment's reads of the next row are in flight (counted
        // lgkmcnt: the 6 newest reads may still be outstanding), and four accumulator chains keep the matrix pipe fed.
        for (int f0 = fg; f0 < a.nfr; f0 += 4) {
            f32x4 acc[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
            int ohl[2], ow[2];
            bool inside[2];
            unsigned ad[2][3][2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                int f = f0 + 2 * u;
                inside[u] = f < a.nfr;                               // odd tail: the second slot repeats the first (not stored)
                if (!inside[u]) f = f0;
                const int l0 = 16 * f;
                const int ohl0 = l0 / W;                             // wave-uniform
                ow[u] = l0 - ohl0 * W + i16;
                ohl[u] = ohl0;
                if (ow[u] >= W) { ow[u] -= W; ++ohl[u]; }            // W >= 16: a fragment spans at most two rows
                if (ohl[u] >= a.R) { inside[u] = false; ohl[u] = a.R - 1; }      // the last fragment may run past the strip
                // The row pitch P is a multiple of 8 slots, so the swizzle key (position & 7) is the slot's (ow + s) & 7
                // whatever the filter row: three addresses per K half (s = 0, 1, 2), the filter row r is the IMMEDIATE offset
                // r * P * 128 of the read (the first version computed all 18 addresses: 5.8 VALU instructions per MFMA)
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int slot = ow[u] + s;
                    const unsigned off = (unsigned)((ohl[u] * P + slot) * 128 + ((qd ^ (slot & 7)) << 4));
                    ad[u][s][0] = bufa + off;
                    ad[u][s][1] = bufa + (off ^ 64u);
                }
            }
            // (inline asm: every register an asm read writes stays live until its wait; a set is reloaded only after the
            // MFMAs that read it have all been ISSUED -- they fetch their operands in their first passes, an LDS read
            // returns 64+ cycles later)
            c3_s16x8 fa[3][2], fb[3][2];
#define C3_READ(DST, U_, R_)                                                                              \
            _Pragma("unroll") for (int s = 0; s < 3; ++s) {                                                \
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[s][0]) : "v"(ad[U_][s][0]), "n"((R_) * PB)); \
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[s][1]) : "v"(ad[U_][s][1]), "n"((R_) * PB)); \
            }
#define C3_MFMA(SRC, U_, R_)                                                                              \
            _Pragma("unroll") for (int t = 0; t < 3; ++t)                                                  \
                _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                           \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                          \
                        acc[U_][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[3 * (R_) + t][kh][j], SRC[t][kh], acc[U_][j], 0, 0, 0);
#define C3_WAIT(N_)                                                                                       \
            asm volatile("s_waitcnt lgkmcnt(" #N_ ")" ::: "memory");                                       \
            __builtin_amdgcn_sched_barrier(0);
            C3_READ(fa, 0, 0)
            C3_READ(fb, 1, 0)
            C3_WAIT(6)
            C3_MFMA(fa, 0, 0)
            __builtin_amdgcn_sched_barrier(0);
            C3_READ(fa, 0, 1)
            C3_WAIT(6)
            C3_MFMA(fb, 1, 0)
            __builtin_amdgcn_sched_barrier(0);
            C3_READ(fb, 1, 1)
            C3_WAIT(6)
            C3_MFMA(fa, 0, 1)
            __builtin_amdgcn_sched_barrier(0);
            C3_READ(fa, 0, 2)
            C3_WAIT(6)
            C3_MFMA(fb, 1, 1)
            __builtin_amdgcn_sched_barrier(0);
            C3_READ(fb, 1, 2)
            C3_WAIT(6)
            C3_MFMA(fa, 0, 2)
            C3_WAIT(0)
            C3_MFMA(fb, 1, 2)
#undef C3_READ
#undef C3_MFMA
#undef C3_WAIT
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int oh = oh0 + ohl[u];
                if (inside[u] && oh < a.H) {
                    u32x4 o0;
                    o0[0] = pack_bf16x2(acc[u][0][0], acc[u][0][1]); o0[1] = pack_bf16x2(acc[u][0][2], acc[u][0][3]);
                    o0[2] = pack_bf16x2(acc[u][1][0], acc[u][1][1]); o0[3] = pack_bf16x2(acc[u][1][2], acc[u][1][3]);
                    unsigned short* yp = a.y + DTF_IDX((((int64_t)n * a.H + oh) * W + ow[u]) * C3_C + chalf * 32 + qd * 8,
                                                       (int64_t)a.NB * a.H * W * C3_C - 7, "K16 output strip edge");
                    *reinterpret_cast<u32x4*>(yp) = o0;
                    if (a.partial) {          // statistics of what the consumer reads: the rounded values
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float a0 = __uint_as_float(o0[k] << 16), a1 = __uint_as_float(o0[k] & 0xffff0000u);
                            s1[2 * k] += a0; s2[2 * k] = fmaf(a0, a0, s2[2 * k]);
                            s1[2 * k + 1] += a1; s2[2 * k + 1] = fmaf(a1, a1, s2[2 * k + 1]);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();              // every wave is done with this buffer before it is restaged
    }

    // ---- per-block statistics row: lanes of a DPP row (same qd) -> the two fragment-lane waves of a channel half.
    // s[k]: channel chalf * 32 + qd * 8 + k
    if (!a.partial) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* red = reinterpret_cast<float*>(c3_smem);    // [2 fragment lanes][2][64]
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float t1 = c3_row16_sum(s1[k]), t2 = c3_row16_sum(s2[k]);
        if (i16 == 0) {
            const int c = chalf * 32 + qd * 8 + k;
            red[(fg * 2 + 0) * C3_C + c] = t1;
            red[(fg * 2 + 1) * C3_C + c] = t2;
        }
    }
    __syncthreads();
    if (tid < 2 * C3_C) {
        const int stat = tid >> 6, c = tid & 63;
        a.partial[((int64_t)blockIdx.x * 2 + stat) * C3_C + c] = red[(0 * 2 + stat) * C3_C + c] + red[(1 * 2 + stat) * C3_C + c];
    }
}

static int c3_plan(int NB, int H, int W, C3Args* a, int* lds, int* grid) {
    if (NB < 1 || H < 1 || W < 16 || W > 1024) return -1;
    a->NB = NB; a->H = H; a->W = W; a->P = W + 2 <= 24 ? 24 : (W + 2 <= 32 ? 32 : 64);            // the instantiated pitches
    if (W + 2 > 64) return -1;
    static int rows = 0;
    if (rows == 0) { const char* e = getenv("DTF_C3_ROWS"); rows = e ? atoi(e) : 3; if (rows < 1 || rows > 8) rows = 3; }
    int R = rows < H ? rows : H;
    for (;; --R) {                                     // two buffers, two blocks per CU
        a->npieces = ((R + 2) * a->P + 7) / 8;
        a->bufbytes = a->npieces * 1024;
        if ((2 * a->bufbytes <= 163840 && (a->npieces + 3) / 4 <= C3_MAXP) || R == 1) break;
    }
    if (2 * a->bufbytes > 163840) return -1;
    a->R = R;
    a->npw = (a->npieces + 3) / 4;
    if (a->npw > C3_MAXP) return -1;
    a->strips_per_img = (H + R - 1) / R;
    if ((int64_t)NB * a->strips_per_img > 0x7fffffff) return -1;
    a->nstrips = NB * a->strips_per_img;
    a->nfr = (R * W + 15) / 16;
    *lds = 2 * a->bufbytes;
    if (*lds < 2 * 2 * C3_C * 4) *lds = 2 * 2 * C3_C * 4;
    int cu = 0, dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
    const int per_cu = 2 * *lds <= 163840 ? 2 : 1;
    *grid = per_cu * cu < a->nstrips ? per_cu * cu : a->nstrips;
    return 0;
}

extern "C" {

int dtf_conv3x3_c64_ok(int NB, int H, int W) {
    C3Args a{};
    int lds, grid;
    return c3_plan(NB, H, W, &a, &lds, &grid) == 0;
}

// partial rows ([blocks][2][64] fp32) the kernel writes
int64_t dtf_conv3x3_c64_blocks(int NB, int H, int W) {
    C3Args a{};
    int lds, grid = 0;
    if (c3_plan(NB, H, W, &a, &lds, &grid) != 0) return 0;
    return grid;
}

// y[NB, H, W, 64] = conv3x3 (x[NB, H, W, 64], w[64][3][3][64]), stride 1, padding 1; zero: >= 16 bytes of zeros; partial
// (optional): column sums / sums of squares of y, one row per block
int dtf_conv3x3_c64(const void* x, const void* w, const void* zero, void* y, float* partial, int NB, int H, int W,
                    hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    C3Args a{};
    int lds = 0, grid = 0;
    if (c3_plan(NB, H, W, &a, &lds, &grid) != 0) return -80;
    if ((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)y) | ((uintptr_t)partial) | ((uintptr_t)zero)) & 15) return -16;
    a.x = (const unsigned short*)x; a.w = (const unsigned short*)w; a.zero = (const unsigned short*)zero;
    a.y = (unsigned short*)y; a.partial = partial;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)conv3x3_c64_kernel<24 * 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        (void)hipFuncSetAttribute((const void*)conv3x3_c64_kernel<32 * 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        (void)hipFuncSetAttribute((const void*)conv3x3_c64_kernel<64 * 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        attr = true;
    }
    if (a.P == 24) hipLaunchKernelGGL((conv3x3_c64_kernel<24 * 128>), dim3(grid), dim3(256), lds, stream, a);
    else if (a.P == 32) hipLaunchKernelGGL((conv3x3_c64_kernel<32 * 128>), dim3(grid), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL((conv3x3_c64_kernel<64 * 128>), dim3(grid), dim3(256), lds, stream, a);
    DTF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
