// K14: DIRECT weight gradient of the narrow 3x3 stride-1 convolutions (Cin, Cout multiples of 64, NHWC bf16; a block
// works on one 64 x 64 channel slice, the blocks are divided among the (Cout / 64) x (Cin / 64) slices):
//
//     dW[co][r][s][ci] = sum_{n, h, w} dY[n, h, w, co] * X[n, h + r - 1, w + s - 1, ci]
//
// K6T-conv (gemm_tn_kernels.hip) runs this as an implicit GEMM whose patch matrix re-stages every input pixel once
// per filter tap; on a 64-channel layer that is 32 KB through the per-CU load path for 8 MFMAs per wave and K step,
// and the load path, not HBM or the matrix pipe, is the limit (docs/STATUS.md).  Here every pixel is staged ONCE:
//
//   * a block owns strips of R output rows of one image (R = 4, fewer where a strip would not fit LDS).  The strip's R dY rows and R + 2 input rows go into LDS as
//     FLATTENED position lists with a row pitch of P = W + 2: the two extra columns hold the convolution's zero padding
//     (X) / zeros (dY).  A filter tap (r, s) is then a UNIFORM shift of the flattened index, q = p + r P + s - 1, and
//     the products that wrap around a row end vanish because dY is zero there -- no per-pixel masks anywhere;
//   * K step = 32 flattened positions; both operands have the reduction index on the LDS rows, so the MFMA fragments
//     come from ds_read_b64_tr_b16 (the K6T recipe) out of 128-byte position rows, 32-byte chunks XOR-swizzled by
//     bits 1 and 3 of the position (conflict-free for the 8 rows a 32-lane half reads);
//   * the whole filter gradient stays in registers: 9 taps x 4 x 4 fragments of 16 x 16 = 144 fragments, 18 per wave
//     (wave = input-channel group x half of the (tap, output-channel group) list: 4 dY + 5 X fragment reads feed 18
//     MFMAs per K step);
//   * strips are double-buffered by LDS-DMA (the next strip streams in under the current one's MFMAs: counted vmcnt,
//     raw s_barrier); each block finally writes ONE fp32 slab, summed in fixed order by tn_reduce (deterministic).
#include "dtf_common.h"

typedef __attribute__((ext_vector_type(8))) short wd_s16x8;
typedef __attribute__((ext_vector_type(4))) short wd_s16x4;
typedef __attribute__((address_space(3))) void wd_lds_void;
typedef const __attribute__((address_space(1))) void wd_glb_void;

#define WD_R 4            // output rows per strip
#define WD_C 64           // channels per slice (both sides)
#define WD_MAXP 16        // LDS-DMA pieces per wave and strip (upper bound of the unrolled staging loop)

struct WdArgs {
    const unsigned short* dy; const unsigned short* x; const unsigned short* zero; float* slab;
    int NB, H, W, P, nch, xpos, ypos, nstrips, strips_per_img, npw;      // npw: pieces per wave (uniform); H, W: OUTPUT size
    int R, Hx, Wx, pad;                                                  // rows per strip; input size; padding (0 / 1)
    int Cin, Cout, nsl_ci, nsl, bps;                                     // channel slices; bps: blocks per slice
};

__device__ __forceinline__ wd_s16x4 wd_tr_read(unsigned addr) {
    wd_s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

// chunk ch of a strip sits ch * 4096 bytes on: an immediate offset, so the 18 fragment addresses are computed once
// per strip, not per read (the first version spent 58 VALU instructions per 18 MFMAs on them)
template <int OFF>
__device__ __forceinline__ wd_s16x4 wd_tr_read_o(unsigned addr) {
    wd_s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

__device__ __forceinline__ int wd_swz(int pos) { return ((pos >> 1) & 1) | (((pos >> 3) & 1) << 1); }

__global__ __launch_bounds__(512) void conv_wrw_direct_kernel(WdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cig = wave & 3, half = wave >> 2;
    const int P = a.P, W = a.W, H = a.H;
    const int bufbytes = (a.xpos + a.ypos) * 128;
    const int slice = blockIdx.x % a.nsl, bis = blockIdx.x / a.nsl;       // channel slice of this block, its index inside
    const int co0 = (slice / a.nsl_ci) * WD_C, ci0 = (slice % a.nsl_ci) * WD_C;
    const int ybase = a.xpos * 128;               // dY image behind the X image inside a buffer

    // zero both buffers once: slack positions and padding columns are never written again and must stay finite
    for (int i = tid; i < 2 * bufbytes / 16; i += 512) reinterpret_cast<u32x4*>(smem)[i] = (u32x4){0u, 0u, 0u, 0u};

    // ---- staging plan (strip-independent): piece j = 8 consecutive LDS positions of one image; lane -> position
    // 8 j + (lane >> 3), 16-byte slot lane & 7.  X pieces first (xpos / 8), then dY pieces (ypos / 8); wave w takes
    // pieces w, w + 8, ...  Per piece and lane: pixel offset inside the strip window (or -1: padding / slack) and the
    // local row (validity against the image's top / bottom edge is per strip).
    const int nxp = a.xpos >> 3, npieces = nxp + (a.ypos >> 3);
    int pc_off[WD_MAXP], pc_row[WD_MAXP], pc_lds[WD_MAXP];
#pragma unroll
    for (int k = 0; k < WD_MAXP; ++k) {
        int j = wave + 8 * k;
        if (j >= npieces) j = npieces - 1;          // surplus slots repeat the last piece (identical data)
        const bool isx = j < nxp;
        const int pos = (isx ? j : j - nxp) * 8 + (lane >> 3);      // LDS position inside its image
        const int q = isx ? pos - 1 : pos;                          // X image has one position of front slack
        const int rows = isx ? a.R + 2 : a.R;
        int off = -1, rl = 0;
        if (q >= 0 && q < rows * P) {
            rl = q / P;
            const int col = q - rl * P;
            // dY: columns 1 .. W are the row, 0 and W + 1 zeros.  X: column c is input column c - pad (padding 1: the two
            // outer columns are the convolution's zero padding; padding 0: all W + 2 columns are image)
            if (isx) { if ((unsigned)(col - a.pad) < (unsigned)a.Wx) off = rl * a.Wx + col - a.pad; }
            else if (col >= 1 && col <= W) off = rl * W + col - 1;
        }
        const int d16 = lane & 7;
        const int c = (d16 >> 1) ^ wd_swz(pos);                     // source 32-byte chunk that lands in this slot
        // channels past the tensor's (a 96-channel layer fills half of its second slice) read zeros
        const int chn = (isx ? ci0 : co0) + (c * 2 + (d16 & 1)) * 8;
        if (chn >= (isx ? a.Cin : a.Cout)) off = -1;
        pc_off[k] = off < 0 ? -1 : off * (isx ? a.Cin : a.Cout) + chn;
        pc_row[k] = rl | (isx ? 0 : 256);
        pc_lds[k] = (isx ? 0 : ybase) + (isx ? j : j - nxp) * 1024;
    }
    auto stage = [&](int strip, int buf) {
        const int n = strip / a.strips_per_img, h0 = (strip - n * a.strips_per_img) * a.R;
        const unsigned short* xw = a.x + ((int64_t)n * a.Hx + (h0 - a.pad)) * a.Wx * a.Cin;     // window origin: input row h0 - pad
        const unsigned short* yw = a.dy + ((int64_t)n * H + h0) * W * a.Cout;
        unsigned char* base = smem + buf * bufbytes;
#pragma unroll
        for (int k = 0; k < WD_MAXP; ++k) {
            if (k < a.npw) {
                const bool isx = pc_row[k] < 256;
                const int rl = pc_row[k] & 255;
                const int ih = isx ? h0 - a.pad + rl : h0 + rl;
                const unsigned short* src = (isx ? xw : yw) + pc_off[k];
                if (pc_off[k] < 0 || (unsigned)ih >= (unsigned)(isx ? a.Hx : H)) src = a.zero;
                __builtin_amdgcn_global_load_lds((wd_glb_void*)src, (wd_lds_void*)(base + pc_lds[k]), 16, 0, 0);
            }
        }
    };

    // ---- fragment addressing (ds_read_b64_tr_b16: lane = 16 g + 4 q + p supplies row 8 g + 4 h + q, columns 4 p ..)
    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const unsigned lds0 = (unsigned)(uintptr_t)(wd_lds_void*)smem;
    int offA[4][2], offB[5][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = 8 * g + 4 * h + qq;
#pragma unroll
        for (int c = 0; c < 4; ++c) {        // dY fragment of this wave's c-th output-channel group (half 1: reversed list)
            const int cog = half ? 3 - c : c;
            offA[c][h] = ybase + row * 128 + ((cog ^ wd_swz(row)) << 5) + 8 * pp;
        }
#pragma unroll
        for (int t = 0; t < 5; ++t) {        // X fragment of this wave's t-th tap (half 0: taps 0..4, half 1: 8..4), channel group cig
            const int tap = half ? 8 - t : t, r = tap / 3, s = tap - r * 3;
            const int xrow = 1 + r * P + s - 1 + row;               // + 1: front slack position
            offB[t][h] = xrow * 128 + ((cig ^ wd_swz(xrow)) << 5) + 8 * pp;
        }
    }

    f32x4 acc[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int strip = bis;
    __syncthreads();                               // the zero fill is complete before the first DMA lands
    if (strip < a.nstrips) stage(strip, 0);
    int it = 0;
    for (; strip < a.nstrips; strip += a.bps, ++it) {
        const int buf = it & 1;
        const int next = strip + a.bps;
        if (next < a.nstrips) {
            stage(next, buf ^ 1);                  // streams in under this strip's MFMAs
            // this strip's pieces were issued before the npw of the next one
            switch (a.npw) {
#define WD_CASE(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
            WD_CASE(1) WD_CASE(2) WD_CASE(3) WD_CASE(4) WD_CASE(5) WD_CASE(6) WD_CASE(7) WD_CASE(8) WD_CASE(9) WD_CASE(10)
            WD_CASE(11) WD_CASE(12) WD_CASE(13) WD_CASE(14) WD_CASE(15) WD_CASE(16)
#undef WD_CASE
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned bb = lds0 + (unsigned)(buf * bufbytes);
        unsigned adA[4][2], adB[5][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int c = 0; c < 4; ++c) adA[c][h] = bb + (unsigned)offA[c][h];
#pragma unroll
            for (int t = 0; t < 5; ++t) adB[t][h] = bb + (unsigned)offB[t][h];
        }
        // software pipeline over the chunks: the fragment reads of chunk c + 1 are issued BEFORE the MFMAs of chunk c
        // (two register sets), so a wave's LDS latency hides behind its own 18 MFMAs
        wd_s16x4 la0[4], ha0[4], lb0[5], hb0[5], la1[4], ha1[4], lb1[5], hb1[5];
#define WD_READ(CH, S)                                                                                                \
        if (a.nch > CH) {                                                                                             \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                           \
                la##S[c] = wd_tr_read_o<(CH) * 4096>(adA[c][0]);                                                      \
                ha##S[c] = wd_tr_read_o<(CH) * 4096>(adA[c][1]);                                                      \
            }                                                                                                         \
            _Pragma("unroll") for (int t = 0; t < 5; ++t) {                                                           \
                lb##S[t] = wd_tr_read_o<(CH) * 4096>(adB[t][0]);                                                      \
                hb##S[t] = wd_tr_read_o<(CH) * 4096>(adB[t][1]);                                                      \
            }                                                                                                         \
        }
#define WD_STEP(CH, S, T)                                                                                             \
        if (a.nch > CH) {                                                                                             \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            wd_s16x8 fa[4], fb[5];                                                                                    \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) fa[c] = __builtin_shufflevector(la##S[c], ha##S[c], 0, 1, 2, 3, 4, 5, 6, 7); \
            _Pragma("unroll") for (int t = 0; t < 5; ++t) fb[t] = __builtin_shufflevector(lb##S[t], hb##S[t], 0, 1, 2, 3, 4, 5, 6, 7); \
            WD_READ((CH) + 1, T)                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            /* the 36 (tap, output-channel group) pairs: half 0 walks pairs 0 .. 17 upwards, half 1 pairs 35 .. 18 */ \
            /* downwards -- the same (k >> 2, k & 3) indexing of the two (reversed) fragment lists */                 \
            _Pragma("unroll") for (int k = 0; k < 18; ++k)                                                            \
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[k & 3], fb[k >> 2], acc[k], 0, 0, 0);             \
        }
        WD_READ(0, 0)
        WD_STEP(0, 0, 1) WD_STEP(1, 1, 0) WD_STEP(2, 0, 1) WD_STEP(3, 1, 0) WD_STEP(4, 0, 1) WD_STEP(5, 1, 0) WD_STEP(6, 0, 1)
        WD_STEP(7, 1, 0)
#undef WD_STEP
#undef WD_READ
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();              // every wave is done with this buffer before it is restaged
    }

    // ---- slab[slice][block in slice][co][tap][ci]; C map of the 16 x 16 fragment: row (co) = 4 g + r, column (ci) = lane & 15
    float* out = a.slab + ((int64_t)slice * a.bps + bis) * (9 * WD_C * WD_C);
#pragma unroll
    for (int k = 0; k < 18; ++k) {
        const int tap = half ? 8 - (k >> 2) : (k >> 2), cog = half ? 3 - (k & 3) : (k & 3);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = cog * 16 + g * 4 + r, ci = cig * 16 + (lane & 15);
            out[((int64_t)co * 9 + tap) * WD_C + ci] = acc[k][r];
        }
    }
}

// out[i] = sum_b slab[b][i] in a fixed order, one rounding.  One slab per CU means 256 slabs of 147 KB: a thread that
// walks them alone issues 256 dependent-latency loads (the first version: ~75 us of the kernel pair's 140).  Here a block
// covers 64 float4 columns x 4 slab lanes; a lane sums the slabs b = lane (mod 4), eight loads in flight, and the four
// lanes are folded through LDS in lane order.
// blockIdx.y = channel slice: its slabs are [nb][64][9][64], its results go to out[co0 + co][tap][ci0 + ci] of [Cout][9][Cin].
__global__ __launch_bounds__(256) void wd_reduce_kernel(const float* __restrict__ slab, int nb, int n4, void* __restrict__ out,
                                                        int out_dtype, int Cin, int Cout, int nsl_ci) {
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    const int slice = blockIdx.y;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (col < n4) {
        const f32x4* p = reinterpret_cast<const f32x4*>(slab) + (int64_t)slice * nb * n4 + col;
        int b = sl;
        for (; b + 28 < nb; b += 32) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(b + 4 * u) * n4];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < nb; b += 4) s += p[(int64_t)b * n4];
    }
    __shared__ f32x4 red[4][64];
    red[sl][threadIdx.x & 63] = s;
    __syncthreads();
    if (sl != 0 || col >= n4) return;
    s = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    // column = (co * 9 + tap) * 16 + ci / 4 inside the slice
    const int ci4 = col & 15, cotap = col >> 4;
    const int co0 = (slice / nsl_ci) * WD_C, ci0 = (slice % nsl_ci) * WD_C;
    if (co0 + cotap / 9 >= Cout || ci0 + ci4 * 4 >= Cin) return;      // partial last slices
    const int64_t o = ((int64_t)co0 * 9 + cotap) * Cin + ci0 + ci4 * 4;
    if (out_dtype == DTF_BF16) {
        u32x2 r;
        r[0] = pack_bf16x2(s[0], s[1]);
        r[1] = pack_bf16x2(s[2], s[3]);
        *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(out) + o) = r;
    } else {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + o) = s;
    }
}

// H, W: OUTPUT size; pad 1 ('same': input H x W) or 0 ('valid': input (H + 2) x (W + 2))
static int wd_plan(int NB, int H, int W, int Cin, int Cout, int pad, WdArgs* a, int* grid) {
    if (Cin < 32 || Cout < 32 || Cin % 32 || Cout % 32 || Cin > 256 || Cout > 256 || (pad != 0 && pad != 1)) return -1;
    const int P = W + 2;
    // rows per strip: the most (<= WD_R) whose two buffers fit the 160 KB of LDS and whose chunk / piece counts fit the
    // unrolled loops (wide images -- Inception's 147 x 147 stem layers -- run one row per strip)
    int R = WD_R < H ? WD_R : H, nch = 0, xpos = 0, ypos = 0, npw = 0;
    for (;; --R) {
        if (R < 1) return -1;
        nch = (R * P + 31) / 32;
        xpos = (1 + nch * 32 + 2 * P + 2 + 7) / 8 * 8;      // largest read: position nch*32 - 1 shifted by 2 P + 1 (+ slack)
        ypos = nch * 32;
        npw = ((xpos + ypos) / 8 + 7) / 8;
        if (2 * (xpos + ypos) * 128 <= 163840 && xpos >= 1 + (R + 2) * P && nch <= 8 && npw <= WD_MAXP && npw >= 1) break;
    }
    a->NB = NB; a->H = H; a->W = W; a->P = P; a->nch = nch; a->xpos = xpos; a->ypos = ypos; a->npw = npw;
    a->R = R; a->pad = pad; a->Hx = H + 2 - 2 * pad; a->Wx = W + 2 - 2 * pad;
    a->strips_per_img = (H + R - 1) / R;
    a->nstrips = NB * a->strips_per_img;
    int cu = 0, dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
    a->Cin = Cin; a->Cout = Cout; a->nsl_ci = (Cin + WD_C - 1) / WD_C; a->nsl = a->nsl_ci * ((Cout + WD_C - 1) / WD_C);
    int bps = cu / a->nsl;                                          // blocks per channel slice
    if (bps > a->nstrips) bps = a->nstrips;
    if (bps < 1) return -1;
    a->bps = bps;
    *grid = bps * a->nsl;
    return 0;
}

extern "C" {

// 1 when the direct kernel covers this geometry (3x3, stride 1, padding 1 or 0, channels in multiples of 32 up to 256, a
// strip fits LDS); H, W: INPUT size
int dtf_conv_wrw_direct_ok(int NB, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int ph, int pw) {
    if (KH != 3 || KW != 3 || stride != 1 || ph != pw || (ph != 0 && ph != 1) || NB < 1) return 0;
    const int OH = H - 2 + 2 * ph, OW = W - 2 + 2 * ph;
    if (OH < 1 || OW < 1) return 0;
    WdArgs a{};
    int grid;
    return wd_plan(NB, OH, OW, Cin, Cout, ph, &a, &grid) == 0;
}

// fp32 workspace floats: one slab per block (H, W: INPUT size)
int64_t dtf_conv_wrw_direct_workspace2(int NB, int H, int W, int Cin, int Cout, int pad) {
    WdArgs a{};
    int grid = 0;
    if (wd_plan(NB, H - 2 + 2 * pad, W - 2 + 2 * pad, Cin, Cout, pad, &a, &grid) != 0) return 0;
    return (int64_t)grid * 9 * WD_C * WD_C;
}
int64_t dtf_conv_wrw_direct_workspace(int NB, int H, int W, int Cin, int Cout) {
    return dtf_conv_wrw_direct_workspace2(NB, H, W, Cin, Cout, 1);
}

// dw[Cout][3][3][Cin] (out_dtype: bf16 / fp32) from dy[NB, OH, OW, Cout] and x[NB, H, W, Cin] (OH = H - 2 + 2 pad);
// zero: >= 16 bytes of zeros
int dtf_conv_wrw_direct2(const void* dy, const void* x, void* dw, int out_dtype, float* workspace, const void* zero, int NB,
                         int H, int W, int Cin, int Cout, int pad, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    WdArgs a{};
    int grid = 0;
    if (wd_plan(NB, H - 2 + 2 * pad, W - 2 + 2 * pad, Cin, Cout, pad, &a, &grid) != 0) return -80;
    if ((((uintptr_t)dy) | ((uintptr_t)x) | ((uintptr_t)dw) | ((uintptr_t)workspace) | ((uintptr_t)zero)) & 15) return -16;
    a.dy = (const unsigned short*)dy; a.x = (const unsigned short*)x; a.zero = (const unsigned short*)zero; a.slab = workspace;
    const int lds = 2 * (a.xpos + a.ypos) * 128;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)conv_wrw_direct_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        attr = true;
    }
    hipLaunchKernelGGL(conv_wrw_direct_kernel, dim3(grid), dim3(512), lds, stream, a);
    const int n4 = 9 * WD_C * WD_C / 4;
    hipLaunchKernelGGL(wd_reduce_kernel, dim3((n4 + 63) / 64, a.nsl), dim3(256), 0, stream, workspace, a.bps, n4, dw, out_dtype,
                       Cin, Cout, a.nsl_ci);
    DTF_CHECK_LAUNCH();
    return 0;
}

int dtf_conv_wrw_direct(const void* dy, const void* x, void* dw, int out_dtype, float* workspace, const void* zero, int NB, int H,
                        int W, int Cin, int Cout, hipStream_t stream) {
    return dtf_conv_wrw_direct2(dy, x, dw, out_dtype, workspace, zero, NB, H, W, Cin, Cout, 1, stream);
}

}  // extern "C"
