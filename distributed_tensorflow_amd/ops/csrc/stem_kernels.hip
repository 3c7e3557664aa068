// K15: the ResNet stem -- 7x7, stride 2, padding 3, 3 -> 64 channels on NHWC bf16 -- as two direct kernels.
//
// The library runs this layer at 359 us forward / 354 us weight gradient (batch 256, 224^2; MIOpen igemm asm), the
// BatchNorm statistics pass over its 411 MB output takes another 111 us and the BatchNorm backward apply pass that
// produces the weight gradient's operand 184 us: 1.0 ms of a 21.6 ms step for 0.12 TFLOP of work whose HBM floor is
// ~0.25 ms.  Nothing in the GEMM pipelines fits a 3-channel input (a pixel is 6 bytes), so the layer gets its own
// formulation:
//
//   * forward (stem_fwd_kernel): a block owns R output rows of one image.  The 2R + 5 input rows go into LDS once
//     (shifted by one element so that every window start is dword-aligned: a window starts at element 6 ow - 9).
//     A filter ROW is one K step: its 7 taps x 3 channels are 21 CONSECUTIVE elements of the input row, padded to 32
//     with zeros (K = 7 x 32); a lane's MFMA operand is 8 consecutive elements of its pixel's window -- 4 dword reads,
//     no gather.  The filter (28 fragments) lives in registers for the whole block.  Operands are swapped
//     (D = W . X^T) and the filter rows permuted so that a lane ends up with 2 x 8 CONSECUTIVE output channels of one
//     pixel: two 16-byte stores, each store instruction covering 64 contiguous bytes of every pixel's line.
//     The epilogue also reduces the bf16-rounded outputs per channel (sum, sum of squares) into one partial row per
//     block in the layout of the K7 finalize kernel: the BatchNorm statistics pass over the output disappears.
//   * weight gradient (stem_wrw_kernel, below): see there.
#include "dtf_common.h"

typedef __attribute__((ext_vector_type(8))) short st_s16x8;

#define ST_C 64           // output channels
#define ST_KH 7
#define ST_T 21           // taps x channels of one filter row
#define ST_STAGE_U 8      // staging tasks in flight per thread

typedef __attribute__((address_space(3))) void st_lds_void;
typedef const __attribute__((address_space(1))) void st_glb_void;

struct StemFwdArgs {
    const unsigned short* x; const unsigned short* wp; const unsigned short* zero; unsigned short* y; float* partial;
    int NB, H, W, OH, OW, R, strips_per_img, nstrips, pitch, nrows, owf, bufbytes, npieces, npw;
};

__device__ __forceinline__ float st_row16_sum(float v) {      // sum over the 16 lanes of a DPP row (lane & 15)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    return v;
}

// wp[r][o][1 + t] = w[o][r][t], t = 0..20 (w: [64][7][7][3], the channels-last parameter); wp[r][o][0] and [22..31] zero:
// the forward kernel's K step starts ONE element before the window (an even element: dword-aligned in LDS)
__global__ __launch_bounds__(256) void stem_pack_w_kernel(const unsigned short* __restrict__ w, unsigned short* __restrict__ wp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ST_KH * ST_C * 32) return;
    const int t = i & 31, o = (i >> 5) & 63, r = i >> 11;
    wp[i] = (t >= 1 && t <= ST_T) ? w[o * (ST_KH * ST_T) + r * ST_T + t - 1] : (unsigned short)0;
}

// channel of accumulator (j, i) of the lane with quarter qd: (j >> 1) * 32 + qd * 8 + (j & 1) * 4 + i
//
// Staging is LDS-DMA, double-buffered (the K14 recipe): a buffer is nrows rows of `pitch` bytes -- 32 bytes of left
// padding, the image row, >= 32 bytes of right padding -- and is filled as a sequence of contiguous 1 KB pieces whose
// lanes fetch either 16 bytes of the image or 16 bytes of a zero page (padding columns, rows outside the image), so no
// lane ever needs a register and the next strip streams in under the current strip's MFMAs.  The image rows keep their
// dword alignment that way, and a window (element 6 ow - 9 of its row) starts on an ODD element: a lane reads 5 dwords
// and funnel-shifts them into its 8-element operand.
#define ST_MAXP 8         // LDS-DMA pieces per wave and strip (upper bound of the unrolled staging loop)
extern __shared__ __attribute__((aligned(16))) unsigned st_smem[];

// PAR: column parity of this wave's fragments (wave & 1)
template <int PAR>
__device__ __forceinline__ void stem_fwd_body(const StemFwdArgs& a) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rowbytes = a.W * 6;

    // 4 waves = 2 fragment lanes x 2 channel halves: a wave keeps the filter fragments of 32 output channels (56
    // registers), three blocks of R = 4 rows fit a CU (12 waves: the LDS latency of the window reads hides behind the
    // other waves; with all 64 channels per wave -- 228 registers, 8 waves per CU -- the kernel was bound by that
    // latency: 139 us even without its stores).
    const int fg = wave & 1, chalf = wave >> 1;
    // ---- staging plan (strip-independent): piece j = bytes [1024 j, 1024 j + 1024) of a buffer; wave w takes pieces
    // w, w + 4, ...; lane -> 16 bytes at 1024 j + 16 lane = (row, column) of the buffer
    // (recomputed per strip: one division per piece and lane is cheaper than 18 live registers)
    auto stage = [&](int strip, int buf) {
        const int n = strip / a.strips_per_img, oh0 = (strip - n * a.strips_per_img) * a.R;
        const int ih0 = 2 * oh0 - 3;
        const unsigned short* ximg = a.x + (int64_t)n * a.H * (a.W * 3);
        unsigned char* base = reinterpret_cast<unsigned char*>(st_smem) + buf * a.bufbytes;
#pragma unroll
        for (int k = 0; k < ST_MAXP; ++k) {
            if (k < a.npw) {
                int j = wave + 4 * k;
                if (j >= a.npieces) j = a.npieces - 1;        // surplus slots repeat the last piece (identical data)
                const int o = j * 1024 + lane * 16;
                const int row = o / a.pitch, col = o - row * a.pitch - 32;
                const int ih = ih0 + row;
                const unsigned short* src = ximg + (int64_t)ih * (a.W * 3) + (col >> 1);
                if (row >= a.nrows || col < 0 || col >= rowbytes || (unsigned)ih >= (unsigned)a.H) src = a.zero;
                __builtin_amdgcn_global_load_lds((st_glb_void*)src, (st_lds_void*)(base + j * 1024), 16, 0, 0);
            }
        }
    };

    // ---- the filter: fragment (r, j) row i = output channel (j >> 1) * 32 + (i >> 2) * 8 + (j & 1) * 4 + (i & 3).
    // Blocks are persistent (two per CU walk the strip list): the 28 fragments are fetched once per block.
    const int i16 = lane & 15, qd = lane >> 4;
    st_s16x8 wf[ST_KH][2];
#pragma unroll
    for (int r = 0; r < ST_KH; ++r)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int och = chalf * 32 + (i16 >> 2) * 8 + j * 4 + (i16 & 3);
            wf[r][j] = *reinterpret_cast<const st_s16x8*>(a.wp + ((r * ST_C + och) * 32 + qd * 8));
        }
    float s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
    // the filter has landed before the counted waits below start counting
    __builtin_amdgcn_s_waitcnt(0x0F70);

    // A K step is 32 elements of an input row from element 6 ow - 10 on: element 0 is one before the window (zero
    // weight), 1 .. 21 the window, 22 .. 31 padding (zero weights; the quarter qd = 3 re-reads the qd = 2 chunk).
    // The spare elements are NOT masked: a non-finite pixel next to a window would reach that window's output, but
    // such a pixel already makes the batch statistics of every channel -- and with them the whole layer -- non-finite.
    const int ngr = (a.R * a.OW + 31) >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(st_lds_void*)st_smem;
    int strip = blockIdx.x;
    if (strip < a.nstrips) stage(strip, 0);
    int it = 0;
    for (; strip < a.nstrips; strip += gridDim.x, ++it) {
        const int buf = it & 1;
        const int next = strip + gridDim.x;
        if (next < a.nstrips) {
            stage(next, buf ^ 1);                  // streams in under this strip's MFMAs
            switch (a.npw) {                       // this strip's pieces were issued before the npw of the next one
#define ST_CASE(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
            ST_CASE(1) ST_CASE(2) ST_CASE(3) ST_CASE(4) ST_CASE(5) ST_CASE(6) ST_CASE(7) ST_CASE(8)
#undef ST_CASE
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int n = strip / a.strips_per_img, oh0 = (strip - n * a.strips_per_img) * a.R;
        // A fragment is 16 pixels of ONE column parity (linear index 32 g + 2 i + par over the strip's rows): element
        // e of a row sits at byte 32 + 2 e, so a K step starts at byte 12 ow + 12 (+ 16 qd) -- 8-byte aligned for odd
        // ow, 4 bytes past that for even ow.  With the parity uniform per wave (wave fg takes parity fg) the four dwords
        // come from fixed pieces: odd columns two aligned ds_read_b64, even columns b32 + b64 + b32 -- no per-lane
        // select.  (Per-lane parity cost 8 VALU operations per K step and made the kernel VALU-bound: 116 us without
        // its stores; ds_read_b64 moves 256 B/clk over 64 banks, dword reads 128 B/clk over 32.  Inline asm: the compiler
        // fuses plain 8-byte loads into the slow ds_read2_b64; every register an asm read writes must stay LIVE until
        // the wait -- the compiler reuses a dead one while the read is still in flight.)
        for (int g = 0; g < ngr; ++g) {
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            const int l0 = 32 * g;
            const int ohl0 = l0 / a.OW, ow0 = l0 - ohl0 * a.OW;      // wave-uniform
            int ow = ow0 + 2 * i16 + PAR, ohl = ohl0;
            if (ow >= a.OW) { ow -= a.OW; ++ohl; }                   // OW >= 32: a fragment spans at most two rows
            const bool inside = ohl < a.R;                           // the last fragment may run past the strip
            if (!inside) ohl = a.R - 1;
            const int qq = qd < 3 ? qd : 2;
            const unsigned ad = lds0 + (unsigned)(buf * a.bufbytes + (2 * ohl) * a.pitch + 12 * ow + 12 + 16 * qq);
            unsigned ea[4], ed[4];
            u32x2 eb[4], ec[4];
            st_s16x8 fx[4];
#define ST_READ(R0, NR)                                                                                   \
            _Pragma("unroll") for (int r = 0; r < NR; ++r) {                                               \
                const unsigned ar = ad + (unsigned)((R0 + r) * a.pitch);                                   \
                if (PAR) {                                                                                 \
                    asm volatile("ds_read_b64 %0, %1" : "=v"(eb[r]) : "v"(ar));                            \
                    asm volatile("ds_read_b64 %0, %1 offset:8" : "=v"(ec[r]) : "v"(ar));                   \
                } else {                                                                                   \
                    asm volatile("ds_read_b32 %0, %1" : "=v"(ea[r]) : "v"(ar));                            \
                    asm volatile("ds_read_b64 %0, %1 offset:4" : "=v"(eb[r]) : "v"(ar));                   \
                    asm volatile("ds_read_b32 %0, %1 offset:12" : "=v"(ed[r]) : "v"(ar));                  \
                }                                                                                          \
            }
#define ST_PICK(NR)                                                                                       \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            _Pragma("unroll") for (int r = 0; r < NR; ++r) {                                               \
                u32x4 v;                                                                                   \
                if (PAR) { v[0] = eb[r][0]; v[1] = eb[r][1]; v[2] = ec[r][0]; v[3] = ec[r][1]; }           \
                else { v[0] = ea[r]; v[1] = eb[r][0]; v[2] = eb[r][1]; v[3] = ed[r]; }                     \
                fx[r] = __builtin_bit_cast(st_s16x8, v);                                                   \
            }
#define ST_MFMA(R0, NR)                                                                                   \
            _Pragma("unroll") for (int r = 0; r < NR; ++r)                                                 \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                              \
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[R0 + r][j], fx[r], acc[j], 0, 0, 0);
            ST_READ(0, 4)
            ST_PICK(4)
            ST_READ(4, 3)
            __builtin_amdgcn_sched_barrier(0);
            ST_MFMA(0, 4)
            ST_PICK(3)
            ST_MFMA(4, 3)
#undef ST_READ
#undef ST_PICK
#undef ST_MFMA
            const int oh = oh0 + ohl;
            if (inside && oh < a.OH) {
                u32x4 o0;
                o0[0] = pack_bf16x2(acc[0][0], acc[0][1]); o0[1] = pack_bf16x2(acc[0][2], acc[0][3]);
                o0[2] = pack_bf16x2(acc[1][0], acc[1][1]); o0[3] = pack_bf16x2(acc[1][2], acc[1][3]);
                unsigned short* yp = a.y + DTF_IDX((((int64_t)n * a.OH + oh) * a.OW + ow) * ST_C + chalf * 32 + qd * 8,
                                                   (int64_t)a.NB * a.OH * a.OW * ST_C - 7, "K15 output strip edge");
                *reinterpret_cast<u32x4*>(yp) = o0;
                {
                // statistics of what the consumer reads: the rounded values
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float a0 = __uint_as_float(o0[k] << 16), a1 = __uint_as_float(o0[k] & 0xffff0000u);
                    s1[2 * k] += a0; s2[2 * k] = fmaf(a0, a0, s2[2 * k]);
                    s1[2 * k + 1] += a1; s2[2 * k + 1] = fmaf(a1, a1, s2[2 * k + 1]);
                }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();              // every wave is done with this buffer before it is restaged
    }

    // ---- per-block statistics row: lanes of a DPP row (same qd) -> the 4 fragment-lane waves of a channel half in order.
    // s[k]: channel chalf * 32 + qd * 8 + k
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* red = reinterpret_cast<float*>(st_smem);    // [2 fragment lanes][2][64]
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float t1 = st_row16_sum(s1[k]), t2 = st_row16_sum(s2[k]);
        if (i16 == 0) {
            const int c = chalf * 32 + qd * 8 + k;
            red[(fg * 2 + 0) * ST_C + c] = t1;
            red[(fg * 2 + 1) * ST_C + c] = t2;
        }
    }
    __syncthreads();
    if (tid < 2 * ST_C) {
        const int stat = tid >> 6, c = tid & 63;
        const float s = red[(0 * 2 + stat) * ST_C + c] + red[(1 * 2 + stat) * ST_C + c];
        a.partial[((int64_t)blockIdx.x * 2 + stat) * ST_C + c] = s;
    }
}

// The two parities are two instantiations of the same body: every wave executes the same sequence of barriers
// (s_barrier counts arrivals, whatever the program counter).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void stem_fwd_kernel(StemFwdArgs a) {
    if ((threadIdx.x >> 6) & 1) stem_fwd_body<1>(a);
    else stem_fwd_body<0>(a);
}

// ---------------------------------------------------------------------------
// Weight gradient with the BatchNorm backward applied on load:
//
//     dW[o][r][s][c] = sum_{n, oh, ow} dz[n, oh, ow, o] * X[n, 2 oh - 3 + r, 2 ow - 3 + s, c],
//     dz = k1[o] * dy' + k2[o] * z + k3[o],   dy' = dy where z * scale[o] + shift[o] > 0, else 0
//
// (z: the convolution output, dy: the gradient w.r.t. the BatchNorm + ReLU output, k1..k3: the K7 backward finalize's
// coefficients) -- the K7 apply pass that would write dz (read 2 x 411 MB, write 411 MB at batch 256) and the library
// kernel that would read it back (354 us) become ONE pass over dy and z.  GEMM view: M = 64 output channels, N = 147
// filter elements (10 fragments of 16), K = the pixels.  A block owns strips of R = 2 output rows of one image:
//   * dy / z / the 9 input rows of the NEXT strip are prefetched into registers while the current strip computes;
//   * A operand: dz in bf16 as [pixel][64 channels] LDS rows (32-byte chunks XOR-swizzled like K14), read with
//     ds_read_b64_tr_b16 -- the reduction index is the LDS row;
//   * B operand: the image rows are re-laid out in LDS as T[input row][t = 3 s + c][ow] = X[row][6 ow - 9 + t] (zero where
//     that leaves the row), so the 8 consecutive pixels a lane needs of one filter element are 16 contiguous bytes;
//   * wave w owns the filter-element fragments w, w + 4, w + 8 (12 or 8 accumulators: with 20 per wave plus the
//     prefetch registers the kernel needed 342 registers and one block per CU); one fp32 slab per block, summed in
//     fixed order by stem_wrw_reduce_kernel (deterministic).
// ---------------------------------------------------------------------------
#define SW_R 2            // output rows per strip
#define SW_NF 10          // filter-element fragments (16 columns each; 147 of the 160 columns are real)
#define SW_MAXK 8         // K steps (32 pixels) per strip at most
#define SW_RAWU 4         // 16-byte pieces of the input rows per thread and strip at most

struct StemWrwArgs {
    const unsigned short* dy; const unsigned short* z; const unsigned short* x;
    const float* scale; const float* shift; const float* coef; float* slab;
    int NB, H, W, OH, OW, strips_per_img, nstrips, ksteps, tpb, cpr, a_bytes, t_bytes;
};

typedef __attribute__((ext_vector_type(4))) short st_s16x4;

__device__ __forceinline__ int sw_swz(int pos) { return ((pos >> 1) & 1) | (((pos >> 3) & 1) << 1); }

template <int MAXK, int RAWU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void stem_wrw_kernel(StemWrwArgs a) {
    unsigned char* smem = reinterpret_cast<unsigned char*>(st_smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool has3 = wave < 2;                      // fragments wave, wave + 4 and (waves 0, 1) wave + 8
    const int OW = a.OW, njg = OW >> 3, nrows_in = 2 * SW_R + 5;
    const int rowelems = a.W * 3, rowbytes = a.W * 6;
    unsigned char* regA = smem;
    unsigned char* regT = smem + a.a_bytes;
    float* tab = reinterpret_cast<float*>(smem + a.a_bytes + a.t_bytes);      // [5][64]: scale, shift, k1, k2, k3
    for (int i = tid; i < 5 * ST_C; i += 256) {
        const int q = i >> 6, c = i & 63;
        tab[i] = q == 0 ? a.scale[c] : (q == 1 ? a.shift[c] : a.coef[(q - 2) * ST_C + c]);
    }

    // ---- strip-independent addressing
    const int i16 = lane & 15, qd = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const unsigned lds0 = (unsigned)(uintptr_t)(st_lds_void*)st_smem;
    unsigned offA[4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = 8 * qd + 4 * h + qq;
#pragma unroll
        for (int mg = 0; mg < 4; ++mg) offA[mg][h] = lds0 + (unsigned)(row * 128 + ((mg ^ sw_swz(row)) << 5) + 8 * pp);
    }
    unsigned kpart[MAXK], npart[3];
#pragma unroll
    for (int ks = 0; ks < MAXK; ++ks) {
        const int G = ks * 4 + qd;
        int ohl = G / njg, ow0 = (G - ohl * njg) * 8;
        if (ohl >= SW_R) { ohl = 0; ow0 = 0; }          // K steps past the strip: dz is zero there, any finite operand
        kpart[ks] = lds0 + (unsigned)(a.a_bytes + (2 * ohl * ST_T) * a.tpb + ow0 * 2);
    }
    bool ncol_ok = true;
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        int ng = (wave + 4 * f) * 16 + i16;
        if (f == 2) ncol_ok = ng < ST_KH * ST_T;
        if (ng > ST_KH * ST_T - 1) ng = ST_KH * ST_T - 1;
        npart[f] = (unsigned)(ng * a.tpb);
    }
    // the T image: 252 threads = 12 x 21: thread -> filter-row element t, (input row, 8-pixel group) pairs sub, sub + 12, ..
    const int tt = tid % ST_T, tsub = tid / ST_T;

    f32x4 acc[4][3];
#pragma unroll
    for (int mg = 0; mg < 4; ++mg)
#pragma unroll
        for (int f = 0; f < 3; ++f) acc[mg][f] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- register prefetch of one strip: dy / z (position (tid >> 3) + 32 u, channels 8 (tid & 7) ..) and the input rows
    u32x4 rdy[MAXK], rz[MAXK], rraw[RAWU];
    const int chunk8 = tid & 7, c0 = chunk8 * 8;
    auto prefetch = [&](int strip) {
        const int n = strip / a.strips_per_img, oh0 = (strip - n * a.strips_per_img) * SW_R;
        const int vrows = a.OH - oh0 < SW_R ? a.OH - oh0 : SW_R;
        // the input rows first: they are consumed first (vmcnt retires in order)
        const int ih0 = 2 * oh0 - 3;
#pragma unroll
        for (int u = 0; u < RAWU; ++u) {
            const int idx = tid + 256 * u;
            rraw[u] = (u32x4){0u, 0u, 0u, 0u};
            if (idx < nrows_in * a.cpr) {
                const int row = idx / a.cpr, ch = idx - row * a.cpr, ih = ih0 + row;
                if ((unsigned)ih < (unsigned)a.H)
                    rraw[u] = *reinterpret_cast<const u32x4*>(a.x + ((int64_t)n * a.H + ih) * rowelems + ch * 8);
            }
        }
        const int64_t base = (((int64_t)n * a.OH + oh0) * OW) * ST_C + c0;
#pragma unroll
        for (int u = 0; u < MAXK; ++u) {
            const int pos = (tid >> 3) + 32 * u;
            rdy[u] = (u32x4){0u, 0u, 0u, 0u};
            rz[u] = (u32x4){0u, 0u, 0u, 0u};
            if (u < a.ksteps && pos < vrows * OW) {
                rdy[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.dy + base + (int64_t)pos * ST_C));
                rz[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.z + base + (int64_t)pos * ST_C));
            }
        }
    };

    int strip = blockIdx.x;
    if (strip < a.nstrips) prefetch(strip);
    for (; strip < a.nstrips; strip += gridDim.x) {
        const int n = strip / a.strips_per_img, oh0 = (strip - n * a.strips_per_img) * SW_R;
        const int vrows = a.OH - oh0 < SW_R ? a.OH - oh0 : SW_R;
        // 1. the input rows, as they are, into region A
#pragma unroll
        for (int u = 0; u < RAWU; ++u) {
            const int idx = tid + 256 * u;
            if (idx < nrows_in * a.cpr) {
                const int row = idx / a.cpr, ch = idx - row * a.cpr;
                *reinterpret_cast<u32x4*>(regA + row * rowbytes + ch * 16) = rraw[u];
            }
        }
        __syncthreads();
        // 2. T[i][t][ow] = X[i][6 ow - 9 + t]
        if (tid < 12 * ST_T) {
            for (int ij = tsub; ij < nrows_in * njg; ij += 12) {
                const int i = ij / njg, jg = ij - i * njg;
                const unsigned short* rawrow = reinterpret_cast<const unsigned short*>(regA + i * rowbytes);
                const int e0 = 48 * jg + tt - 9;
                unsigned v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int e = e0 + 6 * k;
                    const bool ok = (unsigned)e < (unsigned)rowelems;
                    const unsigned val = rawrow[ok ? e : 0];
                    v[k] = ok ? val : 0u;
                }
                u32x4 o;
                o[0] = v[0] | (v[1] << 16); o[1] = v[2] | (v[3] << 16); o[2] = v[4] | (v[5] << 16); o[3] = v[6] | (v[7] << 16);
                *reinterpret_cast<u32x4*>(regT + (i * ST_T + tt) * a.tpb + jg * 16) = o;
            }
        }
        __syncthreads();
        // 3. dz from the prefetched dy / z, bf16, into region A (over the raw rows)
        {
            float sc[8], sh[8], k1[8], k2[8], k3[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                sc[k] = tab[0 * ST_C + c0 + k]; sh[k] = tab[1 * ST_C + c0 + k];
                k1[k] = tab[2 * ST_C + c0 + k]; k2[k] = tab[3 * ST_C + c0 + k]; k3[k] = tab[4 * ST_C + c0 + k];
            }
#pragma unroll
            for (int u = 0; u < MAXK; ++u) {
                if (u < a.ksteps) {
                    const int pos = (tid >> 3) + 32 * u;
                    u32x4 o = {0u, 0u, 0u, 0u};
                    if (pos < vrows * OW) {
                        float f[8];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float z0 = __uint_as_float(rz[u][k] << 16), z1 = __uint_as_float(rz[u][k] & 0xffff0000u);
                            float g0 = __uint_as_float(rdy[u][k] << 16), g1 = __uint_as_float(rdy[u][k] & 0xffff0000u);
                            g0 = fmaf(z0, sc[2 * k], sh[2 * k]) > 0.f ? g0 : 0.f;
                            g1 = fmaf(z1, sc[2 * k + 1], sh[2 * k + 1]) > 0.f ? g1 : 0.f;
                            f[2 * k] = fmaf(k1[2 * k], g0, fmaf(k2[2 * k], z0, k3[2 * k]));
                            f[2 * k + 1] = fmaf(k1[2 * k + 1], g1, fmaf(k2[2 * k + 1], z1, k3[2 * k + 1]));
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = pack_bf16x2(f[2 * k], f[2 * k + 1]);
                    }
                    *reinterpret_cast<u32x4*>(regA + pos * 128 + (((chunk8 >> 1) ^ sw_swz(pos)) << 5) + (chunk8 & 1) * 16) = o;
                }
            }
        }
        // 4. the next strip streams into the registers under the MFMAs
        if (strip + (int)gridDim.x < a.nstrips) prefetch(strip + gridDim.x);
        __syncthreads();
        // 5. the K steps
#pragma unroll
        for (int ks = 0; ks < MAXK; ++ks) {
            if (ks < a.ksteps) {
                st_s16x4 la[4], ha[4];
                st_s16x8 fb[3];
#pragma unroll
                for (int mg = 0; mg < 4; ++mg) {
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(la[mg]) : "v"(offA[mg][0]), "n"(ks * 4096));
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ha[mg]) : "v"(offA[mg][1]), "n"(ks * 4096));
                }
#pragma unroll
                for (int f = 0; f < 3; ++f)      // (waves 2, 3 read a third fragment they do not use: every asm output stays live)
                    asm volatile("ds_read_b128 %0, %1" : "=v"(fb[f]) : "v"(kpart[ks] + npart[f]));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (!ncol_ok) fb[2] = (st_s16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int mg = 0; mg < 4; ++mg) {
                    const st_s16x8 fa = __builtin_shufflevector(la[mg], ha[mg], 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[mg][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[0], acc[mg][0], 0, 0, 0);
                    acc[mg][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[1], acc[mg][1], 0, 0, 0);
                    if (has3) acc[mg][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[2], acc[mg][2], 0, 0, 0);
                }
            }
        }
        __syncthreads();            // every wave is done with regions A and T before the next strip rewrites them
    }

    // ---- one slab [64][160] per block.  C map of a fragment: row (output channel) = 16 mg + 4 qd + i, column (filter
    // element) = 16 (wave + 4 f) + i16
    float* out = a.slab + (int64_t)blockIdx.x * (ST_C * SW_NF * 16);
#pragma unroll
    for (int mg = 0; mg < 4; ++mg)
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            if (f == 2 && !has3) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                out[(mg * 16 + 4 * qd + i) * (SW_NF * 16) + (wave + 4 * f) * 16 + i16] = acc[mg][f][i];
        }
}

// dw[o][147] (bf16 / fp32, the channels-last parameter layout) = sum over the blocks' slabs in a fixed order: a block
// covers 64 float4 columns x 4 slab lanes (the K14 reduce)
__global__ __launch_bounds__(256) void stem_wrw_reduce_kernel(const float* __restrict__ slab, int nb, void* __restrict__ dw,
                                                              int out_dtype) {
    const int n4 = ST_C * SW_NF * 16 / 4;
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (col < n4) {
        const f32x4* p = reinterpret_cast<const f32x4*>(slab) + col;
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
    const int o = col / (SW_NF * 4), nq = (col - o * (SW_NF * 4)) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int nn = nq + i;
        if (nn >= ST_KH * ST_T) continue;
        if (out_dtype == DTF_BF16) reinterpret_cast<unsigned short*>(dw)[o * (ST_KH * ST_T) + nn] = f32_to_bf16(s[i]);
        else reinterpret_cast<float*>(dw)[o * (ST_KH * ST_T) + nn] = s[i];
    }
}

static int stem_blocks(int nstrips) {
    int cu = 0, dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
    static int per_cu = 0;
    if (per_cu == 0) { const char* e = getenv("DTF_STEM_BLOCKS_PER_CU"); per_cu = e ? atoi(e) : 3; if (per_cu < 1) per_cu = 3; }
    const int g = cu * per_cu;
    return g < nstrips ? g : nstrips;
}

static int stem_fwd_plan(int NB, int H, int W, StemFwdArgs* a, int* lds) {
    if (NB < 1 || H < 7 || W < 64 || (W & 7)) return -1;          // OW >= 32: a fragment spans at most two rows
    a->NB = NB; a->H = H; a->W = W;
    a->OH = (H + 6 - 7) / 2 + 1;
    a->OW = (W + 6 - 7) / 2 + 1;
    static int rows = 0;
    if (rows == 0) { const char* e = getenv("DTF_STEM_ROWS"); rows = e ? atoi(e) : 4; if (rows < 1 || rows > 8) rows = 4; }
    a->R = rows < a->OH ? rows : a->OH;
    a->strips_per_img = (a->OH + a->R - 1) / a->R;
    a->owf = (a->OW + 15) / 16;
    // row pitch: 32 B left padding + the row + right padding up to the last byte a (padded) window reads:
    // dword 3 ow + 3 + 4 * 2 + 4 of the last column fragment's last pixel
    int p = (48 * a->owf + 16) * 4;
    if (p < W * 6 + 64) p = W * 6 + 64;
    a->pitch = (p + 15) & ~15;
    a->nrows = 2 * a->R + 5;
    a->npieces = (a->nrows * a->pitch + 1023) / 1024;
    a->bufbytes = a->npieces * 1024;
    a->npw = (a->npieces + 3) / 4;
    if (a->npw > 8) return -1;
    *lds = 2 * a->bufbytes;
    if (*lds > 81920) return -1;
    if ((int64_t)NB * a->strips_per_img > 0x7fffffff) return -1;
    a->nstrips = NB * a->strips_per_img;
    return 0;
}


static int stem_wrw_plan(int NB, int H, int W, StemWrwArgs* a, int* lds, int* grid) {
    if (NB < 1 || H < 7 || W < 64 || (W & 15) || W > 256) return -1;
    a->NB = NB; a->H = H; a->W = W;
    a->OH = (H - 1) / 2 + 1;
    a->OW = W / 2;
    a->strips_per_img = (a->OH + SW_R - 1) / SW_R;
    a->nstrips = NB * a->strips_per_img;
    a->ksteps = (SW_R * a->OW + 31) / 32;
    if (a->ksteps > SW_MAXK) return -1;
    a->tpb = a->OW * 2 + 16;
    a->cpr = W * 6 / 16;
    if ((2 * SW_R + 5) * a->cpr > 256 * SW_RAWU) return -1;
    int ab = a->ksteps * 4096;
    if (ab < (2 * SW_R + 5) * W * 6) ab = (2 * SW_R + 5) * W * 6;
    a->a_bytes = (ab + 15) & ~15;
    a->t_bytes = (2 * SW_R + 5) * ST_T * a->tpb;
    *lds = a->a_bytes + a->t_bytes + 5 * ST_C * 4;
    if (*lds > 163840) return -1;
    int cu = 0, dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
    const int per_cu = 163840 / *lds >= 2 ? 2 : 1;
    *grid = cu * per_cu < a->nstrips ? cu * per_cu : a->nstrips;
    return 0;
}

extern "C" {

int dtf_stem_ok(int NB, int H, int W) {
    StemFwdArgs a{};
    int lds;
    return stem_fwd_plan(NB, H, W, &a, &lds) == 0;
}

// partial rows ([blocks][2][64] fp32) the forward writes
int64_t dtf_stem_fwd_blocks(int NB, int H, int W) {
    StemFwdArgs a{};
    int lds;
    if (stem_fwd_plan(NB, H, W, &a, &lds) != 0) return 0;
    return (int64_t)stem_blocks(a.nstrips);
}

// wp: [7][64][32] bf16 workspace (the repacked filter)
int dtf_stem_pack_w(const void* w, void* wp, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    hipLaunchKernelGGL(stem_pack_w_kernel, dim3(ST_KH * ST_C * 32 / 256), dim3(256), 0, stream, (const unsigned short*)w,
                       (unsigned short*)wp);
    DTF_CHECK_LAUNCH();
    return 0;
}

// zero: >= 16 bytes of zeros.  y[NB, OH, OW, 64] = conv7x7/2 (x[NB, H, W, 3], w); partial: dtf_stem_fwd_blocks rows of column sums / sums of squares of y
int dtf_stem_fwd(const void* x, const void* wp, void* y, float* partial, const void* zero, int NB, int H, int W,
                 hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    StemFwdArgs a{};
    int lds = 0;
    if (stem_fwd_plan(NB, H, W, &a, &lds) != 0) return -80;
    if ((((uintptr_t)x) | ((uintptr_t)wp) | ((uintptr_t)y) | ((uintptr_t)partial) | ((uintptr_t)zero)) & 15) return -16;
    a.zero = (const unsigned short*)zero;
    a.x = (const unsigned short*)x; a.wp = (const unsigned short*)wp; a.y = (unsigned short*)y; a.partial = partial;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)stem_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
        attr = true;
    }
    hipLaunchKernelGGL(stem_fwd_kernel, dim3(stem_blocks(a.nstrips)), dim3(256), lds, stream, a);
    DTF_CHECK_LAUNCH();
    return 0;
}

// 1 when the fused weight-gradient kernel covers this image size
int dtf_stem_wrw_ok(int NB, int H, int W) {
    StemWrwArgs a{};
    int lds, grid;
    return stem_wrw_plan(NB, H, W, &a, &lds, &grid) == 0;
}

// fp32 workspace floats: one [64][160] slab per block
int64_t dtf_stem_wrw_workspace(int NB, int H, int W) {
    StemWrwArgs a{};
    int lds, grid = 0;
    if (stem_wrw_plan(NB, H, W, &a, &lds, &grid) != 0) return 0;
    return (int64_t)grid * ST_C * SW_NF * 16;
}

// dw[64][7][7][3] (out_dtype: bf16 / fp32) from dy, z [NB, OH, OW, 64], x [NB, H, W, 3] and the BatchNorm backward's
// per-channel terms (scale, shift: the forward's; coef: [3][64] of the K7 backward finalize)
int dtf_stem_wrw(const void* dy, const void* z, const void* x, const float* scale, const float* shift, const float* coef,
                 void* dw, int out_dtype, float* workspace, int NB, int H, int W, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    StemWrwArgs a{};
    int lds = 0, grid = 0;
    if (stem_wrw_plan(NB, H, W, &a, &lds, &grid) != 0) return -80;
    if ((((uintptr_t)dy) | ((uintptr_t)z) | ((uintptr_t)x) | ((uintptr_t)workspace)) & 15) return -16;
    a.dy = (const unsigned short*)dy; a.z = (const unsigned short*)z; a.x = (const unsigned short*)x;
    a.scale = scale; a.shift = shift; a.coef = coef; a.slab = workspace;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)stem_wrw_kernel<7, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        (void)hipFuncSetAttribute((const void*)stem_wrw_kernel<SW_MAXK, SW_RAWU>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        attr = true;
    }
    if (a.ksteps <= 7 && (2 * SW_R + 5) * a.cpr <= 256 * 3)
        hipLaunchKernelGGL((stem_wrw_kernel<7, 3>), dim3(grid), dim3(256), lds, stream, a);
    else
        hipLaunchKernelGGL((stem_wrw_kernel<SW_MAXK, SW_RAWU>), dim3(grid), dim3(256), lds, stream, a);
    const int n4 = ST_C * SW_NF * 16 / 4;
    hipLaunchKernelGGL(stem_wrw_reduce_kernel, dim3((n4 + 63) / 64), dim3(256), 0, stream, workspace, grid, dw, out_dtype);
    DTF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
