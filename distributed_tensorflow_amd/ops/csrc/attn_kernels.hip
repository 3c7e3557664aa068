// K11: self-attention for BERT-sized heads on gfx950 -- head_dim 64, sequence length 128 / 256 / 384 / 512,
// optional key-padding mask (one valid length per sequence), no dropout (config #4).  One workgroup owns one (batch, head): K and V of the whole
// sequence stay resident in LDS (2 x 64 KB at S = 512), so they are read from HBM/L2 exactly once per
// head, and every wave computes COMPLETE score rows -- no online-softmax rescaling:
//
//   S^T = K . Q^T      (MFMA operands swapped: a lane owns ONE query column and, over the 32 key tiles,
//                       128 of its 512 scores in registers; the row maximum / sum are a register
//                       reduction plus two cross-lane steps)
//   P   = exp2((S^T - max) * scale * log2 e)      (unnormalised, bf16)
//   O^T = V^T . P^T    (the accumulator tile IS the next MFMA's B operand: a lane's 4 + 4 scores of two
//                       key tiles are the 8 k-elements of one 32-key step, in the permuted k order
//                       kappa(g, j) = 32 s + 16 (j >> 2) + 4 g + (j & 3); the matching A operand V^T comes
//                       from two ds_read_b64_tr_b16 of the row-major V image at rows 32 s + 4 g (+16))
//
// Q, K, V are read through row strides, so the packed projection output [B, S, 3, A, D] is consumed in
// place, and O is written token-major [B, S, A*D]: no transposes or copies around the kernel.
// The per-row log-sum-exp (exp2 domain) is stored for the backward kernels.
#include "dtf_common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) short at_s16x8;
typedef __attribute__((ext_vector_type(4))) short at_s16x4;
typedef __attribute__((address_space(3))) void at_lds_void;

#define AT_D 64
#define AT_MAXS 512
#define AT_QT 1          // query tiles per wave pass

struct AttnArgs {
    const unsigned short* Q; const unsigned short* K; const unsigned short* V;   // head (b, h) at + b*bs + h*AT_D
    unsigned short* O; float* LSE;        // O: [B, S, A*64]; LSE: [B, A, S]
    int64_t q_bs, k_bs, v_bs;             // batch strides (elements)
    int q_rs, k_rs, v_rs;                 // row (token) strides (elements)
    int B, A, S;
    float scale_log2e;
    const int* kv_len;                    // optional [B]: keys >= kv_len[b] are masked out (padding); null = none
};

__device__ __forceinline__ at_s16x4 at_tr_read(unsigned addr) {
    at_s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
// same, 16 rows (2048 B) further down the image
__device__ __forceinline__ at_s16x4 at_tr_read_16(unsigned addr) {
    at_s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(v) : "v"(addr));
    return v;
}

// K image: [key][128 B], 16-B chunk c of row r at chunk c ^ ((r >> 1) & 7): conflict-free ds_read_b128 of
// an A fragment (16 keys x 32 d).  V image: [key][128 B], 32-B chunk c of row r at c ^ ((r >> 1) & 3):
// conflict-free ds_read_b64_tr_b16 of a 4-key x 16-d block pair.
__device__ __forceinline__ int at_k_off(int row, int c16) { return row * 128 + ((c16 ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int at_v_off(int row, int c16) { return row * 128 + ((((c16 >> 1) ^ ((row >> 1) & 3)) << 5) | ((c16 & 1) << 4)); }

template <int NKT, int NT, bool MASKED>      // key tiles of 16 = S / 16; threads; key-padding mask compiled in
__global__ __launch_bounds__(NT) void attn_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Kimg = smem;
    unsigned char* Vimg = smem + AT_MAXS * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / a.A, h = blockIdx.x - b * a.A;
    const int S = a.S;
    const unsigned short* Qh = a.Q + b * a.q_bs + h * AT_D;
    const unsigned short* Kh = a.K + b * a.k_bs + h * AT_D;
    const unsigned short* Vh = a.V + b * a.v_bs + h * AT_D;

    // K, V of this head -> LDS (once): S rows x 8 chunks of 16 B each.  All loads of a batch are issued
    // before the first LDS store (a load -> store -> load chain pays the full memory latency per chunk:
    // ~30 us per head, more than the attention itself).
    constexpr int CH = (NKT * 16 * 8 + NT - 1) / NT;    // chunks per thread and matrix
    constexpr int LB = CH < 8 ? CH : 8;             // chunks in flight per batch
#pragma unroll
    for (int i0 = 0; i0 < CH; i0 += LB) {
        u32x4 kv[LB], vv[LB];
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            int i = (i0 + u) * NT + tid;
            if (i > NKT * 128 - 1) i = NKT * 128 - 1;          // clamp (duplicates rewrite the same bytes)
            const int row = i >> 3, c = i & 7;
            kv[u] = *reinterpret_cast<const u32x4*>(Kh + (int64_t)row * a.k_rs + c * 8);
            vv[u] = *reinterpret_cast<const u32x4*>(Vh + (int64_t)row * a.v_rs + c * 8);
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            int i = (i0 + u) * NT + tid;
            if (i > NKT * 128 - 1) i = NKT * 128 - 1;
            const int row = i >> 3, c = i & 7;
            *reinterpret_cast<u32x4*>(Kimg + at_k_off(row, c)) = kv[u];
            *reinterpret_cast<u32x4*>(Vimg + at_v_off(row, c)) = vv[u];
        }
    }
    __syncthreads();

    const int fr = lane & 15, g = lane >> 4;
    const unsigned vbase = (unsigned)(uintptr_t)(at_lds_void*)Vimg;
    // transposed-read lane address inside a [4 keys x 16 d] block: lane 16 g + 4 q + p -> row q, cols 4 p ..
    const int tq = (lane >> 2) & 3, tp = lane & 3;

    // Two passes over the keys per query tile, both streaming (no 128-register score tile, so several
    // waves fit per SIMD and one wave's softmax VALU work hides under another's MFMAs -- the kernel is
    // VALU-bound: PMC shows 2.3x more VALU than MFMA cycles):
    //   pass 1: S^T tiles -> row maximum only;
    //   pass 2: S^T tiles again (the extra 64 MFMAs are cheaper than holding the scores), p = exp2(.),
    //           row sum, bf16 P fragment, O^T += V^T . P^T.
    const float c = a.scale_log2e;
    const int klen = MASKED ? a.kv_len[b] : S;      // block-uniform
    for (int qt = wave; qt < NKT; qt += NT / 64) {
        const int q0 = qt * 16;
        // Q fragments as the B operand: lane (q = fr, group g) holds d = 32 ks + 8 g .. + 7
        at_s16x8 qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[ks] = *reinterpret_cast<const at_s16x8*>(Qh + (int64_t)(q0 + fr) * a.q_rs + ks * 32 + g * 8);

        float mx = -3.0e38f;
#pragma unroll
        for (int tb = 0; tb < NKT; tb += 4) {
            at_s16x8 kf[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    kf[u][ks] = *reinterpret_cast<const at_s16x8*>(Kimg + at_k_off((tb + u) * 16 + fr, ks * 4 + g));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[u][ks], qf[ks], sc, 0, 0, 0);
                if (MASKED && 16 * (tb + u + 1) > klen) {          // tile reaches into the padding (uniform branch)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[r] = (16 * (tb + u) + 4 * g + r < klen) ? sc[r] : -3.0e38f;
                }
                mx = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), mx);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mc = mx * c;

        float sum = 0.f;
        // transposed-read addresses: the image swizzle depends on (row >> 1) & 3 only, which neither the
        // 32-row step nor the +16-row second read changes -> one running address per d tile
        unsigned vb[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vb[dt] = vbase + at_v_off(4 * g + tq, dt * 2) + tp * 8;
        f32x4 oacc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int s = 0; s < NKT / 2; ++s) {
            at_s16x8 kf[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    kf[u][ks] = *reinterpret_cast<const at_s16x8*>(Kimg + at_k_off((2 * s + u) * 16 + fr, ks * 4 + g));
            at_s16x4 lo[4], hi[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                lo[dt] = at_tr_read(vb[dt]);
                hi[dt] = at_tr_read_16(vb[dt]);
                vb[dt] += 4096;                       // 32 keys further
            }
            f32x4 sc[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                sc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) sc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[u][ks], qf[ks], sc[u], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (MASKED && 16 * (2 * s + u + 1) > klen) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[u][r] = (16 * (2 * s + u) + 4 * g + r < klen) ? sc[u][r] : -3.0e38f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sc[u][r] = __builtin_amdgcn_exp2f(fmaf(sc[u][r], c, -mc));      // arguments <= 0
                    sum += sc[u][r];
                }
            }
            const u32x4 pk = {pack_bf16x2(sc[0][0], sc[0][1]), pack_bf16x2(sc[0][2], sc[0][3]),
                              pack_bf16x2(sc[1][0], sc[1][1]), pack_bf16x2(sc[1][2], sc[1][3])};
            const at_s16x8 pf = __builtin_bit_cast(at_s16x8, pk);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const at_s16x8 vf = __builtin_shufflevector(lo[dt], hi[dt], 0, 1, 2, 3, 4, 5, 6, 7);
                oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[dt], 0, 0, 0);
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        unsigned short* orow = a.O + ((int64_t)b * S + q0 + fr) * (a.A * AT_D) + h * AT_D;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            u32x2 pk;
            pk[0] = pack_bf16x2(oacc[dt][0] * inv, oacc[dt][1] * inv);
            pk[1] = pack_bf16x2(oacc[dt][2] * inv, oacc[dt][3] * inv);
            *reinterpret_cast<u32x2*>(orow + dt * 16 + g * 4) = pk;
        }
        if (g == 0) a.LSE[((int64_t)b * a.A + h) * S + q0 + fr] = mc + log2f(sum);
    }
}

// ---------------------------------------------------------------------------
// Backward: two kernels, both one workgroup per (batch, head) with two [S][64] images resident in LDS
// and both streaming over 32-wide steps of the other sequence axis; P is recomputed from the saved
// log-sum-exp, so no S x S tensor ever exists in memory.
//   dQ kernel   : K, V images.  Per 16-query tile (one per lane column, as in the forward):
//                 S^T = K.Q^T, dP^T = V.dO^T, dS^T = P^T o (dP^T - D) * scale, dQ^T += K^T . dS^T
//                 (K^T by transposing reads of the same K image).  Also writes D = rowsum(dO o O).
//   dK/dV kernel: Q, dO images.  Per 16-key tile: S = Q.K^T, dP = dO.V^T (rows = queries, so L and D are
//                 per register and come from a small LDS table), dV^T += dO^T . P, dK^T += Q^T . dS.
// Every accumulator tile feeds the next MFMA directly as its B operand (permuted-k trick of the forward).
// Each output element is written exactly once: no atomics, deterministic.
// dQ, dK, dV go straight into the packed [B, S, 3, A, 64] gradient of the fused QKV projection.
// ---------------------------------------------------------------------------
struct AttnBwdArgs {
    const unsigned short* Q; const unsigned short* K; const unsigned short* V;
    const unsigned short* O; const unsigned short* dO;      // [B, S, A*64]
    const float* LSE; float* Dsum;                            // [B, A, S]
    unsigned short* dQ; unsigned short* dK; unsigned short* dV;
    int64_t q_bs, k_bs, v_bs, dq_bs, dk_bs, dv_bs;
    int q_rs, k_rs, v_rs, dq_rs, dk_rs, dv_rs;
    int B, A, S;
    float scale_log2e, scale;
    const int* kv_len;
};

// dual-use image: [row][128 B], 32-B chunk c of row r at c ^ ((r >> 1) & 3); conflict-free for the
// ds_read_b128 fragment reads and for the ds_read_b64_tr_b16 block reads
template <int NKT, int NT>
__device__ __forceinline__ void at_stage_image(unsigned char* img, const unsigned short* src, int rs, int tid) {
    constexpr int CH = (NKT * 16 * 8 + NT - 1) / NT;
    constexpr int LB = CH < 8 ? CH : 8;
#pragma unroll
    for (int i0 = 0; i0 < CH; i0 += LB) {
        u32x4 v[LB];
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            int i = (i0 + u) * NT + tid;
            if (i > NKT * 128 - 1) i = NKT * 128 - 1;
            v[u] = *reinterpret_cast<const u32x4*>(src + (int64_t)(i >> 3) * rs + (i & 7) * 8);
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            int i = (i0 + u) * NT + tid;
            if (i > NKT * 128 - 1) i = NKT * 128 - 1;
            *reinterpret_cast<u32x4*>(img + at_v_off(i >> 3, i & 7)) = v[u];
        }
    }
}

__device__ __forceinline__ at_s16x8 at_pack8(const f32x4& a, const f32x4& b) {
    const u32x4 pk = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
    return __builtin_bit_cast(at_s16x8, pk);
}

__device__ __forceinline__ float at_dot8(const at_s16x8& x, const at_s16x8& y) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s = fmaf(bf16_to_f32((unsigned short)x[e]), bf16_to_f32((unsigned short)y[e]), s);
    return s;
}

template <int NKT, int NT, bool MASKED>
__global__ __launch_bounds__(NT) void attn_bwd_dq_kernel(AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Kimg = smem;
    unsigned char* Vimg = smem + AT_MAXS * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / a.A, h = blockIdx.x - b * a.A;
    const int S = a.S, H = a.A * AT_D;
    at_stage_image<NKT, NT>(Kimg, a.K + b * a.k_bs + h * AT_D, a.k_rs, tid);
    at_stage_image<NKT, NT>(Vimg, a.V + b * a.v_bs + h * AT_D, a.v_rs, tid);
    __syncthreads();

    const int fr = lane & 15, g = lane >> 4;
    const int tq = (lane >> 2) & 3, tp = lane & 3;
    const unsigned kbase = (unsigned)(uintptr_t)(at_lds_void*)Kimg;
    const float c = a.scale_log2e;
    const int klen = MASKED ? a.kv_len[b] : S;
    for (int qt = wave; qt < NKT; qt += NT / 64) {
        const int q = qt * 16 + fr;
        const unsigned short* qrow = a.Q + b * a.q_bs + h * AT_D + (int64_t)q * a.q_rs;
        const unsigned short* dorow = a.dO + ((int64_t)b * S + q) * H + h * AT_D;
        const unsigned short* orow = a.O + ((int64_t)b * S + q) * H + h * AT_D;
        at_s16x8 qf[2], dof[2];
        float dsum = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[ks] = *reinterpret_cast<const at_s16x8*>(qrow + ks * 32 + g * 8);
            dof[ks] = *reinterpret_cast<const at_s16x8*>(dorow + ks * 32 + g * 8);
            const at_s16x8 of = *reinterpret_cast<const at_s16x8*>(orow + ks * 32 + g * 8);
            dsum += at_dot8(dof[ks], of);
        }
        dsum += __shfl_xor(dsum, 16, 64);
        dsum += __shfl_xor(dsum, 32, 64);
        const int64_t li = ((int64_t)b * a.A + h) * S + q;
        if (g == 0) a.Dsum[li] = dsum;
        const float l2 = a.LSE[li];
        const float dss = dsum * a.scale;            // dS = P o (dP - D) * scale = P o fma(dP, scale, -D * scale)

        unsigned kb[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) kb[dt] = kbase + at_v_off(4 * g + tq, dt * 2) + tp * 8;
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int s = 0; s < NKT / 2; ++s) {
            at_s16x8 kf[2][2], vf[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int off = at_v_off((2 * s + u) * 16 + fr, ks * 4 + g);
                    kf[u][ks] = *reinterpret_cast<const at_s16x8*>(Kimg + off);
                    vf[u][ks] = *reinterpret_cast<const at_s16x8*>(Vimg + off);
                }
            at_s16x4 lo[4], hi[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                lo[dt] = at_tr_read(kb[dt]);
                hi[dt] = at_tr_read_16(kb[dt]);
                kb[dt] += 4096;
            }
            f32x4 sc[2], dp[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                sc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dp[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    sc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[u][ks], qf[ks], sc[u], 0, 0, 0);
                    dp[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[u][ks], dof[ks], dp[u], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = __builtin_amdgcn_exp2f(fmaf(sc[u][r], c, -l2));
                    if (MASKED && 16 * (2 * s + u + 1) > klen) p = (16 * (2 * s + u) + 4 * g + r < klen) ? p : 0.f;
                    sc[u][r] = p * fmaf(dp[u][r], a.scale, -dss);
                }
            const at_s16x8 dsf = at_pack8(sc[0], sc[1]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const at_s16x8 kt = __builtin_shufflevector(lo[dt], hi[dt], 0, 1, 2, 3, 4, 5, 6, 7);
                dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsf, dq[dt], 0, 0, 0);
            }
        }
        unsigned short* dqrow = a.dQ + b * a.dq_bs + h * AT_D + (int64_t)q * a.dq_rs;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            u32x2 pk;
            pk[0] = pack_bf16x2(dq[dt][0], dq[dt][1]);
            pk[1] = pack_bf16x2(dq[dt][2], dq[dt][3]);
            *reinterpret_cast<u32x2*>(dqrow + dt * 16 + g * 4) = pk;
        }
    }
}

template <int NKT, int NT, bool MASKED>
__global__ __launch_bounds__(NT) void attn_bwd_dkv_kernel(AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Qimg = smem;
    unsigned char* Gimg = smem + AT_MAXS * 128;                 // dO
    float* Ls = reinterpret_cast<float*>(smem + 2 * AT_MAXS * 128);
    float* Ds = Ls + AT_MAXS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / a.A, h = blockIdx.x - b * a.A;
    const int S = a.S, H = a.A * AT_D;
    at_stage_image<NKT, NT>(Qimg, a.Q + b * a.q_bs + h * AT_D, a.q_rs, tid);
    at_stage_image<NKT, NT>(Gimg, a.dO + (int64_t)b * S * H + h * AT_D, H, tid);
    for (int i = tid; i < S; i += NT) {
        Ls[i] = a.LSE[((int64_t)b * a.A + h) * S + i];
        Ds[i] = a.Dsum[((int64_t)b * a.A + h) * S + i] * a.scale;
    }
    __syncthreads();

    const int fr = lane & 15, g = lane >> 4;
    const int tq = (lane >> 2) & 3, tp = lane & 3;
    const unsigned qbase = (unsigned)(uintptr_t)(at_lds_void*)Qimg;
    const unsigned gbase = (unsigned)(uintptr_t)(at_lds_void*)Gimg;
    const float c = a.scale_log2e;
    const int klen = MASKED ? a.kv_len[b] : S;
    for (int kt = wave; kt < NKT; kt += NT / 64) {
        const int key = kt * 16 + fr;
        const bool kvalid = !MASKED || key < klen;           // padded keys take no probability mass: dK = dV = 0
        const unsigned short* krow = a.K + b * a.k_bs + h * AT_D + (int64_t)key * a.k_rs;
        const unsigned short* vrow = a.V + b * a.v_bs + h * AT_D + (int64_t)key * a.v_rs;
        at_s16x8 kf[2], vf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[ks] = *reinterpret_cast<const at_s16x8*>(krow + ks * 32 + g * 8);
            vf[ks] = *reinterpret_cast<const at_s16x8*>(vrow + ks * 32 + g * 8);
        }
        unsigned tb[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) tb[dt] = qbase + at_v_off(4 * g + tq, dt * 2) + tp * 8;
        const unsigned gdelta = gbase - qbase;
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 2
        for (int s = 0; s < NKT / 2; ++s) {
            at_s16x8 qa[2][2], ga[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int off = at_v_off((2 * s + u) * 16 + fr, ks * 4 + g);
                    qa[u][ks] = *reinterpret_cast<const at_s16x8*>(Qimg + off);
                    ga[u][ks] = *reinterpret_cast<const at_s16x8*>(Gimg + off);
                }
            at_s16x4 qlo[4], qhi[4], glo[4], ghi[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                qlo[dt] = at_tr_read(tb[dt]);
                qhi[dt] = at_tr_read_16(tb[dt]);
                glo[dt] = at_tr_read(tb[dt] + gdelta);
                ghi[dt] = at_tr_read_16(tb[dt] + gdelta);
                tb[dt] += 4096;
            }
            f32x4 sc[2], dp[2], l2[2], dd[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                sc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dp[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {       // rows = queries 32 s + 16 u + 4 g + r, column = this lane's key
                    sc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[u][ks], kf[ks], sc[u], 0, 0, 0);
                    dp[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga[u][ks], vf[ks], dp[u], 0, 0, 0);
                }
                l2[u] = *reinterpret_cast<const f32x4*>(Ls + 32 * s + 16 * u + 4 * g);
                dd[u] = *reinterpret_cast<const f32x4*>(Ds + 32 * s + 16 * u + 4 * g);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = __builtin_amdgcn_exp2f(fmaf(sc[u][r], c, -l2[u][r]));
                    if (MASKED) p = kvalid ? p : 0.f;
                    sc[u][r] = p;
                    dp[u][r] = p * fmaf(dp[u][r], a.scale, -dd[u][r]);
                }
            const at_s16x8 pf = at_pack8(sc[0], sc[1]);
            const at_s16x8 dsf = at_pack8(dp[0], dp[1]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const at_s16x8 gt = __builtin_shufflevector(glo[dt], ghi[dt], 0, 1, 2, 3, 4, 5, 6, 7);
                const at_s16x8 qt2 = __builtin_shufflevector(qlo[dt], qhi[dt], 0, 1, 2, 3, 4, 5, 6, 7);
                dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt, pf, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt2, dsf, dk[dt], 0, 0, 0);
            }
        }
        unsigned short* dkrow = a.dK + b * a.dk_bs + h * AT_D + (int64_t)key * a.dk_rs;
        unsigned short* dvrow = a.dV + b * a.dv_bs + h * AT_D + (int64_t)key * a.dv_rs;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            u32x2 pk;
            pk[0] = pack_bf16x2(dk[dt][0], dk[dt][1]);
            pk[1] = pack_bf16x2(dk[dt][2], dk[dt][3]);
            *reinterpret_cast<u32x2*>(dkrow + dt * 16 + g * 4) = pk;
            pk[0] = pack_bf16x2(dv[dt][0], dv[dt][1]);
            pk[1] = pack_bf16x2(dv[dt][2], dv[dt][3]);
            *reinterpret_cast<u32x2*>(dvrow + dt * 16 + g * 4) = pk;
        }
    }
}

extern "C" {

// q, k, v: bf16, head (b, h) of tensor T starts at T + b*T_bs + h*64, token rows T_rs elements apart.
// o: bf16 [B, S, A*64]; lse: fp32 [B, A, S] (log2 domain, includes the scale).
int dtf_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int64_t q_bs, int64_t k_bs,
                 int64_t v_bs, int q_rs, int k_rs, int v_rs, int B, int A, int S, int D, float scale,
                 const int* kv_len, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (D != AT_D || (S != 512 && S != 384 && S != 256 && S != 128) || B <= 0 || A <= 0) return -70;
    if ((q_rs | k_rs | v_rs) & 7) return -71;
    if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)o)) & 15) return -16;
    AttnArgs a{(const unsigned short*)q, (const unsigned short*)k, (const unsigned short*)v, (unsigned short*)o, lse,
               q_bs, k_bs, v_bs, q_rs, k_rs, v_rs, B, A, S, scale * 1.4426950408889634f, kv_len};
    static bool attr_set = false;
    static int nt = 1024;
    const int lds = 2 * AT_MAXS * 128;
#define AT_FOR_ALL(X) X(32, 512) X(24, 512) X(16, 512) X(8, 512) X(32, 1024) X(24, 1024) X(16, 1024) X(8, 1024)
    if (!attr_set) {
#define AT_ATTR(N, T)                                                                                                  \
        hipFuncSetAttribute((const void*)attn_fwd_kernel<N, T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        hipFuncSetAttribute((const void*)attn_fwd_kernel<N, T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        AT_FOR_ALL(AT_ATTR)
#undef AT_ATTR
        const char* e = getenv("DTF_ATTN_THREADS");
        if (e && atoi(e) == 512) nt = 512;
        attr_set = true;
    }
#define AT_LAUNCH(N, T)                                                                                        \
    if (S == N * 16 && nt == T) {                                                                              \
        if (kv_len) hipLaunchKernelGGL((attn_fwd_kernel<N, T, true>), dim3(B * A), dim3(T), lds, stream, a);   \
        else hipLaunchKernelGGL((attn_fwd_kernel<N, T, false>), dim3(B * A), dim3(T), lds, stream, a);         \
    }
    AT_FOR_ALL(AT_LAUNCH)
#undef AT_LAUNCH
#undef AT_FOR_ALL
    DTF_CHECK_LAUNCH();
    return 0;
}

// Backward.  o, do_: bf16 [B, S, A*64]; lse from the forward; dsum: fp32 [B, A, S] workspace;
// dq / dk / dv: bf16, strided like q / k / v (e.g. slices of a packed [B, S, 3, A, 64] gradient).
int dtf_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* do_, const float* lse,
                 float* dsum, void* dq, void* dk, void* dv, int64_t q_bs, int64_t k_bs, int64_t v_bs, int q_rs, int k_rs,
                 int v_rs, int64_t dq_bs, int64_t dk_bs, int64_t dv_bs, int dq_rs, int dk_rs, int dv_rs, int B, int A,
                 int S, int D, float scale, const int* kv_len, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (D != AT_D || (S != 512 && S != 384 && S != 256 && S != 128) || B <= 0 || A <= 0) return -70;
    if ((q_rs | k_rs | v_rs | dq_rs | dk_rs | dv_rs) & 7) return -71;
    if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)o) | ((uintptr_t)do_) | ((uintptr_t)dq) |
         ((uintptr_t)dk) | ((uintptr_t)dv)) & 15) return -16;
    AttnBwdArgs a{(const unsigned short*)q, (const unsigned short*)k, (const unsigned short*)v, (const unsigned short*)o,
                  (const unsigned short*)do_, lse, dsum, (unsigned short*)dq, (unsigned short*)dk, (unsigned short*)dv,
                  q_bs, k_bs, v_bs, dq_bs, dk_bs, dv_bs, q_rs, k_rs, v_rs, dq_rs, dk_rs, dv_rs, B, A, S,
                  scale * 1.4426950408889634f, scale, kv_len};
    static bool attr_set = false;
    static int dq_nt = 1024;
    const int lds1 = 2 * AT_MAXS * 128, lds2 = 2 * AT_MAXS * 128 + 2 * AT_MAXS * 4;
#define AT_FOR_S(X) X(32) X(24) X(16) X(8)
    if (!attr_set) {
#define AT_ATTR1(N, M)                                                                                                  \
        hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<N, 512, M>, hipFuncAttributeMaxDynamicSharedMemorySize, lds1);  \
        hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<N, 1024, M>, hipFuncAttributeMaxDynamicSharedMemorySize, lds1); \
        hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<N, 512, M>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
#define AT_ATTR(N) AT_ATTR1(N, false) AT_ATTR1(N, true)
        AT_FOR_S(AT_ATTR)
#undef AT_ATTR
#undef AT_ATTR1
        const char* e = getenv("DTF_ATTN_DQ_THREADS");
        if (e && atoi(e) == 512) dq_nt = 512;
        attr_set = true;
    }
#define AT_LAUNCH1(N, M)                                                                                                \
        if (dq_nt == 1024) hipLaunchKernelGGL((attn_bwd_dq_kernel<N, 1024, M>), dim3(B * A), dim3(1024), lds1, stream, a); \
        else hipLaunchKernelGGL((attn_bwd_dq_kernel<N, 512, M>), dim3(B * A), dim3(512), lds1, stream, a);              \
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<N, 512, M>), dim3(B * A), dim3(512), lds2, stream, a);
#define AT_LAUNCH(N)                                  \
    if (S == N * 16) {                                \
        if (kv_len) { AT_LAUNCH1(N, true) }           \
        else { AT_LAUNCH1(N, false) }                 \
    }
    AT_FOR_S(AT_LAUNCH)
#undef AT_LAUNCH1
#undef AT_LAUNCH
#undef AT_FOR_S
    DTF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
