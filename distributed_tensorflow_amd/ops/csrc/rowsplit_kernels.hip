// K20: the seam between the fused embedding rows and the deep tower of Wide&Deep (config #3), one pass each way.
//
// A fused table row is [D deep values | wide weight | pad] (D + 8 bf16 = 9 chunks of 16 B at D = 64).  The model needs
//   h    [B][F*D + tail]   the deep tower's input row: the F deep vectors, then `tail` more columns = the dense
//                          features followed by zeros up to a whole number of GEMM K tiles
//   wide [B]               fp32 sum over the F features of the wide weights
// and backward the reverse: de [B][F][D + 8] from d_h and d_wide.  As tensor expressions (strided slice copies, a strided
// sum, three slice writes into the gradient) these were ten elementwise launches and 330 us of a 3 ms step, most of
// them touching one or seven elements of every 144-B row; here each direction reads its input once in 16-B chunks and
// writes its output once.
//
// Mapping: one wave per batch row, chunk t of the row's F * (D/8 + 1) chunks on lane t % 64 (consecutive lanes read
// consecutive 16-B chunks of e / write consecutive chunks of de); the lanes that hold a feature's wide chunk add its
// first element to a per-lane partial, one xor-shuffle tree per row sums them in a fixed order (deterministic).
#include "dtf_common.h"

__global__ __launch_bounds__(256) void emb_split_fwd_kernel(const unsigned short* __restrict__ e,
                                                            const unsigned short* __restrict__ dense,
                                                            unsigned short* __restrict__ h, float* __restrict__ wide,
                                                            int64_t B, int F, int D, int nd, int tail, int64_t ldh) {
    const int lane = threadIdx.x & 63;
    const int cpf = D / 8 + 1;                       // chunks per feature
    const int nchunk = F * cpf, ntail = tail / 8;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t b = wave0; b < B; b += nwave) {
        const unsigned short* er = e + b * (int64_t)nchunk * 8;
        unsigned short* hr = h + b * ldh;
        float w = 0.f;
        for (int t = lane; t < nchunk; t += 64) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(er + (int64_t)t * 8);
            const int f = t / cpf, c = t - f * cpf;
            if (c < cpf - 1) *reinterpret_cast<u32x4*>(hr + (int64_t)f * D + c * 8) = v;
            else w += __uint_as_float(v[0] << 16);
        }
        // tail columns: dense features, then zeros (lanes 0 .. ntail-1 build one 16-B chunk each)
        if (lane < ntail) {
            unsigned short t8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = lane * 8 + k;
                t8[k] = j < nd ? dense[b * nd + j] : (unsigned short)0;
            }
            u32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (unsigned)t8[2 * k] | ((unsigned)t8[2 * k + 1] << 16);
            *reinterpret_cast<u32x4*>(hr + (int64_t)F * D + lane * 8) = v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o, 64);
        if (lane == 0) wide[b] = w;
    }
}

__global__ __launch_bounds__(256) void emb_split_bwd_kernel(const unsigned short* __restrict__ dh,
                                                            const float* __restrict__ dwide,
                                                            unsigned short* __restrict__ de, int64_t B, int F, int D,
                                                            int64_t lddh) {
    const int lane = threadIdx.x & 63;
    const int cpf = D / 8 + 1;
    const int nchunk = F * cpf;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t b = wave0; b < B; b += nwave) {
        const unsigned short* hr = dh ? dh + b * lddh : nullptr;
        unsigned short* er = de + b * (int64_t)nchunk * 8;
        const unsigned dw = dwide ? (unsigned)f32_to_bf16(dwide[b]) : 0u;
        for (int t = lane; t < nchunk; t += 64) {
            const int f = t / cpf, c = t - f * cpf;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (c < cpf - 1) { if (hr) v = *reinterpret_cast<const u32x4*>(hr + (int64_t)f * D + c * 8); }
            else v[0] = dw;
            *reinterpret_cast<u32x4*>(er + (int64_t)t * 8) = v;
        }
    }
}

static inline int rowsplit_grid(int64_t B) {
    int64_t blocks = (B + 3) / 4;                    // 4 waves = 4 rows per block pass
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

extern "C" {

// 0 when the layout is one the kernels handle
int dtf_emb_split_ok(int64_t B, int F, int D, int W, int nd, int tail, int64_t ldh) {
    if (B <= 0 || F <= 0 || D <= 0 || (D & 7) || W != D + 8) return -60;
    if (tail < 0 || (tail & 7) || nd < 0 || nd > tail || tail > 512) return -61;
    if (ldh != (int64_t)F * D + tail || (ldh & 7)) return -62;
    return 0;
}

int dtf_emb_split_fwd(const void* e, const void* dense, void* h, float* wide, int64_t B, int F, int D, int nd, int tail,
                      int64_t ldh, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    const int rc = dtf_emb_split_ok(B, F, D, D + 8, nd, tail, ldh);
    if (rc) return rc;
    if ((((uintptr_t)e) | ((uintptr_t)h)) & 15) return -16;
    if (nd > 0 && !dense) return -63;
    hipLaunchKernelGGL(emb_split_fwd_kernel, dim3(rowsplit_grid(B)), dim3(256), 0, stream, (const unsigned short*)e,
                       (const unsigned short*)dense, (unsigned short*)h, wide, B, F, D, nd, tail, ldh);
    DTF_CHECK_LAUNCH();
    return 0;
}

// dh / dwide may be null (that output of the forward was not used): the matching part of de is zero
int dtf_emb_split_bwd(const void* dh, const float* dwide, void* de, int64_t B, int F, int D, int64_t lddh,
                      hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (B <= 0 || F <= 0 || D <= 0 || (D & 7)) return -60;
    if (dh && (lddh < (int64_t)F * D || (lddh & 7))) return -62;
    if ((((uintptr_t)dh) | ((uintptr_t)de)) & 15) return -16;
    hipLaunchKernelGGL(emb_split_bwd_kernel, dim3(rowsplit_grid(B)), dim3(256), 0, stream, (const unsigned short*)dh,
                       dwide, (unsigned short*)de, B, F, D, lddh);
    DTF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
