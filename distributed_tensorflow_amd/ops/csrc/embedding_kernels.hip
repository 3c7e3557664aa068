// K3 / K4: embedding lookup and sparse gradient apply for gfx950.
// They replace what the TF1 PS does for partitioned embedding variables:
// Gather / embedding_lookup, and ScatterAdd / SparseApplyAdagrad (externals X3/X4).
//
// K3  embedding_gather      out[i,:] = sum_{j<L} table[idx[i*L+j], :]  (L = bag length, 1 = plain)
//     one 16-B (8 x bf16) or 32-B (8 x fp32) chunk per lane, consecutive lanes walk
//     one row, so a wave reads whole rows as contiguous 128-1024 B segments; tables are
//     far larger than the 256 MiB Infinity Cache, i.e. this runs at the HBM random-row rate.
//
// K4  two variants, chosen by the caller:
//  (a) deterministic sort/segment form (any optimizer; bitwise reproducible):
//      ids are sorted (rocPRIM radix sort through torch.sort); the sorted run of each id
//      is cut into chunks of <= EMB_CHUNK rows; `emb_chunk_partial` sums each chunk's
//      gradient rows in fixed order into an fp32 slab (long runs of a hot id are spread
//      over many lane groups); `emb_segment_apply` adds the slabs of each id in order and
//      applies SGD / Adagrad / row-wise Adagrad with ONE plain store per unique row.
//  (b) atomic form (SGD on fp32 tables only): global_atomic_add_f32 with one dword per
//      lane and consecutive lanes on consecutive floats of a row (256 contiguous bytes per
//      wave-instruction for D >= 64 -- the shape that reaches the ~1.3 TB/s atomic rate;
//      one-lane-per-row would be ~17x slower).  Not reproducible bit-for-bit.
#include "dtf_common.h"

#define EMB_CHUNK 64

enum { EMB_OPT_SGD = 0, EMB_OPT_ADAGRAD = 1, EMB_OPT_ROWWISE_ADAGRAD = 2 };

template <int TDT, int ODT>
__global__ __launch_bounds__(256) void embedding_gather_kernel(const void* __restrict__ table,
                                                               const int64_t* __restrict__ idx,
                                                               void* __restrict__ out, int64_t nbags, int bag,
                                                               int D, int64_t R, float scale) {
    const int dv = D >> 3;
    const int64_t total = nbags * dv;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t b = t / dv;
        const int v = (int)(t - b * dv);
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int j = 0; j < bag; ++j) {
            const int64_t r = idx[b * bag + j];
            if (r < 0 || r >= R) continue;              // padding id
            float x[8];
            load8<TDT>(table, r * D + (v << 3), x);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += x[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] *= scale;
        store8<ODT>(out, b * D + (v << 3), acc);
    }
}

// any D (e.g. the D = 1 "wide" weights): one element per lane
template <int TDT, int ODT>
__global__ __launch_bounds__(256) void embedding_gather_scalar_kernel(const void* __restrict__ table,
                                                                      const int64_t* __restrict__ idx,
                                                                      void* __restrict__ out, int64_t nbags,
                                                                      int bag, int D, int64_t R, float scale) {
    const int64_t total = nbags * D;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t b = t / D;
        const int d = (int)(t - b * D);
        float acc = 0.f;
        for (int j = 0; j < bag; ++j) {
            const int64_t r = idx[b * bag + j];
            if (r < 0 || r >= R) continue;
            acc += load1<TDT>(table, r * D + d);
        }
        store1<ODT>(out, b * D + d, acc * scale);
    }
}

// (a.1) partial[c, :] = sum over sorted positions j in [chunk_begin[c], chunk_begin[c+1]) of
//       grad[perm[j] / bag, :]   (fixed order).  chunk_begin has n+1 entries, unused = n.
template <int GDT>
__global__ __launch_bounds__(256) void emb_chunk_partial_kernel(const void* __restrict__ grad,
                                                                const int64_t* __restrict__ perm,
                                                                const int64_t* __restrict__ chunk_begin,
                                                                float* __restrict__ partial, int64_t n, int D,
                                                                int bag) {
    const int dv = (D + 7) >> 3;
    const int64_t total = n * dv;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool vec = (D & 7) == 0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t c = t / dv;
        const int v = (int)(t - c * dv);
        const int64_t j0 = chunk_begin[c];
        if (j0 >= n) continue;                       // beyond the last chunk
        const int64_t j1 = chunk_begin[c + 1];
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int64_t j = j0; j < j1; ++j) {
            const int64_t src = perm[j] / bag;
            if (vec) {
                float x[8];
                load8<GDT>(grad, src * D + (v << 3), x);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += x[k];
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if ((v << 3) + k < D) acc[k] += load1<GDT>(grad, src * D + (v << 3) + k);
            }
        }
        if (vec) store8<DTF_F32>(partial, c * D + (v << 3), acc);
        else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if ((v << 3) + k < D) partial[c * D + (v << 3) + k] = acc[k];
        }
    }
}

struct EmbApplyArgs {
    void* table; float* acc;                       // acc: [R, D] (Adagrad) or [R] (row-wise) or null
    const float* partial;
    const int64_t* sorted_ids; const int64_t* seg_begin; const int64_t* seg_first_chunk;
    const int64_t* nchunks_ptr;                    // device scalar: number of chunks
    int64_t n, R; int D, opt;
    float lr, eps, grad_scale, wd;
};

// (a.2) one lane group per unique id: g = ordered sum of its chunk slabs, then the optimizer.
template <int TDT>
__global__ __launch_bounds__(256) void emb_segment_apply_kernel(EmbApplyArgs a) {
    const int D = a.D;
    const int dv = (D + 7) >> 3;
    const int64_t total = a.n * dv;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t nchunks = *a.nchunks_ptr;
    const bool vec = (D & 7) == 0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t s = t / dv;
        const int v = (int)(t - s * dv);
        const int64_t j0 = a.seg_begin[s];
        if (j0 >= a.n) continue;                    // beyond the last unique id
        const int64_t row = a.sorted_ids[j0];
        if (row < 0 || row >= a.R) continue;        // padding id: no update
        const int64_t c0 = a.seg_first_chunk[s];
        int64_t c1 = a.seg_first_chunk[s + 1];
        if (c1 > nchunks) c1 = nchunks;
        float g[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = 0.f;
        for (int64_t c = c0; c < c1; ++c) {
            if (vec) {
                float x[8];
                load8<DTF_F32>(a.partial, c * D + (v << 3), x);
#pragma unroll
                for (int k = 0; k < 8; ++k) g[k] += x[k];
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if ((v << 3) + k < D) g[k] += a.partial[c * D + (v << 3) + k];
            }
        }
        float w[8];
        const int64_t off = row * D + (v << 3);
        if (vec) load8<TDT>(a.table, off, w);
        else {
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = ((v << 3) + k < D) ? load1<TDT>(a.table, off + k) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = fmaf(g[k], a.grad_scale, a.wd * w[k]);
        if (a.opt == EMB_OPT_SGD) {
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = fmaf(-a.lr, g[k], w[k]);
        } else if (a.opt == EMB_OPT_ADAGRAD) {
            float h[8];
            if (vec) load8<DTF_F32>(a.acc, off, h);
            else {
#pragma unroll
                for (int k = 0; k < 8; ++k) h[k] = ((v << 3) + k < D) ? a.acc[off + k] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                h[k] = fmaf(g[k], g[k], h[k]);
                w[k] = fmaf(-a.lr, g[k] * rsqrtf(h[k] + a.eps), w[k]);
            }
            if (vec) store8<DTF_F32>(a.acc, off, h);
            else {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if ((v << 3) + k < D) a.acc[off + k] = h[k];
            }
        } else {
            // row-wise Adagrad: one accumulator per row = mean over the row of g^2.  The lanes of
            // a row's group reduce their partial sums through shuffles (dv <= 64, power of two
            // group widths are padded by the xor pattern staying inside the group when dv is a
            // power of two; otherwise fall back to a serial read of the other lanes).
            float ss = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) ss = ((v << 3) + k < D) ? fmaf(g[k], g[k], ss) : ss;
            float tot = ss;
            if ((dv & (dv - 1)) == 0 && dv <= 64) {
                for (int o = 1; o < dv; o <<= 1) tot += __shfl_xor(tot, o, 64);
            } else {
                const int lane = threadIdx.x & 63;
                const int base = lane - v;
                tot = 0.f;
                for (int q = 0; q < dv; ++q) {
                    const int src = base + q;
                    // lanes of this group may straddle a wave boundary only when dv does not
                    // divide 64; the host refuses that shape for the row-wise optimizer
                    tot += __shfl(ss, src & 63, 64);
                }
            }
            const float hnew = a.acc[row] + tot / (float)D;
            const float rs = rsqrtf(hnew + a.eps);
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = fmaf(-a.lr, g[k] * rs, w[k]);
            if (v == 0) a.acc[row] = hnew;
        }
        if (vec) store8<TDT>(a.table, off, w);
        else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if ((v << 3) + k < D) store1<TDT>(a.table, off + k, w[k]);
        }
    }
}

// (b) SGD by float atomics: table[idx[i], d] -= lr * grad[i / bag, d]
template <int GDT>
__global__ __launch_bounds__(256) void emb_sgd_atomic_kernel(float* __restrict__ table,
                                                             const int64_t* __restrict__ idx,
                                                             const void* __restrict__ grad, int64_t n, int D,
                                                             int bag, int64_t R, float neg_lr_scale) {
    const int64_t total = n * D;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t i = t / D;
        const int d = (int)(t - i * D);
        const int64_t r = idx[i];
        if (r < 0 || r >= R) continue;
        const float g = load1<GDT>(grad, (i / bag) * D + d);
        atomicAdd(table + r * D + d, g * neg_lr_scale);
    }
}

extern "C" {

int dtf_embedding_gather(const void* table, int table_dtype, const int64_t* idx, void* out, int out_dtype,
                         int64_t nbags, int bag, int D, int64_t R, float scale, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (nbags <= 0) return 0;
    const bool vec = (D % 8 == 0) && ((((uintptr_t)table) | ((uintptr_t)out)) & 15) == 0;
    const int64_t items = vec ? nbags * (D / 8) : nbags * D;
    const int grid = dtf_stream_grid(items, 256);
#define CALL(T, O)                                                                                              \
    do {                                                                                                        \
        if (vec) hipLaunchKernelGGL((embedding_gather_kernel<T, O>), dim3(grid), dim3(256), 0, stream, table,   \
                                    idx, out, nbags, bag, D, R, scale);                                         \
        else hipLaunchKernelGGL((embedding_gather_scalar_kernel<T, O>), dim3(grid), dim3(256), 0, stream,       \
                                table, idx, out, nbags, bag, D, R, scale);                                      \
    } while (0)
    if (table_dtype == DTF_BF16 && out_dtype == DTF_BF16) CALL(DTF_BF16, DTF_BF16);
    else if (table_dtype == DTF_BF16) CALL(DTF_BF16, DTF_F32);
    else if (out_dtype == DTF_BF16) CALL(DTF_F32, DTF_BF16);
    else CALL(DTF_F32, DTF_F32);
#undef CALL
    DTF_CHECK_LAUNCH();
    return 0;
}

// Deterministic sparse apply.  Index arrays are produced by ops/embedding.py with fixed-size
// torch ops (no host sync): sorted_ids[n], perm[n], chunk_begin[n+1], seg_begin[n+1],
// seg_first_chunk[n+1], nchunks (device scalar).  partial: fp32 [n, D] workspace.
int dtf_embedding_sparse_apply(void* table, int table_dtype, float* acc, const void* grad, int grad_dtype,
                               const int64_t* sorted_ids, const int64_t* perm, const int64_t* chunk_begin,
                               const int64_t* seg_begin, const int64_t* seg_first_chunk,
                               const int64_t* nchunks_ptr, float* partial, int64_t n, int D, int bag, int64_t R,
                               int opt, float lr, float eps, float grad_scale, float wd, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (n <= 0) return 0;
    if (opt != EMB_OPT_SGD && !acc) return -30;
    const int dv = (D + 7) / 8;
    if (opt == EMB_OPT_ROWWISE_ADAGRAD && (64 % dv) != 0) return -31;
    if ((D % 8 == 0) && (((uintptr_t)table | (uintptr_t)grad | (uintptr_t)partial | (uintptr_t)acc) & 15)) return -16;
    const int grid = dtf_stream_grid(n * dv, 256);
    if (grad_dtype == DTF_BF16)
        hipLaunchKernelGGL((emb_chunk_partial_kernel<DTF_BF16>), dim3(grid), dim3(256), 0, stream, grad, perm,
                           chunk_begin, partial, n, D, bag);
    else
        hipLaunchKernelGGL((emb_chunk_partial_kernel<DTF_F32>), dim3(grid), dim3(256), 0, stream, grad, perm,
                           chunk_begin, partial, n, D, bag);
    EmbApplyArgs a{table, acc, partial, sorted_ids, seg_begin, seg_first_chunk, nchunks_ptr, n, R, D, opt,
                   lr, eps, grad_scale, wd};
    if (table_dtype == DTF_BF16)
        hipLaunchKernelGGL((emb_segment_apply_kernel<DTF_BF16>), dim3(grid), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((emb_segment_apply_kernel<DTF_F32>), dim3(grid), dim3(256), 0, stream, a);
    DTF_CHECK_LAUNCH();
    return 0;
}

int dtf_embedding_sgd_atomic(float* table, const int64_t* idx, const void* grad, int grad_dtype, int64_t n, int D,
                             int bag, int64_t R, float lr, float grad_scale, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (n <= 0) return 0;
    const int grid = dtf_stream_grid(n * D, 256);
    const float s = -lr * grad_scale;
    if (grad_dtype == DTF_BF16)
        hipLaunchKernelGGL((emb_sgd_atomic_kernel<DTF_BF16>), dim3(grid), dim3(256), 0, stream, table, idx, grad, n,
                           D, bag, R, s);
    else
        hipLaunchKernelGGL((emb_sgd_atomic_kernel<DTF_F32>), dim3(grid), dim3(256), 0, stream, table, idx, grad, n,
                           D, bag, R, s);
    DTF_CHECK_LAUNCH();
    return 0;
}

int dtf_embedding_chunk_rows() { return EMB_CHUNK; }

}  // extern "C"
