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
                                                                const int64_t* __restrict__ sorted_ids,
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
        // a chunk that IS its whole run (unique or rarely repeated ids: almost every row of an HBM-sized table) has no
        // slab: the apply kernel sums those few gradient rows itself -- no fp32 round trip through `partial`
        {
            const int64_t id = sorted_ids[j0];
            const bool starts = j0 == 0 || sorted_ids[j0 - 1] != id;
            const bool ends = j1 >= n || sorted_ids[j1] != id;
            if (starts && ends) continue;
        }
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        int64_t j = j0;
        if (vec) {
            // four gathers in flight per lane (the loop is a chain of dependent loads: perm[j], then that gradient row --
            // one at a time it ran at 0.7 TB/s); the adds stay in position order, so the sums are bit for bit the old ones
            for (; j + 4 <= j1; j += 4) {
                int64_t src[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) src[u] = DTF_IDX(perm[j + u] / bag, n / bag, "emb_chunk_partial perm");
                float x[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) load8<GDT>(grad, src[u] * D + (v << 3), x[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] += x[u][k];
                }
            }
        }
        for (; j < j1; ++j) {
            const int64_t src = DTF_IDX(perm[j] / bag, n / bag, "emb_chunk_partial perm");
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

// ---------------------------------------------------------------------------
// Segment plan of the deterministic sparse apply: from the SORTED ids, the run (= unique id) and chunk
// boundaries the two kernels above walk.  The framework-op version (a cumsum, three scatter_reduce, a
// gather and a handful of elementwise kernels over n positions) cost 1.3 ms of a 5.3 ms Wide&Deep step;
// these four small kernels do it in two block-level passes over the ids:
//   1. last run head of every 1024-position block (the run that enters the next block starts there);
//   2. heads / chunk heads per block (needs 1 for the carried-in run start);
//   3. exclusive scan of the per-block counts (one block);
//   4. emit seg_begin / seg_first_chunk at the run heads, chunk_begin at the chunk heads, the sentinels
//      (= n) everywhere else, and the chunk count.
// ---------------------------------------------------------------------------
#define PLAN_T 256
#define PLAN_E 4
#define PLAN_BLK (PLAN_T * PLAN_E)

__device__ __forceinline__ int plan_scan_max_excl(int v, int* sh) {      // exclusive max-scan over the block, identity -1
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int o = 1; o < PLAN_T; o <<= 1) {
        const int x = t >= o ? sh[t - o] : -1;
        __syncthreads();
        if (x > sh[t]) sh[t] = x;
        __syncthreads();
    }
    const int r = t > 0 ? sh[t - 1] : -1;
    __syncthreads();
    return r;
}

__device__ __forceinline__ int plan_scan_sum_excl(int v, int* sh, int* total) {
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int o = 1; o < PLAN_T; o <<= 1) {
        const int x = t >= o ? sh[t - o] : 0;
        __syncthreads();
        sh[t] += x;
        __syncthreads();
    }
    const int r = t > 0 ? sh[t - 1] : 0;
    *total = sh[PLAN_T - 1];
    __syncthreads();
    return r;
}

// heads of the thread's PLAN_E consecutive positions; returns the last head position (or -1)
__device__ __forceinline__ int plan_heads(const int64_t* __restrict__ sid, int n, int i0, bool (&head)[PLAN_E]) {
    int last = -1;
    int64_t prev = (i0 > 0 && i0 <= n) ? sid[i0 - 1] : 0;
#pragma unroll
    for (int e = 0; e < PLAN_E; ++e) {
        const int i = i0 + e;
        head[e] = false;
        if (i < n) {
            const int64_t cur = sid[i];
            head[e] = (i == 0) || (cur != prev);
            prev = cur;
            if (head[e]) last = i;
        }
    }
    return last;
}

__global__ __launch_bounds__(PLAN_T) void plan_last_head_kernel(const int64_t* __restrict__ sid, int n,
                                                                 int* __restrict__ last_head) {
    __shared__ int sh[PLAN_T];
    bool head[PLAN_E];
    const int last = plan_heads(sid, n, blockIdx.x * PLAN_BLK + threadIdx.x * PLAN_E, head);
    sh[threadIdx.x] = last;
    __syncthreads();
    for (int o = PLAN_T / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o && sh[threadIdx.x + o] > sh[threadIdx.x]) sh[threadIdx.x] = sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) last_head[blockIdx.x] = sh[0];
}

// shared part of passes 2 and 4: heads, chunk heads and their per-thread counts
__device__ __forceinline__ void plan_flags(const int64_t* __restrict__ sid, int n, const int* __restrict__ last_head,
                                           int chunk, int* sh, bool (&head)[PLAN_E], bool (&chead)[PLAN_E], int& nh,
                                           int& nc) {
    __shared__ int carry;
    if (threadIdx.x == 0) {          // start of the run that enters this block: nearest earlier block with a head
        int c = 0;
        for (int b = (int)blockIdx.x - 1; b >= 0; --b)
            if (last_head[b] >= 0) { c = last_head[b]; break; }
        carry = c;
    }
    const int i0 = blockIdx.x * PLAN_BLK + threadIdx.x * PLAN_E;
    const int last = plan_heads(sid, n, i0, head);
    int start = plan_scan_max_excl(last, sh);          // includes the barrier that publishes carry
    if (start < 0) start = carry;
    nh = 0; nc = 0;
#pragma unroll
    for (int e = 0; e < PLAN_E; ++e) {
        const int i = i0 + e;
        chead[e] = false;
        if (i < n) {
            if (head[e]) start = i;
            chead[e] = ((i - start) % chunk) == 0;
            nh += head[e] ? 1 : 0;
            nc += chead[e] ? 1 : 0;
        }
    }
}

__global__ __launch_bounds__(PLAN_T) void plan_count_kernel(const int64_t* __restrict__ sid, int n,
                                                             const int* __restrict__ last_head, int chunk,
                                                             int* __restrict__ counts /* [2][nblk + 1] */, int nblk) {
    __shared__ int sh[PLAN_T];
    bool head[PLAN_E], chead[PLAN_E];
    int nh, nc;
    plan_flags(sid, n, last_head, chunk, sh, head, chead, nh, nc);
    int th, tc;
    plan_scan_sum_excl(nh, sh, &th);
    plan_scan_sum_excl(nc, sh, &tc);
    if (threadIdx.x == 0) { counts[blockIdx.x] = th; counts[nblk + 1 + blockIdx.x] = tc; }
}

// in-place exclusive scan of the two count rows (nblk entries each, total appended at [nblk]); one block
__global__ __launch_bounds__(PLAN_T) void plan_scan_blocks_kernel(int* __restrict__ counts, int nblk) {
    __shared__ int sh[PLAN_T];
    const int per = (nblk + PLAN_T - 1) / PLAN_T;
    for (int row = 0; row < 2; ++row) {
        int* c = counts + row * (nblk + 1);
        const int b0 = threadIdx.x * per;
        int s = 0;
        for (int k = 0; k < per; ++k)
            if (b0 + k < nblk) s += c[b0 + k];
        int total;
        int off = plan_scan_sum_excl(s, sh, &total);
        for (int k = 0; k < per; ++k)
            if (b0 + k < nblk) { const int v = c[b0 + k]; c[b0 + k] = off; off += v; }
        if (threadIdx.x == 0) c[nblk] = total;
        __syncthreads();
    }
}

__global__ __launch_bounds__(PLAN_T) void plan_emit_kernel(const int64_t* __restrict__ sid, int n,
                                                            const int* __restrict__ last_head, int chunk,
                                                            const int* __restrict__ counts, int nblk,
                                                            int64_t* __restrict__ chunk_begin, int64_t* __restrict__ seg_begin,
                                                            int64_t* __restrict__ seg_first_chunk, int64_t* __restrict__ nchunks) {
    __shared__ int sh[PLAN_T];
    bool head[PLAN_E], chead[PLAN_E];
    int nh, nc;
    plan_flags(sid, n, last_head, chunk, sh, head, chead, nh, nc);
    int th, tc;
    int seg = counts[blockIdx.x] + plan_scan_sum_excl(nh, sh, &th);             // id of the thread's first head
    int chk = counts[nblk + 1 + blockIdx.x] + plan_scan_sum_excl(nc, sh, &tc);
    const int nseg_total = counts[nblk], nchunk_total = counts[2 * nblk + 1];
    const int i0 = blockIdx.x * PLAN_BLK + threadIdx.x * PLAN_E;
#pragma unroll
    for (int e = 0; e < PLAN_E; ++e) {
        const int i = i0 + e;
        if (i >= n) break;
        if (chead[e]) {
            chunk_begin[chk] = i;
            if (head[e]) { seg_begin[seg] = i; seg_first_chunk[seg] = chk; ++seg; }
            ++chk;
        }
        // unused tail of the three arrays (n + 1 entries each) = n
        if (i >= nseg_total) { seg_begin[i] = n; seg_first_chunk[i] = n; }
        if (i >= nchunk_total) chunk_begin[i] = n;
        if (i == n - 1) { seg_begin[n] = n; seg_first_chunk[n] = n; chunk_begin[n] = n; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) nchunks[0] = nchunk_total;
}

struct EmbApplyArgs {
    void* table; float* acc;                       // acc: [R, D] (Adagrad) or [R] (row-wise) or null
    const float* partial;
    const void* grad; const int64_t* perm; const int64_t* chunk_begin; int bag;
    const int64_t* sorted_ids; const int64_t* seg_begin; const int64_t* seg_first_chunk;
    const int64_t* nchunks_ptr;                    // device scalar: number of chunks
    int64_t n, R; int D, opt;
    float lr, eps, grad_scale, wd;
};

// (a.2) one lane group per unique id: g = ordered sum of its chunk slabs, then the optimizer.
template <int TDT, int GDT>
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
        const int64_t row = a.sorted_ids[DTF_IDX(j0, a.n, "emb_segment_apply seg_begin")];
        if (row < 0 || row >= a.R) continue;        // padding id: no update
        const int64_t c0 = DTF_IDX(a.seg_first_chunk[s], a.n + 1, "emb_segment_apply first chunk");
        int64_t c1 = a.seg_first_chunk[s + 1];
        if (c1 > nchunks) c1 = nchunks;
        // the row's weights (and Adagrad accumulator) are requested FIRST: their addresses need only the id, so the
        // random HBM row fetches overlap with the perm -> gradient-row chain below instead of following it
        float w[8], h[8];
        const int64_t off = row * D + (v << 3);
        if (vec) load8<TDT>(a.table, off, w);
        else {
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = ((v << 3) + k < D) ? load1<TDT>(a.table, off + k) : 0.f;
        }
        if (a.opt == EMB_OPT_ADAGRAD) {
            if (vec) load8<DTF_F32>(a.acc, off, h);
            else {
#pragma unroll
                for (int k = 0; k < 8; ++k) h[k] = ((v << 3) + k < D) ? a.acc[off + k] : 0.f;
            }
        }
        float g[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = 0.f;
        if (c1 - c0 == 1) {
            // single-chunk run: the gradient rows straight from `grad`, in sorted order (bitwise the slab's sum)
            const int64_t j1 = a.chunk_begin[c0 + 1];
            for (int64_t j = j0; j < j1; ++j) {
                const int64_t src = DTF_IDX(a.perm[j] / a.bag, a.n / a.bag, "emb_segment_apply perm");
                if (vec) {
                    float x[8];
                    load8<GDT>(a.grad, src * D + (v << 3), x);
#pragma unroll
                    for (int k = 0; k < 8; ++k) g[k] += x[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if ((v << 3) + k < D) g[k] += load1<GDT>(a.grad, src * D + (v << 3) + k);
                }
            }
        } else
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
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = fmaf(g[k], a.grad_scale, a.wd * w[k]);
        if (a.opt == EMB_OPT_SGD) {
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = fmaf(-a.lr, g[k], w[k]);
        } else if (a.opt == EMB_OPT_ADAGRAD) {
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

// DTF_EMB_GRID_CAP: blocks of the gather / sparse-apply kernels.  Default: no cap, one loop body per lane -- at HBM
// scale (180 GB of tables, profiles/r2_straggler_kernels.txt) the gather went 4.09 -> 4.54 TB/s and sort + segment
// apply 0.998 -> 0.958 ms against the 2048-block grid-strided shape.
static inline int emb_grid(int64_t items) {
    static int64_t cap = 0;
    if (cap == 0) {
        const char* e = getenv("DTF_EMB_GRID_CAP");
        cap = e ? atoll(e) : (1ll << 30);
        if (cap < 1) cap = 1ll << 30;
    }
    int64_t b = (items + 255) / 256;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

extern "C" {

int dtf_embedding_gather(const void* table, int table_dtype, const int64_t* idx, void* out, int out_dtype,
                         int64_t nbags, int bag, int D, int64_t R, float scale, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (nbags <= 0) return 0;
    const bool vec = (D % 8 == 0) && ((((uintptr_t)table) | ((uintptr_t)out)) & 15) == 0;
    const int64_t items = vec ? nbags * (D / 8) : nbags * D;
    const int grid = emb_grid(items);
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
    const int grid = emb_grid(n * dv);
    if (grad_dtype == DTF_BF16)
        hipLaunchKernelGGL((emb_chunk_partial_kernel<DTF_BF16>), dim3(grid), dim3(256), 0, stream, grad, perm,
                           chunk_begin, sorted_ids, partial, n, D, bag);
    else
        hipLaunchKernelGGL((emb_chunk_partial_kernel<DTF_F32>), dim3(grid), dim3(256), 0, stream, grad, perm,
                           chunk_begin, sorted_ids, partial, n, D, bag);
    EmbApplyArgs a{table, acc, partial, grad, perm, chunk_begin, bag, sorted_ids, seg_begin, seg_first_chunk,
                   nchunks_ptr, n, R, D, opt, lr, eps, grad_scale, wd};
#define APPLY(T, G) hipLaunchKernelGGL((emb_segment_apply_kernel<T, G>), dim3(grid), dim3(256), 0, stream, a)
    if (table_dtype == DTF_BF16) { if (grad_dtype == DTF_BF16) APPLY(DTF_BF16, DTF_BF16); else APPLY(DTF_BF16, DTF_F32); }
    else { if (grad_dtype == DTF_BF16) APPLY(DTF_F32, DTF_BF16); else APPLY(DTF_F32, DTF_F32); }
#undef APPLY
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

// ints of scratch for dtf_embedding_segment_plan
int64_t dtf_embedding_plan_scratch_ints(int64_t n) {
    const int64_t nblk = (n + PLAN_BLK - 1) / PLAN_BLK;
    return nblk + 2 * (nblk + 1);
}

// sorted_ids[n] -> chunk_begin[n+1], seg_begin[n+1], seg_first_chunk[n+1], nchunks[1] (see the kernels above)
int dtf_embedding_segment_plan(const int64_t* sorted_ids, int64_t n, int chunk, int* scratch, int64_t* chunk_begin,
                               int64_t* seg_begin, int64_t* seg_first_chunk, int64_t* nchunks, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (n <= 0 || n >= (1ll << 31) - PLAN_BLK || chunk < 1) return -32;
    const int nblk = (int)((n + PLAN_BLK - 1) / PLAN_BLK);
    int* last_head = scratch;
    int* counts = scratch + nblk;
    hipLaunchKernelGGL(plan_last_head_kernel, dim3(nblk), dim3(PLAN_T), 0, stream, sorted_ids, (int)n, last_head);
    hipLaunchKernelGGL(plan_count_kernel, dim3(nblk), dim3(PLAN_T), 0, stream, sorted_ids, (int)n, last_head, chunk, counts, nblk);
    hipLaunchKernelGGL(plan_scan_blocks_kernel, dim3(1), dim3(PLAN_T), 0, stream, counts, nblk);
    hipLaunchKernelGGL(plan_emit_kernel, dim3(nblk), dim3(PLAN_T), 0, stream, sorted_ids, (int)n, last_head, chunk, counts, nblk,
                       chunk_begin, seg_begin, seg_first_chunk, nchunks);
    DTF_CHECK_LAUNCH();
    return 0;
}

int dtf_embedding_chunk_rows() { return EMB_CHUNK; }

}  // extern "C"
