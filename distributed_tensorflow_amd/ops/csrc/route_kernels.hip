// Embedding id routing for row-sharded tables (gfx950): replaces the
// sort / bincount / .tolist() / index_put chain around the three all-to-alls
// of parallel/embedding.py with a STABLE W-way bucket-by-owner split that
// never leaves the device (SURVEY 2.4 R3, 5.8 "no host sync in the step").
//
//   owner(g)  = g % W           (TF `mod` partition)   local = g / W
//             = g / rows_per    (TF `div` partition)   local = g - owner * rows_per
//
// Every (source rank, owner) pair exchanges a FIXED number of slots `cap`
// (capacity factor x N / W, like an MoE dispatch), so the all-to-all split
// sizes are host constants.  Request i goes to slot  owner * cap + rank_i,
// where rank_i = number of earlier requests with the same owner: positions
// are a pure function of the input (no atomics), so the order in which the
// owner sums duplicate-row gradients -- and with it the table -- is bitwise
// reproducible.  Requests beyond `cap` are dropped and counted in
// `overflow` (polled by the host every few hundred steps; it then raises or
// grows the capacity).
//
//   route_count    per 1024-id block: histogram over owners
//   route_scan     one wave per owner: exclusive scan of the block histograms
//   route_scatter  recompute the stable ranks, write send_ids / slot / inv
#include "dtf_common.h"

#define ROUTE_BLOCK 256
#define ROUTE_ITERS 4
#define ROUTE_TILE (ROUTE_BLOCK * ROUTE_ITERS)
#define ROUTE_MAX_W 64

struct RouteArgs {
    const int64_t* ids;     // [n] global rows (negative = padding)
    int64_t n;
    int W;                  // owners
    int div;                // 0: mod partition, 1: div partition
    int64_t rows_per;       // div: rows per part
    int64_t total_rows;
    int64_t cap;            // slots per owner
    int32_t* hist;          // [nblk, W] counts, then (after scan) exclusive bases
    int64_t* totals;        // [W] requests per owner (before capping)
    int64_t* send_ids;      // [W, cap] local row at the owner, -1 = empty slot
    int64_t* slot;          // [n] slot of request i, -1 = dropped / padding
    int64_t* inv;           // [W, cap] request index held by a slot, -1 = empty
    int64_t* overflow;      // [1] += number of dropped requests
};

__device__ __forceinline__ int route_owner(const RouteArgs& a, int64_t g, int64_t& local) {
    if (g < 0 || g >= a.total_rows) { local = -1; return -1; }
    if (a.div) {
        int o = (int)DTF_IDX(g / a.rows_per, a.W, "route div owner");
        local = g - (int64_t)o * a.rows_per;
        return o;
    }
    local = g / a.W;
    return (int)(g % a.W);
}

__global__ __launch_bounds__(ROUTE_BLOCK) void route_count_kernel(RouteArgs a) {
    __shared__ int cnt[ROUTE_MAX_W];
    for (int j = threadIdx.x; j < a.W; j += ROUTE_BLOCK) cnt[j] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * ROUTE_TILE;
#pragma unroll
    for (int e = 0; e < ROUTE_ITERS; ++e) {
        const int64_t i = base + e * ROUTE_BLOCK + threadIdx.x;
        int64_t local;
        const int o = i < a.n ? route_owner(a, a.ids[i], local) : -1;
        for (int j = 0; j < a.W; ++j) {
            const unsigned long long m = __ballot(o == j);
            if ((threadIdx.x & 63) == 0 && m) atomicAdd(&cnt[j], __popcll(m));   // LDS atomic, integer: order-free
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < a.W; j += ROUTE_BLOCK) a.hist[(int64_t)blockIdx.x * a.W + j] = cnt[j];
}

// grid = W blocks of one wave: exclusive scan over the blocks of column `owner`
__global__ __launch_bounds__(64) void route_scan_kernel(RouteArgs a, int nblk) {
    const int j = blockIdx.x;
    const int lane = threadIdx.x;
    int64_t running = 0;
    for (int b0 = 0; b0 < nblk; b0 += 64) {
        const int b = b0 + lane;
        int v = b < nblk ? a.hist[(int64_t)b * a.W + j] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (b < nblk) {
            // bases beyond INT32 cannot occur: n < 2^31 is checked on the host
            a.hist[(int64_t)b * a.W + j] = (int)(running + incl - v);
        }
        running += __shfl(incl, 63, 64);
    }
    if (lane == 0) {
        a.totals[j] = running;
        if (running > a.cap) atomicAdd((unsigned long long*)a.overflow, (unsigned long long)(running - a.cap));
    }
}

__global__ __launch_bounds__(ROUTE_BLOCK) void route_scatter_kernel(RouteArgs a, int nblk) {
    __shared__ int wcnt[ROUTE_BLOCK / 64][ROUTE_MAX_W];
    __shared__ int running[ROUTE_MAX_W];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int)blockIdx.x < nblk) {
        for (int j = threadIdx.x; j < a.W; j += ROUTE_BLOCK) running[j] = a.hist[(int64_t)blockIdx.x * a.W + j];
        const int64_t base = (int64_t)blockIdx.x * ROUTE_TILE;
        const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
        for (int e = 0; e < ROUTE_ITERS; ++e) {
            const int64_t i = base + e * ROUTE_BLOCK + threadIdx.x;
            int64_t local = -1;
            const int o = i < a.n ? route_owner(a, a.ids[i], local) : -1;
            int rank = 0;
            for (int j = 0; j < a.W; ++j) {
                const unsigned long long m = __ballot(o == j);
                if (o == j) rank = __popcll(m & lt);
                if (lane == 0) wcnt[wave][j] = __popcll(m);
            }
            __syncthreads();                       // wcnt of this iteration + running visible
            if (o >= 0) {
                int64_t p = running[o] + rank;
                for (int w = 0; w < wave; ++w) p += wcnt[w][o];
                if (p < a.cap) {
                    const int64_t s = DTF_IDX((int64_t)o * a.cap + p, (int64_t)a.W * a.cap, "route slot");
                    a.send_ids[s] = local;
                    a.inv[s] = i;
                    a.slot[i] = s;
                } else {
                    a.slot[i] = -1;
                }
            } else if (i < a.n) {
                a.slot[i] = -1;
            }
            __syncthreads();                       // everyone has read running / wcnt
            for (int j = threadIdx.x; j < a.W; j += ROUTE_BLOCK) {
                int s = 0;
#pragma unroll
                for (int w = 0; w < ROUTE_BLOCK / 64; ++w) s += wcnt[w][j];
                running[j] += s;
            }
            __syncthreads();                       // wcnt is rewritten by the next iteration
        }
    }
    // empty slots of every owner's list: grid-stride over [W, cap]
    const int64_t stride = (int64_t)gridDim.x * ROUTE_BLOCK;
    const int64_t all = (int64_t)a.W * a.cap;
    for (int64_t s = (int64_t)blockIdx.x * ROUTE_BLOCK + threadIdx.x; s < all; s += stride) {
        const int j = (int)(s / a.cap);
        if (s - (int64_t)j * a.cap >= a.totals[j]) { a.send_ids[s] = -1; a.inv[s] = -1; }
    }
}

extern "C" {

int dtf_route_tile() { return ROUTE_TILE; }

// hist: int32 [ceil(n / ROUTE_TILE), W];  totals: int64 [W];  overflow: int64 [1] (accumulates)
int dtf_embedding_route(const int64_t* ids, int64_t n, int W, int div, int64_t rows_per, int64_t total_rows,
                        int64_t cap, int32_t* hist, int64_t* totals, int64_t* send_ids, int64_t* slot,
                        int64_t* inv, int64_t* overflow, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (W < 1 || W > ROUTE_MAX_W) return -41;
    if (n < 0 || n >= (1LL << 31) - ROUTE_TILE || cap < 1) return -42;
    if (div && rows_per < 1) return -43;
    const int nblk = (int)((n + ROUTE_TILE - 1) / ROUTE_TILE);
    RouteArgs a{ids, n, W, div, rows_per, total_rows, cap, hist, totals, send_ids, slot, inv, overflow};
    if (nblk > 0) hipLaunchKernelGGL(route_count_kernel, dim3(nblk), dim3(ROUTE_BLOCK), 0, stream, a);
    hipLaunchKernelGGL(route_scan_kernel, dim3(W), dim3(64), 0, stream, a, nblk);
    int64_t fill_blocks = ((int64_t)W * cap + ROUTE_BLOCK - 1) / ROUTE_BLOCK;
    if (fill_blocks > 2048) fill_blocks = 2048;
    const int grid = (int)(nblk > fill_blocks ? nblk : fill_blocks);
    hipLaunchKernelGGL(route_scatter_kernel, dim3(grid > 0 ? grid : 1), dim3(ROUTE_BLOCK), 0, stream, a, nblk);
    DTF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
