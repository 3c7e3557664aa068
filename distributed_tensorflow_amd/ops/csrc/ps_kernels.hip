// Parameter-server data-plane kernels for gfx950 (MI355X):
//   K1  fused SGD-momentum shard apply      (replaces TF ApplyMomentum on the PS)
//   K2  fused Adam / AdamW shard apply      (replaces TF ApplyAdam)
//   K5  multi-tensor bucket pack / unpack   (replaces per-variable RecvTensor/send)
//   K8  sum-of-squares, cast/scale helpers  (clip_by_global_norm, casts)
// SURVEY.md 2.4.  All of them are pure HBM streaming kernels: 16 B per lane per
// access, 8 elements per thread per iteration, one block per 2048 elements (dtf_flat_grid),
// no LDS, so the occupancy is register-limited only (target: >= 80 % of the
// ~6.3 TB/s achievable HBM rate on multi-MB shards).
//
// Optional W-way source sum: the gradient operand may be `nsrc` slabs of the
// same shard `src_stride` elements apart (the receive buffer of an all-to-all
// push, one slab per peer over its own xGMI link); the kernel adds them in
// fixed slab order while it streams, so the reduce costs no extra pass and the
// result is bitwise reproducible.
#include "dtf_common.h"

// hyper[] layout (device memory, optional; lets a captured hipGraph replay with
// new values): [0] learning rate, [1] extra gradient scale, [2] step count
#define HYPER_LR 0
#define HYPER_GSCALE 1
#define HYPER_STEP 2

struct SgdParams {
    const void* g; int64_t src_stride; int nsrc;
    float* p32; float* m; void* p_out; int64_t n;
    const float* hyper; const float* gnorm_sq;
    float lr, momentum, wd, grad_scale, max_norm; int nesterov;
};

__device__ __forceinline__ float clip_coef(const float* gnorm_sq, float max_norm) {
    if (gnorm_sq == nullptr || max_norm <= 0.f) return 1.f;
    float nrm = sqrtf(*gnorm_sq);
    return nrm > max_norm ? max_norm / (nrm + 1e-6f) : 1.f;
}

template <int GDT, int ODT>
__global__ __launch_bounds__(256) void sgd_momentum_apply_kernel(SgdParams a) {
    const float lr = a.hyper ? a.hyper[HYPER_LR] : a.lr;
    const float gs = a.grad_scale * (a.hyper ? a.hyper[HYPER_GSCALE] : 1.f) * clip_coef(a.gnorm_sq, a.max_norm);
    const float mu = a.momentum, wd = a.wd;
    const int64_t nvec = a.n >> 3;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        const int64_t i = v << 3;
        float g[8], p[8], m[8];
        load8<GDT>(a.g, i, g);
        for (int s = 1; s < a.nsrc; ++s) {
            float t[8];
            load8<GDT>(a.g, i + s * a.src_stride, t);
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] += t[k];
        }
        load8<DTF_F32>(a.p32, i, p);
        load8<DTF_F32>(a.m, i, m);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float gk = fmaf(g[k], gs, wd * p[k]);
            m[k] = fmaf(mu, m[k], gk);
            float upd = a.nesterov ? fmaf(mu, m[k], gk) : m[k];
            p[k] = fmaf(-lr, upd, p[k]);
        }
        store8<DTF_F32>(a.p32, i, p);
        store8<DTF_F32>(a.m, i, m);
        if (a.p_out) store8<ODT>(a.p_out, i, p);
    }
    // scalar tail (n not a multiple of 8)
    const int64_t tail0 = nvec << 3;
    if (blockIdx.x == 0) {
        for (int64_t i = tail0 + threadIdx.x; i < a.n; i += blockDim.x) {
            float g = 0.f;
            for (int s = 0; s < a.nsrc; ++s) g += load1<GDT>(a.g, i + s * a.src_stride);
            float p = a.p32[i], m = a.m[i];
            float gk = fmaf(g, gs, wd * p);
            m = fmaf(mu, m, gk);
            float upd = a.nesterov ? fmaf(mu, m, gk) : m;
            p = fmaf(-lr, upd, p);
            a.p32[i] = p; a.m[i] = m;
            if (a.p_out) store1<ODT>(a.p_out, i, p);
        }
    }
}

struct AdamParams {
    const void* g; int64_t src_stride; int nsrc;
    float* p32; float* m; float* v; void* p_out; int64_t n;
    const float* hyper; const float* gnorm_sq;
    float lr, beta1, beta2, eps, wd, grad_scale, max_norm, step;
    int adamw, bias_correction;
};

template <int GDT, int ODT>
__global__ __launch_bounds__(256) void adam_apply_kernel(AdamParams a) {
    const float lr = a.hyper ? a.hyper[HYPER_LR] : a.lr;
    const float step = a.hyper ? a.hyper[HYPER_STEP] : a.step;
    const float gs = a.grad_scale * (a.hyper ? a.hyper[HYPER_GSCALE] : 1.f) * clip_coef(a.gnorm_sq, a.max_norm);
    const float b1 = a.beta1, b2 = a.beta2, eps = a.eps, wd = a.wd;
    float bc1 = 1.f, bc2s = 1.f;
    if (a.bias_correction) {
        bc1 = 1.f - powf(b1, step);
        bc2s = sqrtf(1.f - powf(b2, step));
    }
    const float step_size = lr / bc1;
    const float l2 = a.adamw ? 0.f : wd;
    const float decay = a.adamw ? lr * wd : 0.f;
    const int64_t nvec = a.n >> 3;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t vi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; vi < nvec; vi += stride) {
        const int64_t i = vi << 3;
        float g[8], p[8], m[8], v[8];
        load8<GDT>(a.g, i, g);
        for (int s = 1; s < a.nsrc; ++s) {
            float t[8];
            load8<GDT>(a.g, i + s * a.src_stride, t);
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] += t[k];
        }
        load8<DTF_F32>(a.p32, i, p);
        load8<DTF_F32>(a.m, i, m);
        load8<DTF_F32>(a.v, i, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float gk = fmaf(g[k], gs, l2 * p[k]);
            m[k] = fmaf(b1, m[k], (1.f - b1) * gk);
            v[k] = fmaf(b2, v[k], (1.f - b2) * gk * gk);
            float denom = sqrtf(v[k]) / bc2s + eps;
            p[k] = p[k] - decay * p[k] - step_size * (m[k] / denom);
        }
        store8<DTF_F32>(a.p32, i, p);
        store8<DTF_F32>(a.m, i, m);
        store8<DTF_F32>(a.v, i, v);
        if (a.p_out) store8<ODT>(a.p_out, i, p);
    }
    const int64_t tail0 = nvec << 3;
    if (blockIdx.x == 0) {
        for (int64_t i = tail0 + threadIdx.x; i < a.n; i += blockDim.x) {
            float g = 0.f;
            for (int s = 0; s < a.nsrc; ++s) g += load1<GDT>(a.g, i + s * a.src_stride);
            float p = a.p32[i], m = a.m[i], v = a.v[i];
            float gk = fmaf(g, gs, l2 * p);
            m = fmaf(b1, m, (1.f - b1) * gk);
            v = fmaf(b2, v, (1.f - b2) * gk * gk);
            float denom = sqrtf(v) / bc2s + eps;
            p = p - decay * p - step_size * (m / denom);
            a.p32[i] = p; a.m[i] = m; a.v[i] = v;
            if (a.p_out) store1<ODT>(a.p_out, i, p);
        }
    }
}

// ---------------------------------------------------------------------------
// K5: multi-tensor pack / unpack.  One launch moves up to PACK_MAX_T tensors;
// the descriptor table rides in the kernel arguments (no H2D copy, no 161 tiny
// launches at ~1.2-1.5 us of boundary each).  A block owns one PACK_CHUNK-element
// chunk of one tensor and finds it by a wave-uniform binary search over the
// chunk prefix sums.
// ---------------------------------------------------------------------------
#define PACK_MAX_T 96
#define PACK_CHUNK 8192   // elements per block: 256 threads x 8 elems x 4 iterations

struct PackTable {
    const void* ptr[PACK_MAX_T];        // per-tensor storage (source for pack, destination for unpack)
    int64_t flat_off[PACK_MAX_T];       // element offset inside the flat bucket buffer
    int32_t numel[PACK_MAX_T];
    int32_t chunk_start[PACK_MAX_T + 1];
    uint8_t dtype[PACK_MAX_T];          // dtype of the per-tensor storage
    int32_t ntensors;
};

template <int SDT, int DDT>
__device__ __forceinline__ void copy_chunk(const void* src, int64_t s0, void* dst, int64_t d0,
                                           int64_t count, float scale, bool vec_ok) {
    if (vec_ok) {
        const int64_t nvec = count >> 3;
        for (int64_t v = threadIdx.x; v < nvec; v += blockDim.x) {
            float x[8];
            load8<SDT>(src, s0 + (v << 3), x);
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] *= scale;
            store8<DDT>(dst, d0 + (v << 3), x);
        }
        for (int64_t i = (nvec << 3) + threadIdx.x; i < count; i += blockDim.x)
            store1<DDT>(dst, d0 + i, load1<SDT>(src, s0 + i) * scale);
    } else {
        for (int64_t i = threadIdx.x; i < count; i += blockDim.x)
            store1<DDT>(dst, d0 + i, load1<SDT>(src, s0 + i) * scale);
    }
}

__device__ __forceinline__ int find_tensor(const PackTable& t, int chunk) {
    int lo = 0, hi = t.ntensors;   // chunk_start[lo] <= chunk < chunk_start[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (t.chunk_start[mid] <= chunk) lo = mid; else hi = mid;
    }
    return lo;
}

// UNPACK=false: tensors -> flat (pack, with scale);  UNPACK=true: flat -> tensors
template <int FDT, bool UNPACK>
__global__ __launch_bounds__(256) void bucket_pack_kernel(PackTable t, void* flat, float scale) {
    const int chunk = blockIdx.x;
    const int ti = (int)DTF_IDX(find_tensor(t, chunk), t.ntensors, "bucket_pack tensor of chunk");
    const int64_t e0 = (int64_t)(chunk - t.chunk_start[ti]) * PACK_CHUNK;
    int64_t count = (int64_t)t.numel[ti] - e0;
    if (count > PACK_CHUNK) count = PACK_CHUNK;
    if (count <= 0) return;
    void* tp = const_cast<void*>(t.ptr[ti]);
    const int tdt = t.dtype[ti];
    const int64_t foff = t.flat_off[ti] + e0;
    const int tbytes = tdt == DTF_BF16 ? 2 : 4;
    const int fbytes = FDT == DTF_BF16 ? 2 : 4;
    const bool vec_ok = ((((uintptr_t)tp + (uintptr_t)e0 * tbytes) & 15) == 0) &&
                        ((((uintptr_t)flat + (uintptr_t)foff * fbytes) & 15) == 0);
    if constexpr (!UNPACK) {
        if (tdt == DTF_BF16) copy_chunk<DTF_BF16, FDT>(tp, e0, flat, foff, count, scale, vec_ok);
        else copy_chunk<DTF_F32, FDT>(tp, e0, flat, foff, count, scale, vec_ok);
    } else {
        if (tdt == DTF_BF16) copy_chunk<FDT, DTF_BF16>(flat, foff, tp, e0, count, scale, vec_ok);
        else copy_chunk<FDT, DTF_F32>(flat, foff, tp, e0, count, scale, vec_ok);
    }
}

// ---------------------------------------------------------------------------
// K8 helpers
// ---------------------------------------------------------------------------
// nsrc > 1: x is `nsrc` slabs `src_stride` elements apart (the receive buffer of an all-to-all push);
// the slabs are summed in fixed order first, so the norm is the norm of the REDUCED gradient and the
// clipped path needs no separate slab-sum pass.
template <int DT>
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const void* x, int64_t n, int nsrc, int64_t src_stride,
                                                            float* partials) {
    float acc = 0.f;
    const int64_t nvec = n >> 3;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        float t[8];
        load8<DT>(x, v << 3, t);
        for (int s = 1; s < nsrc; ++s) {
            float u[8];
            load8<DT>(x, (v << 3) + s * src_stride, u);
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] += u[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = fmaf(t[k], t[k], acc);
    }
    if (blockIdx.x == 0)
        for (int64_t i = (nvec << 3) + threadIdx.x; i < n; i += blockDim.x) {
            float t = 0.f;
            for (int s = 0; s < nsrc; ++s) t += load1<DT>(x, i + s * src_stride);
            acc = fmaf(t, t, acc);
        }
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// fixed-order final sum -> bitwise reproducible norm; accumulate != 0 adds to *out
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* partials, int n, float* out,
                                                           int accumulate) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += partials[i];
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = red[0] + red[1] + red[2] + red[3];
        out[0] = accumulate ? out[0] + s : s;
    }
}

template <int SDT, int DDT>
__global__ __launch_bounds__(256) void cast_scale_kernel(const void* src, void* dst, int64_t n, float scale) {
    const int64_t nvec = n >> 3;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        float t[8];
        load8<SDT>(src, v << 3, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] *= scale;
        store8<DDT>(dst, v << 3, t);
    }
    if (blockIdx.x == 0)
        for (int64_t i = (nvec << 3) + threadIdx.x; i < n; i += blockDim.x)
            store1<DDT>(dst, i, load1<SDT>(src, i) * scale);
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

#define DISPATCH2(A, B, CALL)                                              \
    do {                                                                   \
        if ((A) == DTF_BF16 && (B) == DTF_BF16) { CALL(DTF_BF16, DTF_BF16); } \
        else if ((A) == DTF_BF16 && (B) == DTF_F32) { CALL(DTF_BF16, DTF_F32); } \
        else if ((A) == DTF_F32 && (B) == DTF_BF16) { CALL(DTF_F32, DTF_BF16); } \
        else { CALL(DTF_F32, DTF_F32); }                                    \
    } while (0)

extern "C" {

int dtf_sgd_momentum_apply(const void* g, int g_dtype, int nsrc, int64_t src_stride, float* p32, float* m,
                           void* p_out, int out_dtype, int64_t n, const float* hyper, const float* gnorm_sq,
                           float lr, float momentum, float wd, float grad_scale, float max_norm, int nesterov,
                           hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (n <= 0) return 0;
    if (!aligned16(g) || !aligned16(p32) || !aligned16(m) || !aligned16(p_out)) return -16;
    if (nsrc > 1 && (src_stride & 7)) return -17;
    SgdParams a{g, src_stride, nsrc < 1 ? 1 : nsrc, p32, m, p_out, n, hyper, gnorm_sq,
                lr, momentum, wd, grad_scale, max_norm, nesterov};
    const int grid = dtf_flat_grid(n >> 3, 256);
#define CALL(G, O) hipLaunchKernelGGL((sgd_momentum_apply_kernel<G, O>), dim3(grid), dim3(256), 0, stream, a)
    DISPATCH2(g_dtype, out_dtype, CALL);
#undef CALL
    DTF_CHECK_LAUNCH();
    return 0;
}

int dtf_adam_apply(const void* g, int g_dtype, int nsrc, int64_t src_stride, float* p32, float* m, float* v,
                   void* p_out, int out_dtype, int64_t n, const float* hyper, const float* gnorm_sq, float lr,
                   float beta1, float beta2, float eps, float wd, float grad_scale, float max_norm, float step,
                   int adamw, int bias_correction, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (n <= 0) return 0;
    if (!aligned16(g) || !aligned16(p32) || !aligned16(m) || !aligned16(v) || !aligned16(p_out)) return -16;
    if (nsrc > 1 && (src_stride & 7)) return -17;
    AdamParams a{g, src_stride, nsrc < 1 ? 1 : nsrc, p32, m, v, p_out, n, hyper, gnorm_sq,
                 lr, beta1, beta2, eps, wd, grad_scale, max_norm, step, adamw, bias_correction};
    const int grid = dtf_flat_grid(n >> 3, 256);
#define CALL(G, O) hipLaunchKernelGGL((adam_apply_kernel<G, O>), dim3(grid), dim3(256), 0, stream, a)
    DISPATCH2(g_dtype, out_dtype, CALL);
#undef CALL
    DTF_CHECK_LAUNCH();
    return 0;
}

// ptrs[i]: storage of tensor i; numel[i]; flat_off[i] element offset in `flat`;
// dtypes[i] of the per-tensor storage.  unpack=0: tensors -> flat (x scale).
int dtf_bucket_pack(int ntensors, const void* const* ptrs, const int64_t* numel, const int64_t* flat_off,
                    const int32_t* dtypes, void* flat, int flat_dtype, float scale, int unpack,
                    hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    int done = 0;
    while (done < ntensors) {
        PackTable t;
        int nt = 0, chunks = 0;
        while (done + nt < ntensors && nt < PACK_MAX_T) {
            const int64_t ne = numel[done + nt];
            if (ne < 0 || ne > 0x7fffffffLL) return -18;
            t.ptr[nt] = ptrs[done + nt];
            t.flat_off[nt] = flat_off[done + nt];
            t.numel[nt] = (int32_t)ne;
            t.dtype[nt] = (uint8_t)dtypes[done + nt];
            t.chunk_start[nt] = chunks;
            chunks += (int)((ne + PACK_CHUNK - 1) / PACK_CHUNK);
            ++nt;
        }
        t.chunk_start[nt] = chunks;
        t.ntensors = nt;
        if (chunks > 0) {
            if (flat_dtype == DTF_BF16) {
                if (unpack) hipLaunchKernelGGL((bucket_pack_kernel<DTF_BF16, true>), dim3(chunks), dim3(256), 0, stream, t, flat, scale);
                else hipLaunchKernelGGL((bucket_pack_kernel<DTF_BF16, false>), dim3(chunks), dim3(256), 0, stream, t, flat, scale);
            } else {
                if (unpack) hipLaunchKernelGGL((bucket_pack_kernel<DTF_F32, true>), dim3(chunks), dim3(256), 0, stream, t, flat, scale);
                else hipLaunchKernelGGL((bucket_pack_kernel<DTF_F32, false>), dim3(chunks), dim3(256), 0, stream, t, flat, scale);
            }
            DTF_CHECK_LAUNCH();
        }
        done += nt;
    }
    return 0;
}

// partials: workspace of >= 2048 floats.  out[0] (+)= sum((sum_s x_s)^2)
int dtf_sumsq(const void* x, int dtype, int64_t n, int nsrc, int64_t src_stride, float* partials, float* out,
              int accumulate, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (!aligned16(x)) return -16;
    if (nsrc < 1) nsrc = 1;
    if (nsrc > 1 && (src_stride & 7)) return -17;
    const int grid = dtf_stream_grid(n >> 3, 256);
    if (dtype == DTF_BF16) hipLaunchKernelGGL((sumsq_partial_kernel<DTF_BF16>), dim3(grid), dim3(256), 0, stream, x, n, nsrc, src_stride, partials);
    else hipLaunchKernelGGL((sumsq_partial_kernel<DTF_F32>), dim3(grid), dim3(256), 0, stream, x, n, nsrc, src_stride, partials);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, stream, partials, grid, out, accumulate);
    DTF_CHECK_LAUNCH();
    return 0;
}

int dtf_cast_scale(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, float scale,
                   hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (n <= 0) return 0;
    if (!aligned16(src) || !aligned16(dst)) return -16;
    const int grid = dtf_flat_grid(n >> 3, 256);
#define CALL(S, D) hipLaunchKernelGGL((cast_scale_kernel<S, D>), dim3(grid), dim3(256), 0, stream, src, dst, n, scale)
    DISPATCH2(src_dtype, dst_dtype, CALL);
#undef CALL
    DTF_CHECK_LAUNCH();
    return 0;
}

int dtf_abi_version() {
    DTF_CLEAR_ERRORS(); return 3; }

}  // extern "C"
