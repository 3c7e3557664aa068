// K18: 1x1 convolution ONTO 64 channels as a register-streaming kernel:  y[M, 64] = x[M, K] . w[64, K]^T,  K = 64 or 256.
// ResNet's stage-1 squeeze layers and the data gradients of its expand layers (256 -> 64 channels at 56 x 56, batch 256:
// 411 MB in, 103 MB out) are one read stream with a 32 KB filter; the library's best solutions take 125 us (K = 256) and
// 64 / 76 us (K = 64) where the stream alone is ~86 / ~40 us.  No LDS at all here: a wave keeps the WHOLE filter in registers
// (K = 256: 32 fragments = 128 registers) and owns 16-pixel groups; a lane's MFMA operand of a group is 16 contiguous bytes
// of its pixel's row, loaded straight from HBM into the operand registers one group (K = 64: four groups) ahead; operands
// swapped with permuted filter rows (as K16 / K17) so a lane stores 8 consecutive channels of its pixel.
#include "dtf_common.h"

typedef __attribute__((ext_vector_type(8))) short n64_s16x8;

struct N64Args {
    const unsigned short* x; const unsigned short* w; unsigned short* y;
    int64_t M, ngroups;
};

// KS: K steps of 32 input channels; U: 16-pixel groups per iteration (loads of 2 U groups in flight per wave)
template <int KS, int U, bool NT>
__global__ __launch_bounds__(256) void conv1x1_n64_kernel(N64Args a) {
    constexpr int K = 32 * KS;
    const int lane = threadIdx.x & 63, i16 = lane & 15, qd = lane >> 4;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;

    // filter fragment (s, f) row r = output channel 32 (f / 2) + 8 (r / 4) + 4 (f % 2) + (r % 4): the accumulator rows
    // 4 qd .. 4 qd + 3 of fragments 2 h, 2 h + 1 are then the 8 consecutive channels 32 h + 8 qd ..
    n64_s16x8 wf[KS][4];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int och = 32 * (f >> 1) + 8 * (i16 >> 2) + 4 * (f & 1) + (i16 & 3);
            wf[s][f] = *reinterpret_cast<const n64_s16x8*>(a.w + och * K + s * 32 + qd * 8);
        }

    const int64_t niter = (a.ngroups + U - 1) / U;
    auto load = [&](int64_t t, n64_s16x8 (&xb)[U][KS]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t g = t * U + u;
            if (g >= a.ngroups) g = a.ngroups - 1;            // surplus groups repeat the last one; their stores are skipped
            const unsigned short* p = a.x + DTF_IDX((g * 16 + i16) * K + qd * 8, a.M * K - 7 - 32 * (KS - 1), "K18 input group");
#pragma unroll
            for (int s = 0; s < KS; ++s)
                xb[u][s] = NT ? __builtin_nontemporal_load(reinterpret_cast<const n64_s16x8*>(p + s * 32))
                              : *reinterpret_cast<const n64_s16x8*>(p + s * 32);
        }
    };
    auto compute = [&](int64_t t, n64_s16x8 (&xb)[U][KS]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4 acc[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][f], xb[u][s], acc[f], 0, 0, 0);
            const int64_t g = t * U + u;
            if (g < a.ngroups) {
                unsigned short* yp = a.y + DTF_IDX((g * 16 + i16) * 64 + 8 * qd, a.M * 64 - 39, "K18 output group");
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u32x4 o;
                    o[0] = pack_bf16x2(acc[2 * h][0], acc[2 * h][1]); o[1] = pack_bf16x2(acc[2 * h][2], acc[2 * h][3]);
                    o[2] = pack_bf16x2(acc[2 * h + 1][0], acc[2 * h + 1][1]); o[3] = pack_bf16x2(acc[2 * h + 1][2], acc[2 * h + 1][3]);
                    *reinterpret_cast<u32x4*>(yp + 32 * h) = o;
                }
            }
        }
    };

    n64_s16x8 xb0[U][KS], xb1[U][KS];
    int64_t t = wave;
    if (t < niter) load(t, xb0);
    while (t < niter) {
        const int64_t t1 = t + nwaves;
        if (t1 < niter) load(t1, xb1);
        compute(t, xb0);
        const int64_t t2 = t1 + nwaves;
        if (t2 < niter) load(t2, xb0);
        if (t1 < niter) compute(t1, xb1);
        t = t2;
    }
}

static int n64_blocks_per_cu(int K) {
    static int env = -1;
    if (env < 0) { const char* e = getenv("DTF_N64_BLOCKS_PER_CU"); env = e ? atoi(e) : 0; }
    if (env > 0 && env <= 8) return env;
    return K == 256 ? 1 : 2;
}

extern "C" {

int dtf_conv1x1_n64_ok(int64_t M, int K, int N) {
    return N == 64 && (K == 64 || K == 256) && M >= 16 && M % 16 == 0 && M < ((int64_t)1 << 31);
}

// x: bf16 [M, K] row-major; w: bf16 [64, K] row-major; y: bf16 [M, 64]
int dtf_conv1x1_n64(const void* x, const void* w, void* y, int64_t M, int K, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (!dtf_conv1x1_n64_ok(M, K, 64)) return -81;
    if ((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)y)) & 15) return -16;
    N64Args a;
    a.x = (const unsigned short*)x; a.w = (const unsigned short*)w; a.y = (unsigned short*)y;
    a.M = M; a.ngroups = M / 16;
    static int u256 = 0;
    if (u256 == 0) { const char* e = getenv("DTF_N64_U"); u256 = e && atoi(e) == 2 ? 2 : 1; }
    const int U = K == 256 ? u256 : 4;
    const int64_t niter = (a.ngroups + U - 1) / U;
    int64_t grid = (niter + 3) / 4;
    int cu = 0, dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
    const int64_t cap = (int64_t)cu * n64_blocks_per_cu(K);
    if (grid > cap) grid = cap;
    static int nt = -1;
    if (nt < 0) { const char* e = getenv("DTF_N64_NT"); nt = e ? atoi(e) != 0 : 0; }
    if (K == 256) {
        if (U == 2) hipLaunchKernelGGL((conv1x1_n64_kernel<8, 2, false>), dim3((unsigned)grid), dim3(256), 0, stream, a);
        else if (nt) hipLaunchKernelGGL((conv1x1_n64_kernel<8, 1, true>), dim3((unsigned)grid), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((conv1x1_n64_kernel<8, 1, false>), dim3((unsigned)grid), dim3(256), 0, stream, a);
    } else {
        if (nt) hipLaunchKernelGGL((conv1x1_n64_kernel<2, 4, true>), dim3((unsigned)grid), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((conv1x1_n64_kernel<2, 4, false>), dim3((unsigned)grid), dim3(256), 0, stream, a);
    }
    DTF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
