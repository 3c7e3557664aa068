// K13: convolution forward (and, with the flipped / transposed filter, the stride-1 data gradient) as an implicit
// GEMM on the K6 persistent LDS-DMA pipeline (gemm256p.h), optionally with the BatchNorm statistics of the output
// reduced in the epilogue (SURVEY 2.4 K7 "conv epilogue"; replaces TF's Conv2D + FusedBatchNorm statistics pass):
//
//     Y[n, oh, ow, co] = sum_{r, s, c} X[n, oh*S - ph + r, ow*S - pw + s, c] * W[co, r, s, c]        NHWC bf16, fp32 acc
//
// GEMM view: M = NB*OH*OW output pixels, N = Cout, K = KH*KW*Cin ordered (r, s, c) -- exactly the memory order of a
// channels-last filter, so the B operand is the filter as it lies in memory and the A operand is never materialised:
// K tile t is 32 channels of one tap, every lane of the staging wave adds the wave-uniform tap offset to the address
// of its row's patch origin (or reads a zero page when the tap falls into the padding).
// Requirements: Cin % 32 == 0, Cout % 8 == 0, KH*KW <= 32, 16-B aligned tensors, M < 2^31 - 256.
#include "gemm256p.h"
#include <stdlib.h>
#include <string.h>

namespace {

struct Variant { int nj, mi; double cost; };
// (NJ, MI) tiles built below with the measured time of ONE round of such tiles over the CUs (us at K = 4608,
// scripts/conv_fwd_bench.py; only the ratios matter).  Not proportional to the tile area: the pipeline is bound by LDS
// traffic, i.e. by MI + NJ fragment reads and by the number of LDS-DMA pieces per wave (MI = 4 stages ONE piece of A).
const Variant kVariants[] = {{4, 8, 90.}, {4, 7, 81.}, {4, 6, 79.}, {4, 5, 77.}, {4, 4, 64.}, {3, 8, 83.}, {3, 7, 77.},
                             {3, 6, 75.}, {3, 5, 74.}, {3, 4, 59.}, {2, 8, 70.}, {2, 4, 51.5}};
int force_nj = 0, force_mi = 0;
int num_cu = 0;
bool inited = false;

template <bool STATS, int NJ, int MI>
void launch(const GemmArgs& a, hipStream_t stream) {
    static bool attr = false;
    auto k = gemm256p_kernel<ACT_NONE, STATS, NJ, MI, true>;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        attr = true;
    }
    const int ntiles = a.ntm * a.ntn;
    hipLaunchKernelGGL(k, dim3(ntiles < num_cu ? ntiles : num_cu), dim3(512), 131072, stream, a, ntiles);
}

void init_once() {
    if (inited) return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || num_cu <= 0) num_cu = 256;
    const char* e = getenv("DTF_CONV_FWD_TILE");          // "nj,mi": pin the tile variant (A/B runs)
    if (e && force_nj == 0) {
        force_nj = atoi(e);
        const char* c = strchr(e, ',');
        force_mi = c ? atoi(c + 1) : 8;
    }
    inited = true;
}

// cheapest tiling of an M x N problem: rounds over the CUs x work per tile / tile efficiency
void pick(int64_t M, int N, int* nj, int* mi) {
    if (force_nj) { *nj = force_nj; *mi = force_mi ? force_mi : 8; return; }
    double best = 1e300;
    for (const Variant& v : kVariants) {
        const int64_t tiles = ((M + 32 * v.mi - 1) / (32 * v.mi)) * ((N + 64 * v.nj - 1) / (64 * v.nj));
        const int64_t rounds = (tiles + num_cu - 1) / num_cu;
        const double cost = (double)rounds * v.cost;
        if (cost < best) { best = cost; *nj = v.nj; *mi = v.mi; }
    }
}

}  // namespace

extern "C" {

void dtf_conv_fwd_set_tile(int nj, int mi) { init_once(); force_nj = nj; force_mi = mi; }

// rows of the statistics partials for an M-pixel output: two half tiles per row tile of the chosen variant
int64_t dtf_conv_fwd_stats_blocks(int64_t M, int N) {
    init_once();
    int nj = 4, mi = 8;
    pick(M, N, &nj, &mi);
    return 2 * ((M + 32 * mi - 1) / (32 * mi));
}

// y = conv2d(x, w, stride, padding) on NHWC bf16 tensors; w is [Cout][KH][KW][Cin] (a channels-last filter).
// stats (nullable): fp32 [dtf_conv_fwd_stats_blocks(M, Cout)][2][Cout] column sums / sums of squares of y.
// zero: device pointer to >= 64 B of zeros.
int dtf_conv_fwd(const void* x, const void* w, const void* zero, void* y, float* stats, int NB, int H, int W, int Cin,
                 int OH, int OW, int Cout, int KH, int KW, int stride, int ph, int pw, hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    init_once();
    const int64_t M64 = (int64_t)NB * OH * OW;
    if (M64 <= 0 || Cout <= 0) return 0;
    if (Cin <= 0 || (Cin & 31) || (Cout & 7) || KH * KW > 32 || KH < 1 || KW < 1 || stride < 1) return -50;
    if (M64 >= (1ll << 31) - 256) return -51;
    if ((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)y) | ((uintptr_t)zero) | ((uintptr_t)stats)) & 15) return -16;
    if (!zero) return -53;
    int nj = 4, mi = 8;
    pick(M64, Cout, &nj, &mi);
    GemmArgs a{};
    a.X = (const unsigned short*)x; a.W = (const unsigned short*)w; a.Y = (unsigned short*)y;
    a.M = (int)M64; a.N = Cout; a.K = KH * KW * Cin; a.act = ACT_NONE;
    a.ntm = (int)((M64 + 32 * mi - 1) / (32 * mi)); a.ntn = (Cout + 64 * nj - 1) / (64 * nj);
    a.stats = stats; a.zero = (const unsigned short*)zero;
    { static int gm = 0; if (gm <= 0) { const char* e = getenv("DTF_CONV_FWD_GROUP_M"); gm = e ? atoi(e) : 4; if (gm < 1) gm = 1; } a.group_m = gm; }
    a.cH = H; a.cW = W; a.cC = Cin; a.cOH = OH; a.cOW = OW; a.cKH = KH; a.cKW = KW; a.cS = stride; a.cPH = ph; a.cPW = pw;
#define CF_CASE(NJ_, MI_)                                                         \
    if (nj == NJ_ && mi == MI_) {                                                 \
        if (stats) launch<true, NJ_, MI_>(a, stream);                             \
        else launch<false, NJ_, MI_>(a, stream);                                  \
    } else
    CF_CASE(4, 8) CF_CASE(4, 7) CF_CASE(4, 6) CF_CASE(4, 5) CF_CASE(4, 4) CF_CASE(3, 8) CF_CASE(3, 7) CF_CASE(3, 6) CF_CASE(3, 5)
    CF_CASE(3, 4) CF_CASE(2, 8) CF_CASE(2, 4) return -54;
#undef CF_CASE
    DTF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
