// Id sort for the K4 sparse apply (SURVEY 2.4 scatter-add path): a stable sort of the looked-up row ids with their
// positions.  torch.sort on int64 runs the library radix sort over all 64 key bits (11 one-sweep launches, 0.21-0.23 ms
// for 1.7 M ids = 7 % of a Wide&Deep step); a table of R rows needs floor(log2 R) + 1 bits -- 3 passes of 8 bits for the
// 2.6 M rows of config #3, 4 for the 312 M rows of the 180 GB run.  rocPRIM's device radix sort (header-only, compiled
// here for gfx950) with the bit range restricted, ids outside [0, R) (padding slots of the routed exchange) folded onto
// the single key R by a transform iterator so that no second pass over the ids is needed, and a counting iterator
// as the value input (no arange tensor).
#include "dtf_common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

struct DtfFoldInvalid {
    int64_t R;
    __host__ __device__ int64_t operator()(int64_t v) const { return (v < 0 || v >= R) ? R : v; }
};

static inline unsigned sort_bits(int64_t R) {
    unsigned b = 1;
    while (b < 63 && ((int64_t)1 << b) <= R) ++b;       // 2^b > R: covers the fold key R itself
    return b;
}

extern "C" {

int64_t dtf_sort_ids_temp_bytes(int64_t n, int64_t R) {
    if (n <= 0 || R <= 0) return 0;
    size_t bytes = 0;
    rocprim::transform_iterator<const int64_t*, DtfFoldInvalid, int64_t> kin((const int64_t*)nullptr, DtfFoldInvalid{R});
    rocprim::counting_iterator<int64_t> vin(0);
    const hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, kin, (int64_t*)nullptr, vin, (int64_t*)nullptr,
                                                   (size_t)n, 0u, sort_bits(R), (hipStream_t)0, false);
    return e == hipSuccess ? (int64_t)bytes : -1;
}

// sid[n]: ids sorted ascending (ids outside [0, R) appear as R, after every valid id), perm[n]: their positions in
// `ids`, stable.  temp: dtf_sort_ids_temp_bytes(n, R) bytes.
int dtf_sort_ids(const int64_t* ids, int64_t* sid, int64_t* perm, int64_t n, int64_t R, void* temp, int64_t temp_bytes,
                 hipStream_t stream) {
    DTF_CLEAR_ERRORS();
    if (n <= 0) return 0;
    if (R <= 0 || !ids || !sid || !perm || !temp) return -70;
    size_t bytes = (size_t)temp_bytes;
    rocprim::transform_iterator<const int64_t*, DtfFoldInvalid, int64_t> kin(ids, DtfFoldInvalid{R});
    rocprim::counting_iterator<int64_t> vin(0);
    const hipError_t e = rocprim::radix_sort_pairs(temp, bytes, kin, sid, vin, perm, (size_t)n, 0u, sort_bits(R), stream,
                                                   false);
    if (e != hipSuccess) return -71;
    DTF_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
