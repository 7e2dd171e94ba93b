// scan_win.hip -- translation unit of scan_win_kernel (scan_win.h): the device types and helpers of search.hip, the kernel, and the
// launcher search.hip calls.  A unit of its own so that the dominant kernel of C3 compiles in seconds, not with the other eight.
#include <hip/hip_runtime.h>
#include <cstdio>

#include "vbm25_internal.h"

namespace vbm25 {
#include "device_types.h"
#include "decode.h"
#include "topk_lds.h"
#include "block_fetch.h"
#include "topk_reg.h"
#include "scan_win.h"
#include "scan_win_launch.h"

// mt: the most indexed terms of a query of the batch -- the kernel is compiled for 2, 4, 5 and 8 run loads per window; k: the top-k
// lives in 1, 2 or 4 register rows of 64 entries (the eight-load kernel has registers for one row only: scan_win_max_k)
template <int MT>
static void launch_mt(const DevIndex &ix, const DevBatch &bt, uint32_t grid, hipStream_t st) {
    // (grid: workgroups; a workgroup is wn_waves(MT) independent waves)
    if (bt.k <= 64u) scan_win_kernel<MT, 1><<<grid, wn_waves(MT) * 64, 0, st>>>(ix, bt);
    else if constexpr (MT <= 5) {
        if (bt.k <= 128u) scan_win_kernel<MT, 2><<<grid, wn_waves(MT) * 64, 0, st>>>(ix, bt);
        else scan_win_kernel<MT, 4><<<grid, wn_waves(MT) * 64, 0, st>>>(ix, bt);
    }
}
hipError_t scan_win_launch(const DevIndex &ix, const DevBatch &bt, uint32_t mt, uint32_t grid, hipStream_t st) {
    if (bt.k > scan_win_max_k(mt)) return hipErrorInvalidValue;
    if (mt <= 2) launch_mt<2>(ix, bt, grid, st);
    else if (mt <= 4) launch_mt<4>(ix, bt, grid, st);
    else if (mt <= 5) launch_mt<5>(ix, bt, grid, st);
    else launch_mt<8>(ix, bt, grid, st);
    return hipGetLastError();
}
static uint32_t waves_of(uint32_t mt) { return uint32_t(mt <= 2 ? wn_waves(2) : mt <= 4 ? wn_waves(4) : mt <= 5 ? wn_waves(5) : wn_waves(8)); }
uint32_t scan_win_resident_waves(uint32_t mt) { return WN_GRID * waves_of(mt); }
uint32_t scan_win_max_terms() { return WN_T; }
uint32_t scan_win_max_k(uint32_t mt) { return mt <= 5 ? 256u : 64u; }
uint32_t scan_win_wg(uint32_t mt) { return waves_of(mt); }
}  // namespace vbm25
