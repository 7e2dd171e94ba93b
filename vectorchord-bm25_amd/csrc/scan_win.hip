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

// mt: the most indexed terms of a query of the batch; k: the top-k lives in 1, 2 or 4 register rows of 64 entries (beyond five run
// loads the kernel has registers for one row only: scan_win_max_k)
template <int MT>
static void launch_mt(const DevIndex &ix, const DevBatch &bt, uint32_t grid, hipStream_t st) {
    // (grid: workgroups; a workgroup is wn_waves(MT) independent waves)
    if (bt.k <= 64u) scan_win_kernel<MT, 1><<<grid, wn_waves(MT, 1) * 64, 0, st>>>(ix, bt);
    else if (bt.k <= 128u) scan_win_kernel<MT, 2><<<grid, wn_waves(MT, 2) * 64, 0, st>>>(ix, bt);
    else scan_win_kernel<MT, 4><<<grid, wn_waves(MT, 4) * 64, 0, st>>>(ix, bt);
}
// mt: the most indexed terms of a query of the batch: the kernel compiled for exactly that many run loads per window (2 .. 8; shorter
// queries get null terms.  One term: the two-load kernel -- compiled for a single run load the compiler copies the buffer's
// registers across the loop while their loads are in flight; tools/check_inflight.py finds it)
hipError_t scan_win_launch(const DevIndex &ix, const DevBatch &bt, uint32_t mt, uint32_t grid, hipStream_t st) {
    if (bt.k > scan_win_max_k(mt) || mt == 0 || mt > (uint32_t)WN_T) return hipErrorInvalidValue;
    switch (mt) {
        case 1:
        case 2: launch_mt<2>(ix, bt, grid, st); break;
        case 3: launch_mt<3>(ix, bt, grid, st); break;
        case 4: launch_mt<4>(ix, bt, grid, st); break;
        case 5: launch_mt<5>(ix, bt, grid, st); break;
        case 6: launch_mt<6>(ix, bt, grid, st); break;
        case 7: launch_mt<7>(ix, bt, grid, st); break;
        default: launch_mt<8>(ix, bt, grid, st); break;
    }
    return hipGetLastError();
}
// (the waves of a workgroup depend on the instantiation: the run loads mt the kernel is compiled for and the register rows of its top-k)
static uint32_t waves_of(uint32_t mt, uint32_t k) {
    const int rk = k <= 64u ? 1 : k <= 128u ? 2 : 4;
    switch (mt) {
        case 0:
        case 1:
        case 2: return uint32_t(wn_waves(2, rk));
        case 3: return uint32_t(wn_waves(3, rk));
        case 4: return uint32_t(wn_waves(4, rk));
        case 5: return uint32_t(wn_waves(5, rk));
        case 6: return uint32_t(wn_waves(6, rk));
        case 7: return uint32_t(wn_waves(7, rk));
        default: return uint32_t(wn_waves(8, rk));
    }
}
uint32_t scan_win_resident_waves(uint32_t mt, uint32_t k) { return WN_GRID * waves_of(mt, k); }
uint32_t scan_win_max_terms() { return WN_T; }
uint32_t scan_win_max_k(uint32_t) { return 256u; }  // (round 6: four register rows of the top-k with any number of run loads)
uint32_t scan_win_wg(uint32_t mt, uint32_t k) { return waves_of(mt, k); }
}  // namespace vbm25
