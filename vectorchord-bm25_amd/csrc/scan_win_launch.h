// scan_win_launch.h -- what search.hip sees of scan_win.hip (included inside namespace vbm25, after device_types.h)
hipError_t scan_win_launch(const DevIndex &ix, const DevBatch &bt, uint32_t mt, uint32_t grid, hipStream_t st);  // mt: the most indexed terms of a query of the batch
uint32_t scan_win_resident_waves(uint32_t mt, uint32_t k);  // persistent waves of a full grid (mt: as above; k: the batch's)
uint32_t scan_win_max_terms();       // indexed terms per query
uint32_t scan_win_max_k(uint32_t mt);  // the largest k of a batch whose queries have at most mt indexed terms
uint32_t scan_win_wg(uint32_t mt, uint32_t k);  // waves (= work items in flight) per workgroup
