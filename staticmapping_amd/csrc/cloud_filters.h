// cloud_filters.h -- internal interface of the device-side pre-filters (cloud_filters.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

struct smhip_filter_desc;

namespace smhip {
struct FilterWorkspace;
FilterWorkspace* filt_create(int max_points);
void filt_destroy(FilterWorkspace* w);
// staged_dev: n device rows (x, y, z, intensity); factor: host array of n floats or NULL (= i / n, the collector's rule)
hipError_t filt_set_input(FilterWorkspace* w, hipStream_t st, const float4* staged_dev, const float* factor_host_or_null, int n);
// applies the filters in order to the current cloud; blocks until the size is known
hipError_t filt_run_chain(FilterWorkspace* w, hipStream_t st, const smhip_filter_desc* chain, int nf, int* n_out);
const float4* filt_points(const FilterWorkspace* w);         // x y z intensity
const float* filt_factors(const FilterWorkspace* w);
const int32_t* filt_source_index(const FilterWorkspace* w);  // row of the ORIGINAL input each point came from (-1 after VoxelGrid)
int filt_count(const FilterWorkspace* w);
bool filt_has_index(const FilterWorkspace* w);
}  // namespace smhip
