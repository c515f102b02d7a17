// prep_normals.h -- internal interface of the device-side CalculateNormals (prep_normals.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace smhip {
struct PrepWorkspace;
PrepWorkspace* prep_create(int max_points);
void prep_destroy(PrepWorkspace* w);
// raw: n device points (xyz in .x .y .z).  out_p / out_n: device arrays with room for n / 4 + 8 entries.
// Blocks until *m_host (number of surviving points) is known.
hipError_t prep_calculate_normals(PrepWorkspace* w, hipStream_t st, const float4* raw, int n, float4* out_p,
                                  float4* out_n, int* m_host);
// The same for S scans in one pass (one forest, one sort per tree level for all of them): scan s =
// raw[offset[s] .. offset[s] + n[s]), written to out_p / out_n at out_offset[s]; m_host[s] survivors.
// The workspace must hold sum(n) points.
hipError_t prep_calculate_normals_batch(PrepWorkspace* w, hipStream_t st, const float4* raw, int S, const int* offset,
                                        const int* n, const int* out_offset, float4* out_p, float4* out_n, int* m_host);
// allocates what a batch of >= 32 scans needs (the one-workgroup-per-scan forest) ahead of the first such batch
hipError_t prep_reserve_forest(PrepWorkspace* w);
// Morton-order `raw` into `out` (out[k].w = index of the point in `raw`); asynchronous on `st`.
hipError_t prep_morton_sort(PrepWorkspace* w, hipStream_t st, const float4* raw, int n, float4* out);
// The same for S clouds at once: cloud s = stage[stage_off[s] .. + n[s]) (device), Morton-ordered into out_base + out_off[s];
// one key pass, ONE radix sort on (cloud, Morton key) and one gather for the whole batch.  Within a cloud the order equals
// prep_morton_sort's whenever the cloud spans < 16 km (18 instead of 21 key bits per axis make room for the cloud id).
// The workspace must hold sum(n) points; S <= 512.  Asynchronous on `st`.
hipError_t prep_morton_sort_batch(PrepWorkspace* w, hipStream_t st, const float4* stage, int S, const int* stage_off, const int* n,
                                  const long long* out_off, float4* out_base);
// Random sub-sampling of a Morton-ordered cloud (in[k].w = caller index): keeps the points whose counter-based uniform of
// (seed, caller index) is < prob, in the input's order, with out[k].w = index in the sampled cloud in caller order.
// Blocks until *m_host (points kept) is known.  in != out.
hipError_t prep_sample_morton(PrepWorkspace* w, hipStream_t st, const float4* in, int n, float prob, uint32_t seed, float4* out, int* m_host);
// pcl::ApproximateVoxelGrid (leaf x leaf x leaf) of `raw` (n points, arrival order = array order) into `out`
// (room for n entries; out[k].w = k).  Blocks until *m_host (number of centroids) is known.
hipError_t prep_approx_voxel_grid(PrepWorkspace* w, hipStream_t st, const float4* raw, int n, float leaf, float4* out, int* m_host);
// The same for S clouds at once (S <= 64; the workspace must hold sum(n) points): cloud c = raw[c][0 .. n[c]) (device pointers) into
// out[c] (room for n[c] entries), m_host[c] centroids; per cloud exactly prep_approx_voxel_grid's output.  One synchronise for the batch.
hipError_t prep_approx_voxel_grid_batch(PrepWorkspace* w, hipStream_t st, int S, const float4* const* raw, const int* n, float leaf,
                                        float4* const* out, int* m_host);
// Generic use of the workspace's radix sort by other builders (the NDT voxel grid): fill prep_keys(w, 0) / prep_values(w, 0)
// with n (key, value) pairs, call prep_sort_pairs, read the sorted pairs from prep_keys(w, 1) / prep_values(w, 1).
unsigned long long* prep_keys(PrepWorkspace* w, int which);
int32_t* prep_values(PrepWorkspace* w, int which);
hipError_t prep_sort_pairs(PrepWorkspace* w, hipStream_t st, int n, int end_bit);
}  // namespace smhip
