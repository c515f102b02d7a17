// smhip_filter_api.hip -- C ABI of the device pre-filters (included by smhip_api.hip); kernels in cloud_filters.hip.
// Mirrors pre_processers::filter::{Range, AxisRange, BoundingBoxRemoval, RandomSampler, VoxelGrid, Factory}
// (/root/reference/pre_processors/filter_*.cc): constructor defaults, ConfigsValid() and Filter().
#include <cfloat>

extern "C" {

void smhip_filter_default(int type, smhip_filter_desc* f) {
  if (!f) return;
  std::memset(f, 0, sizeof(*f));
  f->type = type;
  switch (type) {
    case SMHIP_FILTER_RANGE: f->p[0] = 0.f; f->p[1] = FLT_MAX; break;                         // filter_range.cc:33-36
    case SMHIP_FILTER_AXIS_RANGE: f->p[0] = -FLT_MAX; f->p[1] = FLT_MAX; f->axis_index = 2; break;   // filter_axis_range.h:56-57, .cc:28
    case SMHIP_FILTER_RANDOM_SAMPLER: f->p[0] = 1.f; break;                                   // filter_random_sample.cc:28
    case SMHIP_FILTER_VOXEL_GRID: f->p[0] = 0.1f; break;                                      // filter_voxel_grid.h:55
    case SMHIP_FILTER_BOUNDING_BOX_REMOVAL:                                                   // filter_bounding_box.h:53-58
      f->p[0] = f->p[1] = f->p[2] = -FLT_MAX; f->p[3] = f->p[4] = f->p[5] = FLT_MAX; break;
    default: f->type = 0; break;
  }
}

int smhip_filter_config_valid(const smhip_filter_desc* f) {
  if (!f) return 0;
  switch (f->type) {
    case SMHIP_FILTER_RANGE: return 1;                                                        // no override: Interface default
    case SMHIP_FILTER_AXIS_RANGE: return (f->p[1] > f->p[0]) && f->axis_index >= 0 && f->axis_index <= 2;   // filter_axis_range.cc:40-42
    case SMHIP_FILTER_RANDOM_SAMPLER: return f->p[0] >= 0.f && f->p[0] <= 1.f;                 // filter_random_sample.cc:34-36
    case SMHIP_FILTER_VOXEL_GRID: return f->p[0] > 1.e-6;                                      // filter_voxel_grid.cc:36
    case SMHIP_FILTER_BOUNDING_BOX_REMOVAL: return f->p[0] < f->p[3] && f->p[1] < f->p[4] && f->p[2] < f->p[5];   // filter_bounding_box.cc:49-51
  }
  return 0;
}

static smhip_status filter_ensure(smhip_handle h) {
  smhip_status s = prep_ensure(h);
  if (s) return s;
  if (h->filt) return SMHIP_OK;
  h->filt = filt_create(std::max(h->dev.ns_cap, h->dev.nt_cap));
  if (!h->filt) { h->err = "filter workspace allocation failed"; return SMHIP_ERR_HIP; }
  return SMHIP_OK;
}

smhip_status smhip_filter_chain_f32(smhip_handle h, const float* points, int stride_floats, int n, const smhip_filter_desc* chain,
                                    int n_filters, int* n_out) {
  if (!h || !points || n < 0 || n_filters < 0 || (n_filters > 0 && !chain) || (stride_floats != 4 && stride_floats != 5)) {
    if (h) h->err = "bad arguments (stride must be 4 = x y z intensity or 5 = InnerPointType)";
    return SMHIP_ERR_INVALID_ARGUMENT;
  }
  for (int k = 0; k < n_filters; ++k)
    if (!smhip_filter_config_valid(&chain[k])) { h->err = "filter " + std::to_string(k) + ": ConfigsValid() is false"; return SMHIP_ERR_INVALID_ARGUMENT; }
  if (n > std::max(h->dev.ns_cap, h->dev.nt_cap)) { h->err = "cloud larger than the handle's capacity"; return SMHIP_ERR_CAPACITY; }
  HIPCHK(h, hipSetDevice(h->device));
  smhip_status s = filter_ensure(h);
  if (s) return s;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  // stage rows as (x, y, z, intensity); the factors of stride-5 rows travel in the second half of the staging buffer
  float* fac = reinterpret_cast<float*>(h->stage + std::max(h->dev.ns_cap, h->dev.nt_cap));
  for (int i = 0; i < n; ++i) {
    const float* r = points + (size_t)stride_floats * i;
    h->stage[i] = make_float4(r[0], r[1], r[2], r[3]);
    if (stride_floats == 5) fac[i] = r[4];
  }
  HIPCHK(h, hipMemcpyAsync(h->prep_raw, h->stage, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, h->stream));
  hipError_t e = filt_set_input(h->filt, h->stream, h->prep_raw, stride_floats == 5 ? fac : nullptr, n);
  if (e == hipSuccess) e = filt_run_chain(h->filt, h->stream, chain, n_filters, n_out);
  if (e != hipSuccess) { h->err = std::string("filter chain: ") + hipGetErrorString(e); return e == hipErrorInvalidValue ? SMHIP_ERR_INVALID_ARGUMENT : SMHIP_ERR_HIP; }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (n_out) *n_out = filt_count(h->filt);
  return SMHIP_OK;
}

smhip_status smhip_filter_get_output(smhip_handle h, float* points5, int32_t* source_index, int n) {
  if (!h || !h->filt) { if (h) h->err = "no filter chain has run"; return h ? SMHIP_ERR_NOT_READY : SMHIP_ERR_INVALID_ARGUMENT; }
  if (n != filt_count(h->filt)) { h->err = "n must equal the filtered size"; return SMHIP_ERR_INVALID_ARGUMENT; }
  if (n == 0) return SMHIP_OK;
  HIPCHK(h, hipSetDevice(h->device));
  std::vector<float4> p((size_t)n);
  std::vector<float> f((size_t)n);
  HIPCHK(h, hipMemcpyAsync(p.data(), filt_points(h->filt), sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(f.data(), filt_factors(h->filt), sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  if (source_index) HIPCHK(h, hipMemcpyAsync(source_index, filt_source_index(h->filt), sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (points5)
    for (int i = 0; i < n; ++i) {
      points5[5 * (size_t)i] = p[i].x; points5[5 * (size_t)i + 1] = p[i].y; points5[5 * (size_t)i + 2] = p[i].z;
      points5[5 * (size_t)i + 3] = p[i].w; points5[5 * (size_t)i + 4] = f[i];
    }
  return SMHIP_OK;
}

// the filtered cloud becomes SetInputSource of `slot` without leaving the device
smhip_status smhip_filter_output_to_source(smhip_handle h, int slot) {
  smhip_status s = check_slot(h, slot);
  if (s) return s;
  if (!h->filt) { h->err = "no filter chain has run"; return SMHIP_ERR_NOT_READY; }
  const int n = filt_count(h->filt);
  if (n <= 0) { h->err = "the filtered cloud is empty"; return SMHIP_ERR_INVALID_ARGUMENT; }
  if (n > h->dev.ns_cap) { h->err = "filtered cloud larger than max_source_points"; return SMHIP_ERR_CAPACITY; }
  HIPCHK(h, hipSetDevice(h->device));
  const hipError_t e = prep_morton_sort(h->prep, h->stream, filt_points(h->filt), n, const_cast<float4*>(h->dev.src) + (size_t)slot * h->dev.ns_cap);
  if (e != hipSuccess) { h->err = std::string("prep_morton_sort: ") + hipGetErrorString(e); return SMHIP_ERR_HIP; }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->ns[slot] = n;
  touch_source(h, slot);
  return SMHIP_OK;
}

}  // extern "C"
