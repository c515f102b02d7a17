// smhip_ndt_api.hip -- host side of registrators::Ndt on the C ABI (included by smhip_api.hip).
//
// Ndt::Align (/root/reference/registrators/ndt.cc:38-64) = convert clouds, setInputTarget (voxel grid
// build, every call), pclomp NDT align, getFitnessScore.  The whole Align is ONE submission: table build, the rounds of the
// device-resident Newton / More-Thuente driver (ndt_kernels.hip: ndt_derivatives_ctl + ndt_ctl_step per round, a state machine
// per job in device memory), the fitness pass that takes its pose from where the driver left it, one synchronise at the end.
// The host enqueues as many rounds as the handle's previous Align needed (a front end's calls are alike), then the fitness
// pass; the jobs' flags in page-locked memory say whether that was enough -- if not, more rounds and the fitness pass again.
// K Aligns (smhip_ndt_align_batch; smhip_ndt_align is the batch of one) share every launch (grid.y = job): each job's
// evaluation sequence is exactly the one the reference walks, and a batch returns the single calls' bits.
#include "ndt_kernels.hip"

namespace {

struct NdtSlotMeta {                  // what the voxel table resident in a slot was built from
  bool valid = false;
  unsigned long long gen = 0;         // tgt_gen[slot]
  float resolution = 0.f; int min_points = 0; float eig_mult = 0.f;
};

struct NdtHost {
  int cap = 0;                        // pair slots [0, cap) have table storage
  NdtDev* devs_dev = nullptr;         // [cap] device array the kernels index
  NdtDev* devs_host = nullptr;        // [cap] page-locked mirror
  std::vector<NdtDev> devs_sent;      // what devs_dev holds
  NdtGridInfo* info_all = nullptr;    // per-slot storage, contiguous: [cap] ...
  uint32_t* bits_all = nullptr;       // [cap][kNdtMaxWords]
  uint2* words_all = nullptr;
  uint32_t* vstart_all = nullptr;     // [cap][nt_cap + 1]
  float4* vpts_all = nullptr;         // [cap][nt_cap]
  NdtVoxel* vox_all = nullptr;
  NdtCell* cells_all = nullptr;       // [cap][cells_cap] hash tables of the mid cells
  size_t cells_cap = 0;               // entries per slot: the power of two >= 2 nt_cap
  uint32_t* fpos_all = nullptr;       // [cap][nt_cap]
  float4* vbox_all = nullptr;         // [cap][nt_cap]
  double* vsum_all = nullptr;         // [cap][nt_cap][12]
  uint32_t* big_all = nullptr;        // [cap][nt_cap / kNdtBigVoxel + 1]
  uint32_t* qlist_all = nullptr;      // [cap][2][ns_cap]
  float* bbox_all = nullptr;          // [cap][kNdtBoxBlocks][8]
  float* fit_d2_all = nullptr;        // [cap][ns_cap]
  double* icovd_all = nullptr;        // [cap][nt_cap][6]
  double* partials_all = nullptr;     // [cap][rows][kNdtCols], rows = a workgroup of ndt_derivatives_ctl per 256 source points
  int rows = 0;
  double* out_all = nullptr;          // [cap][kNdtOutCols]
  NdtCtl* ctl_dev = nullptr; NdtCtl* ctl_host = nullptr;              // [cap] the jobs' state machines (job k of a batch = entry k)
  uint32_t* flags_pinned = nullptr;   // [cap] (round + 1) << 8 | phase, written by ndt_ctl_step
  NdtResult* res_pinned = nullptr;    // [cap]
  double* out_pinned = nullptr;       // [cap][kNdtOutCols] (test hook)
  NdtGridInfo* info_pinned = nullptr; // [cap]
  double* fit_pinned = nullptr;       // [cap][128]
  int32_t* vkey = nullptr;            // [nt_cap] linear voxel index of every occupied voxel of slot 0 (tests)
  PrepWorkspace* prep = nullptr;      // radix-sort workspace for cap * nt_cap (voxel code, point) pairs
  std::vector<void*> dev_allocs, host_allocs;
  std::vector<NdtSlotMeta> meta;
  smhip_ndt_options opts{};
  int deriv_calls = 0;                // of the last single Align (statistics)
  double last_pairs = 0;
  bool double_math = false;           // stock pcl::NormalDistributionsTransform arithmetic (NdtWithGicp)
  int predicted_rounds = 6;           // rounds the last Align of this handle needed: enqueued before the first look at the flags
  int last_rounds = 0, last_submissions = 0;   // statistics of the last Align: rounds enqueued, synchronisations it took
};

void euler_xyz_f32(const float* T /*row-major 4x4*/, float* e) {
  // Eigen 3.3 eulerAngles(0, 1, 2) on a Matrix3f (:109)
  auto R = [&](int r, int c) { return T[4 * r + c]; };
  float r0 = std::atan2(R(1, 2), R(2, 2));
  const float c2 = std::sqrt(R(0, 0) * R(0, 0) + R(0, 1) * R(0, 1));
  float r1;
  if (r0 > 0.f) { r0 -= (float)M_PI; r1 = std::atan2(-R(0, 2), -c2); }
  else r1 = std::atan2(-R(0, 2), c2);
  const float s1 = std::sin(r0), c1 = std::cos(r0);
  const float r2 = std::atan2(s1 * R(2, 0) - c1 * R(1, 0), c1 * R(1, 1) - s1 * R(2, 1));
  e[0] = -r0; e[1] = -r1; e[2] = -r2;
}

// the angular derivative tables of a pose (computeAngleDerivatives, :288-393), host side: the first evaluation of a job
void angle_derivatives(const double* p, bool dbl, NdtPose& P) {
  double cx, cy, cz, sx, sy, sz;
  if (std::fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = std::cos(p[3]); sx = std::sin(p[3]); }
  if (std::fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = std::cos(p[4]); sy = std::sin(p[4]); }
  if (std::fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = std::cos(p[5]); sz = std::sin(p[5]); }
  ndt_angle_tables(cx, sx, cy, sy, cz, sz, dbl, P);
}

}  // namespace

struct smhip_ndt_state { NdtHost n; };

namespace {

NdtHost& ndt_of(smhip_context* h);

void ndt_release(NdtHost& n) {
  for (void* p : n.dev_allocs) (void)hipFree(p);
  for (void* p : n.host_allocs) (void)hipHostFree(p);
  n.dev_allocs.clear(); n.host_allocs.clear();
  if (n.prep) { prep_destroy(n.prep); n.prep = nullptr; }
  n.cap = 0; n.meta.clear(); n.devs_sent.clear();
}

// table storage for pair slots [0, need): allocated on first use, re-allocated (all tables dropped) when a later batch needs more
smhip_status ndt_ensure(smhip_context* h, int need = 1) {
  NdtHost& n = ndt_of(h);
  if (need <= n.cap) return SMHIP_OK;
  if (h->dev.nt_cap > (1 << 24)) { h->err = "NDT: more than 2^24 target points per slot (the derivative kernel's pair entries hold 24-bit voxel slots)"; return SMHIP_ERR_CAPACITY; }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  ndt_release(n);
  const size_t K = (size_t)need, NT = h->dev.nt_cap;
  bool ok = true;
  auto D = [&](auto** p, size_t count) {
    void* v = nullptr;
    if (ok && hipMalloc(&v, count * sizeof(**p)) == hipSuccess) { n.dev_allocs.push_back(v); *p = reinterpret_cast<decltype(*p)>(v); } else ok = false;
  };
  auto P = [&](auto** p, size_t count) {
    void* v = nullptr;
    if (ok && hipHostMalloc(&v, count * sizeof(**p)) == hipSuccess) { n.host_allocs.push_back(v); *p = reinterpret_cast<decltype(*p)>(v); } else ok = false;
  };
  D(&n.devs_dev, K); D(&n.info_all, K); D(&n.bits_all, K * kNdtMaxWords); D(&n.words_all, K * kNdtMaxWords);
  const size_t NS = h->dev.ns_cap, NBIG = NT / kNdtBigVoxel + 1;
  D(&n.vstart_all, K * (NT + 1)); D(&n.vpts_all, K * NT); D(&n.vox_all, K * NT); D(&n.icovd_all, K * NT * 6);
  n.cells_cap = 1; while (n.cells_cap < 2 * NT) n.cells_cap <<= 1;
  D(&n.cells_all, K * n.cells_cap); D(&n.fpos_all, K * NT); D(&n.vbox_all, K * NT); D(&n.vsum_all, K * NT * 12); D(&n.big_all, K * NBIG); D(&n.qlist_all, K * 2 * NS); D(&n.bbox_all, K * (size_t)kNdtBoxBlocks * 8);
  D(&n.fit_d2_all, K * NS);
  n.rows = ceil_div(h->dev.ns_cap, kNdtDerivThreads);
  D(&n.partials_all, K * (size_t)n.rows * kNdtCols); D(&n.out_all, K * kNdtOutCols);
  D(&n.ctl_dev, K); D(&n.vkey, NT);
  P(&n.devs_host, K); P(&n.ctl_host, K); P(&n.flags_pinned, K); P(&n.res_pinned, K); P(&n.out_pinned, K * kNdtOutCols); P(&n.info_pinned, K); P(&n.fit_pinned, K * 128);
  if (ok && K * NT <= (size_t)0x7fffffff) { n.prep = prep_create((int)(K * NT)); ok = n.prep != nullptr; } else ok = false;
  if (!ok) { ndt_release(n); h->err = "NDT table allocation failed"; return SMHIP_ERR_HIP; }
  n.cap = need;
  n.meta.assign(K, NdtSlotMeta{});
  n.devs_sent.assign(K, NdtDev{});
  for (size_t k = 0; k < K; ++k) {
    NdtDev& d = n.devs_host[k];
    d = NdtDev{};
    d.info = n.info_all + k; d.bits = n.bits_all + k * kNdtMaxWords; d.words = n.words_all + k * kNdtMaxWords;
    d.vstart = n.vstart_all + k * (NT + 1); d.vpts = n.vpts_all + k * NT; d.vox = n.vox_all + k * NT; d.icovd = n.icovd_all + k * NT * 6;
    d.partials = n.partials_all + k * (size_t)n.rows * kNdtCols; d.out = n.out_all + k * kNdtOutCols;
    d.tgt = h->dev.tgt_p + k * NT; d.src = h->dev.src + k * NS;
    d.cells = n.cells_all + k * n.cells_cap; d.fpos = n.fpos_all + k * NT; d.vbox = n.vbox_all + k * NT; d.vsum = n.vsum_all + k * NT * 12; d.big = n.big_all + k * NBIG;
    d.qlist = n.qlist_all + k * 2 * NS; d.qleft = d.qlist + NS; d.bbox = n.bbox_all + k * (size_t)kNdtBoxBlocks * 8; d.fit_d2 = n.fit_d2_all + k * NS;
  }
  return SMHIP_OK;
}

__global__ void ndt_voxel_keys(const NdtDev* devs, int32_t* keys) {
  const NdtDev d = devs[0];
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= d.info->nocc) return;
  const float4 p = d.vpts[d.vstart[v]];
  int i0, i1, i2;
  ndt_voxel_of(d.info, p.x, p.y, p.z, i0, i1, i2);
  keys[v] = i0 + i1 * d.info->div_b[0] + i2 * d.info->div_b[0] * d.info->div_b[1];   // :223
}

bool ndt_table_current(const smhip_context* h, const NdtHost& n, int slot) {
  const NdtSlotMeta& m = n.meta[slot];
  return h->target_cache && m.valid && m.gen == h->tgt_gen[slot] && m.resolution == n.opts.resolution &&
         m.min_points == n.opts.min_points_per_voxel && m.eig_mult == n.opts.min_covar_eigvalue_mult;
}

// per-slot sizes / options into the device array the kernels index (a copy only when something changed since the last one)
smhip_status ndt_push_devs(smhip_context* h, int first, int K) {
  NdtHost& n = ndt_of(h);
  int off = 0, nt_max = 0;
  bool same = true;
  for (int k = first; k < first + K; ++k) nt_max = std::max(nt_max, h->nt[k]);
  int sbits = 1;
  while ((1ll << sbits) <= (long long)nt_max) ++sbits;        // slots < nt_max < 2^sbits: the all-ones key below never names a voxel
  for (int k = first; k < first + K; ++k) {
    NdtDev& d = n.devs_host[k];
    d.nt = h->nt[k]; d.ns = h->ns[k];
    d.min_points = n.opts.min_points_per_voxel; d.eig_mult = n.opts.min_covar_eigvalue_mult;
    d.key_off = off; off += d.nt;
    d.key_bits = sbits + 12;
    d.log2cells = 2; while ((1ll << d.log2cells) < 2ll * d.nt) ++d.log2cells;      // (never above cells_cap: nt <= nt_cap)
    same = same && std::memcmp(&d, &n.devs_sent[k], sizeof(NdtDev)) == 0;
  }
  if (same) return SMHIP_OK;
  HIPCHK(h, hipMemcpyAsync(n.devs_dev + first, n.devs_host + first, sizeof(NdtDev) * K, hipMemcpyHostToDevice, h->stream));
  for (int k = first; k < first + K; ++k) n.devs_sent[k] = n.devs_host[k];
  return SMHIP_OK;
}

// the pair input rows of slots [first, first + K) (sizes for tgt_reduce / the fitness search; the pose the search uses is written
// over `guess` on the device by the job's last step)
smhip_status ndt_push_inputs(smhip_context* h, int first, int K, const double* guesses) {
  for (int k = 0; k < K; ++k) {
    PairInput& in = h->in_pinned[first + k];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) in.guess[4 * r + c] = guesses ? guesses[16 * k + 4 * c + r] : (r == c ? 1.0 : 0.0);
    in.ns = h->ns[first + k]; in.nt = h->nt[first + k]; in.has_normals = 1; in.pad = 0;
  }
  HIPCHK(h, hipMemcpyAsync(const_cast<PairInput*>(h->dev.in) + first, h->in_pinned + first, sizeof(PairInput) * K, hipMemcpyHostToDevice, h->stream));
  return SMHIP_OK;
}

// VoxelGridCovariance::filter(true) on the targets of slots [first, first + K) (ndt_omp.h:117-122 -> init()): every kernel once
// for the whole batch (grid.y = table), ONE radix sort of all the (table, voxel code, point) keys.  Enqueue only: the tables'
// status words come back with the Align's one synchronise (ndt_tables_built); a box larger than the bit grid leaves an empty
// table that every later kernel walks harmlessly.  The pair input rows must be on the device (tgt_reduce reads nt).
smhip_status ndt_enqueue_grids(smhip_context* h, int first, int K) {
  NdtHost& n = ndt_of(h);
  smhip_status s = ndt_push_devs(h, first, K);
  if (s) return s;
  int nt_max = 0, nt_sum = 0;
  for (int k = first; k < first + K; ++k) { nt_max = std::max(nt_max, h->nt[k]); nt_sum += h->nt[k]; n.meta[k].valid = false; }
  const NdtDev* devs = n.devs_dev + first;
  const int gb = ceil_div(nt_max, 256);
  hipLaunchKernelGGL(ndt_bbox, dim3(kNdtBoxBlocks, K), dim3(256), 0, h->stream, devs);
  hipLaunchKernelGGL(ndt_voxel_setup, dim3(K), dim3(64), 0, h->stream, devs, n.opts.resolution);
  hipLaunchKernelGGL(ndt_cells_clear, dim3(std::max(64, 4096 / K), K), dim3(256), 0, h->stream, devs);
  hipLaunchKernelGGL(ndt_voxel_mark, dim3(ceil_div(nt_max, kNdtMarkThreads), K), dim3(kNdtMarkThreads), 0, h->stream, devs);
  hipLaunchKernelGGL(ndt_voxel_rank, dim3(K), dim3(1024), 0, h->stream, devs);
  // (slot << 6 | sub-cell, point) pairs sorted with the rocPRIM radix sort of the workspace: as few key bits as the largest
  // target of the batch can need.  The sorted points ARE vpts
  hipLaunchKernelGGL(ndt_voxel_keys64, dim3(gb, K), dim3(256), 0, h->stream, devs, prep_keys(n.prep, 0), prep_values(n.prep, 0));
  int kbits = 0;
  while ((1 << kbits) < K) ++kbits;
  const hipError_t e = prep_sort_pairs(n.prep, h->stream, nt_sum, n.devs_host[first].key_bits + kbits);
  if (e != hipSuccess) { h->err = std::string("NDT voxel sort: ") + hipGetErrorString(e); return SMHIP_ERR_HIP; }
  hipLaunchKernelGGL(ndt_voxel_heads, dim3(gb, K), dim3(256), 0, h->stream, devs, prep_keys(n.prep, 1), prep_values(n.prep, 1));
  // one wave per occupied voxel (nocc <= nt), a workgroup per crowded one
  hipLaunchKernelGGL(ndt_voxel_stats, dim3(std::min(ceil_div(nt_max, 4), std::max(64, 8192 / K)), K), dim3(256), 0, h->stream, devs);
  hipLaunchKernelGGL(ndt_voxel_stats_big, dim3(std::max(8, 256 / K), K), dim3(1024), 0, h->stream, devs);
  hipLaunchKernelGGL(ndt_voxel_leaves, dim3(std::min(ceil_div(nt_max, 64), std::max(32, 2048 / K)), K), dim3(64), 0, h->stream, devs);
  HIPCHK(h, hipGetLastError());
  return SMHIP_OK;
}
// ... after the stream has been synchronised
smhip_status ndt_tables_built(smhip_context* h, int first, int K) {
  NdtHost& n = ndt_of(h);
  for (int k = first; k < first + K; ++k) {
    if (n.info_pinned[k].status) { h->err = "NDT voxel box exceeds the bit grid (leaf size too small for the target extent)"; return SMHIP_ERR_CAPACITY; }
    NdtSlotMeta& m = n.meta[k];
    m.valid = true; m.gen = h->tgt_gen[k];
    m.resolution = n.opts.resolution; m.min_points = n.opts.min_points_per_voxel; m.eig_mult = n.opts.min_covar_eigvalue_mult;
  }
  return SMHIP_OK;
}
smhip_status ndt_build_grids(smhip_context* h, int first, int K) {      // build + wait (test hooks, GICP's stand-alone uses)
  smhip_status s = ndt_push_inputs(h, first, K, nullptr);
  if (s == SMHIP_OK) s = ndt_enqueue_grids(h, first, K);
  if (s) return s;
  HIPCHK(h, hipMemcpyAsync(ndt_of(h).info_pinned + first, ndt_of(h).info_all + first, sizeof(NdtGridInfo) * K, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return ndt_tables_built(h, first, K);
}

NdtCtlOpts ndt_ctl_opts(const NdtHost& n) {
  NdtCtlOpts o{};
  o.step_size = n.opts.step_size; o.trans_eps = n.opts.transformation_epsilon; o.max_iterations = n.opts.max_iterations;
  const double c1 = 10.0 * (1 - (double)n.opts.outlier_ratio);                     // :86-93
  const double c2 = (double)n.opts.outlier_ratio / std::pow((double)n.opts.resolution, 3);
  const double d3 = -std::log(c2);
  o.d1 = -std::log(c1 + c2) - d3;
  o.d2 = -2 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - d3) / o.d1);
  o.res2 = n.opts.resolution * n.opts.resolution;
  o.double_math = n.double_math ? 1 : 0;
  return o;
}

// a job's state before its first evaluation (:98-119): final_transformation_ = guess.cast<float>() (ndt.cc:58), p from its
// translation and Euler angles, the first computeDerivatives at p
void ndt_ctl_start(const NdtHost& n, const NdtCtlOpts& o, NdtCtl& j, int slot, const double* guess_cm, const double* pose6, int32_t phase) {
  std::memset(&j, 0, sizeof(j));
  j.slot = slot;
  if (guess_cm) {
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) j.Tf[4 * r + c] = (float)guess_cm[4 * c + r];
    float e[3];
    euler_xyz_f32(j.Tf, e);
    j.p[0] = j.Tf[3]; j.p[1] = j.Tf[7]; j.p[2] = j.Tf[11]; j.p[3] = e[0]; j.p[4] = e[1]; j.p[5] = e[2];   // :107-111
  } else {
    for (int i = 0; i < 6; ++i) j.p[i] = pose6[i];
    const float a = (float)pose6[3], b = (float)pose6[4], c = (float)pose6[5];
    ndt_pose_matrix_f32(pose6, std::cos(a), std::sin(a), std::cos(b), std::sin(b), std::cos(c), std::sin(c), j.Tf);
  }
  for (int i = 0; i < 6; ++i) j.eval_p[i] = j.p[i];
  j.phase = phase;
  NdtPose& P = j.pose;
  for (int i = 0; i < 12; ++i) P.T[i] = j.Tf[i];
  angle_derivatives(j.p, n.double_math, P);
  P.d1d = o.d1; P.d2d = o.d2; P.d1 = (float)o.d1; P.d2 = (float)o.d2;
  P.res2 = o.res2;
  P.compute_hessian = 1;
}

// one round for jobs [0, K) of the batch whose first slot is `first`
void ndt_enqueue_round(smhip_context* h, int first, int K, int blocks, const NdtCtlOpts& o, int round, double* out_host) {
  NdtHost& n = ndt_of(h);
  const dim3 g(blocks, K);
  if (n.double_math) hipLaunchKernelGGL(ndt_derivatives_ctl<double>, g, dim3(kNdtDerivThreads), 0, h->stream, n.devs_dev, n.ctl_dev + first);
  else hipLaunchKernelGGL(ndt_derivatives_ctl<float>, g, dim3(kNdtDerivThreads), 0, h->stream, n.devs_dev, n.ctl_dev + first);
  hipLaunchKernelGGL(ndt_ctl_step, dim3(K), dim3(kNdtStepThreads), 0, h->stream, n.devs_dev, n.ctl_dev + first, blocks, o, round,
                     const_cast<PairInput*>(h->dev.in), n.flags_pinned + first, n.res_pinned + first, out_host);
}

// pcl::Registration::getFitnessScore(): mean squared 1-NN distance of each slot's source, moved by its pose, to the slot's raw
// target (ndt.cc:60, ndt_gicp.cc:88,101), for slots [first, first + K) in one pass.  In two parts: the search structure over the
// raw targets (independent of the pose: enqueued before the rounds), and the search itself with the pose that stands in the
// slots' pair input rows on the device.
smhip_status fitness_enqueue_search(smhip_context* h, int first, int K) {
  NdtHost& n = ndt_of(h);
  int ns_max = 0;
  for (int k = first; k < first + K; ++k) ns_max = std::max(ns_max, h->ns[k]);
  PairInput* in = const_cast<PairInput*>(h->dev.in);
  hipLaunchKernelGGL(ndt_fit_reset, dim3(ceil_div(K, 64)), dim3(64), 0, h->stream, n.devs_dev, first, K);
  hipLaunchKernelGGL(ndt_fit_near, dim3(ceil_div(ns_max, 256), K), dim3(256), 0, h->stream, n.devs_dev, in, first);
  hipLaunchKernelGGL(ndt_fit_mid, dim3(std::max(64, kNdtFitMidBlocks / K), K), dim3(256), 0, h->stream, n.devs_dev, in, first);
  hipLaunchKernelGGL(ndt_fit_far, dim3(std::max(64, kNdtFitFarBlocks / K), K), dim3(256), 0, h->stream, n.devs_dev, in, first);
  hipLaunchKernelGGL(fitness_partial, dim3(64, K), dim3(256), 0, h->stream, n.devs_dev, first, n.fit_pinned);
  HIPCHK(h, hipGetLastError());
  return SMHIP_OK;
}

void fitness_collect(smhip_context* h, int K, double* out) {        // after the synchronise
  NdtHost& n = ndt_of(h);
  for (int k = 0; k < K; ++k) {
    double ssum = 0, cnt = 0;
    for (int b = 0; b < 64; ++b) { ssum += n.fit_pinned[128 * k + 2 * b]; cnt += n.fit_pinned[128 * k + 2 * b + 1]; }
    out[k] = cnt > 0 ? ssum / cnt : 1.7976931348623157e308;
  }
  h->ev_used = 0;
}

// the fitness score of slots [first, first + K) at host-side poses T (column-major 4x4 each): GICP's closing score, the C ABI's
// stand-alone fitness call
smhip_status fitness_scores(smhip_context* h, int first, int K, const double* T, double* out) {
  for (int k = first; k < first + K; ++k)
    if (h->ns[k] <= 0 || h->nt[k] <= 0) { h->err = "fitness score before SetInputSource/SetInputTarget"; return SMHIP_ERR_NOT_READY; }
  smhip_status s = ndt_ensure(h, first + K);
  if (s) return s;
  NdtHost& n = ndt_of(h);
  s = ndt_push_inputs(h, first, K, T);
  if (s) return s;
  // the search structure is the slots' voxel tables: built here unless they are current
  bool current = true;
  for (int k = first; k < first + K; ++k) current = current && ndt_table_current(h, n, k);
  if (current) s = ndt_push_devs(h, first, K);
  else s = ndt_enqueue_grids(h, first, K);
  if (s == SMHIP_OK) s = fitness_enqueue_search(h, first, K);
  if (s) return s;
  if (!current) HIPCHK(h, hipMemcpyAsync(n.info_pinned + first, n.info_all + first, sizeof(NdtGridInfo) * K, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (!current) { s = ndt_tables_built(h, first, K); if (s) return s; }
  fitness_collect(h, K, out);
  return SMHIP_OK;
}
smhip_status fitness_score(smhip_context* h, const double* T, double* out) { return fitness_scores(h, 0, 1, T, out); }

// K Aligns in lock-step on slots [first, first + K)
smhip_status ndt_align_slots(smhip_context* h, int first, int K, const double* guesses, double* results, double* scores, smhip_ndt_stats* stats) {
  for (int k = first; k < first + K; ++k)
    if (h->ns[k] <= 0 || h->nt[k] <= 0) { h->err = "Ndt::Align before SetInputSource/SetInputTarget"; return SMHIP_ERR_NOT_READY; }   // ndt.cc:40-42
  HIPCHK(h, hipSetDevice(h->device));
  smhip_status s = ndt_ensure(h, first + K);
  if (s) return s;
  NdtHost& n = ndt_of(h);
  s = ndt_push_inputs(h, first, K, guesses);
  if (s) return s;
  // setInputTarget -> init() on every Align (ndt.cc:54).  The voxel table is a pure function of the target and the
  // options, so it is kept while the slot's target is unchanged (smhip_set_target_cache(h, 0) = rebuild every time); a batch
  // keeps its tables when every one of them is current, else rebuilds them all in one pass
  bool current = true;
  for (int k = first; k < first + K; ++k) current = current && ndt_table_current(h, n, k);
  if (current) {
    s = ndt_push_devs(h, first, K);
    if (s) return s;
    h->cache_hits++;
  } else {
    s = ndt_enqueue_grids(h, first, K);
    if (s) return s;
  }
  const NdtCtlOpts o = ndt_ctl_opts(n);
  int ns_max = 0;
  for (int k = first; k < first + K; ++k) ns_max = std::max(ns_max, h->ns[k]);
  // (workgroups per evaluation from the LARGEST source: a smaller cloud's surplus workgroups add zero rows, and the fold of the
  // rows is grouped by workgroup index, so a pair's sums are the bits its single call gives)
  const int blocks = std::max(1, ceil_div(ns_max, kNdtDerivThreads));
  for (int k = 0; k < K; ++k) {
    ndt_ctl_start(n, o, n.ctl_host[first + k], first + k, guesses + 16 * k, nullptr, kNdtInit);
    n.flags_pinned[first + k] = 0;
  }
  HIPCHK(h, hipMemcpyAsync(n.ctl_dev + first, n.ctl_host + first, sizeof(NdtCtl) * K, hipMemcpyHostToDevice, h->stream));
  // every evaluation of a job is a round; the reference's own bounds: max_iterations + 2 Newton iterations of at most 1 + 10 + 1
  const int round_cap = (std::max(0, o.max_iterations) + 2) * 12 + 2;
  int rounds = 0, want = std::max(1, std::min(n.predicted_rounds, round_cap));
  n.last_submissions = 0;
  for (bool speculate = true;; speculate = false) {
    for (; rounds < want; ++rounds) ndt_enqueue_round(h, first, K, blocks, o, rounds, nullptr);
    // the fitness pass behind the predicted rounds: valid if every job has ended by then (a handle's Aligns are alike: it nearly
    // always has).  After a wrong prediction: rounds only, twice as many each time, and the fitness pass once the flags say done
    if (speculate) { s = fitness_enqueue_search(h, first, K); if (s) return s; }
    HIPCHK(h, hipMemcpyAsync(n.info_pinned + first, n.info_all + first, sizeof(NdtGridInfo) * K, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    n.last_submissions++;
    bool all_done = true;
    for (int k = 0; k < K; ++k) all_done = all_done && (n.flags_pinned[first + k] & 0xffu) == (uint32_t)kNdtDone;
    if (all_done) {
      if (!speculate) {
        s = fitness_enqueue_search(h, first, K);
        if (s) return s;
        HIPCHK(h, hipMemcpyAsync(n.info_pinned + first, n.info_all + first, sizeof(NdtGridInfo) * K, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        n.last_submissions++;
      }
      break;
    }
    if (rounds >= round_cap) { h->err = "NDT: a job did not end within the reference's own iteration bounds"; return SMHIP_ERR_HIP; }
    want = std::min(round_cap, 2 * rounds);
  }
  if (!current) { s = ndt_tables_built(h, first, K); if (s) return s; }
  if (std::getenv("SMHIP_NDT_DEBUG")) {
    std::fprintf(stderr, "ndt align of %d job(s): %d rounds enqueued, %d submissions, predicted %d\n", K, rounds, n.last_submissions, n.predicted_rounds);
    for (int k = 0; k < std::min(K, 8); ++k) {
      const NdtGridInfo& gi = n.info_pinned[first + k];
      std::fprintf(stderr, "ndt job %d: ns %d nt %d voxels %d big %d; fitness: %u queries past the fine cube, %u past the mid shells\n", k, h->ns[first + k], h->nt[first + k], gi.nocc, gi.nbig, gi.nlist, gi.nleft);
    }
  }
  int needed = 1;
  for (int k = 0; k < K; ++k) needed = std::max(needed, n.res_pinned[first + k].done_round + 1);
  n.predicted_rounds = needed;
  n.last_rounds = rounds;
  // getFinalTransformation().cast<double>(), column-major out (ndt.cc:61)
  for (int k = 0; k < K; ++k)
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) results[16 * k + 4 * c + r] = (double)n.res_pinned[first + k].Tf[4 * r + c];
  std::vector<double> fit(K);
  fitness_collect(h, K, fit.data());
  for (int k = 0; k < K; ++k) {
    const NdtResult& r = n.res_pinned[first + k];
    if (scores) scores[k] = fit[k];
    if (stats) {
      stats[k].iterations = r.it;
      stats[k].derivative_calls = r.deriv_calls;
      stats[k].voxels = n.info_pinned[first + k].nocc;
      stats[k].status = 0;
      stats[k].trans_probability = r.sc / (double)h->ns[first + k];      // :170
      stats[k].pairs_last = r.last_pairs;
    }
  }
  n.deriv_calls = n.res_pinned[first].deriv_calls; n.last_pairs = n.res_pinned[first].last_pairs;
  return SMHIP_OK;
}

}  // namespace

extern "C" {

void smhip_ndt_default_options(smhip_ndt_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->resolution = 1.0f;               // ndt.cc:31
  o->step_size = 0.1f;                // ndt_omp_impl.hpp:50
  o->outlier_ratio = 0.55f;           // :51
  o->transformation_epsilon = 0.1f;   // :71
  o->max_iterations = 35;             // :72
  o->min_points_per_voxel = 6;        // voxel_grid_covariance_omp.h:204
  o->min_covar_eigvalue_mult = 0.01f; // :205
}

smhip_status smhip_ndt_set_options(smhip_handle h, const smhip_ndt_options* o) {
  if (!h || !o) return SMHIP_ERR_INVALID_ARGUMENT;
  if (!(o->resolution > 0) || !(o->step_size > 0) || o->max_iterations < 0 || o->min_points_per_voxel < 3) {
    h->err = "bad NDT options";
    return SMHIP_ERR_INVALID_ARGUMENT;
  }
  NdtHost& n = ndt_of(h);
  n.opts = *o;
  for (auto& m : n.meta) m.valid = false;
  return SMHIP_OK;
}

smhip_status smhip_ndt_align(smhip_handle h, const double guess[16], double result[16], double* score, smhip_ndt_stats* stats) {
  if (!h || !guess || !result) return SMHIP_ERR_INVALID_ARGUMENT;
  return ndt_align_slots(h, 0, 1, guess, result, score, stats);
}

smhip_status smhip_ndt_align_batch(smhip_handle h, int first_slot, int npairs, const double* guesses, double* results, double* scores, smhip_ndt_stats* stats) {
  if (!h || !guesses || !results || npairs < 1 || first_slot < 0 || first_slot + npairs > h->dev.slots) {
    if (h) h->err = "bad slot range / guesses";
    return SMHIP_ERR_INVALID_ARGUMENT;
  }
  return ndt_align_slots(h, first_slot, npairs, guesses, results, scores, stats);
}

// Test hooks: build the voxel grid of slot 0's target / evaluate computeDerivatives at a pose.
smhip_status smhip_ndt_build_voxels(smhip_handle h, int* n_voxels) {
  if (!h) return SMHIP_ERR_INVALID_ARGUMENT;
  if (h->nt[0] <= 0) { h->err = "target not set"; return SMHIP_ERR_NOT_READY; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  smhip_status s = ndt_ensure(h, 1);
  if (s) return s;
  s = ndt_build_grids(h, 0, 1);
  if (s) return s;
  if (n_voxels) *n_voxels = ndt_of(h).info_pinned[0].nocc;
  return SMHIP_OK;
}

smhip_status smhip_ndt_get_voxels(smhip_handle h, int capacity, int32_t* keys, int32_t* counts, double* means, float* icovs, float* centroids) {
  if (!h) return SMHIP_ERR_INVALID_ARGUMENT;
  NdtHost& n = ndt_of(h);
  if (n.cap < 1 || !n.meta[0].valid) { h->err = "voxel grid not built"; return SMHIP_ERR_NOT_READY; }
  const int nocc = n.info_pinned[0].nocc;
  if (capacity < nocc) { h->err = "capacity too small"; return SMHIP_ERR_CAPACITY; }
  HIPCHK(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(ndt_voxel_keys, dim3(ceil_div(nocc, 256)), dim3(256), 0, h->stream, n.devs_dev, n.vkey);
  std::vector<NdtVoxel> vox(nocc);
  std::vector<int32_t> k(nocc);
  HIPCHK(h, hipMemcpyAsync(vox.data(), n.vox_all, sizeof(NdtVoxel) * nocc, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(k.data(), n.vkey, sizeof(int32_t) * nocc, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (int v = 0; v < nocc; ++v) {
    if (keys) keys[v] = k[v];
    if (counts) counts[v] = vox[v].n;
    if (means) for (int c = 0; c < 3; ++c) means[3 * v + c] = vox[v].mean[c];
    if (icovs) for (int c = 0; c < 6; ++c) icovs[6 * v + c] = vox[v].icov[c];
    if (centroids) for (int c = 0; c < 3; ++c) centroids[3 * v + c] = vox[v].centroid[c];
  }
  return SMHIP_OK;
}

smhip_status smhip_ndt_compute_derivatives(smhip_handle h, const double pose6[6], int compute_hessian, double* score, double grad[6], double hess[36]) {
  if (!h || !pose6 || !score || !grad || !hess) return SMHIP_ERR_INVALID_ARGUMENT;
  NdtHost& n = ndt_of(h);
  if (n.cap < 1 || !n.meta[0].valid) { h->err = "voxel grid not built"; return SMHIP_ERR_NOT_READY; }
  if (h->ns[0] <= 0) { h->err = "source not set"; return SMHIP_ERR_NOT_READY; }
  HIPCHK(h, hipSetDevice(h->device));
  smhip_status s = ndt_push_devs(h, 0, 1);
  if (s) return s;
  // one round of the device driver with a job that only wants this evaluation
  const NdtCtlOpts o = ndt_ctl_opts(n);
  ndt_ctl_start(n, o, n.ctl_host[0], 0, nullptr, pose6, kNdtEvalOnly);
  n.ctl_host[0].pose.compute_hessian = compute_hessian != 0;
  n.flags_pinned[0] = 0;
  HIPCHK(h, hipMemcpyAsync(n.ctl_dev, n.ctl_host, sizeof(NdtCtl), hipMemcpyHostToDevice, h->stream));
  ndt_enqueue_round(h, 0, 1, std::max(1, ceil_div(h->ns[0], kNdtDerivThreads)), o, 0, n.out_pinned);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipGetLastError());
  *score = n.out_pinned[0];
  for (int i = 0; i < 6; ++i) grad[i] = n.out_pinned[1 + i];
  for (int i = 0; i < 36; ++i) hess[i] = compute_hessian ? n.out_pinned[7 + i] : 0.0;
  n.last_pairs = n.out_pinned[43];
  n.deriv_calls++;
  return SMHIP_OK;
}

smhip_status smhip_ndt_time_derivatives(smhip_handle h, int first_slot, int npairs, int launches, double* ms_per_launch, double* pairs_per_launch) {
  if (!h || !ms_per_launch || npairs < 1 || launches < 1 || first_slot < 0 || first_slot + npairs > h->dev.slots) return SMHIP_ERR_INVALID_ARGUMENT;
  NdtHost& n = ndt_of(h);
  if (n.cap < first_slot + npairs) { h->err = "no NDT tables for these slots"; return SMHIP_ERR_NOT_READY; }
  for (int k = first_slot; k < first_slot + npairs; ++k)
    if (!n.meta[k].valid || h->ns[k] <= 0) { h->err = "voxel table / source of a slot missing"; return SMHIP_ERR_NOT_READY; }
  HIPCHK(h, hipSetDevice(h->device));
  smhip_status s = ndt_push_devs(h, first_slot, npairs);
  if (s) return s;
  // jobs that stay in the 'evaluation wanted' state (no control launch runs): every launch does the same work
  const NdtCtlOpts o = ndt_ctl_opts(n);
  int ns_max = 0;
  for (int k = 0; k < npairs; ++k) {
    double G[16];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) G[4 * c + r] = (double)n.res_pinned[first_slot + k].Tf[4 * r + c];
    ndt_ctl_start(n, o, n.ctl_host[first_slot + k], first_slot + k, G, nullptr, kNdtTrial);
    ns_max = std::max(ns_max, h->ns[first_slot + k]);
  }
  HIPCHK(h, hipMemcpyAsync(n.ctl_dev + first_slot, n.ctl_host + first_slot, sizeof(NdtCtl) * npairs, hipMemcpyHostToDevice, h->stream));
  const dim3 g(std::max(1, ceil_div(ns_max, kNdtDerivThreads)), npairs);
  hipEvent_t a, b;
  HIPCHK(h, hipEventCreate(&a)); HIPCHK(h, hipEventCreate(&b));
  auto launch = [&]() {
    if (n.double_math) hipLaunchKernelGGL(ndt_derivatives_ctl<double>, g, dim3(kNdtDerivThreads), 0, h->stream, n.devs_dev, n.ctl_dev + first_slot);
    else hipLaunchKernelGGL(ndt_derivatives_ctl<float>, g, dim3(kNdtDerivThreads), 0, h->stream, n.devs_dev, n.ctl_dev + first_slot);
  };
  launch();                                                  // warm
  HIPCHK(h, hipEventRecord(a, h->stream));
  for (int r = 0; r < launches; ++r) launch();
  HIPCHK(h, hipEventRecord(b, h->stream));
  // the pair count of these evaluations: one control round on top (it ends the jobs' EvalOnly lives and leaves the sums)
  for (int k = 0; k < npairs; ++k) n.ctl_host[first_slot + k].phase = kNdtEvalOnly;
  HIPCHK(h, hipMemcpyAsync(n.ctl_dev + first_slot, n.ctl_host + first_slot, sizeof(NdtCtl) * npairs, hipMemcpyHostToDevice, h->stream));
  ndt_enqueue_round(h, first_slot, npairs, (int)g.x, o, 0, n.out_pinned);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipGetLastError());
  float ms = 0;
  HIPCHK(h, hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  *ms_per_launch = (double)ms / launches;
  if (pairs_per_launch) { double p = 0; for (int k = 0; k < npairs; ++k) p += n.out_pinned[(size_t)k * kNdtOutCols + 43]; *pairs_per_launch = p; }
  return SMHIP_OK;
}

// tuning hook (not in smhip.h): the rows of partial sums the last evaluation of slot 0 left, as raw 8-byte words
smhip_status smhip_ndt_debug_rows(smhip_handle h, unsigned long long* out, int max_rows, int* rows) {
  if (!h || !out || !rows) return SMHIP_ERR_INVALID_ARGUMENT;
  NdtHost& n = ndt_of(h);
  if (n.cap < 1) { h->err = "no NDT state"; return SMHIP_ERR_NOT_READY; }
  *rows = std::min(max_rows, std::max(1, ceil_div(h->ns[0], kNdtDerivThreads)));
  HIPCHK(h, hipMemcpy(out, n.partials_all, sizeof(double) * kNdtCols * (size_t)*rows, hipMemcpyDeviceToHost));
  return SMHIP_OK;
}

}  // extern "C"
