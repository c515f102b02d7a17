// smhip_api.hip -- host side of the C ABI declared in include/smhip.h.
//
// One smhip_context = one matcher instance in the sense of
// static_map::registrator::Interface (/root/reference/registrators/interface.h:67-116):
// it owns a stream, a device arena sized at creation (no hipMalloc inside Align) and
// `slots` independent scan pairs.  The IcpFast loop (icp_fast.cc:455-529) stays resident on
// the device; the host only enqueues launches and, when early exit is on, polls one word
// every `check_every` iterations.
#include "icp_kernels.hip"
#include "icp_one.hip"
#include "nabo_kernels.hip"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/smhip.h"
#include "prep_normals.h"
#include "cloud_filters.h"

using namespace smhip;

struct smhip_ndt_state;
struct smhip_gicp_state;

struct smhip_context {
  smhip_ndt_state* ndt = nullptr;
  smhip_gicp_state* gicp = nullptr;
  PrepWorkspace* prep = nullptr;          // device CalculateNormals workspace (allocated on first use)
  PrepWorkspace* prep_batch = nullptr;    // the same sized for every slot at once (batched target preparation)
  FilterWorkspace* filt = nullptr;        // device pre-filters (allocated on first use)
  float4* prep_raw = nullptr;             // raw scan staging on the device
  float4* raw_batch = nullptr;            // the same for a whole batch of scans (smhip_set_sources_f32_batch; allocated on first use)
  hipStream_t copy_stream = nullptr;      // host-to-device copies of a batch of page-locked scans (overlap the handle's stream)
  hipEvent_t ev_copied = nullptr, ev_raw_free = nullptr;
  bool raw_in_use = false;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // a batch is split into up to kMaxParts parts on separate streams so that the latency-bound kernels of one part
  // (finalize, validate, grid build) hide behind the NN / accumulate kernels of the others
  static constexpr int kMaxParts = 4;
  hipStream_t side[kMaxParts - 1] = {};
  hipEvent_t ev_fork = nullptr, ev_join[kMaxParts - 1] = {};
  int n_side = 0;
  IcpDev dev{};
  KdDev kd{};                    // SMHIP_NN_NABO: tree arrays, allocated on first use
  bool kd_allocated = false;
  smhip_icp_options opts{};
  std::vector<int> ns, nt, has_normals;
  // Target-side structures are kept across calls while a slot's target is unchanged (single-pair calls only: the front end
  // aligns scan after scan against one key frame, map_builder.cc:379-392).  tgt_gen[slot] changes whenever the slot's target
  // does; grid_gen / grid_cell / grid_sorted describe the search structure currently resident in the slot.
  std::vector<unsigned long long> tgt_gen, grid_gen;
  std::vector<unsigned long long> src_gen, src3_gen;   // same idea for the packed 12-byte copy of a slot's source (pack_source)
  std::vector<float> grid_cell_built;
  std::vector<int> grid_sorted, grid_rows, grid_mode;   // grid_rows: row-occupancy bitmap built too; grid_mode: nn_mode of the structure
  unsigned long long gen_counter = 0;
  int nabo_listed_blocks = kNaboListedBlocks;   // workgroups per pair of the list walk (SMHIP_NABO_LISTED_BLOCKS overrides, tuning only)
  int target_cache = 1;             // smhip_set_target_cache
  unsigned long long cache_hits = 0;
  PairInput* in_pinned = nullptr;
  PairState* state_pinned = nullptr;
  int hist_mode = 0;             // nn_mode of the batch whose searched-query history is waiting in hist_pinned
  int nabo_fused_from = 6;       // reference-search mode: the first iteration of a batch that runs the fused certificate pass (fused_iteration)
  float split_share = 0.2f;      // auto split: the first iteration whose median searched share falls below this runs certify + listed search
  int sums_blocks = kSumsBlocks;  // workgroups of iteration_sums (SMHIP_SUMS_BLOCKS)
  int sums_long_for = 3;         // fused iterations of a batch whose missed pairs iteration_sums cuts into long blocks (SMHIP_SUMS_LONG_FOR)
  int use_shadow = 1;            // fused certificate pass reads the 4-byte shadow of (bound, match) where every target is small enough (SMHIP_SHADOW)
  int one_blocks = 0;            // workgroups of the single-pair persistent kernel the device holds at once (0: not available)
  int one_used = 0;              // the last single-pair enqueue went through it
  int one_blocks_allowed = 1;    // 0: fine-grained memory could not be had at smhip_create
  long long one_launches = 0;    // enqueues that went through it (smhip_icp_single_launch_counts)
  int one_fallbacks = 0;         // Aligns done again as separate launches because the launch stopped itself (see fetch_range)
  int one_enabled = 1;           // SMHIP_ONE_PAIR=0: single pairs through the separate launches (measurement aid)
  int one_groups_want = 0;       // SMHIP_ONE_GROUPS: groups of its barrier (tuning)
  int one_blocks_want = 0;       // SMHIP_ONE_BLOCKS: its grid (tuning; 0 = as many as a round each needs, at most what is resident)
  int wave_search = 0;           // batches: the every-query-searches iterations through nn_ball_lds (0, default: 5-25 % faster on the bench scans)
                                 // or nn_ball_wave (1; SMHIP_WAVE_SEARCH=1) -- same results
  float4* stage = nullptr;       // pinned staging for uploads, 2 * max(ns_cap, nt_cap)
  uint32_t* done_pinned = nullptr;
  // split_after = 0: where the batched iterations switch from the fused search to certify + listed search follows the
  // previous batch (the share of queries that needed a search per iteration, search_hist): a front end's guesses are
  // alike from call to call.  Results do not depend on it, only the time.
  uint32_t* hist_pinned = nullptr;
  int hist_first = 0;
  int hist_pairs = 0;                  // pairs whose rows the last enqueue copied to hist_pinned (0 = none)
  std::vector<int> hist_ns;            // their sources' sizes at that enqueue (the slots may hold other clouds by the time the rows are read)
  int hist_iters = 0;                  // iterations that enqueue ran
  int auto_split = 2;
  int32_t* ids_pinned = nullptr;
  float* d2_pinned = nullptr;
  int32_t* ids_dev = nullptr;    // scratch for exported matches
  float* d2_dev = nullptr;
  std::vector<void*> allocs;
  std::string err;
  int last_npairs = 0;
  // profiling
  int profile = 0;               // 0 off, 1 every launch, 2 the dominant NN kernel only
  struct Ev { hipEvent_t a, b; int cat; int np; };
  std::vector<Ev> ev_pool;
  size_t ev_used = 0;
  smhip_icp_profile prof{};
};

namespace {

thread_local std::string g_create_error;

#define HIPCHK(h, expr)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                       \
      return SMHIP_ERR_HIP;                                                               \
    }                                                                                     \
  } while (0)

template <typename T>
smhip_status dev_alloc(smhip_context* h, T** p, size_t count) {
  void* v = nullptr;
  HIPCHK(h, hipMalloc(&v, count * sizeof(T)));
  h->allocs.push_back(v);
  *p = reinterpret_cast<T*>(v);
  return SMHIP_OK;
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// profiling brackets: category 0 prepare, 1 the refinement launches of FindClosests (validate / ring / fallback), 2 error_elements,
// 3 solve, 4 the main NN kernel (fused search, or the full libnabo walk), 5 the certificate pass, 6 the listed search / list walk
struct Bracket {
  smhip_context* h;
  smhip_context::Ev* ev = nullptr;
  hipStream_t st;
  Bracket(smhip_context* h_, int cat, hipStream_t st_, int np = 0) : h(h_), st(st_) {
    // 1: every launch.  2 / 3 / 4: ONE kernel class only -- the NN kernels proper (fused search / full walk and the certificate
    // pass), accumulate, or the listed search -- which costs a timed region one event pair per iteration and part (every
    // bracket is a barrier between two launches: all classes at once took 6 % off the batch rate)
    if (!h->profile) return;
    if (h->profile == 2 && cat != 4 && cat != 5) return;
    if (h->profile == 3 && cat != 2) return;
    if (h->profile == 4 && cat != 6) return;
    if (h->ev_used == h->ev_pool.size()) {
      smhip_context::Ev e{};
      if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
      h->ev_pool.push_back(e);
    }
    ev = &h->ev_pool[h->ev_used++];
    ev->cat = cat;
    ev->np = np;
    (void)hipEventRecord(ev->a, st);
  }
  ~Bracket() {
    if (ev) (void)hipEventRecord(ev->b, st);
  }
};

void collect_profile(smhip_context* h) {
  if (!h->profile) return;
  for (size_t k = 0; k < h->ev_used; ++k) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, h->ev_pool[k].a, h->ev_pool[k].b) != hipSuccess) continue;
    switch (h->ev_pool[k].cat) {
      case 0: h->prof.ms_prepare += ms; break;
      case 1: h->prof.ms_find_closests += ms; h->prof.launches_find_closests++; h->prof.ms_nn_refine += ms; h->prof.launches_nn_refine++; break;
      case 6: h->prof.ms_find_closests += ms; h->prof.launches_find_closests++;
              h->prof.ms_nn_listed += ms; h->prof.launches_nn_listed++; h->prof.pairs_nn_listed += h->ev_pool[k].np; break;
      case 4: h->prof.ms_find_closests += ms; h->prof.launches_find_closests++;
              h->prof.ms_nn_main += ms; h->prof.launches_nn_main++; h->prof.pairs_nn_main += h->ev_pool[k].np; break;
      case 5: h->prof.ms_find_closests += ms; h->prof.launches_find_closests++;
              h->prof.ms_nn_certify += ms; h->prof.launches_nn_certify++; h->prof.pairs_nn_certify += h->ev_pool[k].np; break;
      case 2: h->prof.ms_error_elements += ms; h->prof.launches_error_elements++; h->prof.pairs_error_elements += h->ev_pool[k].np; break;
      case 3: h->prof.ms_solve += ms; h->prof.launches_solve++; break;
    }
  }
  h->ev_used = 0;
}

inline void touch_target(smhip_context* h, int slot) { h->tgt_gen[slot] = ++h->gen_counter; }
inline void touch_source(smhip_context* h, int slot) { h->src_gen[slot] = ++h->gen_counter; }
inline void touch_grid(smhip_context* h, int first, int np) {        // the slots' search structures are (re)built / overwritten
  for (int p = first; p < first + np; ++p) { h->grid_gen[p] = 0; h->grid_cell_built[p] = 0.f; h->grid_sorted[p] = 0; h->grid_rows[p] = 0; h->grid_mode[p] = -1; }
}

smhip_status check_slot(smhip_context* h, int slot) {
  if (!h) return SMHIP_ERR_INVALID_ARGUMENT;
  if (slot < 0 || slot >= h->dev.slots) { h->err = "slot out of range"; return SMHIP_ERR_INVALID_ARGUMENT; }
  return SMHIP_OK;
}

// One half of a batch: a by-value copy of the device view restricted to pairs [pair_base, pair_base + np)
// and the stream its launches go to.
struct Half {
  IcpDev d;
  hipStream_t stream;
  int np;
  bool small = false;      // few workgroups per launch: use the single-round NN / short-chunk accumulate variants
  int first_fused = -1;    // the first iteration of this Align that ran the fused path
};

// the ICP iteration kernels stream the 12-byte copy of the sources: repacked here (main stream, the PairInput rows already
// on their way) for launches that cover a slot whose source changed since it was last packed
smhip_status ensure_packed(smhip_context* h, int first, int np) {
  bool stale = false;
  for (int p = first; p < first + np; ++p) stale = stale || h->src3_gen[p] != h->src_gen[p];
  if (!stale) return SMHIP_OK;
  hipLaunchKernelGGL(pack_source, dim3(std::min(8192, 128 * np)), dim3(256), 0, h->stream, h->dev, first, np);
  HIPCHK(h, hipGetLastError());
  for (int p = first; p < first + np; ++p) h->src3_gen[p] = h->src_gen[p];
  return SMHIP_OK;
}

// per-call resets for pairs [0, np) (main stream, before the halves fork)
smhip_status enqueue_resets(smhip_context* h, int np, int first = 0) {
  IcpDev& d = h->dev;
  touch_grid(h, first, np);               // bits / ccount are zeroed below: whatever structure was resident is gone
  HIPCHK(h, hipMemcpyAsync(const_cast<PairInput*>(d.in) + first, h->in_pinned + first, sizeof(PairInput) * np, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(reset_scratch, dim3(std::min(4096, 256 * np)), dim3(256), 0, h->stream, d, first, np);
  return ensure_packed(h, first, np);
}

// target centring + search-structure build for one half
smhip_status kd_ensure(smhip_context* h) {
  if (h->kd_allocated) return SMHIP_OK;
  const size_t B = h->dev.slots, NT = h->dev.nt_cap;
  h->kd.node_cap = (int32_t)(NT / 2 + 8);       // a split node has >= 9 points and gives each child >= 4: <= nt / 4 leaves, 2 leaves - 1 nodes
  h->kd.seg_cap = (int32_t)(NT / 4 + 8);
  smhip_status s = SMHIP_OK;
  auto A = [&](smhip_status r) { if (s == SMHIP_OK) s = r; };
  A(dev_alloc(h, &h->kd.nodes, B * (size_t)h->kd.node_cap));
  A(dev_alloc(h, &h->kd.segs, B * 2 * (size_t)h->kd.seg_cap));
  A(dev_alloc(h, &h->kd.alt, B * NT));
  A(dev_alloc(h, &h->kd.cnt, B * 2 * (size_t)h->kd.seg_cap));
  h->kd.leaf_cap = (int32_t)(NT / 4 + 2);       // block index = bucket start >> 2
  A(dev_alloc(h, &h->kd.leaf, B * (size_t)h->kd.leaf_cap * 24));
  A(dev_alloc(h, &h->dev.nabo_work, B * (size_t)h->dev.ns_cap));
  if (s == SMHIP_OK) h->kd_allocated = true;
  return s;
}

smhip_status enqueue_grid_build(smhip_context* h, const Half& f, int nt_max) {
  const IcpDev& d = f.d;
  const int np = f.np;
  Bracket br(h, 0, f.stream);
  const dim3 gpts(ceil_div(nt_max, 256), np);
  hipLaunchKernelGGL(tgt_reduce, dim3(kTgtReduceBlocks, np), dim3(256), 0, f.stream, d);
  hipLaunchKernelGGL(grid_setup, dim3(ceil_div(np, 64)), dim3(64), 0, f.stream, d, np);
  if (h->opts.nn_mode == SMHIP_NN_NABO) {
    // the reference's own structure: libnabo's kd-tree over the centred target, rebuilt per Align (icp_fast.cc:464-467)
    if (!h->kd_allocated) { h->err = "nn_mode NABO: tree arrays missing (smhip_icp_set_options allocates them)"; return SMHIP_ERR_NOT_READY; }
    // the search keeps one pending sibling per tree level in kKdStack LDS slots per query; a median split halves (rounding
    // up) until a bucket holds <= 8 points, so 8 << kKdStack points is the deepest tree the stack can follow
    if (nt_max > (kKdBucket << kKdStack)) { h->err = "nn_mode NABO: target larger than 8 << 18 points (the search stack holds 18 tree levels)"; return SMHIP_ERR_CAPACITY; }
    hipLaunchKernelGGL(kd_build, dim3(np), dim3(kKdThreads), 0, f.stream, d, h->kd);
    HIPCHK(h, hipGetLastError());
    for (int p = d.pair_base; p < d.pair_base + np; ++p) {
      h->grid_gen[p] = h->tgt_gen[p]; h->grid_cell_built[p] = d.grid_cell; h->grid_sorted[p] = 1; h->grid_rows[p] = 0; h->grid_mode[p] = SMHIP_NN_NABO;
    }
    return SMHIP_OK;
  }
  hipLaunchKernelGGL(grid_mark, gpts, dim3(256), 0, f.stream, d);
  // the ring searches (NDT fitness, GICP neighbourhoods / correspondences) skip empty grid rows through a bitmap; the ball
  // search of IcpFast visits a handful of rows per query and does not need it
  if (d.have_rowbits) hipLaunchKernelGGL(grid_rowbits, dim3(ceil_div(kMaxRowWords, 256), np), dim3(256), 0, f.stream, d);
  hipLaunchKernelGGL(grid_rank, dim3(std::max(1, std::min(16, 64 / np)), np), dim3(1024), 0, f.stream, d);   // segments per pair when pairs are few
  hipLaunchKernelGGL(grid_count, gpts, dim3(256), 0, f.stream, d);
  hipLaunchKernelGGL(grid_cscan, dim3(np), dim3(1024), 0, f.stream, d);
  if (d.sort_cells) {
    hipLaunchKernelGGL(grid_scatter_idx, gpts, dim3(256), 0, f.stream, d);
    hipLaunchKernelGGL(grid_place, gpts, dim3(256), 0, f.stream, d);
  } else {
    hipLaunchKernelGGL(grid_scatter, gpts, dim3(256), 0, f.stream, d);
  }
  HIPCHK(h, hipGetLastError());
  for (int p = d.pair_base; p < d.pair_base + np; ++p) {
    h->grid_gen[p] = h->tgt_gen[p]; h->grid_cell_built[p] = d.grid_cell; h->grid_sorted[p] = d.sort_cells; h->grid_rows[p] = d.have_rowbits;
    h->grid_mode[p] = SMHIP_NN_GRID;
  }
  return SMHIP_OK;
}

Half whole_batch(smhip_context* h, int np, int first);

// is the search structure resident in `slot` the one a build with the current settings would produce?
bool grid_cached(smhip_context* h, int slot) {
  const int want_mode = h->opts.nn_mode == SMHIP_NN_NABO ? SMHIP_NN_NABO : SMHIP_NN_GRID;   // which structure: kd-tree or grid
  if (h->grid_mode[slot] != want_mode) return false;
  if (want_mode == SMHIP_NN_NABO) return h->target_cache && h->grid_gen[slot] != 0 && h->grid_gen[slot] == h->tgt_gen[slot];
  return h->target_cache && h->grid_gen[slot] != 0 && h->grid_gen[slot] == h->tgt_gen[slot] &&
         h->grid_cell_built[slot] == h->dev.grid_cell && h->grid_sorted[slot] >= h->dev.sort_cells &&
         h->grid_rows[slot] >= (h->dev.use_ball ? 0 : 1);
}

// single-pair form of enqueue_resets + enqueue_grid_build that skips the build when the slot's target is unchanged
smhip_status enqueue_prepare_one(smhip_context* h, int slot, int nt_max) {
  if (!grid_cached(h, slot)) {
    smhip_status s = enqueue_resets(h, 1, slot);
    if (s) return s;
    return enqueue_grid_build(h, whole_batch(h, 1, slot), nt_max);
  }
  IcpDev d = h->dev; d.npairs = 1; d.pair_base = slot; d.have_rowbits = h->dev.use_ball ? 0 : 1;
  HIPCHK(h, hipMemcpyAsync(const_cast<PairInput*>(d.in) + slot, h->in_pinned + slot, sizeof(PairInput), hipMemcpyHostToDevice, h->stream));
  { smhip_status ps = ensure_packed(h, slot, 1); if (ps) return ps; }
  hipLaunchKernelGGL(reset_scratch_light, dim3(8), dim3(256), 0, h->stream, d, slot, 1);
  hipLaunchKernelGGL(pose_setup, dim3(1), dim3(64), 0, h->stream, d, 1);
  HIPCHK(h, hipGetLastError());
  h->cache_hits++;
  return SMHIP_OK;
}

Half whole_batch(smhip_context* h, int np, int first) {
  Half f;
  f.d = h->dev; f.d.npairs = np; f.d.pair_base = first;
  f.d.have_rowbits = h->dev.use_ball ? 0 : 1;     // ring-search contexts build and use the row-occupancy bitmap
  f.stream = h->stream; f.np = np;
  return f;
}

// single-stream convenience used by find_closests / the NDT fitness pass
smhip_status enqueue_prepare(smhip_context* h, int np, int nt_max) {
  if (np == 1) return enqueue_prepare_one(h, 0, nt_max);
  smhip_status s = enqueue_resets(h, np);
  if (s) return s;
  return enqueue_grid_build(h, whole_batch(h, np, 0), nt_max);
}

smhip_status enqueue_find_closests_half(smhip_context* h, const Half& f, int ns_max, int iteration);

inline int nt_max_of(const smhip_context* h, int first, int np) {
  int m = 0;
  for (int p = first; p < first + np; ++p) m = std::max(m, h->nt[p]);
  return m;
}

smhip_status enqueue_find_closests(smhip_context* h, int np, int ns_max) {
  return enqueue_find_closests_half(h, whole_batch(h, np, 0), ns_max, 0);
}

smhip_status enqueue_find_closests_half(smhip_context* h, const Half& f, int ns_max, int iteration) {
  const IcpDev& d = f.d;
  const int np = f.np;
  hipStream_t st = f.stream;
  const dim3 g(ceil_div(ns_max, kNnThreads), np);
  if (h->opts.nn_mode == SMHIP_NN_NABO) {
    // knn(k = 1, epsilon) through libnabo's tree: what it returns IS the match (no bounds, nothing to refine).  Iteration 0
    // walks every query and records its traversal certificate; later iterations re-walk only the queries that have moved
    // further than their certificate allows (nabo_kernels.hip)
    KdDev kd = h->kd;
    const float e = h->opts.nn_epsilon >= 0.f ? h->opts.nn_epsilon : 3.16f;
    kd.max_error2 = (1.0f + e) * (1.0f + e);
    const int nb1 = ceil_div(ns_max, kNnThreads);
    const bool shallow = nt_max_of(h, d.pair_base, np) <= (kKdBucket << 12);     // 12 stack levels (40 KiB of LDS) cover the target
    if (d.certify && iteration > 0) {
      {
        Bracket br(h, 5, st, np);
        if (f.small) {
          hipLaunchKernelGGL((nn_certify<1, true>), dim3(nb1 * 8 * ceil_div(np, 8)), dim3(kNnThreads), 0, st, d, nb1);
        } else if (d.fused) {
          // certificate pass + the sums below the predicted quantile band in one pass over the source (fused_iteration decides)
          const int nbc = ceil_div(ns_max, kNnThreads * kCertifyItems);
          hipLaunchKernelGGL((nn_certify_acc<kCertifyItems, true>), dim3(nbc * 8 * ceil_div(np, 8)), dim3(kNnThreads), 0, st, d, nbc);
        } else {
          const int nbc = ceil_div(ns_max, kNnThreads * kCertifyItems);
          hipLaunchKernelGGL((nn_certify<kCertifyItems, true>), dim3(nbc * 8 * ceil_div(np, 8)), dim3(kNnThreads), 0, st, d, nbc);
        }
      }
      {
        Bracket br(h, 6, st, np);
        const int nbl = f.small ? nb1 : h->nabo_listed_blocks;
        const dim3 gl(nbl * 8 * ceil_div(np, 8));
        if (shallow) hipLaunchKernelGGL((nn_nabo<1, true, 12>), gl, dim3(kNnThreads), 0, st, d, kd, nbl);
        else hipLaunchKernelGGL((nn_nabo<1, true, kKdStack>), gl, dim3(kNnThreads), 0, st, d, kd, nbl);
      }
      if (d.fused) {
        // the walked queries by the fused pass's rule, and the check of its prediction (this mode's nn_validate)
        Bracket br(h, 1, st);
        hipLaunchKernelGGL(nabo_validate, dim3(np), dim3(kAccThreads), 0, st, d);
        hipLaunchKernelGGL(accumulate_listed, dim3(kNaboAccBlocks, np), dim3(kAccThreads), 0, st, d);
      }
    } else {
      Bracket br(h, 4, st, np);
      if (f.small) {
        const dim3 g1(nb1 * 8 * ceil_div(np, 8));
        if (shallow) hipLaunchKernelGGL((nn_nabo<1, false, 12>), g1, dim3(kNnThreads), 0, st, d, kd, nb1);
        else hipLaunchKernelGGL((nn_nabo<1, false, kKdStack>), g1, dim3(kNnThreads), 0, st, d, kd, nb1);
      } else {
        const int nb4 = ceil_div(ns_max, kNnThreads * 4);
        const dim3 g4(nb4 * 8 * ceil_div(np, 8));
        if (shallow) hipLaunchKernelGGL((nn_nabo<4, false, 12>), g4, dim3(kNnThreads), 0, st, d, kd, nb4);
        else hipLaunchKernelGGL((nn_nabo<4, false, kKdStack>), g4, dim3(kNnThreads), 0, st, d, kd, nb4);
      }
    }
  } else if (h->opts.nn_mode == SMHIP_NN_GRID) {
    if (d.use_ball) {
      const int nblk = ceil_div(ns_max, kNnThreads * kBallItems);
      const dim3 gx(nblk * 8 * ceil_div(np, 8));
      const dim3 glist(kListedBlocks * 8 * ceil_div(np, 8));
      // the two-launch form pays from ~16 pairs per launch on (measured with the many-lanes-per-query listed search: equal for
      // one pair, +4 % at 16, +1 % at 32, +6 % at 2 x 32, +9 % at 256 pairs); an explicit split_after option is honoured
      // for any size
      const bool split_now = d.certify && iteration >= d.split_after && (h->opts.split_after > 0 || np >= 16);
      if (d.lds_table && !split_now) {
        // certificate, in-workgroup compaction of the failing queries and LDS-staged search in one launch
        Bracket br(h, 4, st, np);
        if (f.small) {
          const int nb1 = ceil_div(ns_max, kNnThreads);
          hipLaunchKernelGGL(nn_ball_lds<1>, dim3(nb1 * 8 * ceil_div(np, 8)), dim3(kNnThreads), 0, st, d, nb1);
        } else if (h->wave_search) {
          // a wave per 64 queries walks the box of its balls once, candidates broadcast from the wave's LDS strip (nn_ball_wave)
          if (iteration == 0) hipLaunchKernelGGL((nn_ball_wave<kBallItems, true>), gx, dim3(kNnThreads), 0, st, d, nblk);
          else hipLaunchKernelGGL((nn_ball_wave<kBallItems, false>), gx, dim3(kNnThreads), 0, st, d, nblk);
        } else if (iteration == 0) {
          hipLaunchKernelGGL((nn_ball_lds<kBallItems, true>), gx, dim3(kNnThreads), 0, st, d, nblk);
        } else {
          hipLaunchKernelGGL(nn_ball_lds<kBallItems>, gx, dim3(kNnThreads), 0, st, d, nblk);
        }
      } else if (d.certify && iteration > 0) {
        // global-memory variant: certificate pass, then a search over the compacted failing queries
        // (also what the converged iterations of the LDS variant use: a streaming certificate pass at full occupancy and a
        // near-empty listed search beat the fused kernel once only a handful of certificates fail)
        {
          Bracket br(h, d.lds_table ? 5 : 4, st, np);
          if (d.fused) {
            // certificate pass + the sums below the predicted quantile band in one pass over the source (fused_iteration decides)
            const int nbc = ceil_div(ns_max, kNnThreads * kCertifyItems);
            // (every target of the launch below 32 767 points: the 4-byte shadow of bound + match instead of the two arrays.  Only behind
            // the LDS-table ball search, whose kernels -- with the listed search and the refinement kernels -- write the shadow with every
            // match (st_match); nn_ring_wide, nn_brute and nn_nabo set idx / lb alone and never run in such an Align)
            if (h->use_shadow && d.lds_table && nt_max_of(h, d.pair_base, np) < 0x7fff)
              hipLaunchKernelGGL((nn_certify_acc<kCertifyItems, false, true>), dim3(nbc * 8 * ceil_div(np, 8)), dim3(kNnThreads), 0, st, d, nbc);
            else
              hipLaunchKernelGGL(nn_certify_acc<kCertifyItems>, dim3(nbc * 8 * ceil_div(np, 8)), dim3(kNnThreads), 0, st, d, nbc);
          } else if (f.small) {
            const int nb1 = ceil_div(ns_max, kNnThreads);
            hipLaunchKernelGGL(nn_certify<1>, dim3(nb1 * 8 * ceil_div(np, 8)), dim3(kNnThreads), 0, st, d, nb1);
          } else {
            const int nbc = ceil_div(ns_max, kNnThreads * kCertifyItems);
            hipLaunchKernelGGL(nn_certify<kCertifyItems>, dim3(nbc * 8 * ceil_div(np, 8)), dim3(kNnThreads), 0, st, d, nbc);
          }
        }
        {
          Bracket br(h, 6, st, np);
          if (d.fused) {
            // the lists' lengths differ by an order of magnitude between the pairs of a launch: cut into equal items first
            hipLaunchKernelGGL(listed_plan, dim3(np), dim3(256), 0, st, d);
            hipLaunchKernelGGL(nn_ball_listed_items, dim3(kListedItemBlocks), dim3(kNnThreads), 0, st, d);
          } else {
            hipLaunchKernelGGL(nn_ball_listed, glist, dim3(kNnThreads), 0, st, d, kListedBlocks);
          }
        }
      } else {
        Bracket br(h, 4, st, np);
        hipLaunchKernelGGL(nn_ball, gx, dim3(kNnThreads), 0, st, d, nblk);
      }
      if ((f.small || d.fused) && !d.exact_all) {
        // a few pairs: validate + ring + fallback as ONE launch, a workgroup per pair (near-empty launches cost ~5 us each there).
        // The same in a batch's fused iterations: the pose has settled there, a quantile that reaches a lower bound is the rare
        // case, and the two spread-out launches cost 20-25 us each of a ~700 us iteration whether they do anything or not
        // (8 192 and 16 384 workgroups that look at one flag).
        Bracket br(h, 1, st);
        hipLaunchKernelGGL(nn_refine_one, dim3(np), dim3(kNnThreads), 0, st, d);
        return SMHIP_OK;
      }
      { Bracket br(h, 1, st); hipLaunchKernelGGL(nn_validate, dim3(np), dim3(256), 0, st, d); }
      { Bracket br(h, 1, st); hipLaunchKernelGGL(nn_ring<true>, dim3(32, np), dim3(kNnThreads), 0, st, d); }
    } else if ((long long)np * ns_max < (1ll << 21)) {
      // few queries in the whole launch: several lanes per query keep the SIMDs busy
      Bracket br(h, 4, st, np);
      hipLaunchKernelGGL(nn_ring_coop, dim3(ceil_div(ns_max, kNnThreads / kCoopLanes), np), dim3(kNnThreads), 0, st, d);
      hipLaunchKernelGGL(nn_ring_wide, dim3(kWideBlocks, np), dim3(kNnThreads), 0, st, d);
    } else {
      Bracket br(h, 4, st, np);
      hipLaunchKernelGGL(nn_ring<false>, g, dim3(kNnThreads), 0, st, d);
    }
    { Bracket br(h, 1, st); hipLaunchKernelGGL(nn_fallback, dim3(kFallbackSlices, np), dim3(kNnThreads), 0, st, d); }
  } else {
    Bracket br(h, 4, st, np);
    hipLaunchKernelGGL(nn_brute, g, dim3(kNnThreads), 0, st, d);
  }
  return SMHIP_OK;
}

// Does iteration `iteration` of this batch part run the fused path (nn_certify_acc + nn_ball_listed_items: certificate pass and
// normal-equation sums in one pass over the source)?  Exactly where the two-launch certificate form runs in a batch, unless every
// bound is refined in every iteration anyway (nothing to speculate on) or the cloud has more record segments than finalize indexes.
bool fused_iteration(const smhip_context* h, const Half& f, int ns_max, int iteration) {
  const IcpDev& d = f.d;
  if (h->opts.nn_mode == SMHIP_NN_NABO) {
    // the reference-search form: every certificate iteration of a batch (the walk has no bounds to refine); rows of partials for
    // the certificate pass's workgroups + accumulate_listed's
    // ... from the iteration on in which the previous batch's median pair walked fewer than a fifth of its queries again: the walked
    // queries are summed from their lists (accumulate_listed: scattered reads, ten times a streamed point's bytes), which only
    // pays once they are few; before that `accumulate` streams every point
    if (!d.certify || h->opts.no_fused_sums || f.small || iteration < std::max(1, h->nabo_fused_from) || f.np < 16) return false;
    const int nbc = ceil_div(ns_max, kNnThreads * kCertifyItems);
    return nbc + kNaboAccBlocks <= d.part_stride && nbc * (kNnThreads / 64) + kNaboAccBlocks * (kAccThreads / 64) <= std::min(kFinalizeMaxSeg, (int)d.seg_stride);
  }
  if (h->opts.nn_mode != SMHIP_NN_GRID || !d.use_ball || !d.lds_table || !d.certify || h->opts.no_fused_sums || d.exact_all || f.small) return false;
  if (iteration < 1 || iteration < d.split_after || !(h->opts.split_after > 0 || f.np >= 16)) return false;
  if (f.np > kListedMaxPairs) return false;
  // finalize's segment table: the certificate pass's waves + the listed search's items of a list nn_validate accepts
  return ceil_div(ns_max, kNnThreads * kCertifyItems) * (kNnThreads / 64) + kListedMaxItems <= std::min(kFinalizeMaxSeg, (int)d.seg_stride);
}

smhip_status fill_inputs(smhip_context* h, int np, const double* guesses, int* ns_max, int* nt_max, int first = 0) {
  *ns_max = 0; *nt_max = 0;
  for (int p = first; p < first + np; ++p) {
    if (h->ns[p] <= 0 || h->nt[p] <= 0) { h->err = "Align before SetInputSource/SetInputTarget"; return SMHIP_ERR_NOT_READY; }
    if (!h->has_normals[p]) { h->err = "IcpFast target has no normals (icp_fast.cc:430)"; return SMHIP_ERR_NO_NORMALS; }
    PairInput& in = h->in_pinned[p];
    const double* g = guesses + 16 * (p - first);
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) in.guess[4 * r + c] = g[4 * c + r];   // column-major in
    in.ns = h->ns[p]; in.nt = h->nt[p]; in.has_normals = 1; in.pad = 0;
    *ns_max = std::max(*ns_max, h->ns[p]);
    *nt_max = std::max(*nt_max, h->nt[p]);
  }
  return SMHIP_OK;
}

void sync_options(smhip_context* h) {
  h->dev.max_iteration = std::max(1, h->opts.max_iteration);
  h->dev.early_exit = h->opts.early_exit;
  h->dev.max_ring = std::max(1, h->opts.grid_max_ring);
  h->dev.rho = h->opts.dist_outlier_ratio;
  h->dev.grid_cell = h->opts.grid_cell > 0 ? h->opts.grid_cell : 0.5f;
  h->dev.use_ball = h->opts.use_ball;
  h->dev.sort_cells = 1;
  h->dev.certify = h->opts.no_certify ? 0 : 1;
  h->dev.split_after = h->opts.split_after > 0 ? h->opts.split_after : (h->opts.split_after < 0 ? 1 << 30 : 2);
  h->dev.lds_table = h->opts.no_lds_table ? 0 : 1;
  h->dev.cap_factor = h->opts.ball_cap_factor > 1.0f ? h->opts.ball_cap_factor : 1.5f;
  h->dev.exact_all = h->opts.exact_matches;
  h->dev.ball_radius = h->opts.ball_radius > 0 ? h->opts.ball_radius : 0.3f;
  // (the ball search walks a query's rows through a 32-bit mask: at most 14 cells of radius, 3.5 m with the default cell -- a larger
  // radius only moves the point from which the ring search takes over, never the result)
  h->dev.ball_radius = std::min(h->dev.ball_radius, 14.0f * h->dev.grid_cell);
  { const char* e = std::getenv("SMHIP_DEBUG_FLAGS"); h->dev.debug_flags = e ? std::atoi(e) : 0; }
  h->dev.fused = 0; h->dev.fused_nabo = 0;
  h->dev.band_pad = 0.1f; h->dev.band_gain = 1.5f;              // tuning only: results do not depend on the band, only how often it holds
  { const char* e = std::getenv("SMHIP_BAND_PAD"); if (e && std::atof(e) >= 0.0) h->dev.band_pad = (float)std::atof(e); }
  // lanes per query of the balanced listed search: round 4 measured it flat from 1 024 to 8 192 (3.13-3.25 ms per step), slower beyond
  // (16 384: 3.5, 32 768: 4.2); with the level-by-level row walk of round 5: 512 / 1 024 / 2 048 / 4 096 = +2.6 / +3.1 / +0.6 / 0 % alignments/s
  // on the bench batch together with 2 048 sums workgroups (one lane walks a short ball's few rows at no loss now).  At most kListedLaneBudgetMax: a pair's list is cut into at most budget / 256 + 1 items
  // (kListedMaxItems: finalize's segment table and the handle's seg_stride are sized for that)
  h->dev.listed_lane_budget = 1024;
  { const char* e = std::getenv("SMHIP_LISTED_LANES"); if (e && std::atoi(e) >= 256) h->dev.listed_lane_budget = std::min(std::atoi(e), kListedLaneBudgetMax); }
  h->split_share = 0.2f;
  { const char* e = std::getenv("SMHIP_SPLIT_SHARE"); if (e && std::atof(e) > 0.0) h->split_share = (float)std::atof(e); }
  h->sums_long_for = 3;
  { const char* e = std::getenv("SMHIP_SUMS_LONG_FOR"); if (e) h->sums_long_for = std::atoi(e); }
  h->use_shadow = 1;
  { const char* e = std::getenv("SMHIP_SHADOW"); if (e) h->use_shadow = std::atoi(e); }
  h->wave_search = 0;
  { const char* e = std::getenv("SMHIP_WAVE_SEARCH"); if (e) h->wave_search = std::atoi(e); }
  h->one_enabled = 1; h->one_blocks_want = 0;
  { const char* e = std::getenv("SMHIP_ONE_PAIR"); if (e) h->one_enabled = std::atoi(e); }
  { const char* e = std::getenv("SMHIP_ONE_BLOCKS"); if (e) h->one_blocks_want = std::atoi(e); }
  h->one_groups_want = 0;
  { const char* e = std::getenv("SMHIP_ONE_GROUPS"); if (e) h->one_groups_want = std::atoi(e); }
  h->sums_blocks = kSumsBlocks;
  { const char* e = std::getenv("SMHIP_SUMS_BLOCKS"); if (e && std::atoi(e) >= 8) h->sums_blocks = (std::min(std::atoi(e), 65536) / 8) * 8; }
  h->dev.listed_grain = 1;
  { const char* e = std::getenv("SMHIP_LISTED_GRAIN"); if (e && std::atoi(e) >= 0) h->dev.listed_grain = std::atoi(e); }
  { const char* e = std::getenv("SMHIP_BAND_GAIN"); if (e && std::atof(e) >= 0.0) h->dev.band_gain = (float)std::atof(e); }
  { const char* e = std::getenv("SMHIP_NABO_LISTED_BLOCKS"); if (e && std::atoi(e) > 0) h->nabo_listed_blocks = std::max(8, std::min(4096, std::atoi(e))); }   // >= 8: a workgroup's 16-bit histogram bins
}

// FindClosests output in the caller's order: source i was uploaded from caller index src.w,
// target position j holds caller index tq.w.
__global__ void export_matches(IcpDev b, int pair, int32_t* out_ids, float* out_d2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.state[pair].ns) return;
  const int orig = __float_as_int(b.src[(size_t)pair * b.ns_cap + i].w);
  const int j = b.idx[(size_t)pair * b.ns_cap + i];
  out_ids[orig] = j < 0 ? -1 : __float_as_int(b.tq[(size_t)pair * b.nt_cap + j].w);
  out_d2[orig] = b.d2[(size_t)pair * b.ns_cap + i];
}

smhip_status fetch_matches(smhip_context* h, int slot, int32_t* ids, float* d2, int n) {
  if (n != h->ns[slot]) { h->err = "n must equal the slot's source size"; return SMHIP_ERR_INVALID_ARGUMENT; }
  hipLaunchKernelGGL(export_matches, dim3(ceil_div(n, 256)), dim3(256), 0, h->stream, h->dev, slot, h->ids_dev, h->d2_dev);
  HIPCHK(h, hipMemcpyAsync(h->ids_pinned, h->ids_dev, sizeof(int32_t) * n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d2_pinned, h->d2_dev, sizeof(float) * n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (ids) std::memcpy(ids, h->ids_pinned, sizeof(int32_t) * n);
  if (d2) std::memcpy(d2, h->d2_pinned, sizeof(float) * n);
  return SMHIP_OK;
}

}  // namespace

extern "C" {

int smhip_version(void) { return 100; }

const char* smhip_status_string(smhip_status s) {
  switch (s) {
    case SMHIP_OK: return "ok";
    case SMHIP_ERR_INVALID_ARGUMENT: return "invalid argument";
    case SMHIP_ERR_NO_DEVICE: return "no gfx950 device";
    case SMHIP_ERR_HIP: return "HIP runtime error";
    case SMHIP_ERR_NOT_READY: return "source/target not set";
    case SMHIP_ERR_NO_NORMALS: return "target has no normals";
    case SMHIP_ERR_NO_MATCH: return "no finite correspondence";
    case SMHIP_ERR_CAPACITY: return "cloud exceeds handle capacity";
    default: return "unknown";
  }
}

int smhip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return -1;
  int good = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, i) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ++good;
  }
  return good;
}

void smhip_icp_default_options(smhip_icp_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->max_iteration = 100;          // icp_fast.h:58
  o->dist_outlier_ratio = 0.7f;    // icp_fast.h:59
  o->early_exit = 1;
  o->nn_mode = SMHIP_NN_GRID;
  o->grid_cell = 0.25f;
  o->grid_max_ring = 8;
  o->check_every = 8;
  o->use_ball = 1;
  o->exact_matches = 0;
  o->ball_radius = 0.3f;
  o->ball_cap_factor = 1.5f;
  o->no_certify = 0;
  o->nn_epsilon = 3.16f;           // icp_fast.cc:174 (used by SMHIP_NN_NABO only)
}

smhip_status smhip_create(int device, void* stream, int pair_slots, int max_source_points, int max_target_points,
                          smhip_handle* out) {
  if (!out || pair_slots < 1 || max_source_points < 1 || max_target_points < 1) return SMHIP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  // finalize indexes at most kFinalizeMaxSeg accumulate waves per pair: 4 Mi source points per cloud
  if (max_source_points > kFinalizeMaxSeg * 64 * kAccItemsBatch) return SMHIP_ERR_INVALID_ARGUMENT;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SMHIP_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SMHIP_ERR_NO_DEVICE;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SMHIP_ERR_NO_DEVICE;   // gfx950-only code objects
  if (hipSetDevice(device) != hipSuccess) return SMHIP_ERR_NO_DEVICE;
  smhip_context* h = new smhip_context();
  h->device = device;
  if (stream) { h->stream = reinterpret_cast<hipStream_t>(stream); h->own_stream = false; }
  else {
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return SMHIP_ERR_HIP; }
    h->own_stream = true;
  }
  if (hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess) h->ev_fork = nullptr;
  // side streams are created when a batch first asks for them (ensure_side_streams): the runtime maps streams onto a
  // handful of hardware queues (GPU_MAX_HW_QUEUES, default 4), so idle streams are not free
  smhip_icp_default_options(&h->opts);
  IcpDev& d = h->dev;
  d.slots = pair_slots; d.ns_cap = max_source_points; d.nt_cap = max_target_points;
  d.acc_blocks = ceil_div(max_source_points, kAccThreads * kAccItemsSmall);
  d.acc_items = kAccItemsSmall;
  // rows of partials: accumulate's workgroups, or the fused certificate pass's plus accumulate_listed's (reference-search mode)
  d.part_stride = std::max(d.acc_blocks, ceil_div(max_source_points, kNnThreads * kCertifyItems) + std::max(kNaboAccBlocks, kFusedListedMax / kListedSumChunk));
  d.dl_stride = ceil_div(max_source_points, kNnThreads * kCertifyItems) * (kNnThreads * kCertifyItems);
  d.bl_stride = std::max(ceil_div(max_source_points, kAccThreads * kAccItemsBatch) * (kAccThreads * kAccItemsBatch), d.dl_stride);
  // one segment per producing wave: accumulate with short chunks makes the most; the fused path has its certificate pass's waves
  // plus the listed search's
  d.seg_stride = std::max(d.acc_blocks * (kAccThreads / 64),
                          ceil_div(max_source_points, kNnThreads * kCertifyItems) * (kNnThreads / 64) + std::max(kListedBlocks * (kNnThreads / 64), kListedMaxItems) + 1);
  const size_t B = pair_slots, NS = max_source_points, NT = max_target_points;
  smhip_status s = SMHIP_OK;
  auto A = [&](smhip_status r) { if (s == SMHIP_OK) s = r; };
  A(dev_alloc(h, &d.state, B));
  A(dev_alloc(h, const_cast<PairInput**>(&d.in), B));
  A(dev_alloc(h, const_cast<float4**>(&d.src), B * NS));
  A(dev_alloc(h, const_cast<float**>(&d.src3), B * NS * 3));
  A(dev_alloc(h, const_cast<float4**>(&d.tgt_p), B * NT));
  A(dev_alloc(h, const_cast<float4**>(&d.tgt_n), B * NT));
  A(dev_alloc(h, &d.tq, B * NT));
  A(dev_alloc(h, &d.tn, B * NT));
  A(dev_alloc(h, &d.tcell, B * NT));
  A(dev_alloc(h, &d.tslot, B * NT));
  A(dev_alloc(h, &d.tord, B * NT));
  A(dev_alloc(h, &d.bits, B * kMaxGridWords));
  A(dev_alloc(h, &d.words, B * kMaxGridWords));
  A(dev_alloc(h, &d.rowbits, B * (size_t)kMaxRowWords));
  A(dev_alloc(h, &d.ccount, B * (NT + 1)));
  A(dev_alloc(h, &d.cstart, B * (NT + 1)));
  A(dev_alloc(h, &d.d2, B * NS));
  A(dev_alloc(h, &d.lb, B * NS));
  A(dev_alloc(h, &d.search_hist, B * kSearchHist));
  A(dev_alloc(h, &d.idx, B * NS));
  A(dev_alloc(h, &d.mb, B * NS));
  A(dev_alloc(h, &d.hist, B * kHistBins));
  A(dev_alloc(h, &d.dlist, B * (size_t)d.dl_stride));
  A(dev_alloc(h, &d.hlist, B * NS));
  A(dev_alloc(h, &d.ulist, B * NS));
  A(dev_alloc(h, &d.ukeys, B * NS));
  A(dev_alloc(h, &d.rec_a, B * 2 * (size_t)d.bl_stride));
  A(dev_alloc(h, &d.rec_j, B * 2 * (size_t)d.bl_stride));
  A(dev_alloc(h, &d.gcount, B * (size_t)d.seg_stride));
  A(dev_alloc(h, &d.dcount, B * (size_t)d.seg_stride));
  A(dev_alloc(h, &d.litems, B));
  A(dev_alloc(h, &d.partials, B * (size_t)d.part_stride * kAccCols));
  A(dev_alloc(h, &d.tpart, B * kTgtReduceBlocks * 16));
  A(dev_alloc(h, &d.done_count, 4));
  {   // everything of the single-pair cooperative launch that crosses workgroups: ONE fine-grained allocation (see IcpDev::one_ctr)
    const size_t w_sync = (size_t)kOneSyncWords * kOnePairs * 4, w_rows = (size_t)kOnePairs * (kOneMaxBlocks + 32) * kAccCols * 8,
                 w_hist = (size_t)kHistBins * kOnePairs * 4, w_keys = (size_t)kOnePairs * 2 * (size_t)d.bl_stride * 4, w_ctr = sizeof(PairState) * kOnePairs;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t total = up(w_sync) + up(w_rows) + up(w_hist) + up(w_keys) + up(w_ctr);
    void* v = nullptr;
    if (s == SMHIP_OK) {
      if (hipExtMallocWithFlags(&v, total, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        v = nullptr;
        if (hipMalloc(&v, total) != hipSuccess) { s = SMHIP_ERR_HIP; v = nullptr; }
        h->one_blocks_allowed = 0;           // (no fine-grained memory: single pairs keep to the separate launches)
      }
      if (v) {
        h->allocs.push_back(v);
        char* c = reinterpret_cast<char*>(v);
        d.one_sync = reinterpret_cast<uint32_t*>(c); c += up(w_sync);
        d.one_rows = reinterpret_cast<double*>(c); c += up(w_rows);
        d.one_hist = reinterpret_cast<uint32_t*>(c); c += up(w_hist);
        d.one_keys = reinterpret_cast<uint32_t*>(c); c += up(w_keys);
        d.one_ctr = reinterpret_cast<PairState*>(c);
        if (hipMemsetAsync(v, 0, total, h->stream) != hipSuccess) s = SMHIP_ERR_HIP;
      }
    }
  }
  A(dev_alloc(h, &h->ids_dev, NS));
  A(dev_alloc(h, &h->d2_dev, NS));
  if (s == SMHIP_OK) {
    const size_t stage_n = 2 * std::max(NS, NT);
    if (hipHostMalloc(reinterpret_cast<void**>(&h->stage), stage_n * sizeof(float4)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&h->in_pinned), B * sizeof(PairInput)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&h->state_pinned), B * sizeof(PairState)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&h->done_pinned), 64) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&h->hist_pinned), B * kSearchHist * sizeof(uint32_t)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&h->ids_pinned), NS * sizeof(int32_t)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&h->d2_pinned), NS * sizeof(float)) != hipSuccess)
      s = SMHIP_ERR_HIP;
  }
  if (s == SMHIP_OK) {
    // the single-pair persistent kernel meets at grid barriers: its grid is what the device holds at once (cooperative launch)
    int per_cu = 0, coop = 0;
    if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, device) == hipSuccess && coop &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, icp_one, kNnThreads, 0) == hipSuccess && per_cu > 0)
      h->one_blocks = (std::min(per_cu * prop.multiProcessorCount, kOneMaxBlocks) / 32) * 32;
    (void)hipGetLastError();
    if (!h->one_blocks_allowed) h->one_blocks = 0;
  }
  if (s == SMHIP_OK && hipMemsetAsync(d.state, 0, B * sizeof(PairState), h->stream) != hipSuccess) s = SMHIP_ERR_HIP;
  if (s == SMHIP_OK && hipStreamSynchronize(h->stream) != hipSuccess) s = SMHIP_ERR_HIP;
  if (s != SMHIP_OK) { smhip_destroy(h); return s; }
  h->ns.assign(B, 0); h->nt.assign(B, 0); h->has_normals.assign(B, 0);
  h->tgt_gen.assign(B, 0); h->grid_gen.assign(B, 0); h->src_gen.assign(B, 1); h->src3_gen.assign(B, 0); h->grid_cell_built.assign(B, 0.f); h->grid_sorted.assign(B, 0); h->grid_rows.assign(B, 0); h->grid_mode.assign(B, -1);
  sync_options(h);
  *out = h;
  return SMHIP_OK;
}

extern "C" void smhip_internal_free_ndt(smhip_context* h);
extern "C" void smhip_internal_free_gicp(smhip_context* h);

smhip_status smhip_destroy(smhip_handle h) {
  if (!h) return SMHIP_ERR_INVALID_ARGUMENT;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  smhip_internal_free_ndt(h);
  smhip_internal_free_gicp(h);
  if (h->prep) prep_destroy(h->prep);
  if (h->prep_batch) prep_destroy(h->prep_batch);
  if (h->filt) filt_destroy(h->filt);
  for (void* p : h->allocs) (void)hipFree(p);
  if (h->stage) (void)hipHostFree(h->stage);
  if (h->in_pinned) (void)hipHostFree(h->in_pinned);
  if (h->state_pinned) (void)hipHostFree(h->state_pinned);
  if (h->done_pinned) (void)hipHostFree(h->done_pinned);
  if (h->hist_pinned) (void)hipHostFree(h->hist_pinned);
  if (h->ids_pinned) (void)hipHostFree(h->ids_pinned);
  if (h->d2_pinned) (void)hipHostFree(h->d2_pinned);
  for (auto& e : h->ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (int k = 0; k < h->n_side; ++k) {
    (void)hipStreamSynchronize(h->side[k]); (void)hipStreamDestroy(h->side[k]); (void)hipEventDestroy(h->ev_join[k]);
  }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
  if (h->ev_copied) (void)hipEventDestroy(h->ev_copied);
  if (h->ev_raw_free) (void)hipEventDestroy(h->ev_raw_free);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return SMHIP_OK;
}

const char* smhip_last_error(smhip_handle h) { return h ? h->err.c_str() : "null handle"; }

smhip_status smhip_icp_set_options(smhip_handle h, const smhip_icp_options* o) {
  if (!h || !o) return SMHIP_ERR_INVALID_ARGUMENT;
  if (!(o->dist_outlier_ratio >= 0.f && o->dist_outlier_ratio <= 1.f)) {     // icp_fast.cc:68 CHECK
    h->err = "dist_outlier_ratio must be in [0, 1]";
    return SMHIP_ERR_INVALID_ARGUMENT;
  }
  if (o->nn_mode != SMHIP_NN_BRUTE && o->nn_mode != SMHIP_NN_GRID && o->nn_mode != SMHIP_NN_NABO) { h->err = "bad nn_mode"; return SMHIP_ERR_INVALID_ARGUMENT; }
  if (o->nn_mode == SMHIP_NN_NABO && !(o->nn_epsilon >= 0.f)) { h->err = "nn_epsilon must be >= 0"; return SMHIP_ERR_INVALID_ARGUMENT; }
  h->opts = *o;
  if (h->opts.check_every < 1) h->opts.check_every = 8;
  sync_options(h);
  if (o->nn_mode == SMHIP_NN_NABO) {      // the mode's arrays (tree, work classes) are allocated here, never inside Align
    HIPCHK(h, hipSetDevice(h->device));
    return kd_ensure(h);
  }
  return SMHIP_OK;
}

smhip_status smhip_synchronize(smhip_handle h) {
  if (!h) return SMHIP_ERR_INVALID_ARGUMENT;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return SMHIP_OK;
}

// ---- uploads ---------------------------------------------------------------------------
static smhip_status prep_ensure(smhip_handle h);

// staged source -> device, Morton-ordered there (spatially coherent wavefronts; w = caller index)
static smhip_status upload_source(smhip_handle h, int slot, int n) {
  smhip_status s = prep_ensure(h);
  if (s) return s;
  HIPCHK(h, hipMemcpyAsync(h->prep_raw, h->stage, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, h->stream));
  const hipError_t e = prep_morton_sort(h->prep, h->stream, h->prep_raw, n, const_cast<float4*>(h->dev.src) + (size_t)slot * h->dev.ns_cap);
  if (e != hipSuccess) { h->err = std::string("prep_morton_sort: ") + hipGetErrorString(e); return SMHIP_ERR_HIP; }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->ns[slot] = n;
  touch_source(h, slot);
  return SMHIP_OK;
}

static smhip_status upload(smhip_handle h, const float4* dst_dev, const float4* staged, int n) {
  HIPCHK(h, hipMemcpyAsync(const_cast<float4*>(dst_dev), staged, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, h->stream));
  return SMHIP_OK;
}

smhip_status smhip_set_source_f64(smhip_handle h, int slot, const double* xyz, int n) {
  smhip_status s = check_slot(h, slot);
  if (s) return s;
  if (!xyz || n <= 0) { h->err = "empty source cloud"; return SMHIP_ERR_INVALID_ARGUMENT; }
  if (n > h->dev.ns_cap) { h->err = "source larger than max_source_points"; return SMHIP_ERR_CAPACITY; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));   // staging buffer reuse
  for (int i = 0; i < n; ++i) h->stage[i] = make_float4((float)xyz[3 * i], (float)xyz[3 * i + 1], (float)xyz[3 * i + 2], 0.f);
  return upload_source(h, slot, n);
}

smhip_status smhip_set_source_f32(smhip_handle h, int slot, const float* xyz, int stride, int n) {
  smhip_status s = check_slot(h, slot);
  if (s) return s;
  if (!xyz || n <= 0 || stride < 3) { h->err = "empty source cloud / bad stride"; return SMHIP_ERR_INVALID_ARGUMENT; }
  if (n > h->dev.ns_cap) { h->err = "source larger than max_source_points"; return SMHIP_ERR_CAPACITY; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (stride == 4) std::memcpy(h->stage, xyz, sizeof(float4) * (size_t)n);      // KITTI rows are already float4 (w is overwritten on the device)
  else for (int i = 0; i < n; ++i) h->stage[i] = make_float4(xyz[(size_t)stride * i], xyz[(size_t)stride * i + 1], xyz[(size_t)stride * i + 2], 0.f);
  return upload_source(h, slot, n);
}

smhip_status smhip_reserve_batch_workspaces(smhip_handle h) {
  if (!h) return SMHIP_ERR_INVALID_ARGUMENT;
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->prep_batch) {
    h->prep_batch = prep_create(h->dev.slots * h->dev.ns_cap);
    if (!h->prep_batch) { h->err = "batch workspace allocation failed"; return SMHIP_ERR_HIP; }
  }
  if (prep_reserve_forest(h->prep_batch) != hipSuccess) { h->err = "batch workspace allocation failed (CalculateNormals forest)"; return SMHIP_ERR_HIP; }
  if (!h->raw_batch) {
    smhip_status s = dev_alloc(h, &h->raw_batch, (size_t)h->dev.slots * h->dev.ns_cap);
    if (s) return s;
  }
  return SMHIP_OK;
}

smhip_status smhip_set_sources_f32_batch(smhip_handle h, int count, const int* slots, const float* const* rows, const int* n) {
  if (!h || !slots || !rows || !n || count < 1 || count > h->dev.slots || count > 512) { if (h) h->err = "bad batch of sources"; return SMHIP_ERR_INVALID_ARGUMENT; }
  long long total = 0;
  for (int k = 0; k < count; ++k) {
    smhip_status s = check_slot(h, slots[k]);
    if (s) return s;
    if (!rows[k] || n[k] <= 0) { h->err = "empty source cloud"; return SMHIP_ERR_INVALID_ARGUMENT; }
    if (n[k] > h->dev.ns_cap) { h->err = "source larger than max_source_points"; return SMHIP_ERR_CAPACITY; }
    total += n[k];
  }
  if (total > 0x7fffffffll) { h->err = "batch of sources holds more than 2^31 points (the staging offsets are 32-bit)"; return SMHIP_ERR_CAPACITY; }
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->prep_batch) {
    h->prep_batch = prep_create(h->dev.slots * h->dev.ns_cap);
    if (!h->prep_batch) { h->err = "batched upload workspace allocation failed"; return SMHIP_ERR_HIP; }
  }
  if (!h->raw_batch) {
    smhip_status s = dev_alloc(h, &h->raw_batch, (size_t)h->dev.slots * h->dev.ns_cap);
    if (s) return s;
  }
  // rows -> the device staging array, cloud after cloud.  Pinned rows are copied from where they lie (no host copy, the call
  // returns at once); pageable rows go through the handle's one pinned buffer, which must have left the host before it is
  // filled again.
  std::vector<int> stage_off(count);
  std::vector<long long> out_off(count);
  std::vector<char> is_pinned(count);
  bool all_pinned = true;
  for (int k = 0; k < count; ++k) {
    hipPointerAttribute_t attr{};
    is_pinned[k] = hipPointerGetAttributes(&attr, rows[k]) == hipSuccess && attr.type == hipMemoryTypeHost;
    if (!is_pinned[k]) { (void)hipGetLastError(); all_pinned = false; }
  }
  // All rows page-locked: the copies go to a stream of their own, so the DMA engines bring the next batch in while the
  // handle's stream is still aligning the previous one (a batch of 256 scans is 0.5 GB: 10 ms of a 50 ms batch).  The staging
  // array is free once the previous batch's Morton ordering has read it (ev_raw_free), and the ordering of this batch waits
  // for the copies (ev_copied).
  hipStream_t cs = h->stream;
  if (all_pinned) {
    if (!h->copy_stream) {
      if (hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&h->ev_copied, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&h->ev_raw_free, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); h->copy_stream = nullptr; }
    }
    if (h->copy_stream) {
      cs = h->copy_stream;
      if (h->raw_in_use) HIPCHK(h, hipStreamWaitEvent(cs, h->ev_raw_free, 0));
    }
  }
  long long at = 0;
  for (int k = 0; k < count; ++k) {
    stage_off[k] = (int)at;
    out_off[k] = (long long)slots[k] * h->dev.ns_cap;
    const void* from = rows[k];
    if (!is_pinned[k]) {
      HIPCHK(h, hipStreamSynchronize(h->stream));
      std::memcpy(h->stage, rows[k], sizeof(float4) * (size_t)n[k]);
      from = h->stage;
    }
    HIPCHK(h, hipMemcpyAsync(h->raw_batch + at, from, sizeof(float4) * (size_t)n[k], hipMemcpyHostToDevice, cs));
    at += n[k];
  }
  if (cs != h->stream) {
    HIPCHK(h, hipEventRecord(h->ev_copied, cs));
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_copied, 0));
  }
  const hipError_t e = prep_morton_sort_batch(h->prep_batch, h->stream, h->raw_batch, count, stage_off.data(), n, out_off.data(), const_cast<float4*>(h->dev.src));
  if (e != hipSuccess) { h->err = std::string("prep_morton_sort_batch: ") + hipGetErrorString(e); return SMHIP_ERR_HIP; }
  if (h->ev_raw_free) { HIPCHK(h, hipEventRecord(h->ev_raw_free, h->stream)); h->raw_in_use = true; }
  for (int k = 0; k < count; ++k) { h->ns[slots[k]] = n[k]; touch_source(h, slots[k]); }
  return SMHIP_OK;
}

smhip_status smhip_set_target_f64(smhip_handle h, int slot, const double* xyz, const double* nrm, int n) {
  smhip_status s = check_slot(h, slot);
  if (s) return s;
  if (!xyz || n <= 0) { h->err = "empty target cloud"; return SMHIP_ERR_INVALID_ARGUMENT; }
  if (n > h->dev.nt_cap) { h->err = "target larger than max_target_points"; return SMHIP_ERR_CAPACITY; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float4* sp = h->stage;
  float4* sn = h->stage + std::max(h->dev.ns_cap, h->dev.nt_cap);
  bool finite = true;
  for (int i = 0; i < n; ++i) {
    sp[i] = make_float4((float)xyz[3 * i], (float)xyz[3 * i + 1], (float)xyz[3 * i + 2], 0.f);
    sn[i] = nrm ? make_float4((float)nrm[3 * i], (float)nrm[3 * i + 1], (float)nrm[3 * i + 2], 0.f) : make_float4(0, 0, 0, 0);
    finite = finite && std::isfinite(sp[i].x) && std::isfinite(sp[i].y) && std::isfinite(sp[i].z);
  }
  if (!finite) { h->err = "target cloud has NaN / Inf coordinates"; return SMHIP_ERR_INVALID_ARGUMENT; }
  s = upload(h, h->dev.tgt_p + (size_t)slot * h->dev.nt_cap, sp, n);
  if (s) return s;
  s = upload(h, h->dev.tgt_n + (size_t)slot * h->dev.nt_cap, sn, n);
  if (s) return s;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->nt[slot] = n;
  h->has_normals[slot] = nrm != nullptr;
  touch_target(h, slot);
  return SMHIP_OK;
}

smhip_status smhip_set_target_f32(smhip_handle h, int slot, const float* xyz, int stride, const float* nrm, int nstride, int n) {
  smhip_status s = check_slot(h, slot);
  if (s) return s;
  if (!xyz || n <= 0 || stride < 3 || (nrm && nstride < 3)) { h->err = "empty target cloud / bad stride"; return SMHIP_ERR_INVALID_ARGUMENT; }
  if (n > h->dev.nt_cap) { h->err = "target larger than max_target_points"; return SMHIP_ERR_CAPACITY; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float4* sp = h->stage;
  float4* sn = h->stage + std::max(h->dev.ns_cap, h->dev.nt_cap);
  bool finite = true;
  for (int i = 0; i < n; ++i) {
    sp[i] = make_float4(xyz[(size_t)stride * i], xyz[(size_t)stride * i + 1], xyz[(size_t)stride * i + 2], 0.f);
    sn[i] = nrm ? make_float4(nrm[(size_t)nstride * i], nrm[(size_t)nstride * i + 1], nrm[(size_t)nstride * i + 2], 0.f) : make_float4(0, 0, 0, 0);
    finite = finite && std::isfinite(sp[i].x) && std::isfinite(sp[i].y) && std::isfinite(sp[i].z);
  }
  if (!finite) { h->err = "target cloud has NaN / Inf coordinates"; return SMHIP_ERR_INVALID_ARGUMENT; }
  s = upload(h, h->dev.tgt_p + (size_t)slot * h->dev.nt_cap, sp, n);
  if (s) return s;
  s = upload(h, h->dev.tgt_n + (size_t)slot * h->dev.nt_cap, sn, n);
  if (s) return s;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->nt[slot] = n;
  h->has_normals[slot] = nrm != nullptr;
  touch_target(h, slot);
  return SMHIP_OK;
}

static smhip_status prep_ensure(smhip_handle h) {
  if (h->prep) return SMHIP_OK;
  const int cap = std::max(h->dev.ns_cap, h->dev.nt_cap);
  h->prep = prep_create(cap);
  if (!h->prep) { h->err = "device CalculateNormals workspace allocation failed"; return SMHIP_ERR_HIP; }
  return dev_alloc(h, &h->prep_raw, (size_t)cap);
}

static smhip_status prep_run(smhip_handle h, const float4* raw_dev, int n, int slot, int* n_out) {
  // the leaf count is only known afterwards: run into the scratch halves of the staging-sized device buffer
  // when the slot's arrays could overflow, i.e. require nt_cap >= n / 4 + 8 (every leaf holds >= 4 points)
  if (h->dev.nt_cap < n / 4 + 8) { h->err = "max_target_points too small for the prepared target (need n / 4 + 8)"; return SMHIP_ERR_CAPACITY; }
  int m = 0;
  const hipError_t e = prep_calculate_normals(h->prep, h->stream, raw_dev, n,
                                              const_cast<float4*>(h->dev.tgt_p) + (size_t)slot * h->dev.nt_cap,
                                              const_cast<float4*>(h->dev.tgt_n) + (size_t)slot * h->dev.nt_cap, &m);
  // from here on the slot's target arrays have been written: whatever target it held is gone, also on the error paths
  touch_target(h, slot);
  h->nt[slot] = 0; h->has_normals[slot] = 0;
  if (e != hipSuccess) { h->err = std::string("prep_calculate_normals: ") + hipGetErrorString(e); return SMHIP_ERR_HIP; }
  if (m <= 0) { h->err = "CalculateNormals produced no target points"; return SMHIP_ERR_INVALID_ARGUMENT; }
  h->nt[slot] = m;
  h->has_normals[slot] = 1;
  touch_target(h, slot);
  if (n_out) *n_out = m;
  return SMHIP_OK;
}

smhip_status smhip_prepare_target_f32(smhip_handle h, int slot, const float* xyz, int stride, int n, int* n_out) {
  smhip_status s = check_slot(h, slot);
  if (s) return s;
  if (!xyz || n <= 0 || stride < 3) { h->err = "empty cloud / bad stride"; return SMHIP_ERR_INVALID_ARGUMENT; }
  if (n > std::max(h->dev.ns_cap, h->dev.nt_cap)) { h->err = "scan larger than the handle's capacity"; return SMHIP_ERR_CAPACITY; }
  HIPCHK(h, hipSetDevice(h->device));
  s = prep_ensure(h);
  if (s) return s;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (stride == 4) std::memcpy(h->stage, xyz, sizeof(float4) * (size_t)n);
  else for (int i = 0; i < n; ++i) h->stage[i] = make_float4(xyz[(size_t)stride * i], xyz[(size_t)stride * i + 1], xyz[(size_t)stride * i + 2], 0.f);
  HIPCHK(h, hipMemcpyAsync(h->prep_raw, h->stage, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, h->stream));
  return prep_run(h, h->prep_raw, n, slot, n_out);
}

smhip_status smhip_prepare_target_from_source(smhip_handle h, int from, int to, int* n_out) {
  smhip_status s = check_slot(h, from);
  if (s) return s;
  s = check_slot(h, to);
  if (s) return s;
  if (h->ns[from] <= 0) { h->err = "source slot is empty"; return SMHIP_ERR_NOT_READY; }
  HIPCHK(h, hipSetDevice(h->device));
  s = prep_ensure(h);
  if (s) return s;
  return prep_run(h, h->dev.src + (size_t)from * h->dev.ns_cap, h->ns[from], to, n_out);
}

smhip_status smhip_prepare_targets_from_sources(smhip_handle h, int count, const int* from_slots, const int* to_slots, int* n_out) {
  if (!h || !from_slots || !to_slots || count < 1 || count > h->dev.slots) return SMHIP_ERR_INVALID_ARGUMENT;
  HIPCHK(h, hipSetDevice(h->device));
  std::vector<int> off(count), n(count), out_off(count), m(count, 0);
  for (int k = 0; k < count; ++k) {
    smhip_status s = check_slot(h, from_slots[k]);
    if (s) return s;
    s = check_slot(h, to_slots[k]);
    if (s) return s;
    if (h->ns[from_slots[k]] <= 0) { h->err = "source slot is empty"; return SMHIP_ERR_NOT_READY; }
    n[k] = h->ns[from_slots[k]];
    if (h->dev.nt_cap < n[k] / 4 + 8) { h->err = "max_target_points too small for the prepared target (need n / 4 + 8)"; return SMHIP_ERR_CAPACITY; }
    off[k] = from_slots[k] * h->dev.ns_cap;
    out_off[k] = to_slots[k] * h->dev.nt_cap;
  }
  if (!h->prep_batch) {
    h->prep_batch = prep_create(h->dev.slots * h->dev.ns_cap);
    if (!h->prep_batch) { h->err = "batched CalculateNormals workspace allocation failed"; return SMHIP_ERR_HIP; }
  }
  const hipError_t e = prep_calculate_normals_batch(h->prep_batch, h->stream, h->dev.src, count, off.data(), n.data(), out_off.data(),
                                                    const_cast<float4*>(h->dev.tgt_p), const_cast<float4*>(h->dev.tgt_n), m.data());
  for (int k = 0; k < count; ++k) { touch_target(h, to_slots[k]); h->nt[to_slots[k]] = 0; h->has_normals[to_slots[k]] = 0; }   // overwritten, whatever follows
  if (e != hipSuccess) { h->err = std::string("prep_calculate_normals_batch: ") + hipGetErrorString(e); return SMHIP_ERR_HIP; }
  for (int k = 0; k < count; ++k) {
    if (m[k] <= 0) { h->err = "CalculateNormals produced no target points"; return SMHIP_ERR_INVALID_ARGUMENT; }
    h->nt[to_slots[k]] = m[k];
    h->has_normals[to_slots[k]] = 1;
    touch_target(h, to_slots[k]);
    if (n_out) n_out[k] = m[k];
  }
  return SMHIP_OK;
}

smhip_status smhip_prepare_target_from_target(smhip_handle h, int from, int to, int* n_out) {
  smhip_status s = check_slot(h, from);
  if (s) return s;
  s = check_slot(h, to);
  if (s) return s;
  if (from == to) { h->err = "from_slot == to_slot: the raw target would be overwritten while it is read"; return SMHIP_ERR_INVALID_ARGUMENT; }
  if (h->nt[from] <= 0) { h->err = "target slot is empty"; return SMHIP_ERR_NOT_READY; }
  HIPCHK(h, hipSetDevice(h->device));
  s = prep_ensure(h);
  if (s) return s;
  return prep_run(h, h->dev.tgt_p + (size_t)from * h->dev.nt_cap, h->nt[from], to, n_out);
}

smhip_status smhip_sample_source(smhip_handle h, int from, int to, float prob, uint32_t seed, int* n_out) {
  smhip_status s = check_slot(h, from);
  if (s) return s;
  s = check_slot(h, to);
  if (s) return s;
  if (from == to) { h->err = "from_slot == to_slot"; return SMHIP_ERR_INVALID_ARGUMENT; }
  if (h->ns[from] <= 0) { h->err = "source slot is empty"; return SMHIP_ERR_NOT_READY; }
  if (!(prob > 0.f)) { h->err = "sampling probability must be > 0"; return SMHIP_ERR_INVALID_ARGUMENT; }
  HIPCHK(h, hipSetDevice(h->device));
  s = prep_ensure(h);
  if (s) return s;
  int m = 0;
  const hipError_t e = prep_sample_morton(h->prep, h->stream, h->dev.src + (size_t)from * h->dev.ns_cap, h->ns[from], prob, seed,
                                          const_cast<float4*>(h->dev.src) + (size_t)to * h->dev.ns_cap, &m);
  if (e != hipSuccess) { h->err = std::string("prep_sample_morton: ") + hipGetErrorString(e); return SMHIP_ERR_HIP; }
  if (m <= 0) { h->err = "sampling kept no point"; return SMHIP_ERR_INVALID_ARGUMENT; }
  h->ns[to] = m;
  touch_source(h, to);
  if (n_out) *n_out = m;
  return SMHIP_OK;
}

smhip_status smhip_get_target_f32(smhip_handle h, int slot, float* xyz, float* normals, int n) {
  smhip_status s = check_slot(h, slot);
  if (s) return s;
  if (n != h->nt[slot] || n <= 0) { h->err = "n must equal the slot's target size"; return SMHIP_ERR_INVALID_ARGUMENT; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float4* sp = h->stage;
  float4* sn = h->stage + std::max(h->dev.ns_cap, h->dev.nt_cap);
  HIPCHK(h, hipMemcpyAsync(sp, h->dev.tgt_p + (size_t)slot * h->dev.nt_cap, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(sn, h->dev.tgt_n + (size_t)slot * h->dev.nt_cap, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (int i = 0; i < n; ++i) {
    if (xyz) { xyz[3 * i] = sp[i].x; xyz[3 * i + 1] = sp[i].y; xyz[3 * i + 2] = sp[i].z; }
    if (normals) { normals[3 * i] = sn[i].x; normals[3 * i + 1] = sn[i].y; normals[3 * i + 2] = sn[i].z; }
  }
  return SMHIP_OK;
}

smhip_status smhip_copy_slot(smhip_handle h, int from, int to) {
  smhip_status s = check_slot(h, from);
  if (s) return s;
  s = check_slot(h, to);
  if (s) return s;
  if (from == to) return SMHIP_OK;
  HIPCHK(h, hipSetDevice(h->device));
  IcpDev& d = h->dev;
  hipLaunchKernelGGL(copy_slot, dim3(ceil_div(std::max(d.ns_cap, d.nt_cap), 256)), dim3(256), 0, h->stream, d,
                     const_cast<float4*>(d.src), const_cast<float4*>(d.tgt_p), const_cast<float4*>(d.tgt_n), from, to);
  HIPCHK(h, hipGetLastError());
  h->ns[to] = h->ns[from]; h->nt[to] = h->nt[from]; h->has_normals[to] = h->has_normals[from];
  touch_source(h, to);
  touch_target(h, to);
  return SMHIP_OK;
}

// fewer side streams than asked for = less overlap, still correct
static void ensure_side_streams(smhip_context* h, int n) {
  if (!h->ev_fork) return;
  for (int k = h->n_side; k < n && k < smhip_context::kMaxParts - 1; ++k) {
    if (hipStreamCreateWithFlags(&h->side[k], hipStreamNonBlocking) != hipSuccess) { h->side[k] = nullptr; return; }
    if (hipEventCreateWithFlags(&h->ev_join[k], hipEventDisableTiming) != hipSuccess) { (void)hipStreamDestroy(h->side[k]); h->side[k] = nullptr; return; }
    h->n_side = k + 1;
  }
}

// ---- Align -------------------------------------------------------------------------------
static smhip_status enqueue_range(smhip_handle h, int first, int npairs, const double* guesses) {
  if (!h || !guesses || npairs < 1 || first < 0 || first + npairs > h->dev.slots) {
    if (h) h->err = "bad slot range / guesses";
    return SMHIP_ERR_INVALID_ARGUMENT;
  }
  HIPCHK(h, hipSetDevice(h->device));
  int ns_max = 0, nt_max = 0;
  // in_pinned may still be in flight from a previous enqueue on this stream
  HIPCHK(h, hipStreamSynchronize(h->stream));
  smhip_status s = fill_inputs(h, npairs, guesses, &ns_max, &nt_max, first);
  if (s) return s;
  if (h->hist_pairs > 0) {
    // the previous batch is complete (stream synchronised above): the first iteration k >= 1 in which the median pair searched
    // fewer than a fifth of its queries is where certify + listed search starts to beat the fused kernel (measured: the listed
    // search costs ~1 ms per 64 pairs with every query listed, the fused kernel 0.25-0.35 ms whatever the share)
    int k = 1;
    std::vector<float> share((size_t)h->hist_pairs);
    for (; k < std::min(h->hist_iters, kSearchHist); ++k) {
      for (int p = 0; p < h->hist_pairs; ++p) share[p] = (float)h->hist_pinned[(size_t)p * kSearchHist + k] / (float)std::max(1, h->hist_ns[p]);
      std::nth_element(share.begin(), share.begin() + share.size() / 2, share.end());
      if (share[share.size() / 2] < h->split_share) break;
    }
    if (h->hist_mode == SMHIP_NN_NABO) h->nabo_fused_from = std::max(1, std::min(k, 12));   // (see fused_iteration)
    else h->auto_split = std::max(1, std::min(k, 8));         // 8: from there on the two-launch form won on every workload measured
    h->hist_pairs = 0;
  }
  if (h->opts.split_after == 0) h->dev.split_after = h->auto_split;
  h->prof.split_after_used = npairs >= 16 && h->dev.certify ? h->dev.split_after : 0;
  const bool cached_one = npairs == 1 && grid_cached(h, first);
  if (cached_one) s = enqueue_prepare_one(h, first, nt_max);     // target unchanged: pose + scratch reset only
  else s = enqueue_resets(h, npairs, first);
  if (s) return s;
  // Split the batch over several streams: the latency-bound launches of one part (finalize, validate, grid
  // build, near-empty refinement kernels) overlap the throughput-bound NN / accumulate of the others.
  // Parts are multiples of 8 pairs (the XCD mapping) and at least 16 pairs each.
  // (default two.  Measured on 512-pair batches with the fixed-grid tail kernels of round 5: 2 / 3 / 4 parts = 25.3 / 25.9 / 26.0 k
  // alignments/s, identity guesses 15.3 / 15.8 / 15.7 k, mixed 17.4 / 18.1 / 18.4 k -- overlap_streams = 4 is worth 1.5-5 % there; the
  // sequence driver's 256-pair batches lose 12 % with four parts of 64 pairs, so the default stays where every batch size is served)
  int want = h->opts.no_overlap ? 1 : (h->opts.overlap_streams > 0 ? h->opts.overlap_streams : 2);
  want = std::min(want, smhip_context::kMaxParts);
  if (want > 1 && npairs >= 32) ensure_side_streams(h, want - 1);
  want = std::min(want, 1 + h->n_side);
  while (want > 1 && npairs < 16 * want) --want;
  Half halves[smhip_context::kMaxParts];
  int nh = want;
  {
    int done = 0;
    for (int k = 0; k < nh; ++k) {
      int np = (k == nh - 1) ? npairs - done : (((npairs - done) / (nh - k) + 7) / 8) * 8;
      np = std::min(np, npairs - done);
      halves[k] = whole_batch(h, np, first + done);
      halves[k].stream = k == 0 ? h->stream : h->side[k - 1];
      done += np;
    }
  }
  auto fork = [&]() -> smhip_status {
    if (nh > 1) {
      HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
      for (int k = 1; k < nh; ++k) HIPCHK(h, hipStreamWaitEvent(h->side[k - 1], h->ev_fork, 0));
    }
    return SMHIP_OK;
  };
  auto join = [&]() -> smhip_status {
    for (int k = 1; k < nh; ++k) {
      HIPCHK(h, hipEventRecord(h->ev_join[k - 1], h->side[k - 1]));
      HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join[k - 1], 0));
    }
    return SMHIP_OK;
  };
  s = fork();
  if (s) return s;
  for (int k = 0; k < nh; ++k) {   // launches of fewer than ~2 workgroups per CU take the small-launch kernel variants
    halves[k].small = halves[k].np * ceil_div(ns_max, kNnThreads * kBallItems) < 512;
    // long accumulate chunks once they still leave >= 3 workgroups per CU
    // (or when short chunks would make more segments than finalize indexes)
    halves[k].d.acc_items = (halves[k].np * ceil_div(ns_max, kAccThreads * kAccItemsBatch) >= 768 ||
                             ceil_div(ns_max, kAccThreads * kAccItemsSmall) * (kAccThreads / 64) > kFinalizeMaxSeg) ? kAccItemsBatch : kAccItemsSmall;
  }
  const int max_it = h->dev.max_iteration;
  bool grid_built = false;
  // One pair (the front end's call, map_builder.cc:317-333): the whole loop and the score as ONE cooperative launch whose workgroups
  // meet at grid barriers (icp_one.hip) -- the same matches, distances and kept sets as the launches below.
  h->one_used = 0;
  // (up to kOnePairs pairs per launch, a row of the grid each: the back end's handful of concurrent submap pairs -- 6 pairs 1.40 ms
  // against 1.71 as separate launches; SMHIP_ONE_PAIRS=n lowers the limit, 1 = single pairs only)
  const int one_pairs_max = std::getenv("SMHIP_ONE_PAIRS") ? std::min(std::max(std::atoi(std::getenv("SMHIP_ONE_PAIRS")), 1), kOnePairs) : kOnePairs;
  if (npairs <= one_pairs_max && h->one_blocks > 0 && !h->opts.no_single_kernel && h->one_enabled && h->opts.nn_mode == SMHIP_NN_GRID && h->dev.use_ball &&
      h->dev.lds_table && h->dev.certify && !h->dev.exact_all && h->profile == 0) {
    const int nrounds = ceil_div(ns_max, kNnThreads);
    // two rounds of 256 points per workgroup (measured on 120 000 points, 20 iterations, target kept: 472 workgroups of one round
    // 1.15-1.18 ms, 320: 1.11-1.15, 240: 1.06-1.11, 160: 1.04-1.13 -- a barrier waits for the slowest workgroup, and two rounds
    // even out what one round's few searching queries cost); a multiple of 8: the barrier's groups.  Several pairs (up to
    // kOnePairs: the back end's handful of concurrent submap pairs) share what the device holds at once, a row of the grid each.
    int G = h->one_blocks_want > 0 ? std::max(8, (h->one_blocks_want / 8) * 8) : ((ceil_div(nrounds, 2) + 7) / 8) * 8;
    if (!(h->one_blocks_want > 0 && std::getenv("SMHIP_ONE_IDLE"))) G = std::min(G, ((nrounds + 7) / 8) * 8);   // (SMHIP_ONE_IDLE: tests run small clouds on a grid of mostly idle workgroups)
    if (G >= 64) G = ((G + 31) / 32) * 32;                 // (whole groups of the barrier; workgroups beyond the rounds only take part in the barriers)
    G = std::min(G, ((h->one_blocks / npairs) / 8) * 8);
    // (at most 12 rounds per workgroup -- the kernel holds up to kOneMaxRounds = 16 --: measured on 120 000-point pairs, 6 pairs at 12
    // rounds 1.52 ms against 1.71 as separate launches, 8 pairs at 15 rounds 1.87 against 1.73)
    if (G >= 8 && ceil_div(nrounds, G) <= std::min(12, kOneMaxRounds)) {
      if (!cached_one) { s = enqueue_grid_build(h, halves[0], nt_max); if (s) return s; }
      grid_built = true;
      IcpDev d1 = halves[0].d;
      d1.fused = 0; d1.fused_nabo = 0;
      // the barrier's groups: 8 (measured on 256 / 480 workgroups: 8 or 16 groups equal, 32 groups 7 % slower -- the barriers wait for
      // the slowest workgroup, not for their own atomics; SMHIP_ONE_GROUPS overrides)
      int groups = 8;
      if (h->one_groups_want > 0 && (h->one_groups_want & (h->one_groups_want - 1)) == 0 && h->one_groups_want <= 32 && G % h->one_groups_want == 0) groups = h->one_groups_want;
      void* args[] = {&d1, &groups};
      if (hipLaunchCooperativeKernel(reinterpret_cast<const void*>(icp_one), dim3(G, npairs), dim3(kNnThreads), args, 0, h->stream) == hipSuccess) {
        h->one_used = 1;
        h->one_launches += 1;
        h->last_npairs = npairs;
        return SMHIP_OK;
      }
      // the runtime refused the cooperative launch (it cannot place the grid): not an error of the Align -- the same iterations as
      // separate launches below, and no further attempts on this handle
      (void)hipGetLastError();
      h->one_blocks = 0;
    }
  }
  // one iteration of one part: FindClosests, the sums, finalize
  auto enqueue_iteration = [&](Half& f, int it) -> smhip_status {
    f.d.fused = fused_iteration(h, f, ns_max, it) ? 1 : 0;     // every launch of this iteration and part sees the same flag
    f.d.fused_nabo = f.d.fused && h->opts.nn_mode == SMHIP_NN_NABO ? 1 : 0;
    if (f.d.fused && !f.d.fused_nabo) {
      // A band needs two quantiles: the iteration after the first has none, so no pair's sums can come from the fused pass -- the
      // plain accumulate launch for all of them (fused = 0 for the sums only would change what finalize expects: keep the flag,
      // it reads spec_ok = 0).  The next `sums_long_for` fused iterations most predictions still miss (the quantile moves by more
      // than a bin): long blocks; after that short ones (iteration_sums).
      f.first_fused = f.first_fused < 0 ? it : f.first_fused;
      // (short blocks only while their record segments -- four per block -- fit finalize's table: clouds of up to a million points)
      const bool short_fits = ceil_div(ns_max, kAccThreads * kAccItemsSmall) * (kAccThreads / 64) <= kFinalizeMaxSeg;
      f.d.sums_items = (f.d.acc_items == kAccItemsBatch && (it - f.first_fused < h->sums_long_for || !short_fits)) ? kAccItemsBatch : kAccItemsSmall;
      if (it < 2) f.d.sums_items = f.d.acc_items;
    }
    smhip_status r = enqueue_find_closests_half(h, f, ns_max, it);
    if (r) return r;
    {
      Bracket br(h, 2, f.stream, f.np);
      const int nblk = ceil_div(ns_max, kAccThreads * f.d.acc_items);
      if (f.d.fused && !f.d.fused_nabo && it >= 2) {
        // fused iteration: only the pairs whose prediction missed need `accumulate`, the others the sums of their listed matches --
        // one fixed grid that takes both kinds of work (iteration_sums) instead of nblk workgroups per pair that look at a flag
        if (f.d.sums_items == kAccItemsBatch) hipLaunchKernelGGL(iteration_sums<kAccItemsBatch>, dim3(h->sums_blocks), dim3(kAccThreads), 0, f.stream, f.d);
        else hipLaunchKernelGGL(iteration_sums<kAccItemsSmall>, dim3(h->sums_blocks), dim3(kAccThreads), 0, f.stream, f.d);
      } else if (f.d.acc_items == kAccItemsBatch) hipLaunchKernelGGL(accumulate<kAccItemsBatch>, dim3(nblk * 8 * ceil_div(f.np, 8)), dim3(kAccThreads), 0, f.stream, f.d, nblk);
      else hipLaunchKernelGGL(accumulate<kAccItemsSmall>, dim3(nblk * 8 * ceil_div(f.np, 8)), dim3(kAccThreads), 0, f.stream, f.d, nblk);
    }
    { Bracket br(h, 3, f.stream); hipLaunchKernelGGL(finalize, dim3(f.np), dim3(256), 0, f.stream, f.d); }
    return SMHIP_OK;
  };
  // (The parts march in lock-step.  Tried in round 5: part k + 1 started two to five iterations behind part k, so that one part's
  // searching iterations -- bound by vector-instruction issue -- would run beside another's streaming ones -- bound by HBM: no gain
  // with 2, 3, 4 or 8 parts (24.0 / 23.9 / 23.5 / 18.7 k alignments/s against 24.4 k in lock-step).  Side by side the two kernels
  // split the registers: the certificate pass needs all of its waves to keep ~10 MB in flight, the search all of its to fill the
  // SIMDs, and each runs as much slower as the other gains.)
  if (!cached_one && !grid_built) for (int k = 0; k < nh; ++k) { s = enqueue_grid_build(h, halves[k], nt_max); if (s) return s; }
  for (int it = 0; it < max_it; ++it) {
    for (int k = 0; k < nh; ++k) {
      s = enqueue_iteration(halves[k], it);
      if (s) return s;
    }
    if (h->dev.early_exit && (it + 1) % h->opts.check_every == 0 && it + 1 < max_it) {
      s = join();
      if (s) return s;
      HIPCHK(h, hipMemcpyAsync(h->done_pinned, h->dev.done_count, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      if (*h->done_pinned >= (uint32_t)npairs) break;
      s = fork();      // re-fork for the next chunk of iterations
      if (s) return s;
    }
  }
  // the score of the iteration every pair left the loop with (icp_fast.cc:516-522), from that iteration's distances
  for (int k = 0; k < nh; ++k) {
    Half& f = halves[k];
    Bracket br(h, 3, f.stream);
    hipLaunchKernelGGL(final_score, dim3(f.np * kScoreParts), dim3(kAccThreads), 0, f.stream, f.d);
    hipLaunchKernelGGL(score_fold, dim3(ceil_div(f.np, 256)), dim3(256), 0, f.stream, f.d, f.np);
  }
  s = join();
  if (s) return s;
  // (only batches that ran the ball search with certificates say anything about where its two forms cross)
  if (npairs >= 16 && h->dev.certify &&
      ((h->opts.split_after == 0 && h->opts.nn_mode == SMHIP_NN_GRID && h->dev.use_ball && h->dev.lds_table) || h->opts.nn_mode == SMHIP_NN_NABO)) {
    h->hist_mode = h->opts.nn_mode;
    HIPCHK(h, hipMemcpyAsync(h->hist_pinned, h->dev.search_hist + (size_t)first * kSearchHist, sizeof(uint32_t) * kSearchHist * (size_t)npairs,
                             hipMemcpyDeviceToHost, h->stream));
    h->hist_first = first; h->hist_pairs = npairs; h->hist_iters = max_it;
    h->hist_ns.assign(h->ns.begin() + first, h->ns.begin() + first + npairs);
  }
  HIPCHK(h, hipGetLastError());
  h->last_npairs = npairs;
  return SMHIP_OK;
}

smhip_status smhip_icp_enqueue_batch(smhip_handle h, int npairs, const double* guesses) { return enqueue_range(h, 0, npairs, guesses); }

static smhip_status fetch_range(smhip_handle h, int first, int npairs, double* results, double* scores, smhip_icp_stats* stats) {
  if (!h || npairs < 1 || first < 0 || first + npairs > h->dev.slots) return SMHIP_ERR_INVALID_ARGUMENT;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(h->state_pinned, h->dev.state + first, sizeof(PairState) * npairs, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->one_used && !std::getenv("SMHIP_ONE_NO_RETRY")) {
    // The cooperative launch stopped itself (its barrier watchdog or its workgroups' consistency check, icp_one.hip): never seen
    // on a single pair, but the protocol rests on the timing of agent-scope loads, not on fences -- so the Align is simply done
    // again as separate launches (whose result differs from the launch's in the sums' order: ~1e-15) and the handle keeps to them.
    bool stopped = false;
    for (int p = 0; p < npairs; ++p)
      stopped = stopped || (h->state_pinned[p].done && (h->state_pinned[p].status == SMHIP_ERR_HIP || (h->state_pinned[p].status == SMHIP_OK && h->state_pinned[p].score_mismatch)));
    if (stopped) {
      std::vector<double> g(16 * (size_t)npairs);
      for (int p = 0; p < npairs; ++p)
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) g[16 * (size_t)p + 4 * c + r] = h->in_pinned[first + p].guess[4 * r + c];
      h->one_blocks = 0;
      h->one_fallbacks += 1;
      const smhip_status rs = enqueue_range(h, first, npairs, g.data());
      if (rs) return rs;
      HIPCHK(h, hipMemcpyAsync(h->state_pinned, h->dev.state + first, sizeof(PairState) * npairs, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
    }
  }
  collect_profile(h);
  smhip_status worst = SMHIP_OK;
  for (int p = 0; p < npairs; ++p) {
    const PairState& st = h->state_pinned[p];
    if (results) std::memcpy(results + 16 * p, st.result, sizeof(double) * 16);
    if (scores) scores[p] = st.score;
    if (stats) {
      stats[p].iterations = st.iter;
      stats[p].kept = st.kept;
      float lim; uint32_t key = st.limit_key; std::memcpy(&lim, &key, 4);
      stats[p].limit_d2 = lim;
      stats[p].fallback_queries = (int32_t)st.fallback_total;
      stats[p].status = st.status;
      stats[p].hard_queries = (int32_t)st.hard_total;
      stats[p].refined_iterations = (int32_t)st.refine_total;
      stats[p].searched_queries = (int32_t)st.searched_total;
      stats[p].fused_iterations = (int32_t)st.spec_hits;
    }
    if (st.status != SMHIP_OK && worst == SMHIP_OK) {
      worst = st.status;
      h->err = st.status == SMHIP_ERR_INVALID_ARGUMENT ? "pair failed: target cloud has NaN / Inf coordinates"
             : (st.status == SMHIP_ERR_HIP ? "pair failed: the single-pair launch stopped itself (a grid barrier did not complete, or its workgroups disagreed; internal)" : "pair failed: no finite correspondence");
      if (st.status == SMHIP_ERR_HIP && h->one_used) {
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpy(w, h->dev.one_sync + (size_t)p * kOneSyncWords + kSyncAbort, sizeof(w), hipMemcpyDeviceToHost);
        char buf[256];
        if (w[0] == 2u)
          std::snprintf(buf, sizeof(buf), " [pair %d: workgroups disagree -- workgroup %u at barrier %u has quantile key %u / %u distances, column %u averages %u]", p, w[1], w[2], w[3], w[4], w[5], w[6]);
        else
          std::snprintf(buf, sizeof(buf), " [pair %d: noticed by workgroup %u at barrier %u: group arrivals %u of %u, groups arrived %u of %u]", p, w[1], w[2], w[3], w[4], w[5], w[6]);
        h->err += buf;
#if SMHIP_ONE_CHECKS
        {   // (diagnostic build) the barriers every workgroup entered, in order: where do they part?
          const int G = 64, N = 64;
          std::vector<uint32_t> tr((size_t)G * N * 4);
          (void)hipMemcpy(tr.data(), h->dev.rec_a + (size_t)p * 2 * h->dev.bl_stride, tr.size() * 4, hipMemcpyDeviceToHost);
          for (int n = 0; n < N; ++n) {
            bool same = true;
            for (int w = 1; w < 40; ++w) for (int c = 0; c < 3; ++c) same = same && tr[((size_t)w * N + n) * 4 + c] == tr[(size_t)n * 4 + c];
            if (!same) {
              std::fprintf(stderr, "[icp_one trace] pair %d: workgroups part at their barrier %d:", p, n + 1);
              for (int w = 0; w < 40; ++w) std::fprintf(stderr, " %u/%u/%u/%u", tr[((size_t)w * N + n) * 4], tr[((size_t)w * N + n) * 4 + 1], tr[((size_t)w * N + n) * 4 + 2], tr[((size_t)w * N + n) * 4 + 3]);
              std::fprintf(stderr, "\n");
              break;
            }
          }
        }
#endif
      }
    }
    if (!st.done && worst == SMHIP_OK) { worst = SMHIP_ERR_HIP; h->err = "pair did not finish (internal)"; }
    if (st.done && st.status == SMHIP_OK && st.score_mismatch && worst == SMHIP_OK) {
      worst = SMHIP_ERR_HIP; h->err = "score: the matches summed are not the matches finalize kept (internal)";
      char buf[160];
      std::snprintf(buf, sizeof(buf), " [pair %d: %u distances at or below the quantile, %d kept by the last iteration (%d iterations, %u fused)]", p, st.score_cnt[0], st.kept, st.iter, st.spec_hits);
      h->err += buf;
    }
  }
  return worst;
}

smhip_status smhip_icp_fetch_batch(smhip_handle h, int npairs, double* results, double* scores, smhip_icp_stats* stats) {
  return fetch_range(h, 0, npairs, results, scores, stats);
}

smhip_status smhip_icp_align_range(smhip_handle h, int first_slot, int npairs, const double* guesses, double* results,
                                   double* scores, smhip_icp_stats* stats) {
  smhip_status s = enqueue_range(h, first_slot, npairs, guesses);
  if (s) return s;
  return fetch_range(h, first_slot, npairs, results, scores, stats);
}

__global__ void fill_unit_z_normals(float4* n, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) n[i] = make_float4(0.f, 0.f, 1.f, 0.f);
}

// The post-hoc score of IcpUsingPointMatcher::Align (icp_pointmatcher.cc:112-143): ONE FindClosests + TrimmedDist pass of
// the slot's source moved by T against the slot's target; score = exp(-mean distance over the kept matches).  That is
// exactly what a one-iteration IcpFast pass reports (its score comes from the matches made before the pose update), so
// the pass runs through the same kernels; normals play no part in it (a target without normals gets unit-z placeholders).
smhip_status smhip_icp_trimmed_score(smhip_handle h, int slot, const double T[16], float dist_outlier_ratio, double* score, int32_t* kept) {
  smhip_status s = check_slot(h, slot);
  if (s) return s;
  if (!T || !score) return SMHIP_ERR_INVALID_ARGUMENT;
  if (!(dist_outlier_ratio >= 0.f && dist_outlier_ratio <= 1.f)) { h->err = "dist_outlier_ratio must be in [0, 1]"; return SMHIP_ERR_INVALID_ARGUMENT; }
  if (h->ns[slot] <= 0 || h->nt[slot] <= 0) { h->err = "trimmed score before SetInputSource/SetInputTarget"; return SMHIP_ERR_NOT_READY; }
  HIPCHK(h, hipSetDevice(h->device));
  const int had = h->has_normals[slot];
  if (!had) {
    hipLaunchKernelGGL(fill_unit_z_normals, dim3(ceil_div(h->nt[slot], 256)), dim3(256), 0, h->stream,
                       const_cast<float4*>(h->dev.tgt_n) + (size_t)slot * h->dev.nt_cap, h->nt[slot]);
    h->has_normals[slot] = 1;
  }
  const smhip_icp_options saved = h->opts;
  smhip_icp_options o = saved;
  o.max_iteration = 1; o.early_exit = 0; o.dist_outlier_ratio = dist_outlier_ratio;
  h->opts = o;
  sync_options(h);
  double result[16];
  smhip_icp_stats st{};
  s = enqueue_range(h, slot, 1, T);
  if (s == SMHIP_OK) s = fetch_range(h, slot, 1, result, score, &st);
  h->opts = saved;
  sync_options(h);
  h->has_normals[slot] = had;
  if (kept) *kept = st.kept;
  return s;
}

smhip_status smhip_set_target_cache(smhip_handle h, int enable) {
  if (!h) return SMHIP_ERR_INVALID_ARGUMENT;
  h->target_cache = enable ? 1 : 0;
  return SMHIP_OK;
}

smhip_status smhip_icp_forget_search_history(smhip_handle h) {
  if (!h) return SMHIP_ERR_INVALID_ARGUMENT;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));       // (a batch's rows may still be on their way to hist_pinned)
  h->hist_pairs = 0;
  h->auto_split = 2;
  h->nabo_fused_from = 6;
  return SMHIP_OK;
}

smhip_status smhip_icp_single_launch_counts(smhip_handle h, int64_t* launches_used, int64_t* fallbacks) {
  if (!h) return SMHIP_ERR_INVALID_ARGUMENT;
  if (launches_used) *launches_used = h->one_launches;
  if (fallbacks) *fallbacks = h->one_fallbacks;
  return SMHIP_OK;
}

smhip_status smhip_get_capacity(smhip_handle h, int* pair_slots, int* max_source_points, int* max_target_points) {
  if (!h) return SMHIP_ERR_INVALID_ARGUMENT;
  if (pair_slots) *pair_slots = h->dev.slots;
  if (max_source_points) *max_source_points = h->dev.ns_cap;
  if (max_target_points) *max_target_points = h->dev.nt_cap;
  return SMHIP_OK;
}

smhip_status smhip_get_cloud_sizes(smhip_handle h, int slot, int* n_source, int* n_target, int* has_normals) {
  smhip_status s = check_slot(h, slot);
  if (s) return s;
  if (n_source) *n_source = h->ns[slot];
  if (n_target) *n_target = h->nt[slot];
  if (has_normals) *has_normals = h->has_normals[slot];
  return SMHIP_OK;
}

smhip_status smhip_icp_align_batch(smhip_handle h, int npairs, const double* guesses, double* results, double* scores,
                                   smhip_icp_stats* stats) {
  smhip_status s = smhip_icp_enqueue_batch(h, npairs, guesses);
  if (s) return s;
  return smhip_icp_fetch_batch(h, npairs, results, scores, stats);
}

smhip_status smhip_icp_align(smhip_handle h, const double guess[16], double result[16], double* score, smhip_icp_stats* stats) {
  return smhip_icp_align_batch(h, 1, guess, result, score, stats);
}

__global__ void export_results(IcpDev b, int npairs, double* out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npairs) return;
  const PairState* st = &b.state[p];
  for (int k = 0; k < 16; ++k) out[18 * p + k] = st->result[k];
  out[18 * p + 16] = st->score;
  out[18 * p + 17] = (double)st->iter;
}

smhip_status smhip_icp_export_results_device(smhip_handle h, int npairs, void* dev_out) {
  if (!h || !dev_out || npairs < 1 || npairs > h->dev.slots) return SMHIP_ERR_INVALID_ARGUMENT;
  HIPCHK(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(export_results, dim3(ceil_div(npairs, 64)), dim3(64), 0, h->stream, h->dev, npairs,
                     reinterpret_cast<double*>(dev_out));
  HIPCHK(h, hipGetLastError());
  return SMHIP_OK;
}

smhip_status smhip_icp_get_matches(smhip_handle h, int slot, int32_t* ids, float* d2, int n) {
  smhip_status s = check_slot(h, slot);
  if (s) return s;
  HIPCHK(h, hipSetDevice(h->device));
  return fetch_matches(h, slot, ids, d2, n);
}

smhip_status smhip_icp_find_closests(smhip_handle h, int slot, const double T[16], int32_t* ids, float* d2, int n) {
  smhip_status s = check_slot(h, slot);
  if (s) return s;
  if (!T) return SMHIP_ERR_INVALID_ARGUMENT;
  if (slot != 0) { h->err = "find_closests works on slot 0"; return SMHIP_ERR_INVALID_ARGUMENT; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  int ns_max = 0, nt_max = 0;
  // normals are irrelevant for the NN pass
  const int had = h->has_normals[0];
  h->has_normals[0] = 1;
  s = fill_inputs(h, 1, T, &ns_max, &nt_max);
  h->has_normals[0] = had;
  if (s) return s;
  s = enqueue_prepare(h, 1, nt_max);
  if (s) return s;
  const int exact_was = h->dev.exact_all;
  h->dev.exact_all = 1;                 // FindClosests contract: every match exact
  s = enqueue_find_closests(h, 1, ns_max);
  h->dev.exact_all = exact_was;
  if (s) return s;
  s = fetch_matches(h, 0, ids, d2, n);
  // leave the per-iteration scratch clean
  HIPCHK(h, hipMemsetAsync(h->dev.hist, 0, sizeof(uint32_t) * kHistBins, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->ev_used = 0;
  return s;
}

smhip_status smhip_icp_enable_profile(smhip_handle h, int enable) {
  if (!h) return SMHIP_ERR_INVALID_ARGUMENT;
  h->profile = (enable >= 2 && enable <= 4) ? enable : (enable != 0 ? 1 : 0);
  h->prof = smhip_icp_profile{};
  h->ev_used = 0;
  return SMHIP_OK;
}

smhip_status smhip_icp_get_search_counts(smhip_handle h, int slot, uint32_t* counts) {
  smhip_status s = check_slot(h, slot);
  if (s) return s;
  if (!counts) return SMHIP_ERR_INVALID_ARGUMENT;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(counts, h->dev.search_hist + (size_t)slot * kSearchHist, sizeof(uint32_t) * kSearchHist, hipMemcpyDeviceToHost));
  return SMHIP_OK;
}

smhip_status smhip_icp_get_profile(smhip_handle h, smhip_icp_profile* out) {
  if (!h || !out) return SMHIP_ERR_INVALID_ARGUMENT;
  *out = h->prof;
  return SMHIP_OK;
}

}  // extern "C"

// ---- registrators::Ndt --------------------------------------------------------------------------
#include "smhip_ndt_api.hip"

namespace {
NdtHost& ndt_of(smhip_context* h) {
  if (!h->ndt) { h->ndt = new smhip_ndt_state(); smhip_ndt_default_options(&h->ndt->n.opts); }
  return h->ndt->n;
}
}  // namespace

extern "C" void smhip_internal_free_ndt(smhip_context* h) {
  if (!h || !h->ndt) return;
  ndt_release(h->ndt->n);
  delete h->ndt;
  h->ndt = nullptr;
}

// ---- pre_processers::filter -----------------------------------------------------------------------
#include "smhip_filter_api.hip"

// ---- registrators::NdtWithGicp ------------------------------------------------------------------
#include "smhip_gicp_api.hip"

struct smhip_gicp_state { GicpHost g; };

namespace {
GicpHost& gicp_of(smhip_context* h) {
  if (!h->gicp) { h->gicp = new smhip_gicp_state(); smhip_ndt_gicp_default_options(&h->gicp->g.opts); }
  return h->gicp->g;
}
}  // namespace

extern "C" void smhip_internal_free_gicp(smhip_context* h) {
  if (!h || !h->gicp) return;
  GicpHost& g = h->gicp->g;
  if (g.out_pinned) (void)hipHostFree(g.out_pinned);
  if (g.count_pinned) (void)hipHostFree(g.count_pinned);
  if (g.prep_avg) prep_destroy(g.prep_avg);
  delete h->gicp;
  h->gicp = nullptr;
}
