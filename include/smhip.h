/* smhip.h -- C ABI of the MI355X (gfx950) scan-matching backend.
 *
 * This is the drop-in boundary for StaticMapping's per-frame registration hot
 * path.  Nothing like it exists in the reference: the reference
 * reaches its registrators through the C++ class
 *   static_map::registrator::Interface      /root/reference/registrators/interface.h:67-116
 * created by
 *   CreateMatcher(const MatcherOptions&)    /root/reference/registrators/interface.cc:139-173
 * The C++ mirror of that class lives in include/smhip/registrator.h and is a
 * thin adapter over the entry points below; INTEGRATION.md shows the lines a
 * maintainer adds to interface.cc to select it.
 *
 * Conventions
 *   - plain C types only; no exceptions cross the boundary; every call returns
 *     an smhip_status (0 = ok) and smhip_last_error(h) gives the text;
 *   - 4x4 transforms are COLUMN-major doubles (Eigen::Matrix4d storage), and
 *     map SOURCE-frame points into the TARGET frame
 *     (pose_source = pose_target * align_result, builder/map_builder.cc:354);
 *   - a handle owns one HIP stream (or borrows the caller's), all its device
 *     buffers and a fixed number of independent "pair slots"; slot 0 is what the
 *     single-pair registrator::Interface adapter uses, slots 0..B-1 are used by
 *     the batched / sharded benchmark path.  One thread per handle; different
 *     handles are independent (the reference runs up to 6 matchers
 *     concurrently, builder/map_builder.cc:655,706-708).
 */
#ifndef SMHIP_H_
#define SMHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int smhip_status;
enum {
  SMHIP_OK = 0,
  SMHIP_ERR_INVALID_ARGUMENT = 1,
  SMHIP_ERR_NO_DEVICE = 2,      /* no gfx950 device / HIP runtime failure at create */
  SMHIP_ERR_HIP = 3,            /* a HIP call failed; see smhip_last_error */
  SMHIP_ERR_NOT_READY = 4,      /* Align before SetInputSource/SetInputTarget */
  SMHIP_ERR_NO_NORMALS = 5,     /* IcpFast target without normals (icp_fast.cc:430 CHECK) */
  SMHIP_ERR_NO_MATCH = 6,       /* no finite correspondence (icp_fast.cc:81 CHECK(!values.empty())) */
  SMHIP_ERR_CAPACITY = 7        /* cloud larger than the handle was created for */
};

typedef struct smhip_context* smhip_handle;

/* nn_mode values */
enum {
  SMHIP_NN_BRUTE = 0,           /* LDS-tiled exact brute force (BASELINE config #2) */
  SMHIP_NN_GRID = 1,            /* bit-rank voxel grid, exact with brute-force fallback */
  SMHIP_NN_NABO = 2             /* the reference's own search: libnabo 1.0.7's KDTREE_LINEAR_HEAP tree rebuilt per Align and
                                   its epsilon-approximate knn (icp_fast.cc:169-180, 464-467), nn_epsilon = 3.16 there.
                                   APPROXIMATE by design: the mode that reproduces what a libnabo build of the reference
                                   returns; the exact modes differ from it by millimetres */
};

/* Options of the IcpFast-equivalent matcher.  max_iteration / dist_outlier_ratio
 * mirror IcpFast::options_ (/root/reference/registrators/icp_fast.h:56-60) and
 * are the names registered at icp_fast.cc:407-419. */
typedef struct smhip_icp_options {
  int32_t max_iteration;        /* default 100 */
  float dist_outlier_ratio;     /* default 0.7f */
  int32_t early_exit;           /* 1 = CheckConvergence enabled (reference behaviour, icp_fast.cc:377-405);
                                   0 = run exactly max_iteration iterations (throughput runs) */
  int32_t nn_mode;              /* SMHIP_NN_GRID (default), SMHIP_NN_BRUTE or SMHIP_NN_NABO */
  float grid_cell;              /* voxel edge in metres for SMHIP_NN_GRID (default 0.25) */
  int32_t grid_max_ring;        /* largest ring searched in the grid before the brute-force fallback (default 8: rings 1, 2, 4, 8) */
  int32_t check_every;          /* host polls the device "all done" word every this many iterations (default 8) */
  int32_t use_ball;             /* 1 (default): ball-bounded search seeded by the previous iteration's match, with
                                   certified trimming (matches beyond the quantile carry a proven lower bound
                                   instead of an exact distance); 0: exact ring search for every query */
  int32_t exact_matches;        /* 1: refine every lower bound to the exact match in every iteration (default 0;
                                   the transform / score are identical either way, only rejected matches differ) */
  float ball_radius;            /* largest search radius of the ball search in metres (default 0.3; queries with nothing that close
                                   are lower-bounded and, if the trimming quantile reaches the bound, refined exactly) */
  float ball_cap_factor;        /* next iteration's search-radius cap = factor x this iteration's quantile distance (default 1.5) */
  int32_t no_certify;           /* 1 = search every query in every iteration instead of first trying the nearest-neighbour
                                   certificate (runner-up bound minus the query's motion); default 0 = certificates on */
  int32_t no_lds_table;         /* 1: voxel lookups from global memory (nn_ball) instead of LDS row tables (nn_ball_lds) */
  int32_t no_overlap;           /* 1: keep a batch on one stream (default 0: a batch is split into parts of >= 16 pairs on
                                   separate streams so one part's latency-bound launches hide behind the others' NN) */
  int32_t overlap_streams;      /* number of such parts, 1..4; 0 = default (2; 4 is 1.5-5 % faster on 512-pair batches, slower on 256-pair ones).
                                 * More than 2 parts (or RCCL beside the matcher) need more hardware queues than the runtime's default 4, or two
                                 * parts share a queue and run one after the other: export GPU_MAX_HW_QUEUES=8 before the process's first HIP
                                 * call (smhip_shard and bench.py do) */
  int32_t split_after;          /* batches: iterations >= this run the certificate pass and the search of the failing queries as two
                                   launches instead of the fused kernel (pays once few certificates fail).  0 = automatic: 2 for a
                                   handle's first batch, then the first iteration in which the median pair of the PREVIOUS batch
                                   searched fewer than a fifth of its queries (smhip_icp_profile.split_after_used reports it);
                                   negative = never.  Results are identical either way. */
  float nn_epsilon;             /* SMHIP_NN_NABO only: libnabo's epsilon (default 3.16, icp_fast.cc:174; 0 = exact through the tree) */
  int32_t no_fused_sums;        /* batches: 1 = certificate pass and normal-equation sums as two passes over the source in every iteration.
                                   Default 0: from split_after on ONE pass does both, summing the matches below a predicted band of
                                   histogram bins around the trimming quantile and leaving the band's members for the exact select
                                   (a missed prediction costs that pair one plain pass).  The same matches, distances and kept set
                                   either way; the 29 sums are added in a different (fixed) order: poses agree to ~1e-12.
                                   SMHIP_NN_NABO has the same form (traversal certificates instead of distance bounds), from the
                                   iteration on in which the previous batch's median pair walked fewer than a fifth of its queries. */
  int32_t no_single_kernel;     /* a single pair (smhip_icp_align) or a batch of up to 8 pairs: 1 = the iteration loop as separate launches per
                                   iteration.  Default 0: the whole loop of icp_fast.cc:484-523 and the score as ONE cooperative launch whose
                                   workgroups stay resident and meet at grid barriers, a row of the grid per pair (exact-search mode with
                                   certificates; other modes and clouds of more than 16 rounds of 256 points per resident workgroup take the
                                   separate launches).  The same matches, distances and kept sets either way; the sums are added in a different
                                   (fixed) order: poses agree to ~1e-14.  Cooperative launches of several handles run one after the other. */
} smhip_icp_options;

/* Per-call statistics (all optional to read). */
typedef struct smhip_icp_stats {
  int32_t iterations;           /* iterations executed by this pair */
  int32_t kept;                 /* correspondences kept by the trimmed-distance filter in the last iteration */
  double limit_d2;              /* the quantile (squared distance) of the last iteration */
  int32_t fallback_queries;     /* queries resolved by the brute-force fallback, summed over iterations */
  int32_t status;               /* per-pair smhip_status */
  int32_t hard_queries;         /* matches recorded as lower bounds by the ball search, summed over iterations */
  int32_t refined_iterations;   /* iterations in which those bounds had to be refined to exact matches */
  int32_t searched_queries;     /* queries that needed a search (certificate failed / first iteration), summed over iterations */
  int32_t fused_iterations;     /* iterations whose sums came from the fused certificate pass (its quantile band held) */
} smhip_icp_stats;

/* Kernel-time breakdown collected when profiling is enabled (HIP events around
 * every launch, recorded on the stream the launch goes to: a batch of >= 16 pairs
 * runs as two halves on two streams, so the sums can exceed the wall time).  Names follow the reference's
 * REGISTER_BLOCK labels where one exists (icp_fast.cc:103,171,261,484). */
typedef struct smhip_icp_profile {
  double ms_prepare;            /* target centring + grid build (replaces kd-tree build, icp_fast.cc:464-467) */
  double ms_find_closests;      /* "FindClosests": all NN kernels (main + fallback), summed over launches */
  double ms_error_elements;     /* "ErrorElements"+"ComputePointToPlane" accumulate kernel */
  double ms_solve;              /* select + 6x6 solve + convergence kernel */
  int32_t launches_find_closests;
  int32_t launches_error_elements;
  int32_t launches_solve;
  int32_t launches_nn_main;     /* launches of the dominant NN kernel alone (nn_ball_lds; the certificate / listed-search
                                   launches of the converged iterations count under find_closests only) */
  double ms_nn_main;            /* its summed duration (subset of ms_find_closests) */
  double pairs_nn_main;         /* pairs those launches covered, summed (pairs per launch = this / launches_nn_main) */
  double ms_nn_certify;         /* the certificate pass (nn_certify) alone: summed duration (subset of ms_find_closests) */
  double pairs_nn_certify;      /* pairs its launches covered, summed */
  int32_t launches_nn_certify;
  int32_t split_after_used;     /* iteration from which the last batched enqueue ran certificate pass + listed search (filled with or
                                   without profiling; 0 = the last enqueue was not a batch of >= 16 pairs) */
  double ms_nn_listed;          /* the search of the queries whose certificate failed (nn_ball_listed; SMHIP_NN_NABO: the list walk) */
  double pairs_nn_listed;
  double ms_nn_refine;          /* the refinement launches of an iteration (nn_validate + nn_ring + nn_fallback), near-empty when the
                                   quantile stays below every recorded lower bound */
  double pairs_error_elements;  /* pairs the accumulate launches covered, summed */
  int32_t launches_nn_listed;
  int32_t launches_nn_refine;
} smhip_icp_profile;

/* ---- library / device ------------------------------------------------- */
int smhip_version(void);
const char* smhip_status_string(smhip_status s);
/* number of visible HIP devices whose gcnArchName starts with "gfx950"; <0 on runtime failure */
int smhip_device_count(void);

/* ---- handle ------------------------------------------------------------ */
/* stream: a hipStream_t to borrow (e.g. torch's current stream) or NULL to own one.
 * pair_slots >= 1; max_source_points / max_target_points size the device arena once (max_source_points <= 4194304).
 * (no hipMalloc happens inside Align). */
smhip_status smhip_create(int device, void* stream, int pair_slots, int max_source_points,
                          int max_target_points, smhip_handle* out);
smhip_status smhip_destroy(smhip_handle h);
const char* smhip_last_error(smhip_handle h);
void smhip_icp_default_options(smhip_icp_options* o);
smhip_status smhip_icp_set_options(smhip_handle h, const smhip_icp_options* o);
smhip_status smhip_synchronize(smhip_handle h);

/* ---- inputs (host memory; copied to the device immediately) ------------
 * replaces IcpFast::SetInputSource / SetInputTarget (icp_fast.cc:421-453), which
 * deep-copy the caller's EigenPointCloud (3xN column-major f64 = xyzxyz...). */
smhip_status smhip_set_source_f64(smhip_handle h, int slot, const double* xyz_colmajor_3xN, int n);
smhip_status smhip_set_target_f64(smhip_handle h, int slot, const double* xyz_colmajor_3xN,
                                  const double* normals_colmajor_3xN, int n);
/* The source is re-ordered along a Morton curve on upload (spatially coherent wavefronts); ids / d2
 * returned by smhip_icp_get_matches are in the CALLER's order again.
 * float32 points with a stride in floats: stride 4 = KITTI .bin rows (ros_node/kitti_reader.cc:91-121),
 * stride 5 = data::InnerPointType AoS {x,y,z,intensity,factor} (builder/data/cloud_types.h:46-56). */
smhip_status smhip_set_source_f32(smhip_handle h, int slot, const float* xyz, int stride_floats, int n);
/* `count` source clouds at once (the sequence driver's batch: SetInputSource of many pairs): cloud k = rows[k], n[k] rows of 4
 * floats (KITTI rows) into slots[k].  One host-to-device copy per cloud, ONE Morton ordering for the batch (a single upload
 * costs ~25 small launches).  rows[k] in pinned memory (hipHostMalloc / hipHostRegister) are copied from where they lie,
 * asynchronously: they must stay untouched until the handle's stream has passed the copies (smhip_synchronize, or any call
 * that blocks on the stream, e.g. smhip_prepare_targets_from_sources); pageable rows go through the handle's staging buffer
 * one cloud at a time.  Needs stride 4. */
smhip_status smhip_set_sources_f32_batch(smhip_handle h, int count, const int* slots, const float* const* rows, const int* n);
/* allocates the workspaces of the batched calls (smhip_set_sources_f32_batch, smhip_prepare_targets_from_sources) now rather
 * than inside the first batch -- several GB for a handle of hundreds of slots */
smhip_status smhip_reserve_batch_workspaces(smhip_handle h);
smhip_status smhip_set_target_f32(smhip_handle h, int slot, const float* xyz, int xyz_stride_floats,
                                  const float* normals, int normals_stride_floats, int n);
/* Re-use slot `from`'s device-resident clouds in slot `to` (benchmark replication; no host copy). */
smhip_status smhip_copy_slot(smhip_handle h, int from, int to);

/* ---- IcpFast::Align (icp_fast.cc:455-529) ------------------------------
 * Aligns slots [0, npairs).  guesses/results: npairs * 16 doubles, column-major.
 * scores: exp(-mean sqrt(d2)) over kept matches of the last iteration (icp_fast.cc:518-521).
 * stats may be NULL.  Blocking: returns after the results are in host memory. */
smhip_status smhip_icp_align(smhip_handle h, const double guess[16], double result[16], double* score,
                             smhip_icp_stats* stats);
smhip_status smhip_icp_align_batch(smhip_handle h, int npairs, const double* guesses, double* results,
                                   double* scores, smhip_icp_stats* stats);
/* The same for the pair slots [first_slot, first_slot + npairs): lets one handle keep several independent jobs resident
 * (the back end's concurrent SubmapPairMatch tasks, builder/map_builder.cc:655; an ICP pair beside its score pair). */
smhip_status smhip_icp_align_range(smhip_handle h, int first_slot, int npairs, const double* guesses, double* results,
                                   double* scores, smhip_icp_stats* stats);
/* IcpUsingPointMatcher::Align's post-hoc score (icp_pointmatcher.cc:112-143): one FindClosests + TrimmedDist(ratio) pass of
 * the slot's source moved by T (column-major) against the slot's target; *score = exp(-mean distance of the kept matches),
 * *kept (may be NULL) = their number.  The target needs no normals. */
smhip_status smhip_icp_trimmed_score(smhip_handle h, int slot, const double T[16], float dist_outlier_ratio, double* score,
                                     int32_t* kept);
/* Asynchronous halves of the batch call: enqueue leaves everything on the stream (needs
 * early_exit = 0 or accepts running all max_iteration launches), fetch blocks and copies out. */
smhip_status smhip_icp_enqueue_batch(smhip_handle h, int npairs, const double* guesses);
smhip_status smhip_icp_fetch_batch(smhip_handle h, int npairs, double* results, double* scores,
                                   smhip_icp_stats* stats);

/* Device-resident copy of the last results: writes npairs * 18 doubles to device memory at
 * dev_out (16 column-major transform + score + iterations) on the handle's stream, so a collective
 * (RCCL gather of the poses) can consume them without a host round trip. */
smhip_status smhip_icp_export_results_device(smhip_handle h, int npairs, void* dev_out);

/* ---- target preparation (host) -------------------------------------------
 * EigenPointCloud::CalculateNormals (builder/data/cloud_types.cc:347-368): kd-box subsampling +
 * unconstrained-LS normals.  The reference's CALLER runs it before SetInputTarget
 * (builder/map_builder.cc:286,389), so it is a free function, not part of Align.
 * xyz: 3xN column-major; outputs need room for n points; *n_out = surviving points. */
smhip_status smhip_calculate_normals_f64(const double* xyz_colmajor_3xN, int n, double* out_xyz,
                                         double* out_normals, int* n_out);

/* Device-side target preparation: upload a raw scan (float32 rows, stride in
 * floats) and run CalculateNormals on the GPU; the surviving points + normals become `slot`'s target.
 * *n_out = number of target points.  smhip_prepare_target_from_source does the same from the source
 * cloud already resident in slot `from` (scan-to-scan sequences: scan i is the source of pair i-1 and the
 * target of pair i, uploaded once). */
smhip_status smhip_prepare_target_f32(smhip_handle h, int slot, const float* xyz, int stride_floats, int n, int* n_out);
smhip_status smhip_prepare_target_from_source(smhip_handle h, int from_slot, int to_slot, int* n_out);
/* Batched form: count targets in ONE pass (one kd forest, one sort per tree level for all scans);
 * target of to_slots[k] = CalculateNormals(source cloud of from_slots[k]); n_out[k] = its size. */
smhip_status smhip_prepare_targets_from_sources(smhip_handle h, int count, const int* from_slots, const int* to_slots, int* n_out);

/* CalculateNormals of the RAW target resident in from_slot becomes the (points + normals) target of to_slot -- the
 * reference filter of the IcpUsingPointMatcher chain (SamplingSurfaceNormal, icp_pointmatcher.cc:176-184) without a
 * second upload of the reference cloud.  from_slot != to_slot. */
smhip_status smhip_prepare_target_from_target(smhip_handle h, int from_slot, int to_slot, int* n_out);
/* RandomSamplingDataPointsFilter (icp_pointmatcher.cc:170-174) on the device: the source of to_slot = the points of
 * from_slot's source whose uniform draw is < prob (prob >= 1 keeps everything).  The draw is a pure function of
 * (seed, the point's index in the cloud as uploaded): splitmix64(seed << 32 | index) >> 11, times 2^-53. */
smhip_status smhip_sample_source(smhip_handle h, int from_slot, int to_slot, float prob, uint32_t seed, int* n_out);
/* Target-side structures (the ICP search grid, the NDT voxel table; for NdtWithGicp also the down-sampled target, the
 * correspondence grid and the target's GICP covariances) are pure functions of the target cloud and the options;
 * single-pair calls keep them while a slot's target is unchanged -- the front end aligns scan after scan against one key
 * frame / submap (builder/map_builder.cc:379-392) -- and rebuild them whenever the target is set, prepared or copied again
 * or an option they depend on changes.  enable = 0: nothing survives from one Align to the next, as in the reference
 * (icp_fast.cc:464-467, ndt.cc:54, ndt_gicp.cc:55-81); default 1.  Results are identical either way. */
smhip_status smhip_set_target_cache(smhip_handle h, int enable);
/* IcpFast Aligns of up to 8 pairs run as ONE cooperative launch (csrc/icp_one.hip: the loop of icp_fast.cc:484-523 + the score).
 * Should that launch stop itself (its barrier watchdog, its workgroups' consistency columns, a score whose count differs from the
 * kept count), the fetch runs the Align again as separate launches and the handle keeps to them: *launches_used = Aligns that went
 * through the one launch, *fallbacks = how often that happened (0 on every run recorded so far).  Either pointer may be NULL. */
smhip_status smhip_icp_single_launch_counts(smhip_handle h, int64_t* launches_used, int64_t* fallbacks);
/* With split_after = 0 a batch of >= 16 pairs switches from the every-query search to certificates + listed search at the iteration
 * the PREVIOUS batch suggests (a front end's guesses are alike from call to call); results never depend on it, only the time.  This
 * call forgets what earlier batches taught the handle -- e.g. after a warm-up batch of made-up clouds (smhip_shard) -- so that the
 * next batch starts from the defaults a new handle has. */
smhip_status smhip_icp_forget_search_history(smhip_handle h);
/* what the handle was created with / what a slot currently holds (any pointer may be NULL) */
smhip_status smhip_get_capacity(smhip_handle h, int* pair_slots, int* max_source_points, int* max_target_points);
smhip_status smhip_get_cloud_sizes(smhip_handle h, int slot, int* n_source, int* n_target, int* has_normals);

/* ---- introspection for parity tests ------------------------------------
 * Matches of the LAST executed iteration of `slot` (FindClosests output, icp_fast.cc:169-180):
 * ids index the target cloud in the caller's order, d2 are squared distances (float32).  Every match
 * the trimmed-distance filter KEPT is exact; with use_ball = 1 and exact_matches = 0 a rejected match
 * may carry a certified lower bound (> the quantile) and the best target seen so far instead. */
smhip_status smhip_icp_get_matches(smhip_handle h, int slot, int32_t* ids, float* d2, int n);
/* One FindClosests pass only: transform the slot's source by T (column-major 4x4, applied AFTER
 * centring exactly as Align does) and return ids / d2 without running ICP. */
smhip_status smhip_icp_find_closests(smhip_handle h, int slot, const double T[16], int32_t* ids, float* d2, int n);

/* ---- registrators::Ndt (pclomp NDT, /root/reference/registrators/ndt.cc:29-64) -------------------
 * Uses slot 0's clouds (upload them with smhip_set_source_f32 / smhip_set_target_f32: the reference's
 * Ndt converts the InnerPointType AoS to pcl::PointXYZ, ndt.cc:44-51; the target needs no normals).
 * Defaults are the wrapper's: resolution 1.0 (ndt.cc:31), KDTREE neighbourhood (ndt.cc:33), step 0.1,
 * outlier ratio 0.55, transformation epsilon 0.1, 35 iterations (pclomp/ndt_omp_impl.hpp:47-76). */
typedef struct smhip_ndt_options {
  float resolution;
  float step_size;
  float outlier_ratio;
  float transformation_epsilon;
  int32_t max_iterations;
  int32_t min_points_per_voxel;      /* voxel_grid_covariance_omp.h:204 */
  float min_covar_eigvalue_mult;     /* voxel_grid_covariance_omp.h:205 */
  int32_t reserved[5];
} smhip_ndt_options;

typedef struct smhip_ndt_stats {
  int32_t iterations;                /* nr_iterations_ */
  int32_t derivative_calls;          /* computeDerivatives evaluations (1 + one per Newton iteration) */
  int32_t voxels;                    /* occupied voxels of the target */
  int32_t status;
  double trans_probability;          /* score / N (ndt_omp_impl.hpp:170) */
  double pairs_last;                 /* (point, voxel) pairs of the last evaluation = N * mean neighbours */
} smhip_ndt_stats;

void smhip_ndt_default_options(smhip_ndt_options* o);
smhip_status smhip_ndt_set_options(smhip_handle h, const smhip_ndt_options* o);
/* Ndt::Align: voxel grid build + Newton iterations + getFitnessScore.  *score = mean squared 1-NN
 * distance of the aligned source to the raw target (LOWER is better, unlike the ICP score). */
smhip_status smhip_ndt_align(smhip_handle h, const double guess[16], double result[16], double* score,
                             smhip_ndt_stats* stats);
/* npairs independent Ndt::Align calls -- pair slots first_slot .. first_slot + npairs - 1, clouds set with smhip_set_source_f32 /
 * smhip_set_target_f32 on those slots -- advanced in lock-step: the back end runs up to six SubmapPairMatch tasks at once with
 * whichever matcher is configured (builder/map_builder.cc:399-446, 655).  Every pair's Newton / More-Thuente state machine
 * (pclomp/ndt_omp_impl.hpp:81-171, 757-916) lives in device memory and asks for the evaluations the reference would make, in the
 * reference's order; a round = one computeDerivatives launch over all pairs still running + one control launch (fold, 6x6 solve,
 * line-search decision, next pose), the voxel tables are built in one pass and the fitness scores in another: the whole batch is one
 * submission and one synchronise.  Results are the single calls' bit for bit.
 * guesses / results: npairs column-major 4x4; scores / stats: npairs entries (may be NULL). */
smhip_status smhip_ndt_align_batch(smhip_handle h, int first_slot, int npairs, const double* guesses, double* results,
                                   double* scores, smhip_ndt_stats* stats);
/* parity-test hooks: VoxelGridCovariance::applyFilter output and one computeDerivatives evaluation
 * at pose6 = (tx, ty, tz, rx, ry, rz).  icovs hold xx xy xz yy yz zz as float; hess is row-major 6x6. */
smhip_status smhip_ndt_build_voxels(smhip_handle h, int* n_voxels);
smhip_status smhip_ndt_get_voxels(smhip_handle h, int capacity, int32_t* keys, int32_t* counts, double* means,
                                  float* icovs, float* centroids);
/* copy slot's current target (as uploaded / prepared) back to the host: xyz and normals, 3 floats each */
smhip_status smhip_get_target_f32(smhip_handle h, int slot, float* xyz, float* normals, int n);
smhip_status smhip_ndt_compute_derivatives(smhip_handle h, const double pose6[6], int compute_hessian,
                                           double* score, double grad[6], double hess[36]);
/* measurement hook: `launches` back-to-back computeDerivatives launches (with Hessian) over slots first_slot .. + npairs - 1 at the
 * poses their last Align ended with, HIP events around them on the handle's stream: *ms_per_launch = the average duration of
 * the kernel, *pairs_per_launch = the (point, voxel) pairs one launch works through (for its algorithmic bytes).  The slots'
 * voxel tables must be current (an Align with the target cache on, or smhip_ndt_build_voxels for slot 0). */
smhip_status smhip_ndt_time_derivatives(smhip_handle h, int first_slot, int npairs, int launches, double* ms_per_launch,
                                        double* pairs_per_launch);

/* ---- registrators::NdtWithGicp (ndt_gicp.cc:28-112) ------------------------------------------
 * ApproximateVoxelGrid (0.2 m) on both clouds -> stock pcl NDT -> stock pcl GICP; PCL is not vendored by the
 * reference, the arithmetic follows PCL 1.8.1 (see oracle/ndt_gicp.py for the pinning).  The matcher keeps its
 * raw clouds in private device buffers and uses two pair slots of the handle per job as working space (job j: slot j and
 * slot pair_slots / 2 + j; the single Align is job 0), so the handle must be created with pair_slots >= 2,
 * max_source_points / max_target_points >= the raw cloud sizes and max_target_points >= the down-sampled source size.
 * One grid cell (gicp_search_cell) serves every search of an Align, so a target's search grid is built once per Align. */
typedef struct smhip_ndt_gicp_options {
  float voxel_resolution;              /* 0.2  ndt_gicp.h:73 */
  int32_t using_voxel_filter;          /* 1    ndt_gicp.h:74 */
  int32_t use_ndt;                     /* 1    ndt_gicp.h:75 */
  float ndt_transformation_epsilon;    /* 0.01 ndt_gicp.cc:38 */
  float ndt_step_size;                 /* 0.1  :39 */
  float ndt_resolution;                /* 1.0  :40 */
  int32_t ndt_max_iterations;          /* 35   :41 */
  int32_t gicp_max_iterations;         /* 35   :44 */
  double gicp_rotation_epsilon;        /* 1e-3 :43 */
  double gicp_transformation_epsilon;  /* 5e-4 PCL default (pclomp/gicp_omp.h:117) */
  double gicp_epsilon;                 /* 1e-3 (gicp_omp.h:109) */
  double gicp_corr_dist_threshold;     /* 5 m  (gicp_omp.h:118) */
  int32_t gicp_max_inner_iterations;   /* 20   (gicp_omp.h:112) */
  int32_t gicp_k_correspondences;      /* 20   (gicp_omp.h:108) */
  float gicp_search_cell;              /* grid cell (m) of the k-NN search behind the covariances; 0 = 3 x voxel_resolution.
                                          Performance only: the search is exact for any cell. */
  int32_t reserved[3];
} smhip_ndt_gicp_options;

typedef struct smhip_ndt_gicp_stats {
  int32_t ok;                          /* Align's bool: 0 when the NDT fitness was > 1 (result = guess) */
  int32_t n_source, n_target;          /* cloud sizes after the voxel filter */
  int32_t ndt_iterations;
  int32_t gicp_iterations;             /* outer iterations (nr_iterations_) */
  int32_t gicp_function_evaluations;   /* functor evaluations of all BFGS runs */
  int32_t gicp_correspondences;        /* kept correspondences of the last outer iteration */
  int32_t reserved;
  double ndt_score;                    /* ndt_.getFitnessScore() (0.9 when use_ndt = 0) */
  double gicp_score;                   /* gicp_.getFitnessScore() (10 when skipped) */
} smhip_ndt_gicp_stats;

void smhip_ndt_gicp_default_options(smhip_ndt_gicp_options* o);
smhip_status smhip_ndt_gicp_set_options(smhip_handle h, const smhip_ndt_gicp_options* o);
/* SetInputSource / SetInputTarget of this matcher: raw clouds (stride 3, 4 or 5 floats), kept as handed over */
smhip_status smhip_ndt_gicp_set_source_f32(smhip_handle h, const float* xyz, int stride_floats, int n);
smhip_status smhip_ndt_gicp_set_target_f32(smhip_handle h, const float* xyz, int stride_floats, int n);
/* NdtWithGicp::Align.  *score = exp(-GICP fitness) (exp(-10) when stats->ok == 0, ndt_gicp.cc:103,107). */
smhip_status smhip_ndt_gicp_align(smhip_handle h, const double guess[16], double result[16], double* score,
                                  smhip_ndt_gicp_stats* stats);
/* Several NdtWithGicp pairs on one handle: a handle created with S pair slots runs S / 2 JOBS (smhip_ndt_gicp_jobs); job j has
 * its own clouds (the calls above are job 0's) and keeps its own target-derived structures between Aligns.
 * smhip_ndt_gicp_align_batch aligns jobs first_job .. first_job + njobs - 1 in lock-step -- the stages of
 * NdtWithGicp::Align (ndt_gicp.cc:55-112) run over all jobs at once, and inside NDT and GICP every job keeps the reference's own
 * sequence of evaluations while each round of evaluations is ONE launch and one hand-back for all jobs that are still
 * running (the back end's six concurrent SubmapPairMatch tasks, map_builder.cc:399-446, 655, as one call).  Results are
 * bit-identical to the single calls'.  guesses / results: njobs column-major 4x4; scores / stats: njobs entries (nullable). */
int smhip_ndt_gicp_jobs(smhip_handle h);
smhip_status smhip_ndt_gicp_set_source_f32_job(smhip_handle h, int job, const float* xyz, int stride_floats, int n);
smhip_status smhip_ndt_gicp_set_target_f32_job(smhip_handle h, int job, const float* xyz, int stride_floats, int n);
smhip_status smhip_ndt_gicp_align_batch(smhip_handle h, int first_job, int njobs, const double* guesses, double* results,
                                        double* scores, smhip_ndt_gicp_stats* stats);
/* pcl GICP alone on slot 0's clouds (smhip_set_source_f32 / smhip_set_target_f32); *fitness = getFitnessScore() */
smhip_status smhip_gicp_align(smhip_handle h, const double guess[16], double result[16], double* fitness,
                              smhip_ndt_gicp_stats* stats);
/* parity-test hooks: the working clouds of the last smhip_ndt_gicp_align in the filter's output order (which = 0
 * source, 1 target; xyz may be NULL to query the size) and the GICP covariances (n x 6 doubles: xx xy xz yy yz zz)
 * in that same order (or the upload order for smhip_gicp_align) */
smhip_status smhip_ndt_gicp_get_downsampled(smhip_handle h, int which, float* xyz, int capacity, int* n_out);
/* the GICP functor (value and 6-gradient, state x = tx ty tz roll pitch yaw) over the correspondences of the last outer
 * iteration of the last GICP run, with base_transformation_ = guess */
smhip_status smhip_gicp_evaluate(smhip_handle h, const double guess[16], const double x[6], double* f, double grad[6]);
smhip_status smhip_gicp_get_covariances(smhip_handle h, int which, double* cov, int n);

/* ---- pre_processers::filter (pre_processors/filter_*.cc): the front end's pre-filters -----------
 * Range / AxisRange / BoundingBoxRemoval / RandomSampler / VoxelGrid applied in order to one cloud on the device,
 * as filter::Factory::Filter does (filter_factory.cc:83-106).  p[] carries each filter's float parameters:
 *   RANGE                 p[0] min_range, p[1] max_range
 *   AXIS_RANGE            p[0] min, p[1] max, axis_index 0 / 1 / 2
 *   RANDOM_SAMPLER        p[0] sampling_rate; seed selects the (counter-based) random stream
 *   VOXEL_GRID            p[0] voxel_size
 *   BOUNDING_BOX_REMOVAL  p[0..2] min_x min_y min_z, p[3..5] max_x max_y max_z */
enum { SMHIP_FILTER_RANGE = 1, SMHIP_FILTER_AXIS_RANGE = 2, SMHIP_FILTER_RANDOM_SAMPLER = 3, SMHIP_FILTER_VOXEL_GRID = 4,
       SMHIP_FILTER_BOUNDING_BOX_REMOVAL = 5 };
typedef struct smhip_filter_desc {
  int32_t type;
  int32_t axis_index;
  uint32_t seed;
  int32_t reserved;
  float p[6];
} smhip_filter_desc;
/* the filter's constructor defaults / its ConfigsValid() (1 = valid) */
void smhip_filter_default(int type, smhip_filter_desc* f);
int smhip_filter_config_valid(const smhip_filter_desc* f);
/* points: n rows of stride 4 (x y z intensity: KITTI; factor = i / n as the collector sets it,
 * builder/data/data_collector.h:202-204) or 5 floats (InnerPointType).  *n_out = size of the filtered cloud. */
smhip_status smhip_filter_chain_f32(smhip_handle h, const float* points, int stride_floats, int n,
                                    const smhip_filter_desc* chain, int n_filters, int* n_out);
/* the filtered cloud as InnerPointType rows (n x 5 floats) and, per point, the row of the input it came from
 * (inliers composed over the chain; -1 after a VoxelGrid).  Either pointer may be NULL. */
smhip_status smhip_filter_get_output(smhip_handle h, float* points5, int32_t* source_index, int n);
/* the filtered cloud becomes SetInputSource of `slot` without leaving the device */
smhip_status smhip_filter_output_to_source(smhip_handle h, int slot);

/* ---- static_map::MultiResolutionVoxelMap (builder/multi_resolution_voxel_map.{h,cc}) ----------
 * The probabilistic hit / miss voxel map with ray casting behind the reference's static-map output (one
 * InsertPointCloud per frame, builder/map_builder.cc:832-900), on the device.  Results equal the reference's insert loop
 * executed in point order (its OpenMP form races on the probabilities, multi_resolution_voxel_map.cc:76-94).
 * Field names and defaults: MrvmSettings, multi_resolution_voxel_map.h:54-65. */
typedef struct smhip_mrvm_settings {
  float prob_threshold;          /* 0.6 */
  float high_resolution;         /* 0.1 m voxels */
  float hit_prob;                /* 0.55, clamped to [0.501, 0.9] (.cc:50) */
  float miss_prob;               /* 0.48, clamped to [0.1, 0.499] (.cc:51) */
  float z_offset;                /* added to the origin's z (.cc:66-67) */
  int32_t max_point_num_in_cell; /* 10 */
  int32_t use_max_intensity;     /* 1: output points carry their voxel's max intensity (.cc:160-163) */
  int32_t reserved;
} smhip_mrvm_settings;
typedef struct smhip_mrvm_context* smhip_mrvm_handle;
void smhip_mrvm_default_settings(smhip_mrvm_settings* s);
/* table_log2: the open-addressing voxel table STARTS with 2^table_log2 slots and doubles between inserts, like the reference's
 * map grows: before an insert it is made large enough for the voxels it holds plus one per point of the coming cloud to fill at
 * most half of it, up to 2^28 slots (smhip_mrvm_set_max_table_log2 lowers that; a slot costs 37 + 20 * max_point_num_in_cell
 * bytes).  Only a table that can no longer grow (the limit, or the device's memory) reports "> 70 % full" as a warning and, when
 * a voxel finds no slot, SMHIP_ERR_CAPACITY from then on (the voxel is lost, the rest of the cloud is applied).
 * max_cloud_points: the largest cloud one InsertPointCloud may hand over */
smhip_status smhip_mrvm_create(int device, int table_log2, int max_cloud_points, const smhip_mrvm_settings* settings, smhip_mrvm_handle* out);
smhip_status smhip_mrvm_destroy(smhip_mrvm_handle h);
const char* smhip_mrvm_last_error(smhip_mrvm_handle h);
void smhip_mrvm_set_offset_z(smhip_mrvm_handle h, float offset);                         /* SetOffsetZ, .cc:55-57 */
/* InsertPointCloud(cloud, origin), .cc:59-131: n rows of `stride_floats` >= 4 floats (x y z intensity [factor]: InnerPointType is
 * stride 5), origin = the sensor position of the frame.  Non-finite points are skipped.  Refused BEFORE the map is touched
 * (status != OK, map unchanged): empty cloud, cloud larger than max_cloud_points, origin not finite or beyond +-2^20 voxels.
 * Applied with a warning (status OK, text in smhip_mrvm_last_error): points beyond +-2^20 voxels are skipped (their number:
 * smhip_mrvm_last_skipped; the flag does not carry over to the next insert), a table > 70 % full that cannot grow. */
smhip_status smhip_mrvm_insert_f32(smhip_mrvm_handle h, const float* points, int stride_floats, int n, const float origin[3]);
smhip_status smhip_mrvm_last_skipped(smhip_mrvm_handle h, int* n);
smhip_status smhip_mrvm_voxel_count(smhip_mrvm_handle h, int* n);
smhip_status smhip_mrvm_set_max_table_log2(smhip_mrvm_handle h, int max_table_log2);   /* 10..28; default 28 */
int smhip_mrvm_table_log2(smhip_mrvm_handle h);                                        /* log2 of the table's current size */
/* OutputToPointCloud(threshold, PointXYZI cloud) without averaging, .cc:133-170: rows x y z intensity of every stored point of
 * every voxel with probability byte >= uint8(threshold * 256), in no particular order (the reference iterates an unordered
 * map).  capacity = 0 only counts. */
smhip_status smhip_mrvm_output(smhip_mrvm_handle h, float threshold, float* xyzi, int capacity, int* n_out);
/* Both OutputToPointCloud overloads with MrvmSettings::output_average, .cc:125-216.  flags: SMHIP_MRVM_AVERAGE = one row per
 * voxel, the float mean of its stored points (summed in their order, divided by float(size)); its 4th column is the voxel's max
 * intensity, or 0 without use_max_intensity (the reference never assigns it).  SMHIP_MRVM_RGB = the PointXYZRGB overload: the
 * 4th column holds the bits of the packed colour a << 24 | r << 16 | g << 8 | b with r = g = b = min(255, uint32(max_intensity * 1.4))
 * and a = 255 (pcl::PointXYZRGB's `rgb` float; the reference assigns r, g, b only, so the alpha byte is what the PCL >= 1.8
 * constructor puts there). */
#define SMHIP_MRVM_AVERAGE 1
#define SMHIP_MRVM_RGB 2
smhip_status smhip_mrvm_output_ex(smhip_mrvm_handle h, float threshold, int flags, float* rows, int capacity, int* n_out);
/* parity-test hook: every voxel of the map -- key (3 ints), probability byte, max intensity, number of stored points and
 * the points themselves (max_point_num_in_cell x 5 floats per voxel), in no particular order */
smhip_status smhip_mrvm_dump(smhip_mrvm_handle h, int32_t* keys3, uint8_t* prob, int32_t* max_intensity, int32_t* npoints, float* points5,
                             int capacity, int* n_out);

/* ---- profiling ----------------------------------------------------------- */
/* enable: 0 off, 1 events around every launch; events around ONE kernel class only, cheap enough to leave on inside a timed
 * region: 2 the NN kernels proper (fused search / full walk, certificate pass), 3 accumulate, 4 the listed search */
smhip_status smhip_icp_enable_profile(smhip_handle h, int enable);
smhip_status smhip_icp_get_profile(smhip_handle h, smhip_icp_profile* out);
/* queries that went through a search (exact modes: certificate failed; NABO: walked again) in iterations 0..11 of the
 * slot's last Align; counts[12] */
smhip_status smhip_icp_get_search_counts(smhip_handle h, int slot, uint32_t* counts);

#ifdef __cplusplus
}
#endif
#endif /* SMHIP_H_ */
