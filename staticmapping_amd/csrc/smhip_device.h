// smhip_device.h -- device-side data layout shared by the HIP kernels and the host shim.
// MI355X / gfx950 only (wave64, 160 KiB LDS, 8 XCDs); no portability layer on purpose.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smhip {

constexpr int kHistBins = 2048;          // level-1 histogram of d2 keys: float bits >> 20 (8 exp + 3 mantissa bits)
constexpr int kHistShift = 20;
constexpr int kSearchHist = 12;          // iterations whose searched-query counts are kept per pair
constexpr int kNnThreads = 256;          // one query per thread, 4 waves per workgroup
constexpr int kAccThreads = 256;
// source points per thread in the accumulate kernel.  Every workgroup ends with a 29-column f64 reduction that costs about
// as many vector instructions as 8 points, so batches use long chunks (32 points per thread: 105 -> 62 us per 64 pairs);
// a single pair keeps short ones, or it would have 15 workgroups for 256 CUs.
constexpr int kAccItemsSmall = 8;
constexpr int kAccItemsBatch = 32;
constexpr int kFinalizeKeyCap = 8192;     // band records whose keys finalize keeps in LDS between its radix-select passes
constexpr int kFinalizeMaxSeg = 2048;     // record segments (one per producing wave) per pair finalize can index: 4 Mi source points at 32 per
                                          // accumulate thread; the fused path (kCertifyItems = 32 per thread, four segments per workgroup, + the listed search's
                                          // kListedMaxItems = 129 items: fused_iteration() checks the sum) up to (2048 - 129) / 4 workgroups of 8 192 = 3.9 Mi points
constexpr int kScoreParts = 16;          // workgroups per pair of final_score
constexpr int kAccCols = 32;             // 21 (A upper) + 6 (b) + 1 (unused: the score is formed once, by final_score) + 1 (count) padded to 32
constexpr int kMaxGridWords = 1 << 18;   // 32-cell words per pair (8 Mi cells)
constexpr int kMaxRowWords = (kMaxGridWords >> 5) + 2;   // 32-row words of the row-occupancy bitmap (rows = ny * nz <= kMaxGridWords), + slack for two-word reads
constexpr int kTgtReduceBlocks = 32;     // partial blocks for the target mean / bbox
constexpr int kBruteTile = 1024;         // target points staged in LDS per tile of the brute-force kernels (16 KiB as float4)
constexpr int kFallbackSlices = 64;      // target slices the fallback sweep is spread over
constexpr int kLdsTableCap = 1024;        // u32 entries of the per-workgroup row table in nn_ball_lds (4 KiB)
constexpr int kLdsPointCap = 512;         // target points staged per round (8 KiB)
constexpr int kLdsRowCap = 256;           // grid rows of the box whose runs are staged (<= workgroup size)
constexpr int kWideBlocks = 512;         // workgroups (4 waves = 4 queries at a time) per pair of nn_ring_wide
constexpr int kNaboAccBlocks = 8;         // workgroups per pair of accumulate_listed (reference-search mode, fused path)
constexpr int kListedMaxPairs = 1024;   // pairs per launch of the balanced listed search (nn_ball_listed_items: its plan sits in LDS)
constexpr int kListedItemBlocks = 1280;  // its workgroups: 5 per CU (LDS), each takes an equal run of the launch's items
constexpr int kListedBlocks = 32;        // workgroups per pair of the listed search (nn_ball_listed): it strides over the list
constexpr int kBallItems = 2;            // rounds of 256 queries per workgroup in nn_ball_lds / nn_ball (prefetch across rounds; 2 measured best: 4 = +8 %, 1 = +5 %)
constexpr int kListedLaneBudgetMax = 32768;   // largest lane budget of the balanced listed search (SMHIP_LISTED_LANES; default 1024, sync_options)
constexpr int kFusedListedMax = 16384;   // fused path: with more failing certificates than this in a pair and iteration the sums are left to `accumulate`
                                         // (their listed matches below the band are summed kListedSumChunk entries per work item of iteration_sums,
                                         // a row of partials each)
// most work items listed_plan can cut a pair's list into (= record segments of region 1 finalize has to index): a list of `count`
// entries searched with L lanes each (L * count <= the budget, or L = 1) makes ceil(count * L / 256) items
constexpr int kListedMaxItems = (kFusedListedMax / kNnThreads > kListedLaneBudgetMax / kNnThreads ? kFusedListedMax / kNnThreads : kListedLaneBudgetMax / kNnThreads) + 1;
constexpr int kListedSumChunk = 4096;    // fused path: listed queries per work item of iteration_sums (a row of partials each: at most kFusedListedMax / this)
constexpr int kSumsBlocks = 2048;        // its workgroups (a multiple of 8).  768 (three per CU at 164 VGPRs: all resident) / 1 536 / 2 048 / 3 072: 25.5 / 26.3 / 26.0 /
                                         // 25.8 k alignments/s on the bench batch, 17.8 / 18.2 / 18.5 / 18.5 k mixed: short items, so a second round of workgroups balances them
constexpr int kCertifyItems = 32;        // rounds per workgroup of the certificate pass: its per-workgroup costs (histogram zero + flush, pipeline fill, the
                                         // fused pass's 29-column reduction) are large next to a round's.  The plain pass (round 3): 8: 80, 12: 57, 16: 57,
                                         // 20: 49, 24: 48, 28: 52, 32: 53 us per 64 pairs (120 k points); the fused pass with the shadow (round 5), per
                                         // 512-pair step: 16: 8.41, 20: 8.21, 32: 7.97, 40: 7.98 ms (at most 32: one bit per round in nabo_list_append)

// the single-pair persistent kernel (icp_one.hip)
constexpr int kOneMaxRounds = 16;        // rounds of 256 points a workgroup may own (sizes the LDS list of its in-bin points: 16 KiB)
constexpr int kOneSyncWords = 4096;      // IcpDev::one_sync: the grid barrier's counters and flags + the length of the key list, a 128-byte line each; zeroed per Align
constexpr int kOneMaxBlocks = 1024;      // the largest grid (workgroups per pair) of the kernel
constexpr int kOnePairs = 8;             // pairs one launch of it can hold (grid.y; each with its own barrier lines and rows)

// Per-pair device state.  Everything an iteration needs and everything the host reads back.
struct PairState {
  // transforms, row-major
  double M[12];          // T_iter * G : applied to the raw source every iteration
  double M_prev[12];     // the same of the previous iteration (the global-memory search variants size their search margin with it)
  // How far a source point s can have moved since a certificate bound was recorded: |M_n s - M_k s| <= (pot_a[n] - pot_a[k]) |s|
  // + (pot_b[n] - pot_b[k]), with pot_a = the sum over the iterations so far of ||dR||_F and pot_b of |dt| (finalize).  A bound L
  // recorded at iteration k is stored as L + pot_a[k] |s| + pot_b[k]; at iteration n it is worth that minus pot_a[n] |s| +
  // pot_b[n] -- nothing to rewrite while a query stays certified, and no second transform to measure its motion.
  double pot_a, pot_b;
  double step_a, step_b; // the last iteration's two norms alone (how far a query can have moved in THIS iteration: search margin)
  double T_iter[16];
  double G[16];          // T(-mu) * guess
  double mu[3];          // target mean (icp_fast.cc:457-458)
  double guess[16];      // row-major copy of the caller's guess
  // CheckConvergence history (icp_fast.cc:377-405): last 5 entries
  double quat[5][4];
  double trans[5][3];
  int32_t n_hist;
  // loop control
  int32_t iter;
  int32_t done;
  int32_t status;
  int32_t ns;
  int32_t nt;
  int32_t has_normals;
  // grid geometry (bit-rank voxel grid over the centred target)
  float origin[3];
  float h;
  float inv_h;
  int32_t nx, ny, nz, wx, nw;
  int32_t nocc;
  // selection / lists
  uint32_t unresolved_count;
  uint32_t fallback_ticket;  // slices of nn_fallback that have finished (the last one writes the results back)
  uint32_t fallback_total;
  uint32_t searched_total;   // queries that went through a search, summed over iterations
  uint32_t deferred_count;   // queries whose certificate failed (nn_certify), searched by nn_ball_listed
  uint32_t hard_count;       // queries nn_ball recorded with a lower bound this iteration
  uint32_t hard_total;
  uint32_t min_lb_key;       // smallest such bound (float bits)
  int32_t refine;            // 1 = this iteration's bounds must be refined to exact matches
  uint32_t refine_total;
  float rcap2;               // squared search-radius cap of nn_ball for the next iteration
  int32_t kept;
  uint32_t limit_key;
  uint32_t nabo_count[4];    // SMHIP_NN_NABO: queries to walk again this iteration, by the work class of their last walk (nn_certify<., true>)
  int32_t grid_invalid;      // 1 = grid_setup found a non-finite target box: no search structure was built for this target, and
                             //     every Align on it fails (pose_setup re-asserts it when the structure is "kept")
  // Fused certificate pass + sums (nn_certify_acc): finalize predicts the histogram bins the NEXT iteration's quantile will fall
  // in, [band_lo, band_hi] (band_lo = 0: no prediction).  The fused pass sums the matches below the band on the spot and leaves
  // the band's members as records; nn_validate, once every distance is known, sets spec_ok when the quantile's bin did land in
  // the band (and nothing had to be refined) -- otherwise `accumulate` redoes the iteration's sums the plain way.
  int32_t band_lo, band_hi;
  int32_t spec_ok;
  uint32_t spec_hits;        // iterations whose sums came from the fused pass (statistics)
  uint32_t listed_ticket;    // (the launch's first pair) next unclaimed item of nn_ball_listed_items; listed_plan resets it
  uint32_t pad2_;

  // outputs
  double score_part[kScoreParts];   // final_score: the sums of sqrt(d2) over the kept matches of the pair's kScoreParts point ranges
  uint32_t score_cnt[kScoreParts];  // ... and how many matches each range summed (score_fold divides by their total)
  uint32_t score_mismatch;          // 1 = that total differed from finalize's kept count (never expected; tests look at it)
  uint32_t pad_score_;
  double score;
  double result[16];     // column-major (Eigen layout)
};

// Host-written per-pair inputs of one Align call.
struct PairInput {
  double guess[16];      // row-major
  int32_t ns;
  int32_t nt;
  int32_t has_normals;
  int32_t pad;
};

// Pointers + capacities handed to every kernel by value.
struct IcpDev {
  int32_t slots, ns_cap, nt_cap;
  int32_t npairs;            // pairs this launch covers (XCD-aware kernels pad the grid to a multiple of 8)
  int32_t pair_base;         // first pair slot of this launch (the batch is split over two streams)
  int32_t acc_blocks;        // ceil(ns_cap / (kAccThreads * kAccItemsSmall))
  int32_t part_stride;       // rows of `partials` per pair (= acc_blocks: accumulate with short chunks makes the most)
  int32_t bl_stride;         // record slots per region and pair: ns_cap rounded up to whole accumulate AND certificate-pass chunks
  int32_t dl_stride;         // dlist entries per pair as the fused path indexes it: ns_cap rounded up to a whole certificate-pass chunk
  int32_t seg_stride;        // gcount / dcount entries per pair
  int32_t fused;             // 1 = this iteration's launches belong to the fused path (nn_certify_acc + nn_ball_listed_items):
                             //     nn_validate decides spec_ok, accumulate returns at once when it holds, finalize reads either form
  float band_pad;            // half-width of the predicted band in bins, at least (tuning; default 0.1)
  int32_t fused_nabo;        // the fused path's reference-search form (nn_certify_acc<., true> + nn_nabo + accumulate_listed)
  int32_t listed_lane_budget; // lanes a pair's listed search may spread its queries over (sets the lanes per query of nn_ball_listed_items)
  int32_t listed_grain;      // items a workgroup of nn_ball_listed_items claims at a time (0: equal runs fixed in advance)
  float band_gain;           // ... and this many times the quantile's last move (default 1.5)
  int32_t acc_items;         // points per thread of the accumulate launches of this batch part (finalize folds accordingly)
  int32_t sums_items;        // fused iterations: points per thread of the blocks iteration_sums cuts a missed pair into (this launch)
  PairState* state;
  const PairInput* in;
  const float4* src;         // [slots][ns_cap] raw source xyz, w = caller index bits (uploads, the prep / filter / NDT / GICP kernels)
  const float* src3;         // [slots][ns_cap][3] the same points packed to 12 bytes: what the ICP iteration kernels stream
                             //                 (20 iterations x 2 passes read it; packed by pack_source when a slot's source changed)
  const float4* tgt_p;       // [slots][nt_cap] raw target xyz
  const float4* tgt_n;       // [slots][nt_cap] raw target normals
  float4* tq;                // [slots][nt_cap] centred target, cell-sorted; w = original index bits
  float4* tn;                // [slots][nt_cap] normals, same order
  uint32_t* tcell;           // [slots][nt_cap] (word << 5) | bit
  uint32_t* tslot;           // [slots][nt_cap] occupied-cell slot
  uint32_t* tord;            // [slots][nt_cap] ordinal inside the cell
  uint32_t* bits;            // [slots][kMaxGridWords]
  uint2* words;              // [slots][kMaxGridWords] {occupancy bits, exclusive rank}
  uint32_t* rowbits;         // [slots][kMaxRowWords] bit (z * ny + y) = grid row (y, z) holds at least one point (built for the ring searches)
  int32_t have_rowbits;      // 1 = rowbits describes the resident grid
  uint32_t* ccount;          // [slots][nt_cap + 1]
  uint32_t* cstart;          // [slots][nt_cap + 1]
  float* lb;                 // [slots][ns_cap] > 0: every target point other than the match is at least this far;
                             //                 < 0: no exact match, EVERY target point is at least -lb away; 0: unknown
  float* d2;                 // [slots][ns_cap]
  int32_t* idx;              // [slots][ns_cap] index into tq/tn (sorted order)
  uint32_t* mb;              // [slots][ns_cap] the 4-byte shadow of (idx, lb) the fused certificate pass streams instead of the two: 15 bits of
                             //                 match (0x7fff: none) + the bound as a 17-bit float whose magnitude is rounded TOWARDS ZERO (a smaller
                             //                 bound certifies less, never wrongly).  Every writer of idx / lb writes it (st_match); it is read only
                             //                 by launches whose targets all have fewer than 32 767 points
  uint32_t* hist;            // [slots][kHistBins]
  int32_t* dlist;            // [slots][ns_cap] deferred queries (searched by nn_ball_listed)
  int32_t* hlist;            // [slots][ns_cap] queries the tile phase could not certify (ring search)
  int32_t* ulist;            // [slots][ns_cap] unresolved queries (brute-force fallback)
  unsigned long long* ukeys; // [slots][ns_cap] fallback winners: (d2 bits << 32) | original target index
  // Band records: the matches whose d2 falls in the histogram bins the quantile can fall in (for `accumulate`, which runs with
  // the quantile's bin known: that bin alone), left for finalize with everything it needs of them -- source point, d2, match --
  // by the wave that met them, in its own segment, in the order it met them: no atomics, an order fixed by the data.
  // Region 0 ([0, bl_stride)): the segments of accumulate's or the fused pass's waves (64 * points-per-thread slots each);
  // region 1 ([bl_stride, 2 bl_stride)): the segments of the listed search's waves (bl_stride / 128 slots each).
  float4* rec_a;             // [slots][2 bl_stride] {source x, y, z, d2}
  int32_t* rec_j;            // [slots][2 bl_stride] matched target position
  uint32_t* gcount;          // [slots][seg_stride] records per segment
  int32_t* dcount;           // [slots][seg_stride] fused path: queries whose certificate failed, per wave of nn_certify_acc (its segment of
                             //                 dlist: dl_stride entries per pair, 64 * kCertifyItems slots per wave); listed_plan turns
                             //                 the counts into exclusive offsets with the total behind them
  uint32_t* litems;          // [slots] fused path: work items (one workgroup pass each) of the pair's listed search (listed_plan)
  double* partials;          // [slots][part_stride][kAccCols]: one row per workgroup of accumulate / of the fused certificate pass
  double* tpart;             // [slots][kTgtReduceBlocks][16]
  uint32_t* done_count;      // number of finished pairs
  uint32_t* one_sync;        // [kOnePairs][kOneSyncWords] grid barrier + key-list length of the persistent kernel (icp_one), per pair of its launch
  double* one_rows;          // [kOnePairs][kOneMaxBlocks + 32][kAccCols] its workgroups' rows of sums, then its groups'
  uint32_t* one_hist;        // [kOnePairs][kHistBins] its own level-1 histogram (cumulative inside a launch)
  uint32_t* one_keys;        // [kOnePairs][2 bl_stride] its key lists (two parities)
  PairState* one_ctr;        // [kOnePairs] the counters its workgroups add to (hard_count, deferred_count, unresolved_count)
                             // -- everything of icp_one that crosses workgroups lies in FINE-GRAINED device memory (one allocation,
                             //    hipDeviceMallocFinegrained): the XCDs' L2s do not keep such lines, so an agent-scope load behind a
                             //    barrier cannot be served a copy from before it
  uint8_t* nabo_work;        // [slots][ns_cap] SMHIP_NN_NABO: buckets the query's last walk scanned (capped at 255); null until the mode is used
  uint32_t* search_hist;     // [slots][kSearchHist] queries that needed a search in iteration k of the last Align (finalize; read back by the
                             //                 host to place the switch from the fused search to certify + listed search, split_after = 0)
  // options
  int32_t max_iteration;
  int32_t early_exit;
  int32_t max_ring;
  float nn_cutoff2;          // > 0: the ring searches may stop once every unseen point is provably farther than this (squared)
  int32_t sort_cells;        // 1 = order points inside a cell by caller index (deterministic tie rule)
  int32_t use_ball;          // 1 = ball-bounded search with certified trimming; 0 = ring search over every query
  int32_t lds_table;         // 1 = nn_ball_lds (row tables staged in LDS), 0 = nn_ball (global lookups)
  int32_t certify;           // 1 = iterations >= 1 run nn_certify and search only the queries whose certificate fails
  int32_t split_after;       // iterations >= this run the certificate pass and the listed search as two launches
  int32_t exact_all;         // 1 = every match exact (no lower bounds survive), e.g. find_closests
  float ball_radius;         // largest search radius of nn_ball (first iteration / clamp)
  float cap_factor;          // next cap = cap_factor x quantile distance
  float rho;                 // dist_outlier_ratio as float (widened to double exactly like the reference)
  float grid_cell;
  int32_t debug_flags;       // diagnostic builds only (SMHIP_DEBUG_FLAGS with -DSMHIP_PHASE_TIMING=1): bit 4 = per-phase timing of nn_ball_lds
};

}  // namespace smhip
