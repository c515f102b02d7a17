// icp_one.hip -- the single-pair IcpFast::Align as ONE cooperative launch.
//
// The reference's front end aligns one scan against one key frame at a time (builder/map_builder.cc:317-333; sequential by
// :302-307, 379-392), so the call that matters there is a single `Align`, and on the device a single Align is latency, not
// bandwidth: 120 000 source points are 2.4 MB, the target 0.7 MB.  As separate launches an iteration is four of them (search,
// refinement check, sums, finalize: 37 + 5 + 11 + 28 us, profiles/r05_single_pair_kernel_stats.txt) with one workgroup doing the
// whole of `finalize`.  Here the whole loop of icp_fast.cc:484-523 is one kernel whose workgroups stay resident (cooperative
// launch: the grid is sized to what the device holds at once) and meet at grid barriers; everything a workgroup can compute from
// what all of them published it computes itself instead of waiting for one workgroup to do it and a barrier to hand it round:
//
//   S   every workgroup: the rounds of nn_ball_lds it owns (certificate, LDS-staged ball search of the failing queries), its
//       histogram added to the pair's                                                                    -- barrier 1 --
//   V   every workgroup: the quantile's bin from the pair's histogram, the check of the lower bounds against it (nn_validate);
//       in the rare iteration that must refine bounds: every workgroup refines its own queries' (ring search, then a sweep of the
//       target for what the rings leave open), one more barrier, the bin again
//   A   every workgroup: the sums of its points below the bin; the keys of its points inside the bin appended to the pair's key
//       list (ONE returning atomic per workgroup), the points themselves remembered in LDS                -- barrier 2 --
//   F1  every workgroup: the exact rank inside the bin by radix select over the key list (~2 000 keys: the whole list sits in
//       LDS) -- the same list in every workgroup, in whatever order it was appended: the selected VALUE does not depend on it --
//       then its own in-bin points at or below the limit added to its sums; its row of 29 sums published   -- barrier 3 --
//   F2  every workgroup: the rows folded in a fixed order, the 6x6 solve, the pose update, CheckConvergence (finalize_tail) on
//       its OWN copy of the pair's state in LDS.  Identical inputs, identical instruction sequence: identical new pose in every
//       workgroup, no barrier before the next iteration's S.  Workgroup 0 alone writes the state back to global memory.
//
// Matches, distances, histogram, quantile and kept set are those of the separate kernels bit for bit (the same bodies:
// ball_lds_rounds, find_quantile_bin, ring_body, fallback_body, accumulate_terms, finalize_tail); the 29 sums are added in a
// different -- fixed -- order, so poses agree to ~1e-15 and a run is reproducible bit for bit.
//
// Coherence without cache-wide fences.  The eight XCDs' L2s are not coherent with each other, and an agent-scope release /
// acquire pair is a write-back plus an invalidate of a whole L2 by every wave: as first built -- fences at every barrier -- a
// barrier of 472 workgroups cost 45-65 us and every load behind it missed (6.8 ms per Align).  Instead every word that crosses
// workgroups is named: it is written with agent-scope (write-through) stores or atomics and read with agent-scope loads, a wave
// waits for its own such stores to complete (s_waitcnt) before its workgroup arrives at the barrier, and nothing else is flushed
// or invalidated.  A workgroup's matches, distances and bounds (written in S, read in A and F1 by the SAME workgroup: one CU,
// one L1, one L2) and the read-only target stay in the caches.  The rows of sums -- 472 x 232 bytes read by every workgroup --
// go to a FRESH address range each iteration (IcpDev::one_rows, 16 MiB): a line nobody has read since the kernel began is in no
// L2, so plain loads are safe and each XCD fetches the rows once for its 59 workgroups (when the range wraps, an acquire fence).
// The rare refinement of lower bounds is done by the queries' owners for the same reason (the separate launches spread the list
// of lower-bounded queries over all workgroups, which then write other workgroups' matches).
// Barrier: one returning arrival per workgroup on a monotonic counter; the last arrival publishes the epoch in a second word, the
// others spin on that word (not on the counter the arrivals queue on).
#include <cstddef>
#include "smhip_device.h"

namespace smhip {

// a word published for other workgroups by an exchange whose old value is waited for: the write has been PERFORMED where every XCD
// sees it before the wave goes on to its barrier (a store's completion count says it was accepted, not where it stands against a
// returning atomic on another channel)
__device__ __forceinline__ void pub_dev(uint32_t* p, uint32_t v) {
  const uint32_t o = __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" :: "v"(o));
}
__device__ __forceinline__ void pub_dev(double* p, double v) {
  const unsigned long long o = __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" :: "v"(o));
}
#ifndef SMHIP_ONE_SCOPE
#define SMHIP_ONE_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
#if defined(SMHIP_ONE_RMW_READS)
__device__ __forceinline__ uint32_t ld_dev(const uint32_t* p) { return __hip_atomic_fetch_or(const_cast<uint32_t*>(p), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
__device__ __forceinline__ uint32_t ld_dev(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, SMHIP_ONE_SCOPE); }
#endif
__device__ __forceinline__ unsigned long long ld_dev64(const double* p) {
#if defined(SMHIP_ONE_RMW_READS)
  return __hip_atomic_fetch_or(reinterpret_cast<unsigned long long*>(const_cast<double*>(p)), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, SMHIP_ONE_SCOPE);
#endif
}
__device__ __forceinline__ void st_dev(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, SMHIP_ONE_SCOPE); }
__device__ __forceinline__ void st_dev(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, SMHIP_ONE_SCOPE);
}

// IcpDev::one_sync, in 128-byte lines (32 words): [0] arrivals of the groups' last workgroups, [32] length of the key list,
// [kSyncArrive + 32 g] arrivals of group g's workgroups, [kSyncFlag + 32 g] the last completed epoch as group g polls it.  Group =
// blockIdx % NG; measured (tools/grid_barrier_probe.hip, 472 workgroups): every arrival on one counter
// 6.2 us per barrier -- same-address atomics complete one every ~13 ns --, two levels 2.1 us.
constexpr int kOneMaxGroups = 32;
constexpr int kSyncArrive = 64, kSyncFlag = kSyncArrive + 32 * kOneMaxGroups, kSyncCount = kSyncFlag + 32 * kOneMaxGroups, kSyncTotal = kSyncCount + 32 * kOneMaxGroups;
constexpr int kSyncKeys = kSyncTotal + 32;        // + 32 p: length of the key list of the iterations of parity p
constexpr int kSyncMinLb = kSyncKeys + 64;        // + 32 p: ~(the smallest lower bound recorded) of the iterations of parity p (0: none)
constexpr int kSyncAbort = kSyncMinLb + 64;       // != 0: a barrier timed out -- every workgroup leaves; [+1..+7] what the first to notice saw
static_assert(kSyncAbort + 32 <= kOneSyncWords, "IcpDev::one_sync holds the barrier's lines");
struct OneGrid { uint32_t G, NG; };       // workgroups, groups (a power of two that divides G)
__device__ __forceinline__ void one_arrive_and_wait(uint32_t* sync, uint32_t epoch, OneGrid og, bool group_last, uint32_t* s_abort) {
  const uint32_t g = blockIdx.x & (og.NG - 1);
  if (group_last) {
    const uint32_t old = __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == epoch * og.NG)
      for (uint32_t k = 0; k < og.NG; ++k) st_dev(&sync[kSyncFlag + 32 * k], epoch);
  }
  // (a watchdog instead of an endless spin: a barrier that has not completed after 0.2 s -- a thousand times the longest
  // iteration -- raises the launch's abort word with what it saw; every workgroup looks at that word while it spins and leaves, the
  // pair ends with status SMHIP_ERR_HIP: an error at the host, not a hang and not a dead context)
  const unsigned long long t0 = wall_clock64();
  uint32_t polls = 0;
  while (ld_dev(&sync[kSyncFlag + 32 * g]) < epoch) {
    __builtin_amdgcn_s_sleep(1);
    if ((++polls & 0x3ffu) == 0u) {
      if (ld_dev(&sync[kSyncAbort]) != 0u) { *s_abort = 1u; return; }
      if (wall_clock64() - t0 > 20000000ull) {
        if (__hip_atomic_exchange(&sync[kSyncAbort], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
          st_dev(&sync[kSyncAbort + 1], (uint32_t)blockIdx.x); st_dev(&sync[kSyncAbort + 2], epoch);
          st_dev(&sync[kSyncAbort + 3], ld_dev(&sync[kSyncArrive + 32 * g])); st_dev(&sync[kSyncAbort + 4], epoch * (og.G / og.NG));
          st_dev(&sync[kSyncAbort + 5], ld_dev(&sync[0])); st_dev(&sync[kSyncAbort + 6], epoch * og.NG);
        }
        *s_abort = 1u;
        return;
      }
    }
  }
}
__device__ __forceinline__ void one_grid_sync(uint32_t* sync, uint32_t& epoch, OneGrid og, uint32_t* s_abort) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // this wave's write-through stores and atomics have completed
  __syncthreads();
  ++epoch;
  if (threadIdx.x == 0) {
    const uint32_t g = blockIdx.x & (og.NG - 1);
    const uint32_t old = __hip_atomic_fetch_add(&sync[kSyncArrive + 32 * g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    one_arrive_and_wait(sync, epoch, og, old + 1u == epoch * (og.G / og.NG), s_abort);
  }
  __syncthreads();
}
// The barrier with five counts riding on it (the fused form: how many distances in all, below the predicted band, in each of its
// bins): integers, so atomics add them in any order to the same totals.  Two levels like the arrivals: a workgroup adds its counts
// to its group's line, the group's last arrival moves the line's contents (exchanged for zero) to the top line, which is never
// cleared -- every workgroup remembers what it last read there (cprev, threads 0..4) and takes the difference.
// sync[kSyncCount + 32 g + t]: group g's counts, sync[kSyncTotal + t]: the totals.
__device__ __forceinline__ void one_grid_sync_counts(uint32_t* sync, uint32_t& epoch, OneGrid og, const uint32_t* s_cnt, uint32_t* s_out, uint32_t& cprev,
                                                    uint32_t* s_flag, uint32_t* s_abort) {
  const uint32_t G = og.G;
  const uint32_t g = blockIdx.x & (og.NG - 1);
  uint32_t sink = 0;
  if (threadIdx.x < 5 && s_cnt[threadIdx.x]) sink = __hip_atomic_fetch_add(&sync[kSyncCount + 32 * g + threadIdx.x], s_cnt[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" :: "v"(sink) : "memory");      // (the old values have come back: the additions are done)
  __syncthreads();
  ++epoch;
  if (threadIdx.x == 0) {
    const uint32_t old = __hip_atomic_fetch_add(&sync[kSyncArrive + 32 * g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_flag = old + 1u == epoch * (G / og.NG) ? 1u : 0u;
  }
  __syncthreads();
  const bool last = *s_flag != 0u;
  if (last) {
    if (threadIdx.x < 5) {
      const uint32_t v = __hip_atomic_exchange(&sync[kSyncCount + 32 * g + threadIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sink = v ? __hip_atomic_fetch_add(&sync[kSyncTotal + threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" :: "v"(sink) : "memory");
    __syncthreads();
  }
  if (threadIdx.x == 0) one_arrive_and_wait(sync, epoch, og, last, s_abort);
  __syncthreads();
  if (threadIdx.x < 5) {
    const uint32_t cur = ld_dev(&sync[kSyncTotal + threadIdx.x]);
    s_out[threadIdx.x] = cur - cprev;
    cprev = cur;
  }
  __syncthreads();
}
// The barrier behind the rows of sums, with the first level of their fold inside it: the last workgroup of a group to arrive adds
// the group's rows -- in the order of their workgroups, whoever arrives last -- and publishes the group's row before it arrives
// for the group.  (Every workgroup reading all 472 rows would pull 57 MB through the fabric per iteration.)
__device__ __forceinline__ void one_grid_sync_fold(uint32_t* sync, uint32_t& epoch, OneGrid og, const double* rows, double* grows,
                                                  double (*s_part)[32], uint32_t* s_flag, uint32_t* s_abort) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  ++epoch;
  const uint32_t G = og.G;
  const uint32_t g = blockIdx.x & (og.NG - 1);
  if (threadIdx.x == 0) {
    const uint32_t old = __hip_atomic_fetch_add(&sync[kSyncArrive + 32 * g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_flag = old + 1u == epoch * (G / og.NG) ? 1u : 0u;
  }
  __syncthreads();
  const bool last = *s_flag != 0u;
  if (last) {
    const int col = threadIdx.x & 31, sub = threadIdx.x >> 5;
    double s = 0;
#pragma unroll 8
    for (uint32_t w = g + og.NG * sub; w < G; w += og.NG * 8)
      s += __longlong_as_double((long long)ld_dev64(rows + (size_t)w * kAccCols + col));
    s_part[sub][col] = s;
    __syncthreads();
    if (threadIdx.x < kAccCols) {
      double t = 0;
      for (int k = 0; k < 8; ++k) t += s_part[k][threadIdx.x];
      pub_dev(&grows[(size_t)g * kAccCols + threadIdx.x], t);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (threadIdx.x == 0) one_arrive_and_wait(sync, epoch, og, last, s_abort);
  __syncthreads();
}
__device__ __forceinline__ double uniform_f64(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// the serial tail as a CALL: inlined, its few hundred live doubles (the 6x6 factorisation, the pose chain) set the register
// allocation of the whole kernel -- 256 VGPRs and 89 spilled ones in the loops every thread runs
__device__ __noinline__ void one_tail(TailOpts o, PairState* ls, int pair, const double* s_tot, uint32_t n_valid, uint32_t limit_key, int ns, bool publish) {
  (void)finalize_tail(o, ls, pair, s_tot, n_valid, limit_key, false, ns, publish);
}

#ifndef SMHIP_ONE_TIMING
#define SMHIP_ONE_TIMING 0
#endif
// diagnostic build (-DSMHIP_ONE_CHECKS=1): every index the kernel takes from memory is checked against its range before it is used
#ifndef SMHIP_ONE_CHECKS
#define SMHIP_ONE_CHECKS 0
#endif
#if SMHIP_ONE_CHECKS
// every workgroup's barriers, in order: (index of the barrier it is about to enter, iteration, which barrier of the code, a value)
// in the pair's (otherwise unused) rec_a array, 64 entries per workgroup; the host prints them when the pair stops itself
#define SMHIP_OTRACE(code, val) do { if (threadIdx.x == 0 && target < 64u) reinterpret_cast<uint4*>(b.rec_a + (size_t)pair * 2 * b.bl_stride)[(size_t)blockIdx.x * 64 + target] = make_uint4(target + 1u, (uint32_t)ls.iter, (uint32_t)(code), (uint32_t)(val)); } while (0)
#define SMHIP_OCHK(cond, what, v) do { if (!(cond)) { printf("[icp_one] CHECK %s failed: value %lld (pair row %u workgroup %u thread %u iteration %d)\n", what, (long long)(v), (unsigned)blockIdx.y, (unsigned)blockIdx.x, (unsigned)threadIdx.x, ls.iter); __builtin_trap(); } } while (0)
#else
#define SMHIP_OTRACE(code, val) do { } while (0)
#define SMHIP_OCHK(cond, what, v) do { } while (0)
#endif
#if SMHIP_ONE_TIMING
#define SMHIP_OPH(k) do { if (otime) { const unsigned long long now_ = wall_clock64(); oacc[k] += now_ - oprev; oprev = now_; } } while (0)
#else
#define SMHIP_OPH(k) do { } while (0)
#endif

__global__ __launch_bounds__(kNnThreads, 1) void icp_one(IcpDev b, int groups) {   // (one workgroup per CU: 305 registers, none spilled; at two per CU 39 spilled in the loops every thread runs)
  const int pair = b.pair_base + (int)blockIdx.y;          // (a launch holds up to kOnePairs pairs: a row of the grid each, nothing shared)
  const uint32_t G = gridDim.x;
  const OneGrid og = {G, (uint32_t)groups};
  PairState* st = &b.state[pair];
  __shared__ PairState ls;
  __shared__ uint32_t s_hist[kHistBins];
  __shared__ uint32_t s_w[17];
  __shared__ uint32_t s_q[4];
  __shared__ uint32_t s_h[1024];
  __shared__ uint32_t s_sel[2];
  __shared__ double s_red[4][29];
  __shared__ double s_tot[kAccCols];
  __shared__ uint32_t s_cnt[16];                         // the fused form's five counts of this workgroup, [8..12] the pair's
  __shared__ double s_part[8][32];
  __shared__ uint32_t s_keys[kFinalizeKeyCap];           // the key list (select); the refinement's target tile (V)
  __shared__ int s_rec[kNnThreads / 64][64 * kOneMaxRounds];
  __shared__ uint32_t s_wc[kNnThreads / 64];
  __shared__ uint32_t s_misc[8];
  __shared__ uint32_t s_abort;                           // a barrier of this launch timed out (see one_arrive_and_wait): leave
  static_assert(sizeof(float4) * kBruteTile <= sizeof(uint32_t) * kFinalizeKeyCap, "the fallback's tile aliases the key list");
  {
    const uint32_t* g = reinterpret_cast<const uint32_t*>(st);
    uint32_t* l = reinterpret_cast<uint32_t*>(&ls);
    for (int k = threadIdx.x; k < (int)(sizeof(PairState) / 4); k += kNnThreads) l[k] = g[k];
  }
  if (threadIdx.x == 0) s_abort = 0u;
  __syncthreads();
  if (ls.done) return;                                   // (pose_setup: a target without a search structure) -- every workgroup alike
  // behind every barrier: did it complete?  If not the pair ends here, failed (workgroup 0 says so in the pair's state)
  auto bail = [&]() -> bool {
    if (!s_abort) return false;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      st->status = 3;                                    // SMHIP_ERR_HIP
      st->done = 1; st->score = 0; st->iter = ls.iter;
      for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) st->result[4 * c + r] = ls.guess[4 * r + c];
      atomicAdd(b.done_count, 1u);
    }
    return true;
  };
  const int ns = ls.ns;
  const int nrounds = (ns + kNnThreads - 1) / kNnThreads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t so = (size_t)pair * b.ns_cap, to = (size_t)pair * b.nt_cap;
  uint32_t* gh = b.one_hist + (size_t)blockIdx.y * kHistBins;        // (its own histogram, counters and key lists: fine-grained memory, IcpDev::one_ctr)
  PairState* ctr = b.one_ctr + blockIdx.y;
  uint32_t* sync = b.one_sync + (size_t)blockIdx.y * kOneSyncWords;
  uint32_t* gkeys0 = b.one_keys + (size_t)blockIdx.y * 2 * b.bl_stride;
  // What every workgroup adds to is never cleared while any of them may still read it or may already be adding again: the counters
  // of searched / lower-bounded / unresolved queries are cumulative (cnt_prev: what this workgroup had read of them when the
  // iteration began), the key list and the smallest-bound word exist twice -- an iteration uses those of its parity, and workgroup 0
  // clears the other parity's behind the iteration's first barrier, an iteration after their last reader and a barrier before their
  // next writer.  (As first built, workgroup 0 cleared them behind the iteration's LAST barrier: nothing stood between that and the
  // fast workgroups' next additions -- a batch of two pairs with most workgroups idle lost counts and, once, hung.)
  uint32_t cnt_prev[3] = {0u, 0u, 0u};                   // hard_count, deferred_count, unresolved_count
  uint32_t* gkeys = gkeys0;                              // this iteration's key list and its length
  uint32_t* key_count = &sync[kSyncKeys];
  uint32_t* min_lb_word = &sync[kSyncMinLb];
  double* rows = b.one_rows + (size_t)blockIdx.y * (kOneMaxBlocks + 32) * kAccCols;   // [G][kAccCols] the workgroups' rows of sums
  double* grows = rows + (size_t)kOneMaxBlocks * kAccCols;            // [groups][kAccCols] the groups'
  uint32_t target = 0;
  // The pair's histogram is never cleared inside the launch: every workgroup remembers the eight words it owns as it last read them
  // and takes the difference (a store that clears a word other workgroups add to would have to be ordered against their atomics)
  uint32_t hprev[kHistBins / kNnThreads];
#pragma unroll
  for (int k = 0; k < kHistBins / kNnThreads; ++k) hprev[k] = 0u;
#if SMHIP_ONE_TIMING
  const bool otime = (b.debug_flags & 64) && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0;
  unsigned long long oacc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long oprev = otime ? wall_clock64() : 0ull;
#endif
  uint32_t cprev = 0;                                    // (threads 0..4) the pair's cumulative counts as last read
  double Mc[12];
  double acc[29];
  int wcount = 0;                                        // this wave's listed points so far (wave-uniform)
  float pd = INFINITY;                                   // this lane's listed point (the wave's first 64): fetched ahead of the barrier
  float4 ps = make_float4(0, 0, 0, 0), pq = ps, pn = ps;

  // ErrorElements + ComputePointToPlane (icp_fast.cc:100-166, 256-303) of this workgroup's points whose distance lies below
  // histogram bin `lo`, into acc; its points of bins [lo, hi] listed: remembered in LDS, their keys appended to the pair's key list
  // (ONE returning atomic per workgroup), the first 64 of each wave fetched whole.
  auto collect = [&](uint32_t lo, uint32_t hi) {
#pragma unroll
    for (int c = 0; c < 29; ++c) acc[c] = 0.0;
    wcount = 0;
    for (int r = blockIdx.x; r < nrounds; r += (int)G) {
      const int i = r * kNnThreads + (int)threadIdx.x;
      bool listed = false;
      if (i < ns) {
        const float d = b.d2[so + i];
        const int j = b.idx[so + i];
        const float4 s4 = ld_src(b, so + i);
        const uint32_t key = __float_as_uint(d);
        SMHIP_OCHK(j >= -1 && j < ls.nt, "collect: match id", j);
        if (key < 0x7f800000u) {
          const uint32_t bin = key >> kHistShift;
          if (bin < lo) accumulate_terms(Mc, s4, b.tq[to + max(j, 0)], b.tn[to + max(j, 0)], acc);
          else listed = bin <= hi;
        }
      }
      const unsigned long long bm = __ballot(listed);
      if (listed) s_rec[wave][wcount + (int)rank_below(bm)] = i;
      wcount += (int)__popcll(bm);
    }
    if (lane == 0) s_wc[wave] = (uint32_t)wcount;
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t total = s_wc[0] + s_wc[1] + s_wc[2] + s_wc[3];
      s_misc[1] = total ? __hip_atomic_fetch_add(key_count, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    }
    __syncthreads();
    uint32_t base = s_misc[1];
    for (int w = 0; w < wave; ++w) base += s_wc[w];
    SMHIP_OCHK(base + (uint32_t)wcount <= (uint32_t)b.bl_stride, "collect: key list position", base + (uint32_t)wcount);
    SMHIP_OCHK(wcount <= 64 * kOneMaxRounds, "collect: wave list length", wcount);
    for (int k = lane; k < wcount; k += 64) pub_dev(&gkeys[base + k], __float_as_uint(b.d2[so + s_rec[wave][k]]));
    pd = INFINITY;
    if (lane < wcount) {
      const int i = s_rec[wave][lane];
      SMHIP_OCHK(i >= 0 && i < ns, "collect: listed point", i);
      pd = b.d2[so + i];
      const int j = max(b.idx[so + i], 0);
      ps = ld_src(b, so + i); pq = b.tq[to + j]; pn = b.tn[to + j];
    }
  };
  // The exact quantile (icp_fast.cc:65-90): the rank-th smallest key of bin qbin in the pair's key list, by radix select over the
  // list as every workgroup reads it (the order it was appended in does not matter to the VALUE); then this workgroup's listed
  // points at or below it added to acc -- weights = (d2 <= limit), icp_fast.cc:497-498 -- in the order the waves met them.
  auto select_and_add = [&](uint32_t qbin, uint32_t rank) -> uint32_t {
    const int nb = (int)ld_dev(key_count);
    SMHIP_OCHK(nb >= 0 && nb <= b.bl_stride, "select: key list length", nb);
    const bool flat = nb <= kFinalizeKeyCap;
    if (flat) {
      for (int e0 = 0; e0 < nb; e0 += 8 * kNnThreads) {          // eight loads in flight per thread: one pair's ~2 000 keys in one round
        uint32_t kk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) kk[u] = ld_dev(&gkeys[min(e0 + u * kNnThreads + (int)threadIdx.x, nb - 1)]);
#pragma unroll
        for (int u = 0; u < 8; ++u) if (e0 + u * kNnThreads + (int)threadIdx.x < nb) s_keys[e0 + u * kNnThreads + threadIdx.x] = kk[u];
      }
    }
    SMHIP_OPH(6);
#if SMHIP_ONE_CHECKS
    {   // what this workgroup saw of the key list: its length and the sum of its keys (wrapping), for the disagreement report
      __syncthreads();
      uint32_t cs = 0;
      for (int e = threadIdx.x; e < nb; e += kNnThreads) cs += flat ? s_keys[e] : ld_dev(&gkeys[e]);
      for (int off = 32; off > 0; off >>= 1) cs += __shfl_xor(cs, off, 64);
      if (threadIdx.x == 0) s_cnt[14] = 0;
      __syncthreads();
      if (lane == 0) atomicAdd(&s_cnt[14], cs);
      if (threadIdx.x == 0) { s_cnt[15] = (uint32_t)nb; s_cnt[13] = rank; }
      __syncthreads();
    }
#endif
    uint32_t prefix = 0, mask = 0;
    for (int pass = 0; pass < 2; ++pass) {                         // the low 20 key bits, ten at a time
      const int shift = pass == 0 ? 10 : 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) s_h[threadIdx.x + u * kNnThreads] = 0;
      __syncthreads();                                            // (pass 0: also the keys staged above)
      for (int e = threadIdx.x; e < nb; e += kNnThreads) {
        const uint32_t full = flat ? s_keys[e] : ld_dev(&gkeys[e]);
        const uint32_t key = full & 0xfffffu;
        if ((full >> kHistShift) == qbin && (key & mask) == prefix) atomicAdd(&s_h[(key >> shift) & 1023u], 1u);
      }
      __syncthreads();
      {
        uint32_t c[4], v = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { c[u] = s_h[threadIdx.x * 4 + u]; v += c[u]; }
        uint32_t tot;
        const uint32_t excl = block_excl_scan(v, s_w, &tot);
        if (v > 0 && excl <= rank && rank < excl + v) {           // exactly one thread (0 <= rank < the count of keys under the prefix)
          uint32_t run = excl;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (rank < run + c[u]) { s_sel[0] = threadIdx.x * 4 + u; s_sel[1] = run; break; }
            run += c[u];
          }
        }
      }
      __syncthreads();
      prefix |= s_sel[0] << shift;
      mask |= 1023u << shift;
      rank -= s_sel[1];
    }
    const uint32_t limit_key = (qbin << kHistShift) | prefix;
    SMHIP_OPH(7);
    if (lane < wcount && __float_as_uint(pd) <= limit_key) accumulate_terms(Mc, ps, pq, pn, acc);
    for (int k = lane + 64; k < wcount; k += 64) {
      const int i = s_rec[wave][k];
      SMHIP_OCHK(i >= 0 && i < ns, "select: listed point", i);
      const float d = b.d2[so + i];
      if (__float_as_uint(d) <= limit_key) {
        const int j = max(b.idx[so + i], 0);
        accumulate_terms(Mc, ld_src(b, so + i), b.tq[to + j], b.tn[to + j], acc);
      }
    }
    SMHIP_OPH(8);
    return limit_key;
  };
  // this workgroup's row of sums published, the barrier with the groups' fold inside, the groups' rows added in order into dst
  // Columns 29 / 30 carry the exact quantile and the count of distances AS THIS WORKGROUP HAS THEM: every workgroup derives both from
  // what the others published, so they must be the same everywhere, and their sums over the G workgroups must be G times one's own --
  // a workgroup that finds otherwise (it, or another, read something the others did not) raises the abort word: the pair ends with
  // an error instead of workgroups that take different turns at the next iteration's barriers.
  auto publish_and_fold = [&](double* dst, uint32_t my_limit_key, uint32_t my_n_valid) {
    block_reduce29(acc, s_red, dst);
    if (threadIdx.x < kAccCols) {
      double x = threadIdx.x < 29 ? dst[threadIdx.x] : 0.0;
      if (threadIdx.x == 29) x = (double)my_limit_key;
      if (threadIdx.x == 30) x = (double)my_n_valid;
      pub_dev(&rows[(size_t)blockIdx.x * kAccCols + threadIdx.x], x);
    }
    SMHIP_OPH(9);
    SMHIP_OTRACE(5, my_limit_key);
    one_grid_sync_fold(sync, target, og, rows, grows, s_part, &s_misc[3], &s_abort);
    SMHIP_OPH(10);
    if (threadIdx.x < kAccCols) {
      double s = 0;
      for (uint32_t k = 0; k < og.NG; ++k)
        s += __longlong_as_double((long long)ld_dev64(grows + (size_t)k * kAccCols + threadIdx.x));
      dst[threadIdx.x] = s;
      if ((threadIdx.x == 29 && s != (double)my_limit_key * (double)G) || (threadIdx.x == 30 && s != (double)my_n_valid * (double)G)) {
        if (__hip_atomic_exchange(&sync[kSyncAbort], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
          st_dev(&sync[kSyncAbort + 1], (uint32_t)blockIdx.x); st_dev(&sync[kSyncAbort + 2], target);
          st_dev(&sync[kSyncAbort + 3], my_limit_key); st_dev(&sync[kSyncAbort + 4], my_n_valid);
          st_dev(&sync[kSyncAbort + 5], (uint32_t)threadIdx.x); st_dev(&sync[kSyncAbort + 6], (uint32_t)(s / (double)G));
#if SMHIP_ONE_CHECKS
          printf("[icp_one] disagreement seen by pair row %u workgroup %u iteration %d: its key list has %u keys, key sum %u, rank %u, quantile key %u; fused %d\n",
                 (unsigned)blockIdx.y, (unsigned)blockIdx.x, ls.iter, s_cnt[15], s_cnt[14], s_cnt[13], my_limit_key, (int)(ls.band_lo > 0));
#endif
        }
        s_abort = 1u;
      }
    }
    __syncthreads();
    SMHIP_OPH(11);
  };
  // this iteration's counters, as every workgroup has read them, cleared for the next iteration's S: by exchanges, whose old
  // values have come back before this workgroup arrives at the next barrier (a plain store's completion says less about where
  // it stands against another XCD's atomics on the same word)
  // behind the iteration's first barrier: the other parity's key list and smallest-bound word cleared for the next iteration (by
  // exchanges, whose old values have come back before this workgroup arrives at the next barrier)
  auto clear_next_parity = [&]() {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      const uint32_t q = (uint32_t)(ls.iter + 1) & 1u;
      uint32_t o = __hip_atomic_exchange(&sync[kSyncKeys + 32 * q], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      o |= __hip_atomic_exchange(&sync[kSyncMinLb + 32 * q], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_misc[4] = o;
    }
  };
  // (a fused attempt that missed leaves its candidates in the list the plain form is about to fill: cleared behind the attempt's
  // barrier, a barrier before the plain form appends)
  auto clear_key_list = [&]() {
    if (blockIdx.x == 0 && threadIdx.x == 0) s_misc[5] = __hip_atomic_exchange(key_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto flush_own_hist = [&]() {
    __syncthreads();
    uint32_t o = 0;
    for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) {
      const uint32_t v = s_hist[k];
      if (v) o |= __hip_atomic_fetch_add(&gh[k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("" :: "v"(o));                             // (performed, not just accepted, before the barrier)
  };
  auto read_counters = [&]() {                                    // thread 0: this iteration's share of the cumulative counters
    ls.hard_count = ld_dev(&ctr->hard_count) - cnt_prev[0];
    ls.deferred_count = ld_dev(&ctr->deferred_count) - cnt_prev[1];
    ls.unresolved_count = 0;
    ls.min_lb_key = ~ld_dev(min_lb_word);
  };

  for (;;) {
    // ---------------- S: FindClosests (icp_fast.cc:486-493)
#if SMHIP_ONE_TIMING
    const unsigned long long s_t0 = wall_clock64();
#endif
#pragma unroll
    for (int k = 0; k < 12; ++k) Mc[k] = uniform_f64(ls.M[k]);
    for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) s_hist[k] = 0;
    uint32_t min_lb = 0xffffffffu;
    {
      const uint32_t par = (uint32_t)ls.iter & 1u;
      gkeys = gkeys0 + (size_t)par * b.bl_stride;
      key_count = &sync[kSyncKeys + 32 * par];
      min_lb_word = &sync[kSyncMinLb + 32 * par];
    }
    if (ls.iter == 0) {
      // every query searches: the rounds of nn_ball_lds (the workgroup's queries walk the rows of their balls from LDS tables)
      for (int r = blockIdx.x; r < nrounds; r += (int)G) ball_lds_rounds<1, false>(b, ctr, &ls, pair, r * kNnThreads, Mc, s_hist, min_lb);
    } else {
      // Later iterations: the certificate for every query (nn_certify), then the few whose certificate fails (6 % of them in a
      // settled iteration: ~30 of a workgroup's 512) searched with as many lanes each as the workgroup has to spare (the listed
      // search, listed_search_one: a lane takes every L-th row of the query's ball, the lanes' results merged with the sweep's tie
      // rule) -- where nn_ball_lds gives a failing query ONE lane and walks the workgroup through its staging barriers for it
      // (8 us per round against 3).  The same matches either way (every search here is exact); the recorded bounds differ in
      // their search radius, as they do between the two forms of the batched path.
      int* s_fail = &s_rec[0][0];                            // (collect's list: not in use during S)
      uint32_t asink = 0;                                    // (the additions below return their old values: done once those have come back)
      if (threadIdx.x == 0) s_misc[7] = 0;
      __syncthreads();
      const Pot pot = {(float)ls.pot_a, (float)ls.pot_b, 0.f, 0.f};
      const float r_need = 0.9f * sqrtf(ls.rcap2);
      const float4* __restrict__ tq = b.tq + to;
      for (int r = blockIdx.x; r < nrounds; r += (int)G) {
        const int i = r * kNnThreads + (int)threadIdx.x;
        bool hard = false, fail = false;
        if (i < ns) {
          const float4 s = ld_src(b, so + i);
          const float l = ld_lb(b, so + i);
          const int j = b.idx[so + i];
          SMHIP_OCHK(j >= -1 && j < ls.nt, "certify: match id", j);
          const float4 t = tq[max(j, 0)];
          double px, py, pz;
          transform_point(Mc, s, px, py, pz);
          const float qx = (float)px, qy = (float)py, qz = (float)pz;
          const float Lp = bound_now(l, pot_at(pot, norm3(s.x, s.y, s.z)));
          fail = true;
          if (isfinite(qx) && isfinite(qy) && isfinite(qz) && Lp > 0.f) {
            if (l > 0.f && j >= 0) {
              const float d1 = dist2(t, qx, qy, qz);
              if (d1 < Lp * Lp) {                              // still the unique nearest neighbour: exact, no search
                b.d2[so + i] = d1;
                atomicAdd(&s_hist[__float_as_uint(d1) >> kHistShift], 1u);
                fail = false;
              }
            } else if (l < 0.f && Lp >= r_need) {              // still provably farther than the trimming radius
              const float lb2 = Lp * Lp;
              b.d2[so + i] = lb2;
              atomicAdd(&s_hist[__float_as_uint(lb2) >> kHistShift], 1u);
              min_lb = min(min_lb, __float_as_uint(lb2));
              hard = true;
              fail = false;
            }
          }
        }
        const unsigned long long fm = __ballot(fail);
        if (fm) {
          uint32_t basepos = 0;
          if (lane == 0) basepos = atomicAdd(&s_misc[7], (uint32_t)__popcll(fm));
          basepos = (uint32_t)__builtin_amdgcn_readlane((int)basepos, 0);
          if (fail) s_fail[basepos + rank_below(fm)] = i;
        }
        const unsigned long long hm = __ballot(hard);
        if (lane == 0 && hm) asink |= __hip_atomic_fetch_add(&ctr->hard_count, (uint32_t)__popcll(hm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      const int nf = (int)s_misc[7];
      if (threadIdx.x == 0 && nf) asink |= __hip_atomic_fetch_add(&ctr->deferred_count, (uint32_t)nf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (nf > 0) {
        const ListedCtx ctx = listed_ctx(b, &ls, pair);
        int logL = 0;
        while (logL < 4 && (nf << (logL + 1)) <= kNnThreads) ++logL;
        const int L = 1 << logL, per = kNnThreads >> logL;
        const int sub = (int)threadIdx.x & (L - 1), qi = (int)threadIdx.x >> logL;
        SMHIP_OCHK(nf <= kOneMaxRounds * kNnThreads, "search: failing queries", nf);
        for (int base = 0; base < nf; base += per) {
          const int i = base + qi < nf ? s_fail[base + qi] : -1;
          SMHIP_OCHK(i >= -1 && i < ns, "search: listed query", i);
          bool hard = false, band = false;
          float4 srec; float drec; int jrec;
          listed_search_one(b, &ls, ctx, so, i, sub, L, s_hist, min_lb, hard, false, 0, 0, band, srec, drec, jrec);
          const unsigned long long hm = __ballot(hard && sub == 0);
          if (lane == 0 && hm) asink |= __hip_atomic_fetch_add(&ctr->hard_count, (uint32_t)__popcll(hm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      asm volatile("" :: "v"(asink));
      __syncthreads();                                        // (s_fail is collect's list again)
    }
    {   // the smallest lower bound this workgroup recorded -> the iteration's word (kept as its complement: zero = none)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) min_lb = min(min_lb, (uint32_t)__shfl_xor((int)min_lb, off, 64));
      uint32_t o = 0;
      if (lane == 0 && min_lb != 0xffffffffu) o = __hip_atomic_fetch_max(min_lb_word, ~min_lb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" :: "v"(o));
    }
#if SMHIP_ONE_TIMING
    if ((b.debug_flags & 64) && threadIdx.x == 0) {          // how long S took in every workgroup: the largest and the sum (sync[40], [41])
      const uint32_t dtS = (uint32_t)(wall_clock64() - s_t0);
      atomicMax(&sync[40], dtS); atomicAdd(&sync[41], dtS);
    }
#endif
    SMHIP_OPH(0);
    uint32_t n_valid = 0, limit_key = 0;
    bool have_sums = false, first_barrier_done = false;
    // ---------------- the fused form: finalize_tail predicted the histogram bins this iteration's quantile can fall in
    // ([band_lo, band_hi], at most three; band_lo = 0: no prediction).  Then no histogram crosses workgroups at all: the sums below
    // the band stay in registers, the band's members are listed, and five counts from the workgroup's own LDS histogram -- all
    // distances, those below the band, those in each of its bins -- ride on the barrier.  The totals say whether the quantile's bin
    // did land in the band (and no lower bound needs refining): if so the exact select over the band's keys and ONE row of sums
    // finish the iteration with TWO barriers; if not, the iteration is done again the plain way below.  (The plain way's cost is its histogram:
    // ~200 atomics per workgroup on words all 472 workgroups add to -- same-address atomics complete one every ~13 ns.)
    const uint32_t blo = (uint32_t)ls.band_lo, bhi = (uint32_t)ls.band_hi;
    if (ls.band_lo > 0 && !(b.debug_flags & 128)) {
      __syncthreads();
      collect(blo, bhi);
      SMHIP_OPH(4);
      {   // this workgroup's counts from its own histogram: all distances, those below the band (at most 2 048 each: one scan for both)
        uint32_t v = 0;
#pragma unroll
        for (int u = 0; u < kHistBins / kNnThreads; ++u) {
          const uint32_t bin = threadIdx.x * (kHistBins / kNnThreads) + u, c = s_hist[bin];
          v += c + (bin < blo ? c << 16 : 0u);
        }
        uint32_t tot;
        (void)block_excl_scan(v, s_w, &tot);
        if (threadIdx.x < 5) s_cnt[threadIdx.x] = threadIdx.x == 0 ? (tot & 0xffffu) : (threadIdx.x == 1 ? tot >> 16 : (blo + (threadIdx.x - 2) <= bhi ? s_hist[blo + (threadIdx.x - 2)] : 0u));
      }
      SMHIP_OPH(9);
      SMHIP_OTRACE(1, s_cnt[0]);
      one_grid_sync_counts(sync, target, og, s_cnt, s_cnt + 8, cprev, &s_misc[3], &s_abort);
      if (bail()) return;
      SMHIP_OPH(10);
      clear_next_parity();
      first_barrier_done = true;
      if (threadIdx.x == 0) {
        read_counters();
        const uint32_t nv = s_cnt[8], below = s_cnt[9];
        uint32_t hit = 0, qb = 0, rank = 0;
        if (nv > 0) {
          const uint32_t k = (uint32_t)quantile_rank(nv, b.rho);
          uint32_t run = below;
          if (k >= below)
            for (uint32_t c = 0; blo + c <= bhi; ++c) {
              const uint32_t n = s_cnt[10 + c];
              if (k < run + n) { hit = 1; qb = blo + c; rank = k - run; break; }
              run += n;
            }
        }
        // (nn_validate) a lower bound sharing the quantile's bin is not provably above it
        if (hit && ls.hard_count > 0 && (b.exact_all || (ls.min_lb_key >> kHistShift) <= qb)) hit = 0;
        s_q[0] = qb; s_q[1] = rank; s_q[2] = nv; s_q[3] = hit;
      }
      __syncthreads();
      SMHIP_OPH(3);
      if (s_q[3]) {
        n_valid = s_q[2];
        limit_key = select_and_add(s_q[0], s_q[1]);               // acc: the sums below the band + the band's members at or below the exact quantile
        publish_and_fold(s_tot, limit_key, n_valid);
        if (bail()) return;
        if (threadIdx.x == 0) ls.spec_hits += 1;
        have_sums = true;
      } else {
        clear_key_list();                                          // (nobody reads the list on a miss; the plain way appends behind its first barrier)
      }
    }
    if (!have_sums) {
      // ---------------- the plain form: the pair's histogram, the quantile's bin, the sums below it, the exact select inside it
      flush_own_hist();
      SMHIP_OPH(1);
      SMHIP_OTRACE(2, have_sums ? 1 : 0);
      one_grid_sync(sync, target, og, &s_abort);
      if (bail()) return;
      if (!first_barrier_done) clear_next_parity();
      SMHIP_OPH(2);
      // V: the quantile's bin; do the lower bounds stand above it?  (nn_validate)
      uint32_t hraw[kHistBins / kNnThreads], hcnt[kHistBins / kNnThreads];
#pragma unroll
      for (int k = 0; k < kHistBins / kNnThreads; ++k) hraw[k] = ld_dev(&gh[threadIdx.x * (kHistBins / kNnThreads) + k]);
#pragma unroll
      for (int k = 0; k < kHistBins / kNnThreads; ++k) hcnt[k] = hraw[k] - hprev[k];
      find_quantile_bin_counts(hcnt, b.rho, s_w, s_q);
      if (threadIdx.x == 0) {
        read_counters();
        const bool any = ls.hard_count > 0;
        const bool below = (ls.min_lb_key >> kHistShift) <= s_q[0];
        const bool refine = any && (b.exact_all || below || s_q[2] == 0);
        if (refine) ls.refine_total += 1;
        s_misc[0] = refine ? 1u : 0u;
      }
      __syncthreads();
      if (s_misc[0]) {
        // The bounds are refined to matches (nn_ring<true> + nn_fallback) by their OWNERS: every workgroup walks the rings for the
        // lower-bounded queries among its own points (a stored bound < 0 marks them) and sweeps the whole target for those the rings
        // leave open -- nothing but the histogram and a counter crosses workgroups.  Then one more barrier and the quantile again.
        // First a barrier: the refinement takes bounds out of the pair's histogram, and no workgroup may still be reading it for the
        // decision that brought all of them here (without it a slow workgroup read a histogram some bounds had already left, found
        // the quantile below the smallest bound and did NOT refine: workgroups of one pair at different barriers -- seen only in
        // launches of several pairs of mixed sizes, where the small pairs' workgroups run far apart).
        one_grid_sync(sync, target, og, &s_abort);
        if (bail()) return;
        for (int k = threadIdx.x; k < kHistBins; k += kNnThreads) s_hist[k] = 0;
        if (threadIdx.x == 0) s_misc[6] = 0;
        __syncthreads();
        float4* s_t = reinterpret_cast<float4*>(s_keys);
        const float4* __restrict__ tq = b.tq + to;
        const int nt = ls.nt;
        for (int r = blockIdx.x; r < nrounds; r += (int)G) {
          const int i = r * kNnThreads + (int)threadIdx.x;
          const bool hardq = i < ns && b.lb[so + i] < 0.f;
          bool open = false;
          float qx = 0.f, qy = 0.f, qz = 0.f;
          Best best = {INFINITY, -1, INFINITY};
          if (hardq) {
            const uint32_t old = __float_as_uint(b.d2[so + i]);           // the bound leaves the histogram
            if (old < 0x7f800000u) atomicSub(&gh[old >> kHistShift], 1u);
            double px, py, pz;
            transform_point(Mc, ld_src(b, so + i), px, py, pz);
            qx = (float)px; qy = (float)py; qz = (float)pz;
            open = !ring_search_query(b, &ls, pair, qx, qy, qz, best);
          }
          if (__syncthreads_or(open ? 1 : 0)) {                            // workgroup-uniform
            if (open) best = {INFINITY, -1, INFINITY};
            for (int base = 0; base < nt; base += kBruteTile) {
              const int m = min(kBruteTile, nt - base);
              __syncthreads();
              for (int k = threadIdx.x; k < m; k += kNnThreads) s_t[k] = tq[base + k];
              __syncthreads();
              if (open)
#pragma unroll 8
                for (int k = 0; k < m; ++k) test_ascending(s_t[k], base + k, qx, qy, qz, best);
            }
            __syncthreads();
            const unsigned long long om = __ballot(open);
            if (lane == 0 && om) atomicAdd(&s_misc[6], (uint32_t)__popcll(om));
          }
          if (hardq) {
            b.d2[so + i] = best.d2;
            st_match(b, so + i, best.j, 0.f);                              // exact match, no runner-up information: searched again next time
            const uint32_t key = __float_as_uint(best.d2);
            if (key < 0x7f800000u) atomicAdd(&s_hist[key >> kHistShift], 1u);
          }
        }
        __syncthreads();
        if (threadIdx.x == 0 && s_misc[6]) {
          const uint32_t o = __hip_atomic_fetch_add(&ctr->unresolved_count, s_misc[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          asm volatile("" :: "v"(o));
        }
        flush_own_hist();
        SMHIP_OTRACE(3, s_q[0] | ((ls.min_lb_key >> kHistShift) << 12) | (ls.hard_count > 0 ? 1u << 24 : 0u) | ((s_q[2] & 0x7fu) << 25));
        one_grid_sync(sync, target, og, &s_abort);
        if (bail()) return;
#pragma unroll
        for (int k = 0; k < kHistBins / kNnThreads; ++k) hraw[k] = ld_dev(&gh[threadIdx.x * (kHistBins / kNnThreads) + k]);
#pragma unroll
        for (int k = 0; k < kHistBins / kNnThreads; ++k) hcnt[k] = hraw[k] - hprev[k];
        find_quantile_bin_counts(hcnt, b.rho, s_w, s_q);
        if (threadIdx.x == 0) ls.unresolved_count = ld_dev(&ctr->unresolved_count) - cnt_prev[2];
        __syncthreads();
      }
#pragma unroll
      for (int k = 0; k < kHistBins / kNnThreads; ++k) hprev[k] = hraw[k];
      const uint32_t qbin = s_q[0], below = s_q[1], krank = s_q[3];
      n_valid = s_q[2];
      SMHIP_OPH(3);
      // A: the sums below the bin, the bin's members listed
      if (n_valid > 0) collect(qbin, qbin);
      else {
#pragma unroll
        for (int c = 0; c < 29; ++c) acc[c] = 0.0;
        wcount = 0;
      }
      SMHIP_OPH(4);
      SMHIP_OTRACE(4, qbin | ((ls.min_lb_key >> kHistShift) << 12) | (ls.hard_count > 0 ? 1u << 24 : 0u) | ((n_valid & 0x7fu) << 25));
      one_grid_sync(sync, target, og, &s_abort);
      if (bail()) return;
      SMHIP_OPH(5);
      // F1: the exact quantile, the bin's members at or below it; F2: the rows
      if (n_valid > 0) limit_key = select_and_add(qbin, krank - below);
      publish_and_fold(s_tot, limit_key, n_valid);
      if (bail()) return;
    }
    // ---------------- solve, pose update, convergence -- in every workgroup, on its own copy of the state
    if (threadIdx.x == 0) {
      cnt_prev[0] += ls.hard_count; cnt_prev[1] += ls.deferred_count; cnt_prev[2] += ls.unresolved_count;   // (the tail folds them into the totals and zeroes them)
      const TailOpts o = {b.cap_factor, b.ball_radius, b.band_gain, b.band_pad, 0, b.early_exit, b.max_iteration, b.search_hist, b.done_count};
      one_tail(o, &ls, pair, s_tot, n_valid, limit_key, ns, blockIdx.x == 0);
    }
    __syncthreads();
    SMHIP_OPH(12);
    if (blockIdx.x == 0) {
      // the state back to global memory (the host and the next Align's kernels read it there) -- all of it but the counters the
      // other workgroups' next S may already be adding to
      const uint32_t* l = reinterpret_cast<const uint32_t*>(&ls);
      uint32_t* g = reinterpret_cast<uint32_t*>(st);
      for (int k = threadIdx.x; k < (int)(sizeof(PairState) / 4); k += kNnThreads) {
        const size_t o = (size_t)k * 4;
        const bool live = o == offsetof(PairState, hard_count) || o == offsetof(PairState, deferred_count) || o == offsetof(PairState, unresolved_count) ||
                          o == offsetof(PairState, min_lb_key) || o == offsetof(PairState, fallback_ticket);
        if (!live) g[k] = l[k];
      }
    }
    if (ls.done) break;
  }
  // ---------------- the score of the iteration the loop ended with: exp(-mean distance of its kept matches) (icp_fast.cc:516-522)
  if (ls.status != 0) return;
  // (a barrier first: the score's fold writes the groups' rows again, and a group whose workgroups are all here already would
  // write its row while a slow workgroup of another group is still reading the LAST iteration's -- inside the loop a whole
  // barrier always lies between two folds, here none did: a pair of a launch of seven ended with 10 729 of its 14 000 kept
  // matches in its state, which the score's own count gave away)
  one_grid_sync(sync, target, og, &s_abort);
  if (bail()) return;
  {
    const uint32_t limit_key = ls.limit_key;
    double s = 0.0;
    uint32_t cnt = 0;
    for (int r = blockIdx.x; r < nrounds; r += (int)G) {
      const int i = r * kNnThreads + (int)threadIdx.x;
      if (i < ns) {
        const float d = b.d2[so + i];
        if (__float_as_uint(d) <= limit_key) { s += sqrt((double)d); ++cnt; }
      }
    }
    s = wave_sum_to_last(s);
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if (lane == 63) s_red[wave][0] = s;
    if (lane == 0) s_wc[wave] = cnt;
    __syncthreads();
    if (threadIdx.x < 29) {
      double v = 0.0;
      if (threadIdx.x == 0) v = ((s_red[0][0] + s_red[1][0]) + s_red[2][0]) + s_red[3][0];
      if (threadIdx.x == 1) v = (double)(((s_wc[0] + s_wc[1]) + s_wc[2]) + s_wc[3]);
      pub_dev(&rows[(size_t)blockIdx.x * kAccCols + threadIdx.x], v);
    }
    one_grid_sync_fold(sync, target, og, rows, grows, s_part, &s_misc[3], &s_abort);
    if (bail()) return;
    if (blockIdx.x != 0) return;
    if (threadIdx.x < 2) {
      double t = 0;
      for (uint32_t k = 0; k < og.NG; ++k)
        t += __longlong_as_double((long long)ld_dev64(grows + (size_t)k * kAccCols + threadIdx.x));
      s_part[0][threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const double tot = s_part[0][0], n = s_part[0][1];
      st->score_mismatch = (uint32_t)n != (uint32_t)ls.kept ? 1u : 0u;
      st->score_cnt[0] = (uint32_t)n;                         // (what the error text shows beside the kept count)
      st->score = n > 0 ? exp(-tot / n) : 0.0;
    }
  }
#if SMHIP_ONE_TIMING
  if (otime) {
    unsigned long long tot = 0;
    for (int k = 0; k < 13; ++k) tot += oacc[k];
    const double u = 0.01 / ls.iter;
    printf("[icp_one] S over the workgroups: largest %.1f us (of any iteration), mean %.1f us\n", ld_dev(&sync[40]) * 0.01, ld_dev(&sync[41]) * 0.01 / ((double)G * ls.iter));
    printf("[icp_one] iterations %d grid %u us per iteration: S %.1f flush %.1f bar1 %.1f V %.1f A %.1f bar2 %.1f F1 (keys %.1f select %.1f records %.1f reduce %.1f) bar3 %.1f fold %.1f tail %.1f | total %.1f us\n",
           ls.iter, G, oacc[0] * u, oacc[1] * u, oacc[2] * u, oacc[3] * u, oacc[4] * u, oacc[5] * u, oacc[6] * u, oacc[7] * u, oacc[8] * u, oacc[9] * u, oacc[10] * u,
           oacc[11] * u, oacc[12] * u, tot * 0.01);
  }
#endif
}

}  // namespace smhip
