// kd_median_tree.h -- the median-split kd-tree both callers of the registrators build, level by level on the device.
//
// libnabo's buildNodes (nabo/kdtree_cpu.cpp, bucketSize 8: the tree IcpFast searches, /root/reference/registrators/
// icp_fast.cc:464-467) and EigenPointCloud::BuildNormals (/root/reference/builder/data/cloud_types.cc:105-144, leaves of
// <= 7 points: the caller-side CalculateNormals) are the same construction: a node with more points than a bucket splits
// on the widest side of the box it INHERITED (root: the cloud's bounds; a child: the parent's box cut at the cut value;
// argMax from (index 0, value 0)), leftCount = count - count / 2, std::nth_element at that rank, cut value = that
// element's coordinate.  The tree depends only on which points fall on which side of each median, so it is built level by
// level by ONE 1024-thread workgroup per cloud: every position carries its segment, an exact radix select on the
// order-preserving float key of the cut coordinate finds each segment's median (8 / 4 / 2 / 1 bits per pass as the segments
// multiply, all segments of a level in one sweep with their histograms in 64 KiB of LDS), ties on the median value are
// broken by the index the point carries in .w with a second select (block-uniformly skipped when there are none), and
// one partition pass moves the points; once the segments hold <= 64 points each, every element is ranked inside its
// segment by a wave instead (no sweeps: it counts the smaller keys of its segment, laid in LDS).  Many clouds = many workgroups: a batch of >= 256 clouds fills the chip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smhip {

constexpr int kKdThreads = 1024;          // one workgroup builds one cloud's tree
constexpr int kKdHistWords = 16384;       // 64 KiB of LDS histograms: segments per pass x 2^bits bins
#ifndef SMHIP_KD_WN
#define SMHIP_KD_WN 2                  // windows a wave takes per trip where it ranks by counting
#endif
constexpr int kKdCountMax = 64;           // levels whose segments hold at most this many points rank by counting inside a wave's window

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_down(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, 64));
  return v;
}
// `fill` where the DPP control has no source lane (or the row is masked off), else v of the source lane
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or(int fill, int v) {
  return __builtin_amdgcn_update_dpp(fill, v, CTRL, ROW_MASK, 0xf, false);
}
// Inclusive scan over the wave with DPP row operations (register to register): a scan inside every row of 16 lanes
// (row_shr 1, 2, 4, 8), then the row totals carried across (row_bcast 15 into rows 1 and 3, row_bcast 31 into rows 2 and
// 3).  The __shfl_up form went through ds_bpermute: 6 LDS round trips per scan.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int /*lane*/) {
  v += (uint32_t)dpp_or<0x111, 0xf>(0, (int)v);
  v += (uint32_t)dpp_or<0x112, 0xf>(0, (int)v);
  v += (uint32_t)dpp_or<0x114, 0xf>(0, (int)v);
  v += (uint32_t)dpp_or<0x118, 0xf>(0, (int)v);
  v += (uint32_t)dpp_or<0x142, 0xa>(0, (int)v);
  v += (uint32_t)dpp_or<0x143, 0xc>(0, (int)v);
  return v;
}

// Block-wide exclusive scan of one value per thread (blockDim.x multiple of 64, <= 1024).
// Returns the exclusive prefix; *total receives the block sum.  s_w needs 17 words.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_w, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  uint32_t inc = wave_incl_scan(v, lane);
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int w = 0; w < nwave; ++w) { uint32_t t = s_w[w]; s_w[w] = run; run += t; }
    s_w[16] = run;
  }
  __syncthreads();
  uint32_t excl = inc - v + s_w[wave];
  *total = s_w[16];
  __syncthreads();
  return excl;
}


// SMHIP_KD_TIMING (diagnostic build): workgroup 0 adds up where kd_median_build's time goes (wall clock, 10 ns ticks) and prints it
#ifdef SMHIP_KD_TIMING
#define SMHIP_KDPH(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long now_ = wall_clock64(); s_kdph[i] += now_ - s_kdph[15]; s_kdph[15] = now_; } } while (0)
#else
#define SMHIP_KDPH(i) do { } while (0)
#endif

#ifndef SMHIP_KD_SU
#define SMHIP_KD_SU 4
#endif
#ifndef SMHIP_KD_PU
#define SMHIP_KD_PU 4
#endif

struct KdSeg {                            // one node of the current level while the tree is being built
  uint32_t first, count;                  // its points: positions [first, first + count) of the working order
  float mn[3], mx[3];                     // the box it inherited
  uint32_t node;                          // its index in the node array
  uint32_t split;                         // 1 = more than a bucket: splits at this level
  uint32_t dim, left;                     // cut dimension, leftCount
  uint32_t prefix, k, nless, neq;         // radix select state: key bits fixed so far, rank among the still-matching keys,
                                          // keys known to be smaller, keys equal to the selected one
  uint32_t vidx;                          // ties at the median: caller indices below this one go left
  uint32_t rank;                          // number of splitting segments before this one
  uint32_t tie, trank;                    // several points ON the median value of which `trank` (running) must go left
  uint32_t ldim, rdim;                    // the cut dimensions the two children will take (known once the cut value is)
};


__device__ __forceinline__ uint32_t kd_key(float x) {          // order-preserving float -> uint
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float kd_unkey(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ float kd_coord(const float4 p, uint32_t d) { return d == 0 ? p.x : (d == 1 ? p.y : p.z); }


// Preconditions (all threads of the 1024-thread workgroup call it together): cur[i] = point i with .w = its index bits for
// i < n, sid[i] = 0, seg[0] = the root {first 0, count n, node 0, the cloud's bounds}, s_misc[0] = 1 (node count).
// s_hist: kKdHistWords words, s_w: 17, s_misc: 4 (all LDS).  On return cur / oth, sid / sid_o and seg / seg_o have been
// swapped once per level: `cur` holds the final order (every leaf's points contiguous), nodes[] the tree (inner: {cut value
// bits, (left child << 2) | dim}, children side by side; leaf: {first, (count << 2) | 3}), s_misc[0] the node count.
// kk / kk_o: [n] words each, the select's own stream -- kk[i] = the order-preserving key of point i's coordinate on the cut
// dimension of ITS segment.  The sweeps of the radix select (four per level) then move 8 bytes per point (key + segment id)
// instead of 20 (the whole float4 for one coordinate of it).  A child's cut dimension follows from the box it inherits, so
// the partition pass, which has the point in registers, writes the next level's keys as it scatters.
// fetch(i) = point i of the cloud as the working orders hold it (.w = i as int bits) -- not called any more (the last levels kept
// keys, not points, while they sorted; now every lane keeps its point); the callers' argument stays.  n < 2^24 - 1.
// WIDE: the extents of the inherited box compared as exact double differences (cloud_types.cc works on doubles: two sides whose
// float difference rounds to the same value are still told apart, as the sort-per-level builder of prep_normals.hip does); the
// libnabo restatement keeps the float differences it has always used.
template <int BUCKET, bool WIDE, typename Fetch>
__device__ __forceinline__ void kd_median_build(int n, Fetch fetch, float4*& cur, float4*& oth, uint32_t*& sid, uint32_t*& sid_o, KdSeg*& seg, KdSeg*& seg_o,
                                                uint32_t*& kk, uint32_t*& kk_o,
                                                uint2* nodes, uint32_t* cnt_global, int seg_cap, int node_cap,
                                                uint32_t* s_hist, uint32_t* s_w, uint32_t* s_misc, int32_t* status) {
  const int tid = threadIdx.x;
  (void)fetch;
  __shared__ uint32_t s_gst[3 * 64];                         // select state of a group's segments during a sweep
  if (n >= 0xffffff) { if (tid == 0) *status = 3; return; }  // (the sort keys of the last levels hold 24 index bits; the callers' own caps are far below)
  int S = 1;                                                 // segments of the current level
#ifdef SMHIP_KD_TIMING
  __shared__ unsigned long long s_kdph[16];
  if (tid == 0) { for (int k = 0; k < 15; ++k) s_kdph[k] = 0; s_kdph[15] = wall_clock64(); }
#endif
  for (int level = 0; level < 40 && S > 0; ++level) {
    SMHIP_KDPH(7);
    // ---- per segment: leaf or split, cut dimension, leftCount
    uint32_t my_splits = 0;
    if (tid == 0) s_misc[2] = 0;                             // the largest splitting segment of the level
    __syncthreads();
    const int per = (S + kKdThreads - 1) / kKdThreads;
    const int s_lo = min(S, tid * per), s_hi = min(S, s_lo + per);
    for (int s = s_lo; s < s_hi; ++s) {
      KdSeg& g = seg[s];
      if (g.count <= (uint32_t)BUCKET) {
        g.split = 0;
        nodes[g.node] = make_uint2(g.first, (g.count << 2) | 3u);
      } else {
        g.split = 1;
        uint32_t cd = 0;                                     // argMax from (0, 0.)
        if (WIDE) {
          double mv = 0.0;
          for (uint32_t d = 0; d < 3; ++d) { const double e = (double)g.mx[d] - (double)g.mn[d]; if (e > mv) { mv = e; cd = d; } }
        } else {
          float mv = 0.f;
          for (uint32_t d = 0; d < 3; ++d) { const float e = g.mx[d] - g.mn[d]; if (e > mv) { mv = e; cd = d; } }
        }
        g.dim = cd;
        g.left = g.count - g.count / 2;
        g.prefix = 0; g.k = g.left; g.nless = 0; g.neq = 0; g.vidx = 0;
        atomicMax(&s_misc[2], g.count);
        ++my_splits;
      }
    }
    uint32_t nsplit;
    const uint32_t my_rank = block_excl_scan(my_splits, s_w, &nsplit);
    {
      uint32_t r = my_rank;
      for (int s = s_lo; s < s_hi; ++s) if (seg[s].split) seg[s].rank = r++;
    }
    __syncthreads();
    if (nsplit == 0) break;
    if (2 * nsplit > (uint32_t)seg_cap || s_misc[0] + 2 * nsplit > (uint32_t)node_cap) { if (tid == 0) *status = 3; break; }   // cannot happen: caps follow nt_cap

    const uint32_t nc = s_misc[0];
    SMHIP_KDPH(0);
    if (level == 0 && s_misc[2] > 64u) {                     // the root's keys (every later level's come from the partition below)
      const uint32_t d0 = seg[0].dim;
      for (int i = tid; i < n; i += kKdThreads) kk[i] = kd_key(kd_coord(cur[i], d0));
      __syncthreads();
    }
    if (s_misc[2] <= (uint32_t)kKdCountMax) {
      // ---- small segments (<= kKdCountMax points each: the last levels, where the radix select below would need 16-32 sweeps over
      // every point because thousands of segments share the histogram words).  A wave takes a window of 128 consecutive positions
      // (two per lane) and owns the segments that START in its first half -- they end inside the window.  It lays the window's
      // keys ([55:24] the coordinate key, [23:0] the point's index in its cloud: all distinct) in LDS, and every element of an
      // owned segment counts the keys of ITS segment that are smaller than its own -- as many steps as the level's largest
      // segment has points, one 8-byte LDS read and a compare each; the count is its rank: the point, still in the lane's
      // registers, goes straight to first + rank, left child = ranks below `left`, cut value = the coordinate of rank `left`.
      // (Two earlier forms: every element against 128 broadcasts, 2 600 instructions a window; a bitonic network over the
      // window's 128 keys, 28 compare-exchange stages of two 64-bit shuffles each, after which the points had to be picked up
      // again by index -- the four levels of a 120 000-point cloud 4.0 ms, counted 2.8.)  kWN windows a trip, their loads
      // issued together.
      const int lane = tid & 63;
      constexpr int kWN = SMHIP_KD_WN, kE = 2 * kWN;           // element 2 * w + e: window w of the trip, half e
      unsigned long long* wk = reinterpret_cast<unsigned long long*>(s_hist) + 128 * kWN * (tid >> 6);   // the wave's 128 keys per window
      const uint32_t steps = s_misc[2];
      for (uint32_t wbase = 64u * (uint32_t)(tid >> 6); wbase < (uint32_t)n; wbase += (uint32_t)(kWN * kKdThreads)) {
        unsigned long long kq[kE];
        float4 pp[kE];
        uint32_t fl[kE], cn[kE], lf[kE], sr[kE], svv[kE], gd[kE], gf[kE];
        bool own[kE], act[kE];
#pragma unroll
        for (int q = 0; q < kE; ++q) {
          const uint32_t pos = wbase + (uint32_t)((q >> 1) * kKdThreads) + 64u * (q & 1) + lane;
          svv[q] = 0xffffffffu;
          pp[q] = make_float4(0, 0, 0, 0);
          if (pos < (uint32_t)n) { svv[q] = sid[pos]; pp[q] = cur[pos]; }
        }
#pragma unroll
        for (int q = 0; q < kE; ++q) {
          act[q] = false; gd[q] = 0; gf[q] = 0; cn[q] = 0; lf[q] = 0; sr[q] = 0;
          if (svv[q] != 0xffffffffu) {
            const KdSeg& g = seg[svv[q]];
            act[q] = g.split != 0u; gd[q] = g.dim; gf[q] = g.first; cn[q] = g.count; lf[q] = g.left; sr[q] = g.rank;
          }
        }
        bool any = false;
#pragma unroll
        for (int q = 0; q < kE; ++q) {
          const uint32_t w0 = wbase + (uint32_t)((q >> 1) * kKdThreads);
          const uint32_t pos = w0 + 64u * (q & 1) + lane;
          own[q] = false; kq[q] = ~0ull; fl[q] = 0;
          if (act[q]) {
            const uint32_t key = kd_key(kd_coord(pp[q], gd[q])), idx = (uint32_t)__float_as_int(pp[q].w);
            kq[q] = ((unsigned long long)key << 24) | (unsigned long long)idx;
            own[q] = gf[q] >= w0 && gf[q] < w0 + 64u;
            fl[q] = gf[q] - w0;
          }
          if ((q & 1) == 0 && pos < (uint32_t)n && !act[q]) { oth[pos] = pp[q]; sid_o[pos] = 0xffffffffu; }   // in a leaf (now or earlier): stays where it is for good
          wk[128 * (q >> 1) + 64 * (q & 1) + lane] = kq[q];
          any = any || own[q];
        }
        if (__ballot(any) == 0ull) continue;                  // wave-uniform
        uint32_t rk[kE];
#pragma unroll
        for (int q = 0; q < kE; ++q) rk[q] = 0u;
        for (uint32_t t = 0; t < steps; ++t) {
#pragma unroll
          for (int q = 0; q < kE; ++q) {
            const bool in = own[q] && t < cn[q];
            const unsigned long long o = wk[128 * (q >> 1) + (in ? fl[q] + t : 0u)];
            rk[q] += (in && o < kq[q]) ? 1u : 0u;
          }
        }
#pragma unroll
        for (int q = 0; q < kE; ++q) {
          if (!own[q]) continue;
          const uint32_t dst = wbase + (uint32_t)((q >> 1) * kKdThreads) + fl[q] + rk[q];
          oth[dst] = pp[q];
          sid_o[dst] = 2 * sr[q] + (rk[q] < lf[q] ? 0u : 1u);
          if (rk[q] == lf[q]) seg[svv[q]].prefix = (uint32_t)(kq[q] >> 24);      // the nth element: its coordinate is the cut value
        }
      }
      __syncthreads();
      SMHIP_KDPH(8);
    } else {
    SMHIP_KDPH(0);
    // ---- exact radix select of the element of rank `left` on the cut coordinate, all segments of a group at once
    // 8-bit digits while a level has <= 1024 segments: beyond 64 of them the level is swept in groups of 64 segments, each group
    // over its own positions only -- 4 short sweeps per group instead of 8 (4-bit digits) over every point of the cloud
    constexpr int kSU = SMHIP_KD_SU;                         // positions per thread and trip of a sweep
    const int bits = S <= 1024 ? 8 : (S <= 4096 ? 4 : (S <= 8192 ? 2 : 1));
    const int G = kKdHistWords >> bits;                      // segments per group
    const uint32_t mask = (1u << bits) - 1u;
    for (int g0 = 0; g0 < S; g0 += G) {
      const int g1 = min(S, g0 + G);
      const uint32_t p_lo = seg[g0].first, p_hi = seg[g1 - 1].first + seg[g1 - 1].count;
      for (int pass = 0; pass < 2; ++pass) {                 // pass 0: the coordinate key; pass 1 (ties only): the caller index
        if (pass == 1) {
          // does any segment of the group have several points ON its median value of which some must go left?
          if (tid == 0) s_misc[1] = 0;
          __syncthreads();
          for (int s = g0 + tid; s < g1; s += kKdThreads) {
            KdSeg& g = seg[s];
            g.tie = (g.split && g.neq > 1 && g.k > 0) ? 1u : 0u;
            g.trank = g.k;
            g.vidx = 0;                                      // doubles as the prefix of the index select
            if (g.tie) s_misc[1] = 1;
          }
          __syncthreads();
          if (!s_misc[1]) break;                             // block-uniform
        }
        for (int shift = 32 - bits; shift >= 0; shift -= bits) {
          for (int k = tid; k < (g1 - g0) << bits; k += kKdThreads) s_hist[k] = 0;
          // a group of 8-bit digits is at most 64 segments: their select state goes to LDS for the sweep (per position it was a
          // dependent global load between the segment id and the histogram atomic)
          const bool staged = bits == 8;
          if (staged && tid < g1 - g0) {
            const KdSeg& g = seg[g0 + tid];
            s_gst[tid] = g.prefix; s_gst[64 + tid] = pass == 0 ? g.split : (g.split & g.tie); s_gst[128 + tid] = g.vidx;
          }
          __syncthreads();
          // kSU positions per thread and trip, their levels of loads (segment id + key, segment state) issued together:
          // one position at a time every visit was a chain of dependent memory latencies
          for (uint32_t pos0 = p_lo + tid; pos0 < p_hi; pos0 += kSU * kKdThreads) {
            uint32_t sv[kSU], kv[kSU];
#pragma unroll
            for (int u = 0; u < kSU; ++u) {
              const uint32_t pos = pos0 + u * kKdThreads;
              sv[u] = pos < p_hi ? sid[pos] : 0xffffffffu;
              kv[u] = kk[min(pos, p_hi - 1u)];
            }
            uint32_t gpre[kSU], gsel[kSU], gvid[kSU];
#pragma unroll
            for (int u = 0; u < kSU; ++u) {
              if (staged) {
                const uint32_t sl = sv[u] == 0xffffffffu ? 0u : sv[u] - (uint32_t)g0;
                gpre[u] = s_gst[sl]; gvid[u] = s_gst[128 + sl];
                gsel[u] = sv[u] == 0xffffffffu ? 0u : s_gst[64 + sl];
              } else {
                const KdSeg& g = seg[sv[u] == 0xffffffffu ? (uint32_t)g0 : sv[u]];
                gpre[u] = g.prefix; gvid[u] = g.vidx;
                gsel[u] = sv[u] == 0xffffffffu ? 0u : (pass == 0 ? g.split : (g.split & g.tie));
              }
            }
#pragma unroll
            for (int u = 0; u < kSU; ++u) {
              bool sel = gsel[u] != 0u;
              uint32_t bin = 0;
              const uint32_t key = kv[u];
              if (pass == 0) {
                sel = sel && !(shift + bits < 32 && (key >> (shift + bits)) != (gpre[u] >> (shift + bits)));
                bin = ((sv[u] - g0) << bits) + ((key >> shift) & mask);
              } else {
                sel = sel && key == gpre[u];
                uint32_t ik = 0;
                if (sel) ik = (uint32_t)__float_as_int(cur[pos0 + u * kKdThreads].w);           // (ties only: the point's index in its cloud)
                sel = sel && !(shift + bits < 32 && (ik >> (shift + bits)) != (gvid[u] >> (shift + bits)));
                bin = ((sv[u] - g0) << bits) + ((ik >> shift) & mask);
              }
              // neighbours in the working order are neighbours in space: on the upper levels (and in every segment's leading
              // digits) a whole wave lands in one or two bins, and 64 atomics on one LDS word run one after the other -- the
              // lanes that share the first two bins met send one atomic each
              unsigned long long rem = __ballot(sel);
#pragma unroll
              for (int r = 0; r < 2; ++r) {
                if (!rem) break;                                                  // wave-uniform
                const int lead = __ffsll((long long)rem) - 1;
                const uint32_t bl = (uint32_t)__builtin_amdgcn_readlane((int)bin, lead);
                const unsigned long long m = __ballot(sel && bin == bl);
                if ((tid & 63) == lead) atomicAdd(&s_hist[bl], (uint32_t)__popcll(m));
                sel = sel && bin != bl;
                rem &= ~m;
              }
              if (sel) atomicAdd(&s_hist[bin], 1u);
            }
          }
          __syncthreads();
          SMHIP_KDPH(2);
          if (bits == 8) {
            // few segments, 256 bins each: a wave per segment, four bins per lane, one scan (a single thread walking 256
            // bins took longer than the sweep over the points)
            const int lane = tid & 63;
            for (int s = g0 + (tid >> 6); s < g1; s += kKdThreads / 64) {        // wave-uniform
              KdSeg& g = seg[s];
              if (!g.split || (pass == 1 && !g.tie)) continue;
              const uint32_t* hh = &s_hist[(s - g0) << 8];
              const uint32_t c0 = hh[4 * lane], c1 = hh[4 * lane + 1], c2 = hh[4 * lane + 2], c3 = hh[4 * lane + 3];
              const uint32_t want = pass == 0 ? g.k : g.trank;
              const uint32_t incl = wave_incl_scan(c0 + c1 + c2 + c3, lane);
              const uint32_t excl = incl - (c0 + c1 + c2 + c3);
              if (excl <= want && want < incl) {                                  // exactly one lane
                uint32_t cum = excl, d = 4u * lane, c = c0;
                if (cum + c <= want) { cum += c; ++d; c = c1; }
                if (cum + c <= want) { cum += c; ++d; c = c2; }
                if (cum + c <= want) { cum += c; ++d; c = c3; }
                if (pass == 0) { g.prefix |= d << shift; g.nless += cum; g.k -= cum; g.neq = c; }
                else { g.vidx |= d << shift; g.trank -= cum; }
              }
            }
          } else {
            for (int s = g0 + tid; s < g1; s += kKdThreads) {
              KdSeg& g = seg[s];
              if (!g.split || (pass == 1 && !g.tie)) continue;
              const uint32_t* hh = &s_hist[(s - g0) << bits];
              const uint32_t want = pass == 0 ? g.k : g.trank;  // rank among the still-matching keys
              uint32_t cum = 0;
              for (uint32_t d = 0; d <= mask; ++d) {
                const uint32_t c = hh[d];
                if (cum + c > want) {
                  if (pass == 0) { g.prefix |= d << shift; g.nless += cum; g.k -= cum; g.neq = c; }
                  else { g.vidx |= d << shift; g.trank -= cum; }
                  break;
                }
                cum += c;
              }
            }
          }
          __syncthreads();
          SMHIP_KDPH(3);
        }
      }
    }
    // After pass 0: prefix = key of the median element, nless = keys below it, neq = keys equal to it, k = how many of the
    // equal ones go LEFT (0 for tie-free data).  After pass 1 (ties): vidx = caller index of the first equal point that
    // goes right.  Without ties vidx stays 0: no equal point goes left.

    // ---- the children's cut dimensions (their boxes: the parent's with the cut value on one side), for the keys the partition writes
    const bool next_ranks = (s_misc[2] + 1u) / 2u <= (uint32_t)kKdCountMax;    // the next level ranks its segments inside waves: it reads no keys
    if (!next_ranks) {
      for (int s = s_lo; s < s_hi; ++s) {
        KdSeg& g = seg[s];
        if (!g.split) continue;
        const float cut = kd_unkey(g.prefix);
        for (int side = 0; side < 2; ++side) {
          uint32_t cd = 0; float mv = 0.f; double mvw = 0.0;   // (the child's own choice one level on, the same way)
          for (uint32_t d = 0; d < 3; ++d) {
            const float lo = (side == 1 && d == g.dim) ? cut : g.mn[d], hi = (side == 0 && d == g.dim) ? cut : g.mx[d];
            if (WIDE) { const double e = (double)hi - (double)lo; if (e > mvw) { mvw = e; cd = d; } }
            else { const float e = hi - lo; if (e > mv) { mv = e; cd = d; } }
          }
          if (side == 0) g.ldim = cd; else g.rdim = cd;
        }
      }
      __syncthreads();
    }
    SMHIP_KDPH(4);
    // ---- partition into the other buffer; children become the next level's segments
    const bool lds_counters = 2 * S <= kKdHistWords;
    uint32_t* cnt = lds_counters ? s_hist : cnt_global;
    for (int k = tid; k < 2 * S; k += kKdThreads) cnt[k] = 0;
    // while the segments' fields fit next to the fill counters (levels of up to 1 489 segments) the pass reads them from LDS
    const bool pstaged = 11 * S <= kKdHistWords;
    uint32_t* sf = s_hist + 2 * S;                             // [9][S]: split, dim, prefix, vidx, first, left, rank, ldim, rdim
    if (pstaged)
      for (int k = tid; k < S; k += kKdThreads) {
        const KdSeg& g = seg[k];
        sf[k] = g.split; sf[S + k] = g.dim; sf[2 * S + k] = g.prefix; sf[3 * S + k] = g.vidx; sf[4 * S + k] = g.first;
        sf[5 * S + k] = g.left; sf[6 * S + k] = g.rank; sf[7 * S + k] = g.ldim; sf[8 * S + k] = g.rdim;
      }
    __syncthreads();
    constexpr int kPU = SMHIP_KD_PU;                          // positions per thread and trip, their loads issued together
    for (uint32_t pos0 = 0; pos0 < (uint32_t)n; pos0 += kPU * kKdThreads) {     // whole waves take the trip together (ballots below)
      uint32_t sq[kPU];
      float4 pq[kPU];
#pragma unroll
      for (int u = 0; u < kPU; ++u) {
        const uint32_t pos = pos0 + u * kKdThreads + tid;
        sq[u] = 0xffffffffu;
        pq[u] = make_float4(0, 0, 0, 0);
        if (pos < (uint32_t)n) { sq[u] = sid[pos]; pq[u] = cur[pos]; }
      }
#pragma unroll
      for (int u = 0; u < kPU; ++u) {
      const uint32_t pos = pos0 + u * kKdThreads + tid;
      const bool live = pos < (uint32_t)n;
      const uint32_t s = sq[u];
      const float4 p = pq[u];
      bool moving = false, left = false;
      uint32_t first = 0, nleft = 0, rank = 0, ndim = 0;
      if (s != 0xffffffffu) {
        uint32_t g_split, g_dim, g_prefix, g_vidx, g_ldim, g_rdim;
        if (pstaged) {
          g_split = sf[s]; g_dim = sf[S + s]; g_prefix = sf[2 * S + s]; g_vidx = sf[3 * S + s]; first = sf[4 * S + s];
          nleft = sf[5 * S + s]; rank = sf[6 * S + s]; g_ldim = sf[7 * S + s]; g_rdim = sf[8 * S + s];
        } else {
          const KdSeg& g = seg[s];
          g_split = g.split; g_dim = g.dim; g_prefix = g.prefix; g_vidx = g.vidx; first = g.first; nleft = g.left; rank = g.rank;
          g_ldim = g.ldim; g_rdim = g.rdim;
        }
        if (g_split) {
          moving = true;
          const uint32_t key = kd_key(kd_coord(p, g_dim));
          left = key < g_prefix || (key == g_prefix && (uint32_t)__float_as_int(p.w) < g_vidx);
          ndim = left ? g_ldim : g_rdim;
        }
      }
      if (live && !moving) { oth[pos] = p; sid_o[pos] = 0xffffffffu; }   // in a leaf (now or earlier): stays where it is for good
      // fill counters: one atomic per wave and side when the wave's moving lanes share a segment (always so on the upper
      // levels, where the same two counters would otherwise take every point of the cloud), one per lane otherwise
      const unsigned long long mm = __ballot(moving);
      if (mm) {
        const int lane = tid & 63;
        const int lead = __ffsll((long long)mm) - 1;
        const uint32_t s_lead = (uint32_t)__shfl((int)s, lead, 64);
        const bool uniform = __ballot(moving && s != s_lead) == 0ull;
        uint32_t np = 0;
        if (uniform) {
          const unsigned long long ml = __ballot(moving && left), mr = mm & ~ml;
          uint32_t bl = 0, br = 0;
          if (lane == lead) {
            if (ml) bl = atomicAdd(&cnt[2 * s], (uint32_t)__popcll(ml));
            if (mr) br = atomicAdd(&cnt[2 * s + 1], (uint32_t)__popcll(mr));
          }
          bl = (uint32_t)__shfl((int)bl, lead, 64); br = (uint32_t)__shfl((int)br, lead, 64);
          const unsigned long long below = (1ull << lane) - 1ull;
          if (moving) np = left ? first + bl + (uint32_t)__popcll(ml & below) : first + nleft + br + (uint32_t)__popcll(mr & below);
        } else if (moving) {
          np = left ? first + atomicAdd(&cnt[2 * s], 1u) : first + nleft + atomicAdd(&cnt[2 * s + 1], 1u);
        }
        if (moving) {
          oth[np] = p; sid_o[np] = 2 * rank + (left ? 0u : 1u);
          if (!next_ranks) kk_o[np] = kd_key(kd_coord(p, ndim));
        }
      }
      }
    }
    __syncthreads();
    SMHIP_KDPH(5);
    }   // radix select + partition
    __syncthreads();
    for (int s = tid; s < S; s += kKdThreads) {                // (strided: neighbouring lanes, neighbouring segments)
      const KdSeg& g = seg[s];
      if (!g.split) continue;
      const float cut = kd_unkey(g.prefix);
      nodes[g.node] = make_uint2(__float_as_uint(cut), ((nc + 2 * g.rank) << 2) | g.dim);
      KdSeg l{}, r{};
      l.first = g.first; l.count = g.left; l.node = nc + 2 * g.rank;
      r.first = g.first + g.left; r.count = g.count - g.left; r.node = nc + 2 * g.rank + 1;
      for (int d = 0; d < 3; ++d) { l.mn[d] = g.mn[d]; l.mx[d] = g.mx[d]; r.mn[d] = g.mn[d]; r.mx[d] = g.mx[d]; }
      l.mx[g.dim] = cut; r.mn[g.dim] = cut;
      seg_o[2 * g.rank] = l; seg_o[2 * g.rank + 1] = r;
    }
    __syncthreads();
    if (tid == 0) s_misc[0] = nc + 2 * nsplit;
    { float4* t4 = cur; cur = oth; oth = t4; }
    { uint32_t* t1 = sid; sid = sid_o; sid_o = t1; }
    { uint32_t* t1 = kk; kk = kk_o; kk_o = t1; }
    { KdSeg* ts = seg; seg = seg_o; seg_o = ts; }
    S = 2 * (int)nsplit;
    __syncthreads();
    SMHIP_KDPH(6);
  }
#ifdef SMHIP_KD_TIMING
  if (tid == 0 && blockIdx.x == 0)
    printf("[kd timing, n %d] setup %.0f us, small-segment levels %.0f, select sweeps %.0f, select scans %.0f, child dims %.0f, partition %.0f, next segments %.0f, level head %.0f\n", n,
           s_kdph[0] * 0.01, s_kdph[8] * 0.01, s_kdph[2] * 0.01, s_kdph[3] * 0.01, s_kdph[4] * 0.01, s_kdph[5] * 0.01, s_kdph[6] * 0.01, s_kdph[7] * 0.01);
#endif
}

}  // namespace smhip
