// nabo_kernels.hip -- the reference's OWN nearest-neighbour search on the device (nn_mode = SMHIP_NN_NABO).
//
// IcpFast asks libnabo for an APPROXIMATE neighbour: NNS::create(points, dim, KDTREE_LINEAR_HEAP) rebuilt in every Align
// (/root/reference/registrators/icp_fast.cc:464-467) and knn(query, ids, dists, 1, epsilon = 3.16, ALLOW_SELF_MATCH)
// in every iteration (:169-180).  With epsilon = 3.16 a subtree is skipped unless it could hold a point (1 + 3.16)^2 =
// 17.3 times closer (squared) than the best so far, so what comes back is often NOT the nearest neighbour, and a whole
// Align ends 3-4 mm away from one that uses the exact neighbour -- more than the 1e-3 m the
// north star allows.  The exact grid search (icp_kernels.hip) therefore cannot land within tolerance of a libnabo build;
// this file walks libnabo's tree with libnabo's rule instead.
//
// libnabo is a git dependency pinned to tags/1.0.7 (setup/install_libnabo.sh:16-18), not vendored; the algorithm is the
// published one of nabo/kdtree_cpu.cpp (class KDTreeUnbalancedPtInLeavesImplicitBoundsStackOpt), as restated and
// cross-checked in oracle/nabo.py and oracle/csrc/smref_icp.c (nabo_*):
//   buildNodes  leaf when count <= bucketSize (8); cut dimension = argMax(maxValues - minValues) of the box INHERITED from
//               the parent (root: the cloud's bounds; a child gets the parent's box cut at cutVal), argMax starting from
//               (index 0, value 0); leftCount = count - count / 2; std::nth_element at first + leftCount; cutVal = that
//               element's coordinate.
//   recurseKnn  leaf: every bucket entry with dist < best replaces it; inner node: new_off = q[cd] - cutVal, the side of
//               the query first, then rd += new_off^2 - old_off^2 and the other side only if rd (1 + eps)^2 < best.
// The tree depends only on which points fall on which side of each median (tie-free data: uniquely), so it is built
// level by level with an exact radix select per segment instead of a recursive nth_element; the search keeps libnabo's
// visiting order (deepest pending sibling first) with an explicit stack of root paths.
// Arithmetic: float32 on the centred target / transformed query, like every other search here (libnabo: float64): a
// pruning test within one float ulp of its threshold can fall the other way -- tests/test_nabo_gpu.py counts how often.
#pragma once
#include "smhip_device.h"
#include "kd_median_tree.h"

namespace smhip {

constexpr int kKdBucket = 8;              // libnabo's default bucketSize
constexpr int kKdStack = 18;              // pending siblings per query: at most one per tree level (18 levels: > 1 M target points; node < 2^23)
constexpr int kNaboListedBlocks = 96;     // workgroups per pair striding over the lists of queries to walk again (32: -2 %, 64: -1 %, 128: equal)
constexpr int kKdTopNodes = 511;          // tree levels 0..8 staged in LDS by the search kernel (4 KiB)

struct KdDev {
  uint2* nodes;        // [slots][kd_node_cap]  inner: {cut value bits, (left child << 2) | dim}; leaf: {first, (count << 2) | 3}
  KdSeg* segs;         // [slots][2][kd_seg_cap]
  float4* alt;         // [slots][nt_cap]       second working order (ping-pong with tq)
  uint32_t* cnt;       // [slots][2 * seg_cap]  left / right fill counters of a level with more segments than LDS holds
  float* leaf;         // [slots][leaf_cap][24]  the buckets again, as the search scans them: x[8] y[8] z[8] of the bucket that starts at
                       //                        tq position `first` in block first >> 2 (a bucket of a split cloud holds >= 4 points, so
                       //                        blocks are unique), unused entries = +inf (their distance is +inf: never a candidate)
  int32_t node_cap, seg_cap, leaf_cap;
  float max_error2;    // (1 + epsilon)^2
};
// rd of a sibling: the squared distance to its half-space box, updated exactly as recurseKnn does (no contraction, so
// that the pre-filter at push time and the test at pop time see the same number)
__device__ __forceinline__ float kd_rd_step(float rd, float old_off, float new_off) {
  return __fadd_rn(rd, __fadd_rn(-__fmul_rn(old_off, old_off), __fmul_rn(new_off, new_off)));
}

// Build: grid = (pairs), 1024 threads.  Needs tgt_reduce + grid_setup to have run (st->mu, st->nt).
__global__ __launch_bounds__(kKdThreads) void kd_build(IcpDev b, KdDev kd) {
  const int pair = b.pair_base + blockIdx.x;
  PairState* st = &b.state[pair];
  if (st->done) return;
  const int n = st->nt;
  const int tid = threadIdx.x;
  __shared__ __attribute__((aligned(16))) uint32_t s_hist[kKdHistWords];   // (kd_median_build also lays 8-byte keys in it)
  __shared__ uint32_t s_w[17];
  __shared__ float s_box[6][16];
  __shared__ uint32_t s_misc[4];
  const size_t to = (size_t)pair * b.nt_cap;
  float4* cur = b.tq + to;                                   // working order: centred point, w = caller index
  float4* oth = kd.alt + to;
  uint32_t* sid = b.tcell + to;                              // segment of every position (0xffffffff = already in a leaf)
  uint32_t* sid_o = b.tslot + to;
  uint32_t* kk = b.tord + to;                                // the select's keys (the grid's arrays are free in this mode)
  uint32_t* kk_o = b.ccount + (size_t)pair * (b.nt_cap + 1);
  KdSeg* seg = kd.segs + (size_t)pair * 2 * kd.seg_cap;
  KdSeg* seg_o = seg + kd.seg_cap;
  uint2* nodes = kd.nodes + (size_t)pair * kd.node_cap;
  const double mu[3] = {st->mu[0], st->mu[1], st->mu[2]};

  // ---- centred points in caller order + the cloud's bounds (the root's box)
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = tid; i < n; i += kKdThreads) {
    const float3 c = centre_point(b.tgt_p[to + i], mu);
    cur[i] = make_float4(c.x, c.y, c.z, __int_as_float(i));
    sid[i] = 0;
    mn[0] = fminf(mn[0], c.x); mn[1] = fminf(mn[1], c.y); mn[2] = fminf(mn[2], c.z);
    mx[0] = fmaxf(mx[0], c.x); mx[1] = fmaxf(mx[1], c.y); mx[2] = fmaxf(mx[2], c.z);
  }
  for (int d = 0; d < 3; ++d) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
  if ((tid & 63) == 0) for (int d = 0; d < 3; ++d) { s_box[d][tid >> 6] = mn[d]; s_box[3 + d][tid >> 6] = mx[d]; }
  __syncthreads();
  if (tid == 0) {
    KdSeg r{};
    r.first = 0; r.count = (uint32_t)n; r.node = 0;
    for (int d = 0; d < 3; ++d) {
      float a = s_box[d][0], c = s_box[3 + d][0];
      for (int w = 1; w < kKdThreads / 64; ++w) { a = fminf(a, s_box[d][w]); c = fmaxf(c, s_box[3 + d][w]); }
      r.mn[d] = a; r.mx[d] = c;
    }
    seg[0] = r;
    s_misc[0] = 1;                                           // node count
  }
  __syncthreads();

  // ---- libnabo's buildNodes level by level (kd_median_tree.h; bucketSize 8)
  const float4* raw_t = b.tgt_p + to;
  auto fetch = [&](uint32_t i) { const float3 c = centre_point(raw_t[i], mu); return make_float4(c.x, c.y, c.z, __int_as_float((int)i)); };
  kd_median_build<kKdBucket, false>(n, fetch, cur, oth, sid, sid_o, seg, seg_o, kk, kk_o, nodes, kd.cnt + (size_t)pair * 2 * kd.seg_cap, kd.seg_cap, kd.node_cap,
                             s_hist, s_w, s_misc, &st->status);
  // ---- final order into tq (+ normals), bucket entries by caller index (a deterministic stand-in for nth_element's
  // unspecified order inside a bucket; it only matters for exactly equidistant entries)
  float4* tq = b.tq + to;
  if (cur != tq) {
    for (int i = tid; i < n; i += kKdThreads) tq[i] = cur[i];
    __syncthreads();
  }
  const int nn = (int)s_misc[0];
  for (int v = tid; v < nn; v += kKdThreads) {
    const uint2 nd = nodes[v];
    if ((nd.y & 3u) != 3u) continue;
    const uint32_t f = nd.x, c = nd.y >> 2;
    for (uint32_t a = f + 1; a < f + c; ++a) {
      const float4 kq = tq[a];
      const int key = __float_as_int(kq.w);
      uint32_t p = a;
      while (p > f && __float_as_int(tq[p - 1].w) > key) { tq[p] = tq[p - 1]; --p; }
      tq[p] = kq;
    }
  }
  __syncthreads();
  float* leafs = kd.leaf + (size_t)pair * kd.leaf_cap * 24;
  for (int v = tid; v < nn; v += kKdThreads) {
    const uint2 nd = nodes[v];
    if ((nd.y & 3u) != 3u) continue;
    const uint32_t f = nd.x, c = nd.y >> 2;
    float* blk = leafs + (size_t)(f >> 2) * 24;
    for (uint32_t e = 0; e < (uint32_t)kKdBucket; ++e) {
      const float4 p = e < c ? tq[f + e] : make_float4(INFINITY, INFINITY, INFINITY, 0.f);
      blk[e] = p.x; blk[8 + e] = p.y; blk[16 + e] = p.z;
    }
  }
  float4* tn = b.tn + to;
  for (int i = tid; i < n; i += kKdThreads) {
    float4 nr = b.tgt_n[to + __float_as_int(tq[i].w)];
    nr.w = 0.f;
    tn[i] = nr;
  }
  if (tid == 0) st->nocc = nn;                               // node count (the grid's occupied-cell count is unused in this mode)
}

// ------------------------------------------------------------------------------------------------------------------
// knn(k = 1, epsilon, ALLOW_SELF_MATCH): one query per lane, libnabo's visiting order.
//
// Traversal certificates.  libnabo's answer is a deterministic function of the query position: the walk is a sequence of
//   side decisions   new_off > 0 at every inner node it enters (which child is "near"),
//   prune decisions  rd (1 + eps)^2 < best for every sibling it leaves behind, with the best distance found so far,
// and the answer is the first smallest entry of the buckets it scans.  While every one of those comparisons keeps its
// outcome the walk visits the same buckets in the same order, and while the winner of the scanned entries stays ahead
// of the runner-up it returns the same id.  All of them are comparisons of 1-Lipschitz functions of the query:
//   |new_off| changes by at most the query's motion m;  sqrt(rd) (the distance to the sibling's half-space corner) and
//   sqrt(best) (a minimum of point distances over the same scanned set) by at most m each, so a prune decision holds
//   while |sqrt(rd) (1 + eps) - sqrt(best)| > (2 + eps) m;  the winner holds while runner-up - winner > 2 m.
// The walk therefore records the smallest such slack (in metres of query motion, less the float roundings of the
// quantities compared) next to the match, stored like the exact search's bounds with the pair's motion potential added
// (icp_kernels.hip, with_pot / bound_now).  In the following iterations nn_certify<., true> re-derives nothing: a query
// whose accumulated motion is below its slack provably gets the same id from the same walk, so only its distance to
// that id is recomputed (the same dist2 the bucket scan uses: the same bits); the others are compacted and walked again
// (nn_nabo<., true>).  tests/test_nabo_gpu.py::test_nabo_certificates_change_no_bit.
// ------------------------------------------------------------------------------------------------------------------
struct KdNodeView { uint32_t dim, left; float cut; };   // leaf: dim == 3, left = count, cut bits = first

// lower bound on |sqrt(x) - sqrt(y)| for x, y >= 0 computed as floats with a relative error of at most 1e-5 each
// (rd: <= 18 un-contracted update steps; best: three fused multiply-adds): |x - y| / (2 sqrt(max)) less that error
__device__ __forceinline__ float kd_sqrt_gap(float x, float y) {
  const float m = fmaxf(fmaxf(x, y), 1.0e-30f);
  return (fabsf(x - y) - 2.0e-5f * m) * (0.5f * __frsqrt_rn(m));
}

// LISTED = false: queries [blk * 256 * ITEMS, ...) of the pair (iteration 0, find_closests, certificates off);
// LISTED = true: the queries nn_certify<., true> could not certify (dlist), the pair's nblk workgroups striding over the list
// STACK = pending-sibling slots per query (one per tree level): 12 covers targets of up to 8 << 12 points and leaves the
// workgroup at 20 KiB of LDS and 64 VGPRs (8 waves per SIMD; the 18 of the general case: 26 KiB, 6 workgroups per CU).
// Measured on 512 x 120 k queries, every query walked in 20 iterations: 3 workgroups per CU (52 KiB: two-word stack entries,
// 8 KiB histogram) 355 ms, 4 (40 KiB) 264 ms, 6 (24 KiB) 156 ms, 8 (20 KiB, 64 VGPRs, 24 B of scratch) 148 ms.
template <int ITEMS, bool LISTED, int STACK>
__global__ __launch_bounds__(kNnThreads, STACK <= 12 ? 8 : 6) void nn_nabo(IcpDev b, KdDev kd, int nblk) {
  int pair, blk;
  if (!xcd_block(nblk, b.npairs, b.pair_base, pair, blk)) return;
  PairState* st = &b.state[pair];
  if (st->done) return;
  const int ns = st->ns;
  // LISTED: the four class lists laid end to end, heaviest class first, each padded to whole waves -- a wave's queries all
  // come from one class (its walk costs the longest of its lanes' walks)
  int cls_end[4] = {0, 0, 0, 0}, cls_n[4] = {0, 0, 0, 0};
  int count = ns;
  if (LISTED) {
    int acc = 0;
#pragma unroll
    for (int c = 3; c >= 0; --c) { cls_n[c] = (int)st->nabo_count[c]; acc += (cls_n[c] + 63) & ~63; cls_end[c] = acc; }
    count = acc;
  }
  const int base0 = blk * (kNnThreads * ITEMS);
  if (base0 >= count) return;
  double Mc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) Mc[k] = st->M[k];
  const Pot pot = {(float)st->pot_a, (float)st->pot_b, 0.f, 0.f};
  // The kernel's speed is its occupancy (measured: 3 -> 4 workgroups per CU = 1.34 x faster; a wave's walk is a chain of
  // dependent lookups), and its occupancy is its LDS: 12 KiB of stack (ONE word per pending sibling), 8 KiB of top tree
  // levels and a 4 KiB histogram with two 16-bit bins per word (a workgroup adds < 65 536 per bin) = 24 KiB, 6 per CU.
  __shared__ uint32_t s_hist[kHistBins / 2];
  __shared__ uint2 s_top[kKdTopNodes + 1];
  // pending siblings: (tree level << 25) | PARENT node for a sibling of the main path -- which child it is, the cut dimension
  // and the query's offset from the cut plane are re-read from the parent when its turn comes -- or 0x80000000 | root path
  // for a sibling met while exploring another sibling's subtree
  __shared__ uint32_t s_stack[STACK][kNnThreads];
  const uint2* __restrict__ nodes = kd.nodes + (size_t)pair * kd.node_cap;
  const float4* __restrict__ leafs = reinterpret_cast<const float4*>(kd.leaf + (size_t)pair * kd.leaf_cap * 24);
  const int n_nodes = st->nocc;
  for (int k = threadIdx.x; k < kHistBins / 2; k += kNnThreads) s_hist[k] = 0;
  for (int k = threadIdx.x; k < min(n_nodes, kKdTopNodes); k += kNnThreads) s_top[k] = nodes[k];
  __syncthreads();
  const float E2 = kd.max_error2;
  const float inv_prune = 1.0f / (1.0f + sqrtf(E2) * 1.000001f);   // a prune decision moves by at most (1 + (1 + eps)) x the motion
  const size_t so = (size_t)pair * b.ns_cap;
  const int t = threadIdx.x;
  auto node_at = [&](uint32_t v) -> uint2 { return v < (uint32_t)kKdTopNodes ? s_top[v] : nodes[v]; };

  for (int base = base0; base < count; base += LISTED ? nblk * kNnThreads : kNnThreads) {
    if (!LISTED && base >= base0 + kNnThreads * ITEMS) break;
    const int e_ = base + t;
    if (e_ >= count) continue;
    int i = e_;
    if (LISTED) {
      const int c = e_ < cls_end[3] ? 3 : (e_ < cls_end[2] ? 2 : (e_ < cls_end[1] ? 1 : 0));
      const int k = e_ - (cls_end[c] - ((cls_n[c] + 63) & ~63));
      if (k >= cls_n[c]) continue;                            // padding of the class's last wave
      i = *nabo_list_slot(b, so, c, (uint32_t)k);
    }
    const float4 s4 = ld_src(b, so + i);
    double px, py, pz;
    transform_point(Mc, s4, px, py, pz);
    const float q[3] = {(float)px, (float)py, (float)pz};
    float best = INFINITY, second = INFINITY;                // smallest and second-smallest distance over every scanned entry
    int bestj = -1;
    uint32_t buckets = 0;                                    // buckets scanned: the cost class of the query's next walk
    float side = INFINITY;                                   // smallest |new_off| over the inner nodes entered
    float prune = INFINITY;                                  // smallest |sqrt(rd E2) - sqrt(best)| over the prune decisions
    if (isfinite(q[0]) && isfinite(q[1]) && isfinite(q[2]) && n_nodes > 0) {
      // a bucket is scanned from its x[8] y[8] z[8] block: six 16-byte loads issued together (one memory latency per
      // bucket), two entries per packed-fp32 instruction, no count mask (unused entries are +inf away).  Entries are tested
      // in bucket order with a strict "<", as libnabo's leaf loop does; each distance is dist2()'s expression, bit for bit
      // (nn_certify recomputes a certified query's distance with it)
      float rd_pruned = INFINITY;                             // nearest sibling pruned since the last bucket: best is constant in between,
      auto fold_pruned = [&]() {                              // so one gap stands for all of them
        if (rd_pruned < INFINITY) prune = fminf(prune, kd_sqrt_gap(rd_pruned, best));
        rd_pruned = INFINITY;
      };
      const f32x2 qx2 = {q[0], q[0]}, qy2 = {q[1], q[1]}, qz2 = {q[2], q[2]};
      auto scan_leaf = [&](uint2 nd) {
        const uint32_t f = nd.x;
        ++buckets;
        fold_pruned();
        const float4* blk = leafs + (size_t)(f >> 2) * 6;
        const float4 xa = blk[0], xb = blk[1], ya = blk[2], yb = blk[3], za = blk[4], zb = blk[5];
        const f32x2 xs[4] = {{xa.x, xa.y}, {xa.z, xa.w}, {xb.x, xb.y}, {xb.z, xb.w}};
        const f32x2 ys[4] = {{ya.x, ya.y}, {ya.z, ya.w}, {yb.x, yb.y}, {yb.z, yb.w}};
        const f32x2 zs[4] = {{za.x, za.y}, {za.z, za.w}, {zb.x, zb.y}, {zb.z, zb.w}};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f32x2 dx = qx2 - xs[k], dy = qy2 - ys[k], dz = qz2 - zs[k];
          f32x2 d = dx * dx;
          d = __builtin_elementwise_fma(dy, dy, d);
          d = __builtin_elementwise_fma(dz, dz, d);
          second = __builtin_amdgcn_fmed3f(d.x, best, second);   // best <= second always: the median of the three is the new runner-up
          if (d.x < best) { best = d.x; bestj = (int)(f + 2 * k); }
          second = __builtin_amdgcn_fmed3f(d.y, best, second);
          if (d.y < best) { best = d.y; bestj = (int)(f + 2 * k + 1); }
        }
      };
      // (1) descent to the query's own leaf.  On this path no coordinate has an offset yet, so the sibling left behind at a
      //     level has rd = new_off^2 and the offset vector e_cd * new_off: (its node, cd) and new_off are parked in the
      //     lane's stack column, and once the leaf has been scanned the column is compacted in place to the siblings the
      //     leaf's distance cannot prune.
      uint32_t v = 0, path = 1u;
      int depth = 0;
      uint2 nd = node_at(v);
      while ((nd.y & 3u) != 3u) {
        const uint32_t cd = nd.y & 3u;
        const float no = (cd == 0 ? q[0] : (cd == 1 ? q[1] : q[2])) - __uint_as_float(nd.x);
        const uint32_t right = no > 0.f ? 1u : 0u;
        side = fminf(side, fabsf(no));
        if (depth < STACK) s_stack[depth][t] = ((uint32_t)depth << 25) | v;   // level | this node: the parent of the sibling left behind
        else side = -INFINITY;                               // deeper than the stack (the host refuses such targets): never certified
        ++depth;
        path = (path << 1) | right;
        v = (nd.y >> 2) + right;                              // children sit side by side: left, right
        nd = node_at(v);
      }
      scan_leaf(nd);
      int sp = 0;
      for (int l = 0; l < min(depth, STACK); ++l) {            // shallow to deep: the write position never passes the read position
        const uint32_t A = s_stack[l][t];
        const uint2 pn = node_at(A & 0x1ffffffu);
        const uint32_t cd = pn.y & 3u;
        const float no = (cd == 0 ? q[0] : (cd == 1 ? q[1] : q[2])) - __uint_as_float(pn.x);   // the descent's own expression: the same float
        const float rdE = kd_rd_step(0.f, 0.f, no) * E2;
        if (rdE < best) { s_stack[sp][t] = A; ++sp; }
        else rd_pruned = fminf(rd_pruned, rdE);               // pruned now = pruned at its turn (best only shrinks)
      }
      // (2) pending siblings, deepest first; the test is libnabo's, with the best AS OF NOW
      // The main-path sibling whose subtree is being explored: every nested sibling popped before the next main-path one
      // lies below it, so its state is re-derived from there (one offset, a handful of steps) instead of from the root.
      uint32_t a_v = 0, a_cd = 0; float a_no = 0.f; int a_dp = 0;   // node, its parent's cut dimension, the query's offset from that cut, depth
      while (sp > 0) {
        --sp;
        const uint32_t A = s_stack[sp][t];
        float off[3] = {0.f, 0.f, 0.f};
        float rd = 0.f;
        uint32_t pp;                                           // root path of the node being explored (for nested siblings)
        if (!(A & 0x80000000u)) {                              // a sibling of the main path: its state is one offset
          const uint2 pn = node_at(A & 0x1ffffffu);
          const uint32_t cd = pn.y & 3u;
          const float no = (cd == 0 ? q[0] : (cd == 1 ? q[1] : q[2])) - __uint_as_float(pn.x);
          rd = kd_rd_step(0.f, 0.f, no);
          if (!(rd * E2 < best)) { rd_pruned = fminf(rd_pruned, rd * E2); continue; }
          prune = fminf(prune, kd_sqrt_gap(rd * E2, best));
          if (cd == 0) off[0] = no; else if (cd == 1) off[1] = no; else off[2] = no;
          v = (pn.y >> 2) + (no > 0.f ? 0u : 1u);               // the child the descent did not take
          nd = node_at(v);
          const int lv = (int)(A >> 25);                       // its root path: the main path down to its level, last step flipped
          pp = (path >> (depth - lv - 1)) ^ 1u;
          a_v = v; a_no = no; a_cd = cd; a_dp = lv + 1;
        } else {                                               // met inside a main-path sibling's subtree: re-derive its state from that sibling
          const uint32_t P = A & 0x7fffffffu;
          v = a_v; nd = node_at(v);
          rd = kd_rd_step(0.f, 0.f, a_no);
          if (a_cd == 0) off[0] = a_no; else if (a_cd == 1) off[1] = a_no; else off[2] = a_no;
          const int dp = 31 - __clz((int)P);
          for (int l = dp - a_dp - 1; l >= 0; --l) {
            const uint32_t cd = nd.y & 3u;
            const float no = (cd == 0 ? q[0] : (cd == 1 ? q[1] : q[2])) - __uint_as_float(nd.x);
            const uint32_t near = no > 0.f ? 1u : 0u, bit = (P >> l) & 1u;
            if (bit != near) {
              const float oo = cd == 0 ? off[0] : (cd == 1 ? off[1] : off[2]);
              rd = kd_rd_step(rd, oo, no);
              if (cd == 0) off[0] = no; else if (cd == 1) off[1] = no; else off[2] = no;
            }
            v = (nd.y >> 2) + bit;
            nd = node_at(v);
          }
          if (!(rd * E2 < best)) { rd_pruned = fminf(rd_pruned, rd * E2); continue; }
          prune = fminf(prune, kd_sqrt_gap(rd * E2, best));
          pp = P;
        }
        // explore that subtree: near children first; its own far siblings are pushed as root paths (pre-filtered with the
        // current best)
        while ((nd.y & 3u) != 3u) {
          const uint32_t cd = nd.y & 3u;
          const float no = (cd == 0 ? q[0] : (cd == 1 ? q[1] : q[2])) - __uint_as_float(nd.x);
          const uint32_t right = no > 0.f ? 1u : 0u;
          side = fminf(side, fabsf(no));
          const float oo = cd == 0 ? off[0] : (cd == 1 ? off[1] : off[2]);
          const float rdf = kd_rd_step(rd, oo, no);
          if (rdf * E2 < best) {
            if (sp < STACK) { s_stack[sp][t] = 0x80000000u | (pp << 1) | (right ^ 1u); ++sp; }
            else side = -INFINITY;
          } else {
            rd_pruned = fminf(rd_pruned, rdf * E2);             // pruned with today's best = pruned at its turn
          }
          pp = (pp << 1) | right;
          v = (nd.y >> 2) + right;
          nd = node_at(v);
        }
        scan_leaf(nd);
      }
      fold_pruned();
    }
    b.d2[so + i] = best;
    b.idx[so + i] = bestj;
    b.nabo_work[so + i] = (uint8_t)min(buckets, 255u);
    // the certificate: how far the query may move before any decision of this walk, or its winner, can change
    float slack = 0.f;
    if (bestj >= 0) {
      const float win = second < INFINITY ? 0.5f * kd_sqrt_gap(second, best) : INFINITY;
      slack = fminf(fminf(side, prune * inv_prune), win);
      slack = fminf(slack, 1.0e3f);                            // keeps the stored sum a small float
      if (!(slack > 0.f)) slack = 0.f;                         // 0 = unknown: walked again next iteration
    }
    st_lb(b, so + i, with_pot(slack, pot_at(pot, norm3(s4.x, s4.y, s4.z))));
    const uint32_t key = __float_as_uint(best);
    if (key < 0x7f800000u) atomicAdd(&s_hist[key >> (kHistShift + 1)], 1u << (((key >> kHistShift) & 1u) << 4));
  }
  __syncthreads();
  uint32_t* gh = b.hist + (size_t)pair * kHistBins;
  for (int k = threadIdx.x; k < kHistBins / 2; k += kNnThreads) {
    const uint32_t v = s_hist[k];
    if (v & 0xffffu) atomicAdd(&gh[2 * k], v & 0xffffu);
    if (v >> 16) atomicAdd(&gh[2 * k + 1], v >> 16);
  }
}

}  // namespace smhip
