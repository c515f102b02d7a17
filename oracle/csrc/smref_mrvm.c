/* smref_mrvm.c -- TEST ORACLE (not product code): plain-C restatement of static_map::MultiResolutionVoxelMap, the
 * probabilistic hit / miss voxel map with ray casting behind the reference's static-map output
 * (/root/reference/builder/multi_resolution_voxel_map.{h,cc}; caller builder/map_builder.cc:832-900).
 *
 * The reference's insert loop runs under OpenMP with unsynchronised read-modify-writes of the voxel probabilities
 * (multi_resolution_voxel_map.cc:76-78, :94 "not atomic"), so a multi-threaded build of it has no defined result; what is
 * restated here is the loop as written, executed in point order -- the reference compiled without _OPENMP.
 *   Initialise            :46-53   hit / miss probabilities clamped to [0.501, 0.9] / [0.1, 0.499]
 *   constructor           :40-44   odds_table_[i] = log(p / (1 - p)), p = i / 256
 *   InsertPointCloud      :59-131  per point: Bresenham voxels origin -> point; end voxel: need_update = false, max intensity,
 *                                  prob = uint8(update(prob, hit) * 256), first max_point_num_in_cell points kept; every other
 *                                  voxel of the ray that EXISTS and still has need_update: prob = uint8(update(prob, miss) * 256);
 *                                  afterwards need_update = true for the cloud's end voxels
 *   VoxelCastingBresenham common/math.cc:35-93
 *   OutputToPointCloud    :133-170 (PointXYZI): voxels with probability >= uint8(threshold * 256)
 * PARITY UNPINNED: the reference holds no test or fixture for this class. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MRVM_TABLE 256
#define MRVM_UNKNOWN 128

typedef struct { float x, y, z, intensity, factor; } MrvmPoint;              /* data::InnerPointType, cloud_types.h:46-56 */
typedef struct {
  int32_t key[3];
  uint8_t used, prob, need_update;
  int32_t max_intensity, npoints;
  MrvmPoint* points;
} MrvmVoxel;
typedef struct {
  float high_resolution, hit_prob, miss_prob, z_offset;
  int32_t max_point_num_in_cell;
  float odds_table[MRVM_TABLE];
  MrvmVoxel* vox;
  size_t cap, count;
  int32_t* end_list; size_t end_n, end_cap;
} Mrvm;

static float clampf(float v, float lo, float hi) { return v > hi ? hi : (v < lo ? lo : v); }   /* common::Clamp, math.h:66-75 */
/* ProbabilityToOdd (header :133-135): float in, the arithmetic in double (the literal 1. is a double), float out */
static float prob_to_odd(float p) { return (float)log((double)p / (1. - (double)p)); }
/* OddToProbability (:137-139): std::exp(float) is the float overload; the rest is double; float out */
static float odd_to_prob(float odd) { return (float)(1. - 1. / (1. + (double)expf(odd))); }

static size_t hash3(int32_t x, int32_t y, int32_t z) {
  uint64_t h = (uint64_t)(uint32_t)x * 0x9E3779B97F4A7C15ull ^ (uint64_t)(uint32_t)y * 0xC2B2AE3D27D4EB4Full ^ (uint64_t)(uint32_t)z * 0x165667B19E3779F9ull;
  h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
  return (size_t)h;
}
static void grow(Mrvm* m);
static MrvmVoxel* find(Mrvm* m, const int32_t k[3], int create) {
  if (create && (m->count + 1) * 10 > m->cap * 7) grow(m);
  size_t i = hash3(k[0], k[1], k[2]) & (m->cap - 1);
  for (;;) {
    MrvmVoxel* v = &m->vox[i];
    if (!v->used) {
      if (!create) return NULL;
      v->used = 1; v->key[0] = k[0]; v->key[1] = k[1]; v->key[2] = k[2];
      v->prob = MRVM_UNKNOWN; v->need_update = 1; v->max_intensity = 0; v->npoints = 0;          /* HighResolutionVoxel(), header :118-120 */
      v->points = (MrvmPoint*)malloc(sizeof(MrvmPoint) * (size_t)m->max_point_num_in_cell);
      ++m->count;
      return v;
    }
    if (v->key[0] == k[0] && v->key[1] == k[1] && v->key[2] == k[2]) return v;
    i = (i + 1) & (m->cap - 1);
  }
}
static void grow(Mrvm* m) {
  MrvmVoxel* old = m->vox; const size_t oc = m->cap;
  m->cap *= 2; m->vox = (MrvmVoxel*)calloc(m->cap, sizeof(MrvmVoxel));
  for (size_t i = 0; i < oc; ++i) if (old[i].used) {
    size_t j = hash3(old[i].key[0], old[i].key[1], old[i].key[2]) & (m->cap - 1);
    while (m->vox[j].used) j = (j + 1) & (m->cap - 1);
    m->vox[j] = old[i];
  }
  free(old);
}

void* smref_mrvm_create(float high_resolution, float hit_prob, float miss_prob, float z_offset, int max_point_num_in_cell) {
  if (max_point_num_in_cell <= 0 || !(high_resolution > 0.f)) return NULL;                         /* CHECK_GT(max_point_num_in_cell, 0), :48 */
  Mrvm* m = (Mrvm*)calloc(1, sizeof(Mrvm));
  m->high_resolution = high_resolution; m->z_offset = z_offset; m->max_point_num_in_cell = max_point_num_in_cell;
  m->hit_prob = clampf(hit_prob, 0.501f, 0.9f);                                                    /* :50-52 */
  m->miss_prob = clampf(miss_prob, 0.1f, 0.499f);
  for (int i = 0; i < MRVM_TABLE; ++i) m->odds_table[i] = prob_to_odd((float)i / (float)MRVM_TABLE);   /* :41-43 */
  m->cap = 1 << 16; m->vox = (MrvmVoxel*)calloc(m->cap, sizeof(MrvmVoxel));
  return m;
}
void smref_mrvm_free(void* h) {
  Mrvm* m = (Mrvm*)h;
  if (!m) return;
  for (size_t i = 0; i < m->cap; ++i) if (m->vox[i].used) free(m->vox[i].points);
  free(m->vox); free(m->end_list); free(m);
}
/* the byte -> byte maps of one hit / one miss: prob = (Probability)(update_prob(prob, hit) * kTableSize), :68-72, :99, :114 */
void smref_mrvm_tables(void* h, uint8_t* hit_table, uint8_t* miss_table) {
  Mrvm* m = (Mrvm*)h;
  const float hit_log_odd = prob_to_odd(m->hit_prob), miss_log_odd = prob_to_odd(m->miss_prob);
  for (int p = 0; p < MRVM_TABLE; ++p) {
    float odd = m->odds_table[p]; odd += hit_log_odd;
    hit_table[p] = (uint8_t)(clampf(odd_to_prob(odd), 0.1f, 0.9f) * (float)MRVM_TABLE);
    odd = m->odds_table[p]; odd += miss_log_odd;
    miss_table[p] = (uint8_t)(clampf(odd_to_prob(odd), 0.1f, 0.9f) * (float)MRVM_TABLE);
  }
}

static int32_t voxel_of(float c, float step) { return (int32_t)lroundf(floorf(c / step)); }       /* common/math.cc:41-43 */

/* points: n rows of 5 floats (InnerPointType); origin: the sensor position (frame->GlobalTranslation()) */
int smref_mrvm_insert(void* h, const float* points, int n, const float origin[3]) {
  Mrvm* m = (Mrvm*)h;
  if (!m || !points || n <= 0) return 1;                                                           /* "cloud is empty.", :61-64 */
  const float o[3] = {origin[0], origin[1], origin[2] + m->z_offset};                              /* :66-67 */
  const float hit_log_odd = prob_to_odd(m->hit_prob), miss_log_odd = prob_to_odd(m->miss_prob);
  const float res = m->high_resolution;
  m->end_n = 0;
  for (int pi = 0; pi < n; ++pi) {
    const MrvmPoint pt = *(const MrvmPoint*)(points + 5 * (size_t)pi);
    if (!isfinite(pt.x) || !isfinite(pt.y) || !isfinite(pt.z)) continue;                          /* FATAL_CHECK_POINT territory: skipped */
    /* VoxelCastingBresenham(offseted_origin, point_vec, resolution), common/math.cc:35-93 */
    int32_t x0 = voxel_of(o[0], res), y0 = voxel_of(o[1], res), z0 = voxel_of(o[2], res);
    const int32_t xe = voxel_of(pt.x, res), ye = voxel_of(pt.y, res), ze = voxel_of(pt.z, res);
    const int dx = abs(xe - x0), sx = x0 < xe ? 1 : -1;
    const int dy = abs(ye - y0), sy = y0 < ye ? 1 : -1;
    const int dz = abs(ze - z0), sz = z0 < ze ? 1 : -1;
    int dm = dx > dy ? dx : dy; if (dz > dm) dm = dz;
    int ex = dm >> 1, ey = dm >> 1, ez = dm >> 1;
    /* the end voxel first (:86-104), exactly as the loop body orders it */
    {
      const int32_t ke[3] = {xe, ye, ze};
      MrvmVoxel* v = find(m, ke, 1);
      if (v->need_update) {                                  /* first end-voxel visit of this cloud: remember it for the reset at :128-130 */
        if (m->end_n == m->end_cap) { m->end_cap = m->end_cap ? 2 * m->end_cap : 4096; m->end_list = (int32_t*)realloc(m->end_list, sizeof(int32_t) * 3 * m->end_cap); }
        memcpy(&m->end_list[3 * m->end_n++], ke, sizeof(ke));
      }
      v->need_update = 0;                                                                          /* :88 */
      if ((int)pt.intensity > v->max_intensity) v->max_intensity = (int)pt.intensity;              /* :95-98 */
      float odd = m->odds_table[v->prob]; odd += hit_log_odd;
      v->prob = (uint8_t)(clampf(odd_to_prob(odd), 0.1f, 0.9f) * (float)MRVM_TABLE);               /* :99 */
      if (v->npoints < m->max_point_num_in_cell) v->points[v->npoints++] = pt;                     /* :100-103 */
    }
    /* the voxels on the line, all but the last (:106-117) */
    for (int i = dm; i > 0; --i) {
      const int32_t k[3] = {x0, y0, z0};
      MrvmVoxel* v = find(m, k, 0);
      if (v && v->need_update) {
        float odd = m->odds_table[v->prob]; odd += miss_log_odd;
        v->prob = (uint8_t)(clampf(odd_to_prob(odd), 0.1f, 0.9f) * (float)MRVM_TABLE);
      }
      ex -= dx; if (ex < 0) { ex += dm; x0 += sx; }
      ey -= dy; if (ey < 0) { ey += dm; y0 += sy; }
      ez -= dz; if (ez < 0) { ez += dm; z0 += sz; }
    }
  }
  for (size_t e = 0; e < m->end_n; ++e) find(m, &m->end_list[3 * e], 0)->need_update = 1;         /* :128-130 */
  return 0;
}

long smref_mrvm_voxel_count(void* h) { return (long)((Mrvm*)h)->count; }
/* every voxel of the map: keys [count][3], probability, max intensity, stored points; points5 [count][max_point_num_in_cell][5] */
long smref_mrvm_dump(void* h, int32_t* keys, uint8_t* prob, int32_t* max_intensity, int32_t* npoints, float* points5) {
  Mrvm* m = (Mrvm*)h;
  long c = 0;
  for (size_t i = 0; i < m->cap; ++i) {
    const MrvmVoxel* v = &m->vox[i];
    if (!v->used) continue;
    if (keys) memcpy(&keys[3 * c], v->key, sizeof(v->key));
    if (prob) prob[c] = v->prob;
    if (max_intensity) max_intensity[c] = v->max_intensity;
    if (npoints) npoints[c] = v->npoints;
    if (points5) memcpy(&points5[(size_t)c * 5 * (size_t)m->max_point_num_in_cell], v->points, sizeof(MrvmPoint) * (size_t)v->npoints);
    ++c;
  }
  return c;
}
/* OutputToPointCloud, both overloads with settings_.output_average (:125-216).  flags bit 0 = output_average (one row per voxel:
 * float sums of the stored points in their order / float(size)), bit 1 = the PointXYZRGB overload (4th column = the grey level
 * min(255, uint32(max_intensity * 1.4)) as a float); else 4th column = the voxel's max intensity when use_max_intensity, the
 * point's own otherwise (0 for an averaged point: the reference leaves it unassigned). */
long smref_mrvm_output_ex(void* h, float threshold, int use_max_intensity, int flags, float* xyzi, long capacity) {
  Mrvm* m = (Mrvm*)h;
  const uint8_t thr = (uint8_t)(threshold * (float)MRVM_TABLE);
  long c = 0;
  for (size_t i = 0; i < m->cap; ++i) {
    const MrvmVoxel* v = &m->vox[i];
    if (!v->used || v->prob < thr) continue;
    uint32_t intensity = (uint32_t)v->max_intensity;
    intensity = (uint32_t)((double)intensity * 1.4);
    if (intensity > 255) intensity = 255;
    if (flags & 1) {
      if (v->npoints <= 0) continue;
      volatile float ax = 0.f, ay = 0.f, az = 0.f;     /* volatile: every add rounded to float, as the reference's float members are */
      for (int k = 0; k < v->npoints; ++k) { ax = ax + v->points[k].x; ay = ay + v->points[k].y; az = az + v->points[k].z; }
      const float size = (float)v->npoints;
      if (xyzi && c < capacity) {
        xyzi[4 * c] = ax / size; xyzi[4 * c + 1] = ay / size; xyzi[4 * c + 2] = az / size;
        xyzi[4 * c + 3] = (flags & 2) ? (float)intensity : (use_max_intensity ? (float)v->max_intensity : 0.f);
      }
      ++c;
      continue;
    }
    for (int k = 0; k < v->npoints; ++k) {
      if (xyzi && c < capacity) {
        xyzi[4 * c] = v->points[k].x; xyzi[4 * c + 1] = v->points[k].y; xyzi[4 * c + 2] = v->points[k].z;
        xyzi[4 * c + 3] = (flags & 2) ? (float)intensity : (use_max_intensity ? (float)v->max_intensity : v->points[k].intensity);
      }
      ++c;
    }
  }
  return c;
}
long smref_mrvm_output(void* h, float threshold, int use_max_intensity, float* xyzi, long capacity) {
  return smref_mrvm_output_ex(h, threshold, use_max_intensity, 0, xyzi, capacity);
}
