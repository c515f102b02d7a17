// Host-only test of include/smhip/kitti_scans.h: the `.bin` reader (kitti_reader.cc:91-121 semantics: float32 rows of 4,
// at most 1 000 000 floats per file) and the read-ahead pool behind the sharded sequence driver.
// usage: test_kitti_scans <scratch dir>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "smhip/kitti_scans.h"

namespace {

int failures = 0;
#define EXPECT(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

// file k holds `floats` floats, value at position i = 1000 k + i
std::string WriteScan(const std::string& dir, int k, size_t floats) {
  char name[64];
  std::snprintf(name, sizeof(name), "/%06d.bin", k);
  const std::string path = dir + name;
  std::vector<float> v(floats);
  for (size_t i = 0; i < floats; ++i) v[i] = static_cast<float>(1000.0 * k + static_cast<double>(i % 997));
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) { std::perror("fopen"); std::exit(2); }
  std::fwrite(v.data(), sizeof(float), floats, f);
  std::fclose(f);
  return path;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: test_kitti_scans <scratch dir>\n"); return 2; }
  const std::string dir = argv[1];
  using smhip::kitti::kMaxFloatsPerFile;
  // 0: ordinary, 1: not a multiple of 4 floats (the trailing partial row is dropped), 2: empty, 3: longer than the reader's
  // buffer (truncated at 1 000 000 floats = 250 000 points), 4..11: ordinary of different lengths
  std::vector<std::string> files;
  std::vector<size_t> floats = {4 * 1000, 4 * 500 + 3, 0, kMaxFloatsPerFile + 4 * 100};
  for (int k = 4; k < 12; ++k) floats.push_back(4 * static_cast<size_t>(100 * k + 7));
  for (size_t k = 0; k < floats.size(); ++k) files.push_back(WriteScan(dir, static_cast<int>(k), floats[k]));
  files.push_back(dir + "/does_not_exist.bin");                           // index 12: unreadable

  {   // the reader alone
    std::vector<float> rows(kMaxFloatsPerFile);
    EXPECT(smhip::kitti::ReadBin(files[0], rows.data()) == 1000);
    EXPECT(rows[0] == 0.f && rows[5] == 5.f);
    EXPECT(smhip::kitti::ReadBin(files[1], rows.data()) == 500);
    EXPECT(smhip::kitti::ReadBin(files[2], rows.data()) == 0);
    EXPECT(smhip::kitti::ReadBin(files[3], rows.data()) == static_cast<int>(kMaxFloatsPerFile / 4));
    EXPECT(smhip::kitti::ReadBin(files[12], rows.data()) == -1);
  }

  // the order a two-rank, batch-3 driver would produce, with repeats, the odd files and the unreadable one
  std::vector<int> order = {0, 1, 2, 3, 3, 4, 5, 6, 12, 7, 8, 9, 10, 11, 11, 0, 2, 5};
  for (int rep = 0; rep < 6; ++rep) for (int k = 4; k < 12; ++k) order.push_back(k);
  for (int threads : {1, 2, 4, 7}) {
    for (int ring : {2, 3, 16}) {
      smhip::kitti::ScanPrefetcher pf(files, order, threads, ring);
      EXPECT(pf.planned() == order.size());
      for (size_t i = 0; i < order.size(); ++i) {
        int n = -2, fi = -2;
        const float* rows = pf.Next(&n, &fi);
        EXPECT(rows != nullptr);
        if (!rows) break;
        EXPECT(fi == order[i]);
        const int k = order[i];
        if (k == 12) { EXPECT(n == -1); continue; }
        const size_t want = std::min(floats[k], kMaxFloatsPerFile) / 4;
        EXPECT(n == static_cast<int>(want));
        if (n > 0) {
          EXPECT(rows[0] == static_cast<float>(1000.0 * k));
          const size_t last = 4 * static_cast<size_t>(n) - 1;
          EXPECT(rows[last] == static_cast<float>(1000.0 * k + static_cast<double>(last % 997)));
        }
      }
      int n = 0, fi = 0;
      EXPECT(pf.Next(&n, &fi) == nullptr);                                // exhausted
      EXPECT(pf.Next(&n, &fi) == nullptr);
    }
  }
  {   // destroyed with scans still unread: the workers stop
    smhip::kitti::ScanPrefetcher pf(files, order, 3, 4);
    int n = 0, fi = 0;
    EXPECT(pf.Next(&n, &fi) != nullptr);
  }
  {   // hold-until-release over caller-owned buffers (the driver's page-locked ring): every scan handed out since the last
      // ReleaseHeld() keeps its contents while the readers run ahead in the remaining buffers; short buffers truncate the file
    const size_t slot_floats = 4 * 600;                                   // room for 600 points
    std::vector<std::vector<float>> store(7, std::vector<float>(slot_floats));
    std::vector<float*> bufs;
    for (auto& v : store) bufs.push_back(v.data());
    std::vector<int> ord = {4, 5, 6, 7, 8, 9, 10, 11, 0, 4, 5, 6};         // files 4.. hold 407 .. 1107 points, file 0 1000
    smhip::kitti::ScanPrefetcher pf(files, ord, 3, bufs, slot_floats, /*hold_until_release=*/true);
    size_t at = 0;
    while (at < ord.size()) {
      std::vector<const float*> held;
      std::vector<int> held_k;
      for (int b = 0; b < 3 && at < ord.size(); ++b, ++at) {              // a "batch" of three scans, all held at once
        int n = -2, fi = -2;
        const float* rows = pf.Next(&n, &fi);
        EXPECT(rows != nullptr && fi == ord[at]);
        EXPECT(n == static_cast<int>(std::min(floats[ord[at]], slot_floats) / 4));
        held.push_back(rows); held_k.push_back(ord[at]);
      }
      for (size_t b = 0; b < held.size(); ++b) EXPECT(held[b][0] == static_cast<float>(1000.0 * held_k[b]) && held[b][5] == static_cast<float>(1000.0 * held_k[b] + 5));
      pf.ReleaseHeld();
    }
    int n = 0, fi = 0;
    EXPECT(pf.Next(&n, &fi) == nullptr);
  }
  {   // nothing planned
    smhip::kitti::ScanPrefetcher pf(files, {}, 2, 2);
    int n = 0, fi = 0;
    EXPECT(pf.Next(&n, &fi) == nullptr);
  }
  if (failures) { std::fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
  std::printf("kitti_scans: all checks passed\n");
  return 0;
}
