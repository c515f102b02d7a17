// Host-side dry run of the sharded sequence driver at R ranks on ONE host, without a GPU: what smhip_shard's ranks do on the
// CPU before a scan reaches the device -- ShardReadOrder, a ScanPrefetcher pool per rank (kitti_reader.cc:91-149 semantics),
// and the copy of every scan into a staging buffer (smhip_set_source_f32's memcpy into pinned memory) -- for R = 1 and R = 8
// concurrent ranks against a generated drive.  Reports aggregate scans/s, so the host-side ceiling of an 8-GPU node is known
// before one is available; checks that every rank saw exactly its files with the right contents.
// usage: host_dry_run <scratch dir> [scans] [points per scan] [readers per rank] [laps]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "smhip/kitti_scans.h"

namespace {
std::string ScanPath(const std::string& dir, int k) {
  char name[64];
  std::snprintf(name, sizeof(name), "/%06d.bin", k);
  return dir + name;
}
// row i of scan k: (k, i, k + i, 0.5) -- cheap to verify
void WriteScan(const std::string& path, int k, int points) {
  std::vector<float> v(static_cast<size_t>(points) * 4);
  for (int i = 0; i < points; ++i) { v[4 * i] = (float)k; v[4 * i + 1] = (float)(i % 4096); v[4 * i + 2] = (float)(k + i % 4096); v[4 * i + 3] = 0.5f; }
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) { std::perror("fopen"); std::exit(2); }
  std::fwrite(v.data(), sizeof(float), v.size(), f);
  std::fclose(f);
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: host_dry_run <scratch dir> [scans] [points] [readers]\n"); return 2; }
  const std::string dir = argv[1];
  const int n_scans = argc > 2 ? std::atoi(argv[2]) : 65;
  const int points = argc > 3 ? std::atoi(argv[3]) : 120000;
  const int readers = argc > 4 ? std::atoi(argv[4]) : 4;
  const int laps = argc > 5 ? std::atoi(argv[5]) : 6;
  std::vector<std::string> files;
  for (int k = 0; k < n_scans; ++k) { files.push_back(ScanPath(dir, k)); WriteScan(files.back(), k, points); }
  const int n_pairs = n_scans - 1, batch = 64;
  int failures = 0;
  std::printf("{\"scans\": %d, \"points_per_scan\": %d, \"readers_per_rank\": %d, \"laps\": %d, \"runs\": [", n_scans, points, readers, laps);
  bool first = true;
  for (int world : {1, 2, 8}) {
    std::atomic<long> scans_read{0}, bad{0};
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> ranks;
    for (int rank = 0; rank < world; ++rank) {
      ranks.emplace_back([&, rank] {
        const int per = (n_pairs + world - 1) / world;
        const std::vector<int> lap = smhip::kitti::ShardReadOrder(n_pairs, world, rank, std::max(1, std::min(batch, per)));
        std::vector<int> order;                               // the drive several times over: a run long enough to time
        for (int l = 0; l < laps; ++l) order.insert(order.end(), lap.begin(), lap.end());
        smhip::kitti::ScanPrefetcher pf(files, order, readers, 4 * readers + 8);
        std::vector<float> stage(smhip::kitti::kMaxFloatsPerFile);
        int n = 0, fi = -1;
        size_t at = 0;
        while (const float* rows = pf.Next(&n, &fi)) {
          if (at >= order.size() || fi != order[at] || n != points) { ++bad; ++at; continue; }
          std::memcpy(stage.data(), rows, sizeof(float) * 4 * static_cast<size_t>(n));          // the upload's staging copy
          const int i = (fi * 7919) % n;
          if (stage[4 * i] != (float)fi || stage[4 * i + 1] != (float)(i % 4096) || stage[4 * i + 2] != (float)(fi + i % 4096)) ++bad;
          ++at; ++scans_read;
        }
        if (at != order.size()) ++bad;
      });
    }
    for (auto& t : ranks) t.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // every pair needs its two scans: one rank re-reads only batch boundaries, R ranks read both scans of every pair
    std::printf("%s{\"ranks\": %d, \"scans_read\": %ld, \"seconds\": %.4f, \"scans_per_s\": %.1f, \"pairs_per_s\": %.1f, \"GB_per_s\": %.3f, \"bad\": %ld}",
                first ? "" : ", ", world, scans_read.load(), s, scans_read / s, (double)n_pairs * laps / s, scans_read * 16.0 * points / s / 1e9, bad.load());
    first = false;
    failures += bad.load() != 0;
  }
  std::printf("], \"failed\": %d}\n", failures);
  for (const auto& f : files) std::remove(f.c_str());
  return failures ? 1 : 0;
}
