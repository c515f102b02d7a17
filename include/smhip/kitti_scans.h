// kitti_scans.h -- KITTI `.bin` scans (float32 rows x y z reflectance) read ahead by a pool of threads.
//
// Reference: ros_node/kitti_reader.cc:91-149 -- at most 1 000 000 floats per file (:93), files in sorted directory order
// (:124-131).  Host-only C++ (no HIP): used by the sharded sequence driver (csrc/shard_driver.cc) and testable on a CPU box
// (tests/cpp/test_kitti_scans.cc).
#pragma once

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace smhip {
namespace kitti {

constexpr size_t kMaxFloatsPerFile = 1000000;    // kitti_reader.cc:93

// number of points read into rows (kMaxFloatsPerFile floats of room); -1 if the file cannot be opened
inline int ReadBin(const std::string& path, float* rows, size_t room_floats = kMaxFloatsPerFile) {   // kitti_reader.cc:91-121
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return -1;
  const size_t got = std::fread(rows, sizeof(float), std::min(room_floats, kMaxFloatsPerFile), f);
  std::fclose(f);
  return static_cast<int>(got / 4);
}

// The files rank `rank` of `world` reads, in reading order, when the pairs (scan i, scan i + 1) go round-robin over the ranks
// (pair i -> rank i mod world) in batches of `batch` pairs per rank: a pair's target scan is read only when the previous slot
// of the batch did not already hold it as its source (one rank: every pair but the first of a batch).
inline std::vector<int> ShardReadOrder(int n_pairs, int world, int rank, int batch) {
  std::vector<int> order;
  const int per = (n_pairs + world - 1) / world;
  for (int base = 0; base < per; base += batch) {
    int prev_pair = -2;
    for (int k = 0; k < batch && base + k < per; ++k) {
      const int pair = (base + k) * world + rank;
      if (pair >= n_pairs) break;
      if (pair != prev_pair + 1) order.push_back(pair);
      order.push_back(pair + 1);
      prev_pair = pair;
    }
  }
  return order;
}

// The scans a consumer will ask for, in the order it will ask for them, read ahead by a few threads into a ring of buffers.
// (The alignment of a batch takes the GPU a few milliseconds; reading and staging its 64 scans took one host thread 35 ms.)
class ScanPrefetcher {
 public:
  // hold_until_release: Next() no longer hands the previous scan's buffer back to the readers; every scan handed out stays
  // valid until ReleaseHeld() (a batch of scans being copied to the device from these very buffers).  The ring must then
  // hold more scans than are ever outstanding at once.  slot_floats: room per buffer (<= kMaxFloatsPerFile; the reader
  // truncates a longer file there, as it does at kMaxFloatsPerFile).  start = false: the readers start with Start(), after the
  // caller has had a chance to page-lock the buffers (Buffer(k), k < Ring()).
  ScanPrefetcher(const std::vector<std::string>& files, std::vector<int> order, int threads, int ring, bool hold_until_release = false,
                 size_t slot_floats = kMaxFloatsPerFile, bool start = true)
      : files_(files), order_(std::move(order)), ring_(std::max(2, ring)), hold_(hold_until_release),
        slot_floats_(std::min(kMaxFloatsPerFile, std::max<size_t>(4, slot_floats))), threads_(std::max(1, threads)), slots_(ring_) {
    for (auto& sl : slots_) { sl.rows.resize(slot_floats_); sl.data = sl.rows.data(); }
    if (start) Start();
  }
  // The same over buffers the caller owns (e.g. page-locked memory the device copies from directly): buffers.size() slots of
  // slot_floats floats each, which must outlive the prefetcher.
  ScanPrefetcher(const std::vector<std::string>& files, std::vector<int> order, int threads, const std::vector<float*>& buffers, size_t slot_floats,
                 bool hold_until_release)
      : files_(files), order_(std::move(order)), ring_(static_cast<int>(buffers.size())), hold_(hold_until_release),
        slot_floats_(std::min(kMaxFloatsPerFile, std::max<size_t>(4, slot_floats))), threads_(std::max(1, threads)), slots_(buffers.size()) {
    for (size_t k = 0; k < buffers.size(); ++k) slots_[k].data = buffers[k];
    Start();
  }
  void Start() {
    if (!workers_.empty()) return;
    for (int t = 0; t < threads_; ++t) workers_.emplace_back([this] { Work(); });
  }
  int Ring() const { return ring_; }
  float* Buffer(int k) { return slots_[k].data; }
  size_t BufferBytes() const { return slot_floats_ * sizeof(float); }
  // hands every buffer given out since the last call back to the readers
  void ReleaseHeld() {
    std::lock_guard<std::mutex> lk(m_);
    for (long i = released_; i < consumed_; ++i) slots_[i % ring_].state = 0;
    released_ = consumed_;
    held_ = -1;
    cv_free_.notify_all();
  }
  ~ScanPrefetcher() {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
    cv_free_.notify_all();
    for (auto& w : workers_) w.join();
  }
  ScanPrefetcher(const ScanPrefetcher&) = delete;
  ScanPrefetcher& operator=(const ScanPrefetcher&) = delete;
  size_t planned() const { return order_.size(); }
  // The next scan of the order: its rows stay valid until the following Next().  *n = points (-1: unreadable file),
  // *file_index = index into `files`.  nullptr once the order is exhausted.
  const float* Next(int* n, int* file_index) {
    std::unique_lock<std::mutex> lk(m_);
    if (!hold_ && held_ >= 0) { slots_[held_ % ring_].state = 0; held_ = -1; released_ = consumed_; cv_free_.notify_all(); }
    if (consumed_ >= static_cast<long>(order_.size())) return nullptr;
    const long i = consumed_++;
    Slot& sl = slots_[i % ring_];
    cv_ready_.wait(lk, [&] { return sl.state == 2 && sl.item == i; });
    held_ = i;
    *n = sl.n; *file_index = order_[i];
    return sl.data;
  }

 private:
  struct Slot { std::vector<float> rows; float* data = nullptr; int n = 0; long item = -1; int state = 0; };   // 0 free, 1 being read, 2 ready
  void Work() {
    for (;;) {
      long i;
      Slot* sl;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_free_.wait(lk, [&] { return stop_ || next_ >= static_cast<long>(order_.size()) || slots_[next_ % ring_].state == 0; });
        if (stop_ || next_ >= static_cast<long>(order_.size())) return;
        i = next_++;
        sl = &slots_[i % ring_];
        sl->state = 1; sl->item = i;
      }
      const int n = ReadBin(files_[order_[i]], sl->data, slot_floats_);
      { std::lock_guard<std::mutex> lk(m_); sl->n = n; sl->state = 2; }
      cv_ready_.notify_all();
      cv_free_.notify_all();
    }
  }
  const std::vector<std::string>& files_;
  const std::vector<int> order_;
  const int ring_;
  const bool hold_;
  const size_t slot_floats_;
  const int threads_;
  std::vector<Slot> slots_;
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_ready_, cv_free_;
  long next_ = 0, consumed_ = 0, held_ = -1, released_ = 0;
  bool stop_ = false;
};

}  // namespace kitti
}  // namespace smhip
