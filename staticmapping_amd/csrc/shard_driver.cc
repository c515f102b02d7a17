// shard_driver.cc -- BASELINE config #4 as a C++ program: scan-to-scan ICP over a KITTI-format sequence, consecutive pairs
// dealt round-robin over the GPUs of one node, ONE RCCL all-gather of the resulting SE(3) poses.
//
// Reference pieces either side of the registration path that this driver stands in for:
//   ros_node/kitti_reader.cc:91-149   `.bin` scans: float32 rows x y z reflectance, at most 1 000 000 floats per file,
//                                      files in sorted directory order
//   builder/map_builder.cc:354        pose_source = pose_target * align_result
//   builder/map_builder.cc:626-641    kitti_pose.txt: 12 floats per line (row-major top 3x4), setprecision(8)
// The reference itself has no multi-GPU path (its front end is one sequential thread); pairs are independent here because
// every pair's guess is fixed up front (identity or a constant forward step), as SURVEY.md §8(e) lays out.
//
// One process per GPU.  `smhip_shard --gpus G ...` re-executes itself G times (rank r on device r); the ranks can also be
// started by any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK (torchrun's variables).  The ncclUniqueId travels from
// rank 0 to the others through a small file (--id-file, default under /tmp): no MPI, no sockets of our own.
// All arithmetic happens behind the C ABI of include/smhip.h; the gather is ncclAllGather on the doubles
// smhip_icp_export_results_device leaves in device memory -- the poses never visit the host before the collective.
#include <dirent.h>
#include <fcntl.h>
#include <signal.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/smhip.h"
#include "../../include/smhip/kitti_scans.h"

namespace {

constexpr int kPoseDoubles = 18;                 // 16 column-major transform + score + iterations
using smhip::kitti::kMaxFloatsPerFile;
using smhip::kitti::ScanPrefetcher;

struct Args {
  std::string scans_dir, out_path = "kitti_pose.txt", id_file;
  unsigned long long nonce = 0;             // identifies this run's id file (launcher: pid and start time; else MASTER_PORT)
  int gpus = 1, rank = -1, world = -1, local_rank = -1;
  int batch = 256, iterations = 20, early_exit = 0, max_pairs = -1, readers = 8, matchers = 1, warmup = 1, parts = 0;
  double guess_tx = 0.0;
  bool quiet = false;
};

[[noreturn]] void Die(const std::string& m) { std::fprintf(stderr, "smhip_shard: %s\n", m.c_str()); std::exit(2); }

#define HIPOK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) Die(std::string(#e) + ": " + hipGetErrorString(e_)); } while (0)
#define NCCLOK(e) do { ncclResult_t r_ = (e); if (r_ != ncclSuccess) Die(std::string(#e) + ": " + ncclGetErrorString(r_)); } while (0)

std::vector<std::string> ListScans(const std::string& dir) {            // kitti_reader.cc:124-131: sorted listing
  std::vector<std::string> files;
  DIR* d = opendir(dir.c_str());
  if (!d) Die("cannot open " + dir);
  while (dirent* e = readdir(d)) {
    const std::string n = e->d_name;
    if (n.size() > 4 && n.compare(n.size() - 4, 4, ".bin") == 0) files.push_back(dir + "/" + n);
  }
  closedir(d);
  std::sort(files.begin(), files.end());
  return files;
}

void Mul4(const double* a, const double* b, double* out) {              // row-major 4x4
  double r[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += a[4 * i + k] * b[4 * k + j]; r[4 * i + j] = s; }
  std::memcpy(out, r, sizeof(r));
}

Args Parse(int argc, char** argv) {
  Args a;
  for (int i = 1; i < argc; ++i) {
    const std::string k = argv[i];
    auto val = [&]() -> std::string { if (i + 1 >= argc) Die("missing value for " + k); return argv[++i]; };
    if (k == "--scans") a.scans_dir = val();
    else if (k == "--out") a.out_path = val();
    else if (k == "--gpus") a.gpus = std::atoi(val().c_str());
    else if (k == "--rank") a.rank = std::atoi(val().c_str());
    else if (k == "--world") a.world = std::atoi(val().c_str());
    else if (k == "--local-rank") a.local_rank = std::atoi(val().c_str());
    else if (k == "--id-file") a.id_file = val();
    else if (k == "--nonce") a.nonce = std::strtoull(val().c_str(), nullptr, 10);
    else if (k == "--batch") a.batch = std::atoi(val().c_str());
    else if (k == "--iterations") a.iterations = std::atoi(val().c_str());
    else if (k == "--early-exit") a.early_exit = std::atoi(val().c_str());
    else if (k == "--max-pairs") a.max_pairs = std::atoi(val().c_str());
    else if (k == "--readers") a.readers = std::atoi(val().c_str());
    else if (k == "--matchers") a.matchers = std::atoi(val().c_str());
    else if (k == "--warmup") a.warmup = std::atoi(val().c_str());
    else if (k == "--parts") a.parts = std::atoi(val().c_str());
    else if (k == "--guess-tx") a.guess_tx = std::atof(val().c_str());
    else if (k == "--quiet") a.quiet = true;
    else Die("unknown argument " + k + "\nusage: smhip_shard --scans DIR [--gpus G] [--out kitti_pose.txt] [--batch 256] "
             "[--iterations 20] [--early-exit 0|1] [--guess-tx metres] [--max-pairs N] [--readers 8] [--matchers 1|2] [--warmup 1|0] [--parts 0..4]");
  }
  if (a.scans_dir.empty()) Die("--scans DIR is required");
  if (a.rank < 0 && std::getenv("RANK")) a.rank = std::atoi(std::getenv("RANK"));
  if (a.world < 0 && std::getenv("WORLD_SIZE")) a.world = std::atoi(std::getenv("WORLD_SIZE"));
  if (a.local_rank < 0 && std::getenv("LOCAL_RANK")) a.local_rank = std::atoi(std::getenv("LOCAL_RANK"));
  return a;
}

// rank 0 creates the communicator id and publishes it; the others wait for the file.  The file carries a per-run nonce
// (the launcher's, or MASTER_PORT / SMHIP_SHARD_NONCE under an external launcher) in front of the id: a file left behind by
// a crashed run -- or by another run using the same path -- does not match and is waited out instead of being taken for
// this run's id (ncclCommInitRank would hang on a stale one).  Rank 0 replaces any existing file (unlink + O_EXCL on the
// temporary, then rename) and removes it at exit, also on the error paths.
struct IdRecord { unsigned long long nonce; ncclUniqueId id; };
std::string g_id_file_to_remove;
void RemoveIdFile() { if (!g_id_file_to_remove.empty()) std::remove(g_id_file_to_remove.c_str()); }

ncclUniqueId ExchangeId(const Args& a, int rank) {
  IdRecord rec{};
  rec.nonce = a.nonce;
  if (rank == 0) {
    NCCLOK(ncclGetUniqueId(&rec.id));
    const std::string tmp = a.id_file + ".tmp." + std::to_string(static_cast<long>(getpid()));
    std::remove(a.id_file.c_str());
    std::remove(tmp.c_str());
    const int fd = open(tmp.c_str(), O_CREAT | O_EXCL | O_WRONLY, 0600);
    if (fd < 0) Die("cannot create " + tmp);
    const bool ok = write(fd, &rec, sizeof(rec)) == static_cast<ssize_t>(sizeof(rec));
    close(fd);
    if (!ok) { std::remove(tmp.c_str()); Die("cannot write " + tmp); }
    g_id_file_to_remove = a.id_file;
    std::atexit(RemoveIdFile);
    if (std::rename(tmp.c_str(), a.id_file.c_str()) != 0) { std::remove(tmp.c_str()); Die("cannot publish " + a.id_file); }
    return rec.id;
  }
  for (int tries = 0; tries < 6000; ++tries) {                           // <= 60 s
    IdRecord got{};
    std::ifstream f(a.id_file, std::ios::binary);
    if (f && f.read(reinterpret_cast<char*>(&got), sizeof(got)) && f.gcount() == static_cast<std::streamsize>(sizeof(got)) &&
        got.nonce == a.nonce)
      return got.id;
    usleep(10000);
  }
  Die("timed out waiting for " + a.id_file + " (no file with this run's nonce appeared)");
}

int RunRank(const Args& a, int rank, int world, int device) {
  const auto files = ListScans(a.scans_dir);
  if (files.size() < 2) Die("need at least two scans in " + a.scans_dir);
  int n_pairs = static_cast<int>(files.size()) - 1;
  if (a.max_pairs > 0) n_pairs = std::min(n_pairs, a.max_pairs);
  const int per = (n_pairs + world - 1) / world;                         // padded pairs per rank

  HIPOK(hipSetDevice(device));
  ncclComm_t comm;
  const ncclUniqueId id = ExchangeId(a, rank);
  NCCLOK(ncclCommInitRank(&comm, world, id, rank));

  const int B = std::max(1, std::min(std::min(a.batch, 256), per));     // one batched upload holds up to 2 B <= 512 scans
  // --matchers 2: two matchers, each with its own stream, take the batches in turn -- while one runs the alignments of batch k
  // (enqueued, not waited for), the host reads, uploads and prepares the targets of batch k + 1 on the other.
  const int NH = (a.matchers >= 2 && per > B) ? 2 : 1;
  hipStream_t streams[2] = {nullptr, nullptr};
  smhip_handle hs[2] = {nullptr, nullptr};
  for (int k = 0; k < NH; ++k) HIPOK(hipStreamCreateWithFlags(&streams[k], hipStreamNonBlocking));
  hipStream_t stream = streams[0];
  // capacity: the largest scan of the directory (a KITTI scan holds at most 250 000 points: 1 000 000 floats per file are
  // read, kitti_reader.cc:93).  Slots [0, B) hold the pairs of a batch, slots [B, 2 B) park target scans that no pair of the
  // batch holds as its source already.
  size_t max_bytes = 16;
  for (const auto& f : files) { struct stat sb; if (stat(f.c_str(), &sb) == 0) max_bytes = std::max(max_bytes, static_cast<size_t>(sb.st_size)); }
  const size_t slot_floats = std::min(kMaxFloatsPerFile, (max_bytes / 16 + 1) * 4);
  const int cap = static_cast<int>(slot_floats / 4);
  for (int k = 0; k < NH; ++k) {
    const smhip_status s = smhip_create(device, streams[k], 2 * B, cap, cap, &hs[k]);
    if (s != SMHIP_OK) Die(std::string("smhip_create: ") + smhip_status_string(s) + " (is this a gfx950 GPU? there is no CPU fallback)");
    smhip_icp_options o;
    smhip_icp_default_options(&o);
    o.max_iteration = a.iterations;
    o.early_exit = a.early_exit;
    if (a.parts > 0) o.overlap_streams = a.parts;         // parts of a batch on streams of their own (0: the library's default, two)
    if (smhip_icp_set_options(hs[k], &o) != SMHIP_OK) Die(smhip_last_error(hs[k]));
  }

  double* local_dev = nullptr;
  double* all_dev = nullptr;
  HIPOK(hipMalloc(reinterpret_cast<void**>(&local_dev), sizeof(double) * kPoseDoubles * per));
  HIPOK(hipMalloc(reinterpret_cast<void**>(&all_dev), sizeof(double) * kPoseDoubles * per * world));
  HIPOK(hipMemsetAsync(local_dev, 0, sizeof(double) * kPoseDoubles * per, stream));

  // guess: column-major 4x4; identity or a constant forward step (SURVEY.md §8(d) cfg 4; the reference front end always
  // hands its matcher an extrapolated pose, map_builder.cc:302-308)
  std::vector<double> guesses(16 * static_cast<size_t>(B), 0.0);
  for (int k = 0; k < B; ++k) { double* g = &guesses[16 * static_cast<size_t>(k)]; g[0] = g[5] = g[10] = g[15] = 1.0; g[12] = a.guess_tx; }

  // the files this rank reads, in reading order (the same walk as the loop below)
  const std::vector<int> order = smhip::kitti::ShardReadOrder(n_pairs, world, rank, B);
  // The readers fill page-locked buffers and the batch's uploads are DMA'd straight out of them (no staging copy): a
  // buffer is held from Next() until the uploads of its batch have left the host (ReleaseHeld), so the ring holds two
  // batches' worth of scans -- one being uploaded, one being read ahead.
  const int scans_per_batch = world == 1 ? B + 1 : 2 * B;
  const int ring = 2 * scans_per_batch + std::max(1, a.readers);
  std::vector<float*> ring_buffers(ring, nullptr);
  bool pinned = true;
  for (int k = 0; k < ring && pinned; ++k)
    if (hipHostMalloc(reinterpret_cast<void**>(&ring_buffers[k]), slot_floats * sizeof(float), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); pinned = false; }
  std::vector<std::vector<float>> pageable;                  // the host refused that much page-locked memory: ordinary buffers, staged uploads
  if (!pinned) {
    for (float*& b : ring_buffers) { if (b) (void)hipHostFree(b); b = nullptr; }
    pageable.assign(ring, std::vector<float>(slot_floats));
    for (int k = 0; k < ring; ++k) ring_buffers[k] = pageable[k].data();
  }
  for (int k = 0; k < NH; ++k) {
    if (smhip_reserve_batch_workspaces(hs[k]) != SMHIP_OK) Die(smhip_last_error(hs[k]));     // not inside the first batch
    HIPOK(hipStreamSynchronize(streams[k]));
  }
  // Part of bringing the process up, like the handle and its workspaces above: one batch of 32 small made-up scans (three walls of a
  // room, 8 192 points each) through the same four calls as every batch below.  The first use of a kernel loads its code object, the
  // first batched upload creates the copy stream, the first radix sort sizes rocPRIM's workspace -- 25 ms of a first batch that a
  // mapping process pays once, not per sequence.  Nothing of it survives: every slot is set again by the first real batch.
  double warmup_s = 0.0;
  if (a.warmup) {
    const auto w0 = std::chrono::steady_clock::now();
    const int WN = 8192, WS = std::min(32, B);
    // (page-locked like the readers' buffers, so that the upload takes the same road)
    float* wmem = nullptr;
    std::vector<float> wpageable;
    const size_t wfloats = 4 * static_cast<size_t>(WN);
    if (!pinned || hipHostMalloc(reinterpret_cast<void**>(&wmem), wfloats * (WS + 1) * sizeof(float), hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError(); wmem = nullptr; wpageable.resize(wfloats * (WS + 1));
    }
    float* wbase = wmem ? wmem : wpageable.data();
    std::vector<float*> wscan(WS + 1);
    for (int k = 0; k <= WS; ++k) wscan[k] = wbase + wfloats * k;
    uint32_t lcg = 12345u;
    auto u01 = [&]() { lcg = lcg * 1664525u + 1013904223u; return static_cast<float>(lcg >> 8) * (1.0f / 16777216.0f); };
    for (int k = 0; k <= WS; ++k)
      for (int i = 0; i < WN; ++i) {
        float x = 20.f * u01() - 10.f, y = 20.f * u01() - 10.f, z = 4.f * u01();
        switch (i % 3) { case 0: z = 0.002f * u01(); break; case 1: x = 10.f + 0.002f * u01(); break; default: y = 10.f + 0.002f * u01(); break; }
        float* r = &wscan[k][4 * static_cast<size_t>(i)];
        r[0] = x - 0.02f * k; r[1] = y; r[2] = z; r[3] = 0.f;
      }
    for (int hk = 0; hk < NH; ++hk) {
      std::vector<int> wslots, wn, wfrom, wto, wnt(WS);
      std::vector<const float*> wrows;
      wrows.push_back(wscan[0]); wslots.push_back(B); wn.push_back(WN); wfrom.push_back(B); wto.push_back(0);
      for (int k = 0; k < WS; ++k) {
        wrows.push_back(wscan[k + 1]); wslots.push_back(k); wn.push_back(WN);
        if (k > 0) { wfrom.push_back(k - 1); wto.push_back(k); }
      }
      if (smhip_set_sources_f32_batch(hs[hk], static_cast<int>(wslots.size()), wslots.data(), wrows.data(), wn.data()) != SMHIP_OK ||
          smhip_prepare_targets_from_sources(hs[hk], WS, wfrom.data(), wto.data(), wnt.data()) != SMHIP_OK ||
          smhip_icp_enqueue_batch(hs[hk], WS, guesses.data()) != SMHIP_OK ||
          smhip_icp_export_results_device(hs[hk], WS, local_dev) != SMHIP_OK ||
          // (what the made-up batch taught the handle about where to switch search forms is forgotten: the first real batch chooses
          // as it does without the warm-up)
          smhip_icp_forget_search_history(hs[hk]) != SMHIP_OK)
        Die(std::string("warm-up batch: ") + smhip_last_error(hs[hk]));
      HIPOK(hipStreamSynchronize(streams[hk]));
    }
    HIPOK(hipMemsetAsync(local_dev, 0, sizeof(double) * kPoseDoubles * per, stream));
    HIPOK(hipStreamSynchronize(stream));
    if (wmem) (void)hipHostFree(wmem);
    warmup_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
  }
  // the clock starts BEFORE the first file is opened: the readers start here (their head start used to be whatever the set-up above took)
  const auto t0 = std::chrono::steady_clock::now();
  ScanPrefetcher scans(files, order, a.readers, ring_buffers, slot_floats, /*hold_until_release=*/true);
  auto next_scan = [&](int expect, int* n) -> const float* {
    int fi = -1;
    const float* rows = scans.Next(n, &fi);
    if (!rows || fi != expect) Die("prefetcher out of step with the batch loop");
    if (*n < 0) Die("cannot read " + files[fi]);
    return rows;
  };
  double upload_s = 0.0, wait_s = 0.0, set_s = 0.0, prep_s = 0.0;   // rank 0's host-side split: blocked on the readers / uploads / target preparation
  auto since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); };
  int done = 0, my_pairs = 0;
  // the batch period once the pipeline is full: from the first batch's alignments being enqueued to the last batch's, over the pairs of
  // the batches after the first (a run of four batches spends a quarter of its time filling and draining)
  double first_enq_s = 0.0, last_enq_s = 0.0;
  int first_enq_pairs = 0, enq_pairs = 0;
  // SMHIP_SHARD_TIMELINE=1: where the host is (ms since the clock started) as it walks a batch -- to lay next to a kernel trace
  const bool timeline = rank == 0 && std::getenv("SMHIP_SHARD_TIMELINE") != nullptr;
  auto mark = [&](const char* what, int base) { if (timeline) std::fprintf(stderr, "[timeline] %8.3f ms  batch at %d: %s\n", since(t0) * 1e3, base, what); };
  std::vector<int> up_slots, up_n;
  std::vector<const float*> up_rows;
  for (int base = 0, turn = 0; base < per; base += B, ++turn) {
    smhip_handle h = hs[turn % NH];
    int nb = 0;
    const auto u0 = std::chrono::steady_clock::now();
    std::vector<int> from, to, nts;
    up_slots.clear(); up_n.clear(); up_rows.clear();
    int prev_pair = -2;
    auto w0 = std::chrono::steady_clock::now();
    for (int k = 0; k < B && base + k < per; ++k) {
      const int pair = (base + k) * world + rank;                        // round-robin: pair i -> rank i mod G
      if (pair >= n_pairs) break;
      // scan i = target.  When the previous slot's pair is i - 1 its source IS scan i, already part of this upload (one GPU:
      // every pair but the first of a batch); otherwise the scan is parked in slot B + k.  Either way the scans of the whole
      // batch go up in ONE call and its targets are prepared in ONE device pass (CalculateNormals as a forest of kd-trees).
      if (pair == prev_pair + 1) from.push_back(k - 1);
      else {
        int n = 0;
        up_rows.push_back(next_scan(pair, &n)); up_slots.push_back(B + k); up_n.push_back(n);
        from.push_back(B + k);
      }
      to.push_back(k);
      int n = 0;
      up_rows.push_back(next_scan(pair + 1, &n)); up_slots.push_back(k); up_n.push_back(n);   // scan i + 1 = source
      prev_pair = pair;
      ++nb;
    }
    wait_s += since(w0);
    if (nb == 0) break;
    mark("scans at hand", base);
    w0 = std::chrono::steady_clock::now();
    if (smhip_set_sources_f32_batch(h, static_cast<int>(up_slots.size()), up_slots.data(), up_rows.data(), up_n.data()) != SMHIP_OK)
      Die(std::string("upload of batch at pair ") + std::to_string(base * world + rank) + ": " + smhip_last_error(h));
    set_s += since(w0);
    mark("upload + ordering enqueued", base);
    nts.resize(nb);
    {
      w0 = std::chrono::steady_clock::now();
      if (smhip_prepare_targets_from_sources(h, nb, from.data(), to.data(), nts.data()) != SMHIP_OK) Die(std::string("prepare targets: ") + smhip_last_error(h));
      prep_s += since(w0);
      mark("targets prepared (blocking)", base);
    }
    scans.ReleaseHeld();            // prepare_targets blocked on the stream: the uploads have left the host buffers
    upload_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - u0).count();
    if (smhip_icp_enqueue_batch(h, nb, guesses.data()) != SMHIP_OK) Die(std::string("enqueue: ") + smhip_last_error(h));
    if (smhip_icp_export_results_device(h, nb, local_dev + static_cast<size_t>(kPoseDoubles) * base) != SMHIP_OK) Die(smhip_last_error(h));
    mark("alignments enqueued", base);
    if (turn == 0) { first_enq_s = since(t0); first_enq_pairs = nb; }
    last_enq_s = since(t0); enq_pairs += nb;
    done = base + nb;
    my_pairs += nb;
  }
  (void)done;
  for (int k = 1; k < NH; ++k) HIPOK(hipStreamSynchronize(streams[k]));     // (the gather goes to the first matcher's stream)
  // the ONE collective of the path: every rank's padded block of poses, device to device over xGMI
  NCCLOK(ncclAllGather(local_dev, all_dev, static_cast<size_t>(kPoseDoubles) * per, ncclDouble, comm, stream));
  HIPOK(hipStreamSynchronize(stream));
  const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  int rc = 0;
  if (rank == 0) {
    std::vector<double> all(static_cast<size_t>(kPoseDoubles) * per * world);
    HIPOK(hipMemcpy(all.data(), all_dev, sizeof(double) * all.size(), hipMemcpyDeviceToHost));
    // rank r, local slot s -> pair s * world + r;  chain pose_{i+1} = pose_i * T_i
    std::ofstream out(a.out_path);
    if (!out) Die("cannot write " + a.out_path);
    out.precision(8);
    double pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    auto write_pose = [&]() {
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) out << pose[4 * r + c] << ((r == 2 && c == 3) ? "\n" : " ");
    };
    write_pose();
    double score_sum = 0.0, iter_sum = 0.0;
    int bad = 0;
    for (int pair = 0; pair < n_pairs; ++pair) {
      const double* row = &all[static_cast<size_t>(kPoseDoubles) * (static_cast<size_t>(pair % world) * per + pair / world)];
      double T[16];
      for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) T[4 * r + c] = row[4 * c + r];   // column-major -> row-major
      if (!(row[17] >= 1.0)) ++bad;                                      // a pair that never ran left zeros
      Mul4(pose, T, pose);
      write_pose();
      score_sum += row[16]; iter_sum += row[17];
    }
    out.close();
    if (!a.quiet || bad) {
      std::printf("{\"driver\": \"smhip_shard (C++, RCCL all-gather)\", \"n_gpus\": %d, \"pairs\": %d, \"pairs_rank0\": %d, \"seconds\": %.4f, "
                  "\"pairs_per_s\": %.2f, \"read_upload_prepare_s_rank0\": %.4f, \"wait_for_readers_s_rank0\": %.4f, \"upload_s_rank0\": %.4f, "
                  "\"prepare_targets_s_rank0\": %.4f, \"mean_score\": %.6f, \"mean_iterations\": %.2f, "
                  "\"unfinished_pairs\": %d, \"batch\": %d, \"readers\": %d, \"pinned_read_buffers\": %s, \"warmup_batch_before_the_clock_s\": %.4f, \"steady_state_pairs_per_s_rank0\": %.2f, \"poses_file\": \"%s\"}\n",
                  world, n_pairs, my_pairs, elapsed, n_pairs / elapsed, upload_s, wait_s, set_s, prep_s, score_sum / n_pairs, iter_sum / n_pairs, bad, B, a.readers, pinned ? "true" : "false", warmup_s,
                  last_enq_s > first_enq_s ? (enq_pairs - first_enq_pairs) / (last_enq_s - first_enq_s) : 0.0, a.out_path.c_str());
    }
    if (bad) rc = 3;
  }
  (void)hipFree(local_dev); (void)hipFree(all_dev);
  for (int k = 0; k < NH; ++k) smhip_destroy(hs[k]);
  if (pinned) for (float* b : ring_buffers) (void)hipHostFree(b);
  NCCLOK(ncclCommDestroy(comm));
  for (int k = 0; k < NH; ++k) (void)hipStreamDestroy(streams[k]);
  return rc;
}

}  // namespace

int main(int argc, char** argv) {
  // the library splits a batch over up to four streams and RCCL brings its own: more hardware queues than the runtime's default 4, or
  // two of them share a queue and run one after the other (as bench.py does; must be set before the runtime starts)
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  Args a = Parse(argc, argv);
  if (a.id_file.empty()) a.id_file = "/tmp/smhip_shard_id_" + std::to_string(a.rank >= 0 ? static_cast<long>(getppid()) : static_cast<long>(getpid()));
  if (a.rank >= 0) {                                                     // one rank of a launched group
    const int world = a.world > 0 ? a.world : 1;
    if (a.nonce == 0) {                                                  // external launcher: every rank sees the same MASTER_PORT
      const char* e = std::getenv("SMHIP_SHARD_NONCE");
      if (!e) e = std::getenv("MASTER_PORT");
      if (e) a.nonce = std::strtoull(e, nullptr, 10);
    }
    return RunRank(a, a.rank, world, a.local_rank >= 0 ? a.local_rank : a.rank);
  }
  a.nonce = (static_cast<unsigned long long>(getpid()) << 32) ^ static_cast<unsigned long long>(std::chrono::steady_clock::now().time_since_epoch().count());
  if (a.gpus <= 1) return RunRank(a, 0, 1, 0);
  // launcher: one child process per GPU (fresh processes -- no HIP state is inherited across the fork)
  std::remove(a.id_file.c_str());
  std::vector<pid_t> kids;
  for (int r = 0; r < a.gpus; ++r) {
    const pid_t pid = fork();
    if (pid < 0) { for (pid_t k : kids) kill(k, SIGTERM); Die("fork failed"); }
    if (pid == 0) {
      std::vector<std::string> args(argv, argv + argc);
      args.push_back("--rank"); args.push_back(std::to_string(r));
      args.push_back("--world"); args.push_back(std::to_string(a.gpus));
      args.push_back("--local-rank"); args.push_back(std::to_string(r));
      args.push_back("--id-file"); args.push_back(a.id_file);
      args.push_back("--nonce"); args.push_back(std::to_string(a.nonce));
      std::vector<char*> cargs;
      for (auto& s : args) cargs.push_back(const_cast<char*>(s.c_str()));
      cargs.push_back(nullptr);
      execv("/proc/self/exe", cargs.data());
      std::perror("execv");
      _exit(127);
    }
    kids.push_back(pid);
  }
  // a rank that dies before the all-gather (unreadable scan, device error) would leave the others blocked in it for good:
  // the first failure ends the whole group
  int rc = 0;
  size_t left = kids.size();
  while (left > 0) {
    int st = 0;
    const pid_t k = waitpid(-1, &st, 0);
    if (k < 0) break;
    --left;
    const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
    if (code != 0 && rc == 0) {
      rc = code;
      for (pid_t other : kids) if (other != k) kill(other, SIGTERM);
    }
  }
  std::remove(a.id_file.c_str());
  return rc;
}
