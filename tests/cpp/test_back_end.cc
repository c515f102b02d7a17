// The two back-end callers of the registrator boundary (include/smhip/back_end.h) on synthetic submaps.
// argv: target.bin source.bin  tx ty tz yaw_deg (source pose; the target pose is the identity)  matcher_type
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "smhip/back_end.h"

namespace reg = smhip::registrator;
using smhip::data::InnerPointCloudData;
using smhip::data::InnerPointType;

static std::vector<InnerPointType> ReadKittiBin(const char* path) {      // ros_node/kitti_reader.cc:91-121
  std::ifstream f(path, std::ios::binary);
  std::vector<InnerPointType> pts;
  float row[4];
  while (f.read(reinterpret_cast<char*>(row), sizeof(row))) { InnerPointType p; p.x = row[0]; p.y = row[1]; p.z = row[2]; p.intensity = row[3]; pts.push_back(p); }
  for (size_t i = 0; i < pts.size(); ++i) pts[i].factor = static_cast<float>(static_cast<double>(i) / pts.size());
  return pts;
}

static void PrintMatrix(const char* key, const reg::Matrix4d& m, bool last = false) {
  std::printf("\"%s\": [", key);
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) std::printf("%.17g%s", m(r, c), (r == 3 && c == 3) ? "" : ", ");
  std::printf("]%s", last ? "" : ", ");
}

int main(int argc, char** argv) {
  if (argc < 8) return 2;
  InnerPointCloudData::Ptr target(new InnerPointCloudData(ReadKittiBin(argv[1])));
  InnerPointCloudData::Ptr source(new InnerPointCloudData(ReadKittiBin(argv[2])));
  reg::Matrix4d tpose = reg::Matrix4d::Identity(), spose = reg::Matrix4d::Identity();
  const double yaw = std::atof(argv[6]) * M_PI / 180.0;
  spose(0, 0) = std::cos(yaw); spose(0, 1) = -std::sin(yaw); spose(1, 0) = std::sin(yaw); spose(1, 1) = std::cos(yaw);
  spose(0, 3) = std::atof(argv[3]); spose(1, 3) = std::atof(argv[4]); spose(2, 3) = std::atof(argv[5]);

  smhip::back_end::LoopEdge edge;
  smhip::back_end::LoopDetectorSettings settings;
  settings.accept_scan_match_score = 0.8f;                                   // config/lidar_only_kitti.xml:124
  settings.max_points = 1 << 18;
  const bool closed = smhip::back_end::CloseLoop(tpose, target, spose, source, settings, &edge);
  // a hopeless candidate: the same clouds with the source pose 30 m off must be rejected
  reg::Matrix4d far = spose; far(0, 3) += 30.0; far(1, 3) -= 20.0;
  smhip::back_end::LoopEdge bad_edge;
  const bool closed_far = smhip::back_end::CloseLoop(tpose, target, far, source, settings, &bad_edge);

  reg::MatcherOptions mopt;
  mopt.type = static_cast<reg::Type>(std::atoi(argv[7]));
  mopt.accepted_min_score = 0.7f;                                            // config/lidar_only_kitti.xml:95
  mopt.registrator_options_node = mopt.type == reg::kFastIcp ? "<param name=\"max_iteration\"> 50 </param>" : "";
  const auto sub = smhip::back_end::SubmapPairMatch(mopt, source, spose, target, tpose);
  const auto sub_far = smhip::back_end::SubmapPairMatch(mopt, source, far, target, tpose);

  // the same two pairs as ONE batch through a matcher that outlives it (SubmapPairMatchBatch), twice (arena reuse)
  auto pool_matcher = reg::CreateMatcher(mopt, false);
  std::vector<smhip::back_end::SubmapPairJob> jobs(2);
  jobs[0].source_submap_cloud = source; jobs[0].target_submap_cloud = target; jobs[0].source_first_frame_pose = spose; jobs[0].target_first_frame_pose = tpose;
  jobs[1] = jobs[0]; jobs[1].source_first_frame_pose = far;
  auto batch = smhip::back_end::SubmapPairMatchBatch(mopt, pool_matcher, jobs);
  batch = smhip::back_end::SubmapPairMatchBatch(mopt, pool_matcher, jobs);
  // registrators::Ndt has no pair slots: six pairs through a pool of three matchers on three host threads (SubmapMatcherPool)
  // must give what six single SubmapPairMatch calls give
  bool pool_equal = true;
  int pool_accepted = 0;
  {
    reg::MatcherOptions nopt; nopt.type = reg::kNdt; nopt.accepted_min_score = -1.0f;    // Ndt's score is a distance: accept everything
    std::vector<smhip::back_end::SubmapPairJob> njobs(6);
    for (int k = 0; k < 6; ++k) {
      njobs[k] = jobs[0];
      njobs[k].source_first_frame_pose(0, 3) += 0.02 * k; njobs[k].source_first_frame_pose(1, 3) -= 0.01 * k;
    }
    smhip::back_end::SubmapMatcherPool pool(nopt, 3);
    auto pooled = pool.Match(njobs);
    pooled = pool.Match(njobs);                                                            // the matchers outlive a batch
    for (int k = 0; k < 6; ++k) {
      const auto one = smhip::back_end::SubmapPairMatch(nopt, njobs[k].source_submap_cloud, njobs[k].source_first_frame_pose,
                                                         njobs[k].target_submap_cloud, njobs[k].target_first_frame_pose);
      pool_accepted += pooled[k].accepted ? 1 : 0;
      pool_equal = pool_equal && pooled[k].match_score == one.match_score;
      for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) pool_equal = pool_equal && pooled[k].transform_to_next(r, c) == one.transform_to_next(r, c);
    }
  }
  // ... and the same six pairs as ONE lock-step batch through the pair slots of one Ndt matcher (SubmapPairMatchBatch ->
  // NdtHip::AlignBatch -> smhip_ndt_align_batch): every pair's Newton / More-Thuente evaluations in its own order, the launches
  // shared -- again what six single SubmapPairMatch calls give, bit for bit
  bool ndt_batch_equal = true;
  {
    reg::MatcherOptions nopt; nopt.type = reg::kNdt; nopt.accepted_min_score = -1.0f;
    std::vector<smhip::back_end::SubmapPairJob> njobs(6);
    for (int k = 0; k < 6; ++k) {
      njobs[k] = jobs[0];
      njobs[k].source_first_frame_pose(0, 3) += 0.02 * k; njobs[k].source_first_frame_pose(1, 3) -= 0.01 * k;
    }
    auto ndt_matcher = reg::CreateMatcher(nopt, false);
    auto lock = smhip::back_end::SubmapPairMatchBatch(nopt, ndt_matcher, njobs);
    lock = smhip::back_end::SubmapPairMatchBatch(nopt, ndt_matcher, njobs);               // the K-slot handle outlives a batch
    for (int k = 0; k < 6; ++k) {
      const auto one = smhip::back_end::SubmapPairMatch(nopt, njobs[k].source_submap_cloud, njobs[k].source_first_frame_pose,
                                                         njobs[k].target_submap_cloud, njobs[k].target_first_frame_pose);
      ndt_batch_equal = ndt_batch_equal && lock[k].match_score == one.match_score && lock[k].accepted == one.accepted;
      for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) ndt_batch_equal = ndt_batch_equal && lock[k].transform_to_next(r, c) == one.transform_to_next(r, c);
    }
  }
  // the same through registrators::NdtWithGicp (host-driven BFGS per pair): four pairs, two matchers
  bool gicp_pool_equal = true;
  {
    reg::MatcherOptions gopt; gopt.type = reg::kNdtWithGicp; gopt.accepted_min_score = -1.0f;
    std::vector<smhip::back_end::SubmapPairJob> gjobs(4);
    for (int k = 0; k < 4; ++k) { gjobs[k] = jobs[0]; gjobs[k].source_first_frame_pose(0, 3) += 0.03 * k; }
    smhip::back_end::SubmapMatcherPool gpool(gopt, 2);
    const auto pooled = gpool.Match(gjobs);
    for (int k = 0; k < 4; ++k) {
      const auto one = smhip::back_end::SubmapPairMatch(gopt, gjobs[k].source_submap_cloud, gjobs[k].source_first_frame_pose,
                                                         gjobs[k].target_submap_cloud, gjobs[k].target_first_frame_pose);
      gicp_pool_equal = gicp_pool_equal && pooled[k].match_score == one.match_score;
      for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) gicp_pool_equal = gicp_pool_equal && pooled[k].transform_to_next(r, c) == one.transform_to_next(r, c);
    }
  }
  // ... and the same four pairs as ONE lock-step batch through the jobs of one NdtWithGicp matcher (SubmapPairMatchBatch ->
  // NdtGicpHip::AlignBatch -> smhip_ndt_gicp_align_batch): every pair's NDT and BFGS evaluations in its own order, the launches
  // shared -- what four single SubmapPairMatch calls give, bit for bit
  bool gicp_batch_equal = true;
  {
    reg::MatcherOptions gopt; gopt.type = reg::kNdtWithGicp; gopt.accepted_min_score = -1.0f;
    std::vector<smhip::back_end::SubmapPairJob> gjobs(4);
    for (int k = 0; k < 4; ++k) { gjobs[k] = jobs[0]; gjobs[k].source_first_frame_pose(0, 3) += 0.03 * k; }
    auto gicp_matcher = reg::CreateMatcher(gopt, false);
    auto lock = smhip::back_end::SubmapPairMatchBatch(gopt, gicp_matcher, gjobs);
    lock = smhip::back_end::SubmapPairMatchBatch(gopt, gicp_matcher, gjobs);              // the handle outlives a batch (targets kept)
    for (int k = 0; k < 4; ++k) {
      const auto one = smhip::back_end::SubmapPairMatch(gopt, gjobs[k].source_submap_cloud, gjobs[k].source_first_frame_pose,
                                                         gjobs[k].target_submap_cloud, gjobs[k].target_first_frame_pose);
      gicp_batch_equal = gicp_batch_equal && lock[k].match_score == one.match_score && lock[k].accepted == one.accepted;
      for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) gicp_batch_equal = gicp_batch_equal && lock[k].transform_to_next(r, c) == one.transform_to_next(r, c);
    }
  }
  // a loop-closure matcher that outlives its candidates
  reg::IcpPointMatcherHip keep_matcher(settings.device, 1 << 12);            // deliberately too small: must re-size itself
  keep_matcher.InitWithOptions();
  smhip::back_end::LoopEdge edge2, bad_edge2;
  const bool closed2 = smhip::back_end::CloseLoop(tpose, target, spose, source, settings, &edge2, &keep_matcher);
  const bool closed_far2 = smhip::back_end::CloseLoop(tpose, target, far, source, settings, &bad_edge2, &keep_matcher);

  std::printf("{\"pool_equal\": %s, \"ndt_batch_equal\": %s, \"gicp_pool_equal\": %s, \"gicp_batch_equal\": %s, \"pool_accepted\": %d, \"closed\": %s, \"edge_score\": %.17g, \"closed_far\": %s, ", pool_equal ? "true" : "false",
              ndt_batch_equal ? "true" : "false", gicp_pool_equal ? "true" : "false", gicp_batch_equal ? "true" : "false", pool_accepted,
              closed ? "true" : "false", edge.score, closed_far ? "true" : "false");
  PrintMatrix("edge_guess", edge.init_guess);
  PrintMatrix("edge_transform", edge.transform);
  std::printf("\"sub_accepted\": %s, \"sub_score\": %.17g, \"sub_far_accepted\": %s, \"sub_far_score\": %.17g, ",
              sub.accepted ? "true" : "false", sub.match_score, sub_far.accepted ? "true" : "false", sub_far.match_score);
  PrintMatrix("sub_guess", sub.guess);
  PrintMatrix("sub_far_transform", sub_far.transform_to_next);
  PrintMatrix("sub_far_guess", sub_far.guess);
  std::printf("\"batch_accepted\": [%s, %s], \"batch_score\": [%.17g, %.17g], \"closed2\": %s, \"closed_far2\": %s, \"edge2_score\": %.17g, ",
              batch[0].accepted ? "true" : "false", batch[1].accepted ? "true" : "false", batch[0].match_score, batch[1].match_score,
              closed2 ? "true" : "false", closed_far2 ? "true" : "false", edge2.score);
  PrintMatrix("batch0_transform", batch[0].transform_to_next);
  PrintMatrix("batch1_transform", batch[1].transform_to_next);
  PrintMatrix("edge2_transform", edge2.transform);
  PrintMatrix("sub_transform", sub.transform_to_next, true);
  std::printf("}\n");
  return 0;
}
