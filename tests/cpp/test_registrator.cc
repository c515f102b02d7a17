// Drives the C++ registrator::Interface mirror exactly the way builder/map_builder.cc:281-333 drives
// the reference: CreateMatcher(options) -> target->CalculateNormals() -> SetInputTarget ->
// SetInputSource -> Align(guess, result) -> GetFitnessScore().  Clouds come from KITTI-layout .bin
// files (float32 x,y,z,intensity) written by the pytest wrapper; the result goes to stdout as JSON.
#define SMHIP_REGISTRATOR_THROW_ON_CHECK 1
#include <cmath>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "smhip/registrator.h"

using smhip::data::InnerPointCloudData;
using smhip::data::InnerPointType;
namespace reg = smhip::registrator;

static std::vector<InnerPointType> ReadKittiBin(const char* path) {   // ros_node/kitti_reader.cc:91-121
  std::ifstream f(path, std::ios::binary);
  std::vector<InnerPointType> pts;
  float v[4];
  int i = 0;
  while (f.read(reinterpret_cast<char*>(v), sizeof(v))) {
    InnerPointType p; p.x = v[0]; p.y = v[1]; p.z = v[2]; p.intensity = v[3]; p.factor = 0.f;
    pts.push_back(p); ++i;
  }
  for (size_t k = 0; k < pts.size(); ++k) pts[k].factor = static_cast<float>(k) / pts.size();   // data_collector.h:202-204
  return pts;
}

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: %s target.bin source.bin guess_tx [xml]\n", argv[0]); return 2; }
  reg::MatcherOptions opt;
  opt.type = reg::kFastIcp;
  opt.registrator_options_node = argc > 4 ? argv[4] : "";
  // unknown option names are a CHECK failure (interface.cc:66-67)
  bool unknown_caught = false;
  try {
    reg::MatcherOptions bad = opt;
    bad.registrator_options_node = "<param name=\"no_such_option\"> 1 </param>";
    reg::CreateMatcher(bad);
  } catch (const std::exception&) { unknown_caught = true; }
  // wrong type -> nullptr (interface.cc:158-160)
  reg::MatcherOptions wrong; wrong.type = reg::kLegoLoam;
  const bool wrong_null = reg::CreateMatcher(wrong) == nullptr;

  auto matcher = reg::CreateMatcher(opt, true);
  if (!matcher) return 3;
  InnerPointCloudData::Ptr target(new InnerPointCloudData(ReadKittiBin(argv[1])));
  InnerPointCloudData::Ptr source(new InnerPointCloudData(ReadKittiBin(argv[2])));
  // target without normals is a CHECK failure (icp_fast.cc:430)
  bool no_normals_caught = false;
  try { matcher->SetInputTarget(target); } catch (const std::exception&) { no_normals_caught = true; }
  target->CalculateNormals();                       // map_builder.cc:286
  matcher->SetInputTarget(target);                  // :317
  matcher->SetInputSource(source);                  // :329
  reg::Matrix4d guess = reg::Matrix4d::Identity(), result;
  guess(0, 3) = std::atof(argv[3]);
  const bool ok = matcher->Align(guess, result);    // :333
  // a target the device refuses (NaN coordinate) must leave the matcher WITHOUT a target: the next Align -- after a perfectly
  // good SetInputSource -- fails loudly instead of matching the previous key frame; a good target brings it back
  bool refused_target_fails = false, recovers = false;
  {
    InnerPointCloudData::Ptr bad(new InnerPointCloudData(ReadKittiBin(argv[1])));
    bad->CalculateNormals();
    bad->GetEigenCloud()->points[4] = std::nan("");
    matcher->SetInputTarget(bad);
    matcher->SetInputSource(source);
    reg::Matrix4d r2;
    refused_target_fails = !matcher->Align(guess, r2);
    matcher->SetInputTarget(target);
    recovers = matcher->Align(guess, r2);
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) recovers = recovers && r2(r, c) == result(r, c);
  }
  // EnableInnerCompensation selects a branch of the reference that reads an uninitialised transform (cloud_types.cc:312): the
  // mirror refuses to align (false, result = guess) instead of ignoring the flag, and runs again once it is disabled
  bool compensation_refused = false, compensation_off_runs = false;
  {
    reg::Matrix4d r3;
    matcher->EnableInnerCompensation();
    compensation_refused = !matcher->Align(guess, r3);
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) compensation_refused = compensation_refused && r3(r, c) == guess(r, c);
    matcher->DisableInnerCompensation();
    compensation_off_runs = matcher->Align(guess, r3);
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) compensation_off_runs = compensation_off_runs && r3(r, c) == result(r, c);
  }
  // the same clouds through registrators::Ndt (type 5): no normals needed, InnerCloud AoS upload
  reg::MatcherOptions nopt; nopt.type = reg::kNdt;
  auto ndt = reg::CreateMatcher(nopt);
  reg::Matrix4d nres = reg::Matrix4d::Identity();
  bool ndt_ok = false; double ndt_score = -1; int ndt_type = -1;
  if (ndt) {
    InnerPointCloudData::Ptr target2(new InnerPointCloudData(ReadKittiBin(argv[1])));
    ndt->SetInputTarget(target2);
    ndt->SetInputSource(source);
    ndt_ok = ndt->Align(guess, nres);
    ndt_score = ndt->GetFitnessScore();
    ndt_type = (int)ndt->GetType();
  }
  // registrators::IcpUsingPointMatcher (type 1) with sampling off so the run is reproducible
  reg::MatcherOptions popt; popt.type = reg::kIcpPM;
  popt.registrator_options_node = "<param name=\"random_sampling_prob\"> 1.0 </param>";
  auto pm = reg::CreateMatcher(popt);
  reg::Matrix4d pres = reg::Matrix4d::Identity();
  bool pm_ok = false; double pm_score = -1;
  if (pm) {
    InnerPointCloudData::Ptr target3(new InnerPointCloudData(ReadKittiBin(argv[1])));
    pm->SetInputTarget(target3);
    pm->SetInputSource(source);
    pm_ok = pm->Align(guess, pres);
    pm_score = pm->GetFitnessScore();
  }
  // registrators::NdtWithGicp (type 3) with the reference's own option names
  reg::MatcherOptions gopt; gopt.type = reg::kNdtWithGicp;
  gopt.registrator_options_node = "<param name=\"use_ndt\"> true </param><param name=\"voxel_resolution\"> 0.2 </param>";
  auto gm = reg::CreateMatcher(gopt);
  reg::Matrix4d gres = reg::Matrix4d::Identity();
  bool gicp_ok = false; double gicp_score = -1; int gicp_type = -1;
  if (gm) {
    InnerPointCloudData::Ptr target4(new InnerPointCloudData(ReadKittiBin(argv[1])));
    gm->SetInputTarget(target4);
    gm->SetInputSource(source);
    gicp_ok = gm->Align(guess, gres);
    gicp_score = gm->GetFitnessScore();
    gicp_type = (int)gm->GetType();
  }
  std::printf("{\"gicp_ok\": %s, \"gicp_score\": %.17g, \"gicp_type\": %d, \"gicp_result\": [", gicp_ok ? "true" : "false", gicp_score, gicp_type);
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) std::printf("%.17g%s", gres(r, c), (r == 3 && c == 3) ? "" : ", ");
  std::printf("], ");
  std::printf("\"pm_ok\": %s, \"pm_score\": %.17g, \"pm_result\": [", pm_ok ? "true" : "false", pm_score);
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) std::printf("%.17g%s", pres(r, c), (r == 3 && c == 3) ? "" : ", ");
  std::printf("], ");
  std::printf("\"ndt_ok\": %s, \"ndt_score\": %.17g, \"ndt_type\": %d, \"ndt_result\": [", ndt_ok ? "true" : "false", ndt_score, ndt_type);
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) std::printf("%.17g%s", nres(r, c), (r == 3 && c == 3) ? "" : ", ");
  std::printf("], ");
  std::printf("\"ok\": %s, \"score\": %.17g, \"type\": %d, \"unknown_option_check\": %s, \"wrong_type_null\": %s, "
              "\"no_normals_check\": %s, \"refused_target_fails\": %s, \"recovers_after_good_target\": %s, \"compensation_refused\": %s, "
              "\"compensation_off_runs\": %s, \"target_points\": %d, \"result\": [",
              ok ? "true" : "false", matcher->GetFitnessScore(), (int)matcher->GetType(), unknown_caught ? "true" : "false",
              wrong_null ? "true" : "false", no_normals_caught ? "true" : "false", refused_target_fails ? "true" : "false",
              recovers ? "true" : "false", compensation_refused ? "true" : "false", compensation_off_runs ? "true" : "false",
              target->GetEigenCloud()->size());
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) std::printf("%.17g%s", result(r, c), (r == 3 && c == 3) ? "" : ", ");
  std::printf("]}\n");
  return 0;
}
