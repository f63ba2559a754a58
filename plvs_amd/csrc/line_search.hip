// LineMatcher::SearchByKnn(Frame& CurrentFrame, const Frame& LastFrame)
// (reference src/LineMatcher.cc:303-447): the exact k = 2 search over the LBD descriptors
// runs on the device (plvs_hip_hamming_knn2, the reference's multi-index-hash tie order); the
// bookkeeping the reference does on its result — ratio test (ComputeDescriptorMatches,
// :2568-2620), TH_HIGH, "a train line matched again keeps the closer query", rotation
// histogram with its three maxima (:101-145) — is order dependent and stays on the host.
#include <cmath>
#include <vector>

#include "common.hpp"

namespace {
constexpr int kThHigh = 110;       // LineMatcher::TH_HIGH, src/LineMatcher.cc:87
constexpr int kHistoLength = 12;   // :90
}  // namespace

extern "C" int plvs_hip_lines_search_by_knn(const uint8_t* desc_last, int n_last, const uint8_t* valid_last,
                                            const float* angle_last, const uint8_t* desc_cur, int n_cur,
                                            const float* angle_cur, float nn_ratio, int check_orientation,
                                            int32_t* assigned, int* nmatches) {
  PLVS_REQUIRE(assigned && nmatches && n_last >= 0 && n_cur >= 0, "bad argument");
  *nmatches = 0;
  for (int t = 0; t < n_cur; ++t) assigned[t] = -1;
  if (n_last == 0 || n_cur == 0) return PLVS_OK;
  PLVS_REQUIRE(desc_last && valid_last && angle_last && desc_cur && angle_cur, "null argument");
  int num_valid = 0;
  for (int i = 0; i < n_last; ++i) num_valid += valid_last[i] ? 1 : 0;
  if (num_valid == 0) return PLVS_OK;   // "if(numValidLinesInLastFrame == 0) return 0"
  std::vector<int32_t> idx(2 * (size_t)n_last), dist(2 * (size_t)n_last);
  const int rc = plvs_hip_hamming_knn2(desc_last, n_last, desc_cur, n_cur, valid_last, PLVS_TIE_MIH, idx.data(),
                                       dist.data());
  if (rc != PLVS_OK) return rc;
  const float two_pi = (float)(2.0 * M_PI);           // M_2PI, :57
  const float factor = kHistoLength / two_pi;         // :313
  std::vector<float> match_dist(n_cur, 255.f);
  std::vector<int> bin_of(n_cur, -1);
  std::vector<uint8_t> matched(n_cur, 0);
  int hist[kHistoLength] = {0};
  int n = 0;
  for (int q = 0; q < n_last; ++q) {
    if (!valid_last[q]) continue;                     // compactResult: masked queries are absent
    const int t = idx[2 * q];
    if (t < 0) continue;
    const float d0 = (float)dist[2 * q];
    if (idx[2 * q + 1] >= 0 && !(d0 < nn_ratio * (float)dist[2 * q + 1])) continue;
    if (!(d0 < kThHigh)) continue;
    float rot = angle_last[q] - angle_cur[t];
    if (rot < 0.0) rot += two_pi; else if (rot > two_pi) rot -= two_pi;
    int bin = (int)std::round(rot * factor);
    if (bin == kHistoLength) bin = 0;
    if (!matched[t]) {
      matched[t] = 1;
      match_dist[t] = d0;
      assigned[t] = q;
      ++n;
      if (check_orientation) { ++hist[bin]; bin_of[t] = bin; }
    } else if (match_dist[t] > d0) {
      match_dist[t] = d0;
      assigned[t] = q;
      if (check_orientation) { --hist[bin_of[t]]; ++hist[bin]; bin_of[t] = bin; }
    }
  }
  if (check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < kHistoLength; ++i) {
      const int s = hist[i];
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
      else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
      else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
    for (int t = 0; t < n_cur; ++t)
      if (bin_of[t] >= 0 && bin_of[t] != ind1 && bin_of[t] != ind2 && bin_of[t] != ind3) {
        assigned[t] = -1;
        --n;
      }
  }
  *nmatches = n;
  return PLVS_OK;
}
