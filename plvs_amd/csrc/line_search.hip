// The three descriptor searches of LineMatcher built on the exact k = 2 LBD search:
//   SearchByKnn(Frame& CurrentFrame, const Frame& LastFrame)          src/LineMatcher.cc:303-447
//   SearchByKnn(KeyFramePtr& pKF, const Frame& F, vpMapLineMatches)   :156-301
//   SearchStereoMatchesByKnn(frame, vMatches, vValidMatches, dist)    :454-586
// The k = 2 search runs on the device (plvs_hip_hamming_knn2, the reference's multi-index-hash
// tie order); what the reference does on its result — ratio test (ComputeDescriptorMatches,
// :2568-2620), distance threshold, "a train line matched again keeps the closer query",
// rotation histogram with its three maxima (:101-145) — is order dependent, tiny (<= a few
// hundred lines) and stays on the host.  The three functions differ only in the query mask,
// the threshold (< TH_HIGH, <= TH_LOW, < descriptorDist), the stereo octave test and in what
// they hand back, so they share one pass.
#include <cmath>
#include <vector>

#include "common.hpp"

namespace {
constexpr int kThHigh = 110;       // LineMatcher::TH_HIGH, src/LineMatcher.cc:87
constexpr int kThLow = 60;         // LineMatcher::TH_LOW, :88
constexpr int kHistoLength = 12;   // :90

struct KnnPass {
  // inputs
  const uint8_t* desc_q; int nq; const uint8_t* mask_q; const float* angle_q; const int32_t* octave_q;
  const uint8_t* desc_t; int nt; const float* angle_t; const int32_t* octave_t;
  float nn_ratio; int check_orientation; float thresh; bool inclusive;
  // per train line: the query it holds (-1 none), its distance, its rotation bin (-1 none or cut),
  // and the order in which train lines were first matched (vMatches order of the stereo variant)
  std::vector<int32_t> query_of;
  std::vector<float> dist_of;
  std::vector<int> bin_of;
  std::vector<int32_t> first_order;
  int nmatches = 0;
};

int run_pass(KnnPass& P) {
  P.query_of.assign(P.nt, -1);
  P.dist_of.assign(P.nt, 255.f);
  P.bin_of.assign(P.nt, -1);
  P.first_order.clear();
  P.nmatches = 0;
  std::vector<int32_t> idx(2 * (size_t)P.nq), dist(2 * (size_t)P.nq);
  const int rc = plvs_hip_hamming_knn2(P.desc_q, P.nq, P.desc_t, P.nt, P.mask_q, PLVS_TIE_MIH, idx.data(), dist.data());
  if (rc != PLVS_OK) return rc;
  const float two_pi = (float)(2.0 * M_PI);     // M_2PI, :57
  const float factor = kHistoLength / two_pi;   // :174, :313, :471
  int hist[kHistoLength] = {0};
  int n = 0;
  for (int q = 0; q < P.nq; ++q) {
    if (P.mask_q && !P.mask_q[q]) continue;      // compactResult: masked queries are absent
    const int t = idx[2 * q];
    if (t < 0) continue;
    const float d0 = (float)dist[2 * q];
    if (idx[2 * q + 1] >= 0 && !(d0 < P.nn_ratio * (float)dist[2 * q + 1])) continue;   // ComputeDescriptorMatches
    if (P.inclusive ? !(d0 <= P.thresh) : !(d0 < P.thresh)) continue;
    if (P.octave_q && P.octave_q[q] != P.octave_t[t]) continue;                         // :492
    float rot = P.angle_q[q] - P.angle_t[t];
    if (rot < 0.0) rot += two_pi; else if (rot > two_pi) rot -= two_pi;
    int bin = (int)std::round(rot * factor);
    if (bin == kHistoLength) bin = 0;
    if (P.query_of[t] < 0) {
      P.query_of[t] = q;
      P.dist_of[t] = d0;
      P.first_order.push_back(t);
      ++n;
      if (P.check_orientation) { ++hist[bin]; P.bin_of[t] = bin; }
    } else if (P.dist_of[t] > d0) {              // replace with the better match
      P.dist_of[t] = d0;
      P.query_of[t] = q;
      if (P.check_orientation) { --hist[P.bin_of[t]]; ++hist[bin]; P.bin_of[t] = bin; }
    }
  }
  if (P.check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < kHistoLength; ++i) {     // ComputeThreeMaxima
      const int s = hist[i];
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
      else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
      else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
    for (int t = 0; t < P.nt; ++t)
      if (P.bin_of[t] >= 0) {
        if (P.bin_of[t] != ind1 && P.bin_of[t] != ind2 && P.bin_of[t] != ind3) {
          P.bin_of[t] = -2;                      // cut by the rotation check
          --n;
        }
      }
  }
  P.nmatches = n;
  return PLVS_OK;
}

int search_frame_side(const uint8_t* desc_q, int nq, const uint8_t* valid_q, const float* angle_q,
                      const uint8_t* desc_t, int nt, const float* angle_t, float nn_ratio, int check_orientation,
                      float thresh, bool inclusive, int32_t* assigned, int* nmatches) {
  *nmatches = 0;
  for (int t = 0; t < nt; ++t) assigned[t] = -1;
  if (nq == 0 || nt == 0) return PLVS_OK;
  int num_valid = 0;
  for (int i = 0; i < nq; ++i) num_valid += valid_q[i] ? 1 : 0;
  if (num_valid == 0) return PLVS_OK;   // "if(numValidLines... == 0) return 0"
  KnnPass P{desc_q, nq, valid_q, angle_q, nullptr, desc_t, nt, angle_t, nullptr, nn_ratio, check_orientation,
            thresh, inclusive};
  const int rc = run_pass(P);
  if (rc != PLVS_OK) return rc;
  for (int t = 0; t < nt; ++t)
    if (P.query_of[t] >= 0 && P.bin_of[t] != -2) assigned[t] = P.query_of[t];
  *nmatches = P.nmatches;
  return PLVS_OK;
}

}  // namespace

extern "C" {

int plvs_hip_lines_search_by_knn(const uint8_t* desc_last, int n_last, const uint8_t* valid_last,
                                 const float* angle_last, const uint8_t* desc_cur, int n_cur,
                                 const float* angle_cur, float nn_ratio, int check_orientation,
                                 int32_t* assigned, int* nmatches) {
  PLVS_REQUIRE(assigned && nmatches && n_last >= 0 && n_cur >= 0, "bad argument");
  PLVS_REQUIRE((n_last == 0 || n_cur == 0) || (desc_last && valid_last && angle_last && desc_cur && angle_cur),
               "null argument");
  return search_frame_side(desc_last, n_last, valid_last, angle_last, desc_cur, n_cur, angle_cur, nn_ratio,
                           check_orientation, (float)kThHigh, false, assigned, nmatches);   // :345  < TH_HIGH
}

int plvs_hip_lines_search_by_knn_kf(const uint8_t* desc_kf, int n_kf, const uint8_t* valid_kf,
                                    const float* angle_kf, const uint8_t* desc_f, int n_f,
                                    const float* angle_f, float nn_ratio, int check_orientation,
                                    int32_t* assigned, int* nmatches) {
  PLVS_REQUIRE(assigned && nmatches && n_kf >= 0 && n_f >= 0, "bad argument");
  PLVS_REQUIRE((n_kf == 0 || n_f == 0) || (desc_kf && valid_kf && angle_kf && desc_f && angle_f), "null argument");
  return search_frame_side(desc_kf, n_kf, valid_kf, angle_kf, desc_f, n_f, angle_f, nn_ratio, check_orientation,
                           (float)kThLow, true, assigned, nmatches);                        // :198  <= TH_LOW
}

int plvs_hip_lines_search_stereo_by_knn(const uint8_t* desc_left, int n_left, const float* angle_left,
                                        const int32_t* octave_left, const uint8_t* desc_right, int n_right,
                                        const float* angle_right, const int32_t* octave_right, float nn_ratio,
                                        int check_orientation, int descriptor_dist, int32_t* match_query,
                                        int32_t* match_train, float* match_distance, uint8_t* match_valid,
                                        int cap, int* n_out, int* nmatches) {
  PLVS_REQUIRE(n_out && nmatches && n_left >= 0 && n_right >= 0 && cap >= 0, "bad argument");
  *n_out = 0;
  *nmatches = 0;
  if (n_left == 0 || n_right == 0) return PLVS_OK;
  PLVS_REQUIRE(desc_left && angle_left && octave_left && desc_right && angle_right && octave_right, "null argument");
  PLVS_REQUIRE(match_query && match_train && match_distance && match_valid, "null output");
  KnnPass P{desc_left, n_left, nullptr, angle_left, octave_left, desc_right, n_right, angle_right, octave_right,
            nn_ratio, check_orientation, (float)descriptor_dist, false};                    // :490  < descriptorDist
  const int rc = run_pass(P);
  if (rc != PLVS_OK) return rc;
  if ((int)P.first_order.size() > cap) {
    plvs::set_error("stereo line matches: capacity %d < %d", cap, (int)P.first_order.size());
    return PLVS_ERR_CAPACITY;
  }
  int k = 0;
  for (int t : P.first_order) {   // vMatches keeps the slot of the first match of a train line (:500-505, :528)
    match_query[k] = P.query_of[t];
    match_train[k] = t;
    match_distance[k] = P.dist_of[t];
    match_valid[k] = P.bin_of[t] != -2;
    ++k;
  }
  *n_out = k;
  *nmatches = P.nmatches;
  return PLVS_OK;
}

}  // extern "C"
