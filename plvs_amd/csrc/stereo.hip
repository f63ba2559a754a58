// Sparse stereo matching of ORB keypoints on a rectified pair.
//
// Replaces Frame::ComputeStereoMatches, src/Frame.cc:1780-1975 (CPU branches): for every
// left keypoint the best right keypoint in its row band by Hamming distance, an 11x11 L1
// block correlation over 11 shifts on the left keypoint's pyramid level, a parabola
// sub-pixel fit, disparity -> depth, then the 1.5 * 1.4 * median cut on the correlation score.
//
// The reference walks the left keypoints one by one through a per-row candidate table.
// Every left keypoint is independent (nothing is "claimed"), so here one 64-lane wavefront
// takes one left keypoint:
//   * lanes stride over ALL right keypoints and evaluate the row-band membership directly
//     (the table lists iR in increasing order, so "first minimum" = lowest iR among the
//     minima = the minimum of the composite key dist << 32 | iR);
//   * the 121 window pixels are spread over the lanes, each lane accumulates its share of
//     the 11 shifted absolute differences, 11 xor-shuffle reductions give the exact sums;
//   * lane 0 does the float arithmetic of :1937-1962 in the reference's operation order.
// Both pyramids stay where the two extractors left them in HBM (no image leaves the device).
// The median cut needs one order statistic of <= N scores; it runs on the host on the
// downloaded scores.
#include <algorithm>
#include <climits>
#include <vector>

#include "common.hpp"
#include "orb_internal.hpp"

namespace {

constexpr int kThHigh = 100, kThLow = 50;  // src/ORBmatcher.cc:57-58
constexpr int kWavesPerBlock = 4;

struct StereoLevels {
  const uint8_t* left[plvs::kMaxOrbLevels];
  const uint8_t* right[plvs::kMaxOrbLevels];
  int w[plvs::kMaxOrbLevels], h[plvs::kMaxOrbLevels];
  int pitch_left[plvs::kMaxOrbLevels], pitch_right[plvs::kMaxOrbLevels];
  float scale[plvs::kMaxOrbLevels], inv_scale[plvs::kMaxOrbLevels];
  int nlevels;
};

__global__ __launch_bounds__(64 * kWavesPerBlock) void stereo_match_kernel(
    StereoLevels lv, const plvs_keypoint* __restrict__ kl, const uint4* __restrict__ dl, int n_left,
    const plvs_keypoint* __restrict__ kr, const uint4* __restrict__ dr, int n_right, float mb,
    float mbf, float* __restrict__ u_right, float* __restrict__ depth, int* __restrict__ score) {
  const int lane = threadIdx.x & 63;
  const int iL = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (iL >= n_left) return;
  if (lane == 0) {
    u_right[iL] = -1.0f;  // :1782-1783
    depth[iL] = -1.0f;
    score[iL] = -1;
  }
  const plvs_keypoint kpL = kl[iL];
  const int levelL = kpL.octave;
  const float vL = kpL.y, uL = kpL.x;
  const int nRows = lv.h[0];
  if (!(vL >= 0.0f) || vL >= (float)nRows) return;  // vRowIndices[vL]
  const int row = (int)vL;
  const float maxD = mbf / mb;  // :1813-1815
  const float minD = 0.0f;
  const float minU = uL - maxD, maxU = uL - minD;
  if (maxU < 0) return;  // :1836

  // ---- Hamming argmin over the row band (:1840-1867)
  const uint4 qa = dl[2 * iL], qb = dl[2 * iL + 1];
  unsigned long long best = ((unsigned long long)kThHigh << 32);
  for (int iR = lane; iR < n_right; iR += 64) {
    const plvs_keypoint kpR = kr[iR];
    const float r = 2.0f * lv.scale[kpR.octave];
    const int maxr = (int)ceilf(kpR.y + r);  // :1804-1806
    const int minr = (int)floorf(kpR.y - r);
    if (row < minr || row > maxr) continue;
    if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
    if (!(kpR.x >= minU && kpR.x <= maxU)) continue;
    const uint4 ta = dr[2 * iR], tb = dr[2 * iR + 1];
    const int d = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
                  __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
    const unsigned long long key = ((unsigned long long)d << 32) | (unsigned)iR;
    best = key < best ? key : best;  // dist < bestDist, first minimum in iR order
  }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(best, o);
    best = other < best ? other : best;
  }
  const int bestDist = (int)(best >> 32);
  if (!(bestDist < (kThHigh + kThLow) / 2)) return;  // :1870
  const int bestIdxR = (int)(best & 0xffffffffu);

  // ---- block correlation on the pyramid level of the left keypoint (:1872-1932)
  const float uR0 = kr[bestIdxR].x;
  const float scaleFactor = lv.inv_scale[levelL];
  const float scaleduL = roundf(kpL.x * scaleFactor);
  const float scaledvL = roundf(kpL.y * scaleFactor);
  const float scaleduR0 = roundf(uR0 * scaleFactor);
  constexpr int w = 5, L = 5;
  const int lw = lv.w[levelL], lh = lv.h[levelL];
  const float iniu = scaleduR0 + L - w;
  const float endu = scaleduR0 + L + w + 1;
  if (iniu < 0 || endu >= lw) return;
  const int r0 = (int)(scaledvL - w), c0 = (int)(scaleduL - w);
  // A window leaving the level would throw in cv::Mat::rowRange / colRange; restated as "no match".
  if (r0 < 0 || r0 + 2 * w + 1 > lh || c0 < 0 || c0 + 2 * w + 1 > lw || scaleduR0 - L - w < 0) return;
  const int cr0 = (int)scaleduR0 - L - w;  // first column of the 21-wide strip in the right image

  const uint8_t* imL = lv.left[levelL];
  const uint8_t* imR = lv.right[levelL];
  const int pl = lv.pitch_left[levelL], pr = lv.pitch_right[levelL];
  int sad[2 * L + 1];
#pragma unroll
  for (int s = 0; s < 2 * L + 1; ++s) sad[s] = 0;
  for (int p = lane; p < 121; p += 64) {
    const int y = p / 11, x = p - 11 * y;
    const int a = imL[(size_t)(r0 + y) * pl + c0 + x];
    const uint8_t* rr = imR + (size_t)(r0 + y) * pr + cr0 + x;
#pragma unroll
    for (int s = 0; s < 2 * L + 1; ++s) sad[s] += abs(a - (int)rr[s]);
  }
#pragma unroll
  for (int s = 0; s < 2 * L + 1; ++s)
    for (int o = 32; o > 0; o >>= 1) sad[s] += __shfl_xor(sad[s], o);
  if (lane != 0) return;

  int bestDistC = INT_MAX, bestincR = 0;
  float vDists[2 * L + 1];
#pragma unroll
  for (int s = 0; s < 2 * L + 1; ++s) {
    const float dist = (float)sad[s];
    if (dist < (float)bestDistC) {  // float dist < int bestDist
      bestDistC = (int)dist;
      bestincR = s - L;
    }
    vDists[s] = dist;
  }
  if (bestincR == -L || bestincR == L) return;  // :1934

  float dist1 = 0.0f, dist2 = 0.0f, dist3 = 0.0f;
#pragma unroll
  for (int s = 1; s < 2 * L; ++s)
    if (s == L + bestincR) {
      dist1 = vDists[s - 1];
      dist2 = vDists[s];
      dist3 = vDists[s + 1];
    }
  const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));  // :1942
  if (deltaR < -1 || deltaR > 1) return;  // NaN (flat window) passes, as there

  float bestuR = lv.scale[levelL] * (scaleduR0 + (float)bestincR + deltaR);  // :1948
  float disparity = uL - bestuR;
  if (disparity >= minD && disparity < maxD) {
    if (disparity <= 0) {
      disparity = 0.01f;                        // double literal rounded to float
      bestuR = (float)((double)uL - 0.01);      // double subtraction, then float
    }
    depth[iL] = mbf / disparity;
    u_right[iL] = bestuR;
    score[iL] = bestDistC;
  }
}

}  // namespace

struct plvs_stereo {
  plvs_orb* left = nullptr;
  plvs_orb* right = nullptr;
  hipStream_t stream = nullptr;
  plvs::DevBuf<plvs_keypoint> kl, kr;
  plvs::DevBuf<uint8_t> dl, dr;
  plvs::DevBuf<float> u_right, depth;
  plvs::DevBuf<int> score;
  std::vector<int> h_score;
};

extern "C" {

int plvs_hip_stereo_create(plvs_orb* left, plvs_orb* right, plvs_stereo** out) {
  PLVS_REQUIRE(left != nullptr && right != nullptr && out != nullptr, "stereo_create arguments");
  plvs_stereo* s = new plvs_stereo();
  s->left = left;
  s->right = right;
  hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    plvs::set_error("stereo_create: %s", hipGetErrorString(e));
    delete s;
    return PLVS_ERR_HIP;
  }
  *out = s;
  return PLVS_OK;
}

int plvs_hip_stereo_destroy(plvs_stereo* s) {
  if (s == nullptr) return PLVS_OK;
  if (s->stream) (void)hipStreamDestroy(s->stream);
  s->kl.release();
  s->kr.release();
  s->dl.release();
  s->dr.release();
  s->u_right.release();
  s->depth.release();
  s->score.release();
  delete s;
  return PLVS_OK;
}

int plvs_hip_stereo_matches(plvs_stereo* s, const plvs_keypoint* keys_left, const uint8_t* desc_left,
                            int n_left, const plvs_keypoint* keys_right, const uint8_t* desc_right,
                            int n_right, float mb, float mbf, float* u_right, float* depth,
                            int* n_matched) {
  PLVS_REQUIRE(s != nullptr, "handle is null");
  PLVS_REQUIRE(n_left >= 0 && n_right >= 0, "negative keypoint count");
  PLVS_REQUIRE(n_left == 0 || (keys_left && desc_left && u_right && depth), "left arrays are null");
  PLVS_REQUIRE(n_right == 0 || (keys_right && desc_right), "right arrays are null");
  PLVS_REQUIRE(mb > 0.0f, "baseline must be positive");
  if (n_matched) *n_matched = 0;
  if (n_left == 0) return PLVS_OK;

  plvs::OrbPyramidView vl, vr;
  if (!plvs::orb_pyramid_view(s->left, &vl) || !plvs::orb_pyramid_view(s->right, &vr)) {
    plvs::set_error("stereo_matches: both extractors must have processed the current pair");
    return PLVS_ERR_INVALID_ARG;
  }
  PLVS_REQUIRE(vl.nlevels == vr.nlevels, "extractors differ in the number of levels");
  StereoLevels lv;
  lv.nlevels = vl.nlevels;
  for (int l = 0; l < vl.nlevels; ++l) {
    PLVS_REQUIRE(vl.w[l] == vr.w[l] && vl.h[l] == vr.h[l], "left / right pyramids differ in size");
    lv.left[l] = vl.level[l];
    lv.right[l] = vr.level[l];
    lv.w[l] = vl.w[l];
    lv.h[l] = vl.h[l];
    lv.pitch_left[l] = vl.pitch[l];
    lv.pitch_right[l] = vr.pitch[l];
    lv.scale[l] = vl.scale[l];
    lv.inv_scale[l] = vl.inv_scale[l];
  }
  for (int i = 0; i < n_left; ++i)
    PLVS_REQUIRE(keys_left[i].octave >= 0 && keys_left[i].octave < lv.nlevels, "left octave out of range");
  for (int i = 0; i < n_right; ++i)
    PLVS_REQUIRE(keys_right[i].octave >= 0 && keys_right[i].octave < lv.nlevels, "right octave out of range");

  hipStream_t st = s->stream;
  PLVS_HIP_TRY(s->kl.reserve((size_t)n_left));
  PLVS_HIP_TRY(s->dl.reserve((size_t)n_left * 32));
  PLVS_HIP_TRY(s->kr.reserve((size_t)std::max(n_right, 1)));
  PLVS_HIP_TRY(s->dr.reserve((size_t)std::max(n_right, 1) * 32));
  PLVS_HIP_TRY(s->u_right.reserve((size_t)n_left));
  PLVS_HIP_TRY(s->depth.reserve((size_t)n_left));
  PLVS_HIP_TRY(s->score.reserve((size_t)n_left));
  PLVS_HIP_TRY(hipMemcpyAsync(s->kl.p, keys_left, sizeof(plvs_keypoint) * (size_t)n_left, hipMemcpyHostToDevice, st));
  PLVS_HIP_TRY(hipMemcpyAsync(s->dl.p, desc_left, (size_t)n_left * 32, hipMemcpyHostToDevice, st));
  if (n_right > 0) {
    PLVS_HIP_TRY(hipMemcpyAsync(s->kr.p, keys_right, sizeof(plvs_keypoint) * (size_t)n_right, hipMemcpyHostToDevice, st));
    PLVS_HIP_TRY(hipMemcpyAsync(s->dr.p, desc_right, (size_t)n_right * 32, hipMemcpyHostToDevice, st));
  }
  const int blocks = (n_left + kWavesPerBlock - 1) / kWavesPerBlock;
  stereo_match_kernel<<<blocks, 64 * kWavesPerBlock, 0, st>>>(
      lv, s->kl.p, reinterpret_cast<const uint4*>(s->dl.p), n_left, s->kr.p,
      reinterpret_cast<const uint4*>(s->dr.p), n_right, mb, mbf, s->u_right.p, s->depth.p, s->score.p);
  PLVS_KERNEL_CHECK();
  s->h_score.resize((size_t)n_left);
  PLVS_HIP_TRY(hipMemcpyAsync(u_right, s->u_right.p, sizeof(float) * (size_t)n_left, hipMemcpyDeviceToHost, st));
  PLVS_HIP_TRY(hipMemcpyAsync(depth, s->depth.p, sizeof(float) * (size_t)n_left, hipMemcpyDeviceToHost, st));
  PLVS_HIP_TRY(hipMemcpyAsync(s->h_score.data(), s->score.p, sizeof(int) * (size_t)n_left, hipMemcpyDeviceToHost, st));
  PLVS_HIP_TRY(hipStreamSynchronize(st));

  // sort(vDistIdx); median = vDistIdx[size/2].first; reset everything >= 1.5*1.4*median (:1966-1980).
  std::vector<int> scores;
  scores.reserve((size_t)n_left);
  for (int i = 0; i < n_left; ++i)
    if (s->h_score[i] >= 0) scores.push_back(s->h_score[i]);
  int kept = (int)scores.size();
  if (!scores.empty()) {
    std::nth_element(scores.begin(), scores.begin() + scores.size() / 2, scores.end());
    const float median = (float)scores[scores.size() / 2];
    const float thDist = 1.5f * 1.4f * median;
    for (int i = 0; i < n_left; ++i)
      if (s->h_score[i] >= 0 && !((float)s->h_score[i] < thDist)) {
        u_right[i] = -1.0f;
        depth[i] = -1.0f;
        --kept;
      }
  }
  if (n_matched) *n_matched = kept;
  return PLVS_OK;
}

}  // extern "C"
