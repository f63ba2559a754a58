/* TEST INFRASTRUCTURE ONLY — CPU restatement of
 *   LineMatcher::SearchByKnn(Frame& CurrentFrame, const Frame& LastFrame)
 *   (reference src/LineMatcher.cc:303-447) with LineMatcher::ComputeDescriptorMatches
 *   (:2568-2620) and ComputeThreeMaxima (:101-145), for single-camera frames.
 * Plain C, the reference's loop order.  Never linked into the product.  Parity unpinned by
 * the reference (no test ships for it); pinned by tests/test_line_search.py. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define TH_HIGH 110        /* LineMatcher::TH_HIGH, src/LineMatcher.cc:87 */
#define HISTO_LENGTH 12    /* :90 */

extern void oracle_knn2_mih(const uint8_t* q, int nq, const uint8_t* t, int nt, const uint8_t* qmask,
                            int32_t* idx, int32_t* dist);

/* last frame = query side (descriptors, angle of mvKeyLinesUn, valid = has a map line and is
 * not an outlier), current frame = train side.  assigned[t] (out) = index of the last-frame
 * line whose map line goes to current line t, or -1.  Returns the reference's return value. */
int oracle_lines_search_by_knn(const uint8_t* desc_last, int n_last, const uint8_t* valid_last,
                               const float* angle_last, const uint8_t* desc_cur, int n_cur,
                               const float* angle_cur, float nn_ratio, int check_orientation,
                               int32_t* assigned) {
  const float M_2PI_F = (float)(2.0 * 3.14159265358979323846);   /* M_2PI, :57 */
  const float factor = HISTO_LENGTH / M_2PI_F;          /* :313 */
  for (int t = 0; t < n_cur; ++t) assigned[t] = -1;
  int num_valid_last = 0;
  for (int i = 0; i < n_last; ++i) num_valid_last += valid_last[i] ? 1 : 0;
  if (num_valid_last == 0 || n_cur == 0) return 0;
  int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)n_last);
  int32_t* dist = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)n_last);
  oracle_knn2_mih(desc_last, n_last, desc_cur, n_cur, valid_last, idx, dist);
  float* match_dist = (float*)malloc(sizeof(float) * (size_t)n_cur);
  int* bin_of = (int*)malloc(sizeof(int) * (size_t)n_cur);   /* rotation bin a matched train line sits in */
  uint8_t* matched = (uint8_t*)calloc((size_t)n_cur, 1);
  for (int t = 0; t < n_cur; ++t) { match_dist[t] = 255; bin_of[t] = -1; }
  int hist[HISTO_LENGTH] = {0};
  int n = 0;
  for (int q = 0; q < n_last; ++q) {
    if (!valid_last[q]) continue;                       /* compactResult: masked queries are absent */
    const int t = idx[2 * q];
    if (t < 0) continue;
    const float d0 = (float)dist[2 * q];
    if (idx[2 * q + 1] >= 0 && !(d0 < nn_ratio * (float)dist[2 * q + 1])) continue;   /* ComputeDescriptorMatches */
    if (!(d0 < TH_HIGH)) continue;
    float rot = angle_last[q] - angle_cur[t];
    if (rot < 0.0) rot += M_2PI_F; else if (rot > M_2PI_F) rot -= M_2PI_F;
    int bin = (int)roundf(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    if (!matched[t]) {
      matched[t] = 1;
      match_dist[t] = d0;
      assigned[t] = q;
      ++n;
      if (check_orientation) { hist[bin]++; bin_of[t] = bin; }
    } else if (match_dist[t] > d0) {                    /* a train line matched again: keep the closer query */
      match_dist[t] = d0;
      assigned[t] = q;
      if (check_orientation) { hist[bin_of[t]]--; hist[bin]++; bin_of[t] = bin; }
    }
  }
  if (check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < HISTO_LENGTH; ++i) {            /* ComputeThreeMaxima */
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
  free(idx); free(dist); free(match_dist); free(bin_of); free(matched);
  return n;
}

/* ---- the three-maxima cut shared by the two restatements below (ComputeThreeMaxima, :101-145) */
static void three_maxima12(const int* hist, int* ind1, int* ind2, int* ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  *ind1 = *ind2 = *ind3 = -1;
  for (int i = 0; i < HISTO_LENGTH; ++i) {
    const int s = hist[i];
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
    else if (s > max3) { max3 = s; *ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

static int rot_bin(float angle_q, float angle_t) {
  const float M_2PI_F = (float)(2.0 * 3.14159265358979323846);
  const float factor = HISTO_LENGTH / M_2PI_F;
  float rot = angle_q - angle_t;
  if (rot < 0.0) rot += M_2PI_F; else if (rot > M_2PI_F) rot -= M_2PI_F;
  int bin = (int)roundf(rot * factor);
  if (bin == HISTO_LENGTH) bin = 0;
  return bin;
}

/* LineMatcher::SearchByKnn(KeyFramePtr& pKF, const Frame& F, vpMapLineMatches), :156-301.
 * Key frame = query side, frame = train side; a match needs distance <= TH_LOW (:198). */
#define TH_LOW 60          /* LineMatcher::TH_LOW, :88 */
int oracle_lines_search_by_knn_kf(const uint8_t* desc_kf, int n_kf, const uint8_t* valid_kf,
                                  const float* angle_kf, const uint8_t* desc_f, int n_f, const float* angle_f,
                                  float nn_ratio, int check_orientation, int32_t* assigned) {
  for (int t = 0; t < n_f; ++t) assigned[t] = -1;
  int num_valid = 0;
  for (int i = 0; i < n_kf; ++i) num_valid += valid_kf[i] ? 1 : 0;            /* :177-186 */
  if (num_valid == 0 || n_f == 0) return 0;
  int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)n_kf);
  int32_t* dist = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)n_kf);
  oracle_knn2_mih(desc_kf, n_kf, desc_f, n_f, valid_kf, idx, dist);
  float* match_dist = (float*)malloc(sizeof(float) * (size_t)n_f);
  int* bin_of = (int*)malloc(sizeof(int) * (size_t)n_f);
  uint8_t* matched = (uint8_t*)calloc((size_t)n_f, 1);
  for (int t = 0; t < n_f; ++t) { match_dist[t] = 255; bin_of[t] = -1; }
  int hist[HISTO_LENGTH] = {0};
  int n = 0;
  for (int q = 0; q < n_kf; ++q) {
    if (!valid_kf[q]) continue;
    const int t = idx[2 * q];
    if (t < 0) continue;
    const float d0 = (float)dist[2 * q];
    if (idx[2 * q + 1] >= 0 && !(d0 < nn_ratio * (float)dist[2 * q + 1])) continue;
    if (!(d0 <= TH_LOW)) continue;                                             /* :198 */
    const int bin = rot_bin(angle_kf[q], angle_f[t]);
    if (!matched[t]) {
      matched[t] = 1;
      match_dist[t] = d0;
      assigned[t] = q;
      ++n;
      if (check_orientation) { hist[bin]++; bin_of[t] = bin; }
    } else if (match_dist[t] > d0) {                                           /* :236 */
      match_dist[t] = d0;
      assigned[t] = q;
      if (check_orientation) { hist[bin_of[t]]--; hist[bin]++; bin_of[t] = bin; }
    }
  }
  if (check_orientation) {
    int ind1, ind2, ind3;
    three_maxima12(hist, &ind1, &ind2, &ind3);
    for (int t = 0; t < n_f; ++t)
      if (bin_of[t] >= 0 && bin_of[t] != ind1 && bin_of[t] != ind2 && bin_of[t] != ind3) {
        assigned[t] = -1;
        --n;
      }
  }
  free(idx); free(dist); free(match_dist); free(bin_of); free(matched);
  return n;
}

/* LineMatcher::SearchStereoMatchesByKnn(frame, vMatches, vValidMatches, descriptorDist), :454-586,
 * with USE_REPLACE_WITH_BETTER_IN_STEREO_MATCHING 1 (:42).  Left = query (mask all ones), right =
 * train.  vMatches / vValidMatches come back as four arrays of *n_out entries (<= n_right).
 * Returns numValidMatches. */
int oracle_lines_search_stereo_by_knn(const uint8_t* desc_left, int n_left, const float* angle_left,
                                      const int32_t* octave_left, const uint8_t* desc_right, int n_right,
                                      const float* angle_right, const int32_t* octave_right, float nn_ratio,
                                      int check_orientation, int descriptor_dist, int32_t* match_query,
                                      int32_t* match_train, float* match_distance, uint8_t* match_valid,
                                      int* n_out) {
  *n_out = 0;
  if (n_left == 0 || n_right == 0) return 0;
  int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)n_left);
  int32_t* dist = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)n_left);
  oracle_knn2_mih(desc_left, n_left, desc_right, n_right, NULL, idx, dist);
  uint8_t* matched = (uint8_t*)calloc((size_t)n_right, 1);                    /* vbMatched */
  int* stored = (int*)calloc((size_t)n_right, sizeof(int));                   /* vStoredMatchIndex */
  int* bin_of_match = (int*)malloc(sizeof(int) * (size_t)n_right);            /* bin of vMatches[k], -1 none */
  int hist[HISTO_LENGTH] = {0};
  int nm = 0, num_valid = 0;
  for (int q = 0; q < n_left; ++q) {
    const int t = idx[2 * q];
    if (t < 0) continue;
    const float d0 = (float)dist[2 * q];
    if (idx[2 * q + 1] >= 0 && !(d0 < nn_ratio * (float)dist[2 * q + 1])) continue;   /* vValidDescriptorMatches */
    if (!(d0 < descriptor_dist)) continue;                                     /* :490 */
    if (octave_left[q] != octave_right[t]) continue;                           /* :492 */
    if (!matched[t]) {
      matched[t] = 1;
      match_query[nm] = q; match_train[nm] = t; match_distance[nm] = d0; match_valid[nm] = 1;
      bin_of_match[nm] = -1;
      num_valid++;
      stored[t] = nm;
      if (check_orientation) {
        const int bin = rot_bin(angle_left[q], angle_right[t]);
        hist[bin]++;
        bin_of_match[nm] = bin;
      }
      nm++;
    } else {
      const int old = stored[t];
      if (match_distance[old] > d0) {                                          /* :525 */
        match_query[old] = q; match_train[old] = t; match_distance[old] = d0;
        if (check_orientation) {
          hist[bin_of_match[old]]--;
          const int bin = rot_bin(angle_left[q], angle_right[t]);
          hist[bin]++;
          bin_of_match[old] = bin;
        }
      }
    }
  }
  if (check_orientation) {
    int ind1, ind2, ind3;
    three_maxima12(hist, &ind1, &ind2, &ind3);
    for (int k = 0; k < nm; ++k)
      if (bin_of_match[k] != ind1 && bin_of_match[k] != ind2 && bin_of_match[k] != ind3) {
        match_valid[k] = 0;                                                    /* :577 */
        num_valid--;
      }
  }
  *n_out = nm;
  free(idx); free(dist); free(matched); free(stored); free(bin_of_match);
  return num_valid;
}
