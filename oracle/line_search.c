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
