/* ORACLE — test infrastructure only.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call this; the product (plvs_amd/) never does.
 *
 * CPU restatement of Frame::ComputeStereoMatches, src/Frame.cc:1780-1975:
 * row-band candidate table, Hamming argmin below (TH_HIGH + TH_LOW) / 2, 11x11 L1 block
 * correlation over 11 shifts on the pyramid level of the left keypoint, parabola sub-pixel
 * fit, disparity -> depth, and the 1.5 * 1.4 * median cut on the correlation score.
 * The CPU (#ifndef USE_CUDA) branches are the ones followed.  mMedianDepth (:1968-1971) is
 * outside this function's outputs and not restated.
 *
 * Parity unpinned: the reference holds no test or stored output for this function.
 * cv::norm(IL, IR, NORM_L1) on CV_8U is the exact integer sum of absolute differences.
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} stereo_kp; /* cv::KeyPoint, 28 bytes */

int oracle_descriptor_distance(const uint8_t* a, const uint8_t* b); /* hamming.c */

enum { TH_HIGH = 100, TH_LOW = 50 }; /* src/ORBmatcher.cc:57-58 */

typedef struct {
  int dist, idx;
} dist_idx;

static int cmp_dist_idx(const void* a, const void* b) { /* std::pair<int,int> operator< */
  const dist_idx *p = (const dist_idx*)a, *q = (const dist_idx*)b;
  if (p->dist != q->dist) return p->dist < q->dist ? -1 : 1;
  if (p->idx != q->idx) return p->idx < q->idx ? -1 : 1;
  return 0;
}

/* pyr_left / pyr_right: mvImagePyramid of the two extractors (unblurred levels, tightly
 * packed, level_w[l] x level_h[l]).  u_right / depth: n_left floats (mvuRight, mvDepth).
 * score (nullable): the correlation score pushed into vDistIdx, -1 where none was.
 * Returns the number of left keypoints that keep a depth. */
int oracle_stereo_matches(const stereo_kp* keys_left, const uint8_t* desc_left, int n_left,
                          const stereo_kp* keys_right, const uint8_t* desc_right, int n_right,
                          const uint8_t* const* pyr_left, const uint8_t* const* pyr_right,
                          const int* level_w, const int* level_h, int nlevels, const float* scale,
                          const float* inv_scale, float mb, float mbf, float* u_right, float* depth,
                          int* score) {
  (void)nlevels;
  for (int i = 0; i < n_left; i++) {
    u_right[i] = -1.0f;                                                       /* :1782-1783 */
    depth[i] = -1.0f;
    if (score) score[i] = -1;
  }
  const int thOrbDist = (TH_HIGH + TH_LOW) / 2;                               /* :1787 */
  const int nRows = level_h[0];                                               /* :1789 */

  /* vRowIndices, :1792-1810: two passes (count, fill) give the same per-row order. */
  int* row_count = (int*)calloc((size_t)nRows + 1, sizeof(int));
  for (int iR = 0; iR < n_right; iR++) {
    const float kpY = keys_right[iR].y;
    const float r = 2.0f * scale[keys_right[iR].octave];
    const int maxr = (int)ceil(kpY + r);                                      /* ceil / floor of a float sum */
    const int minr = (int)floor(kpY - r);
    for (int yi = minr; yi <= maxr; yi++)
      if (yi >= 0 && yi < nRows) row_count[yi]++;                             /* the reference does not check */
  }
  int* row_start = (int*)malloc(((size_t)nRows + 1) * sizeof(int));
  row_start[0] = 0;
  for (int y = 0; y < nRows; y++) row_start[y + 1] = row_start[y] + row_count[y];
  int* row_items = (int*)malloc((size_t)(row_start[nRows] > 0 ? row_start[nRows] : 1) * sizeof(int));
  memset(row_count, 0, ((size_t)nRows + 1) * sizeof(int));
  for (int iR = 0; iR < n_right; iR++) {
    const float kpY = keys_right[iR].y;
    const float r = 2.0f * scale[keys_right[iR].octave];
    const int maxr = (int)ceil(kpY + r);
    const int minr = (int)floor(kpY - r);
    for (int yi = minr; yi <= maxr; yi++)
      if (yi >= 0 && yi < nRows) row_items[row_start[yi] + row_count[yi]++] = iR;
  }

  const float minZ = mb;                                                      /* :1813-1815 */
  const float minD = 0;
  const float maxD = mbf / minZ;

  dist_idx* vDistIdx = (dist_idx*)malloc((size_t)(n_left > 0 ? n_left : 1) * sizeof(dist_idx));
  int nDistIdx = 0;

  for (int iL = 0; iL < n_left; iL++) {                                       /* :1821 */
    const stereo_kp* kpL = &keys_left[iL];
    const int levelL = kpL->octave;
    const float vL = kpL->y, uL = kpL->x;
    const size_t row = (size_t)vL;                                            /* vRowIndices[vL] */
    if (row >= (size_t)nRows) continue;                                       /* out of the table (UB there) */
    const int* cand = row_items + row_start[row];
    const int ncand = row_start[row + 1] - row_start[row];
    if (ncand == 0) continue;                                                 /* :1830 */
    const float minU = uL - maxD, maxU = uL - minD;
    if (maxU < 0) continue;                                                   /* :1836 */

    int bestDist = TH_HIGH;
    size_t bestIdxR = 0;
    const uint8_t* dL = desc_left + 32 * (size_t)iL;
    for (int iC = 0; iC < ncand; iC++) {                                      /* :1845 */
      const int iR = cand[iC];
      const stereo_kp* kpR = &keys_right[iR];
      if (kpR->octave < levelL - 1 || kpR->octave > levelL + 1) continue;
      const float uR = kpR->x;
      if (uR >= minU && uR <= maxU) {
        const int dist = oracle_descriptor_distance(dL, desc_right + 32 * (size_t)iR);
        if (dist < bestDist) {
          bestDist = dist;
          bestIdxR = (size_t)iR;
        }
      }
    }

    if (bestDist < thOrbDist) {                                               /* :1870 */
      const float uR0 = keys_right[bestIdxR].x;
      const float scaleFactor = inv_scale[kpL->octave];
      const float scaleduL = roundf(kpL->x * scaleFactor);
      const float scaledvL = roundf(kpL->y * scaleFactor);
      const float scaleduR0 = roundf(uR0 * scaleFactor);
      const int w = 5;
      const int lw = level_w[kpL->octave];
      const uint8_t* imL = pyr_left[kpL->octave];
      const uint8_t* imR = pyr_right[kpL->octave];
      /* rowRange / colRange take ints: the float bounds are truncated (they are integral). */
      const int r0 = (int)(scaledvL - w), c0 = (int)(scaleduL - w);

      int bestDistC = INT_MAX;                                                /* :1896 (shadows) */
      int bestincR = 0;
      const int L = 5;
      float vDists[11];
      const float iniu = scaleduR0 + L - w;                                   /* :1902-1905 */
      const float endu = scaleduR0 + L + w + 1;
      if (iniu < 0 || endu >= lw) continue;
      /* A window leaving the level image makes cv::Mat::rowRange / colRange throw there; ORB
       * keypoints sit >= 16 level pixels from the border, so this does not happen on the
       * extractor's own output.  Restated as "no match". */
      if (r0 < 0 || r0 + 2 * w + 1 > level_h[kpL->octave] || c0 < 0 || c0 + 2 * w + 1 > lw ||
          scaleduR0 - L - w < 0)
        continue;

      for (int incR = -L; incR <= +L; incR++) {                               /* :1907 */
        const int cr = (int)(scaleduR0 + incR - w);
        long sad = 0;
        for (int y = 0; y < 2 * w + 1; y++)
          for (int x = 0; x < 2 * w + 1; x++)
            sad += abs((int)imL[(size_t)(r0 + y) * lw + c0 + x] - (int)imR[(size_t)(r0 + y) * lw + cr + x]);
        const float dist = (float)(double)sad;                                /* float dist = cv::norm(...) */
        if (dist < bestDistC) {                                               /* int -> float comparison */
          bestDistC = (int)dist;
          bestincR = incR;
        }
        vDists[L + incR] = dist;
      }
      if (bestincR == -L || bestincR == L) continue;                          /* :1934 */

      const float dist1 = vDists[L + bestincR - 1];
      const float dist2 = vDists[L + bestincR];
      const float dist3 = vDists[L + bestincR + 1];
      const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));   /* :1942 */
      if (deltaR < -1 || deltaR > 1) continue;

      float bestuR = scale[kpL->octave] * ((float)scaleduR0 + (float)bestincR + deltaR); /* :1948 */
      float disparity = (uL - bestuR);
      if (disparity >= minD && disparity < maxD) {
        if (disparity <= 0) {
          disparity = 0.01;                                                   /* double literal -> float */
          bestuR = uL - 0.01;                                                 /* double subtraction -> float */
        }
        depth[iL] = mbf / disparity;
        u_right[iL] = bestuR;
        if (score) score[iL] = bestDistC;
        vDistIdx[nDistIdx].dist = bestDistC;
        vDistIdx[nDistIdx].idx = iL;
        nDistIdx++;
      }
    }
  }

  int kept = nDistIdx;
  if (nDistIdx > 0) {                 /* the reference indexes an empty vector here (UB); nothing to cut */
    qsort(vDistIdx, (size_t)nDistIdx, sizeof(dist_idx), cmp_dist_idx);        /* :1966 */
    const float median = (float)vDistIdx[nDistIdx / 2].dist;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = nDistIdx - 1; i >= 0; i--) {
      if (vDistIdx[i].dist < thDist) break;
      u_right[vDistIdx[i].idx] = -1;
      depth[vDistIdx[i].idx] = -1;
      kept--;
    }
  }
  free(vDistIdx);
  free(row_items);
  free(row_start);
  free(row_count);
  return kept;
}
