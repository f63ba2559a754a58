/* ORACLE — test infrastructure only (see oracle/README in DESIGN.md §3): a CPU restatement of the
 * reference's guided line searches for single-camera frames (NlinesLeft == -1, no mpCamera2):
 *
 *   LineMatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, bLargerSearch, bMono)
 *                                                                         src/LineMatcher.cc:837-1230
 *   LineMatcher::SearchByProjection(Frame& F, const std::vector<MapLinePtr>&, bLargerSearch)
 *                                                                         src/LineMatcher.cc:1286-1560
 *   Frame::AssignFeaturesToGrid (lines) / PosLineInGrid / GetLineFeaturesInArea
 *                                                                         src/Frame.cc:756-778, 1326-1495
 *   Geom2DUtils::GetLine2dRepresentation                                  include/Geom2DUtils.h:135-159
 *   ComputeThreeMaxima                                                    src/LineMatcher.cc:101-145
 *
 * The projection of the map lines into the current frame (LineProjection::ProjectLineWithCheck,
 * MapLine::mTrackProj*) is the caller's, as in the point searches.  Parity unpinned: the reference holds
 * no test or stored output for these functions. */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GRID_ROWS 36 /* LINE_THETA_GRID_ROWS, include/Frame.h:72 */
#define GRID_COLS 160 /* LINE_D_GRID_COLS, :73 */
#define TH_HIGH 110
#define HISTO_LENGTH 12

typedef struct {
  float angle;
  int32_t class_id, octave;
  float pt_x, pt_y, response, size;
  float startPointX, startPointY, endPointX, endPointY;
  float sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
  float lineLength;
  int32_t numOfPixels;
} KeyLine;

typedef struct { float theta, d, nx, ny; } LineRep;

/* Geom2DUtils::GetLine2dRepresentation (float throughout; atan2 of floats is the float overload) */
static void line_rep(float xs, float ys, float xe, float ye, LineRep* r) {
  r->nx = (ye - ys);
  r->ny = (xs - xe);
  if (r->nx < 0) { r->nx *= -1.0f; r->ny *= -1.0f; }
  const float inv = 1.0f / (float)sqrt((double)(r->nx * r->nx + r->ny * r->ny));   /* sqrt(float) in <cmath>: float overload */
  r->nx *= inv;
  r->ny *= inv;
  r->d = r->nx * xe + r->ny * ye;
  r->theta = atan2f(r->ny, r->nx);
}

typedef struct {
  int n;
  const KeyLine* kl;        /* mvKeyLinesUn */
  const uint8_t* desc;      /* mLineDescriptors */
  const float* ur_start;    /* mvuRightLineStart (may be NULL = empty) */
  const float* ur_end;
  float bf;
  const float* scale;       /* mvLineScaleFactors */
  const float* inv_sigma2;  /* mvLineInvLevelSigma2 */
  float theta_inv, d_inv, max_diag;
  int* cell_first;          /* [GRID_COLS*GRID_ROWS+1] */
  int* cell_items;
} LFrame;

static int popcount256(const uint8_t* a, const uint8_t* b) {
  int d = 0;
  for (int i = 0; i < 32; ++i) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
  return d;
}

/* Frame::AssignFeaturesToGrid, the line part (Frame.cc:756-778) with PosLineInGrid (:1477-1495) */
static void build_grid(LFrame* F) {
  const int ncell = GRID_COLS * GRID_ROWS;
  int* cnt = (int*)calloc((size_t)ncell + 1, sizeof(int));
  int* cell = (int*)malloc(sizeof(int) * (size_t)(F->n > 0 ? F->n : 1));
  for (int i = 0; i < F->n; ++i) {
    LineRep r;
    line_rep(F->kl[i].startPointX, F->kl[i].startPointY, F->kl[i].endPointX, F->kl[i].endPointY, &r);
    const int row = (int)round((r.theta - (-M_PI_2)) * F->theta_inv);   /* double arithmetic: LINE_THETA_MIN is a double */
    const int col = (int)round((r.d + F->max_diag) * F->d_inv);
    cell[i] = -1;
    if (row < 0 || row >= GRID_ROWS || col < 0 || col >= GRID_COLS) continue;
    cell[i] = col * GRID_ROWS + row;
    cnt[cell[i] + 1]++;
  }
  for (int c = 0; c < ncell; ++c) cnt[c + 1] += cnt[c];
  F->cell_first = cnt;
  F->cell_items = (int*)malloc(sizeof(int) * (size_t)(F->n > 0 ? F->n : 1));
  int* fill = (int*)calloc((size_t)ncell, sizeof(int));
  for (int i = 0; i < F->n; ++i)
    if (cell[i] >= 0) F->cell_items[cnt[cell[i]] + fill[cell[i]]++] = i;
  free(fill);
  free(cell);
}

/* Frame::GetLineFeaturesInArea(thetaMin, thetaMax, dMin, dMax, ...) (Frame.cc:1352-1475) */
static void area(const LFrame* F, float thetaMin, float thetaMax, float dMin, float dMax, int check, int minLevel,
                 int maxLevel, int* out, int* n) {
  if (thetaMin < -M_PI_2) {
    area(F, (float)(thetaMin + M_PI), (float)(M_PI_2 - 1.1920929e-07f), -dMax, -dMin, check, minLevel, maxLevel, out, n);
    area(F, (float)(-M_PI_2 + 1.1920929e-07f), thetaMax, dMin, dMax, check, minLevel, maxLevel, out, n);
    return;
  }
  if (thetaMax > M_PI_2) {
    area(F, (float)(-M_PI_2 + 1.1920929e-07f), (float)(thetaMax - M_PI), -dMax, -dMin, check, minLevel, maxLevel, out, n);
    area(F, thetaMin, (float)(M_PI_2 - 1.1920929e-07f), dMin, dMax, check, minLevel, maxLevel, out, n);
    return;
  }
  int r0 = (int)floor((thetaMin - (-M_PI_2)) * F->theta_inv);
  if (r0 < 0) r0 = 0;
  if (r0 >= GRID_ROWS) return;
  int r1 = (int)floor((thetaMax - (-M_PI_2)) * F->theta_inv);
  if (r1 > GRID_ROWS - 1) r1 = GRID_ROWS - 1;
  if (r1 < 0) return;
  int c0 = (int)floor((dMin + F->max_diag) * F->d_inv);
  if (c0 < 0) c0 = 0;
  if (c0 >= GRID_COLS) return;
  int c1 = (int)floor((dMax + F->max_diag) * F->d_inv);
  if (c1 > GRID_COLS - 1) c1 = GRID_COLS - 1;
  if (c1 < 0) return;
  for (int ix = c0; ix <= c1; ++ix)
    for (int iy = r0; iy <= r1; ++iy) {
      const int c = ix * GRID_ROWS + iy;
      for (int k = F->cell_first[c]; k < F->cell_first[c + 1]; ++k) {
        const int j = F->cell_items[k];
        if (check && (F->kl[j].octave < minLevel || F->kl[j].octave > maxLevel)) continue;
        out[(*n)++] = j;
      }
    }
}

static int features_in_area(const LFrame* F, const LineRep* r, float dtheta, float dd, int minLevel, int maxLevel, int* out) {
  int n = 0;
  const int check = (minLevel > 0) || (maxLevel < 2147483647);
  const float tmin = r->theta - dtheta, tmax = r->theta + dtheta;
  if (fabs((double)(tmin - tmax)) > M_PI) return -1;   /* the reference exits the process here */
  area(F, tmin, tmax, r->d - dd, r->d + dd, check, minLevel, maxLevel, out, &n);
  return n;
}

/* the two point-line tests of a candidate (LineMatcher.cc:990-1046 / 1372-1428): 0 = rejected */
static int candidate_ok(const LFrame* F, int i2, const LineRep* pr, float inv_s2, float th, float uS, float vS, float uE,
                        float vE, float shiftS, float shiftE) {   /* shift = mbf * invSz (:1011) or mbf / mTrackStartDepth (:1419) */
  const KeyLine* k = &F->kl[i2];
  const float distS = pr->nx * k->startPointX + pr->ny * k->startPointY - pr->d;
  const float distE = pr->nx * k->endPointX + pr->ny * k->endPointY - pr->d;
  if (distS * distS * inv_s2 > th || distE * distE * inv_s2 > th) return 0;
  if (F->ur_start != NULL && F->ur_start[i2] >= 0 && F->ur_end[i2] >= 0) {
    LineRep rr;
    line_rep(uS - shiftS, vS, uE - shiftE, vE, &rr);
    const float dSr = rr.nx * F->ur_start[i2] + rr.ny * k->startPointY - rr.d;
    const float dEr = rr.nx * F->ur_end[i2] + rr.ny * k->endPointY - rr.d;
    if (dSr * dSr * inv_s2 > th || dEr * dEr * inv_s2 > th) return 0;
  }
  return 1;
}

static void three_maxima(const int* cnt, int* i1, int* i2, int* i3) {   /* ComputeThreeMaxima, :101-145 */
  int max1 = 0, max2 = 0, max3 = 0;
  *i1 = *i2 = *i3 = -1;
  for (int i = 0; i < HISTO_LENGTH; ++i) {
    const int s = cnt[i];
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; *i3 = *i2; *i2 = *i1; *i1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; *i3 = *i2; *i2 = i; }
    else if (s > max3) { max3 = s; *i3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { *i2 = -1; *i3 = -1; }
  else if (max3 < 0.1f * (float)max1) { *i3 = -1; }
}

static void frame_init(LFrame* F, int n, const KeyLine* kl, const uint8_t* desc, const float* urs, const float* ure, float bf,
                       const float* scale, const float* inv_sigma2, float max_diag) {
  F->n = n; F->kl = kl; F->desc = desc; F->ur_start = urs; F->ur_end = ure; F->bf = bf; F->scale = scale;
  F->inv_sigma2 = inv_sigma2; F->max_diag = max_diag;
  F->theta_inv = (float)GRID_ROWS / (float)M_PI;          /* Frame.cc:264 */
  F->d_inv = (float)GRID_COLS / (2.0f * max_diag);        /* :265 */
  build_grid(F);
}
static void frame_free(LFrame* F) { free(F->cell_first); free(F->cell_items); }

/* LineMatcher::SearchByProjection(CurrentFrame, LastFrame, bLargerSearch, bMono).
 * Per last-frame line i: valid[i] = mvpMapLines[i] && !mvbLineOutlier[i] && ProjectLineWithCheck succeeded;
 * proj[6 i ..] = uS, vS, uE, vE, invSz, invEz; octave / angle of mvKeyLinesUn[i]; desc = pML->GetDescriptor();
 * has_obs[i] = pML->Observations() > 0.  occupied[i2]: the current line already holds a map line with
 * observations.  direction: 0 none, 1 bForward, 2 bBackward.  assigned[i2] (out) = last-frame line whose map
 * line goes to current line i2, or -1.  Returns nmatches (-1: the theta interval error of the reference). */
int oracle_lines_search_by_projection_ff(int n_cur, const KeyLine* cur_kl, const uint8_t* cur_desc, const float* ur_start,
                                         const float* ur_end, float bf, const float* scale, const float* inv_sigma2,
                                         float max_diag, const uint8_t* occupied, int n_last, const uint8_t* valid,
                                         const float* proj, const int32_t* octave, const float* angle, const uint8_t* desc,
                                         const uint8_t* has_obs, int larger, int direction, float nn_ratio,
                                         int check_orientation, int32_t* assigned) {
  LFrame F;
  frame_init(&F, n_cur, cur_kl, cur_desc, ur_start, ur_end, bf, scale, inv_sigma2, max_diag);
  const float th = larger ? 5.024f : 3.84f;                 /* kChiSquareLinePointProj(Larger), :98-99 */
  const float factor = (float)(HISTO_LENGTH / (2.0 * M_PI));
  uint8_t* occ = (uint8_t*)malloc((size_t)(n_cur > 0 ? n_cur : 1));
  int* bin_of = (int*)malloc(sizeof(int) * (size_t)(n_cur > 0 ? n_cur : 1));
  int* order = (int*)malloc(sizeof(int) * (size_t)(n_last > 0 ? n_last : 1));   /* rotHist pushes, in order */
  int* order_bin = (int*)malloc(sizeof(int) * (size_t)(n_last > 0 ? n_last : 1));
  int norder = 0;
  int* cand = (int*)malloc(sizeof(int) * (size_t)(n_cur > 0 ? 2 * n_cur : 1));
  for (int i = 0; i < n_cur; ++i) { assigned[i] = -1; occ[i] = occupied ? occupied[i] : 0; bin_of[i] = -1; }
  int nmatches = 0, bad = 0;
  for (int i = 0; i < n_last && !bad; ++i) {
    if (!valid[i]) continue;
    const float* p = proj + 6 * i;
    LineRep pr;
    line_rep(p[0], p[1], p[2], p[3], &pr);
    const int lo = octave[i];
    const float sc = F.scale[lo];
    const float dT = (float)(10 * M_PI / 180.f) * sc, dD = 100.0f * sc;      /* Frame::kDeltaTheta / kDeltaD */
    int nc;
    if (direction == 1) nc = features_in_area(&F, &pr, dT, dD, lo, 2147483647, cand);
    else if (direction == 2) nc = features_in_area(&F, &pr, dT, dD, 0, lo, cand);
    else nc = features_in_area(&F, &pr, dT, dD, lo - 1, lo + 1, cand);
    if (nc < 0) { bad = 1; break; }
    if (nc == 0) continue;
    int bestDist = 256, bestIdx = -1, bestDist2 = 256;
    for (int c = 0; c < nc; ++c) {
      const int i2 = cand[c];
      if (occ[i2]) continue;
      if (!candidate_ok(&F, i2, &pr, F.inv_sigma2[lo], th, p[0], p[1], p[2], p[3], F.bf * p[4], F.bf * p[5])) continue;
      const int dist = popcount256(desc + 32 * (size_t)i, F.desc + 32 * (size_t)i2);
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx = i2; }
      else if (dist < bestDist2) { bestDist2 = dist; }
    }
    if (bestDist <= TH_HIGH && bestIdx >= 0) {
      if ((float)bestDist > nn_ratio * (float)bestDist2) continue;
      assigned[bestIdx] = i;
      occ[bestIdx] = has_obs ? has_obs[i] : 1;
      nmatches++;
      if (check_orientation) {
        float rot = angle[i] - F.kl[bestIdx].angle;
        if (rot < 0.0) rot += (float)(2.0 * M_PI); else if (rot > (float)(2.0 * M_PI)) rot -= (float)(2.0 * M_PI);
        int bin = (int)round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        order[norder] = bestIdx;
        order_bin[norder++] = bin;
      }
    }
  }
  if (!bad && check_orientation) {
    int cnt[HISTO_LENGTH] = {0}, i1, i2, i3;
    for (int k = 0; k < norder; ++k) cnt[order_bin[k]]++;
    three_maxima(cnt, &i1, &i2, &i3);
    for (int k = 0; k < norder; ++k)
      if (order_bin[k] != i1 && order_bin[k] != i2 && order_bin[k] != i3) {
        assigned[order[k]] = -1;     /* (a line pushed twice is cleared twice and counted twice, as in the reference) */
        nmatches--;
      }
  }
  free(occ); free(bin_of); free(order); free(order_bin); free(cand);
  frame_free(&F);
  return bad ? -1 : nmatches;
}

/* LineMatcher::SearchByProjection(F, vpMapLines, bLargerSearch), left image only.  Per map line: in_view =
 * mbTrackInView && !isBad(); proj[6 m ..] = mTrackProjStartX, StartY, EndX, EndY, mTrackStartDepth, mTrackEndDepth (the
 * stereo gate divides mbf by them, src/LineMatcher.cc:1419-1423); level = mnTrackScaleLevel. */
int oracle_lines_search_by_projection_map(int n_cur, const KeyLine* cur_kl, const uint8_t* cur_desc, const float* ur_start,
                                          const float* ur_end, float bf, const float* scale, const float* inv_sigma2,
                                          float max_diag, const uint8_t* occupied, int n_map, const uint8_t* in_view,
                                          const float* proj, const int32_t* level, const uint8_t* desc, const uint8_t* has_obs,
                                          int larger, float nn_ratio, int32_t* assigned) {
  LFrame F;
  frame_init(&F, n_cur, cur_kl, cur_desc, ur_start, ur_end, bf, scale, inv_sigma2, max_diag);
  const float th = larger ? 5.024f : 3.84f;
  uint8_t* occ = (uint8_t*)malloc((size_t)(n_cur > 0 ? n_cur : 1));
  int* cand = (int*)malloc(sizeof(int) * (size_t)(n_cur > 0 ? 2 * n_cur : 1));
  for (int i = 0; i < n_cur; ++i) { assigned[i] = -1; occ[i] = occupied ? occupied[i] : 0; }
  int nmatches = 0, bad = 0;
  for (int m = 0; m < n_map && !bad; ++m) {
    if (!in_view[m]) continue;
    const float* p = proj + 6 * m;
    const int lv = level[m];
    LineRep pr;
    line_rep(p[0], p[1], p[2], p[3], &pr);
    const float sc = F.scale[lv];
    const float dT = (float)(10 * M_PI / 180.f) * sc, dD = 100.0f * sc;
    const int nc = features_in_area(&F, &pr, dT, dD, lv - 1, lv, cand);
    if (nc < 0) { bad = 1; break; }
    if (nc == 0) continue;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int c = 0; c < nc; ++c) {
      const int idx = cand[c];
      if (occ[idx]) continue;
      if (!candidate_ok(&F, idx, &pr, F.inv_sigma2[lv], th, p[0], p[1], p[2], p[3], F.bf / p[4], F.bf / p[5])) continue;
      const int dist = popcount256(desc + 32 * (size_t)m, F.desc + 32 * (size_t)idx);
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = F.kl[idx].octave; bestIdx = idx; }
      else if (dist < bestDist2) { bestLevel2 = F.kl[idx].octave; bestDist2 = dist; }
    }
    if (bestDist <= TH_HIGH && bestIdx >= 0) {
      if (bestLevel == bestLevel2 && (float)bestDist > nn_ratio * (float)bestDist2) continue;
      assigned[bestIdx] = m;
      occ[bestIdx] = has_obs ? has_obs[m] : 1;
      nmatches++;
    }
  }
  free(occ); free(cand);
  frame_free(&F);
  return bad ? -1 : nmatches;
}
