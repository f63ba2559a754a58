/* TEST INFRASTRUCTURE ONLY — CPU restatement of
 *   ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints)
 *   (reference src/ORBmatcher.cc:71-244), monocular / RGB-D frames (F.Nleft == -1),
 * with the pieces of Frame it reads: AssignFeaturesToGrid / PosInGrid (src/Frame.cc:717-752,
 * 1305-1316) and GetFeaturesInArea (src/Frame.cc:1231-1303).  Plain C, sequential, the
 * reference's loop order.  Never linked into the product.  Parity unpinned by the reference
 * (it ships no test for this function); pinned by the checks in tests/test_orb_search.py. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GRID_COLS 64   /* FRAME_GRID_COLS, include/Frame.h:68 */
#define GRID_ROWS 48   /* FRAME_GRID_ROWS, include/Frame.h:67 */
#define TH_HIGH 100    /* src/ORBmatcher.cc:57 */
#define TH_LOW 50      /* :58 */

extern int oracle_descriptor_distance(const uint8_t* a, const uint8_t* b);

typedef struct {
  int n;
  const float* x;          /* mvKeysUn[i].pt */
  const float* y;
  const int32_t* octave;   /* mvKeysUn[i].octave */
  const float* u_right;    /* mvuRight[i] (<= 0: no stereo coordinate) */
  const uint8_t* desc;     /* mDescriptors, n x 32 */
  float min_x, min_y, grid_w_inv, grid_h_inv;   /* mnMinX, mnMinY, mfGridElementWidthInv / HeightInv */
  const float* scale_factors;                   /* mvScaleFactors */
} frame_view;

typedef struct {
  int m;
  const uint8_t* track_in_view;   /* mbTrackInView */
  const uint8_t* bad;             /* isBad() */
  const float* proj_x;            /* mTrackProjX, mTrackProjY, mTrackProjXR */
  const float* proj_y;
  const float* proj_xr;
  const float* view_cos;          /* mTrackViewCos */
  const float* track_depth;       /* mTrackDepth */
  const int32_t* level;           /* mnTrackScaleLevel */
  const uint8_t* desc;            /* GetDescriptor(), m x 32 */
  const uint8_t* has_obs;         /* Observations() > 0 (true for every local-map point) */
} mappoint_view;

/* occupied[i] (in): keypoint i already holds a map point whose Observations() > 0.
 * assigned[i] (out): index into the map-point list given to keypoint i by this call, or -1.
 * Returns nmatches. */
int oracle_orb_search_by_projection(const frame_view* F, const mappoint_view* M, float th, int far_points,
                                    float th_far, float nn_ratio, const uint8_t* occupied, int32_t* assigned) {
  /* ---- AssignFeaturesToGrid: cell lists in keypoint order */
  int* cell_of = (int*)malloc(sizeof(int) * (size_t)(F->n > 0 ? F->n : 1));
  int* start = (int*)calloc(GRID_COLS * GRID_ROWS + 1, sizeof(int));
  int* members = (int*)malloc(sizeof(int) * (size_t)(F->n > 0 ? F->n : 1));
  for (int i = 0; i < F->n; ++i) {
    const int px = (int)roundf((F->x[i] - F->min_x) * F->grid_w_inv);
    const int py = (int)roundf((F->y[i] - F->min_y) * F->grid_h_inv);
    cell_of[i] = (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) ? -1 : px * GRID_ROWS + py;
    if (cell_of[i] >= 0) start[cell_of[i] + 1]++;
  }
  for (int c = 0; c < GRID_COLS * GRID_ROWS; ++c) start[c + 1] += start[c];
  {
    int* fill = (int*)malloc(sizeof(int) * GRID_COLS * GRID_ROWS);
    memcpy(fill, start, sizeof(int) * GRID_COLS * GRID_ROWS);
    for (int i = 0; i < F->n; ++i)
      if (cell_of[i] >= 0) members[fill[cell_of[i]]++] = i;
    free(fill);
  }
  uint8_t* blocked = (uint8_t*)malloc((size_t)(F->n > 0 ? F->n : 1));
  for (int i = 0; i < F->n; ++i) {
    blocked[i] = occupied ? occupied[i] : 0;
    assigned[i] = -1;
  }
  int nmatches = 0;
  const int factor = th != 1.0f;
  for (int k = 0; k < M->m; ++k) {
    if (!M->track_in_view[k]) continue;
    if (far_points && M->track_depth[k] > th_far) continue;
    if (M->bad[k]) continue;
    const int level = M->level[k];
    /* RadiusByViewingCos (:246-252): the float viewCos is compared with the double 0.998 */
    float r = ((double)M->view_cos[k] > 0.998) ? 2.5f : 4.0f;
    if (factor) r *= th;
    /* ---- GetFeaturesInArea(x, y, r * scale[level], level-1, level) */
    const float x = M->proj_x[k], y = M->proj_y[k], rr = r * F->scale_factors[level];
    const int min_level = level - 1, max_level = level;
    int c0 = (int)floorf((x - F->min_x - rr) * F->grid_w_inv);
    if (c0 < 0) c0 = 0;
    if (c0 >= GRID_COLS) continue;
    int c1 = (int)ceilf((x - F->min_x + rr) * F->grid_w_inv);
    if (c1 > GRID_COLS - 1) c1 = GRID_COLS - 1;
    if (c1 < 0) continue;
    int r0 = (int)floorf((y - F->min_y - rr) * F->grid_h_inv);
    if (r0 < 0) r0 = 0;
    if (r0 >= GRID_ROWS) continue;
    int r1 = (int)ceilf((y - F->min_y + rr) * F->grid_h_inv);
    if (r1 > GRID_ROWS - 1) r1 = GRID_ROWS - 1;
    if (r1 < 0) continue;
    const int check_levels = (min_level > 0) || (max_level >= 0);
    int best = 256, best2 = 256, best_level = -1, best_level2 = -1, best_idx = -1, any = 0;
    for (int ix = c0; ix <= c1; ++ix)
      for (int iy = r0; iy <= r1; ++iy) {
        const int c = ix * GRID_ROWS + iy;
        for (int q = start[c]; q < start[c + 1]; ++q) {
          const int idx = members[q];
          if (check_levels && (F->octave[idx] < min_level || F->octave[idx] > max_level)) continue;
          const float dx = F->x[idx] - x, dy = F->y[idx] - y;
          if (!(fabsf(dx) < rr && fabsf(dy) < rr)) continue;
          any = 1;
          /* ---- the candidate loop of SearchByProjection */
          if (blocked[idx]) continue;   /* F.mvpMapPoints[idx] && Observations() > 0 */
          if (F->u_right[idx] > 0) {
            const float er = fabsf(M->proj_xr[k] - F->u_right[idx]);
            if (er > r * F->scale_factors[level]) continue;
          }
          const int dist = oracle_descriptor_distance(M->desc + 32 * (size_t)k, F->desc + 32 * (size_t)idx);
          if (dist < best) {
            best2 = best; best = dist;
            best_level2 = best_level; best_level = F->octave[idx];
            best_idx = idx;
          } else if (dist < best2) {
            best_level2 = F->octave[idx];
            best2 = dist;
          }
        }
      }
    if (!any) continue;
    if (best <= TH_HIGH) {
      if (best_level == best_level2 && (float)best > nn_ratio * (float)best2) continue;
      if (best_level != best_level2 || (float)best <= nn_ratio * (float)best2) {
        assigned[best_idx] = k;
        blocked[best_idx] = M->has_obs ? M->has_obs[k] : 1;
        nmatches++;
      }
    }
  }
  free(cell_of); free(start); free(members); free(blocked);
  return nmatches;
}


/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)
 * (reference src/ORBmatcher.cc:1774-1993), single-camera frames (Nleft == -1).  The
 * projection of the last frame's map points into the current frame (Tcw * x3Dw,
 * mpCamera->project) is the caller's: it hands over u, v and 1/z per last-frame keypoint. */
typedef struct {
  int n;                     /* LastFrame.N */
  const uint8_t* valid;      /* LastFrame.mvpMapPoints[i] && !LastFrame.mvbOutlier[i] */
  const float* u;            /* uv(0), uv(1) of the map point in the current frame */
  const float* v;
  const float* invz;         /* 1.0 / x3Dc(2) */
  const int32_t* octave;     /* LastFrame.mvKeys[i].octave */
  const float* angle;        /* LastFrame.mvKeysUn[i].angle (degrees) */
  const uint8_t* desc;       /* pMP->GetDescriptor(), n x 32 */
  const uint8_t* has_obs;    /* pMP->Observations() > 0 (NULL: all) */
} lastframe_view;

static void build_grid(const frame_view* F, int* start, int* members) {
  int* cell_of = (int*)malloc(sizeof(int) * (size_t)(F->n > 0 ? F->n : 1));
  memset(start, 0, sizeof(int) * (GRID_COLS * GRID_ROWS + 1));
  for (int i = 0; i < F->n; ++i) {
    const int px = (int)roundf((F->x[i] - F->min_x) * F->grid_w_inv);
    const int py = (int)roundf((F->y[i] - F->min_y) * F->grid_h_inv);
    cell_of[i] = (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) ? -1 : px * GRID_ROWS + py;
    if (cell_of[i] >= 0) start[cell_of[i] + 1]++;
  }
  for (int c = 0; c < GRID_COLS * GRID_ROWS; ++c) start[c + 1] += start[c];
  int* fill = (int*)malloc(sizeof(int) * GRID_COLS * GRID_ROWS);
  memcpy(fill, start, sizeof(int) * GRID_COLS * GRID_ROWS);
  for (int i = 0; i < F->n; ++i)
    if (cell_of[i] >= 0) members[fill[cell_of[i]]++] = i;
  free(fill);
  free(cell_of);
}

/* Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel) (src/Frame.cc:1231-1303) -> out[], count.
 * (default arguments: minLevel = -1, maxLevel = kMaxInt, include/Frame.h:173). */
static int features_in_area(const frame_view* F, const int* start, const int* members, float x, float y, float r,
                            int min_level, int max_level, int* out) {
  int c0 = (int)floorf((x - F->min_x - r) * F->grid_w_inv);
  if (c0 < 0) c0 = 0;
  if (c0 >= GRID_COLS) return 0;
  int c1 = (int)ceilf((x - F->min_x + r) * F->grid_w_inv);
  if (c1 > GRID_COLS - 1) c1 = GRID_COLS - 1;
  if (c1 < 0) return 0;
  int r0 = (int)floorf((y - F->min_y - r) * F->grid_h_inv);
  if (r0 < 0) r0 = 0;
  if (r0 >= GRID_ROWS) return 0;
  int r1 = (int)ceilf((y - F->min_y + r) * F->grid_h_inv);
  if (r1 > GRID_ROWS - 1) r1 = GRID_ROWS - 1;
  if (r1 < 0) return 0;
  const int check_levels = (min_level > 0) || (max_level >= 0);
  int n = 0;
  for (int ix = c0; ix <= c1; ++ix)
    for (int iy = r0; iy <= r1; ++iy) {
      const int c = ix * GRID_ROWS + iy;
      for (int q = start[c]; q < start[c + 1]; ++q) {
        const int idx = members[q];
        if (check_levels) {
          if (F->octave[idx] < min_level) continue;
          if (F->octave[idx] > max_level) continue;
        }
        const float dx = F->x[idx] - x, dy = F->y[idx] - y;
        if (fabsf(dx) < r && fabsf(dy) < r) out[n++] = idx;
      }
    }
  return n;
}

/* cur_angle: CurrentFrame.mvKeysUn[i].angle; max_x / max_y: mnMaxX / mnMaxY; mbf: CurrentFrame.mbf;
 * forward / backward: bForward / bBackward as the reference derives them from the two poses.
 * assigned[i2] (out): index of the last-frame keypoint whose map point goes to current keypoint
 * i2, or -1.  Returns nmatches. */
int oracle_orb_search_by_projection_ff(const frame_view* F, const float* cur_angle, float max_x, float max_y,
                                       float mbf, const lastframe_view* L, float th, int forward, int backward,
                                       int check_orientation, const uint8_t* occupied, int32_t* assigned) {
  int* start = (int*)malloc(sizeof(int) * (GRID_COLS * GRID_ROWS + 1));
  int* members = (int*)malloc(sizeof(int) * (size_t)(F->n > 0 ? F->n : 1));
  int* cand = (int*)malloc(sizeof(int) * (size_t)(F->n > 0 ? F->n : 1));
  uint8_t* blocked = (uint8_t*)malloc((size_t)(F->n > 0 ? F->n : 1));
  build_grid(F, start, members);
  for (int i = 0; i < F->n; ++i) { blocked[i] = occupied ? occupied[i] : 0; assigned[i] = -1; }
  /* rotHist: per bin the pushed current-keypoint indices (duplicates possible) */
  int* hist_items = (int*)malloc(sizeof(int) * (size_t)(L->n > 0 ? L->n : 1));
  int* hist_bin = (int*)malloc(sizeof(int) * (size_t)(L->n > 0 ? L->n : 1));
  int nhist = 0, nmatches = 0;
  const float factor = 12 / 360.0f;   /* HISTO_LENGTH / 360.0f */
  for (int i = 0; i < L->n; ++i) {
    if (!L->valid[i]) continue;
    const float invzc = L->invz[i];
    if (invzc < 0) continue;
    const float u = L->u[i], v = L->v[i];
    if (u < F->min_x || u > max_x) continue;
    if (v < F->min_y || v > max_y) continue;
    const int oct = L->octave[i];
    const float radius = th * F->scale_factors[oct];
    int nc;
    if (forward) nc = features_in_area(F, start, members, u, v, radius, oct, 2147483647 /* kMaxInt, the default */, cand);
    else if (backward) nc = features_in_area(F, start, members, u, v, radius, 0, oct, cand);
    else nc = features_in_area(F, start, members, u, v, radius, oct - 1, oct + 1, cand);
    if (nc == 0) continue;
    int best = 256, best_idx = -1;
    for (int c = 0; c < nc; ++c) {
      const int i2 = cand[c];
      if (blocked[i2]) continue;
      if (F->u_right[i2] > 0) {
        const float ur = u - mbf * invzc;
        const float er = fabsf(ur - F->u_right[i2]);
        if (er > radius) continue;
      }
      const int dist = oracle_descriptor_distance(L->desc + 32 * (size_t)i, F->desc + 32 * (size_t)i2);
      if (dist < best) { best = dist; best_idx = i2; }
    }
    if (best <= TH_HIGH) {
      assigned[best_idx] = i;
      blocked[best_idx] = L->has_obs ? L->has_obs[i] : 1;
      nmatches++;
      if (check_orientation) {
        float rot = L->angle[i] - cur_angle[best_idx];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        if (bin == 12) bin = 0;
        hist_items[nhist] = best_idx;
        hist_bin[nhist++] = bin;
      }
    }
  }
  if (check_orientation) {
    int count[12] = {0};
    for (int k = 0; k < nhist; ++k) count[hist_bin[k]]++;
    int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;   /* ComputeThreeMaxima, :2123-2170 */
    for (int b = 0; b < 12; ++b) {
      const int s = count[b];
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = b; }
      else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = b; }
      else if (s > max3) { max3 = s; ind3 = b; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
    for (int k = 0; k < nhist; ++k)
      if (hist_bin[k] != ind1 && hist_bin[k] != ind2 && hist_bin[k] != ind3) {
        assigned[hist_items[k]] = -1;
        nmatches--;
      }
  }
  free(start); free(members); free(cand); free(blocked); free(hist_items); free(hist_bin);
  return nmatches;
}

/* ---- ORBmatcher::SearchByBoW(KeyFramePtr& pKF, Frame& F, vpMapPointMatches), src/ORBmatcher.cc:300-507
 * (SURVEY §8 row M4), single-camera frames (F.Nleft == -1 branch, :351-374).
 *
 * DBoW2::FeatureVector is a std::map<NodeId, std::vector<unsigned int>>: here its in-order
 * traversal — node ids ascending, offsets (nnodes + 1) into the concatenated index lists.  The
 * two-iterator walk with lower_bound (:321-489) visits exactly the common node ids in ascending
 * order.  kf_valid[i] = vpMapPointsKF[i] && !isBad() (:338-344); angles are kp.angle of
 * pKF->mvKeysUn and F.mvKeys (:415-427).  assigned[iF] (out, f_n entries) = the key-frame
 * keypoint whose map point F keypoint iF receives, or -1.  Returns nmatches. */
typedef struct {
  int32_t nnodes;
  const uint32_t* node_id;
  const int32_t* offset;
  const uint32_t* index;
} featvec_view;

int oracle_orb_search_by_bow(const featvec_view* KV, const uint8_t* kf_desc, int kf_n, const uint8_t* kf_valid,
                             const float* kf_angle, const featvec_view* FV, const uint8_t* f_desc, int f_n,
                             const float* f_angle, float nn_ratio, int check_orientation, int32_t* assigned) {
  (void)kf_n;
  for (int i = 0; i < f_n; ++i) assigned[i] = -1;
  int* hist_items = (int*)malloc(sizeof(int) * (size_t)(f_n > 0 ? f_n : 1));
  int* hist_bin = (int*)malloc(sizeof(int) * (size_t)(f_n > 0 ? f_n : 1));
  int nhist = 0, nmatches = 0;
  const float factor = 12 / 360.0f;                                    /* :313-315 */
  int a = 0, b = 0;
  while (a < KV->nnodes && b < FV->nnodes) {                           /* :327 */
    if (KV->node_id[a] == FV->node_id[b]) {
      for (int ik = KV->offset[a]; ik < KV->offset[a + 1]; ++ik) {     /* :334 */
        const unsigned realIdxKF = KV->index[ik];
        if (!kf_valid[realIdxKF]) continue;
        const uint8_t* dKF = kf_desc + 32 * (size_t)realIdxKF;
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
        for (int jf = FV->offset[b]; jf < FV->offset[b + 1]; ++jf) {   /* :357 */
          const unsigned realIdxF = FV->index[jf];
          if (assigned[realIdxF] >= 0) continue;                       /* vpMapPointMatches[realIdxF] */
          const int dist = oracle_descriptor_distance(dKF, f_desc + 32 * (size_t)realIdxF);
          if (dist < bestDist1) {
            bestDist2 = bestDist1;
            bestDist1 = dist;
            bestIdxF = (int)realIdxF;
          } else if (dist < bestDist2) {
            bestDist2 = dist;
          }
        }
        if (bestDist1 <= TH_LOW) {                                     /* :407 */
          if ((float)bestDist1 < nn_ratio * (float)bestDist2) {
            assigned[bestIdxF] = (int32_t)realIdxKF;
            if (check_orientation) {
              float rot = kf_angle[realIdxKF] - f_angle[bestIdxF];
              if (rot < 0.0) rot += 360.0f;
              int bin = (int)roundf(rot * factor);
              if (bin == 12) bin = 0;
              hist_items[nhist] = bestIdxF;
              hist_bin[nhist++] = bin;
            }
            nmatches++;
          }
        }
      }
      a++;
      b++;
    } else if (KV->node_id[a] < FV->node_id[b]) {
      while (a < KV->nnodes && KV->node_id[a] < FV->node_id[b]) a++;   /* lower_bound */
    } else {
      while (b < FV->nnodes && FV->node_id[b] < KV->node_id[a]) b++;
    }
  }
  if (check_orientation) {                                              /* :491-507 */
    int count[12] = {0};
    for (int k = 0; k < nhist; ++k) count[hist_bin[k]]++;
    int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
    for (int c = 0; c < 12; ++c) {
      const int s = count[c];
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = c; }
      else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = c; }
      else if (s > max3) { max3 = s; ind3 = c; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
    for (int k = 0; k < nhist; ++k)
      if (hist_bin[k] != ind1 && hist_bin[k] != ind2 && hist_bin[k] != ind3) {
        assigned[hist_items[k]] = -1;
        nmatches--;
      }
  }
  free(hist_items);
  free(hist_bin);
  return nmatches;
}
