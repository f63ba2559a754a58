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
