/* ORACLE — test infrastructure only (see oracle/README in DESIGN.md §3).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this; the product
 * (plvs_amd/) never does.
 *
 * CPU restatement of the step that feeds the TSDF back ends:
 *   PointCloudMapping::InitCamGridPoints               src/PointCloudMapping.cc:796-905
 *   PointCloudMapping::GeneratePointCloudInCameraFrameBGRA   :929-1226
 * with the build's compile-time switches (include/PointDefinitions.h:27-57: PointT =
 * pcl::PointSurfelSegment, COMPUTE_NORMALS 1), NeighborhoodT = EigthNeighborhoodIndicesFast
 * (src/PointCloudMapping.cc:100, include/Neighborhood.h:54-78) and the YAML defaults
 * Segmentation.on 0, filterDepth.on 0 (Examples_old/RGB-D/TUM1.yaml:186,203), so the
 * COMPUTE_SEGMENTS block (:1033-1216) and FilterDepthimage (:939-942) do not run.
 *
 * PINNED (round 5): the two functions are cut verbatim out of the reference's src/PointCloudMapping.cc at build time and
 * compiled against stand-ins for OpenCV / Eigen / PCL (oracle/ref/Makefile, oracle/ref/cloudgen_ref_wrap.cpp ->
 * oracle/_ref/libcloudgen_ref.so); tests/test_oracle_pinned_cloudgen.py compares the grid table, every byte of every point
 * record and pixelToPointIndex on images with holes at steps 1 - 4, odd sizes, padded rows and cutting limits, and
 * reference-made digests are committed (scripts/make_cloudgen_golden.py -> tests/golden/cloudgen_reference_digests.json).
 * What stays a reading is of Eigen 3.3, encoded in the stand-in (oracle/ref/eigen_full): a reduction of three terms is
 * a0 + (a1 + a2); normalize() leaves a zero vector untouched.  No FMA contraction (oracle/Makefile).
 * One thing the pin showed about the CALLER: the reference's fx, fy, cx, cy are doubles holding the FLOAT entries of K
 * (src/PointCloudMapping.cc:177-180) — oracle_cam_grid_points takes doubles: pass (double)(float)fx for the reference's table.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* pcl::PointSurfelSegment, include/PointSurfelSegment.h:63-94 — 48 bytes. */
typedef struct {
  float x, y, z;
  uint32_t kfid;
  float normal[3], normal_pad;
  uint8_t b, g, r, a; /* PCL_ADD_UNION_RGB member order in memory */
  float depth;
  uint32_t label, label_confidence;
} oracle_surfel;

/* InitCamGridPoints without distortion (mDistCoef[0] == 0 skips cv::undistortPoints, :853)
 * and without rectification: the z = 1 back-projection of every grid pixel, :871-884.
 * fx.. are `const double` there (:806-809); the pixel coordinate was stored as float (:842-843);
 * the Eigen::Vector3f constructor rounds the double quotient to float. */
void oracle_cam_grid_points(int width, int height, int step, double fx, double fy, double cx,
                            double cy, float* grid /* ngrid x 2 */) {
  int ii = 0;
  for (int m = 0; m < height; m += step)
    for (int n = 0; n < width; n += step, ii++) {
      grid[2 * ii + 0] = (float)(((float)n - cx) / fx);
      grid[2 * ii + 1] = (float)(((float)m - cy) / fy);
    }
}

static const int kDm[8] = {-1, -1, -1, 0, 1, 1, 1, 0}; /* Neighborhood.h:77 */
static const int kDn[8] = {1, 0, -1, -1, -1, 0, 1, 1}; /* Neighborhood.h:78 */
enum { kSize = 8, kiStartNormal = 1, kDiNormal = 2 };   /* Neighborhood.h:57-60 */

/* Returns the number of points.  depth: height x width f32 (row pitch in floats);
 * bgr: height x width x 3 u8 (row pitch in bytes); grid: matCamGridPoints_ after
 * InitCamGridPoints; out: capacity >= number of grid points; pixel_to_point: height x width
 * int32 (may be NULL). */
int oracle_cloudgen(const float* depth, int depth_pitch, const uint8_t* bgr, int bgr_pitch,
                    int width, int height, int step, const float* grid, double min_depth,
                    double max_depth, uint32_t kfid, oracle_surfel* out, int32_t* pixel_to_point) {
  const int gcols = (width + step - 1) / step, grows = (height + step - 1) / step;
  const int ngrid = gcols * grows;
  int* idx_cloud = (int*)malloc(sizeof(int) * (size_t)(ngrid > 0 ? ngrid : 1));
  for (int i = 0; i < ngrid; i++) idx_cloud[i] = -1;                          /* :951 */
  if (pixel_to_point)
    for (long i = 0; i < (long)width * height; i++) pixel_to_point[i] = -1;   /* :948 */

  int count = 0, ii = 0;
  for (int m = 0; m < height; m += step) {                                    /* :957 */
    const float* drow = depth + (size_t)m * depth_pitch;
    const uint8_t* crow = bgr + (size_t)m * bgr_pitch;
    for (int n = 0; n < width; n += step, ii++) {
      const float d = drow[n];
      if (((double)d > min_depth) && ((double)d < max_depth)) {               /* :967 */
        oracle_surfel p;
        memset(&p, 0, sizeof p);                 /* PointSurfelSegment(), PointSurfelSegment.h:124-138 */
        p.z = d;
        p.x = grid[2 * ii + 0] * d;                                            /* :973 */
        p.y = grid[2 * ii + 1] * d;
        p.r = crow[n * 3 + 0];                                                 /* :978  "B" */
        p.g = crow[n * 3 + 1];
        p.b = crow[n * 3 + 2];
        p.kfid = kfid;                                                         /* :983 */
        p.depth = d;                                                           /* :984 */
        out[count] = p;
        idx_cloud[ii] = count;                                                 /* :990 */
        if (pixel_to_point) pixel_to_point[(size_t)m * width + n] = count;     /* :991 */
        count++;
      }
    }
  }

  ii = 0;
  for (int m = 0; m < height; m += step)                                      /* :1000 */
    for (int n = 0; n < width; n += step, ii++) {
      if (idx_cloud[ii] < 0) continue;
      oracle_surfel* pc = &out[idx_cloud[ii]];
      const double c[3] = {pc->x, pc->y, pc->z};
      double normal[3] = {0.0, 0.0, 0.0};
      for (int kk = kiStartNormal; kk < kSize; kk += kDiNormal) {             /* :1010 */
        const int q1 = kk, q2 = (kk + kDiNormal) % kSize;
        const int m1 = m + kDm[q1] * step, n1 = n + kDn[q1] * step;           /* :888-889 */
        const int m2 = m + kDm[q2] * step, n2 = n + kDn[q2] * step;
        if (m1 < 0 || m1 >= height || n1 < 0 || n1 >= width) continue;        /* :890, :1014 */
        if (m2 < 0 || m2 >= height || n2 < 0 || n2 >= width) continue;
        const int i1 = idx_cloud[(m1 / step) * gcols + n1 / step];
        const int i2 = idx_cloud[(m2 / step) * gcols + n2 / step];
        if (i1 < 0 || i2 < 0) continue;                                        /* :1018 */
        const oracle_surfel* p1 = &out[i1];
        const oracle_surfel* p2 = &out[i2];
        const double a[3] = {p1->x - c[0], p1->y - c[1], p1->z - c[2]};
        const double b[3] = {p2->x - c[0], p2->y - c[1], p2->z - c[2]};
        normal[0] += a[1] * b[2] - a[2] * b[1];                                /* :1025 */
        normal[1] += a[2] * b[0] - a[0] * b[2];
        normal[2] += a[0] * b[1] - a[1] * b[0];
      }
      const double z2 = normal[0] * normal[0] + (normal[1] * normal[1] + normal[2] * normal[2]);
      if (z2 > 0.0) {                                                          /* :1027 normalize() */
        const double len = sqrt(z2);
        normal[0] /= len;
        normal[1] /= len;
        normal[2] /= len;
      }
      pc->normal[0] = (float)normal[0];                                        /* :1028-1030 */
      pc->normal[1] = (float)normal[1];
      pc->normal[2] = (float)normal[2];
    }
  free(idx_cloud);
  return count;
}
