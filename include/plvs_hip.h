/*
 * plvs_hip.h — C ABI of the MI355X (gfx950) hot path for PLVS.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ types, no
 * torch types.  Every entry point replaces one reference call site (cited as
 * file:line relative to the PLVS tree).  All functions return an int status
 * (PLVS_OK == 0, < 0 on error), never throw, and write only into
 * caller-provided memory.  One handle per logical instance; handles are
 * thread-compatible (no shared mutable globals).
 *
 * Two flavours per operation:
 *   - plain      : host pointers in/out, synchronous (what PLVS' CPU code
 *                  hands over today);
 *   - `_dev`     : device pointers in/out + a hipStream_t (passed as void*),
 *                  asynchronous on that stream (inputs already resident in
 *                  HBM — what bench.py times).
 */
#ifndef PLVS_HIP_H_
#define PLVS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
#define PLVS_OK 0
#define PLVS_ERR_INVALID_ARG (-1)
#define PLVS_ERR_HIP (-2)         /* a HIP runtime call failed; see plvs_hip_last_error */
#define PLVS_ERR_NO_DEVICE (-3)
#define PLVS_ERR_CAPACITY (-4)    /* a fixed-capacity device structure overflowed */
#define PLVS_ERR_EMPTY (-5)       /* empty input where the reference bails out */
#define PLVS_ERR_HALO (-6)        /* sharded map: chunks of other ranks are needed first (plvs_hip_tsdf_chisel_halo_*) */
#define PLVS_ERR_COMM_FATAL (-7)  /* a sharded step failed past its counts exchange with no way through its collectives:  */
                                  /* abort / destroy the communicator on every rank, rebuild the sharded map           */

/* Human-readable description of the last error on the calling thread. */
const char* plvs_hip_last_error(void);
/* ABI version of this header (bumped on incompatible change). */
int plvs_hip_abi_version(void);
int plvs_hip_device_count(int* count);
int plvs_hip_set_device(int device);
/* hipDeviceSynchronize on the current device. */
int plvs_hip_synchronize(void);
/* Frees the calling thread's staging buffers (pinned host + device block + stream) that the host
 * flavours of the small matcher calls keep between calls. */
int plvs_hip_release_thread_buffers(void);

/* Thin device-memory helpers so that a host (C++/cgo/ctypes) caller can keep
 * inputs resident without linking HIP itself. */
int plvs_hip_malloc(void** dptr, size_t bytes);
int plvs_hip_free(void* dptr);
int plvs_hip_memcpy_h2d(void* dst, const void* src, size_t bytes);
int plvs_hip_memcpy_d2h(void* dst, const void* src, size_t bytes);
int plvs_hip_memset(void* dst, int value, size_t bytes);

/* ------------------------------------------------------- Hamming k=2 search
 * Exact 2-nearest-neighbour search in Hamming space over 256-bit (32-byte)
 * descriptors; every query is compared against the whole train set.
 *
 * Replaces:
 *   - BinaryDescriptorMatcher::knnMatch(query, train, matches, k=2, mask, compact)
 *       Thirdparty/line_descriptor/src/binary_descriptor_matcher_custom.cpp:258-336
 *       (called from LineMatcher::ComputeDescriptorMatches, src/LineMatcher.cc:2572)
 *       -> tie_rule = PLVS_TIE_MIH
 *   - cv::BFMatcher(NORM_HAMMING).knnMatch(k=2)  (Frame::BFmatcher, src/Frame.cc:118, 2977)
 *       -> tie_rule = PLVS_TIE_LOWEST_INDEX
 *   - the distance itself is ORBmatcher::DescriptorDistance, src/ORBmatcher.cc:2198-2225
 *
 * query  : nq x 32 bytes, row-major.  train : nt x 32 bytes.
 * qmask  : NULL or nq bytes; qmask[i]==0 means "skip query i" (the
 *          reference's Qx1 mask): its outputs are set to -1.
 * idx    : nq x 2 int32, train index of best / second best (-1 if absent).
 * dist   : nq x 2 int32, Hamming distances 0..256 (-1 if absent).
 *
 * Tie rule among equal distances:
 *   PLVS_TIE_LOWEST_INDEX  lowest train index first.
 *   PLVS_TIE_MIH           the discovery order of the reference's
 *                          multi-index-hash search (Mihasher(256,32)::query,
 *                          binary_descriptor_matcher_custom.cpp:633-752):
 *                          (min per-byte distance s, first byte k reaching it,
 *                          enumeration rank of that byte's xor pattern,
 *                          train index).
 * Errors: nq==0 or nt==0 -> PLVS_ERR_EMPTY (reference prints and returns,
 * :263-267), outputs untouched.
 */
#define PLVS_TIE_LOWEST_INDEX 0
#define PLVS_TIE_MIH 1

int plvs_hip_hamming_knn2(const uint8_t* query, int nq, const uint8_t* train, int nt,
                          const uint8_t* qmask, int tie_rule, int32_t* idx, int32_t* dist);
int plvs_hip_hamming_knn2_dev(const uint8_t* d_query, int nq, const uint8_t* d_train, int nt,
                              const uint8_t* d_qmask, int tie_rule, int32_t* d_idx,
                              int32_t* d_dist, void* stream);

/* ------------------------------------------------------------ ORB extraction
 * Replaces ORBextractor (include/ORBextractor.h:68-170, src/ORBextractor.cc):
 *   ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)   :446
 *   int ORBextractor::operator()(image, mask [ignored], keypoints, descriptors,
 *                                vLappingArea)                                          :1245
 * called from Frame::ExtractORB (src/Frame.cc:806-813).  CPU-build semantics
 * (the `#ifndef USE_CUDA` branches): 8-bit bilinear pyramid, cv::FAST per 35-px
 * cell with the two-threshold rule, quadtree distribution, intensity-centroid
 * angle, 7x7 sigma-2 blur, 256-bit rBRIEF.
 *
 * plvs_keypoint is cv::KeyPoint field for field (pt.x, pt.y, size, angle,
 * response, octave, class_id = -1), 28 bytes.
 */
typedef struct plvs_keypoint {
  float x, y;
  float size;
  float angle;
  float response;
  int32_t octave;
  int32_t class_id;
} plvs_keypoint;

typedef struct plvs_orb plvs_orb;

int plvs_hip_orb_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast,
                        int min_th_fast, plvs_orb** out);
int plvs_hip_orb_destroy(plvs_orb* h);
/* Getters of ORBextractor.h:90-113. */
int plvs_hip_orb_get_levels(plvs_orb* h);
float plvs_hip_orb_get_scale_factor(plvs_orb* h);
int plvs_hip_orb_get_scale_tables(plvs_orb* h, float* scale, float* inv_scale, float* sigma2,
                                  float* inv_sigma2);
int plvs_hip_orb_features_per_level(plvs_orb* h, int* out);

/* operator().  image: h rows of `stride` bytes, CV_8UC1.  (lap0, lap1) is
 * vLappingArea: keypoints with lap0 <= x <= lap1 are packed from the back
 * (ORBextractor.cc:1357-1378).  kps / desc hold `cap` entries (desc 32 bytes
 * each).  *n = number of keypoints found (nothing is written if n > cap),
 * *mono_index = the reference's return value.  Empty image -> PLVS_ERR_EMPTY
 * and *mono_index = -1 (the reference returns -1). */
int plvs_hip_orb_extract(plvs_orb* h, const uint8_t* image, int w, int hh, int stride, int lap0,
                         int lap1, plvs_keypoint* kps, uint8_t* desc, int cap, int* n,
                         int* mono_index);
/* Same with the image already resident in device memory (it must be complete,
 * i.e. the producing stream synchronised, on entry).  Outputs are host arrays:
 * the keypoints feed PLVS's host-side tracking. */
int plvs_hip_orb_extract_dev(plvs_orb* h, const uint8_t* d_image, int w, int hh, int stride,
                             int lap0, int lap1, plvs_keypoint* kps, uint8_t* desc, int cap, int* n,
                             int* mono_index);
/* Wall-clock split of the last call, ms: [0] pyramid+FAST+cells, [1] host
 * quadtree, [2] orientation, [3] cos/sin + descriptors, [4] packing. */
int plvs_hip_orb_last_stage_ms(plvs_orb* h, double* ms, int cap);
/* Parity / debug accessors for the last call: pyramid level (mvImagePyramid /
 * mvImagePyramidFiltered), FAST candidates per level (x, y, response relative
 * to the detection region, the vToDistributeKeys of ORBextractor.cc:879). */
int plvs_hip_orb_level_size(plvs_orb* h, int level, int* w, int* hh);
int plvs_hip_orb_download_level(plvs_orb* h, int level, int blurred, uint8_t* out);
int plvs_hip_orb_last_candidates(plvs_orb* h, int level, float* xyr, int cap, int* n);

/* ------------------------------------------------- ORB search by projection
 * ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, th,
 * bFarPoints, thFarPoints) (src/ORBmatcher.cc:71-244) for monocular / RGB-D frames
 * (F.Nleft == -1), called from Tracking::SearchLocalPoints.  The views list what the
 * function reads of F and of the map points; all arrays are host memory. */
typedef struct plvs_frame_view {
  int32_t n;                    /* F.N                                               */
  const float* x;               /* F.mvKeysUn[i].pt.x / .y                           */
  const float* y;
  const int32_t* octave;        /* F.mvKeysUn[i].octave                              */
  const float* u_right;         /* F.mvuRight[i] (<= 0: none)                        */
  const uint8_t* desc;          /* F.mDescriptors, n x 32                            */
  float min_x, min_y;           /* F.mnMinX, F.mnMinY                                */
  float grid_w_inv, grid_h_inv; /* F.mfGridElementWidthInv / HeightInv (64 x 48)     */
  const float* scale_factors;   /* F.mvScaleFactors                                  */
} plvs_frame_view;

typedef struct plvs_mappoint_view {
  int32_t m;
  const uint8_t* track_in_view; /* pMP->mbTrackInView                                */
  const uint8_t* bad;           /* pMP->isBad()                                      */
  const float* proj_x;          /* pMP->mTrackProjX / Y / XR                         */
  const float* proj_y;
  const float* proj_xr;
  const float* view_cos;        /* pMP->mTrackViewCos                                */
  const float* track_depth;     /* pMP->mTrackDepth                                  */
  const int32_t* level;         /* pMP->mnTrackScaleLevel                            */
  const uint8_t* desc;          /* pMP->GetDescriptor(), m x 32                      */
  const uint8_t* has_obs;       /* pMP->Observations() > 0; NULL = all               */
} plvs_mappoint_view;

/* occupied[i] != 0: keypoint i already holds a map point with observations (NULL = none).
 * assigned[i] (n entries, out): index of the map point given to keypoint i, or -1
 * (the caller stores F.mvpMapPoints[i] = vpMapPoints[assigned[i]]).  *nmatches = the
 * reference's return value. */
int plvs_hip_orb_search_by_projection(const plvs_frame_view* F, const plvs_mappoint_view* M, float th,
                                      int far_points, float th_far, float nn_ratio,
                                      const uint8_t* occupied, int32_t* assigned, int* nmatches);
/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)
 * (src/ORBmatcher.cc:1774-1993), single-camera frames; called by Tracking::TrackWithMotionModel.
 * The projection of the last frame's map points (Tcw * x3Dw, mpCamera->project) is the
 * caller's: u, v, 1/z per last-frame keypoint. */
typedef struct plvs_lastframe_view {
  int32_t n;                    /* LastFrame.N                                        */
  const uint8_t* valid;         /* LastFrame.mvpMapPoints[i] && !mvbOutlier[i]        */
  const float* u;               /* uv(0), uv(1) of the map point in the current frame  */
  const float* v;
  const float* invz;            /* 1.0 / x3Dc(2)                                      */
  const int32_t* octave;        /* LastFrame.mvKeys[i].octave                         */
  const float* angle;           /* LastFrame.mvKeysUn[i].angle (degrees)              */
  const uint8_t* desc;          /* pMP->GetDescriptor(), n x 32                       */
  const uint8_t* has_obs;       /* pMP->Observations() > 0; NULL = all                */
} plvs_lastframe_view;
/* cur_angle[i] = CurrentFrame.mvKeysUn[i].angle; max_x / max_y = mnMaxX / mnMaxY; mbf =
 * CurrentFrame.mbf; forward / backward = bForward / bBackward (:1795-1796).  assigned[i2]
 * (out) = index of the last-frame keypoint whose map point goes to current keypoint i2. */
int plvs_hip_orb_search_by_projection_ff(const plvs_frame_view* F, const float* cur_angle, float max_x,
                                         float max_y, float mbf, const plvs_lastframe_view* L, float th,
                                         int forward, int backward, int check_orientation,
                                         const uint8_t* occupied, int32_t* assigned, int* nmatches);
/* ORBmatcher::SearchByBoW(KeyFramePtr& pKF, Frame& F, vpMapPointMatches)
 * (src/ORBmatcher.cc:300-507), single-camera frames; called by
 * Tracking::TrackReferenceKeyFrame and Relocalization.  A DBoW2::FeatureVector
 * (std::map<NodeId, std::vector<unsigned>>) is handed over as its in-order
 * traversal: node ids ascending, offsets (nnodes + 1) into the concatenated
 * feature-index lists. */
typedef struct plvs_featvec_view {
  int32_t nnodes;
  const uint32_t* node_id;
  const int32_t* offset;
  const uint32_t* index;
} plvs_featvec_view;
/* kf_valid[i] = vpMapPointsKF[i] && !isBad(); kf_angle = pKF->mvKeysUn[i].angle,
 * f_angle = F.mvKeys[i].angle (degrees; only read when check_orientation);
 * nn_ratio = mfNNratio.  assigned[iF] (out, f_n) = key-frame keypoint whose map
 * point goes to frame keypoint iF, or -1; *nmatches = the return value. */
int plvs_hip_orb_search_by_bow(const plvs_featvec_view* kf_vec, const uint8_t* kf_desc, int kf_n,
                               const uint8_t* kf_valid, const float* kf_angle,
                               const plvs_featvec_view* f_vec, const uint8_t* f_desc, int f_n,
                               const float* f_angle, float nn_ratio, int check_orientation,
                               int32_t* assigned, int* nmatches);
/* The batched primitive under it: dist[p] = DescriptorDistance(query[pair_q[p]],
 * train[pair_t[p]]) (src/ORBmatcher.cc:2198-2225) for arbitrary candidate lists. */
int plvs_hip_hamming_pairs(const uint8_t* query, int nq, const uint8_t* train, int nt,
                           const int32_t* pair_q, const int32_t* pair_t, int npairs, int32_t* dist);

/* ----------------------------------------------------------- Line extraction
 * Replaces LineExtractor (include/LineExtractor.h:48-83, src/LineExtractor.cc):
 *   LineExtractor::LineExtractor(numLinefeatures, LSDOptions&)                :104-145
 *   void LineExtractor::operator()(image, keylines, descriptors)             :150, 199-289
 * called from Frame::ExtractLSD (src/Frame.cc:815-837), for the default
 * configuration (Line.LSD.on = 0: EDLines detector of
 * Thirdparty/line_descriptor/src/binary_descriptor_custom.cpp, LBD descriptor,
 * Line.pyramidPrecomputation = 0).  The per-pixel maps and the LBD descriptors
 * are computed on the device; anchor linking, line fitting / validation and the
 * octave grouping are sequential and run on host threads inside the library.
 *
 * plvs_keyline is cv::line_descriptor_c::KeyLine field for field
 * (descriptor_custom.hpp:104-172), 68 bytes.
 */
typedef struct plvs_keyline {
  float angle;
  int32_t class_id;
  int32_t octave;
  float pt_x, pt_y;
  float response;
  float size;
  float startPointX, startPointY, endPointX, endPointY;
  float sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
  float lineLength;
  int32_t numOfPixels;
} plvs_keyline;

typedef struct plvs_lines plvs_lines;

/* nfeatures = Line.nfeatures (0 keeps all), nlevels = Line.nLevels, scale_factor =
 * Line.scaleFactor, min_line_length = Line.minLineLength (relative to the image
 * size), line_fit_err_threshold = LSDOptions::lineFitErrThreshold (1.6). */
int plvs_hip_lines_create(int nfeatures, int nlevels, float scale_factor, double min_line_length,
                          double line_fit_err_threshold, plvs_lines** out);
int plvs_hip_lines_destroy(plvs_lines* h);
/* keylines / desc hold `cap` entries (desc 32 bytes each); *n = number of lines
 * (nothing is written if n > cap).  No lines -> *n = 0 (the reference prints
 * "no lines!" and returns).  Empty image -> PLVS_ERR_EMPTY. */
int plvs_hip_lines_extract(plvs_lines* h, const uint8_t* image, int w, int hh, int stride,
                           plvs_keyline* keylines, uint8_t* desc, int cap, int* n);
int plvs_hip_lines_extract_dev(plvs_lines* h, const uint8_t* d_image, int w, int hh, int stride,
                               plvs_keyline* keylines, uint8_t* desc, int cap, int* n);
/* LineExtractor::SetGaussianPyramid (include/LineExtractor.h:83, src/LineExtractor.cc:170) as used by
 * Frame::PrecomputeGaussianPyramid (src/Frame.cc:841-865) when Line.pyramidPrecomputation is on
 * (USE_UNFILTERED_PYRAMID_FOR_LINES 1): octave i of the line detector is level i of `orb`'s
 * unblurred pyramid — read in place on the device — instead of the detector's own blur -> resize
 * chain; the number of octaves becomes min(own, orb levels) and the scale factor the extractor's
 * (BinaryDescriptor::setGaussianPyramid, binary_descriptor_custom.cpp:1491-1533, border 0).
 * `orb` must have processed the same image before plvs_hip_lines_extract[_dev] is called
 * (plvs_hip_frame_extract_dev orders the two itself).  orb = NULL restores the own pyramid. */
int plvs_hip_lines_set_gaussian_pyramid(plvs_lines* h, plvs_orb* orb);
/* ms of the last call: [0] device maps + D2H, [1] host linking/fitting/grouping, [2] LBD;
 * split of [1]: [3] EdgeDrawing of octave 0, [4] tail of the (overlapped) line fitting,
 * [5] grouping + selection. */
int plvs_hip_lines_last_stage_ms(plvs_lines* h, double* ms, int cap);
/* Parity accessors of the last call: which = 0 blurred octave image (u8), 1 dx,
 * 2 dy (s16), 3 packed map (u16: bits 0-8 gradient/4, bit 15 = |dx| < |dy|). */
int plvs_hip_lines_octave_size(plvs_lines* h, int octave, int* w, int* hh);
int plvs_hip_lines_download_map(plvs_lines* h, int octave, int which, void* out);
int plvs_hip_lines_num_in_octave(plvs_lines* h, int octave);

/* ---------------------------------------------------------------------------
 * The LSD detector (Line.LSD.on: 1 -> LineExtractor::skUseLsdExtractor, src/LineExtractor.cc:203-206, 275-279;
 * src/Tracking.cc reads the flag from the settings file).  Replaces
 *   cv::lsd::LineSegmentDetectorImpl::detect   Thirdparty/line_descriptor/src/lsd_custom.cpp:433-1081
 *   LSDDetectorC::detect / detectImpl           Thirdparty/line_descriptor/src/LSDDetector_custom.cpp:50-298
 * and, for plvs_hip_lsd_extract, LineExtractor::operator() with that detector in front of the LBD descriptor.
 * Device: the Gaussian + INTER_LINEAR_EXACT rescaling in front of the detector, the level-line field (gradient norm in
 * double, angle by cv::fastAtan2), the pyramid, Sobel and LBD.  Host threads inside the library: the pseudo-ordering
 * (the reference's unstable std::sort, reproduced by running it) and the sequential region growing / rectangle
 * refinement / NFA loop.  Results are those of the reference's sources bit for bit (tests/test_lsd.py).
 */
typedef struct plvs_lsd plvs_lsd;
typedef struct plvs_lsd_options {   /* LSDDetectorC::LSDOptions (descriptor_custom.hpp:928-957) */
  int refine;                       /* cv::LSD_REFINE_NONE / STD / ADV = 0 / 1 / 2; default 2      */
  double scale, sigma_scale, quant, ang_th, log_eps, density_th;   /* 0.8, 0.6, 2.0, 22.5, 0, 0.7 */
  int n_bins;                       /* 1024                                                      */
} plvs_lsd_options;
int plvs_hip_lsd_default_options(plvs_lsd_options* out);
int plvs_hip_lsd_create(plvs_lsd** out);
int plvs_hip_lsd_destroy(plvs_lsd* h);
/* LineSegmentDetector::detect(image, lines): segments = cap x 4 floats (x1, y1, x2, y2 — cv::Vec4f), in the
 * reference's order; *n = their number (nothing is written if n > cap).  options = NULL: the defaults. */
int plvs_hip_lsd_segments(plvs_lsd* h, const uint8_t* image, int w, int hh, int stride, const plvs_lsd_options* options,
                          float* segments, int cap, int* n);
/* LSDDetectorC::detect(image, keylines, scale, numOctaves, opts): the detector's own pyramid, LSD on every level, KeyLines
 * of the segments longer than min_length x the image diagonal (opts.min_length).  The pyramid as the reference's loop
 * leaves it (LSDDetector_custom.cpp:56-83: it blurs in place an image it has already stored): every level but the last
 * is detected on its 5 x 5, sigma 1 blurred image, level k + 1 = INTER_LINEAR resize by 1.f / scale of that blurred
 * level, the last level is not blurred. */
int plvs_hip_lsd_detect(plvs_lsd* h, const uint8_t* image, int w, int hh, int stride, int num_octaves, float scale,
                        const plvs_lsd_options* options, double min_length, plvs_keyline* keylines, int cap, int* n);
/* LineExtractor::operator() with the LSD detector, options as Tracking fills them (src/Tracking.cc:1458-1485): num_octaves =
 * Line.nLevels; options->scale = Line.scaleFactor — the pyramid scale (narrowed to float, as detectLineFeatures passes it) AND
 * the detector's own rescaling; the other members = the Line.LSD.* keys.  Detection, detectLineFeatures' selection of the
 * nfeatures strongest, LBD descriptors on BinaryDescriptor's own (unblurred) pyramid.  Outputs as plvs_hip_lines_extract. */
int plvs_hip_lsd_extract(plvs_lsd* h, const uint8_t* image, int w, int hh, int stride, int nfeatures, int num_octaves,
                         const plvs_lsd_options* options, double min_length, plvs_keyline* keylines, uint8_t* desc, int cap,
                         int* n);
/* ... with the image in device memory (stride in bytes; the work that produced it must be complete), as plvs_hip_lines_extract_dev */
int plvs_hip_lsd_extract_dev(plvs_lsd* h, const uint8_t* d_image, int w, int hh, int stride, int nfeatures, int num_octaves,
                             const plvs_lsd_options* options, double min_length, plvs_keyline* keylines, uint8_t* desc, int cap,
                             int* n);
/* ms of the last call: [0] device maps + copies, [1] host ordering + region loop (summed over levels), [2] selection + LBD */
int plvs_hip_lsd_last_stage_ms(plvs_lsd* h, double* ms, int cap);

/* LineMatcher::SearchByKnn(Frame& CurrentFrame, const Frame& LastFrame)
 * (src/LineMatcher.cc:303-447), single-camera frames.  Last frame = query side:
 * LBD descriptors (n_last x 32), valid_last[i] = LastFrame.mvpMapLines[i] &&
 * !LastFrame.mvbLineOutlier[i], angle_last[i] = mvKeyLinesUn[i].angle (radians); current
 * frame = train side.  assigned[t] (n_cur entries, out) = index of the last-frame line
 * whose map line goes to current line t (CurrentFrame.mvpMapLines[t] =
 * LastFrame.mvpMapLines[assigned[t]]) or -1; *nmatches = the reference's return value.
 * nn_ratio / check_orientation are the LineMatcher constructor arguments. */
int plvs_hip_lines_search_by_knn(const uint8_t* desc_last, int n_last, const uint8_t* valid_last,
                                 const float* angle_last, const uint8_t* desc_cur, int n_cur,
                                 const float* angle_cur, float nn_ratio, int check_orientation,
                                 int32_t* assigned, int* nmatches);

/* LineMatcher::SearchByKnn(KeyFramePtr& pKF, const Frame& F, vpMapLineMatches)
 * (src/LineMatcher.cc:156-301), called by Tracking::TrackReferenceKeyFrame: key frame =
 * query side (valid_kf[i] = vpMapLinesKF[i] && !isBad()), frame = train side; a match needs
 * distance <= TH_LOW (60).  assigned[t] (n_f entries) = key-frame line whose map line goes
 * to frame line t, or -1. */
int plvs_hip_lines_search_by_knn_kf(const uint8_t* desc_kf, int n_kf, const uint8_t* valid_kf,
                                    const float* angle_kf, const uint8_t* desc_f, int n_f,
                                    const float* angle_f, float nn_ratio, int check_orientation,
                                    int32_t* assigned, int* nmatches);
/* LineMatcher::SearchStereoMatchesByKnn(frame, vMatches, vValidMatches, descriptorDist)
 * (src/LineMatcher.cc:454-586): left lines = query, right lines = train; angle / octave of
 * mvKeyLinesUn / mvKeyLinesRightUn.  Outputs = vMatches (queryIdx, trainIdx, distance) and
 * vValidMatches in the reference's order, *n_out entries (cap >= n_right always suffices);
 * *nmatches = the return value (valid entries). */
int plvs_hip_lines_search_stereo_by_knn(const uint8_t* desc_left, int n_left, const float* angle_left,
                                        const int32_t* octave_left, const uint8_t* desc_right, int n_right,
                                        const float* angle_right, const int32_t* octave_right, float nn_ratio,
                                        int check_orientation, int descriptor_dist, int32_t* match_query,
                                        int32_t* match_train, float* match_distance, uint8_t* match_valid,
                                        int cap, int* n_out, int* nmatches);

/* LineMatcher::SearchByProjection — the guided line searches Tracking runs on every frame
 * (single-camera frames: NlinesLeft == -1, no mpCamera2).  The view lists what they read of the frame;
 * the projection of the map lines into it (LineProjection::ProjectLineWithCheck, MapLine::mTrackProj*)
 * is the caller's, as in the point searches.  All arrays are host memory. */
typedef struct plvs_line_frame_view {
  int32_t n;                          /* F.Nlines                                                  */
  const plvs_keyline* keylines_un;    /* F.mvKeyLinesUn                                            */
  const uint8_t* descriptors;         /* F.mLineDescriptors, n x 32                                */
  const float* u_right_start;         /* F.mvuRightLineStart / End (< 0: none); NULL = empty       */
  const float* u_right_end;
  float bf;                           /* F.mbf                                                     */
  int32_t n_levels;                   /* entries of the two tables below                           */
  const float* line_scale_factors;    /* F.mvLineScaleFactors                                      */
  const float* line_inv_level_sigma2; /* F.mvLineInvLevelSigma2                                    */
  float max_diag;                     /* Frame::mnMaxDiag (the (theta, d) grid spans [-diag, diag]) */
} plvs_line_frame_view;
/* LineMatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, bLargerSearch, bMono)
 * (src/LineMatcher.cc:837-1230), called by Tracking::TrackWithMotionModel (src/Tracking.cc:3653).
 * Per last-frame line i: valid[i] = mvpMapLines[i] && !mvbLineOutlier[i] && ProjectLineWithCheck
 * succeeded; proj[6 i ..] = uS, vS, uE, vE, invSz, invEz of the projection; octave[i] / angle[i] of
 * LastFrame.mvKeyLinesUn[i]; desc = pML->GetDescriptor() (n_last x 32); has_obs[i] =
 * pML->Observations() > 0 (NULL = all).  occupied[i2] != 0: current line i2 already holds a map line
 * with observations (NULL = none).  direction: 0, 1 = bForward, 2 = bBackward (:860-861).
 * assigned[i2] (out, F->n entries) = last-frame line whose map line goes to current line i2, or -1;
 * *nmatches = the reference's return value. */
int plvs_hip_lines_search_by_projection_ff(const plvs_line_frame_view* F, const uint8_t* occupied, int n_last,
                                           const uint8_t* valid, const float* proj, const int32_t* octave,
                                           const float* angle, const uint8_t* desc, const uint8_t* has_obs,
                                           int larger_search, int direction, float nn_ratio, int check_orientation,
                                           int32_t* assigned, int* nmatches);
/* LineMatcher::SearchByProjection(Frame& F, const std::vector<MapLinePtr>&, bLargerSearch)
 * (src/LineMatcher.cc:1286-1560, left image), called by Tracking::SearchLocalLines (:4576).  Per map
 * line m: in_view[m] = mbTrackInView && !isBad(); proj[6 m ..] = mTrackProjStartX, StartY, EndX, EndY,
 * mTrackStartDepth, mTrackEndDepth (the depths themselves: the stereo gate shifts the end points by mbf / depth as
 * :1419-1423 does — not by mbf * (1 / depth), which rounds differently); level[m] = mnTrackScaleLevel. */
int plvs_hip_lines_search_by_projection(const plvs_line_frame_view* F, const uint8_t* occupied, int n_map,
                                        const uint8_t* in_view, const float* proj, const int32_t* level,
                                        const uint8_t* desc, const uint8_t* has_obs, int larger_search,
                                        float nn_ratio, int32_t* assigned, int* nmatches);

/* ------------------------------------------------------------ Frame extraction
 * Points and lines of one image, extracted concurrently on two host threads
 * (each extractor drives its own stream), as Frame::Frame does with threadLeft /
 * threadLines (src/Frame.cc:503-508).  Arguments as in plvs_hip_orb_extract_dev
 * and plvs_hip_lines_extract_dev; the ORB status is returned first. */
int plvs_hip_frame_extract_dev(plvs_orb* orb, plvs_lines* lines, const uint8_t* d_image, int w,
                               int hh, int stride, int lap0, int lap1, plvs_keypoint* kps,
                               uint8_t* desc, int kp_cap, int* n_kp, int* mono_index,
                               plvs_keyline* keylines, uint8_t* line_desc, int line_cap,
                               int* n_lines);
/* The same with a hook: after_points(user, orb_status) runs on the calling thread as soon as the points are out
 * (kps / desc / *n_kp / *mono_index written) while the line thread is still at work — the place for what Tracking does
 * with the points alone (ORBmatcher::SearchByProjection of TrackWithMotionModel, src/Tracking.cc) when the caller
 * wants it off the frame's critical path; the reference joins both threads first (src/Frame.cc:507-508) and gets the
 * same results later. */
int plvs_hip_frame_extract_dev_hook(plvs_orb* orb, plvs_lines* lines, const uint8_t* d_image, int w,
                                    int hh, int stride, int lap0, int lap1, plvs_keypoint* kps,
                                    uint8_t* desc, int kp_cap, int* n_kp, int* mono_index,
                                    plvs_keyline* keylines, uint8_t* line_desc, int line_cap,
                                    int* n_lines, void (*after_points)(void* user, int orb_status), void* user);

/* ------------------------------------------------------------ Frame glue
 * What Frame::Frame runs between ExtractORB / ExtractLSD and the first search (src/Frame.cc:541-580): host flavours — the
 * reference's Frame holds host vectors — with the arithmetic on the device.  K4 = fx, fy, cx, cy (mpCamera->toK(), which
 * for a pinhole camera is also toLinearK()); dist = mDistCoef: ndist = 4, 5 or 8 coefficients k1 k2 p1 p2 [k3 [k4 k5 k6]]
 * (ndist = 0 or dist[0] == 0: no distortion, as :1510 / :1560 / :1752 test it).
 *   _undistort_keypoints   Frame::UndistortKeyPoints :1507-1552 (cv::undistortPoints, five iterations in double)
 *   _compute_image_bounds  Frame::ComputeImageBounds :1749-1778 -> bounds5 = mnMinX, mnMaxX, mnMinY, mnMaxY, mnMaxDiag
 *   _undistort_keylines    Frame::UndistortKeyLines :1555-1700 (single pinhole camera): end points undistorted, the angle
 *                          from them (cv::fastAtan2 * DEG2RAD), lines outside bounds4 = mnMinX, mnMaxX, mnMinY, mnMaxY
 *                          dropped; kept_index[j] = the input line behind output j (the caller compacts mvKeyLines and
 *                          mLineDescriptors with it, :1649-1655)
 *   _assign_features_to_grid  Frame::AssignFeaturesToGrid :716-746, the 64 x 48 key-point grid as a CSR: cell = column * 48
 *                          + row (mGrid[ix][iy]), cell_start[3073], members in key-point order (push_back); the (theta, d)
 *                          line grid stays inside the line searches.
 *   _undistort_points      cv::undistortPoints(src, dst, K, dist, noArray(), K) on n points (x, y interleaved). */
int plvs_hip_frame_undistort_points(const float* xy, int n, const float* K4, const float* dist, int ndist, float* xy_out);
int plvs_hip_frame_undistort_keypoints(const plvs_keypoint* kps, int n, const float* K4, const float* dist, int ndist,
                                       plvs_keypoint* kps_un);
int plvs_hip_frame_compute_image_bounds(int width, int height, const float* K4, const float* dist, int ndist, float* bounds5);
int plvs_hip_frame_undistort_keylines(const plvs_keyline* keylines, int n, const float* K4, const float* dist, int ndist,
                                      const float* bounds4, plvs_keyline* keylines_un, int32_t* kept_index, int* n_kept);
int plvs_hip_frame_assign_features_to_grid(const plvs_keypoint* kps_un, int n, float min_x, float min_y, float grid_w_inv,
                                           float grid_h_inv, int32_t* cell_start, int32_t* cell_items, int* n_items);

/* Device self-test backing the TSDF chain kernel: counts the binary32 significands b =
 * 1.m * 2^exponent for which the kernel's reciprocal (v_rcp_f32 + one Newton step) differs
 * from the correctly rounded 1/b.  Expected: 0 for every exponent the kernel admits. */
int plvs_hip_selftest_rcp(int exponent, uint32_t* mismatches);
/* The walk's square root and quotient (normal-range forms without the compiler's range scaffolding) against
 * sqrtf and `/` on the device, 2.7e8 pseudo-random operand pairs per call: mismatches[0] sqrt, [1] division. */
int plvs_hip_selftest_walk_math(uint32_t seed, uint32_t* mismatches_sqrt_div);
/* The device-wide stable radix sort behind the run / record orderings of both TSDF back ends (two scatter paths: wide
 * digits for short arrays, 8-bit digits reordered in LDS from 2^20 pairs on): n pseudo-random keys below 2^bit_hi with
 * their positions as values (32-bit, or 64-bit with wide_values), sorted on bits [bit_lo, bit_hi).  mismatches2[0] =
 * neighbours out of order or equal keys whose positions are not ascending (stability), [1] = pairs torn apart. */
int plvs_hip_selftest_radix_sort(uint32_t n, int bit_lo, int bit_hi, int wide_values, uint32_t seed, uint32_t* mismatches2);

/* --------------------------------------------------------- TSDF (open_chisel)
 * Chunked (16^3) spatially hashed TSDF with per-point ray integration.
 *
 * Replaces:
 *   PointCloudMapChisel::InsertCloud            src/PointCloudMapChisel.cc:76-98
 *    -> ChiselServer::SetPointCloud             Thirdparty/chisel_server/src/ChiselServer.cpp:561
 *    -> ChiselServer::IntegrateLastPointCloud   ChiselServer.cpp:664
 *    -> chisel::Chisel::IntegratePointCloudWidthDepth (point-cloud part)
 *                                               Thirdparty/open_chisel/src/Chisel.cpp:442-585
 *   with chisel::Raycast                        open_chisel/src/geometry/Raycast.cpp:65-182
 *        DistVoxel::Integrate / ColorVoxel::IntegrateSimple
 *                                               include/open_chisel/DistVoxel.h:91, ColorVoxel.h:91
 *
 * Voxel payload (logical 16 B): {f32 sdf (init 99999), f32 weight (0 =
 * unknown), u32 kfid, u8 r,g,b,colour-weight}.  Chunks are 16x16x16 voxels,
 * linear id (z*16+y)*16+x (Chunk.h:90-93).
 */
typedef struct plvs_tsdf_chisel_params {
  float resolution;      /* voxel edge in metres (PointCloudMapping.resolution)            */
  float trunc_quad;      /* QuadraticTruncator coefficients, ChiselServer.cpp:56-59:       */
  float trunc_linear;    /*   0.0019, -0.00152, 0.001504, scale 6.0                         */
  float trunc_const;
  float trunc_scale;
  float weight;          /* ConstantWeighter weight (1), ChiselServer.cpp:60               */
  int32_t max_chunks;    /* capacity of the device chunk pool (64 KiB of HBM each)         */
  int32_t shard_rank;    /* multi-GPU sharding: this instance only owns chunks with        */
  int32_t shard_count;   /*   (ChunkHasher(id) mod shard_count) == shard_rank; 0/1 = all   */
  int32_t order_free;    /* 0 (default): every voxel update applied in the reference's order,
                          *   results bit-identical to the CPU loop.  1: the visits of a call
                          *   are summed per voxel and applied in one update — sdf / weight
                          *   within float rounding of the reference (tolerance in
                          *   tests/test_tsdf_chisel.py), kfid and colour still exact.       */
} plvs_tsdf_chisel_params;

typedef struct plvs_tsdf_chisel plvs_tsdf_chisel;

/* Fills *p with the constants PLVS uses (ChiselServer.cpp:50-69,
 * PointCloudMapChisel.cc:46-61) for the given resolution. */
int plvs_hip_tsdf_chisel_default_params(float resolution, plvs_tsdf_chisel_params* p);
int plvs_hip_tsdf_chisel_create(const plvs_tsdf_chisel_params* p, plvs_tsdf_chisel** out);
int plvs_hip_tsdf_chisel_destroy(plvs_tsdf_chisel* h);
/* Drops every chunk (PointCloudMap::Clear). */
int plvs_hip_tsdf_chisel_clear(plvs_tsdf_chisel* h);

/* Integrate one camera-frame cloud.  Twc is the 3x4 row-major camera pose
 * [R|t] (Sophus::SE3f Twc of InsertCloud).  xyz: n x 3 f32 camera-frame
 * points in cloud order; rgb: n x 3 u8, the r,g,b members of the pcl point;
 * kfid: n u32.  Points with z < 0.01 are skipped (Chisel.cpp:475).  Host
 * pointers; synchronous. */
int plvs_hip_tsdf_chisel_integrate(plvs_tsdf_chisel* h, const float* xyz, const uint8_t* rgb,
                                   const uint32_t* kfid, int n, const float* Twc);

/* Queued integration — the drop-in shape of PointCloudMapChisel.  PLVS hands over ONE key frame per InsertCloud and reads
 * the map only in UpdateMap, after at most five of them (src/PointCloudMapping.cc:540-552, 594-598).  _queue uploads the
 * cloud (host pointers, as _integrate) and returns without integrating; _flush integrates everything queued in ONE call of
 * the batch pipeline: the same map as integrating the clouds one by one — bit for bit in the ordered mode (the batch applies
 * every update in point order across the clouds), within the stated tolerance in the order-free mode — at a fraction of the
 * per-call launch chain.  Every entry point that reads or changes the map (integrate, carve, mesh, download, chunk lists,
 * deform, last_stats, ...) flushes first — so a READER can return an integrate error (capacity, a non-finite point): the
 * message then names the flush and the number of queued clouds, none of which was integrated; they are dropped (queue them
 * again after clearing / enlarging the map).  _clear drops the queue with the map.  _queued: clouds waiting. */
int plvs_hip_tsdf_chisel_queue(plvs_tsdf_chisel* h, const float* xyz, const uint8_t* rgb, const uint32_t* kfid, int n,
                               const float* Twc);
int plvs_hip_tsdf_chisel_flush(plvs_tsdf_chisel* h);
int plvs_hip_tsdf_chisel_queued(plvs_tsdf_chisel* h, int* nclouds);
/* Batched, device-resident form: `nclouds` clouds integrated in order, cloud c
 * being points [offsets[c], offsets[c+1]) of the concatenated device arrays,
 * with pose d_Twc[12*c .. 12*c+11].  `offsets` is a host array of nclouds+1
 * ints.  Result is identical to nclouds successive calls of the function
 * above.  Asynchronous on `stream` except for one device->host read of the
 * visit count. */
int plvs_hip_tsdf_chisel_integrate_batch_dev(plvs_tsdf_chisel* h, const float* d_xyz,
                                             const uint8_t* d_rgb, const uint32_t* d_kfid,
                                             const int32_t* offsets, int nclouds,
                                             const float* d_Twc, void* stream);

/* Depth images straight into the map: PointCloudMapping::GeneratePointCloudInCameraFrameBGRA's cloud
 * (src/PointCloudMapping.cc:957-996: p = (gx d, gy d, d) for every pixel (m, n) of the stride-`step` grid with
 * min_depth < d < max_depth, in raster order, coloured r, g, b = bytes 0, 1, 2 of the pixel) integrated as
 * PointCloudMapChisel::InsertCloud would (src/PointCloudMapChisel.cc:100-133 -> Chisel.cpp:442-585) — for `nclouds`
 * images with poses d_Twc[12 c ..], in order — WITHOUT the cloud ever being written to memory.  The map is, bit for bit,
 * the map of plvs_hip_cloudgen_generate_dev + plvs_hip_tsdf_chisel_integrate_batch_dev on the same images, in both modes
 * (tests/test_tsdf_chisel_depth.py).  On an order-free handle the walk's tiles are 32 x 16 blocks of grid pixels instead of
 * 512 consecutive points (neighbouring rays share their voxels: fewer table entries, records and segments per tile),
 * the point order the colours and key-frame ids need is recovered from (image, pixel).  Everything is DEVICE memory:
 *   d_depth        image c at d_depth + c * depth_image_stride (floats), rows depth_pitch floats apart
 *   d_bgr          image c at d_bgr + c * bgr_image_stride (bytes), rows bgr_pitch bytes apart, 3 bytes per pixel
 *   d_grid_points  matCamGridPoints_ (plvs_hip_cloudgen_grid_points), ceil(height / step) x ceil(width / step) x 2 floats
 *   d_kfid         one key-frame id per image (may be NULL: 0)
 * Asynchronous on `stream` except for the call's one read of its counters (an ordered, sharded or deforming handle builds
 * the reference's clouds in scratch memory first and waits once more, for their sizes).  plvs_tsdf_stats.points is 0 after a call on
 * an order-free handle (the cloud is never formed); ordered handles report the points of the clouds they built. */
typedef struct plvs_depth_batch {
  const float* d_depth;
  const uint8_t* d_bgr;
  size_t depth_image_stride, bgr_image_stride;
  int depth_pitch, bgr_pitch;
  int width, height, step;
  const float* d_grid_points;
  double min_depth, max_depth;
  const uint32_t* d_kfid;
} plvs_depth_batch;
int plvs_hip_tsdf_chisel_integrate_depth_batch_dev(plvs_tsdf_chisel* h, const plvs_depth_batch* in, int nclouds,
                                                   const float* d_Twc, void* stream);

/* A cloud WITH NORMALS, as PointCloudMapChisel::LoadMap feeds the saved map through
 * (src/PointCloudMapChisel.cc:527-546 -> ChiselServer::IntegrateWorldPointCloud, ChiselServer.cpp:587-615, Twc =
 * identity there -> Chisel::IntegrateWorldPointCloudWithNormals, Chisel.cpp:238-376): every point casts the
 * segment point -/+ 4 voxels along its normal (n x 3 f32, normalised here as there), u = (centre - point) . normal,
 * truncation 4 * resolution, no depth test, no carving, ColorVoxel::Integrate (not IntegrateSimple).  Runs the
 * ordered pipeline on any handle (bit-exact); the updated-chunk list is set as by the integrate calls.  Host
 * pointers and synchronous / device pointers and asynchronous on `stream` as above. */
int plvs_hip_tsdf_chisel_integrate_world_normals(plvs_tsdf_chisel* h, const float* xyz, const uint8_t* rgb,
                                                 const uint32_t* kfid, const float* normals, int n, const float* Twc);
int plvs_hip_tsdf_chisel_integrate_world_normals_dev(plvs_tsdf_chisel* h, const float* d_xyz, const uint8_t* d_rgb,
                                                     const uint32_t* d_kfid, const float* d_normals, int n,
                                                     const float* d_Twc, void* stream);

/* Counters of the last integrate call: voxel read-modify-write visits applied
 * (the "Mvoxels" unit of the metric), points consumed, chunks newly created,
 * chunks updated, distinct voxels updated, longest per-voxel update run. */
typedef struct plvs_tsdf_stats {
  int64_t visits;
  int64_t points;
  int32_t new_chunks;
  int32_t updated_chunks;
  int32_t voxels;    /* distinct voxels updated                                            */
  int32_t max_run;   /* most updates any single voxel took (its updates are inherently     */
                     /* sequential: the serial-latency floor of the call)                  */
} plvs_tsdf_stats;
int plvs_hip_tsdf_chisel_last_stats(plvs_tsdf_chisel* h, plvs_tsdf_stats* s);

/* Tuning knob of the order_free apply stage: a chunk that collects more than `min_segments` (tile, chunk) segments in
 * a call is applied in parts of `part_segments` segments (defaults 2048 / 256); results do not depend on it. */
int plvs_hip_tsdf_chisel_set_apply_parts(plvs_tsdf_chisel* h, int part_segments, int min_segments);

/* Optional per-stage timing with HIP events recorded on the caller's stream
 * (what bench.py uses for the live roofline figure).  Enabling resets the
 * accumulators.  stage_ms returns the milliseconds accumulated per pipeline
 * stage and the number of integrate calls they cover. */
int plvs_hip_tsdf_chisel_set_profiling(plvs_tsdf_chisel* h, int enable);
int plvs_hip_tsdf_chisel_stage_ms(plvs_tsdf_chisel* h, double* ms, int cap, int* nstages,
                                  int64_t* calls);
const char* plvs_hip_tsdf_chisel_stage_name(int i);
/* Stage names of the pipeline the handle last ran (they differ between the ordered and the order-free mode). */
const char* plvs_hip_tsdf_chisel_stage_name_of(plvs_tsdf_chisel* h, int i);

int plvs_hip_tsdf_chisel_num_chunks(plvs_tsdf_chisel* h, int* n);
/* Chunk ids (x,y,z int32 triples) of all chunks, in pool-slot order. */
int plvs_hip_tsdf_chisel_chunk_ids(plvs_tsdf_chisel* h, int32_t* ids_xyz, int cap, int* n);
/* Chunk ids updated by the last integrate call (the "updated block list"
 * that is all-gathered across GPUs; also the mesh-dirty seed set of
 * Chisel.cpp:553-572). */
int plvs_hip_tsdf_chisel_updated_chunk_ids(plvs_tsdf_chisel* h, int32_t* ids_xyz, int cap, int* n);
/* Same list written to a device buffer (cap triples), asynchronous on `stream`:
 * the payload of the cross-GPU all-gather. */
int plvs_hip_tsdf_chisel_updated_chunk_ids_dev(plvs_tsdf_chisel* h, int32_t* d_ids_xyz, int cap,
                                               int* n, void* stream);
/* Download one chunk: 4096 entries each of sdf, weight, kfid and rgbw
 * (r | g<<8 | b<<16 | colour-weight<<24).  Returns PLVS_ERR_INVALID_ARG if
 * the chunk does not exist. */
int plvs_hip_tsdf_chisel_download_chunk(plvs_tsdf_chisel* h, int cx, int cy, int cz, float* sdf,
                                        float* weight, uint32_t* kfid, uint32_t* rgbw);

/* Depth-image carving, the first part of Chisel::IntegratePointCloudWidthDepth when the
 * integrator has carving enabled (Chisel.cpp:394-438; PointCloudMapping.useCarving, off in the
 * shipped settings): every existing chunk on the camera-frustum list goes through
 * ProjectionIntegrator::CarveWithDepth (ProjectionIntegrator.h:271-338) — known voxels more than
 * truncation + carving_dist in front of the measured depth, with sdf < 1e-5, are Reset().  Call it
 * before the integrate of the same keyframe, as the reference does.  depth: height rows of
 * width floats, NaN = no measurement; fx..cy: the depth camera's intrinsics; near / far:
 * ChiselServer's plane distances (0.05 / 5.0); Twc 3x4 row-major (host memory in both flavours);
 * carving_dist: 0.05 in PLVS.  *carved_chunks = "carved in N chunks"; afterwards
 * plvs_hip_tsdf_chisel_updated_chunk_ids[_dev] lists them (meshesToUpdate). */
int plvs_hip_tsdf_chisel_carve(plvs_tsdf_chisel* h, const float* depth, int width, int height, float fx,
                               float fy, float cx, float cy, float near_dist, float far_dist,
                               const float* Twc, float carving_dist, int* carved_chunks);
int plvs_hip_tsdf_chisel_carve_dev(plvs_tsdf_chisel* h, const float* d_depth, int width, int height,
                                   float fx, float fy, float cx, float cy, float near_dist,
                                   float far_dist, const float* Twc, float carving_dist, void* stream,
                                   int* carved_chunks);

/* ------------------------------------------- multi-GPU: the ray-sharded integrate (chisel, order_free = 1)
 * New design (the reference is one process): the MAP is sharded by chunk (shard_rank / shard_count, owner =
 * ChunkHasher(id) mod N, ChunkManager.h:42-54), the WORK by tile of the point stream — rank r walks the tiles
 * t = r (mod N) of 512 consecutive points of every call, through whatever chunks their rays cross, and sends
 * what it collected to the chunk owners.  Every rank is given the same clouds.  One call =
 *   shard_walk    this rank's tiles, then its own aggregation: one sum per touched voxel;
 *                 send_counts[3 * N] = {descriptors, voxel sums, colour-run records} per destination
 *   shard_pack    the three send buffers, each grouped by destination in rank order:
 *                 descriptors 32 B (one per chunk and slab of 512 voxels), sums 32 B, colour-run records 24 B:
 *                 {chunk key 8 B, voxel | tile << 12, six 16-bit spans (first ray | (length - 1) << 9, 0xFFFF =
 *                 none) of the tile's rays that saw the voxel}; a run with more than six spans takes further
 *                 records (round 4: 80 B with the 512-bit ray mask before; at most 2^20 tiles per call)
 *   all-to-all    counts, then the three buffers (the caller's transport; ..._integrate_sharded does it over RCCL)
 *   shard_apply   the three receive buffers grouped by source in rank order, recv_counts[3 * N]
 *   shard_saturated / all-gather / shard_note_saturated  (or _saturated_message / all-gather / _note_gathered)
 *                 the voxels whose colour weight reached 254 in the call: walkers stop sending their runs
 *                 (late knowledge only costs surplus runs).
 * The partial sums are integers: the union of the shards is bit-identical to the single-device order_free map
 * for every N (tests/test_shard_rays.py).  shard_count = 1 is allowed (the rank sends to itself).  Each rank
 * also keeps a walk directory (ids + 512 B of bits per chunk it has walked through, up to max_chunks x
 * shard_count chunks); the voxel data stay with the owner.  InsertCloud equivalent: Chisel.cpp:442-585. */
int plvs_hip_tsdf_chisel_shard_walk(plvs_tsdf_chisel* h, const float* d_xyz, const int32_t* offsets, int nclouds,
                                    const float* d_Twc, int64_t* send_counts, void* stream);
int plvs_hip_tsdf_chisel_shard_pack(plvs_tsdf_chisel* h, void* d_seg_dst, void* d_rec_dst, void* d_run_dst,
                                    void* stream);
int plvs_hip_tsdf_chisel_shard_apply(plvs_tsdf_chisel* h, const void* d_seg_src, const void* d_rec_src,
                                     const void* d_run_src, const int64_t* recv_counts, const uint8_t* d_rgb,
                                     const uint32_t* d_kfid, void* stream);
/* *n = voxels that saturated in the last shard_apply; written to d_voxels as {chunk x, y, z, voxel} int32
 * quadruples when cap >= *n (PLVS_ERR_CAPACITY otherwise: call once with cap 0 for the count). */
int plvs_hip_tsdf_chisel_shard_saturated(plvs_tsdf_chisel* h, int32_t* d_voxels, int cap, int* n, void* stream);
int plvs_hip_tsdf_chisel_shard_note_saturated(plvs_tsdf_chisel* h, const int32_t* d_voxels, int n, void* stream);
/* The same feedback without a host read (round 4): _saturated_message moves the last shard_apply's voxels to a waiting
 * list kept with the handle and writes this rank's fixed-size message — d_msg[0 .. k) = the first k <= rows waiting
 * voxels, d_msg[rows] = {k, 0, 0, 0} (int32 quadruples, rows + 1 of them); what does not fit waits for the next step.
 * _note_gathered notes the all-gathered messages of nranks ranks (nranks x (rows + 1) quadruples), every length read
 * on the device. */
int plvs_hip_tsdf_chisel_shard_saturated_message(plvs_tsdf_chisel* h, int32_t* d_msg, int rows, void* stream);
int plvs_hip_tsdf_chisel_shard_note_gathered(plvs_tsdf_chisel* h, const int32_t* d_gathered, int nranks, int rows,
                                             void* stream);
/* All of the above with the exchanges over RCCL (grouped ncclSend / ncclRecv, ncclAllGather); rccl_comm is the
 * caller's ncclComm_t, one process per GPU, its size and rank those of the map's shard_count / shard_rank. */
int plvs_hip_tsdf_chisel_integrate_sharded(plvs_tsdf_chisel* h, void* rccl_comm, const float* d_xyz,
                                           const uint8_t* d_rgb, const uint32_t* d_kfid, const int32_t* offsets,
                                           int nclouds, const float* d_Twc, void* stream);
/* Errors of the *_integrate_sharded calls.  Before a step's counts exchange a failing rank announces it (-1 counts): no rank
 * applies anything, the failing rank returns its error, the others PLVS_ERR_HALO, the maps are as before.  After it a rank
 * that cannot go through the step's collectives (exchange buffers cannot be allocated, RCCL itself fails) returns
 * PLVS_ERR_COMM_FATAL and LEAVES THE COMMUNICATOR ALONE — it is the caller's (e.g. torch's ProcessGroup): the caller must
 * abort or destroy it on every rank (the peers are inside collectives this rank will not enter; they leave them through
 * their own abort or RCCL's watchdog) and clear or rebuild the sharded map.  plvs_hip_exchange_abort_on_fatal(1) opts in to
 * ncclCommAbort on the failing rank's communicator before the call returns (process-wide; for callers that own the
 * communicator outright and never touch it again). */
int plvs_hip_exchange_abort_on_fatal(int enable);

/* ------------------------------------------- multi-GPU: the block-list exchange
 * The sharded TSDF path (shard_rank / shard_count of either back end; owner(block) =
 * three-prime hash(block id) mod N, ChunkManager.h:42-54 / block_hash.h:15-26) has ONE
 * exchange step: every rank contributes the ids of the blocks its last integrate call
 * updated (plvs_hip_tsdf_*_updated_*_ids_dev) and receives everybody's, over RCCL / xGMI
 * (two fixed-size ncclAllGather: counts, padded lists).  rccl_comm is the caller's
 * ncclComm_t (one process per GPU); RCCL is resolved at run time, the library has no
 * link-time dependency on it.  d_all_ids: [world * cap * 3], d_all_counts: [world]. */
int plvs_hip_tsdf_exchange_block_lists(void* rccl_comm, const int32_t* d_local_ids, int local_count, int cap,
                                       int32_t* d_all_ids, int32_t* d_all_counts, void* stream);
int plvs_hip_rccl_world_size(void* rccl_comm, int* world);
/* The global block directory a rank keeps from the gathered lists: block id -> owner rank
 * (which blocks exist on which GPU: what the host needs to schedule meshing and to route
 * queries).  merge is asynchronous on `stream`; count / list synchronise. */
typedef struct plvs_block_directory plvs_block_directory;
int plvs_hip_block_directory_create(int max_blocks, plvs_block_directory** out);
int plvs_hip_block_directory_destroy(plvs_block_directory* d);
int plvs_hip_block_directory_merge(plvs_block_directory* d, const int32_t* d_all_ids, const int32_t* d_counts, int world,
                                   int cap, void* stream);
int plvs_hip_block_directory_count(plvs_block_directory* d, int* n);
int plvs_hip_block_directory_list(plvs_block_directory* d, int32_t* ids_xyz, int32_t* owners, int cap, int* n);

/* ------------------------------------------------- sparse stereo matching (M5)
 * Replaces Frame::ComputeStereoMatches src/Frame.cc:1780-1975 (caller: the
 * stereo Frame constructor, after the two ExtractORB threads joined).  The two
 * extractors must have processed the current rectified pair: their device
 * pyramids (mvImagePyramid) are read in place.  keys / desc are the packed
 * outputs of plvs_hip_orb_extract (mvKeys / mDescriptors, mvKeysRight /
 * mDescriptorsRight); mb = baseline in metres, mbf = fx * baseline.
 * u_right / depth: n_left floats = mvuRight / mvDepth (-1 = no match), after
 * the 1.5 * 1.4 * median cut.  *n_matched (nullable) = keypoints with depth. */
typedef struct plvs_stereo plvs_stereo;
int plvs_hip_stereo_create(plvs_orb* left, plvs_orb* right, plvs_stereo** out);
int plvs_hip_stereo_destroy(plvs_stereo* s);
int plvs_hip_stereo_matches(plvs_stereo* s, const plvs_keypoint* keys_left, const uint8_t* desc_left,
                            int n_left, const plvs_keypoint* keys_right, const uint8_t* desc_right,
                            int n_right, float mb, float mbf, float* u_right, float* depth,
                            int* n_matched);

/* Surface extraction: ChunkManager::RecomputeMesh (Thirdparty/open_chisel/src/ChunkManager.cpp:116-170:
 * GenerateMesh with one kfid per cube, ColorizeMesh, ComputeNormalsFromGradients) for every chunk of
 * the list — what Chisel::UpdateMeshes (Chisel.cpp:57-65) does for meshesToUpdate (the 27-neighbourhood
 * of the updated chunks, :553-568) when PointCloudMapChisel::UpdateMap calls it
 * (src/PointCloudMapChisel.cc:236).  chunk_ids_xyz: host, nchunks x 3.  Outputs (host): the
 * chisel::Mesh arrays of the chunks back to back, in list order — vertices / normals / colors
 * (n x 3 f32, colours in [0, 1]) and kfids (n); chunk c owns [chunk_first[c], chunk_first[c+1])
 * (chunk_first: nchunks + 1 ints); a chunk that does not exist owns nothing.  *nvertices = n.  If
 * n > capacity nothing is written but chunk_first / *nvertices, and PLVS_ERR_CAPACITY is returned.
 * Sharded map (shard_count > 1): the list holds chunks of THIS rank (another rank's chunk owns nothing); the
 * cubes on a chunk's +x / +y / +z faces, the colour interpolation and the gradient normals read neighbour chunks
 * that live on other ranks.  A call that met a foreign chunk it does not hold returns PLVS_ERR_HALO; the caller
 * fetches the chunks plvs_hip_tsdf_chisel_halo_missing lists from their owners (halo_export there, halo_import
 * here) and calls again — until no rank misses anything (two or three rounds: new vertices can reach further
 * chunks).  The result is then the single-device mesh of the chunk, byte for byte. */
int plvs_hip_tsdf_chisel_mesh_chunks(plvs_tsdf_chisel* h, const int32_t* chunk_ids_xyz, int nchunks,
                                     float* vertices, float* normals, float* colors, uint32_t* kfids,
                                     int capacity, int32_t* chunk_first, int* nvertices);

/* ---- Chisel::Deform: PointCloudMapChisel::OnMapChange with key-frame adjustment (src/PointCloudMapChisel.cc:389-470
 * -> ChiselServer::Deform, ChiselServer.cpp:617-621 -> Chisel::Deform, Chisel.cpp:588-591 -> ChunkManager::Deform,
 * ChunkManager.cpp:918-1063).  Every known voxel whose kfid has an entry in the deformation map moves to
 * R * centre + t; the first voxel to land in a new voxel is copied, later ones are merged into it
 * (DistVoxel::Integrate, SetKfid, ColorVoxel::Integrate).  Who is first follows the reference's walk over `chunks`, a
 * std::unordered_map whose iteration order is a function of its insert / erase history: with deform enabled the map
 * keeps that container (ids only, host side) — every integrate call then reports the chunks it VISITS in first-visit
 * order (one more walk of the rays; one cloud per call), and upload_chunk / clear update it as CreateChunk / Reset do.
 * Enable on the EMPTY map (unsharded, max_chunks < 2^20).  Results equal the reference's voxel for voxel.
 *
 * deform: kfids[n_map] strictly increasing, Rt n_map x 12 floats (R row-major, then t) — the MapKfidRt OnMapChange
 * fills (Rt = Twc_new * Tcw_at_integration).  Voxels whose kfid has no entry are dropped, as in the reference
 * (stats.discarded, its "num discarded voxels").  stats.undefined counts voxels whose new chunk (floor(p / (16 res)))
 * and new voxel (floor(p / res)) roundings disagree so that the reference indexes outside the chunk's voxel array
 * (undefined behaviour there): they are dropped.  PLVS_ERR_CAPACITY if the deformed map needs more than max_chunks
 * chunks; the map is then unchanged.  The updated-chunk list is empty afterwards (the reference leaves
 * meshesToUpdate alone and moves the stored meshes instead: deform_mesh). */
typedef struct plvs_tsdf_deform_stats {
  int32_t new_chunks;   /* chunks of the deformed map                                  */
  int64_t moved;        /* voxels that moved (copied or merged)                        */
  int64_t discarded;    /* known voxels without a transformation                       */
  int64_t undefined;    /* see above                                                   */
} plvs_tsdf_deform_stats;
int plvs_hip_tsdf_chisel_enable_deform(plvs_tsdf_chisel* h);
int plvs_hip_tsdf_chisel_deform(plvs_tsdf_chisel* h, const uint32_t* kfids, const float* Rt, int n_map,
                                plvs_tsdf_deform_stats* stats);
/* The chunk ids in the iteration order of the reference's container (deform enabled). */
int plvs_hip_tsdf_chisel_chunk_order(plvs_tsdf_chisel* h, int32_t* ids_xyz, int cap, int* n);
/* The mesh half of ChunkManager::Deform (:1020-1051): vertex = R * vertex + t, normal = R * normal for the vertices
 * whose kfid has a transformation (in place; the others stay).  Host and device-pointer flavours. */
int plvs_hip_tsdf_chisel_deform_mesh(float* vertices, float* normals, const uint32_t* vertex_kfid, int n,
                                     const uint32_t* kfids, const float* Rt, int n_map);
int plvs_hip_tsdf_chisel_deform_mesh_dev(float* d_vertices, float* d_normals, const uint32_t* d_vertex_kfid, int n,
                                         const uint32_t* d_kfids, const float* d_Rt, int n_map, void* stream);

/* Creates or REPLACES one chunk with the given voxel planes (host, 4096 each, id = (z * 16 + y) * 16 + x): the
 * counterpart of download_chunk (a volume snapshot coming back; the reference itself has no volume file for chisel),
 * and what lets tests put analytic distance fields on the device. */
int plvs_hip_tsdf_chisel_upload_chunk(plvs_tsdf_chisel* h, int cx, int cy, int cz, const float* sdf, const float* weight,
                                      const uint32_t* kfid, const uint32_t* rgbw);

/* Halo of a sharded map, for meshing (see plvs_hip_tsdf_chisel_mesh_chunks).
 *   halo_missing  the chunk ids (host, n x 3) the last mesh_chunks call looked for on other ranks and did not hold;
 *                 PLVS_ERR_CAPACITY with *n set if cap is too small.
 *   halo_lookup   owner side: d_found[i] = 1 if this rank has the i-th of the n requested ids (device, n x 3), else 0.
 *   halo_export   owner side: the found chunks' four planes (sdf, weight, kfid, rgbw: 4096 words each = one payload
 *                 row of 16384 words), packed in request order into d_payload (rows = number of found flags set;
 *                 may be null when none is).
 *   halo_import   requester side: the ids it asked for, the found flags and the nfound payload rows received from
 *                 the owners become ghost chunks (read-only copies in free slots of the pool); "found = 0" is
 *                 remembered as "exists nowhere".  PLVS_ERR_CAPACITY if the pool has fewer than nfound free slots.
 *                 All three are asynchronous on `stream`.
 *   halo_clear    drops every ghost (also done by the next integrate / shard_apply / clear call).
 * The exchange itself is halo_gather below, or the caller's own all-to-all (plvs_amd/shard.py: gather_mesh_halo over
 * torch.distributed). */
/*   mesh_probe    the stages of mesh_chunks on the device only (no outputs, no capacity): *nmissing = the foreign
 *                 chunks the list's meshes reach for and this rank does not hold (0: mesh_chunks will succeed).
 *   halo_gather   the whole exchange over an RCCL communicator (ncclComm_t), collective: every rank calls it with
 *                 its own list (possibly empty); rounds of mesh_probe, requests to the owners (ncclSend / ncclRecv),
 *                 halo_lookup / halo_export there, halo_import here, until no rank misses anything.  *fetched = chunks imported.
 *                 Afterwards mesh_chunks(the same list) is a local call. */
int plvs_hip_tsdf_chisel_mesh_probe(plvs_tsdf_chisel* h, const int32_t* chunk_ids_xyz, int nchunks, int* nmissing);
int plvs_hip_tsdf_chisel_halo_gather(plvs_tsdf_chisel* h, void* rccl_comm, const int32_t* chunk_ids_xyz, int nchunks,
                                     int* fetched, void* stream);
int plvs_hip_tsdf_chisel_halo_missing(plvs_tsdf_chisel* h, int32_t* ids_xyz, int cap, int* n);
int plvs_hip_tsdf_chisel_halo_lookup(plvs_tsdf_chisel* h, const int32_t* d_ids_xyz, int n, uint32_t* d_found, void* stream);
int plvs_hip_tsdf_chisel_halo_export(plvs_tsdf_chisel* h, const int32_t* d_ids_xyz, const uint32_t* d_found, int n,
                                     uint32_t* d_payload, void* stream);
int plvs_hip_tsdf_chisel_halo_import(plvs_tsdf_chisel* h, const int32_t* d_ids_xyz, const uint32_t* d_found,
                                     const uint32_t* d_payload, int n, int nfound, void* stream);
int plvs_hip_tsdf_chisel_halo_clear(plvs_tsdf_chisel* h);

/* ------------------------------------------------- dense stereo (libelas: the part its GPU build moves to the device)
 * Replaces the two methods the reference's own accelerated build overrides — class ElasGPU : public Elas,
 * Thirdparty/libelas-gpu/GPU/elas_gpu.h:41-45:
 *   computeDisparity   Elas::computeDisparity, Thirdparty/libelas-gpu/CPU/elas.cpp:840-968 (findMatch :739-837)
 *   adaptiveMean       Elas::adaptiveMean,     Thirdparty/libelas-gpu/CPU/elas.cpp:1349-1572
 * as PointCloudKeyFrame::ProcessStereoLibelas reaches them (src/PointCloudKeyFrame.cc:335-432 ->
 * libelas::ElasInterface::process -> Elas::process, elas.cpp:36-159) — and, beyond that build, the other stages a pair
 * goes through (entry points further down): the descriptor images (plvs_hip_elas_set_images), the candidate loop of
 * computeSupportMatches (_support_candidates), leftRightConsistencyCheck (_left_right_check), removeSmallSegments
 * (_remove_small_segments) and gapInterpolation (_gap_interpolation).  The support filters, the Delaunay triangulation,
 * the planes and the grid stay the caller's host code.  All pointers are HOST pointers (the reference's call sites hand
 * over host memory); results are bit-identical to the CPU methods — of a reference whose uninitialised reads see zeros
 * (oracle/ref/elas_zero_malloc.h: its sources read allocated memory they never wrote; parity is pinned through that
 * hook).  Preconditions the post-processing stages share with the reference: invalid pixels hold -10;
 * speckle_sim_threshold < 10.
 *   params            the fields of Elas::Parameters (elas.h:62-90) the two methods read
 *   compute_disparity support: n_support x {u, v, d} (Elas::support_pt); tri: n_tri x 36-byte Elas::triangle records
 *                     {c1, c2, c3, t1a, t1b, t1c, t2a, t2b, t2c}; disparity_grid / grid_dims as Elas::createGrid leaves
 *                     them; I1_desc / I2_desc: Descriptor::I_desc, 16 * width * height bytes each — pass both with the
 *                     first call of an image pair and NULL with the second (the right image's): the staged copies are
 *                     reused.  D: width x height floats (width/2 x height/2 with subsampling), -10 where no triangle
 *                     reaches, -1 where no candidate matched.
 *   adaptive_mean     D in place; width x height is the IMAGE size (D is half of it with subsampling).  Where the
 *                     reference reads scratch memory it never wrote (column 3 and the first / last three rows of its
 *                     D_tmp) the value is 0.0, the content of fresh pages. */
typedef struct plvs_elas plvs_elas;
typedef struct {
  int32_t subsampling;   /* Parameters::subsampling (PLVS: skDownsampleStep even) */
  int32_t grid_size;     /* 20 */
  int32_t match_texture; /* 1 (ROBOTICS) */
  float beta;            /* 0.02 */
  float gamma;           /* 3 */
  float sigma;           /* 1 */
  float sradius;         /* 2 */
  /* support_candidates only: */
  int32_t disp_min, disp_max;    /* 0, 255 */
  int32_t candidate_stepsize;    /* 5 */
  int32_t support_texture;       /* 10 */
  int32_t lr_threshold;          /* 2 (also left_right_check) */
  float support_threshold;       /* 0.85 (ROBOTICS) */
  /* the post-processing entry points only: */
  float speckle_sim_threshold;   /* 1 */
  int32_t speckle_size;          /* 200 */
  int32_t ipol_gap_width;        /* 3 (ROBOTICS), 5000 (MIDDLEBURY) */
  int32_t add_corners;           /* 0 (ROBOTICS), 1 (MIDDLEBURY) */
} plvs_elas_params;
int plvs_hip_elas_create(const plvs_elas_params* params, plvs_elas** out);
int plvs_hip_elas_destroy(plvs_elas* e);
int plvs_hip_elas_compute_disparity(plvs_elas* e, const int32_t* support, int n_support, const void* tri, int n_tri,
                                    const int32_t* disparity_grid, const int32_t* grid_dims, const uint8_t* I1_desc,
                                    const uint8_t* I2_desc, int width, int height, int right_image, float* D);
int plvs_hip_elas_adaptive_mean(plvs_elas* e, float* D, int width, int height);
/* One step further than ElasGPU: the candidate loop of Elas::computeSupportMatches (elas.cpp:434-456 with
 * computeMatchingDisparity :296-410) — for every point of the regular grid the disparity of the best forward match that
 * the backward match confirms, -1 otherwise.  D_can: D_can_width x D_can_height int16 with the counts of elas.cpp:425-428
 * (ceil(width / step) x ceil(height / step), step = candidate_stepsize, + 1 if odd with subsampling); row 0 and column 0
 * come back 0 as the reference's calloc leaves them.  The caller goes on with the reference's own
 * removeInconsistentSupportPoints / removeRedundantSupportPoints / addCornerSupportPoints on this grid.  The descriptor
 * images stay staged: the compute_disparity calls of the same pair pass NULL. */
int plvs_hip_elas_support_candidates(plvs_elas* e, const uint8_t* I1_desc, const uint8_t* I2_desc, int width, int height,
                                     int16_t* D_can);
/* The descriptor images on the device: libelas::Descriptor (descriptor.cpp:30-131 — filter::sobel3x3 over the zero-padded
 * buffer of Elas::process, elas.cpp:39-57, and 16 samples per pixel) of both images, `stride` bytes per line.  They stay
 * staged: support_candidates and compute_disparity then take NULL descriptor pointers.  Elas::process builds its
 * Descriptor objects itself (no virtual method reaches them), so this serves a caller that drives the stages on its own —
 * a replacement of ElasInterface::process — and saves the CPU descriptors and the 15 MB upload.
 * download_descriptors: the staged images (16 * width * height bytes each), for parity checks. */
int plvs_hip_elas_set_images(plvs_elas* e, const uint8_t* I1, const uint8_t* I2, int width, int height, int stride);
int plvs_hip_elas_download_descriptors(plvs_elas* e, uint8_t* I1_desc, uint8_t* I2_desc);
/* The post-processing between computeDisparity and adaptiveMean, also virtual in Elas: leftRightConsistencyCheck
 * (elas.cpp:971-1040; both maps in place), removeSmallSegments (:1043-1160), gapInterpolation (:1163-1347).  width x height
 * is the IMAGE size (the maps are half of it with subsampling).  remove_small_segments expects what the left/right check
 * leaves — every invalid pixel at -10 — and a similarity threshold below 10: the reference's segments are then the
 * connected components of a symmetric relation, which is what the device computes. */
int plvs_hip_elas_left_right_check(plvs_elas* e, float* D1, float* D2, int width, int height);
int plvs_hip_elas_remove_small_segments(plvs_elas* e, float* D, int width, int height);
int plvs_hip_elas_gap_interpolation(plvs_elas* e, float* D, int width, int height);
/* The post-processing of Elas::process (elas.cpp:100-135) in ONE call on the maps the two plvs_hip_elas_compute_disparity
 * calls of the pair left in HBM (they always do; pass D = NULL there to skip the download): leftRightConsistencyCheck
 * (lr_threshold >= 0), removeSmallSegments (speckle_size > 0), gapInterpolation (ipol_gap_width > 0), adaptiveMean
 * (filter_adaptive_mean) — the right map only unless postprocess_only_left (PLVS sets it, src/PointCloudKeyFrame.cc:349).
 * D1 / D2: host destinations of the finished maps, may be NULL; the maps stay in HBM. */
int plvs_hip_elas_postprocess(plvs_elas* e, int width, int height, int postprocess_only_left, int filter_adaptive_mean,
                              float* D1, float* D2);
/* PointCloudKeyFrame::ProcessStereoLibelas' disparity -> depth (src/PointCloudKeyFrame.cc:399-420) from the left map in HBM
 * into d_depth (DEVICE, width x height floats: what plvs_hip_cloudgen_generate_dev reads): bf / d, with subsampling only
 * at rows step * m1 and pixels step * n1, step * n1 + 1 (zero elsewhere); step = PointCloudMapping::skDownsampleStep. */
int plvs_hip_elas_depth_dev(plvs_elas* e, float bf, int step, float* d_depth, int width, int height, void* stream);

/* ------------------------------------------------- dense stereo (semi-global matching)
 * Replaces sgm::StereoSGM as PointCloudKeyFrame::ProcessStereoLibsgm uses it
 * (src/PointCloudKeyFrame.cc:435-481): StereoSGM(width, height, 64, 8, 8, HOST2HOST) with
 * Parameters(P1 = 10, P2 = 120, uniqueness = 0.95f) (Thirdparty/libsgm/include/libsgm.h:62-88),
 * execute(left, right, dst) (src/stereo_sgm.cpp:133-181): census, 8-path aggregation,
 * winner-takes-all + uniqueness, 3x3 median, left-right check.  Images are width x height u8,
 * tightly packed; the disparity is u8, 0 = invalid (PLVS: depth = bf / disparity). */
typedef struct plvs_sgm plvs_sgm;
int plvs_hip_sgm_create(int width, int height, int disparity_size, int p1, int p2, float uniqueness,
                        plvs_sgm** out);
int plvs_hip_sgm_destroy(plvs_sgm* s);
/* Host images in, host disparity out; synchronous (EXECUTE_INOUT_HOST2HOST). */
int plvs_hip_sgm_execute(plvs_sgm* s, const uint8_t* left, const uint8_t* right, uint8_t* disparity);
/* Device images / disparity (EXECUTE_INOUT_CUDA2CUDA); asynchronous on `stream`. */
int plvs_hip_sgm_execute_dev(plvs_sgm* s, const uint8_t* d_left, const uint8_t* d_right,
                             uint8_t* d_disparity, void* stream);
/* Parity accessors of the last call: which = 0 / 1 census left / right (u32 per pixel), 2 summed path
 * costs (u16, width * height * 64), 3 / 4 raw left / right disparity, 5 / 6 after the median (u8). */
int plvs_hip_sgm_download(plvs_sgm* s, int which, void* out);

/* ------------------------------------------------- depth image -> cloud (T0)
 * Replaces PointCloudMapping::GeneratePointCloudInCameraFrameBGRA
 * src/PointCloudMapping.cc:929-1031 (caller IntegratePointCloudKeyframe path,
 * :660-677 via GeneratePointCloudInCameraFrame :1228) for the shipped build:
 * pcl::PointSurfelSegment with normals, EigthNeighborhoodIndicesFast,
 * Segmentation.on 0, filterDepth.on 0.  Every grid pixel (m, n), m and n
 * multiples of `step` (PointCloudMapping.downSampleStep, 2), with
 * min_depth < d < max_depth (double limits, PointCloudMapTypes.h:59-60)
 * becomes one point, in row-major grid order: p = (gx*d, gy*d, d), the
 * r, g, b members take the image's B, G, R bytes (:978-980), normals are the
 * area-weighted cross products of the up/left/down/right neighbours in
 * double (:1007-1030), label 0.                                            */

/* pcl::PointSurfelSegment (include/PointSurfelSegment.h:63-94), 48 bytes. */
typedef struct plvs_point_surfel {
  float x, y, z;
  uint32_t kfid;
  float normal_x, normal_y, normal_z, normal_pad;
  uint8_t b, g, r, a;
  float depth;
  uint32_t label, label_confidence;
} plvs_point_surfel;

typedef struct plvs_cloudgen plvs_cloudgen;

/* Rows of matCamGridPoints_ actually used: ceil(w/step) * ceil(h/step). */
int plvs_hip_cloudgen_num_grid_points(int width, int height, int step);
/* InitCamGridPoints (src/PointCloudMapping.cc:796-905) for an undistorted,
 * unrectified camera (mDistCoef[0] == 0): grid[2*ii] = (n - cx)/fx,
 * grid[2*ii+1] = (m - cy)/fy.  Host function, no GPU.  A PLVS build with
 * distortion hands over its own matCamGridPoints_ instead.
 * NOTE (pinned by the reference's own source, tests/test_oracle_pinned_cloudgen.py): PLVS fills fx, fy, cx, cy from the
 * FLOAT entries of K (src/PointCloudMapping.cc:177-180: `pCameraParams_->fx = mK.at<float>(0,0)`), so the doubles it divides
 * by are float values: pass (double)(float)fx ... for a table identical to the reference's. */
int plvs_hip_cloudgen_grid_points(int width, int height, int step, double fx, double fy, double cx,
                                  double cy, float* grid);
/* grid_points: host, num_grid_points x 2 f32 = matCamGridPoints_ after
 * InitCamGridPoints (computed once per camera there too). */
int plvs_hip_cloudgen_create(int width, int height, int step, const float* grid_points,
                             plvs_cloudgen** out);
int plvs_hip_cloudgen_destroy(plvs_cloudgen* h);
/* Host flavour: depth (f32, pitch in floats) and colour (BGR u8, pitch in
 * bytes) are host images; out receives *n records (capacity >=
 * num_grid_points); pixel_to_point (height x width int32, -1 = no point) may
 * be NULL.  Synchronous. */
int plvs_hip_cloudgen_generate(plvs_cloudgen* h, const float* depth, int depth_pitch,
                               const uint8_t* bgr, int bgr_pitch, double min_depth, double max_depth,
                               uint32_t kfid, plvs_point_surfel* out, int capacity,
                               int32_t* pixel_to_point, int* n);
/* Device flavour: images resident in HBM; writes the arrays the TSDF entry
 * points consume (d_xyz n x 3; d_rgb n x 3 for chisel and/or d_rgba n x 4 for
 * voxblox; d_kfid n) plus optional d_normals (n x 3), d_point_depth (n),
 * d_pixel_to_point; every output except d_xyz may be NULL.  The point count
 * goes to *d_count (device, may be NULL) and, if n != NULL, to *n after a
 * stream synchronise (the batch integrate needs host offsets). */
int plvs_hip_cloudgen_generate_dev(plvs_cloudgen* h, const float* d_depth, int depth_pitch,
                                   const uint8_t* d_bgr, int bgr_pitch, double min_depth,
                                   double max_depth, uint32_t kfid, float* d_xyz, uint8_t* d_rgb,
                                   uint8_t* d_rgba, uint32_t* d_kfid, float* d_normals,
                                   float* d_point_depth, int32_t* d_pixel_to_point, int* d_count,
                                   void* stream, int* n);

/* ------------------------------------------------------------- TSDF (voxblox)
 * Block-hashed (16^3) TSDF layer with per-point ray-cast integration.
 *
 * Replaces:
 *   PointCloudMapVoxblox::InsertCloud        src/PointCloudMapVoxblox.cc:81-99
 *    -> TsdfServer::insertPointCloud         Thirdparty/voxblox_server/src/tsdf_server.cc:476-555
 *    -> SimpleTsdfIntegrator::integratePointCloud
 *                                            Thirdparty/voxblox/src/integrator/tsdf_integrator.cc:266-327
 *   with RayCaster (integrator_utils.cc:137-235) and updateTsdfVoxel (:173-232).
 * Voxel payload 12 B: {f32 distance, f32 weight, u8 r,g,b,a} (core/voxel.h:12-18).
 *
 * Integration method: "simple" in the single-thread visiting order of the
 * reference (ThreadSafeIndex mixed order).  "merged" iterates a
 * std::unordered_map and "fast" (PLVS's YAML default) is racy by design
 * (tsdf_integrator.cc:505-569): neither has a reproducible result, see DESIGN.md.
 */
typedef struct plvs_tsdf_voxblox_params {
  float voxel_size;        /* tsdf_voxel_size (PointCloudMapping.resolution)                  */
  float truncation;        /* default_truncation_distance, 0.1  (PointCloudMapVoxblox.cc:57)  */
  float max_weight;        /* 10000                                                     (:58) */
  float min_ray_length;    /* 0.1                                                       (:60) */
  float max_ray_length;    /* 5.0                                                       (:61) */
  int32_t voxel_carving;   /* PointCloudMapping.useCarving                              (:59) */
  int32_t max_blocks;      /* capacity of the device block pool (48 KiB of HBM each)          */
  int32_t shard_rank;      /* multi-GPU: owner(block) = three-prime hash(id) mod shard_count  */
  int32_t shard_count;
} plvs_tsdf_voxblox_params;

typedef struct plvs_tsdf_voxblox plvs_tsdf_voxblox;

int plvs_hip_tsdf_voxblox_default_params(float voxel_size, int use_carving,
                                         plvs_tsdf_voxblox_params* p);
int plvs_hip_tsdf_voxblox_create(const plvs_tsdf_voxblox_params* p, plvs_tsdf_voxblox** out);
int plvs_hip_tsdf_voxblox_destroy(plvs_tsdf_voxblox* h);
int plvs_hip_tsdf_voxblox_clear(plvs_tsdf_voxblox* h);
/* xyz: n x 3 f32 camera-frame points (must be finite: PLVS's generator only
 * emits valid depths; PLVS_ERR_INVALID_ARG otherwise); rgba: n x 4 u8, the
 * r,g,b,a members of the pcl point; Twc 3x4 row-major. */
int plvs_hip_tsdf_voxblox_integrate(plvs_tsdf_voxblox* h, const float* xyz, const uint8_t* rgba,
                                    int n, const float* Twc);
int plvs_hip_tsdf_voxblox_integrate_batch_dev(plvs_tsdf_voxblox* h, const float* d_xyz,
                                              const uint8_t* d_rgba, const int32_t* offsets,
                                              int nclouds, const float* d_Twc, void* stream);
/* Queued insertion, as plvs_hip_tsdf_chisel_queue / _flush for the chisel map: PointCloudMapVoxblox::InsertCloud
 * (src/PointCloudMapVoxblox.cc:81) is called once per key frame, the layer is read only by UpdateMap (:160) after at most
 * kMaxNumKeyFramesToInsertInMapInOneStep of them (src/PointCloudMapping.cc:537-556, 594-598).  _queue uploads the cloud
 * and returns; _flush integrates everything queued as ONE batch of the simple integrator, every cloud a scan of its own,
 * in order — the layer of the call-by-call sequence bit for bit, at the per-key-frame cost of a batch.  Every entry point
 * that reads or changes the map (the integrates, block lists, download / upload, the meshers, the halo calls) flushes
 * first; plvs_hip_tsdf_voxblox_clear drops the queue.  Host pointers; arguments as plvs_hip_tsdf_voxblox_integrate. */
int plvs_hip_tsdf_voxblox_queue(plvs_tsdf_voxblox* h, const float* xyz, const uint8_t* rgba, int n, const float* Twc);
int plvs_hip_tsdf_voxblox_flush(plvs_tsdf_voxblox* h);
int plvs_hip_tsdf_voxblox_queued(plvs_tsdf_voxblox* h, int* nclouds);
/* ------------------------------------------- multi-GPU: the ray-sharded integrate (voxblox, "simple"; round 4)
 * New design (the reference is one process): the MAP is sharded by block (shard_rank / shard_count, owner =
 * AnyIndexHash(block id) mod N, block_hash.h:21-24), the WORK by key frame — rank r casts the rays of the clouds
 * c = r (mod N) of every call, through whatever blocks they cross, and every voxel visit travels to the block's owner as
 * a 16-byte record {block id packed in 8 B, voxel | cloud << 12, sequence number of the ray}.  Every rank is given the
 * same clouds.  updateTsdfVoxel is order dependent: the owner applies a voxel's visits in the reference's order (cloud
 * after cloud, the mixed point order inside), so the union of the shards is the single-device layer bit for bit for
 * every N (tests/test_tsdf_voxblox_shard.py).  One call =
 *   shard_walk    this rank's rays; send_counts[N] = records per destination
 *   shard_pack    the records grouped by destination in rank order, 16 B each (d_send: sum of send_counts records)
 *   all-to-all    counts, then the records (the caller's transport: plvs_amd/shard.py does it over torch.distributed)
 *   shard_apply   the records grouped by source in rank order, recv_counts[N]; the clouds again (the operands of a visit
 *                 are recomputed at the owner)
 * shard_count = 1 is allowed (the rank sends to itself).  Before round 4 a sharded voxblox map had every rank cast every
 * ray and keep its own blocks' visits (plvs_hip_tsdf_voxblox_integrate_batch_dev on a handle with shard_count > 1 still
 * does that). */
int plvs_hip_tsdf_voxblox_shard_walk(plvs_tsdf_voxblox* h, const float* d_xyz, const int32_t* offsets, int nclouds,
                                     const float* d_Twc, int64_t* send_counts, void* stream);
int plvs_hip_tsdf_voxblox_shard_pack(plvs_tsdf_voxblox* h, void* d_send, void* stream);
int plvs_hip_tsdf_voxblox_shard_apply(plvs_tsdf_voxblox* h, const void* d_recv, const int64_t* recv_counts,
                                      const float* d_xyz, const uint8_t* d_rgba, const int32_t* offsets, int nclouds,
                                      const float* d_Twc, void* stream);
/* The three phases with the exchanges over RCCL (grouped ncclSend / ncclRecv); rccl_comm is the caller's ncclComm_t, one
 * process per GPU, its size and rank those of the map's shard_count / shard_rank. */
int plvs_hip_tsdf_voxblox_integrate_sharded(plvs_tsdf_voxblox* h, void* rccl_comm, const float* d_xyz, const uint8_t* d_rgba,
                                            const int32_t* offsets, int nclouds, const float* d_Twc, void* stream);

/* The "fast" integration method (PointCloudMapping.voxbloxIntegrationMethod: "fast" — the default of PLVS's YAML files):
 * FastTsdfIntegrator::integratePointCloud (Thirdparty/voxblox/src/integrator/tsdf_integrator.cc:505-605) in its
 * integrator_threads = 1 schedule, with start_voxel_subsampling_factor 2, max_consecutive_ray_collisions 2 and
 * clear_checks_every_n_frames 1 as src/PointCloudMapVoxblox.cc:70-72 sets them and no time limit: per scan a point casts
 * a ray only if no earlier point of the scan lies in the same voxel of half the voxel size; the ray runs from its far end
 * towards the sensor and stops at the third voxel in a row an earlier ray of the scan went through.  Both tests go
 * through the reference's ApproxHashSet<20, 10000> (utils/approx_hash_array.h) — lossy, and kept from scan to scan —
 * which this library reproduces word for word: the maps are bit-identical to the reference's with one integrator thread
 * (with more threads the reference itself is a race; measured here: 99.6 % of the voxels identical).  Every cloud of a
 * batch is one scan.  Same arguments as plvs_hip_tsdf_voxblox_integrate_batch_dev / _integrate.  The integrator exists to
 * spare a CPU nine tenths of the voxel updates; on the device it costs MORE time than "simple" (it needs several
 * rounds over the scan, plvs_hip_tsdf_voxblox_fast_rounds) — choose it for maps equal to a PLVS run, not for speed. */
int plvs_hip_tsdf_voxblox_integrate_fast_batch_dev(plvs_tsdf_voxblox* h, const float* d_xyz, const uint8_t* d_rgba,
                                                   const int32_t* offsets, int nclouds, const float* d_Twc, void* stream);
int plvs_hip_tsdf_voxblox_integrate_fast(plvs_tsdf_voxblox* h, const float* xyz, const uint8_t* rgba, int n,
                                         const float* Twc);
/* Rounds the last fast call needed to settle the rays' stopping points (diagnostic). */
int plvs_hip_tsdf_voxblox_fast_rounds(plvs_tsdf_voxblox* h, int* rounds);
/* The "merged" integration method (PointCloudMapping.voxbloxIntegrationMethod: "merged"):
 * MergedTsdfIntegrator::integratePointCloud (Thirdparty/voxblox/src/integrator/tsdf_integrator.cc:329-492) with
 * integrator_threads = 1 and enable_anti_grazing off (src/PointCloudMapVoxblox.cc:67): the points that end in the
 * same voxel are folded into one weighted point and colour and cast as ONE ray (clearing rays: the first point of
 * their voxel); the bundles are integrated in the iteration order of the reference's
 * std::unordered_map<AnyIndex, ..., AnyIndexHash> filled in the mixed visiting order.  That order is produced on the
 * host by filling the same container (libstdc++'s, with the reference's hash) in the same sequence — it is the one
 * host step; validity / end voxel per point, the fold of each bundle, the rays and the voxel updates run on the
 * device.  Host pointers; synchronous.  Same arguments as plvs_hip_tsdf_voxblox_integrate. */
int plvs_hip_tsdf_voxblox_integrate_merged(plvs_tsdf_voxblox* h, const float* xyz, const uint8_t* rgba, int n,
                                           const float* Twc);

/* A cloud WITH NORMALS, as PointCloudMapVoxblox::LoadMap feeds the saved cloud through
 * (src/PointCloudMapVoxblox.cc:233-258 -> TsdfServer::insertWorldPointCloud, tsdf_server.cc:577-660, T = identity
 * there -> TsdfIntegratorBase::integrateWorlPointCloud, tsdf_integrator.cc:35-82): points in cloud order, each
 * casting point + normal * truncation -> point - normal * truncation (normals n x 3 f32, normalised here as
 * there), weight 1, ray_start in the sensor origin's place; no ray-length test.  Host pointers; synchronous.
 * Non-finite points are refused (the reference drops them). */
int plvs_hip_tsdf_voxblox_integrate_world_normals(plvs_tsdf_voxblox* h, const float* xyz, const uint8_t* rgba,
                                                  const float* normals, int n, const float* Twc);
/* TsdfIntegratorBase::integrateWorlPointCloud (tsdf_integrator.cc:35-82) leaves the blocks it creates in the
 * integrator's temp_block_map_: it never calls updateLayerWithStoredBlocks, so they join the layer only with the next
 * integratePointCloud call (:306 / :343) — after PointCloudMapVoxblox::LoadMap the map's block list, its meshes and
 * Block::updated() miss them until a camera cloud arrives.  enable = 1 reproduces that: blocks created by
 * plvs_hip_tsdf_voxblox_integrate_world_normals stay out of num_blocks / block_ids / download_block /
 * updated_block_ids / the meshing calls until the next integrate / integrate_batch_dev / integrate_merged (their
 * voxels accumulate meanwhile, as the reference's do); enable = 0 (default) shows them at once and makes any waiting
 * blocks visible. */
int plvs_hip_tsdf_voxblox_set_deferred_world_blocks(plvs_tsdf_voxblox* h, int enable);
int plvs_hip_tsdf_voxblox_last_stats(plvs_tsdf_voxblox* h, plvs_tsdf_stats* s);
int plvs_hip_tsdf_voxblox_num_blocks(plvs_tsdf_voxblox* h, int* n);
int plvs_hip_tsdf_voxblox_block_ids(plvs_tsdf_voxblox* h, int32_t* ids_xyz, int cap, int* n);
/* The blocks the last integrate call visited (Block::updated() = true, tsdf_integrator.cc:151): host / device list. */
int plvs_hip_tsdf_voxblox_updated_block_ids(plvs_tsdf_voxblox* h, int32_t* ids_xyz, int cap, int* n);
int plvs_hip_tsdf_voxblox_updated_block_ids_dev(plvs_tsdf_voxblox* h, int32_t* d_ids_xyz, int cap,
                                                int* n, void* stream);
int plvs_hip_tsdf_voxblox_download_block(plvs_tsdf_voxblox* h, int bx, int by, int bz,
                                         float* distance, float* weight, uint32_t* rgba);

/* Surface extraction: MeshIntegrator<TsdfVoxel>::updateMeshForBlock
 * (Thirdparty/voxblox/include/voxblox/mesh/mesh_integrator.h:231-251: extractBlockMesh :165-229 with
 * MarchingCubes::meshCube mesh/marching_cubes.h:66-102, updateMeshColor :348-368; min_weight 1e-4, use_color)
 * for every block of the list — what TsdfServer::updateMesh (Thirdparty/voxblox_server/src/tsdf_server.cc:775-787)
 * runs on the blocks integrated into since the last call (generateMesh(only_mesh_updated_blocks = true,
 * clear_updated_flag = true), mesh_integrator.h:111-139) when PointCloudMapVoxblox::UpdateMap calls it
 * (src/PointCloudMapVoxblox.cc:168).  block_ids_xyz: host, nblocks x 3 (the caller accumulates
 * plvs_hip_tsdf_voxblox_updated_block_ids_dev over its integrate calls, as Block::updated() does).
 * Outputs (host): the voxblox::Mesh arrays of the blocks back to back, in list order — vertices / normals
 * (n x 3 f32; three consecutive vertices are one triangle, Mesh::indices is 0 .. n-1) and colors (n x 4 u8:
 * Color r, g, b, a); block c owns [block_first[c], block_first[c+1]) (block_first: nblocks + 1 ints); a block
 * that does not exist owns nothing.  *nvertices = n.  If n > capacity nothing is written but block_first /
 * *nvertices, and PLVS_ERR_CAPACITY is returned.  Sharded map (shard_count > 1): the list holds blocks of THIS rank
 * (another rank's block owns nothing); the cubes on a block's +x / +y / +z faces read up to seven neighbour blocks,
 * which the caller brings in first from their owners (plvs_hip_tsdf_voxblox_halo_* below; a neighbour that was not
 * brought in counts as non-existent). */
int plvs_hip_tsdf_voxblox_mesh_blocks(plvs_tsdf_voxblox* h, const int32_t* block_ids_xyz, int nblocks,
                                      float* vertices, float* normals, uint8_t* colors_rgba, int capacity,
                                      int32_t* block_first, int* nvertices);

/* Creates or REPLACES one block with the given voxel planes (host, 4096 each, index x + 16 * (y + 16 * z)) — the
 * device side of TsdfServer::loadMap (tsdf_server.cc:865-872: io::LoadBlocksFromFile with
 * BlockMergingStrategy::kReplace, core/layer_inl.h:195-197) once the `.proto` file has been read on the host;
 * download_block is the device side of saveMap (:859-863).  The caller marks the block updated (layer_inl.h:215). */
int plvs_hip_tsdf_voxblox_upload_block(plvs_tsdf_voxblox* h, int bx, int by, int bz, const float* distance,
                                       const float* weight, const uint32_t* rgba);

/* Halo of a sharded voxblox map, for meshing.  The ids to fetch are known on the host: the seven +x / +y / +z
 * neighbours of every block to mesh that another rank owns (three-prime block hash mod shard_count).
 *   halo_lookup   owner side: d_found[i] = 1 if this rank has the i-th of the n requested ids (device, n x 3).
 *   halo_export   owner side: the found blocks' three planes (distance, weight, rgba: 4096 words each = one payload
 *                 row of 12288 words), packed in request order (rows = number of flags set; may be null when none is).
 *   halo_import   requester side: ids / found flags / nfound payload rows become ghost blocks (read-only copies in
 *                 free pool slots).  PLVS_ERR_CAPACITY if the pool has fewer than nfound free slots.
 *   halo_clear    drops every ghost (also done by the next integrate / clear call).
 * All asynchronous on `stream` but halo_clear.  plvs_amd/shard.py: sharded_mesh_blocks runs the exchange over
 * torch.distributed. */
int plvs_hip_tsdf_voxblox_halo_lookup(plvs_tsdf_voxblox* h, const int32_t* d_ids_xyz, int n, uint32_t* d_found, void* stream);
int plvs_hip_tsdf_voxblox_halo_export(plvs_tsdf_voxblox* h, const int32_t* d_ids_xyz, const uint32_t* d_found, int n,
                                      uint32_t* d_payload, void* stream);
int plvs_hip_tsdf_voxblox_halo_import(plvs_tsdf_voxblox* h, const int32_t* d_ids_xyz, const uint32_t* d_found,
                                      const uint32_t* d_payload, int n, int nfound, void* stream);
int plvs_hip_tsdf_voxblox_halo_clear(plvs_tsdf_voxblox* h);
/* The exchange over an RCCL communicator (ncclComm_t), collective: every rank calls it with the blocks it is about
 * to mesh (possibly none); one round of requests / flags / payload rows (ncclSend / ncclRecv groups).  *fetched =
 * blocks imported.  Afterwards mesh_blocks(the same list) is a local call. */
int plvs_hip_tsdf_voxblox_halo_gather(plvs_tsdf_voxblox* h, void* rccl_comm, const int32_t* block_ids_xyz, int nblocks,
                                      int* fetched, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PLVS_HIP_H_ */
