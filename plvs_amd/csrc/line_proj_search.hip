// The two guided line searches Tracking runs on every frame (single-camera frames):
//   LineMatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, bLargerSearch, bMono)
//                          src/LineMatcher.cc:837-1230, called by Tracking::TrackWithMotionModel (:3653)
//   LineMatcher::SearchByProjection(Frame& F, const std::vector<MapLinePtr>&, bLargerSearch)
//                          src/LineMatcher.cc:1286-1560, called by Tracking::SearchLocalLines (:4576)
// Same split as the point searches (orb_search.hip): the candidate windows — the (theta, d) line grid
// of Frame::GetLineFeaturesInArea, src/Frame.cc:1326-1475 — and the two point-to-line gates are
// evaluated on the host for every projected line first; ALL candidate descriptor distances then go
// through one batched launch (plvs_hip_hamming_pairs); the order-dependent part — best / second best,
// ratio test, "a line that already holds an observed map line is skipped", rotation histogram — replays
// the reference's loop over those distances.
#include <cmath>
#include <vector>

#include "common.hpp"

namespace {

constexpr int kRows = 36;       // LINE_THETA_GRID_ROWS, include/Frame.h:72
constexpr int kCols = 160;      // LINE_D_GRID_COLS, :73
constexpr int kThHigh = 110;    // LineMatcher::TH_HIGH
constexpr int kHisto = 12;      // HISTO_LENGTH
constexpr float kEps = 1.1920929e-07f;   // std::numeric_limits<float>::epsilon()

// All query x train distances of two small descriptor sets (a few hundred lines each), started BEFORE the host evaluates
// its windows and gates and collected after: the launch-and-wait round trip of the device (~40 us, more than the 75 000
// popcounts take) runs beside the ~20 us of host work instead of behind it.  The u16 matrix is written straight into the
// pinned block.
__global__ __launch_bounds__(256) void hamming_matrix_kernel(const uint4* __restrict__ query, int nq, const uint4* __restrict__ train,
                                                             int nt, uint16_t* __restrict__ out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nq * nt) return;
  const int q = p / nt, t = p - q * nt;
  const uint4 a0 = query[2 * q], a1 = query[2 * q + 1], b0 = train[2 * t], b1 = train[2 * t + 1];
  out[p] = (uint16_t)(__popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
                      __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w));
}
constexpr int kMatrixMaxPairs = 1 << 18;   // (512 x 512 lines; beyond that the candidate pairs go through plvs_hip_hamming_pairs)
struct MatrixJob {
  const uint16_t* dist = nullptr;   // nq x nt, valid after matrix_wait
  int nt = 0;
  bool started = false;
};
int matrix_begin(MatrixJob& J, const uint8_t* query, int nq, const uint8_t* train, int nt) {
  J.started = false;
  if (nq <= 0 || nt <= 0 || (long long)nq * nt > kMatrixMaxPairs) return PLVS_OK;
  auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  const size_t o_q = 0, o_t = o_q + up16((size_t)nq * 32), o_d = o_t + up16((size_t)nt * 32), total = o_d + up16(sizeof(uint16_t) * (size_t)nq * nt);
  plvs::HostStage& st = plvs::thread_stage();
  PLVS_HIP_TRY(st.reserve(total));
  memcpy(st.pinned + o_q, query, (size_t)nq * 32);
  memcpy(st.pinned + o_t, train, (size_t)nt * 32);
  // (the descriptors are copied: every one is read by hundreds of threads, and a read of pinned host memory is not cached)
  PLVS_HIP_TRY(hipMemcpyAsync(st.dev, st.pinned, o_d, hipMemcpyHostToDevice, st.stream));
  hipLaunchKernelGGL(hamming_matrix_kernel, dim3(plvs::ceil_div((size_t)nq * nt, 256)), dim3(256), 0, st.stream,
                     reinterpret_cast<const uint4*>(st.dev + o_q), nq, reinterpret_cast<const uint4*>(st.dev + o_t), nt,
                     reinterpret_cast<uint16_t*>(st.pinned + o_d));
  PLVS_KERNEL_CHECK();
  J.dist = reinterpret_cast<const uint16_t*>(st.pinned + o_d);
  J.nt = nt;
  J.started = true;
  return PLVS_OK;
}
int matrix_wait(MatrixJob& J) {
  if (J.started) PLVS_HIP_TRY(hipStreamSynchronize(plvs::thread_stage().stream));
  return PLVS_OK;
}
struct MatrixGuard {   // no call leaves with its kernel still writing into the thread's staging block (error returns)
  MatrixJob& j;
  ~MatrixGuard() { if (j.started) (void)hipStreamSynchronize(plvs::thread_stage().stream); }
};

struct Rep { float theta, d, nx, ny; };

// Geom2DUtils::GetLine2dRepresentation (include/Geom2DUtils.h:135-159): unit normal with nx >= 0
Rep representation(float xs, float ys, float xe, float ye) {
  Rep r;
  r.nx = ye - ys;
  r.ny = xs - xe;
  if (r.nx < 0) { r.nx *= -1.0f; r.ny *= -1.0f; }
  const float inv = 1.0f / std::sqrt(r.nx * r.nx + r.ny * r.ny);
  r.nx *= inv;
  r.ny *= inv;
  r.d = r.nx * xe + r.ny * ye;
  r.theta = std::atan2(r.ny, r.nx);
  return r;
}

struct LineGrid {   // Frame::mLineGrid as one CSR table, cells in (column, row) order
  const plvs_line_frame_view* F;
  float theta_inv, d_inv;
  std::vector<int> first, items;

  explicit LineGrid(const plvs_line_frame_view* f) : F(f) {
    theta_inv = (float)kRows / (float)M_PI;           // mfLineGridElementThetaInv, src/Frame.cc:264
    d_inv = (float)kCols / (2.0f * f->max_diag);      // mfLineGridElementDInv, :265
    std::vector<int> cell((size_t)f->n, -1);
    first.assign((size_t)kRows * kCols + 1, 0);
    for (int i = 0; i < f->n; ++i) {                  // PosLineInGrid, :1477-1495
      const plvs_keyline& k = f->keylines_un[i];
      const Rep r = representation(k.startPointX, k.startPointY, k.endPointX, k.endPointY);
      const int row = (int)std::round((r.theta - (-M_PI_2)) * theta_inv);
      const int col = (int)std::round((r.d + f->max_diag) * d_inv);
      if (row < 0 || row >= kRows || col < 0 || col >= kCols) continue;
      cell[i] = col * kRows + row;
      ++first[(size_t)cell[i] + 1];
    }
    for (size_t c = 0; c + 1 < first.size(); ++c) first[c + 1] += first[c];
    items.resize((size_t)f->n);
    std::vector<int> fill((size_t)kRows * kCols, 0);
    for (int i = 0; i < f->n; ++i)
      if (cell[i] >= 0) items[(size_t)first[cell[i]] + fill[cell[i]]++] = i;
  }

  // Frame::GetLineFeaturesInArea on the unwrapped interval (:1352-1475): the (theta, d) manifold wraps at
  // +-pi/2 with a sign flip of d
  void window(float tmin, float tmax, float dmin, float dmax, bool check, int lo, int hi, std::vector<int>& out) const {
    if (tmin < -M_PI_2) {
      window((float)(tmin + M_PI), (float)(M_PI_2 - kEps), -dmax, -dmin, check, lo, hi, out);
      window((float)(-M_PI_2 + kEps), tmax, dmin, dmax, check, lo, hi, out);
      return;
    }
    if (tmax > M_PI_2) {
      window((float)(-M_PI_2 + kEps), (float)(tmax - M_PI), -dmax, -dmin, check, lo, hi, out);
      window(tmin, (float)(M_PI_2 - kEps), dmin, dmax, check, lo, hi, out);
      return;
    }
    const int r0 = std::max(0, (int)std::floor((tmin - (-M_PI_2)) * theta_inv));
    if (r0 >= kRows) return;
    const int r1 = std::min(kRows - 1, (int)std::floor((tmax - (-M_PI_2)) * theta_inv));
    if (r1 < 0) return;
    const int c0 = std::max(0, (int)std::floor((dmin + F->max_diag) * d_inv));
    if (c0 >= kCols) return;
    const int c1 = std::min(kCols - 1, (int)std::floor((dmax + F->max_diag) * d_inv));
    if (c1 < 0) return;
    // (the cells of a column are consecutive in the table: rows r0 .. r1 of column ix are ONE range of items, in the
    // reference's order — a fifth of the loop trips of a cell-by-cell walk)
    for (int ix = c0; ix <= c1; ++ix) {
      const size_t ca = (size_t)ix * kRows + (size_t)r0, cb = (size_t)ix * kRows + (size_t)r1 + 1;
      for (int k = first[ca]; k < first[cb]; ++k) {
        const int j = items[(size_t)k];
        if (check && (F->keylines_un[j].octave < lo || F->keylines_un[j].octave > hi)) continue;
        out.push_back(j);
      }
    }
  }

  // false: the interval is wider than pi (the reference prints an error and leaves the process)
  bool search(const Rep& r, float dtheta, float dd, int lo, int hi, std::vector<int>& out) const {
    out.clear();
    const bool check = (lo > 0) || (hi < 2147483647);
    const float tmin = r.theta - dtheta, tmax = r.theta + dtheta;
    if (std::fabs(tmin - tmax) > M_PI) return false;
    window(tmin, tmax, r.d - dd, r.d + dd, check, lo, hi, out);
    return true;
  }
};

// the point-to-line gates of a candidate (:990-1046, 1372-1428): both end points of the frame line within the
// chi-square bound of the projected line, and the same on the right image when the line has a stereo match
// shift_s / shift_e: the disparity of the projected end points — mbf * invSz (frame to frame, :1011-1013: the
// projection's own inverse depths) or mbf / mTrackStartDepth (map lines, :1419-1423: a division by the stored depth)
// (the right-image representation of the projected line depends on the projection alone: the reference forms it for every
// candidate, here it is formed once per projected line, and without the angle it never uses)
struct RightRep {
  bool have = false;
  float nx = 0, ny = 0, d = 0;
};
bool passes_gates(const plvs_line_frame_view* F, int i2, const Rep& pr, float inv_sigma2, float th, const float* p,
                  float shift_s, float shift_e, RightRep& rr) {
  const plvs_keyline& k = F->keylines_un[i2];
  const float ds = pr.nx * k.startPointX + pr.ny * k.startPointY - pr.d;
  const float de = pr.nx * k.endPointX + pr.ny * k.endPointY - pr.d;
  if (ds * ds * inv_sigma2 > th || de * de * inv_sigma2 > th) return false;
  if (F->u_right_start != nullptr && F->u_right_start[i2] >= 0 && F->u_right_end[i2] >= 0) {
    if (!rr.have) {   // Geom2DUtils::GetLine2dRepresentation of the shifted end points (representation() above, less theta)
      const float xs = p[0] - shift_s, ys = p[1], xe = p[2] - shift_e, ye = p[3];
      float nx = ye - ys, ny = xs - xe;
      if (nx < 0) { nx *= -1.0f; ny *= -1.0f; }
      const float inv = 1.0f / std::sqrt(nx * nx + ny * ny);
      nx *= inv;
      ny *= inv;
      rr.nx = nx; rr.ny = ny; rr.d = nx * xe + ny * ye;
      rr.have = true;
    }
    const float dsr = rr.nx * F->u_right_start[i2] + rr.ny * k.startPointY - rr.d;
    const float der = rr.nx * F->u_right_end[i2] + rr.ny * k.endPointY - rr.d;
    if (dsr * dsr * inv_sigma2 > th || der * der * inv_sigma2 > th) return false;
  }
  return true;
}

struct Candidates {   // per projected line: its gated candidates, and one distance per candidate
  std::vector<int> first;          // [nq + 1]
  std::vector<int32_t> q, t, dist;
};

int candidate_distances(const plvs_line_frame_view* F, const uint8_t* qdesc, int nq, Candidates& C, MatrixJob& J) {
  C.dist.assign(C.q.size(), 0);
  if (J.started) {   // (the matrix has been on its way since the call began)
    const int rc = matrix_wait(J);
    if (rc != PLVS_OK) return rc;
    for (size_t c = 0; c < C.q.size(); ++c) C.dist[c] = (int)J.dist[(size_t)C.q[c] * (size_t)J.nt + (size_t)C.t[c]];
    return PLVS_OK;
  }
  if (C.q.empty()) return PLVS_OK;
  return plvs_hip_hamming_pairs(qdesc, nq, F->descriptors, F->n, C.q.data(), C.t.data(), (int)C.q.size(), C.dist.data());
}

bool view_ok(const plvs_line_frame_view* F) {
  return F != nullptr && F->n >= 0 && (F->n == 0 || (F->keylines_un && F->descriptors)) && F->line_scale_factors &&
         F->line_inv_level_sigma2 && F->max_diag > 0.0f;
}

}  // namespace

extern "C" {

int plvs_hip_lines_search_by_projection_ff(const plvs_line_frame_view* F, const uint8_t* occupied, int n_last,
                                           const uint8_t* valid, const float* proj, const int32_t* octave,
                                           const float* angle, const uint8_t* desc, const uint8_t* has_obs,
                                           int larger_search, int direction, float nn_ratio, int check_orientation,
                                           int32_t* assigned, int* nmatches) {
  PLVS_REQUIRE(view_ok(F), "bad frame view");
  PLVS_REQUIRE(nmatches && (F->n == 0 || assigned), "null output");
  PLVS_REQUIRE(n_last >= 0 && (n_last == 0 || (valid && proj && octave && angle && desc)), "bad last-frame arrays");
  *nmatches = 0;
  for (int i = 0; i < F->n; ++i) assigned[i] = -1;
  if (F->n == 0 || n_last == 0) return PLVS_OK;
  const float th = larger_search ? 5.024f : 3.84f;   // kChiSquareLinePointProj(Larger), :98-99
  MatrixJob job;
  MatrixGuard guard{job};
  {
    const int rcj = matrix_begin(job, desc, n_last, F->descriptors, F->n);
    if (rcj != PLVS_OK) return rcj;
  }
  const LineGrid grid(F);
  // ---- pass 1: windows and gates for every projected line
  Candidates C;
  C.first.assign((size_t)n_last + 1, 0);
  std::vector<int> win;
  for (int i = 0; i < n_last; ++i) {
    C.first[(size_t)i] = (int)C.q.size();
    if (!valid[i]) continue;
    const float* p = proj + 6 * (size_t)i;
    const Rep pr = representation(p[0], p[1], p[2], p[3]);
    const int lo = octave[i];
    PLVS_REQUIRE(lo >= 0 && lo < F->n_levels, "octave of a last-frame line outside the frame's levels");
    const float sc = F->line_scale_factors[lo];
    const float dtheta = (float)(10 * M_PI / 180.f) * sc, dd = 100.0f * sc;   // Frame::kDeltaTheta, kDeltaD
    bool ok;
    if (direction == 1) ok = grid.search(pr, dtheta, dd, lo, 2147483647, win);        // bForward
    else if (direction == 2) ok = grid.search(pr, dtheta, dd, 0, lo, win);            // bBackward
    else ok = grid.search(pr, dtheta, dd, lo - 1, lo + 1, win);
    if (!ok) {
      plvs::set_error("GetLineFeaturesInArea: search over the full theta interval (the reference terminates here)");
      return PLVS_ERR_INVALID_ARG;
    }
    RightRep rr;
    for (int i2 : win)
      if (passes_gates(F, i2, pr, F->line_inv_level_sigma2[lo], th, p, F->bf * p[4], F->bf * p[5], rr)) {
        C.q.push_back(i);
        C.t.push_back(i2);
      }
  }
  C.first[(size_t)n_last] = (int)C.q.size();
  // ---- all candidate distances in one launch
  const int rc = candidate_distances(F, desc, n_last, C, job);
  if (rc != PLVS_OK) return rc;
  // ---- pass 2: the reference's loop
  std::vector<uint8_t> occ((size_t)F->n, 0);
  if (occupied)
    for (int i = 0; i < F->n; ++i) occ[(size_t)i] = occupied[i];
  const float factor = (float)(kHisto / (2.0 * M_PI));
  std::vector<int> pushed, pushed_bin;
  int n = 0;
  for (int i = 0; i < n_last; ++i) {
    int best = 256, best2 = 256, best_idx = -1;
    for (int c = C.first[(size_t)i]; c < C.first[(size_t)i + 1]; ++c) {
      const int i2 = C.t[(size_t)c];
      if (occ[(size_t)i2]) continue;
      const int d = C.dist[(size_t)c];
      if (d < best) { best2 = best; best = d; best_idx = i2; }
      else if (d < best2) { best2 = d; }
    }
    if (best <= kThHigh && best_idx >= 0) {
      if ((float)best > nn_ratio * (float)best2) continue;
      assigned[best_idx] = i;
      occ[(size_t)best_idx] = has_obs ? has_obs[i] : 1;
      ++n;
      if (check_orientation) {
        float rot = angle[i] - F->keylines_un[best_idx].angle;
        if (rot < 0.0) rot += (float)(2.0 * M_PI); else if (rot > (float)(2.0 * M_PI)) rot -= (float)(2.0 * M_PI);
        int bin = (int)std::round(rot * factor);
        if (bin == kHisto) bin = 0;
        pushed.push_back(best_idx);
        pushed_bin.push_back(bin);
      }
    }
  }
  if (check_orientation) {   // ComputeThreeMaxima (:101-145) and the cut of the other bins (:1215-1225)
    int cnt[kHisto] = {0};
    for (int b : pushed_bin) ++cnt[b];
    int i1 = -1, i2 = -1, i3 = -1, m1 = 0, m2 = 0, m3 = 0;
    for (int b = 0; b < kHisto; ++b) {
      const int s = cnt[b];
      if (s > m1) { m3 = m2; m2 = m1; m1 = s; i3 = i2; i2 = i1; i1 = b; }
      else if (s > m2) { m3 = m2; m2 = s; i3 = i2; i2 = b; }
      else if (s > m3) { m3 = s; i3 = b; }
    }
    if (m2 < 0.1f * (float)m1) { i2 = -1; i3 = -1; }
    else if (m3 < 0.1f * (float)m1) { i3 = -1; }
    for (size_t k = 0; k < pushed.size(); ++k)
      if (pushed_bin[k] != i1 && pushed_bin[k] != i2 && pushed_bin[k] != i3) {
        assigned[pushed[k]] = -1;
        --n;
      }
  }
  *nmatches = n;
  return PLVS_OK;
}

int plvs_hip_lines_search_by_projection(const plvs_line_frame_view* F, const uint8_t* occupied, int n_map,
                                        const uint8_t* in_view, const float* proj, const int32_t* level,
                                        const uint8_t* desc, const uint8_t* has_obs, int larger_search,
                                        float nn_ratio, int32_t* assigned, int* nmatches) {
  PLVS_REQUIRE(view_ok(F), "bad frame view");
  PLVS_REQUIRE(nmatches && (F->n == 0 || assigned), "null output");
  PLVS_REQUIRE(n_map >= 0 && (n_map == 0 || (in_view && proj && level && desc)), "bad map-line arrays");
  *nmatches = 0;
  for (int i = 0; i < F->n; ++i) assigned[i] = -1;
  if (F->n == 0 || n_map == 0) return PLVS_OK;
  const float th = larger_search ? 5.024f : 3.84f;
  MatrixJob job;
  MatrixGuard guard{job};
  {
    const int rcj = matrix_begin(job, desc, n_map, F->descriptors, F->n);
    if (rcj != PLVS_OK) return rcj;
  }
  const LineGrid grid(F);
  Candidates C;
  C.first.assign((size_t)n_map + 1, 0);
  std::vector<int> win;
  for (int m = 0; m < n_map; ++m) {
    C.first[(size_t)m] = (int)C.q.size();
    if (!in_view[m]) continue;
    const float* p = proj + 6 * (size_t)m;
    const int lv = level[m];
    PLVS_REQUIRE(lv >= 0 && lv < F->n_levels, "mnTrackScaleLevel outside the frame's levels");
    const Rep pr = representation(p[0], p[1], p[2], p[3]);
    const float sc = F->line_scale_factors[lv];
    if (!grid.search(pr, (float)(10 * M_PI / 180.f) * sc, 100.0f * sc, lv - 1, lv, win)) {
      plvs::set_error("GetLineFeaturesInArea: search over the full theta interval (the reference terminates here)");
      return PLVS_ERR_INVALID_ARG;
    }
    RightRep rr;
    for (int idx : win)
      if (passes_gates(F, idx, pr, F->line_inv_level_sigma2[lv], th, p, F->bf / p[4], F->bf / p[5], rr)) {
        C.q.push_back(m);
        C.t.push_back(idx);
      }
  }
  C.first[(size_t)n_map] = (int)C.q.size();
  const int rc = candidate_distances(F, desc, n_map, C, job);
  if (rc != PLVS_OK) return rc;
  std::vector<uint8_t> occ((size_t)F->n, 0);
  if (occupied)
    for (int i = 0; i < F->n; ++i) occ[(size_t)i] = occupied[i];
  int n = 0;
  for (int m = 0; m < n_map; ++m) {
    int best = 256, best2 = 256, lvl = -1, lvl2 = -1, best_idx = -1;
    for (int c = C.first[(size_t)m]; c < C.first[(size_t)m + 1]; ++c) {
      const int idx = C.t[(size_t)c];
      if (occ[(size_t)idx]) continue;
      const int d = C.dist[(size_t)c];
      const int oct = F->keylines_un[idx].octave;
      if (d < best) { best2 = best; best = d; lvl2 = lvl; lvl = oct; best_idx = idx; }
      else if (d < best2) { lvl2 = oct; best2 = d; }
    }
    if (best <= kThHigh && best_idx >= 0) {
      if (lvl == lvl2 && (float)best > nn_ratio * (float)best2) continue;   // ratio only inside one scale level
      assigned[best_idx] = m;
      occ[(size_t)best_idx] = has_obs ? has_obs[m] : 1;
      ++n;
    }
  }
  *nmatches = n;
  return PLVS_OK;
}

}  // extern "C"
