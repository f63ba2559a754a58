// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints)
// (reference src/ORBmatcher.cc:71-244) and SearchByProjection(CurrentFrame, LastFrame, th, bMono)
// (:1774-1993) for monocular / RGB-D frames, replaced at the
// search-function level: the Hamming distances of every (map point, candidate
// keypoint) pair are computed in one batched launch; the parts that are sequential by
// definition — the candidate windows on the 64x48 frame grid (Frame::GetFeaturesInArea,
// src/Frame.cc:1231-1303) and the greedy best / second-best assignment, in which a keypoint
// claimed by an earlier map point is skipped by the later ones — stay on the host, in the
// reference's order.
#include <cmath>
#include <vector>

#include "common.hpp"

namespace {

constexpr int kGridCols = 64;   // FRAME_GRID_COLS, include/Frame.h:68
constexpr int kGridRows = 48;   // FRAME_GRID_ROWS, include/Frame.h:67
constexpr int kThHigh = 100;    // ORBmatcher::TH_HIGH, src/ORBmatcher.cc:57
constexpr int kThLow = 50;      // ORBmatcher::TH_LOW, :58
constexpr int kHistoLength = 12; // ORBmatcher::HISTO_LENGTH, :61

// dist[p] = ORBmatcher::DescriptorDistance(query[pair_q[p]], train[pair_t[p]]) (src/ORBmatcher.cc:2198)
__global__ __launch_bounds__(256) void hamming_pairs_kernel(const uint4* __restrict__ query,
                                                            const uint4* __restrict__ train,
                                                            const int32_t* __restrict__ pair_q,
                                                            const int32_t* __restrict__ pair_t, int npairs,
                                                            int32_t* __restrict__ dist) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npairs) return;
  const uint4 a0 = query[2 * pair_q[p]], a1 = query[2 * pair_q[p] + 1];
  const uint4 b0 = train[2 * pair_t[p]], b1 = train[2 * pair_t[p] + 1];
  dist[p] = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
            __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// Frame::AssignFeaturesToGrid (src/Frame.cc:717-752): cell lists in keypoint order
struct FrameGrid {
  // mGrid as one array in the reference's enumeration order — column ix, row iy, insertion order inside the cell —
  // with the fields the window test reads next to the index: a window is one contiguous range per column.
  struct Member { float x, y; int octave, idx; };
  std::vector<int> start;
  std::vector<Member> members;
  explicit FrameGrid(const plvs_frame_view* F) : start(kGridCols * kGridRows + 1, 0), members(F->n) {
    std::vector<int> cell(F->n);
    for (int i = 0; i < F->n; ++i) {
      const int px = (int)std::round((F->x[i] - F->min_x) * F->grid_w_inv);   // PosInGrid, :1305-1316
      const int py = (int)std::round((F->y[i] - F->min_y) * F->grid_h_inv);
      cell[i] = (px < 0 || px >= kGridCols || py < 0 || py >= kGridRows) ? -1 : px * kGridRows + py;
      if (cell[i] >= 0) ++start[cell[i] + 1];
    }
    for (int c = 0; c < kGridCols * kGridRows; ++c) start[c + 1] += start[c];
    std::vector<int> fill(start.begin(), start.end() - 1);
    for (int i = 0; i < F->n; ++i)
      if (cell[i] >= 0) members[fill[cell[i]]++] = Member{F->x[i], F->y[i], F->octave[i], i};
  }
  // Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel) (src/Frame.cc:1231-1303), appended to `out`
  // in the reference's order (defaults: minLevel = -1, maxLevel = kMaxInt, include/Frame.h:173)
  template <typename Fn>
  void for_each_in_area(const plvs_frame_view* F, float x, float y, float r, int min_level, int max_level, Fn&& fn) const {
    int c0 = (int)std::floor((x - F->min_x - r) * F->grid_w_inv);
    if (c0 < 0) c0 = 0;
    if (c0 >= kGridCols) return;
    int c1 = (int)std::ceil((x - F->min_x + r) * F->grid_w_inv);
    if (c1 > kGridCols - 1) c1 = kGridCols - 1;
    if (c1 < 0) return;
    int r0 = (int)std::floor((y - F->min_y - r) * F->grid_h_inv);
    if (r0 < 0) r0 = 0;
    if (r0 >= kGridRows) return;
    int r1 = (int)std::ceil((y - F->min_y + r) * F->grid_h_inv);
    if (r1 > kGridRows - 1) r1 = kGridRows - 1;
    if (r1 < 0) return;
    const bool check_levels = (min_level > 0) || (max_level >= 0);
    for (int ix = c0; ix <= c1; ++ix) {
      const Member* m = members.data() + start[ix * kGridRows + r0];
      const Member* const end = members.data() + start[ix * kGridRows + r1 + 1];   // rows r0 .. r1 of the column
      for (; m < end; ++m) {
        if (check_levels && (m->octave < min_level || m->octave > max_level)) continue;
        const float dx = m->x - x, dy = m->y - y;
        if (std::fabs(dx) < r && std::fabs(dy) < r) fn(m->idx);
      }
    }
  }
};

// Frame::GetFeaturesInArea + the stereo gate + the Hamming distance of every candidate, on the device: a wave per
// query walks the query's grid window in the reference's order — column after column, each column's rows one
// contiguous range of the CSR members —, and the accepted candidates leave as (index, distance) in that order (ballot
// compaction keeps it).  Two passes over the window: count, reserve the query's output range, emit.
struct WinQuery {
  float u, v, r, ur;            // window centre, radius, predicted right coordinate
  int min_level, max_level;     // inclusive; tested only when check_levels
  int src;                      // descriptor row of the query
  int check_levels;
};
struct WinFrame {
  const FrameGrid::Member* members;
  const int* start;
  const float* u_right;
  const uint4* desc;            // two uint4 per keypoint
  float min_x, min_y, grid_w_inv, grid_h_inv;
};
__global__ __launch_bounds__(256) void orb_window_candidates(const WinQuery* __restrict__ queries, int nq, WinFrame F,
                                                             const uint4* __restrict__ qdesc, uint32_t* __restrict__ cursor,
                                                             int32_t* __restrict__ q_first, int32_t* __restrict__ q_count,
                                                             int2* __restrict__ out, uint32_t out_cap) {
  const int lane = threadIdx.x & 63;
  const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= nq) return;
  const WinQuery q = queries[qi];
  int c0 = (int)floorf((q.u - F.min_x - q.r) * F.grid_w_inv);
  int c1 = (int)ceilf((q.u - F.min_x + q.r) * F.grid_w_inv);
  int r0 = (int)floorf((q.v - F.min_y - q.r) * F.grid_h_inv);
  int r1 = (int)ceilf((q.v - F.min_y + q.r) * F.grid_h_inv);
  bool empty = c0 >= kGridCols || c1 < 0 || r0 >= kGridRows || r1 < 0;
  c0 = max(c0, 0); c1 = min(c1, kGridCols - 1); r0 = max(r0, 0); r1 = min(r1, kGridRows - 1);
  auto accept = [&](const FrameGrid::Member& m) {
    if (q.check_levels && (m.octave < q.min_level || m.octave > q.max_level)) return false;
    const float dx = m.x - q.u, dy = m.y - q.v;
    if (!(fabsf(dx) < q.r && fabsf(dy) < q.r)) return false;
    const float ur = F.u_right[m.idx];
    if (ur > 0) {
      const float er = fabsf(q.ur - ur);
      if (er > q.r) return false;
    }
    return true;
  };
  uint32_t total = 0;
  if (!empty)
    for (int ix = c0; ix <= c1; ++ix) {
      const int s = F.start[ix * kGridRows + r0], e = F.start[ix * kGridRows + r1 + 1];
      for (int b = s; b < e; b += 64) {
        const bool ok = b + lane < e && accept(F.members[b + lane]);
        total += (uint32_t)__popcll(__ballot(ok));
      }
    }
  uint32_t first = 0;
  if (lane == 0) {
    first = total ? atomicAdd(cursor, total) : 0u;
    q_first[qi] = (int32_t)first;
    q_count[qi] = (int32_t)total;
  }
  first = (uint32_t)__shfl((int)first, 0);
  if (total == 0 || first + total > out_cap) return;   // (the host sees cursor > out_cap and repeats with more room)
  const uint4 qa = qdesc[2 * (size_t)q.src], qb = qdesc[2 * (size_t)q.src + 1];
  uint32_t at = first;
  for (int ix = c0; ix <= c1; ++ix) {
    const int s = F.start[ix * kGridRows + r0], e = F.start[ix * kGridRows + r1 + 1];
    for (int b = s; b < e; b += 64) {
      FrameGrid::Member m{};
      bool ok = false;
      if (b + lane < e) {
        m = F.members[b + lane];
        ok = accept(m);
      }
      const unsigned long long mask = __ballot(ok);
      if (ok) {
        const uint4 ta = F.desc[2 * (size_t)m.idx], tb = F.desc[2 * (size_t)m.idx + 1];
        const int d = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
                      __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
        out[at + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = make_int2(m.idx, d);
      }
      at += (uint32_t)__popcll(mask);
    }
  }
}

// The windows of `wq` on frame F (orb_window_candidates): per query k the accepted candidates pair_t / dist
// [q_at[k], q_at[k] + q_n[k]) in the reference's enumeration order.  qdesc: the queries' descriptor rows (n_qdesc x 32).
int window_candidates(const plvs_frame_view* F, const FrameGrid& grid, const std::vector<WinQuery>& wq, const uint8_t* qdesc,
                      int n_qdesc, std::vector<int32_t>& q_at, std::vector<int32_t>& q_n, std::vector<int32_t>& pair_t,
                      std::vector<int32_t>& dist) {
  const int nq = (int)wq.size();
  q_at.assign((size_t)nq, 0);
  q_n.assign((size_t)nq, 0);
  pair_t.clear();
  dist.clear();
  if (nq == 0) return PLVS_OK;
  auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  const size_t nm = grid.members.size();
  plvs::HostStage& st = plvs::thread_stage();
  size_t out_cap = (size_t)nq * 32 + 4096;
  for (int attempt = 0;; ++attempt) {
    const size_t o_wq = 0, o_mem = o_wq + up16(sizeof(WinQuery) * (size_t)nq), o_start = o_mem + up16(sizeof(FrameGrid::Member) * nm),
                 o_ur = o_start + up16(sizeof(int) * grid.start.size()), o_fd = o_ur + up16(sizeof(float) * (size_t)F->n),
                 o_qd = o_fd + up16((size_t)F->n * 32), o_cur = o_qd + up16((size_t)n_qdesc * 32), o_first = o_cur + 16,
                 o_count = o_first + up16(sizeof(int32_t) * (size_t)nq), o_out = o_count + up16(sizeof(int32_t) * (size_t)nq),
                 total = o_out + up16(sizeof(int2) * out_cap);
    PLVS_HIP_TRY(st.reserve(total));
    memcpy(st.pinned + o_wq, wq.data(), sizeof(WinQuery) * (size_t)nq);
    memcpy(st.pinned + o_mem, grid.members.data(), sizeof(FrameGrid::Member) * nm);
    memcpy(st.pinned + o_start, grid.start.data(), sizeof(int) * grid.start.size());
    memcpy(st.pinned + o_ur, F->u_right, sizeof(float) * (size_t)F->n);
    memcpy(st.pinned + o_fd, F->desc, (size_t)F->n * 32);
    memcpy(st.pinned + o_qd, qdesc, (size_t)n_qdesc * 32);
    memset(st.pinned + o_cur, 0, 16);
    PLVS_HIP_TRY(hipMemcpyAsync(st.dev, st.pinned, o_first, hipMemcpyHostToDevice, st.stream));
    const WinFrame wf{reinterpret_cast<const FrameGrid::Member*>(st.dev + o_mem), reinterpret_cast<const int*>(st.dev + o_start),
                      reinterpret_cast<const float*>(st.dev + o_ur), reinterpret_cast<const uint4*>(st.dev + o_fd),
                      F->min_x, F->min_y, F->grid_w_inv, F->grid_h_inv};
    hipLaunchKernelGGL(orb_window_candidates, dim3(plvs::ceil_div((size_t)nq, 4)), dim3(256), 0, st.stream,
                       reinterpret_cast<const WinQuery*>(st.dev + o_wq), nq, wf, reinterpret_cast<const uint4*>(st.dev + o_qd),
                       reinterpret_cast<uint32_t*>(st.dev + o_cur), reinterpret_cast<int32_t*>(st.pinned + o_first),
                       reinterpret_cast<int32_t*>(st.pinned + o_count), reinterpret_cast<int2*>(st.pinned + o_out), (uint32_t)out_cap);
    // (first / count / records go straight into the pinned block — posted writes over the link, a few tens of KB —:
    // one copy in, one launch, one wait; the copies back and the second wait they needed cost more than the search's
    // arithmetic.  The cursor stays in HBM: it is the target of the waves' atomics.)
    PLVS_KERNEL_CHECK();
    PLVS_HIP_TRY(hipStreamSynchronize(st.stream));
    const int32_t* qf = reinterpret_cast<const int32_t*>(st.pinned + o_first);
    const int32_t* qc = reinterpret_cast<const int32_t*>(st.pinned + o_count);
    size_t found = 0;
    for (int k = 0; k < nq; ++k) found += (size_t)qc[k];
    if (found > out_cap) {   // more candidates than room: once more with what is needed
      PLVS_REQUIRE(attempt == 0, "candidate count changed between two identical launches");
      out_cap = found + 64;
      continue;
    }
    const int2* rec = reinterpret_cast<const int2*>(st.pinned + o_out);
    pair_t.resize(found);
    dist.resize(found);
    // (the ranges were reserved in the order the waves arrived: copied here query after query)
    uint32_t at = 0;
    for (int k = 0; k < nq; ++k) {
      q_at[k] = (int32_t)at;
      q_n[k] = qc[k];
      for (int c = 0; c < qc[k]; ++c) {
        pair_t[at + c] = rec[qf[k] + c].x;
        dist[at + c] = rec[qf[k] + c].y;
      }
      at += (uint32_t)qc[k];
    }
    return PLVS_OK;
  }
}

// ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:2123-2170) on bin counts
void three_maxima(const int* count, int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  ind1 = ind2 = ind3 = -1;
  for (int i = 0; i < L; ++i) {
    const int s = count[i];
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

}  // namespace

constexpr size_t kDirectReadBytes = 96 * 1024;   // inputs up to this size are read in place from pinned host memory

// The pair distances of plvs_hip_hamming_pairs in two steps, for a caller that knows the descriptor sets before it knows
// the pairs (SearchByBoW enumerates them node by node): begin() stages the descriptors and starts their copy, the host
// enumerates, finish() sends the pair lists behind them, launches and waits.  max_pairs bounds the pairs of finish().
struct PairsJob {
  size_t o_pq = 0, o_pt = 0, o_d = 0;
  int nq = 0, nt = 0, max_pairs = 0;
};
int pairs_begin(PairsJob& J, const uint8_t* query, int nq, const uint8_t* train, int nt, int max_pairs) {
  auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  const size_t o_q = 0, o_t = o_q + up16((size_t)nq * 32);
  J.o_pq = o_t + up16((size_t)nt * 32);
  J.o_pt = J.o_pq + up16(sizeof(int32_t) * (size_t)max_pairs);
  J.o_d = J.o_pt + up16(sizeof(int32_t) * (size_t)max_pairs);
  J.nq = nq; J.nt = nt; J.max_pairs = max_pairs;
  plvs::HostStage& st = plvs::thread_stage();
  PLVS_HIP_TRY(st.reserve(J.o_d + up16(sizeof(int32_t) * (size_t)max_pairs)));
  memcpy(st.pinned + o_q, query, (size_t)nq * 32);
  memcpy(st.pinned + o_t, train, (size_t)nt * 32);
  PLVS_HIP_TRY(hipMemcpyAsync(st.dev, st.pinned, J.o_pq, hipMemcpyHostToDevice, st.stream));
  return PLVS_OK;
}
int pairs_finish(PairsJob& J, const int32_t* pair_q, const int32_t* pair_t, int npairs, int32_t* dist) {
  plvs::HostStage& st = plvs::thread_stage();
  if (npairs == 0) {
    PLVS_HIP_TRY(hipStreamSynchronize(st.stream));
    return PLVS_OK;
  }
  PLVS_REQUIRE(npairs <= J.max_pairs, "more pairs than announced");
  memcpy(st.pinned + J.o_pq, pair_q, sizeof(int32_t) * (size_t)npairs);
  memcpy(st.pinned + J.o_pt, pair_t, sizeof(int32_t) * (size_t)npairs);
  // (the two lists lie apart — max_pairs each —: one copy over both when they are nearly full, two otherwise)
  if ((size_t)npairs * 2 >= (size_t)J.max_pairs) {
    PLVS_HIP_TRY(hipMemcpyAsync(st.dev + J.o_pq, st.pinned + J.o_pq, (J.o_pt - J.o_pq) + sizeof(int32_t) * (size_t)npairs, hipMemcpyHostToDevice, st.stream));
  } else {
    PLVS_HIP_TRY(hipMemcpyAsync(st.dev + J.o_pq, st.pinned + J.o_pq, sizeof(int32_t) * (size_t)npairs, hipMemcpyHostToDevice, st.stream));
    PLVS_HIP_TRY(hipMemcpyAsync(st.dev + J.o_pt, st.pinned + J.o_pt, sizeof(int32_t) * (size_t)npairs, hipMemcpyHostToDevice, st.stream));
  }
  hipLaunchKernelGGL(hamming_pairs_kernel, dim3(plvs::ceil_div(npairs, 256)), dim3(256), 0, st.stream,
                     reinterpret_cast<const uint4*>(st.dev), reinterpret_cast<const uint4*>(st.dev + ((size_t)J.nq * 32 + 15) / 16 * 16),
                     reinterpret_cast<const int32_t*>(st.dev + J.o_pq), reinterpret_cast<const int32_t*>(st.dev + J.o_pt), npairs,
                     reinterpret_cast<int32_t*>(st.pinned + J.o_d));
  PLVS_KERNEL_CHECK();
  PLVS_HIP_TRY(hipStreamSynchronize(st.stream));
  memcpy(dist, st.pinned + J.o_d, sizeof(int32_t) * (size_t)npairs);
  return PLVS_OK;
}
struct PairsGuard {   // (an error return between begin and finish must not leave the copy in flight on the staging block)
  bool armed = false;
  ~PairsGuard() { if (armed) (void)hipStreamSynchronize(plvs::thread_stage().stream); }
};

extern "C" {

int plvs_hip_hamming_pairs(const uint8_t* query, int nq, const uint8_t* train, int nt, const int32_t* pair_q,
                           const int32_t* pair_t, int npairs, int32_t* dist) {
  PLVS_REQUIRE(nq >= 0 && nt >= 0 && npairs >= 0, "negative size");
  if (npairs == 0) return PLVS_OK;
  PLVS_REQUIRE(query && train && pair_q && pair_t && dist, "null argument");
  for (int p = 0; p < npairs; ++p)
    PLVS_REQUIRE(pair_q[p] >= 0 && pair_q[p] < nq && pair_t[p] >= 0 && pair_t[p] < nt, "pair index out of range");
  // one staged block: [query | train | pair_q | pair_t] in, [dist] out
  auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  const size_t o_q = 0, o_t = o_q + up16((size_t)nq * 32), o_pq = o_t + up16((size_t)nt * 32),
               o_pt = o_pq + up16(sizeof(int32_t) * (size_t)npairs), o_d = o_pt + up16(sizeof(int32_t) * (size_t)npairs),
               total = o_d + up16(sizeof(int32_t) * (size_t)npairs);
  plvs::HostStage& st = plvs::thread_stage();
  PLVS_HIP_TRY(st.reserve(total));
  memcpy(st.pinned + o_q, query, (size_t)nq * 32);
  memcpy(st.pinned + o_t, train, (size_t)nt * 32);
  memcpy(st.pinned + o_pq, pair_q, sizeof(int32_t) * (size_t)npairs);
  memcpy(st.pinned + o_pt, pair_t, sizeof(int32_t) * (size_t)npairs);
  // The distances go straight into the pinned block (posted writes over the link); a small call — the line searches: a few
  // hundred descriptors, a thousand pairs — reads its inputs from there as well: one launch and one wait instead of a copy
  // in, a launch, a copy back and a wait (60 us for 5 us of arithmetic).
  const bool direct = o_d <= kDirectReadBytes;   // (measured: 6 us saved at 26 KB, 6 us lost at 230 KB)
  char* const in = direct ? st.pinned : st.dev;
  if (!direct) PLVS_HIP_TRY(hipMemcpyAsync(st.dev, st.pinned, o_d, hipMemcpyHostToDevice, st.stream));
  hipLaunchKernelGGL(hamming_pairs_kernel, dim3(plvs::ceil_div(npairs, 256)), dim3(256), 0, st.stream,
                     reinterpret_cast<const uint4*>(in + o_q), reinterpret_cast<const uint4*>(in + o_t),
                     reinterpret_cast<const int32_t*>(in + o_pq), reinterpret_cast<const int32_t*>(in + o_pt), npairs,
                     reinterpret_cast<int32_t*>(st.pinned + o_d));
  PLVS_KERNEL_CHECK();
  PLVS_HIP_TRY(hipStreamSynchronize(st.stream));   // (spinning on hipStreamQuery instead: no difference, measured)
  memcpy(dist, st.pinned + o_d, sizeof(int32_t) * (size_t)npairs);
  return PLVS_OK;
}

int plvs_hip_orb_search_by_projection(const plvs_frame_view* F, const plvs_mappoint_view* M, float th,
                                      int far_points, float th_far, float nn_ratio, const uint8_t* occupied,
                                      int32_t* assigned, int* nmatches) {
  PLVS_REQUIRE(F && M && assigned && nmatches, "null argument");
  PLVS_REQUIRE(F->n >= 0 && M->m >= 0, "negative size");
  *nmatches = 0;
  for (int i = 0; i < F->n; ++i) assigned[i] = -1;
  if (F->n == 0 || M->m == 0) return PLVS_OK;
  PLVS_REQUIRE(F->x && F->y && F->octave && F->u_right && F->desc && F->scale_factors, "null frame array");
  PLVS_REQUIRE(M->track_in_view && M->bad && M->proj_x && M->proj_y && M->proj_xr && M->view_cos &&
                   M->track_depth && M->level && M->desc,
               "null map-point array");
  const FrameGrid grid(F);
  // ---- candidate windows (GetFeaturesInArea) of every map point the reference would process,
  // in the reference's order; everything but the "already claimed" test is decided here
  struct Query { int k; float r_scaled; int first, count; };
  std::vector<WinQuery> wq;
  wq.reserve((size_t)M->m);
  const bool factor = th != 1.0f;
  for (int k = 0; k < M->m; ++k) {
    if (!M->track_in_view[k]) continue;
    if (far_points && M->track_depth[k] > th_far) continue;
    if (M->bad[k]) continue;
    const int level = M->level[k];
    PLVS_REQUIRE(level >= 0, "negative predicted level");
    float r = ((double)M->view_cos[k] > 0.998) ? 2.5f : 4.0f;   // RadiusByViewingCos, :246-252
    if (factor) r *= th;
    const float rr = r * F->scale_factors[level];
    // GetFeaturesInArea(x, y, rr, level - 1, level); a keypoint with a stereo coordinate must agree with the
    // point's predicted one within rr (RGB-D / stereo frames)
    wq.push_back(WinQuery{M->proj_x[k], M->proj_y[k], rr, M->proj_xr[k], level - 1, level, k, (level - 1 > 0 || level >= 0) ? 1 : 0});
  }
  std::vector<Query> queries;
  std::vector<int32_t> pair_t, dist, q_at, q_n;
  int rc = window_candidates(F, grid, wq, M->desc, M->m, q_at, q_n, pair_t, dist);
  if (rc != PLVS_OK) return rc;
  for (size_t k = 0; k < wq.size(); ++k)
    if (q_n[k]) queries.push_back(Query{wq[k].src, wq[k].r, q_at[k], q_n[k]});
  // ---- greedy assignment in map-point order (:106-163)
  std::vector<uint8_t> blocked(F->n);
  for (int i = 0; i < F->n; ++i) blocked[i] = occupied ? occupied[i] : 0;
  int n = 0;
  for (const Query& q : queries) {
    int best = 256, best2 = 256, best_level = -1, best_level2 = -1, best_idx = -1;
    for (int p = q.first; p < q.first + q.count; ++p) {
      const int idx = pair_t[p];
      if (blocked[idx]) continue;   // F.mvpMapPoints[idx] && Observations() > 0
      const int d = dist[p];
      if (d < best) {
        best2 = best; best = d;
        best_level2 = best_level; best_level = F->octave[idx];
        best_idx = idx;
      } else if (d < best2) {
        best_level2 = F->octave[idx];
        best2 = d;
      }
    }
    if (best <= kThHigh) {
      if (best_level == best_level2 && (float)best > nn_ratio * (float)best2) continue;
      if (best_level != best_level2 || (float)best <= nn_ratio * (float)best2) {
        assigned[best_idx] = q.k;
        blocked[best_idx] = M->has_obs ? M->has_obs[q.k] : 1;
        ++n;
      }
    }
  }
  *nmatches = n;
  return PLVS_OK;
}

int plvs_hip_orb_search_by_projection_ff(const plvs_frame_view* F, const float* cur_angle, float max_x, float max_y,
                                         float mbf, const plvs_lastframe_view* L, float th, int forward,
                                         int backward, int check_orientation, const uint8_t* occupied,
                                         int32_t* assigned, int* nmatches) {
  PLVS_REQUIRE(F && L && assigned && nmatches, "null argument");
  PLVS_REQUIRE(F->n >= 0 && L->n >= 0, "negative size");
  *nmatches = 0;
  for (int i = 0; i < F->n; ++i) assigned[i] = -1;
  if (F->n == 0 || L->n == 0) return PLVS_OK;
  PLVS_REQUIRE(F->x && F->y && F->octave && F->u_right && F->desc && F->scale_factors && cur_angle,
               "null frame array");
  PLVS_REQUIRE(L->valid && L->u && L->v && L->invz && L->octave && L->angle && L->desc, "null last-frame array");
  const FrameGrid grid(F);
  // ---- the queries the reference would run (:1800-1840), then their windows and distances on the device
  struct Query { int i; int first, count; };
  std::vector<WinQuery> wq;
  std::vector<int> wq_src;
  wq.reserve((size_t)L->n);
  for (int i = 0; i < L->n; ++i) {
    if (!L->valid[i]) continue;
    const float invzc = L->invz[i];
    if (invzc < 0) continue;
    const float u = L->u[i], v = L->v[i];
    if (u < F->min_x || u > max_x) continue;
    if (v < F->min_y || v > max_y) continue;
    const int oct = L->octave[i];
    PLVS_REQUIRE(oct >= 0, "negative octave");
    const float radius = th * F->scale_factors[oct];   // :1826
    const int min_level = forward ? oct : (backward ? 0 : oct - 1);
    const int max_level = forward ? 2147483647 : (backward ? oct : oct + 1);
    wq.push_back(WinQuery{u, v, radius, u - mbf * invzc, min_level, max_level, i, (min_level > 0 || max_level >= 0) ? 1 : 0});
  }
  std::vector<Query> queries;
  std::vector<int32_t> pair_t, dist, q_at, q_n;
  int rc = window_candidates(F, grid, wq, L->desc, L->n, q_at, q_n, pair_t, dist);
  if (rc != PLVS_OK) return rc;
  for (size_t k = 0; k < wq.size(); ++k)
    if (q_n[k]) queries.push_back(Query{wq[k].src, q_at[k], q_n[k]});
  std::vector<uint8_t> blocked(F->n);
  for (int i = 0; i < F->n; ++i) blocked[i] = occupied ? occupied[i] : 0;
  std::vector<int> hist_item, hist_bin;   // rotHist: what was pushed, in order (duplicates possible)
  const float factor = kHistoLength / 360.0f;
  int n = 0;
  for (const Query& q : queries) {
    int best = 256, best_idx = -1;
    for (int p = q.first; p < q.first + q.count; ++p) {
      const int i2 = pair_t[p];
      if (blocked[i2]) continue;   // CurrentFrame.mvpMapPoints[i2] && Observations() > 0
      if (dist[p] < best) { best = dist[p]; best_idx = i2; }
    }
    if (best <= kThHigh) {
      assigned[best_idx] = q.i;
      blocked[best_idx] = L->has_obs ? L->has_obs[q.i] : 1;
      ++n;
      if (check_orientation) {
        float rot = L->angle[q.i] - cur_angle[best_idx];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == kHistoLength) bin = 0;
        hist_item.push_back(best_idx);
        hist_bin.push_back(bin);
      }
    }
  }
  if (check_orientation) {
    int count[kHistoLength] = {0};
    for (int b : hist_bin) ++count[b];
    int ind1, ind2, ind3;
    three_maxima(count, kHistoLength, ind1, ind2, ind3);
    for (size_t k = 0; k < hist_bin.size(); ++k)
      if (hist_bin[k] != ind1 && hist_bin[k] != ind2 && hist_bin[k] != ind3) {
        assigned[hist_item[k]] = -1;
        --n;
      }
  }
  *nmatches = n;
  return PLVS_OK;
}

int plvs_hip_orb_search_by_bow(const plvs_featvec_view* KV, const uint8_t* kf_desc, int kf_n,
                               const uint8_t* kf_valid, const float* kf_angle, const plvs_featvec_view* FV,
                               const uint8_t* f_desc, int f_n, const float* f_angle, float nn_ratio,
                               int check_orientation, int32_t* assigned, int* nmatches) {
  PLVS_REQUIRE(KV && FV && nmatches, "null argument");
  PLVS_REQUIRE(kf_n >= 0 && f_n >= 0 && KV->nnodes >= 0 && FV->nnodes >= 0, "negative size");
  PLVS_REQUIRE(f_n == 0 || assigned, "assigned is null");
  *nmatches = 0;
  for (int i = 0; i < f_n; ++i) assigned[i] = -1;
  if (kf_n == 0 || f_n == 0 || KV->nnodes == 0 || FV->nnodes == 0) return PLVS_OK;
  PLVS_REQUIRE(KV->node_id && KV->offset && KV->index && FV->node_id && FV->offset && FV->index,
               "null feature-vector array");
  PLVS_REQUIRE(kf_desc && kf_valid && f_desc, "null descriptor / validity array");
  PLVS_REQUIRE(!check_orientation || (kf_angle && f_angle), "angles needed for the orientation check");
  for (int a = 0; a < KV->nnodes; ++a) {
    PLVS_REQUIRE(KV->offset[a] <= KV->offset[a + 1], "key-frame offsets not monotone");
    PLVS_REQUIRE(a == 0 || KV->node_id[a - 1] < KV->node_id[a], "key-frame node ids must ascend (std::map order)");
  }
  for (int b = 0; b < FV->nnodes; ++b) {
    PLVS_REQUIRE(FV->offset[b] <= FV->offset[b + 1], "frame offsets not monotone");
    PLVS_REQUIRE(b == 0 || FV->node_id[b - 1] < FV->node_id[b], "frame node ids must ascend (std::map order)");
  }
  for (int k = KV->offset[0]; k < KV->offset[KV->nnodes]; ++k)
    PLVS_REQUIRE(KV->index[k] < (uint32_t)kf_n, "key-frame feature index out of range");
  for (int k = FV->offset[0]; k < FV->offset[FV->nnodes]; ++k)
    PLVS_REQUIRE(FV->index[k] < (uint32_t)f_n, "frame feature index out of range");

  // Every (key-frame feature, frame feature) pair of every common vocabulary node, in the
  // reference's visiting order; one launch for all distances, then the greedy pass on the host.
  struct Query { int kf; int first, count; };
  std::vector<Query> queries;
  std::vector<int32_t> pair_q, pair_t;
  // the descriptors are on their way to the device while the pairs are enumerated (a walk over the nodes alone bounds them)
  PairsJob job;
  PairsGuard guard;
  {
    long long bound = 0;
    int a0 = 0, b0 = 0;
    while (a0 < KV->nnodes && b0 < FV->nnodes) {
      if (KV->node_id[a0] == FV->node_id[b0]) {
        bound += (long long)(KV->offset[a0 + 1] - KV->offset[a0]) * (FV->offset[b0 + 1] - FV->offset[b0]);
        ++a0; ++b0;
      } else if (KV->node_id[a0] < FV->node_id[b0]) ++a0;
      else ++b0;
    }
    PLVS_REQUIRE(bound < (1ll << 30), "too many candidate pairs");
    if (bound == 0) return PLVS_OK;
    const int rcb = pairs_begin(job, kf_desc, kf_n, f_desc, f_n, (int)bound);
    if (rcb != PLVS_OK) return rcb;
    guard.armed = true;
    pair_q.reserve((size_t)bound);
    pair_t.reserve((size_t)bound);
  }
  int a = 0, b = 0;
  while (a < KV->nnodes && b < FV->nnodes) {   // the lower_bound walk of :327-489
    if (KV->node_id[a] == FV->node_id[b]) {
      for (int ik = KV->offset[a]; ik < KV->offset[a + 1]; ++ik) {
        const int real_kf = (int)KV->index[ik];
        if (!kf_valid[real_kf]) continue;   // !pMP || pMP->isBad()
        Query q{real_kf, (int)pair_q.size(), FV->offset[b + 1] - FV->offset[b]};
        for (int jf = FV->offset[b]; jf < FV->offset[b + 1]; ++jf) {
          pair_q.push_back(real_kf);
          pair_t.push_back((int32_t)FV->index[jf]);
        }
        if (q.count) queries.push_back(q);
      }
      ++a;
      ++b;
    } else if (KV->node_id[a] < FV->node_id[b]) {
      ++a;
    } else {
      ++b;
    }
  }
  std::vector<int32_t> dist(pair_q.size());
  int rc = pairs_finish(job, pair_q.data(), pair_t.data(), (int)pair_q.size(), dist.data());
  guard.armed = false;
  if (rc != PLVS_OK) return rc;
  std::vector<int> hist_item, hist_bin;
  const float factor = kHistoLength / 360.0f;   // USE_NEW_HISTOGRAM_FACTOR, :52, :313
  int n = 0;
  for (const Query& q : queries) {
    int best1 = 256, best_idx = -1, best2 = 256;
    for (int p = q.first; p < q.first + q.count; ++p) {
      const int i_f = pair_t[p];
      if (assigned[i_f] >= 0) continue;   // vpMapPointMatches[realIdxF]
      const int d = dist[p];
      if (d < best1) {
        best2 = best1;
        best1 = d;
        best_idx = i_f;
      } else if (d < best2) {
        best2 = d;
      }
    }
    if (best1 <= kThLow && (float)best1 < nn_ratio * (float)best2) {
      assigned[best_idx] = q.kf;
      ++n;
      if (check_orientation) {
        float rot = kf_angle[q.kf] - f_angle[best_idx];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == kHistoLength) bin = 0;
        hist_item.push_back(best_idx);
        hist_bin.push_back(bin);
      }
    }
  }
  if (check_orientation) {
    int count[kHistoLength] = {0};
    for (int bn : hist_bin) ++count[bn];
    int ind1, ind2, ind3;
    three_maxima(count, kHistoLength, ind1, ind2, ind3);
    for (size_t k = 0; k < hist_bin.size(); ++k)
      if (hist_bin[k] != ind1 && hist_bin[k] != ind2 && hist_bin[k] != ind3) {
        assigned[hist_item[k]] = -1;
        --n;
      }
  }
  *nmatches = n;
  return PLVS_OK;
}

}  // extern "C"
