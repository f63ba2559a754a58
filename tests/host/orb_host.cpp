// Host compile of the product's quadtree distribution (plvs_amd/csrc/orb_octree.hpp)
// for a CPU-side agreement check against the oracle.  Test infrastructure only.
#include <cstdint>
#include <vector>

#include "../../plvs_amd/csrc/orb_octree.hpp"

extern "C" int hostorb_distribute(const float* xyr, int n, int minX, int maxX, int minY, int maxY,
                                  int N, float* out_xyr, int cap) {
  std::vector<plvs::orb::Cand> c(n);
  for (int i = 0; i < n; ++i) c[i] = plvs::orb::Cand{xyr[3 * i], xyr[3 * i + 1], xyr[3 * i + 2]};
  static plvs::orb::QuadTree scratch;   // as the extractor does: storage kept across calls
  std::vector<plvs::orb::Cand> r = plvs::orb::distribute_quadtree(c, minX, maxX, minY, maxY, N, &scratch);
  for (size_t i = 0; i < r.size() && (int)i < cap; ++i) {
    out_xyr[3 * i] = r[i].x;
    out_xyr[3 * i + 1] = r[i].y;
    out_xyr[3 * i + 2] = r[i].response;
  }
  return (int)r.size();
}
