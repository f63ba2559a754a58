// Read-only view of an ORB extractor's device pyramid for the other translation units of
// the library (mvImagePyramid is a public member of ORBextractor for the same reason:
// Frame::ComputeStereoMatches reads it, src/Frame.cc:1789, 1885-1913).
#pragma once
#include <cstddef>
#include <cstdint>

struct plvs_orb;

namespace plvs {

constexpr int kMaxOrbLevels = 16;

struct OrbPyramidView {
  int nlevels = 0;
  const uint8_t* level[kMaxOrbLevels] = {};  // pixel (0,0) of the unblurred level (device)
  int w[kMaxOrbLevels] = {}, h[kMaxOrbLevels] = {}, pitch[kMaxOrbLevels] = {};
  float scale[kMaxOrbLevels] = {}, inv_scale[kMaxOrbLevels] = {};
};

// False if the extractor has not processed an image yet (no pyramid).
bool orb_pyramid_view(const plvs_orb* o, OrbPyramidView* v);

}  // namespace plvs
