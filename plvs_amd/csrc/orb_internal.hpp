// Read-only view of an ORB extractor's device pyramid for the other translation units of
// the library (mvImagePyramid is a public member of ORBextractor for the same reason:
// Frame::ComputeStereoMatches reads it, src/Frame.cc:1789, 1885-1913).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <functional>

struct plvs_orb;
struct plvs_lines;

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

// The event recorded on the extractor's stream right after the pyramid kernels of the image being
// extracted; a consumer stream waits on it before reading the levels.
hipEvent_t orb_pyramid_event(const plvs_orb* o);
// One-shot hook run by the extracting thread right after that event has been recorded (so another
// host thread knows the event now belongs to the current image).  Cleared when it fires.
void orb_set_pyramid_hook(plvs_orb* o, std::function<void()> hook);

// The extractor whose pyramid a line extractor shares (plvs_hip_lines_set_gaussian_pyramid), or null.
plvs_orb* lines_shared_orb(const plvs_lines* o);

}  // namespace plvs
