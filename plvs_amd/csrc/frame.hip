// Frame-level extraction: the ORB extractor and the line extractor of one image run on
// two host threads, each driving its own stream — what Frame::Frame does with
// threadLeft / threadLines (reference src/Frame.cc:503-508 mono, :290-330 stereo).
#include <atomic>
#include <cstdio>
#include <future>
#include <thread>

#include "common.hpp"
#include "orb_internal.hpp"

static int frame_extract(plvs_orb* orb, plvs_lines* lines, const uint8_t* d_image, int w, int h, int stride, int lap0,
                         int lap1, plvs_keypoint* kps, uint8_t* desc, int kp_cap, int* n_kp, int* mono_index,
                         plvs_keyline* keylines, uint8_t* line_desc, int line_cap, int* n_lines,
                         void (*after_points)(void*, int), void* user) {
  PLVS_REQUIRE(orb && lines && n_kp && mono_index && n_lines, "null argument");
  int rc_lines = PLVS_OK;
  char lines_error[512] = "";
  // Line.pyramidPrecomputation (Frame::PrecomputeGaussianPyramid, src/Frame.cc:841-865): the line
  // extractor reads the ORB pyramid.  The reference builds it before starting the two threads;
  // here the line thread starts as soon as the ORB thread has enqueued the pyramid kernels and
  // orders its device work after them through the extractor's event.
  const bool shared = plvs::lines_shared_orb(lines) == orb;
  std::promise<void> pyramid_enqueued;
  std::future<void> pyramid_ready = pyramid_enqueued.get_future();
  std::atomic<bool> fired{false};
  auto fire = [&]() {
    if (!fired.exchange(true)) pyramid_enqueued.set_value();
  };
  if (shared) plvs::orb_set_pyramid_hook(orb, fire);
  int device = 0;
  PLVS_HIP_TRY(hipGetDevice(&device));   // the current device is per thread: the line thread inherits the caller's
  std::thread tl([&]() {
    if (hipSetDevice(device) != hipSuccess) {
      rc_lines = PLVS_ERR_HIP;
      snprintf(lines_error, sizeof lines_error, "frame: hipSetDevice(%d) failed on the line thread", device);
      return;
    }
    if (shared) pyramid_ready.wait();
    rc_lines = plvs_hip_lines_extract_dev(lines, d_image, w, h, stride, keylines, line_desc, line_cap, n_lines);
    if (rc_lines != PLVS_OK) snprintf(lines_error, sizeof lines_error, "%s", plvs_hip_last_error());   // thread-local
  });
  const int rc_orb = plvs_hip_orb_extract_dev(orb, d_image, w, h, stride, lap0, lap1, kps, desc, kp_cap, n_kp,
                                              mono_index);
  if (shared) {
    plvs::orb_set_pyramid_hook(orb, nullptr);
    fire();   // the extractor returned before reaching its pyramid (error paths)
  }
  if (after_points) after_points(user, rc_orb);   // the caller's work on the points, beside the line thread
  tl.join();
  if (rc_orb != PLVS_OK) return rc_orb;
  if (rc_lines != PLVS_OK) plvs::set_error("%s", lines_error);
  return rc_lines;
}

extern "C" int plvs_hip_frame_extract_dev(plvs_orb* orb, plvs_lines* lines, const uint8_t* d_image, int w,
                                          int h, int stride, int lap0, int lap1, plvs_keypoint* kps,
                                          uint8_t* desc, int kp_cap, int* n_kp, int* mono_index,
                                          plvs_keyline* keylines, uint8_t* line_desc, int line_cap,
                                          int* n_lines) {
  return frame_extract(orb, lines, d_image, w, h, stride, lap0, lap1, kps, desc, kp_cap, n_kp, mono_index, keylines,
                       line_desc, line_cap, n_lines, nullptr, nullptr);
}

extern "C" int plvs_hip_frame_extract_dev_hook(plvs_orb* orb, plvs_lines* lines, const uint8_t* d_image, int w,
                                               int h, int stride, int lap0, int lap1, plvs_keypoint* kps,
                                               uint8_t* desc, int kp_cap, int* n_kp, int* mono_index,
                                               plvs_keyline* keylines, uint8_t* line_desc, int line_cap,
                                               int* n_lines, void (*after_points)(void* user, int orb_status),
                                               void* user) {
  return frame_extract(orb, lines, d_image, w, h, stride, lap0, lap1, kps, desc, kp_cap, n_kp, mono_index, keylines,
                       line_desc, line_cap, n_lines, after_points, user);
}
