// Frame-level extraction: the ORB extractor and the line extractor of one image run on
// two host threads, each driving its own stream — what Frame::Frame does with
// threadLeft / threadLines (reference src/Frame.cc:503-508 mono, :290-330 stereo).
#include <atomic>
#include <cstdio>
#include <future>
#include <thread>

#include <condition_variable>
#include <functional>
#include <mutex>

#include "common.hpp"
#include "orb_internal.hpp"

namespace {
// The line thread of a frame, kept between frames (creating and joining a thread per frame costs the caller ~25 us
// before its own extraction starts).  One helper for the process: a second caller that finds it busy (two Frames under
// construction at once) starts a thread of its own, as before.
class LineWorker {
 public:
  ~LineWorker() {
    {
      std::lock_guard<std::mutex> lk(m_);
      quit_ = true;
    }
    go_.notify_all();
    if (th_.joinable()) th_.join();
  }
  bool try_start(const std::function<void()>* job) {
    std::unique_lock<std::mutex> lk(m_);
    if (busy_) return false;
    if (!th_.joinable()) th_ = std::thread([this]() { run(); });
    busy_ = true;
    done_flag_ = false;
    job_ = job;
    lk.unlock();
    go_.notify_all();
    return true;
  }
  void wait() {
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return done_flag_; });
    busy_ = false;
  }

 private:
  void run() {
    for (;;) {
      const std::function<void()>* job;
      {
        std::unique_lock<std::mutex> lk(m_);
        go_.wait(lk, [&] { return quit_ || job_ != nullptr; });
        if (quit_) return;
        job = job_;
        job_ = nullptr;
      }
      (*job)();
      {
        std::lock_guard<std::mutex> lk(m_);
        done_flag_ = true;
      }
      done_.notify_all();
    }
  }
  std::mutex m_;
  std::condition_variable go_, done_;
  std::thread th_;
  const std::function<void()>* job_ = nullptr;
  bool busy_ = false, done_flag_ = false, quit_ = false;
};
LineWorker g_line_worker;
}  // namespace

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
  const std::function<void()> line_job = [&]() {
    if (hipSetDevice(device) != hipSuccess) {
      rc_lines = PLVS_ERR_HIP;
      snprintf(lines_error, sizeof lines_error, "frame: hipSetDevice(%d) failed on the line thread", device);
      return;
    }
    if (shared) pyramid_ready.wait();
    rc_lines = plvs_hip_lines_extract_dev(lines, d_image, w, h, stride, keylines, line_desc, line_cap, n_lines);
    if (rc_lines != PLVS_OK) snprintf(lines_error, sizeof lines_error, "%s", plvs_hip_last_error());   // thread-local
  };
  std::thread own;
  const bool pooled = g_line_worker.try_start(&line_job);
  if (!pooled) own = std::thread(line_job);
  const int rc_orb = plvs_hip_orb_extract_dev(orb, d_image, w, h, stride, lap0, lap1, kps, desc, kp_cap, n_kp,
                                              mono_index);
  if (shared) {
    plvs::orb_set_pyramid_hook(orb, nullptr);
    fire();   // the extractor returned before reaching its pyramid (error paths)
  }
  if (after_points) after_points(user, rc_orb);   // the caller's work on the points, beside the line thread
  if (pooled) g_line_worker.wait(); else own.join();
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
