/*
 * oracle/orb.cpp — CPU restatement of PLVS's ORB extractor (CPU build, i.e. the
 * `#ifndef USE_CUDA` branches).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under plvs_amd/ may call into this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it.
 *
 * Parity status: UNPINNED by the reference (no test, no stored outputs, and the
 * OpenCV primitives it calls are not in the tree) — see cv_primitives.hpp.  C++
 * rather than C because DistributeOctTree's result depends on std::list
 * insertion order and on std::sort (libstdc++ introsort) with a comparator that
 * leaves ties unordered; using the same library calls reproduces that.
 *
 * Follows src/ORBextractor.cc (paths relative to the PLVS tree):
 *   :106-108   PATCH_SIZE 31, HALF_PATCH_SIZE 15, EDGE_THRESHOLD 19
 *   :110-137   IC_Angle            :141-182  computeOrbDescriptor
 *   :446-523   constructor (scale tables, features per level, umax)
 *   :536-592   ExtractorNode::DivideNode      :594-609  compareNodes
 *   :611-865   DistributeOctTree   :867-1052 ComputeKeyPointsOctTree (CPU branch)
 *   :1245-1389 operator()          :1481-1506 ComputePyramid
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <list>
#include <utility>
#include <vector>

#include "cv_primitives.hpp"

using namespace ocv;

namespace {

const int PATCH_SIZE = 31;
const int HALF_PATCH_SIZE = 15;
const int EDGE_THRESHOLD = 19;

const int bit_pattern_31_[256 * 4] = {
#include "orb_pattern.inc"
};

struct KeyPoint {  /* cv::KeyPoint */
  float x = 0, y = 0, size = 0, angle = -1, response = 0;
  int octave = 0, class_id = -1;
};

struct Pt {
  int x = 0, y = 0;
};

struct ExtractorNode {
  std::vector<KeyPoint> vKeys;
  Pt UL, UR, BL, BR;
  std::list<ExtractorNode>::iterator lit;
  bool bNoMore = false;
  void DivideNode(ExtractorNode& n1, ExtractorNode& n2, ExtractorNode& n3, ExtractorNode& n4);
};

/* :536-592 */
void ExtractorNode::DivideNode(ExtractorNode& n1, ExtractorNode& n2, ExtractorNode& n3,
                               ExtractorNode& n4) {
  const int halfX = (int)std::ceil(static_cast<float>(UR.x - UL.x) / 2);
  const int halfY = (int)std::ceil(static_cast<float>(BR.y - UL.y) / 2);
  n1.UL = UL;
  n1.UR = Pt{UL.x + halfX, UL.y};
  n1.BL = Pt{UL.x, UL.y + halfY};
  n1.BR = Pt{UL.x + halfX, UL.y + halfY};
  n2.UL = n1.UR;
  n2.UR = UR;
  n2.BL = n1.BR;
  n2.BR = Pt{UR.x, UL.y + halfY};
  n3.UL = n1.BL;
  n3.UR = n1.BR;
  n3.BL = BL;
  n3.BR = Pt{n1.BR.x, BL.y};
  n4.UL = n3.UR;
  n4.UR = n2.BR;
  n4.BL = n3.BR;
  n4.BR = BR;
  for (size_t i = 0; i < vKeys.size(); i++) {
    const KeyPoint& kp = vKeys[i];
    if (kp.x < n1.UR.x) {
      if (kp.y < n1.BR.y) n1.vKeys.push_back(kp);
      else n3.vKeys.push_back(kp);
    } else if (kp.y < n1.BR.y)
      n2.vKeys.push_back(kp);
    else
      n4.vKeys.push_back(kp);
  }
  if (n1.vKeys.size() == 1) n1.bNoMore = true;
  if (n2.vKeys.size() == 1) n2.bNoMore = true;
  if (n3.vKeys.size() == 1) n3.bNoMore = true;
  if (n4.vKeys.size() == 1) n4.bNoMore = true;
}

/* :594-609 */
bool compareNodes(std::pair<int, ExtractorNode*>& e1, std::pair<int, ExtractorNode*>& e2) {
  if (e1.first < e2.first) return true;
  else if (e1.first > e2.first) return false;
  else return e1.second->UL.x < e2.second->UL.x;
}

struct Extractor {
  int nfeatures, nlevels, iniThFAST, minThFAST;
  float scaleFactor;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
  std::vector<int> mnFeaturesPerLevel, umax;
  std::vector<Image> mvImagePyramid, mvImagePyramidFiltered;
  std::vector<std::vector<KeyPoint>> lastCandidates; /* vToDistributeKeys per level (tests) */

  Extractor(int nf, float sf, int nl, int ini, int mn)
      : nfeatures(nf), nlevels(nl), iniThFAST(ini), minThFAST(mn), scaleFactor(sf) {
    /* :455-470 */
    mvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvScaleFactor[0] = 1.0f;
    mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
      mvScaleFactor[i] = mvScaleFactor[i - 1] * scaleFactor;
      mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
    }
    mvInvScaleFactor.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {
      mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
      mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
    }
    mvImagePyramid.resize(nlevels);
    mvImagePyramidFiltered.resize(nlevels);
    /* :481-493 */
    mnFeaturesPerLevel.resize(nlevels);
    float factor = 1.0f / scaleFactor;
    float nDesiredFeaturesPerScale =
        nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sumFeatures = 0;
    for (int level = 0; level < nlevels - 1; level++) {
      mnFeaturesPerLevel[level] = cv_round(nDesiredFeaturesPerScale);
      sumFeatures += mnFeaturesPerLevel[level];
      nDesiredFeaturesPerScale *= factor;
    }
    mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);
    /* :501-517 umax */
    umax.resize(HALF_PATCH_SIZE + 1);
    int v, v0, vmax = cv_floor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
      while (umax[v0] == umax[v0 + 1]) ++v0;
      umax[v] = v0;
      ++v0;
    }
  }

  /* :1481-1506.  The 19-px REFLECT_101 borders the reference adds around every
   * level are never read by the CPU extraction path (FAST, IC_Angle and the
   * descriptor stay >= 4 px inside; the blur runs on a border-less clone), so
   * only the inner images are kept here. */
  void ComputePyramid(const Image& image) {
    for (int level = 0; level < nlevels; ++level) {
      const float scale = mvInvScaleFactor[level];
      const int sw = cv_round((float)image.w * scale), sh = cv_round((float)image.h * scale);
      if (level != 0) {
        mvImagePyramid[level] = Image(sw, sh);
        resize_linear_u8(mvImagePyramid[level - 1], mvImagePyramid[level]);
      } else {
        mvImagePyramid[level] = image;
      }
    }
  }

  /* :611-865 */
  std::vector<KeyPoint> DistributeOctTree(const std::vector<KeyPoint>& vToDistributeKeys, int minX,
                                          int maxX, int minY, int maxY, int N) {
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    if (nIni == 0) return std::vector<KeyPoint>();
    const float hX = static_cast<float>(maxX - minX) / nIni;
    std::list<ExtractorNode> lNodes;
    std::vector<ExtractorNode*> vpIniNodes(nIni);
    for (int i = 0; i < nIni; i++) {
      lNodes.emplace_back(ExtractorNode());
      vpIniNodes[i] = &lNodes.back();
      vpIniNodes[i]->UL = Pt{(int)(hX * static_cast<float>(i)), 0};
      vpIniNodes[i]->UR = Pt{(int)(hX * static_cast<float>(i + 1)), 0};
      /* the reference fills BL/BR x from a default-constructed node (0); only
       * their y is ever read */
      vpIniNodes[i]->BL = Pt{0, maxY - minY};
      vpIniNodes[i]->BR = Pt{0, maxY - minY};
    }
    for (size_t i = 0; i < vToDistributeKeys.size(); i++) {
      const KeyPoint& kp = vToDistributeKeys[i];
      vpIniNodes[(int)(kp.x / hX)]->vKeys.push_back(kp);
    }
    std::list<ExtractorNode>::iterator lit = lNodes.begin();
    while (lit != lNodes.end()) {
      if (lit->vKeys.size() == 1) { lit->bNoMore = true; lit++; }
      else if (lit->vKeys.empty()) lit = lNodes.erase(lit);
      else lit++;
    }
    bool bFinish = false;
    std::vector<std::pair<int, ExtractorNode*>> vSizeAndPointerToNode;
    vSizeAndPointerToNode.reserve(lNodes.size() * 4);

    auto addChild = [&](ExtractorNode& n, int* nToExpand) {
      if (n.vKeys.size() > 0) {
        lNodes.emplace_front(n);
        if (n.vKeys.size() > 1) {
          if (nToExpand) (*nToExpand)++;
          vSizeAndPointerToNode.emplace_back((int)n.vKeys.size(), &lNodes.front());
          lNodes.front().lit = lNodes.begin();
        }
      }
    };

    while (!bFinish) {
      int prevSize = (int)lNodes.size();
      lit = lNodes.begin();
      int nToExpand = 0;
      vSizeAndPointerToNode.clear();
      while (lit != lNodes.end()) {
        if (lit->bNoMore) { lit++; continue; }
        ExtractorNode n1, n2, n3, n4;
        lit->DivideNode(n1, n2, n3, n4);
        addChild(n1, &nToExpand);
        addChild(n2, &nToExpand);
        addChild(n3, &nToExpand);
        addChild(n4, &nToExpand);
        lit = lNodes.erase(lit);
      }
      if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
        bFinish = true;
      } else if (((int)lNodes.size() + nToExpand * 3) > N) {
        while (!bFinish) {
          prevSize = (int)lNodes.size();
          std::vector<std::pair<int, ExtractorNode*>> vPrev = vSizeAndPointerToNode;
          vSizeAndPointerToNode.clear();
          std::sort(vPrev.begin(), vPrev.end(), compareNodes);
          for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
            ExtractorNode n1, n2, n3, n4;
            vPrev[j].second->DivideNode(n1, n2, n3, n4);
            addChild(n1, nullptr);
            addChild(n2, nullptr);
            addChild(n3, nullptr);
            addChild(n4, nullptr);
            lNodes.erase(vPrev[j].second->lit);
            if ((int)lNodes.size() >= N) break;
          }
          if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
        }
      }
    }
    std::vector<KeyPoint> vResultKeys;
    vResultKeys.reserve(nfeatures);
    for (lit = lNodes.begin(); lit != lNodes.end(); lit++) {
      std::vector<KeyPoint>& vNodeKeys = lit->vKeys;
      KeyPoint* pKP = &vNodeKeys[0];
      float maxResponse = pKP->response;
      for (size_t k = 1; k < vNodeKeys.size(); k++)
        if (vNodeKeys[k].response > maxResponse) {
          pKP = &vNodeKeys[k];
          maxResponse = vNodeKeys[k].response;
        }
      vResultKeys.emplace_back(*pKP);
    }
    return vResultKeys;
  }

  /* :110-137 */
  float IC_Angle(const Image& image, float ptx, float pty) const {
    int m_01 = 0, m_10 = 0;
    const int step = image.w;
    const uint8_t* center = image.row(cv_round(pty)) + cv_round(ptx);
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
      int v_sum = 0;
      const int d = umax[v];
      for (int u = -d; u <= d; ++u) {
        const int val_plus = center[u + v * step], val_minus = center[u - v * step];
        v_sum += (val_plus - val_minus);
        m_10 += u * (val_plus + val_minus);
      }
      m_01 += v * v_sum;
    }
    return fast_atan2((float)m_01, (float)m_10);
  }

  /* :867-1052, CPU branch */
  void ComputeKeyPointsOctTree(std::vector<std::vector<KeyPoint>>& allKeypoints) {
    allKeypoints.assign(nlevels, std::vector<KeyPoint>());
    lastCandidates.assign(nlevels, std::vector<KeyPoint>());
    const float W = 35;
    const int minBorderX = EDGE_THRESHOLD - 3;
    const int minBorderY = minBorderX;
    for (int level = 0; level < nlevels; ++level) {
      const Image& img = mvImagePyramid[level];
      const int maxBorderX = img.w - EDGE_THRESHOLD + 3;
      const int maxBorderY = img.h - EDGE_THRESHOLD + 3;
      std::vector<KeyPoint> vToDistributeKeys;
      const float width = (float)(maxBorderX - minBorderX);
      const float height = (float)(maxBorderY - minBorderY);
      if ((width <= 0) || (height <= 0)) continue;
      const int nCols = (int)(width / W);
      const int nRows = (int)(height / W);
      if ((nCols == 0) || (nRows == 0)) continue;
      const int wCell = (int)std::ceil(width / nCols);
      const int hCell = (int)std::ceil(height / nRows);
      for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
          const float iniX = (float)(minBorderX + j * wCell);
          float maxX = iniX + wCell + 6;
          if (iniX >= maxBorderX - 6) continue;
          if (maxX > maxBorderX) maxX = (float)maxBorderX;
          std::vector<FastKp> vKeysCell;
          const int x0 = (int)iniX, x1 = (int)maxX, y0 = (int)iniY, y1 = (int)maxY;
          fast_9_16(img.row(y0) + x0, img.w, x1 - x0, y1 - y0, iniThFAST, true, vKeysCell);
          if (vKeysCell.empty())
            fast_9_16(img.row(y0) + x0, img.w, x1 - x0, y1 - y0, minThFAST, true, vKeysCell);
          for (const FastKp& k : vKeysCell) {
            KeyPoint kp;
            kp.x = k.x + j * wCell;
            kp.y = k.y + i * hCell;
            kp.size = 7.f;
            kp.angle = -1;
            kp.response = k.response;
            vToDistributeKeys.push_back(kp);
          }
        }
      }
      lastCandidates[level] = vToDistributeKeys;
      std::vector<KeyPoint>& keypoints = allKeypoints[level];
      keypoints = DistributeOctTree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                    mnFeaturesPerLevel[level]);
      const int scaledPatchSize = (int)(PATCH_SIZE * mvScaleFactor[level]);
      for (KeyPoint& kp : keypoints) {
        kp.x += minBorderX;
        kp.y += minBorderY;
        kp.octave = level;
        kp.size = (float)scaledPatchSize;
      }
    }
    for (int level = 0; level < nlevels; ++level)
      for (KeyPoint& kp : allKeypoints[level]) kp.angle = IC_Angle(mvImagePyramid[level], kp.x, kp.y);
  }

  /* :141-182 */
  static void computeOrbDescriptor(const KeyPoint& kpt, const Image& img, uint8_t* desc) {
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float angle = (float)kpt.angle * factorPI;
    const float a = cosf(angle), b = sinf(angle); /* `using namespace std` picks the float overloads */
    const uint8_t* center = img.row(cv_round(kpt.y)) + cv_round(kpt.x);
    const int step = img.w;
    const int* pat = bit_pattern_31_;
    auto get = [&](int idx) -> int {
      const int px = pat[2 * idx], py = pat[2 * idx + 1];
      return center[cv_round(px * b + py * a) * step + cv_round(px * a - py * b)];
    };
    for (int i = 0; i < 32; ++i, pat += 32) {
      int val = 0;
      for (int bit = 0; bit < 8; ++bit) {
        const int t0 = get(2 * bit), t1 = get(2 * bit + 1);
        val |= (t0 < t1) << bit;
      }
      desc[i] = (uint8_t)val;
    }
  }

  /* :1245-1389; returns monoIndex, -1 on an empty image */
  int extract(const Image& image, int lap0, int lap1, std::vector<KeyPoint>& _keypoints,
              std::vector<uint8_t>& descriptors) {
    if (image.w == 0 || image.h == 0) return -1;
    ComputePyramid(image);
    std::vector<std::vector<KeyPoint>> allKeypoints;
    ComputeKeyPointsOctTree(allKeypoints);
    int nkeypoints = 0;
    for (int level = 0; level < nlevels; ++level) nkeypoints += (int)allKeypoints[level].size();
    descriptors.assign((size_t)nkeypoints * 32, 0);
    _keypoints.assign(nkeypoints, KeyPoint());
    int monoIndex = 0, stereoIndex = nkeypoints - 1;
    for (int level = 0; level < nlevels; ++level) {
      std::vector<KeyPoint>& keypoints = allKeypoints[level];
      const int nkeypointsLevel = (int)keypoints.size();
      if (nkeypointsLevel == 0) continue;
      gaussian_blur_u8(mvImagePyramid[level], mvImagePyramidFiltered[level], 7, 2.0);
      std::vector<uint8_t> desc((size_t)nkeypointsLevel * 32);
      for (int i = 0; i < nkeypointsLevel; i++)
        computeOrbDescriptor(keypoints[i], mvImagePyramidFiltered[level], &desc[(size_t)i * 32]);
      const float scale = mvScaleFactor[level];
      int i = 0;
      for (KeyPoint& kp : keypoints) {
        if (level != 0) { kp.x *= scale; kp.y *= scale; }
        int dst;
        if (kp.x >= lap0 && kp.x <= lap1) dst = stereoIndex--;
        else dst = monoIndex++;
        _keypoints[dst] = kp;
        memcpy(&descriptors[(size_t)dst * 32], &desc[(size_t)i * 32], 32);
        i++;
      }
    }
    return monoIndex;
  }
};

}  // namespace

extern "C" {

struct oracle_kp {
  float x, y, size, angle, response;
  int32_t octave, class_id;
};

void* oracle_orb_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST) {
  return new Extractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST);
}
void oracle_orb_destroy(void* h) { delete (Extractor*)h; }

int oracle_orb_features_per_level(void* h, int* out) {
  Extractor* e = (Extractor*)h;
  for (int i = 0; i < e->nlevels; i++) out[i] = e->mnFeaturesPerLevel[i];
  return e->nlevels;
}
void oracle_orb_umax(void* h, int* out) {
  Extractor* e = (Extractor*)h;
  for (int i = 0; i <= HALF_PATCH_SIZE; i++) out[i] = e->umax[i];
}

/* image: h rows of `stride` bytes.  kps/desc hold `cap` entries.  *n = number of
 * keypoints (may exceed cap: then nothing is written).  Returns monoIndex or -1. */
int oracle_orb_extract(void* h, const uint8_t* img, int w, int hh, int stride, int lap0, int lap1,
                       oracle_kp* kps, uint8_t* desc, int cap, int* n) {
  Extractor* e = (Extractor*)h;
  Image im(w, hh);
  for (int y = 0; y < hh; y++) memcpy(im.row(y), img + (size_t)y * stride, w);
  std::vector<KeyPoint> k;
  std::vector<uint8_t> d;
  const int mono = e->extract(im, lap0, lap1, k, d);
  if (mono < 0) { *n = 0; return -1; }
  *n = (int)k.size();
  if ((int)k.size() <= cap) {
    for (size_t i = 0; i < k.size(); i++)
      kps[i] = oracle_kp{k[i].x, k[i].y, k[i].size, k[i].angle, k[i].response, k[i].octave, k[i].class_id};
    if (!d.empty()) memcpy(desc, d.data(), d.size());
  }
  return mono;
}

/* Intermediates of the last extract (for stage-by-stage parity tests). */
int oracle_orb_level_size(void* h, int level, int* w, int* hh) {
  Extractor* e = (Extractor*)h;
  *w = e->mvImagePyramid[level].w;
  *hh = e->mvImagePyramid[level].h;
  return 0;
}
void oracle_orb_get_level(void* h, int level, int blurred, uint8_t* out) {
  Extractor* e = (Extractor*)h;
  const Image& im = blurred ? e->mvImagePyramidFiltered[level] : e->mvImagePyramid[level];
  if (!im.d.empty()) memcpy(out, im.d.data(), im.d.size());
}
int oracle_orb_num_candidates(void* h, int level) { return (int)((Extractor*)h)->lastCandidates[level].size(); }
void oracle_orb_get_candidates(void* h, int level, float* xyr) {
  const std::vector<KeyPoint>& c = ((Extractor*)h)->lastCandidates[level];
  for (size_t i = 0; i < c.size(); i++) {
    xyr[3 * i] = c[i].x;
    xyr[3 * i + 1] = c[i].y;
    xyr[3 * i + 2] = c[i].response;
  }
}

/* ---- primitives exposed for unit checks ---------------------------------- */
void oracle_resize_linear_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
  Image s(sw, sh), d(dw, dh);
  memcpy(s.d.data(), src, s.d.size());
  resize_linear_u8(s, d);
  memcpy(dst, d.d.data(), d.d.size());
}
void oracle_gaussian_blur_u8(const uint8_t* src, int w, int h, int ksize, double sigma, uint8_t* dst) {
  Image s(w, h), d;
  memcpy(s.d.data(), src, s.d.size());
  gaussian_blur_u8(s, d, ksize, sigma);
  memcpy(dst, d.d.data(), d.d.size());
}
void oracle_gaussian_kernel_q8(int n, double sigma, int* out) {
  std::vector<int> k = gaussian_kernel_q8(n, sigma);
  for (int i = 0; i < n; i++) out[i] = k[i];
}
float oracle_fast_atan2(float y, float x) { return fast_atan2(y, x); }
/* xyr: up to cap (x, y, response) triples; returns the number of keypoints */
int oracle_fast(const uint8_t* img, int stride, int cols, int rows, int threshold, int nonmax,
                float* xyr, int cap) {
  std::vector<FastKp> k;
  fast_9_16(img, stride, cols, rows, threshold, nonmax != 0, k);
  for (size_t i = 0; i < k.size() && (int)i < cap; i++) {
    xyr[3 * i] = k[i].x;
    xyr[3 * i + 1] = k[i].y;
    xyr[3 * i + 2] = k[i].response;
  }
  return (int)k.size();
}

}  // extern "C"
