// Host-side keypoint distribution of the ORB extractor (quadtree culling).
//
// This stage of ORBextractor::DistributeOctTree (src/ORBextractor.cc:611-865,
// ExtractorNode::DivideNode :536-592, compareNodes :594-609) is inherently
// sequential and order dependent: nodes live in a std::list that is grown by
// push-front while it is being walked, the "expand the largest nodes first"
// phase relies on std::sort with a comparator that leaves (count, UL.x) ties
// unordered, and the survivor of a node is the FIRST maximum-response key.  It
// touches a few thousand candidates per level, so it stays on the host (one
// thread per pyramid level) and uses the very same library containers so that
// ties fall exactly as they do in the reference build.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <list>
#include <utility>
#include <vector>

namespace plvs {
namespace orb {

struct Cand {  // FAST survivor, coordinates relative to the level's detection region
  float x, y, response;
};

struct QuadNode {
  std::vector<Cand> keys;
  int ulx = 0, uly = 0, urx = 0, bry = 0;  // the only corner coordinates the algorithm reads
  std::list<QuadNode>::iterator self;
  bool leaf = false;  // bNoMore

  void split(QuadNode out[4]) const {
    const int halfX = (int)std::ceil(static_cast<float>(urx - ulx) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(bry - uly) / 2);
    const int midx = ulx + halfX, midy = uly + halfY;
    // n1 = upper-left, n2 = upper-right, n3 = lower-left, n4 = lower-right
    out[0].ulx = ulx;  out[0].uly = uly;  out[0].urx = midx; out[0].bry = midy;
    out[1].ulx = midx; out[1].uly = uly;  out[1].urx = urx;  out[1].bry = midy;
    out[2].ulx = ulx;  out[2].uly = midy; out[2].urx = midx; out[2].bry = bry;
    out[3].ulx = midx; out[3].uly = midy; out[3].urx = urx;  out[3].bry = bry;
    for (const Cand& k : keys) {
      const int q = (k.x < midx ? 0 : 1) + (k.y < midy ? 0 : 2);
      out[q].keys.push_back(k);
    }
    for (int q = 0; q < 4; ++q) out[q].leaf = out[q].keys.size() == 1;
  }
};

inline bool node_less(std::pair<int, QuadNode*>& a, std::pair<int, QuadNode*>& b) {
  if (a.first < b.first) return true;
  if (a.first > b.first) return false;
  return a.second->ulx < b.second->ulx;
}

// Returns at most ~N keys (one per surviving node), in list order.
inline std::vector<Cand> distribute_quadtree(const std::vector<Cand>& cands, int minX, int maxX,
                                             int minY, int maxY, int N) {
  std::vector<Cand> result;
  const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
  if (nIni == 0) return result;
  const float hX = static_cast<float>(maxX - minX) / nIni;
  std::list<QuadNode> nodes;
  std::vector<QuadNode*> roots(nIni);
  for (int i = 0; i < nIni; ++i) {
    nodes.emplace_back();
    QuadNode* n = &nodes.back();
    n->ulx = (int)(hX * static_cast<float>(i));
    n->urx = (int)(hX * static_cast<float>(i + 1));
    n->uly = 0;
    n->bry = maxY - minY;
    roots[i] = n;
  }
  for (const Cand& k : cands) roots[(int)(k.x / hX)]->keys.push_back(k);
  for (auto it = nodes.begin(); it != nodes.end();) {
    if (it->keys.size() == 1) { it->leaf = true; ++it; }
    else if (it->keys.empty()) it = nodes.erase(it);
    else ++it;
  }
  std::vector<std::pair<int, QuadNode*>> expandable;
  expandable.reserve(nodes.size() * 4);
  auto push_children = [&](QuadNode child[4], int* n_expand) {
    for (int q = 0; q < 4; ++q) {
      if (child[q].keys.empty()) continue;
      nodes.emplace_front(child[q]);
      if (child[q].keys.size() > 1) {
        if (n_expand) ++*n_expand;
        expandable.emplace_back((int)child[q].keys.size(), &nodes.front());
        nodes.front().self = nodes.begin();
      }
    }
  };
  bool finished = false;
  while (!finished) {
    int prev = (int)nodes.size();
    int n_expand = 0;
    expandable.clear();
    for (auto it = nodes.begin(); it != nodes.end();) {
      if (it->leaf) { ++it; continue; }
      QuadNode child[4];
      it->split(child);
      push_children(child, &n_expand);
      it = nodes.erase(it);
    }
    if ((int)nodes.size() >= N || (int)nodes.size() == prev) {
      finished = true;
    } else if ((int)nodes.size() + n_expand * 3 > N) {
      while (!finished) {
        prev = (int)nodes.size();
        std::vector<std::pair<int, QuadNode*>> todo = expandable;
        expandable.clear();
        std::sort(todo.begin(), todo.end(), node_less);
        for (int j = (int)todo.size() - 1; j >= 0; --j) {
          QuadNode child[4];
          todo[j].second->split(child);
          push_children(child, nullptr);
          nodes.erase(todo[j].second->self);
          if ((int)nodes.size() >= N) break;
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prev) finished = true;
      }
    }
  }
  result.reserve(nodes.size());
  for (const QuadNode& n : nodes) {
    const Cand* best = &n.keys[0];
    for (size_t k = 1; k < n.keys.size(); ++k)
      if (n.keys[k].response > best->response) best = &n.keys[k];
    result.push_back(*best);
  }
  return result;
}

}  // namespace orb
}  // namespace plvs
