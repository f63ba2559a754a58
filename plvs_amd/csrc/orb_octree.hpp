// Host-side keypoint distribution of the ORB extractor (quadtree culling).
//
// This stage of ORBextractor::DistributeOctTree (src/ORBextractor.cc:611-865,
// ExtractorNode::DivideNode :536-592, compareNodes :594-609) is inherently
// sequential and order dependent: nodes live in a std::list that is grown by
// push-front while it is being walked, the "expand the largest nodes first"
// phase relies on std::sort with a comparator that leaves (count, UL.x) ties
// unordered, and the survivor of a node is the FIRST maximum-response key.  It
// touches a few thousand candidates per level, so it stays on the host.  The
// list order, the split order and the std::sort call (same sequence, same
// comparator outcomes) are the reference's, so ties fall exactly as they do
// in the reference build; only the storage differs (see QuadNode).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

namespace plvs {
namespace orb {

struct Cand {  // FAST survivor, coordinates relative to the level's detection region
  float x, y, response;
};

// A node of the quadtree.  The keys of a node are a contiguous range of one shared pool (a split appends
// the four children's keys, each in the parent's order, and never moves anything), and the nodes form an
// intrusive doubly-linked list over one array — the operations the reference performs on its std::list
// (push_front while walking, erase of the walked node) without a heap allocation per node and per key vector.
struct QuadNode {
  int ulx = 0, uly = 0, urx = 0, bry = 0;  // the only corner coordinates the algorithm reads
  uint32_t kb = 0, ke = 0;                 // keys: pool[kb, ke)
  int prev = -1, next = -1;                // list links (indices into the node array)
  bool leaf = false;                       // bNoMore
};

class QuadTree {
 public:
  std::vector<QuadNode> nodes;
  std::vector<Cand> pool;
  int head = -1, tail = -1, count = 0;

  void reset() {   // keeps the capacity
    nodes.clear();
    pool.clear();
    head = tail = -1;
    count = 0;
  }
  int push_back(const QuadNode& n) {
    const int i = (int)nodes.size();
    nodes.push_back(n);
    nodes[i].prev = tail; nodes[i].next = -1;
    if (tail >= 0) nodes[tail].next = i; else head = i;
    tail = i;
    ++count;
    return i;
  }
  int push_front(const QuadNode& n) {
    const int i = (int)nodes.size();
    nodes.push_back(n);
    nodes[i].prev = -1; nodes[i].next = head;
    if (head >= 0) nodes[head].prev = i; else tail = i;
    head = i;
    ++count;
    return i;
  }
  int erase(int i) {   // returns the next node
    const int p = nodes[i].prev, n = nodes[i].next;
    if (p >= 0) nodes[p].next = n; else head = n;
    if (n >= 0) nodes[n].prev = p; else tail = p;
    --count;
    return n;
  }
  // ExtractorNode::DivideNode (:536-592): children n1 = upper-left, n2 = upper-right, n3 = lower-left,
  // n4 = lower-right; a key goes to the child its coordinates fall in, in the parent's order.
  void split(int i, QuadNode out[4]) {
    const QuadNode n = nodes[i];
    const int halfX = (int)std::ceil(static_cast<float>(n.urx - n.ulx) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(n.bry - n.uly) / 2);
    const int midx = n.ulx + halfX, midy = n.uly + halfY;
    out[0].ulx = n.ulx; out[0].uly = n.uly; out[0].urx = midx;  out[0].bry = midy;
    out[1].ulx = midx;  out[1].uly = n.uly; out[1].urx = n.urx; out[1].bry = midy;
    out[2].ulx = n.ulx; out[2].uly = midy;  out[2].urx = midx;  out[2].bry = n.bry;
    out[3].ulx = midx;  out[3].uly = midy;  out[3].urx = n.urx; out[3].bry = n.bry;
    uint32_t cnt[4] = {0, 0, 0, 0};
    for (uint32_t k = n.kb; k < n.ke; ++k) ++cnt[(pool[k].x < midx ? 0 : 1) + (pool[k].y < midy ? 0 : 2)];
    const uint32_t base = (uint32_t)pool.size();
    pool.resize(base + (n.ke - n.kb));
    uint32_t at[4];
    uint32_t o = base;
    for (int q = 0; q < 4; ++q) {
      out[q].kb = at[q] = o;
      o += cnt[q];
      out[q].ke = o;
      out[q].leaf = cnt[q] == 1;
    }
    for (uint32_t k = n.kb; k < n.ke; ++k) {
      const Cand c = pool[k];
      pool[at[(c.x < midx ? 0 : 1) + (c.y < midy ? 0 : 2)]++] = c;
    }
  }
};

// Returns at most ~N keys (one per surviving node), in list order.
// `scratch` (optional) keeps the node array and key pool across calls: a caller that extracts frame after frame
// passes one per level and pays no allocation.
inline std::vector<Cand> distribute_quadtree(const std::vector<Cand>& cands, int minX, int maxX,
                                             int minY, int maxY, int N, QuadTree* scratch = nullptr) {
  std::vector<Cand> result;
  const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
  if (nIni == 0) return result;
  const float hX = static_cast<float>(maxX - minX) / nIni;
  QuadTree local;
  QuadTree& T = scratch ? *scratch : local;
  T.reset();
  T.nodes.reserve(8 * (size_t)std::max(N, 1) + 64);
  T.pool.reserve(8 * cands.size() + 64);
  // the initial nodes and their keys (counting sort by root, stable)
  std::vector<uint32_t> first(nIni + 1, 0);
  for (const Cand& k : cands) ++first[(int)(k.x / hX) + 1];
  for (int i = 0; i < nIni; ++i) first[i + 1] += first[i];
  T.pool.resize(cands.size());
  {
    std::vector<uint32_t> at(first.begin(), first.end() - 1);
    for (const Cand& k : cands) T.pool[at[(int)(k.x / hX)]++] = k;
  }
  for (int i = 0; i < nIni; ++i) {
    QuadNode n;
    n.ulx = (int)(hX * static_cast<float>(i));
    n.urx = (int)(hX * static_cast<float>(i + 1));
    n.uly = 0;
    n.bry = maxY - minY;
    n.kb = first[i]; n.ke = first[i + 1];
    T.push_back(n);
  }
  for (int it = T.head; it >= 0;) {
    const uint32_t sz = T.nodes[it].ke - T.nodes[it].kb;
    if (sz == 1) { T.nodes[it].leaf = true; it = T.nodes[it].next; }
    else if (sz == 0) it = T.erase(it);
    else it = T.nodes[it].next;
  }
  // (key count, node): the reference's vSizeAndPointerToNode; compareNodes reads the count and UL.x
  std::vector<std::pair<int, int>> expandable, todo;
  auto node_less = [&](const std::pair<int, int>& a, const std::pair<int, int>& b) {
    if (a.first < b.first) return true;
    if (a.first > b.first) return false;
    return T.nodes[a.second].ulx < T.nodes[b.second].ulx;
  };
  auto push_children = [&](const QuadNode child[4], int* n_expand) {
    for (int q = 0; q < 4; ++q) {
      const int sz = (int)(child[q].ke - child[q].kb);
      if (sz == 0) continue;
      const int id = T.push_front(child[q]);
      if (sz > 1) {
        if (n_expand) ++*n_expand;
        expandable.emplace_back(sz, id);
      }
    }
  };
  bool finished = false;
  while (!finished) {
    int prev = T.count;
    int n_expand = 0;
    expandable.clear();
    for (int it = T.head; it >= 0;) {
      if (T.nodes[it].leaf) { it = T.nodes[it].next; continue; }
      QuadNode child[4];
      T.split(it, child);
      push_children(child, &n_expand);
      it = T.erase(it);
    }
    if (T.count >= N || T.count == prev) {
      finished = true;
    } else if (T.count + n_expand * 3 > N) {
      while (!finished) {
        prev = T.count;
        todo = expandable;
        expandable.clear();
        std::sort(todo.begin(), todo.end(), node_less);
        for (int j = (int)todo.size() - 1; j >= 0; --j) {
          QuadNode child[4];
          T.split(todo[j].second, child);
          push_children(child, nullptr);
          T.erase(todo[j].second);
          if (T.count >= N) break;
        }
        if (T.count >= N || T.count == prev) finished = true;
      }
    }
  }
  result.reserve(T.count);
  for (int it = T.head; it >= 0; it = T.nodes[it].next) {
    const QuadNode& n = T.nodes[it];
    const Cand* best = &T.pool[n.kb];
    for (uint32_t k = n.kb + 1; k < n.ke; ++k)
      if (T.pool[k].response > best->response) best = &T.pool[k];
    result.push_back(*best);
  }
  return result;
}

}  // namespace orb
}  // namespace plvs
