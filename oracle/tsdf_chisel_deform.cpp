// oracle/tsdf_chisel_deform.cpp — the ORDER half of ChunkManager::Deform's restatement.  TEST INFRASTRUCTURE ONLY
// (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library).
//
// Deform (src/ChunkManager.cpp:918-1017) walks `chunks`, a std::unordered_map<ChunkID, ChunkPtr, ChunkHasher>
// (include/open_chisel/ChunkManager.h:42-56), and where several old voxels land in one new voxel the result depends on
// who comes first.  That order is libstdc++'s: a function of the container's whole insert / erase history.  The
// reference inserts a chunk when a raycast voxel first VISITS it (GetOrCreateChunkAt for every voxel, Chisel.cpp:505,
// :305) and erases the ones no voxel update reached when the call ends (GarbageCollect, Chisel.cpp:574-585, :366-376;
// erasing does not move the others, but the insert may have triggered a rehash).  This file keeps a
// std::unordered_map with the same hash beside an oracle_chisel map (whose own table is order-free), fed by the
// oracle's visit hook, and hands its iteration order to oracle_chisel_deform() (oracle/tsdf_chisel.c).  After a
// deform the container is `newChunks`: a fresh map filled in first-claim order (:981-990), then swapped in.
// Pinned: tests/test_oracle_pinned_chisel_map.py compares maps deformed here with chisel::Chisel::Deform of the
// compiled reference library, voxel by voxel.
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

extern "C" {
struct oracle_chisel;
void oracle_chisel_set_visit_hook(oracle_chisel* o, void (*hook)(void*, const int32_t*), void* ctx);
int oracle_chisel_has_chunk(const oracle_chisel* o, int cx, int cy, int cz);
int oracle_chisel_num_chunks(const oracle_chisel* o);
int oracle_chisel_deform(oracle_chisel* o, const int32_t* order, int n_order, const uint32_t* kfids, const float* Rt,
                         int n_map, int32_t* new_order, int new_cap, int64_t* stats);
}

namespace {

struct Id {
  int32_t x, y, z;
  bool operator==(const Id& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct IdHash {   // ChunkHasher: int * size_t, i.e. the sign-extended coordinate times the prime
  std::size_t operator()(const Id& k) const {
    return ((std::size_t)(int64_t)k.x * 73856093u) ^ ((std::size_t)(int64_t)k.y * 19349663u) ^
           ((std::size_t)(int64_t)k.z * 83492791u);
  }
};
using Shadow = std::unordered_map<Id, bool, IdHash>;

struct Ordered {
  oracle_chisel* map;
  Shadow chunks;
  std::vector<Id> fresh;   // inserted during the running call
};

void on_visit(void* ctx, const int32_t* id) {
  Ordered* h = static_cast<Ordered*>(ctx);
  const Id k{id[0], id[1], id[2]};
  if (h->chunks.find(k) == h->chunks.end()) {
    h->chunks.insert(std::make_pair(k, true));   // CreateChunk, ChunkManager.h:99
    h->fresh.push_back(k);
  }
}

}  // namespace

extern "C" {

// Attach to a (fresh, unsharded) oracle map: from here on its integrate calls report their visits.
void* oracle_chisel_ordered_attach(oracle_chisel* o) {
  Ordered* h = new Ordered{o, {}, {}};
  oracle_chisel_set_visit_hook(o, on_visit, h);
  return h;
}
void oracle_chisel_ordered_detach(void* p) {
  Ordered* h = static_cast<Ordered*>(p);
  oracle_chisel_set_visit_hook(h->map, nullptr, nullptr);
  delete h;
}
// After EVERY integrate call on the map: the garbage collection of Chisel.cpp:574-585.
void oracle_chisel_ordered_end_call(void* p) {
  Ordered* h = static_cast<Ordered*>(p);
  for (const Id& k : h->fresh)
    if (!oracle_chisel_has_chunk(h->map, k.x, k.y, k.z)) h->chunks.erase(k);
  h->fresh.clear();
}
// Chisel::Reset -> ChunkManager::Reset: chunks.clear() (the bucket array stays); call with oracle_chisel_clear.
void oracle_chisel_ordered_reset(void* p) {
  Ordered* h = static_cast<Ordered*>(p);
  h->chunks.clear();
  h->fresh.clear();
}
int oracle_chisel_ordered_size(void* p) { return (int)static_cast<Ordered*>(p)->chunks.size(); }
void oracle_chisel_ordered_order(void* p, int32_t* ids) {
  size_t k = 0;
  for (const auto& kv : static_cast<Ordered*>(p)->chunks) {
    ids[3 * k] = kv.first.x; ids[3 * k + 1] = kv.first.y; ids[3 * k + 2] = kv.first.z;
    ++k;
  }
}
// Chisel::Deform.  kfids strictly increasing, Rt n x 12 (R row-major, t).  stats: 2 x int64 (see oracle_chisel_deform).
int oracle_chisel_ordered_deform(void* p, const uint32_t* kfids, const float* Rt, int n, int64_t* stats) {
  Ordered* h = static_cast<Ordered*>(p);
  std::vector<int32_t> order(3 * h->chunks.size() + 3);
  oracle_chisel_ordered_order(p, order.data());
  // (every voxel could found a chunk of its own)
  const size_t cap = (size_t)oracle_chisel_num_chunks(h->map) * 4096 + 1;
  std::vector<int32_t> fresh(3 * cap);
  const int n_new = oracle_chisel_deform(h->map, order.data(), (int)h->chunks.size(), kfids, Rt, n, fresh.data(),
                                         (int)cap, stats);
  Shadow next;
  for (int i = 0; i < n_new; ++i) next.insert(std::make_pair(Id{fresh[3 * i], fresh[3 * i + 1], fresh[3 * i + 2]}, true));
  h->chunks.swap(next);
  return n_new;
}

}  // extern "C"
