// The one exchange step of the sharded TSDF path, behind the C ABI (BASELINE north star: "RCCL all-gather
// of updated block lists over xGMI"): every rank contributes the ids of the blocks its
// last integrate call updated, every rank receives all lists and folds them into its copy of the GLOBAL
// block directory (block id -> owner rank) — what a C++ PLVS host needs to know which blocks exist on which
// GPU and to schedule meshing.  New design: the reference is single-process.
//
// RCCL is resolved at run time (the symbols of the process if RCCL is already loaded — e.g. by torch —,
// librccl.so.1 otherwise): the library carries no link-time dependency on it.
#include <atomic>
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <vector>

#include "common.hpp"
#include "tsdf_chisel_core.hpp"
#include "tsdf_directory.hpp"

namespace {

using plvs::tsdf::kEmptyKey;
using plvs::tsdf::pack_block;

// the RCCL entry points used (rccl.h: ncclResult_t = int, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4)
using AllGatherFn = int (*)(const void*, void*, size_t, int, void*, hipStream_t);
using SendFn = int (*)(const void*, size_t, int, int, void*, hipStream_t);
using RecvFn = int (*)(void*, size_t, int, int, void*, hipStream_t);
using GroupFn = int (*)();
using CommCountFn = int (*)(void*, int*);
using CommAbortFn = int (*)(void*);
using ErrStrFn = const char* (*)(int);
struct Rccl {
  AllGatherFn all_gather = nullptr;
  SendFn send = nullptr;
  RecvFn recv = nullptr;
  GroupFn group_start = nullptr, group_end = nullptr;
  CommCountFn comm_count = nullptr, comm_rank = nullptr;
  CommAbortFn comm_abort = nullptr;
  ErrStrFn err = nullptr;
};

// Once the counts of a sharded step have been exchanged every peer is committed to the collectives that follow with
// the sizes this rank announced.  From there on a local error must not return past a collective the peers will enter:
//   soft errors (a HIP call, packing, applying) are NOTED and the rank keeps going through every collective with the
//     announced sizes (the data it sends is then unspecified; its return value and message say the map is inconsistent);
//   hard errors (the exchange buffers cannot be allocated, RCCL itself returned an error) leave no way to honour the
//     announced sizes.  The communicator is the CALLER's (torch's ProcessGroup owns it in plvs_amd/shard.py and bench.py): the
//     library does not touch it — an aborted communicator that its owner later uses or destroys is undefined behaviour, and
//     a local ncclCommAbort does not reliably release remote ranks that are already inside the group call either.  The rank
//     returns PLVS_ERR_COMM_FATAL: the caller must abort (or destroy) the communicator on EVERY rank — the peers of this
//     step are inside collectives this rank will not enter and leave them through their own abort or RCCL's watchdog —
//     and clear or rebuild the sharded map.  A caller that owns its communicator outright can opt in to the old behaviour
//     (plvs_hip_exchange_abort_on_fatal(1): ncclCommAbort before returning, which fails this rank's side at once).
struct LateErrors {
  int rc = PLVS_OK;          // the first error noted
  bool hard = false;
  char msg[400] = {0};
  void note(int e, bool is_hard) {
    if (e == PLVS_OK) return;
    if (rc == PLVS_OK) {
      rc = e;
      snprintf(msg, sizeof msg, "%s", plvs::last_error_buf());
    }
    hard = hard || is_hard;
  }
  void note_hip(hipError_t e, const char* what, bool is_hard) {
    if (e == hipSuccess) return;
    plvs::set_error("%s failed: %s", what, hipGetErrorString(e));
    note(PLVS_ERR_HIP, is_hard);
  }
  void note_rccl(const Rccl* r, int e, const char* what) {
    if (e == 0) return;
    plvs::set_error("%s failed: %s", what, r->err ? r->err(e) : "?");
    note(PLVS_ERR_HIP, true);
  }
};

std::atomic<int> g_abort_on_fatal{0};

int abort_exchange(const Rccl* r, void* comm, const LateErrors& L) {
  const bool opted = g_abort_on_fatal.load() != 0;
  const bool aborted = opted && r->comm_abort != nullptr && r->comm_abort(comm) == 0;
  plvs::set_error("%s — after the counts of the sharded step had been exchanged and with no way to go through its "
                  "collectives: %s; other ranks may have applied this step and may still be inside its collectives: abort or "
                  "destroy the communicator on every rank, then clear or rebuild the sharded map with a new one", L.msg,
                  aborted ? "this rank's communicator was aborted (plvs_hip_exchange_abort_on_fatal)"
                          : "the communicator was left as it is (it is the caller's)");
  return PLVS_ERR_COMM_FATAL;
}

const Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    void* h = RTLD_DEFAULT;
    if (dlsym(h, "ncclAllGather") == nullptr) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (h == nullptr) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (h != nullptr) {
      r.all_gather = reinterpret_cast<AllGatherFn>(dlsym(h, "ncclAllGather"));
      r.comm_count = reinterpret_cast<CommCountFn>(dlsym(h, "ncclCommCount"));
      r.comm_rank = reinterpret_cast<CommCountFn>(dlsym(h, "ncclCommUserRank"));
      r.err = reinterpret_cast<ErrStrFn>(dlsym(h, "ncclGetErrorString"));
      r.send = reinterpret_cast<SendFn>(dlsym(h, "ncclSend"));
      r.recv = reinterpret_cast<RecvFn>(dlsym(h, "ncclRecv"));
      r.group_start = reinterpret_cast<GroupFn>(dlsym(h, "ncclGroupStart"));
      r.group_end = reinterpret_cast<GroupFn>(dlsym(h, "ncclGroupEnd"));
      r.comm_abort = reinterpret_cast<CommAbortFn>(dlsym(h, "ncclCommAbort"));
    }
  }
  return (r.all_gather && r.comm_count && r.comm_rank) ? &r : nullptr;
}

// Global directory: open addressing, packed block id -> owner rank.
__global__ void directory_merge(const int32_t* __restrict__ all_ids, const int32_t* __restrict__ counts, int world,
                                int cap, unsigned long long* __restrict__ keys, int32_t* __restrict__ owner,
                                uint32_t mask, uint32_t* __restrict__ nblocks, uint32_t* __restrict__ err) {
  const int r = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= world || i >= min(counts[r], cap)) return;
  const int32_t* id = all_ids + ((size_t)r * cap + i) * 3;
  unsigned long long key;
  if (!pack_block(id[0], id[1], id[2], &key)) {
    atomicOr(err, plvs::tsdf::kErrCoordRange);
    return;
  }
  uint32_t h = plvs::tsdf::dir_hash(id[0], id[1], id[2], mask);
  for (uint32_t probe = 0; probe <= mask; ++probe) {
    const unsigned long long cur = atomicCAS(&keys[h], kEmptyKey, key);
    if (cur == kEmptyKey) {
      owner[h] = r;
      atomicAdd(nblocks, 1u);
      return;
    }
    if (cur == key) return;
    h = (h + 1) & mask;
  }
  atomicOr(err, plvs::tsdf::kErrPoolFull);
}

__global__ void directory_list(const unsigned long long* __restrict__ keys, const int32_t* __restrict__ owner, uint32_t capacity,
                               int32_t* __restrict__ ids, int32_t* __restrict__ owners, int cap_out,
                               uint32_t* __restrict__ n) {
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= capacity || keys[h] == kEmptyKey) return;
  const uint32_t at = atomicAdd(n, 1u);
  if ((int)at >= cap_out) return;
  const unsigned long long k = keys[h];
  ids[3 * at + 0] = (int)((k >> 42) & 0x1FFFFFu) - plvs::tsdf::kCoordBias;
  ids[3 * at + 1] = (int)((k >> 21) & 0x1FFFFFu) - plvs::tsdf::kCoordBias;
  ids[3 * at + 2] = (int)(k & 0x1FFFFFu) - plvs::tsdf::kCoordBias;
  owners[at] = owner[h];
}

}  // namespace

struct plvs_block_directory {
  unsigned long long* keys = nullptr;
  int32_t* owner = nullptr;
  uint32_t* ctr = nullptr;   // [0] blocks, [1] error bits, [2] list cursor
  uint32_t mask = 0;
};

namespace {
__global__ void store_word(int32_t* p, int32_t v) { *p = v; }
}  // namespace

extern "C" {

int plvs_hip_exchange_abort_on_fatal(int enable) {
  g_abort_on_fatal.store(enable ? 1 : 0);
  return PLVS_OK;
}


int plvs_hip_block_directory_create(int max_blocks, plvs_block_directory** out) {
  PLVS_REQUIRE(out && max_blocks > 0 && max_blocks <= (1 << 24), "bad arguments");
  plvs_block_directory* d = new plvs_block_directory();
  uint32_t cap = 1024;
  while (cap < 2u * (uint32_t)max_blocks) cap <<= 1;
  d->mask = cap - 1;
  if (hipMalloc((void**)&d->keys, (size_t)cap * 8) != hipSuccess || hipMalloc((void**)&d->owner, (size_t)cap * 4) != hipSuccess ||
      hipMalloc((void**)&d->ctr, 3 * sizeof(uint32_t)) != hipSuccess) {
    plvs::set_error("block directory: out of device memory");
    (void)hipFree(d->keys); (void)hipFree(d->owner); (void)hipFree(d->ctr);
    delete d;
    return PLVS_ERR_HIP;
  }
  PLVS_HIP_TRY(hipMemset(d->keys, 0xFF, (size_t)cap * 8));
  PLVS_HIP_TRY(hipMemset(d->ctr, 0, 3 * sizeof(uint32_t)));
  *out = d;
  return PLVS_OK;
}

int plvs_hip_block_directory_destroy(plvs_block_directory* d) {
  if (!d) return PLVS_OK;
  (void)hipFree(d->keys); (void)hipFree(d->owner); (void)hipFree(d->ctr);
  delete d;
  return PLVS_OK;
}

int plvs_hip_block_directory_merge(plvs_block_directory* d, const int32_t* d_all_ids, const int32_t* d_counts, int world,
                                   int cap, void* stream) {
  PLVS_REQUIRE(d && d_all_ids && d_counts && world > 0 && cap > 0, "bad arguments");
  hipLaunchKernelGGL(directory_merge, dim3(plvs::ceil_div((size_t)cap, 256), (unsigned)world), dim3(256), 0,
                     static_cast<hipStream_t>(stream), d_all_ids, d_counts, world, cap, d->keys, d->owner, d->mask, d->ctr,
                     d->ctr + 1);
  PLVS_KERNEL_CHECK();
  return PLVS_OK;
}

int plvs_hip_block_directory_count(plvs_block_directory* d, int* n) {
  PLVS_REQUIRE(d && n, "null argument");
  uint32_t c[2] = {0, 0};
  PLVS_HIP_TRY(hipMemcpy(c, d->ctr, sizeof(c), hipMemcpyDeviceToHost));
  if (c[1]) {
    plvs::set_error("block directory: %s", (c[1] & plvs::tsdf::kErrPoolFull) ? "full" : "block id out of range");
    return PLVS_ERR_CAPACITY;
  }
  *n = (int)c[0];
  return PLVS_OK;
}

int plvs_hip_block_directory_list(plvs_block_directory* d, int32_t* ids_xyz, int32_t* owners, int cap, int* n) {
  PLVS_REQUIRE(d && n && cap >= 0 && (cap == 0 || (ids_xyz && owners)), "bad arguments");
  int32_t *di = nullptr, *dow = nullptr;
  PLVS_HIP_TRY(hipMemset(d->ctr + 2, 0, sizeof(uint32_t)));
  PLVS_HIP_TRY(hipMalloc((void**)&di, (size_t)(cap > 0 ? cap : 1) * 3 * sizeof(int32_t)));
  PLVS_HIP_TRY(hipMalloc((void**)&dow, (size_t)(cap > 0 ? cap : 1) * sizeof(int32_t)));
  hipLaunchKernelGGL(directory_list, dim3(plvs::ceil_div((size_t)d->mask + 1, 256)), dim3(256), 0, nullptr, d->keys, d->owner,
                     d->mask + 1, di, dow, cap, d->ctr + 2);
  uint32_t got = 0;
  hipError_t e = hipMemcpy(&got, d->ctr + 2, sizeof(uint32_t), hipMemcpyDeviceToHost);
  const int m = (int)got < cap ? (int)got : cap;
  if (e == hipSuccess && m > 0) e = hipMemcpy(ids_xyz, di, (size_t)m * 3 * sizeof(int32_t), hipMemcpyDeviceToHost);
  if (e == hipSuccess && m > 0) e = hipMemcpy(owners, dow, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost);
  (void)hipFree(di);
  (void)hipFree(dow);
  PLVS_HIP_TRY(e);
  *n = (int)got;
  return PLVS_OK;
}

int plvs_hip_tsdf_exchange_block_lists(void* rccl_comm, const int32_t* d_local_ids, int local_count, int cap,
                                       int32_t* d_all_ids, int32_t* d_all_counts, void* stream) {
  PLVS_REQUIRE(rccl_comm && d_local_ids && d_all_ids && d_all_counts && cap > 0, "bad arguments");
  PLVS_REQUIRE(local_count >= 0 && local_count <= cap, "the local list does not fit the per-rank capacity");
  const Rccl* r = rccl();
  if (r == nullptr) {
    plvs::set_error("RCCL is not available in this process (ncclAllGather / librccl.so.1 not found)");
    return PLVS_ERR_NO_DEVICE;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  // two fixed-size collectives: the counts, then the padded lists (a few KB per rank: latency-bound on xGMI)
  int rank = 0;
  if (r->comm_rank(rccl_comm, &rank) != 0) {
    plvs::set_error("ncclCommUserRank failed");
    return PLVS_ERR_HIP;
  }
  // in place: this rank's slot of the gathered counts (the value travels as a kernel argument: an asynchronous copy
  // from a stack variable could outlive this function)
  hipLaunchKernelGGL(store_word, dim3(1), dim3(1), 0, s, d_all_counts + rank, (int32_t)local_count);
  int rc = r->all_gather(d_all_counts + rank, d_all_counts, 1, /*ncclInt32*/ 2, rccl_comm, s);
  if (rc == 0) rc = r->all_gather(d_local_ids, d_all_ids, (size_t)cap * 3, /*ncclInt32*/ 2, rccl_comm, s);
  if (rc != 0) {
    plvs::set_error("ncclAllGather failed: %s", r->err ? r->err(rc) : "?");
    return PLVS_ERR_HIP;
  }
  return PLVS_OK;
}

// The ray-sharded integrate with its exchanges over RCCL (tsdf_shard.hpp): shard_walk, the all-to-all of the
// counts, shard_pack, the all-to-all of descriptors / voxel sums / runs (grouped ncclSend / ncclRecv pairs, uint32
// words), shard_apply, and the all-gather of the voxels whose colour saturated in the call.
int plvs_hip_tsdf_chisel_integrate_sharded(plvs_tsdf_chisel* h, void* rccl_comm, const float* d_xyz, const uint8_t* d_rgb,
                                           const uint32_t* d_kfid, const int32_t* offsets, int nclouds, const float* d_Twc,
                                           void* stream) {
  PLVS_REQUIRE(h && rccl_comm, "null argument");
  const Rccl* r = rccl();
  if (r == nullptr || !r->send || !r->recv || !r->group_start || !r->group_end) {
    plvs::set_error("RCCL is not available in this process (ncclSend / ncclRecv / librccl.so.1 not found)");
    return PLVS_ERR_NO_DEVICE;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  int world = 0, rank = 0;
  if (r->comm_count(rccl_comm, &world) != 0 || r->comm_rank(rccl_comm, &rank) != 0 || world < 1 || world > 64) {
    plvs::set_error("bad RCCL communicator (1..64 ranks)");
    return PLVS_ERR_INVALID_ARG;
  }
  struct Scratch {   // grow-only device buffers of this thread
    plvs::DevBuf<long long> cnt;
    plvs::DevBuf<uint32_t> send[3], recv[3];
    plvs::DevBuf<int32_t> sat, all_sat;
  };
  static thread_local Scratch B;
  constexpr size_t kWords[3] = {8, 8, 6};   // uint32 words of a descriptor, a voxel sum, a run record (kWireRun)
  int64_t sc[3 * 64], rcv[3 * 64];
  // A rank whose walk fails must not leave its peers blocked inside a collective: it goes through the exchanges with
  // nothing to send and says so in its counts (-1).  Every rank learns of it in the counts exchange and NOBODY applies
  // anything: the step is void on every rank (the maps stay as they were, the step can be repeated); the failing rank
  // returns its own error, the others PLVS_ERR_HALO.  A failure AFTER the counts exchange (packing, applying, a HIP or
  // RCCL error inside this function) cannot be taken back on the ranks that have applied: the error message says that
  // the sharded map is inconsistent and must be cleared / rebuilt.
  int failed = plvs_hip_tsdf_chisel_shard_walk(h, d_xyz, offsets, nclouds, d_Twc, sc, stream);
  constexpr int kSatRows = 16384;
  const size_t msg_words = 4 * ((size_t)kSatRows + 1);
  if (failed == PLVS_OK) {
    // Every buffer whose size this rank knows BEFORE the counts exchange is reserved before it: running out of
    // memory for them is then an announced failure — the step is void on every rank — and not a rank that cannot go through
    // collectives its peers are committed to.  Only the receive buffers wait for the peers' counts.
    size_t own[3] = {0, 0, 0};
    for (int p = 0; p < world; ++p)
      for (int k = 0; k < 3; ++k) own[k] += (size_t)sc[3 * p + k];
    hipError_t e = hipSuccess;
    for (int k = 0; k < 3 && e == hipSuccess; ++k) e = B.send[k].reserve(own[k] * kWords[k] + 4);
    if (e == hipSuccess) e = B.sat.reserve(msg_words);
    if (e == hipSuccess) e = B.all_sat.reserve(msg_words * world);
    if (e == hipSuccess) e = B.cnt.reserve((size_t)6 * world);
    if (e != hipSuccess) {
      plvs::set_error("sharded integrate: reserving the send buffers failed: %s (announced: the step is void on every rank)",
                      hipGetErrorString(e));
      failed = PLVS_ERR_HIP;
    }
  }
  if (failed != PLVS_OK)
    for (int p = 0; p < world; ++p) sc[3 * p] = -1, sc[3 * p + 1] = 0, sc[3 * p + 2] = 0;
  int rc = PLVS_OK;
  bool peer_failed = false;
#define RCCL_TRY(call)                                                      \
  do {                                                                      \
    const int e_ = (call);                                                  \
    if (e_ != 0) {                                                          \
      plvs::set_error("%s failed: %s", #call, r->err ? r->err(e_) : "?");   \
      return PLVS_ERR_HIP;                                                  \
    }                                                                       \
  } while (0)
  // ---- counts
  PLVS_HIP_TRY(B.cnt.reserve((size_t)6 * world));
  PLVS_HIP_TRY(hipMemcpyAsync(B.cnt.p, sc, (size_t)3 * world * sizeof(long long), hipMemcpyHostToDevice, s));
  RCCL_TRY(r->group_start());
  for (int p = 0; p < world; ++p) {
    RCCL_TRY(r->send(B.cnt.p + 3 * p, 3, /*ncclInt64*/ 4, p, rccl_comm, s));
    RCCL_TRY(r->recv(B.cnt.p + 3 * world + 3 * p, 3, /*ncclInt64*/ 4, p, rccl_comm, s));
  }
  RCCL_TRY(r->group_end());
  PLVS_HIP_TRY(hipMemcpyAsync(rcv, B.cnt.p + 3 * world, (size_t)3 * world * sizeof(long long), hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  for (int p = 0; p < world; ++p) {
    if (rcv[3 * p] < 0) {
      peer_failed = peer_failed || p != rank;
      rcv[3 * p] = rcv[3 * p + 1] = rcv[3 * p + 2] = 0;
    }
    if (sc[3 * p] < 0) sc[3 * p] = 0;
  }
  // ---- payloads.  From here on no error returns before the last collective (LateErrors, above).
#undef RCCL_TRY
  LateErrors L;
  size_t stot[3] = {0, 0, 0}, rtot[3] = {0, 0, 0};
  for (int p = 0; p < world; ++p)
    for (int k = 0; k < 3; ++k) {
      stot[k] += (size_t)sc[3 * p + k];
      rtot[k] += (size_t)rcv[3 * p + k];
    }
  for (int k = 0; k < 3; ++k) {
    L.note_hip(B.send[k].reserve(stot[k] * kWords[k] + 4), "reserving the send buffer", true);
    L.note_hip(B.recv[k].reserve(rtot[k] * kWords[k] + 4), "reserving the receive buffer", true);
  }
  L.note_hip(B.sat.reserve(msg_words), "reserving the saturation message", true);
  L.note_hip(B.all_sat.reserve(msg_words * world), "reserving the gathered saturation messages", true);
  if (L.hard) return abort_exchange(r, rccl_comm, L);
  const bool walked = failed == PLVS_OK;   // (a failed walk was announced: the step is void everywhere, not a late failure)
  if (walked) L.note(plvs_hip_tsdf_chisel_shard_pack(h, B.send[0].p, B.send[1].p, B.send[2].p, stream), false);
  // (the peers already expect this rank's counts: what goes out after a failed pack is unspecified data of the announced size)
  L.note_rccl(r, r->group_start(), "ncclGroupStart");
  {
    size_t so[3] = {0, 0, 0}, ro[3] = {0, 0, 0};
    for (int p = 0; p < world; ++p)
      for (int k = 0; k < 3; ++k) {
        const size_t ns = (size_t)sc[3 * p + k] * kWords[k], nr = (size_t)rcv[3 * p + k] * kWords[k];
        if (ns) L.note_rccl(r, r->send(B.send[k].p + so[k], ns, /*ncclUint32*/ 3, p, rccl_comm, s), "ncclSend");
        if (nr) L.note_rccl(r, r->recv(B.recv[k].p + ro[k], nr, /*ncclUint32*/ 3, p, rccl_comm, s), "ncclRecv");
        so[k] += ns;
        ro[k] += nr;
      }
  }
  L.note_rccl(r, r->group_end(), "ncclGroupEnd");
  if (L.hard) return abort_exchange(r, rccl_comm, L);
  // (a peer announced a failed walk: the step is void — nothing is applied here either, no saturation is reported)
  const auto live = [&]() { return walked && !peer_failed && L.rc == PLVS_OK; };
  if (live()) L.note(plvs_hip_tsdf_chisel_shard_apply(h, B.recv[0].p, B.recv[1].p, B.recv[2].p, rcv, d_rgb, d_kfid, stream), false);
  // ---- voxels whose colour saturated: every rank notes every list.  ONE fixed-size all-gather (up to kSatRows voxels + their
  // number in a last row; a longer list waits in the handle for the next step — the list is advisory: a run sent for a
  // saturated voxel is a no-op at its owner), the lengths read on the device: no host read in the feedback.
  L.note_hip(hipMemsetAsync(B.sat.p + 4 * (size_t)kSatRows, 0, 4 * sizeof(int32_t), s), "hipMemsetAsync", false);   // (a failed rank announces none)
  if (live()) L.note(plvs_hip_tsdf_chisel_shard_saturated_message(h, B.sat.p, kSatRows, stream), false);
  L.note_rccl(r, r->all_gather(B.sat.p, B.all_sat.p, msg_words, /*ncclInt32*/ 2, rccl_comm, s), "ncclAllGather");
  if (L.hard) return abort_exchange(r, rccl_comm, L);
  if (live()) L.note(plvs_hip_tsdf_chisel_shard_note_gathered(h, B.all_sat.p, world, kSatRows, stream), false);
  (void)rc;
  if (!walked) return failed;   // its own walk's error; announced, nothing was applied anywhere
  if (L.rc != PLVS_OK) {   // (keep this rank's own message, add what it means for the job)
    plvs::set_error("%s — after the exchange had begun: other ranks may have applied this step, the sharded map is "
                    "inconsistent (clear or rebuild it on every rank)", L.msg);
    return L.rc;
  }
  if (peer_failed) {
    plvs::set_error("a peer rank failed in its walk of the sharded integrate: nothing was applied on any rank that reports "
                    "this (the step is void and can be repeated once the peer recovers)");
    return PLVS_ERR_HALO;
  }
  return PLVS_OK;
}

// The ray-sharded voxblox integrate with its exchange over RCCL (tsdf_voxblox_shard.hpp): shard_walk, the all-to-all of the
// counts, shard_pack, the all-to-all of the 16-byte visit records (grouped ncclSend / ncclRecv, uint32 words), shard_apply.
// A rank whose walk fails goes through both exchanges with nothing to send and says so in its counts (-1): nobody applies
// anything, the step is void on every rank (the failing rank returns its own error, the others PLVS_ERR_HALO).
int plvs_hip_tsdf_voxblox_integrate_sharded(plvs_tsdf_voxblox* h, void* rccl_comm, const float* d_xyz, const uint8_t* d_rgba,
                                            const int32_t* offsets, int nclouds, const float* d_Twc, void* stream) {
  PLVS_REQUIRE(h && rccl_comm, "null argument");
  const Rccl* r = rccl();
  if (r == nullptr || !r->send || !r->recv || !r->group_start || !r->group_end) {
    plvs::set_error("RCCL is not available in this process (ncclSend / ncclRecv / librccl.so.1 not found)");
    return PLVS_ERR_NO_DEVICE;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  int world = 0, rank = 0;
  if (r->comm_count(rccl_comm, &world) != 0 || r->comm_rank(rccl_comm, &rank) != 0 || world < 1 || world > 64) {
    plvs::set_error("bad RCCL communicator (1..64 ranks)");
    return PLVS_ERR_INVALID_ARG;
  }
  struct Scratch {   // grow-only device buffers of this thread
    plvs::DevBuf<long long> cnt;
    plvs::DevBuf<uint32_t> send, recv;
  };
  static thread_local Scratch B;
  constexpr size_t kWords = 4;   // uint32 words of a visit record (kVbWire)
  int64_t sc[64], rcv[64];
  int failed = plvs_hip_tsdf_voxblox_shard_walk(h, d_xyz, offsets, nclouds, d_Twc, sc, stream);
  if (failed == PLVS_OK) {   // (the send buffer's size is known before the counts exchange: out of memory for it is an announced failure)
    size_t own = 0;
    for (int p = 0; p < world; ++p) own += (size_t)sc[p];
    hipError_t e = B.send.reserve(own * kWords + 4);
    if (e == hipSuccess) e = B.cnt.reserve((size_t)2 * world);
    if (e != hipSuccess) {
      plvs::set_error("sharded integrate: reserving the send buffer failed: %s (announced: the step is void on every rank)",
                      hipGetErrorString(e));
      failed = PLVS_ERR_HIP;
    }
  }
  if (failed != PLVS_OK)
    for (int p = 0; p < world; ++p) sc[p] = -1;
  bool peer_failed = false;
#define RCCL_TRY(call)                                                      \
  do {                                                                      \
    const int e_ = (call);                                                  \
    if (e_ != 0) {                                                          \
      plvs::set_error("%s failed: %s", #call, r->err ? r->err(e_) : "?");   \
      return PLVS_ERR_HIP;                                                  \
    }                                                                       \
  } while (0)
  PLVS_HIP_TRY(B.cnt.reserve((size_t)2 * world));
  PLVS_HIP_TRY(hipMemcpyAsync(B.cnt.p, sc, (size_t)world * sizeof(long long), hipMemcpyHostToDevice, s));
  RCCL_TRY(r->group_start());
  for (int p = 0; p < world; ++p) {
    RCCL_TRY(r->send(B.cnt.p + p, 1, /*ncclInt64*/ 4, p, rccl_comm, s));
    RCCL_TRY(r->recv(B.cnt.p + world + p, 1, /*ncclInt64*/ 4, p, rccl_comm, s));
  }
  RCCL_TRY(r->group_end());
  PLVS_HIP_TRY(hipMemcpyAsync(rcv, B.cnt.p + world, (size_t)world * sizeof(long long), hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  size_t stot = 0, rtot = 0;
  for (int p = 0; p < world; ++p) {
    if (rcv[p] < 0) {
      peer_failed = peer_failed || p != rank;
      rcv[p] = 0;
    }
    if (sc[p] < 0) sc[p] = 0;
    stot += (size_t)sc[p];
    rtot += (size_t)rcv[p];
  }
  // From here on no error returns before the last collective (LateErrors, above).
#undef RCCL_TRY
  LateErrors L;
  L.note_hip(B.send.reserve(stot * kWords + 4), "reserving the send buffer", true);
  L.note_hip(B.recv.reserve(rtot * kWords + 4), "reserving the receive buffer", true);
  if (L.hard) return abort_exchange(r, rccl_comm, L);
  const bool walked = failed == PLVS_OK;   // (a failed walk was announced: the step is void everywhere, not a late failure)
  if (walked) L.note(plvs_hip_tsdf_voxblox_shard_pack(h, B.send.p, stream), false);
  L.note_rccl(r, r->group_start(), "ncclGroupStart");
  {
    size_t so = 0, ro = 0;
    for (int p = 0; p < world; ++p) {
      const size_t ns = (size_t)sc[p] * kWords, nr = (size_t)rcv[p] * kWords;
      if (ns) L.note_rccl(r, r->send(B.send.p + so, ns, /*ncclUint32*/ 3, p, rccl_comm, s), "ncclSend");
      if (nr) L.note_rccl(r, r->recv(B.recv.p + ro, nr, /*ncclUint32*/ 3, p, rccl_comm, s), "ncclRecv");
      so += ns;
      ro += nr;
    }
  }
  L.note_rccl(r, r->group_end(), "ncclGroupEnd");
  if (L.hard) return abort_exchange(r, rccl_comm, L);
  if (walked && !peer_failed && L.rc == PLVS_OK)
    L.note(plvs_hip_tsdf_voxblox_shard_apply(h, B.recv.p, rcv, d_xyz, d_rgba, offsets, nclouds, d_Twc, stream), false);
  if (!walked) return failed;   // its own walk's error; announced, nothing was applied anywhere
  if (L.rc != PLVS_OK) {
    plvs::set_error("%s — after the exchange had begun: other ranks may have applied this step, the sharded map is "
                    "inconsistent and must be cleared / rebuilt", L.msg);
    return L.rc;
  }
  if (peer_failed) {
    plvs::set_error("a peer rank failed in its walk of the sharded integrate: nothing was applied on any rank that reports "
                    "this (the step is void and can be repeated once the peer recovers)");
    return PLVS_ERR_HALO;
  }
  return PLVS_OK;
}

}  // extern "C"

namespace {

// One round of a meshing halo over RCCL, shared by the two back ends: this rank's requests (host ids, any order) go to
// their owners (three-prime hash mod world), the owners look them up, the found flags travel back, then one payload row
// per block that exists; the answers are imported here.  Grouped ncclSend / ncclRecv for every step.
struct HaloOps {
  void* map;
  size_t words;   // uint32 words of a block on the wire
  int (*lookup)(void* map, const int32_t* d_ids, int n, uint32_t* d_found, void* stream);
  int (*pack)(void* map, const int32_t* d_ids, const uint32_t* d_found, int n, uint32_t* d_payload, void* stream);
  int (*import)(void* map, const int32_t* d_ids, const uint32_t* d_found, const uint32_t* d_payload, int n, int nfound,
                void* stream);
};

struct HaloScratch {   // grow-only device buffers of the calling thread
  plvs::DevBuf<int32_t> cnt, req_ids, got_ids;
  plvs::DevBuf<uint32_t> got_found, got_payload, back_found, back_payload;
};

int halo_round(const Rccl* r, void* comm, int world, hipStream_t s, HaloScratch& B, const HaloOps& ops,
               const std::vector<int32_t>& miss, int* imported) {
#define RCCL_TRY(call)                                                      \
  do {                                                                      \
    const int e_ = (call);                                                  \
    if (e_ != 0) {                                                          \
      plvs::set_error("%s failed: %s", #call, r->err ? r->err(e_) : "?");   \
      return PLVS_ERR_HIP;                                                  \
    }                                                                       \
  } while (0)
  void* const stream = static_cast<void*>(s);
  const int nmiss = (int)(miss.size() / 3);
  *imported = 0;
  PLVS_HIP_TRY(B.cnt.reserve((size_t)4 * world + 4));
  std::vector<int32_t> req((size_t)3 * (nmiss > 0 ? nmiss : 1));
  int32_t req_cnt[64] = {0}, got_cnt[64] = {0}, at[64];
  std::vector<int> owner((size_t)(nmiss > 0 ? nmiss : 1));
  for (int i = 0; i < nmiss; ++i) {
    owner[(size_t)i] = plvs::chisel::shard_of(plvs::chisel::chunk_hash(miss[3 * i], miss[3 * i + 1], miss[3 * i + 2]), world);
    ++req_cnt[owner[(size_t)i]];
  }
  at[0] = 0;
  for (int p = 1; p < world; ++p) at[p] = at[p - 1] + req_cnt[p - 1];
  for (int i = 0; i < nmiss; ++i) {
    const int o = at[owner[(size_t)i]]++;
    req[3 * (size_t)o] = miss[3 * i];
    req[3 * (size_t)o + 1] = miss[3 * i + 1];
    req[3 * (size_t)o + 2] = miss[3 * i + 2];
  }
  // ---- request counts
  int32_t* d_req_cnt = B.cnt.p + world;
  int32_t* d_got_cnt = B.cnt.p + 2 * world;
  PLVS_HIP_TRY(hipMemcpyAsync(d_req_cnt, req_cnt, (size_t)world * sizeof(int32_t), hipMemcpyHostToDevice, s));
  RCCL_TRY(r->group_start());
  for (int p = 0; p < world; ++p) {
    RCCL_TRY(r->send(d_req_cnt + p, 1, /*ncclInt32*/ 2, p, comm, s));
    RCCL_TRY(r->recv(d_got_cnt + p, 1, /*ncclInt32*/ 2, p, comm, s));
  }
  RCCL_TRY(r->group_end());
  PLVS_HIP_TRY(hipMemcpyAsync(got_cnt, d_got_cnt, (size_t)world * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  size_t ngot = 0;
  for (int p = 0; p < world; ++p) ngot += (size_t)got_cnt[p];
  // ---- the ids
  PLVS_HIP_TRY(B.req_ids.reserve(3 * (size_t)nmiss + 4));
  PLVS_HIP_TRY(B.got_ids.reserve(3 * ngot + 4));
  if (nmiss > 0)
    PLVS_HIP_TRY(hipMemcpyAsync(B.req_ids.p, req.data(), (size_t)3 * nmiss * sizeof(int32_t), hipMemcpyHostToDevice, s));
  RCCL_TRY(r->group_start());
  {
    size_t so = 0, ro = 0;
    for (int p = 0; p < world; ++p) {
      if (req_cnt[p]) RCCL_TRY(r->send(B.req_ids.p + 3 * so, (size_t)3 * req_cnt[p], /*ncclInt32*/ 2, p, comm, s));
      if (got_cnt[p]) RCCL_TRY(r->recv(B.got_ids.p + 3 * ro, (size_t)3 * got_cnt[p], /*ncclInt32*/ 2, p, comm, s));
      so += (size_t)req_cnt[p];
      ro += (size_t)got_cnt[p];
    }
  }
  RCCL_TRY(r->group_end());
  // ---- owners look the ids up; the flags travel the reverse way, then one payload row per block that exists
  PLVS_HIP_TRY(B.got_found.reserve(ngot + 4));
  PLVS_HIP_TRY(B.back_found.reserve((size_t)nmiss + 4));
  int rc = ops.lookup(ops.map, B.got_ids.p, (int)ngot, B.got_found.p, stream);
  if (rc != PLVS_OK) return rc;
  RCCL_TRY(r->group_start());
  {
    size_t so = 0, ro = 0;
    for (int p = 0; p < world; ++p) {
      if (got_cnt[p]) RCCL_TRY(r->send(B.got_found.p + so, (size_t)got_cnt[p], /*ncclUint32*/ 3, p, comm, s));
      if (req_cnt[p]) RCCL_TRY(r->recv(B.back_found.p + ro, (size_t)req_cnt[p], /*ncclUint32*/ 3, p, comm, s));
      so += (size_t)got_cnt[p];
      ro += (size_t)req_cnt[p];
    }
  }
  RCCL_TRY(r->group_end());
  std::vector<uint32_t> got_found(ngot + 1), back_found((size_t)nmiss + 1);
  if (ngot) PLVS_HIP_TRY(hipMemcpyAsync(got_found.data(), B.got_found.p, ngot * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  if (nmiss) PLVS_HIP_TRY(hipMemcpyAsync(back_found.data(), B.back_found.p, (size_t)nmiss * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  size_t send_rows[64] = {0}, recv_rows[64] = {0}, nsend = 0, nrecv = 0;
  {
    size_t so = 0, ro = 0;
    for (int p = 0; p < world; ++p) {
      for (int i = 0; i < got_cnt[p]; ++i) send_rows[p] += got_found[so + (size_t)i] ? 1 : 0;
      for (int i = 0; i < req_cnt[p]; ++i) recv_rows[p] += back_found[ro + (size_t)i] ? 1 : 0;
      so += (size_t)got_cnt[p];
      ro += (size_t)req_cnt[p];
      nsend += send_rows[p];
      nrecv += recv_rows[p];
    }
  }
  PLVS_HIP_TRY(B.got_payload.reserve(nsend * ops.words + 4));
  PLVS_HIP_TRY(B.back_payload.reserve(nrecv * ops.words + 4));
  rc = ops.pack(ops.map, B.got_ids.p, B.got_found.p, (int)ngot, B.got_payload.p, stream);
  if (rc != PLVS_OK) return rc;
  RCCL_TRY(r->group_start());
  {
    size_t so = 0, ro = 0;
    for (int p = 0; p < world; ++p) {
      if (send_rows[p]) RCCL_TRY(r->send(B.got_payload.p + so * ops.words, send_rows[p] * ops.words, /*ncclUint32*/ 3, p, comm, s));
      if (recv_rows[p]) RCCL_TRY(r->recv(B.back_payload.p + ro * ops.words, recv_rows[p] * ops.words, /*ncclUint32*/ 3, p, comm, s));
      so += send_rows[p];
      ro += recv_rows[p];
    }
  }
  RCCL_TRY(r->group_end());
  rc = ops.import(ops.map, B.req_ids.p, B.back_found.p, B.back_payload.p, nmiss, (int)nrecv, stream);
  if (rc != PLVS_OK) return rc;
  PLVS_HIP_TRY(hipStreamSynchronize(s));
  *imported = (int)nrecv;
  return PLVS_OK;
#undef RCCL_TRY
}

int halo_comm(void* rccl_comm, const Rccl** r_out, int* world, int* rank) {
  const Rccl* r = rccl();
  if (r == nullptr || !r->send || !r->recv || !r->group_start || !r->group_end) {
    plvs::set_error("RCCL is not available in this process (ncclSend / ncclRecv / librccl.so.1 not found)");
    return PLVS_ERR_NO_DEVICE;
  }
  if (r->comm_count(rccl_comm, world) != 0 || r->comm_rank(rccl_comm, rank) != 0 || *world < 1 || *world > 64) {
    plvs::set_error("bad RCCL communicator (1..64 ranks)");
    return PLVS_ERR_INVALID_ARG;
  }
  *r_out = r;
  return PLVS_OK;
}

}  // namespace

extern "C" {

// The meshing halo of a sharded chisel map over RCCL (include/plvs_hip.h: halo_gather): rounds of
//   mesh_probe -> all-gather of the miss counts -> halo_round (requests to the owners, answers back, import)
// until no rank misses anything.  A chunk travels as 64 KiB (four planes of 4096 words).
int plvs_hip_tsdf_chisel_halo_gather(plvs_tsdf_chisel* h, void* rccl_comm, const int32_t* chunk_ids_xyz, int nchunks,
                                     int* fetched, void* stream) {
  PLVS_REQUIRE(h && rccl_comm && fetched, "null argument");
  *fetched = 0;
  const Rccl* r = nullptr;
  int world = 0, rank = 0;
  int rc = halo_comm(rccl_comm, &r, &world, &rank);
  if (rc != PLVS_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  static thread_local HaloScratch B;
  const HaloOps ops{h, (size_t)4 * 4096,
                    [](void* m, const int32_t* ids, int n, uint32_t* f, void* st) {
                      return plvs_hip_tsdf_chisel_halo_lookup(static_cast<plvs_tsdf_chisel*>(m), ids, n, f, st);
                    },
                    [](void* m, const int32_t* ids, const uint32_t* f, int n, uint32_t* p, void* st) {
                      return plvs_hip_tsdf_chisel_halo_export(static_cast<plvs_tsdf_chisel*>(m), ids, f, n, p, st);
                    },
                    [](void* m, const int32_t* ids, const uint32_t* f, const uint32_t* p, int n, int nf, void* st) {
                      return plvs_hip_tsdf_chisel_halo_import(static_cast<plvs_tsdf_chisel*>(m), ids, f, p, n, nf, st);
                    }};
  PLVS_HIP_TRY(B.cnt.reserve((size_t)4 * world + 4));
  for (int round = 0; round < 8; ++round) {
    int nmiss = 0;
    rc = plvs_hip_tsdf_chisel_mesh_probe(h, chunk_ids_xyz, nchunks, &nmiss);
    if (rc != PLVS_OK) return rc;
    // does anybody miss anything?
    int32_t all_miss[64];
    PLVS_HIP_TRY(hipMemcpyAsync(B.cnt.p + rank, &nmiss, sizeof(int32_t), hipMemcpyHostToDevice, s));
    if (r->all_gather(B.cnt.p + rank, B.cnt.p, 1, /*ncclInt32*/ 2, rccl_comm, s) != 0) {
      plvs::set_error("ncclAllGather failed");
      return PLVS_ERR_HIP;
    }
    PLVS_HIP_TRY(hipMemcpyAsync(all_miss, B.cnt.p, (size_t)world * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    PLVS_HIP_TRY(hipStreamSynchronize(s));
    int any = 0;
    for (int p = 0; p < world; ++p) any |= all_miss[p];
    if (any == 0) return PLVS_OK;
    std::vector<int32_t> miss((size_t)3 * (nmiss > 0 ? nmiss : 0));
    if (nmiss > 0) {
      int got_n = 0;
      rc = plvs_hip_tsdf_chisel_halo_missing(h, miss.data(), nmiss, &got_n);
      if (rc != PLVS_OK) return rc;
      miss.resize((size_t)3 * got_n);
    }
    int imported = 0;
    rc = halo_round(r, rccl_comm, world, s, B, ops, miss, &imported);
    if (rc != PLVS_OK) return rc;
    *fetched += imported;
  }
  plvs::set_error("halo_gather: the halo did not settle in 8 rounds");
  return PLVS_ERR_CAPACITY;
}

// The meshing halo of a block-sharded voxblox map over RCCL: the blocks to fetch are known (the seven +x / +y / +z
// neighbours of every block of the list that another rank owns), so it is ONE halo_round.  A block travels as 48 KiB.
int plvs_hip_tsdf_voxblox_halo_gather(plvs_tsdf_voxblox* h, void* rccl_comm, const int32_t* block_ids_xyz, int nblocks,
                                      int* fetched, void* stream) {
  PLVS_REQUIRE(h && rccl_comm && fetched && nblocks >= 0 && (nblocks == 0 || block_ids_xyz), "bad arguments");
  *fetched = 0;
  const Rccl* r = nullptr;
  int world = 0, rank = 0;
  int rc = halo_comm(rccl_comm, &r, &world, &rank);
  if (rc != PLVS_OK) return rc;
  static thread_local HaloScratch B;
  const HaloOps ops{h, (size_t)3 * 4096,
                    [](void* m, const int32_t* ids, int n, uint32_t* f, void* st) {
                      return plvs_hip_tsdf_voxblox_halo_lookup(static_cast<plvs_tsdf_voxblox*>(m), ids, n, f, st);
                    },
                    [](void* m, const int32_t* ids, const uint32_t* f, int n, uint32_t* p, void* st) {
                      return plvs_hip_tsdf_voxblox_halo_export(static_cast<plvs_tsdf_voxblox*>(m), ids, f, n, p, st);
                    },
                    [](void* m, const int32_t* ids, const uint32_t* f, const uint32_t* p, int n, int nf, void* st) {
                      return plvs_hip_tsdf_voxblox_halo_import(static_cast<plvs_tsdf_voxblox*>(m), ids, f, p, n, nf, st);
                    }};
  // the foreign forward neighbours, once each
  std::vector<std::array<int32_t, 3>> need;
  need.reserve((size_t)nblocks * 7);
  for (int i = 0; i < nblocks; ++i)
    for (int d = 1; d < 8; ++d) {
      const std::array<int32_t, 3> nb = {block_ids_xyz[3 * i] + (d & 1), block_ids_xyz[3 * i + 1] + ((d >> 1) & 1),
                                         block_ids_xyz[3 * i + 2] + ((d >> 2) & 1)};
      if (plvs::chisel::shard_of(plvs::chisel::chunk_hash(nb[0], nb[1], nb[2]), world) != rank) need.push_back(nb);
    }
  std::sort(need.begin(), need.end());
  need.erase(std::unique(need.begin(), need.end()), need.end());
  std::vector<int32_t> miss;
  miss.reserve(need.size() * 3);
  for (const auto& nb : need) miss.insert(miss.end(), nb.begin(), nb.end());
  return halo_round(r, rccl_comm, world, static_cast<hipStream_t>(stream), B, ops, miss, fetched);
}

int plvs_hip_rccl_world_size(void* rccl_comm, int* world) {
  PLVS_REQUIRE(rccl_comm && world, "null argument");
  const Rccl* r = rccl();
  if (r == nullptr) {
    plvs::set_error("RCCL is not available in this process");
    return PLVS_ERR_NO_DEVICE;
  }
  if (r->comm_count(rccl_comm, world) != 0) {
    plvs::set_error("ncclCommCount failed");
    return PLVS_ERR_HIP;
  }
  return PLVS_OK;
}

}  // extern "C"
