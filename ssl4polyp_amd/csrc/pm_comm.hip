// pm_comm.hip -- gradient exchange of the data-parallel step for hosts WITHOUT torch.distributed (SURVEY 8-b lists `pm_comm_*` in
// the C-ABI): a thin handle over RCCL -- ncclAllReduce(SUM, f32) of slices of the caller's flat gradient range on the caller's
// stream, over xGMI.  Replaces what DistributedDataParallel does for the reference (mae/main_pretrain.py:212-214,
// train_classification.py:5746-5750); the Python product path does the same through torch.distributed (parallel.GradSync: same
// bucket slices, same side stream / event fencing), which IS RCCL on ROCm -- this file is that machinery for a C / C++ host.
// RCCL is bound at run time (dlopen), so the library loads on boxes without it and a process that already holds an RCCL (PyTorch
// bundles one) is not forced onto a second copy at link time.  Host code only.
#include "pm_common.h"
#include <dlfcn.h>
#include <string.h>
#include <new>

namespace {

typedef struct { char internal[128]; } pm_nccl_id;   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* pm_nccl_comm;                          // ncclComm_t
struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(pm_nccl_id*) = nullptr;
  int (*CommInitRank)(pm_nccl_comm*, int, pm_nccl_id, int) = nullptr;
  int (*CommDestroy)(pm_nccl_comm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, pm_nccl_comm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  bool ok = false;
};

const Rccl& rccl() {  // resolved once (read-only afterwards)
  static const Rccl r = [] {
    Rccl x;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      x.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (x.handle) break;
    }
    if (!x.handle) return x;
    x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(x.handle, "ncclGetUniqueId"));
    x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(x.handle, "ncclCommInitRank"));
    x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.handle, "ncclCommDestroy"));
    x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(dlsym(x.handle, "ncclAllReduce"));
    x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(x.handle, "ncclGroupStart"));
    x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(x.handle, "ncclGroupEnd"));
    x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllReduce && x.GroupStart && x.GroupEnd;
    return x;
  }();
  return r;
}

constexpr int kNcclSum = 0, kNcclFloat32 = 7;  // rccl.h: ncclSum, ncclFloat32

}  // namespace

struct pm_comm {
  pm_nccl_comm comm;
  int rank, world;
};

extern "C" int pm_comm_unique_id(void* id128) {
  if (!id128) return PM_EINVAL;
  const Rccl& r = rccl();
  if (!r.ok) return PM_EARCH;  // no RCCL on this box
  pm_nccl_id id;
  if (r.GetUniqueId(&id) != 0) return PM_ELAUNCH;
  memcpy(id128, &id, sizeof(id));
  return PM_OK;
}

extern "C" int pm_comm_create(pm_comm** out, const void* id128, int rank, int world) {
  if (!out || !id128) return PM_EINVAL;
  if (world <= 0 || rank < 0 || rank >= world) return PM_ESHAPE;
  const Rccl& r = rccl();
  if (!r.ok) return PM_EARCH;
  pm_nccl_id id;
  memcpy(&id, id128, sizeof(id));
  pm_nccl_comm c = nullptr;
  if (r.CommInitRank(&c, world, id, rank) != 0) return PM_ELAUNCH;
  pm_comm* h = new (std::nothrow) pm_comm{c, rank, world};
  if (!h) {
    r.CommDestroy(c);
    return PM_ELAUNCH;
  }
  *out = h;
  return PM_OK;
}

extern "C" int pm_comm_destroy(pm_comm* c) {
  if (!c) return PM_EINVAL;
  const int st = rccl().CommDestroy(c->comm);
  delete c;
  return st == 0 ? PM_OK : PM_ELAUNCH;
}

extern "C" int pm_comm_world(const pm_comm* c, int* rank, int* world) {
  if (!c) return PM_EINVAL;
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  return PM_OK;
}

// In-place SUM all-reduce of n_buckets slices base[lo[i] .. hi[i]) (element offsets into one f32 range), one RCCL group, on
// `stream` (the caller fences it against its compute stream with events, as parallel.GradSync does).
extern "C" int pm_comm_allreduce_f32(pm_comm* c, float* base, const long* lo, const long* hi, int n_buckets, void* stream) {
  if (!c || !base || !lo || !hi) return PM_EINVAL;
  if (n_buckets <= 0) return PM_ESHAPE;
  for (int i = 0; i < n_buckets; ++i)
    if (lo[i] < 0 || hi[i] <= lo[i]) return PM_ESHAPE;
  const Rccl& r = rccl();
  if (r.GroupStart() != 0) return PM_ELAUNCH;
  int bad = 0;
  for (int i = 0; i < n_buckets; ++i)
    bad |= r.AllReduce(base + lo[i], base + lo[i], (size_t)(hi[i] - lo[i]), kNcclFloat32, kNcclSum, c->comm, pm_stream(stream));
  if (r.GroupEnd() != 0 || bad) return PM_ELAUNCH;
  return PM_OK;
}
