// The head collective of the batch-sharded forward: RCCL bound by dlopen and driven from C, so that a
// step is "graph launch, collect the logits rows, ncclAllGather" enqueued on one stream by one C call.
// No RCCL header is used: the five entry points are declared here with the published NCCL signatures
// (ncclUniqueId = 128 opaque bytes passed BY VALUE to ncclCommInitRank; ncclUint8 = 1).
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "pv_common.h"

int pv_set_text_error(const char* text);   // pv_plan.hip

namespace {

struct NcclId { char internal[128]; };
typedef int (*fn_get_unique_id)(NcclId*);
typedef int (*fn_comm_init_rank)(void**, int, NcclId, int);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_comm_destroy)(void*);
typedef const char* (*fn_error_string)(int);
typedef int (*fn_get_version)(int*);
constexpr int kNcclUint8 = 1;
// The signatures above (ncclUniqueId = 128 bytes BY VALUE, ncclUint8 = 1) are those of NCCL / RCCL 2.x; a library that
// reports another major version is refused instead of being called through a guessed ABI.
constexpr int kNcclMajorKnown = 2;

struct Rccl {
  void* handle = nullptr;
  std::string path;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_all_gather all_gather = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_error_string error_string = nullptr;
  int version = 0;        // ncclGetVersion: major * 10000 + minor * 100 + patch (2.9+), 0 if the symbol is absent
};

// One binding per distinct candidate list (in practice: one).  Candidates are tried in order; for each, a copy that
// is ALREADY mapped (RTLD_NOLOAD: torch's librccl when torch.distributed is in the process) wins over loading a
// second RCCL instance.
int bind_rccl_uncached(const std::string& all, Rccl* out);

// Bound once per candidate list for the life of the process (every pv_comm_* call used to dlopen again and leak a handle
// reference per call); the handle is never closed: communicators hold function pointers into it.
int bind_rccl(const char* lib_paths, Rccl* out) {
  static std::mutex mu;
  static std::map<std::string, Rccl> cache;
  const std::string all = (lib_paths && *lib_paths) ? lib_paths : "librccl.so:librccl.so.1";
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(all);
  if (it == cache.end()) {
    Rccl r;
    const int rc = bind_rccl_uncached(all, &r);
    if (rc != PV_OK) return rc;
    if (lib_paths && *lib_paths)      // an explicit library list (PV_RCCL_LIB on the binding side: the CPU tests' double) is said out loud
      fprintf(stderr, "pv_comm: RCCL bound from the caller's list \"%s\" -> %s (version %d)\n", all.c_str(), r.path.c_str(), r.version);
    it = cache.emplace(all, r).first;
  }
  *out = it->second;
  return PV_OK;
}

int bind_rccl_uncached(const std::string& all, Rccl* out) {
  std::vector<std::string> cands;
  size_t pos = 0;
  while (pos <= all.size()) {
    const size_t end = all.find(':', pos);
    const std::string c = all.substr(pos, end == std::string::npos ? std::string::npos : end - pos);
    if (!c.empty()) cands.push_back(c);
    if (end == std::string::npos) break;
    pos = end + 1;
  }
  void* h = nullptr;
  std::string used;
  for (const auto& c : cands) {
    h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (h) { used = c; break; }
  }
  std::string errs;
  for (size_t i = 0; !h && i < cands.size(); ++i) {
    h = dlopen(cands[i].c_str(), RTLD_NOW | RTLD_LOCAL);
    if (h) used = cands[i];
    else { const char* e = dlerror(); errs += std::string(e ? e : "?") + "; "; }
  }
  if (!h) {
    pv_set_text_error(("pv_comm: no RCCL library could be loaded (" + errs + ")").c_str());
    return PV_ERR_HIP;
  }
  out->handle = h;
  out->path = used;
  out->get_unique_id = reinterpret_cast<fn_get_unique_id>(dlsym(h, "ncclGetUniqueId"));
  out->comm_init_rank = reinterpret_cast<fn_comm_init_rank>(dlsym(h, "ncclCommInitRank"));
  out->all_gather = reinterpret_cast<fn_all_gather>(dlsym(h, "ncclAllGather"));
  out->comm_destroy = reinterpret_cast<fn_comm_destroy>(dlsym(h, "ncclCommDestroy"));
  out->error_string = reinterpret_cast<fn_error_string>(dlsym(h, "ncclGetErrorString"));
  if (!out->get_unique_id || !out->comm_init_rank || !out->all_gather || !out->comm_destroy) {
    pv_set_text_error(("pv_comm: " + used + " lacks an ncclGetUniqueId / ncclCommInitRank / ncclAllGather / "
                       "ncclCommDestroy symbol").c_str());
    dlclose(h);
    return PV_ERR_HIP;
  }
  if (auto get_version = reinterpret_cast<fn_get_version>(dlsym(h, "ncclGetVersion"))) {
    int v = 0;
    if (get_version(&v) == 0) out->version = v;
    const int major = v >= 10000 ? v / 10000 : v / 1000;      // (before 2.9 the encoding was major * 1000 + ...)
    if (v > 0 && major != kNcclMajorKnown) {
      char buf[256];
      snprintf(buf, sizeof(buf), "pv_comm: %s reports NCCL version %d; only major version %d is known to have the call "
               "signatures this file declares", used.c_str(), v, kNcclMajorKnown);
      pv_set_text_error(buf);
      dlclose(h);
      return PV_ERR_HIP;
    }
  }
  return PV_OK;
}

int nccl_fail(const Rccl& r, int code, const char* what) {
  char buf[512];
  snprintf(buf, sizeof(buf), "%s: RCCL error %d (%s) [%s]", what, code,
           r.error_string ? r.error_string(code) : "?", r.path.c_str());
  pv_set_text_error(buf);
  return PV_ERR_HIP;
}

}  // namespace

struct pv_comm {
  Rccl rccl;
  void* comm = nullptr;
  int rank = 0, world = 1;
};

extern "C" int pv_comm_probe(const char* lib_paths) {
  Rccl r;
  return bind_rccl(lib_paths, &r);     // binding only: no RCCL call is made
}

extern "C" int pv_comm_unique_id(void* id128, const char* lib_paths) {
  if (!id128) return PV_ERR_INVALID;
  Rccl r;
  const int rc = bind_rccl(lib_paths, &r);
  if (rc != PV_OK) return rc;
  NcclId id;
  memset(&id, 0, sizeof(id));
  const int e = r.get_unique_id(&id);
  if (e != 0) return nccl_fail(r, e, "ncclGetUniqueId");   // the handle stays mapped: pv_comm_create reuses it
  memcpy(id128, &id, sizeof(id));
  return PV_OK;
}

extern "C" int pv_comm_create(pv_comm** out, const void* id128, int rank, int world, const char* lib_paths) {
  if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return PV_ERR_INVALID;
  *out = nullptr;
  pv_comm* c = new pv_comm();
  int rc = bind_rccl(lib_paths, &c->rccl);
  if (rc != PV_OK) { delete c; return rc; }
  NcclId id;
  memcpy(&id, id128, sizeof(id));
  const int e = c->rccl.comm_init_rank(&c->comm, world, id, rank);
  if (e != 0) { rc = nccl_fail(c->rccl, e, "ncclCommInitRank"); delete c; return rc; }
  c->rank = rank;
  c->world = world;
  *out = c;
  return PV_OK;
}

extern "C" void pv_comm_destroy(pv_comm* c) {
  if (!c) return;
  if (c->comm) (void)c->rccl.comm_destroy(c->comm);
  delete c;      // the library handle stays mapped (RCCL keeps threads / registrations alive past ncclCommDestroy)
}

extern "C" int pv_comm_rank(const pv_comm* c) { return c ? c->rank : PV_ERR_INVALID; }
extern "C" int pv_comm_world(const pv_comm* c) { return c ? c->world : PV_ERR_INVALID; }
extern "C" const char* pv_comm_library(const pv_comm* c) { return c ? c->rccl.path.c_str() : ""; }

extern "C" int pv_comm_all_gather(pv_comm* c, const void* send, void* recv, size_t bytes_per_rank, pv_stream_t stream) {
  if (!c || !c->comm || !send || !recv) return PV_ERR_INVALID;
  if (bytes_per_rank == 0) return PV_OK;
  const int e = c->rccl.all_gather(send, recv, bytes_per_rank, kNcclUint8, c->comm, static_cast<hipStream_t>(stream));
  if (e != 0) return nccl_fail(c->rccl, e, "ncclAllGather");
  return PV_OK;
}

extern "C" int pv_forward_gather(pv_plan* p, pv_joint* j, pv_comm* c, const pv_gather_src* srcs, int n_srcs,
                                 void* staging, void* recv, pv_stream_t stream) {
  if ((p != nullptr) == (j != nullptr) || !srcs || n_srcs <= 0 || n_srcs > 16 || !recv) return PV_ERR_INVALID;
  const bool exchange = c != nullptr && c->world > 1;
  if (exchange && !staging) return PV_ERR_INVALID;
  size_t total = 0;
  for (int i = 0; i < n_srcs; ++i) {
    if (!srcs[i].ptr || srcs[i].rows < 0 || srcs[i].row_pitch < srcs[i].row_bytes) return PV_ERR_INVALID;
    total += srcs[i].row_bytes * (size_t)srcs[i].rows;
  }
  int rc = p ? pv_plan_graph_launch(p, stream) : pv_joint_launch(j, stream);
  if (rc != PV_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  char* dst = static_cast<char*>(exchange ? staging : (c ? static_cast<char*>(recv) + (size_t)c->rank * total : recv));
  for (int i = 0; i < n_srcs; ++i) {
    const pv_gather_src& g = srcs[i];
    const size_t bytes = g.row_bytes * (size_t)g.rows;
    if (bytes == 0) continue;
    if (g.row_pitch == g.row_bytes || g.rows == 1) {
      PV_HIP_CHECK(hipMemcpyAsync(dst, g.ptr, bytes, hipMemcpyDeviceToDevice, s));
    } else {
      PV_HIP_CHECK(hipMemcpy2DAsync(dst, g.row_bytes, g.ptr, g.row_pitch, g.row_bytes, (size_t)g.rows,
                                    hipMemcpyDeviceToDevice, s));
    }
    dst += bytes;
  }
  if (!exchange) return PV_OK;
  return pv_comm_all_gather(c, staging, recv, total, stream);
}
