// Library bookkeeping + the execution plan: a deploy-form model is a fixed list of kernel
// launches (it is specialised to one input size), so the host side records the descriptors
// once and replays them from C++ -- eagerly, or as one captured hipGraph.
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <string>
#include <vector>
#include "pv_common.h"

namespace {
thread_local std::string g_last_error;
constexpr int kAbiVersion = 32;
}  // namespace

int pv_set_hip_error(hipError_t e, const char* what) {
  char buf[512];
  snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
  g_last_error = buf;
  return PV_ERR_HIP;
}

int pv_set_text_error(const char* text) {
  g_last_error = text ? text : "";
  return PV_ERR_HIP;
}

namespace {
thread_local const char* g_last_kernel = "";
}  // namespace
void pv_note_kernel(const char* name) { g_last_kernel = name; }

// Development knobs.  The table is process-global and may be written through the public ABI while another thread
// launches ops, so every access takes the mutex (a launch reads a handful of knobs: nanoseconds against a
// microsecond-scale hipLaunchKernel).  Knob values are baked into a graph at capture time: set them BEFORE building.
namespace {
struct TuneEntry { std::string key; int value; };
std::vector<TuneEntry>& tune_table() { static std::vector<TuneEntry> t; return t; }
std::mutex& tune_mutex() { static std::mutex m; return m; }
}  // namespace

int pv_tune(const char* key, int dflt) {
  std::lock_guard<std::mutex> lock(tune_mutex());
  for (const auto& e : tune_table())
    if (e.key == key) return e.value;
  return dflt;
}

extern "C" int pv_tune_set(const char* key, int value) {
  if (!key || !*key || strlen(key) > 48) return PV_ERR_INVALID;
  std::lock_guard<std::mutex> lock(tune_mutex());
  for (auto& e : tune_table())
    if (e.key == key) { e.value = value; return PV_OK; }
  tune_table().push_back({key, value});
  return PV_OK;
}

extern "C" int pv_tune_clear(void) {
  std::lock_guard<std::mutex> lock(tune_mutex());
  tune_table().clear();
  return PV_OK;
}

extern "C" int pv_version(void) { return kAbiVersion; }
extern "C" const char* pv_last_error(void) { return g_last_error.c_str(); }
extern "C" int pv_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

struct pv_plan {
  struct Op {
    int kind;
    std::vector<unsigned char> desc;
    std::string kernel;      // symbol of the (last) kernel the op launched; filled by pv_plan_profile
  };
  std::vector<Op> ops;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

namespace {

size_t desc_size(int kind) {
  switch (kind) {
    case PV_OP_CONV3D: return sizeof(pv_conv3d_desc);
    case PV_OP_DWCONV3D: return sizeof(pv_dwconv3d_desc);
    case PV_OP_SE_GATE: return sizeof(pv_se_gate_desc);
    case PV_OP_POOL3D: return sizeof(pv_pool3d_desc);
    case PV_OP_LAYERNORM:
    case PV_OP_SOFTMAX_ROWS:
    case PV_OP_MEAN_ROWS:
    case PV_OP_AFFINE_ROWS: return sizeof(pv_rows_desc);
    case PV_OP_POSENC: return sizeof(pv_posenc_desc);
    case PV_OP_ATTENTION: return sizeof(pv_attention_desc);
    case PV_OP_ADD_ACT: return sizeof(pv_add_desc);
    case PV_OP_INGEST:
    case PV_OP_EGRESS: return sizeof(pv_layout_desc);
    case PV_OP_TOKEN_POOL: return sizeof(pv_token_pool_desc);
    case PV_OP_ROI_ALIGN: return sizeof(pv_roi_align_desc);
    case PV_OP_LATERAL: return sizeof(pv_lateral_desc);
    case PV_OP_MLP_ROWS: return sizeof(pv_mlp_desc);
    case PV_OP_LN_LINEAR: return sizeof(pv_ln_linear_desc);
    case PV_OP_BOTTLENECK: return sizeof(pv_bottleneck_desc);
    default: return 0;
  }
}

int run_op(const pv_plan::Op& op, pv_stream_t s) {
  const void* p = op.desc.data();
  switch (op.kind) {
    case PV_OP_CONV3D: return pv_conv3d(static_cast<const pv_conv3d_desc*>(p), s);
    case PV_OP_DWCONV3D: return pv_dwconv3d(static_cast<const pv_dwconv3d_desc*>(p), s);
    case PV_OP_SE_GATE: return pv_se_gate(static_cast<const pv_se_gate_desc*>(p), s);
    case PV_OP_POOL3D: return pv_pool3d(static_cast<const pv_pool3d_desc*>(p), s);
    case PV_OP_LAYERNORM: return pv_layernorm(static_cast<const pv_rows_desc*>(p), s);
    case PV_OP_SOFTMAX_ROWS: return pv_softmax_rows(static_cast<const pv_rows_desc*>(p), s);
    case PV_OP_MEAN_ROWS: return pv_mean_rows(static_cast<const pv_rows_desc*>(p), s);
    case PV_OP_POSENC: return pv_add_posenc(static_cast<const pv_posenc_desc*>(p), s);
    case PV_OP_ATTENTION: return pv_attention(static_cast<const pv_attention_desc*>(p), s);
    case PV_OP_ADD_ACT: return pv_add_act(static_cast<const pv_add_desc*>(p), s);
    case PV_OP_INGEST: return pv_ingest_ncdhw(static_cast<const pv_layout_desc*>(p), s);
    case PV_OP_EGRESS: return pv_egress_ncdhw(static_cast<const pv_layout_desc*>(p), s);
    case PV_OP_TOKEN_POOL: return pv_token_pool(static_cast<const pv_token_pool_desc*>(p), s);
    case PV_OP_ROI_ALIGN: return pv_roi_align(static_cast<const pv_roi_align_desc*>(p), s);
    case PV_OP_LATERAL: return pv_lateral_fuse(static_cast<const pv_lateral_desc*>(p), s);
    case PV_OP_AFFINE_ROWS: return pv_affine_rows(static_cast<const pv_rows_desc*>(p), s);
    case PV_OP_MLP_ROWS: return pv_mlp_rows(static_cast<const pv_mlp_desc*>(p), s);
    case PV_OP_LN_LINEAR: return pv_ln_linear_rows(static_cast<const pv_ln_linear_desc*>(p), s);
    case PV_OP_BOTTLENECK: return pv_bottleneck(static_cast<const pv_bottleneck_desc*>(p), s);
    default: return PV_ERR_INVALID;
  }
}

void drop_graph(pv_plan* p) {
  if (p->exec) { (void)hipGraphExecDestroy(p->exec); p->exec = nullptr; }
  if (p->graph) { (void)hipGraphDestroy(p->graph); p->graph = nullptr; }
}

}  // namespace

extern "C" pv_plan* pv_plan_create(void) { return new pv_plan(); }

extern "C" void pv_plan_destroy(pv_plan* p) {
  if (!p) return;
  drop_graph(p);
  delete p;
}

extern "C" int pv_plan_add(pv_plan* p, int op_kind, const void* desc, size_t desc_bytes) {
  if (!p || !desc) return PV_ERR_INVALID;
  const size_t want = desc_size(op_kind);
  if (want == 0 || want != desc_bytes) return PV_ERR_INVALID;  // ABI mismatch guard
  pv_plan::Op op;
  op.kind = op_kind;
  op.desc.assign(static_cast<const unsigned char*>(desc), static_cast<const unsigned char*>(desc) + want);
  p->ops.push_back(std::move(op));
  drop_graph(p);
  return (int)p->ops.size() - 1;
}

extern "C" int pv_plan_size(const pv_plan* p) { return p ? (int)p->ops.size() : PV_ERR_INVALID; }

extern "C" const char* pv_plan_op_kernel(const pv_plan* p, int i) {
  if (!p || i < 0 || i >= (int)p->ops.size()) return "";
  return p->ops[i].kernel.c_str();
}

extern "C" int pv_plan_launch_range(pv_plan* p, int first, int last, pv_stream_t stream) {
  if (!p || first < 0 || last > (int)p->ops.size() || first > last) return PV_ERR_INVALID;
  for (int i = first; i < last; ++i) {
    const int r = run_op(p->ops[i], stream);
    if (r != PV_OK) return r;
  }
  return PV_OK;
}

extern "C" int pv_plan_launch(pv_plan* p, pv_stream_t stream) {
  if (!p) return PV_ERR_INVALID;
  return pv_plan_launch_range(p, 0, (int)p->ops.size(), stream);
}

extern "C" int pv_plan_graph_build(pv_plan* p, pv_stream_t stream) {
  if (!p) return PV_ERR_INVALID;
  drop_graph(p);
  // Capture on a private non-blocking stream: the caller's stream may be the legacy default
  // stream (torch's current stream usually is), which cannot be captured.  The instantiated
  // graph can then be launched into any stream, the default one included.
  (void)stream;
  hipStream_t cs = nullptr;
  PV_HIP_CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) {
    (void)hipStreamDestroy(cs);
    return pv_set_hip_error(e, "hipStreamBeginCapture");
  }
  const int r = pv_plan_launch(p, cs);
  hipGraph_t g = nullptr;
  e = hipStreamEndCapture(cs, &g);
  (void)hipStreamDestroy(cs);
  if (r != PV_OK) {
    if (g) (void)hipGraphDestroy(g);
    return r;
  }
  if (e != hipSuccess) return pv_set_hip_error(e, "hipStreamEndCapture");
  p->graph = g;
  PV_HIP_CHECK(hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0));
  return PV_OK;
}

// A joint graph: the plans of several sub-batches as parallel branches of ONE hipGraph.  It is its own object with
// its own graph / exec (round 3; it used to live in plans[0]'s slot, where a later pv_plan_graph_build of that plan
// silently replaced it): building or dropping the graph of a member plan does not touch it, and it holds no
// reference to the plans beyond the captured kernel nodes -- rebuild it after pv_plan_add on a member.
struct pv_joint {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  int branches = 0;
};

namespace {
void drop_joint(pv_joint* j) {
  if (j->exec) { (void)hipGraphExecDestroy(j->exec); j->exec = nullptr; }
  if (j->graph) { (void)hipGraphDestroy(j->graph); j->graph = nullptr; }
  j->branches = 0;
}
}  // namespace

extern "C" pv_joint* pv_joint_create(void) { return new pv_joint(); }

extern "C" void pv_joint_destroy(pv_joint* j) {
  if (!j) return;
  drop_joint(j);
  delete j;
}

extern "C" int pv_joint_branches(const pv_joint* j) { return j ? j->branches : PV_ERR_INVALID; }

extern "C" int pv_joint_build(pv_joint* j, pv_plan* const* plans, int n) {
  if (!j || !plans || n <= 0 || n > 16) return PV_ERR_INVALID;
  for (int i = 0; i < n; ++i)
    if (!plans[i]) return PV_ERR_INVALID;
  drop_joint(j);
  // fork / join by events inside the capture: branch i > 0 is recorded on its own stream behind a fork event of the
  // origin stream, and the origin stream waits for every branch's last event before the capture ends
  std::vector<hipStream_t> ss(n, nullptr);
  std::vector<hipEvent_t> ev(n, nullptr);
  auto cleanup = [&]() {
    for (auto& s : ss) if (s) (void)hipStreamDestroy(s);
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
  };
  for (int i = 0; i < n; ++i) {
    hipError_t e = hipStreamCreateWithFlags(&ss[i], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
    if (e != hipSuccess) { cleanup(); return pv_set_hip_error(e, "joint graph: stream / event"); }
  }
  hipError_t e = hipStreamBeginCapture(ss[0], hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) { cleanup(); return pv_set_hip_error(e, "hipStreamBeginCapture"); }
  int r = PV_OK;
  hipError_t he = hipEventRecord(ev[0], ss[0]);
  for (int i = 1; i < n && he == hipSuccess; ++i) he = hipStreamWaitEvent(ss[i], ev[0], 0);
  for (int i = n - 1; i >= 0 && he == hipSuccess && r == PV_OK; --i) {
    r = pv_plan_launch(plans[i], ss[i]);
    if (i > 0 && r == PV_OK) he = hipEventRecord(ev[i], ss[i]);
  }
  // join only after the origin stream's own branch is recorded (a wait placed earlier would order that branch
  // behind the others)
  for (int i = 1; i < n && he == hipSuccess && r == PV_OK; ++i) he = hipStreamWaitEvent(ss[0], ev[i], 0);
  hipGraph_t g = nullptr;
  e = hipStreamEndCapture(ss[0], &g);
  cleanup();
  if (r != PV_OK || he != hipSuccess) {
    if (g) (void)hipGraphDestroy(g);
    return r != PV_OK ? r : pv_set_hip_error(he, "joint graph: fork / join");
  }
  if (e != hipSuccess) return pv_set_hip_error(e, "hipStreamEndCapture");
  j->graph = g;
  PV_HIP_CHECK(hipGraphInstantiate(&j->exec, j->graph, nullptr, nullptr, 0));
  j->branches = n;
  return PV_OK;
}

extern "C" int pv_joint_launch(pv_joint* j, pv_stream_t stream) {
  if (!j || !j->exec) return PV_ERR_INVALID;
  PV_HIP_CHECK(hipGraphLaunch(j->exec, static_cast<hipStream_t>(stream)));
  return PV_OK;
}

extern "C" int pv_plan_graph_launch(pv_plan* p, pv_stream_t stream) {
  if (!p || !p->exec) return PV_ERR_INVALID;
  PV_HIP_CHECK(hipGraphLaunch(p->exec, static_cast<hipStream_t>(stream)));
  return PV_OK;
}

// Per-op device time.  Every op is timed IN SITU (the ops before it have just run, so its operands sit in the
// caches exactly as in a replay) between two events of its own, and the host is kept out of the measurement:
// each measured pass is queued behind one un-instrumented replay of the whole plan, so the device still has
// milliseconds of queued work when the host records the events (an event per op, as a first version did, makes
// the "kernel time" of a 20 us kernel the host's launch + record time on a slow host).  One pass instruments
// every kStride-th op; the minimum over `iters` passes is reported, minus the null interval of an event pair
// with nothing between (the marker packets' own cost), clamped at 0.
extern "C" int pv_plan_profile(pv_plan* p, pv_stream_t stream, int iters, float* ms_per_op) {
  if (!p || !ms_per_op || iters <= 0) return PV_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int n = (int)p->ops.size();
  constexpr int kStride = 8;
  const int slots = (n + kStride - 1) / kStride + 1;   // +1: the null pair
  std::vector<hipEvent_t> e0(slots), e1(slots);
  for (auto& e : e0) PV_HIP_CHECK(hipEventCreate(&e));
  for (auto& e : e1) PV_HIP_CHECK(hipEventCreate(&e));
  for (int i = 0; i < n; ++i) ms_per_op[i] = 1e30f;
  float null_ms = 1e30f;
  int rc = PV_OK;
  for (int it = 0; it < iters && rc == PV_OK; ++it) {
    for (int phase = 0; phase < kStride && phase < n && rc == PV_OK; ++phase) {
      rc = pv_plan_launch(p, stream);                  // queue filler: the host runs ahead of the device
      if (rc != PV_OK) break;
      int slot = 0;
      for (int i = 0; i < n; ++i) {
        const bool timed = (i % kStride) == phase;
        if (timed) PV_HIP_CHECK(hipEventRecord(e0[slot], s));
        g_last_kernel = "";
        rc = run_op(p->ops[i], stream);
        if (rc != PV_OK) break;
        if (timed) {
          PV_HIP_CHECK(hipEventRecord(e1[slot++], s));
          // "(gemm_glds_kernel<true, 2>)" -> "gemm_glds_kernel": the symbol without parentheses / template arguments
          std::string k = g_last_kernel;
          while (!k.empty() && (k.front() == '(' || k.front() == ' ')) k.erase(k.begin());
          const size_t cut = k.find_first_of("<) ");
          if (cut != std::string::npos) k.resize(cut);
          p->ops[i].kernel = k;
        }
      }
      if (rc != PV_OK) break;
      PV_HIP_CHECK(hipEventRecord(e0[slot], s));       // null pair
      PV_HIP_CHECK(hipEventRecord(e1[slot], s));
      PV_HIP_CHECK(hipStreamSynchronize(s));
      float ms = 0.f;
      PV_HIP_CHECK(hipEventElapsedTime(&ms, e0[slot], e1[slot]));
      if (ms < null_ms) null_ms = ms;
      slot = 0;
      for (int i = phase; i < n; i += kStride, ++slot) {
        PV_HIP_CHECK(hipEventElapsedTime(&ms, e0[slot], e1[slot]));
        if (ms < ms_per_op[i]) ms_per_op[i] = ms;
      }
    }
  }
  for (auto& e : e0) (void)hipEventDestroy(e);
  for (auto& e : e1) (void)hipEventDestroy(e);
  if (rc == PV_OK) {
    if (null_ms > 1e29f) null_ms = 0.f;
    for (int i = 0; i < n; ++i) ms_per_op[i] = ms_per_op[i] > null_ms ? ms_per_op[i] - null_ms : 0.f;
  }
  return rc;
}
