// Sequence-parallel forward with the collective INSIDE the library: the engine's layer loop (mc_blocks_sp) driven by an RCCL
// communicator this file owns, so that a sharded forward is ONE call from the host (mc_forward_sp_rccl) -- no re-entry into
// Python for the per-layer K|V all-gather (round 5: 60 GIL-bound re-entries + 30 NCCL enqueues from Python per forward).
//
// This file is a C host of the public ABI and nothing more: it uses mc_embed / mc_blocks_sp / mc_head / mc_unpatchify /
// mc_buffer_info exactly as a third-party C++ caller would (INTEGRATION.md section 4 shows it as the template).  RCCL is
// bound at run time with dlopen -- the copy already mapped into the process (PyTorch-ROCm ships its own librccl.so and its
// own HIP runtime: the communicator must live on THAT runtime) or the system one -- so libmagcache_hip.so carries no link
// dependency on it and single-GPU users never load it.
//
// Reference counterpart: the USP path of MagCache4Wan2.1/magcache_generate.py:813-829,891 (xfuser's all-to-all / ring
// attention); BASELINE.json's north_star asks for "RCCL all-gather over xGMI only for the full-sequence attention join".
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/magcache_hip.h"

namespace mc {
mc_status set_error_v(mc_status s, const char* fmt, va_list ap);   // engine.cpp: the text mc_last_error() reports
}

namespace {

mc_status failf(mc_status s, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  s = mc::set_error_v(s, fmt, ap);
  va_end(ap);
  return s;
}

struct Rccl {
  void* h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  char where[256] = "";
};

Rccl g_rccl;

// the library already in the process first (RTLD_NOLOAD): two RCCL copies would sit on two HIP runtimes
mc_status load_rccl() {
  if (g_rccl.h) return MC_OK;
  const char* names[] = {"librccl.so", "librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) { snprintf(g_rccl.where, sizeof(g_rccl.where), "%s (already mapped)", n); break; }
  if (!h) {
    const char* env = getenv("MAGCACHE_RCCL_LIB");
    const char* more[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : more)
      if (n && (h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) { snprintf(g_rccl.where, sizeof(g_rccl.where), "%s", n); break; }
  }
  if (!h) return failf(MC_ESTATE, "librccl not found (dlopen: %s); set MAGCACHE_RCCL_LIB", dlerror());
#define SYM(f)                                                                                  \
  g_rccl.f = (decltype(g_rccl.f))dlsym(h, "nccl" #f);                                           \
  if (!g_rccl.f) return failf(MC_ESTATE, "librccl (%s) lacks nccl" #f, g_rccl.where)
  SYM(GetUniqueId); SYM(CommInitRank); SYM(CommDestroy); SYM(AllGather); SYM(AllReduce); SYM(GetErrorString); SYM(GetVersion);
#undef SYM
  g_rccl.h = h;
  return MC_OK;
}

#define NCCL_TRY(expr)                                                                                          \
  do {                                                                                                          \
    ncclResult_t _r = (expr);                                                                                   \
    if (_r != ncclSuccess) return failf(MC_EHIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
  } while (0)
#define HIP_TRY(expr)                                                                                           \
  do {                                                                                                          \
    hipError_t _e = (expr);                                                                                     \
    if (_e != hipSuccess) return failf(MC_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

}  // namespace

struct mc_sp_comm {
  ncclComm_t comm = nullptr;
  int nranks = 0, rank = 0;
  hipStream_t cs = nullptr;            // the collective's own stream: the gather rounds run beside the launch stream
  hipEvent_t ready = nullptr;          // launch stream -> cs: "kv_local" is complete
  std::vector<hipEvent_t> done;        // cs -> launch stream: round c has landed
  // the engine this communicator last served and its buffers (resolved once per engine)
  mc_engine* e = nullptr;
  char* kv_local = nullptr;
  char* kv_gather = nullptr;
  int d2 = 0;                          // 2 * dim: elements per K|V row
  int n_rounds = 0, chunk_rows = 0;
  int rc = 0;                          // first failure inside a callback
};

namespace {

mc_status bind_engine(mc_sp_comm* c, mc_engine* e) {
  size_t off = 0, bytes = 0, off_g = 0, bytes_g = 0;
  char* base = (char*)mc_workspace_base(e);
  if (!base) return failf(MC_ESTATE, "mc_set_workspace must run first");
  if (mc_status st = mc_buffer_info(e, "kv_local", &off, &bytes); st != MC_OK) return st;
  if (mc_status st = mc_buffer_info(e, "kv_gather", &off_g, &bytes_g); st != MC_OK) return st;
  int rows = 0, nr = 0, dim = 0, sp = 0;
  if (mc_status st = mc_sp_round_info(e, 0, &nr, &rows, nullptr); st != MC_OK) return st;
  if (mc_status st = mc_sp_geometry(e, nullptr, nullptr, nullptr, nullptr, &dim, &sp); st != MC_OK) return st;
  if (nr < 1) return failf(MC_ESTATE, "the engine is not sequence parallel (sp_size 1)");
  if (sp != c->nranks) return failf(MC_EINVAL, "communicator of %d ranks on an engine with sp_size %d", c->nranks, sp);
  c->e = e; c->kv_local = base + off; c->kv_gather = base + off_g;
  c->d2 = 2 * dim;
  c->n_rounds = nr; c->chunk_rows = rows;
  while ((int)c->done.size() < nr) {
    hipEvent_t ev;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    c->done.push_back(ev);
  }
  return MC_OK;
}

// mc_sp_gather_fn: phase 2 c = start round c on the communicator's stream, 2 c + 1 = the launch stream waits for it
int gather_cb(void* user, int /*layer*/, int phase, mc_stream stream_) {
  mc_sp_comm* c = (mc_sp_comm*)user;
  hipStream_t s = (hipStream_t)stream_;
  const int r = phase >> 1;
  if (r < 0 || r >= c->n_rounds) return (c->rc = 100);
  auto bad = [&](int code) { c->rc = code; return code; };
  if (phase & 1) return hipStreamWaitEvent(s, c->done[r], 0) == hipSuccess ? 0 : bad(3);
  if (r == 0) {   // the k|v rows are complete once what `stream` holds so far has run
    if (hipEventRecord(c->ready, s) != hipSuccess || hipStreamWaitEvent(c->cs, c->ready, 0) != hipSuccess) return bad(1);
  }
  const size_t count = (size_t)c->chunk_rows * c->d2;                      // bf16 elements one rank sends in a round
  const char* send = c->kv_local + (size_t)r * count * 2;
  char* recv = c->kv_gather + (size_t)r * c->nranks * count * 2;
  if (g_rccl.AllGather(send, recv, count, ncclBfloat16, c->comm, c->cs) != ncclSuccess) return bad(2);
  return hipEventRecord(c->done[r], c->cs) == hipSuccess ? 0 : bad(4);
}

}  // namespace

extern "C" {

int mc_sp_rccl_available(void) { return load_rccl() == MC_OK ? 1 : 0; }

mc_status mc_sp_comm_id(void* id_out) {
  if (!id_out) return failf(MC_EINVAL, "null id");
  if (mc_status st = load_rccl(); st != MC_OK) return st;
  ncclUniqueId id;
  NCCL_TRY(g_rccl.GetUniqueId(&id));
  static_assert(sizeof(id) == MC_SP_ID_BYTES, "ncclUniqueId is 128 bytes");
  memcpy(id_out, &id, sizeof(id));
  return MC_OK;
}

mc_status mc_sp_comm_create(const void* id_in, int nranks, int rank, mc_sp_comm** out) {
  if (!id_in || !out || nranks < 1 || rank < 0 || rank >= nranks) return failf(MC_EINVAL, "bad communicator arguments");
  if (mc_status st = load_rccl(); st != MC_OK) return st;
  ncclUniqueId id;
  memcpy(&id, id_in, sizeof(id));
  mc_sp_comm* c = new mc_sp_comm();
  c->nranks = nranks; c->rank = rank;
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);      // on the calling thread's current device
  if (r != ncclSuccess) { delete c; return failf(MC_EHIP, "ncclCommInitRank(%d ranks, rank %d): %s", nranks, rank, g_rccl.GetErrorString(r)); }
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  // highest priority: the collective's few workgroups must get their CUs between the attention workgroups queued behind them
  hipError_t e1 = hipStreamCreateWithPriority(&c->cs, hipStreamNonBlocking, hi);
  hipError_t e2 = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
  if (e1 != hipSuccess || e2 != hipSuccess) {
    mc_sp_comm_destroy(c);
    return failf(MC_EHIP, "stream / event for the communicator: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
  }
  *out = c;
  return MC_OK;
}

void mc_sp_comm_destroy(mc_sp_comm* c) {
  if (!c) return;
  if (c->cs) (void)hipStreamSynchronize(c->cs);
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  for (hipEvent_t ev : c->done) (void)hipEventDestroy(ev);
  if (c->ready) (void)hipEventDestroy(c->ready);
  if (c->cs) (void)hipStreamDestroy(c->cs);
  delete c;
}

const char* mc_sp_comm_info(const mc_sp_comm* c) {
  static char buf[400];
  int v = 0;
  if (g_rccl.GetVersion) (void)g_rccl.GetVersion(&v);
  snprintf(buf, sizeof(buf), "rccl %d from %s; %d ranks, rank %d", v, g_rccl.where, c ? c->nranks : 0, c ? c->rank : -1);
  return buf;
}

mc_status mc_blocks_sp_rccl(mc_engine* e, mc_sp_comm* c, int layer_begin, int layer_end, int branch, mc_mode mode, int overlap,
                            mc_stream stream) {
  if (!e || !c) return failf(MC_EINVAL, "null argument");
  if (mc_status st = bind_engine(c, e); st != MC_OK) return st;     // cheap; follows mc_sp_set_chunks
  c->rc = 0;
  return mc_blocks_sp(e, layer_begin, layer_end, branch, mode, overlap, gather_cb, c, stream);
}

mc_status mc_forward_sp_rccl(mc_engine* e, mc_sp_comm* c, const float* latent_dev, const float* t_dev, double t_host,
                             const void* context_dev, mc_dtype ctx_dtype, int ctx_len, int branch, mc_mode mode, int overlap,
                             float* tokens_full_dev, float* out_dev, mc_stream stream) {
  if (!e || !c || !tokens_full_dev || !out_dev) return failf(MC_EINVAL, "null argument");
  hipStream_t s = (hipStream_t)stream;
  mc_status st = mc_embed(e, latent_dev, t_dev, t_host, context_dev, ctx_dtype, ctx_len, stream);
  if (st != MC_OK) return st;
  int n_layers = 0, seq_len = 0, rows = 0, stride = 0;
  if ((st = mc_sp_geometry(e, &n_layers, &seq_len, &rows, &stride, nullptr, nullptr)) != MC_OK) return st;
  char* base = (char*)mc_workspace_base(e);
  size_t off = 0, bytes = 0;
  if (mode != MC_MODE_SKIP) {
    if ((st = mc_blocks_sp_rccl(e, c, 0, n_layers, branch, mode, overlap, stream)) != MC_OK) return st;
    if (mode == MC_MODE_CALIB) {
      int has = 0;
      if ((st = mc_calib_ready(e, branch, &has)) != MC_OK) return st;
      if (has) {   // (sum rho, sum rho^2, sum 1 - cos, count) over all ranks, then the triple from the sums
        if ((st = mc_buffer_info(e, "calib_sums", &off, &bytes)) != MC_OK) return st;
        NCCL_TRY(g_rccl.AllReduce(base + off, base + off, 4, ncclDouble, ncclSum, c->comm, s));
        if ((st = mc_calib_finalize(e, branch, stream)) != MC_OK) return st;
      }
    }
  }
  if ((st = mc_head(e, branch, mode, stream)) != MC_OK) return st;
  if ((st = mc_buffer_info(e, "head_tokens", &off, &bytes)) != MC_OK) return st;
  // every rank's [rows][stride] fp32 head output -> tokens_full [P * rows][stride], in rank order = token order
  NCCL_TRY(g_rccl.AllGather(base + off, tokens_full_dev, (size_t)rows * stride, ncclFloat, c->comm, s));
  return mc_unpatchify(e, tokens_full_dev, 0, seq_len, out_dev, stream);
}

}  // extern "C"
