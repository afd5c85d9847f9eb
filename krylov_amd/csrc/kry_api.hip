// C ABI of libkrylov_amd: handle management and the host-side orchestration of
// the device kernels.  No host synchronisation happens inside a GMRES cycle or
// inside an implicit-Euler step; results are read back only where the ABI
// hands host memory to the caller.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <dlfcn.h>

#include <atomic>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include <rccl/rccl.h>  // types and prototypes only: the library is loaded on demand

#include "kry_internal.h"

using namespace kry;

static thread_local std::string g_err;

// Bumped by every setter that changes which kernels a solve launches or how (links per
// sweep, hand-over to the one-sweep steps, generated bands, the 512-thread hand-over): a
// captured step is recorded again when the count differs from the one it was captured at.
static std::atomic<uint64_t> g_config_epoch{1};
static void config_changed() { g_config_epoch.fetch_add(1, std::memory_order_relaxed); }

static int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                                  \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess)                                                             \
      return fail(KRY_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                \
  } while (0)

#define KCHK(expr)            \
  do {                        \
    int s_ = (expr);          \
    if (s_ != KRY_OK) return s_; \
  } while (0)

static const double kSNaN = std::numeric_limits<double>::signaling_NaN();

// Scratch of kry_apply_delta_noalloc, an entry point for callers that cannot allocate
// device memory: the per-workgroup partial results of the state update (2 * kMaxParts
// doubles) followed by the three scalars of the fold (k_max_fold).  Every GPU the library
// runs on holds its own copy of this array at its own address until the process ends.
// In host code the name below is only a handle for hipGetSymbolAddress; the address the
// kernels get is looked up once per device in kry_ctx_create (scratch_of_device).
constexpr int kNoAllocScratchDoubles = 2 * kMaxParts + 4;
static __device__ __align__(256) double scratch[kNoAllocScratchDoubles];
constexpr int kMaxDevices = 64;
static double *g_scratch_ptr[kMaxDevices];  // device address of `scratch` per device id
static std::mutex g_scratch_mutex;

// Address of this device's copy of `scratch`; `device` must be the current device.
static int scratch_of_device(int device, double **out) {
  if (device < 0 || device >= kMaxDevices) return KRY_ERR_INVALID;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[device]) {
    double *p = nullptr;
    const hipError_t e = hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) return KRY_ERR_HIP;
    g_scratch_ptr[device] = p;
  }
  *out = g_scratch_ptr[device];
  return KRY_OK;
}

// ---------------------------------------------------------------------------
// Handles
// ---------------------------------------------------------------------------
struct KernelRec {
  int cls;
  hipEvent_t a, b;
};

struct kry_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // optional per-kernel-class timing with HIP events on `stream`
  // handles created on this context keep it alive: destroy order is free
  int refs = 0;
  bool released = false;
  bool timing = false;
  std::vector<hipEvent_t> free_events;
  std::vector<KernelRec> recs;
  // row-block decomposition over several ranks (one process per GPU): the
  // caller supplies the collectives, the library decides when to call them
  int rank = 0, nranks = 1;
  kry_allreduce_fn allreduce = nullptr;
  kry_sendrecv_fn sendrecv = nullptr;
  void *comm_user = nullptr;
  // ... or the library talks to RCCL itself on `stream` (kry_ctx_set_comm_rccl)
  ncclComm_t nccl = nullptr;
  // this device's copy of the file-scope scratch array (kry_apply_delta_noalloc)
  double *scratch = nullptr;
  // Row blocks: a second stream on which a halo exchange travels while the rows that do
  // not need it are worked on (created on first use; fork: main stream -> this one,
  // join: back)
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

// RCCL entry points, resolved at run time so that single-GPU users never load
// the (large) collective library.  A copy that is already in the process - the
// one PyTorch links - is reused.
struct RcclApi {
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
};

static const RcclApi &rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *nm : names)
      if (!api.handle) api.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
    for (const char *nm : names)
      if (!api.handle) api.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (!api.handle) return;
#define KRY_RCCL_SYM(field, sym) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, #sym))
    KRY_RCCL_SYM(GetUniqueId, ncclGetUniqueId);
    KRY_RCCL_SYM(CommInitRank, ncclCommInitRank);
    KRY_RCCL_SYM(CommDestroy, ncclCommDestroy);
    KRY_RCCL_SYM(AllReduce, ncclAllReduce);
    KRY_RCCL_SYM(Send, ncclSend);
    KRY_RCCL_SYM(Recv, ncclRecv);
    KRY_RCCL_SYM(GroupStart, ncclGroupStart);
    KRY_RCCL_SYM(GroupEnd, ncclGroupEnd);
    KRY_RCCL_SYM(GetErrorString, ncclGetErrorString);
#undef KRY_RCCL_SYM
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.Send && api.Recv &&
             api.GroupStart && api.GroupEnd && api.GetErrorString;
  });
  return api;
}

#define NCCLCHK(call)                                                                     \
  do {                                                                                    \
    ncclResult_t r_ = (call);                                                             \
    if (r_ != ncclSuccess) return fail(KRY_ERR_HIP, "%s: %s", #call, rccl_api().GetErrorString(r_)); \
  } while (0)

// Brackets the launches issued in its scope with two events on the context's
// stream when timing is enabled; a no-op otherwise.
struct ScopedKernelTimer {
  kry_ctx *c;
  int cls;
  hipEvent_t a = nullptr, b = nullptr;
  static hipEvent_t get(kry_ctx *c) {
    hipEvent_t e = nullptr;
    if (!c->free_events.empty()) {
      e = c->free_events.back();
      c->free_events.pop_back();
    } else if (hipEventCreate(&e) != hipSuccess) {
      e = nullptr;
    }
    return e;
  }
  ScopedKernelTimer(kry_ctx *ctx, int k) : c(ctx), cls(k) {
    if (!c->timing) return;
    a = get(c);
    b = get(c);
    if (a) (void)hipEventRecord(a, c->stream);
  }
  ~ScopedKernelTimer() {
    if (!c->timing || !a || !b) return;
    (void)hipEventRecord(b, c->stream);
    c->recs.push_back({cls, a, b});
  }
};

struct kry_op {
  kry_ctx *ctx = nullptr;
  int kind = 0;  // 0 stencil, 1 csr
  int64_t n = 0;
  int rows = 0, cols = 0;
  double *band[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  bool own_bands = false;
  int64_t *rowptr = nullptr;
  int32_t *col = nullptr;
  double *val = nullptr;
  int64_t nnz = 0;
  double *tx = nullptr, *ty = nullptr;  // staging for the host apply
  bool slab = false;  // local rows of a row-block decomposition
  // bands s and n are zero where they would couple the two ends of a grid row
  // (column 0 / the last column): the operator is a true 2-D stencil
  bool wrap_free = false;
  // interior band entries can be evaluated instead of read (operators assembled by a step object)
  BandGen gen;
  bool has_gen = false, use_gen = true;
  // the same recipe with strip-major velocity snapshots (StripMap), for solvers whose basis is strip-major
  BandGen gen_strips;
  bool has_gen_strips = false;
};

struct kry_gmres {
  kry_ctx *ctx = nullptr;
  int64_t n = 0, ld = 0;
  // slab vectors carry `halo` padding entries in front of and behind the n owned
  // ones, the innermost halo_row of which are the neighbours' boundary rows
  int64_t halo = 0, halo_row = 0;
  bool offset_applied = false;  // w / vars / z already point past their front halo
  double *tot = nullptr;  // kMaxRed totals handed to the all-reduce
  double *G = nullptr;    // (m+1) x (m+1) Gram entries <v_a, v_b> of basis vectors in the same group, a < b
  int m = 0, flags = 0;
  int kblock = kMgsBlock;  // Gram-Schmidt links fused per sweep (1 = the plain chain), <= kMgsBlock
  int nvec = 0;  // subspace.size()
  double *V = nullptr;      // (m+1) x ld
  double *H = nullptr;      // (m+1) x m
  double *w = nullptr;      // Arnoldi work vector
  double *z = nullptr;      // preconditioned vector
  double *vars = nullptr;   // Solver::vars
  double *parts = nullptr;  // 2 x kMaxRed x kMaxParts (ping-pong partial sums)
  double *beta = nullptr;   // 1
  double *y = nullptr;      // m
  double *work = nullptr;   // 3 m^2 + 3 m
  double *bbuf = nullptr;   // staging of a host right-hand side
  // tridiagonal preconditioner
  double *lo = nullptr, *di = nullptr, *up = nullptr;  // owned only for CSR operators
  bool own_tri = false;
  double *dd = nullptr, *ud = nullptr;
  int64_t nlines = 0, L = 0;
  bool tdp_ready = false;
  // restarted solver with residual monitor (KRY_GMRES_EXTENDED)
  double rtol = 1e-10;
  int max_cycles = 20;
  ExtState *ext = nullptr;    // device
  ExtState *ext_h = nullptr;  // pinned
  double *R = nullptr;        // (m+1) x m rotated Hessenberg | cs (m) | sn (m) | gv (m+1)
  int last_iters = 0, last_cycles = 0, last_converged = 0;
  double last_relres = 0.0;
  // one-sweep Arnoldi steps (no preconditioner, grid stencil operator): first step that
  // uses them (< 0: never; fparts != nullptr: available), partial sums, totals, coefficients
  int fused_from = -1;
  double *fparts = nullptr, *ftot = nullptr, *fcoef = nullptr;
  // the sweeps leave their vectors un-normalised: slot k of V holds fsigma[k] * v_k
  // (fsigma[k] = 1 for vectors stored by the chain); valid while scaled is set
  double *fsigma = nullptr;
  bool scaled = false;
  // row blocks: sweep the interior bands while the halo rows of the new vector travel
  bool overlap_halo = true;
  // Strip-major basis (StripMap): solvers of a step object on one rank keep the Krylov basis of a solve
  // strip by strip when all of its steps are one-sweep steps with generated bands, so that the sweeps'
  // workgroups stream whole consecutive 2 KB lines.  grid_*: the grid the solver was created for (0: none);
  // strips: the current basis is strip-major; v0: the first basis vector in grid order (the hand-over's
  // operator kernel reads it).
  int grid_rows = 0, grid_cols = 0;
  bool strips_capable = false, strips = false;
  double *v0 = nullptr;
  // cil: the current basis is chunk-interleaved (il1 in kry_device.h): solves on one rank whose every step is a
  // step of the Gram-Schmidt chain.  The newest vector is then kept contiguous as well, in v0, for the kernels
  // that take it as their operand (operator, line preconditioner).
  bool cil_capable = false, cil = false;
  bool cil_always = false;  // environment KRY_CHAIN_INTERLEAVE=2 when the solver is created: wherever possible
  // Between two chain steps of a fixed cycle with the line preconditioner: w holds the next basis vector
  // un-normalised and pend_part the partial sums of its squares; the preconditioner kernel of the next step
  // normalises it on the way in (k_tdp_apply2 with SC) instead of a pass of its own.
  bool scale_pending = false;
  bool fuse_scale = true;  // environment KRY_TDP_FUSE_SCALE=0 when the solver is created: off
  const double *pend_part = nullptr;
  int pend_np = 0;
  // The norm of a cycle's last vector from the last sweep's sums (launch_fused_coeffs) instead of from the
  // closing kernel: for solves with the operator the solver is bound to (bound_op: the operator of the step
  // object that owns the solver, which lives as long as the solver does), because the vector itself is then
  // formed only when somebody asks for it (kry_gmres_subspace_vector), which takes the operator again.
  // fskip: the device flag that turns the closing kernel into a no-op; last_pending: slot m of the basis
  // has not been filled by the last solve.
  kry_op *bound_op = nullptr;
  bool norm_from_sums = true;
  int *fskip = nullptr;
  bool last_pending = false;
};

// What the host keeps about the contents of a solver's basis: written when a solve is queued, and
// therefore stale after a captured step is replayed if something else used the solver in between.
struct BasisState {
  int nvec = 0;
  bool scaled = false, strips = false, last_pending = false, cil = false;
};
static BasisState basis_state(const kry_gmres *g) {
  return BasisState{g->nvec, g->scaled, g->strips, g->last_pending, g->cil};
}
static void set_basis_state(kry_gmres *g, const BasisState &b) {
  g->nvec = b.nvec;
  g->scaled = b.scaled;
  g->strips = b.strips;
  g->cil = b.cil;
  g->last_pending = b.last_pending;
}

struct kry_ie {
  kry_ctx *ctx = nullptr;
  IeGeom g;  // cx, cy global; row0/rows: the padded rows this rank owns
  IeBc bc;
  double min_x, min_y;
  int64_t nloc = 0;  // rows * (cy + 2) owned entries
  int64_t neq = 0;   // unknowns of the whole mesh
  int64_t halo = 0;  // padding in front of / behind every field
  double *fld[3] = {nullptr, nullptr, nullptr};  // first owned row (halo row in front and behind)
  double *band[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  // u and v as they were when the off-diagonals were formed (the reference never refreshes
  // them, include/time_disc.hpp:246-274), laid out like fld[]: the one-sweep kernels
  // evaluate the interior band entries from these instead of reading four bands
  double *snap[2] = {nullptr, nullptr};
  double *snap_strips[2] = {nullptr, nullptr};  // the same, strip-major (for solvers with a strip-major basis)
  double *rhs = nullptr;
  double *bcvals = nullptr;    // device, 3 * 2 (cx+cy), whole mesh
  double *bcvals_h = nullptr;  // pinned staging
  bool bc_dirty = true;
  double *wrap = nullptr;  // right boundary column of T, u, v: 3 * cy
  double *parts = nullptr;
  double *maxd = nullptr;    // device scalar
  double *maxd_h = nullptr;  // pinned
  double time = 0.0;
  kry_op *op = nullptr;
  kry_gmres *solver = nullptr;
  bool pending = false;  // work queued whose result has not been waited for
  // hipGraph replay of the step (single rank only)
  bool use_graph = false;
  hipGraphExec_t graph_exec = nullptr;
  double graph_dt = 0.0;
  uint64_t graph_epoch = 0;  // g_config_epoch at capture
  BasisState graph_basis;    // what a replay leaves in the solver's basis (recorded at capture)
  bool whole() const { return g.row0 == 0 && g.rows == g.cx + 2; }
};

// ---------------------------------------------------------------------------
extern "C" {

const char *kry_last_error(void) { return g_err.c_str(); }
const char *kry_version(void) { return "krylov_amd 0.1 (gfx950)"; }

int kry_device_count(int *count) {
  if (!count) return fail(KRY_ERR_INVALID, "count is NULL");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *count = 0;
    return fail(KRY_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = c;
  return KRY_OK;
}

int kry_ctx_create(int device, void *stream, kry_ctx **out) {
  if (!out) return fail(KRY_ERR_INVALID, "out is NULL");
  *out = nullptr;
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess || c <= 0)
    return fail(KRY_ERR_NO_DEVICE, "no HIP device is visible; the HIP path cannot run");
  if (device < 0 || device >= c) return fail(KRY_ERR_INVALID, "device %d out of range (%d devices)", device, c);
  HIPCHK(hipSetDevice(device));
  kry_ctx *ctx = new kry_ctx;
  ctx->device = device;
  if (device < kMaxDevices && scratch_of_device(device, &ctx->scratch) != KRY_OK) {
    delete ctx;
    return fail(KRY_ERR_HIP, "hipGetSymbolAddress of the library's scratch array failed on device %d", device);
  }
  if (stream) {
    ctx->stream = (hipStream_t)stream;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete ctx;
      return fail(KRY_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    ctx->own_stream = true;
  }
  *out = ctx;
  return KRY_OK;
}

static void ctx_free(kry_ctx *ctx);

static void ctx_ref(kry_ctx *ctx) { ctx->refs++; }
static void ctx_unref(kry_ctx *ctx) {
  if (--ctx->refs == 0 && ctx->released) ctx_free(ctx);
}

void kry_ctx_destroy(kry_ctx *ctx) {
  if (!ctx || ctx->released) return;
  ctx->released = true;
  if (ctx->refs == 0) ctx_free(ctx);
}

static void ctx_free(kry_ctx *ctx) {
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto &r : ctx->recs) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  for (auto e : ctx->free_events) (void)hipEventDestroy(e);
  if (ctx->nccl) (void)rccl_api().CommDestroy(ctx->nccl);
  if (ctx->comm_stream) {
    (void)hipStreamSynchronize(ctx->comm_stream);
    (void)hipStreamDestroy(ctx->comm_stream);
  }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int kry_ctx_synchronize(kry_ctx *ctx) {
  if (!ctx) return fail(KRY_ERR_INVALID, "ctx is NULL");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return KRY_OK;
}

void *kry_ctx_stream(kry_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

static int comm_stream_ready(kry_ctx *ctx);

int kry_ctx_set_comm(kry_ctx *ctx, int rank, int nranks, kry_allreduce_fn allreduce,
                     kry_sendrecv_fn sendrecv, void *user) {
  if (!ctx) return fail(KRY_ERR_INVALID, "ctx is NULL");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(KRY_ERR_INVALID, "bad rank %d of %d", rank, nranks);
  if (nranks > 1 && (!allreduce || !sendrecv)) return fail(KRY_ERR_INVALID, "collectives are required for nranks > 1");
  if (ctx->nccl) return fail(KRY_ERR_STATE, "the context already communicates through RCCL");
  ctx->rank = rank;
  ctx->nranks = nranks;
  ctx->allreduce = allreduce;
  ctx->sendrecv = sendrecv;
  ctx->comm_user = user;
  if (nranks > 1) {
    HIPCHK(hipSetDevice(ctx->device));
    KCHK(comm_stream_ready(ctx));
  }
  return KRY_OK;
}

int kry_rccl_unique_id(unsigned char id[128]) {
  if (!id) return fail(KRY_ERR_INVALID, "id is NULL");
  const RcclApi &api = rccl_api();
  if (!api.ok) return fail(KRY_ERR_STATE, "librccl could not be loaded: %s", dlerror());
  static_assert(sizeof(ncclUniqueId) == 128, "unique id size");
  ncclUniqueId u;
  NCCLCHK(api.GetUniqueId(&u));
  memcpy(id, &u, sizeof u);
  return KRY_OK;
}

int kry_ctx_set_comm_rccl(kry_ctx *ctx, int rank, int nranks, const unsigned char id[128]) {
  if (!ctx || !id) return fail(KRY_ERR_INVALID, "NULL argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(KRY_ERR_INVALID, "bad rank %d of %d", rank, nranks);
  if (ctx->nccl) return fail(KRY_ERR_STATE, "the context already communicates through RCCL");
  const RcclApi &api = rccl_api();
  if (!api.ok) return fail(KRY_ERR_STATE, "librccl could not be loaded: %s", dlerror());
  HIPCHK(hipSetDevice(ctx->device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  ncclComm_t comm = nullptr;
  NCCLCHK(api.CommInitRank(&comm, nranks, u, rank));
  ctx->nccl = comm;
  ctx->rank = rank;
  ctx->nranks = nranks;
  ctx->allreduce = nullptr;
  ctx->sendrecv = nullptr;
  ctx->comm_user = nullptr;
  KCHK(comm_stream_ready(ctx));
  return KRY_OK;
}

int kry_ctx_comm_is_native(const kry_ctx *ctx) { return ctx && ctx->nccl ? 1 : 0; }

int kry_ctx_memory_info(kry_ctx *ctx, int64_t *free_bytes, int64_t *total_bytes) {
  if (!ctx) return fail(KRY_ERR_INVALID, "ctx is NULL");
  HIPCHK(hipSetDevice(ctx->device));
  size_t f = 0, t = 0;
  HIPCHK(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = (int64_t)f;
  if (total_bytes) *total_bytes = (int64_t)t;
  return KRY_OK;
}

int kry_ctx_rank(const kry_ctx *ctx) { return ctx ? ctx->rank : 0; }
int kry_ctx_nranks(const kry_ctx *ctx) { return ctx ? ctx->nranks : 1; }

int kry_ctx_set_kernel_timing(kry_ctx *ctx, int on) {
  if (!ctx) return fail(KRY_ERR_INVALID, "ctx is NULL");
  ctx->timing = on != 0;
  return KRY_OK;
}

int kry_ctx_get_kernel_times(kry_ctx *ctx, double ms[], int launches[]) {
  if (!ctx || !ms || !launches) return fail(KRY_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < KRY_K_COUNT; k++) {
    ms[k] = 0.0;
    launches[k] = 0;
  }
  for (auto &r : ctx->recs) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
      ms[r.cls] += (double)t;
      launches[r.cls] += 1;
    }
    ctx->free_events.push_back(r.a);
    ctx->free_events.push_back(r.b);
  }
  ctx->recs.clear();
  return KRY_OK;
}

}  // extern "C"

static int dev_alloc(double **p, int64_t count) {
  *p = nullptr;
  HIPCHK(hipMalloc((void **)p, sizeof(double) * (size_t)(count > 0 ? count : 1)));
  return KRY_OK;
}

// In-place reduction of `count` doubles on the device over all ranks.
static int comm_allreduce(kry_ctx *ctx, double *buf, int64_t count, int op) {
  if (ctx->nccl) {  // also with one rank: the caller asked for RCCL
    if (ctx->nranks == 1) return KRY_OK;
    NCCLCHK(rccl_api().AllReduce(buf, buf, (size_t)count, ncclDouble, op == KRY_REDUCE_MAX ? ncclMax : ncclSum,
                                 ctx->nccl, ctx->stream));
    return KRY_OK;
  }
  if (ctx->nranks == 1) return KRY_OK;
  const int rc = ctx->allreduce(ctx->comm_user, buf, count, op, (void *)ctx->stream);
  if (rc != 0) return fail(KRY_ERR_STATE, "allreduce callback failed with %d", rc);
  return KRY_OK;
}

// Fill the halo rows of a slab vector: `local` addresses the first owned row,
// rows hold P entries, nloc entries are owned.  The lower neighbour (rank-1)
// receives my first row as its upper halo, the upper neighbour my last row.
// Through RCCL all four transfers form one group (one launch on the stream).
// st: the stream the transfers are ordered on (nullptr: the context's stream).
static int comm_halo(kry_ctx *ctx, double *local, int64_t P, int64_t nloc, hipStream_t st = nullptr) {
  if (ctx->nranks == 1) return KRY_OK;
  if (!st) st = ctx->stream;
  const int down = ctx->rank > 0 ? ctx->rank - 1 : -1;
  const int up = ctx->rank + 1 < ctx->nranks ? ctx->rank + 1 : -1;
  if (ctx->nccl) {
    const RcclApi &api = rccl_api();
    NCCLCHK(api.GroupStart());
    ncclResult_t r = ncclSuccess;
    if (down >= 0) {
      if (r == ncclSuccess) r = api.Send(local, (size_t)P, ncclDouble, down, ctx->nccl, st);
      if (r == ncclSuccess) r = api.Recv(local - P, (size_t)P, ncclDouble, down, ctx->nccl, st);
    }
    if (up >= 0) {
      if (r == ncclSuccess) r = api.Send(local + nloc - P, (size_t)P, ncclDouble, up, ctx->nccl, st);
      if (r == ncclSuccess) r = api.Recv(local + nloc, (size_t)P, ncclDouble, up, ctx->nccl, st);
    }
    const ncclResult_t e = api.GroupEnd();
    if (r != ncclSuccess) return fail(KRY_ERR_HIP, "ncclSend/ncclRecv: %s", api.GetErrorString(r));
    if (e != ncclSuccess) return fail(KRY_ERR_HIP, "ncclGroupEnd: %s", api.GetErrorString(e));
    return KRY_OK;
  }
  int rc = ctx->sendrecv(ctx->comm_user, local, down, local + nloc, up, P, (void *)st);
  if (rc == 0) rc = ctx->sendrecv(ctx->comm_user, local + nloc - P, up, local - P, down, P, (void *)st);
  if (rc != 0) return fail(KRY_ERR_STATE, "sendrecv callback failed with %d", rc);
  return KRY_OK;
}

// The second stream and its two events, created on first use.
static int comm_stream_ready(kry_ctx *ctx) {
  if (ctx->comm_stream) return KRY_OK;
  HIPCHK(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  return KRY_OK;
}

// The same exchange for several slab vectors of one shape at once (the three mesh fields
// after a ghost fill): through RCCL all their transfers form ONE group, i.e. one launch on
// the stream instead of one per vector.
static int comm_halo_multi(kry_ctx *ctx, double *const *locals, int count, int64_t P, int64_t nloc) {
  if (ctx->nranks == 1) return KRY_OK;
  if (!ctx->nccl) {
    for (int f = 0; f < count; f++) KCHK(comm_halo(ctx, locals[f], P, nloc));
    return KRY_OK;
  }
  const int down = ctx->rank > 0 ? ctx->rank - 1 : -1;
  const int up = ctx->rank + 1 < ctx->nranks ? ctx->rank + 1 : -1;
  const RcclApi &api = rccl_api();
  NCCLCHK(api.GroupStart());
  ncclResult_t r = ncclSuccess;
  for (int f = 0; f < count; f++) {
    double *local = locals[f];
    if (down >= 0) {
      if (r == ncclSuccess) r = api.Send(local, (size_t)P, ncclDouble, down, ctx->nccl, ctx->stream);
      if (r == ncclSuccess) r = api.Recv(local - P, (size_t)P, ncclDouble, down, ctx->nccl, ctx->stream);
    }
    if (up >= 0) {
      if (r == ncclSuccess) r = api.Send(local + nloc - P, (size_t)P, ncclDouble, up, ctx->nccl, ctx->stream);
      if (r == ncclSuccess) r = api.Recv(local + nloc, (size_t)P, ncclDouble, up, ctx->nccl, ctx->stream);
    }
  }
  const ncclResult_t e = api.GroupEnd();
  if (r != ncclSuccess) return fail(KRY_ERR_HIP, "ncclSend/ncclRecv: %s", api.GetErrorString(r));
  if (e != ncclSuccess) return fail(KRY_ERR_HIP, "ncclGroupEnd: %s", api.GetErrorString(e));
  return KRY_OK;
}

// One send and one receive of `count` doubles (a peer < 0: no such transfer).
static int comm_sendrecv(kry_ctx *ctx, const double *send, int send_peer, double *recv, int recv_peer, int64_t count) {
  if (ctx->nccl) {
    const RcclApi &api = rccl_api();
    NCCLCHK(api.GroupStart());
    ncclResult_t r = ncclSuccess;
    if (send_peer >= 0) r = api.Send(send, (size_t)count, ncclDouble, send_peer, ctx->nccl, ctx->stream);
    if (recv_peer >= 0 && r == ncclSuccess) r = api.Recv(recv, (size_t)count, ncclDouble, recv_peer, ctx->nccl, ctx->stream);
    const ncclResult_t e = api.GroupEnd();
    if (r != ncclSuccess) return fail(KRY_ERR_HIP, "ncclSend/ncclRecv: %s", api.GetErrorString(r));
    if (e != ncclSuccess) return fail(KRY_ERR_HIP, "ncclGroupEnd: %s", api.GetErrorString(e));
    return KRY_OK;
  }
  if (!ctx->sendrecv) return fail(KRY_ERR_STATE, "no collectives installed");
  const int rc = ctx->sendrecv(ctx->comm_user, send, send_peer, recv, recv_peer, count, (void *)ctx->stream);
  if (rc != 0) return fail(KRY_ERR_STATE, "sendrecv callback failed with %d", rc);
  return KRY_OK;
}

extern "C" int kry_ctx_comm_selftest(kry_ctx *ctx) {
  if (!ctx) return fail(KRY_ERR_INVALID, "ctx is NULL");
  if (!ctx->nccl && ctx->nranks > 1 && (!ctx->allreduce || !ctx->sendrecv))
    return fail(KRY_ERR_STATE, "no collectives installed");
  HIPCHK(hipSetDevice(ctx->device));
  const int n = ctx->nranks, me = ctx->rank;
  double *buf = nullptr;
  HIPCHK(hipMalloc((void **)&buf, 4 * sizeof(double)));
  double h[4] = {me + 1.0, me + 1.0, 100.0 + me, -1.0};
  int st = KRY_OK;
  if (hipMemcpyAsync(buf, h, sizeof h, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) st = fail(KRY_ERR_HIP, "copy failed");
  if (st == KRY_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = fail(KRY_ERR_HIP, "sync failed");
  if (st == KRY_OK) {
    if (ctx->nccl) {  // drive RCCL even in a group of one
      const RcclApi &api = rccl_api();
      ncclResult_t r = api.AllReduce(buf, buf, 1, ncclDouble, ncclSum, ctx->nccl, ctx->stream);
      if (r == ncclSuccess) r = api.AllReduce(buf + 1, buf + 1, 1, ncclDouble, ncclMax, ctx->nccl, ctx->stream);
      if (r != ncclSuccess) st = fail(KRY_ERR_HIP, "ncclAllReduce: %s", api.GetErrorString(r));
    } else if (n > 1) {
      st = comm_allreduce(ctx, buf, 1, KRY_REDUCE_SUM);
      if (st == KRY_OK) st = comm_allreduce(ctx, buf + 1, 1, KRY_REDUCE_MAX);
    }
  }
  if (st == KRY_OK && (ctx->nccl || n > 1)) st = comm_sendrecv(ctx, buf + 2, (me + 1) % n, buf + 3, (me + n - 1) % n, 1);
  if (st == KRY_OK && hipMemcpyAsync(h, buf, sizeof h, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) st = fail(KRY_ERR_HIP, "copy failed");
  if (st == KRY_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = fail(KRY_ERR_HIP, "sync failed");
  (void)hipFree(buf);
  if (st != KRY_OK) return st;
  const double want_sum = 0.5 * n * (n + 1.0), want_max = (double)n, want_tok = 100.0 + (me + n - 1) % n;
  if (h[0] != want_sum || h[1] != want_max || ((ctx->nccl || n > 1) && h[3] != want_tok))
    return fail(KRY_ERR_STATE, "collectives returned sum %g (want %g), max %g (want %g), ring token %g (want %g)", h[0],
                want_sum, h[1], want_max, h[3], want_tok);
  return KRY_OK;
}

static int launch_ok(const char *what) {
  const char *prep = "";
  hipError_t e = take_launch_failure(&prep);
  if (e != hipSuccess) return fail(KRY_ERR_HIP, "%s could not be set in %s: %s", prep, what, hipGetErrorString(e));
  e = hipGetLastError();
  if (e != hipSuccess) return fail(KRY_ERR_HIP, "kernel launch failed in %s: %s", what, hipGetErrorString(e));
  return KRY_OK;
}

extern "C" {

int kry_selftest_shared_division(kry_ctx *ctx, double dx, double dy, uint64_t seed, int64_t count, int mode,
                                 int64_t *mismatches, int64_t *fast_groups) {
  if (!ctx || !mismatches) return fail(KRY_ERR_INVALID, "NULL argument");
  if (count < 0 || mode < 0 || mode > 3) return fail(KRY_ERR_INVALID, "count must not be negative, mode in 0..3");
  HIPCHK(hipSetDevice(ctx->device));
  unsigned long long *dev = nullptr, h[2] = {0, 0};
  HIPCHK(hipMalloc((void **)&dev, sizeof(h)));
  int st = KRY_OK;
  if (hipMemsetAsync(dev, 0, sizeof(h), ctx->stream) != hipSuccess) st = fail(KRY_ERR_HIP, "hipMemsetAsync failed");
  if (st == KRY_OK) {
    launch_selftest_division(ctx->stream, dx, dy, seed, count, mode, dev);
    st = launch_ok("kry_selftest_shared_division");
  }
  if (st == KRY_OK && hipMemcpyAsync(h, dev, sizeof(h), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
    st = fail(KRY_ERR_HIP, "hipMemcpyAsync failed");
  if (st == KRY_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = fail(KRY_ERR_HIP, "sync failed");
  (void)hipFree(dev);
  if (st != KRY_OK) return st;
  *mismatches = (int64_t)h[0];
  if (fast_groups) *fast_groups = (int64_t)h[1];
  return KRY_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Operators
// ---------------------------------------------------------------------------
extern "C" {

int kry_op_create_stencil(kry_ctx *ctx, int rows, int cols, const double *w, const double *s,
                          const double *c, const double *n, const double *e, kry_op **out) {
  if (!ctx || !out || !w || !s || !c || !n || !e) return fail(KRY_ERR_INVALID, "NULL argument");
  if (rows < 1 || cols < 1) return fail(KRY_ERR_INVALID, "rows/cols must be positive");
  HIPCHK(hipSetDevice(ctx->device));
  kry_op *op = new kry_op;
  op->ctx = ctx;
  ctx_ref(ctx);
  op->kind = 0;
  op->rows = rows;
  op->cols = cols;
  op->n = (int64_t)rows * cols;
  op->own_bands = true;
  op->wrap_free = true;
  for (int r = 0; r < rows && op->wrap_free; r++)
    if (s[(int64_t)r * cols] != 0.0 || n[(int64_t)r * cols + cols - 1] != 0.0) op->wrap_free = false;
  const double *src[5] = {w, s, c, n, e};
  for (int b = 0; b < 5; b++) {
    int st = dev_alloc(&op->band[b], op->n);
    if (st == KRY_OK && hipMemcpy(op->band[b], src[b], sizeof(double) * op->n, hipMemcpyHostToDevice) != hipSuccess)
      st = fail(KRY_ERR_HIP, "band upload failed");
    if (st != KRY_OK) {
      kry_op_destroy(op);
      return st;
    }
  }
  *out = op;
  return KRY_OK;
}

int kry_op_create_csr(kry_ctx *ctx, int64_t n, const int64_t *rowptr, const int32_t *col,
                      const double *val, kry_op **out) {
  if (!ctx || !out || !rowptr || (!col && rowptr[n] > 0) || (!val && rowptr[n] > 0))
    return fail(KRY_ERR_INVALID, "NULL argument");
  if (n < 1) return fail(KRY_ERR_INVALID, "n must be positive");
  if (rowptr[0] != 0) return fail(KRY_ERR_INVALID, "rowptr[0] must be 0");
  for (int64_t r = 0; r < n; r++) {
    if (rowptr[r + 1] < rowptr[r]) return fail(KRY_ERR_INVALID, "rowptr not monotone at row %lld", (long long)r);
    for (int64_t p = rowptr[r]; p < rowptr[r + 1]; p++) {
      if (col[p] < 0 || col[p] >= n) return fail(KRY_ERR_INVALID, "column index out of range in row %lld", (long long)r);
      if (p > rowptr[r] && col[p] <= col[p - 1])
        return fail(KRY_ERR_INVALID, "columns of row %lld are not strictly ascending", (long long)r);
    }
  }
  HIPCHK(hipSetDevice(ctx->device));
  kry_op *op = new kry_op;
  op->ctx = ctx;
  ctx_ref(ctx);
  op->kind = 1;
  op->n = n;
  op->nnz = rowptr[n];
  const size_t nz = (size_t)(op->nnz > 0 ? op->nnz : 1);
  if (hipMalloc((void **)&op->rowptr, sizeof(int64_t) * (size_t)(n + 1)) != hipSuccess ||
      hipMalloc((void **)&op->col, sizeof(int32_t) * nz) != hipSuccess ||
      hipMalloc((void **)&op->val, sizeof(double) * nz) != hipSuccess ||
      hipMemcpy(op->rowptr, rowptr, sizeof(int64_t) * (size_t)(n + 1), hipMemcpyHostToDevice) != hipSuccess ||
      (op->nnz > 0 && hipMemcpy(op->col, col, sizeof(int32_t) * nz, hipMemcpyHostToDevice) != hipSuccess) ||
      (op->nnz > 0 && hipMemcpy(op->val, val, sizeof(double) * nz, hipMemcpyHostToDevice) != hipSuccess)) {
    kry_op_destroy(op);
    return fail(KRY_ERR_HIP, "CSR upload failed");
  }
  *out = op;
  return KRY_OK;
}

int kry_op_create_dense(kry_ctx *ctx, int64_t n, const double *a, kry_op **out) {
  if (!ctx || !out || !a) return fail(KRY_ERR_INVALID, "NULL argument");
  if (n < 1) return fail(KRY_ERR_INVALID, "n must be positive");
  std::vector<int64_t> rowptr((size_t)n + 1, 0);
  std::vector<int32_t> col;
  std::vector<double> val;
  for (int64_t r = 0; r < n; r++) {
    for (int64_t c = 0; c < n; c++) {
      const double v = a[r * n + c];
      if (v != 0.0) {  // NaN != 0 is true: non-finite entries are kept
        col.push_back((int32_t)c);
        val.push_back(v);
      }
    }
    rowptr[(size_t)r + 1] = (int64_t)col.size();
  }
  return kry_op_create_csr(ctx, n, rowptr.data(), col.data(), val.data(), out);
}

void kry_op_destroy(kry_op *op) {
  if (!op) return;
  (void)hipSetDevice(op->ctx->device);
  if (op->own_bands)
    for (int b = 0; b < 5; b++) (void)hipFree(op->band[b]);
  (void)hipFree(op->rowptr);
  (void)hipFree(op->col);
  (void)hipFree(op->val);
  (void)hipFree(op->tx);
  (void)hipFree(op->ty);
  kry_ctx *ctx = op->ctx;
  delete op;
  ctx_unref(ctx);
}

int64_t kry_op_size(const kry_op *op) { return op ? op->n : 0; }

}  // extern "C"

// y = A x; with kc > 0 also the partial sums of the first Gram-Schmidt group
static void op_apply_async(kry_op *op, const double *x, double *y, int kc, const double *cbase,
                           int64_t ld, double *part, int *nparts, bool x_has_halo, int64_t pitch = 0) {
  hipStream_t st = op->ctx->stream;
  if (op->kind == 0) {
    Bands b{op->band[0], op->band[1], op->band[2], op->band[3], op->band[4]};
    // operators assembled by a step object: the band entries evaluated from two velocity snapshots instead of
    // five streamed bands (same bits; experimentation knob KRY_SPMV_GEN=0 reads the bands)
    // Only where the apply streams enough partner vectors for the evaluation (eight divisions per lane and
    // round) to hide behind the memory traffic: with a single partner the kernel takes 0.79 instead of 0.61 ms
    // on an 8192 x 8192 grid; from four partners on it pays (GMRES(50) with the line preconditioner on that
    // grid: 355.5 ms per step with stored bands, 348.0 from four partners on, 348.5 from eight, 350.0 from sixteen).
    // KRY_SPMV_GEN=k: from k partners on (0: never).
    static const int gen_min_kc = getenv("KRY_SPMV_GEN") ? atoi(getenv("KRY_SPMV_GEN")) : 4;
    const BandGen *gen = (gen_min_kc > 0 && kc >= gen_min_kc && op->has_gen && op->use_gen) ? &op->gen : nullptr;
    launch_spmv_stencil(st, b, op->n, op->cols, x, y, kc, cbase, ld, part, nparts, x_has_halo, gen, pitch);
  } else {
    launch_spmv_csr(st, op->rowptr, op->col, op->val, op->n, x, y, kc, cbase, ld, part, nparts, pitch);
  }
}

extern "C" {

int kry_op_apply_device(kry_op *op, const double *x_dev, double *y_dev) {
  if (!op || !x_dev || !y_dev) return fail(KRY_ERR_INVALID, "NULL argument");
  if (x_dev == y_dev) return fail(KRY_ERR_INVALID, "x and y must not alias");
  HIPCHK(hipSetDevice(op->ctx->device));
  {
    ScopedKernelTimer t(op->ctx, KRY_K_SPMV);
    op_apply_async(op, x_dev, y_dev, 0, nullptr, 0, nullptr, nullptr, false);
  }
  return launch_ok("kry_op_apply_device");
}

int kry_op_apply(kry_op *op, const double *x_host, double *y_host) {
  if (!op || !x_host || !y_host) return fail(KRY_ERR_INVALID, "NULL argument");
  if (op->slab && op->ctx->nranks > 1)
    return fail(KRY_ERR_STATE, "a slab operator needs its neighbours' rows; use the solver entry points");
  HIPCHK(hipSetDevice(op->ctx->device));
  if (!op->tx) KCHK(dev_alloc(&op->tx, op->n));
  if (!op->ty) KCHK(dev_alloc(&op->ty, op->n));
  hipStream_t st = op->ctx->stream;
  HIPCHK(hipMemcpyAsync(op->tx, x_host, sizeof(double) * op->n, hipMemcpyHostToDevice, st));
  op_apply_async(op, op->tx, op->ty, 0, nullptr, 0, nullptr, nullptr, false);
  KCHK(launch_ok("kry_op_apply"));
  HIPCHK(hipMemcpyAsync(y_host, op->ty, sizeof(double) * op->n, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return KRY_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// GMRES
// ---------------------------------------------------------------------------
static double *basis(kry_gmres *g, int i) { return g->V + (int64_t)i * g->ld + g->halo; }
// Vector i of a strip-major basis (StripMap): the rows of the m + 1 vectors are interleaved, vector i starts i rows
// of a strip into the array.
static double *strip_vec(kry_gmres *g, int i) { return g->V + g->halo + (int64_t)i * kBlock; }
// Vector i as the chain's kernels address it, the distance between two vectors and between two chunks of one
// vector (0: contiguous vectors)
static double *chain_vec(kry_gmres *g, int i) { return g->cil ? g->V + g->halo + (int64_t)i * kChunk : basis(g, i); }
static int64_t chain_ld(const kry_gmres *g) { return g->cil ? (int64_t)kChunk : g->ld; }
static int64_t chain_pitch(const kry_gmres *g) { return g->cil ? (int64_t)(g->m + 1) * kChunk : 0; }

// Latest step at which the one-sweep kernels can take over: the hand-over applies the
// operator with the dots of all earlier vectors in one kernel (at most kMgsBlock of
// them), and at least one full sweep has to follow before the closing one.
static int fused_from_max(int m) { return m - 2 < kMgsBlock - 1 ? m - 2 : kMgsBlock - 1; }
static bool norm_from_sums(const kry_gmres *g, const kry_op *op);

extern "C" {

static int gmres_create(kry_ctx *ctx, int64_t n, int m, int flags, int64_t halo_row, kry_gmres **out, int grid_rows = 0,
                        int grid_cols = 0);

int kry_gmres_create(kry_ctx *ctx, int64_t n, int m, int flags, kry_gmres **out) {
  return gmres_create(ctx, n, m, flags, 0, out);
}

// grid_rows x grid_cols: the grid of the step object the solver belongs to (0: a stand-alone solver)
static int gmres_create(kry_ctx *ctx, int64_t n, int m, int flags, int64_t halo_row, kry_gmres **out, int grid_rows,
                        int grid_cols) {
  if (!ctx || !out) return fail(KRY_ERR_INVALID, "NULL argument");
  *out = nullptr;
  if (n < 1) return fail(KRY_ERR_INVALID, "system size must be positive");
  if (m < 1 || m > kMaxM) return fail(KRY_ERR_INVALID, "num_iters must be in [1, %d]", kMaxM);
  HIPCHK(hipSetDevice(ctx->device));
  kry_gmres *g = new kry_gmres;
  g->ctx = ctx;
  ctx_ref(ctx);
  g->n = n;
  g->halo_row = halo_row;
  g->halo = (halo_row + 31) / 32 * 32;
  // 256-byte multiples: every vector starts 16-byte aligned, halo included
  g->ld = g->halo + (n + 31) / 32 * 32 + g->halo;
  if (ctx->nranks == 1 && !(getenv("KRY_CHAIN_INTERLEAVE") && atoi(getenv("KRY_CHAIN_INTERLEAVE")) == 0)) {
    // whole chunks, so that the m + 1 vectors fit the same array chunk-interleaved
    g->cil_capable = true;
    g->cil_always = getenv("KRY_CHAIN_INTERLEAVE") && atoi(getenv("KRY_CHAIN_INTERLEAVE")) == 2;
    g->ld = g->halo + (n + kChunk - 1) / kChunk * kChunk + g->halo;
  }
  if (getenv("KRY_TDP_FUSE_SCALE") && atoi(getenv("KRY_TDP_FUSE_SCALE")) == 0) g->fuse_scale = false;  // experimentation knob
  g->m = m;
  g->flags = flags;
  const bool one_sweep_possible = !(flags & KRY_GMRES_TDP) && m <= kFusedMaxM && m >= 3;
  // (the kernels hold positions within a strip's block of the interleaved basis, and within a strip-major snapshot,
  // in 32-bit integers, and reach one block to either side)
  if (one_sweep_possible && ctx->nranks == 1 && grid_rows > 0 && (int64_t)grid_rows * grid_cols == n &&
      (int64_t)(grid_rows + 2) * (m + 1) * kBlock < 1000000000LL && strip_major_size(grid_rows, grid_cols) < 2147483000LL &&
      !(getenv("KRY_STRIP_MAJOR") && atoi(getenv("KRY_STRIP_MAJOR")) == 0)) {
    // room for the strip-major form of every basis vector
    g->grid_rows = grid_rows;
    g->grid_cols = grid_cols;
    g->strips_capable = true;
    const int64_t sms = (strip_major_size(grid_rows, grid_cols) + 31) / 32 * 32;
    if (g->halo + sms + g->halo > g->ld) g->ld = g->halo + sms + g->halo;
  }
  // Thirty-two links per sweep: 16.5 bytes per unknown and link (the limit for ever larger
  // groups is 16) and 1/32 of the reductions (all-reduces on several ranks) of the plain
  // chain.
  g->kblock = kMgsBlock;
  if (const char *e = getenv("KRY_HALO_OVERLAP")) g->overlap_halo = atoi(e) != 0;  // experimentation knob
  if (const char *e = getenv("KRY_MGS_BLOCK")) {  // experimentation knob, same range as the setter
    const int k = atoi(e);
    if (mgs_block_supported(k)) g->kblock = k;
  }
  int st = KRY_OK;
  if (st == KRY_OK) st = dev_alloc(&g->V, (int64_t)(m + 1) * g->ld);
  if (st == KRY_OK) st = dev_alloc(&g->H, (int64_t)(m + 1) * m);
  if (st == KRY_OK) st = dev_alloc(&g->w, g->ld);
  if (st == KRY_OK) st = dev_alloc(&g->vars, g->ld);
  if (st == KRY_OK) st = dev_alloc(&g->parts, 2 * (int64_t)kMaxRed * kMaxParts);
  if (st == KRY_OK) st = dev_alloc(&g->tot, kMaxRed);
  if (st == KRY_OK) st = dev_alloc(&g->G, (int64_t)(m + 1) * (m + 1));
  if (st == KRY_OK) st = dev_alloc(&g->beta, 1);
  if (st == KRY_OK) st = dev_alloc(&g->y, m);
  if (st == KRY_OK) st = dev_alloc(&g->work, 3 * (int64_t)m * m + 3 * m);
  if (st == KRY_OK && (flags & KRY_GMRES_EXTENDED)) {
    st = dev_alloc(&g->R, (int64_t)(m + 1) * m + 3 * (int64_t)m + 1);
    if (st == KRY_OK && hipMalloc((void **)&g->ext, sizeof(ExtState)) != hipSuccess)
      st = fail(KRY_ERR_HIP, "hipMalloc of the solver state failed");
    if (st == KRY_OK && hipHostMalloc((void **)&g->ext_h, sizeof(ExtState)) != hipSuccess)
      st = fail(KRY_ERR_HIP, "hipHostMalloc of the solver state failed");
    if (st == KRY_OK) {
      (void)hipMemsetAsync(g->ext, 0, sizeof(ExtState), ctx->stream);
      (void)hipMemsetAsync(g->R, 0, sizeof(double) * ((size_t)(m + 1) * m + 3 * (size_t)m + 1), ctx->stream);
    }
  }
  if (st == KRY_OK && (flags & KRY_GMRES_TDP)) {
    st = dev_alloc(&g->z, g->ld);
    if (st == KRY_OK) st = dev_alloc(&g->dd, g->ld);
    if (st == KRY_OK) st = dev_alloc(&g->ud, g->ld);
  }
  if (st == KRY_OK && (g->strips_capable || g->cil_capable)) st = dev_alloc(&g->v0, g->ld);
  if (st == KRY_OK && one_sweep_possible) {
    // hand-over from the chain: after two steps, or at once where the basis can be strip-major
    int from = g->strips_capable ? 0 : 2;
    if (const char *e = getenv("KRY_FUSED_FROM")) from = atoi(e);
    if (from > fused_from_max(m)) from = fused_from_max(m);
    st = dev_alloc(&g->fsigma, kFusedCoef);
    if (st == KRY_OK) st = dev_alloc(&g->fparts, (int64_t)kFusedRed * kMaxParts);
    if (st == KRY_OK) st = dev_alloc(&g->ftot, kFusedRed);
    if (st == KRY_OK) st = dev_alloc(&g->fcoef, kFusedCoef);
    if (st == KRY_OK && hipMalloc((void **)&g->fskip, 256) != hipSuccess) st = fail(KRY_ERR_HIP, "hipMalloc of a flag failed");
    if (st == KRY_OK) (void)hipMemsetAsync(g->fskip, 0, 256, ctx->stream);
    if (const char *e = getenv("KRY_NORM_FROM_SUMS")) g->norm_from_sums = atoi(e) != 0;  // experimentation knob
    if (st == KRY_OK) g->fused_from = from < 0 ? -1 : from;
  }
  if (st != KRY_OK) {
    kry_gmres_destroy(g);
    return st;
  }
  hipStream_t s = ctx->stream;
  (void)hipMemsetAsync(g->H, 0, sizeof(double) * (size_t)(m + 1) * m, s);
  (void)hipMemsetAsync(g->G, 0, sizeof(double) * (size_t)(m + 1) * (m + 1), s);
  (void)hipMemsetAsync(g->V, 0, sizeof(double) * (size_t)(m + 1) * g->ld, s);
  (void)hipMemsetAsync(g->w, 0, sizeof(double) * (size_t)g->ld, s);
  (void)hipMemsetAsync(g->vars, 0, sizeof(double) * (size_t)g->ld, s);
  if (g->v0) (void)hipMemsetAsync(g->v0, 0, sizeof(double) * (size_t)g->ld, s);
  if (g->z) (void)hipMemsetAsync(g->z, 0, sizeof(double) * (size_t)g->ld, s);
  if (g->fsigma) launch_fill(s, g->fsigma, 1.0, kFusedCoef);
  // work vectors are addressed at their first owned entry from here on
  g->w += g->halo;
  g->vars += g->halo;
  if (g->z) g->z += g->halo;
  if (g->v0) g->v0 += g->halo;
  g->offset_applied = true;
  launch_fill(s, g->vars, kSNaN, n);  // Solver::Solver, include/solver.hpp:16-18
  st = launch_ok("kry_gmres_create");
  if (st == KRY_OK && hipStreamSynchronize(s) != hipSuccess) st = fail(KRY_ERR_HIP, "sync failed");
  if (st != KRY_OK) {
    kry_gmres_destroy(g);
    return st;
  }
  *out = g;
  return KRY_OK;
}

void kry_gmres_destroy(kry_gmres *g) {
  if (!g) return;
  (void)hipSetDevice(g->ctx->device);
  (void)hipStreamSynchronize(g->ctx->stream);
  const int64_t off = g->offset_applied ? g->halo : 0;
  double *bufs[] = {g->V, g->H, g->w ? g->w - off : nullptr, g->z ? g->z - off : nullptr,
                    g->v0 ? g->v0 - off : nullptr, g->vars ? g->vars - off : nullptr, g->parts, g->tot, g->beta, g->y, g->work, g->bbuf,
                    g->dd, g->ud, g->R, g->G, g->fparts, g->ftot, g->fcoef, g->fsigma};
  for (double *p : bufs) (void)hipFree(p);
  if (g->ext) (void)hipFree(g->ext);
  if (g->fskip) (void)hipFree(g->fskip);
  if (g->ext_h) (void)hipHostFree(g->ext_h);
  if (g->own_tri) {
    (void)hipFree(g->lo);
    (void)hipFree(g->di);
    (void)hipFree(g->up);
  }
  kry_ctx *ctx = g->ctx;
  delete g;
  ctx_unref(ctx);
}

int kry_gmres_set_mgs_block(kry_gmres *g, int k) {
  if (!g) return fail(KRY_ERR_INVALID, "NULL argument");
  if (!mgs_block_supported(k)) return fail(KRY_ERR_INVALID, "block size must be one of 1, 2, 3, 4, 8, 16, 32");
  g->kblock = k;
  config_changed();
  return KRY_OK;
}

int kry_gmres_get_mgs_block(const kry_gmres *g) { return g ? g->kblock : 0; }

int kry_gmres_set_one_sweep(kry_gmres *g, int first_step) {
  if (!g) return fail(KRY_ERR_INVALID, "NULL argument");
  config_changed();
  if (first_step < 0) {
    g->fused_from = -1;
    return KRY_OK;
  }
  if (!g->fparts)
    return fail(KRY_ERR_STATE, "one-sweep steps need a solver without preconditioner and 3 <= num_iters <= %d", kFusedMaxM);
  g->fused_from = first_step > fused_from_max(g->m) ? fused_from_max(g->m) : first_step;
  return KRY_OK;
}

int kry_gmres_set_norm_from_sums(kry_gmres *g, int on) {
  if (!g) return fail(KRY_ERR_INVALID, "NULL argument");
  config_changed();
  g->norm_from_sums = on != 0;
  return KRY_OK;
}

int kry_gmres_get_norm_from_sums(const kry_gmres *g) {
  return g && g->fused_from >= 0 && g->bound_op && norm_from_sums(g, g->bound_op) ? 1 : 0;
}

int kry_gmres_last_norm_from_sums(kry_gmres *g, int *used) {
  if (!g || !used) return fail(KRY_ERR_INVALID, "NULL argument");
  *used = 0;
  if (!g->last_pending) return KRY_OK;  // the last solve did not take the shortcut at all
  HIPCHK(hipSetDevice(g->ctx->device));
  HIPCHK(hipMemcpyAsync(used, g->fskip, sizeof(int), hipMemcpyDeviceToHost, g->ctx->stream));
  HIPCHK(hipStreamSynchronize(g->ctx->stream));
  return KRY_OK;
}

int kry_set_one_sweep_split(int first_step) {
  config_changed();
  set_split_from(first_step);
  return get_split_from();
}

int kry_get_one_sweep_split(int *first_step, int *last_step) {
  if (first_step) *first_step = get_split_from();
  if (last_step) *last_step = get_split_upto();
  return KRY_OK;
}

// The hand-over step in effect.  The first sweep recovers its coefficients from the Gram
// matrix of ALL vectors stored so far, and the chain only carries the Gram entries of
// vectors that share a group of kblock links: the hand-over happens while the basis is
// still within the chain's first group.
static int fused_first(const kry_gmres *g) {
  if (g->fused_from < 0) return -1;
  return g->fused_from < g->kblock - 1 ? g->fused_from : g->kblock - 1;
}

int kry_gmres_get_one_sweep(const kry_gmres *g) { return g && g->fparts ? fused_first(g) : -1; }

int kry_gmres_basis_strip_major(const kry_gmres *g) { return g && g->strips ? 1 : 0; }
int kry_gmres_basis_chunk_interleaved(const kry_gmres *g) { return g && g->cil ? 1 : 0; }

int kry_gmres_set_tolerance(kry_gmres *g, double rtol, int max_cycles) {
  if (!g) return fail(KRY_ERR_INVALID, "NULL argument");
  if (!(g->flags & KRY_GMRES_EXTENDED)) return fail(KRY_ERR_STATE, "solver was created without KRY_GMRES_EXTENDED");
  if (!(rtol >= 0.0) || max_cycles < 1) return fail(KRY_ERR_INVALID, "rtol must be >= 0 and max_cycles >= 1");
  g->rtol = rtol;
  g->max_cycles = max_cycles;
  return KRY_OK;
}

int kry_gmres_last_stats(kry_gmres *g, int *iterations, int *cycles, int *converged, double *rel_residual) {
  if (!g) return fail(KRY_ERR_INVALID, "NULL argument");
  if (!(g->flags & KRY_GMRES_EXTENDED)) return fail(KRY_ERR_STATE, "solver was created without KRY_GMRES_EXTENDED");
  if (iterations) *iterations = g->last_iters;
  if (cycles) *cycles = g->last_cycles;
  if (converged) *converged = g->last_converged;
  if (rel_residual) *rel_residual = g->last_relres;
  return KRY_OK;
}

int64_t kry_gmres_size(const kry_gmres *g) { return g ? g->n : 0; }
int kry_gmres_iterations(const kry_gmres *g) { return g ? g->m : 0; }
const double *kry_gmres_result_device(kry_gmres *g) { return g ? g->vars : nullptr; }

}  // extern "C"

static int check_op(kry_gmres *g, kry_op *op) {
  if (!g || !op) return fail(KRY_ERR_INVALID, "NULL argument");
  if (op->n != g->n) return fail(KRY_ERR_INVALID, "operator size %lld != solver size %lld", (long long)op->n, (long long)g->n);
  if (op->ctx != g->ctx) return fail(KRY_ERR_INVALID, "operator and solver belong to different contexts");
  return KRY_OK;
}

// GMRESTDPSolver::assemble_preconditioner + the rhs-independent half of
// SparseTriDiagSolver::solve.
static int tdp_setup_async(kry_gmres *g, kry_op *op) {
  hipStream_t st = g->ctx->stream;
  if (op->kind == 0) {
    if (g->own_tri) return fail(KRY_ERR_STATE, "solver was set up for a CSR operator");
    g->lo = op->band[1];
    g->di = op->band[2];
    g->up = op->band[3];
    g->nlines = op->rows;
    g->L = op->cols;
  } else {
    if (!g->own_tri) {
      KCHK(dev_alloc(&g->lo, g->ld));
      KCHK(dev_alloc(&g->di, g->ld));
      KCHK(dev_alloc(&g->up, g->ld));
      g->own_tri = true;
    }
    launch_tdp_extract_csr(st, op->rowptr, op->col, op->val, op->n, g->lo, g->di, g->up);
    g->nlines = 1;
    g->L = g->n;
  }
  {
    ScopedKernelTimer t(g->ctx, KRY_K_PRECOND);
    launch_tdp_factor(st, g->lo, g->di, g->up, g->dd, g->ud, g->nlines, g->L);
  }
  g->tdp_ready = true;
  return launch_ok("tdp_setup");
}

// On several ranks the per-workgroup partial sums of nq quantities are
// collapsed to nq totals and summed over the ranks; the consumer kernel then
// reads those (np = 1).  On one rank the consumer sums the partials itself.
static int finish_parts(kry_gmres *g, const double **part, int nq, int *np) {
  if (g->ctx->nranks == 1) return KRY_OK;
  launch_reduce_parts(g->ctx->stream, *part, nq, *np, g->tot);
  KCHK(comm_allreduce(g->ctx, g->tot, nq, KRY_REDUCE_SUM));
  *part = g->tot;
  *np = 1;
  return KRY_OK;
}

// v_0 = src / sqrt(sum of part), *out = that norm.  With a strip-major basis the vector is kept in grid
// order as well (g->v0), for the hand-over's operator kernel.
static void store_v0(kry_gmres *g, const double *src, const double *part, int np, double *out) {
  hipStream_t st = g->ctx->stream;
  if (g->strips) {
    launch_scale_to_strips(st, src, g->v0, strip_vec(g, 0), part, np, g->grid_rows, g->grid_cols, g->m + 1, out);
  } else {
    launch_scale_from(st, src, chain_vec(g, 0), part, np, g->n, out, chain_pitch(g), g->cil ? g->v0 : nullptr);
  }
}

static int initial_subspace_async(kry_gmres *g, const double *b_dev) {
  hipStream_t st = g->ctx->stream;
  int np = 0;
  {
    ScopedKernelTimer t(g->ctx, KRY_K_NORMALIZE);
    launch_nrm2_part(st, b_dev, g->n, g->parts, &np);
    const double *pin = g->parts;
    KCHK(finish_parts(g, &pin, 1, &np));
    store_v0(g, b_dev, pin, np, g->beta);
  }
  g->nvec = 1;
  g->scaled = false;
  g->last_pending = false;
  g->scale_pending = false;
  // Every vector starts a cycle normalised; the one-sweep steps overwrite the scale factors
  // of the slots they fill.  A monitored solve may stop before they run (their kernels,
  // the hand-over included, are predicated), so there the factors are reset here; a fixed
  // cycle always runs the hand-over, which resets them itself.
  if (g->fsigma && (g->flags & KRY_GMRES_EXTENDED)) launch_fill(st, g->fsigma, 1.0, kFusedCoef);
  return launch_ok("initial_subspace");
}

// One Arnoldi step with the fused modified Gram-Schmidt chain.  In the
// restarted solver every kernel of the step is predicated on the device flag
// that the residual monitor raises.
// Whether the chain steps of a fixed cycle leave the normalisation of their vector to the next step's
// preconditioner kernel
static bool tdp_fuses_scale(const kry_gmres *g) {
  // (one rank: on several the norm arrives as an all-reduced total, a combination no test on a mesh of the size the
  // 16-byte-lane kernel needs covers)
  return g->fuse_scale && g->ctx->nranks == 1 && (g->flags & KRY_GMRES_TDP) && !(g->flags & KRY_GMRES_EXTENDED) &&
         g->tdp_ready && g->nlines * g->L == g->n &&
         tdp_scaled_supported(g->lo, g->dd, g->ud, g->w, g->z, g->V + g->halo, g->nlines, g->L);
}

// Whether a solve with this operator keeps its basis chunk-interleaved: every step runs the chain on one rank,
// and the layout pays -- a kernel of the chain reads more vectors than a compute unit's translation caches
// cover (from 32 steps per cycle on), or the copy of the newest vector that the layout otherwise needs falls
// away because the preconditioner kernel normalises it on the way in.  (GMRES(30) on a CSR matrix, 8192^2:
// 153.5 ms per cycle with contiguous vectors, 156.4 interleaved.)
static bool fused_usable(const kry_gmres *g, const kry_op *op);
static bool chain_interleaved(const kry_gmres *g, const kry_op *op) {
  if (!g->cil_capable || g->ctx->nranks != 1 || fused_usable(g, op)) return false;
  return g->cil_always || g->m >= 32 || tdp_fuses_scale(g);
}

// another_follows: the caller queues the next chain step of the same cycle right behind this one (the
// normalisation of this step's vector may then be left to that step's preconditioner kernel).
static int arnoldi_async(kry_gmres *g, kry_op *op, bool another_follows = false) {
  LaunchPredicate pred((g->flags & KRY_GMRES_EXTENDED) && g->ext ? &g->ext->stop : nullptr);
  if (g->nvec < 1) return fail(KRY_ERR_STATE, "add_arnoldi before initial_subspace");
  const int j = g->nvec - 1;
  if (j >= g->m) return fail(KRY_ERR_STATE, "subspace is full (%d vectors for num_iters = %d)", g->nvec, g->m);
  hipStream_t st = g->ctx->stream;
  const int64_t n = g->n;
  double *part[2] = {g->parts, g->parts + (int64_t)kMaxRed * kMaxParts};
  int np[2] = {0, 0};
  const double *src = g->cil ? g->v0 : basis(g, j);
  if (g->flags & KRY_GMRES_TDP) {
    if (!g->tdp_ready) return fail(KRY_ERR_STATE, "preconditioner has not been assembled");
    ScopedKernelTimer t(g->ctx, KRY_K_PRECOND);
    if (g->scale_pending) {
      // v_j = w / ||w||, H(j, j-1) = ||w|| and z = M^-1 v_j in one kernel
      TdpScale sc;
      sc.part = g->pend_part;
      sc.np = g->pend_np;
      sc.vout = chain_vec(g, j);
      sc.pitch = chain_pitch(g);
      sc.nout = g->H + (int64_t)j * g->m + (j - 1);
      launch_tdp_apply_scaled(st, g->lo, g->dd, g->ud, g->w, g->z, g->nlines, g->L, sc);
      g->scale_pending = false;
    } else {
      launch_tdp_apply(st, g->lo, g->dd, g->ud, src, g->z, g->nlines, g->L);
    }
    src = g->z;
  }
  // Orthogonalise against v_0..v_j in groups of up to K vectors (modified
  // Gram-Schmidt order is kept: group after group, vector after vector).
  const int K = g->kblock;
  const int nv = j + 1;
  int ps = 0, kp = nv < K ? nv : K;  // previous group: start, size
  if (g->halo_row > 0) KCHK(comm_halo(g->ctx, const_cast<double *>(src), g->halo_row, n));
  {
    // w = A v_j fused with the dots of the first group
    ScopedKernelTimer t(g->ctx, KRY_K_SPMV);
    op_apply_async(op, src, g->w, kp, chain_vec(g, 0), chain_ld(g), part[0], &np[0], g->halo_row > 0, chain_pitch(g));
  }
  const double *pin = part[0];
  int npin = np[0];
  KCHK(finish_parts(g, &pin, 2 * kp - 1, &npin));
  int cur = 1;
  for (int cs = kp; cs < nv; cs += K, cur++) {
    const int kc = nv - cs < K ? nv - cs : K;
    {
      ScopedKernelTimer t(g->ctx, (kp == K && kc == K) ? KRY_K_MGS : KRY_K_MGS_PART);
      launch_mgs_block(st, kp, kc, g->w, chain_vec(g, ps), chain_vec(g, cs), chain_ld(g), pin, npin, part[cur & 1],
                       &np[cur & 1], g->H + (int64_t)ps * g->m + j, g->m, n, g->G, g->m + 1, ps, chain_pitch(g));
    }
    pin = part[cur & 1];
    npin = np[cur & 1];
    KCHK(finish_parts(g, &pin, 2 * kc - 1, &npin));
    ps = cs;
    kp = kc;
  }
  {
    ScopedKernelTimer t(g->ctx, KRY_K_MGS_LAST);
    launch_mgs_block(st, kp, 0, g->w, chain_vec(g, ps), nullptr, chain_ld(g), pin, npin, part[cur & 1], &np[cur & 1],
                     g->H + (int64_t)ps * g->m + j, g->m, n, g->G, g->m + 1, ps, chain_pitch(g));
  }
  pin = part[cur & 1];
  npin = np[cur & 1];
  KCHK(finish_parts(g, &pin, 1, &npin));
  if (another_follows && j + 1 < g->m && tdp_fuses_scale(g)) {
    g->scale_pending = true;
    g->pend_part = pin;
    g->pend_np = npin;
  } else {
    ScopedKernelTimer t(g->ctx, KRY_K_NORMALIZE);
    launch_scale_from(st, g->w, chain_vec(g, j + 1), pin, npin, n, g->H + (int64_t)(j + 1) * g->m + j, chain_pitch(g),
                      g->cil ? g->v0 : nullptr);
  }
  g->nvec++;
  return launch_ok("add_arnoldi");
}

static bool fused_usable(const kry_gmres *g, const kry_op *op);
static bool strips_usable(const kry_gmres *g, const kry_op *op);
static int fused_steps_async(kry_gmres *g, kry_op *op);

// Restarted GMRES(m) with Givens rotations and a residual monitor on the
// device.  One host round trip per cycle (to decide about a restart), none
// inside a cycle: after convergence the remaining kernels of the cycle return
// immediately.  Returns with the result in g->vars and the stream idle.
static int solve_extended(kry_gmres *g, kry_op *op, const double *b_dev, double *x_dev) {
  hipStream_t st = g->ctx->stream;
  const int m = g->m;
  const int64_t n = g->n;
  double *R = g->R, *cs = R + (int64_t)(m + 1) * m, *sn = cs + m, *gv = sn + m;
  if (g->flags & KRY_GMRES_TDP) KCHK(tdp_setup_async(g, op));
  g->strips = strips_usable(g, op);
  g->cil = chain_interleaved(g, op);
  HIPCHK(hipMemsetAsync(g->vars, 0, sizeof(double) * (size_t)n, st));
  g->last_iters = 0;
  g->last_cycles = 0;
  g->last_converged = 0;
  g->last_relres = 0.0;
  for (int cycle = 0; cycle < g->max_cycles; cycle++) {
    if (cycle == 0) {
      KCHK(initial_subspace_async(g, b_dev));
    } else {
      // v_0 = r / ||r||,  r = b - A x
      if (g->halo_row > 0) KCHK(comm_halo(g->ctx, g->vars, g->halo_row, n));
      int np = 0;
      {
        ScopedKernelTimer t(g->ctx, KRY_K_SPMV);
        op_apply_async(op, g->vars, g->w, 0, nullptr, 0, nullptr, nullptr, g->halo_row > 0);
      }
      ScopedKernelTimer t(g->ctx, KRY_K_NORMALIZE);
      launch_sub_nrm2(st, b_dev, g->w, g->w, n, g->parts, &np);
      const double *pin = g->parts;
      KCHK(finish_parts(g, &pin, 1, &np));
      store_v0(g, g->w, pin, np, g->beta);
      g->nvec = 1;
      g->scaled = false;
      g->last_pending = false;
      if (g->fsigma) launch_fill(st, g->fsigma, 1.0, kFusedCoef);
    }
    {
      ScopedKernelTimer t(g->ctx, KRY_K_OTHER);
      launch_ext_begin(st, g->beta, g->ext, gv, m, cycle == 0, g->rtol);
    }
    HIPCHK(hipMemsetAsync(g->H, 0, sizeof(double) * (size_t)(m + 1) * m, st));
    const int chain_steps = fused_usable(g, op) ? fused_first(g) : m;
    for (int j = 0; j < chain_steps; j++) {
      KCHK(arnoldi_async(g, op));
      ScopedKernelTimer t(g->ctx, KRY_K_OTHER);
      launch_givens(st, g->H, m, j, R, cs, sn, gv, g->ext);
    }
    if (chain_steps < m) KCHK(fused_steps_async(g, op));
    {
      ScopedKernelTimer t(g->ctx, KRY_K_OTHER);
      launch_ext_solve(st, R, gv, m, g->ext, g->y);
    }
    if (g->flags & KRY_GMRES_TDP) {
      // x += M^-1 (V y)
      {
        ScopedKernelTimer t(g->ctx, KRY_K_COMBINE);
        launch_combine(st, chain_vec(g, 0), chain_ld(g), g->y, m, g->w, n, &g->ext->iters, false, nullptr, chain_pitch(g));
      }
      {
        ScopedKernelTimer t(g->ctx, KRY_K_PRECOND);
        launch_tdp_apply(st, g->lo, g->dd, g->ud, g->w, g->w, g->nlines, g->L);
      }
      ScopedKernelTimer t(g->ctx, KRY_K_COMBINE);
      launch_add(st, g->vars, g->w, n);
    } else {
      ScopedKernelTimer t(g->ctx, KRY_K_COMBINE);
      if (g->strips)
        launch_combine_strips(st, strip_vec(g, 0), g->y, m, g->vars, g->grid_rows, g->grid_cols, m + 1, &g->ext->iters, true,
                              g->scaled ? g->fsigma : nullptr);
      else
        launch_combine(st, chain_vec(g, 0), chain_ld(g), g->y, m, g->vars, n, &g->ext->iters, true,
                       g->scaled ? g->fsigma : nullptr, chain_pitch(g));
    }
    KCHK(launch_ok("extended solve"));
    HIPCHK(hipMemcpyAsync(g->ext_h, g->ext, sizeof(ExtState), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    g->last_cycles = cycle + 1;
    g->last_iters += g->ext_h->iters;
    g->last_converged = g->ext_h->converged;
    g->last_relres = g->ext_h->beta0 > 0.0 ? g->ext_h->res / g->ext_h->beta0 : g->ext_h->res;
    if (g->ext_h->stop) break;  // tolerance met, or the iteration broke down (NaN)
  }
  if (x_dev) HIPCHK(hipMemcpyAsync(x_dev, g->vars, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
  return KRY_OK;
}

// The one-sweep steps need a true 2-D stencil and a grid that fits the partial-sum buffers.
static bool fused_usable(const kry_gmres *g, const kry_op *op) {
  return g->fused_from >= 0 && op->kind == 0 && op->wrap_free && fused_parts(op->rows, op->cols) <= kMaxParts &&
         (g->ctx->nranks == 1 || g->halo_row == op->cols);
}

// Whether a solve with this operator keeps its basis strip-major: every step of the cycle is a one-sweep
// step (hand-over at step 0) of the grid the solver was created for, with generated bands whose
// velocity snapshots exist strip-major.
static bool strips_usable(const kry_gmres *g, const kry_op *op) {
  return g->strips_capable && fused_usable(g, op) && fused_first(g) == 0 && g->ctx->nranks == 1 && op->has_gen &&
         op->use_gen && op->has_gen_strips && op->rows == g->grid_rows && op->cols == g->grid_cols;
}

// Whether a solve with this operator takes the norm of the cycle's last vector from the last sweep's sums.
static bool norm_from_sums(const kry_gmres *g, const kry_op *op) {
  return g->norm_from_sums && g->fskip != nullptr && op == g->bound_op && g->m >= 2 && fused_sweep_has_tt(g->m - 2);
}

// The closing kernel of a cycle: the update of step m - 1 on the whole vector (slot m) and its squared norm.
static void launch_close_whole(kry_gmres *g, kry_op *op, int *np, const int *skip) {
  const Bands b{op->band[0], op->band[1], op->band[2], op->band[3], op->band[4]};
  const StripMap smap = g->strips ? strip_map(op->rows, op->cols, g->m + 1) : StripMap();
  launch_arnoldi_close(g->ctx->stream, g->m - 1, b, g->strips ? strip_vec(g, 0) : basis(g, 0), g->strips ? (int64_t)kBlock : g->ld,
                       g->fcoef, g->n, op->cols, 0, g->n, g->ctx->nranks > 1, g->fparts, np, g->strips ? &smap : nullptr, skip);
}

// First step of a cycle whose sweep reads the stored bands although it could evaluate them (experimentation
// knob, env KRY_STORED_FROM; default: never).
static int stored_bands_from() {
  static std::atomic<int> v{-1};
  int r = v.load(std::memory_order_relaxed);
  if (r < 0) {
    r = getenv("KRY_STORED_FROM") ? atoi(getenv("KRY_STORED_FROM")) : 1000;
    v.store(r, std::memory_order_relaxed);
  }
  return r;
}

// Totals of nq quantities from the partial sums of a one-sweep kernel, summed over the ranks.
static int fused_totals(kry_gmres *g, int nq, int np, const int *skip = nullptr) {
  launch_reduce_parts_wide(g->ctx->stream, g->fparts, nq, np, g->ftot, skip);
  if (g->ctx->nranks > 1) KCHK(comm_allreduce(g->ctx, g->ftot, nq, KRY_REDUCE_SUM));
  return KRY_OK;
}

// Steps fused_from .. m-1 of a cycle, one sweep over the basis each (kry_fused.hip).
// On entry the basis holds v_0..v_from; on return v_0..v_m and H are complete.
// In the restarted solver the Givens update of column i follows as soon as H(i+1,i) is
// known, which is one sweep later than with the chain: when the residual monitor raises
// the stop flag, one sweep beyond the last column used has already run (its results
// are simply not used), and everything after it returns at once.
static int fused_steps_async(kry_gmres *g, kry_op *op) {
  const bool ext = (g->flags & KRY_GMRES_EXTENDED) && g->ext;
  LaunchPredicate pred(ext ? &g->ext->stop : nullptr);
  double *R = g->R, *rot_c = ext ? R + (int64_t)(g->m + 1) * g->m : nullptr, *rot_s = ext ? rot_c + g->m : nullptr,
         *gv = ext ? rot_s + g->m : nullptr;
  hipStream_t st = g->ctx->stream;
  const int m = g->m, i0 = fused_first(g);
  const int64_t n = g->n;
  const int64_t P = op->cols;
  const bool ranks = g->ctx->nranks > 1;
  const Bands b{op->band[0], op->band[1], op->band[2], op->band[3], op->band[4]};
  int np = 0;
  if (g->halo_row > 0) KCHK(comm_halo(g->ctx, basis(g, i0), g->halo_row, n));
  {
    // the dots of A v_from against v_0..v_from and the Gram column of v_from (the
    // product itself is recomputed by the first sweep)
    ScopedKernelTimer t(g->ctx, KRY_K_SPMV);
    if (g->strips)  // i0 = 0: v_0 in grid order
      op_apply_async(op, g->v0, g->w, 1, g->v0, g->ld, g->fparts, &np, g->halo_row > 0);
    else
      op_apply_async(op, basis(g, i0), g->w, i0 + 1, basis(g, 0), g->ld, g->fparts, &np, g->halo_row > 0);
  }
  KCHK(fused_totals(g, 2 * (i0 + 1) - 1, np));
  // the last step's norm from the last sweep's sums; the closing kernel then only runs where that cancels
  const int *skip = norm_from_sums(g, op) ? g->fskip : nullptr;
  // On one rank, outside the monitored mode, a sweep evaluates its own coefficients in its prologue (no separate
  // launch between two sweeps); the hand-over step and the last step keep the launch of their own.
  static const bool fold_env = !(getenv("KRY_FOLD_COEFFS") && atoi(getenv("KRY_FOLD_COEFFS")) == 0);
  const bool fold = fold_env && !ranks && !ext;
  const CoefIn cin{g->ftot, g->G, m + 1, g->H, m, g->fsigma};
  for (int i = i0; i < m; i++) {
    const bool boot = i == i0;
    const bool folded = fold && !boot && i < m - 1;
    if (!folded)
      launch_fused_coeffs(st, g->ftot, i, boot, g->G, m + 1, g->H, m, g->fcoef, g->fsigma,
                          (skip && i == m - 1) ? g->fskip : nullptr);
    if (ext && !boot) {
      ScopedKernelTimer t(g->ctx, KRY_K_OTHER);
      launch_givens(st, g->H, m, i - 1, R, rot_c, rot_s, gv, g->ext);
    }
    if (i < m - 1) {
      if (ranks) {
        // the new vector on this rank's first and last row, for the neighbours
        ScopedKernelTimer t(g->ctx, KRY_K_OTHER);
        for (int64_t k0 : {(int64_t)0, n - P}) {
          launch_arnoldi_close(st, i, b, basis(g, 0), g->ld, g->fcoef, n, op->cols, k0, P, true, g->fparts, &np);
          if (n == P) break;
        }
      }
      const BandGen *gen = g->strips ? &op->gen_strips : (op->has_gen && op->use_gen ? &op->gen : nullptr);
      if (i >= stored_bands_from()) gen = nullptr;
      if (ranks && g->overlap_halo && fused_bands_splittable(i, op->rows, op->cols)) {
        // The neighbours' rows of the new vector travel on the second stream while the bands of
        // rows that do not touch them are swept; the first and the last band follow.
        kry_ctx *c = g->ctx;
        KCHK(comm_stream_ready(c));
        HIPCHK(hipEventRecord(c->ev_fork, st));
        HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_fork, 0));
        KCHK(comm_halo(c, basis(g, i + 1), P, n, c->comm_stream));
        HIPCHK(hipEventRecord(c->ev_join, c->comm_stream));
        ScopedKernelTimer t(c, KRY_K_FUSED);
        launch_arnoldi_fused(st, i, op->rows, op->cols, b, basis(g, 0), g->ld, g->fcoef, g->fparts, &np, true, gen,
                             kSweepInterior);
        HIPCHK(hipStreamWaitEvent(st, c->ev_join, 0));
        launch_arnoldi_fused(st, i, op->rows, op->cols, b, basis(g, 0), g->ld, g->fcoef, g->fparts, &np, true, gen,
                             kSweepEdges);
      } else {
        if (ranks) KCHK(comm_halo(g->ctx, basis(g, i + 1), P, n));
        ScopedKernelTimer t(g->ctx, KRY_K_FUSED);
        launch_arnoldi_fused(st, i, op->rows, op->cols, b, g->strips ? strip_vec(g, 0) : basis(g, 0),
                             g->strips ? (int64_t)kBlock : g->ld, g->fcoef, g->fparts, &np, ranks, gen, kSweepAll,
                             g->strips ? m + 1 : 0, folded ? &cin : nullptr);
      }
      KCHK(fused_totals(g, 2 * i + 4 + (fused_sweep_has_tt(i) ? 1 : 0), np));
    } else {
      ScopedKernelTimer t(g->ctx, KRY_K_MGS_LAST);
      launch_close_whole(g, op, &np, skip);
    }
  }
  KCHK(fused_totals(g, 1, np, skip));
  launch_fused_finish(st, g->ftot, m, g->H, g->fsigma, skip);
  g->last_pending = skip != nullptr;
  if (ext) {
    ScopedKernelTimer t(g->ctx, KRY_K_OTHER);
    launch_givens(st, g->H, m, m - 1, R, rot_c, rot_s, gv, g->ext);
  }
  g->scaled = true;
  g->nvec = m + 1;
  return launch_ok("one-sweep Arnoldi steps");
}

static int solve_async(kry_gmres *g, kry_op *op, const double *b_dev, double *x_dev) {
  if (g->flags & KRY_GMRES_EXTENDED) return solve_extended(g, op, b_dev, x_dev);
  hipStream_t st = g->ctx->stream;
  if (g->flags & KRY_GMRES_TDP) KCHK(tdp_setup_async(g, op));
  g->strips = strips_usable(g, op);
  g->cil = chain_interleaved(g, op);
  KCHK(initial_subspace_async(g, b_dev));
  HIPCHK(hipMemsetAsync(g->H, 0, sizeof(double) * (size_t)(g->m + 1) * g->m, st));
  if (fused_usable(g, op)) {
    for (int i = 0; i < fused_first(g); i++) KCHK(arnoldi_async(g, op));
    KCHK(fused_steps_async(g, op));
  } else {
    for (int i = 0; i < g->m; i++) KCHK(arnoldi_async(g, op, i + 1 < g->m));
  }
  {
    ScopedKernelTimer t(g->ctx, KRY_K_OTHER);
    launch_small_solve(st, g->H, g->beta, g->m, g->y, g->work);
  }
  {
    ScopedKernelTimer t(g->ctx, KRY_K_COMBINE);
    if (g->strips)
      launch_combine_strips(st, strip_vec(g, 0), g->y, g->m, g->vars, g->grid_rows, g->grid_cols, g->m + 1, nullptr, false,
                            g->scaled ? g->fsigma : nullptr);
    else
      launch_combine(st, chain_vec(g, 0), chain_ld(g), g->y, g->m, g->vars, g->n, nullptr, false,
                     g->scaled ? g->fsigma : nullptr, chain_pitch(g));
  }
  if ((g->flags & KRY_GMRES_TDP) && (g->flags & KRY_GMRES_FIXED_PRECOND)) {
    ScopedKernelTimer t(g->ctx, KRY_K_PRECOND);
    launch_tdp_apply(st, g->lo, g->dd, g->ud, g->vars, g->vars, g->nlines, g->L);
  }
  KCHK(launch_ok("solve"));
  if (x_dev) HIPCHK(hipMemcpyAsync(x_dev, g->vars, sizeof(double) * g->n, hipMemcpyDeviceToDevice, st));
  return KRY_OK;
}

static int stage_rhs(kry_gmres *g, const double *b_host) {
  if (!g->bbuf) KCHK(dev_alloc(&g->bbuf, g->ld));
  HIPCHK(hipMemcpyAsync(g->bbuf, b_host, sizeof(double) * g->n, hipMemcpyHostToDevice, g->ctx->stream));
  return KRY_OK;
}

extern "C" {

int kry_gmres_initial_subspace(kry_gmres *g, const double *b_host, double *beta) {
  if (!g || !b_host) return fail(KRY_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(g->ctx->device));
  KCHK(stage_rhs(g, b_host));
  g->strips = false;  // the step-wise interface works on the chain's layout, with contiguous vectors
  g->cil = false;
  KCHK(initial_subspace_async(g, g->bbuf));
  double b = 0.0;
  HIPCHK(hipMemcpyAsync(&b, g->beta, sizeof(double), hipMemcpyDeviceToHost, g->ctx->stream));
  HIPCHK(hipStreamSynchronize(g->ctx->stream));
  if (beta) *beta = b;
  return KRY_OK;
}

int kry_gmres_assemble_preconditioner(kry_gmres *g, kry_op *op, double *tdiags_host) {
  KCHK(check_op(g, op));
  if (!(g->flags & KRY_GMRES_TDP)) return fail(KRY_ERR_STATE, "solver was created without KRY_GMRES_TDP");
  HIPCHK(hipSetDevice(g->ctx->device));
  KCHK(tdp_setup_async(g, op));
  HIPCHK(hipStreamSynchronize(g->ctx->stream));
  if (tdiags_host) {
    const int64_t n = g->n;
    std::vector<double> lo((size_t)n), di((size_t)n), up((size_t)n);
    HIPCHK(hipMemcpy(lo.data(), g->lo, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(di.data(), g->di, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(up.data(), g->up, sizeof(double) * n, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; i++) {
      tdiags_host[i * 3 + 0] = lo[(size_t)i];
      tdiags_host[i * 3 + 1] = di[(size_t)i];
      tdiags_host[i * 3 + 2] = up[(size_t)i];
    }
    tdiags_host[0] = kSNaN;               // include/solver.hpp:297
    tdiags_host[(n - 1) * 3 + 2] = kSNaN; // include/solver.hpp:312
  }
  return KRY_OK;
}

int kry_gmres_apply_preconditioner(kry_gmres *g, const double *in_host, double *out_host) {
  if (!g || !in_host || !out_host) return fail(KRY_ERR_INVALID, "NULL argument");
  if (!(g->flags & KRY_GMRES_TDP)) return fail(KRY_ERR_STATE, "solver was created without KRY_GMRES_TDP");
  if (!g->tdp_ready) return fail(KRY_ERR_STATE, "preconditioner has not been assembled");
  HIPCHK(hipSetDevice(g->ctx->device));
  KCHK(stage_rhs(g, in_host));
  launch_tdp_apply(g->ctx->stream, g->lo, g->dd, g->ud, g->bbuf, g->z, g->nlines, g->L);
  KCHK(launch_ok("apply_preconditioner"));
  HIPCHK(hipMemcpyAsync(out_host, g->z, sizeof(double) * g->n, hipMemcpyDeviceToHost, g->ctx->stream));
  HIPCHK(hipStreamSynchronize(g->ctx->stream));
  return KRY_OK;
}

int kry_gmres_add_arnoldi(kry_gmres *g, kry_op *op, double *h_next) {
  KCHK(check_op(g, op));
  HIPCHK(hipSetDevice(g->ctx->device));
  KCHK(arnoldi_async(g, op));
  const int j = g->nvec - 2;
  double h = 0.0;
  HIPCHK(hipMemcpyAsync(&h, g->H + (int64_t)(j + 1) * g->m + j, sizeof(double), hipMemcpyDeviceToHost, g->ctx->stream));
  HIPCHK(hipStreamSynchronize(g->ctx->stream));
  if (h_next) *h_next = h;
  return KRY_OK;
}

int kry_gmres_solve_device(kry_gmres *g, kry_op *op, const double *b_dev, double *x_dev) {
  KCHK(check_op(g, op));
  if (!b_dev) return fail(KRY_ERR_INVALID, "b is NULL");
  HIPCHK(hipSetDevice(g->ctx->device));
  return solve_async(g, op, b_dev, x_dev);
}

int kry_gmres_solve(kry_gmres *g, kry_op *op, const double *b_host, double *x_host) {
  KCHK(check_op(g, op));
  if (!b_host) return fail(KRY_ERR_INVALID, "b is NULL");
  HIPCHK(hipSetDevice(g->ctx->device));
  KCHK(stage_rhs(g, b_host));
  KCHK(solve_async(g, op, g->bbuf, nullptr));
  if (x_host) HIPCHK(hipMemcpyAsync(x_host, g->vars, sizeof(double) * g->n, hipMemcpyDeviceToHost, g->ctx->stream));
  HIPCHK(hipStreamSynchronize(g->ctx->stream));
  return KRY_OK;
}

int kry_gmres_subspace_vector(kry_gmres *g, int i, double *out_host) {
  if (!g || !out_host) return fail(KRY_ERR_INVALID, "NULL argument");
  if (i < 0 || i >= g->nvec) return fail(KRY_ERR_INVALID, "subspace vector %d does not exist (%d vectors)", i, g->nvec);
  HIPCHK(hipSetDevice(g->ctx->device));
  if (i == g->m && g->last_pending) {
    // the last solve took the norm of this vector from sums: form the vector now (its coefficients
    // are still in place, and the operator it takes is the one the solver is bound to)
    int np = 0;
    launch_close_whole(g, g->bound_op, &np, nullptr);
    KCHK(launch_ok("kry_gmres_subspace_vector"));
    g->last_pending = false;
  }
  const double *src = basis(g, i);
  if (g->strips) {
    launch_unscale_strips(g->ctx->stream, strip_vec(g, i), g->scaled ? g->fsigma : nullptr, i, g->w, g->grid_rows, g->grid_cols,
                          g->m + 1);
    KCHK(launch_ok("kry_gmres_subspace_vector"));
    src = g->w;
  } else if (g->cil) {
    launch_unscale(g->ctx->stream, chain_vec(g, i), nullptr, i, g->w, g->n, chain_pitch(g));
    KCHK(launch_ok("kry_gmres_subspace_vector"));
    src = g->w;
  } else if (g->scaled) {
    // the one-sweep steps keep their vectors un-normalised: v_i = slot_i / sigma_i
    launch_unscale(g->ctx->stream, src, g->fsigma, i, g->w, g->n);
    KCHK(launch_ok("kry_gmres_subspace_vector"));
    src = g->w;
  }
  HIPCHK(hipMemcpyAsync(out_host, src, sizeof(double) * g->n, hipMemcpyDeviceToHost, g->ctx->stream));
  HIPCHK(hipStreamSynchronize(g->ctx->stream));
  return KRY_OK;
}

int kry_gmres_hessenberg(kry_gmres *g, int i, int j, double *out) {
  if (!g || !out) return fail(KRY_ERR_INVALID, "NULL argument");
  if (i < 0 || i > g->m || j < 0 || j >= g->m) return fail(KRY_ERR_INVALID, "H(%d,%d) out of range", i, j);
  HIPCHK(hipSetDevice(g->ctx->device));
  HIPCHK(hipMemcpyAsync(out, g->H + (int64_t)i * g->m + j, sizeof(double), hipMemcpyDeviceToHost, g->ctx->stream));
  HIPCHK(hipStreamSynchronize(g->ctx->stream));
  return KRY_OK;
}

int kry_gmres_hessenberg_matrix(kry_gmres *g, double *out_host) {
  if (!g || !out_host) return fail(KRY_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(g->ctx->device));
  HIPCHK(hipMemcpyAsync(out_host, g->H, sizeof(double) * (size_t)(g->m + 1) * g->m, hipMemcpyDeviceToHost, g->ctx->stream));
  HIPCHK(hipStreamSynchronize(g->ctx->stream));
  return KRY_OK;
}

int kry_gmres_reduced_solve(kry_gmres *g, const double *H_host, double beta, double *y_host) {
  if (!g || !H_host || !y_host) return fail(KRY_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(g->ctx->device));
  hipStream_t st = g->ctx->stream;
  const int m = g->m;
  HIPCHK(hipMemcpyAsync(g->H, H_host, sizeof(double) * (size_t)(m + 1) * m, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(g->beta, &beta, sizeof(double), hipMemcpyHostToDevice, st));
  launch_small_solve(st, g->H, g->beta, m, g->y, g->work);
  KCHK(launch_ok("kry_gmres_reduced_solve"));
  HIPCHK(hipMemcpyAsync(y_host, g->y, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return KRY_OK;
}

int kry_gmres_prev_result(kry_gmres *g, double *out_host) {
  if (!g || !out_host) return fail(KRY_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(g->ctx->device));
  HIPCHK(hipMemcpyAsync(out_host, g->vars, sizeof(double) * g->n, hipMemcpyDeviceToHost, g->ctx->stream));
  HIPCHK(hipStreamSynchronize(g->ctx->stream));
  return KRY_OK;
}

// Algorithmic bytes (each operand read once, each result written once per
// kernel) of one fixed-m cycle, per kernel class.
int kry_gmres_cycle_bytes(const kry_gmres *g, const kry_op *op, double bytes[KRY_K_COUNT],
                          int launches[KRY_K_COUNT]) {
  if (!g || !op || !bytes || !launches) return fail(KRY_ERR_INVALID, "NULL argument");
  for (int k = 0; k < KRY_K_COUNT; k++) {
    bytes[k] = 0.0;
    launches[k] = 0;
  }
  const double N = 8.0 * (double)g->n;  // bytes of one vector
  const int m = g->m;
  const int K = g->kblock;
  // bytes of the operator itself: five bands, or CSR values + indices + row pointers
  // (grid operators with a recipe: the chain's operator kernel evaluates the entries from two velocity arrays
  // when the rows have even length)
  const int gen_min_kc = getenv("KRY_SPMV_GEN") ? atoi(getenv("KRY_SPMV_GEN")) : 4;
  const bool spmv_gen = op->kind == 0 && op->has_gen && op->use_gen && op->cols % 2 == 0 && op->n % 2 == 0 && gen_min_kc > 0;
  const double opb = op->kind == 0 ? 5.0 * N : 12.0 * (double)op->nnz + 8.0 * (double)(op->n + 1);
  // bytes of the operator as the chain's apply with kp partner vectors reads it
  auto op_bytes = [&](int kp) { return (spmv_gen && kp >= gen_min_kc) ? 2.0 * N : opb; };
  const bool fused = fused_usable(g, op);
  const int chain_steps = fused ? fused_first(g) : m;
  for (int j = 0; j < chain_steps; j++) {
    const int nv = j + 1;
    int kp = nv < K ? nv : K;
    bytes[KRY_K_SPMV] += op_bytes(kp) + N + N + kp * N;  // operator, x read, y written, first group read
    launches[KRY_K_SPMV] += 1;
    for (int cs = kp; cs < nv; cs += K) {
      const int kc = nv - cs < K ? nv - cs : K;
      const int cls = (kp == K && kc == K) ? KRY_K_MGS : KRY_K_MGS_PART;
      bytes[cls] += (2.0 + kp + kc) * N;  // w read + written, kp applied, kc dotted
      launches[cls] += 1;
      kp = kc;
    }
    bytes[KRY_K_MGS_LAST] += (2.0 + kp) * N;
    launches[KRY_K_MGS_LAST] += 1;
  }
  if (fused) {
    const int i0 = fused_first(g);
    bytes[KRY_K_SPMV] += op_bytes(i0 + 1) + N + N + (i0 + 1) * N;  // t = A v_i0 with its dots
    launches[KRY_K_SPMV] += 1;
    // the operator as a sweep reads it: five bands, or the two velocity arrays its interior
    // entries are evaluated from (no stored band is read then)
    const double opf = op->has_gen && op->use_gen ? 2.0 * N : opb;
    for (int i = i0; i < m - 1; i++) {
      // read: the i + 1 stored vectors, operator; written: the new vector
      bytes[KRY_K_FUSED] += (1.0 + i) * N + opf + N;
      launches[KRY_K_FUSED] += 1;
    }
    // closing sweep: the m stored vectors and the operator read, the new vector written -- unless the norm it
    // exists for comes from the last sweep's sums (the launch then returns at once)
    if (!norm_from_sums(g, op)) bytes[KRY_K_MGS_LAST] += m * N + opb + N;
    launches[KRY_K_MGS_LAST] += 1;
  }
  // one normalisation per chain step (none after the one-sweep steps) + ||b|| and v_0 = b/||b||
  // (with the line preconditioner, all but the last step's is done by the next step's preconditioner kernel)
  const int fused_scales = (!fused && tdp_fuses_scale(g)) ? m - 1 : 0;
  bytes[KRY_K_NORMALIZE] = 2.0 * N * (chain_steps - fused_scales) + 3.0 * N;
  launches[KRY_K_NORMALIZE] = chain_steps - fused_scales + 2;
  bytes[KRY_K_COMBINE] = (m + 1) * N;
  launches[KRY_K_COMBINE] = 1;
  if (g->flags & KRY_GMRES_TDP) {
    // factor: lo,di,up read, dd,ud written; apply: forward sweep reads sol,lo,dd and writes out,
    // backward sweep reads out,ud and writes out
    const int applies = m + ((g->flags & KRY_GMRES_FIXED_PRECOND) ? 1 : 0);
    bytes[KRY_K_PRECOND] = 5.0 * N + applies * 7.0 * N + fused_scales * N;  // + the normalised vector written
    launches[KRY_K_PRECOND] = applies + 1;
  }
  return KRY_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Implicit-Euler step
// ---------------------------------------------------------------------------
static int64_t padded_to_cell(const kry_ie *ie, int ip, int jp) {
  const int cx = ie->g.cx, cy = ie->g.cy, P = cy + 2;
  const bool xedge = ip == 0 || ip == cx + 1, yedge = jp == 0 || jp == cy + 1;
  if (xedge && yedge) return -1;
  if (ip == 0) return jp - 1;
  if (ip == cx + 1) return (int64_t)cy + (int64_t)cx * P + (jp - 1);
  return (int64_t)cy + (int64_t)(ip - 1) * P + jp;
}

extern "C" {

int kry_ie_create_slab(kry_ctx *ctx, double c1x, double c1y, double c2x, double c2y, int cx, int cy,
                       int row_begin, int row_end, const double *T0, const double *u0,
                       const double *v0, double diffusion, double reynolds, double prandtl,
                       double eckert, const int *bc_is_dirichlet, int m, int flags, kry_ie **out) {
  if (!ctx || !out || !bc_is_dirichlet) return fail(KRY_ERR_INVALID, "NULL argument");
  *out = nullptr;
  if (cx < 1 || cy < 1) return fail(KRY_ERR_INVALID, "cells_x and cells_y must be positive");
  if ((int64_t)(cx + 2) * (cy + 2) > 2147483000LL) return fail(KRY_ERR_INVALID, "mesh too large for 32-bit cell indices");
  if (row_begin < 0 || row_end > cx + 2 || row_end <= row_begin)
    return fail(KRY_ERR_INVALID, "padded row range [%d, %d) is not inside [0, %d)", row_begin, row_end, cx + 2);
  if (ctx->nranks == 1 && (row_begin != 0 || row_end != cx + 2))
    return fail(KRY_ERR_INVALID, "a partial row range needs a context with collectives (kry_ctx_set_comm)");
  // interior rows (cell index i) owned by this slab
  const int ib = (row_begin > 1 ? row_begin : 1) - 1, ie_ = (row_end < cx + 1 ? row_end : cx + 1) - 1;
  if (ie_ > ib && (!T0 || !u0 || !v0)) return fail(KRY_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(ctx->device));
  kry_ie *ie = new kry_ie;
  ie->ctx = ctx;
  ctx_ref(ctx);
  // Mesh::Mesh, src/mesh.cpp:11-15
  const double min_x = c1x < c2x ? c1x : c2x, max_x = c1x < c2x ? c2x : c1x;
  const double min_y = c1y < c2y ? c1y : c2y, max_y = c1y < c2y ? c2y : c1y;
  if (!(min_x != max_x) || !(min_y != max_y)) {
    delete ie;
    ctx_unref(ctx);
    return fail(KRY_ERR_INVALID, "degenerate domain");
  }
  ie->min_x = min_x;
  ie->min_y = min_y;
  ie->g.cx = cx;
  ie->g.cy = cy;
  ie->g.row0 = row_begin;
  ie->g.rows = row_end - row_begin;
  ie->g.dx = (max_x - min_x) / (double)cx;
  ie->g.dy = (max_y - min_y) / (double)cy;
  ie->g.diffusion = diffusion;
  ie->g.reynolds = reynolds;
  ie->g.prandtl = prandtl;
  ie->g.eckert = eckert;
  ie->g.inv_rp = 1.0 / (reynolds * prandtl);  // include/space_disc.hpp:61
  for (int k = 0; k < 12; k++) ie->bc.weight[k] = bc_is_dirichlet[k] ? 1.0 : 0.0;  // boundary_conds.hpp:50
  const int P = cy + 2;
  ie->nloc = (int64_t)ie->g.rows * P;
  ie->neq = (int64_t)(cx + 2) * P - 4;
  ie->halo = ((int64_t)P + 31) / 32 * 32;
  const int64_t fld_len = ie->halo + (ie->nloc + 31) / 32 * 32 + ie->halo;
  int st = KRY_OK;
  std::vector<double> host((size_t)fld_len);
  const double *src[3] = {T0, u0, v0};
  for (int f = 0; f < 3 && st == KRY_OK; f++) {
    // ghosts, corners (and the halo rows until the first exchange) start as
    // signalling NaN, src/mesh.cpp:34-51
    for (auto &v : host) v = kSNaN;
    for (int i = ib; i < ie_; i++)
      for (int j = 0; j < cy; j++)
        host[(size_t)ie->halo + (size_t)(i + 1 - row_begin) * P + (j + 1)] = src[f][(size_t)(i - ib) * cy + j];
    double *base = nullptr;
    st = dev_alloc(&base, fld_len);
    if (st == KRY_OK && hipMemcpy(base, host.data(), sizeof(double) * fld_len, hipMemcpyHostToDevice) != hipSuccess)
      st = fail(KRY_ERR_HIP, "field upload failed");
    ie->fld[f] = base ? base + ie->halo : nullptr;
  }
  for (int b = 0; b < 5 && st == KRY_OK; b++) {
    st = dev_alloc(&ie->band[b], ie->nloc);
    if (st == KRY_OK && hipMemset(ie->band[b], 0, sizeof(double) * ie->nloc) != hipSuccess) st = fail(KRY_ERR_HIP, "memset failed");
  }
  if (st == KRY_OK) st = dev_alloc(&ie->rhs, ie->nloc);
  if (st == KRY_OK && hipMemset(ie->rhs, 0, sizeof(double) * ie->nloc) != hipSuccess) st = fail(KRY_ERR_HIP, "memset failed");
  const int nbc = 3 * 2 * (cx + cy);
  if (st == KRY_OK) st = dev_alloc(&ie->bcvals, nbc);
  if (st == KRY_OK && hipHostMalloc((void **)&ie->bcvals_h, sizeof(double) * nbc) != hipSuccess) st = fail(KRY_ERR_HIP, "pinned alloc failed");
  if (st == KRY_OK) memset(ie->bcvals_h, 0, sizeof(double) * nbc);
  if (st == KRY_OK) st = dev_alloc(&ie->wrap, 3 * cy);
  if (st == KRY_OK) st = dev_alloc(&ie->parts, 2 * kMaxParts);
  if (st == KRY_OK) st = dev_alloc(&ie->maxd, 4);
  if (st == KRY_OK && hipHostMalloc((void **)&ie->maxd_h, sizeof(double)) != hipSuccess) st = fail(KRY_ERR_HIP, "pinned alloc failed");
  if (st == KRY_OK) {
    kry_op *op = new kry_op;
    op->ctx = ctx;
    ctx_ref(ctx);
    op->kind = 0;
    op->rows = ie->g.rows;
    op->cols = P;
    op->n = ie->nloc;
    for (int b = 0; b < 5; b++) op->band[b] = ie->band[b];
    op->own_bands = false;
    op->wrap_free = true;  // k_offdiag / k_assemble leave the ghost columns' outward couplings at zero
    op->slab = !ie->whole();
    ie->op = op;
    if (m > 0) st = gmres_create(ctx, ie->nloc, m, flags, P, &ie->solver, ie->g.rows, P);
    if (st == KRY_OK && ie->solver) ie->solver->bound_op = op;
  }
  if (st != KRY_OK) {
    kry_ie_destroy(ie);
    return st;
  }
  *out = ie;
  return KRY_OK;
}

int kry_ie_create(kry_ctx *ctx, double c1x, double c1y, double c2x, double c2y, int cx, int cy,
                  const double *T0, const double *u0, const double *v0, double diffusion,
                  double reynolds, double prandtl, double eckert, const int *bc_is_dirichlet, int m,
                  int flags, kry_ie **out) {
  return kry_ie_create_slab(ctx, c1x, c1y, c2x, c2y, cx, cy, 0, cx + 2, T0, u0, v0, diffusion, reynolds,
                            prandtl, eckert, bc_is_dirichlet, m, flags, out);
}

void kry_ie_destroy(kry_ie *ie) {
  if (!ie) return;
  (void)hipSetDevice(ie->ctx->device);
  (void)hipStreamSynchronize(ie->ctx->stream);
  if (ie->graph_exec) (void)hipGraphExecDestroy(ie->graph_exec);
  kry_gmres_destroy(ie->solver);
  kry_op_destroy(ie->op);
  for (int f = 0; f < 3; f++)
    if (ie->fld[f]) (void)hipFree(ie->fld[f] - ie->halo);
  for (int b = 0; b < 5; b++) (void)hipFree(ie->band[b]);
  for (int f = 0; f < 2; f++)
    if (ie->snap[f]) (void)hipFree(ie->snap[f] - ie->halo);
  for (int f = 0; f < 2; f++) (void)hipFree(ie->snap_strips[f]);
  (void)hipFree(ie->rhs);
  (void)hipFree(ie->bcvals);
  (void)hipFree(ie->wrap);
  (void)hipFree(ie->parts);
  (void)hipFree(ie->maxd);
  if (ie->bcvals_h) (void)hipHostFree(ie->bcvals_h);
  if (ie->maxd_h) (void)hipHostFree(ie->maxd_h);
  kry_ctx *ctx = ie->ctx;
  delete ie;
  ctx_unref(ctx);
}

}  // extern "C"

static int ie_wait(kry_ie *ie) {
  if (ie->pending) {
    HIPCHK(hipStreamSynchronize(ie->ctx->stream));
    ie->pending = false;
  }
  return KRY_OK;
}

// TimeDisc::apply_bcs on a slab: hand the right boundary column to every rank,
// fill the owned ghost cells, then refresh the halo rows of the three fields.
static int ie_apply_bcs_async(kry_ie *ie, bool force_upload) {
  hipStream_t st = ie->ctx->stream;
  const int nbc = 3 * 2 * (ie->g.cx + ie->g.cy);
  if (ie->bc_dirty || force_upload) {
    HIPCHK(hipMemcpyAsync(ie->bcvals, ie->bcvals_h, sizeof(double) * nbc, hipMemcpyHostToDevice, st));
    ie->bc_dirty = false;
  }
  launch_pack_wrap(st, ie->g, ie->fld[0], ie->fld[1], ie->fld[2], ie->wrap);
  KCHK(comm_allreduce(ie->ctx, ie->wrap, 3 * (int64_t)ie->g.cy, KRY_REDUCE_SUM));
  launch_bc_fill(st, ie->g, ie->bc, ie->fld[0], ie->fld[1], ie->fld[2], ie->bcvals, ie->wrap);
  KCHK(comm_halo_multi(ie->ctx, ie->fld, 3, ie->g.cy + 2, ie->nloc));
  return KRY_OK;
}

static int ie_assemble_async(kry_ie *ie, double dt, bool force_upload) {
  KCHK(ie_apply_bcs_async(ie, force_upload));
  launch_assemble(ie->ctx->stream, ie->g, ie->bc, dt, ie->fld[0], ie->fld[1], ie->fld[2], ie->band[0],
                  ie->band[1], ie->band[2], ie->band[3], ie->band[4], ie->rhs);
  return launch_ok("assemble_system");
}

// Second half of ImplicitEuler::timestep (include/time_disc.hpp:281-286): T += delta on
// interior cells and the returned maximum (std::max fold in cell order, see k_update_max),
// for the delta in the solver's result vector; queues the copy of the maximum to the host.
static int ie_update_async(kry_ie *ie) {
  hipStream_t st = ie->ctx->stream;
  {
    ScopedKernelTimer t(ie->ctx, KRY_K_ASSEMBLY);
    int np = 0;
    launch_update_max(st, ie->g, ie->fld[0], ie->solver->vars, ie->parts, &np);
    if (ie->ctx->nranks == 1) {
      launch_max_fold(st, ie->g, ie->solver->vars, ie->parts, np, ie->maxd, 7);
    } else {
      // cell order runs through the ranks' row blocks: every rank needs the last NaN
      // position of the whole mesh before it can take its share of the maximum
      launch_max_fold(st, ie->g, ie->solver->vars, ie->parts, np, ie->maxd, 1);
      KCHK(comm_allreduce(ie->ctx, ie->maxd + 1, 2, KRY_REDUCE_MAX));
      launch_max_fold(st, ie->g, ie->solver->vars, ie->parts, np, ie->maxd, 2);
      KCHK(comm_allreduce(ie->ctx, ie->maxd, 1, KRY_REDUCE_MAX));
      launch_max_fold(st, ie->g, ie->solver->vars, ie->parts, np, ie->maxd, 4);
    }
  }
  KCHK(launch_ok("timestep"));
  HIPCHK(hipMemcpyAsync(ie->maxd_h, ie->maxd, sizeof(double), hipMemcpyDeviceToHost, st));
  return KRY_OK;
}

// everything of ImplicitEuler::timestep that runs on the device
static int ie_step_body(kry_ie *ie, double dt, bool force_upload) {
  {
    ScopedKernelTimer t(ie->ctx, KRY_K_ASSEMBLY);
    KCHK(ie_assemble_async(ie, dt, force_upload));
  }
  KCHK(solve_async(ie->solver, ie->op, ie->rhs, nullptr));
  return ie_update_async(ie);
}

static int ie_step_async(kry_ie *ie, double dt) {
  if (!ie->solver) return fail(KRY_ERR_STATE, "step object was created without a solver");
  hipStream_t st = ie->ctx->stream;
  // the restarted solver reads its state on the host between cycles: no capture
  if (!ie->use_graph || (ie->ctx->nranks > 1 && !ie->ctx->nccl) || (ie->solver->flags & KRY_GMRES_EXTENDED)) {
    KCHK(ie_step_body(ie, dt, false));
  } else {
    const uint64_t epoch = g_config_epoch.load(std::memory_order_relaxed);
    if (!ie->graph_exec || ie->graph_dt != dt || ie->graph_epoch != epoch) {
      if (ie->graph_exec) {
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipGraphExecDestroy(ie->graph_exec));
        ie->graph_exec = nullptr;
      }
      hipGraph_t graph = nullptr;
      const bool was_timing = ie->ctx->timing;
      ie->ctx->timing = false;  // no event records inside a capture
      HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      int rc = ie_step_body(ie, dt, true);
      hipError_t e = hipStreamEndCapture(st, &graph);
      ie->ctx->timing = was_timing;
      if (rc != KRY_OK) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
      }
      if (e != hipSuccess) return fail(KRY_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
      e = hipGraphInstantiate(&ie->graph_exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (e != hipSuccess) return fail(KRY_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
      ie->graph_dt = dt;
      ie->graph_epoch = epoch;
      ie->graph_basis = basis_state(ie->solver);
    }
    ie->bc_dirty = false;  // the graph always re-uploads the staged boundary values
    HIPCHK(hipGraphLaunch(ie->graph_exec, st));
    // A replay refills the basis as the captured solve did, whatever used the solver since (the step-wise
    // interface, a solve with another operator): the host's notes about the basis follow the replay.
    set_basis_state(ie->solver, ie->graph_basis);
  }
  ie->pending = true;
  ie->time += dt;  // include/time_disc.hpp:287 (after the solve: BCs were taken at the old time)
  return KRY_OK;
}

extern "C" {

int kry_ie_set_bc_values(kry_ie *ie, int field, int side, const double *vals, int len) {
  if (!ie || !vals) return fail(KRY_ERR_INVALID, "NULL argument");
  if (field < 0 || field > 2 || side < 0 || side > 3) return fail(KRY_ERR_INVALID, "bad field/side");
  const int cx = ie->g.cx, cy = ie->g.cy;
  const int want = (side == KRY_SIDE_BOTTOM || side == KRY_SIDE_TOP) ? cx : cy;
  if (len != want) return fail(KRY_ERR_INVALID, "side %d of a %dx%d mesh takes %d values, got %d", side, cx, cy, want, len);
  HIPCHK(hipSetDevice(ie->ctx->device));
  KCHK(ie_wait(ie));  // the staging buffer may still be the source of a queued copy
  const int per_field = 2 * (cx + cy);
  const int off = side == KRY_SIDE_BOTTOM ? 0 : side == KRY_SIDE_TOP ? cx : side == KRY_SIDE_LEFT ? 2 * cx : 2 * cx + cy;
  memcpy(ie->bcvals_h + (size_t)field * per_field + off, vals, sizeof(double) * (size_t)len);
  ie->bc_dirty = true;
  return KRY_OK;
}

int kry_ie_apply_bcs(kry_ie *ie) {
  if (!ie) return fail(KRY_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(ie->ctx->device));
  KCHK(ie_apply_bcs_async(ie, false));
  KCHK(launch_ok("apply_bcs"));
  ie->pending = true;
  return KRY_OK;
}

int kry_ie_init_offdiag(kry_ie *ie) {
  if (!ie) return fail(KRY_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(ie->ctx->device));
  KCHK(ie_apply_bcs_async(ie, false));
  launch_offdiag(ie->ctx->stream, ie->g, ie->fld[1], ie->fld[2], ie->band[0], ie->band[1], ie->band[2],
                 ie->band[3], ie->band[4]);
  KCHK(launch_ok("init_offdiag"));
  if (ie->solver) {
    // the recipe the one-sweep kernels (and the chain's operator kernel) evaluate the interior band entries from
    const int P = ie->g.cy + 2;
    const int64_t fld_len = ie->halo + (ie->nloc + 31) / 32 * 32 + ie->halo;
    for (int f = 0; f < 2; f++) {
      if (!ie->snap[f]) {
        double *base = nullptr;
        KCHK(dev_alloc(&base, fld_len));
        ie->snap[f] = base + ie->halo;
      }
      HIPCHK(hipMemcpyAsync(ie->snap[f] - ie->halo, ie->fld[1 + f] - ie->halo, sizeof(double) * fld_len,
                            hipMemcpyDeviceToDevice, ie->ctx->stream));
    }
    BandGen &gen = ie->op->gen;
    gen.u = ie->snap[0];
    gen.v = ie->snap[1];
    const int lr = ie->g.row0 >= 1 ? 0 : 1;  // first local row that may be an interior mesh row
    const bool has_interior = lr < ie->g.rows && ie->g.row0 + lr <= ie->g.cx;
    gen.cdiag = ie->band[2] + (has_interior ? (int64_t)lr * P + 1 : 0);
    gen.kx = ie->g.diffusion * ie->g.inv_rp / ie->g.dx;
    gen.ky = ie->g.diffusion * ie->g.inv_rp / ie->g.dy;
    gen.dx = ie->g.dx;
    gen.dy = ie->g.dy;
    gen.rdx = 1.0 / gen.dx;
    gen.rdy = 1.0 / gen.dy;
    static const bool fast_div = !(getenv("KRY_GEN_DIVIDE") && atoi(getenv("KRY_GEN_DIVIDE")) != 0);  // experimentation knob
    gen.div_ok = fast_div && shared_divisor_in_range(gen.dx) && shared_divisor_in_range(gen.dy);
    gen.row0 = ie->g.row0;
    gen.cx = ie->g.cx;
    gen.cy = ie->g.cy;
    gen.dg_bottom = 2.0 * ie->bc.weight[KRY_SIDE_BOTTOM] - 1.0;
    gen.dg_top = 2.0 * ie->bc.weight[KRY_SIDE_TOP] - 1.0;
    gen.dg_left = 2.0 * ie->bc.weight[KRY_SIDE_LEFT] - 1.0;
    ie->op->has_gen = true;
    if (ie->solver->strips_capable) {
      const int64_t sms = strip_major_size(ie->g.rows, P);
      for (int f = 0; f < 2; f++) {
        if (!ie->snap_strips[f]) KCHK(dev_alloc(&ie->snap_strips[f], sms));
        launch_to_strips(ie->ctx->stream, ie->snap[f], ie->snap_strips[f], ie->g.rows, P, true);
      }
      KCHK(launch_ok("init_offdiag (strip-major snapshots)"));
      ie->op->gen_strips = gen;
      ie->op->gen_strips.u = ie->snap_strips[0];
      ie->op->gen_strips.v = ie->snap_strips[1];
      ie->op->has_gen_strips = true;
    }
  }
  ie->pending = true;
  return KRY_OK;
}

int kry_ie_set_generated_bands(kry_ie *ie, int on) {
  if (!ie) return fail(KRY_ERR_INVALID, "NULL argument");
  config_changed();
  ie->op->use_gen = on != 0;
  return KRY_OK;
}

int kry_ie_get_generated_bands(const kry_ie *ie) {
  return ie && ie->op->has_gen && ie->op->use_gen ? 1 : 0;
}

int kry_ie_assemble_system(kry_ie *ie, double dt) {
  if (!ie) return fail(KRY_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(ie->ctx->device));
  KCHK(ie_assemble_async(ie, dt, false));
  ie->pending = true;
  return KRY_OK;
}

int kry_ie_timestep_async(kry_ie *ie, double dt) {
  if (!ie) return fail(KRY_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(ie->ctx->device));
  return ie_step_async(ie, dt);
}

int kry_ie_last_max_delta(kry_ie *ie, double *max_delta) {
  if (!ie || !max_delta) return fail(KRY_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(ie->ctx->device));
  KCHK(ie_wait(ie));
  *max_delta = *ie->maxd_h;
  return KRY_OK;
}

int kry_ie_timestep(kry_ie *ie, double dt, double *max_delta) {
  if (!ie) return fail(KRY_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(ie->ctx->device));
  KCHK(ie_step_async(ie, dt));
  KCHK(ie_wait(ie));
  if (max_delta) *max_delta = *ie->maxd_h;
  return KRY_OK;
}

int kry_ie_apply_delta(kry_ie *ie, const double *delta_local, double *max_delta) {
  if (!ie || !delta_local) return fail(KRY_ERR_INVALID, "NULL argument");
  if (!ie->solver) return fail(KRY_ERR_STATE, "step object was created without a solver");
  HIPCHK(hipSetDevice(ie->ctx->device));
  KCHK(ie_wait(ie));
  HIPCHK(hipMemcpyAsync(ie->solver->vars, delta_local, sizeof(double) * ie->nloc, hipMemcpyHostToDevice, ie->ctx->stream));
  KCHK(ie_update_async(ie));
  HIPCHK(hipStreamSynchronize(ie->ctx->stream));
  if (max_delta) *max_delta = *ie->maxd_h;
  return KRY_OK;
}

// The state update for callers that cannot allocate: no step object (whose creation
// allocates), device pointers in and out, partial results in the library's file-scope
// scratch array.  Queues three things on the context's stream and returns.
int kry_apply_delta_noalloc(kry_ctx *ctx, int cx, int cy, double *T_dev, const double *delta_dev,
                            double *max_delta_dev) {
  if (!ctx || !T_dev || !delta_dev || !max_delta_dev) return fail(KRY_ERR_INVALID, "NULL argument");
  if (cx < 1 || cy < 1) return fail(KRY_ERR_INVALID, "cells_x and cells_y must be positive");
  if ((int64_t)(cx + 2) * (cy + 2) > KRY_NOALLOC_MAX_PADDED_CELLS)
    return fail(KRY_ERR_INVALID, "a %d x %d mesh has more than %lld padded cells, the limit of kry_apply_delta_noalloc", cx, cy,
                (long long)KRY_NOALLOC_MAX_PADDED_CELLS);
  if (!ctx->scratch) return fail(KRY_ERR_STATE, "no scratch array on device %d (device ids up to %d are served)", ctx->device, kMaxDevices - 1);
  HIPCHK(hipSetDevice(ctx->device));
  IeGeom g{};
  g.cx = cx;
  g.cy = cy;
  g.row0 = 0;
  g.rows = cx + 2;
  double *part = ctx->scratch, *scal = ctx->scratch + 2 * kMaxParts;
  int np = 0;
  // every partial result and scalar the fold reads is written by this call's kernels
  // first: nothing in the scratch array has to start at zero
  launch_update_max(ctx->stream, g, T_dev, delta_dev, part, &np);
  launch_max_fold(ctx->stream, g, delta_dev, part, np, scal, 7);
  KCHK(launch_ok("kry_apply_delta_noalloc"));
  HIPCHK(hipMemcpyAsync(max_delta_dev, scal, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  return KRY_OK;
}

// ImplicitEulerApproxFact::timestep, include/time_disc.hpp:443-485.  The five
// band arrays and the right-hand-side array of the step object serve as the
// line systems' storage (lo, di, up, pivots, normalised upper diagonal,
// right-hand side / solution), which is why the object must not own a solver.
int kry_ie_timestep_approx_fact(kry_ie *ie, double dt, double *max_delta) {
  if (!ie) return fail(KRY_ERR_INVALID, "NULL argument");
  if (ie->solver) return fail(KRY_ERR_STATE, "approximate factorisation needs a step object created with m = 0");
  if (!ie->whole() || ie->ctx->nranks > 1)
    return fail(KRY_ERR_STATE, "approximate factorisation is not available on row blocks (the x sweeps cross ranks)");
  HIPCHK(hipSetDevice(ie->ctx->device));
  hipStream_t st = ie->ctx->stream;
  ScopedKernelTimer t(ie->ctx, KRY_K_ASSEMBLY);
  KCHK(ie_apply_bcs_async(ie, false));
  double *lo = ie->band[0], *di = ie->band[1], *up = ie->band[2], *dd = ie->band[3], *ud = ie->band[4];
  launch_af_rows_y(st, ie->g, ie->bc, dt, ie->fld[0], ie->fld[1], ie->fld[2], lo, di, up, ie->rhs);
  const int64_t lines = ie->g.rows, L = ie->g.cy + 2;
  launch_tdp_factor(st, lo, di, up, dd, ud, lines, L);
  launch_tdp_apply(st, lo, dd, ud, ie->rhs, ie->rhs, lines, L);
  int np = 0;
  launch_af_sweep_x(st, ie->g, ie->bc, dt, ie->fld[0], ie->fld[1], ie->rhs, dd, ud, ie->parts, &np);
  launch_af_max_final(st, ie->g, ie->parts, np, ie->maxd);
  KCHK(launch_ok("approximate factorisation step"));
  HIPCHK(hipMemcpyAsync(ie->maxd_h, ie->maxd, sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  ie->pending = false;
  ie->time += dt;
  if (max_delta) *max_delta = *ie->maxd_h;
  return KRY_OK;
}

// assemble_system_dx / assemble_system_dy, include/time_disc.hpp:487-529, for
// inspection: the three coefficients of every row of one line, computed on the
// host from the device's current velocity field with the expressions of the
// device kernels.
int kry_ie_get_af_system(kry_ie *ie, int along_y, int index, double dt, double *out) {
  if (!ie || !out) return fail(KRY_ERR_INVALID, "NULL argument");
  if (!ie->whole()) return fail(KRY_ERR_STATE, "not available on row blocks");
  const int cx = ie->g.cx, cy = ie->g.cy, P = cy + 2;
  if (index < 0 || index >= (along_y ? cx : cy)) return fail(KRY_ERR_INVALID, "line index %d out of range", index);
  HIPCHK(hipSetDevice(ie->ctx->device));
  KCHK(ie_wait(ie));
  std::vector<double> vel((size_t)(cx + 2) * P);
  HIPCHK(hipMemcpy(vel.data(), ie->fld[along_y ? 2 : 1], sizeof(double) * vel.size(), hipMemcpyDeviceToHost));
  const IeGeom &g = ie->g;
  const int rows = (along_y ? cy : cx) + 2;
  const double *w = ie->bc.weight;
  out[0] = kSNaN;
  out[1] = 1.0;
  out[2] = 2.0 * w[along_y ? KRY_SIDE_BOTTOM : KRY_SIDE_LEFT] - 1.0;
  out[(size_t)(rows - 1) * 3 + 0] = 2.0 * w[along_y ? KRY_SIDE_TOP : KRY_SIDE_RIGHT] - 1.0;
  out[(size_t)(rows - 1) * 3 + 1] = 1.0;
  out[(size_t)(rows - 1) * 3 + 2] = kSNaN;
  for (int r = 1; r < rows - 1; r++) {
    // padded position of the cell and of its two neighbours along the line
    const size_t k = along_y ? (size_t)(index + 1) * P + r : (size_t)r * P + (index + 1);
    const size_t step = along_y ? 1 : (size_t)P;
    const double h = along_y ? g.dy : g.dx;
    const double kh = g.diffusion * g.inv_rp / h;
    out[(size_t)r * 3 + 0] = (-vel[k - step] / 2.0 - kh) / h * dt;
    out[(size_t)r * 3 + 1] = 2.0 * g.diffusion * g.inv_rp / (h * h) * dt + 1.0;
    out[(size_t)r * 3 + 2] = (vel[k + step] / 2.0 - kh) / h * dt;
  }
  return KRY_OK;
}

int kry_ie_use_graph(kry_ie *ie, int on) {
  if (!ie) return fail(KRY_ERR_INVALID, "NULL argument");
  ie->use_graph = on != 0;
  return KRY_OK;
}

double kry_ie_time(const kry_ie *ie) { return ie ? ie->time : 0.0; }
int64_t kry_ie_num_eqns(const kry_ie *ie) { return ie ? ie->neq : 0; }
int64_t kry_ie_local_size(const kry_ie *ie) { return ie ? ie->nloc : 0; }
kry_gmres *kry_ie_solver(kry_ie *ie) { return ie ? ie->solver : nullptr; }
kry_op *kry_ie_operator(kry_ie *ie) { return ie ? ie->op : nullptr; }
const double *kry_ie_rhs_device(kry_ie *ie) { return ie ? ie->rhs : nullptr; }

static int download(kry_ie *ie, const double *dev, double *out, int64_t count) {
  HIPCHK(hipMemcpyAsync(out, dev, sizeof(double) * count, hipMemcpyDeviceToHost, ie->ctx->stream));
  HIPCHK(hipStreamSynchronize(ie->ctx->stream));
  ie->pending = false;
  return KRY_OK;
}

int kry_ie_get_local(kry_ie *ie, int what, double *out) {
  if (!ie || !out) return fail(KRY_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(ie->ctx->device));
  const double *src = nullptr;
  if (what >= KRY_LOCAL_FIELD_T && what <= KRY_LOCAL_FIELD_V) src = ie->fld[what];
  else if (what == KRY_LOCAL_RHS) src = ie->rhs;
  else if (what == KRY_LOCAL_DELTA) src = ie->solver ? ie->solver->vars : nullptr;
  else if (what >= KRY_LOCAL_BAND_W && what <= KRY_LOCAL_BAND_E) src = ie->band[what - KRY_LOCAL_BAND_W];
  if (!src) return fail(KRY_ERR_INVALID, "unknown quantity %d", what);
  return download(ie, src, out, ie->nloc);
}

int kry_ie_get_field(kry_ie *ie, int field, double *out_padded) {
  if (!ie || !out_padded) return fail(KRY_ERR_INVALID, "NULL argument");
  if (field < 0 || field > 2) return fail(KRY_ERR_INVALID, "bad field");
  if (!ie->whole()) return fail(KRY_ERR_STATE, "this rank owns a slab only; use kry_ie_get_local");
  HIPCHK(hipSetDevice(ie->ctx->device));
  return download(ie, ie->fld[field], out_padded, ie->nloc);
}

int kry_ie_get_bands(kry_ie *ie, double *out5) {
  if (!ie || !out5) return fail(KRY_ERR_INVALID, "NULL argument");
  if (!ie->whole()) return fail(KRY_ERR_STATE, "this rank owns a slab only; use kry_ie_get_local");
  HIPCHK(hipSetDevice(ie->ctx->device));
  for (int b = 0; b < 5; b++) KCHK(download(ie, ie->band[b], out5 + (size_t)b * ie->nloc, ie->nloc));
  return KRY_OK;
}

static int compact_download(kry_ie *ie, const double *dev, double *out) {
  if (!ie->whole()) return fail(KRY_ERR_STATE, "this rank owns a slab only; use kry_ie_get_local");
  std::vector<double> host((size_t)ie->nloc);
  KCHK(download(ie, dev, host.data(), ie->nloc));
  const int P = ie->g.cy + 2;
  for (int ip = 0; ip < ie->g.cx + 2; ip++)
    for (int jp = 0; jp < P; jp++) {
      const int64_t c = padded_to_cell(ie, ip, jp);
      if (c >= 0) out[c] = host[(size_t)ip * P + jp];
    }
  return KRY_OK;
}

int kry_ie_get_rhs(kry_ie *ie, double *out) {
  if (!ie || !out) return fail(KRY_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(ie->ctx->device));
  return compact_download(ie, ie->rhs, out);
}

int kry_ie_get_delta(kry_ie *ie, double *out) {
  if (!ie || !out) return fail(KRY_ERR_INVALID, "NULL argument");
  if (!ie->solver) return fail(KRY_ERR_STATE, "no solver");
  HIPCHK(hipSetDevice(ie->ctx->device));
  return compact_download(ie, ie->solver->vars, out);
}

int kry_ie_get_dense_system(kry_ie *ie, double *out) {
  if (!ie || !out) return fail(KRY_ERR_INVALID, "NULL argument");
  if (!ie->whole()) return fail(KRY_ERR_STATE, "this rank owns a slab only");
  HIPCHK(hipSetDevice(ie->ctx->device));
  std::vector<double> bands((size_t)5 * ie->nloc);
  KCHK(kry_ie_get_bands(ie, bands.data()));
  const int64_t N = ie->neq;
  const int P = ie->g.cy + 2, R = ie->g.cx + 2;
  memset(out, 0, sizeof(double) * (size_t)N * (size_t)N);
  const int di[5] = {-1, 0, 0, 0, 1}, dj[5] = {0, -1, 0, 1, 0};
  for (int ip = 0; ip < R; ip++)
    for (int jp = 0; jp < P; jp++) {
      const int64_t r = padded_to_cell(ie, ip, jp);
      if (r < 0) continue;
      for (int b = 0; b < 5; b++) {
        const int ni = ip + di[b], nj = jp + dj[b];
        if (ni < 0 || ni >= R || nj < 0 || nj >= P) continue;
        const int64_t c = padded_to_cell(ie, ni, nj);
        if (c < 0) continue;
        const double v = bands[(size_t)b * ie->nloc + (size_t)ip * P + jp];
        if (v != 0.0) out[r * N + c] = v;
      }
    }
  return KRY_OK;
}

}  // extern "C"
