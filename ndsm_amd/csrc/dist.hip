// RCCL transport of the z-slab decomposition: point-to-point plane exchange
// between z-neighbours and with rank 0, and the 2-value reduction of the
// convergence metric.  One process per GPU; every call is enqueued on the
// library stream, so it is ordered with the kernels without host waits.
//
// The reference has no distributed mode (shared-memory OpenMP only); this is
// the xGMI-native counterpart SURVEY 8e specifies.  xGMI is point-to-point: a
// z-slab chain uses one direct link per neighbour, so a halo exchange is a
// grouped ncclSend/ncclRecv pair per neighbour - no ring collective is involved
// in the data path.
#include "common.hpp"

#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <vector>

namespace {

struct Dist {
  bool up = false;
  int rank = 0, size = 1;
  ncclComm_t comm = nullptr;
  double *d_red = nullptr;  // 2 doubles
  int group_depth = 0;      // ncclGroupStart calls not yet matched (ndsmk_dist_group_abort closes them)
};
Dist g_d;

// the runtime is shutting down or re-targeting with a live communicator: take it down with the device
void dist_release() {
  if (!g_d.up) return;
  ncclCommDestroy(g_d.comm);
  (void)hipFree(g_d.d_red);
  g_d = Dist();
}

int nccl_fail(ncclResult_t r, const char *file, int line) {
  return ndsm::fail(NDSMK_ENCCL, ncclGetErrorString(r), file, line);
}

}  // namespace

#define NDSM_NCCL(call)                                                  \
  do {                                                                   \
    ncclResult_t r_ = (call);                                            \
    if (r_ != ncclSuccess) return nccl_fail(r_, __FILE__, __LINE__);     \
  } while (0)

extern "C" {

// 128 bytes, to be created on rank 0 and handed to every rank by the launcher
// (bench.py / tests broadcast it through torch.distributed's gloo group).
int ndsmk_dist_unique_id(void *out128) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  ncclUniqueId id;
  NDSM_NCCL(ncclGetUniqueId(&id));
  std::memcpy(out128, &id, sizeof(id));
  return 0;
}

int ndsmk_dist_init(int rank, int nranks, const void *id128) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(nranks >= 1 && rank >= 0 && rank < nranks);
  if (g_d.up) return (g_d.rank == rank && g_d.size == nranks) ? 0
                     : ndsm::fail(NDSMK_EARG, "RCCL communicator already initialised differently", __FILE__, __LINE__);
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  NDSM_NCCL(ncclCommInitRank(&g_d.comm, nranks, id, rank));
  NDSM_HIP(hipMalloc((void **)&g_d.d_red, 2 * sizeof(double)));
  g_d.rank = rank;
  g_d.size = nranks;
  g_d.up = true;
  ndsm::at_reset(dist_release);
  return 0;
}

int ndsmk_dist_finalize(void) {
  if (!g_d.up) return 0;
  (void)ndsmk_sync();
  dist_release();
  return 0;
}

// Which shared objects this library's HIP and RCCL calls are bound to.  A
// process that also imports PyTorch holds a second ROCm stack (torch bundles its
// own libamdhip64 / librccl); streams must never cross from one stack to the
// other, so both answers have to name the same stack (tests check this).
int ndsmk_bound_libs(char *buf, int len) {
  Dl_info a, b;
  const char *ha = "?", *hb = "?";
  if (dladdr(reinterpret_cast<void *>(&hipStreamSynchronize), &a) && a.dli_fname) ha = a.dli_fname;
  if (dladdr(reinterpret_cast<void *>(&ncclGetUniqueId), &b) && b.dli_fname) hb = b.dli_fname;
  std::snprintf(buf, (size_t)len, "hip=%s;rccl=%s", ha, hb);
  return 0;
}

// what the COMMUNICATOR says (ncclCommUserRank / ncclCommCount), not what the caller passed to
// ndsmk_dist_init: *nranks = 0 without a communicator
int ndsmk_dist_info(int *rank, int *nranks) {
  *rank = 0;
  *nranks = 0;
  if (!g_d.up) return 0;
  NDSM_NCCL(ncclCommUserRank(g_d.comm, rank));
  NDSM_NCCL(ncclCommCount(g_d.comm, nranks));
  return 0;
}

int ndsmk_dist_rank(void) { return g_d.up ? g_d.rank : 0; }
int ndsmk_dist_size(void) { return g_d.up ? g_d.size : 1; }

int ndsmk_dist_group_start(void) {
  NDSM_CHECK_ARG(g_d.up);
  NDSM_NCCL(ncclGroupStart());
  ++g_d.group_depth;
  return 0;
}
int ndsmk_dist_group_end(void) {
  NDSM_CHECK_ARG(g_d.up);
  if (g_d.group_depth > 0) --g_d.group_depth;
  NDSM_NCCL(ncclGroupEnd());
  return 0;
}
// error paths: close whatever group an aborted sequence left open (the calls collected so far are
// issued; the caller is about to report its error anyway) - an open group would swallow every later call
int ndsmk_dist_group_abort(void) {
  while (g_d.up && g_d.group_depth > 0) {
    --g_d.group_depth;
    (void)ncclGroupEnd();
  }
  return 0;
}

int ndsmk_dist_send(const double *p, size_t count, int peer) {
  NDSM_CHECK_ARG(g_d.up && peer >= 0 && peer < g_d.size && peer != g_d.rank);
  NDSM_NCCL(ncclSend(p, count, ncclDouble, peer, g_d.comm, ndsm::stream()));
  return 0;
}

int ndsmk_dist_recv(double *p, size_t count, int peer) {
  NDSM_CHECK_ARG(g_d.up && peer >= 0 && peer < g_d.size && peer != g_d.rank);
  NDSM_NCCL(ncclRecv(p, count, ncclDouble, peer, g_d.comm, ndsm::stream()));
  return 0;
}

// the same for arrays that are not fp64 (the fp32 correction of the mixed-precision mode)
int ndsmk_dist_send_bytes(const void *p, size_t nbytes, int peer) {
  NDSM_CHECK_ARG(g_d.up && peer >= 0 && peer < g_d.size && peer != g_d.rank);
  NDSM_NCCL(ncclSend(p, nbytes, ncclInt8, peer, g_d.comm, ndsm::stream()));
  return 0;
}

int ndsmk_dist_recv_bytes(void *p, size_t nbytes, int peer) {
  NDSM_CHECK_ARG(g_d.up && peer >= 0 && peer < g_d.size && peer != g_d.rank);
  NDSM_NCCL(ncclRecv(p, nbytes, ncclInt8, peer, g_d.comm, ndsm::stream()));
  return 0;
}

// blocking: h_ms[0] <- max over ranks, h_ms[1] <- sum over ranks
int ndsmk_dist_allreduce_max_sum(double *h_ms) {
  NDSM_CHECK_ARG(g_d.up);
  hipStream_t s = ndsm::stream();
  NDSM_HIP(hipMemcpyAsync(g_d.d_red, h_ms, 2 * sizeof(double), hipMemcpyHostToDevice, s));
  NDSM_NCCL(ncclAllReduce(g_d.d_red, g_d.d_red, 1, ncclDouble, ncclMax, g_d.comm, s));
  NDSM_NCCL(ncclAllReduce(g_d.d_red + 1, g_d.d_red + 1, 1, ncclDouble, ncclSum, g_d.comm, s));
  NDSM_HIP(hipMemcpyAsync(h_ms, g_d.d_red, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
  NDSM_HIP(hipStreamSynchronize(s));
  return 0;
}

// Transport self-test on the LIVE communicator (any size, meant for the 1-rank bring-up on a one-GPU
// box): every rank sends `nelem` doubles to itself and receives them back through the same grouped
// ncclSend / ncclRecv pair a halo exchange uses - first on the main stream, then on the communication
// stream between two fences, the order an overlapped pass issues them in - and the 2-value all-reduce
// runs once.  Returns 0 if every byte arrived and the all-reduce returned (max, sum x size) of its input.
int ndsmk_dist_selftest(int nelem) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(g_d.up && nelem > 0);
  const size_t n = (size_t)nelem;
  std::vector<double> h(n), back(n);
  for (size_t i = 0; i < n; ++i) h[i] = 1.0 + 1e-3 * (double)i + (double)g_d.rank;
  double *src = nullptr, *dst = nullptr;
  NDSM_HIP(hipMalloc((void **)&src, n * sizeof(double)));
  NDSM_HIP(hipMalloc((void **)&dst, n * sizeof(double)));
  int rc = 0;
  for (int pass = 0; pass < 2 && !rc; ++pass) {
    rc = ndsmk_h2d(src, h.data(), n * sizeof(double));
    if (!rc) rc = ndsmk_fill0(dst, n * sizeof(double));
    if (!rc && pass == 1) rc = ndsmk_stream_fence(0, 1);
    if (!rc && pass == 1) rc = ndsmk_select_stream(1);
    if (!rc) {
      ncclResult_t r = ncclGroupStart();
      if (r == ncclSuccess) r = ncclSend(src, n, ncclDouble, g_d.rank, g_d.comm, ndsm::stream());
      if (r == ncclSuccess) r = ncclRecv(dst, n, ncclDouble, g_d.rank, g_d.comm, ndsm::stream());
      const ncclResult_t e = ncclGroupEnd();
      if (r == ncclSuccess) r = e;
      if (r != ncclSuccess) rc = nccl_fail(r, __FILE__, __LINE__);
    }
    if (pass == 1) {
      (void)ndsmk_select_stream(0);
      if (!rc) rc = ndsmk_stream_fence(1, 0);
    }
    if (!rc) rc = ndsmk_d2h(back.data(), dst, n * sizeof(double));
    if (!rc && std::memcmp(back.data(), h.data(), n * sizeof(double)) != 0)
      rc = ndsm::fail(NDSMK_ENCCL, pass ? "self send/recv on the communication stream returned other bytes"
                                        : "self send/recv on the main stream returned other bytes", __FILE__, __LINE__);
  }
  (void)hipFree(src);
  (void)hipFree(dst);
  if (rc) return rc;
  double ms[2] = {3.5 + g_d.rank, 0.25};
  rc = ndsmk_dist_allreduce_max_sum(ms);
  if (rc) return rc;
  if (ms[0] != 3.5 + (g_d.size - 1) || ms[1] != 0.25 * g_d.size)
    return ndsm::fail(NDSMK_ENCCL, "2-value all-reduce returned a wrong (max, sum)", __FILE__, __LINE__);
  return 0;
}

}  // extern "C"
