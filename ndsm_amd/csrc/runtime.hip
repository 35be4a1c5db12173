// Device runtime of libndsm_hip: device selection, the library stream, memory
// and copies, events.  No CPU fallback exists anywhere in this library: if no
// MI355X is visible ndsmk_init fails and every entry point above it reports
// the error to the caller.
#include "common.hpp"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

struct Runtime {
  bool up = false;
  int device = -1;
  int ncu = 256;  // compute units of the device (MI355X: 256)
  hipStream_t stream = nullptr;
  hipStream_t comm = nullptr;   // halo traffic that overlaps kernels of the main stream (z-slab worlds)
  bool on_comm = false;         // enqueue copies / RCCL calls on `comm` for now (ndsmk_select_stream)
  hipEvent_t ev0 = nullptr, ev1 = nullptr, evx = nullptr;
  char err[512] = "no error";
};

Runtime g_rt;
std::mutex g_mu;
int g_epoch = 0;                       // see common.hpp
std::vector<void (*)()> g_reset_hooks;

// the runtime is going away (shutdown or re-target): drain, let every translation unit drop what it
// holds on this device, then destroy streams and events
void tear_down() {
  (void)hipStreamSynchronize(g_rt.stream);
  (void)hipStreamSynchronize(g_rt.comm);
  std::vector<void (*)()> hooks;
  hooks.swap(g_reset_hooks);
  for (auto fn : hooks) fn();
  (void)hipStreamDestroy(g_rt.stream);
  (void)hipStreamDestroy(g_rt.comm);
  (void)hipEventDestroy(g_rt.evx);
  (void)hipEventDestroy(g_rt.ev0);
  (void)hipEventDestroy(g_rt.ev1);
  g_rt.up = false;
}

}  // namespace

namespace ndsm {

hipStream_t stream() { return g_rt.on_comm ? g_rt.comm : g_rt.stream; }
bool ready() { return g_rt.up; }
int cu_count() { return g_rt.ncu > 0 ? g_rt.ncu : 256; }
int epoch() { return g_epoch; }
void at_reset(void (*fn)()) {
  for (auto f : g_reset_hooks)
    if (f == fn) return;
  g_reset_hooks.push_back(fn);
}

int fail(int code, const char *what, const char *file, int line) {
  const char *base = std::strrchr(file, '/');
  std::snprintf(g_rt.err, sizeof(g_rt.err), "%s (code %d) at %s:%d", what, code, base ? base + 1 : file, line);
  if (const char *v = std::getenv("NDSM_HIP_VERBOSE"))
    if (v[0] == '1') std::fprintf(stderr, "ERROR(libndsm_hip):%s\n", g_rt.err);
  return code ? code : NDSMK_EARG;
}

int not_ready(const char *file, int line) {
  return fail(NDSMK_ENODEV, "HIP runtime not initialised (ndsmk_init failed or was not called)", file, line);
}

}  // namespace ndsm

extern "C" {

const char *ndsmk_last_error(void) { return g_rt.err; }

int ndsmk_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int ndsmk_init(int device) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_rt.up && (device < 0 || device == g_rt.device)) return 0;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return ndsm::fail(NDSMK_ENODEV, "no HIP device visible - libndsm_hip has no CPU path", __FILE__, __LINE__);
  if (device < 0) {
    const char *lr = std::getenv("LOCAL_RANK");
    device = lr ? std::atoi(lr) % n : 0;
  }
  NDSM_CHECK_ARG(device < n);
  if (g_rt.up) tear_down();  // re-target: everything bound to the old device goes (scratch, RCCL communicator, streams)
  NDSM_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  NDSM_HIP(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return ndsm::fail(NDSMK_ENODEV, "device is not gfx950 (this library carries MI355X code objects only)", __FILE__, __LINE__);
  NDSM_HIP(hipStreamCreateWithFlags(&g_rt.stream, hipStreamNonBlocking));
  {
    int lo = 0, hi = 0;  // numerically lower = higher priority: halo messages should not queue behind a full grid
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    NDSM_HIP(hipStreamCreateWithPriority(&g_rt.comm, hipStreamNonBlocking, hi));
  }
  NDSM_HIP(hipEventCreateWithFlags(&g_rt.evx, hipEventDisableTiming));
  g_rt.on_comm = false;
  NDSM_HIP(hipEventCreate(&g_rt.ev0));
  NDSM_HIP(hipEventCreate(&g_rt.ev1));
  g_rt.device = device;
  g_rt.ncu = prop.multiProcessorCount;
  ++g_epoch;
  g_rt.up = true;
  return 0;
}

int ndsmk_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_rt.up) return 0;
  tear_down();
  g_rt = Runtime();
  return 0;
}

int ndsmk_device_name(char *buf, int len) {
  NDSM_REQUIRE_READY();
  hipDeviceProp_t prop;
  NDSM_HIP(hipGetDeviceProperties(&prop, g_rt.device));
  std::snprintf(buf, (size_t)len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return 0;
}

void *ndsmk_stream(void) { return (void *)g_rt.stream; }

int ndsmk_alloc(void **p, size_t bytes) {
  NDSM_REQUIRE_READY();
  *p = nullptr;
  if (bytes == 0) bytes = 8;
  NDSM_HIP(hipMalloc(p, bytes));
  return 0;
}

int ndsmk_free(void *p) {
  if (!p) return 0;
  NDSM_REQUIRE_READY();
  NDSM_HIP(hipStreamSynchronize(g_rt.comm));   // halo copies / RCCL calls may still read it
  NDSM_HIP(hipStreamSynchronize(g_rt.stream));
  NDSM_HIP(hipFree(p));
  return 0;
}

int ndsmk_h2d(void *dst, const void *h_src, size_t bytes) {
  NDSM_REQUIRE_READY();
  NDSM_HIP(hipMemcpyAsync(dst, h_src, bytes, hipMemcpyHostToDevice, g_rt.stream));
  NDSM_HIP(hipStreamSynchronize(g_rt.stream));
  return 0;
}

int ndsmk_d2h(void *h_dst, const void *src, size_t bytes) {
  NDSM_REQUIRE_READY();
  NDSM_HIP(hipMemcpyAsync(h_dst, src, bytes, hipMemcpyDeviceToHost, g_rt.stream));
  NDSM_HIP(hipStreamSynchronize(g_rt.stream));
  return 0;
}

int ndsmk_d2d(void *dst, const void *src, size_t bytes) {
  NDSM_REQUIRE_READY();
  NDSM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ndsm::stream()));
  return 0;
}

int ndsmk_fill0(void *p, size_t bytes) {
  NDSM_REQUIRE_READY();
  NDSM_HIP(hipMemsetAsync(p, 0, bytes, g_rt.stream));
  return 0;
}

int ndsmk_sync(void) {
  NDSM_REQUIRE_READY();
  NDSM_HIP(hipStreamSynchronize(g_rt.comm));
  NDSM_HIP(hipStreamSynchronize(g_rt.stream));
  return 0;
}

// Two streams, one enqueueing thread.  which = 1: device copies and RCCL calls issued from now on
// go to the communication stream; 0: back to the main stream (kernels are always launched while
// 0 is selected).  ndsmk_stream_fence(from, to): everything enqueued on `from` so far happens
// before anything enqueued on `to` from now on (0 = main, 1 = communication).
int ndsmk_select_stream(int which) {
  NDSM_REQUIRE_READY();
  g_rt.on_comm = which != 0;
  return 0;
}
int ndsmk_stream_fence(int from, int to) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG((from == 0 || from == 1) && (to == 0 || to == 1) && from != to);
  NDSM_HIP(hipEventRecord(g_rt.evx, from ? g_rt.comm : g_rt.stream));
  NDSM_HIP(hipStreamWaitEvent(to ? g_rt.comm : g_rt.stream, g_rt.evx, 0));
  return 0;
}

int ndsmk_timer_start(void) {
  NDSM_REQUIRE_READY();
  NDSM_HIP(hipEventRecord(g_rt.ev0, g_rt.stream));
  return 0;
}

int ndsmk_timer_stop(double *ms) {
  NDSM_REQUIRE_READY();
  NDSM_HIP(hipEventRecord(g_rt.ev1, g_rt.stream));
  NDSM_HIP(hipEventSynchronize(g_rt.ev1));
  float f = 0;
  NDSM_HIP(hipEventElapsedTime(&f, g_rt.ev0, g_rt.ev1));
  *ms = (double)f;
  return 0;
}

}  // extern "C"
