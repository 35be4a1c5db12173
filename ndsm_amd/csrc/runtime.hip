// Device runtime of libndsm_hip: device selection, the library stream, memory
// and copies, events.  No CPU fallback exists anywhere in this library: if no
// MI355X is visible ndsmk_init fails and every entry point above it reports
// the error to the caller.
#include "common.hpp"

#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
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
  // lanes: a few more streams for INDEPENDENT small solves that are pure dispatch latency one after the
  // other (the six 2-D face problems of the vector potential); while a lane is selected everything the
  // library enqueues goes to that lane's stream (ndsmk_select_lane), created on first use
  hipStream_t lane[NDSMK_LANES] = {};
  hipEvent_t lane_ev[NDSMK_LANES] = {};
  int cur_lane = -1;
  char err[512] = "no error";
};

Runtime g_rt;
std::mutex g_mu;
// ---- background transfers (ndsmk_bg_*): one worker thread with a copy stream of its own moves the
// caller's host arrays while the main thread drives the solves (DESIGN.md "end to end") ----
struct BgJob {
  int kind = 0;                 // 0 upload-unless-zero, 1 download, 2 first touch of a host array that will be overwritten
  void *h = nullptr;
  void *d = nullptr;
  size_t bytes = 0;
  hipEvent_t after = nullptr;   // download: main-stream work that must be complete first
  int rc = 0;
  int flag = 0;                 // upload: 1 = the host array was all zero (nothing was copied)
  bool done = false;
};
struct BgState {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::deque<std::shared_ptr<BgJob>> queue;
  std::vector<std::shared_ptr<BgJob>> tickets;
  bool quit = false, running = false;
  hipStream_t copy = nullptr;
  int device = 0;
};
// on the heap and never destroyed: at process exit the worker may still sit in cv_work.wait, and
// destroying a condition variable (or a joinable std::thread) under it hangs or aborts the process
BgState &g_bg = *new BgState;

// all-zero test of a host array, up to four threads (3 GiB of initial guess: ~0.1 s on one core)
bool host_all_zero(const void *p, size_t bytes) {
  const size_t nw = bytes / 8;
  const uint64_t *w = static_cast<const uint64_t *>(p);
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt > 4 ? 4 : (nt < 1 ? 1 : nt);
  if (nw < (size_t)1 << 22) nt = 1;
  std::vector<int> nz(nt, 0);
  auto scan = [&](unsigned t) {
    const size_t a = nw * t / nt, b = nw * (t + 1) / nt;
    uint64_t acc = 0;
    size_t i = a;
    for (; i + 8 <= b; i += 8) {
      acc |= w[i] | w[i + 1] | w[i + 2] | w[i + 3] | w[i + 4] | w[i + 5] | w[i + 6] | w[i + 7];
      if (acc) break;                       // leave early once something was seen
    }
    for (; i < b && !acc; ++i) acc |= w[i];
    nz[t] = acc != 0;
  };
  std::vector<std::thread> th;
  for (unsigned t = 1; t < nt; ++t) th.emplace_back(scan, t);
  scan(0);
  for (auto &x : th) x.join();
  for (unsigned t = 0; t < nt; ++t)
    if (nz[t]) return false;
  const unsigned char *tail = static_cast<const unsigned char *>(p) + nw * 8;
  for (size_t i = 0; i < bytes - nw * 8; ++i)
    if (tail[i]) return false;
  return true;   // note: -0.0 has a non-zero bit pattern and counts as data
}

// First touch of a host array the call is going to overwrite completely (numpy.zeros / numpy.empty hand over
// untouched pages): a device-to-host copy into untouched pages runs at ~13 GB/s on this box (the page faults are
// taken one by one inside the copy), into touched ones at ~50.  Four threads rewrite one word per 4 KiB page with
// its own value: the caller's data is intact if the call fails later.
void host_first_touch(void *p, size_t bytes) {
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt > 4 ? 4 : (nt < 1 ? 1 : nt);
  const size_t page = 4096;
  char *base = static_cast<char *>(p);
  auto touch = [&](unsigned t) {
    const size_t a = (bytes / nt) * t, b = t + 1 == nt ? bytes : (bytes / nt) * (t + 1);
    size_t o = ((reinterpret_cast<size_t>(base) + a + page - 1) & ~(page - 1)) - reinterpret_cast<size_t>(base);
    for (; o + 8 <= b; o += page) {   // write back what is there: the page is faulted in, its contents stay
      volatile unsigned long long *w = reinterpret_cast<volatile unsigned long long *>(base + o);
      *w = *w;
    }
  };
  std::vector<std::thread> th;
  for (unsigned t = 1; t < nt; ++t) th.emplace_back(touch, t);
  touch(0);
  for (auto &x : th) x.join();
}

void bg_worker() {
  (void)hipSetDevice(g_bg.device);
  for (;;) {
    std::shared_ptr<BgJob> j;
    {
      std::unique_lock<std::mutex> lk(g_bg.mu);
      g_bg.cv_work.wait(lk, [] { return g_bg.quit || !g_bg.queue.empty(); });
      if (g_bg.queue.empty()) return;
      j = g_bg.queue.front();
      g_bg.queue.pop_front();
    }
    hipError_t e = hipSuccess;
    if (j->kind == 2) {
      host_first_touch(j->h, j->bytes);
    } else if (j->kind == 0) {
      j->flag = host_all_zero(j->h, j->bytes) ? 1 : 0;
      if (!j->flag) {
        e = hipMemcpyAsync(j->d, j->h, j->bytes, hipMemcpyHostToDevice, g_bg.copy);
        if (e == hipSuccess) e = hipStreamSynchronize(g_bg.copy);
      }
    } else {
      e = hipStreamWaitEvent(g_bg.copy, j->after, 0);
      if (e == hipSuccess) e = hipMemcpyAsync(j->h, j->d, j->bytes, hipMemcpyDeviceToHost, g_bg.copy);
      if (e == hipSuccess) e = hipStreamSynchronize(g_bg.copy);
    }
    {
      std::lock_guard<std::mutex> lk(g_bg.mu);
      j->rc = (int)e;
      j->done = true;
    }
    g_bg.cv_done.notify_all();
  }
}

void bg_stop() {
  if (!g_bg.running) return;
  {
    std::lock_guard<std::mutex> lk(g_bg.mu);
    g_bg.quit = true;
  }
  g_bg.cv_work.notify_all();
  g_bg.th.join();
  for (auto &j : g_bg.tickets)
    if (j->after) (void)hipEventDestroy(j->after);
  g_bg.tickets.clear();
  g_bg.queue.clear();
  if (g_bg.copy) (void)hipStreamDestroy(g_bg.copy);
  g_bg.copy = nullptr;
  g_bg.quit = false;
  g_bg.running = false;
}

std::vector<void (*)()> g_lowmem_hooks;   // called when a device allocation fails, before it is retried once
int g_epoch = 0;                       // see common.hpp
std::vector<void (*)()> g_reset_hooks;

// the runtime is going away (shutdown or re-target): drain, let every translation unit drop what it
// holds on this device, then destroy streams and events
void tear_down() {
  bg_stop();
  (void)hipStreamSynchronize(g_rt.stream);
  (void)hipStreamSynchronize(g_rt.comm);
  for (int l = 0; l < NDSMK_LANES; ++l)
    if (g_rt.lane[l]) (void)hipStreamSynchronize(g_rt.lane[l]);
  std::vector<void (*)()> hooks;
  hooks.swap(g_reset_hooks);
  for (auto fn : hooks) fn();
  (void)hipStreamDestroy(g_rt.stream);
  (void)hipStreamDestroy(g_rt.comm);
  for (int l = 0; l < NDSMK_LANES; ++l) {
    if (g_rt.lane[l]) (void)hipStreamDestroy(g_rt.lane[l]);
    if (g_rt.lane_ev[l]) (void)hipEventDestroy(g_rt.lane_ev[l]);
    g_rt.lane[l] = nullptr;
    g_rt.lane_ev[l] = nullptr;
  }
  g_rt.cur_lane = -1;
  (void)hipEventDestroy(g_rt.evx);
  (void)hipEventDestroy(g_rt.ev0);
  (void)hipEventDestroy(g_rt.ev1);
  g_rt.up = false;
}

}  // namespace

namespace ndsm {

hipStream_t stream() {
  if (g_rt.on_comm) return g_rt.comm;
  return g_rt.cur_lane >= 0 ? g_rt.lane[g_rt.cur_lane] : g_rt.stream;
}
int lane() { return g_rt.cur_lane; }
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

// host-side argument errors of the Fortran layer: leave their text where ndsmk_last_error finds it
int ndsmk_note_error(int code, const char *what) {
  std::snprintf(g_rt.err, sizeof(g_rt.err), "%s (code %d)", what ? what : "error", code);
  if (const char *v = std::getenv("NDSM_HIP_VERBOSE"))
    if (v[0] == '1') std::fprintf(stderr, "ERROR(libndsm_hip):%s\n", g_rt.err);
  return code;
}

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
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipErrorOutOfMemory && !g_lowmem_hooks.empty()) {
    // memory the library keeps for the caller's convenience (the cached vector-potential context) goes first
    (void)hipGetLastError();
    for (auto fn : g_lowmem_hooks) fn();
    e = hipMalloc(p, bytes);
  }
  NDSM_HIP(e);
  return 0;
}

// fn: release whatever is held only as a cache (must be safe to call at any allocation; a holder that is in
// the middle of using its memory does nothing)
void ndsmk_on_low_memory(void (*fn)(void)) {
  for (auto f : g_lowmem_hooks)
    if (f == fn) return;
  g_lowmem_hooks.push_back(fn);
}

int ndsmk_free(void *p) {
  if (!p) return 0;
  NDSM_REQUIRE_READY();
  NDSM_HIP(hipStreamSynchronize(g_rt.comm));   // halo copies / RCCL calls may still read it
  NDSM_HIP(hipStreamSynchronize(g_rt.stream));
  for (int l = 0; l < NDSMK_LANES; ++l)
    if (g_rt.lane[l]) NDSM_HIP(hipStreamSynchronize(g_rt.lane[l]));
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
  NDSM_HIP(hipMemsetAsync(p, 0, bytes, g_rt.cur_lane >= 0 ? g_rt.lane[g_rt.cur_lane] : g_rt.stream));
  return 0;
}

int ndsmk_sync(void) {
  NDSM_REQUIRE_READY();
  NDSM_HIP(hipStreamSynchronize(g_rt.comm));
  NDSM_HIP(hipStreamSynchronize(g_rt.stream));
  for (int l = 0; l < NDSMK_LANES; ++l)
    if (g_rt.lane[l]) NDSM_HIP(hipStreamSynchronize(g_rt.lane[l]));
  return 0;
}

// Lanes.  ndsmk_select_lane(l), 0 <= l < NDSMK_LANES: kernels, device copies and fills issued from now on go
// to lane l's stream (and the reductions use lane l's scratch); -1: back to the main stream.
// ndsmk_lane_fence(l, 0): lane l waits for everything enqueued on the main stream so far;
// ndsmk_lane_fence(l, 1): the main stream waits for everything enqueued on lane l so far.
// ndsmk_lane_sync(l): the host waits for lane l.
int ndsmk_select_lane(int lane) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(lane >= -1 && lane < NDSMK_LANES);
  if (lane >= 0 && !g_rt.lane[lane]) {
    NDSM_HIP(hipStreamCreateWithFlags(&g_rt.lane[lane], hipStreamNonBlocking));
    NDSM_HIP(hipEventCreateWithFlags(&g_rt.lane_ev[lane], hipEventDisableTiming));
  }
  g_rt.cur_lane = lane;
  return 0;
}
int ndsmk_lane_fence(int lane, int to_main) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(lane >= 0 && lane < NDSMK_LANES && g_rt.lane[lane]);
  if (to_main) {
    NDSM_HIP(hipEventRecord(g_rt.lane_ev[lane], g_rt.lane[lane]));
    NDSM_HIP(hipStreamWaitEvent(g_rt.stream, g_rt.lane_ev[lane], 0));
  } else {
    NDSM_HIP(hipEventRecord(g_rt.lane_ev[lane], g_rt.stream));
    NDSM_HIP(hipStreamWaitEvent(g_rt.lane[lane], g_rt.lane_ev[lane], 0));
  }
  return 0;
}
// has everything enqueued on the lane's stream finished?  1 yes, 0 not yet (never blocks), < 0: error
int ndsmk_lane_idle(int lane) {
  if (!ndsm::ready() || lane < 0 || lane >= NDSMK_LANES || !g_rt.lane[lane]) return -1;
  const hipError_t e = hipStreamQuery(g_rt.lane[lane]);
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) {
    (void)hipGetLastError();
    return 0;
  }
  return -1;
}
// Graphs: a launch-bound sequence that repeats unchanged (one V-cycle + metric of a small 2-D solve: ~190
// launches of a few microseconds) is recorded once from the selected lane's stream and replayed as ONE graph launch.
// ndsmk_capture_begin(): everything enqueued on the selected lane from now on is recorded, not executed;
// ndsmk_capture_end(&exec): stop recording and instantiate; ndsmk_graph_launch(exec): replay on the selected
// lane; ndsmk_graph_destroy(exec).  The recorded calls must not allocate, synchronise or set kernel
// attributes: the caller runs the sequence once un-recorded first (every lazily created scratch then exists).
int ndsmk_capture_begin(void) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(g_rt.cur_lane >= 0);
  NDSM_HIP(hipStreamBeginCapture(g_rt.lane[g_rt.cur_lane], hipStreamCaptureModeRelaxed));
  return 0;
}
int ndsmk_capture_end(void **exec) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(g_rt.cur_lane >= 0 && exec);
  *exec = nullptr;
  hipGraph_t g = nullptr;
  NDSM_HIP(hipStreamEndCapture(g_rt.lane[g_rt.cur_lane], &g));
  if (std::getenv("NDSM_HIP_FAKE_GRAPH_FAILURE")) {   // tests: the caller's fall-back to plain enqueueing
    (void)hipGraphDestroy(g);
    return ndsm::fail(NDSMK_EARG, "graph instantiation failed (simulated)", __FILE__, __LINE__);
  }
  hipGraphExec_t e = nullptr;
  hipError_t rc = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  NDSM_HIP(rc);
  *exec = (void *)e;
  return 0;
}
int ndsmk_graph_launch(void *exec) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(exec);
  NDSM_HIP(hipGraphLaunch((hipGraphExec_t)exec, ndsm::stream()));
  return 0;
}
int ndsmk_graph_destroy(void *exec) {
  if (!exec) return 0;
  NDSM_HIP(hipGraphExecDestroy((hipGraphExec_t)exec));
  return 0;
}

int ndsmk_lane_sync(int lane) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(lane >= 0 && lane < NDSMK_LANES && g_rt.lane[lane]);
  NDSM_HIP(hipStreamSynchronize(g_rt.lane[lane]));
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

// ---- background transfers ------------------------------------------------------------------
static int bg_submit(std::shared_ptr<BgJob> j, int *ticket) {
  if (!g_bg.running) {
    g_bg.device = g_rt.device;
    NDSM_HIP(hipStreamCreateWithFlags(&g_bg.copy, hipStreamNonBlocking));
    g_bg.quit = false;
    g_bg.th = std::thread(bg_worker);
    g_bg.running = true;
  }
  {
    std::lock_guard<std::mutex> lk(g_bg.mu);
    g_bg.tickets.push_back(j);
    *ticket = (int)g_bg.tickets.size() - 1;
    g_bg.queue.push_back(j);
  }
  g_bg.cv_work.notify_one();
  return 0;
}

// queue: if the host array is not all zero, copy it to d_dst (the caller zero-fills d_dst itself when
// ndsmk_bg_wait reports flag = 1).  Returns at once; *ticket names the job.
int ndsmk_bg_upload_unless_zero(const void *h_src, void *d_dst, size_t bytes, int *ticket) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(h_src && d_dst && ticket);
  auto j = std::make_shared<BgJob>();
  j->kind = 0;
  j->h = const_cast<void *>(h_src);
  j->d = d_dst;
  j->bytes = bytes;
  return bg_submit(j, ticket);
}

// queue: once everything enqueued on the main stream SO FAR has finished, copy d_src to the host array
int ndsmk_bg_download(void *h_dst, const void *d_src, size_t bytes, int *ticket) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(h_dst && d_src && ticket);
  auto j = std::make_shared<BgJob>();
  j->kind = 1;
  j->h = h_dst;
  j->d = const_cast<void *>(d_src);
  j->bytes = bytes;
  NDSM_HIP(hipEventCreateWithFlags(&j->after, hipEventDisableTiming));
  NDSM_HIP(hipEventRecord(j->after, g_rt.stream));
  return bg_submit(j, ticket);
}

// queue: touch every page of a host array that this call will overwrite completely (a write fault per page;
// the contents are not changed) - ahead of the downloads into it
int ndsmk_bg_first_touch(void *h_dst, size_t bytes, int *ticket) {
  NDSM_REQUIRE_READY();
  NDSM_CHECK_ARG(h_dst && ticket);
  auto j = std::make_shared<BgJob>();
  j->kind = 2;
  j->h = h_dst;
  j->bytes = bytes;
  return bg_submit(j, ticket);
}

// blocking: job `ticket` has finished; *flag (may be NULL) = its all-zero verdict (uploads)
int ndsmk_bg_wait(int ticket, int *flag) {
  NDSM_REQUIRE_READY();
  std::shared_ptr<BgJob> j;
  {
    std::unique_lock<std::mutex> lk(g_bg.mu);
    NDSM_CHECK_ARG(ticket >= 0 && ticket < (int)g_bg.tickets.size());
    j = g_bg.tickets[ticket];
    g_bg.cv_done.wait(lk, [&] { return j->done; });
  }
  if (flag) *flag = j->flag;
  if (j->rc) return ndsm::fail(j->rc, hipGetErrorString((hipError_t)j->rc), __FILE__, __LINE__);
  return 0;
}

// blocking: every queued job has finished; tickets are void afterwards.  Returns the first error.
int ndsmk_bg_drain(void) {
  if (!g_bg.running) return 0;
  int rc = 0;
  std::vector<std::shared_ptr<BgJob>> all;
  {
    std::unique_lock<std::mutex> lk(g_bg.mu);
    g_bg.cv_done.wait(lk, [] {
      for (auto &j : g_bg.tickets)
        if (!j->done) return false;
      return true;
    });
    all.swap(g_bg.tickets);
  }
  for (auto &j : all) {
    if (j->after) (void)hipEventDestroy(j->after);
    if (j->rc && !rc) rc = j->rc;
  }
  if (rc) return ndsm::fail(rc, hipGetErrorString((hipError_t)rc), __FILE__, __LINE__);
  return 0;
}

// pinned host memory (staging of the six boundary faces), device memory figures
int ndsmk_host_alloc(void **p, size_t bytes) {
  NDSM_REQUIRE_READY();
  *p = nullptr;
  NDSM_HIP(hipHostMalloc(p, bytes ? bytes : 8, hipHostMallocDefault));
  return 0;
}
int ndsmk_host_free(void *p) {
  if (!p) return 0;
  NDSM_HIP(hipHostFree(p));
  return 0;
}
int ndsmk_mem_info(size_t *free_bytes, size_t *total_bytes) {
  NDSM_REQUIRE_READY();
  NDSM_HIP(hipMemGetInfo(free_bytes, total_bytes));
  return 0;
}
// asynchronous on the main stream (h_src must be pinned and stay untouched until the stream has passed it)
int ndsmk_h2d_async(void *dst, const void *h_src, size_t bytes) {
  NDSM_REQUIRE_READY();
  NDSM_HIP(hipMemcpyAsync(dst, h_src, bytes, hipMemcpyHostToDevice, g_rt.stream));
  return 0;
}
void ndsmk_at_reset(void (*fn)(void)) { ndsm::at_reset(fn); }

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
