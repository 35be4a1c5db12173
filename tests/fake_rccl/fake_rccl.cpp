// TEST DOUBLE - not part of the product.  A stand-in for the handful of RCCL
// entry points libndsm_hip calls, for rehearsing the multi-rank code path on a
// box with ONE GPU (RCCL itself refuses two ranks on one device: "Duplicate GPU
// detected").  Linked under the product's own objects (libndsm_hip_fake.so,
// see the Makefile) it moves the data through a POSIX shared-memory file instead of xGMI, with RCCL's matching rules kept
// strict so that the mistakes a real run would hang or corrupt on fail here:
//   * point-to-point operations between a pair of ranks match in issue order
//     and must agree on the element count (checked);
//   * a send completes only when the peer has received it (rendezvous) - an
//     ungrouped send/send pattern deadlocks here as it would on the device;
//   * the operations of one ncclGroupStart/End progress together;
//   * every wait has a deadline and returns ncclInternalError instead of hanging.
// Operations run on the host: the stream is drained, the payload copied D2H /
// H2D with hipMemcpy.  ncclDouble and ncclInt8 payloads (all the library uses).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr int MAXR = 8;
constexpr int NSLOT = 4;

struct SlotHdr {
  std::atomic<int> full;   // 0 empty, 1 holds a message
  size_t count;
};

struct Shared {
  std::atomic<int> joined;
  std::atomic<int> left;
  std::atomic<unsigned> bar_count, bar_gen;
  int nranks;
  size_t slot_bytes;
  double red[MAXR][2];
  SlotHdr slot[MAXR][MAXR][NSLOT];   // [src][dst][ring]
  // payload area follows
};

struct Op {
  bool send;
  void *p;
  size_t count;   // BYTES
  int peer;
  hipStream_t s;
  int slot = -1;        // ring position taken by this op
  int state = 0;        // send: 0 pending, 1 posted, 2 consumed; recv: 0 pending, 2 done
};

struct Comm {
  Shared *sh = nullptr;
  size_t map_bytes = 0;
  char name[64];
  int rank = 0, n = 1;
  unsigned head[MAXR] = {0}, tail[MAXR] = {0};   // next ring slot to post to / read from, per peer
  std::vector<double> stage;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
Comm *g_comm = nullptr;

double timeout_s() {
  const char *e = getenv("FAKE_RCCL_TIMEOUT");
  return e ? atof(e) : 60.0;
}

size_t slot_bytes_env() {
  const char *e = getenv("FAKE_RCCL_SLOT_MB");
  return (size_t)(e ? atoi(e) : 4) << 20;
}

char *payload(Comm *c, int src, int dst, int k) {
  char *base = reinterpret_cast<char *>(c->sh) + ((sizeof(Shared) + 4095) & ~size_t(4095));
  return base + (((size_t)src * MAXR + dst) * NSLOT + k) * c->sh->slot_bytes;
}

using Clock = std::chrono::steady_clock;
bool expired(Clock::time_point t0) {
  return std::chrono::duration<double>(Clock::now() - t0).count() > timeout_s();
}

ncclResult_t complain(const char *what) {
  fprintf(stderr, "fake_rccl[rank %d]: %s\n", g_comm ? g_comm->rank : -1, what);
  return ncclInternalError;
}

ncclResult_t barrier(Comm *c) {
  Shared *s = c->sh;
  unsigned gen = s->bar_gen.load();
  if (s->bar_count.fetch_add(1) + 1 == (unsigned)c->n) {
    s->bar_count.store(0);
    s->bar_gen.fetch_add(1);
    return ncclSuccess;
  }
  auto t0 = Clock::now();
  while (s->bar_gen.load() == gen) {
    if (expired(t0)) return complain("barrier timed out (a rank is missing from a collective)");
    std::this_thread::yield();
  }
  return ncclSuccess;
}

// progress engine: returns when every op of the batch is complete
ncclResult_t run_ops(Comm *c, std::vector<Op> &ops) {
  for (auto &o : ops)
    if (hipStreamSynchronize(o.s) != hipSuccess) return complain("stream sync failed");
  auto t0 = Clock::now();
  size_t left = ops.size();
  while (left) {
    bool moved = false;
    // per-peer FIFO order: an op may only start if no earlier op of the same kind to/from that peer is pending
    for (size_t i = 0; i < ops.size(); ++i) {
      Op &o = ops[i];
      if (o.state == 2) continue;
      bool blocked = false;
      for (size_t j = 0; j < i; ++j)
        if (ops[j].send == o.send && ops[j].peer == o.peer && ops[j].state == 0) blocked = true;
      if (o.send) {
        if (o.state == 0 && !blocked) {
          int k = c->head[o.peer] % NSLOT;
          SlotHdr &h = c->sh->slot[c->rank][o.peer][k];
          if (h.full.load(std::memory_order_acquire) == 0) {
            if (o.count > c->sh->slot_bytes) return complain("message larger than FAKE_RCCL_SLOT_MB");
            if (hipMemcpy(payload(c, c->rank, o.peer, k), o.p, o.count, hipMemcpyDeviceToHost) != hipSuccess)
              return complain("D2H failed");
            h.count = o.count;
            h.full.store(1, std::memory_order_release);
            o.slot = k; o.state = 1; c->head[o.peer]++;
            moved = true;
          }
        } else if (o.state == 1) {
          if (c->sh->slot[c->rank][o.peer][o.slot].full.load(std::memory_order_acquire) == 0) {
            o.state = 2; --left; moved = true;
          }
        }
      } else if (!blocked) {
        int k = c->tail[o.peer] % NSLOT;
        SlotHdr &h = c->sh->slot[o.peer][c->rank][k];
        if (h.full.load(std::memory_order_acquire) == 1) {
          if (h.count != o.count) {
            fprintf(stderr, "fake_rccl[rank %d]: recv of %zu bytes from rank %d matched a send of %zu\n",
                    c->rank, o.count, o.peer, h.count);
            return ncclInternalError;
          }
          if (hipMemcpy(o.p, payload(c, o.peer, c->rank, k), o.count, hipMemcpyHostToDevice) != hipSuccess)
            return complain("H2D failed");
          h.full.store(0, std::memory_order_release);
          o.state = 2; c->tail[o.peer]++; --left; moved = true;
        }
      }
    }
    if (!moved) {
      if (expired(t0)) {
        for (auto &o : ops)
          if (o.state != 2)
            fprintf(stderr, "fake_rccl[rank %d]: stuck %s peer %d count %zu state %d\n", c->rank,
                    o.send ? "send to" : "recv from", o.peer, o.count, o.state);
        return complain("point-to-point batch timed out (unmatched send/recv = a deadlock on the device)");
      }
      std::this_thread::yield();
    }
  }
  return ncclSuccess;
}

ncclResult_t submit(Op o) {
  if (!g_comm) return complain("no communicator");
  if (g_depth > 0) {
    g_ops.push_back(o);
    return ncclSuccess;
  }
  std::vector<Op> one{o};
  return run_ops(g_comm, one);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  memset(id, 0, sizeof(*id));
  auto now = std::chrono::high_resolution_clock::now().time_since_epoch().count();
  snprintf(id->internal, sizeof(id->internal), "/ndsm_fake_rccl_%d_%llx", (int)getpid(), (unsigned long long)now);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  Comm *c = new Comm;
  c->rank = rank; c->n = nranks;
  snprintf(c->name, sizeof(c->name), "%s", id.internal);
  size_t sb = slot_bytes_env();
  c->map_bytes = ((sizeof(Shared) + 4095) & ~size_t(4095)) + (size_t)MAXR * MAXR * NSLOT * sb;
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return complain("shm_open failed");
  if (ftruncate(fd, (off_t)c->map_bytes) != 0) return complain("ftruncate failed");   // zero-filled, sparse
  void *m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return complain("mmap failed");
  c->sh = static_cast<Shared *>(m);
  c->sh->nranks = nranks;
  c->sh->slot_bytes = sb;
  g_comm = c;
  c->sh->joined.fetch_add(1);
  auto t0 = Clock::now();
  while (c->sh->joined.load() < nranks) {
    if (expired(t0)) return complain("ncclCommInitRank timed out waiting for the other ranks");
    std::this_thread::yield();
  }
  *comm = reinterpret_cast<ncclComm_t>(c);
  fprintf(stderr, "fake_rccl: rank %d of %d up (TEST DOUBLE over %s, not xGMI)\n", rank, nranks, c->name);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  if (!c) return ncclSuccess;
  if (c->sh->left.fetch_add(1) + 1 == c->n) shm_unlink(c->name);
  munmap(c->sh, c->map_bytes);
  if (g_comm == c) g_comm = nullptr;
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
  *count = reinterpret_cast<const Comm *>(comm)->n;
  return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank) {
  *rank = reinterpret_cast<const Comm *>(comm)->rank;
  return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclInvalidArgument: return "invalid argument";
    case ncclInternalError: return "internal error (fake_rccl: see stderr)";
    default: return "error";
  }
}

ncclResult_t ncclGroupStart() {
  ++g_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return complain("ncclGroupEnd without ncclGroupStart");
  if (--g_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(g_ops);
  if (ops.empty()) return ncclSuccess;
  return run_ops(g_comm, ops);
}

static size_t esize(ncclDataType_t dt) { return dt == ncclDouble ? 8 : (dt == ncclInt8 ? 1 : 0); }

ncclResult_t ncclSend(const void *p, size_t count, ncclDataType_t dt, int peer, ncclComm_t, hipStream_t s) {
  if (!esize(dt)) return complain("only ncclDouble and ncclInt8 are modelled");
  Op o{true, const_cast<void *>(p), count * esize(dt), peer, s};
  return submit(o);
}

ncclResult_t ncclRecv(void *p, size_t count, ncclDataType_t dt, int peer, ncclComm_t, hipStream_t s) {
  if (!esize(dt)) return complain("only ncclDouble and ncclInt8 are modelled");
  Op o{false, p, count * esize(dt), peer, s};
  return submit(o);
}

ncclResult_t ncclAllReduce(const void *in, void *out, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t s) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  if (dt != ncclDouble || count > 2 || (op != ncclMax && op != ncclSum)) return complain("allreduce shape not modelled");
  if (hipStreamSynchronize(s) != hipSuccess) return complain("stream sync failed");
  double v[2] = {0, 0};
  if (hipMemcpy(v, in, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return complain("D2H failed");
  for (size_t i = 0; i < count; ++i) c->sh->red[c->rank][i] = v[i];
  std::atomic_thread_fence(std::memory_order_seq_cst);
  ncclResult_t r = barrier(c);
  if (r != ncclSuccess) return r;
  for (size_t i = 0; i < count; ++i) {
    double acc = c->sh->red[0][i];
    for (int k = 1; k < c->n; ++k) acc = (op == ncclMax) ? (c->sh->red[k][i] > acc ? c->sh->red[k][i] : acc) : acc + c->sh->red[k][i];
    v[i] = acc;
  }
  r = barrier(c);
  if (r != ncclSuccess) return r;
  if (hipMemcpy(out, v, count * 8, hipMemcpyHostToDevice) != hipSuccess) return complain("H2D failed");
  return ncclSuccess;
}

}  // extern "C"
