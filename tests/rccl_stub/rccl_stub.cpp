// TEST-ONLY stand-in for librccl (tests/test_gpu_world2.py): the five entry points libdsgd_hip resolves with dlsym
// (ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy / ncclGetErrorString), implemented over a POSIX
// shared-memory segment between PROCESSES OF ONE HOST that may all sit on the same GPU.  RCCL refuses two ranks on
// one device, and the build environment reaches exactly one GPU: this shim is what lets the world = 2 arithmetic of
// the in-library collective path (reduce -> all-reduce -> apply with k_total = workers x world, all-reduced column
// ranking and dimSparsity counts, summed evaluation tallies, the asynchronous exchange) execute on real kernels.
// Selected only by an explicit DSGD_RCCL_LIB=<path> in the environment; never built or loaded by the product.
//
// ncclGroupStart / ncclGroupEnd (ONE host thread driving several ranks: dsgd_*_devices): inside a group the calls are
// recorded and return at once -- a blocking barrier would deadlock the one thread -- and ncclGroupEnd runs them phase by
// phase: every recorded rank stages its buffer and ARRIVES at the barrier (without waiting), then all wait, then all sum.
//
// ncclAllReduce here is synchronous and host-staged: wait for the stream, copy the send buffer into this rank's slot
// of the segment, barrier, add the slots in RANK ORDER (every rank computes the identical sum, as a real ring /
// tree all-reduce delivers identical results to all ranks), copy back, barrier.  A stream-ordered collective that has
// completed when it returns is a valid (slow) implementation of the stream-ordered contract.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace {
constexpr size_t kSlotBytes = 1u << 20;   // per rank: 262,144 floats (D + 1 = 47,237 for RCV1)
constexpr int kMaxRanks = 8;
constexpr double kTimeoutS = 120.0;

struct Header {
  std::atomic<int> ready;      // ranks attached
  std::atomic<int> count;      // barrier arrivals
  std::atomic<int> sense;
  int n_ranks;
};

struct Comm {
  Header* hdr = nullptr;
  char* slots = nullptr;
  size_t bytes = 0;
  int rank = 0, n = 1, local_sense = 0;
  char name[128];
  std::mutex mu;   // a communicator is used by one thread at a time (the exchange helper thread vs the caller's)
};

// the HIP runtime the process already holds (libdsgd_hip.so loaded it): no second runtime, no link-time dependency
typedef int (*memcpy_fn)(void*, const void*, size_t, int);
typedef int (*sync_fn)(void*);
memcpy_fn hip_memcpy = nullptr;
sync_fn hip_stream_sync = nullptr;
// DSGD_RCCL_STUB_HOSTMEM=1 (the CPU test of the shim itself, tests/test_rccl_stub.py): buffers are host memory
int host_memcpy(void* d, const void* s, size_t n, int) {
  memcpy(d, s, n);
  return 0;
}
int host_sync(void*) { return 0; }
bool resolve_hip() {
  if (hip_memcpy && hip_stream_sync) return true;
  const char* hm = getenv("DSGD_RCCL_STUB_HOSTMEM");
  if (hm && hm[0] == '1') {
    hip_memcpy = host_memcpy;
    hip_stream_sync = host_sync;
    return true;
  }
  const char* names[] = {"libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"};
  void* h = nullptr;
  for (const char* n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  if (!h) h = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return false;
  hip_memcpy = (memcpy_fn)dlsym(h, "hipMemcpy");
  hip_stream_sync = (sync_fn)dlsym(h, "hipStreamSynchronize");
  return hip_memcpy && hip_stream_sync;
}

void arrive(Comm* c) {   // the last arrival of a round opens it for everybody
  c->local_sense = !c->local_sense;
  if (c->hdr->count.fetch_add(1) + 1 == c->n) {
    c->hdr->count.store(0);
    c->hdr->sense.store(c->local_sense);
  }
}
bool wait_round(Comm* c) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 0; c->hdr->sense.load() != c->local_sense; ++spin) {
    if ((spin & 1023) == 1023) {
      std::this_thread::sleep_for(std::chrono::microseconds(50));
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutS) return false;
    }
  }
  return true;
}
bool barrier(Comm* c) {
  arrive(c);
  return wait_round(c);
}

// calls recorded inside ncclGroupStart / ncclGroupEnd (per thread, as in NCCL)
struct Op {
  int kind;   // 0: communicator waiting for its peers, 1: all-reduce
  Comm* c;
  const void* send;
  void* recv;
  size_t count;
  int dtype;
  void* stream;
};
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

std::atomic<int> g_ids{0};
thread_local char g_msg[160] = "no error";
int err(const char* m) {
  snprintf(g_msg, sizeof(g_msg), "rccl_stub: %s", m);
  return 2;   // ncclSystemError
}
}  // namespace

extern "C" {

struct StubUniqueId {
  char internal[128];
};

int ncclGetUniqueId(StubUniqueId* id) {
  memset(id->internal, 0, sizeof(id->internal));
  snprintf(id->internal, sizeof(id->internal), "/dsgd_rccl_stub_%d_%d", (int)getpid(), g_ids.fetch_add(1));
  return 0;
}

static int wait_ready(Comm* c) {
  const auto t0 = std::chrono::steady_clock::now();
  while (c->hdr->ready.load() < c->n) {
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutS) return err("timed out waiting for the peers");
  }
  return 0;
}

int ncclCommInitRank(void** out, int n_ranks, StubUniqueId id, int rank) {
  if (n_ranks < 1 || n_ranks > kMaxRanks || rank < 0 || rank >= n_ranks) return err("bad rank / world size");
  if (!resolve_hip()) return err("HIP runtime not found in this process");
  Comm* c = new Comm();
  c->rank = rank;
  c->n = n_ranks;
  memcpy(c->name, id.internal, sizeof(c->name) - 1);   // (Comm() is value-initialised: the last byte stays 0)
  c->bytes = sizeof(Header) + 64 + (size_t)n_ranks * kSlotBytes;
  int fd = -1;
  if (rank == 0) {
    fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) {
      delete c;
      return err("shm_open / ftruncate (rank 0)");
    }
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    struct stat st;
    for (;;) {   // rank 0 may not have created (or sized) the segment yet
      fd = shm_open(c->name, O_RDWR, 0600);
      if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= c->bytes) break;
      if (fd >= 0) close(fd);
      fd = -1;
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutS) {
        delete c;
        return err("timed out waiting for rank 0's segment");
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
  }
  void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    delete c;
    return err("mmap");
  }
  c->hdr = static_cast<Header*>(p);   // (a fresh segment is zero-filled: ready = count = sense = 0)
  c->slots = static_cast<char*>(p) + ((sizeof(Header) + 63) / 64) * 64;
  if (rank == 0) c->hdr->n_ranks = n_ranks;
  c->hdr->ready.fetch_add(1);
  *out = c;
  if (g_depth > 0) {   // the peers may be further ranks of THIS thread: wait at ncclGroupEnd
    g_ops.push_back(Op{0, c, nullptr, nullptr, 0, 0, nullptr});
    return 0;
  }
  return wait_ready(c);
}

int ncclCommDestroy(void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return 0;
  munmap(c->hdr, c->bytes);
  if (c->rank == 0) shm_unlink(c->name);
  delete c;
  return 0;
}

// datatype / op codes of rccl.h: ncclUint32 = 3, ncclInt64 = 4, ncclFloat32 = 7, ncclFloat64 = 8; ncclSum = 0
static size_t elem_size(int dtype) { return dtype == 3 ? 4 : dtype == 4 ? 8 : dtype == 7 ? 4 : dtype == 8 ? 8 : 0; }
// first half: this rank's buffer into its slot (after everything enqueued on the stream in front of the collective)
static int ar_stage(const Op& o) {
  const size_t bytes = elem_size(o.dtype) * o.count;
  if (hip_stream_sync(o.stream) != 0) return err("hipStreamSynchronize");
  char* mine = o.c->slots + (size_t)o.c->rank * kSlotBytes;
  if (hip_memcpy(mine, o.send, bytes, 2 /* hipMemcpyDeviceToHost */) != 0) return err("hipMemcpy D2H");
  return 0;
}
// second half: the slots added in RANK ORDER (identical on every rank), the sum back on the device
static int ar_combine(const Op& o) {
  const size_t bytes = elem_size(o.dtype) * o.count;
  Comm* c = o.c;
  std::vector<char> acc(bytes);
  memcpy(acc.data(), c->slots, bytes);
  for (int r = 1; r < c->n; ++r) {
    const char* s = c->slots + (size_t)r * kSlotBytes;
    for (size_t i = 0; i < o.count; ++i) {
      switch (o.dtype) {
        case 3: reinterpret_cast<uint32_t*>(acc.data())[i] += reinterpret_cast<const uint32_t*>(s)[i]; break;
        case 4: reinterpret_cast<int64_t*>(acc.data())[i] += reinterpret_cast<const int64_t*>(s)[i]; break;
        case 7: reinterpret_cast<float*>(acc.data())[i] += reinterpret_cast<const float*>(s)[i]; break;
        default: reinterpret_cast<double*>(acc.data())[i] += reinterpret_cast<const double*>(s)[i]; break;
      }
    }
  }
  if (hip_memcpy(o.recv, acc.data(), bytes, 1 /* hipMemcpyHostToDevice */) != 0) return err("hipMemcpy H2D");
  return 0;
}

// datatype / op codes of rccl.h: ncclUint32 = 3, ncclInt64 = 4, ncclFloat32 = 7, ncclFloat64 = 8; ncclSum = 0
int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, void* stream) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return err("null communicator");
  if (op != 0) return err("only ncclSum");
  const size_t es = elem_size(dtype);
  if (!es) return err("unsupported datatype");
  if (es * count > kSlotBytes) return err("message larger than the stub's slot");
  const Op o{1, c, send, recv, count, dtype, stream};
  if (g_depth > 0) {   // recorded: a blocking barrier here would deadlock a thread that drives several ranks
    for (const Op& q : g_ops)
      if (q.kind == 1 && q.c == c) return err("two all-reduces on one communicator inside one group (one slot per rank)");
    g_ops.push_back(o);
    return 0;
  }
  std::lock_guard<std::mutex> lk(c->mu);
  if (int r = ar_stage(o)) return r;
  if (!barrier(c)) return err("barrier timeout (before the sum)");
  if (int r = ar_combine(o)) return r;
  if (!barrier(c)) return err("barrier timeout (after the sum)");   // nobody refills its slot before everybody has read
  return 0;
}

int ncclGroupStart() {
  ++g_depth;
  return 0;
}

int ncclGroupEnd() {
  if (g_depth <= 0) return err("ncclGroupEnd without ncclGroupStart");
  if (--g_depth > 0) return 0;
  std::vector<Op> ops;
  ops.swap(g_ops);
  int rc = 0;
  for (const Op& o : ops)
    if (o.kind == 0 && !rc) rc = wait_ready(o.c);
  std::vector<Op> ars;
  for (const Op& o : ops)
    if (o.kind == 1) ars.push_back(o);
  if (rc || ars.empty()) return rc;
  // phase by phase over all recorded ranks: stage + arrive (nobody waits yet) ... wait ... combine + arrive ... wait
  for (const Op& o : ars) {
    o.c->mu.lock();
    if (!rc) rc = ar_stage(o);
    arrive(o.c);
  }
  for (const Op& o : ars)
    if (!wait_round(o.c) && !rc) rc = err("barrier timeout (before the sum)");
  for (const Op& o : ars) {
    if (!rc) rc = ar_combine(o);
    arrive(o.c);
  }
  for (const Op& o : ars) {
    if (!wait_round(o.c) && !rc) rc = err("barrier timeout (after the sum)");
    o.c->mu.unlock();
  }
  return rc;
}

const char* ncclGetErrorString(int code) { return code == 0 ? "no error" : g_msg; }

}  // extern "C"
