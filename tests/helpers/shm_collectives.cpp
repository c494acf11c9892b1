// shm_collectives.cpp — TEST HELPER, not part of the product.
// The six RCCL entry points libcalico_rp.so binds, implemented over POSIX shared memory
// between processes of one host, so that the multi-rank control flow of the library and of
// bench.py can run with two ranks on ONE GPU (RCCL refuses "Duplicate GPU").  Selected with
// CALICO_RP_COLLECTIVES_LIB.  Sums are taken in rank order (deterministic).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

namespace {
constexpr size_t SLOT_BYTES = (size_t)48 << 20;  // per rank
struct Header {
  std::atomic<int> ready, count, gen, aborted;
};
struct FakeComm {
  int nranks, rank;
  Header* hdr;
  char* slots;
  size_t map_bytes;
  std::string name;
};
// These collectives block on the host, so the library's own time-out (a polled stream wait behind
// an asynchronous RCCL call) never sees them: a peer that died without aborting the communicator
// would leave the others spinning for ever, test runner gone or not.  Every wait here gives up after
// CALICO_RP_COMM_TIMEOUT_S seconds (default 120) and marks the communicator aborted.
double wait_limit_s() {
  const char* e = getenv("CALICO_RP_COMM_TIMEOUT_S");
  const double v = e ? atof(e) : 0.0;
  return v > 0.0 ? v : 120.0;
}
struct Deadline {
  std::chrono::steady_clock::time_point end =
      std::chrono::steady_clock::now() + std::chrono::duration_cast<std::chrono::steady_clock::duration>(
                                             std::chrono::duration<double>(wait_limit_s()));
  unsigned spins = 0;
  bool over() { return (++spins & 1023u) == 0 && std::chrono::steady_clock::now() > end; }
};
// false: a rank has aborted the communicator (ncclCommAbort) or has not turned up in time: nobody waits any longer
bool barrier(FakeComm* c) {
  const int gen = c->hdr->gen.load();
  if (c->hdr->count.fetch_add(1) + 1 == c->nranks) {
    c->hdr->count.store(0);
    c->hdr->gen.fetch_add(1);
  } else {
    Deadline d;
    while (c->hdr->gen.load() == gen) {
      if (c->hdr->aborted.load()) return false;
      if (d.over()) {
        c->hdr->aborted.store(1);
        return false;
      }
      sched_yield();
    }
  }
  return !c->hdr->aborted.load();
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id, 0, sizeof *id);
  std::random_device rd;
  unsigned long long tag = ((unsigned long long)rd() << 32) ^ rd() ^ (unsigned long long)getpid();
  std::snprintf(id->internal, sizeof id->internal, "/calico_shm_%016llx", tag);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  FakeComm* c = new FakeComm;
  c->nranks = nranks;
  c->rank = rank;
  c->name = id.internal;
  c->map_bytes = 4096 + SLOT_BYTES * (size_t)nranks;
  int fd = -1;
  if (rank == 0) {
    fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) return ncclSystemError;
  } else {
    for (int tries = 0; tries < 20000 && fd < 0; ++tries) {
      fd = shm_open(c->name.c_str(), O_RDWR, 0600);
      struct stat st;
      if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < c->map_bytes)) {
        close(fd);
        fd = -1;
      }
      if (fd < 0) usleep(1000);
    }
    if (fd < 0) return ncclSystemError;
  }
  void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  c->hdr = reinterpret_cast<Header*>(p);
  c->slots = reinterpret_cast<char*>(p) + 4096;
  c->hdr->ready.fetch_add(1);
  {
    Deadline d;
    while (c->hdr->ready.load() < nranks) {
      if (d.over()) {
        if (rank == 0) shm_unlink(c->name.c_str());
        munmap(p, c->map_bytes);
        delete c;
        return ncclSystemError;
      }
      sched_yield();
    }
  }
  barrier(c);
  if (rank == 0) shm_unlink(c->name.c_str());  // everyone has it mapped
  *out = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

// As with RCCL inside one node, a peer only learns of the abort because its own collective stops
// making progress; here the flag in the shared header ends the peers' waits at once.
ncclResult_t ncclCommAbort(ncclComm_t comm) {
  FakeComm* c = reinterpret_cast<FakeComm*>(comm);
  c->hdr->aborted.store(1);
  munmap(c->hdr, c->map_bytes);
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  FakeComm* c = reinterpret_cast<FakeComm*>(comm);
  munmap(c->hdr, c->map_bytes);
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
  FakeComm* c = reinterpret_cast<FakeComm*>(comm);
  if (dt != ncclDouble || op != ncclSum || count * 8 > SLOT_BYTES) return ncclInvalidArgument;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(c->slots + SLOT_BYTES * c->rank, send, count * 8, hipMemcpyDeviceToHost) != hipSuccess)
    return ncclUnhandledCudaError;
  if (!barrier(c)) return ncclSystemError;
  std::vector<double> sum(count, 0.0);
  for (int r = 0; r < c->nranks; ++r) {
    const double* s = reinterpret_cast<const double*>(c->slots + SLOT_BYTES * r);
    for (size_t i = 0; i < count; ++i) sum[i] += s[i];
  }
  if (!barrier(c)) return ncclSystemError;
  if (hipMemcpy(recv, sum.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t dt, ncclComm_t comm,
                           hipStream_t stream) {
  FakeComm* c = reinterpret_cast<FakeComm*>(comm);
  if (dt != ncclDouble || sendcount * 8 > SLOT_BYTES) return ncclInvalidArgument;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(c->slots + SLOT_BYTES * c->rank, send, sendcount * 8, hipMemcpyDeviceToHost) != hipSuccess)
    return ncclUnhandledCudaError;
  if (!barrier(c)) return ncclSystemError;
  std::vector<double> all(sendcount * c->nranks);
  for (int r = 0; r < c->nranks; ++r) std::memcpy(&all[sendcount * r], c->slots + SLOT_BYTES * r, sendcount * 8);
  if (!barrier(c)) return ncclSystemError;
  if (hipMemcpy(recv, all.data(), all.size() * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "shm_collectives error"; }

}  // extern "C"
