// fake_rccl.cc — CPU-TEST-ONLY stand-in for librccl: the nine entry points libheifhip's grid_rccl.hip resolves with dlsym, implemented between PROCESSES of
// one host through files in a directory the unique id names.  It exists so that the SPMD grid path (hipdec_grid_*_rccl: one process per GPU, grouped
// ncclSend / ncclRecv gather, geometry and status all-reduces) can run with more than one rank on the CPU, against the emulated library
// (tests/emu/libheifhip_emu.so, HIPDEC_RCCL_LIBRARY=<this>): the host logic on both sides of the exchange executes for real, the transport is a toy.
// Synchronous like the emulated runtime: a send is complete when its file is in place, a receive when it has read it.  NOT part of the product.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>
#include <thread>
#include <atomic>
#include <unistd.h>
#include <sys/stat.h>

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclSystemError = 2, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
struct FakeComm {
  std::string dir;
  int rank = 0, nranks = 1;
  std::vector<uint64_t> sent, received;   // messages so far to / from every peer
  uint64_t reductions = 0;
};
typedef FakeComm* ncclComm_t;
typedef void* hipStream_t;
}

namespace {
struct Op { bool send; void* buf; size_t bytes; int peer; FakeComm* comm; };
thread_local int g_group = 0;
thread_local std::vector<Op> g_ops;
std::atomic<unsigned> g_ids{0};

size_t dtype_bytes(ncclDataType_t t) { return t <= ncclUint8 ? 1 : (t <= ncclUint32 ? 4 : 8); }

bool write_file(const std::string& path, const void* p, size_t n)
{
  const std::string tmp = path + ".part";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = fwrite(p, 1, n, f) == n || n == 0;
  fclose(f);
  return ok && rename(tmp.c_str(), path.c_str()) == 0;
}
bool read_file(const std::string& path, void* p, size_t n, bool remove_after)
{
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(120);
  for (;;) {
    FILE* f = fopen(path.c_str(), "rb");
    if (f) {
      const bool ok = fread(p, 1, n, f) == n || n == 0;
      fclose(f);
      if (remove_after) unlink(path.c_str());
      return ok;
    }
    if (std::chrono::steady_clock::now() > deadline) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
}
ncclResult_t run(const Op& o)
{
  FakeComm* c = o.comm;
  char name[96];
  if (o.send) {
    snprintf(name, sizeof name, "/msg_%d_%d_%llu", c->rank, o.peer, (unsigned long long)c->sent[(size_t)o.peer]++);
    return write_file(c->dir + name, o.buf, o.bytes) ? ncclSuccess : ncclSystemError;
  }
  snprintf(name, sizeof name, "/msg_%d_%d_%llu", o.peer, c->rank, (unsigned long long)c->received[(size_t)o.peer]++);
  return read_file(c->dir + name, o.buf, o.bytes, true) ? ncclSuccess : ncclSystemError;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
  memset(id, 0, sizeof(*id));
  const char* base = getenv("TMPDIR");
  snprintf(id->internal, sizeof(id->internal), "%s/fake_rccl_%d_%u", base && *base ? base : "/tmp", (int)getpid(), g_ids++);
  return mkdir(id->internal, 0700) == 0 ? ncclSuccess : ncclSystemError;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
  if (nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  FakeComm* c = new FakeComm();
  id.internal[sizeof(id.internal) - 1] = 0;
  c->dir = id.internal; c->rank = rank; c->nranks = nranks;
  c->sent.assign((size_t)nranks, 0); c->received.assign((size_t)nranks, 0);
  *comm = c;
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return ncclSuccess; }
ncclResult_t ncclGroupStart() { g_group++; return ncclSuccess; }
ncclResult_t ncclGroupEnd()
{
  if (--g_group > 0) return ncclSuccess;
  ncclResult_t rc = ncclSuccess;
  for (const Op& o : g_ops) if (o.send) { const ncclResult_t r = run(o); if (r) rc = r; }       // every send first: nobody waits for a receiver
  for (const Op& o : g_ops) if (!o.send) { const ncclResult_t r = run(o); if (r) rc = r; }
  g_ops.clear();
  return rc;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t)
{
  if (peer < 0 || peer >= comm->nranks) return ncclInvalidArgument;
  const Op o{true, (void*)buf, count * dtype_bytes(t), peer, comm};
  if (g_group > 0) { g_ops.push_back(o); return ncclSuccess; }
  return run(o);
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t)
{
  if (peer < 0 || peer >= comm->nranks) return ncclInvalidArgument;
  const Op o{false, buf, count * dtype_bytes(t), peer, comm};
  if (g_group > 0) { g_ops.push_back(o); return ncclSuccess; }
  return run(o);
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t)
{
  if (t != ncclInt64 && t != ncclUint64) return ncclInvalidArgument;      // (all grid_rccl.hip uses)
  const uint64_t seq = c->reductions++;
  char name[96];
  snprintf(name, sizeof name, "/red_%llu_%d", (unsigned long long)seq, c->rank);
  if (!write_file(c->dir + name, send, count * 8)) return ncclSystemError;
  std::vector<int64_t> acc(count), in(count);
  for (int r = 0; r < c->nranks; r++) {
    snprintf(name, sizeof name, "/red_%llu_%d", (unsigned long long)seq, r);
    if (!read_file(c->dir + name, in.data(), count * 8, false)) return ncclSystemError;
    for (size_t i = 0; i < count; i++) {
      if (r == 0) acc[i] = in[i];
      else if (op == ncclSum) acc[i] += in[i];
      else if (op == ncclProd) acc[i] *= in[i];
      else if (op == ncclMax) acc[i] = acc[i] > in[i] ? acc[i] : in[i];
      else acc[i] = acc[i] < in[i] ? acc[i] : in[i];
    }
  }
  memcpy(recv, acc.data(), count * 8);
  return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : (r == ncclInvalidArgument ? "invalid argument (fake rccl)" : "system error (fake rccl)"); }

}  // extern "C"
