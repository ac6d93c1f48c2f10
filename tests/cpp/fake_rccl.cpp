// fake_rccl.cpp -- TEST DOUBLE for librccl, used by tests/test_gpu_multi.py only (never by the product: libis3d_amd.so loads it only when
// the test sets IS3D_RCCL_LIBRARY).  It lets several PROCESSES that share one GPU run the library's multi-rank control flow
// (is3d_comm_create / is3d_plan_execute_allreduce / is3d_comm_check / is3d_comm_abort) for real -- something RCCL itself refuses on one
// device ("duplicate GPU") -- so that the error word, the zero contribution of a failed rank and the abort path are exercised with two
// ranks.  What it is NOT: a test of RCCL, of xGMI or of performance.
//
// The nine entry points cf_multi.hip binds.  A communicator is a POSIX shared-memory segment named after the unique id: a generation-counting
// barrier, an abort flag, and one slot of doubles per rank.  ncclAllReduce(sum, double) synchronises the stream, copies the device buffer into
// the rank's slot, meets the other ranks, sums the slots in rank order, copies the sum back and meets again; a rank that finds the abort flag
// while waiting returns ncclSystemError.  THAT IS A PROPERTY OF THIS DOUBLE, NOT OF RCCL: a real ncclAllReduce is enqueued and returns
// ncclSuccess at once, and a peer's local ncclCommAbort does not unblock it -- the library's protection against a rank that never joins is
// the deadline of its own host-side waits (cf_multi.hip::comm_wait), which this double cannot exercise because its all-reduce is host-synchronous.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace {
constexpr size_t kSlotDoubles = (size_t)5 << 20;   // 42 MB per rank (pages are committed when touched): the config-3 spectrum is 4.92e6 doubles
constexpr int kMaxRanks = 8;
struct Shared {
    std::atomic<int> arrived, generation, aborted, attached;
    double slot[kMaxRanks][kSlotDoubles];
};
struct Fake {
    Shared *sh = nullptr;
    int n = 1, rank = 0;
    std::string name;
};
struct Pending { Fake *c; void *buf; size_t count; hipStream_t st; };
thread_local std::vector<Pending> g_group;
thread_local int g_depth = 0;

bool barrier(Fake *c)   // false: aborted (or a peer vanished for 60 s)
{
    Shared *s = c->sh;
    const int gen = s->generation.load();
    if (s->arrived.fetch_add(1) + 1 == c->n) {
        s->arrived.store(0);
        s->generation.fetch_add(1);
        return !s->aborted.load();
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (s->generation.load() == gen) {
        if (s->aborted.load()) return false;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    return !s->aborted.load();
}

ncclResult_t run(const Pending &p)
{
    Fake *c = p.c;
    if (p.count > kSlotDoubles) return ncclInvalidArgument;
    if (hipStreamSynchronize(p.st) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->sh->slot[c->rank], p.buf, p.count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    std::vector<double> sum(p.count, 0.0);
    for (int r = 0; r < c->n; r++)
        for (size_t i = 0; i < p.count; i++) sum[i] += c->sh->slot[r][i];
    if (!barrier(c)) return ncclSystemError;          // everybody has read the slots: they may be overwritten by the next collective
    if (hipMemcpy(p.buf, sum.data(), p.count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof *id);
    std::random_device rd;
    snprintf(id->internal, sizeof id->internal, "/is3d_fake_rccl_%08x%08x", rd(), rd());
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Fake *c = new Fake;
    c->n = nranks; c->rank = rank; c->name = id.internal;
    int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) { delete c; return ncclSystemError; }
    c->sh = (Shared *)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->sh == MAP_FAILED) { delete c; return ncclSystemError; }
    c->sh->attached.fetch_add(1);                      // a fresh segment is zero-filled: the atomics start at 0
    const auto t0 = std::chrono::steady_clock::now();
    while (c->sh->attached.load() < nranks) {          // the real call is collective too
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return ncclSystemError;
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t *, int, const int *) { return ncclInvalidUsage; }   // one process, several devices: not what this double is for
ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Fake *c = (Fake *)comm;
    if (!c) return ncclSuccess;
    if (c->rank == 0) shm_unlink(c->name.c_str());
    munmap(c->sh, sizeof(Shared));
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t comm)
{
    Fake *c = (Fake *)comm;
    if (!c) return ncclSuccess;
    c->sh->aborted.store(1);
    if (c->rank == 0) shm_unlink(c->name.c_str());
    munmap(c->sh, sizeof(Shared));
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd()
{
    if (--g_depth > 0) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    for (const Pending &p : g_group)
        if (rc == ncclSuccess) rc = run(p);
    g_group.clear();
    return rc;
}
ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
    if (sendbuff != recvbuff || datatype != ncclDouble || op != ncclSum || !comm) return ncclInvalidArgument;
    Pending p{(Fake *)comm, recvbuff, count, stream};
    if (g_depth > 0) { g_group.push_back(p); return ncclSuccess; }
    return run(p);
}
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclSystemError ? "fake rccl: aborted or a rank vanished" : "fake rccl: error"; }
}
