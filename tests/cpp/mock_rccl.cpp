// mock_rccl.cpp — TEST INFRASTRUCTURE: a stand-in for librccl.so with the handful of entry points szg_comm.cpp binds,
// so that N ranks can rehearse the REAL C-ABI collective path (szg_rowtile_comm_*, szg_rowtile_gather,
// szg_skyview_allgather_lut_rows with its status-word exchange) on a box with ONE GPU, where RCCL itself refuses to put two
// ranks on a device. Loaded only when a test sets SZG_RCCL_LIBRARY to it; never part of the product.
//
// Transport: every collective is host-staged through a file in /tmp that all ranks map (device -> mapped file -> device),
// between two process-shared barriers. Synchronous and slow, but it keeps RCCL's contracts that matter to the callers:
// in-place all-gather (rank r's chunk already sits at r * bytes of the receive buffer), gather into the root only, stream
// order with the kernels around it (the stream is drained before the copy out and the copy back is enqueued on it).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

namespace
{
constexpr size_t HEADER_BYTES = 4096;
constexpr size_t SLOT_BYTES = (size_t)160 << 20; // per rank: an 8K RGBA16 half frame is 133 MB

struct Header
{
    std::atomic<unsigned> arrived;
    std::atomic<unsigned> generation;
};

struct MockComm
{
    int rank = 0, nranks = 1;
    int fd = -1;
    unsigned char* map = nullptr;
    size_t bytes = 0;
    char path[128] = "";
};

Header* header(MockComm* c) { return reinterpret_cast<Header*>(c->map); }
unsigned char* slot(MockComm* c, int r) { return c->map + HEADER_BYTES + (size_t)r * SLOT_BYTES; }

void barrier(MockComm* c)
{
    Header* h = header(c);
    unsigned const gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1u, std::memory_order_acq_rel) + 1u == (unsigned)c->nranks)
    {
        h->arrived.store(0u, std::memory_order_relaxed);
        h->generation.fetch_add(1u, std::memory_order_release);
        return;
    }
    while (h->generation.load(std::memory_order_acquire) == gen)
    {
        usleep(50);
    }
}
} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    static std::atomic<unsigned> counter{0};
    std::memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/tmp/szg_mock_rccl_%d_%u", (int)getpid(), counter.fetch_add(1u));
    // rank 0 creates the file now, sized and zeroed, so that every rank finds a valid header
    int const fd = open(id->internal, O_CREAT | O_RDWR | O_TRUNC, 0600);
    if (fd < 0)
    {
        return ncclSystemError;
    }
    if (ftruncate(fd, (off_t)HEADER_BYTES) != 0)
    {
        close(fd);
        return ncclSystemError;
    }
    close(fd);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    MockComm* c = new (std::nothrow) MockComm();
    if (c == nullptr)
    {
        return ncclSystemError;
    }
    c->rank = rank;
    c->nranks = nranks;
    snprintf(c->path, sizeof c->path, "%s", id.internal);
    c->bytes = HEADER_BYTES + (size_t)nranks * SLOT_BYTES;
    c->fd = open(c->path, O_RDWR);
    if (c->fd < 0 || (rank == 0 && ftruncate(c->fd, (off_t)c->bytes) != 0))
    {
        delete c;
        return ncclSystemError;
    }
    // the other ranks wait until rank 0 has grown the file (sparse: pages appear when touched)
    for (int spin = 0; spin < 200000; spin++)
    {
        struct stat st;
        if (fstat(c->fd, &st) == 0 && (size_t)st.st_size >= c->bytes)
        {
            break;
        }
        usleep(100);
    }
    c->map = static_cast<unsigned char*>(mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0));
    if (c->map == MAP_FAILED)
    {
        delete c;
        return ncclSystemError;
    }
    *comm = reinterpret_cast<ncclComm_t>(c);
    barrier(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    MockComm* c = reinterpret_cast<MockComm*>(comm);
    if (c == nullptr)
    {
        return ncclSuccess;
    }
    if (c->map != nullptr)
    {
        munmap(c->map, c->bytes);
    }
    if (c->fd >= 0)
    {
        close(c->fd);
    }
    if (c->rank == 0)
    {
        unlink(c->path);
    }
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count)
{
    *count = reinterpret_cast<MockComm*>(comm)->nranks;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream)
{
    MockComm* c = reinterpret_cast<MockComm*>(comm);
    if (datatype != ncclUint8 || sendcount > SLOT_BYTES)
    {
        return ncclInvalidArgument;
    }
    if (hipStreamSynchronize(stream) != hipSuccess ||
        hipMemcpy(slot(c, c->rank), sendbuff, sendcount, hipMemcpyDeviceToHost) != hipSuccess)
    {
        return ncclUnhandledCudaError;
    }
    barrier(c);
    for (int r = 0; r < c->nranks; r++)
    {
        char* const dst = static_cast<char*>(recvbuff) + (size_t)r * sendcount;
        if (r == c->rank && dst == sendbuff)
        {
            continue; // in place
        }
        if (hipMemcpy(dst, slot(c, r), sendcount, hipMemcpyHostToDevice) != hipSuccess)
        {
            return ncclUnhandledCudaError;
        }
    }
    barrier(c);
    return ncclSuccess;
}

ncclResult_t ncclGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, int root, ncclComm_t comm,
                        hipStream_t stream)
{
    MockComm* c = reinterpret_cast<MockComm*>(comm);
    if (datatype != ncclUint8 || sendcount > SLOT_BYTES)
    {
        return ncclInvalidArgument;
    }
    if (hipStreamSynchronize(stream) != hipSuccess ||
        hipMemcpy(slot(c, c->rank), sendbuff, sendcount, hipMemcpyDeviceToHost) != hipSuccess)
    {
        return ncclUnhandledCudaError;
    }
    barrier(c);
    if (c->rank == root)
    {
        for (int r = 0; r < c->nranks; r++)
        {
            if (hipMemcpy(static_cast<char*>(recvbuff) + (size_t)r * sendcount, slot(c, r), sendcount, hipMemcpyHostToDevice) != hipSuccess)
            {
                return ncclUnhandledCudaError;
            }
        }
    }
    barrier(c);
    return ncclSuccess;
}

// Grouped point-to-point calls (szg_rowtile_gather's path when the library has no ncclGather, or SZG_RCCL_NO_GATHER is set):
// like RCCL, Send / Recv inside a group only RECORD the operation; ncclGroupEnd carries the whole group out. Every rank of the
// communicator takes part in every group (true for the gather: the root receives from all, the others send to it): senders
// stage their bytes in their slot, a barrier, receivers copy from the peers' slots, a barrier. A call outside a group is a
// group of one. An operation on a peer outside the communicator fails at the call, inside the open group - which the caller
// must still close.
namespace
{
struct Pending
{
    bool send;
    const void* src;
    void* dst;
    size_t bytes;
    int peer;
    MockComm* comm;
    hipStream_t stream;
};
thread_local int g_depth = 0;
thread_local Pending g_ops[64];
thread_local int g_nops = 0;

ncclResult_t flush()
{
    ncclResult_t result = ncclSuccess;
    MockComm* c = g_nops > 0 ? g_ops[0].comm : nullptr;
    for (int i = 0; i < g_nops; i++)
    {
        const Pending& o = g_ops[i];
        if (o.send && (hipStreamSynchronize(o.stream) != hipSuccess ||
                       hipMemcpy(slot(o.comm, o.comm->rank), o.src, o.bytes, hipMemcpyDeviceToHost) != hipSuccess))
        {
            result = ncclUnhandledCudaError;
        }
    }
    if (c != nullptr)
    {
        barrier(c);
    }
    for (int i = 0; i < g_nops; i++)
    {
        const Pending& o = g_ops[i];
        if (!o.send && (hipStreamSynchronize(o.stream) != hipSuccess || hipMemcpy(o.dst, slot(o.comm, o.peer), o.bytes, hipMemcpyHostToDevice) != hipSuccess))
        {
            result = ncclUnhandledCudaError;
        }
    }
    if (c != nullptr)
    {
        barrier(c);
    }
    g_nops = 0;
    return result;
}

ncclResult_t record(bool send, const void* src, void* dst, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    MockComm* c = reinterpret_cast<MockComm*>(comm);
    if (datatype != ncclUint8 || count > SLOT_BYTES || peer < 0 || peer >= c->nranks || peer == c->rank || g_nops >= 64)
    {
        return ncclInvalidArgument;
    }
    g_ops[g_nops++] = Pending{send, src, dst, count, peer, c, stream};
    return g_depth == 0 ? flush() : ncclSuccess;
}
} // namespace

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    return record(true, sendbuff, nullptr, count, datatype, peer, comm, stream);
}
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    return record(false, nullptr, recvbuff, count, datatype, peer, comm, stream);
}
ncclResult_t ncclGroupStart()
{
    g_depth++;
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0)
    {
        return ncclInvalidUsage;
    }
    return --g_depth == 0 ? flush() : ncclSuccess;
}
ncclResult_t ncclGetVersion(int* version)
{
    *version = -1; // not a real RCCL
    return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t e) { return e == ncclSuccess ? "success" : "mock RCCL error"; }

} // extern "C"
