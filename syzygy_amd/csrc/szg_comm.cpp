// szg_comm.cpp — the two collectives of the row-tiled multi-GPU frame behind the C-ABI (szg/abi.h "Multi-GPU
// collectives"): one gather of the RGBA16 row tiles to the root, and the all-gather of the sky-view LUT row slices.
// No reference counterpart (the reference is single GPU); BASELINE north_star: "C++ host code ... a
// single RCCL gather over xGMI for the composed image".
//
// RCCL is bound at run time, so that libszg_hip.so itself has no link-time dependency on it: single-GPU callers and the
// CPU-only ABI tests never touch RCCL. Binding order (round 3; two RCCLs in one process - PyTorch carries its own copy next
// to ROCm's - is the first thing that goes wrong on a real node): (1) SZG_RCCL_LIBRARY if set; (2) an RCCL that is ALREADY
// MAPPED into the process, found by walking the loaded objects (dl_iterate_phdr) and re-opened by its own path with
// RTLD_NOLOAD - inside a PyTorch process that is the instance torch.distributed uses; (3) librccl.so.1 / librccl.so by
// name. szg_rowtile_comm_backend() reports which object was bound and the version it reports.
#include <dlfcn.h>
#include <link.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types and prototypes only; every call goes through the table below

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

#include "szg/abi.h"
#include "szg_internal.hpp"

namespace
{
struct Rccl
{
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGather) Gather = nullptr; // optional: grouped send / recv otherwise
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr; // optional
    char why[256] = "";
    char info[512] = ""; // "<path> (RCCL <version>; <how it was found>)"
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rcclOnce;

template <typename F> bool bind(void* handle, const char* name, F& out, bool required, char* why, size_t whyBytes)
{
    out = reinterpret_cast<F>(dlsym(handle, name));
    if (out == nullptr && required)
    {
        snprintf(why, whyBytes, "librccl has no symbol %s", name);
        return false;
    }
    return true;
}

// An RCCL that is already mapped into this process (an object whose file name starts with "librccl.so"): its path, or "".
int find_mapped_rccl(struct dl_phdr_info* info, size_t, void* out)
{
    const char* const path = info->dlpi_name;
    if (path == nullptr || path[0] == '\0')
    {
        return 0;
    }
    const char* const slash = strrchr(path, '/');
    const char* const base = slash != nullptr ? slash + 1 : path;
    if (strncmp(base, "librccl.so", 10) == 0) // librccl.so, librccl.so.1, librccl.so.1.0.x - not librccl-net.so and the like
    {
        snprintf(static_cast<char*>(out), 400, "%s", path);
        return 1;
    }
    return 0;
}

const Rccl& rccl()
{
    std::call_once(g_rcclOnce, [] {
        Rccl& r = g_rccl;
        const char* how = "";
        char bound[400] = "";
        const char* const override_path = getenv("SZG_RCCL_LIBRARY");
        if (override_path != nullptr && override_path[0] != '\0')
        {
            r.handle = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
            how = "SZG_RCCL_LIBRARY";
            snprintf(bound, sizeof bound, "%s", override_path);
            if (r.handle == nullptr)
            {
                snprintf(r.why, sizeof r.why, "SZG_RCCL_LIBRARY=%s cannot be loaded: %s", override_path, dlerror());
                return; // an explicit choice that fails is an error, not a reason to bind some other RCCL silently
            }
        }
        if (r.handle == nullptr)
        {
            char mapped[400] = "";
            if (dl_iterate_phdr(find_mapped_rccl, mapped) != 0 && mapped[0] != '\0')
            {
                r.handle = dlopen(mapped, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
                if (r.handle != nullptr)
                {
                    how = "already mapped in this process";
                    snprintf(bound, sizeof bound, "%s", mapped);
                }
            }
        }
        const char* const names[] = {"librccl.so.1", "librccl.so"};
        for (const char* name : names)
        {
            if (r.handle == nullptr)
            {
                r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (r.handle != nullptr)
                {
                    how = "loaded by name";
                    snprintf(bound, sizeof bound, "%s", name);
                }
            }
        }
        if (r.handle == nullptr)
        {
            snprintf(r.why, sizeof r.why, "RCCL is not available: %s", dlerror());
            return;
        }
        r.ok = bind(r.handle, "ncclGetUniqueId", r.GetUniqueId, true, r.why, sizeof r.why) &&
               bind(r.handle, "ncclCommInitRank", r.CommInitRank, true, r.why, sizeof r.why) &&
               bind(r.handle, "ncclCommDestroy", r.CommDestroy, true, r.why, sizeof r.why) &&
               bind(r.handle, "ncclCommCount", r.CommCount, true, r.why, sizeof r.why) &&
               bind(r.handle, "ncclAllGather", r.AllGather, true, r.why, sizeof r.why) &&
               bind(r.handle, "ncclSend", r.Send, true, r.why, sizeof r.why) &&
               bind(r.handle, "ncclRecv", r.Recv, true, r.why, sizeof r.why) &&
               bind(r.handle, "ncclGroupStart", r.GroupStart, true, r.why, sizeof r.why) &&
               bind(r.handle, "ncclGroupEnd", r.GroupEnd, true, r.why, sizeof r.why) &&
               bind(r.handle, "ncclGetErrorString", r.GetErrorString, true, r.why, sizeof r.why) &&
               bind(r.handle, "ncclGather", r.Gather, false, r.why, sizeof r.why) &&
               bind(r.handle, "ncclGetVersion", r.GetVersion, false, r.why, sizeof r.why);
        if (getenv("SZG_RCCL_NO_GATHER") != nullptr) // tests: take the grouped send / recv path although ncclGather exists
        {
            r.Gather = nullptr;
        }
        // the object the symbols really came from (dladdr), which is what matters when two RCCLs are around
        Dl_info where;
        if (r.GetUniqueId != nullptr && dladdr(reinterpret_cast<void*>(r.GetUniqueId), &where) != 0 && where.dli_fname != nullptr)
        {
            snprintf(bound, sizeof bound, "%s", where.dli_fname);
        }
        int version = 0;
        if (r.GetVersion != nullptr)
        {
            (void)r.GetVersion(&version);
        }
        snprintf(r.info, sizeof r.info, "%s (RCCL version code %d; %s; gather: %s)", bound, version, how,
                 r.Gather != nullptr ? "ncclGather" : "grouped ncclSend/ncclRecv");
        if (getenv("SZG_LOG") != nullptr)
        {
            fprintf(stderr, "[szg] collectives bound to %s\n", r.info);
        }
    });
    return g_rccl;
}

// DeviceGuard of szg_api.cpp: a communicator lives on the device it was created on
struct DeviceGuard
{
    int previous = -1;
    bool switched = false;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&previous) == hipSuccess && previous != device)
        {
            switched = hipSetDevice(device) == hipSuccess;
        }
    }
    ~DeviceGuard()
    {
        if (switched)
        {
            (void)hipSetDevice(previous);
        }
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

int fail(int code, const char* fmt, const char* a = "", const char* b = "")
{
    char text[512];
    snprintf(text, sizeof text, fmt, a, b);
    szg::set_last_error(text);
    return code;
}

int fail_nccl(const Rccl& r, ncclResult_t e, const char* what)
{
    return fail(SZG_ERR_HIP, "%s: %s", what, r.GetErrorString != nullptr ? r.GetErrorString(e) : "RCCL error");
}
#define SZG_NCCL(expr)                                                                                                         \
    do                                                                                                                         \
    {                                                                                                                          \
        ncclResult_t const _e = (expr);                                                                                        \
        if (_e != ncclSuccess)                                                                                                 \
        {                                                                                                                      \
            return fail_nccl(r, _e, #expr);                                                                                    \
        }                                                                                                                      \
    } while (0)
} // namespace

// Two communicators, one per collective: the LUT all-gather of frame k and the tile gather of frame k-1 are in flight at
// the same time on different streams, and one communicator serialises its operations.
struct szg_rowtile_comm
{
    int device = 0;
    int rank = 0, nranks = 1;
    ncclComm_t lut = nullptr;
    ncclComm_t tiles = nullptr;
};

extern "C" {

int szg_rowtile_comm_unique_id(void* out_id)
{
    if (out_id == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_rowtile_comm_unique_id: out_id is NULL");
    }
    const Rccl& r = rccl();
    if (!r.ok)
    {
        return fail(SZG_ERR_NO_DEVICE, "%s", r.why);
    }
    static_assert(SZG_ROWTILE_COMM_ID_BYTES == 2 * NCCL_UNIQUE_ID_BYTES, "two RCCL unique ids");
    ncclUniqueId ids[2];
    SZG_NCCL(r.GetUniqueId(&ids[0]));
    SZG_NCCL(r.GetUniqueId(&ids[1]));
    std::memcpy(out_id, ids, sizeof ids);
    return SZG_OK;
}

const char* szg_rowtile_comm_backend(void)
{
    const Rccl& r = rccl();
    return r.ok ? r.info : r.why;
}

// ncclCommInitRank blocks until every rank has joined; a rank that never arrives (a process that died at start-up, a wrong
// rank count) would hang the others for ever. The two initialisations therefore run on a helper thread and the caller waits
// for it with a deadline. On a timeout the thread cannot be cancelled (RCCL offers no cancellation for the blocking call): it
// is detached, keeps its state alive through a shared_ptr, and the CALLER IS EXPECTED TO EXIT with a non-zero status - which
// is what makes a launcher (torch.distributed.run, mpirun) tear the other ranks down. Never re-exec a process that has
// touched the GPU.
int szg_rowtile_comm_create_deadline(szg_rowtile_comm_t** out, int rank, int nranks, const void* unique_id, int device, int timeout_ms)
{
    if (out == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_rowtile_comm_create: out is NULL");
    }
    *out = nullptr;
    if (unique_id == nullptr || nranks < 1 || rank < 0 || rank >= nranks)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_rowtile_comm_create: NULL id or rank outside [0, nranks)");
    }
    const Rccl& r = rccl();
    if (!r.ok)
    {
        return fail(SZG_ERR_NO_DEVICE, "%s", r.why);
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count)
    {
        return fail(SZG_ERR_NO_DEVICE, "szg_rowtile_comm_create: no such HIP device (one process per GPU; no CPU fallback)");
    }
    struct Init
    {
        std::mutex m;
        std::condition_variable cv;
        bool done = false;
        ncclResult_t result = ncclSuccess;
        ncclComm_t lut = nullptr, tiles = nullptr;
        ncclUniqueId ids[2];
    };
    // (nothing may throw across the C boundary: allocation of the shared state and creation of the thread are guarded)
    std::shared_ptr<Init> st;
    std::thread worker;
    try
    {
        st = std::make_shared<Init>();
    }
    catch (...)
    {
        return fail(SZG_ERR_OUT_OF_MEMORY, "szg_rowtile_comm_create: host allocation failed");
    }
    std::memcpy(st->ids, unique_id, sizeof st->ids);
    const Rccl* const rp = &r;
    try
    {
        worker = std::thread([st, rp, rank, nranks, device] {
            ncclResult_t e = ncclUnhandledCudaError;
            if (hipSetDevice(device) == hipSuccess)
            {
                e = rp->CommInitRank(&st->lut, nranks, st->ids[0], rank);
                if (e == ncclSuccess)
                {
                    e = rp->CommInitRank(&st->tiles, nranks, st->ids[1], rank);
                }
            }
            std::lock_guard<std::mutex> lock(st->m);
            st->result = e;
            st->done = true;
            st->cv.notify_all();
        });
    }
    catch (...)
    {
        return fail(SZG_ERR_HIP, "szg_rowtile_comm_create: the helper thread of the rendezvous could not be started");
    }
    bool finished;
    {
        std::unique_lock<std::mutex> lock(st->m);
        if (timeout_ms > 0)
        {
            finished = st->cv.wait_for(lock, std::chrono::milliseconds(timeout_ms), [&] { return st->done; });
        }
        else
        {
            st->cv.wait(lock, [&] { return st->done; });
            finished = true;
        }
    }
    if (!finished)
    {
        worker.detach();
        char text[256];
        snprintf(text, sizeof text,
                 "szg_rowtile_comm_create: rank %d of %d: the other ranks did not join within %d ms (ncclCommInitRank still blocked); "
                 "exit this process with a non-zero status",
                 rank, nranks, timeout_ms);
        szg::set_last_error(text);
        return SZG_ERR_TIMEOUT;
    }
    worker.join();
    szg_rowtile_comm* c = new (std::nothrow) szg_rowtile_comm();
    if (c == nullptr)
    {
        return fail(SZG_ERR_OUT_OF_MEMORY, "szg_rowtile_comm_create: host allocation failed");
    }
    c->device = device;
    c->rank = rank;
    c->nranks = nranks;
    c->lut = st->lut;
    c->tiles = st->tiles;
    if (st->result != ncclSuccess)
    {
        szg_rowtile_comm_destroy(c);
        return fail_nccl(r, st->result, "szg_rowtile_comm_create: ncclCommInitRank");
    }
    *out = c;
    return SZG_OK;
}

int szg_rowtile_comm_create(szg_rowtile_comm_t** out, int rank, int nranks, const void* unique_id, int device)
{
    // SZG_COMM_TIMEOUT_S: deadline of the rendezvous in seconds (default 300; 0 = wait for ever)
    int seconds = 300;
    if (const char* t = getenv("SZG_COMM_TIMEOUT_S"))
    {
        seconds = atoi(t);
    }
    return szg_rowtile_comm_create_deadline(out, rank, nranks, unique_id, device, seconds > 0 ? seconds * 1000 : 0);
}

void szg_rowtile_comm_destroy(szg_rowtile_comm_t* c)
{
    if (c == nullptr)
    {
        return;
    }
    const Rccl& r = rccl();
    if (r.ok)
    {
        if (c->lut != nullptr)
        {
            (void)r.CommDestroy(c->lut);
        }
        if (c->tiles != nullptr)
        {
            (void)r.CommDestroy(c->tiles);
        }
    }
    delete c;
}

int szg_rowtile_comm_rank(const szg_rowtile_comm_t* c) { return c != nullptr ? c->rank : -1; }

int szg_rowtile_comm_size(const szg_rowtile_comm_t* c)
{
    if (c == nullptr)
    {
        return -1;
    }
    // what RCCL itself reports for the communicator, not what the caller said at creation
    const Rccl& r = rccl();
    int n = 0;
    if (!r.ok || r.CommCount(c->tiles, &n) != ncclSuccess)
    {
        return -1;
    }
    return n;
}

int szg_rowtile_gather(szg_rowtile_comm_t* c, void* stream, const void* tile, size_t tile_bytes, void* gathered, int root)
{
    if (c == nullptr || tile == nullptr || root < 0 || root >= c->nranks || (c->rank == root && gathered == nullptr))
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_rowtile_gather: NULL argument or root outside the communicator");
    }
    if (tile_bytes == 0u)
    {
        return SZG_OK;
    }
    const Rccl& r = rccl();
    DeviceGuard const guard(c->device);
    hipStream_t const s = static_cast<hipStream_t>(stream);
    if (r.Gather != nullptr)
    {
        SZG_NCCL(r.Gather(tile, gathered, tile_bytes, ncclUint8, root, c->tiles, s));
        return SZG_OK;
    }
    // N - 1 point-to-point streams into the root, one per xGMI link. The root's own tile is an ordinary copy on the same
    // stream, OUTSIDE the group (a group holds RCCL calls only). Whatever fails inside the group, the group is closed before
    // returning: the first error is kept, ncclGroupEnd is always called.
    if (c->rank == root)
    {
        char* const own = static_cast<char*>(gathered) + (size_t)root * tile_bytes;
        if (own != tile && hipMemcpyAsync(own, tile, tile_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess)
        {
            return fail(SZG_ERR_HIP, "szg_rowtile_gather: copy of the root's own tile failed");
        }
    }
    SZG_NCCL(r.GroupStart());
    ncclResult_t first = ncclSuccess;
    const char* what = "";
    if (c->rank == root)
    {
        for (int peer = 0; peer < c->nranks && first == ncclSuccess; peer++)
        {
            if (peer != root)
            {
                first = r.Recv(static_cast<char*>(gathered) + (size_t)peer * tile_bytes, tile_bytes, ncclUint8, peer, c->tiles, s);
                what = "szg_rowtile_gather: ncclRecv";
            }
        }
    }
    else
    {
        first = r.Send(tile, tile_bytes, ncclUint8, root, c->tiles, s);
        what = "szg_rowtile_gather: ncclSend";
    }
    ncclResult_t const closed = r.GroupEnd();
    if (first != ncclSuccess)
    {
        return fail_nccl(r, first, what);
    }
    if (closed != ncclSuccess)
    {
        return fail_nccl(r, closed, "szg_rowtile_gather: ncclGroupEnd");
    }
    return SZG_OK;
}

int szg_rowtile_allgather(szg_rowtile_comm_t* c, void* stream, void* buffer, size_t bytes_per_rank)
{
    if (c == nullptr || buffer == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_rowtile_allgather: NULL argument");
    }
    if (bytes_per_rank == 0u)
    {
        return SZG_OK;
    }
    const Rccl& r = rccl();
    DeviceGuard const guard(c->device);
    // in place: rank r's contribution already sits at r * bytes_per_rank of the receive buffer
    const char* const mine = static_cast<const char*>(buffer) + (size_t)c->rank * bytes_per_rank;
    SZG_NCCL(r.AllGather(mine, buffer, bytes_per_rank, ncclUint8, c->lut, static_cast<hipStream_t>(stream)));
    return SZG_OK;
}

} // extern "C"
