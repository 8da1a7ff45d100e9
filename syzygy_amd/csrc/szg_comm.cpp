// szg_comm.cpp — the two collectives of the row-tiled multi-GPU frame behind the C-ABI (szg/abi.h "Multi-GPU
// collectives"): one gather of the RGBA16 row tiles to the root, and the all-gather of the sky-view LUT row slices.
// No reference counterpart (the reference is single GPU); BASELINE north_star: "C++ host code ... a
// single RCCL gather over xGMI for the composed image".
//
// RCCL is bound at run time (dlopen of librccl.so.1, the SONAME of both the ROCm library and the copy PyTorch carries:
// inside a PyTorch process this resolves to the instance that is already loaded), so that libszg_hip.so itself has no
// link-time dependency on it: single-GPU callers and the CPU-only ABI tests never touch RCCL.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types and prototypes only; every call goes through the table below

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

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
    char why[256] = "";
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

const Rccl& rccl()
{
    std::call_once(g_rcclOnce, [] {
        Rccl& r = g_rccl;
        const char* const override_path = getenv("SZG_RCCL_LIBRARY");
        const char* const names[] = {override_path, "librccl.so.1", "librccl.so"};
        for (const char* name : names)
        {
            if (name != nullptr && name[0] != '\0' && r.handle == nullptr)
            {
                r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
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
               bind(r.handle, "ncclGather", r.Gather, false, r.why, sizeof r.why);
    });
    return g_rccl;
}

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

int szg_rowtile_comm_create(szg_rowtile_comm_t** out, int rank, int nranks, const void* unique_id, int device)
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
    if (hipSetDevice(device) != hipSuccess)
    {
        return fail(SZG_ERR_HIP, "szg_rowtile_comm_create: hipSetDevice failed");
    }
    szg_rowtile_comm* c = new (std::nothrow) szg_rowtile_comm();
    if (c == nullptr)
    {
        return fail(SZG_ERR_OUT_OF_MEMORY, "szg_rowtile_comm_create: host allocation failed");
    }
    c->device = device;
    c->rank = rank;
    c->nranks = nranks;
    ncclUniqueId ids[2];
    std::memcpy(ids, unique_id, sizeof ids);
    ncclResult_t e = r.CommInitRank(&c->lut, nranks, ids[0], rank);
    if (e == ncclSuccess)
    {
        e = r.CommInitRank(&c->tiles, nranks, ids[1], rank);
    }
    if (e != ncclSuccess)
    {
        szg_rowtile_comm_destroy(c);
        return fail_nccl(r, e, "szg_rowtile_comm_create: ncclCommInitRank");
    }
    *out = c;
    return SZG_OK;
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
    hipStream_t const s = static_cast<hipStream_t>(stream);
    if (r.Gather != nullptr)
    {
        SZG_NCCL(r.Gather(tile, gathered, tile_bytes, ncclUint8, root, c->tiles, s));
        return SZG_OK;
    }
    // N - 1 point-to-point streams into the root, one per xGMI link
    SZG_NCCL(r.GroupStart());
    if (c->rank == root)
    {
        for (int peer = 0; peer < c->nranks; peer++)
        {
            char* const slot = static_cast<char*>(gathered) + (size_t)peer * tile_bytes;
            if (peer != root)
            {
                SZG_NCCL(r.Recv(slot, tile_bytes, ncclUint8, peer, c->tiles, s));
            }
            else if (slot != tile && hipMemcpyAsync(slot, tile, tile_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess)
            {
                (void)r.GroupEnd();
                return fail(SZG_ERR_HIP, "szg_rowtile_gather: copy of the root's own tile failed");
            }
        }
    }
    else
    {
        SZG_NCCL(r.Send(tile, tile_bytes, ncclUint8, root, c->tiles, s));
    }
    SZG_NCCL(r.GroupEnd());
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
    // in place: rank r's contribution already sits at r * bytes_per_rank of the receive buffer
    const char* const mine = static_cast<const char*>(buffer) + (size_t)c->rank * bytes_per_rank;
    SZG_NCCL(r.AllGather(mine, buffer, bytes_per_rank, ncclUint8, c->lut, static_cast<hipStream_t>(stream)));
    return SZG_OK;
}

} // extern "C"
