// sqy_rccl.cpp -- the multi-GPU step of the path for C / C++ callers: z-slabs are encoded independently, one process per GPU
// (SURVEY.md 8(e)); what crosses GPUs is ONE variable-length gather of the compressed slab blobs to a root rank over RCCL / xGMI:
// an all-gather of the 8-byte blob sizes, then grouped point-to-point sends straight into the root's buffer (xGMI is
// point-to-point: seven links into the root, no ring).  The reference has no counterpart (single process, OpenMP only).
//
// RCCL is loaded at first use (dlopen "librccl.so.1": the copy the process already has -- PyTorch-ROCm brings its own under the
// same soname -- or ROCm's), so that single-GPU users of libsqeazy_amd.so need no RCCL at all.
#include "../../include/sqeazy_amd.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() {
        for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) { std::fprintf(stderr, "[sqeazy]\t RCCL (librccl.so.1) not found: %s\n", dlerror()); return; }
#define SQY_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, #sym)); if (!r.field) { std::fprintf(stderr, "[sqeazy]\t RCCL lacks %s\n", #sym); return; }
        SQY_SYM(GetUniqueId, ncclGetUniqueId)
        SQY_SYM(CommInitRank, ncclCommInitRank)
        SQY_SYM(CommDestroy, ncclCommDestroy)
        SQY_SYM(CommCount, ncclCommCount)
        SQY_SYM(CommUserRank, ncclCommUserRank)
        SQY_SYM(AllGather, ncclAllGather)
        SQY_SYM(Send, ncclSend)
        SQY_SYM(Recv, ncclRecv)
        SQY_SYM(GroupStart, ncclGroupStart)
        SQY_SYM(GroupEnd, ncclGroupEnd)
        SQY_SYM(GetErrorString, ncclGetErrorString)
#undef SQY_SYM
        r.ok = true;
    });
    return r;
}

#define SQY_NCCL(call)                                                                                               \
    do {                                                                                                             \
        ncclResult_t r_ = (call);                                                                                    \
        if (r_ != ncclSuccess) {                                                                                     \
            std::fprintf(stderr, "[sqeazy]\t RCCL error %s at %s:%d\n", R.GetErrorString(r_), __FILE__, __LINE__);   \
            return 1;                                                                                                \
        }                                                                                                            \
    } while (0)
#define SQY_HIPC(call)                                                                                               \
    do {                                                                                                             \
        hipError_t e_ = (call);                                                                                      \
        if (e_ != hipSuccess) {                                                                                      \
            std::fprintf(stderr, "[sqeazy]\t HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);   \
            return 1;                                                                                                \
        }                                                                                                            \
    } while (0)

template <class F>
int guarded(F&& f) noexcept
{
    try { return f(); }
    catch (const std::exception& e) { std::fprintf(stderr, "[sqeazy]\t %s\n", e.what()); return 1; }
    catch (...) { std::fprintf(stderr, "[sqeazy]\t unknown exception\n"); return 1; }
}

} // namespace

extern "C" {

int SQYAMD_Comm_UniqueId(char* id128)
{
    return guarded([&]() -> int {
    Rccl& R = rccl();
    if (!R.ok || !id128) return 1;
    static_assert(sizeof(ncclUniqueId) == 128, "the id travels as 128 opaque bytes");
    ncclUniqueId id;
    SQY_NCCL(R.GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return 0;
    });
}

int SQYAMD_Comm_Init(void** comm, int world, int rank, const char* id128)
{
    return guarded([&]() -> int {
    Rccl& R = rccl();
    if (!R.ok || !comm || !id128 || world <= 0 || rank < 0 || rank >= world) return 1;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    SQY_NCCL(R.CommInitRank(&c, world, id, rank));        // (binds to the current HIP device)
    *comm = c;
    return 0;
    });
}

int SQYAMD_Comm_Destroy(void* comm)
{
    return guarded([&]() -> int {
    Rccl& R = rccl();
    if (!R.ok || !comm) return 1;
    SQY_NCCL(R.CommDestroy(static_cast<ncclComm_t>(comm)));
    return 0;
    });
}

int SQYAMD_Gather_Blobs(void* comm_v, int root, const void* d_blob, long nbytes, void* d_recv, long recv_capacity, long* sizes,
                        void* hip_stream)
{
    return guarded([&]() -> int {
    Rccl& R = rccl();
    // Without a communicator this rank cannot take part in anything; every OTHER local failure is carried through the protocol,
    // so that all ranks return 1 together instead of one rank leaving and the others waiting for it until the RCCL time-out
    // (round-3 advice): bad arguments and allocation failures travel as the size -1 in the first all-gather, anything that goes
    // wrong after it as a 0 in the second one, and a group that has been started is always ended.
    if (!R.ok || !comm_v) return 1;
    ncclComm_t comm = static_cast<ncclComm_t>(comm_v);
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    int world = 0, rank = -1;
    SQY_NCCL(R.CommCount(comm, &world));
    SQY_NCCL(R.CommUserRank(comm, &rank));
    bool local_ok = nbytes >= 0 && sizes && !(nbytes > 0 && !d_blob) && root >= 0 && root < world;
    // 1. the sizes: one 8-byte all-gather (device buffers: a small allocation of this call)
    long long* d_sizes = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d_sizes), sizeof(long long) * (size_t)(world + 1)) != hipSuccess) {
        std::fprintf(stderr, "[sqeazy]\t gather: no device memory for the size exchange\n");
        return 1;                                                         // (nothing to send the news with)
    }
    struct Free { long long* p; ~Free() { if (p) (void)hipFree(p); } } free_sizes{d_sizes};
    const long long mine = local_ok ? nbytes : -1;
    SQY_HIPC(hipMemcpyAsync(d_sizes + world, &mine, sizeof(mine), hipMemcpyHostToDevice, stream));
    SQY_NCCL(R.AllGather(d_sizes + world, d_sizes, 1, ncclInt64, comm, stream));
    std::vector<long long> h(world);
    SQY_HIPC(hipMemcpyAsync(h.data(), d_sizes, sizeof(long long) * (size_t)world, hipMemcpyDeviceToHost, stream));
    SQY_HIPC(hipStreamSynchronize(stream));
    uint64_t total = 0;
    bool all_ok = true;
    for (int r = 0; r < world; ++r) {
        if (h[r] < 0) { all_ok = false; continue; }
        if (sizes) sizes[r] = (long)h[r];
        total += (uint64_t)h[r];
    }
    if (!all_ok) {
        if (local_ok) std::fprintf(stderr, "[sqeazy]\t gather: another rank reported a failure\n");
        return 1;                                                         // every rank sees the same sizes: all return 1 here
    }
    const bool root_ok = rank != root || (d_recv && total <= (uint64_t)(recv_capacity < 0 ? 0 : recv_capacity));
    if (!root_ok) std::fprintf(stderr, "[sqeazy]\t gather: %llu bytes do not fit the root's buffer\n", (unsigned long long)total);
    // 2. the blobs: grouped point-to-point transfers into the root's buffer, rank order; the root's own blob is a device copy.
    //    Whether everybody is still fit for it -- a root without room receives into nothing -- is settled by a second all-gather
    //    BEFORE anybody posts a send (no rank may be left waiting in a send that is never matched).
    long long* d_flag = d_sizes;                                          // reuse: world + 1 words
    const long long okflag = root_ok ? 1 : 0;
    SQY_HIPC(hipMemcpyAsync(d_flag + world, &okflag, sizeof(okflag), hipMemcpyHostToDevice, stream));
    SQY_NCCL(R.AllGather(d_flag + world, d_flag, 1, ncclInt64, comm, stream));
    std::vector<long long> f(world);
    SQY_HIPC(hipMemcpyAsync(f.data(), d_flag, sizeof(long long) * (size_t)world, hipMemcpyDeviceToHost, stream));
    SQY_HIPC(hipStreamSynchronize(stream));
    for (int r = 0; r < world; ++r)
        if (!f[r]) return 1;                                              // every rank returns 1 together
    int rc = 0;
    auto nccl_ok = [&](ncclResult_t r_, const char* what) {
        if (r_ != ncclSuccess) { std::fprintf(stderr, "[sqeazy]\t RCCL error %d in %s\n", (int)r_, what); rc = 1; }
    };
    nccl_ok(R.GroupStart(), "ncclGroupStart");
    if (rc == 0) {
        if (rank == root) {
            uint64_t off = 0;
            for (int r = 0; r < world && rc == 0; ++r) {
                if (r != root && h[r] > 0) nccl_ok(R.Recv(static_cast<char*>(d_recv) + off, (size_t)h[r], ncclChar, r, comm, stream), "ncclRecv");
                off += (uint64_t)h[r];
            }
        } else if (nbytes > 0) {
            nccl_ok(R.Send(d_blob, (size_t)nbytes, ncclChar, root, comm, stream), "ncclSend");
        }
        nccl_ok(R.GroupEnd(), "ncclGroupEnd");                            // (always: an open group would stay with the communicator)
    }
    if (rc) return 1;
    if (rank == root && nbytes > 0) {
        uint64_t off = 0;
        for (int r = 0; r < root; ++r) off += (uint64_t)h[r];
        SQY_HIPC(hipMemcpyAsync(static_cast<char*>(d_recv) + off, d_blob, (size_t)nbytes, hipMemcpyDeviceToDevice, stream));
    }
    SQY_HIPC(hipStreamSynchronize(stream));
    return 0;
    });
}

} // extern "C"
