// somar_amd/csrc/comm_rccl.cpp -- one-process-per-GPU transport over RCCL (xGMI inside a node).
//
// Replaces the reference's flat-MPI communication inside Chombo:
//   LevelData::exchange(Copier)            -> one grouped ncclSend/ncclRecv per neighbouring rank,
//                                             packed by k_pack_items, all on the solver's stream
//   MPI_Allreduce(MAX) of the residual norm (MappedAMRMultiGrid.H:825)          -> ncclAllReduce(max)
//   2 x MPI_Allreduce(SUM) in ZeroAvgConstInterpPS (ProlongationStrategy.cpp:141,148)
//                                          -> ONE 2-element ncclAllReduce(sum), result stays in HBM
// xGMI is point-to-point (7 links/GPU): a box layout sharded in x/y blocks talks to <= 8
// neighbours, each message <= 1 MiB, so the exchange is latency-bound; everything is enqueued
// on the compute stream so no host round trip sits between pack, wire and unpack.
//
// RCCL is bound at run time (dlopen) so that a process that already carries a copy of
// librccl (e.g. through PyTorch) does not end up with two.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "level.h"

namespace somar {

struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    const char* (*GetErrorString)(ncclResult_t);
    void* handle = nullptr;
};

static RcclApi& api()
{
    static RcclApi a;
    if (a.handle) return a;
    const char* names[] = {getenv("SOMAR_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n) continue;
        a.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (a.handle) break;
    }
    if (!a.handle) throw Error(-3, std::string("cannot dlopen RCCL: ") + dlerror());
#define SYM(field, name)                                                       \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.handle, name));      \
    if (!a.field) throw Error(-3, std::string("RCCL symbol missing: ") + name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    return a;
}

#define SOMAR_NCCL(call)                                                                       \
    do {                                                                                       \
        ncclResult_t r_ = (call);                                                              \
        if (r_ != ncclSuccess)                                                                 \
            throw Error(-3, std::string("RCCL error ") + api().GetErrorString(r_) + " at " + \
                                __FILE__ + ":" + std::to_string(__LINE__));                   \
    } while (0)

struct RcclComm : Comm {
    ncclComm_t comm = nullptr;
    ~RcclComm() override
    {
        if (comm) api().CommDestroy(comm);
    }
    void allreduce(double* dbuf, int n, int op, hipStream_t st) override
    {
        if (size == 1) return;
        allreduce_raw(dbuf, n, op, st);
    }
    void allreduce_raw(double* dbuf, int n, int op, hipStream_t st) override
    {
        SOMAR_NCCL(api().AllReduce(dbuf, dbuf, (size_t)n, ncclDouble, op ? ncclMax : ncclSum, comm, st));
    }
    void neighbor_exchange(const double* sendbuf, double* recvbuf, const std::vector<int>& peers,
                           const std::vector<long long>& soff, const std::vector<long long>& scount,
                           const std::vector<long long>& roff, const std::vector<long long>& rcount,
                           hipStream_t st) override
    {
        SOMAR_NCCL(api().GroupStart());
        for (size_t q = 0; q < peers.size(); ++q) {
            if (scount[q]) SOMAR_NCCL(api().Send(sendbuf + soff[q], (size_t)scount[q], ncclDouble, peers[q], comm, st));
            if (rcount[q]) SOMAR_NCCL(api().Recv(recvbuf + roff[q], (size_t)rcount[q], ncclDouble, peers[q], comm, st));
        }
        SOMAR_NCCL(api().GroupEnd());
    }
};

void rccl_unique_id(unsigned char* id128)
{
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    SOMAR_NCCL(api().GetUniqueId(&id));
    std::memcpy(id128, &id, 128);
}

Comm* rccl_create(const unsigned char* id128, int rank, int nranks, int device)
{
    SOMAR_HIP(hipSetDevice(device));
    RcclComm* c = new RcclComm;
    c->rank = rank;
    c->size = nranks;
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    // RCCL polls hipGetLastError() during init and reports whatever it finds as its own "unhandled cuda error":
    // a sticky error left behind by an unrelated earlier call in this process must not take the communicator down
    const hipError_t stale = hipGetLastError();
    try {
        ncclResult_t r_ = api().CommInitRank(&c->comm, nranks, id, rank);
        if (r_ != ncclSuccess)
            throw Error(-3, std::string("RCCL error ") + api().GetErrorString(r_) + " in ncclCommInitRank (HIP error pending before the call: " +
                                hipGetErrorString(stale) + ")");
    } catch (...) {
        delete c;
        throw;
    }
    return c;
}

}  // namespace somar
