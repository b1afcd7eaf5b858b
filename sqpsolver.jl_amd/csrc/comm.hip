// comm.hip -- the only exchange between ranks: an all-gather of the per-instance convergence status over RCCL
// (SURVEY.md section 8b sqphip_gather_status, kernel K10).
//
// The reference is a single process; the scaling axis of the hot path is a batch of independent NLP instances
// (ACOPF contingency scenarios), cut into contiguous blocks, one rank per GPU.  No iterate ever crosses ranks: each
// rank reports int32 (ret, iter, done) per instance of its block -- a few hundred bytes, latency-bound, xGMI bandwidth
// irrelevant.  RCCL is loaded at run time (dlopen of librccl.so.1) on the first sqphip_comm_* call, so that a
// single-GPU host needs no RCCL at all.
#include "ctx.hpp"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <algorithm>
#include <cstring>

using namespace sqphip;

namespace {

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool load()
    {
        if (h) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) { err = std::string("dlopen(librccl.so.1) failed: ") + dlerror(); return false; }
        GetUniqueId = (decltype(GetUniqueId))dlsym(h, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(h, "ncclCommInitRank");
        AllGather = (decltype(AllGather))dlsym(h, "ncclAllGather");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !AllGather || !CommDestroy || !GetErrorString) {
            err = "librccl lacks an expected symbol"; dlclose(h); h = nullptr; return false;
        }
        return true;
    }
};
Rccl g_rccl;

// contiguous blocks, sizes differing by at most one (sqpsolver.jl_amd/shard.py, shard_range)
void block_of(int total, int world, int rank, int &lo, int &hi)
{
    const int base = total / world, extra = total % world;
    lo = rank * base + std::min(rank, extra);
    hi = lo + base + (rank < extra ? 1 : 0);
}

}  // namespace

// 1 when librccl can be loaded in this process, else 0.  Hosts call it on every rank and agree (MIN over ranks, by their
// own channel) BEFORE anybody enters sqphip_comm_init, which is collective: a rank that cannot load RCCL would otherwise
// leave the others waiting inside ncclCommInitRank.
extern "C" int sqphip_comm_available(void) { return g_rccl.load() ? 1 : 0; }

extern "C" int sqphip_comm_unique_id(void *id128)
{
    if (!id128) return SQPHIP_EINVAL;
    if (!g_rccl.load()) { fprintf(stderr, "sqphip_comm_unique_id: %s\n", g_rccl.err.c_str()); return SQPHIP_ESTATE; }
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) { fprintf(stderr, "sqphip_comm_unique_id: %s\n", g_rccl.GetErrorString(r)); return SQPHIP_EHIP; }
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id128, &id, 128);
    return SQPHIP_OK;
}

extern "C" int sqphip_comm_init(sqphip_ctx *h, const void *id128, int32_t world, int32_t rank)
{
    if (!h || !id128 || world < 1 || rank < 0 || rank >= world) return SQPHIP_EINVAL;
    Ctx &C = h->c;
    if (C.comm) { C.err = "sqphip_comm_init: communicator already initialised"; return SQPHIP_ESTATE; }
    if (!g_rccl.load()) { C.err = g_rccl.err; return SQPHIP_ESTATE; }
    if (hipSetDevice(C.opt.device) != hipSuccess) { C.err = "hipSetDevice failed"; return SQPHIP_EHIP; }
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = g_rccl.CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) { C.err = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r); return SQPHIP_EHIP; }
    C.comm = comm; C.comm_world = world; C.comm_rank = rank;
    return SQPHIP_OK;
}

void sqphip::comm_release(Ctx &C)
{
    if (C.comm) { g_rccl.CommDestroy((ncclComm_t)C.comm); C.comm = nullptr; }
    if (C.comm_buf) { hipFree(C.comm_buf); C.comm_buf = nullptr; }
    C.comm_world = 1; C.comm_rank = 0;
}

extern "C" int sqphip_comm_destroy(sqphip_ctx *h)
{
    if (!h) return SQPHIP_EINVAL;
    comm_release(h->c);
    return SQPHIP_OK;
}

// (ret, iter, done) of all `total` instances of the job, ordered by global instance id: this context holds the block
// of its rank (total split into world contiguous blocks; the context's batch must equal its block).  Without a
// communicator (world 1) it is the local table.
extern "C" int sqphip_gather_status(sqphip_ctx *h, int32_t total, int32_t *ret_codes, int32_t *iters, int32_t *done)
{
    if (!h || total < 1) return SQPHIP_EINVAL;
    Ctx &C = h->c;
    const int world = C.comm ? C.comm_world : 1, rank = C.comm ? C.comm_rank : 0;
    int lo, hi;
    block_of(total, world, rank, lo, hi);
    if (hi - lo != C.d.B) { C.err = "sqphip_gather_status: the batch of this context is not its rank's block of `total`"; return SQPHIP_EINVAL; }
    try {
        SQPHIP_HIP_OK(hipSetDevice(C.opt.device));
        const int cap = (total + world - 1) / world;                    // largest block
        std::vector<SqpState> S(C.d.B);
        SQPHIP_HIP_OK(hipMemcpyAsync(S.data(), C.d.sst, sizeof(SqpState) * C.d.B, hipMemcpyDeviceToHost, C.stream));
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        std::vector<int32_t> loc(3 * (size_t)cap, 0), all(3 * (size_t)cap * world, 0);
        for (int b = 0; b < C.d.B; ++b) { loc[3 * b] = S[b].ret; loc[3 * b + 1] = S[b].iter; loc[3 * b + 2] = S[b].done; }
        if (!C.comm) all = loc;                 // no communicator: the local table.  With one, the collective runs even for world == 1
        else {
            // the device buffer grows with the largest block seen (a later call may name a larger `total`)
            if (C.comm_buf && C.comm_cap < cap) { SQPHIP_HIP_OK(hipFree(C.comm_buf)); C.comm_buf = nullptr; }
            if (!C.comm_buf) { SQPHIP_HIP_OK(hipMalloc(&C.comm_buf, sizeof(int32_t) * 3 * (size_t)cap * (world + 1))); C.comm_cap = cap; }
            int32_t *send = (int32_t *)C.comm_buf, *recv = send + 3 * (size_t)cap;
            SQPHIP_HIP_OK(hipMemcpyAsync(send, loc.data(), sizeof(int32_t) * 3 * (size_t)cap, hipMemcpyHostToDevice, C.stream));
            const ncclResult_t r = g_rccl.AllGather(send, recv, 3 * (size_t)cap, ncclInt32, (ncclComm_t)C.comm, C.stream);
            if (r != ncclSuccess) { C.err = std::string("ncclAllGather: ") + g_rccl.GetErrorString(r); return SQPHIP_EHIP; }
            SQPHIP_HIP_OK(hipMemcpyAsync(all.data(), recv, sizeof(int32_t) * all.size(), hipMemcpyDeviceToHost, C.stream));
            SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        }
        for (int r = 0; r < world; ++r) {
            int rlo, rhi;
            block_of(total, world, r, rlo, rhi);
            for (int b = 0; b < rhi - rlo; ++b) {
                const int32_t *t = all.data() + 3 * ((size_t)r * cap + b);
                if (ret_codes) ret_codes[rlo + b] = t[0];
                if (iters) iters[rlo + b] = t[1];
                if (done) done[rlo + b] = t[2];
            }
        }
        return SQPHIP_OK;
    } catch (const std::string &e) {
        C.err = e;
        return SQPHIP_EHIP;
    }
}
