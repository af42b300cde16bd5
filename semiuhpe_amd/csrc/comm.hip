// suhpe_comm_*: the data-parallel exchange of the training step behind the C ABI -- one RCCL communicator
// per rank (one process per GPU), float32 all-reduce (sum / average) for the gradient buckets and the BatchNorm buffers, all-gather
// for the per-rank entropy vectors of the dynamic threshold, broadcast for the initial weights.  New capability: the reference
// trains on one GPU (train.py:24-25).  RCCL is resolved at run time (dlopen of librccl.so.1: the copy torch already mapped when the
// caller is a PyTorch process), so libsuhpe.so loads -- and every other entry point works -- on a box without it.
#include "common.h"
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace suhpe {

// the slice of rccl.h this file uses (ABI of RCCL 2.x: ncclUniqueId is 128 opaque bytes, passed by value)
struct UniqueId { char bytes[128]; };
typedef void* Comm;
enum { kFloat = 7, kSum = 0, kAvg = 4 };
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    char why[384] = "librccl.so.1 not found (dlopen)";
};

static Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // SUHPE_RCCL_LIB: another build of the collective library (any shared object that exports the seven nccl* entry points used below);
        // tests/test_comm_abi_two_ranks.py points it at a host-memory transport to drive this ABI with two ranks on a box without GPUs
        const char* override_path = getenv("SUHPE_RCCL_LIB");
        if (override_path && *override_path) {
            // an override that does not load is an ERROR, never a silent fall back to the system RCCL (a typo would run the host-buffer
            // tests against real RCCL with host pointers): the communicator stays unavailable and every suhpe_comm_* call says why
            r.handle = dlopen(override_path, RTLD_NOW | RTLD_GLOBAL);
            if (!r.handle) { const char* why = dlerror(); snprintf(r.why, sizeof(r.why), "SUHPE_RCCL_LIB=%s did not load (%s)", override_path, why ? why : "?"); return; }
        }
        if (!r.handle)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (r.handle) break;
            }
        if (!r.handle) return;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.handle, "ncclAllReduce"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.handle, "ncclAllGather"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(dlsym(r.handle, "ncclBroadcast"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.AllGather && r.Broadcast && r.GetErrorString;
    });
    return r;
}

struct CommState { Comm comm; int rank, world; };

static int rccl_rc(int rc, const char* what) {
    if (rc == 0) return SUHPE_OK;
    set_error("%s: RCCL error %d (%s)", what, rc, rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
    return SUHPE_ELAUNCH;
}

}  // namespace suhpe

using namespace suhpe;

extern "C" {

int suhpe_comm_available(void) { return rccl().ok ? 1 : 0; }

int suhpe_comm_unique_id(void* id128) {
    SUHPE_REQUIRE(id128, "suhpe_comm_unique_id: null pointer");
    SUHPE_REQUIRE(rccl().ok, "suhpe_comm_unique_id: %s", rccl().why);
    UniqueId id;
    const int rc = rccl_rc(rccl().GetUniqueId(&id), "suhpe_comm_unique_id");
    if (rc == SUHPE_OK) memcpy(id128, id.bytes, sizeof(id.bytes));
    return rc;
}

int suhpe_comm_init(suhpe_comm_t* comm, const void* id128, int rank, int world) {
    SUHPE_REQUIRE(comm && id128, "suhpe_comm_init: null pointer");
    SUHPE_REQUIRE(world >= 1 && rank >= 0 && rank < world, "suhpe_comm_init: rank %d outside a world of %d", rank, world);
    SUHPE_REQUIRE(rccl().ok, "suhpe_comm_init: %s", rccl().why);
    UniqueId id;
    memcpy(id.bytes, id128, sizeof(id.bytes));
    Comm c = nullptr;
    const int rc = rccl_rc(rccl().CommInitRank(&c, world, id, rank), "suhpe_comm_init");       // binds to the calling thread's current device
    if (rc != SUHPE_OK) return rc;
    *comm = reinterpret_cast<suhpe_comm_t>(new CommState{c, rank, world});
    return SUHPE_OK;
}

int suhpe_comm_rank(suhpe_comm_t comm) { return comm ? reinterpret_cast<CommState*>(comm)->rank : -1; }
int suhpe_comm_world(suhpe_comm_t comm) { return comm ? reinterpret_cast<CommState*>(comm)->world : 0; }

int suhpe_comm_allreduce(suhpe_comm_t comm, float* buf, int64_t count, int average, suhpe_stream_t stream) {
    SUHPE_REQUIRE(comm && (buf || count == 0) && count >= 0, "suhpe_comm_allreduce: null pointer / negative count");
    if (count == 0) return SUHPE_OK;
    CommState* s = reinterpret_cast<CommState*>(comm);
    return rccl_rc(rccl().AllReduce(buf, buf, (size_t)count, kFloat, average ? kAvg : kSum, s->comm, as_stream(stream)), "suhpe_comm_allreduce");
}

int suhpe_comm_allgather(suhpe_comm_t comm, const float* send, float* recv, int64_t count_per_rank, suhpe_stream_t stream) {
    SUHPE_REQUIRE(comm && send && recv && count_per_rank > 0, "suhpe_comm_allgather: null pointer / empty block");
    CommState* s = reinterpret_cast<CommState*>(comm);
    return rccl_rc(rccl().AllGather(send, recv, (size_t)count_per_rank, kFloat, s->comm, as_stream(stream)), "suhpe_comm_allgather");
}

int suhpe_comm_broadcast(suhpe_comm_t comm, float* buf, int64_t count, int root, suhpe_stream_t stream) {
    SUHPE_REQUIRE(comm && (buf || count == 0) && count >= 0, "suhpe_comm_broadcast: null pointer / negative count");
    CommState* s = reinterpret_cast<CommState*>(comm);
    SUHPE_REQUIRE(root >= 0 && root < s->world, "suhpe_comm_broadcast: root %d outside a world of %d", root, s->world);
    if (count == 0) return SUHPE_OK;
    return rccl_rc(rccl().Broadcast(buf, buf, (size_t)count, kFloat, root, s->comm, as_stream(stream)), "suhpe_comm_broadcast");
}

int suhpe_comm_destroy(suhpe_comm_t comm) {
    if (!comm) return SUHPE_OK;
    CommState* s = reinterpret_cast<CommState*>(comm);
    const int rc = rccl_rc(rccl().CommDestroy(s->comm), "suhpe_comm_destroy");
    delete s;
    return rc;
}

}  // extern "C"
