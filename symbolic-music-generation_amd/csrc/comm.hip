// Gradient exchange of the data-parallel training step over RCCL, behind the C-ABI: one communicator per process
// (one process per GPU), buckets of the flat fp32 gradient buffer reduced on the communicator's own HIP stream so that
// they overlap the rest of backward.
//
// Reference: lightning's DDP strategy behind `lightning.Trainer(...)` (/root/reference/src/train.py:63-72,
// src/train_aria.py:322-336): torch DistributedDataParallel all-reduces gradient buckets during backward and divides by
// the world size.
//
// RCCL is opened at run time (dlopen) the first time a communicator is made: the library has no link-time dependency
// on it, and a process that already carries an RCCL (PyTorch's) keeps using that one.
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.handle) return MGEN_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    MGEN_REQUIRE(h != nullptr, "mgen_comm: cannot open librccl (%s)", dlerror());
#define MGEN_SYM(field, name)                                                        \
    *(void**)(&g_rccl.field) = dlsym(h, name);                                       \
    MGEN_REQUIRE(g_rccl.field != nullptr, "mgen_comm: librccl has no symbol %s", name)
    MGEN_SYM(GetUniqueId, "ncclGetUniqueId");
    MGEN_SYM(CommInitRank, "ncclCommInitRank");
    MGEN_SYM(AllReduce, "ncclAllReduce");
    MGEN_SYM(CommDestroy, "ncclCommDestroy");
    MGEN_SYM(CommAbort, "ncclCommAbort");
    MGEN_SYM(CommGetAsyncError, "ncclCommGetAsyncError");
    MGEN_SYM(GetErrorString, "ncclGetErrorString");
    MGEN_SYM(GetVersion, "ncclGetVersion");
#undef MGEN_SYM
    g_rccl.handle = h;
    return MGEN_OK;
}

}  // namespace

struct mgen_comm {
    ncclComm_t comm;
    hipStream_t stream;     // the reductions run here
    hipEvent_t ready;       // recorded on the compute stream when a bucket's gradients are final
    hipEvent_t done;        // recorded on the reduction stream behind the last bucket
    int rank, world, device;
    long bytes;             // reduced since the last wait (bookkeeping for the caller)
    int buckets;
    bool aborted;           // an asynchronous RCCL error was seen: the communicator has been aborted
};

// A rank that died or a link error shows up as an asynchronous error of the communicator: abort it (which unblocks
// the kernels of the collectives still queued, so the process can report the failure instead of hanging) and fail.
static int comm_check_async(mgen_comm* c, const char* what) {
    if (c->aborted) {
        mgen_set_error("%s: the communicator was aborted after an RCCL error", what);
        return MGEN_ELAUNCH;
    }
    ncclResult_t st = ncclSuccess;
    const ncclResult_t r = g_rccl.CommGetAsyncError(c->comm, &st);
    if (r == ncclSuccess && (st == ncclSuccess || st == ncclInProgress)) return MGEN_OK;
    mgen_set_error("%s: RCCL reports an asynchronous error (%s); communicator aborted", what,
                   g_rccl.GetErrorString(r != ncclSuccess ? r : st));
    g_rccl.CommAbort(c->comm);
    c->aborted = true;
    return MGEN_ELAUNCH;
}

#define MGEN_NCCL(call, what)                                                                    \
    do {                                                                                         \
        ncclResult_t r_ = (call);                                                                \
        if (r_ != ncclSuccess) {                                                                 \
            mgen_set_error("%s: %s", what, g_rccl.GetErrorString(r_));                           \
            return MGEN_ELAUNCH;                                                                 \
        }                                                                                        \
    } while (0)
#define MGEN_HIP(call, what)                                                                     \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            mgen_set_error("%s: %s", what, hipGetErrorString(e_));                               \
            return MGEN_ELAUNCH;                                                                 \
        }                                                                                        \
    } while (0)

// 128 bytes that rank 0 creates and hands to every rank out of band (a file, a TCP store, torch.distributed)
extern "C" int mgen_comm_unique_id(void* id128) {
    MGEN_REQUIRE(id128 != nullptr, "mgen_comm_unique_id: null pointer");
    if (int e = load_rccl()) return e;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    MGEN_NCCL(g_rccl.GetUniqueId(&id), "mgen_comm_unique_id");
    memcpy(id128, &id, sizeof(id));
    return MGEN_OK;
}

extern "C" int mgen_comm_rccl_version(int* version) {
    MGEN_REQUIRE(version != nullptr, "mgen_comm_rccl_version: null pointer");
    if (int e = load_rccl()) return e;
    MGEN_NCCL(g_rccl.GetVersion(version), "mgen_comm_rccl_version");
    return MGEN_OK;
}

extern "C" int mgen_comm_init(mgen_comm** out, const void* id128, int rank, int world, int device) {
    MGEN_REQUIRE(out && id128 && world >= 1 && rank >= 0 && rank < world && device >= 0, "mgen_comm_init: bad arguments");
    if (int e = load_rccl()) return e;
    int prev = -1;
    MGEN_HIP(hipGetDevice(&prev), "mgen_comm_init: hipGetDevice");
    MGEN_HIP(hipSetDevice(device), "mgen_comm_init: hipSetDevice");
    mgen_comm* c = new mgen_comm();
    c->rank = rank; c->world = world; c->device = device; c->bytes = 0; c->buckets = 0; c->aborted = false;
    c->stream = nullptr; c->ready = nullptr; c->done = nullptr;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    int rc = MGEN_OK;
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        mgen_set_error("mgen_comm_init: ncclCommInitRank: %s", g_rccl.GetErrorString(r));
        rc = MGEN_ELAUNCH;
    } else if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
               hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
               hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        mgen_set_error("mgen_comm_init: cannot create the stream / events");
        if (c->done) (void)hipEventDestroy(c->done);      // whichever of the three exist
        if (c->ready) (void)hipEventDestroy(c->ready);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        g_rccl.CommDestroy(c->comm);
        rc = MGEN_ELAUNCH;
    }
    (void)hipSetDevice(prev);   // the caller's current device is left as it was
    if (rc != MGEN_OK) {
        delete c;
        return rc;
    }
    *out = c;
    return MGEN_OK;
}

// Mean (average != 0) or sum over the ranks of buf[0 .. n), in place, fp32. The reduction is enqueued on the
// communicator's stream behind everything `compute_stream` holds at this moment; the call returns at once.
extern "C" int mgen_comm_allreduce_bucket(mgen_comm* c, float* buf, long n, int average, hipStream_t compute_stream) {
    MGEN_REQUIRE(c && buf && n > 0, "mgen_comm_allreduce_bucket: bad arguments");
    if (int e = comm_check_async(c, "mgen_comm_allreduce_bucket")) return e;
    MGEN_HIP(hipEventRecord(c->ready, compute_stream), "mgen_comm_allreduce_bucket: hipEventRecord");
    MGEN_HIP(hipStreamWaitEvent(c->stream, c->ready, 0), "mgen_comm_allreduce_bucket: hipStreamWaitEvent");
    MGEN_NCCL(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat, average ? ncclAvg : ncclSum, c->comm, c->stream),
              "mgen_comm_allreduce_bucket: ncclAllReduce");
    c->bytes += n * 4;
    c->buckets += 1;
    return MGEN_OK;
}

// Makes `stream` wait for every reduction enqueued so far (no host synchronisation). bytes / buckets (optional) return
// what was reduced since the previous wait.
extern "C" int mgen_comm_wait(mgen_comm* c, hipStream_t stream, long* bytes, int* buckets) {
    MGEN_REQUIRE(c != nullptr, "mgen_comm_wait: null communicator");
    if (int e = comm_check_async(c, "mgen_comm_wait")) return e;
    MGEN_HIP(hipEventRecord(c->done, c->stream), "mgen_comm_wait: hipEventRecord");
    MGEN_HIP(hipStreamWaitEvent(stream, c->done, 0), "mgen_comm_wait: hipStreamWaitEvent");
    if (bytes) *bytes = c->bytes;
    if (buckets) *buckets = c->buckets;
    c->bytes = 0;
    c->buckets = 0;
    return MGEN_OK;
}

// Gives up on the communicator: collectives still queued are unblocked and fail; every later call on it fails.
extern "C" int mgen_comm_abort(mgen_comm* c) {
    MGEN_REQUIRE(c != nullptr, "mgen_comm_abort: null communicator");
    if (!c->aborted) {
        g_rccl.CommAbort(c->comm);
        c->aborted = true;
    }
    return MGEN_OK;
}

extern "C" int mgen_comm_destroy(mgen_comm* c) {
    if (!c) return MGEN_OK;
    if (!c->aborted) (void)hipStreamSynchronize(c->stream);
    if (g_rccl.handle && !c->aborted) g_rccl.CommDestroy(c->comm);
    (void)hipEventDestroy(c->ready);
    (void)hipEventDestroy(c->done);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return MGEN_OK;
}
