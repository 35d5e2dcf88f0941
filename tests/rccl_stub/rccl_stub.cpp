// rccl_stub.cpp — TEST INFRASTRUCTURE, never part of the product: a stand-in for librccl.so that pairs ncclSend / ncclRecv between
// the communicators of ONE process and moves the bytes with hipMemcpyAsync (device to device, one GPU is enough).  librt_amd binds
// RCCL at run time; with RT_RCCL_LIB pointing here, tests/test_gpu_gather_stub.py drives 3 and 8 "ranks" through
// rt_gather_tiles_device on a one-GPU box — the N > 1 branch of rust-tracing_amd/csrc/rt_gather.cpp (peers, offsets, counts, the
// ragged last shard, f64 and RGB8), which needs more GPUs than the builder's box has when the real library is behind it.
//
// What it checks that a real exchange would also trip over: a receive is matched with the send of (same communicator id, source
// rank, destination rank), in order, and their element counts and types must agree (ncclInvalidArgument otherwise); a receive whose
// send never comes gives up after RCCL_STUB_TIMEOUT_S seconds (default 20) with ncclSystemError — the stub never hangs.
// What it does not model: transport, bootstrap, topology.  Sends are buffered BY REFERENCE (the stub never blocks a sender): the
// caller keeps the source buffer unchanged until the matching receive has been enqueued, which the tests do.
// RCCL_STUB_FAIL=init makes ncclCommInitRank fail (the set-up error path of the callers).
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

struct ncclComm {
    std::string id;
    int rank = 0, n = 1;
};

namespace {

struct Deposit {
    const void *src;
    size_t count;
    ncclDataType_t type;
    hipEvent_t ready; // recorded on the sender's stream when the send was issued
};
struct Op {
    bool send;
    const void *src;
    void *dst;
    size_t count;
    ncclDataType_t type;
    int peer;
    ncclComm *comm;
    hipStream_t stream;
};
std::mutex g_mu;
std::condition_variable g_cv;
std::map<std::tuple<std::string, int, int>, std::deque<Deposit>> g_mail; // (communicator id, from, to)
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
unsigned g_next_id = 1;

size_t elem_bytes(ncclDataType_t t) {
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}
double timeout_s() {
    const char *e = getenv("RCCL_STUB_TIMEOUT_S");
    return e ? atof(e) : 20.0;
}

ncclResult_t run(const Op &op) {
    if (!op.comm || op.peer < 0 || op.peer >= op.comm->n || op.peer == op.comm->rank || elem_bytes(op.type) == 0) return ncclInvalidArgument;
    if (op.send) {
        Deposit d{op.src, op.count, op.type, nullptr};
        if (hipEventCreateWithFlags(&d.ready, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
        if (hipEventRecord(d.ready, op.stream) != hipSuccess) return ncclUnhandledCudaError;
        std::lock_guard<std::mutex> lock(g_mu);
        g_mail[{op.comm->id, op.comm->rank, op.peer}].push_back(d);
        g_cv.notify_all();
        return ncclSuccess;
    }
    Deposit d;
    {
        std::unique_lock<std::mutex> lock(g_mu);
        auto &box = g_mail[{op.comm->id, op.peer, op.comm->rank}];
        if (!g_cv.wait_for(lock, std::chrono::duration<double>(timeout_s()), [&] { return !box.empty(); })) return ncclSystemError;
        d = box.front();
        box.pop_front();
    }
    ncclResult_t rc = ncclSuccess;
    if (d.count != op.count || d.type != op.type) rc = ncclInvalidArgument; // the two sides disagree on what is exchanged
    else if (hipStreamWaitEvent(op.stream, d.ready, 0) != hipSuccess ||
             hipMemcpyAsync(op.dst, d.src, d.count * elem_bytes(d.type), hipMemcpyDeviceToDevice, op.stream) != hipSuccess) rc = ncclUnhandledCudaError;
    (void)hipEventDestroy(d.ready);
    return rc;
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *out) {
    if (!out) return ncclInvalidArgument;
    memset(out->internal, 0, sizeof out->internal);
    std::lock_guard<std::mutex> lock(g_mu);
    snprintf(out->internal, sizeof out->internal, "rccl-stub-%u", g_next_id++);
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int n, ncclUniqueId id, int rank) {
    if (!comm || n < 1 || rank < 0 || rank >= n) return ncclInvalidArgument;
    const char *f = getenv("RCCL_STUB_FAIL");
    if (f && strcmp(f, "init") == 0) return ncclSystemError;
    ncclComm *c = new ncclComm();
    c->id.assign(id.internal, sizeof id.internal);
    c->rank = rank; c->n = n;
    *comm = c;
    return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *) {
    ncclUniqueId id;
    ncclResult_t rc = ncclGetUniqueId(&id);
    for (int i = 0; i < n && rc == ncclSuccess; ++i) rc = ncclCommInitRank(&comms[i], n, id, i);
    return rc;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return ncclSuccess; }
ncclResult_t ncclCommCount(const ncclComm_t comm, int *n) { if (!comm || !n) return ncclInvalidArgument; *n = comm->n; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *r) { if (!comm || !r) return ncclInvalidArgument; *r = comm->rank; return ncclSuccess; }
ncclResult_t ncclGroupStart() { ++t_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    ncclResult_t rc = ncclSuccess;
    for (int pass = 0; pass < 2; ++pass) // a group's sends are posted before its receives wait
        for (const Op &op : ops)
            if (op.send == (pass == 0)) { const ncclResult_t e = run(op); if (rc == ncclSuccess) rc = e; }
    return rc;
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    const Op op{true, buf, nullptr, count, type, peer, comm, stream};
    if (t_depth > 0) { t_ops.push_back(op); return ncclSuccess; }
    return run(op);
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    const Op op{false, nullptr, buf, count, type, peer, comm, stream};
    if (t_depth > 0) { t_ops.push_back(op); return ncclSuccess; }
    return run(op);
}
const char *ncclGetErrorString(ncclResult_t e) {
    switch (e) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "rccl stub: a HIP call failed";
    case ncclSystemError: return "rccl stub: no matching send arrived in time (or RCCL_STUB_FAIL)";
    case ncclInvalidArgument: return "rccl stub: invalid argument, or send and receive disagree on count / type";
    case ncclInvalidUsage: return "rccl stub: invalid usage";
    default: return "rccl stub: error";
    }
}

} // extern "C"
