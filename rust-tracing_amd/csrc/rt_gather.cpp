// rt_gather.cpp — the frame-end gather of a tile-sharded frame over RCCL / xGMI (the reference has no
// counterpart: its frame is one Vec<Color> in one address space, src/renderer.rs:26-49).
//
// Each rank renders its RT_OUT_TILES buffer with no communication (tile k -> rank k mod N); at frame end ONE grouped
// exchange brings the N buffers to the root: N - 1 ncclRecv on the root, one ncclSend on every other rank, fused in a
// ncclGroupStart / ncclGroupEnd section.  xGMI links are point to point, so each of the 7 transfers of an 8-GPU node runs on
// its own link (C5: 7 x 7.7 MB) — a ring collective would only serialise them.
//
// RCCL is bound at run time (dlopen): a host with one GPU and no RCCL installed still loads and uses the renderer.
#include "rt_api.hpp"

#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>

using namespace rtapi;

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

// loads librccl once; on failure every later call reports the same reason
const Rccl &rccl() {
    static Rccl r = [] {
        Rccl x;
        // RT_RCCL_LIB names the library to bind (a site's own RCCL build; the test suite's in-process stand-in): then that one or none
        const char *chosen = getenv("RT_RCCL_LIB");
        if (chosen && *chosen) {
            x.handle = dlopen(chosen, RTLD_NOW | RTLD_LOCAL);
        } else {
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                x.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (x.handle) break;
            }
        }
        if (!x.handle) {
            const char *e = dlerror();
            x.error = std::string(chosen && *chosen ? "RT_RCCL_LIB could not be loaded: " : "librccl.so could not be loaded: ") + (e ? e : "unknown reason");
            return x;
        }
        auto sym = [&](const char *name) -> void * {
            void *p = dlsym(x.handle, name);
            if (!p && x.error.empty()) x.error = std::string("librccl.so lacks ") + name;
            return p;
        };
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(sym("ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(sym("ncclCommInitRank"));
        x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(sym("ncclCommInitAll"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
        x.CommCount = reinterpret_cast<decltype(x.CommCount)>(sym("ncclCommCount"));
        x.CommUserRank = reinterpret_cast<decltype(x.CommUserRank)>(sym("ncclCommUserRank"));
        x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
        x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
        x.Send = reinterpret_cast<decltype(x.Send)>(sym("ncclSend"));
        x.Recv = reinterpret_cast<decltype(x.Recv)>(sym("ncclRecv"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
        return x;
    }();
    return r;
}

int rccl_fail(const Rccl &r, const char *what, ncclResult_t e) {
    return fail(RT_ERR_COMM, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(e) : "RCCL error"));
}
#define RCCL_TRY(r, call)                                                                                      \
    do {                                                                                                       \
        ncclResult_t _e = (call);                                                                              \
        if (_e != ncclSuccess) return rccl_fail(r, #call, _e);                                                 \
    } while (0)

} // namespace

struct rt_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, n_ranks = 1, device = 0;
    bool owned = true; // false: the caller's communicator (rt_comm_adopt)
};

extern "C" {

static_assert(RT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "rt_amd.h's id size must be RCCL's");

int rt_comm_get_unique_id(uint8_t out_id[RT_COMM_ID_BYTES]) {
    if (!out_id) return fail(RT_ERR_INVALID_ARGUMENT, "rt_comm_get_unique_id: null argument");
    const Rccl &r = rccl();
    if (!r.error.empty()) return fail(RT_ERR_COMM, r.error);
    ncclUniqueId id;
    RCCL_TRY(r, r.GetUniqueId(&id));
    memcpy(out_id, id.internal, RT_COMM_ID_BYTES);
    return RT_OK;
}

int rt_comm_create(const uint8_t id[RT_COMM_ID_BYTES], int rank, int n_ranks, int device, rt_comm **out_comm) {
    if (!id || !out_comm || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(RT_ERR_INVALID_ARGUMENT, "rt_comm_create: bad argument");
    *out_comm = nullptr;
    const Rccl &r = rccl();
    if (!r.error.empty()) return fail(RT_ERR_COMM, r.error);
    if (device < 0 || device >= rt_device_count()) return fail(RT_ERR_NO_DEVICE, "rt_comm_create: device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId uid;
    memcpy(uid.internal, id, RT_COMM_ID_BYTES);
    rt_comm *c = new rt_comm();
    c->rank = rank; c->n_ranks = n_ranks; c->device = device;
    const ncclResult_t e = r.CommInitRank(&c->comm, n_ranks, uid, rank);
    if (e != ncclSuccess) { delete c; return rccl_fail(r, "ncclCommInitRank", e); }
    *out_comm = c;
    return RT_OK;
}

int rt_comm_create_all(int n_devices, const int *devices, rt_comm **out_comms) {
    if (n_devices < 1 || !out_comms) return fail(RT_ERR_INVALID_ARGUMENT, "rt_comm_create_all: bad argument");
    for (int i = 0; i < n_devices; ++i) out_comms[i] = nullptr;
    const Rccl &r = rccl();
    if (!r.error.empty()) return fail(RT_ERR_COMM, r.error);
    const int ndev = rt_device_count();
    std::vector<int> devs(n_devices);
    for (int i = 0; i < n_devices; ++i) {
        devs[i] = devices ? devices[i] : i;
        if (devs[i] < 0 || devs[i] >= ndev) return fail(RT_ERR_NO_DEVICE, "rt_comm_create_all: device ordinal out of range");
    }
    std::vector<ncclComm_t> comms(n_devices, nullptr);
    RCCL_TRY(r, r.CommInitAll(comms.data(), n_devices, devs.data()));
    for (int i = 0; i < n_devices; ++i) {
        rt_comm *c = new rt_comm();
        c->comm = comms[i]; c->rank = i; c->n_ranks = n_devices; c->device = devs[i];
        out_comms[i] = c;
    }
    return RT_OK;
}

int rt_comm_adopt(void *nccl_comm, int device, rt_comm **out_comm) {
    if (!nccl_comm || !out_comm) return fail(RT_ERR_INVALID_ARGUMENT, "rt_comm_adopt: null argument");
    *out_comm = nullptr;
    const Rccl &r = rccl();
    if (!r.error.empty()) return fail(RT_ERR_COMM, r.error);
    rt_comm *c = new rt_comm();
    c->comm = (ncclComm_t)nccl_comm; c->owned = false; c->device = device;
    ncclResult_t e = r.CommCount(c->comm, &c->n_ranks);
    if (e == ncclSuccess) e = r.CommUserRank(c->comm, &c->rank);
    if (e != ncclSuccess) { delete c; return rccl_fail(r, "rt_comm_adopt", e); }
    *out_comm = c;
    return RT_OK;
}

void rt_comm_destroy(rt_comm *comm) {
    if (!comm) return;
    if (comm->owned && comm->comm && rccl().CommDestroy) (void)rccl().CommDestroy(comm->comm);
    delete comm;
}

int rt_comm_rank(const rt_comm *comm) { return comm ? comm->rank : -1; }
int rt_comm_size(const rt_comm *comm) { return comm ? comm->n_ranks : -1; }

int rt_gather_tiles_device(rt_comm *comm, int32_t width, int32_t height, int32_t elem_bytes, const void *d_tiles, void *d_gathered,
                           int root, void *hip_stream) {
    if (!comm || !d_tiles || width <= 0 || height <= 0) return fail(RT_ERR_INVALID_ARGUMENT, "rt_gather_tiles_device: bad argument");
    if (elem_bytes != 8 && elem_bytes != 1) return fail(RT_ERR_INVALID_ARGUMENT, "rt_gather_tiles_device: elem_bytes must be 8 (f64 sums) or 1 (RGB8)");
    if (root < 0 || root >= comm->n_ranks) return fail(RT_ERR_INVALID_ARGUMENT, "rt_gather_tiles_device: root out of range");
    if (comm->rank == root && !d_gathered) return fail(RT_ERR_INVALID_ARGUMENT, "rt_gather_tiles_device: the root needs d_gathered");
    const Rccl &r = rccl();
    if (!r.error.empty()) return fail(RT_ERR_COMM, r.error);
    const hipStream_t stream = (hipStream_t)hip_stream;
    const int n = comm->n_ranks;
    const ncclDataType_t type = elem_bytes == 8 ? ncclDouble : ncclUint8;
    // every rank's slot is as long as the largest shard (shard 0); a rank sends what it really holds
    const int64_t stride = rt_out_size(width, height, RT_OUT_TILES, 0, n);
    const int64_t mine = rt_out_size(width, height, RT_OUT_TILES, comm->rank, n);
    if (stride < 0 || mine < 0) return fail(RT_ERR_INVALID_ARGUMENT, "rt_gather_tiles_device: bad frame size");
    HIP_TRY(hipSetDevice(comm->device));
    unsigned char *const all = static_cast<unsigned char *>(d_gathered);
    if (comm->rank == root) { // the root's own part never leaves the device
        unsigned char *own = all + (size_t)root * (size_t)stride * (size_t)elem_bytes;
        if (mine > 0 && own != d_tiles)
            HIP_TRY(hipMemcpyAsync(own, d_tiles, (size_t)mine * (size_t)elem_bytes, hipMemcpyDeviceToDevice, stream));
    }
    if (n == 1) return RT_OK;
    ncclResult_t e = r.GroupStart();
    if (e != ncclSuccess) return rccl_fail(r, "ncclGroupStart", e);
    if (comm->rank == root) {
        for (int peer = 0; peer < n && e == ncclSuccess; ++peer) {
            if (peer == root) continue;
            const int64_t count = rt_out_size(width, height, RT_OUT_TILES, peer, n);
            if (count > 0) e = r.Recv(all + (size_t)peer * (size_t)stride * (size_t)elem_bytes, (size_t)count, type, peer, comm->comm, stream);
        }
    } else if (mine > 0) {
        e = r.Send(d_tiles, (size_t)mine, type, root, comm->comm, stream);
    }
    const ncclResult_t e2 = r.GroupEnd(); // (always closed, also after a failed call inside)
    if (e != ncclSuccess) return rccl_fail(r, "ncclSend / ncclRecv", e);
    if (e2 != ncclSuccess) return rccl_fail(r, "ncclGroupEnd", e2);
    return RT_OK;
}

} // extern "C"
