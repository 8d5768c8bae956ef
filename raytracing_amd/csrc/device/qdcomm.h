// Multi-GPU exchange step behind the C-ABI: one process per GPU, RCCL over xGMI.
// Included by qdtrace.hip (uses its set_err / untile_kernel).  The path has ONE exchange: every rank's compacted tile
// shard goes straight to rank 0 -- grouped ncclSend / ncclRecv, i.e. each peer's 1/N of the frame over that peer's own
// xGMI link to the root, not around a ring -- and rank 0 de-interleaves.  The AA pass adds a broadcast of the assembled
// frame and a gather of the taps.  RCCL is loaded on first use (dlopen by soname: a process that already holds an RCCL,
// e.g. torch's, shares it; a single-GPU caller never loads the 0.5 GB library).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

struct qd_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;

int rccl_load() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return 0;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return set_err(std::string("RCCL is not loadable: ") + dlerror());
    RcclApi a;
    a.handle = h;
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) ok = false; return p; };
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
    a.Send = (decltype(a.Send))sym("ncclSend");
    a.Recv = (decltype(a.Recv))sym("ncclRecv");
    a.Broadcast = (decltype(a.Broadcast))sym("ncclBroadcast");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
    if (!ok) return set_err("RCCL library lacks an expected entry point");
    g_rccl = a;
    return 0;
}

#define RCCL_TRY(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return set_err(std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); } while (0)

}  // namespace

extern "C" int qd_comm_unique_id(uint8_t id_out[128]) {
    if (!id_out) return set_err("qd_comm_unique_id: null argument");
    if (rccl_load()) return -1;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclUniqueId id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return 0;
}

extern "C" int qd_comm_create(const uint8_t id_bytes[128], int rank, int world, qd_comm** out) {
    if (!id_bytes || !out || world < 1 || rank < 0 || rank >= world) return set_err("qd_comm_create: bad argument");
    if (g_device < 0) return set_err("qd_comm_create: qd_init has not been called");
    *out = nullptr;
    if (rccl_load()) return -1;
    HIP_TRY(hipSetDevice(g_device));
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    qd_comm* c = new qd_comm;
    c->rank = rank; c->world = world;
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { delete c; return set_err(std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)); }
    *out = c;
    return 0;
}

extern "C" int qd_comm_destroy(qd_comm* c) {
    if (!c) return 0;
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return 0;
}

// count floats from every rank -> dst_dev[r * count ...] on rank 0.  src_dev == dst_dev + rank * count (rank 0 produced
// its block in place) skips the self copy.
extern "C" int qd_comm_gather_async(qd_comm* c, const float* src_dev, float* dst_dev, size_t count, void* hip_stream) {
    if (!c || !src_dev || (c->rank == 0 && !dst_dev)) return set_err("qd_comm_gather_async: null argument");
    hipStream_t stream = (hipStream_t)hip_stream;
    if (count == 0) return 0;
    RCCL_TRY(g_rccl.GroupStart());
    ncclResult_t r = ncclSuccess;
    if (c->rank == 0) {
        for (int p = 0; p < c->world && r == ncclSuccess; ++p) {
            if (p == 0 && src_dev == dst_dev) continue;
            if (p == 0) r = g_rccl.Send(src_dev, count, ncclFloat, 0, c->comm, stream);
            if (r == ncclSuccess) r = g_rccl.Recv(dst_dev + (size_t)p * count, count, ncclFloat, p, c->comm, stream);
        }
    } else r = g_rccl.Send(src_dev, count, ncclFloat, 0, c->comm, stream);
    const ncclResult_t e = g_rccl.GroupEnd();
    if (r != ncclSuccess) return set_err(std::string("ncclSend / ncclRecv: ") + g_rccl.GetErrorString(r));
    if (e != ncclSuccess) return set_err(std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(e));
    return 0;
}

extern "C" int qd_comm_broadcast_async(qd_comm* c, float* buf_dev, size_t count, void* hip_stream) {
    if (!c || !buf_dev) return set_err("qd_comm_broadcast_async: null argument");
    if (count == 0) return 0;
    RCCL_TRY(g_rccl.Broadcast(buf_dev, buf_dev, count, ncclFloat, 0, c->comm, (hipStream_t)hip_stream));
    return 0;
}

// The path's exchange step: shards -> rank 0 (gathered_dev: world blocks of tiles_per_rank_padded tiles), then the
// de-interleave kernel writes the row-major frame.  Enqueue only.
extern "C" int qd_gather_frame_async(qd_comm* c, const float* shard_dev, float* gathered_dev, float* frame_dev,
                                     int width, int height, int tile, void* hip_stream) {
    if (!c || !shard_dev) return set_err("qd_gather_frame_async: null argument");
    if (c->rank == 0 && (!gathered_dev || !frame_dev)) return set_err("qd_gather_frame_async: rank 0 needs the gather and frame buffers");
    if (width <= 0 || height <= 0 || tile < 8 || tile % 8) return set_err("qd_gather_frame_async: bad frame or tile size");
    const int n_tiles = ((width + tile - 1) / tile) * ((height + tile - 1) / tile);
    const int tpr = (n_tiles + c->world - 1) / c->world;
    const size_t count = (size_t)tpr * tile * tile * 3;
    if (qd_comm_gather_async(c, shard_dev, gathered_dev, count, hip_stream)) return -1;
    if (c->rank != 0) return 0;
    return qd_untile_async(gathered_dev, frame_dev, width, height, tile, c->world, tpr, hip_stream);
}
