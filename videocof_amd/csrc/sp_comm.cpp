// Library-owned sequence-parallel communicator ("wan_sp_init(ncclComm/rank/size)",
// "wan_sp_a2a_{scatter_heads,gather_heads}", "SP comm uses one library-owned side stream + events").
//
// Replaces, for a host that is not PyTorch, what videox_fun/dist/fuser.py:35-54 (process-group bootstrap through xfuser) and
// videox_fun/dist/wan_xfuser.py:68-111 (the head all-to-all inside xFuserLongContextAttention) do through torch.distributed.
// The Python host of this package keeps using torch.distributed's process group (videocof_amd/dist.py) -- or, with
// `videocof_amd.dist.LibraryComm`, this layer -- the wire layouts are the same either way (include/wan_hip.h, a21).
//
// One communicator = one RCCL comm + ONE side HIP stream + two events.  An exchange is enqueued on the side stream behind an
// event recorded on the caller's compute stream (so it sees every kernel enqueued so far: the projection that wrote the send
// buffer) and returns at once; the caller keeps enqueueing the next projection on its compute stream -- that is the overlap of
// DESIGN.md section 6 -- and calls wan_sp_wait before the kernel that reads the receive buffers.  Nothing synchronises the host.
//
// RCCL is bound at run time (dlopen "librccl.so.1", the soname both ROCm's and PyTorch's copies carry: inside a PyTorch process
// this resolves to the copy torch already loaded, never to a second one), so libwan_hip.so has no link-time dependency on it
// and single-GPU users never load it.
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>

#include <mutex>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "common.hpp"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllToAll)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    char why[256] = "not attempted";      // the loader's own account of a failure (dlerror() is consumed by reading it: capture it once, here)
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        size_t used = 0;
        r.why[0] = 0;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
            const char* e = dlerror();                 // every candidate's reason is kept, not only the last one's
            const int n = snprintf(r.why + used, sizeof(r.why) - used, "%s%s", used ? "; " : "", e ? e : name);
            if (n > 0) used = used + (size_t)n < sizeof(r.why) ? used + (size_t)n : sizeof(r.why) - 1;
        }
        if (!r.handle) return;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
        r.AllToAll = reinterpret_cast<decltype(r.AllToAll)>(dlsym(r.handle, "ncclAllToAll"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.handle, "ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllToAll && r.AllGather && r.GetErrorString;
        if (!r.ok) snprintf(r.why, sizeof(r.why), "librccl was loaded but lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllToAll / ncclAllGather / ncclGetErrorString");
    });
    return r;
}

}  // namespace

struct wan_sp_comm {
    ncclComm_t comm = nullptr;
    bool owns_comm = false;
    int rank = 0, world = 1, device = 0;
    hipStream_t side = nullptr;
    hipEvent_t ready = nullptr;                      // compute -> side
    // side -> compute: one event per exchange from a small ring, so that a wait joins ITS exchange only (the k exchange can be
    // consumed while the q exchange behind it is still in flight).  A ticket is the 1-based count of exchanges started on this
    // communicator; the event of ticket t lives in slot t % kRing and stays valid until kRing further exchanges have started.
    static constexpr int kRing = 8;
    hipEvent_t done[kRing] = {};
    int64_t ticket = 0;                              // exchanges started so far
    int64_t waited = 0;                              // every ticket <= waited has been joined by a wait on the caller's stream
    // Exchanges that ran INLINE on the caller's stream (round 5: when that stream is being captured into a hipGraph -- or the
    // developer switch sp_inline is on -- the collective is enqueued on the caller's stream itself: no side stream, no event fork /
    // join; a forked side stream under capture is what made hipStreamEndCapture segfault in round 4).  Such a ticket needs no wait.
    bool inline_slot[kRing] = {};
    bool captured = false;                           // some collective of this communicator was recorded into a hipGraph
};

// true: enqueue on the caller's stream (the stream is capturing, or sp_inline = 1)
static bool sp_runs_inline(wan_sp_comm* c, hipStream_t cs) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(cs, &st) != hipSuccess) { (void)hipGetLastError(); st = hipStreamCaptureStatusNone; }
    if (st == hipStreamCaptureStatusActive) { c->captured = true; return true; }
    return wan_tune(WAN_TUNE_SP_INLINE) != 0;
}

#define WAN_SP_HIP(call, what)                                                              \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess) {                                                            \
            wan_set_error("%s: %s", what, hipGetErrorString(e__));                          \
            return WAN_ERR_LAUNCH;                                                          \
        }                                                                                   \
    } while (0)
#define WAN_SP_NCCL(call, what)                                                             \
    do {                                                                                    \
        ncclResult_t r__ = (call);                                                          \
        if (r__ != ncclSuccess) {                                                           \
            wan_set_error("%s: RCCL: %s", what, rccl().GetErrorString(r__));                \
            return WAN_ERR_LAUNCH;                                                          \
        }                                                                                   \
    } while (0)

extern "C" wan_status_t wan_sp_unique_id(void* id128) {
    WAN_REQUIRE(id128 != nullptr, WAN_ERR_INVALID, "wan_sp_unique_id: null buffer");
    WAN_REQUIRE(rccl().ok, WAN_ERR_UNSUPPORTED, "wan_sp_unique_id: librccl.so.1 could not be loaded (%s)", rccl().why);
    static_assert(sizeof(ncclUniqueId) == WAN_SP_UNIQUE_ID_BYTES, "rendezvous token size");
    ncclUniqueId id;
    WAN_SP_NCCL(rccl().GetUniqueId(&id), "wan_sp_unique_id");
    memcpy(id128, &id, sizeof(id));
    return WAN_OK;
}

static wan_status_t finish_init(wan_sp_comm* c) {
    WAN_SP_HIP(hipGetDevice(&c->device), "wan_sp_init: hipGetDevice");
    WAN_SP_HIP(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking), "wan_sp_init: side stream");
    WAN_SP_HIP(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming), "wan_sp_init: event");
    for (int i = 0; i < wan_sp_comm::kRing; ++i) WAN_SP_HIP(hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming), "wan_sp_init: event");
    return WAN_OK;
}

extern "C" wan_status_t wan_sp_init(wan_sp_comm** out, const void* id128, int rank, int world_size) {
    WAN_REQUIRE(out != nullptr && id128 != nullptr, WAN_ERR_INVALID, "wan_sp_init: null argument");
    WAN_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, WAN_ERR_INVALID, "wan_sp_init: rank %d of %d", rank, world_size);
    WAN_REQUIRE(rccl().ok, WAN_ERR_UNSUPPORTED, "wan_sp_init: librccl.so.1 could not be loaded (%s)", rccl().why);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    wan_sp_comm* c = new wan_sp_comm();
    c->rank = rank; c->world = world_size; c->owns_comm = true;
    ncclResult_t r = rccl().CommInitRank(&c->comm, world_size, id, rank);
    if (r != ncclSuccess) {
        wan_set_error("wan_sp_init: RCCL: %s", rccl().GetErrorString(r));
        delete c;
        return WAN_ERR_LAUNCH;
    }
    const wan_status_t st = finish_init(c);
    if (st != WAN_OK) { (void)rccl().CommDestroy(c->comm); delete c; return st; }
    *out = c;
    return WAN_OK;
}

extern "C" wan_status_t wan_sp_init_from_comm(wan_sp_comm** out, void* nccl_comm, int rank, int world_size) {
    WAN_REQUIRE(out != nullptr && nccl_comm != nullptr, WAN_ERR_INVALID, "wan_sp_init_from_comm: null argument");
    WAN_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, WAN_ERR_INVALID, "wan_sp_init_from_comm: rank %d of %d", rank, world_size);
    WAN_REQUIRE(rccl().ok, WAN_ERR_UNSUPPORTED, "wan_sp_init_from_comm: librccl.so.1 could not be loaded (%s)", rccl().why);
    wan_sp_comm* c = new wan_sp_comm();
    c->comm = (ncclComm_t)nccl_comm; c->rank = rank; c->world = world_size; c->owns_comm = false;
    const wan_status_t st = finish_init(c);
    if (st != WAN_OK) { delete c; return st; }
    *out = c;
    return WAN_OK;
}

extern "C" int wan_sp_rank(const wan_sp_comm* c) { return c ? c->rank : -1; }
extern "C" int wan_sp_world_size(const wan_sp_comm* c) { return c ? c->world : 0; }

// slab d of `send` -> rank d, slab s of `recv` <- rank s; bytes_total = world_size equal slabs
static wan_status_t a2a_start(wan_sp_comm* c, const void* send, void* recv, int64_t bytes_total, void* compute_stream, const char* what) {
    WAN_REQUIRE(c != nullptr && send != nullptr && recv != nullptr, WAN_ERR_INVALID, "%s: null argument", what);
    WAN_REQUIRE(bytes_total > 0 && bytes_total % c->world == 0, WAN_ERR_INVALID, "%s: %lld bytes do not split into %d equal slabs",
                what, (long long)bytes_total, c->world);
    WAN_REQUIRE(send != recv, WAN_ERR_INVALID, "%s: in-place exchange is not supported (persistent send / receive pairs)", what);
    hipStream_t cs = (hipStream_t)compute_stream;
    if (sp_runs_inline(c, cs)) {           // in stream order on the caller's stream: nothing to fork, nothing to join
        WAN_SP_NCCL(rccl().AllToAll(send, recv, (size_t)(bytes_total / c->world), ncclInt8, c->comm, cs), what);
        ++c->ticket;
        c->inline_slot[c->ticket % wan_sp_comm::kRing] = true;
        return WAN_OK;
    }
    WAN_SP_HIP(hipEventRecord(c->ready, cs), what);                 // everything enqueued on the compute stream so far ...
    WAN_SP_HIP(hipStreamWaitEvent(c->side, c->ready, 0), what);     // ... precedes the exchange
    WAN_SP_NCCL(rccl().AllToAll(send, recv, (size_t)(bytes_total / c->world), ncclInt8, c->comm, c->side), what);
    ++c->ticket;
    c->inline_slot[c->ticket % wan_sp_comm::kRing] = false;
    WAN_SP_HIP(hipEventRecord(c->done[c->ticket % wan_sp_comm::kRing], c->side), what);
    return WAN_OK;
}

extern "C" wan_status_t wan_sp_a2a_scatter_heads(wan_sp_comm* c, const void* send_wire, void* recv_wire, int64_t bytes_total,
                                                 void* compute_stream) {
    return a2a_start(c, send_wire, recv_wire, bytes_total, compute_stream, "wan_sp_a2a_scatter_heads");
}

extern "C" wan_status_t wan_sp_a2a_gather_heads(wan_sp_comm* c, const void* send_wire, void* recv_wire, int64_t bytes_total,
                                                void* compute_stream) {
    return a2a_start(c, send_wire, recv_wire, bytes_total, compute_stream, "wan_sp_a2a_gather_heads");
}

extern "C" wan_status_t wan_sp_all_gather(wan_sp_comm* c, const void* send, void* recv, int64_t bytes_per_rank, void* compute_stream) {
    WAN_REQUIRE(c != nullptr && send != nullptr && recv != nullptr && bytes_per_rank > 0, WAN_ERR_INVALID, "wan_sp_all_gather: bad argument");
    hipStream_t cs = (hipStream_t)compute_stream;
    if (sp_runs_inline(c, cs)) {
        WAN_SP_NCCL(rccl().AllGather(send, recv, (size_t)bytes_per_rank, ncclInt8, c->comm, cs), "wan_sp_all_gather");
        ++c->ticket;
        c->inline_slot[c->ticket % wan_sp_comm::kRing] = true;
        return WAN_OK;
    }
    WAN_SP_HIP(hipEventRecord(c->ready, cs), "wan_sp_all_gather");
    WAN_SP_HIP(hipStreamWaitEvent(c->side, c->ready, 0), "wan_sp_all_gather");
    WAN_SP_NCCL(rccl().AllGather(send, recv, (size_t)bytes_per_rank, ncclInt8, c->comm, c->side), "wan_sp_all_gather");
    ++c->ticket;
    c->inline_slot[c->ticket % wan_sp_comm::kRing] = false;
    WAN_SP_HIP(hipEventRecord(c->done[c->ticket % wan_sp_comm::kRing], c->side), "wan_sp_all_gather");
    return WAN_OK;
}

// the ticket of the exchange started last (0 before the first one): pass it to wan_sp_wait_for
extern "C" int64_t wan_sp_ticket(const wan_sp_comm* c) { return c ? c->ticket : 0; }

// the caller's stream waits (on the device) for exchange `ticket` -- and, the side stream being in order, for every earlier one;
// exchanges started after it stay in flight.  A communicator is driven by ONE host thread (the reference is single-threaded per
// rank); the tickets are plain counters.
extern "C" wan_status_t wan_sp_wait_for(wan_sp_comm* c, int64_t ticket, void* compute_stream) {
    WAN_REQUIRE(c != nullptr, WAN_ERR_INVALID, "wan_sp_wait_for: null communicator");
    WAN_REQUIRE(ticket >= 0 && ticket <= c->ticket, WAN_ERR_INVALID, "wan_sp_wait_for: ticket %lld of %lld started", (long long)ticket, (long long)c->ticket);
    if (ticket == 0) return WAN_OK;
    // an event slot recycled by later exchanges marks a LATER point of the in-order side stream: waiting on it is still correct
    int64_t t = ticket + wan_sp_comm::kRing <= c->ticket ? c->ticket : ticket;
    // an exchange that ran inline on the caller's stream is already ordered; what may still need a join is the latest SIDE-stream
    // exchange at or before it (only the ring's span is known: older ones than that were recycled, i.e. joined or superseded)
    while (t > c->waited && t > c->ticket - wan_sp_comm::kRing && c->inline_slot[t % wan_sp_comm::kRing]) --t;
    if (t <= c->waited || t <= c->ticket - wan_sp_comm::kRing || c->inline_slot[t % wan_sp_comm::kRing]) return WAN_OK;
    WAN_SP_HIP(hipStreamWaitEvent((hipStream_t)compute_stream, c->done[t % wan_sp_comm::kRing], 0), "wan_sp_wait_for");
    if (t > c->waited) c->waited = t;
    return WAN_OK;
}

// join everything started so far
extern "C" wan_status_t wan_sp_wait(wan_sp_comm* c, void* compute_stream) {
    WAN_REQUIRE(c != nullptr, WAN_ERR_INVALID, "wan_sp_wait: null communicator");
    return wan_sp_wait_for(c, c->ticket, compute_stream);
}

extern "C" wan_status_t wan_sp_destroy(wan_sp_comm* c) {
    if (c == nullptr) return WAN_OK;
    // A communicator whose collectives were recorded into hipGraphs: RCCL's teardown (ncclCommDestroy and ncclCommAbort alike,
    // RCCL 2.26 / ROCm 7.0: profiles/r05/sp_graph_capture_inline.log) blocks for good while a graph that holds one of its kernels is
    // alive, and the library cannot know whether the caller has dropped its graphs (GraphedForward.reset()).  Such a communicator is
    // therefore LEFT ALONE -- its RCCL resources go with the process -- rather than risking a hang at interpreter exit.
    if (c->captured) return WAN_OK;
    (void)hipStreamSynchronize(c->side);
    if (c->owns_comm && c->comm) (void)rccl().CommDestroy(c->comm);
    if (c->ready) (void)hipEventDestroy(c->ready);
    for (hipEvent_t e : c->done) if (e) (void)hipEventDestroy(e);
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
    return WAN_OK;
}
