// comm.hip -- the communicator of the chunk fan-out: the path's ONE collective is the broadcast of the
// transformed ST operand from the rank that holds the ST matrix to the ranks that solve chunks against it
// (/root/reference/cytospace/cytospace.py:438, 446-451 pickles the whole ST matrix to every pool worker instead).
//
// Two kinds behind one handle:
//   RCCL   one rank per GPU -- processes (cyto_comm_init: ncclCommInitRank with an id the launcher hands round) or host
//          threads of one process (cyto_comm_init_local on DISTINCT devices: ncclCommInitAll); collectives are ncclBroadcast /
//          ncclAllReduce over xGMI;
//   LOCAL  host threads of one process whose device list names a device more than once (logical ranks: two workers on one GPU
//          -- what a one-GPU box can run of the multi-rank code, and a way to run two chunk pipelines side by side): RCCL
//          refuses duplicate devices, so the ranks meet in a mutex-protected group and the "broadcast" is a device-to-device
//          (or peer) copy out of the root's buffer.
// Every collective is entered by every rank whatever went wrong on it before (a rank that returned early would leave its peers
// blocked), so nothing on the way INTO a collective may fail: the 64-byte device word the small collectives travel through is
// allocated when the communicator is made.
#include "cyto_common.h"
#include <rccl/rccl.h>
#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <mutex>
#include <new>
#include <set>
#include <vector>

namespace cyto {

namespace {

struct LocalGroup {
    std::mutex m;
    std::condition_variable cv;
    int nranks = 0, refs = 0;
    int arrived = 0;
    uint64_t gen = 0;
    int acc = INT_MIN;
    int result[2] = {0, 0};
    bool aborted = false;
    const void *src = nullptr;         // published by the root of a device broadcast
    int src_dev = 0;
    int32_t words[16] = {};            // published by the root of a host-word broadcast
};

struct Comm;
// RCCL communicators made together in ONE process (cyto_comm_init_local on distinct devices): a rank that fails aborts ALL of them --
// ncclCommAbort of its own communicator alone does not release the peers that already sit in a collective (their kernels wait for a
// rank that will never arrive), and they all live in this process.
struct RcclGroup {
    std::mutex m;
    std::vector<Comm *> members;
    int refs = 0;
};

struct Comm {
    int kind = 0;                      // 0: RCCL, 1: LOCAL
    ncclComm_t nccl = nullptr;
    LocalGroup *grp = nullptr;
    RcclGroup *rgrp = nullptr;         // RCCL kind, in-process siblings (null: one process per GPU)
    int rank = 0, nranks = 1, device = 0;
    void *word = nullptr;              // 64 bytes on `device` (RCCL kind: the small collectives' buffer)
    std::mutex use_m;                  // RCCL kind: held while a call is ENQUEUED on `nccl` (never while its stream is waited for) and by the abort
    std::atomic<int> aborted{0};
};

// RCCL kind: enqueue one collective unless the communicator has been aborted (by this rank or, in one process, by a sibling)
template <typename F> int rccl_enqueue(Comm *c, F &&f) {
    std::lock_guard<std::mutex> lk(c->use_m);
    if (c->aborted.load() || !c->nccl) return CYTO_ERR_PEER;
    return f(c->nccl) == ncclSuccess ? CYTO_OK : CYTO_ERR_HIP;
}
// ncclCommAbort, once: the kernels of a collective under way give up, the thread waiting for its stream returns
void rccl_abort_one(Comm *c) {
    c->aborted.store(1);
    std::lock_guard<std::mutex> lk(c->use_m);
    if (c->nccl) { (void)ncclCommAbort(c->nccl); c->nccl = nullptr; }
}

constexpr int k_local_timeout_s = 900;

// LOCAL kind: every rank contributes `val`, every rank receives the maximum; also the barrier of the group.
// CYTO_ERR_PEER if a rank aborted the group or did not arrive in time.
int local_exchange(LocalGroup *g, int val, int *out) {
    std::unique_lock<std::mutex> lk(g->m);
    if (g->aborted) return CYTO_ERR_PEER;
    g->acc = std::max(g->acc, val);
    const uint64_t my = g->gen;
    if (++g->arrived == g->nranks) {
        g->result[my & 1] = g->acc;
        g->acc = INT_MIN;
        g->arrived = 0;
        g->gen++;
        g->cv.notify_all();
    } else {
        const bool ok = g->cv.wait_for(lk, std::chrono::seconds(k_local_timeout_s), [&] { return g->gen != my || g->aborted; });
        if (!ok || (g->aborted && g->gen == my)) {
            g->aborted = true;
            g->cv.notify_all();
            return CYTO_ERR_PEER;
        }
    }
    if (out) *out = g->result[my & 1];
    return CYTO_OK;
}

Comm *as_comm(void *c) { return reinterpret_cast<Comm *>(c); }

}  // namespace

int comm_device(void *comm) { return comm ? as_comm(comm)->device : -1; }
int comm_rank(void *comm) { return comm ? as_comm(comm)->rank : -1; }

// nwords (<= 16) int32 from `root` to every rank (host memory on both sides).  Always enters the collective.
int comm_bcast_words(void *comm_, int32_t *words, int nwords, int root) {
    Comm *c = as_comm(comm_);
    if (!c || !words || nwords < 1 || nwords > 16 || root < 0 || root >= c->nranks) return CYTO_ERR_BAD_ARG;   // (a caller bug: same on every rank)
    if (c->kind == 1) {
        LocalGroup *g = c->grp;
        if (c->rank == root) { std::lock_guard<std::mutex> lk(g->m); memcpy(g->words, words, sizeof(int32_t) * (size_t)nwords); }
        int rc = local_exchange(g, 0, nullptr);                                // the root's words are published
        if (rc) return rc;
        if (c->rank != root) { std::lock_guard<std::mutex> lk(g->m); memcpy(words, g->words, sizeof(int32_t) * (size_t)nwords); }
        return local_exchange(g, 0, nullptr);                                  // ... and read by everybody before the next use of the slot
    }
    int rc = CYTO_OK;
    if (hipSetDevice(c->device) != hipSuccess) rc = CYTO_ERR_HIP;
    if (!rc && hipMemcpy(c->word, words, sizeof(int32_t) * (size_t)nwords, hipMemcpyHostToDevice) != hipSuccess) rc = CYTO_ERR_HIP;
    // (entered whatever happened above: the peers are on their way in)
    { const int e = rccl_enqueue(c, [&](ncclComm_t nc) { return ncclBroadcast(c->word, c->word, (size_t)nwords, ncclInt32, root, nc, nullptr); }); rc = rc ? rc : e; }
    if (hipStreamSynchronize(nullptr) != hipSuccess) rc = rc ? rc : CYTO_ERR_HIP;
    if (c->aborted.load()) rc = CYTO_ERR_PEER;                                  // (whatever arrived is not the root's word)
    if (!rc && hipMemcpy(words, c->word, sizeof(int32_t) * (size_t)nwords, hipMemcpyDeviceToHost) != hipSuccess) rc = CYTO_ERR_HIP;
    if (rc) { (void)hipGetLastError(); }
    return rc;
}

// *val := max over the ranks' *val.  Always enters the collective.
int comm_allreduce_max(void *comm_, int *val) {
    Comm *c = as_comm(comm_);
    if (!c || !val) return CYTO_ERR_BAD_ARG;
    if (c->kind == 1) return local_exchange(c->grp, *val, val);
    int rc = CYTO_OK;
    int32_t w = *val;
    if (hipSetDevice(c->device) != hipSuccess) rc = CYTO_ERR_HIP;
    if (!rc && hipMemcpy(c->word, &w, 4, hipMemcpyHostToDevice) != hipSuccess) rc = CYTO_ERR_HIP;
    { const int e = rccl_enqueue(c, [&](ncclComm_t nc) { return ncclAllReduce(c->word, c->word, 1, ncclInt32, ncclMax, nc, nullptr); }); rc = rc ? rc : e; }
    if (hipStreamSynchronize(nullptr) != hipSuccess) rc = rc ? rc : CYTO_ERR_HIP;
    if (c->aborted.load()) rc = CYTO_ERR_PEER;
    if (!rc && hipMemcpy(&w, c->word, 4, hipMemcpyDeviceToHost) != hipSuccess) rc = CYTO_ERR_HIP;
    if (rc) { (void)hipGetLastError(); return rc; }
    *val = w;
    return CYTO_OK;
}

// `bytes` of device memory from the root's dev_buf into every other rank's dev_buf; returns when this rank's part is done.
int comm_bcast_dev(void *comm_, void *dev_buf, size_t bytes, int root, hipStream_t stream) {
    Comm *c = as_comm(comm_);
    if (!c || !dev_buf || root < 0 || root >= c->nranks) return CYTO_ERR_BAD_ARG;
    if (c->kind == 1) {
        LocalGroup *g = c->grp;
        int mine = CYTO_OK;
        if (hipSetDevice(c->device) != hipSuccess) mine = CYTO_ERR_HIP;
        if (c->rank == root) {
            // what the root wrote (on `stream`) must be complete before another thread's stream reads it
            if (!mine && hipStreamSynchronize(stream) != hipSuccess) mine = CYTO_ERR_HIP;
            std::lock_guard<std::mutex> lk(g->m);
            g->src = dev_buf; g->src_dev = c->device;
        }
        int rc = local_exchange(g, 0, nullptr);
        if (rc) return rc;
        if (c->rank != root && !mine) {
            const void *src; int sdev;
            { std::lock_guard<std::mutex> lk(g->m); src = g->src; sdev = g->src_dev; }
            hipError_t e = sdev == c->device ? hipMemcpyAsync(dev_buf, src, bytes, hipMemcpyDeviceToDevice, stream)
                                             : hipMemcpyPeerAsync(dev_buf, c->device, src, sdev, bytes, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) { set_hip_error(e, "local broadcast copy"); (void)hipGetLastError(); mine = CYTO_ERR_HIP; }
        }
        int worst = mine;
        rc = local_exchange(g, mine, &worst);          // the root's buffer is free again; everybody learns of a failed copy
        if (rc) return rc;
        return mine ? mine : (worst ? CYTO_ERR_PEER : CYTO_OK);
    }
    int rc = CYTO_OK;
    if (hipSetDevice(c->device) != hipSuccess) rc = CYTO_ERR_HIP;
    { const int e = rccl_enqueue(c, [&](ncclComm_t nc) { return ncclBroadcast(dev_buf, dev_buf, bytes, ncclChar, root, nc, stream); }); rc = rc ? rc : e; }
    if (hipStreamSynchronize(stream) != hipSuccess) rc = rc ? rc : CYTO_ERR_HIP;
    if (c->aborted.load()) rc = CYTO_ERR_PEER;
    if (rc) (void)hipGetLastError();
    return rc;
}

}  // namespace cyto

extern "C" {

using cyto::Comm;
using cyto::LocalGroup;

int cyto_comm_unique_id(char *id128) {
    if (!id128) return CYTO_ERR_BAD_ARG;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return CYTO_ERR_HIP;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, 128);
    return CYTO_OK;
}

// One process per GPU: rank `rank` of `nranks`, on device_id; id128 from cyto_comm_unique_id on one rank.
int cyto_comm_init(const char *id128, int rank, int nranks, int device_id, void **comm_out) {
    if (!id128 || !comm_out || nranks <= 0 || rank < 0 || rank >= nranks) return CYTO_ERR_BAD_ARG;
    *comm_out = nullptr;
    int rc = cyto::select_device(device_id);
    if (rc) return rc;
    Comm *c = new (std::nothrow) Comm();
    if (!c) return CYTO_ERR_NOMEM;
    c->rank = rank; c->nranks = nranks; c->device = device_id;
    if (hipMalloc(&c->word, 64) != hipSuccess) { (void)hipGetLastError(); delete c; return CYTO_ERR_NOMEM; }
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    if (ncclCommInitRank(&c->nccl, nranks, id, rank) != ncclSuccess) { (void)hipFree(c->word); delete c; return CYTO_ERR_HIP; }
    *comm_out = c;
    return CYTO_OK;
}

// One process, one host thread per (logical) rank: comms_out[r] is rank r's handle, on device_ids[r].  Distinct devices:
// RCCL (ncclCommInitAll).  A device named more than once: the LOCAL kind (see the head of this file).
int cyto_comm_init_local(int nranks, const int *device_ids, void **comms_out) {
    if (nranks <= 0 || nranks > 64 || !device_ids || !comms_out) return CYTO_ERR_BAD_ARG;
    for (int r = 0; r < nranks; r++) comms_out[r] = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); return CYTO_ERR_NO_DEVICE; }
    bool distinct = true;
    try {
        std::set<int> seen;
        for (int r = 0; r < nranks; r++) {
            if (device_ids[r] < 0 || device_ids[r] >= count) return CYTO_ERR_BAD_ARG;
            distinct = seen.insert(device_ids[r]).second && distinct;
        }
    } catch (...) { return CYTO_ERR_NOMEM; }
    std::vector<Comm *> cs;
    LocalGroup *grp = nullptr;
    cyto::RcclGroup *rgrp = nullptr;
    int rc = CYTO_OK;
    try {
        cs.assign((size_t)nranks, nullptr);
        // (CYTO_COMM_FORCE_RCCL: developer knob of the test-suite -- a lone rank through ncclCommInitAll, so that a one-GPU box runs the
        //  RCCL kind's enqueue / abort paths)
        const bool local_kind = !distinct || (nranks == 1 && !(CYTO_KNOB("CYTO_COMM_FORCE_RCCL").set && CYTO_KNOB("CYTO_COMM_FORCE_RCCL").value));
        if (local_kind) { grp = new LocalGroup(); grp->nranks = nranks; grp->refs = nranks; }
        for (int r = 0; r < nranks && !rc; r++) {
            Comm *c = cs[(size_t)r] = new Comm();
            c->kind = grp ? 1 : 0; c->grp = grp; c->rank = r; c->nranks = nranks; c->device = device_ids[r];
            if (hipSetDevice(c->device) != hipSuccess || hipMalloc(&c->word, 64) != hipSuccess) { (void)hipGetLastError(); c->word = nullptr; rc = CYTO_ERR_NOMEM; }
        }
        if (!rc && !grp) {
            std::vector<ncclComm_t> nc((size_t)nranks);
            rgrp = new cyto::RcclGroup();
            rgrp->members = cs; rgrp->refs = nranks;
            if (ncclCommInitAll(nc.data(), nranks, device_ids) != ncclSuccess) rc = CYTO_ERR_HIP;
            else for (int r = 0; r < nranks; r++) { cs[(size_t)r]->nccl = nc[(size_t)r]; cs[(size_t)r]->rgrp = rgrp; }
        }
    } catch (...) { rc = CYTO_ERR_NOMEM; }
    if (rc) {
        for (Comm *c : cs) if (c) { if (c->word) { (void)hipSetDevice(c->device); (void)hipFree(c->word); } delete c; }
        delete grp;
        delete rgrp;
        return rc;
    }
    for (int r = 0; r < nranks; r++) comms_out[r] = cs[(size_t)r];
    return CYTO_OK;
}

// Ranks the communicator spans (RCCL: ncclCommCount -- what RCCL itself saw).
int cyto_comm_count(void *comm, int *nranks_out) {
    Comm *c = cyto::as_comm(comm);
    if (!c || !nranks_out) return CYTO_ERR_BAD_ARG;
    if (c->kind == 1) { *nranks_out = c->grp->nranks; return CYTO_OK; }
    int n = 0;
    if (ncclCommCount(c->nccl, &n) != ncclSuccess) return CYTO_ERR_HIP;
    *nranks_out = n;
    return CYTO_OK;
}

// 0: RCCL, 1: LOCAL
int cyto_comm_kind(void *comm, int *kind_out) {
    if (!comm || !kind_out) return CYTO_ERR_BAD_ARG;
    *kind_out = cyto::as_comm(comm)->kind;
    return CYTO_OK;
}

int cyto_comm_bcast_f32(void *comm, float *dev_buf, size_t count, int root, int device_id, void *stream_) {
    if (!comm || !dev_buf) return CYTO_ERR_BAD_ARG;
    (void)device_id;                                   // (the communicator knows its device)
    return cyto::comm_bcast_dev(comm, dev_buf, count * 4, root, reinterpret_cast<hipStream_t>(stream_));
}

// *status := the largest status any rank brought (0 = every rank is fine): how the ranks agree to enter -- or to skip -- a data
// collective together.  Every rank must call it.
int cyto_comm_agree(void *comm, int *status) {
    return cyto::comm_allreduce_max(comm, status);
}

// A rank that cannot reach a collective its peers are (or will be) waiting in calls this instead: LOCAL kind -- the waiting
// ranks return CYTO_ERR_PEER; RCCL kind -- ncclCommAbort of this rank's communicator AND, when the ranks are host threads of this
// process (cyto_comm_init_local), of every sibling's: a peer already inside ncclBroadcast / ncclAllReduce is released only by the
// abort of ITS communicator (its kernel polls that flag), and returns CYTO_ERR_PEER; one process per GPU: the launcher's job
// (the peers' own time-outs).  Every later collective on an aborted handle returns CYTO_ERR_PEER at once.  The handle stays valid
// for cyto_comm_destroy only.
int cyto_comm_abort(void *comm) {
    Comm *c = cyto::as_comm(comm);
    if (!c) return CYTO_ERR_BAD_ARG;
    if (c->kind == 1) {
        std::lock_guard<std::mutex> lk(c->grp->m);
        c->grp->aborted = true;
        c->grp->cv.notify_all();
        return CYTO_OK;
    }
    if (c->rgrp) {
        std::lock_guard<std::mutex> lk(c->rgrp->m);
        for (Comm *m : c->rgrp->members) if (m) m->aborted.store(1);          // (nobody enters another collective ...)
        for (Comm *m : c->rgrp->members) if (m) cyto::rccl_abort_one(m);      // (... and the ones under way give up)
        return CYTO_OK;
    }
    cyto::rccl_abort_one(c);
    return CYTO_OK;
}

// 1 once the communicator (or, in one process, a sibling) has been aborted
int cyto_comm_aborted(void *comm, int *aborted_out) {
    Comm *c = cyto::as_comm(comm);
    if (!c || !aborted_out) return CYTO_ERR_BAD_ARG;
    if (c->kind == 1) { std::lock_guard<std::mutex> lk(c->grp->m); *aborted_out = c->grp->aborted ? 1 : 0; }
    else *aborted_out = c->aborted.load() ? 1 : 0;
    return CYTO_OK;
}

int cyto_comm_destroy(void *comm) {
    Comm *c = cyto::as_comm(comm);
    if (!c) return CYTO_OK;
    if (c->word) { (void)hipSetDevice(c->device); (void)hipFree(c->word); }
    if (c->kind == 1) {
        bool last;
        { std::lock_guard<std::mutex> lk(c->grp->m); last = --c->grp->refs == 0; }
        if (last) delete c->grp;
    } else {
        if (c->rgrp) {
            bool last;
            {
                std::lock_guard<std::mutex> lk(c->rgrp->m);
                for (Comm *&m : c->rgrp->members) if (m == c) m = nullptr;
                last = --c->rgrp->refs == 0;
            }
            if (last) delete c->rgrp;
        }
        if (c->nccl) (void)ncclCommDestroy(c->nccl);
    }
    delete c;
    return CYTO_OK;
}

}  // extern "C"
