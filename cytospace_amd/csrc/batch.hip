// batch.hip -- many independent LAPs on one GPU at once, and the RCCL broadcast of the shared
// standardised spot matrix.
//
// CytoSPACE splits large inputs into independent square sub-LAPs ("chunks") and ships each to a worker
// process (/root/reference/cytospace/cytospace.py:430-451).  On the GPU the sequential part of one solve
// occupies ONE workgroup (one CU of 256), so chunks are solved concurrently: one host thread + one HIP
// stream per chunk in flight; the streaming kernels (column reduction, row-cache build, cost GEMM) of
// different chunks interleave on the rest of the chip.
#include "cyto_common.h"
#include <rccl/rccl.h>
#include <atomic>
#include <thread>
#include <vector>

extern "C" {

// Solve nb independent LAPs.  Arrays of per-problem pointers/sizes; outputs may be NULL like in
// cyto_lap_f32.  max_concurrent <= 0 picks min(nb, 32).  Returns the first non-zero status (all
// problems are attempted); status_out[b] (optional) receives each problem's status.
int cyto_lap_batch_f32(int nb, const int *n, const float *const *cost, const int64_t *ld, int cost_on_device,
                       int32_t *const *rowsol, int32_t *const *colsol, float *const *u, float *const *v, double *total,
                       cyto_lap_info *info, int *status_out, int max_concurrent, int device_id) {
    if (nb < 0 || (nb > 0 && (!n || !cost || !ld))) return CYTO_ERR_BAD_ARG;
    if (nb == 0) return CYTO_OK;
    int rc = cyto::select_device(device_id);
    if (rc) return rc;
    int conc = max_concurrent > 0 ? max_concurrent : 32;
    if (conc > nb) conc = nb;
    std::atomic<int> next(0);
    std::vector<int> st((size_t)nb, CYTO_OK);
    auto worker = [&]() {
        if (hipSetDevice(device_id) != hipSuccess) return;
        hipStream_t stream = nullptr;
        if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) stream = nullptr;
        for (;;) {
            const int b = next.fetch_add(1);
            if (b >= nb) break;
            cyto::tl_single_cu_only = 1;
            st[(size_t)b] = cyto_lap_f32(n[b], cost[b], ld[b], cost_on_device, rowsol ? rowsol[b] : nullptr,
                                         colsol ? colsol[b] : nullptr, u ? u[b] : nullptr, v ? v[b] : nullptr,
                                         total ? &total[b] : nullptr, info ? &info[b] : nullptr, device_id, stream);
        }
        if (stream) (void)hipStreamDestroy(stream);
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < conc; t++) pool.emplace_back(worker);
    for (auto &t : pool) t.join();
    int first = CYTO_OK;
    for (int b = 0; b < nb; b++) {
        if (status_out) status_out[b] = st[(size_t)b];
        if (!first && st[(size_t)b]) first = st[(size_t)b];
    }
    return first;
}

// ---- RCCL (xGMI): the only collective on the path is the broadcast of the shared standardised ST
// matrix to the ranks that solve chunks against it. ----
int cyto_comm_unique_id(char *id128) {
    if (!id128) return CYTO_ERR_BAD_ARG;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return CYTO_ERR_HIP;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, 128);
    return CYTO_OK;
}

int cyto_comm_init(const char *id128, int rank, int nranks, int device_id, void **comm_out) {
    if (!id128 || !comm_out || nranks <= 0 || rank < 0 || rank >= nranks) return CYTO_ERR_BAD_ARG;
    int rc = cyto::select_device(device_id);
    if (rc) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclComm_t comm;
    if (ncclCommInitRank(&comm, nranks, id, rank) != ncclSuccess) return CYTO_ERR_HIP;
    *comm_out = comm;
    return CYTO_OK;
}

int cyto_comm_bcast_f32(void *comm, float *dev_buf, size_t count, int root, int device_id, void *stream_) {
    if (!comm || !dev_buf) return CYTO_ERR_BAD_ARG;
    int rc = cyto::select_device(device_id);
    if (rc) return rc;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (ncclBroadcast(dev_buf, dev_buf, count, ncclFloat, root, reinterpret_cast<ncclComm_t>(comm), stream) != ncclSuccess)
        return CYTO_ERR_HIP;
    CYTO_HIP(hipStreamSynchronize(stream));
    return CYTO_OK;
}

int cyto_comm_destroy(void *comm) {
    if (comm) (void)ncclCommDestroy(reinterpret_cast<ncclComm_t>(comm));
    return CYTO_OK;
}

}  // extern "C"
