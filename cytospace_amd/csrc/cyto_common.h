// cyto_common.h -- shared host-side helpers of libcytohip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/cytohip.h"

namespace cyto {

void set_hip_error(hipError_t e, const char *what);

#define CYTO_HIP(expr)                                       \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) {                              \
            ::cyto::set_hip_error(_e, #expr);                \
            return _e == hipErrorOutOfMemory ? CYTO_ERR_NOMEM : CYTO_ERR_HIP; \
        }                                                    \
    } while (0)

// Device memory comes from a per-device cache of blocks (core.hip): a solve takes its ~15 work buffers from blocks that
// earlier solves returned, so the steady state has no hipMalloc / hipFree at all -- hipFree synchronises the whole device,
// which serialised the chunk solves that run side by side on their own streams.  A block is handed back only when the
// stream it was used on has drained (checked with hipStreamQuery, waited for on error paths), so a new owner on another
// stream never races with work still in flight.  cyto_trim_device_cache() gives everything back to the runtime.
void *cache_alloc(size_t bytes, int *status);
void cache_release(void *p, hipStream_t used_on);

// RAII device buffer -- host-side convenience only.
struct DevBuf {
    void *p = nullptr;
    hipStream_t stream = nullptr;          // the stream the buffer is used on (for the drain check at release)
    bool owned = true;                     // false: a VIEW into another buffer's block (a solve's work buffers are carved out of one block)
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { reset(); }
    void reset() { if (p && owned) cache_release(p, stream); p = nullptr; owned = true; }
    int alloc(size_t bytes, hipStream_t used_on = nullptr) {
        reset();
        int st = CYTO_OK;
        stream = used_on;
        p = cache_alloc(bytes ? bytes : 16, &st);
        return st;
    }
    void view(void *q, hipStream_t used_on = nullptr) { reset(); p = q; stream = used_on; owned = false; }
    template <typename U> U *as() const { return reinterpret_cast<U *>(p); }
};

// RAII events (timing brackets of one call); destroyed on every return path.
template <int N> struct Events {
    hipEvent_t e[N] = {};
    int made = 0;
    ~Events() { for (int i = 0; i < made; i++) (void)hipEventDestroy(e[i]); }
    int create() {
        for (; made < N; made++) CYTO_HIP(hipEventCreate(&e[made]));
        return CYTO_OK;
    }
    hipEvent_t operator[](int i) const { return e[i]; }
};

// A private non-blocking stream for one call when the caller gave none.  The stream is drained before it is destroyed: work that
// an early return left queued must not outlive the call.
// (round 6: the stream comes from a per-device pool and goes back to it -- creating and destroying a stream per call cost a chunk call's
//  LAP ~7 ms, tools/c3_walls.py: the first launches on a fresh stream wait for its hardware queue)
hipStream_t stream_pool_acquire();            // a non-blocking stream of the CURRENT device (created if none is idle); nullptr on failure
void stream_pool_release(hipStream_t s);     // back to the pool of the device it was made on (the caller has drained it)
struct StreamGuard {
    hipStream_t s = nullptr;
    bool own = false;
    int acquire() {                            // CYTO_OK / CYTO_ERR_HIP
        s = stream_pool_acquire();
        own = s != nullptr;
        return own ? CYTO_OK : CYTO_ERR_HIP;
    }
    ~StreamGuard() { if (own && s) { (void)hipStreamSynchronize(s); stream_pool_release(s); } }
};
// Declared AFTER the device buffers a second stream works on (so destroyed BEFORE them): whatever path leaves the scope, the
// stream has drained before a buffer goes back to the block cache, where another thread's call may be handed it.
struct StreamDrain {
    hipStream_t s = nullptr;
    ~StreamDrain() { if (s) (void)hipStreamSynchronize(s); }
};

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-symbol, process-wide setting: set it ONCE per kernel to the most
// the CU offers (160 KB minus the kernel's static LDS) instead of before every launch with a size that depends on n
// (two host threads launching different sizes of the same kernel would race).
constexpr int LDS_DYNAMIC_MAX = 160 * 1024 - 2048;
int set_max_dynamic_lds(const void *kernel);

int select_device(int device_id);
int device_cus(int device_id);          // multiProcessorCount, queried once per device (256 if the query fails)

// lap_jv.hip: float32 problems of ONE size solved as a batch (one launch per chain phase, a workgroup per problem)
int lap_batch_same_n(int n, int nb, const float *const *cost, const int64_t *ld, int cost_on_device, int32_t *const *rowsol,
                     int32_t *const *colsol, float *const *u, float *const *v, double *total, cyto_lap_info *info, int *status,
                     int device_id, hipStream_t stream, const int32_t *const *rowmap = nullptr, const int *nu = nullptr,
                     const cyto_lap_opts *opts = nullptr);
// batch.hip: cyto_lap_batch_f32 with optional row maps
int lap_batch_any(int nb, const int *n, const float *const *cost, const int64_t *ld, int cost_on_device,
                  const int32_t *const *rowmap, const int *nu, int32_t *const *rowsol, int32_t *const *colsol, float *const *u,
                  float *const *v, double *total, cyto_lap_info *info, int *status_out, int max_concurrent, int device_id,
                  const cyto_lap_opts *opts = nullptr);
// comm.hip: the communicator's collectives as the context builder uses them (every rank enters every one of them, whatever
// went wrong on it before)
int comm_device(void *comm);
int comm_rank(void *comm);
int comm_bcast_words(void *comm, int32_t *words, int nwords, int root);          // <= 16 host int32 from root to all
int comm_allreduce_max(void *comm, int *val);                                    // *val := max over ranks
int comm_bcast_dev(void *comm, void *dev_buf, size_t bytes, int root, hipStream_t stream);

// Developer knobs (environment variables of the tools/ scripts): read ONCE per process, at first use -- no libc call per solve, and
// a knob cannot change between two solves of one process.
struct Knob { int value; bool set; };
Knob read_knob(const char *name);
#define CYTO_KNOB(name) ([]() -> const ::cyto::Knob & { static const ::cyto::Knob k = ::cyto::read_knob(name); return k; }())
}  // namespace cyto
