// core.hip -- status strings, device selection and memory plumbing of the C ABI (include/cytohip.h).
#include "cyto_common.h"
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <set>
#include <vector>

namespace cyto {

static thread_local char g_hip_err[512] = "";

void set_hip_error(hipError_t e, const char *what) {
    snprintf(g_hip_err, sizeof g_hip_err, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
}

Knob read_knob(const char *name) {
    const char *e = getenv(name);
    Knob k;
    k.set = e != nullptr && *e != 0;
    k.value = k.set ? atoi(e) : 0;
    return k;
}

int select_device(int device_id) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_hip_error(e, "hipGetDeviceCount");
        return CYTO_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= count) return CYTO_ERR_BAD_ARG;
    CYTO_HIP(hipSetDevice(device_id));
    return CYTO_OK;
}

int device_cus(int device_id) {
    static std::mutex m;
    static std::map<int, int> known;
    std::lock_guard<std::mutex> lk(m);
    auto it = known.find(device_id);
    if (it != known.end()) return it->second;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
    known[device_id] = cus;
    return cus;
}

// ---- device block cache ----
namespace {
struct Block { void *p; size_t bytes; };
struct DeviceCache {
    std::multimap<size_t, void *> free_blocks;     // size -> block
    std::map<void *, size_t> live;                 // every block this cache handed out (or holds): its size
    size_t free_bytes = 0;                         // sum over free_blocks
};
std::mutex g_cache_mutex;
std::map<int, DeviceCache> g_cache;                // by device

// Blocks of up to 16 GiB are kept (config c3's standardised scRNA operand is 4 GB: handing it back to the runtime after every call and
// asking for it again cost the fused call ~10 ms of hipFree / hipMalloc -- round 6, tools/c3_walls.py), as long as the blocks the cache
// holds idle stay under 64 GiB of the device's 288 (beyond that the largest idle blocks go back to the runtime; a materialised 10 GB cost
// matrix of a one-off solve does not starve the caller's own allocations for long).
constexpr size_t k_keep_max = (size_t)16 << 30;
constexpr size_t k_idle_cap = (size_t)64 << 30;

size_t round_size(size_t b) {
    // 256-byte granules below 1 MiB, 1/8-octave steps above: a slightly larger n finds the block of the last solve
    if (b <= (1u << 20)) return (b + 255) & ~(size_t)255;
    size_t step = (size_t)1 << 17;
    while ((step << 4) < b) step <<= 1;
    return (b + step - 1) / step * step;
}

void trim_locked(DeviceCache &c) {
    for (auto &kv : c.free_blocks) { c.live.erase(kv.second); (void)hipFree(kv.second); }
    c.free_blocks.clear();
    c.free_bytes = 0;
}
}  // namespace

void *cache_alloc(size_t bytes, int *status) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { *status = CYTO_ERR_HIP; return nullptr; }
    const size_t want = round_size(bytes);
    {
        std::lock_guard<std::mutex> lk(g_cache_mutex);
        DeviceCache &c = g_cache[dev];
        auto it = c.free_blocks.lower_bound(want);
        // best fit, but never a block more than 25 % larger than asked for (a 10 GB block must not serve a 1 MB request)
        if (it != c.free_blocks.end() && it->first <= want + want / 4 + 4096) {
            void *p = it->second;
            c.free_bytes -= it->first;
            c.free_blocks.erase(it);
            *status = CYTO_OK;
            return p;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipErrorOutOfMemory) {                 // give the cached blocks back and try once more
        (void)hipGetLastError();
        { std::lock_guard<std::mutex> lk(g_cache_mutex); trim_locked(g_cache[dev]); }
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) {
        set_hip_error(e, "hipMalloc");
        *status = e == hipErrorOutOfMemory ? CYTO_ERR_NOMEM : CYTO_ERR_HIP;
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_cache_mutex);
    g_cache[dev].live[p] = want;
    *status = CYTO_OK;
    return p;
}

void cache_release(void *p, hipStream_t used_on) {
    if (!p) return;
    // normal returns have synchronised their stream already (the query is then a cheap "yes"); error returns may not have
    if (hipStreamQuery(used_on) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(used_on); }
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(g_cache_mutex);
        for (auto &kv : g_cache) {                  // the owning device's cache (normally the current device)
            auto it = kv.second.live.find(p);
            if (it == kv.second.live.end()) continue;
            // blocks above 1 GiB (a materialised c3 cost matrix is 10 GB) go straight back to the runtime: the hipFree's
            // device-wide synchronisation is nothing beside the solve that used such a block, and a resident 10 GB block
            // nobody asked for starves the caller's own allocations
            // (developer knob CYTO_CACHE_KEEP_MB: the largest block kept, in MiB -- 1024 = rounds 1-5)
            const size_t keep_max = CYTO_KNOB("CYTO_CACHE_KEEP_MB").set ? (size_t)std::max(1, CYTO_KNOB("CYTO_CACHE_KEEP_MB").value) << 20 : k_keep_max;
            if (it->second > keep_max) { kv.second.live.erase(it); break; }
            kv.second.free_blocks.emplace(it->second, p);
            kv.second.free_bytes += it->second;
            // over the idle cap: the largest idle blocks go back to the runtime (outside the lock, below)
            std::vector<void *> evict;
            while (kv.second.free_bytes > k_idle_cap && !kv.second.free_blocks.empty()) {
                auto big = std::prev(kv.second.free_blocks.end());
                kv.second.free_bytes -= big->first;
                kv.second.live.erase(big->second);
                evict.push_back(big->second);
                kv.second.free_blocks.erase(big);
            }
            if (evict.empty()) return;
            p = nullptr;
            for (void *q : evict) { if (!p) p = q; else (void)hipFree(q); }      // (hipFree synchronises the device: rare by construction)
            break;
        }
    }
    (void)hipFree(p);                               // a large block (or not ours: cannot happen); outside the lock
}

// ---- stream pool: non-blocking streams by device, handed out to the calls that want a private stream and taken back drained ----
namespace {
std::mutex g_stream_mutex;
std::map<int, std::vector<hipStream_t>> g_idle_streams;
std::map<hipStream_t, int> g_stream_device;
}  // namespace

hipStream_t stream_pool_acquire() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    {
        std::lock_guard<std::mutex> lk(g_stream_mutex);
        std::vector<hipStream_t> &idle = g_idle_streams[dev];
        if (!idle.empty()) { hipStream_t s = idle.back(); idle.pop_back(); return s; }
    }
    hipStream_t s = nullptr;
    const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) { set_hip_error(e, "hipStreamCreateWithFlags"); (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lk(g_stream_mutex);
    g_stream_device[s] = dev;
    return s;
}

void stream_pool_release(hipStream_t s) {
    if (!s) return;
    std::lock_guard<std::mutex> lk(g_stream_mutex);
    auto it = g_stream_device.find(s);
    if (it == g_stream_device.end()) return;                      // (not ours: cannot happen)
    std::vector<hipStream_t> &idle = g_idle_streams[it->second];
    if (idle.size() < 64) { idle.push_back(s); return; }            // (64 hardware queues at most: _lib.py sets GPU_MAX_HW_QUEUES)
    g_stream_device.erase(it);
    (void)hipStreamDestroy(s);
}

int set_max_dynamic_lds(const void *kernel) {
    // hipFuncSetAttribute acts on the CURRENT device's copy of the kernel: remembered per (device, kernel), so a process
    // that solves on device 0 and then on device 1 sets it on both
    static std::mutex m;
    static std::set<std::pair<int, const void *>> done;
    std::lock_guard<std::mutex> lk(m);
    int dev = 0;
    CYTO_HIP(hipGetDevice(&dev));
    const std::pair<int, const void *> key(dev, kernel);
    if (done.count(key)) return CYTO_OK;
    // the CU has 160 KB; what the kernel declares statically comes off the dynamic allowance
    hipFuncAttributes fa;
    CYTO_HIP(hipFuncGetAttributes(&fa, kernel));
    const int dyn = 160 * 1024 - (int)((fa.sharedSizeBytes + 255) & ~(size_t)255);
    if (dyn < LDS_DYNAMIC_MAX) return CYTO_ERR_INTERNAL;      // a kernel's static LDS outgrew the 2 KB the planners leave for it
    CYTO_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
    done.insert(key);
    return CYTO_OK;
}

}  // namespace cyto

extern "C" {

int cyto_trim_device_cache(int device_id) {
    int rc = cyto::select_device(device_id);
    if (rc) return rc;
    CYTO_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(cyto::g_cache_mutex);
    auto it = cyto::g_cache.find(device_id);
    if (it != cyto::g_cache.end()) cyto::trim_locked(it->second);
    return CYTO_OK;
}

const char *cyto_strerror(int status) {
    switch (status) {
        case CYTO_OK: return "ok";
        case CYTO_ERR_BAD_ARG: return "bad argument (null pointer, non-square or non-positive size)";
        case CYTO_ERR_NONFINITE: return "cost matrix contains NaN or Inf";
        case CYTO_ERR_NOMEM: return "out of host or device memory";
        case CYTO_ERR_INTERNAL: return "internal solver error";
        case CYTO_ERR_HIP: return "HIP runtime error";
        case CYTO_ERR_NO_DEVICE: return "no HIP device available";
        case CYTO_ERR_UNSUPPORTED: return "problem size not supported by this build";
        case CYTO_ERR_SHAPE: return "the two matrices must have the same number of genes (rows)";
        case CYTO_ERR_PEER: return "another rank of the communicator failed, aborted or did not arrive";
        default: return "unknown status";
    }
}

const char *cyto_last_hip_error(void) { return cyto::g_hip_err; }

const char *cyto_version(void) { return "cytohip 0.2.0 (gfx950)"; }

int cyto_abi_sizes(size_t *lap_info, size_t *lap_opts, size_t *assign_info, size_t *chunk) {
    if (lap_info) *lap_info = sizeof(cyto_lap_info);
    if (lap_opts) *lap_opts = sizeof(cyto_lap_opts);
    if (assign_info) *assign_info = sizeof(cyto_assign_info);
    if (chunk) *chunk = sizeof(cyto_chunk);
    return CYTO_OK;
}

int cyto_device_count(int *count) {
    if (!count) return CYTO_ERR_BAD_ARG;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *count = 0; cyto::set_hip_error(e, "hipGetDeviceCount"); return CYTO_ERR_NO_DEVICE; }
    *count = c;
    return CYTO_OK;
}

int cyto_device_name(int device_id, char *buf, size_t buflen) {
    if (!buf || buflen == 0) return CYTO_ERR_BAD_ARG;
    hipDeviceProp_t prop;
    CYTO_HIP(hipGetDeviceProperties(&prop, device_id));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return CYTO_OK;
}

int cyto_malloc(void **dptr, size_t bytes, int device_id) {
    if (!dptr) return CYTO_ERR_BAD_ARG;
    int rc = cyto::select_device(device_id);
    if (rc) return rc;
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 16);
    if (e == hipErrorOutOfMemory) {                 // the solver's cached work blocks are reclaimable: give them back, retry
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
        { std::lock_guard<std::mutex> lk(cyto::g_cache_mutex); cyto::trim_locked(cyto::g_cache[device_id]); }
        e = hipMalloc(dptr, bytes ? bytes : 16);
    }
    if (e != hipSuccess) {
        *dptr = nullptr;
        (void)hipGetLastError();
        cyto::set_hip_error(e, "hipMalloc");
        return e == hipErrorOutOfMemory ? CYTO_ERR_NOMEM : CYTO_ERR_HIP;
    }
    return CYTO_OK;
}

int cyto_free(void *dptr, int device_id) {
    int rc = cyto::select_device(device_id);
    if (rc) return rc;
    if (dptr) CYTO_HIP(hipFree(dptr));
    return CYTO_OK;
}

int cyto_memcpy_h2d(void *dst, const void *src, size_t bytes, int device_id) {
    if (!dst || !src) return CYTO_ERR_BAD_ARG;
    int rc = cyto::select_device(device_id);
    if (rc) return rc;
    CYTO_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return CYTO_OK;
}

int cyto_memcpy_d2h(void *dst, const void *src, size_t bytes, int device_id) {
    if (!dst || !src) return CYTO_ERR_BAD_ARG;
    int rc = cyto::select_device(device_id);
    if (rc) return rc;
    CYTO_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return CYTO_OK;
}

int cyto_memcpy_d2d(void *dst, const void *src, size_t bytes, int device_id) {
    if (!dst || !src) return CYTO_ERR_BAD_ARG;
    int rc = cyto::select_device(device_id);
    if (rc) return rc;
    CYTO_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice));
    return CYTO_OK;
}

int cyto_device_synchronize(int device_id) {
    int rc = cyto::select_device(device_id);
    if (rc) return rc;
    CYTO_HIP(hipDeviceSynchronize());
    return CYTO_OK;
}

}  // extern "C"
