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

// RAII device buffer (hipFree on scope exit) -- host-side convenience only.
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        CYTO_HIP(hipMalloc(&p, bytes ? bytes : 16));
        return CYTO_OK;
    }
    template <typename U> U *as() const { return reinterpret_cast<U *>(p); }
};

int select_device(int device_id);

// set by callers that run several solves concurrently on one GPU (cyto_lap_batch_f32): the cooperative
// augmentation spins on its peers and must not compete with other cooperative launches for CUs
extern thread_local int tl_single_cu_only;
}  // namespace cyto
