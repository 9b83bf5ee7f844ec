// core.hip -- status strings, device selection and memory plumbing of the C ABI (include/cytohip.h).
#include "cyto_common.h"

namespace cyto {

static thread_local char g_hip_err[512] = "";

void set_hip_error(hipError_t e, const char *what) {
    snprintf(g_hip_err, sizeof g_hip_err, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
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

}  // namespace cyto

extern "C" {

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
        default: return "unknown status";
    }
}

const char *cyto_last_hip_error(void) { return cyto::g_hip_err; }

const char *cyto_version(void) { return "cytohip 0.1.0 (gfx950)"; }

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
    CYTO_HIP(hipMalloc(dptr, bytes ? bytes : 16));
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

int cyto_device_synchronize(int device_id) {
    int rc = cyto::select_device(device_id);
    if (rc) return rc;
    CYTO_HIP(hipDeviceSynchronize());
    return CYTO_OK;
}

}  // extern "C"
