// How fast is a pitched host-to-device copy out of PAGEABLE memory (a block of columns of a row-major matrix)?
// build: hipcc -O2 -o tools/dbg/h2d_2d_bench tools/dbg/h2d_2d_bench.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
int main() {
    const size_t G = 20000, C = 50000, W = 8192;
    float *h = (float *)malloc(G * C * 4);
    memset(h, 1, G * C * 4);
    float *d; hipMalloc(&d, G * C * 4);
    hipStream_t s; hipStreamCreate(&s);
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int rep = 0; rep < 2; rep++) {
        double t = now();
        hipMemcpy(d, h, G * C * 4, hipMemcpyHostToDevice);
        double dt = now() - t;
        printf("contiguous %.2f GB in %.1f ms = %.1f GB/s\n", G * C * 4 / 1e9, dt * 1e3, G * C * 4 / dt / 1e9);
        t = now();
        for (size_t c0 = 0; c0 < C; c0 += W) {
            const size_t w = (c0 + W <= C ? W : C - c0);
            hipMemcpy2DAsync(d + c0 * G, w * 4, h + c0, C * 4, w * 4, G, hipMemcpyHostToDevice, s);
        }
        hipStreamSynchronize(s);
        dt = now() - t;
        printf("column blocks of %zu: %.1f ms = %.1f GB/s\n", W, dt * 1e3, G * C * 4 / dt / 1e9);
    }
    return 0;
}
