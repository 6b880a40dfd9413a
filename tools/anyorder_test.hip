// Does hipExtAnyOrderLaunch let two kernels on ONE stream overlap on gfx950?  (hip_ext.h says "not supported on GFX9xx")
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long cycles, int* out)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (out) out[blockIdx.x] = 1;
}
int main()
{
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int* d; hipMalloc(&d, 4096);
    const long long cyc = 100 * 200;  // wall_clock64 = 100 MHz -> 200 us
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0, st);
            hipLaunchKernelGGL(spin, dim3(4), dim3(64), 0, st, cyc, d);
            if (mode == 0) hipLaunchKernelGGL(spin, dim3(4), dim3(64), 0, st, cyc, d + 64);
            else hipExtLaunchKernelGGL(spin, dim3(4), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d + 64);
            hipEventRecord(e1, st);
            hipStreamSynchronize(st);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("mode %s: %.1f us (%s)\n", mode ? "any-order" : "in-order", ms * 1e3, hipGetErrorString(hipGetLastError()));
        }
    }
    // fork/join over two streams with short kernels: what does the cross-stream dependency cost?
    hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t ef, ej; hipEventCreateWithFlags(&ef, hipEventDisableTiming); hipEventCreateWithFlags(&ej, hipEventDisableTiming);
    const long long c10 = 100 * 10;
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0, st);
            for (int it = 0; it < 20; it++) {
                if (mode == 0) {
                    hipLaunchKernelGGL(spin, dim3(4), dim3(64), 0, st, c10, d);
                    hipLaunchKernelGGL(spin, dim3(4), dim3(64), 0, st, c10, d + 64);
                } else {
                    hipEventRecord(ef, st);
                    hipStreamWaitEvent(s2, ef, 0);
                    hipLaunchKernelGGL(spin, dim3(4), dim3(64), 0, s2, c10, d);
                    hipLaunchKernelGGL(spin, dim3(4), dim3(64), 0, st, c10, d + 64);
                    hipEventRecord(ej, s2);
                    hipStreamWaitEvent(st, ej, 0);
                }
            }
            hipEventRecord(e1, st);
            hipStreamSynchronize(st);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("2 x 10us kernels, %s: %.1f us per pair\n", mode ? "fork/join on two streams" : "in-order", ms * 1e3 / 20);
        }
    }
    return 0;
}
