// shader clock vs wall clock: how fast does the SQ actually tick while short kernels run back to back?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_spin(long long ticks, long long *out)
{
    long long t0 = clock64(), r0 = wall_clock64();
    while (clock64() - t0 < ticks) {}
    long long t1 = clock64(), r1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}
__global__ void k_busy(float *p, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float a = p[i % n];
    for (int j = 0; j < 2000; ++j) a = a * 1.0001f + 0.5f;
    p[i % n] = a;
}
int main()
{
    long long *out; hipMalloc(&out, 16);
    float *p; hipMalloc(&p, 1 << 22);
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    int sclk = 0; hipDeviceGetAttribute(&sclk, hipDeviceAttributeClockRate, 0);
    printf("wall clock rate %d kHz, nominal shader clock %d kHz\n", rate, sclk);
    for (int phase = 0; phase < 3; ++phase) {
        if (phase == 1) for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_busy, 4096, 256, 0, 0, p, 1 << 20);
        if (phase == 2) { hipDeviceSynchronize(); for (int i = 0; i < 200; ++i) { hipLaunchKernelGGL(k_busy, 4096, 256, 0, 0, p, 1 << 20); hipDeviceSynchronize(); } }
        hipLaunchKernelGGL(k_spin, 1, 64, 0, 0, 2000000LL, out);
        long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        printf("phase %d: %lld shader ticks in %lld wall ticks -> %.1f MHz\n", phase, h[0], h[1], (double)h[0] / ((double)h[1] / rate) / 1000.0);
    }
    return 0;
}
