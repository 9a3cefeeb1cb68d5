// LDS atomic throughput under same-address conflicts (u64 / f64 / u32), cycles per wave instruction
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int WAYS>
__global__ void __launch_bounds__(256) k_atom(T *out, long long *cyc)
{
    __shared__ T acc[256];
    acc[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // WAYS lanes share one address; distinct groups hit distinct 8-byte slots
    T *p = &acc[(lane / WAYS) + 64 * (threadIdx.x >> 6)];
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 256; ++i) {
        atomicAdd(p, (T)1);
        atomicAdd(p, (T)1);
        atomicAdd(p, (T)1);
        atomicAdd(p, (T)1);
    }
    __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = acc[threadIdx.x];
}
template <typename T, int WAYS> void run(const char *name)
{
    T *out; long long *cyc;
    (void)hipMalloc(&out, 1024 * 256 * sizeof(T)); (void)hipMalloc(&cyc, 8);
    hipLaunchKernelGGL((k_atom<T, WAYS>), 1, 256, 0, 0, out, cyc);
    long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    // 4 waves x 1024 instructions share one LDS
    printf("%s %2d-way: %.1f cycles per wave-instruction (one CU, 4 waves)\n", name, WAYS, (double)h / 4096.0);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main()
{
    run<unsigned long long, 1>("u64"); run<unsigned long long, 4>("u64"); run<unsigned long long, 16>("u64"); run<unsigned long long, 64>("u64");
    run<double, 1>("f64"); run<double, 4>("f64"); run<double, 16>("f64"); run<double, 64>("f64");
    run<unsigned int, 1>("u32"); run<unsigned int, 16>("u32"); run<unsigned int, 64>("u32");
    return 0;
}
