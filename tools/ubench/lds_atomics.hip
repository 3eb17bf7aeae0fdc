// Micro-benchmark: cycles per wave-instruction of LDS atomics (ds_add_f32 / ds_add_u32 / ds_add_u64 / plain RMW)
// on a [160][32] window, 4 waves per block, addresses spread like the DCN backward scatter (16 channel lanes x 4 rows).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(long long *out, float *sink, int iters)
{
    __shared__ float win[160 * 32];
    __shared__ unsigned long long win64[160 * 32 / 2];
    for (int i = threadIdx.x; i < 160 * 32; i += 256) win[i] = 0.f;
    for (int i = threadIdx.x; i < 160 * 16; i += 256) win64[i] = 0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63, j16 = lane & 15, kq = lane >> 4;
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        seed = seed * 1664525u + 1013904223u;
        const int row = ((it * 7 + kq * 3 + (threadIdx.x >> 6) * 11) % 150) + ((seed >> 28) & 1);   // per 16-lane group
        const int rowu = __shfl(row, lane & 48);
        const int idx = rowu * 32 + (j16 ^ ((rowu & 1) << 4));
        const float v = (float)(seed >> 8) * 1e-9f;
        if (MODE == 0) unsafeAtomicAdd(&win[idx], v);
        if (MODE == 1) atomicAdd(reinterpret_cast<unsigned *>(win) + idx, (unsigned)(seed >> 8));
        if (MODE == 2) atomicAdd(&win64[idx >> 1], (unsigned long long)(seed >> 8));
        if (MODE == 3) win[idx] += v;
        if (MODE == 4) atomicAdd(&win[idx], v);
    }
    __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    float s = 0.f;
    for (int i = threadIdx.x; i < 160 * 32; i += 256) s += win[i] + (float)win64[i >> 1];
    if (s == 12345.f) sink[0] = s;
}

int main()
{
    long long *d; float *sink;
    const int blocks = 512, iters = 512;
    CHECK(hipMalloc(&d, blocks * sizeof(long long)));
    CHECK(hipMalloc(&sink, 4));
    const char *names[] = {"ds_add_f32 (unsafeAtomicAdd)", "ds_add_u32", "ds_add_u64", "plain RMW (racy)", "atomicAdd(float) default"};
    for (int m = 0; m < 5; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, sink, iters);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, sink, iters);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, sink, iters);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, sink, iters);
            if (m == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, sink, iters);
            CHECK(hipDeviceSynchronize());
        }
        std::vector<long long> h(blocks);
        CHECK(hipMemcpy(h.data(), d, blocks * sizeof(long long), hipMemcpyDeviceToHost));
        double s = 0; for (auto v : h) s += v;
        printf("%-32s %8.1f cycles per wave-instruction (4 waves/block, 2 blocks/CU)\n", names[m], s / blocks / iters);
    }
    return 0;
}
