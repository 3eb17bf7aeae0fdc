// How much does a launch pay for a grid that is not a whole number of rounds of the 512 resident workgroups?  One 3x3
// 256 -> 256 convolution (64 x 256 tiles: one workgroup per 64 output pixels) on a 1 x W strip, W = 64 n pixels, for tile
// counts around 512 and 1024: time per launch and time per tile.  Torch-free (C ABI).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/tile_sweep.hip -o tools/ubench/tile_sweep -ldl
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include "../../include/lsnet_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ void fill_kernel(float *p, size_t n, unsigned seed, float scale)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 16, h *= 0x7feb352du, h ^= h >> 15, h *= 0x846ca68bu, h ^= h >> 16;
        p[i] = ((float)(h & 0xffffff) / 8388608.f - 1.f) * scale;
    }
}

int main(int argc, char **argv)
{
    void *h = dlopen(getenv("LSNET_SO") ? getenv("LSNET_SO") : "lsnet_amd/csrc/liblsnet_hip.so", RTLD_NOW);
    if (!h) { printf("dlopen: %s\n", dlerror()); return 2; }
    auto bytes_fn = (int64_t(*)(int, int, int, int, int, int, int, int))dlsym(h, "lsn_conv2d_prepared_bytes");
    auto prep_fn = (int (*)(int, const float *, void *, int, int, int, int, int, int, int, lsn_stream_t))dlsym(h, "lsn_conv2d_prepare_weights");
    auto fwd_fn = (int (*)(int, const lsn_conv_level *, const void *, const float *, int, int, int, int, int, int, int, int, int,
                           lsn_stream_t))dlsym(h, "lsn_conv2d_forward_prepared");
    if (!bytes_fn || !prep_fn || !fwd_fn) return 2;
    const int C = 256, Co = argc > 1 ? atoi(argv[1]) : 256, k = 3, Hh = 8;
    const int tiles[] = {256, 384, 480, 512, 520, 525, 544, 640, 700, 768, 896, 1000, 1024, 1050, 1100, 1280, 1536, 2048};
    float *w, *x, *y;
    void *img;
    CK(hipMalloc(&w, (size_t)Co * k * k * C * 4));
    CK(hipMalloc(&img, (size_t)bytes_fn(0, C, Co, k, k, 1, 1, 1)));
    const size_t maxpx = (size_t)2048 * 64;
    CK(hipMalloc(&x, maxpx * C * 4));
    CK(hipMalloc(&y, maxpx * Co * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, 0, w, (size_t)Co * k * k * C, 1u, 0.05f);
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, x, maxpx * C, 2u, 1.f);
    if (prep_fn(0, w, img, C, Co, k, k, 1, 1, 1, nullptr)) return 3;
    printf("3x3 %d -> %d, H = %d rows; 64-pixel tiles x %d column block(s)\n%8s %10s %10s %8s\n", C, Co, Hh, (Co + 255) / 256, "tiles",
           "us", "us/tile", "TF");
    for (int nt : tiles) {
        const int W = nt * 64 / Hh;
        lsn_conv_level lv = {};
        lv.x = x, lv.out = y, lv.B = 1, lv.H = Hh, lv.W = W;
        auto run = [&] { if (fwd_fn(1, &lv, img, nullptr, C, C, Co, k, k, 1, 1, 1, 0, nullptr)) exit(3); };
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) run();
        CK(hipEventRecord(e0, 0));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) run();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1000.0 / reps;
        printf("%8d %10.1f %10.3f %8.1f\n", nt, us, us / nt, 2.0 * nt * 64 * C * Co * k * k / us * 1e-6);
    }
    return 0;
}
