// Forward and data gradient of every dense-convolution shape of the benchmark step (B = 2,
// 800 x 1344) through the C ABI with prepared weight images -- no torch, the binary starts in a second.  Per shape: time
// per launch (HIP events), TFLOP/s, and the results checked against a double-precision sum on the host for sampled
// elements; at the end the sums weighted by the shape's count in one step.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/conv_step.hip -o tools/ubench/conv_step -ldl
//   tools/ubench/conv_step [reps]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/lsnet_hip.h"

static inline int ck_(hipError_t e, const char *file, int line)
{
    if (e != hipSuccess) {
        printf("HIP error %s at %s:%d\n", hipGetErrorString(e), file, line);
        exit(2);
    }
    return 0;
}
#define CK(x) ck_((x), __FILE__, __LINE__)

__global__ void fill_kernel(float *p, size_t n, unsigned seed, float scale)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 16, h *= 0x7feb352du, h ^= h >> 15, h *= 0x846ca68bu, h ^= h >> 16;
        p[i] = ((float)(h & 0xffffff) / 8388608.f - 1.f) * scale;
    }
}

struct Shape {
    const char *name;
    int C, Co, k, stride, H, W, count_fwd, count_bwd;   // launches per step, forward / data gradient (layer 1 is frozen)
};

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 10;
    void *h = dlopen(getenv("LSNET_SO") ? getenv("LSNET_SO") : "lsnet_amd/csrc/liblsnet_hip.so", RTLD_NOW);
    if (!h) {
        printf("dlopen: %s\n", dlerror());
        return 2;
    }
    auto bytes_fn = (int64_t(*)(int, int, int, int, int, int, int, int))dlsym(h, "lsn_conv2d_prepared_bytes");
    auto prep_fn = (int (*)(int, const float *, void *, int, int, int, int, int, int, int, lsn_stream_t))dlsym(h, "lsn_conv2d_prepare_weights");
    auto fwd_fn = (int (*)(int, const lsn_conv_level *, const void *, const float *, int, int, int, int, int, int, int, int, int,
                           lsn_stream_t))dlsym(h, "lsn_conv2d_forward_prepared");
    auto bwd_fn = (int (*)(int, const lsn_conv_level *, const void *, int, int, int, int, int, int, int, lsn_stream_t))dlsym(
        h, "lsn_conv2d_backward_data_prepared");
    auto err_fn = (const char *(*)(void))dlsym(h, "lsn_last_error");
    if (!bytes_fn || !prep_fn || !fwd_fn || !bwd_fn || !err_fn) return 2;
    const int B = 2;
    const Shape shapes[] = {
        {"l1 1x1 64->64", 64, 64, 1, 1, 200, 336, 3, 0},
        {"l1 3x3 64", 64, 64, 3, 1, 200, 336, 3, 0},
        {"l1 1x1 64->256", 64, 256, 1, 1, 200, 336, 4, 0},
        {"l1 1x1 256->64", 256, 64, 1, 1, 200, 336, 2, 0},
        {"l2 1x1 256->128", 256, 128, 1, 1, 200, 336, 1, 0},
        {"l2 3x3 s2 128", 128, 128, 3, 2, 200, 336, 1, 1},
        {"l2 3x3 128", 128, 128, 3, 1, 100, 168, 3, 3},
        {"l2 1x1 128->512", 128, 512, 1, 1, 100, 168, 4, 4},
        {"l2 1x1 512->128", 512, 128, 1, 1, 100, 168, 3, 3},
        {"l2 ds 1x1 s2 256->512", 256, 512, 1, 2, 200, 336, 1, 0},
        {"l3 1x1 512->256", 512, 256, 1, 1, 100, 168, 1, 1},
        {"l3 3x3 s2 256", 256, 256, 3, 2, 100, 168, 1, 1},
        {"l3 3x3 256", 256, 256, 3, 1, 50, 84, 5, 5},
        {"l3 1x1 256->1024", 256, 1024, 1, 1, 50, 84, 6, 6},
        {"l3 1x1 1024->256", 1024, 256, 1, 1, 50, 84, 5, 5},
        {"l3 ds 1x1 s2 512->1024", 512, 1024, 1, 2, 100, 168, 1, 1},
        {"l4 1x1 1024->512", 1024, 512, 1, 1, 50, 84, 1, 1},
        {"l4 3x3 s2 512", 512, 512, 3, 2, 50, 84, 1, 1},
        {"l4 3x3 512", 512, 512, 3, 1, 25, 42, 2, 2},
        {"l4 1x1 512->2048", 512, 2048, 1, 1, 25, 42, 3, 3},
        {"l4 1x1 2048->512", 2048, 512, 1, 1, 25, 42, 2, 2},
        {"l4 ds 1x1 s2 1024->2048", 1024, 2048, 1, 2, 50, 84, 1, 1},
        {"fpn lat 512->256 P3", 512, 256, 1, 1, 100, 168, 1, 1},
        {"fpn lat 1024->256 P4", 1024, 256, 1, 1, 50, 84, 1, 1},
        {"fpn lat 2048->256 P5", 2048, 256, 1, 1, 25, 42, 1, 1},
        {"fpn 3x3 256 P3", 256, 256, 3, 1, 100, 168, 1, 1},
        {"fpn 3x3 256 P4", 256, 256, 3, 1, 50, 84, 1, 1},
        {"fpn 3x3 256 P5", 256, 256, 3, 1, 25, 42, 1, 1},
        {"head 1x1 768->256 P3", 768, 256, 1, 1, 100, 168, 2, 2},
        {"head 3x3 256->27 P3 (offset conv)", 256, 27, 3, 1, 100, 168, 6, 6},
        {"head 3x3 256->80 P3 (cls out)", 256, 80, 3, 1, 100, 168, 1, 1},
    };
    double tot_f = 0, tot_b = 0;
    printf("%-36s %9s %7s %9s %7s %9s %9s\n", "shape", "fwd us", "TF", "bwd us", "TF", "fwd err", "bwd err");
    for (const Shape &s : shapes) {
        const int K = s.k * s.k, pad = s.k / 2;
        const int Ho = (s.H + 2 * pad - s.k) / s.stride + 1, Wo = (s.W + 2 * pad - s.k) / s.stride + 1;
        const size_t nx = (size_t)B * s.H * s.W * s.C, ny = (size_t)B * Ho * Wo * s.Co, nw = (size_t)s.Co * K * s.C;
        float *x, *y, *w, *gy, *gx, *bias;
        CK(hipMalloc(&x, nx * 4)), CK(hipMalloc(&y, ny * 4)), CK(hipMalloc(&w, nw * 4)), CK(hipMalloc(&gy, ny * 4));
        CK(hipMalloc(&gx, nx * 4)), CK(hipMalloc(&bias, (size_t)s.Co * 4));
        hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, x, nx, 3u, 1.f);
        hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, gy, ny, 4u, 1.f);
        hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, 0, w, nw, 5u, 1.7f / sqrtf((float)(s.C * K)));
        hipLaunchKernelGGL(fill_kernel, dim3(4), dim3(256), 0, 0, bias, (size_t)s.Co, 6u, 0.5f);
        const int64_t b0 = bytes_fn(0, s.C, s.Co, s.k, s.k, s.stride, pad, 1), b1 = bytes_fn(1, s.C, s.Co, s.k, s.k, s.stride, pad, 1);
        const bool can_bwd = b1 > 0 && s.Co % 4 == 0;
        void *img0 = nullptr, *img1 = nullptr;
        if (b0 <= 0) {
            printf("%-36s forward image unsupported\n", s.name);
            continue;
        }
        CK(hipMalloc(&img0, (size_t)b0));
        auto chk = [&](int rc, const char *what) {
            if (rc != 0) {
                printf("%s: %s rc %d: %s\n", s.name, what, rc, err_fn());
                exit(3);
            }
        };
        chk(prep_fn(0, w, img0, s.C, s.Co, s.k, s.k, s.stride, pad, 1, nullptr), "prepare 0");
        if (can_bwd) {
            CK(hipMalloc(&img1, (size_t)b1));
            chk(prep_fn(1, w, img1, s.C, s.Co, s.k, s.k, s.stride, pad, 1, nullptr), "prepare 1");
        }
        lsn_conv_level lf = {}, lb = {};
        lf.x = x, lf.out = y, lf.B = B, lf.H = s.H, lf.W = s.W;
        lb.x = gy, lb.out = gx, lb.B = B, lb.H = s.H, lb.W = s.W;
        auto fwd = [&] { chk(fwd_fn(1, &lf, img0, bias, s.C, s.C, s.Co, s.k, s.k, s.stride, pad, 1, 0, nullptr), "forward"); };
        auto bwd = [&] { chk(bwd_fn(1, &lb, img1, s.C, s.Co, s.k, s.k, s.stride, pad, 1, nullptr), "backward-data"); };
        hipEvent_t e0, e1, e2;
        CK(hipEventCreate(&e0)), CK(hipEventCreate(&e1)), CK(hipEventCreate(&e2));
        fwd(), fwd();
        if (can_bwd) bwd(), bwd();
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) fwd();
        CK(hipEventRecord(e1, 0));
        if (can_bwd)
            for (int i = 0; i < reps; ++i) bwd();
        CK(hipEventRecord(e2, 0));
        CK(hipEventSynchronize(e2));
        float tf = 0, tb = 0;
        CK(hipEventElapsedTime(&tf, e0, e1)), CK(hipEventElapsedTime(&tb, e1, e2));
        const double uf = tf * 1000.0 / reps, ub = tb * 1000.0 / reps;
        // host check on sampled elements
        std::vector<float> hx(nx), hy(ny), hw(nw), hgy(ny), hgx(nx), hb(s.Co);
        CK(hipMemcpy(hx.data(), x, nx * 4, hipMemcpyDeviceToHost)), CK(hipMemcpy(hy.data(), y, ny * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hw.data(), w, nw * 4, hipMemcpyDeviceToHost)), CK(hipMemcpy(hgy.data(), gy, ny * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hgx.data(), gx, nx * 4, hipMemcpyDeviceToHost)), CK(hipMemcpy(hb.data(), bias, (size_t)s.Co * 4, hipMemcpyDeviceToHost));
        double ef = 0, sf = 0, eb = 0, sb = 0;
        unsigned rng = 99991u;
        auto rnd = [&](int n) {
            rng = rng * 1664525u + 1013904223u;
            return (int)((rng >> 8) % (unsigned)n);
        };
        for (int t = 0; t < 64; ++t) {
            const int b = rnd(B), oy = t < 4 ? (t & 1) * (Ho - 1) : rnd(Ho), ox = t < 4 ? (t >> 1) * (Wo - 1) : rnd(Wo), co = rnd(s.Co);
            double acc = hb[co];
            for (int ky = 0; ky < s.k; ++ky)
                for (int kx = 0; kx < s.k; ++kx) {
                    const int iy = oy * s.stride - pad + ky, ix = ox * s.stride - pad + kx;
                    if (iy < 0 || iy >= s.H || ix < 0 || ix >= s.W) continue;
                    const float *xp = &hx[((size_t)(b * s.H + iy) * s.W + ix) * s.C], *wp = &hw[((size_t)co * K + ky * s.k + kx) * s.C];
                    for (int c = 0; c < s.C; ++c) acc += (double)xp[c] * wp[c];
                }
            ef = fmax(ef, fabs(hy[((size_t)(b * Ho + oy) * Wo + ox) * s.Co + co] - acc)), sf = fmax(sf, fabs(acc));
        }
        if (can_bwd)
            for (int t = 0; t < 64; ++t) {
                const int b = rnd(B), iy = t < 4 ? (t & 1) * (s.H - 1) : rnd(s.H), ix = t < 4 ? (t >> 1) * (s.W - 1) : rnd(s.W), c = rnd(s.C);
                double acc = 0;
                for (int ky = 0; ky < s.k; ++ky)
                    for (int kx = 0; kx < s.k; ++kx) {
                        const int ny_ = iy + pad - ky, nx_ = ix + pad - kx;
                        if (ny_ < 0 || nx_ < 0 || ny_ % s.stride || nx_ % s.stride) continue;
                        const int oy = ny_ / s.stride, ox = nx_ / s.stride;
                        if (oy >= Ho || ox >= Wo) continue;
                        const float *gp = &hgy[((size_t)(b * Ho + oy) * Wo + ox) * s.Co];
                        for (int co = 0; co < s.Co; ++co) acc += (double)gp[co] * hw[((size_t)co * K + ky * s.k + kx) * s.C + c];
                    }
                eb = fmax(eb, fabs(hgx[((size_t)(b * s.H + iy) * s.W + ix) * s.C + c] - acc)), sb = fmax(sb, fabs(acc));
            }
        const double fl = 2.0 * B * Ho * Wo * (double)s.C * s.Co * K;
        printf("%-36s %9.1f %7.1f", s.name, uf, fl / uf * 1e-6);
        if (can_bwd)
            printf(" %9.1f %7.1f %9.1e %9.1e\n", ub, fl / ub * 1e-6, ef / sf, eb / sb);
        else
            printf(" %9s %7s %9.1e %9s\n", "-", "-", ef / sf, "-");
        fflush(stdout);
        tot_f += uf * s.count_fwd, tot_b += can_bwd ? ub * s.count_bwd : 0;
        CK(hipFree(x)), CK(hipFree(y)), CK(hipFree(w)), CK(hipFree(gy)), CK(hipFree(gx)), CK(hipFree(bias)), CK(hipFree(img0));
        if (img1) CK(hipFree(img1));
    }
    printf("per step (counts of the benchmark step): forward %.0f us, data gradient %.0f us\n", tot_f, tot_b);
    return 0;
}
