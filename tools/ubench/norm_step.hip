// The streaming normalisation kernels of the benchmark step through the C ABI, torch-free: the ReLU gate pass
// (lsn_relu_gate) at the backbone's stage outputs and GroupNorm (+ReLU) forward / backward
// over the five FPN levels (B = 2, 800 x 1344).  Per call: time (HIP events), algorithmic GB/s (each tensor read or
// written once), and a host check in double precision of sampled elements / channels.  These kernels are HBM-bound: the
// number to look at is GB/s against ~4 - 5 TB/s of achievable stream bandwidth.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/norm_step.hip -o tools/ubench/norm_step -ldl
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
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

__global__ void fill_kernel(float *p, size_t n, unsigned seed, float scale, float shift)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 16, h *= 0x7feb352du, h ^= h >> 15, h *= 0x846ca68bu, h ^= h >> 16;
        p[i] = ((float)(h & 0xffffff) / 8388608.f - 1.f) * scale + shift;
    }
}

struct Buf {
    float *d = nullptr;
    size_t n = 0;
    std::vector<float> h;
    void alloc(size_t n_) { n = n_, CK(hipMalloc(&d, n * 4)); }
    void fill(unsigned seed, float scale, float shift = 0.f)
    {
        hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, d, n, seed, scale, shift);
    }
    void pull() { h.resize(n), CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost)); }
    void release() { CK(hipFree(d)), d = nullptr; }
};

template <class F>
static double time_us(F f, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)), CK(hipEventCreate(&e1));
    f(), f();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.0 / reps;
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 10;
    void *h = dlopen(getenv("LSNET_SO") ? getenv("LSNET_SO") : "lsnet_amd/csrc/liblsnet_hip.so", RTLD_NOW);
    if (!h) {
        printf("dlopen: %s\n", dlerror());
        return 2;
    }
    auto gate = (int (*)(const float *, const float *, float *, int64_t, lsn_stream_t))dlsym(h, "lsn_relu_gate");
    auto gn_ws = (int64_t(*)(int, const lsn_gn_level *, int, int))dlsym(h, "lsn_group_norm_workspace_bytes");
    auto gn_fwd = (int (*)(int, const lsn_gn_level *, int, int, const float *, const float *, float, int, float *, void *,
                           lsn_stream_t))dlsym(h, "lsn_group_norm_forward");
    auto gn_bwd = (int (*)(int, const lsn_gn_level *, int, int, const float *, const float *, int, const float *, float *, float *,
                           void *, int, lsn_stream_t))dlsym(h, "lsn_group_norm_backward");
    auto err = (const char *(*)(void))dlsym(h, "lsn_last_error");
    if (!gate || !gn_ws || !gn_fwd || !gn_bwd || !err) return 2;
    auto chk = [&](int rc, const char *what) {
        if (rc != 0) {
            printf("%s rc %d: %s\n", what, rc, err());
            exit(3);
        }
    };

    // ---- ReLU gate (lsn_relu_gate): the one stand-alone pass per backbone stage that the fused bottleneck backward leaves
    // (the last block of a stage; every other gate rides in a backward-data epilogue) ----
    struct GateShape {
        const char *name;
        int N, C, count;
    };
    const GateShape gs[] = {{"l2 out 100x168x512", 33600, 512, 1}, {"l3 out 50x84x1024", 8400, 1024, 1}, {"l4 out 25x42x2048", 2100, 2048, 1}};
    double tot = 0;
    printf("%-36s %9s %8s %9s\n", "ReLU gate", "us", "GB/s", "mismatch");
    for (const GateShape &s : gs) {
        const size_t n = (size_t)s.N * s.C;
        Buf dy, y, g;
        dy.alloc(n), y.alloc(n), g.alloc(n);
        dy.fill(1, 1.f), y.fill(2, 1.f, 0.2f);
        auto run = [&] { chk(gate(dy.d, y.d, g.d, (int64_t)n, nullptr), "relu gate"); };
        const double us = time_us(run, reps);
        run();
        CK(hipDeviceSynchronize());
        dy.pull(), y.pull(), g.pull();
        size_t bad = 0;
        for (size_t i = 0; i < n; i += 7) bad += g.h[i] != (y.h[i] > 0.f ? dy.h[i] : 0.f);
        printf("%-36s %9.1f %8.0f %9zu\n", s.name, us, 12.0 * n / us * 1e-3, bad);
        fflush(stdout);
        tot += us * s.count;
        for (Buf *b : {&dy, &y, &g}) b->release();
    }
    printf("per step (counts of the benchmark step): %.0f us\n", tot);

    // ---- GroupNorm (+ReLU) over the five FPN levels, C = 256, 32 groups ----
    {
        const int C = 256, G = 32, B = 2;
        const int HW[5] = {100 * 168, 50 * 84, 25 * 42, 13 * 21, 7 * 11};
        Buf x[5], y[5], dy[5], dx[5], gamma, beta, dgamma, dbeta, mr;
        lsn_gn_level lv[5] = {};
        size_t total = 0;
        for (int l = 0; l < 5; ++l) {
            const size_t n = (size_t)B * HW[l] * C;
            total += n;
            x[l].alloc(n), y[l].alloc(n), dy[l].alloc(n), dx[l].alloc(n);
            x[l].fill(10 + l, 1.5f, 0.3f), dy[l].fill(20 + l, 1.f);
            lv[l].x = x[l].d, lv[l].y = y[l].d, lv[l].dy = dy[l].d, lv[l].dx = dx[l].d, lv[l].B = B, lv[l].HW = HW[l];
        }
        gamma.alloc(C), beta.alloc(C), dgamma.alloc(C), dbeta.alloc(C), mr.alloc((size_t)5 * B * G * 2);
        gamma.fill(30, 0.5f, 1.f), beta.fill(31, 0.3f);
        void *ws = nullptr;
        CK(hipMalloc(&ws, (size_t)gn_ws(5, lv, C, G)));
        auto fwd = [&] { chk(gn_fwd(5, lv, C, G, gamma.d, beta.d, 1e-5f, 1, mr.d, ws, nullptr), "gn forward"); };
        auto bwd = [&] { chk(gn_bwd(5, lv, C, G, gamma.d, beta.d, 1, mr.d, dgamma.d, dbeta.d, ws, 0, nullptr), "gn backward"); };
        const double uf = time_us(fwd, reps), ub = time_us(bwd, reps);
        fwd(), bwd();
        CK(hipDeviceSynchronize());
        // host check: level 2, image 1, group 5 (forward values, dx), and dgamma / dbeta of 4 channels over everything
        gamma.pull(), beta.pull(), dgamma.pull(), dbeta.pull();
        for (int l = 0; l < 5; ++l) x[l].pull(), y[l].pull(), dy[l].pull(), dx[l].pull();
        const int cg = C / G;
        double ef = 0, sf = 0, ed = 0, sd = 0, eg = 0, sg = 0, eb = 0, sb = 0;
        auto stats = [&](int l, int b, int g, double &mean, double &rstd) {
            double s1 = 0, s2 = 0;
            for (int p = 0; p < HW[l]; ++p)
                for (int c = g * cg; c < (g + 1) * cg; ++c) {
                    const double v = x[l].h[((size_t)b * HW[l] + p) * C + c];
                    s1 += v, s2 += v * v;
                }
            const double n = (double)HW[l] * cg;
            mean = s1 / n;
            rstd = 1.0 / sqrt(s2 / n - mean * mean + 1e-5);
        };
        {
            const int l = 2, b = 1, g = 5;
            double mean, rstd;
            stats(l, b, g, mean, rstd);
            // dx of GroupNorm: with xh = (x - mean) rstd, dz = dy [y > 0], t = dz gamma:
            //   dx = rstd (t - mean_group(t) - xh mean_group(t xh))
            double m1 = 0, m2 = 0;
            const double n = (double)HW[l] * cg;
            for (int p = 0; p < HW[l]; ++p)
                for (int c = g * cg; c < (g + 1) * cg; ++c) {
                    const size_t o = ((size_t)b * HW[l] + p) * C + c;
                    const double xh = (x[l].h[o] - mean) * rstd, yv = xh * gamma.h[c] + beta.h[c];
                    const double t = (yv > 0 ? dy[l].h[o] : 0.0) * gamma.h[c];
                    m1 += t, m2 += t * xh;
                }
            m1 /= n, m2 /= n;
            for (int p = 0; p < HW[l]; p += 37)
                for (int c = g * cg; c < (g + 1) * cg; ++c) {
                    const size_t o = ((size_t)b * HW[l] + p) * C + c;
                    const double xh = (x[l].h[o] - mean) * rstd, yv = xh * gamma.h[c] + beta.h[c];
                    const double want = yv > 0 ? yv : 0.0;
                    ef = fmax(ef, fabs(y[l].h[o] - want)), sf = fmax(sf, fabs(want));
                    const double t = (yv > 0 ? dy[l].h[o] : 0.0) * gamma.h[c];
                    const double wdx = rstd * (t - m1 - xh * m2);
                    ed = fmax(ed, fabs(dx[l].h[o] - wdx)), sd = fmax(sd, fabs(wdx));
                }
        }
        for (int c : {0, 77, 130, 255}) {
            double dg = 0, db = 0;
            for (int l = 0; l < 5; ++l)
                for (int b = 0; b < B; ++b) {
                    double mean, rstd;
                    stats(l, b, c / cg, mean, rstd);
                    for (int p = 0; p < HW[l]; ++p) {
                        const size_t o = ((size_t)b * HW[l] + p) * C + c;
                        const double xh = (x[l].h[o] - mean) * rstd, yv = xh * gamma.h[c] + beta.h[c];
                        const double dz = yv > 0 ? dy[l].h[o] : 0.0;
                        dg += dz * xh, db += dz;
                    }
                }
            eg = fmax(eg, fabs(dgamma.h[c] - dg)), sg = fmax(sg, fabs(dg));
            eb = fmax(eb, fabs(dbeta.h[c] - db)), sb = fmax(sb, fabs(db));
        }
        printf("GroupNorm+ReLU, 5 levels x 256 ch: forward %.1f us (%.0f GB/s algorithmic: x read twice, y written), backward %.1f us "
               "(%.0f GB/s: x, dy read twice, dx written)\n", uf, 4.0 * total * 3 / uf * 1e-3, ub, 4.0 * total * 5 / ub * 1e-3);
        printf("    against the host: y %.1e  dx %.1e  dgamma %.1e  dbeta %.1e   (x6 per step: 2 towers x 3 layers)\n", ef / sf, ed / sd,
               eg / sg, eb / sb);
    }
    return 0;
}
