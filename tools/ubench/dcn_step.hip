// The deformable-conv launches of ONE benchmark step (tools/step_shapes.py: a DCNv2 tower launch over the five FPN levels
// with fused offset | mask-logit tensors, and a pyramid launch of 15 (level, source) pairs; B = 2, 800 x 1344, 256 -> 256,
// 3x3) replayed through the C ABI only -- no torch, the binary starts in a second -- with
//   * forward / backward wall time per launch (HIP events) and the library's own per-family times (lsn_prof_*),
//   * every result checked against a double-precision evaluation on the host for sampled elements (tools/ubench/dcn_ref.h,
//     itself pinned against the oracle on CPU by tests/test_ubench_ref.py: tools may not link oracle/),
//   * the backward run twice and compared bit for bit,
//   * the same launches with debug bit 28 set (the kernels of dcn_kernels.h) as the A/B partner.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/dcn_step.hip -o tools/ubench/dcn_step -ldl
//   tools/ubench/dcn_step [tower|pyramid|both] [reps]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/lsnet_hip.h"
#include "dcn_ref.h"   // the host evaluation (pinned against the oracle on CPU: tests/test_ubench_ref.py)

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

struct Api {
    int (*fwd)(const lsn_dcn_shape *, int, const lsn_dcn_level *, const float *, const float *, lsn_layout, lsn_stream_t);
    int (*bwd)(const lsn_dcn_shape *, int, const lsn_dcn_level *, const float *, float *, float *, lsn_layout, lsn_stream_t);
    int64_t (*ws_bytes)(const lsn_dcn_shape *, int, const lsn_dcn_level *);
    int (*dbg)(long long *, int);
    const char *(*err)(void);
    int (*prof_enable)(int);
    int (*prof_read)(lsn_prof_entry *, int);
};

struct Buf {   // device tensor + host copy
    float *d = nullptr;
    std::vector<float> h;
    size_t n = 0;
    void alloc(size_t n_) { n = n_, CK(hipMalloc(&d, n * 4)); }
    void fill(unsigned seed, float scale) { hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, d, n, seed, scale); }
    void pull() { h.resize(n), CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost)); }
};

constexpr int B = 2, C = 256, Co = 256, KH = 3, K = 9, PAD = 1;
static const int SZ[5][2] = {{100, 168}, {50, 84}, {25, 42}, {13, 21}, {7, 11}};

struct Level {        // one (source map, offset field, output grid) triple
    int src;          // index into the feature maps
    int H, W, Ho, Wo;
    float sh, sw;
    int och;          // channels of the offset tensor: 18, or 27 with the mask logits behind the offsets
    Buf off, gout, out, goff;
    float *gx;        // grad_input buffer of the source (shared by the levels that sample it)
};

struct Launch {
    std::string name;
    bool fused;       // DCNv2 with mask logits
    std::vector<Level> lv;
    Buf w, bias, gw, gb;
    Buf gxs[5];
    bool src_used[5] = {false, false, false, false, false};
};

static Buf feats[5];

// ---- host reference: tools/ubench/dcn_ref.h over the host copies of this launch's tensors ----
static dcnref::Lv view(const Level &L)
{
    return dcnref::Lv{B, L.H, L.W, L.Ho, L.Wo, L.och, L.sh, L.sw, feats[L.src].h.data(), L.off.h.data(), L.gout.h.data()};
}

struct Err {
    double e = 0, scale = 0;
    void add(double got, double want) { e = fmax(e, fabs(got - want)), scale = fmax(scale, fabs(want)); }
    double rel() const { return scale > 0 ? e / scale : e; }
};

static unsigned g_rng = 2463534242u;
static int rnd(int n)
{
    g_rng ^= g_rng << 13, g_rng ^= g_rng >> 17, g_rng ^= g_rng << 5;
    return (int)(g_rng % (unsigned)n);
}

static void check_forward(const Launch &la, Err &e)
{
    for (int t = 0; t < 96; ++t) {
        const Level &L = la.lv[rnd((int)la.lv.size())];
        const int b = rnd(B), ho = rnd(L.Ho), wo = rnd(L.Wo), co = rnd(Co);
        const double s = dcnref::forward_at(view(L), la.w.h.data(), la.bias.d ? la.bias.h.data() : nullptr, C, Co, b, ho, wo, co);
        e.add(L.out.h[((size_t)(b * L.Ho + ho) * L.Wo + wo) * Co + co], s);
    }
}

static void check_backward(const Launch &la, Err &ew, Err &eb, Err &eo, Err &em, Err &ex)
{
    // weight / bias gradient: summed over the levels
    for (int t = 0; t < 24; ++t) {
        const int co = rnd(Co), k = t < K ? t : rnd(K), c = rnd(C);
        double s = 0, sb = 0;
        for (const Level &L : la.lv) {
            double a, ab;
            dcnref::gw_at(view(L), C, Co, co, k, c, &a, &ab);
            s += a, sb += ab;
        }
        ew.add(la.gw.h[((size_t)co * K + k) * C + c], s);
        if (la.gb.d && t < 6) eb.add(la.gb.h[co], sb);
    }
    // offset / mask gradient
    for (int t = 0; t < 48; ++t) {
        const Level &L = la.lv[rnd((int)la.lv.size())];
        const int b = rnd(B), ho = rnd(L.Ho), wo = rnd(L.Wo), k = rnd(K);
        double gy, gx, gm;
        dcnref::goff_at(view(L), la.w.h.data(), C, Co, b, ho, wo, k, &gy, &gx, &gm);
        const float *go = &L.goff.h[((size_t)(b * L.Ho + ho) * L.Wo + wo) * L.och];
        eo.add(go[2 * k], gy), eo.add(go[2 * k + 1], gx);
        if (la.fused) em.add(go[2 * K + k], gm);
    }
    // input gradient: sampled (source, b, y, x, c); every level that reads the source contributes
    for (int t = 0; t < 10; ++t) {
        int src;
        do src = rnd(5); while (!la.src_used[src]);
        const int H = SZ[src][0], W = SZ[src][1];
        const int b = rnd(B), y = rnd(H), x = rnd(W), c = rnd(C);
        double s = 0;
        for (const Level &L : la.lv)
            if (L.src == src) s += dcnref::gx_at(view(L), la.w.h.data(), C, Co, b, y, x, c);
        ex.add(la.gxs[src].h[((size_t)(b * H + y) * W + x) * C + c], s);
    }
}

static void build_launch(Launch &la, bool tower)
{
    la.fused = tower;
    la.name = tower ? "tower (DCNv2, 5 levels)" : "pyramid (15 pairs)";
    static const int LISTS[5][3] = {{0, 1, 2}, {1, 0, 2}, {2, 1, 3}, {3, 2, 4}, {4, 3, 2}};
    unsigned seed = tower ? 1000u : 2000u;
    la.w.alloc((size_t)Co * K * C), la.w.fill(seed++, 0.035f), la.w.pull();
    la.gw.alloc((size_t)Co * K * C);
    if (tower) la.bias.alloc(Co), la.bias.fill(seed++, 0.5f), la.bias.pull(), la.gb.alloc(Co);
    for (int l = 0; l < 5; ++l)
        for (int j = 0; j < (tower ? 1 : 3); ++j) {
            Level L = {};
            L.src = tower ? l : LISTS[l][j];
            L.H = SZ[L.src][0], L.W = SZ[L.src][1], L.Ho = SZ[l][0], L.Wo = SZ[l][1];
            L.sh = (float)L.H / L.Ho, L.sw = (float)L.W / L.Wo;
            L.och = tower ? 27 : 18;
            const size_t po = (size_t)B * L.Ho * L.Wo;
            const float oscale = tower ? 0.87f : 3.46f * fmaxf((float)L.H / L.Ho, 1.f);   // uniform with the std of step_shapes.py
            L.off.alloc(po * L.och), L.off.fill(seed++, oscale), L.off.pull();
            L.gout.alloc(po * Co), L.gout.fill(seed++, 1.f), L.gout.pull();
            L.out.alloc(po * Co), L.goff.alloc(po * L.och);
            la.src_used[L.src] = true;
            la.lv.push_back(std::move(L));
        }
    for (int s = 0; s < 5; ++s)
        if (la.src_used[s]) la.gxs[s].alloc(feats[s].n);
    for (Level &L : la.lv) L.gx = la.gxs[L.src].d;
}

static lsn_strides4 nhwc_strides(int ch, int H, int W) { return lsn_strides4{(int64_t)H * W * ch, 1, (int64_t)W * ch, ch}; }

static void fill_abi(const Launch &la, lsn_dcn_shape &s, std::vector<lsn_dcn_level> &lv)
{
    memset(&s, 0, sizeof(s));
    s.C = C, s.Co = Co, s.kh = s.kw = KH, s.stride = 1, s.pad = PAD, s.dil = 1, s.groups = 1, s.deformable_groups = 1;
    s.scale_h = s.scale_w = 1.f, s.mask_is_logit = la.fused ? 1 : 0;
    lv.assign(la.lv.size(), lsn_dcn_level{});
    for (size_t i = 0; i < la.lv.size(); ++i) {
        const Level &L = la.lv[i];
        lsn_dcn_level &d = lv[i];
        d.input = feats[L.src].d, d.offset = L.off.d, d.output = L.out.d, d.grad_output = L.gout.d;
        d.grad_input = L.gx, d.grad_offset = L.goff.d;
        d.off_st = nhwc_strides(L.och, L.Ho, L.Wo);
        if (la.fused) d.mask = L.off.d + 2 * K, d.mask_st = d.off_st, d.grad_mask = L.goff.d + 2 * K;
        d.B = B, d.H = L.H, d.W = L.W, d.Ho = L.Ho, d.Wo = L.Wo, d.scale_h = L.sh, d.scale_w = L.sw;
    }
}

int main(int argc, char **argv)
{
    const std::string which = argc > 1 ? argv[1] : "both";
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    void *h = dlopen(getenv("LSNET_SO") ? getenv("LSNET_SO") : "lsnet_amd/csrc/liblsnet_hip.so", RTLD_NOW);
    if (!h) {
        printf("dlopen: %s\n", dlerror());
        return 2;
    }
    Api api;
    api.fwd = (decltype(api.fwd))dlsym(h, "lsn_dcn_forward");
    api.bwd = (decltype(api.bwd))dlsym(h, "lsn_dcn_backward");
    api.ws_bytes = (decltype(api.ws_bytes))dlsym(h, "lsn_dcn_backward_workspace_bytes");
    api.dbg = (decltype(api.dbg))dlsym(h, "lsn_debug_phase_clocks");
    api.err = (decltype(api.err))dlsym(h, "lsn_last_error");
    api.prof_enable = (decltype(api.prof_enable))dlsym(h, "lsn_prof_enable");
    api.prof_read = (decltype(api.prof_read))dlsym(h, "lsn_prof_read");
    if (!api.fwd || !api.bwd || !api.ws_bytes || !api.dbg || !api.err || !api.prof_enable || !api.prof_read) return 2;
    for (int l = 0; l < 5; ++l) {
        feats[l].alloc((size_t)B * SZ[l][0] * SZ[l][1] * C);
        feats[l].fill(7u + l, 1.7f);
        feats[l].pull();
    }
    for (int tower = 1; tower >= 0; --tower) {
        if ((tower && which == "pyramid") || (!tower && which == "tower")) continue;
        Launch la;
        build_launch(la, tower != 0);
        lsn_dcn_shape s;
        std::vector<lsn_dcn_level> lv;
        fill_abi(la, s, lv);
        float *ws;
        CK(hipMalloc(&ws, (size_t)Co * K * C * 8));
        s.workspace = ws;
        const int n = (int)lv.size();
        auto chk = [&](int rc, const char *what) {
            if (rc != 0) {
                printf("%s: %s rc %d: %s\n", la.name.c_str(), what, rc, api.err());
                exit(3);
            }
        };
        std::vector<float> keep[2];   // every gradient of mode 1 (new) and mode 0 (debug bit 28), concatenated
        for (int mode = 1; mode >= 0; --mode) {
            api.dbg(nullptr, mode ? (getenv("DCN_STEP_DBG") ? atoi(getenv("DCN_STEP_DBG")) : 0) : (1 << 28));   // DCN_STEP_DBG: debug bits of the default arm (A/B)
            const int64_t gbytes = api.ws_bytes(&s, n, lv.data());
            void *gws = nullptr;
            if (gbytes > 0) CK(hipMalloc(&gws, (size_t)gbytes));
            s.gather_workspace = gws, s.gather_workspace_bytes = gbytes;
            auto fwd = [&] { chk(api.fwd(&s, n, lv.data(), la.w.d, la.bias.d, LSN_NHWC, nullptr), "forward"); };
            auto bwd = [&] { chk(api.bwd(&s, n, lv.data(), la.w.d, la.gw.d, la.gb.d, LSN_NHWC, nullptr), "backward"); };
            hipEvent_t e0, e1, e2;
            CK(hipEventCreate(&e0)), CK(hipEventCreate(&e1)), CK(hipEventCreate(&e2));
            fwd(), bwd();   // warm-up (scratch growth)
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) fwd();
            CK(hipEventRecord(e1, 0));
            for (int i = 0; i < reps; ++i) bwd();
            CK(hipEventRecord(e2, 0));
            CK(hipEventSynchronize(e2));
            float tf = 0, tb = 0;
            CK(hipEventElapsedTime(&tf, e0, e1)), CK(hipEventElapsedTime(&tb, e1, e2));
            api.prof_enable(1);
            for (int i = 0; i < reps; ++i) fwd(), bwd();
            lsn_prof_entry pe[16];
            const int ne = api.prof_read(pe, 16);
            api.prof_enable(0);
            double px = 0;
            for (const Level &L : la.lv) px += (double)B * L.Ho * L.Wo;
            const double gf = 2.0 * px * Co * C * K * 1e-9;
            printf("%-26s %s  gather workspace %.0f MB  forward %.1f us  backward %.1f us  (%.1f GFLOP per pass)\n", la.name.c_str(),
                   mode ? "default kernels   " : "debug bit 28 (old)", gbytes / 1048576.0, tf * 1000 / reps, tb * 1000 / reps, gf);
            for (int i = 0; i < ne; ++i)
                if (pe[i].launches > 0)
                    printf("    %-14s %3lld launches  %8.1f us each  %6.1f TF\n", pe[i].name, pe[i].launches,
                           pe[i].total_ms * 1000 / pe[i].launches, pe[i].flops / (pe[i].total_ms * 1e-3) * 1e-12);
            // results of one more pair, pulled
            fwd(), bwd();
            CK(hipDeviceSynchronize());
            for (Level &L : la.lv) L.out.pull(), L.goff.pull();
            la.gw.pull();
            if (la.gb.d) la.gb.pull();
            for (int q = 0; q < 5; ++q)
                if (la.src_used[q]) la.gxs[q].pull();
            std::vector<float> &kp = keep[mode];
            auto app = [&](const std::vector<float> &v) { kp.insert(kp.end(), v.begin(), v.end()); };
            for (Level &L : la.lv) app(L.goff.h);
            app(la.gw.h);
            if (la.gb.d) app(la.gb.h);
            for (int q = 0; q < 5; ++q)
                if (la.src_used[q]) app(la.gxs[q].h);
            if (mode == 1) {
                Err ef, ew, eb, eo, em, ex;
                g_rng = 2463534242u;
                check_forward(la, ef);
                check_backward(la, ew, eb, eo, em, ex);
                printf("    against the host (double), relative to the largest sampled value: forward %.1e  grad_weight %.1e  grad_bias %.1e  "
                       "grad_offset %.1e  grad_mask %.1e  grad_input %.1e\n", ef.rel(), ew.rel(), eb.rel(), eo.rel(), em.rel(), ex.rel());
                // the backward again: bit for bit?
                bwd();
                CK(hipDeviceSynchronize());
                std::vector<float> again;
                for (Level &L : la.lv) L.goff.pull(), again.insert(again.end(), L.goff.h.begin(), L.goff.h.end());
                la.gw.pull(), again.insert(again.end(), la.gw.h.begin(), la.gw.h.end());
                if (la.gb.d) la.gb.pull(), again.insert(again.end(), la.gb.h.begin(), la.gb.h.end());
                for (int q = 0; q < 5; ++q)
                    if (la.src_used[q]) la.gxs[q].pull(), again.insert(again.end(), la.gxs[q].h.begin(), la.gxs[q].h.end());
                size_t diff = 0;
                for (size_t i = 0; i < again.size(); ++i) diff += memcmp(&again[i], &kp[i], 4) != 0;
                printf("    backward twice: %zu of %zu gradient elements differ\n", diff, again.size());
            } else {
                double d = 0, m = 0;
                for (size_t i = 0; i < kp.size(); ++i) d = fmax(d, fabs((double)kp[i] - keep[1][i])), m = fmax(m, fabs((double)kp[i]));
                printf("    default vs old kernels, all gradients: max |diff| / max |value| = %.1e\n", d / m);
            }
            if (gws) CK(hipFree(gws));
            fflush(stdout);
        }
        api.dbg(nullptr, 0);
    }
    return 0;
}
