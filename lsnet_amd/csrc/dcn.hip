// Host side of the deformable-convolution family: argument checks (mirroring the reference's
// TORCH_CHECKs, deform_conv_cuda.cpp:92-180,182-272), layout adapters, launch geometry and the
// extern "C" entry points declared in include/lsnet_hip.h.
#include <limits.h>
#include <string.h>

#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "dcn_kernels.h"
#include "dcn_mm_kernels.h"
#include "dcn_grouped_kernels.h"
#include "prof.h"

namespace lsn {

static thread_local char g_err[640];
char *err_buf() { return g_err; }
int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// optional phase-clock capture (diagnostics only; see lsn_debug_phase_clocks)
static long long *g_dbg_buf = nullptr;
static int g_dbg_block = 0;

// arithmetic of the implicit-GEMM contractions: exact fp32 MFMA, or split-bf16 products (common.h)
static int g_math_mode = -1;   // -1: not initialised (LSNET_MATH decides at first use)
static int math_mode()
{
    if (g_math_mode < 0) {
        const char *e = getenv("LSNET_MATH");
        g_math_mode = (e && (!strcmp(e, "fp32") || !strcmp(e, "exact"))) ? LSN_MATH_FP32
                      : (e && !strcmp(e, "bf16x3"))                       ? LSN_MATH_BF16X3
                                                                          : LSN_MATH_BF16X6;
    }
    return g_math_mode;
}
// bf16 products per fp32 product of the current mode (0: exact fp32 MFMA)
static int math_np() { return math_mode() == LSN_MATH_BF16X6 ? 6 : (math_mode() == LSN_MATH_BF16X3 ? 3 : 0); }
int split_np() { return math_np(); }
void dbg_state(long long **buf, int *block) { *buf = g_dbg_buf, *block = g_dbg_block; }

// ---- per-kernel launch timing (prof.h) ----
static const char *const kProfNames[PROF_N] = {"dcn_fwd", "dcn_bwd_data", "dcn_wgrad", "conv_fwd", "conv_bwd_data",
                                               "conv_wgrad", "norm", "gconv"};
static unsigned g_prof_mask = 0;   // bit f: family f records events
static std::vector<ProfRec> g_prof;
bool prof_on(int fam) { return fam >= 0 && fam < 32 && ((g_prof_mask >> fam) & 1u); }
void prof_push(const ProfRec &r) { g_prof.push_back(r); }
static bool g_launch_log = false;
static std::vector<LaunchRec> g_launches;
bool launch_log_on() { return g_launch_log; }
void launch_log_push(const LaunchRec &r) { g_launches.push_back(r); }

static double dcn_flops(const DcnArgs &a)
{
    double px = 0;
    for (int i = 0; i < a.nlv; ++i) px += (double)a.lv[i].B * a.lv[i].Ho * a.lv[i].Wo;
    return 2.0 * px * a.Co * (a.C / a.groups) * a.kh * a.kw;
}

// algorithmic HBM bytes of one launch: every operand read or written once
static double dcn_bytes(const DcnArgs &a, int fam)
{
    const double K = (double)a.kh * a.kw;
    double in = 0, out = 0, om = 0;
    for (int i = 0; i < a.nlv; ++i) {
        const double opx = (double)a.lv[i].B * a.lv[i].Ho * a.lv[i].Wo;
        in += (double)a.lv[i].B * a.lv[i].H * a.lv[i].W * a.C;
        out += opx * a.Co;
        om += opx * a.dg * K * (a.lv[i].msk ? 3 : 2);
    }
    const double w = (double)a.Co * (a.C / a.groups) * K;
    switch (fam) {
    case PROF_FWD: return 4.0 * (in + om + w + out);
    case PROF_BWD_DATA: return 4.0 * (in + om + w + out + in + om);   // + grad_input, grad_offset/mask
    default: return 4.0 * (in + om + out + w);
    }
}

struct ProfScope : ProfSpan {
    ProfScope(int fam, const DcnArgs &a, hipStream_t s) : ProfSpan(fam, dcn_flops(a), dcn_bytes(a, fam), s) {}
};

// in[b][r][s] -> out[b][s][r]   (NCHW <-> NHWC with r = C, s = H*W; weight OIHW <-> OHWI with
// b = Co, r = Cg, s = kh*kw).  32x32 tiles through LDS so both sides are coalesced.
__global__ void permute_rs_kernel(const float *__restrict__ in, float *__restrict__ out, int R, int S)
{
    __shared__ float tile[32][33];
    const size_t base = (size_t)blockIdx.z * R * S;
    const int s0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = r0 + i, s = s0 + threadIdx.x;
        if (r < R && s < S) tile[i][threadIdx.x] = in[base + (size_t)r * S + s];
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int s = s0 + i, r = r0 + threadIdx.x;
        if (r < R && s < S) out[base + (size_t)s * R + r] = tile[threadIdx.x][i];
    }
}

static int permute_rs(const float *in, float *out, int Bn, int R, int S, hipStream_t st)
{
    if (Bn <= 0 || R <= 0 || S <= 0) return 0;
    for (int b0 = 0; b0 < Bn; b0 += 65535) {
        const int nb = Bn - b0 < 65535 ? Bn - b0 : 65535;
        dim3 grid(cdiv(S, 32), cdiv(R, 32), nb), block(32, 8);
        hipLaunchKernelGGL(permute_rs_kernel, grid, block, 0, st, in + (size_t)b0 * R * S,
                           out + (size_t)b0 * R * S, R, S);
    }
    LSN_HIP(hipGetLastError());
    return 0;
}

__global__ void scale_kernel(float *p, size_t n, float s)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] *= s;
}

// stream-ordered scratch: everything is freed (in stream order) when the holder goes out of scope
struct Scratch {
    hipStream_t st;
    std::vector<void *> ptrs;
    explicit Scratch(hipStream_t s) : st(s) {}
    float *get(size_t nfloats)
    {
        void *p = nullptr;
        if (nfloats == 0) nfloats = 1;
        if (hipMallocAsync(&p, nfloats * sizeof(float), st) != hipSuccess) return nullptr;
        lib_stat(STAT_POOL_ALLOCS, 1);
        ptrs.push_back(p);
        return reinterpret_cast<float *>(p);
    }
    ~Scratch()
    {
        for (void *p : ptrs) (void)hipFreeAsync(p, st);
    }
};

static int check_shape(const lsn_dcn_shape &s)
{
    LSN_CHECK(s.kh > 0 && s.kw > 0, "kernel size should be greater than zero, but got kH: %d kW: %d", s.kh, s.kw);
    LSN_CHECK(s.stride > 0, "stride should be greater than zero, but got %d", s.stride);
    LSN_CHECK(s.dil > 0, "dilation should be greater than 0, but got %d", s.dil);
    LSN_CHECK(s.pad >= 0, "padding should be non-negative, but got %d", s.pad);
    LSN_CHECK(s.C > 0 && s.Co > 0, "channels must be positive (C %d, Co %d)", s.C, s.Co);
    LSN_CHECK(s.groups > 0 && s.C % s.groups == 0 && s.Co % s.groups == 0,
              "input/output channels (%d, %d) must be divisible by groups %d", s.C, s.Co, s.groups);
    LSN_CHECK(s.deformable_groups > 0 && s.C % s.deformable_groups == 0,
              "input channels must divide deformable group size");
    const int Cg = s.C / s.groups, cpdg = s.C / s.deformable_groups;
    if (!(Cg % cpdg == 0 || cpdg % Cg == 0))
        return fail(LSN_ERR_UNSUPPORTED, "groups %d / deformable_groups %d do not nest", s.groups,
                    s.deformable_groups);
    if (s.kh * s.kw * s.deformable_groups > 64)
        return fail(LSN_ERR_UNSUPPORTED, "kh*kw*deformable_groups = %d > 64 is not supported",
                    s.kh * s.kw * s.deformable_groups);
    LSN_CHECK(s.out_pitch == 0 || (s.out_pitch >= s.Co && s.out_pitch % 4 == 0),
              "out_pitch %d: 0 or a multiple of 4 >= Co = %d", s.out_pitch, s.Co);
    return 0;
}

static int i32(int64_t v, int *ok)
{
    if (v > INT_MAX || v < INT_MIN) *ok = 0;
    return (int)v;
}

template <typename KernelT>
static int set_lds(KernelT kernel, size_t bytes)
{
    if (bytes > 160 * 1024) return fail(LSN_ERR_UNSUPPORTED, "kernel needs %zu B of LDS (> 160 KiB)", bytes);
    if (bytes > 48 * 1024)
        LSN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

// Fill the per-level part of DcnArgs from the public descriptors (NHWC pointers supplied here).
static int fill_levels(DcnArgs &a, const lsn_dcn_shape &s, int n, const lsn_dcn_level *lv, int tile_px)
{
    LSN_CHECK(n >= 1 && n <= MAXLV, "n_levels must be in [1,%d], got %d", MAXLV, n);
    int ok = 1, tiles = 0;
    for (int i = 0; i < n; ++i) {
        Lvl &L = a.lv[i];
        const lsn_dcn_level &d = lv[i];
        LSN_CHECK(d.B > 0 && d.H > 0 && d.W > 0 && d.Ho > 0 && d.Wo > 0,
                  "level %d: Given input size (%d x %d x %d), calculated output size (%d x %d x %d). "
                  "Output size is too small", i, s.C, d.H, d.W, s.Co, d.Ho, d.Wo);
        LSN_CHECK(d.offset != nullptr, "level %d: offset is NULL", i);
        L.B = d.B; L.H = d.H; L.W = d.W; L.Ho = d.Ho; L.Wo = d.Wo;
        L.P = i32((int64_t)d.B * d.Ho * d.Wo, &ok);
        (void)i32((int64_t)d.B * d.H * d.W * s.C, &ok);
        L.tile0 = tiles;
        tiles += cdiv(L.P, tile_px);
        L.sh = d.scale_h; L.sw = d.scale_w;
        L.off = d.offset; L.msk = d.mask;
        L.osb = i32(d.off_st.b, &ok); L.osc = i32(d.off_st.c, &ok);
        L.osh = i32(d.off_st.h, &ok); L.osw = i32(d.off_st.w, &ok);
        L.msb = i32(d.mask_st.b, &ok); L.msc = i32(d.mask_st.c, &ok);
        L.msh = i32(d.mask_st.h, &ok); L.msw = i32(d.mask_st.w, &ok);
        L.x = L.gout = nullptr;
        L.out = L.gx = nullptr;
        L.goff = d.grad_offset; L.gmsk = d.grad_mask;
    }
    if (!ok) return fail(LSN_ERR_UNSUPPORTED, "tensor too large for 32-bit indexing");
    a.nlv = n;
    a.ntiles = tiles;
    a.C = s.C; a.Co = s.Co; a.kh = s.kh; a.kw = s.kw; a.stride = s.stride; a.pad = s.pad; a.dil = s.dil;
    a.groups = s.groups; a.dg = s.deformable_groups;
    const int Cg = s.C / s.groups, cpdg = s.C / s.deformable_groups;
    a.SL = Cg < cpdg ? Cg : cpdg;
    a.msig = s.mask_is_logit ? 1 : 0;
    a.gcol = nullptr;
    a.gtap = nullptr;
    a.gtap_rows = 0;
    a.wg_vec = 0;
    a.mm = 0;
    a.opitch = s.out_pitch > 0 ? s.out_pitch : s.Co;
    a.wtp_bytes = 0;
    a.wg_part = a.wg_part_b = nullptr;
    a.dbg = g_dbg_buf;
    a.dbg_block = g_dbg_block;
    a.w = a.bias = nullptr;
    a.wtp = nullptr;
    a.gw = a.gb = nullptr;
    return 0;
}

// VEC kernels need 16-byte aligned float4 rows: channel counts that are multiples of 4
static bool vec_ok(const DcnArgs &a)
{
    const int Cg = a.C / a.groups, Cog = a.Co / a.groups;
    return (Cg % 4 == 0) && (Cog % 4 == 0) && (a.Co % 4 == 0) && (a.SL % 4 == 0);
}

// ---- the kernels of dcn_mm_kernels.h (dense-convolution skeleton): conditions, weight image, launches ----
static bool dcn_mm_env()
{
    return !((g_dbg_block >> 28) & 1);   // bit 28 of the debug word: the kernels of dcn_kernels.h only (tests, A/B runs)
}

static bool mm_common_ok(const DcnArgs &a)
{
    if (math_np() == 0 || a.groups != 1 || a.dg < 1 || a.C % a.dg != 0 || !dcn_mm_env()) return false;
    for (int i = 0; i < a.nlv; ++i) {   // 32-bit buffer offsets
        if ((int64_t)a.lv[i].B * a.lv[i].H * a.lv[i].W * a.C * 4 >= ((int64_t)1 << 31)) return false;
        if ((int64_t)a.lv[i].P * a.opitch * 4 >= ((int64_t)1 << 31)) return false;
    }
    return true;
}

static int mm_npl() { return math_np() == 6 ? 3 : 2; }

static bool mm_fwd_ok(const DcnArgs &a)
{
    if (!mm_common_ok(a) || (a.C / a.dg) % 32 != 0 || a.Co % 128 != 0) return false;
    if (dcn_fwd_mm_lds_bytes(mm_npl(), a.kh * a.kw * a.dg) > 80 * 1024) return false;
    return cv_wfrag_bytes(a.Co, a.kh * a.kw, a.C, mm_npl()) < ((size_t)1 << 31);
}

// backward-data GEMM as a 1x1 convolution (conv.hip conv_mm_rows) + corner sums in the gather pass: every level that
// wants offset / mask gradients must also want grad_input (its samples are then on the anchor lists); the fragment image
// has to fit the caller's weight workspace (8 bytes per weight element, include/lsnet_hip.h)
int conv_mm_rows(int n, const float *const *x, float *const *out, const int *rows, int Cr, int xpitch, int N,
                 const unsigned short *wf, hipStream_t st);
static bool mm_bwd_ok(const DcnArgs &a)
{
    if (!mm_common_ok(a) || a.C % 4 != 0 || a.Co % 4 != 0) return false;
    for (int i = 0; i < a.nlv; ++i)
        if ((a.lv[i].goff || a.lv[i].gmsk) && !a.lv[i].gx) return false;
    const size_t wb = cv_wfrag_bytes(a.kh * a.kw * a.C, 1, a.Co, mm_npl());
    return wb < ((size_t)1 << 31) && wb <= (size_t)8 * a.Co * a.kh * a.kw * a.C;
}

// weight image in fragment order into `dst` (forward: (Co, K, C) as it lies; backward: the transposed 1x1 view with
// N = K * C columns and the reduction over Co)
// (prepared: the caller's image is already there -- lsn_dcn_shape.weights_prepared)
static int mm_prepare_weights(DcnArgs &a, bool backward, void *dst, hipStream_t st, bool prepared = false)
{
    const int K = a.kh * a.kw;
    const int Co = backward ? K * a.C : a.Co, Kd = backward ? 1 : K, C = backward ? a.Co : a.C;
    TapSub ts = {0, 1, 1, 0, 1, 1, 1};
    unsigned short *out = reinterpret_cast<unsigned short *>(dst);
    if (!prepared) {
        WfragJob j = {};
        j.w = a.w, j.out = out, j.Co = Co, j.K = Kd, j.C = C, j.flipT = backward ? 1 : 0, j.ts = ts;
        const long long total = wfrag_threads(j);
        const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        if (mm_npl() == 3)
            hipLaunchKernelGGL(conv_wfrag_kernel<3>, dim3(blocks), dim3(256), 0, st, j);
        else
            hipLaunchKernelGGL(conv_wfrag_kernel<2>, dim3(blocks), dim3(256), 0, st, j);
        LSN_HIP(hipGetLastError());
    }
    a.wtp = out;
    a.wtp_bytes = (int)cv_wfrag_bytes(Co, Kd, C, mm_npl());
    a.mm = 1;
    return 0;
}

// Work distribution of a forward launch (dcn_mm_kernels.h DcnSk).  The chip holds 512 workgroups of the kernel; whole rounds
// stay one workgroup per tile, the r tiles of the last round are cut into pieces of >= 4 chunks spread evenly over up to 512
// workgroups.  Measured (tools/ubench/dcn_step, debug bit 19 = whole tiles only, profiles/r6_dcn_sk.txt): the pyramid launch
// (2 100 tiles: four rounds and 52 tiles) 780 -> 759 us; the tower launch (700 tiles: one round and 188 tiles) 278 -> 283 us --
// its second round runs one workgroup per CU, which has the matrix pipe to itself and finishes in ~0.65 of a round, so the
// even split has little to return and the pieces' table rebuild and hand-over cost more.  Hence: launches of two rounds and
// more only (the opposite of the dense kernel's rule, conv.hip sk_plan, whose short tiles lose to the longer prologue there).
int conv_sk_scratch(size_t floats, float **part, unsigned **cnt, hipStream_t st);
int conv_sk_max_tiles();
static bool dcn_sk_env() { return !((g_dbg_block >> 19) & 1); }   // debug bit 19: whole tiles only (A/B)
static void dcn_sk_plan(int ntw, int Tall, DcnSk *sk)
{
    sk->n_dp = ntw, sk->sk_n = 0, sk->sk_tiles = 0, sk->part = nullptr, sk->cnt = nullptr;
    constexpr int SLOTS = 512;
    const bool forced = (g_dbg_block >> 18) & 1;   // debug bit 18: pieces for launches of any size (tests)
    // launches of less than HALF a round (a backbone layer of configs 3 / 4: 8 400 pixels = 132 tiles, 2 100 = 33) leave most
    // of the chip idle as whole tiles: up to four pieces per tile (the dense kernel's rule); between half a round and two
    // rounds (the tower launch) whole tiles win, see above
    if (!dcn_sk_env() || Tall < 8 || (ntw >= SLOTS / 2 && ntw < 2 * SLOTS && !forced)) return;
    const int r = ntw % SLOTS;
    if (r == 0 || r > 448) return;
    int per_tile = Tall / 4;
    const int cap = ntw > SLOTS ? 16 : 4;
    if (per_tile > cap) per_tile = cap;
    const long long pieces = (long long)r * per_tile;
    const int ns = pieces < SLOTS ? (int)pieces : SLOTS;
    if (ns <= r) return;
    if (r > conv_sk_max_tiles() || ((long long)r * Tall + 1) * ns >= ((long long)1 << 31)) return;
    sk->n_dp = ntw - r, sk->sk_n = ns, sk->sk_tiles = r;
}

template <int TM, int TN, int WM, int WN, int NP, bool FINE = false>
static int launch_fwd_mm_cfg(const DcnArgs &a, hipStream_t st)
{
    constexpr int BN = WN * TN * 32;
    const size_t lds = dcn_fwd_mm_lds_bytes(SplitCfg<NP>::NPL, a.kh * a.kw * a.dg);
    const int blocks = a.ntiles * (a.Co / BN);
    DcnSk sk;
    dcn_sk_plan(blocks, a.kh * a.kw * (a.C / 32), &sk);
    if (sk.sk_n) {
        if (int rc = conv_sk_scratch((size_t)2 * sk.sk_n * 64 * BN, &sk.part, &sk.cnt, st)) return rc;
        auto k = dcn_fwd_mm_kernel<TM, TN, WM, WN, NP, FINE, true>;
        if (int rc = set_lds(k, lds)) return rc;
        hipLaunchKernelGGL(k, dim3(sk.n_dp + sk.sk_n), dim3(256), lds, st, a, a.wtp, a.wtp_bytes, sk);
    } else {
        auto k = dcn_fwd_mm_kernel<TM, TN, WM, WN, NP, FINE, false>;
        if (int rc = set_lds(k, lds)) return rc;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, st, a, a.wtp, a.wtp_bytes, sk);
    }
    LSN_HIP(hipGetLastError());
    return 0;
}

static int launch_forward_mm(const DcnArgs &a, hipStream_t st)
{
    ProfScope prof(PROF_FWD, a, st);
    // (the 64 x 256 tile in the fine MFMA / staging interleave, dcn_mm_kernels.h FINE: tower launch 310 -> 280 us, pyramid
    // 818 -> 780 us in the round-4 A/B, profiles/r4_fine.txt)
    const bool wide = a.Co % 256 == 0;
    if (math_np() == 6) return wide ? launch_fwd_mm_cfg<2, 2, 1, 4, 6, true>(a, st) : launch_fwd_mm_cfg<1, 2, 2, 2, 6>(a, st);
    return wide ? launch_fwd_mm_cfg<2, 2, 1, 4, 3, true>(a, st) : launch_fwd_mm_cfg<1, 2, 2, 2, 3>(a, st);
}

template <int BM, int BN, int WM, int WN>
static int launch_forward_t(const DcnArgs &a, hipStream_t st)
{
    const int Cog = a.Co / a.groups, KD = a.kh * a.kw * a.dg;
    const size_t lds = (size_t)(BM + BN) * 33 * 4 + (size_t)BM * KD * sizeof(Tap);
    dim3 grid(a.ntiles, cdiv(Cog, BN), a.groups);
    if (vec_ok(a)) {
        auto k = dcn_fwd_kernel<BM, BN, WM, WN, true>;
        if (int rc = set_lds(k, lds)) return rc;
        hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a);
    } else {
        auto k = dcn_fwd_kernel<BM, BN, WM, WN, false>;
        if (int rc = set_lds(k, lds)) return rc;
        hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a);
    }
    LSN_HIP(hipGetLastError());
    return 0;
}

// the split one-workgroup-per-CU kernel (dcn_fwd_xn_kernel) needs float4 rows and 32-bit byte offsets into x and w
static bool xn_ok(const DcnArgs &a)
{
    if (a.Co / a.groups <= 64 || !vec_ok(a)) return false;
    if ((int64_t)a.Co * a.kh * a.kw * (a.C / a.groups) * 4 >= (int64_t)1 << 31) return false;
    for (int i = 0; i < a.nlv; ++i)
        if ((int64_t)a.lv[i].B * a.lv[i].H * a.lv[i].W * a.C * 4 >= (int64_t)1 << 31) return false;
    return !((g_dbg_block >> 29) & 1);   // bit 29 of the debug word forces the two-workgroup variant (A/B runs)
}

// ---- grouped calls (dcn_grouped_kernels.h): ResNeXt's 64 groups of 8 / 16 / 32 channels ----
static bool grouped_env() { return !((g_dbg_block >> 17) & 1); }   // debug bit 17: the general kernels for grouped calls (tests, A/B)

static bool grouped_fwd_ok(const DcnArgs &a)
{
    if (a.groups <= 1 || !grouped_env() || a.C % a.groups != 0 || a.Co != a.C) return false;
    const int cg = a.C / a.groups;
    if (!(cg == 8 || cg == 16 || cg == 32) || a.C % GF_CH != 0 || a.C % a.dg != 0 || (a.C / a.dg) % GF_CH != 0) return false;
    if (dcn_fwd_grouped_lds_bytes(a.kh * a.kw) > 64 * 1024) return false;
    for (int i = 0; i < a.nlv; ++i)
        if ((int64_t)a.lv[i].B * a.lv[i].H * a.lv[i].W * a.C >= ((int64_t)1 << 31) || (reinterpret_cast<uintptr_t>(a.lv[i].x) & 15) != 0)
            return false;
    return true;
}

static int launch_forward_grouped(const DcnArgs &a_in, hipStream_t st)
{
    DcnArgs a = a_in;
    int tiles = 0;
    for (int i = 0; i < a.nlv; ++i) {
        a.lv[i].tile0 = tiles;
        tiles += cdiv(a.lv[i].P, GF_PX);
    }
    a.ntiles = tiles;
    ProfScope prof(PROF_FWD, a, st);
    const size_t lds = dcn_fwd_grouped_lds_bytes(a.kh * a.kw);
    const dim3 grid(tiles * (a.C / GF_CH));
    auto go = [&](auto kern) -> int {
        if (int rc = set_lds(kern, lds)) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
        LSN_HIP(hipGetLastError());
        return 0;
    };
    switch (a.C / a.groups) {
    case 8: return go(dcn_fwd_grouped_kernel<8>);
    case 16: return go(dcn_fwd_grouped_kernel<16>);
    default: return go(dcn_fwd_grouped_kernel<32>);
    }
}

static int launch_forward(const DcnArgs &a, hipStream_t st)
{
    if (a.mm) return launch_forward_mm(a, st);
    if (grouped_fwd_ok(a)) return launch_forward_grouped(a, st);
    const int np = math_np(), KDf = a.kh * a.kw * a.dg;
    // exact fp32 (LSN_MATH_FP32, narrow outputs, odd channel counts): the fp32-MFMA kernel with two workgroups per CU
    if (np == 0 || !xn_ok(a) || (np == 6 ? xn_lds_bytes<6>(KDf) : xn_lds_bytes<3>(KDf)) > 160 * 1024) {
        ProfScope prof(PROF_FWD, a, st);
        return (a.Co / a.groups <= 64) ? launch_forward_t<64, 64, 2, 2>(a, st) : launch_forward_t<64, 256, 1, 4>(a, st);
    }
    dim3 grid(a.ntiles, cdiv(a.Co / a.groups, PIPE_BN), a.groups);
    ProfScope prof(PROF_FWD, a, st);
    const size_t ldsn = np == 6 ? xn_lds_bytes<6>(KDf) : xn_lds_bytes<3>(KDf);
    auto gox = [&](auto kern) -> int {
        if (int rc = set_lds(kern, ldsn)) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(256), ldsn, st, a);
        LSN_HIP(hipGetLastError());
        return 0;
    };
    if (np == 6) return a.wtp ? gox(dcn_fwd_xn_kernel<true, 6>) : gox(dcn_fwd_xn_kernel<false, 6>);
    return a.wtp ? gox(dcn_fwd_xn_kernel<true, 3>) : gox(dcn_fwd_xn_kernel<false, 3>);
}

template <int RED>
static int launch_bwd_data_t(const DcnArgs &a, hipStream_t st)
{
    const int KD = a.kh * a.kw * a.dg;
    const size_t lds = (size_t)RED * 32 * 4 + (size_t)BWD_BM * KD * (sizeof(Tap) + 12);
    if (vec_ok(a)) {
        auto k = dcn_bwd_data_kernel<RED, true>;
        if (int rc = set_lds(k, lds)) return rc;
        hipLaunchKernelGGL(k, dim3(a.ntiles), dim3(256), lds, st, a);
    } else {
        auto k = dcn_bwd_data_kernel<RED, false>;
        if (int rc = set_lds(k, lds)) return rc;
        hipLaunchKernelGGL(k, dim3(a.ntiles), dim3(256), lds, st, a);
    }
    LSN_HIP(hipGetLastError());
    return 0;
}

static bool bwd_x3_ok(const DcnArgs &a)
{
    if (math_np() == 0 || a.wtp == nullptr || a.groups != 1) return false;
    if (a.Co > 256 || a.Co % 8 != 0 || a.C % 4 != 0) return false;
    if ((int64_t)a.kh * a.kw * a.C * a.Co * 6 >= ((int64_t)1 << 31)) return false;
    if (bwd_xn_lds_bytes(math_np(), a.kh * a.kw * a.dg) > 80 * 1024) return false;
    for (int i = 0; i < a.nlv; ++i) {   // 32-bit buffer offsets into the input and the level's column-gradient rows
        if ((int64_t)a.lv[i].B * a.lv[i].H * a.lv[i].W * a.C * 4 >= ((int64_t)1 << 31)) return false;
        if ((int64_t)a.lv[i].P * a.kh * a.kw * a.C * 4 >= ((int64_t)1 << 31)) return false;
    }
    return true;
}

static bool bwd_colbuf_env()
{
    return !((g_dbg_block >> 23) & 1);   // bit 23 of the debug word forces the atomic scatter kernels (tests)
}

// Exact-fp32 column gradients (dcn_gcol_grouped_kernel: fmaf chains per group) in front of the gather pass: grouped calls
// (config 4: 64 groups) in every math mode, and -- round 6 -- EVERY call of LSN_MATH_FP32: the exact mode used to scatter
// with fp32 atomics (round 1's kernels) and was the less reproducible, less exact of the modes; it now
// shares the atomic-free, bit-reproducible gather with the default mode and differs from it in the GEMM arithmetic only.
// Levels that want offset / mask gradients must want grad_input (anchor lists).
static bool bwd_grouped_ok(const DcnArgs &a)
{
    if (a.groups <= 1 && math_np() != 0) return false;
    if (a.C % a.groups != 0 || a.Co % a.groups != 0 || (a.C / a.groups) % 4 != 0 || a.C % a.dg != 0) return false;
    if ((a.C / a.dg) % 4 != 0 || !bwd_colbuf_env()) return false;
    for (int i = 0; i < a.nlv; ++i) {
        if ((a.lv[i].goff || a.lv[i].gmsk) && !a.lv[i].gx) return false;
        if ((int64_t)a.lv[i].B * a.lv[i].H * a.lv[i].W * a.C * 4 >= ((int64_t)1 << 31)) return false;
        if ((int64_t)a.lv[i].P * a.kh * a.kw * a.C * 4 >= ((int64_t)1 << 31)) return false;
    }
    return true;
}

// the atomic-free path (column gradients + gather): served by the dense GEMM (mm_bwd_ok: any Co) or by round 2's GEMM
static bool bwd_gather_ok(const DcnArgs &a)
{
    if (bwd_grouped_ok(a)) return true;
    if (a.wtp == nullptr) return false;
    if (bwd_x3_ok(a)) return true;
    if (!mm_bwd_ok(a)) return false;
    for (int i = 0; i < a.nlv; ++i)   // 32-bit offsets into the level's column-gradient rows
        if ((int64_t)a.lv[i].P * a.kh * a.kw * a.C * 4 >= ((int64_t)1 << 31)) return false;
    return true;
}

// ---- atomic-free grad_input: workspace plan of the bin / scan / fill / sort / gather sequence (dcn_gather_kernels.h) ----
struct GatherPlan {
    bool ok = false;
    int nsamples = 0, nanchors = 0;
    size_t scan_tmp = 0;
    size_t o_gcol = 0, o_cnt = 0, o_start = 0, o_anchor = 0, o_rank = 0, o_frac = 0, o_ent = 0, o_tmp = 0, o_gtap = 0, bytes = 0;
    size_t o_S = 0, o_H = 0, o_ent2 = 0;
    GatherArgs ga;        // groups gathered per 4x4 pixel block
    AnchorArgs aa;        // groups with long lists: gathered per anchor (dcn_anchor_sum / combine)
    int anchor_pixels = 0;
    int na_long = 0;             // the first na_long anchors of the S buffer belong to long-list groups (4 waves each)
    int64_t block_samples = 0;   // samples (upper bound) of the block-gathered groups
};

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// Fills prow0 / abase of the levels and the plan.  gx pointers: levels with the same grad_input share an anchor grid.
static void gather_plan(DcnArgs &a, GatherPlan &pl)
{
    pl.ok = false;
    const int K = a.kh * a.kw, KD = K * a.dg;
    int64_t prow = 0, anchors = 0, Q = 0;
    GatherArgs &ga = pl.ga;
    ga.ng = 0;
    bool any = false;
    int64_t gsamp[MAXLV] = {};   // samples (upper bound: every tap valid) that scatter into each group
    for (int i = 0; i < a.nlv; ++i) {
        Lvl &L = a.lv[i];
        L.prow0 = (int)prow;
        prow += L.P;
        L.abase = 0;
        if (!L.gx) continue;
        any = true;
        int gi = -1;
        for (int j = 0; j < ga.ng; ++j)
            if (ga.g[j].gx == L.gx) gi = j;
        if (gi < 0) {
            gi = ga.ng++;
            GatherGrp &G = ga.g[gi];
            G.gx = L.gx, G.x = L.x, G.B = L.B, G.H = L.H, G.W = L.W;
            G.abase = (int)anchors, G.blk0 = (int)Q;
            anchors += (int64_t)L.B * (L.H + 1) * (L.W + 1);
            Q += (int64_t)L.B * cdiv(L.H, GT) * cdiv(L.W, GT);
        } else if (ga.g[gi].B != L.B || ga.g[gi].H != L.H || ga.g[gi].W != L.W) {
            return;   // one buffer, two shapes: not a valid call for this path
        }
        L.abase = ga.g[gi].abase;
        gsamp[gi] += (int64_t)L.P * KD;
    }
    if (!any) return;
    if (prow * KD >= ((int64_t)1 << 30) || anchors >= ((int64_t)1 << 30) || Q >= ((int64_t)1 << 30)) return;
    if (a.C % 4 != 0 || (a.C / a.dg) % 4 != 0) return;
    pl.nsamples = (int)(prow * KD);
    pl.nanchors = (int)anchors;
    // Groups with more than ~400 samples per 4x4 pixel block go to the per-anchor gather (the pyramid op's coarse source
    // levels); the others keep the block walk.
    constexpr int anchor_thr = 400;
    AnchorArgs &aa = pl.aa;
    aa.ng = 0, aa.NA = 0;
    pl.anchor_pixels = 0, pl.block_samples = 0;
    {
        // anchor path: every group when the unweighted-GEMM pipeline serves the call (its corner sums are cheapest per
        // anchor), otherwise the groups above the threshold.  Long-list groups first: they get four waves per anchor.
        const bool all_anchor = mm_bwd_ok(a) || bwd_grouped_ok(a);
        GatherGrp keep[MAXLV];
        bool to_anchor[MAXLV], is_long[MAXLV];
        int nk = 0;
        int64_t q = 0;
        for (int j = 0; j < ga.ng; ++j) {
            const GatherGrp &G = ga.g[j];
            const int64_t nblk = (int64_t)G.B * cdiv(G.H, GT) * cdiv(G.W, GT);
            const int64_t nanch = (int64_t)G.B * (G.H + 1) * (G.W + 1);
            to_anchor[j] = all_anchor || (anchor_thr > 0 && gsamp[j] > (int64_t)anchor_thr * nblk);
            constexpr int long_thr = 40;
            is_long[j] = !all_anchor || gsamp[j] > (int64_t)long_thr * nanch;   // mean list length (upper bound)
        }
        for (int pass = 0; pass < 2; ++pass) {
            for (int j = 0; j < ga.ng; ++j) {
                const GatherGrp &G = ga.g[j];
                if (!to_anchor[j] || is_long[j] != (pass == 0)) continue;
                AnchorGrp &A = aa.g[aa.ng++];
                A.gx = G.gx, A.x = G.x, A.B = G.B, A.H = G.H, A.W = G.W, A.abase = G.abase, A.a0 = aa.NA;
                aa.NA += G.B * (G.H + 1) * (G.W + 1);
                pl.anchor_pixels += G.B * G.H * G.W;
            }
            if (pass == 0) pl.na_long = aa.NA;
        }
        for (int j = 0; j < ga.ng; ++j) {
            if (to_anchor[j]) continue;
            const GatherGrp &G = ga.g[j];
            keep[nk] = G;
            keep[nk].blk0 = (int)q;
            q += (int64_t)G.B * cdiv(G.H, GT) * cdiv(G.W, GT);
            pl.block_samples += gsamp[j];
            ++nk;
        }
        for (int j = 0; j < nk; ++j) ga.g[j] = keep[j];
        ga.ng = nk;
        Q = q;
    }
    ga.NB = (int)Q;
    ga.C = a.C, ga.K = K, ga.KD = KD, ga.dg = a.dg;
    aa.C = a.C, aa.K = K, aa.KD = KD, aa.dg = a.dg;
    size_t tmp = 0;
    if (rocprim::exclusive_scan((void *)nullptr, tmp, (int *)nullptr, (int *)nullptr, 0, (size_t)pl.nanchors + 1,
                                rocprim::plus<int>(), (hipStream_t)0) != hipSuccess)
        return;
    pl.scan_tmp = tmp;
    size_t o = 0;
    pl.o_gcol = o, o = align256(o + (size_t)prow * K * a.C * sizeof(float));
    pl.o_cnt = o, o = align256(o + ((size_t)pl.nanchors + 1) * sizeof(int));
    pl.o_start = o, o = align256(o + ((size_t)pl.nanchors + 1) * sizeof(int));
    pl.o_anchor = o, o = align256(o + (size_t)pl.nsamples * sizeof(int));
    pl.o_rank = o, o = align256(o + (size_t)pl.nsamples * sizeof(int));
    pl.o_frac = o, o = align256(o + (size_t)pl.nsamples * sizeof(float2));
    pl.o_ent = o, o = align256(o + (size_t)pl.nsamples * sizeof(GEntry));
    pl.o_tmp = o, o = align256(o + tmp + 256);
    pl.o_gtap = o, o = align256(o + ((size_t)pl.nsamples + 1) * sizeof(Tap));   // + the all-zero entry
    pl.o_S = o, o = align256(o + (size_t)pl.aa.NA * 4 * a.C * sizeof(float));
    // corner sums (dcn_offgrad_kernel): one slot per 256-channel block where the per-anchor sums split them over blockIdx.y
    // (every group on the per-anchor path, C > 256; dcn_gather_kernels.h AnchorArgs::ncb)
    // (debug bit 16: one wave walks all blocks of its anchor, the form until round 6 -- A/B runs; the workspace is sized for
    // the split either way, so a size asked for under one setting serves a launch under the other)
    const int ncb_max = (ga.NB == 0 && aa.ng > 0 && a.C > 256) ? cdiv(a.C, 256) : 1;
    aa.ncb = ((g_dbg_block >> 16) & 1) ? 1 : ncb_max;
    aa.hb_slot = (long long)pl.nsamples * 4;
    pl.o_H = o, o = align256(o + (size_t)pl.nsamples * 4 * sizeof(float) * ncb_max);
    pl.o_ent2 = o, o = align256(o + (size_t)pl.nsamples * sizeof(GEntry));   // ordering of the lists above 64 entries
    pl.bytes = o;
    pl.ok = true;
}


// Tap groups of the split backward-data GEMM (grid.y): the launch runs in ceil(tiles x groups / 512) rounds of the 512
// resident blocks (2 per CU) with blocks 1 / groups as long, plus a per-block prologue (gout tile -> registers, sampling
// table) of ~5 % of a whole tile (measured: 3 groups beat 9 on the tower launch): pick the divisor of K with the smallest
// rounds / groups x (1 + 0.05 groups).
static int bwd_tap_groups(const DcnArgs &a)
{
    const int K = a.kh * a.kw;
    int best = 1;
    double best_cost = 1e30;
    for (int g = 1; g <= K; ++g) {
        if (K % g) continue;
        const double cost = (double)cdiv(a.ntiles * g, 512) / g * (1.0 + 0.05 * g);
        if (cost < best_cost - 1e-9) best_cost = cost, best = g;
    }
    return best;
}

// A second stream per launch stream for the list building of the backward-data path.  bin -> scan -> fill -> sort are
// latency-bound kernels of a few thousand short workgroups (50 - 60 us each on the tower launch, 0.1 - 0.2 of the chip's
// wave slots busy) that depend on the offsets only; the column-gradient GEMM beside them depends on grad_output only and
// leaves ~100 VGPRs per SIMD and 110 KB of LDS per CU free.  Fork after the workspace is known, join before the
// per-anchor sums.  (Not the same experiment as the weight gradients on a side stream, profiles/r4_side_stream.txt: two
// MFMA kernels take the matrix pipe from each other; these kernels wait on memory while the GEMM computes.)
// (The TAIL of the path -- per-anchor sums, combine, offset gradients -- on this stream beside the weight-gradient pass of the
// same call was tried as well, profiles/r4_side_lists.txt: nothing on the tower launch, 995 vs 992 us per backward call (the
// gathers take the CUs from the weight gradient's workgroups instead of sharing them), -120 us on the pyramid launch,
// 0.25 ms per step -- and two kernel families that can no longer be timed apart.  Removed.)
struct SideStream {
    hipStream_t main = nullptr, side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
static SideStream g_side[16];
static int g_nside = 0;
static int side_stream(hipStream_t st, SideStream **out)
{
    for (int i = 0; i < g_nside; ++i)
        if (g_side[i].main == st) {
            *out = &g_side[i];
            return 0;
        }
    if (g_nside == 16) return fail(LSN_ERR_RUNTIME, "side streams: more than 16 streams use the library");
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return fail(LSN_ERR_RUNTIME, "side stream would be created inside a stream capture: run the step eagerly once before capturing");
    SideStream &S = g_side[g_nside];
    int lo = 0, hi = 0;
    LSN_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));   // hi = the numerically lowest = most urgent
    LSN_HIP(hipStreamCreateWithPriority(&S.side, hipStreamNonBlocking, hi));
    LSN_HIP(hipEventCreateWithFlags(&S.fork, hipEventDisableTiming));
    LSN_HIP(hipEventCreateWithFlags(&S.join, hipEventDisableTiming));
    S.main = st;
    ++g_nside;
    *out = &S;
    return 0;
}
static bool side_lists_env() { return !((g_dbg_block >> 21) & 1); }   // debug bit 21: lists on the launch stream (A/B)

// (Round 5 cut the launch into bands -- an anchor range per image of a grad_input map + the GEMM rows that scatter into it --
// and ran the per-anchor sums of band i on the side stream beside the GEMM of band i + 1, the column gradients of a band
// still in the Infinity Cache: tower +7 % slower, pyramid unchanged, profiles/r5_band_pipeline.txt.  GEMM and sums are both
// bound by the memory system; removed.)
template <int NP>
static int launch_bwd_colbuf(DcnArgs &a, GatherPlan &pl, unsigned char *ws, hipStream_t st_main)
{
    a.gcol = reinterpret_cast<float *>(ws + pl.o_gcol);
    int *cnt = reinterpret_cast<int *>(ws + pl.o_cnt), *start = reinterpret_cast<int *>(ws + pl.o_start);
    int *sanchor = reinterpret_cast<int *>(ws + pl.o_anchor), *srank = reinterpret_cast<int *>(ws + pl.o_rank);
    float2 *sfrac = reinterpret_cast<float2 *>(ws + pl.o_frac);
    GEntry *ent = reinterpret_cast<GEntry *>(ws + pl.o_ent);
    Tap *gtap = reinterpret_cast<Tap *>(ws + pl.o_gtap);   // also read by the GEMM below and by the weight-gradient pass
    a.gtap_rows = pl.nsamples / (a.kh * a.kw * a.dg);
    // lists on the side stream only beside the dense GEMM (the other column-gradient kernels read the tap table)
    SideStream *side = nullptr;
    if (a.mm && a.groups == 1 && side_lists_env())
        if (int rc = side_stream(st_main, &side)) return rc;
    hipStream_t st = side ? side->side : st_main;
    if (side) {
        LSN_HIP(hipEventRecord(side->fork, st_main));
        LSN_HIP(hipStreamWaitEvent(side->side, side->fork, 0));
    }
    LSN_HIP(hipMemsetAsync(cnt, 0, ((size_t)pl.nanchors + 1) * sizeof(int), st));
    hipLaunchKernelGGL(dcn_bin_kernel, dim3(cdiv(pl.nsamples, 256)), dim3(256), 0, st, a, pl.nsamples, cnt, sanchor, srank, sfrac,
                       gtap);
    a.gtap = gtap;
    size_t tmp = pl.scan_tmp;
    LSN_HIP(rocprim::exclusive_scan(ws + pl.o_tmp, tmp, cnt, start, 0, (size_t)pl.nanchors + 1, rocprim::plus<int>(), st));
    hipLaunchKernelGGL(dcn_fill_kernel, dim3(cdiv(pl.nsamples, 256)), dim3(256), 0, st, pl.nsamples, start, sanchor, srank, sfrac,
                       ent, gtap, a.kh * a.kw * a.dg, a.gtap_rows);
    int sort_blocks = cdiv(pl.nanchors, 4);
    if (sort_blocks > 4096) sort_blocks = 4096;
    hipLaunchKernelGGL(dcn_sort_lists_kernel, dim3(sort_blocks), dim3(256), 0, st, pl.nanchors, start, ent,
                       reinterpret_cast<GEntry *>(ws + pl.o_ent2));
    if (side) {
        LSN_HIP(hipEventRecord(side->join, side->side));
        st = st_main;
    }
    float *Hb = nullptr;
    const bool grouped = a.groups > 1 || NP == 0;   // (NP = 0: LSN_MATH_FP32, every call)
    if (grouped) {   // exact-fp32 column gradients per group, unweighted; the gather pass does the rest as for the dense GEMM
        int ks = 0;   // k-steps of the fp32-MFMA form (dcn_grouped_kernels.h dcn_gcol_mfma_kernel); 0: the fmaf-chain kernel
        const int Cg = a.C / a.groups, Cog = a.Co / a.groups;
        if (grouped_env() && a.C % GC_COLS == 0 && a.opitch % 4 == 0) {
            if (a.groups == 1 && (a.Co == 64 || a.Co == 128 || a.Co == 256)) ks = a.Co / 4;
            if (a.groups > 1 && Cog == Cg && (Cg == 4 || Cg == 8 || Cg == 16 || Cg == 32)) ks = Cg == 32 ? 8 : 4;
            for (int i = 0; i < a.nlv; ++i)
                if ((reinterpret_cast<uintptr_t>(a.lv[i].gout) & 15) != 0) ks = 0;
        }
        if (ks) {
            DcnArgs g = a;
            int tiles = 0;
            for (int i = 0; i < g.nlv; ++i) {
                g.lv[i].tile0 = tiles;
                tiles += cdiv(g.lv[i].P, GC_PX);
            }
            const int nred = a.groups == 1 ? a.Co : GC_COLS;
            const size_t lds = dcn_gcol_mfma_lds_bytes(nred);
            const dim3 grid(tiles * (a.C / GC_COLS));
            auto go = [&](auto kern) -> int {
                if (int rc = set_lds(kern, lds)) return rc;
                hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, g, nred);
                return 0;
            };
            int rc = 0;
            switch (ks) {
            case 4: rc = go(dcn_gcol_mfma_kernel<4>); break;
            case 8: rc = go(dcn_gcol_mfma_kernel<8>); break;
            case 16: rc = go(dcn_gcol_mfma_kernel<16>); break;
            case 32: rc = go(dcn_gcol_mfma_kernel<32>); break;
            default: rc = go(dcn_gcol_mfma_kernel<64>); break;
            }
            if (rc) return rc;
        } else {
            const long long nquads = (long long)pl.nsamples / a.dg * (a.C / 4);
            const int blocks = (int)((nquads + 255) / 256 < 16384 ? (nquads + 255) / 256 : 16384);
            hipLaunchKernelGGL(dcn_gcol_grouped_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, a, nquads);
        }
        for (int i = 0; i < a.nlv; ++i)
            if (a.lv[i].goff || a.lv[i].gmsk) Hb = reinterpret_cast<float *>(ws + pl.o_H);
    } else if (a.mm) {   // the dense kernel writes the unweighted column gradients; everything else happens in the gather pass
        const float *xs[MAXLV];
        float *outs[MAXLV];
        int rows[MAXLV];
        bool any_off = false;
        for (int i = 0; i < a.nlv; ++i) {
            xs[i] = a.lv[i].gout;
            outs[i] = a.gcol + (size_t)a.lv[i].prow0 * a.kh * a.kw * a.C;
            rows[i] = a.lv[i].P;
            any_off = any_off || a.lv[i].goff || a.lv[i].gmsk;
        }
        if (int rc = conv_mm_rows(a.nlv, xs, outs, rows, a.Co, a.opitch, a.kh * a.kw * a.C, a.wtp, st)) return rc;
        if (side) LSN_HIP(hipStreamWaitEvent(st_main, side->join, 0));
        if (any_off) Hb = reinterpret_cast<float *>(ws + pl.o_H);
    } else if constexpr (NP != 0) {
        size_t lds = bwd_xn_lds_bytes(NP, a.kh * a.kw * a.dg);
        if (int rc = set_lds(dcn_bwd_data_xn_kernel<NP, true>, lds)) return rc;
        hipLaunchKernelGGL((dcn_bwd_data_xn_kernel<NP, true>), dim3(a.ntiles, bwd_tap_groups(a)), dim3(256), lds, st, a);
    } else {
        return fail(LSN_ERR_RUNTIME, "deformable backward: exact mode without the exact column-gradient kernel");
    }
    pl.ga.raw = pl.aa.raw = (a.mm || grouped) ? 1 : 0;
    pl.ga.Hb = pl.aa.Hb = Hb;
    pl.ga.gcol = a.gcol, pl.ga.start = start, pl.ga.ent = ent;
    if (pl.aa.ng > 0) {   // per-anchor sums of all four corners, then four of them per pixel
        pl.aa.gcol = a.gcol, pl.aa.start = start, pl.aa.ent = ent;
        pl.aa.S = reinterpret_cast<float *>(ws + pl.o_S);
        if (pl.na_long > 0)
            hipLaunchKernelGGL(dcn_anchor_sum_kernel<4>, dim3(pl.na_long, pl.aa.ncb), dim3(256), 0, st, pl.aa, 0, pl.na_long);
        if (pl.aa.NA > pl.na_long)
            hipLaunchKernelGGL(dcn_anchor_sum_kernel<1>, dim3(cdiv(pl.aa.NA - pl.na_long, 4), pl.aa.ncb), dim3(256), 0, st, pl.aa,
                               pl.na_long, pl.aa.NA - pl.na_long);
        hipLaunchKernelGGL(dcn_anchor_combine_kernel, dim3(cdiv(pl.anchor_pixels, 4)), dim3(256), 0, st, pl.aa, pl.anchor_pixels);
    }
    if (pl.ga.NB > 0) {
        // medium lists: four waves per pixel block; short ones (the tower launch, ~140 entries per block): one
        const bool split = pl.block_samples > (int64_t)300 * pl.ga.NB;
        if (split)
            hipLaunchKernelGGL(dcn_gather_kernel<4>, dim3(pl.ga.NB), dim3(256), 0, st, pl.ga);
        else
            hipLaunchKernelGGL(dcn_gather_kernel<1>, dim3(cdiv(pl.ga.NB, 4)), dim3(256), 0, st, pl.ga);
    }
    if (Hb)
        hipLaunchKernelGGL(dcn_offgrad_kernel, dim3(cdiv(pl.nsamples, 256)), dim3(256), 0, st, a, pl.nsamples,
                           reinterpret_cast<const float4 *>(Hb), pl.aa.ng > 0 ? pl.aa.ncb : 1);
    LSN_HIP(hipGetLastError());
    return 0;
}

// gather_ws: device scratch of >= the plan's byte count for the atomic-free path, or NULL
static int launch_bwd_data(DcnArgs &a, void *gather_ws, size_t gather_ws_bytes, hipStream_t st)
{
    ProfScope prof(PROF_BWD_DATA, a, st);
    const int np = math_np();
    if ((a.mm || bwd_x3_ok(a) || bwd_grouped_ok(a)) && gather_ws && bwd_colbuf_env()) {
        GatherPlan pl;
        gather_plan(a, pl);
        if (pl.ok && pl.bytes <= gather_ws_bytes) {
            unsigned char *ws = reinterpret_cast<unsigned char *>(gather_ws);
            return np == 3 ? launch_bwd_colbuf<3>(a, pl, ws, st) : np == 6 ? launch_bwd_colbuf<6>(a, pl, ws, st)
                                                                           : launch_bwd_colbuf<0>(a, pl, ws, st);
        }
    }
    if (a.mm) return fail(LSN_ERR_RUNTIME, "deformable backward: fragment-order weights without the gather path");
    a.gcol = nullptr, a.gtap = nullptr;
    for (int i = 0; i < a.nlv; ++i)   // the scatter kernels accumulate: start from zero (once per buffer is enough)
        if (a.lv[i].gx)
            LSN_HIP(hipMemsetAsync(a.lv[i].gx, 0, sizeof(float) * (size_t)a.lv[i].B * a.lv[i].H * a.lv[i].W * a.C, st));
    if (bwd_x3_ok(a)) {
        const size_t lds = bwd_xn_lds_bytes(np, a.kh * a.kw * a.dg);
        auto gox = [&](auto kern) -> int {
            if (int rc = set_lds(kern, lds)) return rc;
            hipLaunchKernelGGL(kern, dim3(a.ntiles, bwd_tap_groups(a)), dim3(256), lds, st, a);
            LSN_HIP(hipGetLastError());
            return 0;
        };
        return np == 6 ? gox(dcn_bwd_data_xn_kernel<6, false>) : gox(dcn_bwd_data_xn_kernel<3, false>);
    }
    return (a.Co / a.groups > 64) ? launch_bwd_data_t<256>(a, st) : launch_bwd_data_t<64>(a, st);
}

// Pixel splits of a weight-gradient launch: the grid (columns x splits x co blocks) fills ONE round of the blocks the chip
// holds at once (256 CUs x per_cu), rounded DOWN.  Measured on the benchmark step (profiles/r2_wgrad_rounds.txt): the
// old cdiv(1024, 36) = 29 splits made 1044 blocks = two full rounds plus a third with 20 blocks (0.88 ms per DCN
// launch); 28 splits 0.68 ms; one round instead of two is the same for the DCN launches and 1.1 ms / step faster over
// the dense layers (half the fp32 atomics of the epilogues).
static int wgrad_splits(int cols, int per_cu)
{
    const int slots = 256 * per_cu;
    int s = slots / cols;
    return s < 1 ? 1 : s;
}

// 8-byte buffer loads in the split weight-gradient kernel: even channel counts, 8-byte aligned tensors, byte offsets < 2^31
static int wgrad_vec_bits(const DcnArgs &a)
{
    bool vx = a.C % 2 == 0, vg = a.Co % 2 == 0;
    for (int i = 0; i < a.nlv; ++i) {
        const Lvl &L = a.lv[i];
        vx = vx && (reinterpret_cast<uintptr_t>(L.x) & 7) == 0 && (long long)L.B * L.H * L.W * a.C < (1ll << 29);
        vg = vg && (reinterpret_cast<uintptr_t>(L.gout) & 7) == 0 && (long long)(L.P + WG_BP) * a.Co < (1ll << 29);
    }
    return (vx ? 1 : 0) | (vg ? 2 : 0);
}

// ---- weight gradient on the kernels of dcn_mm_kernels.h: grad_output in fragment order, split partial tiles ----
int conv_wgrad_reduce(const float *part, float *gw, size_t n, const float *part_b, float *gb, int nb, int splits, int splits_b,
                      int accumulate, hipStream_t st);
bool conv_wgrad_fold_pending();   // conv.hip: lsn_conv2d_backward_weight_bn is the caller
int conv_scratch(size_t floats, float **p, hipStream_t st);

static bool mm_wgrad_ok(const DcnArgs &a)
{
    if (!mm_common_ok(a) || a.Co % 256 != 0 || (a.C / a.dg) % 64 != 0) return false;
    if (a.gtap == nullptr || (reinterpret_cast<uintptr_t>(a.gtap) & 15) != 0) return false;
    return true;
}

// dense: a dense convolution's weight gradient (levels without offsets, no backward-data pass before this one): the
// sampling table is built here, and the launch is accounted to the caller's family
static int launch_wgrad_mm(const DcnArgs &a_in, int nchunks, bool accumulate, hipStream_t st, bool dense = false)
{
    DcnArgs a = a_in;
    const int K = a.kh * a.kw, npl = mm_npl();
    const size_t nW = (size_t)a.Co * K * a.C;
    const size_t img_bytes = (size_t)nchunks * 2 * (a.Co / 32) * npl * 1024;
    if (img_bytes >= ((size_t)1 << 31)) return 1;
    const int ncol = K * (a.C / 64), nz = a.Co / 256;
    int S = (512 + ncol * nz / 2) / (ncol * nz);
    if (S > nchunks / 4) S = nchunks / 4;   // a split should run long enough to amortise its prologue and its partial tile
    if (S < 1) S = 1;
    const int nsteps16 = 2 * nchunks;
    int spb = cdiv(nsteps16, 256);            // pre-pass: ~256 step ranges x (Co / 128) tile groups
    if (spb < 4) spb = 4;
    const int nblk_s = cdiv(nsteps16, spb);
    const size_t img_f = (img_bytes + 15) / 16 * 4, part_f = (size_t)S * nW, pb_f = ((size_t)nblk_s * a.Co + 3) & ~(size_t)3;
    const size_t meta_f = ((size_t)nchunks * 8 + 64 + 3) & ~(size_t)3;
    size_t tap_entries = 0;
    if (dense) {
        int64_t rows = 0;
        for (int i = 0; i < a.nlv; ++i) {
            a.lv[i].prow0 = (int)rows;
            rows += a.lv[i].P;
        }
        if (rows * K * a.dg >= ((int64_t)1 << 26)) return 1;   // (a 2 GB table)
        a.gtap_rows = (int)rows;
        tap_entries = (size_t)rows * K * a.dg;
    }
    float *base = nullptr;
    if (int rc = conv_scratch(img_f + part_f + pb_f + meta_f + (tap_entries ? (tap_entries + 1) * (sizeof(Tap) / 4) : 0), &base, st))
        return rc;
    unsigned short *img = reinterpret_cast<unsigned short *>(base);
    float *part = base + img_f, *part_b = part + part_f;
    int *meta = reinterpret_cast<int *>(part_b + pb_f);
    if (dense) {
        Tap *tab = reinterpret_cast<Tap *>(part_b + pb_f + meta_f);
        const int n = (int)tap_entries > nchunks ? (int)tap_entries : nchunks;
        hipLaunchKernelGGL(dcn_tap_table_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, a, (int)tap_entries, tab, nchunks, meta);
        a.gtap = tab;
    } else {
        hipLaunchKernelGGL(dcn_chunk_meta_kernel, dim3(cdiv(nchunks, 256)), dim3(256), 0, st, a, nchunks, meta);
    }
    ProfScope prof(dense ? -1 : PROF_WGRAD, a, st);
    auto pre = npl == 3 ? dcn_gout_frag_kernel<3> : dcn_gout_frag_kernel<2>;
    hipLaunchKernelGGL(pre, dim3(nblk_s, a.Co / 128), dim3(256), 0, st, a, nsteps16, spb, img, a.gb ? part_b : nullptr);
    const size_t lds = dcn_wgrad_mm_lds_bytes(npl);
    if ((long long)nchunks * (S + 1) >= (1ll << 32))
        return fail(LSN_ERR_UNSUPPORTED, "deformable backward-weight: %d chunks x %d splits exceed the kernel's 32-bit split arithmetic", nchunks, S);
    auto go = [&](auto kern) -> int {
        if (int rc = set_lds(kern, lds)) return rc;
        hipLaunchKernelGGL(kern, dim3(ncol, S, nz), dim3(256), lds, st, a, nchunks, img, (int)img_bytes, part, meta);
        LSN_HIP(hipGetLastError());
        return 0;
    };
    // (FINE = the fine MFMA / staging interleave: tower 360 -> 345 us, pyramid 941 -> 916 us, profiles/r4_fine.txt)
    if (dense) {
        if (int rc = (math_np() == 6 ? go(dcn_wgrad_mm_kernel<6, true, true>) : go(dcn_wgrad_mm_kernel<3, true, true>))) return rc;
    } else if (int rc = (math_np() == 6 ? go(dcn_wgrad_mm_kernel<6, false, true>) : go(dcn_wgrad_mm_kernel<3, false, true>))) {
        return rc;
    }
    return conv_wgrad_reduce(part, a.gw, nW, part_b, a.gb, a.Co, S, nblk_s, accumulate ? 1 : 0, st);
}

static bool grouped_wgrad_ok(const DcnArgs &a)
{
    if (a.groups <= 1 || !grouped_env() || a.C % a.groups != 0 || a.Co != a.C || a.kh * a.kw > 9) return false;
    const int cg = a.C / a.groups;
    if (!(cg == 4 || cg == 8 || cg == 16 || cg == 32) || ((int64_t)a.C * cg) % 256 != 0 || a.C % a.dg != 0) return false;
    const int nch = cg >= 16 ? cg : 256 / cg;
    if ((a.C / a.dg) % nch != 0 || a.C % nch != 0) return false;
    for (int i = 0; i < a.nlv; ++i)
        if ((int64_t)a.lv[i].B * a.lv[i].H * a.lv[i].W * a.C >= ((int64_t)1 << 31) || (reinterpret_cast<uintptr_t>(a.lv[i].x) & 15) != 0)
            return false;
    return true;
}

// grouped weight gradient (dcn_grouped_kernels.h): pixel splits leave partial tiles, the ordered reduce adds them (or queues)
static int launch_wgrad_grouped(const DcnArgs &a_in, bool accumulate, hipStream_t st)
{
    DcnArgs a = a_in;
    int64_t rows = 0;
    for (int i = 0; i < a.nlv; ++i) {
        a.lv[i].prow0 = (int)rows;      // (the numbering of the backward-data pass's table, when there is one)
        rows += a.lv[i].P;
    }
    if (rows >= ((int64_t)1 << 30)) return 1;
    if ((reinterpret_cast<uintptr_t>(a.gtap) & 15) != 0 || a.gtap_rows != (int)rows) a.gtap = nullptr;
    const int K = a.kh * a.kw, cg = a.C / a.groups;
    const int ny = (int)((int64_t)a.C * cg / 256);
    const int nch = cg >= 16 ? cg : 256 / cg, nco = 256 / cg;
    int splits = cdiv(3072, ny);                               // ~12 workgroups per CU: the kernel lives on loads in flight
    int per = cdiv((int)rows, splits);
    per = cdiv(per, GRP_PB) * GRP_PB;
    if (per < 4 * GRP_PB) per = 4 * GRP_PB;
    splits = cdiv((int)rows, per);
    const size_t nW = (size_t)a.Co * K * cg;
    const size_t cap = ((size_t)256 << 20) / 4 / (nW + a.Co);  // partial tiles: at most 256 MB
    if ((size_t)splits > cap) {
        splits = (int)(cap > 0 ? cap : 1);
        per = cdiv(cdiv((int)rows, splits), GRP_PB) * GRP_PB;
        splits = cdiv((int)rows, per);
    }
    float *base = nullptr;
    if (int rc = conv_scratch((size_t)splits * (nW + a.Co) + 16, &base, st)) return rc;
    float *part = base, *part_b = a.gb ? base + (((size_t)splits * nW + 3) & ~(size_t)3) : nullptr;
    ProfScope prof(PROF_WGRAD, a, st);
    const size_t lds = dcn_wgrad_grouped_lds_bytes(K, nch, nco);
    const dim3 grid(splits, ny);
    auto go = [&](auto kern) -> int {
        if (int rc = set_lds(kern, lds)) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a, (int)rows, per, part, part_b);
        LSN_HIP(hipGetLastError());
        return 0;
    };
    int rc;
    switch (cg) {
    case 4: rc = go(dcn_wgrad_grouped_kernel<4, 9>); break;
    case 8: rc = go(dcn_wgrad_grouped_kernel<8, 9>); break;
    case 16: rc = go(dcn_wgrad_grouped_kernel<16, 9>); break;
    default: rc = go(dcn_wgrad_grouped_kernel<32, 9>); break;
    }
    if (rc) return rc;
    return conv_wgrad_reduce(part, a.gw, nW, part_b, a.gb, a.Co, splits, splits, accumulate ? 1 : 0, st);
}

static int launch_wgrad(const DcnArgs &a_in, int nsteps, bool accumulate, hipStream_t st)
{
    if (grouped_wgrad_ok(a_in) && a_in.opitch == a_in.Co && ((size_t)a_in.Co * a_in.kh * a_in.kw * (a_in.C / a_in.groups)) % 4 == 0) {
        const int rc = launch_wgrad_grouped(a_in, accumulate, st);
        if (rc != 1) return rc;
    }
    if (mm_wgrad_ok(a_in)) {
        const int rc = launch_wgrad_mm(a_in, nsteps, accumulate, st);
        if (rc != 1) return rc;   // 1: not served (sizes), fall through
    }
    if (a_in.opitch != a_in.Co)
        return fail(LSN_ERR_UNSUPPORTED, "deformable weight gradient: out_pitch outside the fragment-order kernel");
    DcnArgs a = a_in;
    a.wg_vec = wgrad_vec_bits(a);
    if ((reinterpret_cast<uintptr_t>(a.gtap) & 15) != 0) a.gtap = nullptr;
    const int K = a.kh * a.kw, Cg = a.C / a.groups, Cog = a.Co / a.groups;
    const int segs = Cg / a.SL, ncc = cdiv(a.SL, WG_BN);
    const int ncol = a.groups * K * segs * ncc, nz = cdiv(Cog, WG_BM);
    int splits = wgrad_splits(ncol * nz, 2);   // 2 resident blocks per CU (246 VGPRs, 78 KB LDS)
    if (splits > nsteps) splits = nsteps;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    const size_t lds = (size_t)WG_BP * (WG_BM + WG_BN) * 4 + 2 * WG_BP * sizeof(Tap);
    // split-bf16 kernels: one partial gradient per pixel split + an ordered reduce instead of fp32 atomics (deterministic);
    // a block that runs no step still stores its zero tile, so every partial element is written
    const size_t nW = (size_t)a.Co * K * Cg;
    const bool ordered = !((g_dbg_block >> 30) & 1) && nW % 4 == 0 &&
                         (size_t)splits * (nW + a.Co) * sizeof(float) <= ((size_t)256 << 20);
    if (ordered) {
        float *base = nullptr;
        if (int rc = conv_scratch((size_t)splits * (nW + a.Co) + 16, &base, st)) return rc;
        a.wg_part = base;
        a.wg_part_b = a.gb ? base + (((size_t)splits * nW + 3) & ~(size_t)3) : nullptr;
    } else if (!accumulate) {
        LSN_HIP(hipMemsetAsync(a.gw, 0, sizeof(float) * nW, st));
        if (a.gb) LSN_HIP(hipMemsetAsync(a.gb, 0, sizeof(float) * (size_t)a.Co, st));
    }
    ProfScope prof(PROF_WGRAD, a, st);
    if (ordered && math_np() == 0) {   // exact fp32 (fp32 MFMA), partial gradients per split: deterministic
        if (vec_ok(a))
            hipLaunchKernelGGL(dcn_wgrad_kernel<true>, dim3(ncol, splits, nz), dim3(256), lds, st, a, nsteps);
        else
            hipLaunchKernelGGL(dcn_wgrad_kernel<false>, dim3(ncol, splits, nz), dim3(256), lds, st, a, nsteps);
        LSN_HIP(hipGetLastError());
        return conv_wgrad_reduce(a.wg_part, a.gw, nW, a.wg_part_b, a.gb, a.Co, splits, splits, accumulate ? 1 : 0, st);
    }
    if (ordered) {
        const size_t ldsn = math_np() == 6 ? wgrad_xn_lds_bytes<6>() : wgrad_xn_lds_bytes<3>();
        auto kern = math_np() == 6 ? dcn_wgrad_xn_kernel<false, 6> : dcn_wgrad_xn_kernel<false, 3>;
        if (int rc = set_lds(kern, ldsn)) return rc;
        hipLaunchKernelGGL(kern, dim3(ncol, splits, nz), dim3(256), ldsn, st, a, nsteps);
        LSN_HIP(hipGetLastError());
        return conv_wgrad_reduce(a.wg_part, a.gw, nW, a.wg_part_b, a.gb, a.Co, splits, splits, accumulate ? 1 : 0, st);
    }
    if (math_np() && !((g_dbg_block >> 30) & 1)) {   // bit 30: force the fp32 MFMA kernel
        if (math_np() == 6) {
            const size_t ldsn = wgrad_xn_lds_bytes<6>();
            if (int rc = set_lds(dcn_wgrad_xn_kernel<false, 6>, ldsn)) return rc;
            hipLaunchKernelGGL((dcn_wgrad_xn_kernel<false, 6>), dim3(ncol, splits, nz), dim3(256), ldsn, st, a, nsteps);
        } else {
            const size_t ldsn = wgrad_xn_lds_bytes<3>();
            if (int rc = set_lds(dcn_wgrad_xn_kernel<false, 3>, ldsn)) return rc;
            hipLaunchKernelGGL((dcn_wgrad_xn_kernel<false, 3>), dim3(ncol, splits, nz), dim3(256), ldsn, st, a, nsteps);
        }
        LSN_HIP(hipGetLastError());
        return 0;
    }
    if (vec_ok(a))
        hipLaunchKernelGGL(dcn_wgrad_kernel<true>, dim3(ncol, splits, nz), dim3(256), lds, st, a, nsteps);
    else
        hipLaunchKernelGGL(dcn_wgrad_kernel<false>, dim3(ncol, splits, nz), dim3(256), lds, st, a, nsteps);
    LSN_HIP(hipGetLastError());
    return 0;
}

static size_t n_in(const lsn_dcn_shape &s, const lsn_dcn_level &d) { return (size_t)d.B * s.C * d.H * d.W; }
static size_t n_out(const lsn_dcn_shape &s, const lsn_dcn_level &d) { return (size_t)d.B * s.Co * d.Ho * d.Wo; }

static int dcn_forward_impl(const lsn_dcn_shape &s, int n, const lsn_dcn_level *lv, const float *weight,
                            const float *bias, lsn_layout layout, hipStream_t st)
{
    if (int rc = check_shape(s)) return rc;
    LSN_CHECK(weight != nullptr, "weight is NULL");
    DcnArgs a;
    if (int rc = fill_levels(a, s, n, lv, 64)) return rc;
    Scratch ws(st);
    const int K = s.kh * s.kw, Cg = s.C / s.groups;
    std::vector<float *> out_tmp(n, nullptr);
    if (layout == LSN_NHWC) {
        a.w = weight;
        for (int i = 0; i < n; ++i) {
            LSN_CHECK(lv[i].input && lv[i].output, "level %d: input/output is NULL", i);
            a.lv[i].x = lv[i].input;
            a.lv[i].out = lv[i].output;
        }
    } else {
        float *w2 = ws.get((size_t)s.Co * Cg * K);
        if (!w2) return fail(LSN_ERR_RUNTIME, "workspace allocation failed");
        if (int rc = permute_rs(weight, w2, s.Co, Cg, K, st)) return rc;
        a.w = w2;
        for (int i = 0; i < n; ++i) {
            LSN_CHECK(lv[i].input && lv[i].output, "level %d: input/output is NULL", i);
            float *x2 = ws.get(n_in(s, lv[i]));
            out_tmp[i] = ws.get(n_out(s, lv[i]));
            if (!x2 || !out_tmp[i]) return fail(LSN_ERR_RUNTIME, "workspace allocation failed");
            if (int rc = permute_rs(lv[i].input, x2, lv[i].B, s.C, lv[i].H * lv[i].W, st)) return rc;
            a.lv[i].x = x2;
            a.lv[i].out = out_tmp[i];
        }
    }
    a.bias = bias;
    a.wtp = nullptr;
    if (a.opitch != a.Co && !(layout == LSN_NHWC && s.workspace && mm_fwd_ok(a)))
        return fail(LSN_ERR_UNSUPPORTED, "deformable forward: out_pitch %d != Co %d outside the matrix-pipe kernels "
                    "(lsn_dcn_pitched_ok)", a.opitch, a.Co);
    if (s.weights_prepared && !(layout == LSN_NHWC && s.workspace && mm_fwd_ok(a)))
        return fail(LSN_ERR_UNSUPPORTED, "deformable forward: weights_prepared outside the matrix-pipe kernels (lsn_dcn_prepared_ok)");
    if (s.workspace && mm_fwd_ok(a)) {
        if (int rc = mm_prepare_weights(a, false, s.workspace, st, s.weights_prepared != 0)) return rc;
    } else if (s.workspace && math_np() && (Cg % 8 == 0) && a.Co / a.groups > 64 && xn_ok(a)) {
        const size_t nw = (size_t)s.Co * K * Cg;   // split the weights once instead of in every block
        if (math_np() == 6)
            hipLaunchKernelGGL(dcn_prepare_w_kernel<3>, dim3(512), dim3(256), 0, st, a.w,
                               reinterpret_cast<unsigned short *>(s.workspace), nw);
        else
            hipLaunchKernelGGL(dcn_prepare_w_kernel<2>, dim3(512), dim3(256), 0, st, a.w,
                               reinterpret_cast<unsigned short *>(s.workspace), nw);
        a.wtp = reinterpret_cast<const unsigned short *>(s.workspace);
    }
    if (int rc = launch_forward(a, st)) return rc;
    if (layout == LSN_NCHW)
        for (int i = 0; i < n; ++i)
            if (int rc = permute_rs(out_tmp[i], lv[i].output, lv[i].B, lv[i].Ho * lv[i].Wo, s.Co, st)) return rc;
    return 0;
}

static int dcn_backward_impl(const lsn_dcn_shape &s, int n, const lsn_dcn_level *lv, const float *weight,
                             float *grad_weight, float *grad_bias, lsn_layout layout, hipStream_t st)
{
    if (int rc = check_shape(s)) return rc;
    LSN_CHECK(weight != nullptr, "weight is NULL");
    if (grad_bias && !grad_weight)
        return fail(LSN_ERR_UNSUPPORTED, "grad_bias without grad_weight is not supported");
    DcnArgs a;
    if (int rc = fill_levels(a, s, n, lv, BWD_BM)) return rc;
    Scratch ws(st);
    const int K = s.kh * s.kw, Cg = s.C / s.groups;
    std::vector<float *> gx_tmp(n, nullptr);
    float *gw_tmp = nullptr;
    bool any_data = false;
    for (int i = 0; i < n; ++i) {
        LSN_CHECK(lv[i].input && lv[i].grad_output, "level %d: input/grad_output is NULL", i);
        any_data = any_data || lv[i].grad_input || lv[i].grad_offset || lv[i].grad_mask;
    }
    if (layout == LSN_NHWC) {
        a.w = weight;
        a.gw = grad_weight;
        for (int i = 0; i < n; ++i) {
            a.lv[i].x = lv[i].input;
            a.lv[i].gout = lv[i].grad_output;
            a.lv[i].gx = lv[i].grad_input;
        }
    } else {
        float *w2 = ws.get((size_t)s.Co * Cg * K);
        if (!w2) return fail(LSN_ERR_RUNTIME, "workspace allocation failed");
        if (int rc = permute_rs(weight, w2, s.Co, Cg, K, st)) return rc;
        a.w = w2;
        if (grad_weight) {
            gw_tmp = ws.get((size_t)s.Co * Cg * K);
            if (!gw_tmp) return fail(LSN_ERR_RUNTIME, "workspace allocation failed");
            a.gw = gw_tmp;
        }
        for (int i = 0; i < n; ++i) {
            float *x2 = ws.get(n_in(s, lv[i])), *g2 = ws.get(n_out(s, lv[i]));
            if (!x2 || !g2) return fail(LSN_ERR_RUNTIME, "workspace allocation failed");
            if (int rc = permute_rs(lv[i].input, x2, lv[i].B, s.C, lv[i].H * lv[i].W, st)) return rc;
            if (int rc = permute_rs(lv[i].grad_output, g2, lv[i].B, s.Co, lv[i].Ho * lv[i].Wo, st)) return rc;
            a.lv[i].x = x2;
            a.lv[i].gout = g2;
            if (lv[i].grad_input) {
                gx_tmp[i] = ws.get(n_in(s, lv[i]));
                if (!gx_tmp[i]) return fail(LSN_ERR_RUNTIME, "workspace allocation failed");
                a.lv[i].gx = gx_tmp[i];
            }
        }
    }
    a.gb = grad_bias;

    if (any_data) {
        a.wtp = nullptr;
        a.gcol = nullptr;
        const bool can_split = s.workspace && math_np() && s.groups == 1 && s.Co % 2 == 0;
        if (can_split) a.wtp = reinterpret_cast<const unsigned short *>(s.workspace);   // (format: decided below)
        void *gws = s.gather_workspace;
        size_t gws_bytes = (size_t)(s.gather_workspace_bytes > 0 ? s.gather_workspace_bytes : 0);
        if (!gws && layout == LSN_NCHW && (a.wtp || bwd_grouped_ok(a)) && bwd_colbuf_env()) {   // reference-layout entry points: own scratch
            GatherPlan pl;
            DcnArgs probe = a;
            gather_plan(probe, pl);
            if (pl.ok && (gws = ws.get(pl.bytes / 4 + 64)) != nullptr) gws_bytes = pl.bytes;
        }
        bool mm = false;   // the GEMM of dcn_mm_kernels.h: only together with the gather pass
        if (can_split && gws && bwd_colbuf_env() && mm_bwd_ok(a) && bwd_gather_ok(a)) {
            GatherPlan pl;
            DcnArgs probe = a;
            gather_plan(probe, pl);
            mm = pl.ok && pl.bytes <= gws_bytes;
        }
        if (a.opitch != a.Co && !(mm && layout == LSN_NHWC))
            return fail(LSN_ERR_UNSUPPORTED, "deformable backward: out_pitch %d != Co %d outside the matrix-pipe kernels "
                        "(lsn_dcn_pitched_ok)", a.opitch, a.Co);
        if (s.weights_prepared && !(mm && layout == LSN_NHWC))
            return fail(LSN_ERR_UNSUPPORTED, "deformable backward: weights_prepared outside the matrix-pipe kernels (lsn_dcn_prepared_ok)");
        if (mm) {
            if (int rc = mm_prepare_weights(a, true, s.workspace, st, s.weights_prepared != 0)) return rc;
        } else if (can_split) {
            if (math_np() == 6)
                hipLaunchKernelGGL(dcn_prepare_wt_kernel<3>, dim3(512), dim3(256), 0, st, a.w,
                                   reinterpret_cast<unsigned short *>(s.workspace), s.Co, K, s.C);
            else
                hipLaunchKernelGGL(dcn_prepare_wt_kernel<2>, dim3(512), dim3(256), 0, st, a.w,
                                   reinterpret_cast<unsigned short *>(s.workspace), s.Co, K, s.C);
        }
        if (int rc = launch_bwd_data(a, gws, gws_bytes, st)) return rc;
    } else if (a.opitch != a.Co) {
        return fail(LSN_ERR_UNSUPPORTED, "deformable backward: out_pitch without a data-gradient pass");
    }
    if (a.gw) {
        DcnArgs w = a;  // same levels, step (32-pixel) indexing
        int steps = 0;
        for (int i = 0; i < n; ++i) {
            w.lv[i].tile0 = steps;
            steps += cdiv(w.lv[i].P, WG_BP);
        }
        // (the reference-layout path computes into a temporary that is permuted into grad_weight: no accumulation there)
        if (int rc = launch_wgrad(w, steps, s.accumulate_param_grads != 0 && layout == LSN_NHWC, st)) return rc;
    }
    if (layout == LSN_NCHW) {
        for (int i = 0; i < n; ++i)
            if (gx_tmp[i])
                if (int rc = permute_rs(gx_tmp[i], lv[i].grad_input, lv[i].B, lv[i].H * lv[i].W, s.C, st)) return rc;
        if (gw_tmp)
            if (int rc = permute_rs(gw_tmp, grad_weight, s.Co, K, Cg, st)) return rc;
    }
    return 0;
}

static lsn_strides4 nchw_strides(int Ch, int H, int W)
{
    lsn_strides4 s;
    s.b = (int64_t)Ch * H * W;
    s.c = (int64_t)H * W;
    s.h = W;
    s.w = 1;
    return s;
}

static int conv_out(int in, int k, int stride, int pad, int dil) { return (in + 2 * pad - (dil * (k - 1) + 1)) / stride + 1; }

// shape + single NCHW level for the one-to-one wrappers
static int make_single(lsn_dcn_shape &s, lsn_dcn_level &L, int B, int C, int H, int W, int Co, int Ho, int Wo,
                       int kH, int kW, int dH, int dW, int padH, int padW, int dilH, int dilW, int group, int dg,
                       float scaleH, float scaleW, bool pyramid)
{
    if (dH != dW || padH != padW || dilH != dilW)
        return fail(LSN_ERR_UNSUPPORTED, "different stride/pad/dilation for h and w is not supported");
    memset(&s, 0, sizeof(s));
    memset(&L, 0, sizeof(L));
    LSN_CHECK(kH > 0 && kW > 0, "kernel size should be greater than zero, but got kH: %d kW: %d", kH, kW);
    LSN_CHECK(dH > 0, "stride should be greater than zero, but got dH: %d dW: %d", dH, dW);
    LSN_CHECK(dilH > 0, "dilation should be greater than 0, but got dilationH: %d dilationW: %d", dilH, dilW);
    if (!pyramid) {
        Ho = conv_out(H, kH, dH, padH, dilH);
        Wo = conv_out(W, kW, dW, padW, dilW);
    } else {
        // pyramid_shape_check: the output grid is derived from the OFFSET grid and must equal it
        LSN_CHECK(conv_out(Ho, kH, dH, padH, dilH) == Ho && conv_out(Wo, kW, dW, padW, dilW) == Wo,
                  "invalid spatial size of offset, expected height: %d width: %d, but got height: %d width: %d",
                  conv_out(Ho, kH, dH, padH, dilH), conv_out(Wo, kW, dW, padW, dilW), Ho, Wo);
    }
    LSN_CHECK(Ho >= 1 && Wo >= 1,
              "Given input size: (%d x %d x %d). Calculated output size: (%d x %d x %d). Output size is too small",
              C, H, W, Co, Ho, Wo);
    LSN_CHECK(H >= kH && W >= kW, "input image is smaller than kernel");
    s.B = B; s.C = C; s.H = H; s.W = W; s.Co = Co; s.Ho = Ho; s.Wo = Wo;
    s.kh = kH; s.kw = kW; s.stride = dH; s.pad = padH; s.dil = dilH;
    s.groups = group; s.deformable_groups = dg;
    s.scale_h = scaleH; s.scale_w = scaleW;
    L.B = B; L.H = H; L.W = W; L.Ho = Ho; L.Wo = Wo;
    L.scale_h = scaleH; L.scale_w = scaleW;
    L.off_st = nchw_strides(dg * 2 * kH * kW, Ho, Wo);
    L.mask_st = nchw_strides(dg * kH * kW, Ho, Wo);
    return 0;
}

// weight gradient of a dense convolution through the deformable-conv weight-gradient kernel (PLAIN: no offsets), summed
// over up to MAXLV input maps that share the weight
template <int NP, int BMW>
static int conv_wgrad_launch(DcnArgs a, int nsteps, int C, int Co, int K, bool accumulate, hipStream_t st)
{
    const int ncc = cdiv(C, WG_BN), ncol = K * ncc, nz = cdiv(Co, BMW);
    int splits = wgrad_splits(ncol * nz, BMW == 256 ? 2 : 3);   // resident blocks per CU: registers (246 / 142 VGPRs)
    if (splits > nsteps) splits = nsteps;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    // one partial gradient per pixel split + an ordered reduce (deterministic) where the sizes allow; else fp32 atomics
    const size_t nW = (size_t)Co * K * C;
    const bool ordered = nW % 4 == 0 && (size_t)splits * (nW + Co) * sizeof(float) <= ((size_t)256 << 20);
    if (!ordered && conv_wgrad_fold_pending())
        return fail(LSN_ERR_UNSUPPORTED, "conv2d backward-weight (folded norm): the gradient is too large for the ordered reduce");
    if (ordered) {
        float *base = nullptr;
        if (int rc = conv_scratch((size_t)splits * (nW + Co) + 16, &base, st)) return rc;
        a.wg_part = base;
        a.wg_part_b = a.gb ? base + (((size_t)splits * nW + 3) & ~(size_t)3) : nullptr;
    } else if (!accumulate) {
        LSN_HIP(hipMemsetAsync(a.gw, 0, sizeof(float) * nW, st));
        if (a.gb) LSN_HIP(hipMemsetAsync(a.gb, 0, sizeof(float) * (size_t)Co, st));
    }
    const size_t ldsn = wgrad_xn_lds_bytes<NP, BMW>();
    auto k = dcn_wgrad_xn_kernel<true, NP, BMW>;
    if (int rc = set_lds(k, ldsn)) return rc;
    hipLaunchKernelGGL(k, dim3(ncol, splits, nz), dim3(256), ldsn, st, a, nsteps);
    LSN_HIP(hipGetLastError());
    if (ordered) return conv_wgrad_reduce(a.wg_part, a.gw, nW, a.wg_part_b, a.gb, Co, splits, splits, accumulate ? 1 : 0, st);
    return 0;
}

int conv_wgrad_mm(int n, const lsn_conv_level *lv, float *gw, float *gb, int C, int Co, int kh, int kw, int stride, int pad,
                  int dil, int accumulate, hipStream_t st);   // conv.hip

// bits 26 / 27 of the debug word (tools/ubench/wgrad_ab: A/B runs): 0 never, 1 every shape the kernel serves; default (2):
// the shapes it won on (conv_wgrad_dense_mm)
static int conv_wgrad_mm_env() { return ((g_dbg_block >> 26) & 1) ? 0 : ((g_dbg_block >> 27) & 1) ? 1 : 2; }

// The weight gradient of a dense convolution through dcn_wgrad_mm_kernel<NP, DENSE> (grad_output pre-split once into MFMA
// fragment order, the regular grid as the sampling table).  Returns 1 when the shape is not served (256 | Co, 64 | C,
// 32-bit byte offsets, a split-bf16 math mode) or is faster on the patch kernel of conv_wgrad_kernels.h.  Measured on the
// layer shapes of the benchmark step (tools/ubench/wgrad_ab.hip, profiles/r3_wgrad_ab.txt; batch 2): 3x3 at >= 4096
// output pixels 84 vs 96 us (layer 3), 262 vs 272 (FPN P3), 320 vs 380 (five head levels in one launch); 1x1 from >= 1024
// channels to >= 512: 75 vs 89, 53 vs 56; strided 1x1 from >= 512 channels: 83 vs 89, 78 vs 89.  Slower on the 1x1 layers
// with few input channels and many pixels (76 vs 52 us on 128 -> 512 at 100 x 168), which stay where they were.
static int conv_wgrad_dense_mm(int n, const lsn_conv_level *lv, float *gw, float *gb, int C, int Co, int kh, int kw, int stride,
                               int pad, int dil, bool accumulate, hipStream_t st)
{
    if (!conv_wgrad_mm_env() || n < 1 || n > MAXLV || !lv || !gw || Co % 256 != 0 || C % 64 != 0) return 1;
    if (conv_wgrad_mm_env() != 1) {
        int64_t px = 0;
        for (int i = 0; i < n; ++i) {
            const int Ho = (lv[i].H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1, Wo = (lv[i].W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
            px += (int64_t)lv[i].B * (Ho > 0 ? Ho : 0) * (Wo > 0 ? Wo : 0);
        }
        // (all of them measured at >= 2100 output pixels; smaller launches stay where they were)
        const bool win = kh * kw >= 9 ? px >= 4096 : px >= 2048 && ((C >= 1024 && Co >= 512) || (stride >= 2 && C >= 512));
        if (!win) return 1;
    }
    DcnArgs a = {};
    int chunks = 0;
    for (int i = 0; i < n; ++i) {
        Lvl &L = a.lv[i];
        const int B = lv[i].B, H = lv[i].H, W = lv[i].W;
        if (!lv[i].x || !lv[i].grad_out || B <= 0 || H <= 0 || W <= 0) return 1;
        if (((reinterpret_cast<uintptr_t>(lv[i].x) | reinterpret_cast<uintptr_t>(lv[i].grad_out)) & 15) != 0) return 1;   // 16-byte loads
        const int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1, Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
        if (Ho <= 0 || Wo <= 0) return 1;
        if ((int64_t)B * H * W * C * 4 >= ((int64_t)1 << 31) || (int64_t)B * Ho * Wo * Co * 4 >= ((int64_t)1 << 31)) return 1;
        L.x = lv[i].x, L.gout = lv[i].grad_out, L.off = nullptr, L.msk = nullptr;
        L.B = B, L.H = H, L.W = W, L.Ho = Ho, L.Wo = Wo, L.P = B * Ho * Wo, L.sh = L.sw = 1.f;
        L.tile0 = chunks;
        chunks += cdiv(L.P, WG_BP);
    }
    a.nlv = n;
    a.C = C, a.Co = Co, a.kh = kh, a.kw = kw, a.stride = stride, a.pad = pad, a.dil = dil, a.groups = 1, a.dg = 1;
    a.SL = C;
    a.opitch = Co;
    a.gw = gw, a.gb = gb;
    if (!mm_common_ok(a) || chunks < 8) return 1;
    double px = 0, in_el = 0;
    for (int i = 0; i < n; ++i) px += (double)a.lv[i].P, in_el += (double)a.lv[i].B * a.lv[i].H * a.lv[i].W * C;
    ProfSpan prof(PROF_CONV_WGRAD, 2.0 * px * Co * C * kh * kw, 4.0 * (in_el + px * Co + (double)Co * kh * kw * C), st);
    return launch_wgrad_mm(a, chunks, accumulate, st, true);
}

static int conv_wgrad_xn(int n, const lsn_conv_level *lv, float *gw, float *gb, int C, int Co, int kh, int kw, int stride,
                         int pad, int dil, bool accumulate, hipStream_t st)
{
    {   // wide layers: the fragment-order kernel of the deformable family with the regular grid as its sampling table
        const int rc = conv_wgrad_dense_mm(n, lv, gw, gb, C, Co, kh, kw, stride, pad, dil, accumulate, st);
        if (rc != 1) return rc;
    }
    {   // the patch kernel of conv_wgrad_kernels.h serves up to nine taps; anything else stays here
        const int rc = conv_wgrad_mm(n, lv, gw, gb, C, Co, kh, kw, stride, pad, dil, accumulate ? 1 : 0, st);
        if (rc != 1) return rc;
    }
    LSN_CHECK(n >= 1 && n <= MAXLV && lv && gw, "conv2d backward-weight: bad arguments");
    DcnArgs a = {};
    int steps = 0;
    for (int i = 0; i < n; ++i) {
        Lvl &L = a.lv[i];
        const int B = lv[i].B, H = lv[i].H, W = lv[i].W;
        LSN_CHECK(lv[i].x && lv[i].grad_out && B > 0 && H > 0 && W > 0, "conv2d backward-weight: bad level %d", i);
        const int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1, Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
        LSN_CHECK(Ho > 0 && Wo > 0, "conv2d backward-weight: output size is too small");
        if ((int64_t)B * H * W * C >= ((int64_t)1 << 31) || (int64_t)B * Ho * Wo * Co >= ((int64_t)1 << 31))
            return fail(LSN_ERR_UNSUPPORTED, "conv2d backward-weight: tensor too large for 32-bit indexing");
        L.x = lv[i].x, L.gout = lv[i].grad_out, L.off = nullptr, L.msk = nullptr;
        L.B = B, L.H = H, L.W = W, L.Ho = Ho, L.Wo = Wo, L.P = B * Ho * Wo, L.sh = L.sw = 1.f;
        L.tile0 = steps;
        steps += cdiv(L.P, WG_BP);
    }
    a.nlv = n;
    a.C = C, a.Co = Co, a.kh = kh, a.kw = kw, a.stride = stride, a.pad = pad, a.dil = dil, a.groups = 1, a.dg = 1;
    a.SL = C;
    a.opitch = Co;
    a.gw = gw, a.gb = gb;
    a.wg_vec = wgrad_vec_bits(a);
    const int K = kh * kw;
    double px = 0, in_el = 0;
    for (int i = 0; i < n; ++i) px += (double)a.lv[i].P, in_el += (double)a.lv[i].B * a.lv[i].H * a.lv[i].W * C;
    ProfSpan prof(PROF_CONV_WGRAD, 2.0 * px * Co * C * K, 4.0 * (in_el + px * Co + (double)Co * K * C), st);
    const bool x3 = math_np() == 3;   // exact-mode callers get the fp32-equivalent split: there is no fp32-MFMA dense wgrad
    if (Co <= 64)
        return x3 ? conv_wgrad_launch<3, 64>(a, steps, C, Co, K, accumulate, st) : conv_wgrad_launch<6, 64>(a, steps, C, Co, K, accumulate, st);
    return x3 ? conv_wgrad_launch<3, 256>(a, steps, C, Co, K, accumulate, st)
              : conv_wgrad_launch<6, 256>(a, steps, C, Co, K, accumulate, st);
}

}  // namespace lsn

using namespace lsn;

extern "C" {

const char *lsn_last_error(void) { return lsn::err_buf(); }

int lsn_debug_phase_clocks(long long *device_buf_512, int block)
{
    lsn::g_dbg_buf = device_buf_512;
    lsn::g_dbg_block = block;
    return 0;
}
int lsn_version(void) { return 100; }

int64_t lsn_dcn_backward_workspace_bytes(const lsn_dcn_shape *shape, int n_levels, const lsn_dcn_level *levels)
{
    using namespace lsn;
    if (!shape || !levels || !bwd_colbuf_env()) return 0;
    if (check_shape(*shape) != 0) return 0;
    DcnArgs a;
    if (fill_levels(a, *shape, n_levels, levels, BWD_BM) != 0) return 0;
    for (int i = 0; i < n_levels; ++i) a.lv[i].gx = levels[i].grad_input, a.lv[i].goff = levels[i].grad_offset, a.lv[i].gmsk = levels[i].grad_mask;
    a.wtp = reinterpret_cast<const unsigned short *>(shape);   // any non-NULL value: the caller passes `workspace` too
    if (!bwd_gather_ok(a)) return 0;
    GatherPlan pl;
    gather_plan(a, pl);
    return pl.ok ? (int64_t)pl.bytes : 0;
}

int lsn_dcn_prepared_ok(const lsn_dcn_shape *shape, int n_levels, const lsn_dcn_level *levels, int backward)
{
    using namespace lsn;
    if (!shape || !levels || !shape->workspace || check_shape(*shape) != 0) return 0;
    DcnArgs a;
    if (fill_levels(a, *shape, n_levels, levels, backward ? BWD_BM : 64) != 0) return 0;
    if (!backward) return mm_fwd_ok(a) ? 1 : 0;
    bool any_data = false;
    for (int i = 0; i < n_levels; ++i) {
        a.lv[i].gx = levels[i].grad_input, a.lv[i].goff = levels[i].grad_offset, a.lv[i].gmsk = levels[i].grad_mask;
        any_data = any_data || levels[i].grad_input || levels[i].grad_offset || levels[i].grad_mask;
    }
    a.wtp = reinterpret_cast<const unsigned short *>(shape->workspace);
    if (!any_data || !shape->gather_workspace || !bwd_colbuf_env() || math_np() == 0 || shape->groups != 1 || a.Co % 2 != 0) return 0;
    if (!mm_bwd_ok(a) || !bwd_gather_ok(a)) return 0;
    GatherPlan pl;
    gather_plan(a, pl);
    return (pl.ok && (int64_t)pl.bytes <= shape->gather_workspace_bytes) ? 1 : 0;
}

int lsn_dcn_pitched_ok(const lsn_dcn_shape *shape, int n_levels, const lsn_dcn_level *levels, int backward)
{
    using namespace lsn;
    if (!shape || !levels || !shape->workspace || check_shape(*shape) != 0) return 0;
    DcnArgs a;
    if (fill_levels(a, *shape, n_levels, levels, backward ? BWD_BM : 64) != 0) return 0;
    if (!backward) return mm_fwd_ok(a) ? 1 : 0;
    bool any_data = false;
    for (int i = 0; i < n_levels; ++i) {
        a.lv[i].gx = levels[i].grad_input, a.lv[i].goff = levels[i].grad_offset, a.lv[i].gmsk = levels[i].grad_mask;
        any_data = any_data || levels[i].grad_input || levels[i].grad_offset || levels[i].grad_mask;
    }
    a.wtp = reinterpret_cast<const unsigned short *>(shape->workspace);
    if (!any_data || !shape->gather_workspace || !bwd_colbuf_env() || a.Co % 2 != 0) return 0;
    if (!mm_bwd_ok(a) || !bwd_gather_ok(a)) return 0;
    GatherPlan pl;
    gather_plan(a, pl);
    if (!pl.ok || (int64_t)pl.bytes > shape->gather_workspace_bytes) return 0;
    // the weight gradient on the fragment-order kernel (launch_wgrad_mm)
    if (a.Co % 256 != 0 || (a.C / a.dg) % 64 != 0) return 0;
    int64_t steps = 0;
    for (int i = 0; i < n_levels; ++i) steps += cdiv(a.lv[i].P, WG_BP);
    return steps * 2 * (a.Co / 32) * mm_npl() * 1024 < ((int64_t)1 << 31) ? 1 : 0;
}

int lsn_set_math_mode(int mode)
{
    LSN_CHECK(mode == LSN_MATH_FP32 || mode == LSN_MATH_BF16X3 || mode == LSN_MATH_BF16X6, "unknown math mode %d", mode);
    lsn::g_math_mode = mode;
    return 0;
}

int lsn_get_math_mode(void) { return lsn::math_mode(); }

int lsn_conv2d_backward_weight_multi(int n_levels, const lsn_conv_level *levels, float *grad_w, float *grad_bias, int C,
                                     int Co, int kh, int kw, int stride, int pad, int dil, int accumulate,
                                     lsn_stream_t stream)
{
    LSN_CHECK(C > 0 && Co > 0 && kh > 0 && kw > 0 && stride > 0 && dil > 0 && pad >= 0, "conv2d backward-weight: bad shape");
    return conv_wgrad_xn(n_levels, levels, grad_w, grad_bias, C, Co, kh, kw, stride, pad, dil, accumulate != 0,
                         reinterpret_cast<hipStream_t>(stream));
}

int lsn_conv2d_backward_weight(const float *x, const float *grad_out, float *grad_w, float *grad_bias, int B, int H,
                               int W, int C, int Co, int kh, int kw, int stride, int pad, int dil, int accumulate,
                               lsn_stream_t stream)
{
    lsn_conv_level L = {};
    L.x = x, L.grad_out = grad_out, L.B = B, L.H = H, L.W = W;
    return lsn_conv2d_backward_weight_multi(1, &L, grad_w, grad_bias, C, Co, kh, kw, stride, pad, dil, accumulate, stream);
}

int lsn_prof_enable(int on)
{
    for (auto &r : lsn::g_prof) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    lsn::g_prof.clear();
    lsn::g_prof_mask = on == 1 ? ~0u : (unsigned)on;
    return 0;
}

int lsn_prof_launch_log(int on)
{
    for (auto &r : lsn::g_launches) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    lsn::g_launches.clear();
    lsn::g_launch_log = on != 0;
    return 0;
}

int lsn_prof_read_launches(lsn_prof_launch *out, int max_entries)
{
    using namespace lsn;
    const int n = (int)g_launches.size();
    if (out == nullptr) return n;
    int k = 0;
    for (; k < n && k < max_entries; ++k) {
        const LaunchRec &r = g_launches[k];
        LSN_HIP(hipEventSynchronize(r.e1));
        float ms = 0.f;
        LSN_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
        lsn_prof_launch &o = out[k];
        o.kind = r.v[0], o.C = r.v[1], o.Co = r.v[2], o.kh = r.v[3], o.kw = r.v[4], o.stride = r.v[5], o.pad = r.v[6];
        o.dil = r.v[7], o.relu = r.v[8], o.xpitch = r.v[9], o.n_levels = r.v[10], o.has_residual = r.v[11], o.has_gate = r.v[12];
        for (int i = 0; i < 16; ++i) o.B[i] = r.B[i], o.H[i] = r.H[i], o.W[i] = r.W[i];
        o.ms = ms;
    }
    return k;
}

int lsn_prof_read(lsn_prof_entry *out, int max_entries)
{
    using namespace lsn;
    LSN_CHECK(out != nullptr && max_entries >= PROF_N, "lsn_prof_read needs room for %d entries", PROF_N);
    for (int f = 0; f < PROF_N; ++f) {
        std::memset(&out[f], 0, sizeof(out[f]));
        std::snprintf(out[f].name, sizeof(out[f].name), "%s", kProfNames[f]);
    }
    for (auto &r : g_prof) {
        LSN_HIP(hipEventSynchronize(r.e1));
        float ms = 0.f;
        LSN_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
        lsn_prof_entry &e = out[r.fam];
        e.launches += 1;
        e.total_ms += ms;
        e.flops += r.flops;
        e.bytes += r.bytes;
    }
    return PROF_N;
}

int lsn_dcn_forward(const lsn_dcn_shape *shape, int n_levels, const lsn_dcn_level *levels, const float *weight,
                    const float *bias, lsn_layout layout, lsn_stream_t stream)
{
    if (!shape || !levels) return fail(LSN_ERR_INVALID, "shape/levels is NULL");
    return dcn_forward_impl(*shape, n_levels, levels, weight, bias, layout, stream);
}

int lsn_dcn_backward(const lsn_dcn_shape *shape, int n_levels, const lsn_dcn_level *levels, const float *weight,
                     float *grad_weight, float *grad_bias, lsn_layout layout, lsn_stream_t stream)
{
    if (!shape || !levels) return fail(LSN_ERR_INVALID, "shape/levels is NULL");
    return dcn_backward_impl(*shape, n_levels, levels, weight, grad_weight, grad_bias, layout, stream);
}

int lsn_deform_conv_forward(const float *input, const float *weight, const float *offset, float *output, int B,
                            int C, int H, int W, int Co, int kW, int kH, int dW, int dH, int padW, int padH,
                            int dilW, int dilH, int group, int deformable_group, int im2col_step,
                            lsn_stream_t stream)
{
    (void)im2col_step;
    lsn_dcn_shape s;
    lsn_dcn_level L;
    if (int rc = make_single(s, L, B, C, H, W, Co, 0, 0, kH, kW, dH, dW, padH, padW, dilH, dilW, group,
                             deformable_group, 1.f, 1.f, false))
        return rc;
    L.input = input; L.offset = offset; L.output = output;
    return dcn_forward_impl(s, 1, &L, weight, nullptr, LSN_NCHW, stream);
}

int lsn_deform_conv_backward_input(const float *input, const float *offset, const float *grad_output,
                                   float *grad_input, float *grad_offset, const float *weight, int B, int C,
                                   int H, int W, int Co, int kW, int kH, int dW, int dH, int padW, int padH,
                                   int dilW, int dilH, int group, int deformable_group, int im2col_step,
                                   lsn_stream_t stream)
{
    (void)im2col_step;
    lsn_dcn_shape s;
    lsn_dcn_level L;
    if (int rc = make_single(s, L, B, C, H, W, Co, 0, 0, kH, kW, dH, dW, padH, padW, dilH, dilW, group,
                             deformable_group, 1.f, 1.f, false))
        return rc;
    L.input = input; L.offset = offset; L.grad_output = grad_output;
    L.grad_input = grad_input; L.grad_offset = grad_offset;
    return dcn_backward_impl(s, 1, &L, weight, nullptr, nullptr, LSN_NCHW, stream);
}

static int scale_weight_grad(float *gw, size_t n, float scale, hipStream_t st)
{
    if (scale == 1.f) return 0;
    hipLaunchKernelGGL(scale_kernel, dim3(256), dim3(256), 0, st, gw, n, scale);
    LSN_HIP(hipGetLastError());
    return 0;
}

int lsn_deform_conv_backward_parameters(const float *input, const float *offset, const float *grad_output,
                                        float *grad_weight, int B, int C, int H, int W, int Co, int kW, int kH,
                                        int dW, int dH, int padW, int padH, int dilW, int dilH, int group,
                                        int deformable_group, float scale, int im2col_step, lsn_stream_t stream)
{
    (void)im2col_step;
    lsn_dcn_shape s;
    lsn_dcn_level L;
    if (int rc = make_single(s, L, B, C, H, W, Co, 0, 0, kH, kW, dH, dW, padH, padW, dilH, dilW, group,
                             deformable_group, 1.f, 1.f, false))
        return rc;
    L.input = input; L.offset = offset; L.grad_output = grad_output;
    // the weight values are not needed for dL/dW; pass grad_weight as a placeholder pointer
    if (int rc = dcn_backward_impl(s, 1, &L, grad_weight, grad_weight, nullptr, LSN_NCHW, stream)) return rc;
    return scale_weight_grad(grad_weight, (size_t)Co * (C / group) * kH * kW, scale, stream);
}

int lsn_modulated_deform_conv_forward(const float *input, const float *weight, const float *bias,
                                      const float *offset, const float *mask, float *output, int B, int C, int H,
                                      int W, int Co, int kernel_h, int kernel_w, int stride_h, int stride_w,
                                      int pad_h, int pad_w, int dilation_h, int dilation_w, int group,
                                      int deformable_group, int with_bias, lsn_stream_t stream)
{
    lsn_dcn_shape s;
    lsn_dcn_level L;
    if (int rc = make_single(s, L, B, C, H, W, Co, 0, 0, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                             dilation_h, dilation_w, group, deformable_group, 1.f, 1.f, false))
        return rc;
    LSN_CHECK(mask != nullptr, "mask is NULL");
    L.input = input; L.offset = offset; L.mask = mask; L.output = output;
    return dcn_forward_impl(s, 1, &L, weight, with_bias ? bias : nullptr, LSN_NCHW, stream);
}

int lsn_modulated_deform_conv_backward(const float *input, const float *weight, const float *bias,
                                       const float *offset, const float *mask, float *grad_input,
                                       float *grad_weight, float *grad_bias, float *grad_offset, float *grad_mask,
                                       const float *grad_output, int B, int C, int H, int W, int Co, int kernel_h,
                                       int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w,
                                       int dilation_h, int dilation_w, int group, int deformable_group,
                                       int with_bias, lsn_stream_t stream)
{
    (void)bias;
    lsn_dcn_shape s;
    lsn_dcn_level L;
    if (int rc = make_single(s, L, B, C, H, W, Co, 0, 0, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                             dilation_h, dilation_w, group, deformable_group, 1.f, 1.f, false))
        return rc;
    LSN_CHECK(mask != nullptr, "mask is NULL");
    L.input = input; L.offset = offset; L.mask = mask; L.grad_output = grad_output;
    L.grad_input = grad_input; L.grad_offset = grad_offset; L.grad_mask = grad_mask;
    return dcn_backward_impl(s, 1, &L, weight, grad_weight, with_bias ? grad_bias : nullptr, LSN_NCHW, stream);
}

int lsn_pyramid_deform_conv_forward(const float *input, const float *weight, const float *offset, float *output,
                                    int B, int C, int H, int W, int Co, int Ho, int Wo, int kW, int kH, int dW,
                                    int dH, int padW, int padH, int dilW, int dilH, float scaleW, float scaleH,
                                    int group, int deformable_group, int im2col_step, lsn_stream_t stream)
{
    (void)im2col_step;
    lsn_dcn_shape s;
    lsn_dcn_level L;
    if (int rc = make_single(s, L, B, C, H, W, Co, Ho, Wo, kH, kW, dH, dW, padH, padW, dilH, dilW, group,
                             deformable_group, scaleH, scaleW, true))
        return rc;
    L.input = input; L.offset = offset; L.output = output;
    return dcn_forward_impl(s, 1, &L, weight, nullptr, LSN_NCHW, stream);
}

int lsn_pyramid_deform_conv_backward_input(const float *input, const float *offset, const float *grad_output,
                                           float *grad_input, float *grad_offset, const float *weight, int B,
                                           int C, int H, int W, int Co, int Ho, int Wo, int kW, int kH, int dW,
                                           int dH, int padW, int padH, int dilW, int dilH, float scaleW,
                                           float scaleH, int group, int deformable_group, int im2col_step,
                                           lsn_stream_t stream)
{
    (void)im2col_step;
    lsn_dcn_shape s;
    lsn_dcn_level L;
    if (int rc = make_single(s, L, B, C, H, W, Co, Ho, Wo, kH, kW, dH, dW, padH, padW, dilH, dilW, group,
                             deformable_group, scaleH, scaleW, true))
        return rc;
    L.input = input; L.offset = offset; L.grad_output = grad_output;
    L.grad_input = grad_input; L.grad_offset = grad_offset;
    return dcn_backward_impl(s, 1, &L, weight, nullptr, nullptr, LSN_NCHW, stream);
}

int lsn_pyramid_deform_conv_backward_parameters(const float *input, const float *offset,
                                                const float *grad_output, float *grad_weight, int B, int C,
                                                int H, int W, int Co, int Ho, int Wo, int kW, int kH, int dW,
                                                int dH, int padW, int padH, int dilW, int dilH, float scaleW,
                                                float scaleH, int group, int deformable_group, float scale,
                                                int im2col_step, lsn_stream_t stream)
{
    (void)im2col_step;
    lsn_dcn_shape s;
    lsn_dcn_level L;
    if (int rc = make_single(s, L, B, C, H, W, Co, Ho, Wo, kH, kW, dH, dW, padH, padW, dilH, dilW, group,
                             deformable_group, scaleH, scaleW, true))
        return rc;
    L.input = input; L.offset = offset; L.grad_output = grad_output;
    if (int rc = dcn_backward_impl(s, 1, &L, grad_weight, grad_weight, nullptr, LSN_NCHW, stream)) return rc;
    return scale_weight_grad(grad_weight, (size_t)Co * (C / group) * kH * kW, scale, stream);
}

}  // extern "C"
