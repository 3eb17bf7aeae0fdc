// Dense 2-D convolution (groups = 1) on channels-last fp32 tensors as an implicit GEMM on the bf16 matrix pipe with
// split operands (common.h: NP = 6 products of exact 3-way bf16 splits = fp32-equivalent, the default; NP = 3 products
// of 2-way splits), fp32 accumulation.  Kernels: conv_kernels.h.
//
// The reference runs torch.nn.Conv2d = cuDNN for every dense conv of the path (backbone resnet.py:624-631,
// 261-301; neck fpn.py:171-217; head lsnet_head.py:160-257).
//
//   forward      : out[p][co] = sum_{tap, ci} x[p @ tap][ci] * w[co][tap][ci] (+ bias, ReLU); up to 8 maps per launch
//   backward-data: the same kernel on grad_output with the weights transposed and flipped; a stride-s convolution is
//                  s x s stride-1 convolutions over tap subsets (TapSub), each writing its residue class of input pixels
//   backward-weight: dcn_wgrad_xn_kernel<PLAIN = true> (dcn_kernels.h), reduction over pixels with px-contiguous LDS images
//
// Weights reach the kernel in MFMA fragment order (conv_wfrag_kernel), written by lsn_conv2d_prepare_weights -- once per
// optimizer step when the caller keeps the image (the Python mirror caches it per parameter version), or inside the
// one-shot entry points.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/lsnet_hip.h"
#include "common.h"
#include "conv_kernels.h"
#include "conv_wgrad_kernels.h"
#include "prof.h"

namespace lsn {

int split_np();   // dcn.hip: bf16 products per fp32 product of the current math mode (0: exact fp32)
static int conv_np() { return split_np() == 3 ? 3 : 6; }   // these kernels have no fp32-MFMA variant: exact mode gets x6
static int conv_npl() { return conv_np() == 3 ? 2 : 3; }

// Library-owned scratch for the partial tiles of split reductions: ONE block PER STREAM (callers on different streams must
// not share partial tiles), grown on demand.  A block that is outgrown is
// RETIRED, not freed: a captured hipGraph may hold its address, and no device synchronisation is needed.
// Growing while the stream is being captured is refused -- run the step eagerly once first (the graph wrapper's warm-up
// calls do).  Host-side bookkeeping is not thread-safe (one Python thread drives the library).
static long long g_stat[4] = {0, 0, 0, 0};
void lib_stat(int which, long long add) { g_stat[which & 3] += add; }

struct Scratch {
    hipStream_t st;
    float *p;
    size_t floats;
};
static Scratch g_scr[16];
static int g_nscr = 0;
// ---- deferred weight-gradient reduces (conv_wgrad_kernels.h; include/lsnet_hip.h lsn_wgrad_defer) ----
// While deferral is on for a stream, the partial tiles of its weight-gradient calls (part_buffer(..., keep = true)) come from an
// ARENA that is not reused until the flush, and conv_wgrad_reduce() queues a descriptor instead of launching.
struct RJob {
    bool fold;
    const float *part, *part_b;
    float *gw, *gb;
    size_t n;
    int nb, splits, splits_b, R, Co;
    WgFold f;
};
struct Arena {
    hipStream_t st;
    float *p;
    size_t floats, used;
    long long defer_mb;   // 0: off
    std::vector<RJob> *jobs;
};
static Arena g_arena[16];
static int g_narena = 0;
static Arena *arena_of(hipStream_t st, bool create)
{
    for (int i = 0; i < g_narena; ++i)
        if (g_arena[i].st == st) return &g_arena[i];
    if (!create || g_narena == 16) return nullptr;
    g_arena[g_narena] = Arena{st, nullptr, 0, 0, 0, new std::vector<RJob>()};
    return &g_arena[g_narena++];
}
static int wgrad_flush(Arena *ar);

// `floats` of arena space for the partial tiles of ONE weight-gradient call.  When the arena is full the queued reduces run
// first (they consume every tile handed out so far, in stream order) and the arena starts over; while it is smaller than the
// caller's limit (lsn_wgrad_defer) it is also replaced by a larger block -- nothing refers to the old one after the flush, so it is
// freed once the stream has drained.  That synchronisation happens only while the arena is still finding its size (the first
// step at the largest shape); in the steady state this function does pointer arithmetic and nothing else.
static int arena_alloc(Arena *ar, size_t floats, float **p)
{
    const size_t off = (ar->used + 63) & ~(size_t)63;
    if (off + floats <= ar->floats) {
        *p = ar->p + off;
        ar->used = off + floats;
        return 0;
    }
    if (int rc = wgrad_flush(ar)) return rc;
    ar->used = 0;
    const size_t limit = (size_t)ar->defer_mb << 18;   // MB -> floats
    if (floats > ar->floats || ar->floats < limit) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(ar->st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
            return fail(LSN_ERR_RUNTIME, "scratch would grow inside a stream capture: run the step eagerly once before capturing");
        size_t want = 2 * ar->floats > off + floats ? 2 * ar->floats : off + floats;   // what this step has needed so far, at least doubled
        if (want > limit + floats) want = limit + floats;
        if (want < floats + floats / 4) want = floats + floats / 4;
        want += (size_t)4 << 20;
        if (ar->p) {
            LSN_HIP(hipStreamSynchronize(ar->st));
            lib_stat(STAT_BLOCKING_SYNCS, 1);
            LSN_HIP(hipFree(ar->p));
            lib_stat(STAT_HELD_BYTES, -(long long)(ar->floats * sizeof(float)));
            ar->p = nullptr, ar->floats = 0;
        }
        float *np = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&np), want * sizeof(float)) != hipSuccess) {
            //  no room for the arena: the queue is empty (flushed above), deferral ends for this stream and the call
            // -- like every later one -- takes the per-stream scratch block with an immediate reduce
            (void)hipGetLastError();
            ar->defer_mb = 0, ar->used = 0;
            return 1;
        }
        lib_stat(STAT_MALLOCS, 1), lib_stat(STAT_HELD_BYTES, (long long)(want * sizeof(float)));
        ar->p = np, ar->floats = want;
    }
    *p = ar->p;
    ar->used = floats;
    return 0;
}

static bool stream_capturing(hipStream_t st)
{
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
}

static int part_buffer(size_t floats, float **p, hipStream_t st, bool keep = false)
{
    if (keep) {
        // (never while the stream is being captured: a queued reduce is host state, a replay would not run it)
        Arena *ar = arena_of(st, false);
        if (ar && ar->defer_mb > 0 && !stream_capturing(st)) {
            const int rc = arena_alloc(ar, floats, p);
            if (rc != 1) return rc;   // 1: the arena could not be allocated, deferral is off now
        }
    }
    Scratch *sc = nullptr;
    for (int i = 0; i < g_nscr; ++i)
        if (g_scr[i].st == st) sc = &g_scr[i];
    if (!sc) {
        if (g_nscr == 16) return fail(LSN_ERR_RUNTIME, "scratch: more than 16 streams use the library");
        sc = &g_scr[g_nscr++];
        *sc = Scratch{st, nullptr, 0};
    }
    if (floats > sc->floats) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
            return fail(LSN_ERR_RUNTIME, "scratch would grow inside a stream capture: run the step eagerly once before capturing");
        const size_t want = floats + floats / 4;
        float *np = nullptr;
        LSN_HIP(hipMalloc(reinterpret_cast<void **>(&np), want * sizeof(float)));
        lib_stat(STAT_MALLOCS, 1), lib_stat(STAT_HELD_BYTES, (long long)(want * sizeof(float)));
        sc->p = np, sc->floats = want;   // (the old block stays allocated: see above)
    }
    *p = sc->p;
    return 0;
}

// Per-stream arrival counters of the stream-K tiles (conv_kernels.h ConvArgs): zeroed once, every launch leaves them zero.
constexpr int SK_MAX_TILES = 4096;
constexpr int LIB_TICKETS = 1024;   // behind the stream-K counters: self-resetting tickets of other kernels (norm.hip)
struct SkCounters {
    hipStream_t st;
    unsigned *p;
};
static SkCounters g_skc[16];
static int g_nskc = 0;
static int sk_counters(unsigned **p, hipStream_t st)
{
    for (int i = 0; i < g_nskc; ++i)
        if (g_skc[i].st == st) {
            *p = g_skc[i].p;
            return 0;
        }
    if (g_nskc == 16) return fail(LSN_ERR_RUNTIME, "scratch: more than 16 streams use the library");
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return fail(LSN_ERR_RUNTIME, "scratch would grow inside a stream capture: run the step eagerly once before capturing");
    unsigned *np = nullptr;
    LSN_HIP(hipMalloc(reinterpret_cast<void **>(&np), (SK_MAX_TILES + LIB_TICKETS) * sizeof(unsigned)));
    lib_stat(STAT_MALLOCS, 1), lib_stat(STAT_HELD_BYTES, (long long)((SK_MAX_TILES + LIB_TICKETS) * sizeof(unsigned)));
    LSN_HIP(hipMemsetAsync(np, 0, (SK_MAX_TILES + LIB_TICKETS) * sizeof(unsigned), st));
    g_skc[g_nskc++] = SkCounters{st, np};
    *p = np;
    return 0;
}

// norm.hip: LIB_TICKETS zero-initialised, self-resetting ticket counters of this stream (last-arriver reductions)
int lib_tickets(unsigned **p, hipStream_t st)
{
    unsigned *base = nullptr;
    if (int rc = sk_counters(&base, st)) return rc;
    *p = base + SK_MAX_TILES;
    return 0;
}

// Work distribution of one launch (conv_kernels.h ConvArgs).  The chip holds 512 workgroups of these kernels at once, so
// a launch of W tiles runs in ceil(W / 512) rounds and the last one is rarely full: 525 tiles (every convolution at the
// 100 x 168 maps of two images) cost 226 us against 177 us for 512 (tools/ubench/tile_sweep), 132 or 33 tiles (stages 3 / 4)
// leave most of the chip idle.  Rounds 3 / 4 answered with a reduction split over blockIdx.z plus a reduce launch, and a
// second launch for the tail; both quantise again (the split rule produced 528 workgroups for six layer shapes of the
// step).  Round 5: the last 512 + r tiles of a launch -- or all of them when there are fewer than 512 -- are divided
// EVENLY BY CHUNKS over 512 workgroups, whole tiles before them stay one workgroup each.
// Not for short reductions (< 16 chunks: the layer is bound by its output stream, and a partial tile costs what the tile
// costs) and not when the last round is nearly full.
static void sk_plan(int ntw, int Tall, int *n_dp, int *sk_n, int *sk_tiles)
{
    *n_dp = ntw, *sk_n = 0, *sk_tiles = 0;
#ifdef LSNET_AB_DIST   // build.py --ab -DLSNET_AB_DIST: rounds 3 / 4 work distribution (profiles/r5_sk_*.txt)
    return;
#endif
    constexpr int SLOTS = 512;
    // (launches of two or more whole rounds lose more to the segment loop's longer prologue in EVERY workgroup than the
    // last round can return: profiles/r5_sk_policy.txt, layer-1 rows)
    if (Tall < 8 || ntw >= 2 * SLOTS) return;
    const int r = ntw % SLOTS;
    if (r == 0 || r > 448) return;
    // Pieces per tile: at most four when the pieces are the whole launch (the tile's last arriver adds them on its own), up
    // to sixteen for the few tiles behind a whole round (the chip is idle otherwise); never shorter than four chunks.
    int per_tile = Tall / 4;
    const int cap = ntw > SLOTS ? 16 : 4;
    if (per_tile > cap) per_tile = cap;
    const long long pieces = (long long)r * per_tile;
    const int ns = pieces < SLOTS ? (int)pieces : SLOTS;
    if (ns <= r) return;
    *n_dp = ntw - r, *sk_n = ns, *sk_tiles = r;
}

// Fewer than 512 tiles and more than four pieces per tile (stage 4, FPN P5: 33 pixel tiles under 64 .. 144 chunks): the tile's
// last arriver would add eight partial tiles on its own while the chip idles -- these keep round 3's blockIdx.z split with its
// chip-wide reduce launch, rounded DOWN to one round of workgroups (round 3 / 4 rounded to nearest: 528 on 512 slots for six
// layer shapes of the step).
static int z_split(int ntw, int Tall)
{
    if (ntw >= 128 || Tall < 16) return 1;
    int ks = 512 / ntw;
    if (ks > Tall / 8) ks = Tall / 8;
    if (ks > 16) ks = 16;
    return ks < 1 ? 1 : ks;
}

// One launch over the pixel tiles [tile_base, tile_base + ntiles) of the (tiled) levels, `ks` chunk splits (the round-3
// form with its reduce launch: only where stream-K does not apply).
template <int TM, int TN, int WM, int WN, int NP, bool UNAL, bool FINE, bool TRANS>
static int launch_conv_range(ConvArgs &a, int ks, int tile_base, int ntiles, hipStream_t st)
{
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const size_t lds = (size_t)2 * SplitCfg<NP>::NPL * BM * 64;
    a.colblocks = (a.Co + BN - 1) / BN;
    const int ntw = ntiles * a.colblocks;
    a.n_dp = ntw, a.sk_n = 0, a.sk_tiles = 0, a.sk_part = nullptr, a.sk_cnt = nullptr;
    if (ks == 1 && !TRANS) {
        sk_plan(ntw, a.kh * a.kw * cv_ncc(a.C), &a.n_dp, &a.sk_n, &a.sk_tiles);
        // (the kernel's unit arithmetic is 32-bit: (U + 1) * sk_n must stay below 2^31)
        if (a.sk_tiles > SK_MAX_TILES || ((long long)a.sk_tiles * a.kh * a.kw * cv_ncc(a.C) + 1) * a.sk_n >= ((long long)1 << 31))
            a.n_dp = ntw, a.sk_n = 0, a.sk_tiles = 0;
        if (a.sk_n) {
            if (int rc = part_buffer((size_t)2 * a.sk_n * BM * BN, &a.sk_part, st)) return rc;
            if (int rc = sk_counters(&a.sk_cnt, st)) return rc;
        }
    }
    a.ksplit = ks;
    a.tile_base = tile_base;
    const int pix0 = tile_base * BM;   // (ks > 1: one level)
    const int rows = ks > 1 ? (a.lv[0].P - pix0 < ntiles * BM ? a.lv[0].P - pix0 : ntiles * BM) : 0;
    a.part_pix0 = pix0, a.part_rows = rows;
    const size_t n = (size_t)rows * a.Co;
    if (ks > 1)
        if (int rc = part_buffer(n * ks, &a.part, st)) return rc;
    dim3 grid(a.n_dp + a.sk_n, 1, ks);
    if (a.sk_n) {
        auto k = conv_mm_kernel<TM, TN, WM, WN, NP, UNAL, FINE, TRANS, true>;
        static bool attr_set = false;   // per instantiation
        if (!attr_set) {
            LSN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
        hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a);
    } else {
        auto k = conv_mm_kernel<TM, TN, WM, WN, NP, UNAL, FINE, TRANS, false>;
        static bool attr_set = false;
        if (!attr_set) {
            LSN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
        hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a);
    }
    if (ks > 1) {
        const int blocks = (int)((n / 4 + 255) / 256 < 1024 ? (n / 4 + 255) / 256 : 1024);
        const size_t o = (size_t)pix0 * a.Co;
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, a.part, a.lv[0].out + o, a.bias,
                           a.lv[0].res ? a.lv[0].res + o : nullptr, a.lv[0].gate ? a.lv[0].gate + o : nullptr, (int)n, a.Co, ks,
                           a.relu);
    }
    LSN_HIP(hipGetLastError());
    return 0;
}

#ifdef LSNET_AB_DIST   // build.py --ab -DLSNET_AB_DIST: rounds 3 / 4 work distribution (profiles/r5_sk_*.txt)
// Round 4's tail split (the A/B library keeps rounds 3 / 4's work distribution: blockIdx.z splits, this, no stream-K): a
// single-level launch whose last round would be a small remainder runs its last pixel tiles as a second launch with the
// reduction split `kt` ways (profiles/r4_tail_split.txt).
static int tail_split(int total_wg, int colblocks, int Tall, int *tail_tiles)
{
    const int rem = total_wg % 512;
    if (total_wg <= 512 || rem == 0 || rem > 128 || rem % colblocks != 0 || Tall < 48) return 1;
    int kt = 384 / rem;
    if (kt > Tall / 4) kt = Tall / 4;
    if (kt > 16) kt = 16;
    if (kt < 2) return 1;
    *tail_tiles = rem / colblocks;
    return kt;
}
#endif

template <int TM, int TN, int WM, int WN, int NP, bool UNAL, bool FINE, bool TRANS = false>
static int launch_conv_cfg(ConvArgs &a, int ks, hipStream_t st)
{
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    int tiles = 0;
    for (int i = 0; i < a.nlv; ++i) {
        a.lv[i].tile0 = tiles;
        tiles += (a.lv[i].P + BM - 1) / BM;
    }
    a.ntiles = tiles;
#ifdef LSNET_AB_DIST   // build.py --ab -DLSNET_AB_DIST: rounds 3 / 4 work distribution (profiles/r5_sk_*.txt)
    const int colblocks = (a.Co + BN - 1) / BN;
    int tail = 0;
    const int kt = (a.nlv == 1 && !a.ostep && ks == 1 && !TRANS) ? tail_split(tiles * colblocks, colblocks, a.kh * a.kw * cv_ncc(a.C), &tail)
                                                                : 1;
    if (kt > 1) {
        if (int rc = launch_conv_range<TM, TN, WM, WN, NP, UNAL, FINE, TRANS>(a, 1, 0, tiles - tail, st)) return rc;
        return launch_conv_range<TM, TN, WM, WN, NP, UNAL, FINE, TRANS>(a, kt, tiles - tail, tail, st);
    }
#endif
    return launch_conv_range<TM, TN, WM, WN, NP, UNAL, FINE, TRANS>(a, ks, 0, tiles, st);
}

// FINE: the fine MFMA / staging interleave (conv_kernels.h) -- adopted for the two wide tiles after the round-4 sweep; for
// the two narrow ones (Co <= 64) it measured no difference (4216 / 3095 vs 4218 / 3105 us, profiles/r4_conv_tiles.txt)
template <int TM, int TN, int WM, int WN, bool FINE>
static int launch_conv(ConvArgs &a, int ks, hipStream_t st)
{
    if constexpr (TN == 2 && WN == 2) {   // the unaligned-slab variant exists for the 64 x 128 tile
        if (a.C % 4 != 0 || a.xpitch % 4 != 0)
            return conv_np() == 3 ? launch_conv_cfg<TM, TN, WM, WN, 3, true, false>(a, ks, st)
                                  : launch_conv_cfg<TM, TN, WM, WN, 6, true, false>(a, ks, st);
    }
    if (a.C % 4 != 0 || a.xpitch % 4 != 0)
        return fail(LSN_ERR_UNSUPPORTED, "conv2d: C %% 4 != 0 needs more than 64 output channels");
    return conv_np() == 3 ? launch_conv_cfg<TM, TN, WM, WN, 3, false, FINE>(a, ks, st)
                          : launch_conv_cfg<TM, TN, WM, WN, 6, false, FINE>(a, ks, st);
}

// Tile choice (pixels x output channels per workgroup; two workgroups share a CU, so the chip holds 512 of them at
// once).  Round-4 sweep over the step's layer shapes (tools/ubench/conv_step, profiles/r4_conv_tiles.txt):
//   * 64 x 256 (a pixel slab is fetched and split ONCE for 256 output channels: half the split work per MFMA of the
//     128-wide tiles) wins wherever Co is a multiple of 256 and the reduction is deep (3x3) or the launch fills a round;
//   * 64 x 128 otherwise for Co > 64; 128 x 64 / 128 x 32 for the narrow outputs;
//   * 128 x 128 lost on every row (fewer, longer workgroups) and the one-workgroup-per-CU fat register tiles
//     (256 x 128, 128 x 256) lost by 30 %: both are gone.
// When even the narrow tile leaves most of the chip idle (few pixels under a deep reduction: layer 4, FPN P5 .. P7) the
// chunk range is split over blockIdx.z into partial tiles that a second small kernel adds up
// (profiles/r3_conv_ksplit.txt; forcing other split counts lost on every row of the round-4 sweep).
static int conv_forward(ConvArgs &a, hipStream_t st)
{
    auto blocks = [&](int bm, int bn) {
        int t = 0;
        for (int i = 0; i < a.nlv; ++i) t += (a.lv[i].P + bm - 1) / bm;
        return t * ((a.Co + bn - 1) / bn);
    };
    int cfg;   // 2: 64 x 128, 3: 128 x 64, 4: 128 x 32, 5: 64 x 256
    if (a.Co <= 32) cfg = 4;
    else if (a.Co <= 64) cfg = 3;
    else cfg = 2;
    // (round 5, with stream-K filling the chip either way: 64 x 256 for EVERY Co % 256 == 0 lost 3 % over the step's shapes,
    // forward 3944 -> 4078 us, data gradient 2871 -> 2959 us, tools/ubench/conv_step A/B -- the rule stands)
    if (a.Co % 256 == 0 && a.C % 4 == 0 && a.xpitch % 4 == 0 && (a.kh * a.kw > 1 || blocks(64, 256) >= 512)) cfg = 5;
    const int nb = cfg == 2 ? blocks(64, 128) : cfg == 3 ? blocks(128, 64) : cfg == 5 ? blocks(64, 256) : blocks(128, 32);
    const int Tall = a.kh * a.kw * cv_ncc(a.C);
    int ks = 1;
#ifdef LSNET_AB_DIST   // build.py --ab -DLSNET_AB_DIST: rounds 3 / 4 work distribution (profiles/r5_sk_*.txt)
    if (a.nlv == 1 && !a.ostep && nb <= 320 && Tall >= 16) {
        ks = (512 + nb / 2) / nb;
        if (ks > Tall / 8) ks = Tall / 8;
        if (ks > 16) ks = 16;
        if (ks < 1) ks = 1;
    }
#else
    // round 5: stream-K pieces (sk_plan) wherever a tile gets at most four of them, blockIdx.z splits + a reduce launch below
    if (a.nlv == 1 && !a.ostep) ks = z_split(nb, Tall);
#endif
    switch (cfg) {
    case 2: return launch_conv<1, 2, 2, 2, true>(a, ks, st);
    case 3: return launch_conv<1, 2, 4, 1, false>(a, ks, st);
    case 5: return launch_conv<2, 2, 1, 4, true>(a, ks, st);
    default: return launch_conv<1, 1, 4, 1, false>(a, ks, st);
    }
}

// dcn.hip: out_i[r][0 .. N) = x_i[r][0 .. Cr) . Wt for row blocks i (the backward-data GEMM of the deformable
// convolutions: a 1x1 convolution of grad_output whose N = K * C output columns are the column gradients); wf = the
// fragment image of the (N, 1, Cr) GEMM view.  No profiling span of its own: it runs inside the caller's.
int conv_mm_rows(int n, const float *const *x, float *const *out, const int *rows, int Cr, int xpitch, int N,
                 const unsigned short *wf, hipStream_t st)
{
    LSN_CHECK(n >= 1 && n <= CV_MAXLV && Cr % 4 == 0 && xpitch % 4 == 0 && xpitch >= Cr, "conv_mm_rows: bad arguments");
    ConvArgs a = {};
    a.nlv = n;
    for (int i = 0; i < n; ++i) {
        ConvLvl &L = a.lv[i];
        L.x = x[i], L.out = out[i], L.res = nullptr;
        L.B = 1, L.H = 1, L.W = rows[i], L.Ho = 1, L.Wo = rows[i], L.P = rows[i];
    }
    a.wf = wf;
    a.wf_bytes = (int)cv_wfrag_bytes(N, 1, Cr, conv_npl());
    a.C = Cr, a.Co = N, a.kh = a.kw = 1, a.stride = 1, a.pad_h = a.pad_w = 0, a.dil = 1;
    a.xpitch = xpitch;
    bool rows32 = true;   // the TRANS epilogue stores through a buffer descriptor of each row block: 32-bit byte offsets
    for (int i = 0; i < n; ++i) rows32 = rows32 && (long long)rows[i] * N * 4 < (1ll << 31) - 256;
    if (N % 256 == 0 && rows32) {   // the 64 x 256 tile with whole-line stores (conv_kernels.h TRANS); plain output, one split
#ifdef LSNET_AB_DIST   // build.py --ab -DLSNET_AB_DIST: rounds 3 / 4 work distribution (profiles/r5_sk_*.txt)
        return conv_forward(a, st);
#else
        return conv_np() == 3 ? launch_conv_cfg<2, 2, 1, 4, 3, false, true, true>(a, 1, st)
                              : launch_conv_cfg<2, 2, 1, 4, 6, false, true, true>(a, 1, st);
#endif
    }
    return conv_forward(a, st);
}

// optional folded BatchNorm of a prepare call (all NULL: a plain weight)
struct BnFold {
    const float *gamma = nullptr, *var = nullptr, *beta = nullptr, *mean = nullptr;
    float *shift_out = nullptr;
    float eps = 0.f;
};

static void wfrag_job(WfragJob &j, const float *w, unsigned short *out, int Co, int K, int C, int flipT, const TapSub &ts,
                      const BnFold &bn)
{
    j = WfragJob{};
    j.w = w, j.out = out, j.Co = Co, j.K = K, j.C = C, j.flipT = flipT, j.ts = ts;
    j.bn_gamma = bn.gamma, j.bn_var = bn.var, j.bn_beta = bn.beta, j.bn_mean = bn.mean, j.shift_out = bn.shift_out, j.bn_eps = bn.eps;
}

static void conv_wfrag(const float *w, unsigned short *out, int Co, int K, int C, int flipT, const TapSub &ts, hipStream_t st,
                       const BnFold &bn = BnFold())
{
    WfragJob j;
    wfrag_job(j, w, out, Co, K, C, flipT, ts, bn);
    const long long total = wfrag_threads(j);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    if (conv_npl() == 2)
        hipLaunchKernelGGL(conv_wfrag_kernel<2>, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, j);
    else
        hipLaunchKernelGGL(conv_wfrag_kernel<3>, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, j);
}

// prof.h span of one launch: 2 P Co C K flops; every operand once (the strided backward-data pass counts grad_out once
// per residue class it launches)
struct ConvProf : ProfSpan {
    static double fl(const ConvArgs &a)
    {
        double px = 0;
        for (int i = 0; i < a.nlv; ++i) px += (double)a.lv[i].P;
        return 2.0 * px * a.Co * a.C * a.kh * a.kw;
    }
    static double by(const ConvArgs &a)
    {
        double e = (double)a.Co * a.C * a.kh * a.kw;
        for (int i = 0; i < a.nlv; ++i)
            e += (double)a.lv[i].B * a.lv[i].H * a.lv[i].W * a.xpitch + (double)a.lv[i].P * a.Co;
        return 4.0 * e;
    }
    ConvProf(int fam, const ConvArgs &a, hipStream_t s) : ProfSpan(fam, fl(a), by(a), s) {}
};

// prof.h launch log: one record per forward / backward-data call (all residue classes of a strided data gradient included)
struct LaunchLog {
    LaunchRec r;
    hipStream_t st;
    bool on;
    LaunchLog(int kind, int n, const lsn_conv_level *lv, int C, int xpitch, int Co, int kh, int kw, int stride, int pad, int dil,
              int relu, hipStream_t s) : st(s), on(launch_log_on())
    {
        if (!on) return;
        memset(&r, 0, sizeof(r));
        int res = 0, gate = 0;
        for (int i = 0; i < n && i < 16; ++i) {
            r.B[i] = lv[i].B, r.H[i] = lv[i].H, r.W[i] = lv[i].W;
            res |= lv[i].residual != nullptr, gate |= lv[i].gate != nullptr;
        }
        const int v[13] = {kind, C, Co, kh, kw, stride, pad, dil, relu, xpitch, n, res, gate};
        memcpy(r.v, v, sizeof(v));
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) {
            on = false;
            return;
        }
        (void)hipEventRecord(r.e0, st);
    }
    ~LaunchLog()
    {
        if (!on) return;
        (void)hipEventRecord(r.e1, st);
        launch_log_push(r);
    }
    LaunchLog(const LaunchLog &) = delete;
    LaunchLog &operator=(const LaunchLog &) = delete;
};

static int conv_out_size(int in, int k, int stride, int pad, int dil) { return (in + 2 * pad - (dil * (k - 1) + 1)) / stride + 1; }

static int conv_check(int B, int H, int W, int C, int Co, int kh, int kw, int stride, int pad, int dil, int *Ho,
                      int *Wo)
{
    LSN_CHECK(B > 0 && H > 0 && W > 0 && C > 0 && Co > 0 && kh > 0 && kw > 0, "conv2d: empty tensor");
    LSN_CHECK(stride > 0 && dil > 0 && pad >= 0, "conv2d: bad stride / dilation / padding");
    if (kh * kw > 64) return fail(LSN_ERR_UNSUPPORTED, "conv2d kernel takes at most 64 taps, got %d x %d", kh, kw);
    *Ho = conv_out_size(H, kh, stride, pad, dil);
    *Wo = conv_out_size(W, kw, stride, pad, dil);
    LSN_CHECK(*Ho > 0 && *Wo > 0, "conv2d: output size is too small");
    if ((int64_t)B * H * W * C * 4 >= ((int64_t)1 << 31) || (int64_t)Co * kh * kw * C * 4 >= ((int64_t)1 << 31) ||
        (int64_t)B * *Ho * *Wo * Co * 4 >= ((int64_t)1 << 31))
        return fail(LSN_ERR_UNSUPPORTED, "conv2d: tensor too large for 32-bit buffer offsets");
    return 0;
}

// Scratch for a weight image when the caller of a one-shot entry point passes no workspace: stream-ordered allocation.
struct TempImage {
    void *p = nullptr;
    hipStream_t st;
    explicit TempImage(hipStream_t s) : st(s) {}
    int alloc(size_t bytes)
    {
        LSN_HIP(hipMallocAsync(&p, bytes, st));
        lib_stat(STAT_POOL_ALLOCS, 1);
        return 0;
    }
    ~TempImage()
    {
        if (p) (void)hipFreeAsync(p, st);
    }
};

static int conv_forward_impl(int n, const lsn_conv_level *lv, const void *prepared, const float *bias, int C, int xpitch,
                             int Co, int kh, int kw, int stride, int pad, int dil, int relu, hipStream_t st)
{
    LSN_CHECK(n >= 1 && n <= CV_MAXLV && lv && prepared, "conv2d: bad level list");
    LSN_CHECK(xpitch > 0, "conv2d: bad pixel pitch %d", xpitch);
    LaunchLog log(0, n, lv, C, xpitch, Co, kh, kw, stride, pad, dil, relu, st);
    if ((C % 4 != 0 || xpitch % 4 != 0) && Co <= 64)
        return fail(LSN_ERR_UNSUPPORTED, "conv2d: C %% 4 != 0 needs more than 64 output channels");
    ConvArgs a = {};
    a.nlv = n;
    for (int i = 0; i < n; ++i) {
        ConvLvl &L = a.lv[i];
        LSN_CHECK(lv[i].x && lv[i].out, "conv2d: NULL tensor in level %d", i);
        if (int rc = conv_check(lv[i].B, lv[i].H, lv[i].W, C, Co, kh, kw, stride, pad, dil, &L.Ho, &L.Wo)) return rc;
        if (xpitch != C) {
            // row-merged form: the C "channels" of a tap are C / xpitch horizontally adjacent pixels; they must stay
            // inside the row (the caller pads the image), and the output grid follows the REAL pixels
            LSN_CHECK(kw == 1 && pad == 0 && dil == 1 && C % xpitch == 0, "conv2d: row-merged form needs kw = 1, pad = 0, dil = 1");
            L.Wo = (lv[i].W - C / xpitch) / stride + 1;
            LSN_CHECK(L.Wo > 0, "conv2d: output size is too small");
        }
        L.x = lv[i].x, L.out = lv[i].out, L.B = lv[i].B, L.H = lv[i].H, L.W = lv[i].W;
        L.res = lv[i].residual, L.gate = lv[i].gate;
        L.P = L.B * L.Ho * L.Wo;
    }
    a.bias = bias;
    a.wf = reinterpret_cast<const unsigned short *>(prepared);
    a.wf_bytes = (int)cv_wfrag_bytes(Co, kh * kw, C, conv_npl());
    a.C = C, a.Co = Co, a.kh = kh, a.kw = kw, a.stride = stride, a.pad_h = a.pad_w = pad, a.dil = dil;
    a.xpitch = xpitch;
    a.relu = relu;
    ConvProf prof(PROF_CONV_FWD, a, st);
    return conv_forward(a, st);
}

// Residue classes of the transposed convolution (backward-data of a stride-s convolution):
//   grad_in[y, x] = sum over taps (i, j) with (y + pad - i dil) % s == 0 (same for x) of
//                   grad_out[(y + pad - i dil) / s, (x + pad - j dil) / s] . w[:, i, j, :]
// Each class (y % s, x % s) of input pixels is a stride-1 convolution of grad_out with its own subset of the taps (an
// arithmetic progression): no zero-stuffed samples, no wasted products.  Classes without a tap keep the zeros of a memset.
struct BwdClass {
    TapSub ts;
    int py, px, pad_h, pad_w, dstep;
    size_t wf_off;   // byte offset of the class's weight image
};
struct BwdPlan {
    BwdClass cls[64];
    int ncls;
    bool need_zero;
    size_t bytes;
};

static int bwd_plan(int C, int Co, int kh, int kw, int s, int pad, int dil, BwdPlan *pl)
{
    if (s * s > 64) return fail(LSN_ERR_UNSUPPORTED, "conv2d backward-data: stride %d is not supported", s);
    pl->ncls = 0, pl->need_zero = false, pl->bytes = 0;
    auto taps_of = [&](int p, int k, int &t0, int &tstep, int &nt) {
        t0 = -1, tstep = 1, nt = 0;
        int prev = -1;
        for (int t = 0; t < k; ++t)
            if (((p + pad - t * dil) % s + s) % s == 0) {
                if (nt == 0) t0 = t;
                if (nt == 1) tstep = t - prev;
                prev = t;
                ++nt;
            }
    };
    for (int py = 0; py < s; ++py)
        for (int px = 0; px < s; ++px) {
            BwdClass c;
            c.py = py, c.px = px;
            taps_of(py, kh, c.ts.i0, c.ts.istep, c.ts.ni);
            taps_of(px, kw, c.ts.j0, c.ts.jstep, c.ts.nj);
            c.ts.kw = kw;
            if (c.ts.ni == 0 || c.ts.nj == 0) {
                pl->need_zero = true;
                continue;
            }
            // y_in = yy + q0 - m d' (m-th tap of the subset), d' = istep dil / s; as a correlation over the reversed
            // subset: y_in = yy - P + m' d' with P = (ni - 1) d' - q0
            const int dstep_h = c.ts.istep * dil / s, dstep_w = c.ts.jstep * dil / s;
            if (c.ts.ni > 1 && c.ts.nj > 1 && dstep_h != dstep_w)
                return fail(LSN_ERR_UNSUPPORTED, "conv2d backward-data: unequal tap spacing");
            c.dstep = c.ts.ni > 1 ? dstep_h : (c.ts.nj > 1 ? dstep_w : 1);
            if ((c.ts.ni > 1 && c.ts.istep * dil % s != 0) || (c.ts.nj > 1 && c.ts.jstep * dil % s != 0))
                return fail(LSN_ERR_UNSUPPORTED, "conv2d backward-data: stride / dilation combination");
            const int q0h = (py + pad - c.ts.i0 * dil) / s, q0w = (px + pad - c.ts.j0 * dil) / s;
            c.pad_h = (c.ts.ni - 1) * c.dstep - q0h;
            c.pad_w = (c.ts.nj - 1) * c.dstep - q0w;
            c.wf_off = pl->bytes;
            // the transposed convolution reduces over the forward Co and produces the forward C
            pl->bytes += cv_wfrag_bytes(C, c.ts.ni * c.ts.nj, Co, conv_npl());
            pl->cls[pl->ncls++] = c;
        }
    return 0;
}

// grad_in of every level (B, H, W, C) from grad_out (B, Ho, Wo, Co): lv[i].x = grad_out, lv[i].out = grad_in, and
// lv[i].B / H / W are the INPUT sizes of the forward convolution
static int conv_backward_data_impl(int n, const lsn_conv_level *lv, const void *prepared, int C, int Co, int kh, int kw,
                                   int stride, int pad, int dil, hipStream_t st)
{
    LSN_CHECK(n >= 1 && n <= CV_MAXLV && lv && prepared, "conv2d backward: bad arguments");
    LaunchLog log(1, n, lv, C, C, Co, kh, kw, stride, pad, dil, 0, st);
    if (Co % 4 != 0 && C <= 64) return fail(LSN_ERR_UNSUPPORTED, "conv2d backward-data: Co %% 4 != 0 needs C > 64");
    const int s = stride;
    int Ho[CV_MAXLV], Wo[CV_MAXLV];
    for (int i = 0; i < n; ++i) {
        LSN_CHECK(lv[i].x && lv[i].out, "conv2d backward: NULL tensor in level %d", i);
        if (int rc = conv_check(lv[i].B, lv[i].H, lv[i].W, C, Co, kh, kw, stride, pad, dil, &Ho[i], &Wo[i])) return rc;
    }
    if (s > 1 && n > 1) return fail(LSN_ERR_UNSUPPORTED, "conv2d backward-data: strided convolutions take one level per call");
    BwdPlan pl;
    if (int rc = bwd_plan(C, Co, kh, kw, s, pad, dil, &pl)) return rc;
    const int H = lv[0].H, W = lv[0].W, B = lv[0].B;
    bool epi = false;   // a residual (another path's gradient of the same tensor) / ReLU gate in the epilogue
    for (int i = 0; i < n; ++i) epi = epi || lv[i].residual || lv[i].gate;
    if (epi && pl.need_zero)
        return fail(LSN_ERR_UNSUPPORTED, "conv2d backward-data: residual / gate with residue classes that have no tap");
    if (epi && C % 4 != 0) return fail(LSN_ERR_UNSUPPORTED, "conv2d backward-data: residual / gate need C %% 4 == 0");
    if (pl.need_zero) LSN_HIP(hipMemsetAsync(lv[0].out, 0, sizeof(float) * (size_t)B * H * W * C, st));
    for (int ci = 0; ci < pl.ncls; ++ci) {
        const BwdClass &c = pl.cls[ci];
        const int Hc = c.py < H ? (H - c.py + s - 1) / s : 0, Wc = c.px < W ? (W - c.px + s - 1) / s : 0;
        if (s > 1 && (Hc == 0 || Wc == 0)) continue;
        ConvArgs a = {};
        a.wf = reinterpret_cast<const unsigned short *>(reinterpret_cast<const unsigned char *>(prepared) + c.wf_off);
        a.wf_bytes = (int)cv_wfrag_bytes(C, c.ts.ni * c.ts.nj, Co, conv_npl());
        a.nlv = n;
        for (int i = 0; i < n; ++i) {
            ConvLvl &L = a.lv[i];
            L.x = lv[i].x, L.out = lv[i].out, L.res = lv[i].residual, L.gate = lv[i].gate;
            L.B = lv[i].B, L.H = Ho[i], L.W = Wo[i];
            L.Ho = s > 1 ? Hc : lv[i].H, L.Wo = s > 1 ? Wc : lv[i].W;
            L.P = L.B * L.Ho * L.Wo;
        }
        a.C = Co, a.Co = C, a.kh = c.ts.ni, a.kw = c.ts.nj, a.stride = 1;
        a.xpitch = Co;
        a.pad_h = c.pad_h, a.pad_w = c.pad_w, a.dil = c.dstep;
        if (s > 1) a.ostep = s, a.oy0 = c.py, a.ox0 = c.px, a.OH = H, a.OW = W;
        ConvProf prof(PROF_CONV_BWD_DATA, a, st);
        if (int rc = conv_forward(a, st)) return rc;
    }
    return 0;
}

static int64_t prepared_bytes(int kind, int C, int Co, int kh, int kw, int stride, int pad, int dil)
{
    if (kind == 0) return (int64_t)cv_wfrag_bytes(Co, kh * kw, C, conv_npl());
    if (kind == 2) return (int64_t)cv_wfrag_bytes(kh * kw * C, 1, Co, conv_npl());   // deformable backward GEMM: N = K C columns
    BwdPlan pl;
    if (bwd_plan(C, Co, kh, kw, stride, pad, dil, &pl)) return -1;
    return (int64_t)pl.bytes;
}

static int bn_fold_of(const lsn_conv_wprep &p, BnFold *bn)
{
    *bn = BnFold();
    if (!p.bn_gamma) return 0;
    LSN_CHECK(p.bn_var, "conv2d prepare: incomplete BatchNorm description");
    if (p.kind == 0)
        LSN_CHECK(p.bn_beta && p.bn_mean && p.shift_out, "conv2d prepare: incomplete BatchNorm description");
    else
        LSN_CHECK(!p.shift_out, "conv2d prepare: the shift belongs to the forward image");
    bn->gamma = p.bn_gamma, bn->var = p.bn_var, bn->beta = p.bn_beta, bn->mean = p.bn_mean, bn->shift_out = p.shift_out;
    bn->eps = p.bn_eps;
    return 0;
}

static int prepare_weights(int kind, const float *w, void *prepared, int C, int Co, int kh, int kw, int stride, int pad,
                           int dil, hipStream_t st, const BnFold &bn = BnFold())
{
    LSN_CHECK(w && prepared, "conv2d prepare: NULL pointer");
    LSN_CHECK(C > 0 && Co > 0 && kh > 0 && kw > 0 && kh * kw <= 64, "conv2d prepare: bad weight shape");
    if (prepared_bytes(kind, C, Co, kh, kw, stride, pad, dil) >= ((int64_t)1 << 31))
        return fail(LSN_ERR_UNSUPPORTED, "conv2d: weight too large for 32-bit buffer offsets");
    if (kind == 0) {
        conv_wfrag(w, reinterpret_cast<unsigned short *>(prepared), Co, kh * kw, C, 0, TapSub{}, st, bn);
    } else if (kind == 2) {
        conv_wfrag(w, reinterpret_cast<unsigned short *>(prepared), kh * kw * C, 1, Co, 1, TapSub{0, 1, 1, 0, 1, 1, 1}, st, bn);
    } else {
        BwdPlan pl;
        if (int rc = bwd_plan(C, Co, kh, kw, stride, pad, dil, &pl)) return rc;
        for (int ci = 0; ci < pl.ncls; ++ci)
            conv_wfrag(w, reinterpret_cast<unsigned short *>(reinterpret_cast<unsigned char *>(prepared) + pl.cls[ci].wf_off),
                       C, kh * kw, Co, 1, pl.cls[ci].ts, st, bn);
    }
    LSN_HIP(hipGetLastError());
    return 0;
}

// ---- weight / bias gradient (conv_wgrad_kernels.h) ----
int conv_wgrad_reduce(const float *part, float *gw, size_t n, const float *part_b, float *gb, int nb, int splits, int splits_b,
                      int accumulate, hipStream_t st);
// (state of a folded-norm weight-gradient call, see conv_wgrad_reduce below)
static thread_local const WgFoldJobs *g_wg_fold = nullptr;
bool conv_wgrad_fold_pending() { return g_wg_fold != nullptr; }
static int wg_jobs() { return g_wg_fold ? g_wg_fold->njobs : 1; }
template <int TI, int TJ, int TG, int WI, int WJ, int PMAX, bool UN_OK>
static int launch_wgrad_cfg(WgArgs &a, float *gw, float *gb, int accumulate, hipStream_t st)
{
    const bool un = a.Co % 4 != 0;
    if (un && !UN_OK) return 1;
    constexpr int BN = WI * TI * 32, BM = WJ * TJ * 32;
    const int npl = conv_npl(), K = a.kh * a.kw;
    const int blocks = cdiv(a.Co, BM) * cdiv(a.C, BN);
    const size_t nW = (size_t)a.Co * K * a.C;
    const int nj = wg_jobs();   // > 1: the levels are jobs with their own outputs; every split stays inside one level
#ifdef LSNET_AB_DIST   // build.py --ab -DLSNET_AB_DIST: rounds 3 / 4 work distribution (profiles/r5_sk_*.txt)
    int S = (512 + blocks * nj / 2) / (blocks * nj);
#else
    int S = 512 / (blocks * nj);   // one round of workgroups (rounds 3 / 4 rounded to nearest: 516 .. 540 on 512 slots)
#endif
    const size_t cap = ((size_t)192 << 20) / 4 / (nW + a.Co) / nj;   // partial tiles: at most 192 MB
    if ((size_t)S > cap) S = (int)cap;
    if (S > a.nseg / nj / 6) S = a.nseg / nj / 6;   // a split should run long enough to amortise its prologue and its partial tile
    if (S < 1) S = 1;
    S *= nj;
    float *part = nullptr;
    if (int rc = part_buffer((size_t)S * (nW + a.Co) + 16, &part, st, true)) return rc;
    a.part = part;
    a.part_b = part + (((size_t)S * nW + 3) & ~(size_t)3);
    a.want_bias = gb != nullptr;
    constexpr int XP = 1024 / BN, GP = 1024 / BM;   // pixels per staging pass (conv_wgrad_kernel)
    const size_t lds = (size_t)2 * npl * ((size_t)cdiv(PMAX, XP) * XP * BN * 2 + (size_t)cdiv(16, GP) * GP * BM * 2);
    if (lds > 160 * 1024) return 1;
    dim3 grid(blocks, S);
    LSN_CHECK((long long)a.nseg * (S + 1) < (1ll << 32), "conv2d backward-weight: %d segments x %d splits exceed the kernel's 32-bit split arithmetic", a.nseg, S);
    auto launch = [&](auto kern) -> int {
        LSN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
        return 0;
    };
    int rc;
    // the fine MFMA / staging interleave (conv_wgrad_kernels.h FINE; round-4 A/B over the step's layer shapes 3.57 -> 3.41 ms)
    // wherever it compiles without spills: not the unaligned form, not the 4 x 1 wave layout of the narrow 3x3 form
    constexpr bool FINE = !(TG == 9 && WI == 4);
    if constexpr (UN_OK) {
        if (un)
            rc = npl == 2 ? launch(conv_wgrad_kernel<TI, TJ, TG, WI, WJ, 3, true, PMAX, false>)
                          : launch(conv_wgrad_kernel<TI, TJ, TG, WI, WJ, 6, true, PMAX, false>);
        else
            rc = npl == 2 ? launch(conv_wgrad_kernel<TI, TJ, TG, WI, WJ, 3, false, PMAX, FINE>)
                          : launch(conv_wgrad_kernel<TI, TJ, TG, WI, WJ, 6, false, PMAX, FINE>);
    } else {
        rc = npl == 2 ? launch(conv_wgrad_kernel<TI, TJ, TG, WI, WJ, 3, false, PMAX, FINE>)
                      : launch(conv_wgrad_kernel<TI, TJ, TG, WI, WJ, 6, false, PMAX, FINE>);
    }
    if (rc) return rc;
    return conv_wgrad_reduce(a.part, gw, nW, a.part_b, gb, a.Co, S, S, accumulate, st);
}

// dcn.hip: gw (+)= sum of `splits` partial gradients of n floats (n % 4 == 0), gb (+)= sum of splits_b partial rows of nb
// With a BatchNorm fold pending (lsn_conv2d_backward_weight_bn set g_wg_fold for the duration of its weight-gradient
// call) the reduce also applies the norm's scale and forms grad_gamma / grad_beta (conv_wgrad_reduce_bn_kernel); gb is then
// the entry point's dummy bias gradient (it only makes the main kernels write the per-channel sums of g).

static int reduce_ls(int splits)
{
#ifdef LSNET_PER_LANE
    constexpr int PER_LANE = LSNET_PER_LANE;
#else
    constexpr int PER_LANE = 16;
#endif
    int LS = 1;
    while (LS < 64 && LS * PER_LANE <= splits) LS <<= 1;
    return LS;
}

int conv_wgrad_reduce(const float *part, float *gw, size_t n, const float *part_b, float *gb, int nb, int splits, int splits_b,
                      int accumulate, hipStream_t st)
{
    Arena *ar = arena_of(st, false);
    if (ar && ar->defer_mb > 0 && accumulate && !stream_capturing(st)) {
        // queue instead of launching.  A gradient that already has a queued job is reduced first: two jobs of one launch must
        // not add onto the same addresses.
        const int nj = g_wg_fold ? g_wg_fold->njobs : 1;
        bool clash = false;
        for (const RJob &q : *ar->jobs)
            for (int j = 0; j < nj; ++j) {
                const float *tw = g_wg_fold ? (nj > 1 ? g_wg_fold->gw[j] : gw) : gw;
                clash = clash || q.gw == tw || (gb && q.gb == gb) || (g_wg_fold && q.fold && q.f.dgamma == g_wg_fold->f[j].dgamma);
            }
        if (clash)
            if (int rc = wgrad_flush(ar)) return rc;   // (the arena is NOT rewound here: this call's partial tiles are in it)
        if (g_wg_fold) {
            LSN_CHECK(part_b && nb > 0 && n % ((size_t)nb * 4) == 0 && splits % nj == 0 && splits_b % nj == 0,
                      "conv2d backward-weight (folded norm): bad partial layout");
            for (int j = 0; j < nj; ++j) {
                RJob q = {};
                q.fold = true;
                q.part = part + (size_t)j * (splits / nj) * n, q.part_b = part_b + (size_t)j * (splits_b / nj) * nb;
                q.gw = nj > 1 ? g_wg_fold->gw[j] : gw;
                q.n = n, q.nb = nb, q.splits = splits / nj, q.splits_b = splits_b / nj, q.R = (int)(n / nb), q.Co = nb;
                q.f = g_wg_fold->f[j];
                ar->jobs->push_back(q);
            }
        } else {
            RJob q = {};
            q.fold = false, q.part = part, q.part_b = part_b, q.gw = gw, q.gb = gb, q.n = n, q.nb = nb, q.splits = splits, q.splits_b = splits_b;
            ar->jobs->push_back(q);
        }
        if ((long long)(ar->used * sizeof(float)) > (ar->defer_mb << 20)) {
            if (int rc = wgrad_flush(ar)) return rc;
            ar->used = 0;   // every queued tile has been consumed (in stream order): the arena starts over
        }
        return 0;
    }
    int LS = 1;
    // partial tiles a lane sums on its own before the xor-shuffles (LS = splits / PER_LANE lanes share a float4).  Round-4 sweep
    // (tools/ubench/wgrad_ab rule / bn, us per step of the benchmark's layer mix): 4: 3374 / 2945, 8: 3284 / 2729,
    // 16: 3254 / 2650, 32: 3252 / 2642 -- fewer, longer lanes win until the loads in flight run out.
#ifdef LSNET_PER_LANE
    constexpr int PER_LANE = LSNET_PER_LANE;
#else
    constexpr int PER_LANE = 16;
#endif
    while (LS < 64 && LS * PER_LANE <= splits) LS <<= 1;   // >= PER_LANE / 2 loads per lane; a wave's lanes share 64 / LS elements
    if (g_wg_fold) {
        const int nj = g_wg_fold->njobs;   // (splits / splits_b count ALL jobs' partial tiles)
        LSN_CHECK(part_b && nb > 0 && n % ((size_t)nb * 4) == 0 && splits % nj == 0 && splits_b % nj == 0,
                  "conv2d backward-weight (folded norm): bad partial layout");
        LS = 1;
        while (LS < 64 && LS * PER_LANE <= splits / nj) LS <<= 1;
        hipLaunchKernelGGL(conv_wgrad_reduce_bn_kernel, dim3(nb, nj), dim3(256), 0, st, part, gw, (int)(n / nb), nb, part_b,
                           splits / nj, splits_b / nj, accumulate, LS, *g_wg_fold);
        LSN_HIP(hipGetLastError());
        return 0;
    }
    const size_t thr = n / 4 * LS;
    const int rb = (int)((thr + 255) / 256 < 4096 ? (thr + 255) / 256 : 4096);
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(rb > 0 ? rb : 1), dim3(256), 0, st, part, gw, n, part_b, gb, nb, splits,
                       splits_b, accumulate, LS);
    LSN_HIP(hipGetLastError());
    return 0;
}

static int wgrad_flush(Arena *ar)
{
    std::vector<RJob> &jobs = *ar->jobs;
    if (jobs.empty()) return 0;
    hipStream_t st = ar->st;
    double bytes = 0;
    for (const RJob &q : jobs) bytes += 4.0 * ((double)q.splits * q.n + 2.0 * q.n);
    ProfSpan prof(PROF_CONV_WGRAD, 0.0, bytes, st);   // (the family's time includes its reduces, wherever they are launched)
    RJobsPlain P = {};
    RJobsFold F = {};
    int pblk = 0, fblk = 0;
    auto launch_plain = [&]() {
        if (P.njobs) hipLaunchKernelGGL(conv_wgrad_reduce_multi_kernel, dim3(pblk), dim3(256), 0, st, P);
        P.njobs = 0, pblk = 0;
    };
    auto launch_fold = [&]() {
        if (F.njobs) hipLaunchKernelGGL(conv_wgrad_reduce_bn_multi_kernel, dim3(fblk), dim3(256), 0, st, F);
        F.njobs = 0, fblk = 0;
    };
    for (const RJob &q : jobs) {
        const int LS = reduce_ls(q.splits);
        if (q.fold) {
            if (F.njobs == RJ_MAX) launch_fold();
            RJobFold &d = F.j[F.njobs++];
            d.part = q.part, d.part_b = q.part_b, d.gw = q.gw, d.f = q.f, d.R = q.R, d.Co = q.Co, d.splits = q.splits,
            d.splits_b = q.splits_b, d.LS = LS, d.blk0 = fblk;
            fblk += q.Co;
        } else {
            if (P.njobs == RJ_MAX) launch_plain();
            const size_t thr = q.n / 4 * LS;
            const int rb = (int)((thr + 255) / 256 < 4096 ? (thr + 255) / 256 : 4096);
            RJobPlain &d = P.j[P.njobs++];
            d.part = q.part, d.part_b = q.part_b, d.gw = q.gw, d.gb = q.gb, d.n = q.n, d.nb = q.nb, d.splits = q.splits,
            d.splits_b = q.splits_b, d.LS = LS, d.blk0 = pblk, d.nblk = rb > 0 ? rb : 1;
            pblk += d.nblk;
        }
    }
    launch_plain();
    launch_fold();
    jobs.clear();
    LSN_HIP(hipGetLastError());
    return 0;
}

// dcn.hip (stream-K pieces of the deformable forward): slots in this stream's scratch block and its tile counters
int conv_sk_scratch(size_t floats, float **part, unsigned **cnt, hipStream_t st)
{
    if (int rc = part_buffer(floats, part, st)) return rc;
    return sk_counters(cnt, st);
}
int conv_sk_max_tiles() { return SK_MAX_TILES; }

// library-owned scratch (also for dcn.hip's weight-gradient pass): grows, never shrinks
int conv_scratch(size_t floats, float **p, hipStream_t st) { return part_buffer(floats, p, st, true); }   // (weight-gradient passes of dcn.hip)

// Returns 1 when the shape is not served here (more than nine taps, C % 4 != 0, 64-bit offsets): the caller keeps the
// general kernel of dcn.hip.
int conv_wgrad_mm(int n, const lsn_conv_level *lv, float *gw, float *gb, int C, int Co, int kh, int kw, int stride, int pad,
                  int dil, int accumulate, hipStream_t st)
{
    // 3x3 / stride 1 / dilation 1 and 1x1 (any stride) run here.  (A strided 3x3 needs a 3 x 33 patch: one workgroup per
    // CU and 2-way bank-conflicted tr-reads -- the general kernel of dcn.hip was faster.)
    const bool k33 = kh == 3 && kw == 3 && stride == 1 && dil == 1, k11 = kh == 1 && kw == 1;
    if (n < 1 || n > CV_MAXLV || !(k33 || k11) || C % 4 != 0) return 1;
    WgArgs a = {};
    int seg = 0;
    for (int i = 0; i < n; ++i) {
        WgLvl &L = a.lv[i];
        const int B = lv[i].B, H = lv[i].H, W = lv[i].W;
        if (!lv[i].x || !lv[i].grad_out || B <= 0 || H <= 0 || W <= 0) return 1;
        const int Ho = conv_out_size(H, kh, stride, pad, dil), Wo = conv_out_size(W, kw, stride, pad, dil);
        if (Ho <= 0 || Wo <= 0) return 1;
        if ((int64_t)B * H * W * C * 4 >= ((int64_t)1 << 31) || (int64_t)B * Ho * Wo * Co * 4 >= ((int64_t)1 << 31)) return 1;
        L.x = lv[i].x, L.go = lv[i].grad_out, L.B = B, L.H = H, L.W = W, L.Ho = Ho, L.Wo = Wo;
        L.nsx = cdiv(Wo, 16);
        L.seg0 = seg;
        seg += B * Ho * L.nsx;
    }
    a.nlv = n, a.nseg = seg;
    a.C = C, a.Co = Co, a.kh = kh, a.kw = kw, a.stride = stride, a.pad = pad, a.dil = dil;
    a.cstep = kw == 1 ? stride : 1;
    a.PW = kw == 1 ? 16 : 15 * stride + (kw - 1) * dil + 1;
    if (kh * a.PW > 99) return 1;
    double px = 0, in_el = 0;
    for (int i = 0; i < n; ++i) px += (double)a.lv[i].B * a.lv[i].Ho * a.lv[i].Wo, in_el += (double)a.lv[i].B * a.lv[i].H * a.lv[i].W * C;
    ProfSpan prof(PROF_CONV_WGRAD, 2.0 * px * Co * C * kh * kw, 4.0 * (in_el + px * Co + (double)Co * kh * kw * C), st);
    if (kh * kw == 1) {   // one tap: 32 .. 64 channels per wave on either side, by the layer's width
        // (round-4 sweep, profiles/r4_wgrad_tiles.txt: forcing any one of the four tiles on every layer loses to this
        // width rule by 0 .. 25 %)
        if (C <= 64)
            return Co <= 64 ? launch_wgrad_cfg<1, 1, 1, 2, 2, 16, false>(a, gw, gb, accumulate, st)
                            : launch_wgrad_cfg<1, 2, 1, 2, 2, 16, false>(a, gw, gb, accumulate, st);
        return Co <= 64 ? launch_wgrad_cfg<2, 1, 1, 2, 2, 16, false>(a, gw, gb, accumulate, st)
                        : launch_wgrad_cfg<2, 2, 1, 2, 2, 16, false>(a, gw, gb, accumulate, st);
    }
    if (Co <= 32) return launch_wgrad_cfg<1, 1, 9, 4, 1, 54, true>(a, gw, gb, accumulate, st);
    return launch_wgrad_cfg<1, 1, 9, 2, 2, 54, false>(a, gw, gb, accumulate, st);
}

// ---- every stale image of a step in one launch ----
static WfragJob *g_jobs_dev = nullptr;
static size_t g_jobs_cap = 0;
static std::vector<WfragJob> g_jobs_host;   // what the device table holds

static int prepare_weights_multi(int n, const lsn_conv_wprep *it, hipStream_t st)
{
    LSN_CHECK(n >= 0 && (n == 0 || it), "conv2d prepare: bad item list");
    std::vector<WfragJob> jobs;
    long long total = 0;
    for (int i = 0; i < n; ++i) {
        const lsn_conv_wprep &p = it[i];
        LSN_CHECK(p.w && p.prepared && p.C > 0 && p.Co > 0 && p.kh > 0 && p.kw > 0 && p.kh * p.kw <= 64,
                  "conv2d prepare: bad item %d", i);
        if (prepared_bytes(p.kind, p.C, p.Co, p.kh, p.kw, p.stride, p.pad, p.dil) >= ((int64_t)1 << 31))
            return fail(LSN_ERR_UNSUPPORTED, "conv2d: weight too large for 32-bit buffer offsets");
        BnFold bn;
        if (int rc = bn_fold_of(p, &bn)) return rc;
        if (p.kind == 0) {
            WfragJob j;
            wfrag_job(j, p.w, reinterpret_cast<unsigned short *>(p.prepared), p.Co, p.kh * p.kw, p.C, 0, TapSub{}, bn);
            j.start = total;
            total += wfrag_threads(j);
            jobs.push_back(j);
        } else if (p.kind == 2) {
            WfragJob j;
            wfrag_job(j, p.w, reinterpret_cast<unsigned short *>(p.prepared), p.kh * p.kw * p.C, 1, p.Co, 1, TapSub{0, 1, 1, 0, 1, 1, 1}, bn);
            j.start = total;
            total += wfrag_threads(j);
            jobs.push_back(j);
        } else {
            BwdPlan pl;
            if (int rc = bwd_plan(p.C, p.Co, p.kh, p.kw, p.stride, p.pad, p.dil, &pl)) return rc;
            for (int c = 0; c < pl.ncls; ++c) {
                WfragJob j;
                wfrag_job(j, p.w, reinterpret_cast<unsigned short *>(reinterpret_cast<unsigned char *>(p.prepared) + pl.cls[c].wf_off),
                          p.C, p.kh * p.kw, p.Co, 1, pl.cls[c].ts, bn);
                j.start = total;
                total += wfrag_threads(j);
                jobs.push_back(j);
            }
        }
    }
    if (jobs.empty()) return 0;
    const size_t bytes = jobs.size() * sizeof(WfragJob);
    const bool same = jobs.size() == g_jobs_host.size() && memcmp(jobs.data(), g_jobs_host.data(), bytes) == 0;
    if (!same) {   // the table changes only when the set of weights does (first steps): upload then, reuse afterwards
        if (jobs.size() > g_jobs_cap) {
            // (an outgrown table is retired, not freed: a captured graph may hold its address)
            LSN_HIP(hipMalloc(reinterpret_cast<void **>(&g_jobs_dev), (jobs.size() + 64) * sizeof(WfragJob)));
            lib_stat(STAT_MALLOCS, 1), lib_stat(STAT_HELD_BYTES, (long long)((jobs.size() + 64) * sizeof(WfragJob)));
            g_jobs_cap = jobs.size() + 64;
        }
        LSN_HIP(hipStreamSynchronize(st));   // (an earlier launch may still read the old table)
        lib_stat(STAT_BLOCKING_SYNCS, 1);
        LSN_HIP(hipMemcpy(g_jobs_dev, jobs.data(), bytes, hipMemcpyHostToDevice));
        g_jobs_host = jobs;
    }
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (conv_npl() == 2)
        hipLaunchKernelGGL(conv_wfrag_multi_kernel<2>, dim3(blocks), dim3(256), 0, st, g_jobs_dev, (int)jobs.size(), total);
    else
        hipLaunchKernelGGL(conv_wfrag_multi_kernel<3>, dim3(blocks), dim3(256), 0, st, g_jobs_dev, (int)jobs.size(), total);
    LSN_HIP(hipGetLastError());
    return 0;
}

}  // namespace lsn

extern "C" {

int lsn_wgrad_defer(int max_mbytes, lsn_stream_t stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    lsn::Arena *ar = lsn::arena_of(st, max_mbytes > 0);
    if (!ar) return max_mbytes > 0 ? lsn::fail(LSN_ERR_RUNTIME, "scratch: more than 16 streams use the library") : 0;
    if (max_mbytes > 0) {
        ar->jobs->clear();   // (a step that died half-way may have left descriptors behind: they are dropped, not run)
        ar->used = 0;
        ar->defer_mb = max_mbytes;
        return 0;
    }
    const int rc = lsn::wgrad_flush(ar);
    ar->defer_mb = 0, ar->used = 0;
    return rc;
}

int lsn_wgrad_flush(lsn_stream_t stream)
{
    lsn::Arena *ar = lsn::arena_of(reinterpret_cast<hipStream_t>(stream), false);
    if (!ar) return 0;
    const int rc = lsn::wgrad_flush(ar);
    ar->used = 0;
    return rc;
}

int lsn_scratch_stats(long long *out4)
{
    LSN_CHECK(out4 != nullptr, "lsn_scratch_stats: NULL pointer");
    for (int i = 0; i < 4; ++i) out4[i] = lsn::g_stat[i];
    return 0;
}

int lsn_conv2d_backward_weight_bn(const float *x, const float *g, const float *w, const float *bn_gamma, const float *bn_mean,
                                  const float *bn_var, float bn_eps, float *grad_w, float *grad_gamma, float *grad_beta, int B,
                                  int H, int W, int C, int Co, int kh, int kw, int stride, int pad, int dil, int accumulate,
                                  lsn_stream_t stream)
{
    LSN_CHECK(x && g && w && bn_gamma && bn_mean && bn_var && grad_w && grad_gamma && grad_beta,
              "conv2d backward-weight (folded norm): NULL pointer");
    if (((size_t)kh * kw * C) % 4 != 0)
        return lsn::fail(LSN_ERR_UNSUPPORTED, "conv2d backward-weight (folded norm): kh * kw * C %% 4 != 0");
    lsn::WgFoldJobs f = {};
    f.f[0] = lsn::WgFold{w, bn_gamma, bn_mean, bn_var, grad_gamma, grad_beta, bn_eps};
    f.gw[0] = grad_w, f.njobs = 1;
    lsn::g_wg_fold = &f;
    // grad_beta stands in as the bias gradient: the main kernels then write the per-channel partial sums of g, and the
    // fold-aware reduce is the only writer of grad_w / grad_gamma / grad_beta
    const int rc = lsn_conv2d_backward_weight(x, g, grad_w, grad_beta, B, H, W, C, Co, kh, kw, stride, pad, dil, accumulate, stream);
    lsn::g_wg_fold = nullptr;
    return rc;
}

int lsn_conv2d_backward_weight_bn_jobs(int n_jobs, const lsn_wgrad_bn_job *jobs, int B, int H, int W, int C, int Co, int kh, int kw,
                                       int stride, int pad, int dil, int accumulate, lsn_stream_t stream)
{
    LSN_CHECK(n_jobs >= 1 && n_jobs <= 8 && jobs, "conv2d backward-weight jobs: 1 .. 8 jobs");
    lsn::WgFoldJobs f = {};
    lsn_conv_level lv[8] = {};
    for (int j = 0; j < n_jobs; ++j) {
        const lsn_wgrad_bn_job &q = jobs[j];
        LSN_CHECK(q.x && q.g && q.w && q.bn_gamma && q.bn_mean && q.bn_var && q.grad_w && q.grad_gamma && q.grad_beta,
                  "conv2d backward-weight jobs: NULL pointer in job %d", j);
        f.f[j] = lsn::WgFold{q.w, q.bn_gamma, q.bn_mean, q.bn_var, q.grad_gamma, q.grad_beta, q.bn_eps};
        f.gw[j] = q.grad_w;
        lv[j].x = q.x, lv[j].grad_out = q.g, lv[j].B = B, lv[j].H = H, lv[j].W = W;
    }
    f.njobs = n_jobs;
    int rc = 1;
    if (n_jobs > 1 && ((size_t)kh * kw * C) % 4 == 0) {   // the patch kernel with the jobs as levels; 1: shape not served there
        lsn::g_wg_fold = &f;
        rc = lsn::conv_wgrad_mm(n_jobs, lv, jobs[0].grad_w, jobs[0].grad_beta, C, Co, kh, kw, stride, pad, dil, accumulate,
                                reinterpret_cast<hipStream_t>(stream));
        lsn::g_wg_fold = nullptr;
    }
    if (rc != 1) return rc;
    for (int j = 0; j < n_jobs; ++j) {
        const lsn_wgrad_bn_job &q = jobs[j];
        if (int r = lsn_conv2d_backward_weight_bn(q.x, q.g, q.w, q.bn_gamma, q.bn_mean, q.bn_var, q.bn_eps, q.grad_w, q.grad_gamma,
                                                  q.grad_beta, B, H, W, C, Co, kh, kw, stride, pad, dil, accumulate, stream))
            return r;
    }
    return 0;
}

int lsn_conv2d_prepare_weights_multi(int n_items, const lsn_conv_wprep *items, lsn_stream_t stream)
{
    return lsn::prepare_weights_multi(n_items, items, reinterpret_cast<hipStream_t>(stream));
}

int64_t lsn_conv2d_prepared_bytes(int kind, int C, int Co, int kh, int kw, int stride, int pad, int dil)
{
    return lsn::prepared_bytes(kind, C, Co, kh, kw, stride, pad, dil);
}

int lsn_conv2d_prepare_weights(int kind, const float *w, void *prepared, int C, int Co, int kh, int kw, int stride,
                               int pad, int dil, lsn_stream_t stream)
{
    return lsn::prepare_weights(kind, w, prepared, C, Co, kh, kw, stride, pad, dil, reinterpret_cast<hipStream_t>(stream));
}

int lsn_conv2d_prepare_weights_item(const lsn_conv_wprep *item, lsn_stream_t stream)
{
    LSN_CHECK(item != nullptr, "conv2d prepare: NULL item");
    lsn::BnFold bn;
    if (int rc = lsn::bn_fold_of(*item, &bn)) return rc;
    return lsn::prepare_weights(item->kind, item->w, item->prepared, item->C, item->Co, item->kh, item->kw, item->stride,
                                item->pad, item->dil, reinterpret_cast<hipStream_t>(stream), bn);
}

int lsn_conv2d_forward_prepared(int n_levels, const lsn_conv_level *levels, const void *prepared, const float *bias, int C,
                                int xpitch, int Co, int kh, int kw, int stride, int pad, int dil, int relu,
                                lsn_stream_t stream)
{
    return lsn::conv_forward_impl(n_levels, levels, prepared, bias, C, xpitch, Co, kh, kw, stride, pad, dil, relu,
                                  reinterpret_cast<hipStream_t>(stream));
}

int lsn_conv2d_backward_data_prepared(int n_levels, const lsn_conv_level *levels, const void *prepared, int C, int Co,
                                      int kh, int kw, int stride, int pad, int dil, lsn_stream_t stream)
{
    return lsn::conv_backward_data_impl(n_levels, levels, prepared, C, Co, kh, kw, stride, pad, dil,
                                        reinterpret_cast<hipStream_t>(stream));
}

// ---- one-shot forms: weight image built by the call (into `workspace`, or a stream-ordered temporary) ----
static int one_shot(int kind, int n_levels, const lsn_conv_level *levels, const float *w, const float *bias, void *workspace,
                    int C, int xpitch, int Co, int kh, int kw, int stride, int pad, int dil, int relu, lsn_stream_t stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!w) return lsn::fail(LSN_ERR_INVALID, "conv2d: weight is NULL");
    const int64_t bytes = lsn::prepared_bytes(kind, C, Co, kh, kw, stride, pad, dil);
    if (bytes < 0) return LSN_ERR_UNSUPPORTED;
    lsn::TempImage tmp(st);
    if (!workspace) {
        if (int rc = tmp.alloc((size_t)bytes)) return rc;
        workspace = tmp.p;
    }
    if (int rc = lsn::prepare_weights(kind, w, workspace, C, Co, kh, kw, stride, pad, dil, st)) return rc;
    if (kind == 0)
        return lsn::conv_forward_impl(n_levels, levels, workspace, bias, C, xpitch, Co, kh, kw, stride, pad, dil, relu, st);
    return lsn::conv_backward_data_impl(n_levels, levels, workspace, C, Co, kh, kw, stride, pad, dil, st);
}

int lsn_conv2d_forward_multi(int n_levels, const lsn_conv_level *levels, const float *w, const float *bias, void *workspace,
                             int C, int Co, int kh, int kw, int stride, int pad, int dil, int relu, lsn_stream_t stream)
{
    return one_shot(0, n_levels, levels, w, bias, workspace, C, C, Co, kh, kw, stride, pad, dil, relu, stream);
}

int lsn_conv2d_backward_data_multi(int n_levels, const lsn_conv_level *levels, const float *w, float *wt_workspace, int C,
                                   int Co, int kh, int kw, int stride, int pad, int dil, lsn_stream_t stream)
{
    return one_shot(1, n_levels, levels, w, nullptr, wt_workspace, C, C, Co, kh, kw, stride, pad, dil, 0, stream);
}

int lsn_conv2d_forward_pitched(const float *x, const float *w, const float *bias, float *out, void *workspace, int B,
                               int H, int W, int C, int xpitch, int Co, int kh, int kw, int stride, int pad, int dil,
                               int relu, lsn_stream_t stream)
{
    lsn_conv_level L = {};
    L.x = x, L.out = out, L.B = B, L.H = H, L.W = W;
    return one_shot(0, 1, &L, w, bias, workspace, C, xpitch, Co, kh, kw, stride, pad, dil, relu, stream);
}

int lsn_conv2d_forward(const float *x, const float *w, const float *bias, float *out, void *workspace, int B, int H,
                       int W, int C, int Co, int kh, int kw, int stride, int pad, int dil, int relu,
                       lsn_stream_t stream)
{
    return lsn_conv2d_forward_pitched(x, w, bias, out, workspace, B, H, W, C, C, Co, kh, kw, stride, pad, dil, relu, stream);
}

int lsn_conv2d_backward_data(const float *grad_out, const float *w, float *grad_in, float *wt_workspace, int B,
                             int H, int W, int C, int Co, int kh, int kw, int stride, int pad, int dil,
                             lsn_stream_t stream)
{
    lsn_conv_level L = {};
    L.x = grad_out, L.out = grad_in, L.B = B, L.H = H, L.W = W;
    return lsn_conv2d_backward_data_multi(1, &L, w, wt_workspace, C, Co, kh, kw, stride, pad, dil, stream);
}

}  // extern "C"
