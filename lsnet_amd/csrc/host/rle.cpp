// liblsnet_host.so -- run-length masks for the evaluation path (C ABI in include/lsnet_host.h).
//
// A mask is a vector of run lengths over the column-major pixel order, background first.  Everything here is exact
// integer work except the polygon boundary walk, whose double arithmetic and truncations follow COCO's rasterisation
// rule operation for operation (reference: cocoapi/pycocotools/common/maskApi.c, cited per function in the header):
// the result has to be the same mask, run for run, or AP numbers would not be comparable.
#include "../../../include/lsnet_host.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

namespace {

using Runs = std::vector<uint32_t>;

size_t emit(const Runs &r, uint32_t *out, size_t cap)
{
    const size_t n = std::min(r.size(), cap);
    if (out) std::copy(r.begin(), r.begin() + n, out);
    return r.size();
}

// ---------------------------------------------------------------------------------------------- polygon -> runs
struct Pt { int u, v; };

// Dense integer points along one edge of the 5x up-sampled polygon, in walking order a -> b.  One point per step
// of the major axis; the minor coordinate is interpolated from the end with the SMALLER major coordinate and
// rounded by +.5 and truncation (so an edge rasterises identically in both walking directions).
void walk_edge(int ax, int ay, int bx, int by, std::vector<Pt> &out)
{
    const int dx = std::abs(bx - ax), dy = std::abs(by - ay);
    if (dx == 0 && dy == 0) {          // repeated vertex: one point (its minor coordinate is never looked at)
        out.push_back({ax, ay});
        return;
    }
    if (dx >= dy) {
        const bool rev = ax > bx;
        const int x0 = rev ? bx : ax, y0 = rev ? by : ay, y1 = rev ? ay : by;
        const double slope = static_cast<double>(y1 - y0) / dx;
        for (int d = 0; d <= dx; ++d) {
            const int t = rev ? dx - d : d;
            out.push_back({t + x0, static_cast<int>(y0 + slope * t + .5)});
        }
    } else {
        const bool rev = ay > by;
        const int y0 = rev ? by : ay, x0 = rev ? bx : ax, x1 = rev ? ax : bx;
        const double slope = static_cast<double>(x1 - x0) / dy;
        for (int d = 0; d <= dy; ++d) {
            const int t = rev ? dy - d : d;
            out.push_back({static_cast<int>(x0 + slope * t + .5), t + y0});
        }
    }
}

Runs polygon_runs(const double *xy, size_t k, uint32_t h, uint32_t w)
{
    const double scale = 5;
    std::vector<int> px(k + 1), py(k + 1);
    for (size_t j = 0; j < k; ++j) {
        px[j] = static_cast<int>(scale * xy[2 * j] + .5);
        py[j] = static_cast<int>(scale * xy[2 * j + 1] + .5);
    }
    if (k) { px[k] = px[0]; py[k] = py[0]; }
    std::vector<Pt> pts;
    for (size_t j = 0; j < k; ++j) walk_edge(px[j], py[j], px[j + 1], py[j + 1], pts);

    // wherever the walk moves to another (up-sampled) column it crosses a column boundary; crossings that fall on a
    // pixel-column centre toggle the mask from that row on.  Start positions of the toggles, in pixels:
    std::vector<uint32_t> pos;
    for (size_t j = 1; j < pts.size(); ++j) {
        if (pts[j].u == pts[j - 1].u) continue;
        double xd = static_cast<double>(pts[j].u < pts[j - 1].u ? pts[j].u : pts[j].u - 1);
        xd = (xd + .5) / scale - .5;
        if (std::floor(xd) != xd || xd < 0 || xd > static_cast<double>(w) - 1) continue;
        double yd = static_cast<double>(pts[j].v < pts[j - 1].v ? pts[j].v : pts[j - 1].v);
        yd = (yd + .5) / scale - .5;
        if (yd < 0) yd = 0; else if (yd > h) yd = h;
        yd = std::ceil(yd);
        pos.push_back(static_cast<uint32_t>(static_cast<int>(xd) * static_cast<int>(h) + static_cast<int>(yd)));
    }
    pos.push_back(static_cast<uint32_t>(static_cast<uint64_t>(h) * w));
    std::sort(pos.begin(), pos.end());

    // toggle positions -> run lengths; an empty run means two toggles at one pixel: they cancel, the neighbouring
    // runs fuse
    Runs gaps(pos.size());
    uint32_t prev = 0;
    for (size_t j = 0; j < pos.size(); ++j) { gaps[j] = pos[j] - prev; prev = pos[j]; }
    Runs runs;
    size_t j = 0;
    runs.push_back(gaps[j++]);
    while (j < gaps.size()) {
        if (gaps[j] > 0) { runs.push_back(gaps[j++]); continue; }
        ++j;
        if (j < gaps.size()) runs.back() += gaps[j++];
    }
    return runs;
}

// ---------------------------------------------------------------------------------------------- run arithmetic
// A cursor over one mask's runs: `left` pixels remain in the current run of value `on`.  Advancing never skips a
// run, so empty runs inside an input survive a merge exactly as they do in the reference.
struct Cursor {
    const uint32_t *runs;
    size_t m, next;
    uint32_t left;
    bool on;
    Cursor(const uint32_t *r, size_t n) : runs(r), m(n), next(1), left(n ? r[0] : 0), on(false) {}
    void take(uint32_t c)
    {
        left -= c;
        if (!left && next < m) { left = runs[next++]; on = !on; }
    }
};

Runs merge_two(const uint32_t *a, size_t ma, const uint32_t *b, size_t mb, bool intersect)
{
    Cursor A(a, ma), B(b, mb);
    Runs out;
    bool v = false;
    uint32_t acc = 0, remaining = 1;
    while (remaining > 0) {
        const uint32_t c = std::min(A.left, B.left);
        acc += c;
        A.take(c);
        B.take(c);
        remaining = A.left + B.left;
        const bool was = v;
        v = intersect ? (A.on && B.on) : (A.on || B.on);
        if (v != was || remaining == 0) { out.push_back(acc); acc = 0; }
    }
    return out;
}

void overlap(const uint32_t *a, size_t ma, const uint32_t *b, size_t mb, uint32_t &inter, uint32_t &uni)
{
    Cursor A(a, ma), B(b, mb);
    inter = uni = 0;
    uint32_t remaining = 1;
    while (remaining > 0) {
        const uint32_t c = std::min(A.left, B.left);
        if (A.on || B.on) { uni += c; if (A.on && B.on) inter += c; }
        A.take(c);
        B.take(c);
        remaining = A.left + B.left;
    }
}

uint32_t area_of(const uint32_t *r, size_t m)
{
    uint32_t a = 0;
    for (size_t j = 1; j < m; j += 2) a += r[j];
    return a;
}

// first / last pixel of every foreground run -> extent; a run that continues into the next column covers all rows
void bbox_of(const uint32_t *r, size_t m, uint32_t h, uint32_t w, double *bb)
{
    m = (m / 2) * 2;
    if (m == 0) { bb[0] = bb[1] = bb[2] = bb[3] = 0; return; }
    uint32_t xs = w, ys = h, xe = 0, ye = 0, cc = 0, xp = 0;
    for (size_t j = 0; j < m; ++j) {
        cc += r[j];
        const uint32_t t = cc - static_cast<uint32_t>(j % 2), y = t % h, x = (t - y) / h;
        if (j % 2 == 0) xp = x; else if (xp < x) { ys = 0; ye = h - 1; }
        xs = std::min(xs, x); xe = std::max(xe, x); ys = std::min(ys, y); ye = std::max(ye, y);
    }
    bb[0] = xs; bb[2] = xe - xs + 1;
    bb[1] = ys; bb[3] = ye - ys + 1;
}

double box_iou(const double *d, const double *g, bool crowd)
{
    const double da = d[2] * d[3], ga = g[2] * g[3];
    const double w = std::fmin(d[2] + d[0], g[2] + g[0]) - std::fmax(d[0], g[0]);
    if (w <= 0) return 0;
    const double h = std::fmin(d[3] + d[1], g[3] + g[1]) - std::fmax(d[1], g[1]);
    if (h <= 0) return 0;
    const double i = w * h;
    return i / (crowd ? da : da + ga - i);
}

}  // namespace

extern "C" {

size_t lsn_rle_from_polygon(const double *xy, size_t k, uint32_t h, uint32_t w, uint32_t *counts, size_t cap)
{
    return emit(polygon_runs(xy, k, h, w), counts, cap);
}

size_t lsn_rle_from_bbox(const double *bbox, uint32_t h, uint32_t w, uint32_t *counts, size_t cap)
{
    const double xs = bbox[0], xe = xs + bbox[2], ys = bbox[1], ye = ys + bbox[3];
    const double xy[8] = {xs, ys, xs, ye, xe, ye, xe, ys};
    return emit(polygon_runs(xy, 4, h, w), counts, cap);
}

size_t lsn_rle_merge(const uint32_t *counts, const size_t *offsets, size_t n, int intersect, uint32_t *out, size_t cap)
{
    if (n == 0) return 0;
    Runs acc(counts + offsets[0], counts + offsets[1]);
    for (size_t i = 1; i < n; ++i)
        acc = merge_two(acc.data(), acc.size(), counts + offsets[i], offsets[i + 1] - offsets[i], intersect != 0);
    return emit(acc, out, cap);
}

void lsn_rle_area(const uint32_t *counts, const size_t *offsets, size_t n, uint32_t *area)
{
    for (size_t i = 0; i < n; ++i) area[i] = area_of(counts + offsets[i], offsets[i + 1] - offsets[i]);
}

void lsn_rle_to_bbox(const uint32_t *counts, const size_t *offsets, size_t n, const uint32_t *hs, const uint32_t *ws,
                     double *bbox)
{
    for (size_t i = 0; i < n; ++i)
        bbox_of(counts + offsets[i], offsets[i + 1] - offsets[i], hs[i], ws[i], bbox + 4 * i);
}

void lsn_rle_iou(const uint32_t *dt_counts, const size_t *dt_offsets, const uint32_t *dt_h, const uint32_t *dt_w,
                 size_t m, const uint32_t *gt_counts, const size_t *gt_offsets, const uint32_t *gt_h,
                 const uint32_t *gt_w, size_t n, const uint8_t *iscrowd, double *out)
{
    std::vector<double> db(4 * m), gb(4 * n);
    lsn_rle_to_bbox(dt_counts, dt_offsets, m, dt_h, dt_w, db.data());
    lsn_rle_to_bbox(gt_counts, gt_offsets, n, gt_h, gt_w, gb.data());
    for (size_t d = 0; d < m; ++d)
        for (size_t g = 0; g < n; ++g) {
            const bool crowd = iscrowd && iscrowd[g];
            double &o = out[d * n + g];
            o = box_iou(&db[4 * d], &gb[4 * g], crowd);
            if (!(o > 0)) continue;                       // disjoint extents: disjoint masks
            if (dt_h[d] != gt_h[g] || dt_w[d] != gt_w[g]) { o = -1; continue; }
            const uint32_t *a = dt_counts + dt_offsets[d], *b = gt_counts + gt_offsets[g];
            const size_t ma = dt_offsets[d + 1] - dt_offsets[d], mb = gt_offsets[g + 1] - gt_offsets[g];
            uint32_t inter, uni;
            overlap(a, ma, b, mb, inter, uni);
            if (inter == 0) uni = 1; else if (crowd) uni = area_of(a, ma);
            o = static_cast<double>(inter) / static_cast<double>(uni);
        }
}

void lsn_bbox_iou(const double *dt, size_t m, const double *gt, size_t n, const uint8_t *iscrowd, double *out)
{
    for (size_t d = 0; d < m; ++d)
        for (size_t g = 0; g < n; ++g) out[d * n + g] = box_iou(dt + 4 * d, gt + 4 * g, iscrowd && iscrowd[g]);
}

size_t lsn_rle_encode(const uint8_t *mask, uint32_t h, uint32_t w, uint32_t *counts, size_t cap)
{
    const size_t hw = static_cast<size_t>(h) * w;
    Runs runs;
    uint8_t cur = 0;
    uint32_t len = 0;
    for (size_t j = 0; j < hw; ++j) {
        if (mask[j] != cur) { runs.push_back(len); len = 0; cur = mask[j]; }
        ++len;
    }
    runs.push_back(len);
    return emit(runs, counts, cap);
}

void lsn_rle_decode(const uint32_t *counts, size_t m, uint8_t *mask, size_t hw)
{
    size_t p = 0;
    uint8_t v = 0;
    for (size_t j = 0; j < m; ++j) {
        for (uint32_t k = 0; k < counts[j] && p < hw; ++k) mask[p++] = v;
        v = !v;
    }
}

// 5 payload bits per character (+48), bit 5 = "more follows"; values from the fourth run on are stored as the
// difference to the run two places back (same-valued neighbours are similar), sign-extended two's complement.
size_t lsn_rle_to_string(const uint32_t *counts, size_t m, char *s, size_t cap)
{
    size_t p = 0;
    for (size_t i = 0; i < m; ++i) {
        long x = static_cast<long>(counts[i]);
        if (i > 2) x -= static_cast<long>(counts[i - 2]);
        bool more = true;
        while (more) {
            char c = static_cast<char>(x & 0x1f);
            x >>= 5;
            more = (c & 0x10) ? x != -1 : x != 0;
            if (more) c |= 0x20;
            if (p < cap) s[p] = static_cast<char>(c + 48);
            ++p;
        }
    }
    if (p < cap) s[p] = 0;
    return p;
}

size_t lsn_rle_from_string(const char *s, uint32_t *counts, size_t cap)
{
    size_t m = 0, p = 0;
    std::vector<long> seen;
    while (s[p]) {
        long x = 0;
        int k = 0;
        bool more = true;
        while (more) {
            const char c = static_cast<char>(s[p] - 48);
            x |= static_cast<long>(c & 0x1f) << (5 * k);
            more = (c & 0x20) != 0;
            ++p; ++k;
            if (!more && (c & 0x10)) x |= -1L << (5 * k);
        }
        if (m > 2) x += seen[m - 2];
        seen.push_back(static_cast<long>(static_cast<uint32_t>(x)));
        if (m < cap) counts[m] = static_cast<uint32_t>(x);
        ++m;
    }
    return m;
}

}  // extern "C"
