// Host-side integer curve tables (closed forms; no floating point anywhere).
//
// The reference builds these with pure-Python float64 recursion at module
// construction (src/curves/space_filling_curves.py:134-251,471-491: 0.6 s at
// n=224, 2.2 s at n=384).  Here every curve is generated directly on the
// integer grid:
//   hilbert : classic iterative d2xy, whose (x, y) equals the reference's
//             (row, col) for every order
//   z       : bit de-interleave with the reference's child order (App. A.3)
//   moore   : four order-(k-1) Hilbert frames, integer frame recursion on
//             doubled coordinates (frames of space_filling_curves.py:239-245)
//   peano   : integer 3x3 pattern recursion (patterns of :95-108)
// then pruned to the n x n domain in curve order (embed_and_prune_sfc, :471-491).
#include "common_host.h"

#include <algorithm>
#include <vector>

namespace {

struct Cell { int32_t i, j; };  // i = row (H index), j = column (W index)

int order_for(int n, int base) {
    int order = 0;
    int64_t g = 1;
    while (g < n) { g *= base; order++; }
    return order;
}

void gen_hilbert(int order, std::vector<Cell> &out) {
    const int64_t P = int64_t(1) << order;
    out.resize(size_t(P * P));
    for (int64_t d = 0; d < P * P; d++) {
        int64_t x = 0, y = 0, t = d;
        for (int64_t s = 1; s < P; s <<= 1) {
            int64_t rx = 1 & (t >> 1);
            int64_t ry = 1 & (t ^ rx);
            if (ry == 0) {
                if (rx == 1) { x = s - 1 - x; y = s - 1 - y; }
                std::swap(x, y);
            }
            x += s * rx;
            y += s * ry;
            t >>= 2;
        }
        out[size_t(d)] = {int32_t(x), int32_t(y)};
    }
}

void gen_z(int order, std::vector<Cell> &out) {
    const int64_t P = int64_t(1) << order;
    out.resize(size_t(P * P));
    for (int64_t d = 0; d < P * P; d++) {
        int32_t i = 0, j = 0;
        for (int b = 0; b < order; b++) {
            int q = int((d >> (2 * b)) & 3);
            i |= ((~q) & 1) << b;   // visited (x+1/2,y),(x,y),(x+1/2,y+1/2),(x,y+1/2)
            j |= (q >> 1) << b;
        }
        out[size_t(d)] = {i, j};
    }
}

// Hilbert frame recursion on coordinates doubled so that cell centres are odd
// integers: frame = origin (x0,y0) + axis vectors (xi,xj), (yi,yj).
void hilbert_frame(std::vector<Cell> &out, int64_t x0, int64_t y0, int64_t xi, int64_t xj,
                   int64_t yi, int64_t yj, int n) {
    if (n <= 0) {
        int64_t x = x0 + (xi + yi) / 2, y = y0 + (xj + yj) / 2;   // odd => centre of a cell
        out.push_back({int32_t(x >> 1), int32_t(y >> 1)});
        return;
    }
    hilbert_frame(out, x0, y0, yi / 2, yj / 2, xi / 2, xj / 2, n - 1);
    hilbert_frame(out, x0 + xi / 2, y0 + xj / 2, xi / 2, xj / 2, yi / 2, yj / 2, n - 1);
    hilbert_frame(out, x0 + xi / 2 + yi / 2, y0 + xj / 2 + yj / 2, xi / 2, xj / 2, yi / 2, yj / 2, n - 1);
    hilbert_frame(out, x0 + xi / 2 + yi, y0 + xj / 2 + yj, -yi / 2, -yj / 2, -xi / 2, -xj / 2, n - 1);
}

void gen_moore(int order, std::vector<Cell> &out) {
    const int64_t S = int64_t(2) << order;   // doubled side length
    out.clear();
    out.reserve(size_t(1) << (2 * order));
    if (order == 0) { out.push_back({0, 0}); return; }
    const int64_t xi = S, xj = 0, yi = 0, yj = S;
    // (i, j) = (x, y): the reference's final rotation by 2*pi is the identity
    hilbert_frame(out, xi / 2, xj / 2, -xi / 2, xj / 2, yi / 2, yj / 2, order - 1);
    hilbert_frame(out, xi / 2 + yi / 2, xj / 2 + yj / 2, -xi / 2, xj / 2, yi / 2, yj / 2, order - 1);
    hilbert_frame(out, xi / 2 + yi, xj / 2 + yj, xi / 2, xj / 2, yi / 2, -yj / 2, order - 1);
    hilbert_frame(out, xi / 2 + yi / 2, xj / 2 + yj / 2, xi / 2, xj / 2, yi / 2, -yj / 2, order - 1);
}

const int8_t PEANO[4][9][3] = {
    {{0, 0, 0}, {1, 0, 1}, {2, 0, 0}, {2, 1, 1}, {1, 1, 0}, {0, 1, 1}, {0, 2, 0}, {1, 2, 1}, {2, 2, 0}},
    {{2, 0, 1}, {1, 0, 0}, {0, 0, 1}, {0, 1, 0}, {1, 1, 1}, {2, 1, 0}, {2, 2, 1}, {1, 2, 0}, {0, 2, 1}},
    {{0, 2, 2}, {1, 2, 3}, {2, 2, 2}, {2, 1, 3}, {1, 1, 2}, {0, 1, 3}, {0, 0, 2}, {1, 0, 3}, {2, 0, 2}},
    {{2, 2, 3}, {1, 2, 2}, {0, 2, 3}, {0, 1, 2}, {1, 1, 3}, {2, 1, 2}, {2, 0, 3}, {1, 0, 2}, {0, 0, 3}},
};

// Cells are produced as (x, y); the reference's post-transform swaps them.
void peano_rec(std::vector<Cell> &out, int32_t x, int32_t y, int32_t size, int order, int pattern) {
    if (order == 0) { out.push_back({y, x}); return; }
    size /= 3;
    for (int idx = 0; idx < 9; idx++) {
        size_t start = out.size();
        peano_rec(out, x + PEANO[pattern][idx][0] * size, y + PEANO[pattern][idx][1] * size, size,
                  order - 1, PEANO[pattern][idx][2]);
        if (idx % 3 == 1) std::reverse(out.begin() + long(start), out.end());
    }
}

int build(int curve, int n, std::vector<Cell> &kept) {
    if (n <= 0 || n > 4096) return sfcvit::fail(SFCVIT_EINVAL, "curve table: n=%d out of range (1..4096)", n);
    std::vector<Cell> full;
    switch (curve) {
    case SFCVIT_CURVE_HILBERT: gen_hilbert(order_for(n, 2), full); break;
    case SFCVIT_CURVE_Z: gen_z(order_for(n, 2), full); break;
    case SFCVIT_CURVE_MOORE: gen_moore(order_for(n, 2), full); break;
    case SFCVIT_CURVE_PEANO: {
        int order = order_for(n, 3);
        int32_t P = 1;
        for (int i = 0; i < order; i++) P *= 3;
        full.reserve(size_t(P) * P);
        peano_rec(full, 0, 0, P, order, 0);
        break;
    }
    case SFCVIT_CURVE_RASTER:
        full.resize(size_t(n) * n);
        for (int r = 0; r < n; r++)
            for (int c = 0; c < n; c++) full[size_t(r) * n + c] = {r, c};
        break;
    case SFCVIT_CURVE_SPIRAL:
        // Ring k: bottom row left->right, right column upwards, top row right->left, left column downwards;
        // the next ring starts one cell up-right of where this one ended.
        full.reserve(size_t(n) * n);
        for (int k = 0; 2 * k < n; k++) {
            const int lo = k, hi = n - 1 - k;
            for (int c = lo; c <= hi; c++) full.push_back({hi, c});
            for (int r = hi - 1; r >= lo; r--) full.push_back({r, hi});
            if (hi > lo) {
                for (int c = hi - 1; c >= lo; c--) full.push_back({lo, c});
                for (int r = lo + 1; r <= hi - 1; r++) full.push_back({r, lo});
            }
        }
        break;
    case SFCVIT_CURVE_HILBERT_T: {
        if (n & (n - 1)) return sfcvit::fail(SFCVIT_EINVAL, "transposed Hilbert table: n=%d must be a power of two", n);
        gen_hilbert(order_for(n, 2), full);
        for (Cell &c : full) std::swap(c.i, c.j);
        break;
    }
    default:
        return sfcvit::fail(SFCVIT_EINVAL, "curve table: unknown curve id %d", curve);
    }
    kept.clear();
    kept.reserve(size_t(n) * n);
    for (const Cell &c : full)
        if (c.i >= 0 && c.i < n && c.j >= 0 && c.j < n) kept.push_back(c);
    if (kept.size() != size_t(n) * n)
        return sfcvit::fail(SFCVIT_EINVAL, "curve table: internal error, %zu cells kept for n=%d", kept.size(), n);
    return SFCVIT_OK;
}

}  // namespace

extern "C" int sfcvit_curve_table(int curve, int n, int32_t *out_flat) {
    if (!out_flat) return sfcvit::fail(SFCVIT_EINVAL, "curve table: null output");
    std::vector<Cell> kept;
    if (int rc = build(curve, n, kept)) return rc;
    for (size_t t = 0; t < kept.size(); t++) out_flat[t] = kept[t].i * n + kept[t].j;
    return SFCVIT_OK;
}

extern "C" int sfcvit_curve_table_rc(int curve, int n, int64_t *out_rc) {
    if (!out_rc) return sfcvit::fail(SFCVIT_EINVAL, "curve table: null output");
    std::vector<Cell> kept;
    if (int rc = build(curve, n, kept)) return rc;
    for (size_t t = 0; t < kept.size(); t++) {
        out_rc[2 * t] = kept[t].i;
        out_rc[2 * t + 1] = kept[t].j;
    }
    return SFCVIT_OK;
}

extern "C" int sfcvit_pixel_table(const int32_t *flat, int img, int p, int g, int32_t *out_pix) {
    if (!flat || !out_pix) return sfcvit::fail(SFCVIT_EINVAL, "pixel table: null pointer");
    if (img <= 0 || p <= 0 || g <= 0 || img % p != 0)
        return sfcvit::fail(SFCVIT_EINVAL, "pixel table: img=%d must be a positive multiple of p=%d", img, p);
    const int grid = img / p;
    const int64_t cells = int64_t(grid) * grid;
    if (cells % g != 0)
        return sfcvit::fail(SFCVIT_EINVAL, "pixel table: %lld pre-patches not divisible by group %d", (long long)cells, g);
    const int64_t n_tok = cells / g;
    const int P = g * p * p;
    for (int64_t t = 0; t < n_tok; t++)
        for (int gi = 0; gi < g; gi++) {
            const int32_t s = flat[t * g + gi];
            if (s < 0 || s >= cells) return sfcvit::fail(SFCVIT_EINVAL, "pixel table: index %d out of range", s);
            const int r0 = (s / grid) * p, c0 = (s % grid) * p;
            for (int p1 = 0; p1 < p; p1++)
                for (int p2 = 0; p2 < p; p2++)
                    out_pix[t * P + gi * p * p + p1 * p + p2] = (r0 + p1) * img + (c0 + p2);
        }
    return SFCVIT_OK;
}
