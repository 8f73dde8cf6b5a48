/*
 * oracle/curves_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, float64 recursion exactly as the reference does it)
 * of the reference's space-filling-curve generators and of embed_and_prune_sfc.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this file.  The product generates its tables with closed-form
 * integer code (csrc/curves.cpp); this file is the checker for it.
 *
 * Pinned against: tests/golden/curves_small.npz and tests/golden/curves_sha.json,
 * both produced by tools/make_golden.py from the imported reference.
 *
 * Reference citations (relative to /root/reference):
 *   hilbert_curve        src/curves/space_filling_curves.py:168-202
 *   z_curve              src/curves/space_filling_curves.py:134-165
 *   moore_curve          src/curves/space_filling_curves.py:205-251
 *   peano_curve          src/curves/space_filling_curves.py:74-131
 *   onion_curve          src/curves/space_filling_curves.py:9-71
 *   grid_size            src/curves/space_filling_curves.py:458-468
 *   embed_and_prune_sfc  src/curves/space_filling_curves.py:471-491
 *   spiral walk          src/tokenizers/_1D/onion_embedding1D.py:35-53
 *                        (= src/tokenizers/multiscale/multi_onion.py:68-87)
 *   _2D Hilbert indices  src/tokenizers/_2D/hilbert_embedding.py:30-78
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    double *x, *y;
    int64_t n, cap;
} pts_t;

static void pts_push(pts_t *p, double x, double y) {
    if (p->n == p->cap) {
        p->cap = p->cap ? p->cap * 2 : 1024;
        p->x = (double *)realloc(p->x, sizeof(double) * (size_t)p->cap);
        p->y = (double *)realloc(p->y, sizeof(double) * (size_t)p->cap);
    }
    p->x[p->n] = x;
    p->y[p->n] = y;
    p->n++;
}

static void pts_free(pts_t *p) {
    free(p->x);
    free(p->y);
    memset(p, 0, sizeof(*p));
}

/* fin = M @ [x, y] for a 2x2 matrix M, as np.dot(fin, [x_i, y_i]) does. */
static void pts_transform(pts_t *p, const double m[4]) {
    for (int64_t i = 0; i < p->n; i++) {
        double x = p->x[i], y = p->y[i];
        p->x[i] = m[0] * x + m[1] * y;
        p->y[i] = m[2] * x + m[3] * y;
    }
}

/* mirror @ rotation(deg), both 2x2 row-major (space_filling_curves.py:196-201). */
static void mirror_rot(double mxx, double myy, double deg, double out[4]) {
    double c = cos(deg), s = sin(deg);
    /* rotation = [[c, -s], [s, c]]; mirror = diag(mxx, myy) */
    out[0] = mxx * c;
    out[1] = mxx * -s;
    out[2] = myy * s;
    out[3] = myy * c;
}

/* space_filling_curves.py:182-193 */
static void hilbert_rec(pts_t *p, double x0, double y0, double xi, double xj,
                        double yi, double yj, int n) {
    if (n <= 0) {
        pts_push(p, x0 + (xi + yi) / 2, y0 + (xj + yj) / 2);
        return;
    }
    hilbert_rec(p, x0, y0, yi / 2, yj / 2, xi / 2, xj / 2, n - 1);
    hilbert_rec(p, x0 + xi / 2, y0 + xj / 2, xi / 2, xj / 2, yi / 2, yj / 2, n - 1);
    hilbert_rec(p, x0 + xi / 2 + yi / 2, y0 + xj / 2 + yj / 2, xi / 2, xj / 2,
                yi / 2, yj / 2, n - 1);
    hilbert_rec(p, x0 + xi / 2 + yi, y0 + xj / 2 + yj, -yi / 2, -yj / 2, -xi / 2,
                -xj / 2, n - 1);
}

static void hilbert_curve(pts_t *p, int order, double size) {
    hilbert_rec(p, 0, 0, size, 0, 0, size, order);
    double m[4];
    mirror_rot(-1, 1, M_PI / 2, m); /* :196-201 */
    pts_transform(p, m);
}

/* space_filling_curves.py:148-156 */
static void z_rec(pts_t *p, double x0, double y0, double w, int n) {
    if (n == 0) {
        pts_push(p, x0 + w / 2, y0 + w / 2);
        return;
    }
    double half = w / 2;
    z_rec(p, x0 + half, y0, half, n - 1);
    z_rec(p, x0, y0, half, n - 1);
    z_rec(p, x0 + half, y0 + half, half, n - 1);
    z_rec(p, x0, y0 + half, half, n - 1);
}

static void z_curve(pts_t *p, int order, double size) {
    z_rec(p, 0, 0, size, order);
    double m[4];
    mirror_rot(-1, -1, M_PI, m); /* :159-164 */
    pts_transform(p, m);
}

/* space_filling_curves.py:233-251 */
static void moore_curve(pts_t *p, int order, double size) {
    double x0 = 0, y0 = 0, xi = size, xj = 0, yi = 0, yj = size;
    int n = order;
    if (n <= 0) {
        pts_push(p, x0 + (xi + yi) / 2, y0 + (xj + yj) / 2);
    } else {
        hilbert_rec(p, x0 + xi / 2, y0 + xj / 2, -xi / 2, xj / 2, yi / 2, yj / 2, n - 1);
        hilbert_rec(p, x0 + xi / 2 + yi / 2, y0 + xj / 2 + yj / 2, -xi / 2, xj / 2,
                    yi / 2, yj / 2, n - 1);
        hilbert_rec(p, x0 + xi / 2 + yi, y0 + xj / 2 + yj, xi / 2, xj / 2, yi / 2,
                    -yj / 2, n - 1);
        hilbert_rec(p, x0 + xi / 2 + yi / 2, y0 + xj / 2 + yj / 2, xi / 2, xj / 2,
                    yi / 2, -yj / 2, n - 1);
    }
    double c = cos(M_PI * 2), s = sin(M_PI * 2);
    double m[4] = {c, -s, s, c}; /* :248-251, rotation only */
    pts_transform(p, m);
}

/* space_filling_curves.py:95-108: (dx, dy, next_pattern) per step. */
static const int PEANO[4][9][3] = {
    {{0, 0, 0}, {1, 0, 1}, {2, 0, 0}, {2, 1, 1}, {1, 1, 0}, {0, 1, 1}, {0, 2, 0}, {1, 2, 1}, {2, 2, 0}},
    {{2, 0, 1}, {1, 0, 0}, {0, 0, 1}, {0, 1, 0}, {1, 1, 1}, {2, 1, 0}, {2, 2, 1}, {1, 2, 0}, {0, 2, 1}},
    {{0, 2, 2}, {1, 2, 3}, {2, 2, 2}, {2, 1, 3}, {1, 1, 2}, {0, 1, 3}, {0, 0, 2}, {1, 0, 3}, {2, 0, 2}},
    {{2, 2, 3}, {1, 2, 2}, {0, 2, 3}, {0, 1, 2}, {1, 1, 3}, {2, 1, 2}, {2, 0, 3}, {1, 0, 2}, {0, 0, 3}},
};

/* space_filling_curves.py:86-123.  Returns the sub-curve appended to p; the
 * reference reverses the sub-list of every middle-column step (idx % 3 == 1). */
static void peano_rec(pts_t *p, double x, double y, double size, int order, int pattern) {
    if (order == 0) {
        pts_push(p, x + size / 2, y + size / 2);
        return;
    }
    size /= 3;
    for (int idx = 0; idx < 9; idx++) {
        int dx = PEANO[pattern][idx][0], dy = PEANO[pattern][idx][1];
        int64_t start = p->n;
        peano_rec(p, x + dx * size, y + dy * size, size, order - 1, PEANO[pattern][idx][2]);
        if (idx % 3 == 1) {
            for (int64_t a = start, b = p->n - 1; a < b; a++, b--) {
                double tx = p->x[a], ty = p->y[a];
                p->x[a] = p->x[b];
                p->y[a] = p->y[b];
                p->x[b] = tx;
                p->y[b] = ty;
            }
        }
    }
}

static void peano_curve(pts_t *p, int order, double size) {
    peano_rec(p, 0, 0, size, order, 0);
    double m[4];
    mirror_rot(-1, 1, M_PI / 2, m); /* :125-130 */
    pts_transform(p, m);
}

/* space_filling_curves.py:23-55: ring of a j x j grid, then recurse on j-2. */
static void onion_rec(pts_t *p, int j, int ox, int oy) {
    if (j == 2) {
        pts_push(p, ox, oy);
        pts_push(p, ox + 1, oy);
        pts_push(p, ox + 1, oy + 1);
        pts_push(p, ox, oy + 1);
        return;
    }
    for (int x = 0; x < j; x++) pts_push(p, ox + x, oy);
    for (int y = 1; y < j; y++) pts_push(p, ox + j - 1, oy + y);
    for (int x = j - 2; x >= 0; x--) pts_push(p, ox + x, oy + j - 1);
    for (int y = j - 2; y > 0; y--) pts_push(p, ox, oy + y);
    if (j > 2) onion_rec(p, j - 2, ox + 1, oy + 1);
}

static void onion_curve(pts_t *p, int order, double size) {
    order *= 2; /* :21 */
    onion_rec(p, order, 0, 0);
    double cell = size / order;
    for (int64_t i = 0; i < p->n; i++) {
        p->x[i] = p->x[i] * cell + cell / 2;
        p->y[i] = p->y[i] * cell + cell / 2;
    }
    double c = cos(0.0), s = sin(0.0);
    double m[4] = {c, -s, s, c};
    pts_transform(p, m);
}

enum { SFC_HILBERT = 0, SFC_Z = 1, SFC_MOORE = 2, SFC_PEANO = 3, SFC_ONION = 4 };

/* space_filling_curves.py:458-468 */
static int64_t grid_size(int order, int kind) {
    switch (kind) {
    case SFC_HILBERT:
    case SFC_Z:
    case SFC_MOORE:
        return (int64_t)1 << order;
    case SFC_PEANO: {
        int64_t r = 1;
        for (int i = 0; i < order; i++) r *= 3;
        return r;
    }
    case SFC_ONION:
        return order + (order % 2);
    }
    return -1;
}

/*
 * space_filling_curves.py:471-491.  Writes (i, j) pairs (int64, interleaved)
 * of every kept point into out_ij (capacity width*height pairs); returns the
 * number of pairs, or -1 on a bad kind / overflow of the output.
 */
int64_t oracle_embed_and_prune_sfc(int kind, int width, int height, int64_t *out_ij) {
    if (kind < 0 || kind > SFC_ONION) return -1;
    int order = 0;
    int wh = width > height ? width : height;
    while (grid_size(order, kind) < wh) order++;
    int64_t P = grid_size(order, kind);
    pts_t p;
    memset(&p, 0, sizeof(p));
    switch (kind) {
    case SFC_HILBERT: hilbert_curve(&p, order, (double)P); break;
    case SFC_Z: z_curve(&p, order, (double)P); break;
    case SFC_MOORE: moore_curve(&p, order, (double)P); break;
    case SFC_PEANO: peano_curve(&p, order, (double)P); break;
    case SFC_ONION:
        /* onion_curve(order) doubles its argument (:21) while grid_size says
         * order + order%2: the reference therefore generates a 2*order grid
         * scaled into [0, P] -- restated as is. */
        if (order == 0) { pts_free(&p); return 0; }
        onion_curve(&p, order, (double)P);
        break;
    }
    int64_t cnt = 0, cap = (int64_t)width * height;
    for (int64_t t = 0; t < p.n; t++) {
        int64_t i = (int64_t)floor(p.x[t]), j = (int64_t)floor(p.y[t]);
        if (0 <= i && i < width && 0 <= j && j < height) {
            if (cnt >= cap) { /* the reference would just keep appending */
                cnt++;
                continue;
            }
            out_ij[2 * cnt] = i;
            out_ij[2 * cnt + 1] = j;
            cnt++;
        }
    }
    pts_free(&p);
    return cnt;
}

/*
 * onion_embedding1D.py:35-53: walk from the bottom-left cell, directions right, up, left, down, turning when
 * the next cell is outside or already visited.  Writes the flat indices r*n+c; returns n*n.
 */
int64_t oracle_spiral_flat(int n, int64_t *out) {
    static const int di[4] = {0, -1, 0, 1}, dj[4] = {1, 0, -1, 0};
    char *seen = (char *)calloc((size_t)n * n, 1);
    int i = n - 1, j = 0, d = 0;
    for (int64_t t = 0; t < (int64_t)n * n; t++) {
        out[t] = (int64_t)i * n + j;
        seen[(size_t)i * n + j] = 1;
        int ni = i + di[d], nj = j + dj[d];
        if (0 <= ni && ni < n && 0 <= nj && nj < n && !seen[(size_t)ni * n + nj]) {
            i = ni;
            j = nj;
        } else {
            d = (d + 1) % 4;
            i += di[d];
            j += dj[d];
        }
    }
    free(seen);
    return (int64_t)n * n;
}

/*
 * _2D/hilbert_embedding.py:40-45: order = int(log2(grid)); points of the raw recursion on the unit square
 * (no post-transform); (int(x*grid), int(y*grid)) -> i*grid + j.  Returns the number of indices (4^order).
 */
int64_t oracle_hilbert2d_flat(int grid, int64_t *out) {
    int order = (int)log2((double)grid);
    pts_t p;
    memset(&p, 0, sizeof(p));
    hilbert_rec(&p, 0, 0, 1.0, 0, 0, 1.0, order);
    for (int64_t t = 0; t < p.n; t++)
        out[t] = (int64_t)(p.x[t] * grid) * grid + (int64_t)(p.y[t] * grid);
    int64_t n = p.n;
    pts_free(&p);
    return n;
}
