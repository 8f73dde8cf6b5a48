// Tile descriptors of the tiled patch-embed kernels: the HOST analysis of a pixel table (no device code in this file, so that
// `make asan` can build it with the CPU sanitizers: hostcheck/host_check.cpp).  Consumers: patch_embed_tiled.hip.
#include "common_host.h"

#include <algorithm>
#include <cstring>
#include <vector>

using namespace sfcvit;
using sfcvit::tile_desc::DESC_HDR;
using sfcvit::tile_desc::MAXCLS;

// HOST.  Analyses a pixel table (sfcvit_pixel_table output, host copy): if every token is a 16 x 16 pixel tile of the
// image (any visiting order, <= 8 distinct pixel orders inside a tile) or a strip of 256 consecutive pixels, writes the
// tile descriptor the tiled kernels take (layout below) and returns the number of int32 written; 0 = not tileable
// (use the generic path); < 0 = error.  desc: [0] mode (1 tile, 2 strip) [1] classes [2] k-tile step [3] segment step
// [4] N [5] 256 [6 .. 6+classes] first token (in `toks`) of each class, then N; [16 .. 16+N) token ids grouped by
// class; [16+N .. 16+2N) pixel offset of each token's origin; then classes x 256 curve positions of the tile's pixels
// in raster order, then classes x 256 inverse tables (curve position -> raster pixel).
extern "C" int sfcvit_tile_descriptors(const int32_t *pix, int N, int P, int img_w, int32_t *desc, int capacity) {
    if (!pix || !desc || N <= 0 || img_w <= 0) return -fail(SFCVIT_EINVAL, "tile_descriptors: bad argument");       // < 0: a count is expected on success
    if (P != 256 || (img_w & 7)) return 0;
    std::vector<int32_t> origin(N), cls(N);
    std::vector<std::vector<int32_t>> perms;
    bool strip = true;
    for (int n = 0; n < N && strip; n++)
        for (int k = 0; k < 256; k++)
            if (pix[size_t(n) * 256 + k] != n * 256 + k) { strip = false; break; }
    int mode = 0;
    if (strip) {
        mode = 2;
        std::vector<int32_t> id(256);
        for (int k = 0; k < 256; k++) id[k] = k;
        perms.push_back(id);
        for (int n = 0; n < N; n++) { origin[n] = n * 256; cls[n] = 0; }
    } else {
        mode = 1;
        for (int n = 0; n < N; n++) {
            const int32_t *pp = pix + size_t(n) * 256;
            int r0 = 1 << 30, c0 = 1 << 30;
            for (int k = 0; k < 256; k++) { r0 = std::min(r0, pp[k] / img_w); c0 = std::min(c0, pp[k] % img_w); }
            if (c0 & 7) return 0;                                         // 16-byte vector loads of the row segments
            std::vector<int32_t> pos(256, -1);
            for (int k = 0; k < 256; k++) {
                const int r = pp[k] / img_w - r0, c = pp[k] % img_w - c0;
                if (r >= 16 || c >= 16 || pos[r * 16 + c] >= 0) return 0;   // not a 16 x 16 tile
                pos[r * 16 + c] = k;
            }
            origin[n] = r0 * img_w + c0;
            int found = -1;
            for (size_t c = 0; c < perms.size(); c++)
                if (perms[c] == pos) { found = int(c); break; }
            if (found < 0) {
                if (perms.size() >= size_t(MAXCLS)) return 0;
                perms.push_back(pos);
                found = int(perms.size()) - 1;
            }
            cls[n] = found;
        }
    }
    const int ncls = int(perms.size());
    const int total = DESC_HDR + 2 * N + 2 * ncls * 256;
    if (capacity < total) return -fail(SFCVIT_EINVAL, "tile_descriptors: capacity %d < %d", capacity, total);
    std::memset(desc, 0, sizeof(int32_t) * DESC_HDR);
    desc[0] = mode; desc[1] = ncls;
    desc[2] = mode == 1 ? 4 * img_w : 64;          // k-tile = 4 tile rows of 16 pixels / 64 consecutive pixels
    desc[3] = mode == 1 ? img_w : 16;              // segment = one tile row / 16 consecutive pixels
    desc[4] = N; desc[5] = 256;
    int t = 0;
    for (int c = 0; c < ncls; c++) {
        desc[6 + c] = t;
        for (int n = 0; n < N; n++)
            if (cls[n] == c) desc[DESC_HDR + t++] = n;
    }
    desc[6 + ncls] = t;
    for (int n = 0; n < N; n++) desc[DESC_HDR + N + n] = origin[n];
    for (int c = 0; c < ncls; c++)
        for (int j = 0; j < 256; j++) {
            desc[DESC_HDR + 2 * N + c * 256 + j] = perms[c][j];                              // raster pixel j -> curve position
            desc[DESC_HDR + 2 * N + (ncls + c) * 256 + perms[c][j]] = j;                     // curve position -> raster pixel
        }
    return total;
}
