// Sanitizer driver for the HOST-ONLY code of libsfcvit_hip.so ("Race detection / sanitizers": GPU ASan is not
// available on this pool, so the native host code gets a CPU-side -fsanitize=address,undefined build of its own):
//   sfcvit_curve_table / _rc (curves.cpp), sfcvit_pixel_table (curves.cpp), sfcvit_tile_descriptors (patch_embed_tiled.hip,
//   host part), the error path of common.cpp.  Every output buffer is a heap block of EXACTLY the documented size, so that
// an off-by-one in a generator or in the descriptor writer is a heap-buffer-overflow report instead of silent corruption.
// Built and run by `make asan` (tests/test_host_cpu.py::test_host_code_is_clean_under_address_sanitizer).  No GPU call.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "../../../include/sfcvit.h"

static int g_fail = 0;
#define CHECK(cond, ...) do { if (!(cond)) { std::fprintf(stderr, "host_check: " __VA_ARGS__); std::fprintf(stderr, " (%s:%d)\n", __FILE__, __LINE__); g_fail++; } } while (0)

static bool is_permutation(const int32_t *t, int n2) {
    std::vector<char> seen(n2, 0);
    for (int i = 0; i < n2; i++) {
        if (t[i] < 0 || t[i] >= n2 || seen[t[i]]) return false;
        seen[t[i]] = 1;
    }
    return true;
}

int main() {
    const int curves[] = {SFCVIT_CURVE_HILBERT, SFCVIT_CURVE_Z, SFCVIT_CURVE_MOORE, SFCVIT_CURVE_PEANO, SFCVIT_CURVE_RASTER,
                          SFCVIT_CURVE_SPIRAL, SFCVIT_CURVE_HILBERT_T};
    const int sizes[] = {1, 2, 3, 5, 7, 8, 9, 14, 16, 24, 27, 28, 32, 33, 81, 100, 224};
    for (int c : curves)
        for (int n : sizes) {
            const int n2 = n * n;
            std::unique_ptr<int32_t[]> flat(new int32_t[n2]);
            std::unique_ptr<int64_t[]> rc(new int64_t[2 * size_t(n2)]);
            const int r1 = sfcvit_curve_table(c, n, flat.get());
            if (c == SFCVIT_CURVE_HILBERT_T && (n & (n - 1))) {      // the _2D tokenizer's private generator: powers of two only
                CHECK(r1 != SFCVIT_OK || is_permutation(flat.get(), n2), "curve %d n %d: accepted but not a permutation", c, n);
                continue;
            }
            CHECK(r1 == SFCVIT_OK, "curve %d n %d: %s", c, n, sfcvit_last_error());
            if (r1 != SFCVIT_OK) continue;
            CHECK(is_permutation(flat.get(), n2), "curve %d n %d: not a permutation of the grid", c, n);
            CHECK(sfcvit_curve_table_rc(c, n, rc.get()) == SFCVIT_OK, "curve %d n %d (rc): %s", c, n, sfcvit_last_error());
            for (int t = 0; t < n2; t++)
                if (rc[2 * size_t(t)] * n + rc[2 * size_t(t) + 1] != flat[t]) { CHECK(false, "curve %d n %d: rc / flat disagree at %d", c, n, t); break; }
        }
    // argument errors come back as codes with a message, never as a crash
    CHECK(sfcvit_curve_table(99, 4, nullptr) == SFCVIT_EINVAL, "null output accepted");
    {
        int32_t one[16];
        CHECK(sfcvit_curve_table(99, 4, one) == SFCVIT_EINVAL && std::strlen(sfcvit_last_error()) > 0, "unknown curve accepted");
        CHECK(sfcvit_curve_table(SFCVIT_CURVE_HILBERT, 0, one) != SFCVIT_OK, "n = 0 accepted");
        CHECK(sfcvit_curve_table(SFCVIT_CURVE_HILBERT, -3, one) != SFCVIT_OK, "n < 0 accepted");
    }
    // pixel tables: (img, p, g) of the _1D tokenizers (p = 1, g = pixels per token), ViT/16 patches (p = 16, g = 1), the
    // hierarchical levels of the reference's main.py (32 px: p = 1 g = 16, p = 2 g = 4 ... ) and a grouped case
    struct PT { int img, p, g, curve; };
    const PT pts[] = {{32, 1, 256, SFCVIT_CURVE_HILBERT}, {32, 16, 1, SFCVIT_CURVE_Z}, {32, 2, 16, SFCVIT_CURVE_Z}, {32, 4, 4, SFCVIT_CURVE_MOORE},
                      {224, 1, 256, SFCVIT_CURVE_HILBERT}, {224, 16, 1, SFCVIT_CURVE_HILBERT}, {224, 1, 256, SFCVIT_CURVE_RASTER},
                      {27, 3, 9, SFCVIT_CURVE_PEANO}, {48, 1, 256, SFCVIT_CURVE_SPIRAL}};
    for (const PT &q : pts) {
        const int grid = q.img / q.p, cells = grid * grid, P = q.g * q.p * q.p, N = cells / q.g;
        std::unique_ptr<int32_t[]> flat(new int32_t[cells]);
        CHECK(sfcvit_curve_table(q.curve, grid, flat.get()) == SFCVIT_OK, "pixel table: curve table failed");
        std::unique_ptr<int32_t[]> pix(new int32_t[size_t(N) * P]);
        const int rc = sfcvit_pixel_table(flat.get(), q.img, q.p, q.g, pix.get());
        CHECK(rc == SFCVIT_OK, "pixel table img %d p %d g %d: %s", q.img, q.p, q.g, sfcvit_last_error());
        if (rc != SFCVIT_OK) continue;
        CHECK(is_permutation(pix.get(), q.img * q.img), "pixel table img %d p %d g %d: not a permutation of the pixels", q.img, q.p, q.g);
        if (P == 256 && q.img % 8 == 0) {                          // what the tokenizers hand to sfcvit_tile_descriptors
            const int cap_needed = 16 + 2 * N + 2 * 8 * 256;       // DESC_HDR + 2 N + 2 MAXCLS 256
            std::unique_ptr<int32_t[]> probe(new int32_t[cap_needed]);
            const int total = sfcvit_tile_descriptors(pix.get(), N, P, q.img, probe.get(), cap_needed);
            CHECK(total >= 0, "tile descriptors img %d: %s", q.img, sfcvit_last_error());
            if (total > 0) {
                std::unique_ptr<int32_t[]> exact(new int32_t[total]);   // exactly what it said it writes
                CHECK(sfcvit_tile_descriptors(pix.get(), N, P, q.img, exact.get(), total) == total, "tile descriptors: exact capacity refused");
                CHECK(sfcvit_tile_descriptors(pix.get(), N, P, q.img, exact.get(), total - 1) < 0, "tile descriptors: short capacity accepted");
                CHECK(exact[4] == N && exact[5] == 256 && exact[1] >= 1 && exact[1] <= 8, "tile descriptors: header");
            }
        }
    }
    {
        int32_t flat[4] = {0, 1, 2, 7}, out[4 * 4 * 4];
        CHECK(sfcvit_pixel_table(flat, 4, 2, 1, out) == SFCVIT_EINVAL, "pixel table: out-of-range index accepted");
        CHECK(sfcvit_pixel_table(flat, 5, 2, 1, out) == SFCVIT_EINVAL, "pixel table: img not a multiple of p accepted");
        CHECK(sfcvit_pixel_table(flat, 4, 2, 3, out) == SFCVIT_EINVAL, "pixel table: group that does not divide accepted");
        CHECK(sfcvit_pixel_table(nullptr, 4, 2, 1, out) == SFCVIT_EINVAL, "pixel table: null accepted");
        CHECK(sfcvit_tile_descriptors(nullptr, 4, 256, 32, out, 64) < 0, "tile descriptors: null accepted");
    }
    CHECK(sfcvit_abi_version() == SFCVIT_ABI_VERSION, "abi version");
    if (g_fail) { std::fprintf(stderr, "host_check: %d check(s) failed\n", g_fail); return 1; }
    std::printf("host_check ok\n");
    return 0;
}
