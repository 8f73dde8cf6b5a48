// Host-side helpers shared by every translation unit of libsfcvit_hip.so.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/sfcvit.h"

namespace sfcvit {

namespace tile_desc {          // sfcvit_tile_descriptors (tile_descriptors.cpp) <-> patch_embed_tiled.hip
constexpr int MAXCLS = 8, DESC_HDR = 16;
}

// Records the message for sfcvit_last_error() and returns `code`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();
// hipGetLastError() -> SFCVIT_ELAUNCH with the HIP message, or SFCVIT_OK.
int check_launch(const char *what);
// Compute units of the current device (0 if unknown): the grid of the persistent kernels.
int device_cu_count();
// gemm8p.hip: 16 zero-initialised device words per (device, stream), shared by the persistent kernels (each leaves its words
// zero); allocated at the first call on a device (not inside a hipGraph capture: warm up first).  nullptr if unavailable.
unsigned *stream_counters(void *stream);
// Opt a kernel into `bytes` of dynamic LDS on the current device (once per kernel and device; 0 or an error code).
int raise_lds_limit(const void *kernel, int bytes, const char *what);

// Which GEMM kernel the calling thread's last sfcvit_gemm launched (sfcvit_last_gemm_kernel formats it as the symbol
// rocprofv3 shows): family 1 gemm8p_kernel<a, b, c>, 2 gemm8p_km_kernel<a>, 3 gemm256_kernel<a, b, c, d>, 4 gemm_kernel<a, b, c>.
void note_gemm_kernel(int family, int a = 0, int b = 0, int c = 0, int d = 0);
// Which attention kernel the calling thread's last sfcvit_attention_fwd / _bwd launched (its main kernel, named as
// rocprofv3 names it); fmt is printf-style.
void note_attn_kernel(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
// Which row-wise kernel the calling thread's last sfcvit_layernorm_bwd* launched (sfcvit_last_rowwise_kernel).
void note_rowwise_kernel(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
bool gemm_fused_colsum();     // the kernel noted last was the 8-phase kernel with column sums in its epilogue
bool gemm_fused_actmask();    // ... with the activation bit mask written (act) or read (dact) by its epilogue

// rowwise.hip: out[n] = sum over `nparts` rows of part[nparts][N] in a fixed order; out fp32 or bf16.
int launch_colsum_reduce(const float *part, int nparts, int N, void *out, int out_bf16, void *stream);
// The general form: out[c] = sum_p part[p * ld + c], c < ncols.  Launched now -- or, between sfcvit_reduce_defer(1) and
// sfcvit_reduce_defer(0), queued for ONE batched launch at sfcvit_reduce_flush (every such reduction of a backward pass
// produces a parameter gradient nothing reads before the pass ends: 60 launches of 5 us per ViT-B step become one).
int reduce_cols(const float *part, int nparts, int ld, int ncols, void *out, int out_bf16, void *stream);
bool reduce_deferring();

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace sfcvit
