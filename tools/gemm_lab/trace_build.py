#!/usr/bin/env python3
"""Builds a STAMPED copy of the product's persistent GEMM (csrc/gemm8p.hip, two-phase schedule) into tools/probe/lib_trace.so:
waves 0 and 4 (one per wave group) of workgroup TRACE_WG write s_memtime after every barrier of a k-tile -- start of
X mma, end of X mma, start of Y mma, end of Y mma (or of the epilogue) -- into LDS and, at the end, into a __device__
array that `sfcvit_lab_trace` copies out.  The product source stays free of lab paths: the stamps are inserted here, into
a temporary copy, at textual anchors of the two-phase k-tile.
    python tools/gemm_lab/trace_build.py && SFCVIT_LIB=tools/probe/lib_trace.so python tools/gemm_lab/trace_2phase.py 2304 768"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd", "csrc")
TRACE_WG = int(os.environ.get("TRACE_WG", "8"))

src = open(os.path.join(CSRC, "gemm8p.hip")).read()


def sub(old, new, count=1):
    global src
    assert src.count(old) >= 1, old
    src = src.replace(old, new, count)


sub("namespace p8 {\n", """namespace p8 {
__device__ unsigned long long p8_trace[2][256];
#define P8_STAMP()                                                                                                   \\
    do {                                                                                                             \\
        if ((tid & 255) == 0 && blockIdx.x == %d && tr_n < 256)                                                      \\
            reinterpret_cast<unsigned long long *>(smem + LDS_BIAS + 8192)[(tid >> 8) * 256 + tr_n] = __builtin_amdgcn_s_memtime(); \\
        tr_n++;                                                                                                      \\
    } while (0)
""" % TRACE_WG)
sub("    const uint16_t *A = static_cast<const uint16_t *>(g.a);\n    const uint16_t *B = static_cast<const uint16_t *>(g.b);\n    // Tile queue.",
    "    int tr_n = 0;\n    const uint16_t *A = static_cast<const uint16_t *>(g.a);\n    const uint16_t *B = static_cast<const uint16_t *>(g.b);\n    // Tile queue.")
# the two-phase k-tile: stamps after each of its four barriers
sub("""        wait_lgkm<0>();
        bar();
        mma0(fb0, 0);
        mma0(fb1, 1);
        if (first) set_next(t + 1);           // entry t + 1 was published an epilogue and several barriers ago
        bar();
""", """        wait_lgkm<0>();
        bar();
        P8_STAMP();
        mma0(fb0, 0);
        mma0(fb1, 1);
        if (first) set_next(t + 1);
        bar();
        P8_STAMP();
""")
sub("""        wait_lgkm<0>();
        bar();
        mma1(fb0, 0);
        mma1(fb1, 1);
        if (last) {
            if (wr == 0) {
                bar();
                draw();
            }
            epilogue();
            if (wr == 1) bar();
        } else {
            bar();
        }
    };""", """        wait_lgkm<0>();
        bar();
        P8_STAMP();
        mma1(fb0, 0);
        mma1(fb1, 1);
        if (last) {
            if (wr == 0) {
                bar();
                draw();
            }
            epilogue();
            if (wr == 1) bar();
        } else {
            bar();
        }
        P8_STAMP();
    };""")
sub("""    if (wr == 0) bar();
    wait_vm<0>();
    finish();
}""", """    if (wr == 0) bar();
    wait_vm<0>();
    if ((tid & 255) == 0 && blockIdx.x == %d)
        for (int i = 0; i < 256; i++)
            p8_trace[tid >> 8][i] = i < tr_n ? reinterpret_cast<unsigned long long *>(smem + LDS_BIAS + 8192)[(tid >> 8) * 256 + i] : 0ull;
    finish();
}""" % TRACE_WG)
if os.environ.get("TRACE_EPI"):
    # TRACE_EPI=1: four more stamps per epilogue of the traced waves -- start | arithmetic done | LDS exchange done | stores
    # issued -- into p8_trace_e[group][tile][4] (trace_epilogue.py prints them); a scheduling barrier pins each stamp
    sub("namespace p8 {\n", """namespace p8 {
__device__ unsigned long long p8_trace_e[2][60][4];
#define P8_ESTAMP(k)                                                                                                 \\
    do {                                                                                                             \\
        __builtin_amdgcn_sched_barrier(0);                                                                           \\
        if ((tid & 255) == 0 && blockIdx.x == %d && te_n < 60)                                                       \\
            reinterpret_cast<unsigned long long *>(smem + LDS_BIAS + 12288)[((tid >> 8) * 60 + te_n) * 4 + (k)] = __builtin_amdgcn_s_memtime(); \\
        __builtin_amdgcn_sched_barrier(0);                                                                           \\
    } while (0)
""" % TRACE_WG)
    sub("    int tr_n = 0;\n", "    int tr_n = 0, te_n = 0;\n")
    sub("        auto batch = [&](auto i0c, auto i1c) __attribute__((always_inline)) {\n            constexpr int i0 = decltype(i0c)::value, i1 = decltype(i1c)::value;\n",
        "        auto batch = [&](auto i0c, auto i1c) __attribute__((always_inline)) {\n            constexpr int i0 = decltype(i0c)::value, i1 = decltype(i1c)::value;\n            if (i0 == 0) P8_ESTAMP(0);\n")
    sub("            patch_exchange<i1 - i0>(pk, sm);\n", "            if (i0 == 0) P8_ESTAMP(1);\n            patch_exchange<i1 - i0>(pk, sm);\n            if (i0 == 0) P8_ESTAMP(2);\n")
    sub("                store_row_pair(g, m0 + 16 * i, n0, sm, pk[i - i0][0], pk[i - i0][1]);\n            });\n        };",
        "                store_row_pair(g, m0 + 16 * i, n0, sm, pk[i - i0][0], pk[i - i0][1]);\n            });\n            if (i0 == 0) { P8_ESTAMP(3); te_n++; }\n        };")
    sub("            p8_trace[tid >> 8][i] = i < tr_n ? reinterpret_cast<unsigned long long *>(smem + LDS_BIAS + 8192)[(tid >> 8) * 256 + i] : 0ull;",
        "            p8_trace[tid >> 8][i] = i < tr_n ? reinterpret_cast<unsigned long long *>(smem + LDS_BIAS + 8192)[(tid >> 8) * 256 + i] : 0ull;\n"
        "    if ((tid & 255) == 0 && blockIdx.x == %d)\n        for (int i = 0; i < 240; i++)\n"
        "            (&p8_trace_e[tid >> 8][0][0])[i] = i < 4 * te_n ? reinterpret_cast<unsigned long long *>(smem + LDS_BIAS + 12288)[(tid >> 8) * 240 + i] : 0ull;" % TRACE_WG)
sub("    const int LDS_TOTAL = LDS_BIAS + (a.bias ? a.N * 2 : 0);", "    const int LDS_TOTAL = LDS_MAX;")
src += """
extern "C" int sfcvit_lab_trace(unsigned long long *host) {
    return int(hipMemcpyFromSymbol(host, HIP_SYMBOL(sfcvit::p8::p8_trace), sizeof(unsigned long long) * 512));
}
"""
if os.environ.get("TRACE_EPI"):
    src += """
extern "C" int sfcvit_lab_trace_epilogue(unsigned long long *host) {
    return int(hipMemcpyFromSymbol(host, HIP_SYMBOL(sfcvit::p8::p8_trace_e), sizeof(unsigned long long) * 480));
}
"""
tmp = os.path.join(CSRC, "build", "gemm8p_trace.hip")
os.makedirs(os.path.dirname(tmp), exist_ok=True)
open(tmp, "w").write(src)
flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -mllvm -amdgpu-atomic-optimizer-strategy=None".split()
obj = os.path.join(CSRC, "build", "gemm8p_trace.o")
subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-I" + CSRC, "-c", tmp, "-o", obj])
others = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build")))
          if f.endswith(".o") and f not in ("gemm8p.o", "gemm8p_trace.o")]
out = os.path.join(ROOT, "tools", "probe", "lib_trace.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj, *others])
print("built", out)
