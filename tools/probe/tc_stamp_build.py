"""Builds tools/probe/libvslam_hip_tc.so: the library with time stamps in the chained triangular solve (a scratch
copy of the sources under /tmp is edited, the tree is left alone).  Read the stamps with tools/probe/tc_stamps.py."""
import os
import shutil
import subprocess

root = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
dst = "/tmp/tc"
os.makedirs(dst + "/visualslam_amd", exist_ok=True)
shutil.rmtree(dst + "/visualslam_amd/csrc", ignore_errors=True)
shutil.rmtree(dst + "/include", ignore_errors=True)
shutil.copytree(root + "/visualslam_amd/csrc", dst + "/visualslam_amd/csrc")
shutil.copytree(root + "/include", dst + "/include")
p = dst + "/visualslam_amd/csrc/tri.hip"
s = open(p).read()


def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b, 1)


rep("        if (ctx->chain_used + need > VSLAM_CHAIN_SOL_DOUBLES) {", "        if (true) {   // stamps live beyond the call's own stretch: every call starts the ring afresh")
rep("    double nacc = 0.0; ", "    long long *stamps = reinterpret_cast<long long *>(sol) + 16384;\n    double nacc = 0.0; ")
rep("        for (int i = 0; i < g.nblk; ++i) {\n            const int par = i & 1;", "        for (int i = 0; i < g.nblk; ++i) {\n            if (lane == 0) stamps[i * 16 + 0] = wall_clock64();\n            const int par = i & 1;")
rep("            const int c = g.at(i, lane);\n            const double x = (c >= 0) ? v * rinvs", "            if (lane == 0 && v != 1234.5678) stamps[i * 16 + 1] = wall_clock64();\n            const int c = g.at(i, lane);\n            const double x = (c >= 0) ? v * rinvs")
rep("            const double acc = do_terms(i + 1);",
    "            if (tid == 0) stamps[i * 16 + 2] = wall_clock64();\n            const double acc = do_terms(i + 1);\n            if (tid == 0 && acc != 1234.5678) stamps[i * 16 + 3] = stamps[i * 16 + 4] = wall_clock64();")
rep("            finish_terms(i + 1, acc, h, zv);\n        }\n        lds_barrier();\n    }\n}",
    "            finish_terms(i + 1, acc, h, zv);\n            if (tid == 0) stamps[i * 16 + 5] = wall_clock64();\n            if (lane == 0) stamps[i * 16 + 6 + wave] = wall_clock64();\n        }\n        lds_barrier();\n    }\n}")
rep("    VSLAM_TRY(d2h(ctx, z, dz, (size_t)n * 8));\n    return stream_finish(ctx);",
    "    VSLAM_TRY(d2h(ctx, z, dz, (size_t)n * 8));\n    if (getenv(\"TC_STAMPS\")) { static long long st[16 * 240]; VSLAM_TRY(d2h(ctx, st, ctx->chain_sol + 16384, sizeof(st))); VSLAM_TRY(stream_finish(ctx)); FILE *f = fopen(getenv(\"TC_STAMPS\"), \"w\"); for (int i = 0; i < 240; ++i) { for (int q = 0; q < 14; ++q) fprintf(f, \"%lld \", st[i * 16 + q]); fprintf(f, \"\\n\"); } fclose(f); }\n    return stream_finish(ctx);")
s = s.replace('#include "linalg.hpp"', '#include "linalg.hpp"\n#include <cstdio>\n#include <cstdlib>', 1)
open(p, "w").write(s)
subprocess.check_call(["make", "-C", dst + "/visualslam_amd/csrc", "OUT=" + root + "/tools/probe/libvslam_hip_tc.so"], stdout=subprocess.DEVNULL)
print("built")
