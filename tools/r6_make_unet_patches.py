"""Regenerates tools/r6_unet_p_trace.patch (device-side time line, -DMVS_PTRACE) and tools/r6_unet_p_diag.patch (timing-only
builds, -DDIAG_NOMFMA / NOLOAD / NOSTORE) from the CURRENT mvsnet_amd/csrc/unet2d_p.hip by textual substitution, so that the
patches keep applying when the kernel changes around them.   python tools/r6_make_unet_patches.py"""
import os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mvsnet_amd", "csrc", "unet2d_p.hip")
orig = open(SRC).read()


def sub(s, a, b):
    assert s.count(a) == 1, a[:60]
    return s.replace(a, b)


def emit(text, name):
    open(SRC, "w").write(text)
    try:
        d = subprocess.run(["git", "diff", "--", SRC], cwd=ROOT, capture_output=True, text=True).stdout
    finally:
        open(SRC, "w").write(orig)
    open(os.path.join(ROOT, "tools", name), "w").write(d)
    print(name, len(d.splitlines()), "lines")


# ---- trace
s = orig
s = sub(s, 'namespace {\n\nconstexpr int PTH = 8;', '''#ifdef MVS_PTRACE
// device-side time line (tools/r6_unet_p_trace.py): 32 int64 per workgroup, slot = blockIdx.x (no atomics: a shared slot counter
// serialises the workgroups' starts by ~10 ns each and shows up as a "ramp" that is not there)
__device__ long long* g_ptrace = nullptr;
__device__ int g_ptrace_cap = 0;
extern "C" int mvs_unet_ptrace(void* buf, int cap) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_ptrace), &buf, sizeof(buf));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_ptrace_cap), &cap, sizeof(cap));
    return (int)e;
}
#define PT(i) do { if (tr && tid == 0 && (i) < 31) tr[(i)] = wall_clock64(); } while (0)
#else
#define PT(i) do {} while (0)
#endif

namespace {

constexpr int PTH = 8;''')
s = sub(s, "    // This workgroup's tiles: a contiguous range.", '''#ifdef MVS_PTRACE
    long long* tr = nullptr;
    if (g_ptrace && tid == 0 && (int)blockIdx.x < g_ptrace_cap && blockIdx.y == 0) {
        tr = g_ptrace + 1 + (long long)blockIdx.x * 32;
        tr[31] = ((long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) << 32) | blockIdx.x;
        if (blockIdx.x == 0) g_ptrace[0] = gridDim.x;
    }
    PT(0);
#endif
    // This workgroup's tiles: a contiguous range.''')
s = sub(s, "    affine_end();                                    // (its barriers also publish the weights)",
        "    PT(28);                                          // loads requested, weights written\n    affine_end();                                    // (its barriers also publish the weights)\n    PT(29);")
s = sub(s, "    __syncthreads();\n    if (t0 + 1 < t1) fetch(t0 + 1, pinB, okB);", "    __syncthreads();\n    PT(1);\n    if (t0 + 1 < t1) fetch(t0 + 1, pinB, okB);")
s = sub(s, "        operands(0, 0);\n        if (tile + 2 < t1) fetch(tile + 2, pinF, okF);", "        PT(2 + 3 * (tile - t0));\n        operands(0, 0);\n        if (tile + 2 < t1) fetch(tile + 2, pinF, okF);")
s = sub(s, "        // this tile's results wait in registers", "        PT(3 + 3 * (tile - t0));\n        // this tile's results wait in registers")
s = sub(s, "        __syncthreads();                             // the other slab is complete, this one is free\n        float* t_ = slab_cur;",
        "        __syncthreads();                             // the other slab is complete, this one is free\n        PT(4 + 3 * (tile - t0));\n        float* t_ = slab_cur;")
s = sub(s, "    emit_prev();\n    if (sum_view >= 0 && p.stats) flush(sum_view);\n}", "    emit_prev();\n    if (sum_view >= 0 && p.stats) flush(sum_view);\n    PT(30);\n}")
emit(s, "r6_unet_p_trace.patch")

# ---- timing-only builds
s = orig
s = sub(s, "                        acc[m][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[s_ & 1][m][j], bq[s_ & 1][v][j], acc[m][v], 0, 0, 0);",
        "#ifdef DIAG_NOMFMA\n                        if (p.V < 0)\n#endif\n                        acc[m][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[s_ & 1][m][j], bq[s_ & 1][v][j], acc[m][v], 0, 0, 0);")
s = sub(s, "            pin[i] = *reinterpret_cast<const float4*>(base + off);",
        "#ifdef DIAG_NOLOAD\n            pin[i] = p.V < 0 ? *reinterpret_cast<const float4*>(base + off) : make_float4(1.f, 2.f, 3.f, 4.f);\n#else\n            pin[i] = *reinterpret_cast<const float4*>(base + off);\n#endif")
s = sub(s, "                                                       ry, ok ? (pix + co) * 4 : OOB, 0, 0);",
        "#ifdef DIAG_NOSTORE\n                                                       ry, ok && p.V < 0 ? (pix + co) * 4 : OOB, 0, 0);\n#else\n                                                       ry, ok ? (pix + co) * 4 : OOB, 0, 0);\n#endif")
emit(s, "r6_unet_p_diag.patch")
