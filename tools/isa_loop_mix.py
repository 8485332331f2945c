"""Instruction mix of the tile loop of the fused ConvGRU kernels from hipcc --save-temps assembly:
    cd /tmp && hipcc --offload-arch=gfx950 -O3 -std=c++17 --save-temps -c <repo>/mvsnet_amd/csrc/gru_fused.hip -o /tmp/gf.o
    python tools/isa_loop_mix.py /tmp/gru_fused-hip-amdgcn-amd-amdhsa-gfx950.s
The loop = from the barrier before the first 16x16x4 matrix instruction to the last barrier of the function."""
import collections, re, sys
L = open(sys.argv[1]).read().split("\n")
names = [i for i, l in enumerate(L) if re.match(r"^_Z\S*:", l)]
ends = [i for i, l in enumerate(L) if l.startswith(".Lfunc_end")]
for a, b in zip(names, ends):
    if "gru_fused_kernel" not in L[a]:
        continue
    body = L[a:b]
    bars = [i for i, l in enumerate(body) if "s_barrier" in l]
    mf = [i for i, l in enumerate(body) if "v_mfma_f32_16x16x4" in l]
    start = max(x for x in bars if x < mf[0])
    end = max(x for x in bars if x > mf[-1]) if any(x > mf[-1] for x in bars) else len(body)
    # the last barrier after the loop belongs to the sums' reduction: take the first barrier after the last small-job mfma
    m4 = [i for i, l in enumerate(body) if "v_mfma_f32_4x4x1" in l]
    end = min(x for x in bars if x > max(mf[-1], m4[-1]))
    cnt = collections.Counter()
    for l in body[start:end]:
        t = l.split()
        if not t or t[0].startswith((";", ".")):
            continue
        op = t[0]
        if op.startswith("v_mfma"):
            cnt["mfma16" if "16x16" in op else "mfma4"] += 1
        elif op.startswith(("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log")):
            cnt["trans"] += 1
        elif op.startswith("v_"):
            cnt["valu:" + re.sub(r"_e32|_e64|_dpp|_sdwa", "", op)] += 1
        elif op.startswith("ds_"):
            cnt["lds:" + op] += 1
        elif op.startswith(("buffer_", "global_")):
            cnt["vmem:" + op] += 1
        elif op.startswith("s_"):
            cnt["salu"] += 1
    tot = sum(v for k, v in cnt.items() if k.startswith("valu:"))
    print(L[a][:64], "| loop lines", end - start, "| plain VALU", tot, "transcendental", cnt["trans"], "mfma 16x16x4", cnt["mfma16"],
          "mfma 4x4x1 (all jobs)", cnt["mfma4"], "SALU", cnt["salu"])
    print("    VALU:", ", ".join("%s %d" % (k[5:], v) for k, v in cnt.most_common(60) if k.startswith("valu:")))
    print("    LDS / VMEM:", ", ".join("%s %d" % (k.split(":")[1], v) for k, v in cnt.most_common(80) if k.startswith(("lds", "vmem"))))
