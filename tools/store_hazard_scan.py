"""Scans the gfx950 code objects of libmvsnet_hip.so for the store-data hazard tools/store_hazard_probe.hip measures:
a VMEM store of more than 64 bits whose data registers are written by a VALU / matrix instruction fewer than NEED wait
states later (measured on MI355X: 2 for global_store / buffer_store with soffset 0, 1 with a register soffset; the compiler
inserts 1 and 0).      python tools/store_hazard_scan.py [objects...]      exit code 1 if an instance is found."""
import glob, os, re, subprocess, sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
STORE = re.compile(r"\b(global|buffer|flat|scratch)_store_dwordx([34])\s+(.*)")
REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    m = REG.fullmatch(tok.strip().rstrip(","))
    if not m:
        return set()
    if m.group(3) is not None:
        return {int(m.group(3))}
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


def written(line):
    """VGPRs a VALU / MFMA instruction writes (first operand); empty for everything else."""
    ins = line.split()
    if not ins or not (ins[0].startswith("v_")):
        return set()
    if ins[0].startswith(("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane", "v_nop")):
        return set()
    return regs(ins[1]) if len(ins) > 1 else set()


def scan(path):
    # the device code object sits in the .hip_fatbin section: llvm-objdump --offloading extracts it next to its input
    work = "/tmp/_hazard_%d" % os.getpid()
    os.makedirs(work, exist_ok=True)
    local = os.path.join(work, os.path.basename(path))
    subprocess.run(["cp", path, local], check=True)
    subprocess.run([OBJDUMP, "--offloading", local], capture_output=True, text=True, cwd=work)
    cos = [f for f in glob.glob(local + ".*") if "gfx950" in f]
    dis = "".join(subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", c], capture_output=True, text=True).stdout for c in cos)
    subprocess.run(["rm", "-rf", work])
    lines = [l.split("//")[0].strip() for l in dis.splitlines()]
    found, nstores, func = [], 0, "?"
    for i, l in enumerate(lines):
        if l.endswith(">:"):
            func = l.split("<")[-1][:-2]
        m = STORE.search(l)
        if not m:
            continue
        nstores += 1
        ops = [o.strip() for o in m.group(3).split(",")]
        kind = m.group(1)
        data = regs(ops[0]) if kind == "buffer" else regs(ops[1])
        need = 2
        if kind == "buffer":      # vdata, vaddr, srsrc, soffset
            so = ops[3].split()[0] if len(ops) > 3 else "0"
            need = 1 if so.startswith("s") or so in ("m0",) else 2
        ws, j = 0, i + 1
        while ws < need and j < len(lines):
            t = lines[j]
            j += 1
            if not t or t.endswith(":"):
                continue
            if written(t) & data:
                found.append((func, l, t, ws, need))
                break
            mm = re.match(r"s_nop\s+(\d+)", t)
            ws += int(mm.group(1)) + 1 if mm else 1
            if t.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                break
    return nstores, found


if __name__ == "__main__":
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(HERE, "mvsnet_amd", "csrc", "*.o")))
    total = 0
    for o in objs:
        n, f = scan(o)
        total += len(f)
        print("%-28s %5d wide stores, %d with their data overwritten too early" % (os.path.basename(o), n, len(f)))
        for func, st, wr, ws, need in f[:6]:
            print("    %s\n        %s\n        %s      <- after %d wait state(s), %d needed" % (func[:90], st, wr, ws, need))
    sys.exit(1 if total else 0)
