"""Scans the gfx950 code objects of libmvsnet_hip.so for the store-data hazard tools/store_hazard_probe.hip measures:
a VMEM store of more than 64 bits whose data registers are written by a VALU / matrix instruction fewer than NEED wait
states later (measured on MI355X: 2 for global_store / buffer_store with soffset 0, 1 with a register soffset; the compiler
inserts 1 and 0).  This scan, not the compiler, is what guarantees the wait states.

The walk follows the control flow for the NEED wait states after each store: the fall-through of a conditional branch AND its
target (a loop back-edge: last store of an iteration -> first vector instruction of the next), the target of s_branch; it stops
at s_endpgm / s_setpc.  Data registers in VGPRs (v..) or AGPRs (a..); writers: any v_* instruction's destination, both
operands of v_swap_b32, v_accvgpr_write's AGPR.

    python tools/store_hazard_scan.py [objects...]      exit code 1 if an instance is found."""
import glob, os, re, subprocess, sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
STORE = re.compile(r"\b(global|buffer|flat|scratch)_store_dwordx([34])\s+(.*)")
REG = re.compile(r"([va])\[(\d+):(\d+)\]|([va])(\d+)")
LABEL = re.compile(r"^[0-9a-f]+ <(L\d+)>:$")
FUNC = re.compile(r"^[0-9a-f]+ <([^>]+)>:$")


def regs(tok):
    """Register set of one operand as {('v', 3), ...}."""
    m = REG.fullmatch(tok.strip().rstrip(","))
    if not m:
        return set()
    if m.group(4) is not None:
        return {(m.group(4), int(m.group(5)))}
    return {(m.group(1), k) for k in range(int(m.group(2)), int(m.group(3)) + 1)}


def written(line):
    """Vector registers a VALU / MFMA instruction writes; empty for everything else."""
    ins = line.replace(",", " ").split()
    if not ins or not ins[0].startswith("v_"):
        return set()
    if ins[0].startswith(("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane", "v_nop")):
        return set()
    out = regs(ins[1]) if len(ins) > 1 else set()
    if ins[0].startswith("v_swap") and len(ins) > 2:
        out |= regs(ins[2])
    return out


def scan_text(dis):
    """(number of wide stores, [(function, store, overwriting instruction, wait states seen, needed)]) of a disassembly listing
    in `llvm-objdump -d --no-show-raw-insn --symbolize-operands` form."""
    lines, func_of, label_at = [], [], {}
    func = "?"
    for raw in dis.splitlines():
        t = raw.split("//")[0].strip()
        if not t:
            continue
        m = LABEL.match(t)
        if m:
            label_at[(func, m.group(1))] = len(lines)
            continue
        m = FUNC.match(t)
        if m:
            func = m.group(1)
            continue
        if t.endswith(":"):
            continue
        lines.append(t)
        func_of.append(func)
    found, nstores = [], 0
    for i, l in enumerate(lines):
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
        # depth-first over the paths leaving the store, `need` wait states deep
        stack, seen, hit = [(i + 1, 0)], set(), None
        while stack and hit is None:
            j, ws = stack.pop()
            while ws < need and j < len(lines) and func_of[j] == func_of[i]:
                if (j, ws) in seen:
                    break
                seen.add((j, ws))
                t = lines[j]
                if written(t) & data:
                    hit = (func_of[i], l, t, ws, need)
                    break
                mm = re.match(r"s_nop\s+(\d+)", t)
                ws += int(mm.group(1)) + 1 if mm else 1
                op = t.split()[0]
                if op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
                    break
                if op == "s_branch" or op.startswith("s_cbranch"):
                    tgt = label_at.get((func_of[i], t.split()[1])) if len(t.split()) > 1 else None
                    if tgt is not None:
                        stack.append((tgt, ws))
                    if op == "s_branch":
                        break
                j += 1
        if hit:
            found.append(hit)
    return nstores, found


def have_objdump():
    return os.path.exists(OBJDUMP)


def scan(path):
    # the device code object sits in the .hip_fatbin section: llvm-objdump --offloading extracts it next to its input
    work = "/tmp/_hazard_%d" % os.getpid()
    os.makedirs(work, exist_ok=True)
    local = os.path.join(work, os.path.basename(path))
    subprocess.run(["cp", path, local], check=True)
    subprocess.run([OBJDUMP, "--offloading", local], capture_output=True, text=True, cwd=work)
    cos = [f for f in glob.glob(local + ".*") if "gfx950" in f]
    dis = "".join(subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "--symbolize-operands", c], capture_output=True, text=True).stdout
                  for c in cos)
    subprocess.run(["rm", "-rf", work])
    return scan_text(dis)


if __name__ == "__main__":
    if not have_objdump():
        print("llvm-objdump not found at %s" % OBJDUMP)
        sys.exit(2)
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(HERE, "mvsnet_amd", "csrc", "*.o")))
    total = 0
    for o in objs:
        n, f = scan(o)
        total += len(f)
        print("%-28s %5d wide stores, %d with their data overwritten too early" % (os.path.basename(o), n, len(f)))
        for func, st, wr, ws, need in f[:6]:
            print("    %s\n        %s\n        %s      <- after %d wait state(s), %d needed" % (func[:90], st, wr, ws, need))
    sys.exit(1 if total else 0)
