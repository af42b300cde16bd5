"""Compiler-inserted s_waitcnt vmcnt(..) inside the chunk loops of wino_nt_kernel<*> (outside the hand-written ASMSTART/ASMEND blocks).
    python tools/debug/wino_waits.py [extra -D flags]
    python tools/debug/wino_waits.py --small [extra -D flags]      wino_small_kernel<*> (conv_wino_small.hip): its prologue reads the filter image with
                                                                   loads hipcc sees, so compiler waits OUTSIDE the block loop are expected there
Every wait the compiler adds inside the loop drains the prefetch queue the kernel manages by hand."""
import re, subprocess, sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
SMALL = "--small" in sys.argv
if SMALL:
    sys.argv.remove("--small")
src = os.path.join(ROOT, "semiuhpe_amd", "csrc", "conv_wino_small.hip" if SMALL else "conv_wino.hip")
out = "/tmp/wino_waits_small.s" if SMALL else "/tmp/wino_waits.s"
KERNEL = r"^(_ZN5suhpe17wino_small_kernelILi(\d)ELb([01])ELb([01])E\S*):" if SMALL else r"^(_ZN5suhpe14wino_nt_kernelILi(\d)E\S*):"
def kernel_id(m):
    return int(m.group(2)) * 100 + int(m.group(3)) * 10 + int(m.group(4)) if SMALL else int(m.group(2))
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-S", "--cuda-device-only", "-I" + os.path.dirname(src),
                       "-o", out, src] + sys.argv[1:], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
cur, in_asm, res = None, False, {}
for i, l in enumerate(lines):
    m = re.match(KERNEL, l)
    if m:
        cur = kernel_id(m); res[cur] = []
    if cur is None:
        continue
    if "s_endpgm" in l:
        cur = None; continue
    if "#ASMSTART" in l: in_asm = True
    if "#ASMEND" in l: in_asm = False
    if "s_waitcnt" in l and "vmcnt" in l and not in_asm:
        # context: is it inside the loop? look back for the nearest label comment
        ctx = next((lines[j] for j in range(i, max(0, i - 400), -1) if lines[j].startswith(".LBB")), "")
        res[cur].append((i + 1, l.strip(), "in-loop" if "in Loop" in ctx or "Loop Header" in ctx else "outside", lines[i + 1].strip()[:60]))
    if "scratch_" in l:
        res[cur].append((i + 1, l.strip()[:50], "SCRATCH", ""))
for k in sorted(res):
    print("MODE", k)
    for r in res[k]:
        print("   ", *r)


# ---- audit of the asm register loads: between an unseen load (buffer_load / global_load inside ASMSTART..ASMEND) and the hand-written wait that
# follows it, no compiler instruction may touch the destination registers (a copy or spill there would read stale data)
def regs_of(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        out.add(int(a))
    return out


cur, in_asm, pending, bad = None, False, set(), []
for i, l in enumerate(lines):
    m = re.match(KERNEL, l)
    if m:
        cur, pending = kernel_id(m), set()
    if cur is None:
        continue
    if "s_endpgm" in l:
        cur = None; continue
    if "#ASMSTART" in l: in_asm = True; continue
    if "#ASMEND" in l: in_asm = False; continue
    t = l.split(";")[0].strip()
    if not t or t.startswith("."):
        continue
    covers = "covers-loads" in l
    if in_asm:
        m2 = re.match(r"(buffer_load_dwordx4|buffer_load_dwordx2|global_load_dwordx4)\s+v\[(\d+):(\d+)\]", t)
        if m2:
            pending.update(range(int(m2.group(2)), int(m2.group(3)) + 1))
        if t.startswith("s_waitcnt") and "vmcnt" in t and covers:
            pending = set()                     # (the hand waits that name the destination registers as "+v" operands carry this marker in the source)
    elif pending and (regs_of(t) & pending):
        bad.append((cur, i + 1, t))
print("registers of unseen loads touched before their wait:", bad if bad else "none")
if SMALL:
    print("kernels:", len(res))
    sys.exit(1 if bad or len(res) != 8 or any(r[2] != "outside" for v in res.values() for r in v) else 0)
sys.exit(1 if bad or any(res.values()) else 0)
