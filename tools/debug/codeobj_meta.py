"""Per-kernel metadata of the BUILT library (no recompilation): private (scratch) segment size, SGPR / VGPR spill counts, VGPR count and LDS size of
every kernel in semiuhpe_amd/csrc/libsuhpe.so, read from the code objects embedded in its .hip_fatbin section.
    python tools/debug/codeobj_meta.py [substring ...]          (prints name, vgprs, scratch bytes, spills; exit 1 if a listed kernel has scratch)
Used by tests/test_codeobj_audit.py: the hand-pipelined kernels (inline-asm loads whose registers the compiler must not move) may not spill."""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "semiuhpe_amd", "csrc", "libsuhpe.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernels(lib=LIB):
    out = {}
    with tempfile.TemporaryDirectory() as td:
        fb = os.path.join(td, "fb.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fb}", lib, os.path.join(td, "unused.so")], check=True)
        blob = open(fb, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for n, s in enumerate(starts):
            part = os.path.join(td, f"b{n}.bin")
            open(part, "wb").write(blob[s:starts[n + 1] if n + 1 < len(starts) else len(blob)])
            co = os.path.join(td, f"b{n}.co")
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={part}",
                                f"--output={co}"], capture_output=True, text=True)
            if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk)
                if not name:
                    continue
                f = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1)) if re.search(r"\.%s:\s+(\d+)" % key, blk) else -1
                out[name.group(1)] = {"vgpr": f("vgpr_count"), "scratch": f("private_segment_fixed_size"), "sgpr_spill": f("sgpr_spill_count"),
                                      "vgpr_spill": f("vgpr_spill_count"), "lds": f("group_segment_fixed_size")}
    return out


if __name__ == "__main__":
    ks = kernels()
    pats = sys.argv[1:]
    bad = 0
    for k in sorted(ks):
        if pats and not any(p in k for p in pats):
            continue
        v = ks[k]
        print(f"{k[:110]:110s} vgpr {v['vgpr']:4d} scratch {v['scratch']:5d} sgpr_spill {v['sgpr_spill']:4d} vgpr_spill {v['vgpr_spill']:4d}")
        bad += bool(pats) and (v["scratch"] != 0 or v["vgpr_spill"] != 0)
    sys.exit(1 if bad else 0)
