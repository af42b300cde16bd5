"""Build-time audit of the hand-pipelined Winograd kernel (correctness of the asm loads rested on one compiler's register allocation).
Compiles csrc/conv_wino.hip for gfx950 here (hipcc cross-compiles without a GPU) and requires, for every wino_nt_kernel instantiation:
  * no scratch: ScratchSize 0, no VGPR spills (a spill reload inside the persistent loop also makes hipcc drain the prefetch queue every pass);
  * no s_waitcnt vmcnt(..) written by the compiler anywhere in the kernel (every vector-memory operation is inline asm with a hand-counted wait);
  * no compiler instruction touching the destination registers of an unseen (asm) register load before the hand wait that covers it.
A hipcc upgrade that changes any of this fails loudly instead of racing silently."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_winograd_kernels_have_no_scratch_no_compiler_waits_and_untouched_load_registers():
    src = os.path.join(ROOT, "semiuhpe_amd", "csrc", "conv_wino.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.devnull],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    cur, seen = None, {}
    for line in r.stderr.split("\n"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
        m = re.search(r"(ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill): (\d+)", line)
        if m and cur and "wino_" in cur:
            seen.setdefault(cur, {})[m.group(1)] = int(m.group(2))
    nt = [k for k in seen if "wino_nt_kernel" in k]
    assert len(nt) == 6, sorted(seen)
    for k, v in seen.items():
        assert v.get("ScratchSize [bytes/lane]") == 0 and v.get("VGPRs Spill") == 0, (k, v)
    # the wait / register audit (tools/debug/wino_waits.py exits non-zero and prints what it found)
    a = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "debug", "wino_waits.py")], capture_output=True, text=True, timeout=600)
    assert a.returncode == 0, a.stdout[-3000:] + a.stderr[-1000:]
    assert "registers of unseen loads touched before their wait: none" in a.stdout


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_small_winograd_kernel_has_no_scratch_no_compiler_waits_in_its_loop_and_untouched_load_registers():
    """wino_small_kernel<KS, STATS, RES> (csrc/conv_wino_small.hip): the patch-row and skip loads of its block loop are inline asm behind hand-counted
    waits.  Eight instantiations, none with scratch; the only s_waitcnt vmcnt hipcc itself writes are in the prologue (the filter image is read with
    loads it sees: an explicit __builtin_amdgcn_s_waitcnt keeps their wait out of the loop); no compiler instruction touches an asm load's destination
    before its wait -- including the loads of the block past the last one, which are drained before the final stores."""
    src = os.path.join(ROOT, "semiuhpe_amd", "csrc", "conv_wino_small.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.devnull],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    cur, seen = None, {}
    for line in r.stderr.split("\n"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
        m = re.search(r"(ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill): (\d+)", line)
        if m and cur and "wino_small_kernel" in cur:
            seen.setdefault(cur, {})[m.group(1)] = int(m.group(2))
    assert len(seen) == 8, sorted(seen)
    for k, v in seen.items():
        assert v.get("ScratchSize [bytes/lane]") == 0 and v.get("VGPRs Spill") == 0, (k, v)
    a = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "debug", "wino_waits.py"), "--small"], capture_output=True, text=True, timeout=600)
    assert a.returncode == 0, a.stdout[-3000:] + a.stderr[-1000:]
    assert "registers of unseen loads touched before their wait: none" in a.stdout and "kernels: 8" in a.stdout


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"), reason="llvm tools not installed")
def test_hand_pipelined_kernels_of_the_built_library_have_no_scratch():
    """The BUILT semiuhpe_amd/csrc/libsuhpe.so (what the GPU box loads), read through its code-object metadata (tools/debug/codeobj_meta.py): the
    kernels whose memory operations are inline asm with hand-counted waits -- the Winograd tiles, the LDS-DMA GEMM tiles -- and the register-chained
    attention kernels use no scratch and spill no VGPR (a spill reload is a vector-memory operation the hand counts do not know about; in the
    attention kernels it would undo the register chaining).  SGPR spills go to VGPR lanes (v_writelane / v_readlane), which is why every asm
    VMEM statement that reads an SGPR operand starts with `s_nop 4` (conv_dma.inl, conv_wino.hip)."""
    sys.path.insert(0, os.path.join(ROOT, "tools", "debug"))
    import codeobj_meta
    ks = codeobj_meta.kernels()
    assert len(ks) > 300, len(ks)
    want = {"gemm_nt_dma": 8, "wino_nt_kernel": 6, "wino_small_kernel": 8, "window_attn_fwd_rs": 2, "window_attn_bwd_rs": 2, "wino_wgrad_kernel": 1}
    for pat, least in want.items():
        hit = {k: v for k, v in ks.items() if pat in k}
        assert len(hit) >= least, (pat, sorted(hit))
        for k, v in hit.items():
            assert v["scratch"] == 0 and v["vgpr_spill"] == 0, (k, v)
    # round 6: the fused depthwise backward sits at the 256-register cap of two workgroups per CU by design (filter in LDS, fill in chunks of 7
    # pixels, nothing prefetched across the window loop); the instantiations the training step runs -- SiLU at compile time, both map sizes, both forms --
    # must stay free of scratch (every spilling variant measured slower: DESIGN.md section 3)
    fused = {k: v for k, v in ks.items() if "dw3x3_bwd_tile_kernel" in k and k.rstrip("E").endswith(("Li3E", "Li3EE")) or ("dw3x3_bwd_tile_kernel" in k and "ELi3EEEv" in k)}
    assert len(fused) == 4, sorted(k for k in ks if "dw3x3_bwd_tile_kernel" in k)
    for k, v in fused.items():
        assert v["scratch"] == 0 and v["vgpr_spill"] == 0, (k, v)
