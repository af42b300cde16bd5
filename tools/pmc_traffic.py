"""HBM traffic per launch of every kernel from two rocprofv3 --pmc passes over the same command (FETCH_SIZE and WRITE_SIZE
cannot share a pass on gfx950: TCC has 4 slots, FETCH_SIZE takes 3 and WRITE_SIZE 2):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/f -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/w -- python3 bench.py ...
    python tools/pmc_traffic.py out/f out/w profiles/r01_pmc_traffic.json

Units and corrections (MI355X_MICROARCH.md "HBM", re-checked here with tools/bnbench.py and tools/micro/store_patterns on
known byte counts): both counters are KiB; FETCH_SIZE reports exactly half the bytes of 16-byte-per-lane streaming loads
on gfx950 (every load in these kernels is one), so it is doubled; WRITE_SIZE is exact for the 16-byte stores and for the
dword-per-lane stores of the GEMM epilogue.
"""
import csv, glob, json, re, sys


def normalize(name):
    """rocprofv3 kernel name -> the name bench.py / suhpe_conv2d_kernel_name uses"""
    name = re.sub(r"\(.*$", "", name.replace("void ", "").replace("suhpe::", "")).replace(" ", "")
    m = re.match(r"(igemm_nt_kernel<\d+,\d+,\d+,\d+,\d+),\d+(,(true|false|\d+))*>", name)     # drop MODE and the epilogue / input-transform flags
    if m:
        return m.group(1) + ">"
    m = re.match(r"(igemm_nt3x3_kernel<\d+,\d+,\d+,\d+),\d+(,(true|false))*>", name)          # drop the channel chunk, tile kind, epilogue flag
    if m:
        return m.group(1) + ">"
    m = re.match(r"(igemm_tn_kernel<\d+,\d+,\d+,\d+,\d+,\d+)(,(true|false))?>", name)          # drop the input-transform flag
    if m:
        return m.group(1) + ">"
    m = re.match(r"(gemm_nt_persistent<\d+,\d+,\d+,\d+),(true|false)>", name)
    if m:
        return m.group(1) + ">"
    m = re.match(r"(gemm_nt_resident<\d+,\d+,\d+),(true|false)>", name)
    if m:
        return m.group(1) + ">"
    if re.match(r"wino_nt_kernel<(true|false|\d+)>", name):          # every instantiation (narrow forms, the AFFINE eval-mode ones 4-6) is the menu's one tile
        return "wino_nt_kernel<64,64>"
    if re.match(r"wino_wgrad_kernel<\d+>", name):
        return "wino_wgrad_kernel<64,64>"
    if re.match(r"wino_small_kernel<", name):                        # (KS, statistics, skip add): one name, as suhpe_conv2d_kernel_name reports it
        return "wino_small_kernel"
    return name


def collect(d, counter):
    agg = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = agg.setdefault(normalize(r["Kernel_Name"]), [0.0, 0])
            a[0] += float(r["Counter_Value"]) * 1024.0
            a[1] += 1
    return agg


def step_bytes(d, counter, scale):
    """bytes of ONE steady-state training step: the dispatches between the last two `adam_kernel` launches of the run (a training step ends
    with exactly one), in dispatch order -- the warm-up step before them has already built every workspace and pinned every tile"""
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"]) * 1024.0 * scale))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "adam_kernel" in r[1]]
    if len(marks) < 2:
        return None, 0
    seg = rows[marks[-2] + 1:marks[-1] + 1]
    return sum(r[2] for r in seg), len(seg)


if __name__ == "__main__":
    fdir, wdir, out = sys.argv[1:4]
    fetch, write = collect(fdir, "FETCH_SIZE"), collect(wdir, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        fb, fn = fetch.get(k, [0.0, 0])
        wb, wn = write.get(k, [0.0, 0])
        res[k] = {"launches": max(fn, wn), "fetch_bytes_per_launch": round(2.0 * fb / max(fn, 1)), "write_bytes_per_launch": round(wb / max(wn, 1))}
        res[k]["hbm_bytes_per_launch"] = res[k]["fetch_bytes_per_launch"] + res[k]["write_bytes_per_launch"]
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench                      # csrc_digest(): the kernel sources these passes ran on (bench.py withholds the figure once they change)
    fstep, fcount = step_bytes(fdir, "FETCH_SIZE", 2.0)
    wstep, wcount = step_bytes(wdir, "WRITE_SIZE", 1.0)
    step = None
    if fstep is not None and wstep is not None:
        step = {"hbm_bytes": round(fstep + wstep), "fetch_bytes": round(fstep), "write_bytes": round(wstep), "launches": max(fcount, wcount),
                "how": "dispatches between the last two adam_kernel launches of the profiled run (one steady-state training step)"}
    workload = sys.argv[4] if len(sys.argv) > 4 else "effinetv2 32+128"
    json.dump({"note": "FETCH_SIZE x2 (gfx950 16-B/lane loads) + WRITE_SIZE, KiB -> bytes, averaged per launch over the profiled run",
               "csrc_digest": bench.csrc_digest(), "workload": workload, "step": step, "kernels": res}, open(out, "w"), indent=1)
    if step:
        print(f"one steady-state step: {step['hbm_bytes'] / 1e9:.1f} GB in {step['launches']} launches")
    top = sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]
    for k, v in top:
        print(f"{k[:60]:60s} x{v['launches']:6d} fetch {v['fetch_bytes_per_launch']/1e6:9.2f} MB write {v['write_bytes_per_launch']/1e6:9.2f} MB")
