"""Per-MBConv-block HBM ledger of a C2 training step: every launch of one EfficientNetV2-S MBConv block, forward and
backward, with the bytes the FETCH_SIZE / WRITE_SIZE counters saw for THAT dispatch, against the tensors the launch has to move and against the
ideal-fusion count of SURVEY 8d / tools/count_work.py (every conv input and output once, fp32).

    python tools/block_ledger.py <fetch_counter_collection.csv> <write_counter_collection.csv> [student se_mlp index = 10] > profiles/rNN_block_ledger.txt

Input: the two per-dispatch counter files `tools/profile_round.sh` keeps under profiles/rNN_pmc/ (separate --pmc passes over
`tools/step_bench.py effinetv2 32 128 1 2`, one stream, so dispatch order = program order and the two passes pair up dispatch by dispatch).
The steady-state step = the dispatches between the last two adam_kernel launches.  A block is found through its squeeze-excitation MLP: the
student's merged pass runs it on 160 samples (grid 160 x 1024), the teacher's on 128; the k-th student launch belongs to MBConv block k
(stages 4-6 of torchvision's efficientnet_v2_s as used at /root/reference/src/networks.py:301-303: 6 + 9 + 15 blocks; k = 10 -> features.5.4:
160 -> 960 -> 160 on 14 x 14).  Forward = from the block's expand GEMM to its closing BatchNorm (+ skip) kernel; backward = the mirror range
around the block's se_mlp_bwd launch.  Also prints what the whole step contains besides kernels of this library (copies, torch kernels).
"""
import csv, re, sys
from collections import Counter


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void suhpe::", "").replace("suhpe::", "")
        rows.append((int(r["Dispatch_Id"]), name, int(r["Grid_Size"]), float(r["Counter_Value"]) * 1024.0, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    rows.sort()
    return rows


def main():
    f, w = load(sys.argv[1]), load(sys.argv[2])
    which = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    assert len(f) == len(w) and all(a[1] == b[1] for a, b in zip(f, w)), "the two counter passes do not pair up dispatch by dispatch"
    marks = [i for i, r in enumerate(f) if r[1].startswith("adam_kernel")]
    lo, hi = marks[-2] + 1, marks[-1] + 1
    step = range(lo, hi)
    names = Counter(f[i][1].split("<")[0] for i in step)
    foreign = {k: v for k, v in names.items() if "rocclr" in k or "at::native" in k}
    tot_f, tot_w = sum(2 * f[i][3] for i in step), sum(w[i][3] for i in step)
    print(f"steady-state step: {hi - lo} dispatches, fetch {tot_f / 1e9:.2f} GB (FETCH_SIZE x 2) + write {tot_w / 1e9:.2f} GB = {(tot_f + tot_w) / 1e9:.2f} GB")
    print(f"not this library's kernels in the step: {foreign if foreign else 'none'}"
          f"   (whole profiled run: {sum(1 for r in f if 'copyBuffer' in r[1])} copyBuffer dispatches, all but these before the first step: "
          f"parameter flattening, tile-database pins, workspaces)")
    se_f = [i for i in step if f[i][1].startswith("se_mlp_fwd") and f[i][2] == 160 * 1024]
    se_b = [i for i in step if f[i][1].startswith("se_mlp_bwd")]
    assert len(se_f) == 30 and len(se_b) == 30, (len(se_f), len(se_b))
    i_f, i_b = se_f[which], se_b[len(se_b) - 1 - which]

    def is_tail(i):                         # the block's closing BatchNorm apply (act none, + skip): bn_act_fwd_kernel<4, -1>
        return f[i][1].startswith("bn_act_fwd_kernel<4, -1>")
    a = i_f
    while not is_tail(a - 1):
        a -= 1
    b = i_f
    while not is_tail(b):
        b += 1
    fwd = list(range(a, b + 1))
    # backward: from the closing BatchNorm's backward (bn_bwd_partial<4, -1> of the narrow tensor) to the expand conv's filter gradient
    a = i_b
    while not f[a][1].startswith("bn_bwd_partial<4, -1>"):
        a -= 1
    b = i_b + 1
    while not f[b][1].startswith("bn_bwd_partial<4, -1>"):
        b += 1
    bwd = list(range(a, b))
    E = 160 * 14 * 14 * 960 * 4 / 1e6
    C = 160 * 14 * 14 * 160 * 4 / 1e6
    print(f"\nblock = student MBConv #{which} (features.5.x: 160 -> 960 -> 160 @ 14 x 14, 160 images): expanded tensor E = {E:.1f} MB, block tensor C = {C:.1f} MB")
    for title, rng in (("FORWARD", fwd), ("BACKWARD", bwd)):
        print(f"\n{title}: {len(rng)} launches")
        print(f"{'kernel':58s} {'us':>7s} {'fetch MB':>9s} {'write MB':>9s} {'E-passes':>9s} {'TB/s':>6s}")
        tf = tw = tt = 0.0
        for i in rng:
            fb, wb, us = 2 * f[i][3] / 1e6, w[i][3] / 1e6, f[i][4]
            tf += fb; tw += wb; tt += us
            print(f"{f[i][1][:58]:58s} {us:7.1f} {fb:9.1f} {wb:9.1f} {(fb + wb) / E:9.2f} {(fb + wb) / us:6.2f}")
        print(f"{'sum':58s} {tt:7.1f} {tf:9.1f} {tw:9.1f} {(tf + tw) / E:9.2f}")
    ideal_f = (C + E) + (E + E) + (E + C)                   # expand in / out, depthwise in / out, project in / out
    print(f"\nideal-fusion count (tools/count_work.py: every conv input and output once): forward {ideal_f:.0f} MB = {ideal_f / E:.2f} E-passes, "
          f"backward 2 x that = {2 * ideal_f:.0f} MB")
    sf = sum(2 * f[i][3] + w[i][3] for i in fwd) / 1e6
    sb = sum(2 * f[i][3] + w[i][3] for i in bwd) / 1e6
    print(f"measured: forward {sf:.0f} MB ({sf / ideal_f:.2f} x), backward {sb:.0f} MB ({sb / (2 * ideal_f):.2f} x), block {sf + sb:.0f} MB ({(sf + sb) / (3 * ideal_f):.2f} x)")


if __name__ == "__main__":
    main()
