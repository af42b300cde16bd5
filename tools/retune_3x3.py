"""Re-measure the tile winners of the 3x3 / stride 1 forward and input-gradient GEMMs (the halo-tile kernel joined the menu):
drops those entries from semiuhpe_amd/tiles_mi355x.json, lets the first-use autotuner (kernels._autotune) time the whole menu
on the step's real operands for the three BASELINE backbones, and writes the merged database.

    python tools/retune_3x3.py gpurun_out/tiles_mi355x.json      (needs an MI355X; copy the result over semiuhpe_amd/tiles_mi355x.json)
"""
import json, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
out = os.path.abspath(sys.argv[1])
src = os.path.join(ROOT, "semiuhpe_amd", "tiles_mi355x.json")
db = json.load(open(src))
keep = {}
for k, v in db["tiles"].items():
    p = k.split(",")          # pass,n,h,w,cin,cout,r,s,stride,pad,groups
    if p[0] in ("0", "1", "2") and p[6] == "3" and p[7] == "3" and p[8] == "1" and p[9] == "1":
        continue
    keep[k] = v
print(f"dropped {len(db['tiles']) - len(keep)} of {len(db['tiles'])} entries")
db["tiles"] = keep
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(db, open(out, "w"), indent=0)
runs = [("effinetv2", 32, 128), ("repvgg", 64, 256), ("tinyvit", 32, 128), ("resnet50", 32, 128)] if len(sys.argv) < 3 else [tuple(a.split(":")) for a in sys.argv[2:]]
for net, b, mu in runs:
    env = dict(os.environ, SUHPE_TILE_DB=out, SUHPE_TILE_DB_SAVE=out, SUHPE_OVERLAP="0", SUHPE_WGRAD_OVERLAP="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_bench.py"), net, str(b), str(mu), "2", "3"], env=env, capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:], flush=True)
new = json.load(open(out))["tiles"]
halo = {k: v for k, v in new.items() if "nt3x3" in v or "wino" in v}
print(f"{len(new)} entries, {sum('nt3x3' in v for v in halo.values())} won by the halo kernel, {sum('wino' in v for v in halo.values())} by the Winograd kernel")
for k, v in sorted(halo.items()):
    print(" ", k, v)
