"""Basic blocks of one kernel in a hipcc -S listing: instruction mix, MFMA count and the s_waitcnt sequence per block.
    python tools/debug/isa_blocks.py <file.s> <mangled-name substring> [min instructions]"""
import collections, re, sys
s = open(sys.argv[1]).read()
m = re.search(r"^(\S*%s\S*):" % re.escape(sys.argv[2]), s, re.M)
i = m.start(); j = s.index(".Lfunc_end", i)
minins = int(sys.argv[3]) if len(sys.argv) > 3 else 40
blocks, cur = [], ("entry", [])
for l in s[i:j].splitlines():
    lab = re.match(r"^(\.LBB\d+_\d+):", l)
    if lab:
        blocks.append(cur); cur = (lab.group(1), [])
    elif l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;"):
        cur[1].append(l.strip())
blocks.append(cur)
print(m.group(1), sum(len(b[1]) for b in blocks), "instructions in", len(blocks), "blocks")
for name, ins in blocks:
    if len(ins) < minins:
        continue
    c = collections.Counter(x.split()[0] for x in ins)
    print(name, len(ins), "mfma", sum(v for k, v in c.items() if "mfma" in k), "| waits:", " ".join(x.replace("s_waitcnt ", "") for x in ins if "s_waitcnt" in x)[:300])
    print("     ", c.most_common(10))
