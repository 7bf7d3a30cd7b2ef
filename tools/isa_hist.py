"""Instruction histogram of one kernel in a hipcc -S listing, whole body and per basic block with MFMA/VALU counts.
usage: python tools/isa_hist.py /tmp/mfma.s <mangled-name-prefix> [--blocks]"""
import collections, sys
lines = open(sys.argv[1]).read().split("\n")
pre = sys.argv[2]
i = next(k for k, l in enumerate(lines) if l.startswith(pre) and ":" in l)
end = next(k for k in range(i, len(lines)) if "s_endpgm" in lines[k])
b = lines[i:end]
c = collections.Counter()
for l in b:
    t = l.strip().split()
    if t and not t[0].startswith((";", ".")) and not t[0].endswith(":"):
        c[t[0]] += 1
print(sum(c.values()), "instructions")
for k, v in c.most_common(40):
    print("   ", k, v)
if "--blocks" in sys.argv:
    cur, cnt = "entry", collections.Counter()
    for l in b:
        t = l.strip().split()
        if not t:
            continue
        if t[0].endswith(":") and t[0].startswith(".LBB"):
            if sum(cnt.values()) > 30:
                print(cur, sum(cnt.values()), dict(cnt.most_common(8)))
            cur, cnt = t[0], collections.Counter()
        elif not t[0].startswith((";", ".")):
            cnt[t[0]] += 1
    print(cur, sum(cnt.values()), dict(cnt.most_common(8)))
