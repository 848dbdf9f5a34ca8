#!/usr/bin/env python3
"""Instruction census of one kernel in a hipcc -S dump: totals, per-class counts, and per basic block sizes."""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(key) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end + 1]
blocks, cur = [], ("entry", [])
for l in body[1:]:
    t = l.strip()
    if not t or t.startswith((";", "//", ".")) and not re.match(r"\.LBB\d+_\d+:", t):
        continue
    if re.match(r"\.LBB\d+_\d+:", t):
        blocks.append(cur)
        cur = (t[:-1], [])
        continue
    cur[1].append(t.split()[0])
blocks.append(cur)
tot = collections.Counter()
for name, ins in blocks:
    tot.update(ins)
print("total", sum(tot.values()))
cls = collections.Counter()
for k, v in tot.items():
    if k.startswith("v_") and any(x in k for x in ("rcp", "rsq", "sqrt", "sin", "cos", "exp", "log")):
        cls["trans"] += v
    elif k.startswith("v_"):
        cls["valu"] += v
    elif k.startswith("s_"):
        cls["salu"] += v
    elif k.startswith(("global_", "buffer_", "flat_", "scratch_")):
        cls["vmem:" + k] += v
    elif k.startswith("ds_"):
        cls["lds"] += v
    else:
        cls[k] += v
print(dict(cls))
print("top", tot.most_common(25))
for name, ins in blocks:
    if len(ins) > 30:
        c = collections.Counter(ins)
        tr = sum(v for k, v in c.items() if any(x in k for x in ("rcp", "rsq", "sqrt", "sin", "cos", "exp", "log")))
        print(f"{name:12s} n={len(ins):5d} valu={sum(v for k, v in c.items() if k.startswith('v_')):5d} trans={tr:3d} "
              f"salu={sum(v for k, v in c.items() if k.startswith('s_')):4d} mem={sum(v for k, v in c.items() if k.startswith(('global','scratch','ds_'))):3d}")
