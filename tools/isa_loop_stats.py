#!/usr/bin/env python3
"""Developer aid: instruction histogram of the innermost loop(s) of one kernel in a hipcc --save-temps .s file.
usage: isa_loop_stats.py file.s kernel_name_substring [substring2 ...]"""
import collections
import re
import sys

text = open(sys.argv[1]).read().split("\n")
subs = sys.argv[2:]
start = next(i for i, l in enumerate(text) if re.match(r"^[A-Za-z_][\w.$]*:", l) and all(s in l for s in subs) and not l.startswith("\t"))
end = next(i for i in range(start, len(text)) if ".end_amdhsa_kernel" in text[i])
body = text[start:end]
for i, l in enumerate(body):
    if "Loop Header" in l:
        label = l.split(":")[0].strip()
        back = max(j for j, m in enumerate(body) if re.search(r"s_cbranch\S*\s+" + re.escape(label) + r"\s*$", m) or re.search(r"s_branch\s+" + re.escape(label) + r"\s*$", m))
        loop = body[i:back + 1]
        hist = collections.Counter(m.group(1) for m in (re.match(r"\s+([a-z][a-z0-9_]+)", x) for x in loop) if m)
        n_v = sum(c for k, c in hist.items() if k.startswith("v_"))
        print(f"loop {label}: lines {i}..{back}, {sum(hist.values())} instructions, {n_v} vector")
        for k, c in hist.most_common(40):
            print(f"  {c:5d} {k}")
