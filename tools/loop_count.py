"""Instruction histogram of every loop of one kernel in an assembly listing (hipcc -S
--cuda-device-only): a loop is the stretch between a backward branch and its target.
Usage: python tools/loop_count.py listing.s kernel-name-substring [min-size]"""
import collections
import re
import sys

text = open(sys.argv[1]).read()
want = sys.argv[2]
min_size = int(sys.argv[3]) if len(sys.argv) > 3 else 200
for m in re.finditer(r"^(\S*%s\S*):" % re.escape(want), text, re.M):
    end = text.index(".Lfunc_end", m.end())
    seq = []
    for line in text[m.end():end].split("\n"):
        line = line.split(";")[0].strip()
        if not line or (line.startswith(".") and not line.endswith(":")):
            continue
        seq.append(line)
    labels = {l[:-1]: k for k, l in enumerate(seq) if l.endswith(":")}
    print(m.group(1), len([l for l in seq if not l.endswith(":")]), "instructions")
    for k, l in enumerate(seq):
        b = re.match(r"s_c?branch\S*\s+(\.LBB\S+)", l)
        if b and b.group(1) in labels and labels[b.group(1)] < k:
            body = [x for x in seq[labels[b.group(1)]:k + 1] if not x.endswith(":")]
            if len(body) < min_size:
                continue
            c = collections.Counter(x.split()[0] for x in body)
            print("  loop at", b.group(1), len(body), "instructions,", sum(v for n, v in c.items() if "mfma" in n), "mfma")
            print("   ", sorted(c.items(), key=lambda x: -x[1])[:40])
