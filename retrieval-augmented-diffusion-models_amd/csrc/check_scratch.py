#!/usr/bin/env python3
"""Build guard: parse hipcc -Rpass-analysis=kernel-resource-usage remarks, fail when a kernel spills more than LIMIT bytes
per lane to scratch.  knn KSEL=32 variants (k > 12 neighbours, rare) are allowed their known spill."""
import re, sys
limit = int(sys.argv[1]); bad = []
for f in sys.argv[2:]:
    name = None
    for line in open(f, errors="replace"):
        m = re.search(r"Function Name: (\S+)", line)
        if m: name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            n = int(m.group(1))
            if n > limit and not ("knn_" in name and "ILi32E" in name): bad.append((name, n, f))
if bad:
    for name, n, f in bad: print(f"check_scratch: {name} uses {n} bytes/lane of scratch (> {limit}) [{f}]", file=sys.stderr)
    sys.exit(1)
print(f"check_scratch: ok ({len(sys.argv) - 2} objects, limit {limit} B/lane)")
