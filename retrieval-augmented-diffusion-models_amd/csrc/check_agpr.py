#!/usr/bin/env python3
"""Build check for kernels whose accumulators are asm-owned literal AGPRs (conv_halo4.hip): inside such a kernel hipcc must not
name an AGPR at or above a<first_owned> in an instruction of its own (it allocates AGPRs from a0 upwards for parked VGPRs and
memory-to-memory values -- possibly on top of an accumulator).
usage: check_agpr.py file.s first_owned kernel_name_substring [...]"""
import re, sys

def main():
    path, first_owned, names = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
    text = open(path).read().split("\n")
    bad = 0
    cur = None; inasm = False
    for ln in text:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1) if any(n in m.group(1) for n in names) else None
            inasm = False
            continue
        if cur is None:
            continue
        if ";;#ASMSTART" in ln: inasm = True
        elif ";;#ASMEND" in ln: inasm = False
        elif not inasm:
            code = ln.split(";")[0]
            for m2 in re.finditer(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]", code):
                hi = int(m2.group(1)) if m2.group(1) else int(m2.group(3))
                if hi >= first_owned:
                    print(f"check_agpr: {cur}: compiler-generated '{ln.strip()}'"); bad += 1
        if ".end_amdhsa_kernel" in ln or ln.startswith("\t.section"):
            pass
    if bad:
        print(f"check_agpr: {bad} compiler AGPR accesses inside asm-owned-accumulator kernels"); sys.exit(1)

if __name__ == "__main__":
    main()
