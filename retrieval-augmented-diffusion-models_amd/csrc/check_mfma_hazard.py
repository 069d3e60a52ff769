#!/usr/bin/env python3
"""Build check for the asm-owned-MFMA kernels (lin4.hip, conv_halo4.hip): a compiler-generated VALU write of a VGPR that an inline-asm
v_mfma reads as SrcA / SrcB within the next few instructions is a hazard hipcc does not pad (it cannot see inside the asm statement):
the matrix pipe reads the STALE register.  Fails the build when such a write sits closer than MIN_DIST instructions in front of the
MFMA without an s_nop covering the gap (measured round 5: a v_mov one instruction ahead of a start-value MFMA gave garbage columns).
usage: check_mfma_hazard.py file.s kernel_name_substring [min_dist]"""
import re, sys

def main():
    path, name = sys.argv[1], sys.argv[2]
    min_dist = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    cur, code, bad = None, [], 0
    def scan(kname, code):
        nbad = 0
        for i, l in enumerate(code):
            if not l.startswith("v_mfma"):
                continue
            regs = set()
            for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", l):
                regs |= set(range(int(a), int(b) + 1))
            slack = 0
            for j in range(i - 1, max(-1, i - 1 - min_dist), -1):
                c = code[j]
                m = re.match(r"s_nop (\d+)", c)
                if m:
                    slack += int(m.group(1)) + 1
                    continue
                if slack + (i - 1 - j) >= min_dist:
                    break
                if c.startswith("v_mfma") or not c.startswith("v_"):
                    continue
                m = re.match(r"v_\w+\s+v\[(\d+):(\d+)\]", c) or re.match(r"v_\w+\s+v(\d+)\b", c)
                if not m:
                    continue
                g = [int(x) for x in m.groups()]
                if set(range(g[0], g[-1] + 1)) & regs:
                    print(f"check_mfma_hazard: {kname}: '{c}' {i - j} instruction(s) before '{l[:60]}'")
                    nbad += 1
        return nbad
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            if cur: bad += scan(cur, code)
            cur = m.group(1) if name in m.group(1) else None
            code = []
            continue
        if cur is None:
            continue
        s = ln.strip()
        if not s or s.startswith(";") or s.startswith("."):
            continue
        if ln.startswith("\t") or ln.startswith(" "):
            code.append(s.split(";")[0].strip())
        if ".end_amdhsa_kernel" in ln:
            bad += scan(cur, code); cur = None; code = []
    if bad:
        print(f"check_mfma_hazard: {bad} unpadded VALU -> MFMA operand hazards"); sys.exit(1)
    print("check_mfma_hazard: ok")

if __name__ == "__main__":
    main()
