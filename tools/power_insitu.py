#!/usr/bin/env python3
"""Power / shader clock of the GPU WHILE the headline bench runs (the in-situ companion of power_probe.py): starts `bench.py` as a child
process and samples rocm-smi every 0.25 s until it ends; prints the bench line and mean / max of the samples taken inside the timed steps
(the upper half of the power readings: the setup phase idles).  GPU box only.  usage: power_insitu.py [bench args ...]"""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:] or ["--steps", "6", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
child = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
samples = []
while child.poll() is None:
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
        pw = re.search(r"Power \(W\):\s*([0-9.]+)", out); ck = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", out)
        if pw and ck:
            samples.append((time.time(), float(pw.group(1)), int(ck.group(1))))
    except Exception:
        pass
    time.sleep(0.25)
line = child.stdout.read().strip().splitlines()
print(line[-1] if line else "(no bench line)")
if samples:
    ps = sorted(s[1] for s in samples)
    cut = ps[len(ps) // 2]
    busy = [s for s in samples if s[1] >= cut]
    print(f"{len(samples)} samples over {samples[-1][0] - samples[0][0]:.1f} s; loaded half ({len(busy)} samples): power mean {sum(s[1] for s in busy) / len(busy):.0f} W, "
          f"max {max(s[1] for s in busy):.0f} W; sclk mean {sum(s[2] for s in busy) / len(busy):.0f} MHz, min {min(s[2] for s in busy)} MHz, max {max(s[2] for s in busy)} MHz")
    print("every 8th sample (t, W, MHz):", [(round(s[0] - samples[0][0], 1), s[1], s[2]) for s in samples[::8]])
