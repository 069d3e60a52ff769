#!/usr/bin/env python3
"""Per-op cost table of the UNet executor at the benchmarked batch (GPU box only): HIP-event brackets around every launch of one
guided DDIM step (B = 64 -> UNet batch 128, shared guidance prefix, zero-context shortcut -- exactly what bench.py's step runs),
grouped by the op's role in the graph and its shape (C ABI: rdm_prof_enable / rdm_prof_dump).

    python tools/op_trace.py [--batch 64] [--steps 3] [--out gpurun_out/op_trace.csv]

Prints per (role, shape): launches per forward, average us, ms per forward, achieved TFLOP/s (GEMM-class) or TB/s (norms)."""
import argparse, collections, csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rdm_amd  # noqa: F401
from rdm_amd import _lib, synthetic
from rdm_amd.models.diffusion.ddpm import MinimalRETRODiffusion

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=4, help="DDIM steps traced (must divide 1000)")
ap.add_argument("--k", type=int, default=4)
ap.add_argument("--scale", type=float, default=2.0, help="guidance scale; 1.0 = no CFG (config #4's geometry: UNet batch = --batch)")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "op_trace.csv"))
a = ap.parse_args()
torch.set_grad_enabled(False)
ctx = _lib.Context(0)
model = MinimalRETRODiffusion(unet_config={"params": {}}, first_stage_config={"params": {"ddconfig": {}}}, k_nn=a.k, ctx=ctx)
model.load_unet_state_dict(synthetic.unet_state_dict(model.unet_cfg))
g = torch.Generator(device=ctx.device).manual_seed(0)
x = torch.randn((a.batch, 3, 64, 64), device=ctx.device, generator=g)
cond = torch.randn((a.batch, a.k, 512), device=ctx.device, generator=g) * 0.45
uncond = torch.zeros_like(cond) if a.scale > 1 else None
ctx.ddim_sample(a.steps, x, cond, uncond, model.alphas_cumprod, eta=0.0, scale=a.scale)          # warm-up (derived weights, arena)
torch.cuda.synchronize()
ctx.prof_reset()
ctx.prof_enable(tuple(range(7)))
ctx.ddim_sample(a.steps, x, cond, uncond, model.alphas_cumprod, eta=0.0, scale=a.scale)
torch.cuda.synchronize()
ctx.prof_enable(())
os.makedirs(os.path.dirname(a.out), exist_ok=True)
raw = a.out.replace(".csv", "_raw.csv")
ctx.prof_dump(raw)
acc = collections.OrderedDict()
for r in csv.DictReader(open(raw)):
    key = (int(r["kind"]), r["tag"], int(r["d0"]), int(r["d1"]), int(r["d2"]))
    e = acc.setdefault(key, [0, 0.0, 0.0])
    e[0] += 1; e[1] += float(r["ms"]); e[2] += float(r["work"])
os.remove(raw)
KIND = {0: "conv3x3", 1: "linear", 2: "knn", 3: "attention", 4: "groupnorm", 5: "layernorm", 6: "upsconv"}
rows = []
tot = 0.0
for (kind, tag, d0, d1, d2), (n, ms, work) in acc.items():
    per_fwd = ms / a.steps
    tot += per_fwd
    rate = work / (ms * 1e-3) / 1e12 if ms > 0 else 0.0          # TFLOP/s or TB/s
    rows.append((per_fwd, KIND.get(kind, str(kind)), tag, d0, d1, d2, n / a.steps, ms / n * 1e3, rate))
rows.sort(key=lambda r: -r[0])
with open(a.out, "w") as f:
    f.write("kind,role,d0,d1,d2,launches_per_forward,avg_us,ms_per_forward,tflops_or_tbps\n")
    for per_fwd, kind, tag, d0, d1, d2, n, us, rate in rows:
        f.write(f"{kind},{tag},{d0},{d1},{d2},{n:.1f},{us:.2f},{per_fwd:.4f},{rate:.2f}\n")
print(f"{'kind':10s} {'role':18s} {'shape':>24s} {'n/fwd':>6s} {'avg us':>9s} {'ms/fwd':>8s} {'TF|TB/s':>8s}")
for per_fwd, kind, tag, d0, d1, d2, n, us, rate in rows:
    print(f"{kind:10s} {tag:18s} {f'{d0}x{d1}x{d2}':>24s} {n:6.1f} {us:9.2f} {per_fwd:8.4f} {rate:8.2f}")
print(f"sum of bracketed launches: {tot:.3f} ms per UNet forward (B' = {2 * a.batch})")
by_role = collections.defaultdict(float)
for per_fwd, kind, tag, *_ in rows:
    by_role[tag] += per_fwd
print("by role:", ", ".join(f"{k} {v:.3f}" for k, v in sorted(by_role.items(), key=lambda kv: -kv[1])))
ctx.close()
