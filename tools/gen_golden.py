#!/usr/bin/env python3
"""Generate golden vectors from the reference's IN-TREE classes (runs only in the build
container, where /root/reference exists; the fixtures it writes are what travels).

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py

For every fixture it first asserts  oracle == reference-class  (fp32, atol 2e-5) on the
same seeded weights/inputs — this is what pins the oracle — then stores inputs + the
REFERENCE outputs under tests/golden/.  Weights are not stored: they are regenerated
deterministically by oracle.unet.synth_state_dict(shapes, seed).
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tools", "ldm_shim"))   # shim `main`/`ldm` must shadow the reference's main.py
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import clip as oclip
from oracle import unet as ounet

OUT = os.path.join(ROOT, "tests", "golden")
torch.manual_seed(0)
torch.set_grad_enabled(False)


def _ref_unet(spec):
    from rdm.modules.diffusionmodules.openaimodel import UNetModel
    m = UNetModel(image_size=64, in_channels=spec.in_channels, out_channels=spec.out_channels,
                  model_channels=spec.model_channels, attention_resolutions=list(spec.attention_resolutions),
                  num_res_blocks=spec.num_res_blocks, channel_mult=list(spec.channel_mult),
                  num_head_channels=spec.num_head_channels, use_spatial_transformer=True, transformer_depth=1,
                  context_dim=spec.context_dim, use_checkpoint=True)
    return m.eval()


def check(name, a, b, atol=2e-5, rtol=2e-5):
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    print(f"[{name}] max|oracle-ref| = {err:.3e} (ref max {ref:.3e})")
    assert err <= atol + rtol * ref, name


def gen_unet(tag, spec, B, k, hw, seed):
    shapes = ounet.param_shapes(spec)
    ref = _ref_unet(spec)
    ref_keys = {k_: tuple(v.shape) for k_, v in ref.state_dict().items()}
    assert ref_keys == shapes, (set(ref_keys) ^ set(shapes))
    sd = ounet.synth_state_dict(shapes, seed=seed)
    ref.load_state_dict(sd, strict=True)
    rng = np.random.default_rng(seed + 1)
    x = torch.from_numpy(rng.standard_normal((B, spec.in_channels, hw, hw)).astype(np.float32))
    t = torch.from_numpy(rng.integers(0, 1000, size=(B,)).astype(np.int64))
    ctx = torch.from_numpy((rng.standard_normal((B, k, spec.context_dim)) * 0.45).astype(np.float32))
    y_ref = ref(x, t, context=[ctx.clone()])
    y_or = ounet.unet_forward(sd, spec, x, t, ctx)
    check(f"unet/{tag}", y_or, y_ref, atol=5e-5, rtol=5e-5)
    np.savez_compressed(os.path.join(OUT, f"unet_{tag}.npz"), x=x.numpy(), t=t.numpy(), ctx=ctx.numpy(),
                        eps=y_ref.numpy(), seed=np.int64(seed),
                        n_params=np.int64(sum(int(np.prod(s)) for s in shapes.values())))


def gen_attention(seed=77):
    from rdm.modules.attention import CrossAttention, SpatialTransformer
    rng = np.random.default_rng(seed)
    C, heads, k = 128, 4, 4
    # SpatialTransformer
    st = SpatialTransformer(C, heads, 32, depth=1, context_dim=512).eval()
    shapes = {n: tuple(v.shape) for n, v in st.state_dict().items()}
    sd = ounet.synth_state_dict(shapes, seed=seed)
    st.load_state_dict(sd)
    x = torch.from_numpy(rng.standard_normal((2, C, 8, 8)).astype(np.float32))
    ctx = torch.from_numpy((rng.standard_normal((2, k, 512)) * 0.45).astype(np.float32))
    y_ref = st(x, context=[ctx.clone()])
    sdp = {"st." + n: v for n, v in sd.items()}
    y_or = ounet.spatial_transformer(sdp, "st", x, ctx, heads)
    check("spatial_transformer", y_or, y_ref)
    # CrossAttention self + cross
    ca = CrossAttention(C, context_dim=512, heads=heads, dim_head=32).eval()
    shapes2 = {n: tuple(v.shape) for n, v in ca.state_dict().items()}
    sd2 = ounet.synth_state_dict(shapes2, seed=seed + 1)
    ca.load_state_dict(sd2)
    tok = torch.from_numpy(rng.standard_normal((2, 64, C)).astype(np.float32))
    y2_ref = ca(tok, context=ctx)
    y2_or = ounet.cross_attention({"a." + n: v for n, v in sd2.items()}, "a", tok, ctx, heads)
    check("cross_attention", y2_or, y2_ref)
    np.savez_compressed(os.path.join(OUT, "attention.npz"), st_x=x.numpy(), st_ctx=ctx.numpy(), st_y=y_ref.numpy(),
                        ca_x=tok.numpy(), ca_ctx=ctx.numpy(), ca_y=y2_ref.numpy(), seed=np.int64(seed))


def gen_clip(seed=99):
    from rdm.modules.custom_clip.model import CLIP
    spec = oclip.tiny_clip_spec()
    m = CLIP(spec.embed_dim, spec.image_resolution, spec.vision_layers, spec.vision_width, spec.vision_patch_size,
             spec.context_length, spec.vocab_size, spec.transformer_width, spec.transformer_heads,
             spec.transformer_layers).eval()
    shapes = {n: tuple(v.shape) for n, v in m.state_dict().items() if n != "logit_scale"}
    assert shapes == oclip.clip_param_shapes(spec), set(shapes) ^ set(oclip.clip_param_shapes(spec))
    sd = ounet.synth_state_dict(shapes, seed=seed)
    sd["positional_embedding"] = sd["positional_embedding"] * 0.1
    m.load_state_dict({**sd, "logit_scale": torch.ones([])})
    rng = np.random.default_rng(seed + 1)
    tokens = np.zeros((3, 77), dtype=np.int64)
    for i, L in enumerate((5, 12, 77)):
        tokens[i, :L] = rng.integers(1, spec.vocab_size - 2, size=L)
        tokens[i, 0] = spec.vocab_size - 2
        tokens[i, L - 1] = spec.vocab_size - 1          # EOT = highest id -> argmax position
    tokens = torch.from_numpy(tokens)
    img = torch.from_numpy(rng.standard_normal((2, 3, spec.image_resolution, spec.image_resolution)).astype(np.float32))
    t_ref, i_ref = m.encode_text(tokens), m.encode_image(img)
    check("clip.encode_text", oclip.encode_text(sd, spec, tokens), t_ref, atol=5e-5)
    check("clip.encode_image", oclip.encode_image(sd, spec, img), i_ref, atol=5e-5)
    np.savez_compressed(os.path.join(OUT, "clip_tiny.npz"), tokens=tokens.numpy(), image=img.numpy(),
                        text_out=t_ref.numpy(), image_out=i_ref.numpy(), seed=np.int64(seed))


def gen_tokenizer():
    from rdm.modules.custom_clip.simple_tokenizer import SimpleTokenizer
    tk = SimpleTokenizer()
    caps = ["a happy bear reading a newspaper, oil on canvas", "A photo of a dog.", "  Hello,   WORLD!! 123  ",
            "an armchair in the shape of an avocado", ""]
    sot, eot = tk.encoder["<|startoftext|>"], tk.encoder["<|endoftext|>"]
    rows = np.zeros((len(caps), 77), dtype=np.int64)
    for i, c in enumerate(caps):
        ids = [sot] + tk.encode(c) + [eot]
        rows[i, :len(ids)] = ids
    print("tokenizer KAT row0:", rows[0, :14].tolist())
    np.savez_compressed(os.path.join(OUT, "tokenizer.npz"), captions=np.array(caps), tokens=rows)


def gen_rarm():
    """RARM backbone: the in-tree RetrievalPatchTransformer class (rdm/modules/attention.py:199-272) with the shipped switches
    (continuous=False, causal, cross_attend, positional encodings).  tiny: full forward + a guided, top-k sampled sequence through
    the oracle's sampling loop with the REFERENCE class as the transformer; shipped (--full): logits of an 8-token prefix."""
    from rdm.modules.attention import RetrievalPatchTransformer
    from oracle import rarm as orarm

    def build(spec, seed):
        m = RetrievalPatchTransformer(in_channels=spec.vocab_in, n_heads=spec.n_heads, d_head=spec.d_head, depth=spec.depth,
                                      context_dim=spec.context_dim, positional_encodings=True, sequence_length=spec.sequence_length,
                                      out_channels=spec.vocab_out, cross_attend=True, causal=True, continuous=False).eval()
        shapes = {n: tuple(v.shape) for n, v in m.state_dict().items()}
        assert shapes == orarm.rarm_param_shapes(spec), set(shapes) ^ set(orarm.rarm_param_shapes(spec))
        sd = ounet.synth_state_dict(shapes, seed=seed)
        m.load_state_dict(sd)
        return m, sd

    spec = orarm.tiny_rarm_spec()
    m, sd = build(spec, 777)
    rng = np.random.default_rng(778)
    tokens = torch.from_numpy(rng.integers(0, spec.vocab_out, size=(3, 12)).astype(np.int64))
    tokens[:, 0] = spec.vocab_in - 1                                   # sos
    ctx = torch.from_numpy((rng.standard_normal((3, 4, 512)) * 0.45).astype(np.float32))
    y_ref = m(tokens, context=ctx)
    check("rarm/tiny forward", orarm.rarm_forward(sd, spec, tokens, ctx), y_ref, atol=5e-5, rtol=5e-5)
    steps = 16
    u = torch.from_numpy(rng.random((steps, 3)).astype(np.float32))
    cond = torch.full((3, 1), spec.vocab_in - 1, dtype=torch.long)
    out_ref, lg_ref = orarm.rarm_sample(sd, spec, cond, ctx, steps, u, temperature=0.9, top_k=50, guidance_scale=2.0,
                                        forward=lambda t_, c_: m(t_, context=c_))
    out_or, lg_or = orarm.rarm_sample(sd, spec, cond, ctx, steps, u, temperature=0.9, top_k=50, guidance_scale=2.0)
    assert torch.equal(out_ref, out_or)
    check("rarm/tiny sampled logits", lg_or, lg_ref, atol=1e-4, rtol=1e-4)
    np.savez_compressed(os.path.join(OUT, "rarm_tiny.npz"), tokens=tokens.numpy(), ctx=ctx.numpy(), logits=y_ref.numpy(), uniforms=u.numpy(),
                        sampled=out_ref.numpy(), sampled_logits=lg_ref.numpy(), seed=np.int64(777), temperature=np.float32(0.9),
                        top_k=np.int64(50), guidance_scale=np.float32(2.0))
    if "--full" in sys.argv:
        spec = orarm.shipped_rarm_spec()
        m, sd = build(spec, 777)
        tokens = torch.from_numpy(rng.integers(0, spec.vocab_out, size=(2, 8)).astype(np.int64))
        tokens[:, 0] = spec.vocab_in - 1
        ctx = torch.from_numpy((rng.standard_normal((2, 8, 512)) * 0.45).astype(np.float32))
        y_ref = m(tokens, context=ctx)
        check("rarm/shipped forward", orarm.rarm_forward(sd, spec, tokens, ctx), y_ref, atol=1e-4, rtol=1e-4)
        np.savez_compressed(os.path.join(OUT, "rarm_shipped.npz"), tokens=tokens.numpy(), ctx=ctx.numpy(),
                            logits_last=y_ref[:, -2:].numpy().astype(np.float32), seed=np.int64(777))


def gen_rarm_deep():
    """RARM at the size that is benchmarked (BASELINE config #5: 18 x 768, vocab 16386 -> 16384, k = 8), at DEPTH: the 8-token prefix
    of rarm_shipped.npz exercises neither the K/V cache beyond position 8 nor the sampler at vocab 16384 / top-k 256.  Stored from
    the reference's in-tree RetrievalPatchTransformer (rdm/modules/attention.py:199-272), same weights (seed 777):
      * a full 256-token prefix, one row with random neighbours and one with ZERO neighbours (the unconditional half of a guided
        batch): logits at positions 0, 31, 127, 255;
      * a 32-step guided (scale 2.0), temperature 1.0, top-k 256 sampled continuation from <sos> for two rows through the oracle's
        sampling loop with the REFERENCE class as the transformer (LatentImageRETRO.sample, transformer.py:224-271): all tokens, the
        guided logits at steps 0, 7, 31, and the RAW conditional / unconditional logits at steps 0 and 31 (what the sampler kernel
        combines itself)."""
    from rdm.modules.attention import RetrievalPatchTransformer
    from oracle import rarm as orarm
    spec = orarm.shipped_rarm_spec()
    m = RetrievalPatchTransformer(in_channels=spec.vocab_in, n_heads=spec.n_heads, d_head=spec.d_head, depth=spec.depth,
                                  context_dim=spec.context_dim, positional_encodings=True, sequence_length=spec.sequence_length,
                                  out_channels=spec.vocab_out, cross_attend=True, causal=True, continuous=False).eval()
    shapes = {n: tuple(v.shape) for n, v in m.state_dict().items()}
    assert shapes == orarm.rarm_param_shapes(spec)
    sd = ounet.synth_state_dict(shapes, seed=777)
    m.load_state_dict(sd)
    rng = np.random.default_rng(4242)
    L = spec.sequence_length
    tokens = torch.from_numpy(rng.integers(0, spec.vocab_out, size=(2, L)).astype(np.int64))
    tokens[:, 0] = spec.vocab_in - 1
    ctx = torch.from_numpy((rng.standard_normal((2, 8, 512)) * 0.45).astype(np.float32))
    ctx[1] = 0.0
    y_ref = m(tokens, context=ctx)
    check("rarm/deep forward", orarm.rarm_forward(sd, spec, tokens, ctx), y_ref, atol=2e-4, rtol=2e-4)
    pos = np.array([0, 31, 127, 255])
    steps = 32
    sctx = torch.from_numpy((rng.standard_normal((2, 8, 512)) * 0.45).astype(np.float32))
    u = torch.from_numpy(rng.random((steps, 2)).astype(np.float32))
    cond = torch.full((2, 1), spec.vocab_in - 1, dtype=torch.long)
    raw = {}

    def fwd(t_, c_):
        out = m(t_, context=c_)
        raw[t_.shape[1] - 1] = out[:, -1].clone()          # [2B, vocab]: conditional rows first, then the zero-neighbour rows
        return out
    toks, lg = orarm.rarm_sample(sd, spec, cond, sctx, steps, u, temperature=1.0, top_k=256, guidance_scale=2.0, forward=fwd)
    toks_o, lg_o = orarm.rarm_sample(sd, spec, cond, sctx, steps, u, temperature=1.0, top_k=256, guidance_scale=2.0)
    assert torch.equal(toks, toks_o)
    check("rarm/deep sampled logits", lg_o, lg, atol=3e-4, rtol=3e-4)
    keep = [0, 7, 31]
    np.savez_compressed(os.path.join(OUT, "rarm_shipped_deep.npz"), tokens=tokens.numpy(), ctx=ctx.numpy(), positions=pos,
                        logits_at=y_ref[:, pos].numpy().astype(np.float32), sample_ctx=sctx.numpy(), uniforms=u.numpy(),
                        sampled=toks.numpy(), guided_steps=np.array(keep), guided_logits=lg[:, keep].numpy().astype(np.float32),
                        raw_steps=np.array([0, 31]), raw_logits=torch.stack([raw[0], raw[31]]).numpy().astype(np.float32),
                        seed=np.int64(777), temperature=np.float32(1.0), top_k=np.int64(256), guidance_scale=np.float32(2.0))


def gen_script_flags():
    """Flag table of the reference CLI (scripts/rdm_sample.py:22-143), read from its argparse calls with `ast` (the script
    itself cannot be imported: torchvision / clip / omegaconf are absent).  Stored as data: option strings, type name,
    default, action — what tests/test_host_cpu.py::test_rdm_sample_flags_match_reference compares our parser with."""
    import ast
    import json
    src = open("/root/reference/scripts/rdm_sample.py").read()
    flags = []
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_argument":
            kw = {k.arg: k.value for k in node.keywords}
            ent = {"options": [ast.literal_eval(a) for a in node.args],
                   "type": (kw["type"].id if isinstance(kw.get("type"), ast.Name) else None),
                   "action": ast.literal_eval(kw["action"]) if "action" in kw else None,
                   "default": ast.literal_eval(kw["default"]) if "default" in kw else None}
            flags.append(ent)
    flags.sort(key=lambda e: e["options"][-1])
    with open(os.path.join(OUT, "rdm_sample_flags.json"), "w") as f:
        json.dump(flags, f, indent=1)
    print(f"rdm_sample.py flag table: {len(flags)} flags")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if "--rarm-deep" in sys.argv:                  # only the deep RARM fixture (~2 min of CPU)
        gen_rarm_deep(); sys.exit(0)
    gen_script_flags()
    gen_rarm()
    gen_attention()
    gen_unet("tiny", ounet.tiny_spec(), B=2, k=4, hw=16, seed=1234)
    gen_clip()
    gen_tokenizer()
    if "--full" in sys.argv:
        gen_unet("shipped", ounet.shipped_spec(), B=1, k=4, hw=64, seed=1234)
        gen_rarm_deep()
    print("golden fixtures written to", OUT)
