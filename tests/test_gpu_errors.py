"""Error behaviour of the C ABI (include/rdm_hip.h: every entry returns 0 or a negative code and leaves a message in rdm_last_error)
and of the ctypes binding's argument checks: calls before weights / database are loaded, wrong shapes, out-of-range k / S / positions,
blobs that do not match the manifest -- and the context stays usable after every one of them."""
import numpy as np
import pytest
import torch

from oracle import rarm as orarm
from oracle import unet as ounet
from oracle import vqdecoder as ovq

from _util import spec_to_unet_cfg, spec_to_vq_cfg

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fresh():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import rdm_amd  # noqa: F401
    from rdm_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _err(fn):
    from rdm_amd._lib import RdmError
    with pytest.raises(RdmError) as e:
        fn()
    msg = str(e.value)
    assert len(msg) > 20, msg                   # code + a real message, not an empty string
    return msg


def test_calls_before_load_fail_with_a_message(fresh):
    d = fresh.device
    x = torch.zeros(1, 3, 16, 16, device=d); t = torch.zeros(1, dtype=torch.long, device=d); c = torch.zeros(1, 4, 512, device=d)
    assert "not loaded" in _err(lambda: fresh.unet_forward(x, t, c))
    assert "not loaded" in _err(lambda: fresh.vq_decode(x))
    assert "no database" in _err(lambda: fresh.knn(torch.zeros(2, 512, device=d), 4))
    assert "no database" in _err(lambda: fresh.db_gather(torch.zeros(2, 4, dtype=torch.int32, device=d), 512))
    _err(lambda: fresh.clip_encode_text(torch.zeros(1, 77, dtype=torch.long, device=d)))
    _err(lambda: fresh.rarm_forward(torch.zeros(1, 4, dtype=torch.long, device=d), torch.zeros(1, 2, 512, device=d)))


def test_knn_argument_checks(fresh):
    d = fresh.device
    g = torch.Generator(device=d).manual_seed(1)
    db = torch.randn(1000, 512, device=d, generator=g).half()
    fresh.db_load(db)
    q = torch.randn(3, 512, device=d, generator=g)
    assert "k > 28" in _err(lambda: fresh.knn(q, 29))
    small = torch.randn(5, 512, device=d, generator=g).half()
    fresh.db_load(small)
    assert "exceeds" in _err(lambda: fresh.knn(q, 6))
    # still usable, and exact
    idx, sc = fresh.knn(q, 5)
    ref = (torch.nn.functional.normalize(q.double(), dim=1) @ torch.nn.functional.normalize(small.double(), dim=1).T).argsort(dim=1, descending=True)
    assert torch.equal(idx.cpu().long() & 0xffffffff, ref.cpu())


def test_blob_and_shape_mismatches(fresh):
    from rdm_amd import _lib, packing
    spec = ounet.tiny_spec()
    cfg = spec_to_unet_cfg(spec)
    sd = ounet.synth_state_dict(ounet.param_shapes(spec), seed=3)
    blob = packing.pack("unet", cfg, sd)
    assert "manifest" in _err(lambda: fresh.load_unet(cfg, blob[:-256]))
    fresh.load_unet(cfg, blob)
    d = fresh.device
    B, k = 2, 4
    x = torch.zeros(B, spec.in_channels, 16, 16, device=d); t = torch.zeros(B, dtype=torch.long, device=d)
    c = torch.zeros(B, k, spec.context_dim, device=d)
    fresh.unet_forward(x, t, c)                                            # fine
    _err(lambda: fresh.unet_forward(x, t, c[:, 0]))                        # [B, 512] instead of [B, k, 512]
    _err(lambda: fresh.unet_forward(x, t[:1], c))                          # timestep vector of the wrong length
    _err(lambda: fresh.unet_forward(x, t, c[:1]))                          # conditioning for a different batch
    ac = torch.linspace(0.999, 0.005, 1000)
    _err(lambda: fresh.ddim_sample(4, x, c, c[:, :2], ac, scale=2.0))      # unconditional conditioning with another k
    _err(lambda: fresh.ddim_sample(4, x, c, None, ac, eta=1.0, noise=torch.zeros(3, B, 3, 16, 16, device=d)))   # noise stack shorter than the loop
    z, _, _ = fresh.ddim_sample(4, x, c, None, ac)                         # and the context is still alive
    assert torch.isfinite(z).all()


def test_no_device_memory_growth_across_calls_and_contexts():
    """hipMemGetInfo at the same point of three context lifetimes and over twelve rounds of every sampling entry (DDIM, VQ decode, online
    and bulk kNN, RARM sampling): the library's scratch is sized once and reused, and rdm_ctx_destroy gives everything back."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import rdm_amd  # noqa: F401
    from rdm_amd import _lib, packing

    def free():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0]

    levels = []
    for rep in range(3):
        ctx = _lib.Context(0); d = ctx.device
        spec = ounet.tiny_spec(); cfg = spec_to_unet_cfg(spec)
        ctx.load_unet(cfg, packing.pack("unet", cfg, ounet.synth_state_dict(ounet.param_shapes(spec), seed=1)))
        vspec = ovq.tiny_vq_spec(); vcfg = spec_to_vq_cfg(vspec)
        ctx.load_vq(vcfg, packing.pack("vq", vcfg, ounet.synth_state_dict(ovq.vq_param_shapes(vspec), seed=5)))
        rs = orarm.RarmSpec(vocab_in=514, vocab_out=512, n_heads=2, d_head=64, depth=2, context_dim=512, sequence_length=64)
        rcfg = _lib.make_rarm_cfg(in_channels=514, out_channels=512, n_heads=2, d_head=64, depth=2, context_dim=512, sequence_length=64)
        ctx.load_rarm(rcfg, packing.pack("rarm", rcfg, ounet.synth_state_dict(orarm.rarm_param_shapes(rs), seed=7)))
        g = torch.Generator(device=d).manual_seed(0)
        db = torch.randn(50000, 512, device=d, generator=g).half(); ctx.db_load(db); del db
        x = torch.randn(4, 3, 16, 16, device=d, generator=g); c = torch.randn(4, 4, 512, device=d, generator=g); uc = torch.zeros_like(c)
        ac = torch.linspace(0.9999, 0.005, 1000)
        marks = []
        for it in range(12):
            z = ctx.ddim_sample(4, x, c, uc, ac, scale=2.0)[0]
            img = ctx.vq_decode(z)
            idx, sc = ctx.knn(torch.randn(200 if it % 2 else 7, 512, device=d, generator=g), 20 if it % 2 else 4)
            toks = ctx.rarm_sample(torch.full((3, 1), 513, device=d, dtype=torch.long), c[:3], 16, torch.rand(16, 3, device=d, generator=g),
                                   top_k=50, guidance_scale=2.0)
            del z, img, idx, sc, toks
            torch.cuda.empty_cache()
            marks.append(free())
        assert marks[1] - marks[-1] < (4 << 20), f"device memory grows across calls: {marks}"      # after the second round nothing is allocated
        levels.append(marks[-1])
        ctx.close(); del ctx, x, c, uc
        torch.cuda.empty_cache()
    assert abs(levels[0] - levels[2]) < (4 << 20), f"a closed context leaves device memory behind: {levels}"


def test_token_and_code_ids_out_of_range_are_refused(fresh):
    """nn.Embedding (RetrievalPatchTransformer.proj_in) and taming's get_codebook_entry raise on an id outside the table; the
    kernels would read row 0 instead: the binding checks the range (advisor finding, round 2)."""
    from rdm_amd import _lib, packing
    d = fresh.device
    rs = orarm.RarmSpec(vocab_in=514, vocab_out=512, n_heads=2, d_head=64, depth=2, context_dim=512, sequence_length=64)
    rcfg = _lib.make_rarm_cfg(in_channels=514, out_channels=512, n_heads=2, d_head=64, depth=2, context_dim=512, sequence_length=64)
    fresh.load_rarm(rcfg, packing.pack("rarm", rcfg, ounet.synth_state_dict(orarm.rarm_param_shapes(rs), seed=7)))
    ctx2 = torch.zeros(2, 2, 512, device=d)
    assert "[0, 514)" in _err(lambda: fresh.rarm_forward(torch.full((2, 3), 514, dtype=torch.long, device=d), ctx2))
    assert "[0, 514)" in _err(lambda: fresh.rarm_sample(torch.full((2, 1), -1, dtype=torch.long, device=d), ctx2, 4, torch.rand(4, 2, device=d), top_k=8))
    fresh.rarm_forward(torch.full((2, 3), 513, dtype=torch.long, device=d), ctx2)          # the last valid id passes


def test_single_op_entry_points_reject_unsupported_shapes(fresh):
    """The single-kernel ops added in round 3 state their shape contract instead of running something else."""
    d = fresh.device
    bf = lambda *s: torch.zeros(*s, device=d, dtype=torch.bfloat16)
    # token-major-V flash attention: n must be a multiple of 64
    assert "multiple of 64" in _err(lambda: fresh.op_self_attention_qkv(bf(1, 32, 3 * 64), 2))
    # fused cross-attention: C % 64, softmax group in {1, 2, 4}, LayerNorm form takes no separate residual
    G, U = bf(1, 128, 96), bf(1, 96, 128)
    assert "unsupported shape" in _err(lambda: fresh.op_xattn_fused(bf(1, 32, 96), G, U, None, None, 12, 4))
    G, U = bf(1, 128, 64), bf(1, 64, 128)
    assert "unsupported shape" in _err(lambda: fresh.op_xattn_fused(bf(1, 32, 64), G, U, None, None, 6, 3))
    g = torch.ones(64, device=d)
    assert "res must be null" in _err(lambda: fresh.op_xattn_fused(bf(1, 32, 64), G, U, None, bf(1, 32, 64), 8, 4, ln=(g, g, 1e-5)))
    # fused head: W % 32, C <= 240
    w = torch.zeros(3, 64, 3, 3, device=d)
    assert "unsupported shape" in _err(lambda: fresh.op_head_conv(bf(1, 8, 24, 64), w, None))
    w = torch.zeros(3, 256, 3, 3, device=d)
    assert "unsupported shape" in _err(lambda: fresh.op_head_conv(bf(1, 8, 32, 256), w, None))
    # and the context still works
    out = fresh.op_head_conv(bf(1, 8, 32, 64), torch.zeros(3, 64, 3, 3, device=d), None)
    assert out.shape == (1, 3, 8, 32) and float(out.abs().max()) == 0.0
