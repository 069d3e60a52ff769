"""Multi-GPU sampling: one process per GPU, batch sharded, ONE collective (all-gather of finished images).

The reference has no multi-GPU sampling; samples are independent once their conditioning row
exists, so the batch is split contiguously over ranks, weights and the fp16 DB are replicated, and the only
exchange is an all-gather of the decoded images (RCCL over xGMI on GPUs — backend "nccl" — or gloo in the CPU
tests).  Per-sample RNG streams are a function of (seed, GLOBAL sample index) so results do not depend on the
number of ranks.  Used by `MinimalRETRODiffusion.set_distributed()` (sample_with_query / sample_from_rdata),
`scripts/rdm_sample.py --gpus N` and `bench.py`.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend: str = None):
    """Join the process group described by the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    -> (rank, local_rank).  Backend: RCCL ("nccl") when a HIP device is visible, else gloo.  Must be called before
    the library context is created so that each rank binds its own device."""
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    # test switches for a one-GPU box: every rank on one device, gloo group (RCCL refuses two ranks on one GPU)
    backend = backend or os.environ.get("RDM_DIST_BACKEND") or None
    if os.environ.get("RDM_DIST_DEVICE") is not None:
        local = int(os.environ["RDM_DIST_DEVICE"])
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, local


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def world_rank(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def shard_range(n_total: int, world: int, rank: int):
    """Contiguous balanced split: the first n_total % world ranks get one extra sample."""
    q, r = divmod(n_total, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def shared_seed(device="cpu", group=None) -> int:
    """One draw from torch's global CPU generator, rank 0's value on every rank: the per-call base seed of the
    per-sample noise streams (follows `seed_everything`, so a seeded run repeats and an unseeded one does not)."""
    s = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        s = s.to(device)
        dist.broadcast(s, src=0, group=group)
    return int(s.item())


def broadcast_int_array(arr, src: int = 0, device="cpu", group=None):
    """numpy integer array -> rank `src`'s copy on every rank (same shape everywhere).  Used for anything drawn from a PROCESS-local
    generator that all ranks must agree on (pseudo-query ids from numpy's global RNG: without a seed every rank draws differently)."""
    import numpy as np
    a = np.asarray(arr)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return a
    t = torch.from_numpy(a.astype(np.int64)).to(device)
    if t.is_cuda and dist.get_backend(group) == "gloo":
        t = t.cpu()
    dist.broadcast(t, src=src, group=group)
    return t.cpu().numpy().astype(a.dtype)


def profiler_attached() -> bool:
    """True when a profiler / preloaded tool (rocprofv3, LD_PRELOAD) has very likely initialised the GPU before main():
    starting another program from here would be the exec-after-GPU-init that takes the node down on this pool."""
    import os
    if any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ):
        return True
    pre = os.environ.get("LD_PRELOAD", "")
    return any(t in pre for t in ("rocprof", "roctracer", "rocp", "rocm-sdk"))


def safe_self_launch(script: str, gpus: int, argv, master_port: str = None) -> int:
    """One process per GPU through torch.distributed.run, started as a CHILD before anything in this process touches the GPU.
    Refused under a profiler (profile the per-rank process under torchrun instead).  Returns the child's exit code."""
    import os, subprocess, sys
    if profiler_attached():
        sys.stderr.write("refusing to self-launch torch.distributed.run under a profiler / LD_PRELOAD tool: the preloaded library has "
                         "initialised the GPU, and starting another program from this process is not allowed on this pool.\n"
                         "Profile one rank instead:  python -m torch.distributed.run ... rocprofv3 ... -- python3 <script> ...\n")
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", master_port or os.environ.get("MASTER_PORT", "29541"), os.path.abspath(script)] + list(argv)
    return subprocess.call(cmd)


def per_sample_noise(seed: int, global_indices, shape, device="cpu", dtype=torch.float32):
    """Noise for each global sample index from its own generator (created ON `device`, so a GPU rank draws with the device's
    Philox stream and nothing crosses PCIe): a function of (seed, global index) only, hence invariant to the sharding."""
    device = torch.device(device)
    out = []
    for gi in global_indices:
        g = torch.Generator(device=device).manual_seed((int(seed) * 1_000_003 + int(gi)) % (2 ** 63 - 1))
        out.append(torch.randn(shape, generator=g, dtype=dtype, device=device))
    if not out:
        return torch.empty((0,) + tuple(shape), dtype=dtype, device=device)
    return torch.stack(out)


_abandoned_comm_contexts = []     # communicator contexts whose hand-shake timed out: kept alive (a helper thread may still be inside them), never used again


def attach_library_comm(ctx, group=None) -> bool:
    """Give the library its own RCCL communicator over the ranks of the torch process group (C ABI: rdm_comm_unique_id on rank 0 ->
    broadcast of the 128-byte id over the torch group -> rdm_comm_init on every rank), so that the data path's one collective -- the
    all-gather of the finished images -- is `rdm_comm_all_gather` (`all_gather_images(..., ctx=ctx)`).  Only for an RCCL ("nccl") group
    with one device per rank; on gloo (CPU tests, several ranks sharing one GPU: RCCL refuses two ranks on one device) nothing is created.
    -> True when the communicator exists ON EVERY RANK.  A failure to create it is NOT fatal: the torch.distributed collective stays in
    use (and the caller can report which one ran).

    The communicator lives on a DEDICATED sibling context (`ctx.new_comm_context()`: same device, its own non-blocking side stream),
    stored as `ctx.lib_comm` on success.  The rendezvous and a known-answer probe gather run in a helper thread with a deadline
    (RDM_LIB_COMM_TIMEOUT seconds, default 120): a rendezvous that never completes costs this path, not the run -- the product context is
    never touched by the helper, nothing is ever enqueued on the product's stream, and a sibling whose hand-shake did not finish is
    abandoned (kept referenced, never closed, never used)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1 or dist.get_backend(group) != "nccl":
        return False
    if os.environ.get("RDM_NO_LIB_COMM", "0") not in ("", "0"):          # torch.distributed's collective only
        return False
    if getattr(ctx, "lib_comm_agreed", 0) == dist.get_world_size(group):
        return True
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ok = torch.ones(1, dtype=torch.int32, device=ctx.device)
    uid = torch.zeros(128, dtype=torch.uint8, device=ctx.device)
    cctx = None
    try:
        cctx = ctx.new_comm_context() if hasattr(ctx, "new_comm_context") else ctx
        if rank == 0:
            uid = torch.frombuffer(bytearray(cctx.comm_unique_id()), dtype=torch.uint8).to(ctx.device)
    except Exception:
        ok.zero_()
    dist.broadcast(uid, src=0, group=group)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) == 0:
        if cctx is not None and cctx is not ctx and hasattr(cctx, "close"):
            cctx.close()
        return False
    uid_bytes = bytes(uid.cpu().numpy().tobytes())
    result = {"ok": False}

    def _init_and_probe():
        try:
            cctx.comm_init(uid_bytes, rank, world)
            # one known-answer gather before anything depends on it (this path cannot be exercised on the one-GPU build boxes): rank r
            # contributes [r, r + 0.5, r, r + 0.5]; any other result -> the torch.distributed collective stays in use
            side = getattr(cctx, "_side_stream", None)
            probe = torch.tensor([rank, rank + 0.5, rank, rank + 0.5], dtype=torch.float32, device=ctx.device)
            want = torch.arange(world, dtype=torch.float32, device=ctx.device)[:, None] + torch.tensor([0.0, 0.5, 0.0, 0.5], device=ctx.device)
            if side is not None:
                torch.cuda.current_stream(ctx.device).synchronize()          # the probe operands exist before the side stream reads them
            got = cctx.comm_all_gather(probe, world)
            if side is not None:
                side.synchronize()
            result["ok"] = got.shape == want.shape and bool(torch.equal(got, want))
        except Exception:
            result["ok"] = False

    import threading
    th = threading.Thread(target=_init_and_probe, daemon=True)
    th.start()
    th.join(timeout=float(os.environ.get("RDM_LIB_COMM_TIMEOUT", "120")))
    if th.is_alive() or not result["ok"]:
        ok.zero_()
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)          # all ranks agree on which collective they will call
    if int(ok.item()) == 0:
        if th.is_alive():
            _abandoned_comm_contexts.append((cctx, th))             # still inside the rendezvous: never closed, never used
        else:
            try:
                cctx.comm_destroy()
                if cctx is not ctx and hasattr(cctx, "close"):
                    cctx.close()
            except Exception:
                pass
        ctx.lib_comm_agreed = 0
        ctx.lib_comm = None
        return False
    ctx.lib_comm = cctx
    ctx.lib_comm_agreed = world                                     # (set only on the agreed flag: a late helper thread cannot switch one rank over)
    return True


def library_comm_gather(ctx, local: torch.Tensor, world: int) -> torch.Tensor:
    """[b, ...] -> [world, b, ...] through the library's communicator (`ctx.lib_comm`, the sibling context attach_library_comm created, or the
    context itself): rdm_comm_all_gather on the communicator's own non-blocking side stream, ordered after the producer of `local` on the
    current stream and before its consumers (stream waits, no host synchronisation)."""
    cc = getattr(ctx, "lib_comm", None) or ctx
    side = getattr(cc, "_side_stream", None)
    cur = torch.cuda.current_stream(local.device)
    local = local.contiguous()
    if side is not None:
        side.wait_stream(cur)
    out = cc.comm_all_gather(local, world)
    if side is not None:
        local.record_stream(side); out.record_stream(side)
        cur.wait_stream(side)
    return out


def all_gather_images(local: torch.Tensor, n_total: int = None, group=None, ctx=None) -> torch.Tensor:
    """Gather [b_rank, ...] shards into [n_total, ...] on every rank (shards may differ by one sample).
    ctx: a library context with a communicator (attach_library_comm) -> the gather is rdm_comm_all_gather (RCCL through the C ABI, on
    the library's stream) whenever the shards are equal-sized device tensors; otherwise torch.distributed."""
    if not dist.is_available() or not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    if local.is_cuda and dist.get_backend(group) == "gloo":        # gloo has no device all-gather: stage through the host
        return all_gather_images(local.cpu(), n_total, group).to(local.device)
    if n_total is None:
        n_total = local.shape[0] * world
    counts = [shard_range(n_total, world, r)[1] - shard_range(n_total, world, r)[0] for r in range(world)]
    bmax = max(counts)
    if ctx is not None and local.is_cuda and getattr(ctx, "lib_comm_agreed", 0) == world and all(c == bmax for c in counts):
        return library_comm_gather(ctx, local, world).reshape((world * bmax,) + tuple(local.shape[1:]))
    if all(c == bmax for c in counts):
        out = torch.empty((world * bmax,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    pad = torch.zeros((bmax,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


def all_gather_tensor(local: torch.Tensor, group=None) -> torch.Tensor:
    """[...] on every rank (same shape) -> [world, ...]; a gloo group takes device tensors through the host."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local[None]
    if local.is_cuda and dist.get_backend(group) == "gloo":
        return all_gather_tensor(local.cpu(), group).to(local.device)
    world = dist.get_world_size(group)
    flat = local.contiguous().reshape(1, -1)
    out = torch.empty((world, flat.shape[1]), device=local.device, dtype=local.dtype)      # concatenation along dim 0 (gloo and RCCL)
    dist.all_gather_into_tensor(out, flat, group=group)
    return out.reshape((world,) + tuple(local.shape))


def merge_sharded_topk(idx_global: torch.Tensor, score: torch.Tensor, k: int, group=None):
    """Row-sharded database ("when memory matters"): every rank holds the exact top-k_l of ITS rows for all queries
    -- idx_global int64 [B,k_l] (already offset by the shard's first row; padding = 2^62), score fp64 [B,k_l] (padding = -inf).
    ONE exchange (all-gather of the pairs), then the same total order as the single-GPU search: score descending, ties to the
    lower global index.  -> (idx int64 [B,k], score fp64 [B,k]) identical on every rank."""
    pairs = torch.stack([idx_global.to(torch.float64), score.to(torch.float64)], dim=0)      # indices < 2^53: exact in fp64
    allp = all_gather_tensor(pairs, group)                                                  # [world, 2, B, k_l]
    idx = allp[:, 0].permute(1, 0, 2).reshape(idx_global.shape[0], -1).to(torch.int64)
    sc = allp[:, 1].permute(1, 0, 2).reshape(idx_global.shape[0], -1)
    o1 = torch.argsort(idx, dim=1, stable=True)                                             # secondary key first ...
    idx, sc = torch.gather(idx, 1, o1), torch.gather(sc, 1, o1)
    o2 = torch.argsort(sc, dim=1, descending=True, stable=True)                             # ... then the stable primary sort
    return torch.gather(idx, 1, o2)[:, :k].contiguous(), torch.gather(sc, 1, o2)[:, :k].contiguous()


def average_gradients(grads, bucket_bytes: int = 64 << 20, group=None):
    """Data-parallel gradient averaging for the training step (what DistributedDataParallel does under the reference's
    Lightning trainer): the fp32 gradients of `grads` (dict name -> tensor, identical keys / shapes on every rank) are flattened in
    sorted-name order into buckets of <= bucket_bytes, each bucket ONE all-reduce (RCCL over xGMI on the GPU box: few, large,
    per-link-bound ring collectives instead of one per tensor), then divided by the world size and scattered back in place.
    Deterministic: the bucket layout depends on names and shapes only.  No-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return grads
    world = dist.get_world_size(group)
    names = sorted(grads)
    i = 0
    while i < len(names):
        j, size = i, 0
        while j < len(names) and (j == i or size + grads[names[j]].numel() * 4 <= bucket_bytes):
            size += grads[names[j]].numel() * 4
            j += 1
        flat = torch.cat([grads[n].reshape(-1).float() for n in names[i:j]])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat /= world
        off = 0
        for n in names[i:j]:
            k = grads[n].numel()
            grads[n] = flat[off:off + k].reshape(grads[n].shape).to(grads[n].dtype)
            off += k
        i = j
    return grads
