"""Sample-parallel multi-GPU support: one process per GPU, NO collective inside the denoise loop.

Reference: inference/sample.py:199-202 partitions seeds / classes / output indices / per-sample camera
lists rank-strided (`x[rank::world_size]`) and every rank torch.load()s both checkpoints itself
(sample.py:186,194).  Here the partition is identical, and the checkpoints are read once by rank 0 and
broadcast as ONE packed fp32 blob per model over RCCL/xGMI (torch.distributed backend "nccl"; the
reference's own precedent is trainers/utils.py:11-37 read_file_dist).  1.6 GiB at one-link xGMI speed
(~153 GB/s) is ~11 ms — start-up only.
"""
import os

import torch
import torch.distributed as dist


def rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun-style env vars (RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return rank_world()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))))
    dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=world)
    return rank_world()


def spawn_one_process_per_gpu(script, argv, nproc=None, module=False):
    """The reference's `mp.spawn(main, nprocs=torch.cuda.device_count())` (inference/sample.py:340-348): when this
    process was NOT started by a launcher (no WORLD_SIZE) and the node has more than one GPU, re-exec `script argv`
    under torch.distributed.run with one rank per GPU and return its exit code; otherwise return None and the caller
    carries on as the single rank (or as the rank torchrun made it)."""
    import subprocess
    import sys
    if "WORLD_SIZE" in os.environ:
        return None
    if nproc is None:
        nproc = torch.cuda.device_count() if torch.cuda.is_available() else 1
    if nproc <= 1:
        return None
    # torchrun's own c10d rendezvous picks the port (--standalone): no bind-then-close race with another process
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={nproc}"] + (["-m"] if module else []) + [script] + list(argv)
    env = dict(os.environ)
    if "HSA_ENABLE_IPC_MODE_LEGACY" not in env or "OMP_NUM_THREADS" not in env:
        print("[ivid_amd] starting %d ranks with HSA_ENABLE_IPC_MODE_LEGACY=%s OMP_NUM_THREADS=%s (set them to override)"
              % (nproc, env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), env.get("OMP_NUM_THREADS", "8")), file=sys.stderr)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    # random draws the reference makes ONCE in the parent and hands to every rank (classes / viewset 'random',
    # sample.py:296-336): the ranks re-draw them from this common seed, so each rank's shard is a shard of ONE list
    env.setdefault("IVID_DRAW_SEED", str(int.from_bytes(os.urandom(4), "little")))
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        print("[ivid_amd] the %d-rank launch failed (exit %d).  If torch.distributed / RCCL cannot initialise on this node, run a "
              "single rank instead: WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 python %s%s ... (or CUDA_VISIBLE_DEVICES=<one gpu>)"
              % (nproc, rc, "-m " if module else "", script), file=sys.stderr)
    return rc


def common_draw_seed():
    """One seed for the random draws that must agree on all ranks: IVID_DRAW_SEED when the launcher set it, else rank 0's
    fresh seed (exchanged with one all_gather when there is more than one rank)."""
    seed = int(os.environ["IVID_DRAW_SEED"]) if os.environ.get("IVID_DRAW_SEED") else int.from_bytes(os.urandom(4), "little")
    if rank_world()[1] > 1:
        seed = int(gather_scalars(seed)[0])
    return seed


def shard(seq, rank=None, world=None):
    """Rank-strided shard, exactly `seq[rank::world_size]` (sample.py:199-202); None passes through."""
    if seq is None:
        return None
    if rank is None:
        rank, world = rank_world()
    return seq[rank::world]


def shard_plan(num_samples, world, batchsize):
    """What the rank-strided partition (sample.py:199-202) and sample_all's batching (sample.py:56-58: batches of `batchsize`,
    the last one ragged) give every rank: [{rank, samples, batches, last_batch}].  BASELINE config 4: 10 000 samples on 8 ranks
    in batches of 32 = 1250 samples per rank = 39 full batches + one batch of 2."""
    plan = []
    for r in range(world):
        n = len(range(r, num_samples, world))
        nb = (n + batchsize - 1) // batchsize
        plan.append({"rank": r, "samples": n, "batches": nb, "last_batch": n - (nb - 1) * batchsize if nb else 0})
    return plan


def shard_views(modelviews, rank=None, world=None):
    """Per-sample camera lists are sharded, a shared camera list is not (sample.py:202)."""
    if modelviews and isinstance(modelviews[0], list):
        return shard(modelviews, rank, world)
    return modelviews


def broadcast_state_dict(schema, state_dict=None, device=None, src=0):
    """Broadcast a checkpoint from `src` to every rank as one flat fp32 message.

    schema: [(name, shape)] known on every rank (derived from the config, so no metadata exchange);
    state_dict: the loaded checkpoint on `src`, ignored elsewhere.  Returns {name: tensor} on `device`.
    """
    rank, world = rank_world()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    sizes = []
    for _, shape in schema:
        n = 1
        for s in shape:
            n *= s
        sizes.append(n)
    total = sum(sizes)
    # strict=True semantics on the source rank (missing / unexpected keys, per-tensor shapes), and the verdict is
    # exchanged BEFORE the payload so that a bad checkpoint raises on every rank instead of leaving the others hung
    # in the broadcast
    err = ""
    if rank == src:
        names = {n for n, _ in schema}
        missing = [n for n, _ in schema if n not in state_dict]
        unexpected = [k for k in state_dict if k not in names]
        bad = [n for n, shape in schema if n in state_dict and tuple(state_dict[n].shape) != tuple(shape)]
        if missing or unexpected or bad:
            err = f"checkpoint does not match the config: missing {missing[:3]}, unexpected {unexpected[:3]}, wrong shape {bad[:3]}"
    if world > 1:
        flag = torch.tensor([1 if err else 0], dtype=torch.int32, device=device)
        dist.broadcast(flag, src=src)
        if int(flag.item()) and not err:
            err = f"rank {src} rejected the checkpoint (see its error)"
    if err:
        raise KeyError(err)
    if rank == src:
        flat = torch.cat([state_dict[n].detach().reshape(-1).to(torch.float32) for n, _ in schema]).to(device)
        assert flat.numel() == total
    else:
        flat = torch.empty(total, dtype=torch.float32, device=device)
    if world > 1:
        dist.broadcast(flat, src=src)
    out, off = {}, 0
    for (name, shape), n in zip(schema, sizes):
        out[name] = flat[off:off + n].view(shape)
        off += n
    return out


def gather_scalars(value):
    """all_gather of one float per rank (timings / counters at the end of a run)."""
    rank, world = rank_world()
    if world == 1:
        return [float(value)]
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    outs = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return [float(o.item()) for o in outs]
