"""CPU, world_size 2 over gloo: rank-strided sharding and the one-shot checkpoint broadcast."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import common as C


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, C.ROOT)
    from ivid_amd import parallel
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    parallel.init_from_env("gloo")
    schema = C.schema_for(C.MINI)
    sd = C.synth_weights(C.MINI, 0) if rank == 0 else None
    got = parallel.broadcast_state_dict(schema, sd, device=torch.device("cpu"))
    ref = C.synth_weights(C.MINI, 0)
    same = all(torch.equal(got[k], ref[k]) for k in ref)
    seeds = list(range(9))
    views = [[i, i + 100] for i in range(9)]
    mine = parallel.shard(seeds)
    times = parallel.gather_scalars(1.5 + rank)
    q.put((rank, same, mine, parallel.shard_views(views), parallel.shard_views([1, 2, 3]), times))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_broadcast():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert res[0][2] == [0, 2, 4, 6, 8] and res[1][2] == [1, 3, 5, 7]          # seeds[rank::2] (sample.py:199)
    assert sorted(res[0][2] + res[1][2]) == list(range(9))                      # disjoint cover
    assert res[0][3] == [[0, 100], [2, 102], [4, 104], [6, 106], [8, 108]]      # per-sample views sharded
    assert res[0][4] == [1, 2, 3] and res[1][4] == [1, 2, 3]                    # shared camera list not sharded
    assert res[0][5] == [1.5, 2.5] and res[1][5] == [1.5, 2.5]


def test_single_process_paths():
    from ivid_amd import parallel
    assert parallel.rank_world() == (0, 1)
    assert parallel.shard([1, 2, 3]) == [1, 2, 3] and parallel.shard(None) is None
    sd = C.synth_weights(C.MINI_UNCLASS, 3)
    out = parallel.broadcast_state_dict(C.schema_for(C.MINI_UNCLASS), sd, device=torch.device("cpu"))
    assert all(torch.equal(out[k], sd[k]) for k in sd)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without torchrun must come up as TWO ranks that see each other through a collective
    (here gloo on CPU; on the GPU box the same path runs over RCCL) and must refuse a mismatching WORLD_SIZE."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    bench = os.path.join(C.ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--launcher-dry-run"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["self_launched"] is True
    assert [p[0] for p in out["ranks_seen"]] == [0, 1]
    # an external launcher with the wrong world size is an error, not a silent single-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    r2 = subprocess.run([sys.executable, bench, "--gpus", "2", "--launcher-dry-run"], env=env2, capture_output=True, text=True,
                        timeout=600)
    assert r2.returncode != 0 and "WORLD_SIZE" in (r2.stderr + r2.stdout)


def test_bench_n1_prints_the_same_record_with_and_without_a_launcher():
    """The driver's scaling run starts N = 1 under `torch.distributed.run --nproc-per-node 1` while its plain bench run starts
    `python bench.py`: both must come up as the same single rank and print a record with the same keys, so that SCALE's N = 1
    point can be checked against BENCH.  (CPU: the launcher dry run; the timed path needs an MI355X.)"""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    bench = os.path.join(C.ROOT, "bench.py")
    plain = subprocess.run([sys.executable, bench, "--gpus", "1", "--launcher-dry-run"], env=env, capture_output=True, text=True, timeout=600)
    assert plain.returncode == 0, plain.stderr[-2000:]
    launched = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                               "127.0.0.1", "--master-port", "29731", bench, "--gpus", "1", "--launcher-dry-run"], env=env,
                              capture_output=True, text=True, timeout=600)
    assert launched.returncode == 0, launched.stderr[-2000:]
    a, b = (json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]) for r in (plain, launched))
    assert set(a) == set(b) and a["n_gpus"] == b["n_gpus"] == 1 and a["ranks_seen"] == b["ranks_seen"] == [[0, -1]]
    assert a["config4_full_size_plan"] == b["config4_full_size_plan"]


def test_spawn_helper_is_a_no_op_under_a_launcher_or_on_one_device():
    from ivid_amd import parallel
    assert parallel.spawn_one_process_per_gpu("x.py", [], nproc=1) is None
    old = os.environ.get("WORLD_SIZE")
    os.environ["WORLD_SIZE"] = "4"
    try:
        assert parallel.spawn_one_process_per_gpu("x.py", [], nproc=8) is None
    finally:
        if old is None:
            del os.environ["WORLD_SIZE"]
        else:
            os.environ["WORLD_SIZE"] = old


def _bad_ckpt_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, C.ROOT)
    from ivid_amd import parallel
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    parallel.init_from_env("gloo")
    schema = C.schema_for(C.MINI)
    sd = None
    if rank == 0:
        sd = C.synth_weights(C.MINI, 0)
        sd["out.2.bias"] = torch.zeros(7)              # wrong shape
        sd["not.a.parameter"] = torch.zeros(1)         # unexpected key
        del sd["time_embed.1.bias"]                    # missing key
    try:
        parallel.broadcast_state_dict(schema, sd, device=torch.device("cpu"))
        q.put((rank, "no error"))
    except KeyError as e:
        q.put((rank, str(e)))
    dist.barrier()
    dist.destroy_process_group()


def test_bad_checkpoint_raises_on_every_rank_instead_of_hanging():
    """strict load semantics across ranks: missing / unexpected keys and wrong shapes are detected on the source
    rank and the verdict is exchanged BEFORE the payload, so the other ranks raise too instead of waiting in the broadcast."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_bad_ckpt_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert "missing ['time_embed.1.bias']" in res[0] and "not.a.parameter" in res[0] and "out.2.bias" in res[0]
    assert "rejected the checkpoint" in res[1]


def _c4_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, C.ROOT)
    from ivid_amd import parallel
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    parallel.init_from_env("gloo")
    seeds = list(range(10000))
    mine = parallel.shard(seeds)
    batches = [mine[i:i + 32] for i in range(0, len(mine), 32)]            # sample_all's batching (sample.py:56-58)
    secs = parallel.gather_scalars(100.0 + rank)                            # per-rank clocks as bench.py --config c4 reports them
    seed = parallel.common_draw_seed()                                      # one seed for the draws all ranks must share
    q.put((rank, len(mine), len(batches), len(batches[-1]), mine[:2], mine[-1], secs, seed))
    dist.barrier()
    dist.destroy_process_group()


def test_config4_shard_arithmetic_world_size_8_with_the_ragged_last_batch():
    """BASELINE config 4: 10 000 samples sharded over 8 ranks (seeds[rank::8], sample.py:199-202) in batches of 32: every
    rank gets 1250 samples = 39 full batches + ONE batch of 2.  Dry run over gloo with 8 processes: the shards are a disjoint
    cover, the plan bench.py prints matches what the ranks really get, per-rank timings and the common draw seed arrive
    everywhere."""
    from ivid_amd import parallel
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_c4_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    plan = parallel.shard_plan(10000, 8, 32)
    for r, (rank, n, nb, last, first2, lastseed, secs, seed) in enumerate(res):
        assert rank == r and (n, nb, last) == (1250, 40, 2) == (plan[r]["samples"], plan[r]["batches"], plan[r]["last_batch"])
        assert first2 == [r, r + 8] and lastseed == 9992 + r
        assert secs == [100.0 + k for k in range(8)]
        assert seed == res[0][7]
    assert sum(p["samples"] for p in plan) == 10000
    # ragged totals: 10 001 samples leave rank 0 one more sample; a world that does not divide the batch count
    p2 = parallel.shard_plan(10001, 8, 32)
    assert [p["samples"] for p in p2] == [1251] + [1250] * 7 and p2[0]["last_batch"] == 3
    assert parallel.shard_plan(5, 8, 32)[5] == {"rank": 5, "samples": 0, "batches": 0, "last_batch": 0}
