"""N>1 path on CPU: world_size-2 gloo processes exercise the sharding + single-gather code of rfx.dist."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """An unused TCP port from the kernel (a pid-derived port collided now and then with a socket of an earlier run still in TIME_WAIT)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "ransac-flow_amd"))
    import torch.distributed as dist
    from rfx import dist as rdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_items, h8, w8 = 6, 3, 4
    mine = rdist.shard_indices(n_items, rank, world)
    res = []
    for i in mine:  # fake per-pair results: H = i * ones, flow = i + arange
        ok = (i != 3)
        res.append(dict(H=torch.full((3, 3), float(i)) if ok else None,
                        flowDown=torch.arange(2 * h8 * w8, dtype=torch.float32).view(1, 2, h8, w8) + i))
    rec = rdist.pack_records(res)
    assert rec.shape == (len(mine), rdist.record_width(h8, w8))
    out = rdist.gather_records(rec, dist)
    q.put((rank, out.clone().numpy()))          # by value: a tensor travels as a file descriptor the parent may ask for after this process has exited
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_single_gather_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = {r: torch.from_numpy(o) for r, o in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "ransac-flow_amd"))
    from rfx import dist as rdist
    assert torch.equal(outs[0], outs[1])                       # every rank holds the full result set
    perm = rdist.unshard_order(6, world)
    assert sorted(perm) == list(range(6))
    g = outs[0]
    for k, item in enumerate(perm):
        if item == 3:
            assert g[k, 9] == 1 and g[k, :9].abs().sum() == 0   # failed pair: status 1, zero H
        else:
            assert g[k, 9] == 0 and (g[k, :9] == item).all()
        assert g[k, 10] == item                                # first flow element = item id
    assert rdist.shard_indices(7, 1, 3) == [1, 4]
    assert torch.equal(rdist.gather_records(g, None), g)       # single process: identity


def _pipelined_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "ransac-flow_amd"))
    import torch.distributed as dist
    from rfx import dist as rdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pg = rdist.PipelinedGather(dist)
    got = []
    for step in range(4):                       # every step a FRESH block (as bench.py's step allocates one): [rank, step, row]
        rec = torch.stack([torch.tensor([float(rank), float(step), float(i)]) for i in range(3)])
        prev = pg.push(rec)
        assert (prev is None) == (step == 0)
        if prev is not None:
            got.append(prev.clone())
    got.append(pg.flush().clone())
    assert pg.flush() is None
    q.put((rank, torch.stack(got).numpy()))      # by value (see _worker)
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_gather_returns_every_steps_block_one_step_late_world2():
    """rfx.dist.PipelinedGather (bench.py's timed loop since round 6): the all_gather of step k is started asynchronously and
    collected when step k+1 pushes -- every step's block arrives complete, rank-major, in step order, on both ranks; the last one
    comes out of flush().  A single process runs the same protocol without a collective."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipelined_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = {r: torch.from_numpy(o) for r, o in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert torch.equal(outs[0], outs[1]) and outs[0].shape == (4, 6, 3)
    for step in range(4):
        blk = outs[0][step]
        assert blk[:, 1].eq(step).all() and blk[:, 0].tolist() == [0, 0, 0, 1, 1, 1] and blk[:, 2].tolist() == [0, 1, 2] * 2
    sys.path.insert(0, os.path.join(ROOT, "ransac-flow_amd"))
    from rfx import dist as rdist
    pg = rdist.PipelinedGather(None)
    a, b = torch.ones(2, 3), torch.zeros(2, 3)
    assert pg.push(a) is None and pg.push(b) is a and pg.flush() is b and pg.flush() is None


def test_bench_rank_code_self_spawns_two_gloo_ranks():
    """bench.py's REAL rank code path -- `python bench.py --gpus 2` with no launcher environment must re-exec itself under
    torch.distributed.run, build the process group, run warm-up + timed steps between barriers, gather the records with the
    single all_gather, take the max time over ranks and print ONE JSON line with n_gpus = 2 -- rehearsed on CPU with
    --dry-run (gloo, a stand-in step that fabricates rank-tagged records; everything around the kernels is the code the
    driver's N > 1 runs execute)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1",
                          "--batch", "4"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout                          # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["config"]["ranks_seen_in_gather"] == [0, 1] and j["config"]["gathered_records"] == 8
    assert j["value"] > 0 and abs(j["value"] - 4 * 3 * 2 / (j["ms_per_step"] * 3e-3)) < 1e-2 * j["value"]
    assert "preflight ok: gloo, 2 rank(s)" in out.stderr        # init + 1 MB all_gather + barrier before the workload is built
    # first-N>1-run hygiene: the line explains itself -- ranks in the group, every rank's own clock, the per-rank
    # CPU thread cap, and the score chunk rank 0 resolved and broadcast (all ranks sum their scores the same way)
    c = j["config"]
    assert c["n_ranks_in_process_group"] == 2 and c["backend"] == "gloo" and len(c["ms_per_step_per_rank"]) == 2
    assert max(c["ms_per_step_per_rank"]) == pytest.approx(j["ms_per_step"], rel=1e-3)
    assert c["cpu_threads_per_rank"] >= 1 and c["score_chunk_products"] in (128, 192, 256, 384, 512, 1024)
    assert ("host probe" in c["score_chunk_source"]) or ("fallback" in c["score_chunk_source"])
    assert out.stderr.count("score chunk: %d products" % c["score_chunk_products"]) == 2     # both ranks hold rank 0's value
    # a single rank stays a single process and says so
    out1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "2", "--warmup", "0", "--batch", "3"],
                          capture_output=True, text=True, timeout=120, env=env)
    j1 = json.loads([ln for ln in out1.stdout.splitlines() if ln.startswith("{")][0])
    assert j1["n_gpus"] == 1 and j1["config"]["ranks_seen_in_gather"] == [0] and j1["config"]["gathered_records"] == 3


def test_bench_preflight_diagnoses_a_missing_rank_instead_of_hanging():
    """A rank whose peers never arrive (here: WORLD_SIZE=2 with only rank 0 started) must fail within the preflight limit with the
    one-line diagnosis (stage, backend, visible devices, IPC mode, rendezvous), not hang until the driver's clock runs out."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RFX_PREFLIGHT_TIMEOUT="5")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode != 0
    assert "[bench preflight FAILED] rank 0/2 at init_process_group" in out.stderr, out.stderr[-1500:]
    assert "HSA_ENABLE_IPC_MODE_LEGACY=" in out.stderr and "MASTER=127.0.0.1:%d" % port in out.stderr


def test_committed_profiles_carry_the_kernel_names_bench_prints():
    """The driver's line names its dominant kernel with bench.kernel_name(kernel id) and reads that kernel's HBM traffic from the
    committed PMC summary (bench.pmc_traffic): both are only worth something while the committed rocprofv3 / PMC evidence was
    taken with the SAME kernel instances (round 3's qs summary predated a template-argument change and went stale).  Every
    instance with a share of the config-3 step must be found, by the name bench prints, in profiles/r06_rocprofv3_kernel_stats_
    config3.csv and in profiles/r06_pmc_summary_config3.json; the kernel ids come from the committed bench line itself."""
    import csv
    import json
    sys.path.insert(0, ROOT)
    import bench
    line = [l for l in open(os.path.join(ROOT, "profiles", "r06_bench_default.log")) if l.startswith("{")][-1]
    roof = json.loads(line)["roofline"]
    rows = [ln for ln in open(os.path.join(ROOT, "profiles", "r06_rocprofv3_kernel_stats_config3.csv")) if not ln.startswith("#")]   # "# command" line
    stats = [r["Name"].replace(" ", "") for r in csv.DictReader(rows)]
    pmc = [k.replace(" ", "") for k in json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_summary_config3.json")))["kernels"]]
    checked = 0
    for kid, o in roof["conv_kernels"].items():
        if o["time_share"] < 0.02:
            continue
        name = bench.kernel_name(int(kid)).replace(" ", "")
        assert any(name in s for s in stats), (kid, name)
        assert any(p.startswith(name) for p in pmc), (kid, name)
        checked += 1
    assert checked >= 6
    assert roof["kernel"].replace(" ", "") == bench.kernel_name(65538).replace(" ", "")     # the 128-channel split 3x3 instance (round 6)
    t, src = bench.pmc_traffic(roof["kernel"], "3")
    assert abs(t - roof["traffic"]) < 0.02 * t and src.endswith("r06_pmc_summary_config3.json")    # (the summary was re-taken once more after the line)
