"""CPU, world_size 2 over gloo: the flat-arena gradient reducer of the bundled trainer
(the N>1 path of bench.py) -- coverage of every element exactly once, overlap-order
independence, and the reference's gradient scaling (trainer.py:874-879 with sample_size 1
per rank => mean over ranks)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ifseg_amd.trainer import ArenaReducer
    n = 10_000
    # layer slices with gaps before / between / after (top-level tensors), like the real arena
    slices = {"encoder.": (0, 700), "encoder.layers.0.": (700, 3000), "encoder.layers.1.": (3000, 5200),
              "decoder.": (5200, 5600), "decoder.layers.0.": (5600, 9000)}
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(n, generator=g)
    mine = flat.clone()
    red = ArenaReducer(flat, slices, n)
    # backward order: decoder layers first, then "decoder.", encoder layers, "encoder."
    for p in ("decoder.layers.0.", "decoder.", "encoder.layers.1.", "encoder.layers.0.", "encoder."):
        red.on_ready(p)
    red.finish()
    gathered = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(gathered, mine)
    expect = sum(gathered)
    ok = torch.allclose(flat, expect, atol=1e-6)
    # scaling: sum over ranks * (world / sum(sample_size)) / world with sample_size == 1 per rank -> mean
    mean = flat * (1.0 / world)
    ok = ok and torch.allclose(mean, expect / world, atol=1e-6)
    if rank == 0:
        out.put(bool(ok))
    dist.destroy_process_group()


def _worker_fp32(rank, world, port, out):
    """bf16 arena, fp32-accumulate option: every slice is summed in fp32 over the ranks and rounded to bf16 once"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ifseg_amd.trainer import ArenaReducer
    n = 6000
    slices = {"encoder.": (0, 500), "encoder.layers.0.": (500, 3000), "decoder.layers.0.": (3100, 5800)}
    g = torch.Generator().manual_seed(7 + rank)
    flat = torch.randn(n, generator=g).to(torch.bfloat16)
    mine = flat.float()
    red = ArenaReducer(flat, slices, n, fp32_accumulate=True)
    assert red.fp32
    for p in ("decoder.layers.0.", "encoder.layers.0.", "encoder."):
        red.on_ready(p)
    red.finish()
    gathered = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(gathered, mine)
    expect = sum(gathered).to(torch.bfloat16)              # fp32 sum of the ranks' bf16 values, ONE rounding
    if rank == 0:
        out.put(bool(torch.equal(flat.view(torch.int16), expect.view(torch.int16))))
    dist.destroy_process_group()


def test_arena_reducer_fp32_accumulate_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fp32, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_arena_reducer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _worker_logs(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ifseg_amd.trainer import Trainer
    t = Trainer.__new__(Trainer)                      # only the cross-rank sum of the logging outputs is exercised
    t.world, t.device = world, torch.device("cpu")
    hist = torch.arange(5, dtype=torch.float32) * (rank + 1)
    logs = [{"loss": torch.tensor(1.5 + rank), "ntokens": 100 + rank, "nsentences": 2, "sample_size": 1,
             "area_intersect": hist, "area_union": hist + 1, "note": "strings stay per-rank"},
            {"loss": torch.tensor(0.25), "ntokens": 7, "nsentences": 2, "sample_size": 1, "area_intersect": hist * 2,
             "area_union": hist * 2 + 1, "note": "x"}]
    red = t._sync_logs(logs)
    r = range(world)
    ok = abs(float(red[0]["loss"]) - sum(1.5 + k for k in r)) < 1e-6 and red[0]["ntokens"] == sum(100 + k for k in r)
    ok = ok and red[0]["sample_size"] == world and isinstance(red[0]["ntokens"], int) and red[0]["note"] == logs[0]["note"]
    ok = ok and torch.equal(red[0]["area_intersect"], sum(torch.arange(5.) * (k + 1) for k in r))
    ok = ok and torch.equal(red[1]["area_union"], sum(torch.arange(5.) * (k + 1) * 2 + 1 for k in r))
    # mIoU from the summed histograms (criterions/seg_criterion.py:533-572)
    from ifseg_amd.criterions import SegCriterion
    agg = SegCriterion.reduce_metrics([dict(red[0], imfree_loss=0.0, seg_loss=0.0, nll_loss=0.0,
                                            area_pred_label=red[0]["area_union"], area_label=red[0]["area_union"])])
    ok = ok and abs(agg["mIoU"] - float(torch.nanmean(red[0]["area_intersect"] / red[0]["area_union"]))) < 1e-4
    if rank == 0:
        out.put(bool(ok))
    dist.destroy_process_group()


def test_logging_outputs_and_histograms_are_summed_over_ranks():
    """trainer.py:1368-1406 / fairseq/distributed/utils.py:654-700: one all-reduce over every numeric logging entry,
    including the 4 x nseg area histograms the mIoU is computed from."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_logs, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
