"""Benchmark of the SegOFA fwd+bwd hot path (BASELINE.json metric: images/sec, 512x512,
SegOFA-Base, bf16).  One "step" = criterion forward (HIP model + upsample/CE loss) +
backward (HIP) + gradient all-reduce (N>1) + clip + Adam, on one batch of 8 synthetic
images per GPU that is already resident in HBM.

    python bench.py [--gpus N --steps K --warmup W]
N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / WORLD_SIZE in
the environment) or started plainly -- bench.py then re-executes itself under torch.distributed.run with N ranks, one
per GPU over RCCL, and fails loudly when the box has fewer than N GPUs.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline":     dominant kernel family, algorithmic FLOP / measured kernel time (HIP events on
                  the launch stream inside the timed region) vs the 2.5 PFLOP/s dense bf16 MFMA peak
  "cpu_baseline": the CPU oracle (oracle/segofa_ref.py, fp32, all host cores) timed on a bounded
                  sample (4 images) of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# BASELINE.json configs measured by this file (--config): c2 is the headline one (the driver's default run)
CONFIGS = {
    "c2": dict(arch="segofa_base", nseg=15, size=512, batch=8, tag="BASELINE configs[1]: SegOFA-Base bf16", trunk="ResNet-101"),
    "c3": dict(arch="segofa_base", nseg=150, size=512, batch=8, tag="BASELINE configs[2] per-GPU share (B=8 of global 64): SegOFA-Base bf16, 150 ADE20K classes", trunk="ResNet-101"),
    "c4": dict(arch="segofa_large", nseg=171, size=640, batch=8, tag="BASELINE configs[3] per-GPU share: SegOFA-Large bf16, 171 COCO-Stuff classes", trunk="ResNet-152"),
}
# algorithmic fwd+bwd GFLOP / image, frozen ResNet (tools/count_flops.py re-derives them on the oracle)
# (tools/count_flops.py: 912.0 / 1003.2 / 5245.5 -- the last replaces SURVEY 8d's estimate 5252.8 = fwd + 2 (fwd - conv))
GF_PER_IMG = {("segofa_base", 15): 912.0, ("segofa_base", 150): 1003.2, ("segofa_large", 171): 5245.5}
MFMA_PEAK_TF = 2500.0                      # dense bf16 (MI355X_MICROARCH.md)


def cpu_baseline(nseg, src_len, arch="segofa_base", size=512):
    """SURVEY 8d / BASELINE.md section 4: the CPU restatement (oracle/segofa_ref.py, fp32) on BASELINE configs[0] inputs
    (B = 2, 512x512, frozen trunk, dropout 0), every host core, 1 warm-up + 3 timed fwd+bwd steps (a bounded sample:
    ~10-20 s of CPU work on the GPU box)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import segofa_ref as O
    # every host core up to 32 threads: the GPU box has 256 hardware threads, on which this op mix (many small fp32
    # GEMMs / elementwise ops) is pathologically slow -- measured 279 s per step with 256 threads vs ~2 s with 32
    ncores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(ncores)
    large = arch == "segofa_large"
    cfg = (O.large_config if large else O.base_config)(num_seg_tokens=nseg, patch_image_size=size, orig_patch_image_size=size)
    sd = O.procedural_state_dict(cfg)
    spec = O.state_dict_spec(cfg)
    for k, v in sd.items():
        if v.dtype.is_floating_point and "embed_images" not in k and not spec[k][1].startswith("alias"):
            v.requires_grad_(k.startswith(("encoder.layers", "decoder.layers")))
    nimg, warm, timed = (1, 0, 1) if large else (2, 1, 3)      # Large / 640: ~6x the work per image
    batch = O.synthetic_batch(cfg, nimg, src_len, image_size=size)
    dts = []
    for i in range(warm + timed):
        for v in sd.values():
            v.grad = None
        t0 = time.time()
        logits, extra = O.segofa_forward(sd, cfg, batch["src_tokens"], batch["patch_images"])
        loss, _, _ = O.seg_loss(cfg, logits, batch["target"], size // 16, size // 16, size, size)
        loss.backward()
        if i >= warm:
            dts.append(time.time() - t0)
    dt = sum(dts) / len(dts)
    return {"value": round(nimg / dt, 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%s inputs (B=%d, %dx%d, %d classes): %d warm-up + %d timed fwd+bwd steps of "
                      "oracle/segofa_ref.py, fp32, %d host threads, %.2f s per step" % (
                          "BASELINE configs[0]" if (not large and nseg == 15) else arch, nimg, size, size, nseg, warm, timed,
                          torch.get_num_threads(), dt)}


def _self_spawn(n, argv):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU"""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    shared = os.environ.get("IFSEG_DIST_BACKEND") == "gloo" and os.environ.get("IFSEG_ALLOW_SHARED_GPU") == "1"
    if ndev < n and not shared:
        sys.stderr.write("bench.py --gpus %d: only %d GPU(s) visible on this box -- one rank per GPU is required (functional "
                         "runs of N ranks on one GPU: IFSEG_DIST_BACKEND=gloo IFSEG_ALLOW_SHARED_GPU=1)\n" % (n, ndev))
        sys.exit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    sys.exit(subprocess.call(cmd, env=env))


PROF_STRIDE = 3


TRAFFIC_JSON = os.environ.get("IFSEG_TRAFFIC_JSON", "profiles/round6_hbm_traffic.json")


def _pmc_traffic(kind):
    """HBM bytes per launch of a kernel family.  NOT measured in this run: read from the committed PMC collection
    (TRAFFIC_JSON: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, gfx950 x2 correction on
    FETCH_SIZE, tools/pmc_traffic.sh); the bench line names the file in `traffic_source`.  None if not collected."""
    for rel in (TRAFFIC_JSON, "profiles/round5_hbm_traffic.json"):
        try:
            with open(os.path.join(ROOT, rel)) as f:
                return json.load(f)[kind]["bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            continue
    return None


def _traffic_source():
    for rel in (TRAFFIC_JSON, "profiles/round5_hbm_traffic.json"):
        path = os.path.join(ROOT, rel)
        if os.path.exists(path):
            at = None
            try:
                with open(path) as f:
                    at = json.load(f).get("_collected_at")      # the commit whose tree the PMC passes ran (stamped when the file was committed)
            except (OSError, ValueError):
                pass
            return rel + " (rocprofv3 --pmc passes of this command%s, committed; not collected in this run)" % (" at commit %s" % at if at else "")
    return None


HBM_PEAK_GBS = 8000.0


def _one_roofline(r, steps):
    """one kernel family from a profiling pass after the timed region (event pairs around every launch of the family)"""
    sec = r["ms"] * 1e-3
    if r["kind"].startswith("ln_"):         # HBM-bound row kernels: algorithmic bytes / time vs 8 TB/s
        ach = r["bytes"] / sec / 1e9 if sec > 0 else 0.0
        return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "launches": r["launches"], "avg_launch_us": round(r["ms"] * 1e3 / max(1, r["launches"]), 2),
                "ms_per_step": round(r["ms"] / max(1, steps), 3), "traffic": _pmc_traffic(r["kind"])}
    ach = r["flops"] / sec / 1e12 if sec > 0 else 0.0
    return {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TF, 4),
            "launches": r["launches"], "avg_launch_us": round(r["ms"] * 1e3 / max(1, r["launches"]), 2),
            "ms_per_step": round(r["ms"] / max(1, steps), 3), "traffic": _pmc_traffic(r["kind"])}


def _group_roofline(rs, steps):
    ms = sum(r["ms"] for r in rs)
    nl = sum(r["launches"] for r in rs)
    ach = sum(r["flops"] for r in rs) / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    tr = [_pmc_traffic(r["kind"]) for r in rs]
    return {"bound": "mfma", "kernel": "gemm_kernel", "instantiations": [r["kind"] for r in rs], "achieved": round(ach, 2),
            "peak": MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TF, 4),
            "traffic": round(sum(t * r["launches"] for t, r in zip(tr, rs)) / max(1, nl)) if all(t is not None for t in tr) else None,
            "launches": nl, "avg_launch_us": round(ms * 1e3 / max(1, nl), 2), "ms_per_step": round(ms / max(1, steps), 3),
            "measured": "separate pass of %d steps after the timed region (per-launch events on every GEMM slow the step)" % steps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2",
                    help="BASELINE.json configuration: c2 = configs[1] (headline, default), c3 = configs[2] (150 classes, L=215) "
                         "and c4 = configs[3] (Large, 640x640, 171 classes) at their per-GPU size")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default: the configuration's, 8)")
    ap.add_argument("--nseg", type=int, default=None, help="override the configuration's class count")
    ap.add_argument("--steady-steps", type=int, default=200,
                    help="steps of the steady-state cross-check run right after the timed region (0: skip)")
    ap.add_argument("--dropout", type=float, default=0.1, help="dropout (coco_unseen.sh:22)")
    ap.add_argument("--drop-path", type=float, default=0.1, help="encoder/decoder drop-path rate (coco_unseen.sh:20-21)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefetch", action="store_true", help="run the frozen trunk in line instead of one batch ahead")
    ap.add_argument("--graph", action="store_true",
                    help="replay the HIP-graph-captured update (Trainer.train_step(graph=True): captured once per resident "
                         "batch) instead of enqueueing ~800 launches per step from the host.  Off by default: the step is "
                         "GPU-bound (host enqueue 11.6 ms vs 20.5 ms of GPU time) and hipGraph replays the four-stream step "
                         "with less cross-branch concurrency than the streams themselves (measured 20.8-32 ms vs 20.5 ms)")
    ap.add_argument("--image-free", action="store_true",
                    help="the recipe's image-free step (SURVEY 8f row 1, coco_unseen.sh:51): loss on an artificial image "
                         "(EmbeddingBag patches, no trunk) + a no-grad pass over the real images for the metrics")
    ap.add_argument("--lab", action="store_true",
                    help="laboratory run: accept IFSEG_LAB=1 (the gate of every A/B switch, ifseg_amd/lab.py); the line is "
                         "marked \"lab\": true and is not a record")
    a = ap.parse_args()
    # a record must not come from a run that leaves work out or runs a non-default path: experiment switches (IFSEG_EXP_*:
    # kernels skipped, bias work compiled out) and the laboratory gate (IFSEG_LAB=1, without --lab) abort the run, and every
    # IFSEG_* variable that was set is printed into the line (config.env)
    env_seen = {k: v for k, v in sorted(os.environ.items()) if k.startswith("IFSEG_")}
    exp = [k for k in env_seen if k.startswith("IFSEG_EXP_") or k in ("IFSEG_RING_ABLATE",)]
    if os.environ.get("IFSEG_LAB") == "1" and not a.lab:
        exp.append("IFSEG_LAB")
    if exp:
        sys.stderr.write("bench.py: experiment / laboratory switch(es) %s set -- not the product's default step; refusing to report a number\n" % ", ".join(exp))
        sys.exit(3)
    C = CONFIGS[a.config]
    if a.batch is None:
        a.batch = C["batch"]
    if a.nseg is None:
        a.nseg = C["nseg"]
    arch, size = C["arch"], C["size"]
    gf_img = GF_PER_IMG.get((arch, a.nseg))

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        _self_spawn(a.gpus, sys.argv[1:])
    if int(os.environ.get("WORLD_SIZE", "1")) != a.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %s rank(s)\n" % (a.gpus, os.environ.get("WORLD_SIZE", "1")))
        sys.exit(2)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    local = local % max(1, ndev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        backend = os.environ.get("IFSEG_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)                       # functional test on a 1-GPU box
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from ifseg_amd import hip, lab
    from ifseg_amd.criterions import SegCriterion
    from ifseg_amd.tasks.mm_tasks import SegmentationTask
    from ifseg_amd.trainer import Trainer
    hip.lib().ifseg_experimental_build.restype = ctypes.c_int
    if hip.lib().ifseg_experimental_build():
        sys.stderr.write("bench.py: %s was compiled with a measurement switch that leaves work out (ifseg_experimental_build() = %d); "
                         "refusing to report a number\n" % (os.environ.get("IFSEG_LIB", "libifseg_hip.so"), hip.lib().ifseg_experimental_build()))
        sys.exit(3)

    torch.manual_seed(0)
    task = SegmentationTask(num_seg_tokens=a.nseg, patch_image_size=size, arch=arch)
    model = task.build_model()
    model.cfg.dropout, model.cfg.encoder_drop_path_rate, model.cfg.decoder_drop_path_rate = a.dropout, a.drop_path, a.drop_path
    crit = SegCriterion(task, unsupervised_segmentation=a.image_free, init_seg_with_text=False)
    trainer = Trainer(model, crit, task, device=dev, lazy_logs=True)
    # RING different synthetic batches, cycled: the batches of the next calls are handed to the trainer as `prefetch` (what
    # a buffered data iterator holds ahead), so their frozen-trunk pass runs underneath this step on a second stream -- one
    # pass per IFSEG_TRUNK_LOOKAHEAD (default 2) batches; every batch of the timed region goes through the trunk exactly once
    RING, AHEAD = 4, 4
    ring = []
    for j in range(RING):
        sm = task.synthetic_sample(a.batch, dev, seed=1234 + rank + 7919 * j)
        sm["net_input"]["patch_images"] = sm["net_input"]["patch_images"].to(torch.bfloat16)
        if a.image_free:
            sm.update(task.synthetic_aux_sample(a.batch, dev, seed=4321 + rank + 7919 * j))
        ring.append(sm)
    sample = ring[0]
    step_no = [0]

    use_graph = [False]

    def one_step():
        i = step_no[0]
        step_no[0] += 1
        nxt = None if a.no_prefetch else [ring[(i + k) % RING] for k in range(1, AHEAD + 1)]
        return trainer.train_step([ring[i % RING]], prefetch=nxt, graph=use_graph[0])

    def sync():
        torch.cuda.synchronize()        # (the direct RCCL collectives of a step are stream-ordered: drained before the c10d barrier)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # discovery pass: every family timed over the last TWO warm-up steps -- one trunk pass serves two batches, so a single step
    # either contains a whole pass (conv family counted twice) or none; per-step values are the two-step sums halved
    n_disc = (1 if a.no_prefetch else max(1, trainer.eng.trunk_lookahead))
    n_disc = n_disc if a.warmup >= n_disc else 1
    for i in range(a.warmup):
        if i == a.warmup - n_disc:
            hip.prof_reset(); hip.prof_enable(0x1FF)
        one_step()
    torch.cuda.synchronize()
    fam = [hip.prof_read(k) for k in range(len(hip.PROF_KINDS))]
    for f in fam:
        f["ms"] /= n_disc
    # the `roofline` object goes to the largest family BY MEASURED KERNEL TIME of this run (discovery pass above).  The
    # attention backward is one operation executed as two concurrent kernels (dK/dV and dQ): they count as one family
    # (sum of their kernel times); every family keeps its own entry in roofline_gemm_kernel / roofline_other_kernels.
    kinds = list(hip.PROF_KINDS)
    fam_ms = {n: f["ms"] for n, f in zip(kinds, fam)}
    merged = {n: m for n, m in fam_ms.items() if n not in ("attn_bwd_dkv", "attn_bwd_dq")}
    merged["attn_bwd_dkv"] = fam_ms.get("attn_bwd_dkv", 0.0) + fam_ms.get("attn_bwd_dq", 0.0)
    dominant = kinds.index(max(merged, key=merged.get)) if a.warmup > 0 else kinds.index("attn_bwd_dkv")
    # the dK/dV and dQ kernels of the attention backward run side by side on two streams and share the GPU: they are
    # timed as ONE unit (delta + dK/dV + dQ, an event pair on the main stream around the three launches)
    pair = hip.PROF_KINDS[dominant] in ("attn_bwd_dkv", "attn_bwd_dq") and trainer.eng.overlap
    graphed = world == 1 and a.graph and not a.image_free and a.warmup >= 3
    hip.prof_reset()
    hip.prof_enable(0)

    def dominant_pass(nsteps):
        """event pairs around the dominant kernel (every PROF_STRIDE-th launch), on eagerly enqueued steps"""
        if pair:
            trainer.eng.attn_bwd_timing = {"stride": PROF_STRIDE, "seen": 0, "pairs": []}
        else:
            hip.prof_enable(1 << dominant, stride=PROF_STRIDE)   # an event pair costs two queue packets
        for _ in range(nsteps):
            one_step()
        torch.cuda.synchronize()

    if graphed:
        # the whole update is captured once per resident batch (two graphs: the batches alternate) and replayed in the
        # timed region; kernel timing events cannot live inside a graph, so the dominant kernel is timed on eagerly
        # enqueued steps right after the timed region
        use_graph[0] = True
        for _ in range(4):
            one_step()                        # 2 captures + 2 replays
    else:
        dominant_pass(0)
    sync()
    t0 = time.time()
    for _ in range(a.steps):
        logs = one_step()
    t_enq = time.time() - t0          # the host has ENQUEUED every step (it runs ahead of the GPU unless launches are the limit)
    n_marks = len(trainer.eng.marks) if trainer.eng.marks is not None else 0
    sync()
    dt = time.time() - t0
    tt = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = tt.item()
    loss = float(logs[-1]["loss"])
    # steady-state cross-check of the (short) driver-timed region: the same step, >= 200 times / >= 3 s, measured the same
    # way right after it -- long enough for an external utilisation sampler to see a busy GPU
    steady = None
    if a.steady_steps > 0:
        n_ss, t_ss = 0, 0.0
        sync()
        t1 = time.time()
        while n_ss < a.steady_steps or (time.time() - t1) < 3.0:
            for _ in range(20):
                one_step()
            n_ss += 20
            if n_ss >= 4000:
                break
        sync()
        t_ss = time.time() - t1
        ts = torch.tensor([t_ss], device=dev)
        if world > 1:
            dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        t_ss = ts.item()
        steady = {"steps": n_ss, "seconds": round(t_ss, 3), "ms_per_step": round(t_ss / n_ss * 1e3, 3),
                  "images_per_sec": round(a.batch * world * n_ss / t_ss, 2)}
    # what one step costs the HOST: the enqueue of a step into an EMPTY GPU queue (device drained first), so that no
    # back-pressure from a full queue is inside the figure -- `host_enqueue_ms_per_step` of the timed region converges to the
    # GPU time whenever the host is ahead.  Median of 7, after the timed regions.
    host_empty = []
    for _ in range(7):
        torch.cuda.synchronize()
        th = time.time()
        one_step()
        host_empty.append((time.time() - th) * 1e3)
    torch.cuda.synchronize()
    host_empty.sort()
    if graphed:
        use_graph[0] = False
        dominant_pass(min(10, a.steps))
    if pair:
        prs = trainer.eng.attn_bwd_timing["pairs"]
        trainer.eng.attn_bwd_timing = None
        tr = [_pmc_traffic(k) for k in ("attn_bwd_dkv", "attn_bwd_dq")]
        dom = {"kind": "attn_bwd (delta + dK/dV kernel + dQ kernel, the two side by side on two streams)",
               "ms": sum(a_.elapsed_time(b_) for a_, b_, _ in prs), "flops": sum(f for _, _, f in prs), "launches": len(prs),
               "traffic": (tr[0] + tr[1]) if all(t is not None for t in tr) else None}
    else:
        dom = hip.prof_read(dominant)
        dom["traffic"] = _pmc_traffic(dom["kind"])
    hip.prof_enable(0)
    # second view, outside the timed region (timing ~300 launches per step with events costs ~8 % of the step): the four
    # GEMM-shaped families are instantiations of ONE kernel, gemm_kernel (csrc/gemm.hip); together they are the largest
    # time consumer, so their aggregate MFMA rate is reported next to the dominant single family
    gk = [k for k, n in enumerate(hip.PROF_KINDS) if n.startswith("gemm_") or n == "conv"]
    hip.prof_reset(); hip.prof_enable(sum(1 << k for k in gk))
    extra = min(3, a.steps)
    for _ in range(extra):
        one_step()
    torch.cuda.synchronize()
    grs = [hip.prof_read(k) for k in gk]
    # third view: every other family on its own (attention forward / dQ / dK,dV, LayerNorm), same method
    ok_ = [k for k in range(len(hip.PROF_KINDS)) if k not in gk]
    hip.prof_reset(); hip.prof_enable(sum(1 << k for k in ok_))
    for _ in range(extra):
        one_step()
    torch.cuda.synchronize()
    ors = [hip.prof_read(k) for k in ok_]
    hip.prof_enable(0)
    rccl_stats = trainer.reducer.stats()
    if world > 1 and rccl_stats["mode"] != "direct" and lab.get("REDUCE_MODE") is None:
        raise RuntimeError("bench.py: %d ranks but the gradient all-reduce did not run through the direct RCCL communicator (%r)"
                           % (world, rccl_stats))
    if world > 1 and rccl_stats["ranks"] != world:
        raise RuntimeError("bench.py: the RCCL communicator spans %d ranks, WORLD_SIZE is %d" % (rccl_stats["ranks"], world))
    if rank == 0:
        imgs = a.batch * world * a.steps
        value = imgs / dt
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
        out = {
            "metric": "images/sec (%dx%d, SegOFA-%s fwd+bwd)" % (size, size, "Large" if arch == "segofa_large" else "Base"),
            "value": round(value, 2), "unit": "images/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("IMAGE-FREE step (not the headline config) -- " if a.image_free else "") + C["tag"] + ", batch %d/GPU, %dx%d, %d classes (L=%d), "
                                   "frozen %s trunk%s, dropout %.2f / drop-path %.2f; step = fwd + upsample/CE loss + bwd + clip + Adam%s"
                                   % (a.batch, size, size, a.nseg, task.src_len, C["trunk"],
                                      "" if a.no_prefetch else " (one pass per %d future batches on a second stream; a ring of %d synthetic batches, every batch through the trunk exactly once)" % (trainer.eng.trunk_lookahead, RING),
                                      a.dropout, a.drop_path,
                                      "; every step is one replay of the HIP-graph-captured update" if graphed else "; steps enqueued from the host"),
                       "global_batch": a.batch * world, "parallelism": "dp%d" % world, "loss": round(loss, 4),
                       "env": env_seen},
            "roofline": {"bound": "mfma", "kernel": dom["kind"], "achieved": round(ach, 2), "peak": MFMA_PEAK_TF,
                         "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TF, 4), "traffic": dom["traffic"],
                         "traffic_source": _traffic_source(),
                         "launches": dom["launches"], "avg_launch_us": round(dom["ms"] * 1e3 / max(1, dom["launches"]), 2),
                         "sampled": "every %d. launch timed with a HIP event pair on its stream%s" % (
                             PROF_STRIDE, " (on %d eagerly enqueued steps after the timed graph replays)" % min(10, a.steps) if graphed else ""),
                         "whole_step_frac": round(value / world * gf_img / 1e3 / MFMA_PEAK_TF, 4) if gf_img else None,
                         "gflop_per_image": gf_img},
            **({"lab": True} if os.environ.get("IFSEG_LAB") == "1" else {}),
            "roofline_gemm_kernel": _group_roofline(grs, extra),
            "roofline_other_kernels": {r["kind"]: _one_roofline(r, extra) for r in ors},
            "kernel_families_ms_per_step": {f["kind"]: round(f["ms"], 3) for f in fam},
        }
        # the data-parallel leg as it actually ran (live from the reducer): N ranks in ONE RCCL communicator, how many
        # collectives and bytes per step.  Any N > 1 run that did not go through the direct communicator fails loudly below.
        out["rccl"] = rccl_stats
        out["host_enqueue_ms_per_step"] = round(t_enq / a.steps * 1e3, 3)
        out["host_ms_per_step_empty_queue"] = round(host_empty[len(host_empty) // 2], 3)
        if trainer.eng.marks:                    # IFSEG_PHASE_TIMING=1: main-stream (and host) time between the phase marks
            torch.cuda.synchronize()
            allm = trainer.eng.marks[:n_marks]
            per_step = max(1, len(allm) // max(1, sum(1 for m in allm if m[0] == "step_start")))
            mk = allm[-per_step * min(a.steps, 20):]       # the timed region's last steps
            while mk and mk[0][0] != "step_start":
                mk = mk[1:]
            ph, hp = {}, {}
            for i in range(len(mk) - 1):
                k = mk[i][0] + "->" + mk[i + 1][0]
                ph.setdefault(k, []).append(mk[i][1].elapsed_time(mk[i + 1][1]))
                hp.setdefault(k, []).append((mk[i + 1][2] - mk[i][2]) * 1e3)
            out["phase_ms"] = {k: round(sum(v) / len(v), 3) for k, v in ph.items()}
            out["phase_host_ms"] = {k: round(sum(v) / len(v), 3) for k, v in hp.items()}
        if trainer.eng.drain_timing:             # IFSEG_DRAIN_TIMING=1: main-stream wait for the side queue at the end of the backward
            torch.cuda.synchronize()
            w = [t0.elapsed_time(t1) for t0, t1 in trainer.eng.drain_timing[-a.steps:]]
            out["end_of_backward_wait_ms"] = round(sum(w) / len(w), 3)
        if steady is not None:
            out["steady_state"] = steady
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(a.nseg, task.src_len, arch, size)
        print(json.dumps(out))
    trainer.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
