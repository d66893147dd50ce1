"""Minimal train-step harness standing in for the reference's train.py / trainer.py on a
box without fairseq (last row).  One process per GPU.

train_step semantics reproduced (trainer.py:745-1050):
  seed = seed + num_updates (:1297) -> criterion(model, sample) -> backward ->
  gradient SUM over ranks (DDP, distributed_fairseq_model.py:57-67) ->
  multiply_grads(world / sum(sample_size)) (:874-879; sample_size == 1 per rank, so the
  result is the mean over ranks) -> clip_grad_norm(1.0) (:886) -> Adam(0.9, 0.999,
  eps 1e-8, decoupled wd 0.1) on fp32 masters (fp16_optimizer.py) -> cosine LR (:962).

MI355X realisation: gradients live in one flat bf16 arena (engine.g16); RCCL all-reduce
is issued per layer slice from inside the backward (engine.grad_ready_hook) so it rides
the xGMI links while the remaining backward kernels run; scaling, clipping, Adam and the
bf16 write-back are ONE kernel over the arena with the norm kept on the device.
"""
import contextlib
import math

import torch
import torch.distributed as dist

from . import hip, lab


def layer_slices(eng):
    """prefix -> (lo, hi) element range of the gradient arena (layers are contiguous)"""
    sl = {}
    for n in eng.trainable_names():
        parts = n.split(".")
        key = ".".join(parts[:3]) + "." if parts[1] == "layers" else parts[0] + "."
        if (parts[0], parts[1]) in (("encoder", "token_rel_pos_table_list"), ("encoder", "image_rel_pos_table_list"),
                                    ("decoder", "seg_rel_pos_table_list")):
            # a layer's rel-pos tables sit right behind it in the arena and are final when its hook fires (the engine casts
            # their accumulators into the gradient arena before `_notify`): part of the layer's slice -- as top-level tensors
            # they split every pair of neighbouring layers, and no two layer slices ever merged into one bucket.  (The decoder's
            # token / image tables never receive a gradient and live at the end of the arena: top-level.)
            key = "%s.layers.%s." % (parts[0], parts[2])
        lo, hi = eng.offs[n], eng.offs[n] + math.prod(eng.shapes[n])
        a, b = sl.get(key, (lo, hi))
        sl[key] = (min(a, lo), max(b, hi))
    return sl


def optimizer_plan(eng):
    """[(key, [(lo, hi), ...])] -- the trainable arena cut into the slices the forward first reads, in that order:
    "g0": everything outside the transformer layers except the token table (embedding LayerNorms, position tables and their
    projections, image_proj, ... and EVERY encoder layer's rel-pos tables: the forward gathers them for all layers before
    layer 0), "emb": the token table (the largest tensor: 42 % of Base), "e<l>": encoder layer l, "rest": the decoder.
    Every element of [0, n_train) lies in exactly one range (tests/test_ddp_gloo.py::test_optimizer_plan_*)."""
    key_of = {}
    for n in eng.trainable_names():
        parts = n.split(".")
        if parts[0] == "encoder" and parts[1] == "layers":
            k = "e%s" % parts[2]
        elif parts[0] == "encoder" and parts[1] == "embed_tokens":
            k = "emb"
        elif parts[0] == "encoder":
            k = "g0"
        else:
            k = "rest"
        key_of[n] = k
    order = ["g0", "emb"] + ["e%d" % l for l in range(eng.cfg.enc_layers)] + ["rest"]
    ranges = {k: [] for k in order}
    # the arena pads every tensor to a multiple of 8 elements: a slice runs up to the next tensor's offset
    names = [n for n in eng.order if eng.offs[n] < eng.n_train]
    for i, n in enumerate(names):
        lo = eng.offs[n]
        hi = eng.offs[names[i + 1]] if i + 1 < len(names) else eng.n_train
        k = key_of.get(n, "rest")
        r = ranges[k]
        if r and r[-1][1] == lo:
            r[-1] = (r[-1][0], hi)
        else:
            r.append((lo, hi))
    return [(k, ranges[k]) for k in order if ranges[k]]


class ArenaReducer:
    """Gradient SUM over ranks on a flat arena, overlapped with the backward pass.

    ``on_ready(prefix)`` is called (in backward order) as soon as every gradient of a layer is
    final: the layer's contiguous slice is all-reduced asynchronously (RCCL on its own stream,
    xGMI busy while later backward kernels run).  ``finish()`` reduces whatever ranges were
    not covered (top-level tensors) and waits.  The reference gets the same effect from
    torch DDP's 25 MB buckets (distributed_fairseq_model.py:57-67)."""

    def __init__(self, flat, slices, n, fp32_accumulate=None):
        self.flat, self.slices, self.n = flat, slices, n
        self.works, self.done = [], []
        self._cnt, self._last = [0, 0], (0, 0)      # collectives / bytes of the step in flight, of the last finished step
        # fp32_accumulate (IFSEG_REDUCE_FP32=1): a bf16 ring sum over 8 ranks rounds after every hop (~3 bits of the sum);
        # with this option every slice travels and is summed in fp32 (twice the bytes on the links: 427 MB instead of
        # 213 MB for SegOFA-Base, ~0.7 ms over the full xGMI mesh) and is rounded to bf16 ONCE, after the sum.  The
        # default stays the reference's own behaviour under --bf16 (DDP reduces the bf16 gradients as they are).
        if fp32_accumulate is None:
            fp32_accumulate = lab.get("REDUCE_FP32") == "1"
        self.fp32 = bool(fp32_accumulate) and flat.dtype != torch.float32
        self._wide = []
        # Buckets: layer slices become final in reverse arena order, so adjacent ones are merged until a bucket holds
        # IFSEG_BUCKET_MB (default 48) before its all-reduce is issued -- ~5 collectives per step for SegOFA-Base instead
        # of 14.  Every torch.distributed call costs the enqueueing host thread ~0.1-0.3 ms in the middle of the backward
        # (measured on the RCCL world-1 leg: +4 ms per step with one call per layer) and xGMI rings are per-link bound:
        # fewer, larger collectives.  (torch DDP: 25 MB buckets, distributed_fairseq_model.py:57-67.)
        self.bucket = int(float(lab.get("BUCKET_MB", "48")) * (1 << 20)) // max(1, flat.element_size())
        self._open = None               # [lo, hi) of the bucket being filled
        # IFSEG_REDUCE_MODE: "direct" (default on the nccl backend: ifseg_amd/rccl.py, collectives enqueued in the calling
        # stream), "c10d" (torch.distributed's async all_reduce on its internal stream), and for measurements "none",
        # "fake-extra-stream", "fake-same-stream"
        self.mode = lab.get("REDUCE_MODE", "direct")
        self.direct = None
        if self.mode == "direct" and dist.is_initialized() and dist.get_backend() == "nccl" and flat.is_cuda:
            from .rccl import RcclComm
            import sys
            err = None
            try:
                self.direct = RcclComm(flat.device)
            except (OSError, AttributeError, RuntimeError) as exc:
                err = exc
            # the ranks AGREE on the outcome: a communicator that came up on some ranks only would leave them in
            # ncclAllReduce while the others sit in a c10d collective -- the job hangs.  One MIN over a success flag through
            # the process group that exists anyway; any failure anywhere sends every rank to torch.distributed's all_reduce.
            ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=flat.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            torch.cuda.synchronize(flat.device)
            if int(ok.item()) == 0:
                if self.direct is not None:
                    self.direct.destroy()
                sys.stderr.write("ifseg_amd: direct RCCL communicator unavailable on %s (%s): every rank falls back to "
                                 "torch.distributed's all_reduce (measured ~20 %% slower per step, DESIGN.md section 5)\n"
                                 % ("this rank" if err is not None else "another rank", err))
                self.direct, self.mode = None, "c10d"
        # gloo (functional runs: N ranks on one GPU, CPU tests) has no bf16 device reduction: staged through fp32 host
        # memory, synchronously.  The production backend is "nccl" (= RCCL over xGMI), asynchronous on its own stream.
        self.staged = dist.is_initialized() and dist.get_backend() == "gloo" and flat.is_cuda

    def stats(self):
        """what the last finished step put on the links: {"ranks", "mode", "buckets", "bytes"} (bench.py's `rccl` object)"""
        return {"ranks": self.direct.world if self.direct is not None else (dist.get_world_size() if dist.is_initialized() else 1),
                "mode": self.mode if not self.staged else "gloo-staged", "buckets": self._last[0], "bytes": self._last[1],
                "fp32_accumulate": self.fp32}

    def close(self):
        """destroy the direct communicator (after the last step: nothing of it may still be queued)"""
        if self.direct is not None:
            torch.cuda.synchronize(self.flat.device)
            self.direct.destroy()
            self.direct = None

    def _reduce(self, lo, hi):
        mode = self.mode
        self._cnt[0] += 1
        self._cnt[1] += (hi - lo) * (4 if self.fp32 else self.flat.element_size())
        if mode == "none":                                   # (measurement only: tools/rccl_phase_probe.py)
            return
        if self.staged:
            torch.cuda.current_stream().synchronize()       # the hook runs in weight-gradient-stream order
            cpu = self.flat[lo:hi].float().cpu()
            dist.all_reduce(cpu)
            self.flat[lo:hi].copy_(cpu)
        elif self.direct is not None:
            # RCCL's C API, on the CURRENT stream (the weight-gradient stream inside the backward, the main stream in
            # finish()): stream-ordered behind the kernels that produced the slice, nothing to wait for afterwards
            if self.fp32:
                wide = self.flat[lo:hi].float()
                self.direct.all_reduce_(wide)
                self.flat[lo:hi].copy_(wide)                # one rounding, after the fp32 sum
            else:
                self.direct.all_reduce_(self.flat[lo:hi])
        elif mode.startswith("fake"):
            # the stream choreography of an async c10d collective WITHOUT any collective (measurement only): an extra
            # stream waits for the caller's stream and records an end event that finish() makes the main stream wait for
            if getattr(self, "_comm", None) is None:
                self._comm = torch.cuda.current_stream() if mode == "fake-same-stream" else torch.cuda.Stream()
                self._fake_events, self._fei = [torch.cuda.Event() for _ in range(64)], 0
            ev, end = self._fake_events[self._fei % 64], self._fake_events[(self._fei + 1) % 64]
            self._fei += 2
            ev.record()
            self._comm.wait_event(ev)
            end.record(self._comm)

            class _W:
                def __init__(s_, e):
                    s_.e = e

                def wait(s_):
                    torch.cuda.current_stream().wait_event(s_.e)
            self.works.append(_W(end))
        elif self.fp32:
            wide = self.flat[lo:hi].float()
            # allocated on the calling (weight-gradient) stream, read back by finish() on whatever stream finish() runs on:
            # kept alive in `_wide` until then, and finish() tells the allocator about its stream
            self._wide.append((lo, hi, wide))
            self.works.append(dist.all_reduce(wide, async_op=True))
        else:
            self.works.append(dist.all_reduce(self.flat[lo:hi], async_op=True))

    def on_ready(self, prefix):
        if prefix not in self.slices or prefix in ("encoder.", "decoder."):
            return      # top-level tensors become final only at the end of their half of the backward
        lo, hi = self.slices[prefix]
        self.done.append((lo, hi))
        if self._open is not None and (hi == self._open[0] or lo == self._open[1]):
            self._open = [min(lo, self._open[0]), max(hi, self._open[1])]
        else:
            self._flush_open()
            self._open = [lo, hi]
        if self._open[1] - self._open[0] >= self.bucket:
            self._flush_open()

    def _flush_open(self):
        if self._open is not None:
            lo, hi = self._open
            self._open = None
            self._reduce(lo, hi)

    def finish(self):
        self._flush_open()
        cur = 0
        for lo, hi in sorted(self.done) + [(self.n, self.n)]:
            if lo > cur:
                self._reduce(cur, lo)
            cur = max(cur, hi)
        for w in self.works:
            w.wait()
        for lo, hi, wide in self._wide:
            if wide.is_cuda:
                wide.record_stream(torch.cuda.current_stream(wide.device))
            self.flat[lo:hi].copy_(wide)            # one rounding, after the fp32 sum
        self.works, self.done, self._wide = [], [], []
        self._last, self._cnt = (self._cnt[0], self._cnt[1]), [0, 0]


class Trainer:
    def __init__(self, model, criterion, task, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1,
                 clip_norm=1.0, max_update=2000, min_lr=0.0, seed=1, device=None, lazy_logs=False):
        self.model, self.criterion, self.task = model, criterion, task
        self.lr0, self.betas, self.eps, self.wd, self.clip = lr, betas, eps, weight_decay, clip_norm
        self.max_update, self.min_lr, self.seed = max_update, min_lr, seed
        self.num_updates = 0
        self._defer = False                 # clip + Adam underneath the next forward (train_step(defer_optimizer=...))
        self.lazy_logs = lazy_logs          # data-parallel runs: cross-rank log sums stay on the device (see _sync_logs)
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        model.autograd_mode = "arena"
        eng = model.engine
        if not eng.packed or eng.device != self.device:
            model.to(self.device)
            eng.pack(self.device)
        self.eng = eng
        n = eng.n_train
        self.p32 = eng.master_init                         # fp32 masters (taken over from pack())
        self.m = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.v = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.overflow = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._ovf_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._ovf_events = []
        self._graphs, self._hyper, self._last_gscale = {}, None, None
        eng.master_owned = True                            # this trainer keeps eng.master in step with eng.p16
        self.ws = torch.zeros(1024, dtype=torch.float32, device=self.device)
        self.reducer = ArenaReducer(eng.g16, layer_slices(eng), eng.n_train)
        # IFSEG_FORCE_GRAD_HOOK=1 with an initialised process group of ONE rank: the whole data-parallel leg (per-layer
        # all-reduce issued from the weight-gradient stream, finish()'s waits, the log all-reduce) runs through the real
        # backend -- how the RCCL path is exercised on a one-GPU box (tests/test_configs_gpu.py)
        self.dist_on = self.world > 1 or (dist.is_initialized() and lab.get("FORCE_GRAD_HOOK") == "1")
        if self.dist_on:
            eng.grad_ready_hook = self._on_grads_ready
        if self.world > 1:
            # same start on every rank (DDP broadcasts rank 0's parameters at construction)
            # (the frozen ResNet trunk and its FrozenBN statistics live outside the arena: broadcast too, then re-fold)
            extra = [p.data for n, p in model.named_parameters() if "embed_images" in n]
            extra += [b.data for n, b in model.named_buffers() if "embed_images" in n and b.dtype.is_floating_point]
            if self.reducer.staged:
                for t in [eng.p16.view(torch.float16), self.p32] + extra:      # gloo: no bf16 / int16
                    c = t.cpu()
                    dist.broadcast(c, 0)
                    t.copy_(c)
            else:
                for t in [eng.p16, self.p32] + extra:
                    dist.broadcast(t, 0)
            eng._pack_resnet()
            eng.refresh_frozen()

    # -- DDP over the flat arena ---------------------------------------------------------
    def _on_grads_ready(self, prefix):
        self.reducer.on_ready(prefix)

    # -- schedule -------------------------------------------------------------------------
    def get_lr(self):
        """Learning rate of the NEXT update, as the reference's cosine schedule produces it
        (fairseq/optim/lr_scheduler/cosine_lr_scheduler.py:74-152, period = total updates via `reinit`, train.py:184).
        `trainer.begin_epoch` (train.py:304 -> trainer.py:717-721,1126-1134) steps the scheduler with num_updates = 0
        before the first update and `lr_step_update` steps it after every update (trainer.py:962), so update k (1-based)
        runs at cosine(k - 1): the peak lr first.  Pinned by tests/golden/fixture_optim.npz (the reference's own
        scheduler object driven in that order)."""
        t = self.num_updates
        i = t // self.max_update                               # restarts shrink by lr_shrink = 0.1 (:139-147)
        shrink = 0.1 ** i
        lo, hi = self.min_lr * shrink, self.lr0 * shrink
        return lo + 0.5 * (hi - lo) * (1 + math.cos(math.pi * (t - i * self.max_update) / self.max_update))

    # -- cross-rank sum of the logging outputs ------------------------------------------------
    def _sync_logs(self, logs):
        """trainer.py:1368-1406 (`_fast_stat_sync_sum`) -> fairseq/distributed/utils.py:654-700: every numeric entry
        of the logging outputs (losses, ntokens, sample_size, the 4 x nseg area histograms the mIoU is computed from,
        criterions/seg_criterion.py:590-597) is summed over the ranks in ONE all-reduce."""
        if not getattr(self, "dist_on", self.world > 1):
            return logs
        keys, flat, scalars = [], [], []
        for i, lg in enumerate(logs):
            for k in sorted(lg):
                v = lg[k]
                if isinstance(v, torch.Tensor):
                    flat.append(v.detach().to(self.device, torch.float64).reshape(-1))
                    keys.append((i, k, v.numel(), v.shape, None))
                elif isinstance(v, (int, float)):
                    keys.append((i, k, 1, None, len(scalars)))
                    scalars.append(float(v))
        # the python numbers travel in ONE pinned block (an async copy): `torch.tensor(x, device=...)` per entry is a
        # pageable host-to-device copy each, and every one of them waits for the stream
        nt = sum(t.numel() for t in flat)
        if scalars:
            pin = getattr(self, "_log_pin", None)
            if pin is None or pin.shape[1] < len(scalars):
                pin = self._log_pin = torch.zeros(8, max(64, len(scalars)), dtype=torch.float64)
                if torch.device(self.device).type == "cuda":
                    pin = self._log_pin = pin.pin_memory()
                self._log_pin_i = 0
            r = self._log_pin_i = (self._log_pin_i + 1) % pin.shape[0]
            #  with lazy logs nothing in a step synchronises the host: a row of the ring is reused only after the
            # asynchronous copy that last read it has executed (one event per row)
            evs = getattr(self, "_log_pin_ev", None)
            if evs is None or len(evs) != pin.shape[0]:
                evs = self._log_pin_ev = [None] * pin.shape[0]
            if evs[r] is not None:
                evs[r].synchronize()
            pin[r, :len(scalars)] = torch.tensor(scalars, dtype=torch.float64)
            flat.append(pin[r, :len(scalars)].to(self.device, non_blocking=True))
            if torch.device(self.device).type == "cuda":
                if evs[r] is None:
                    evs[r] = torch.cuda.Event()
                evs[r].record()
        order = [(i, k, n, shape, (None if si is None else nt + si)) for i, k, n, shape, si in keys]
        buf = torch.cat(flat)
        if dist.get_backend() == "gloo" and buf.is_cuda:        # functional runs of N ranks on one GPU
            cpu = buf.cpu()
            dist.all_reduce(cpu)
            buf = cpu.to(self.device)
        elif getattr(getattr(self, "reducer", None), "direct", None) is not None:
            self.reducer.direct.all_reduce_(buf)                # RCCL on the current stream: no c10d stream inside a step
        else:
            dist.all_reduce(buf)
        out, o = [dict(lg) for lg in logs], 0
        lazy = getattr(self, "lazy_logs", False)
        # lazy: no host read-back inside the step -- every numeric entry stays a DEVICE tensor (a view of the reduced
        # buffer) and a consumer converts when it logs; reading them here drains the queue on every update and costs the
        # run-ahead of the host (measured +1.4 ms per step on the RCCL world-1 leg).  Otherwise ONE read-back for all.
        src = buf if lazy else buf.cpu()
        for i, k, n, shape, pos in order:
            if pos is None:
                v = src[o:o + n]
                o += n
                out[i][k] = v.reshape(shape).float() if lazy else v.reshape(shape).float().to(self.device)
            elif lazy:
                out[i][k] = src[pos]
            else:
                out[i][k] = int(src[pos].item()) if isinstance(logs[i][k], int) else float(src[pos].item())
        return out

    def check_overflow(self, wait=False):
        """trainer.py:895-904: non-finite gradient norm -> FloatingPointError.  The norm never leaves the device:
        the Adam kernel skips the update and raises a flag; it is read here once its event has completed (at the latest
        one step late, `wait=True` blocks)."""
        pend, self._ovf_events = self._ovf_events, []
        done = False
        for ev in pend:
            if wait:
                ev.synchronize()
            if ev.query():
                done = True
            else:
                self._ovf_events.append(ev)
        # the flag was copied to pinned host memory by the step itself (async, in stream order): reading it never
        # blocks on the queue (a `.item()` here would drain the whole step and stop the host from running ahead)
        if done and int(self._ovf_host[0]):
            # in-flight copies of the flag (later steps already enqueued) land before the pinned word is cleared, so the
            # same overflow cannot be reported twice
            for ev in self._ovf_events:
                ev.synchronize()
            self._ovf_events = []
            self._ovf_host.zero_()
            self.overflow.zero_()
            # Fatal-only semantics (the reference raises before anything advances, trainer.py:895-904): here the error
            # surfaces up to one step late -- the skipped update left the weights untouched, but num_updates, the
            # schedule, Adam's step count and the dropout seed have advanced and a further update may be applied.
            raise FloatingPointError("gradients are Nan/Inf (the update was skipped; reported one step late -- restart "
                                     "from a checkpoint rather than continuing)")

    # -- steps ------------------------------------------------------------------------------
    def _upload_hyper(self, lr, step, gscale):
        """{lr, 1 - beta1^step, 1 - beta2^step, grad_scale} -> device (async copy from a ring of pinned rows): what a
        captured step's Adam kernel reads on replay"""
        if self._hyper is None:
            self._hyper = torch.zeros(4, dtype=torch.float32, device=self.device)
            self._hyper_pin = torch.zeros(64, 4, dtype=torch.float32).pin_memory()
            self._hyper_i = 0
        i = self._hyper_i = (self._hyper_i + 1) % 64
        row = self._hyper_pin[i]
        row[0], row[1], row[2], row[3] = lr, 1.0 - self.betas[0] ** step, 1.0 - self.betas[1] ** step, gscale
        self._hyper.copy_(row, non_blocking=True)

    def _step_body(self, samples, captured=False):
        """forward + loss + backward (+ cross-rank reduction) + clip + Adam of one update, enqueue only.  `captured`: the
        calls are being recorded into a HIP graph -- per-update scalars come from device memory (`_hyper`, the engine's
        `step_dev`), nothing host-side is read back, and the trunk-prefetch stream is joined at the end."""
        logs, sample_sizes = [], []
        eng = self.eng
        accumulate = len(samples) > 1       # update_freq > 1 (trainer.py:745-830; the shipped recipe uses 1)
        hook = eng.grad_ready_hook
        if accumulate:
            # every backward rewrites the bf16 gradient arena: the micro-batch gradients are summed in fp32 and the
            # ranks are reduced once, on the sum (fairseq's no_sync on all but the last micro-batch)
            eng.grad_ready_hook = None
            if getattr(self, "_gacc", None) is None:
                self._gacc = torch.empty(eng.n_train, dtype=torch.float32, device=self.device)
            self._gacc.zero_()
        try:
            ahead = getattr(self, "_ahead", None)
            for i, sample in enumerate(samples):
                if ahead is not None:
                    # what the forward calls after this one will be handed, in order: the remaining micro-batches of this
                    # update, then the next call's samples (HipEngine._prefetch_request)
                    eng._pf_request = [q["net_input"]["patch_images"] for q in list(samples[i + 1:]) + ahead
                                       if "patch_images" in q.get("net_input", {})]
                eng.mark("step_start")
                loss, ss, lg = self.task.train_step(sample, self.model, self.criterion, None, self.num_updates)
                logs.append(lg)
                sample_sizes.append(ss)
                if accumulate:
                    self._gacc.add_(eng.g16)
        finally:
            eng.grad_ready_hook = hook
        if accumulate:
            eng.g16.copy_(self._gacc)
        total_ss = float(sum(sample_sizes))
        if self.dist_on:
            self.reducer.finish()
            total_ss *= self.world          # every rank reports sample_size 1 (seg_criterion.py:345)
        gscale = 1.0 / total_ss             # sum over ranks * (world / total) / world
        self._last_gscale = gscale
        hip.grad_sumsq(eng.g16, self.ws, self.sumsq)
        lr = self.get_lr()                  # the schedule is stepped after the update (trainer.py:1062-1066)
        step = self.num_updates + 1
        if not captured:
            self._upload_hyper(lr, step, gscale)
        if self._defer and not captured and eng.overlap and torch.device(self.device).type == "cuda":
            self._adam_deferred(lr, step, gscale)
        else:
            hip.adam_step(self.p32, eng.g16, self.m, self.v, eng.p16[: eng.n_train], lr, self.betas[0],
                          self.betas[1], self.eps, self.wd, step, gscale, self.clip, self.sumsq, self.overflow, hyper=self._hyper)
        eng.mark("adam_end")
        if captured and eng._pf is not None:
            torch.cuda.current_stream().wait_event(eng._pf["done"])     # every forked stream rejoins before the capture ends
        return logs

    def _adam_deferred(self, lr, step, gscale):
        """clip + Adam of this update on the optimizer's own stream, one launch per range of `optimizer_plan`, in the order the
        NEXT forward reads the parameters; an event per slice (HipEngine.params_pending): that forward starts while the bulk
        of the update (3 GB of HBM traffic, nothing for the matrix cores) is still running.  Element-wise the same kernel as
        the single launch: the parameters are bit-equal (test_deferred_optimizer_*).  trainer.py:865-907, optim/adam.py:45-110."""
        eng = self.eng
        if getattr(self, "_opt_stream", None) is None:
            # (no stream of its own: a fifth stream shares a hardware queue with one of the other four and serialises against it
            # -- measured 25.7 vs 17.1 ms per step; the dQ stream is idle from the end of a backward to the next one)
            self._opt_stream = eng._dq_stream_get()
            self._opt_plan = optimizer_plan(eng)
            self._opt_events = {}
        main = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(main)                  # the gradient norm and the hyper-parameter upload are queued on the main stream
        self._opt_stream.wait_event(fork)
        events = {}
        with torch.cuda.stream(self._opt_stream):
            prev = hip.set_stream(self._opt_stream.cuda_stream)
            try:
                p16 = eng.p16
                for key, ranges in self._opt_plan:
                    for lo, hi in ranges:
                        hip.adam_step(self.p32[lo:hi], eng.g16[lo:hi], self.m[lo:hi], self.v[lo:hi], p16[lo:hi], lr, self.betas[0],
                                      self.betas[1], self.eps, self.wd, step, gscale, self.clip, self.sumsq, self.overflow,
                                      hyper=self._hyper)
                    ev = self._opt_events.get(key)
                    if ev is None:
                        ev = self._opt_events[key] = torch.cuda.Event()
                    ev.record(self._opt_stream)
                    events[key] = ev
            finally:
                hip.set_stream(prev)
        events["all"] = events[self._opt_plan[-1][0]]
        eng.params_pending(events)

    def params_ready(self):
        """the calling stream waits for a deferred optimizer: before anything but the engine's forward reads the parameters
        (evaluation outside the engine, checkpoints, the fp32 masters / Adam moments of this trainer)"""
        self.eng._params_wait(None)

    def train_step(self, samples, prefetch=None, graph=False, defer_optimizer=None):
        """One update.  `prefetch`: the samples of the NEXT call(s), in order, as far as the data iterator already holds
        them: their frozen-trunk features are computed on a second stream underneath this step (HipEngine.prefetch_trunk);
        with more than one, `IFSEG_TRUNK_LOOKAHEAD` (default 2) batches go through the trunk in one pass.
        `graph=True` (one GPU, update_freq 1): the whole update -- ~800 kernel launches on four streams -- is captured
        into a HIP graph the first time a (sample, prefetch) pair of tensors is seen and REPLAYED afterwards: one graph
        launch instead of ~9 ms of host enqueue per step.  The sample tensors are the graph's static inputs: a data
        iterator copies each new batch into them (bench.py alternates two resident batches)."""
        self.check_overflow()
        self._quiesced_for_eval = False
        # defer_optimizer (default: IFSEG_DEFER_OPTIMIZER, off): clip + Adam run on their own stream underneath the NEXT
        # forward, which waits per parameter slice; the caller must not read parameters / optimizer state from another
        # stream without `params_ready()` (valid_step, close and grad_norm call it)
        self._defer = (lab.get("DEFER_OPTIMIZER", "0") == "1") if defer_optimizer is None else bool(defer_optimizer)
        eng = self.eng
        if not self.model.training:
            self.model.train()
        eng.step_seed = self.seed + self.num_updates
        self._ahead = None
        if graph and self.world == 1 and len(samples) == 1:
            logs = self._graph_step(samples[0], prefetch[0] if prefetch else None)
        else:
            if prefetch and "trunkpf" not in lab.get("EXP_SKIP", ""):
                self._ahead = list(prefetch)
            logs = self._step_body(samples)
        self.num_updates += 1
        ost = getattr(self, "_opt_stream", None) if (self._defer and eng._popt) else None
        with torch.cuda.stream(ost) if ost is not None else contextlib.nullcontext():
            self._ovf_host.copy_(self.overflow, non_blocking=True)      # (behind the optimizer that may raise the flag)
            ev = torch.cuda.Event()
            ev.record()
        self._ovf_events = self._ovf_events[-3:] + [ev]
        return self._sync_logs(logs)

    def _graph_step(self, sample, nxt):
        eng = self.eng
        key = (id(sample), id(nxt))
        ent = self._graphs.get(key)
        # per-update scalars go to the device BEFORE the replay, in stream order
        eng.upload_step_seed()
        self._upload_hyper(self.get_lr(), self.num_updates + 1, self._last_gscale or 1.0)
        if ent is None:
            if self.num_updates < 2:
                raise RuntimeError("Trainer.train_step(graph=True): run at least two eager updates first (workspaces, "
                                   "streams and the trunk prefetch of this batch must exist before the capture)")
            if nxt is not None:
                eng._pf_request = nxt["net_input"]["patch_images"]
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                logs = self._step_body([sample], captured=True)
            if eng._pf is not None:
                # the graph joins the trunk stream itself; its (captured) event must not be waited on by later steps
                eng._pf = dict(eng._pf, done=None)
            ent = self._graphs[key] = (g, logs, sample, nxt, eng._pf)     # keeps the static inputs alive
        else:
            # host-side state the enqueue code would have advanced
            self.criterion.iter += 1
            eng._pf = ent[4]
            self.criterion.effective_iter = self.criterion.iter // self.criterion.criterion_update_freq
            self.model.set_num_updates(self.num_updates)
        ent[0].replay()
        # the captured log tensors are the graph's static outputs, rewritten by every replay: hand out copies, so that a
        # caller aggregating logs over an interval (fairseq's reduce_metrics) keeps each step's values
        return [{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in lg.items()} for lg in ent[1]]

    def close(self):
        """end of training: nothing of the direct RCCL communicator may still be queued when it is destroyed, and no c10d
        collective (a checkpoint barrier, destroy_process_group) may be issued beside queued direct ones"""
        self.params_ready()
        if torch.device(self.device).type == "cuda":
            torch.cuda.synchronize(self.device)
        self.reducer.close()

    def quiesce(self):
        """before a torch.distributed collective OUTSIDE the step (validation, checkpoint barrier): the direct communicator's
        collectives are stream-ordered on this rank's streams, c10d's on its own -- drain the device first"""
        if torch.device(self.device).type == "cuda":
            torch.cuda.synchronize(self.device)

    def grad_norm(self):
        """global gradient norm of the last update, after the world/sample_size scaling (host sync: logging only)"""
        return float(self.sumsq.sqrt().item()) * self._last_gscale

    def valid_step(self, sample):
        self.params_ready()
        if self.dist_on and not getattr(self, "_quiesced_for_eval", False):
            # the first validation batch after training: c10d collectives of the evaluation (log sums, checkpoint barriers)
            # must not be issued beside queued direct-RCCL ones
            self.quiesce()
        self._quiesced_for_eval = True
        return self.task.valid_step(sample, self.model, self.criterion)
