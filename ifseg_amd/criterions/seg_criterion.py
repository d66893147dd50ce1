"""seg_criterion -- caller of the hot path (mirror of criterions/seg_criterion.py).

Supervised branch of the reference criterion (seg_criterion.py:188-192 ->
compute_loss :269-347 -> upsample_logits :237-244, compute_metric :349-362,
reduce_metrics mIoU :533-572): bilinear upsample of the per-patch logits to pixel
resolution, masked cross entropy, per-class area histograms.  ``sample_size`` is 1 per
rank (``ntokens = 1``, :345) exactly like the reference, so after the trainer's
``multiply_grads(world / sum(sample_size))`` the gradient is the mean over ranks.

The loss math is one fused HIP kernel pair (csrc/loss.hip: upsample + CE + gradient +
histograms) whenever the image is exactly 16x the feature grid and
label smoothing is off; otherwise the same math runs as stock PyTorch ops (eval on odd
image sizes).  ``compute_loss_torch`` is kept as the in-framework reference of the kernel.
"""
from dataclasses import dataclass, field
from typing import Optional

import torch
import torch.nn.functional as F

from .. import hip
from ..registry import CriterionBase, DataclassBase, register_criterion, str_bool


def _f(default, help=""):
    return field(default=default, metadata={"help": help})


@dataclass
class SegCriterionConfig(DataclassBase):
    """criterions/seg_criterion.py:32-101 (the 'true' / 'false' string flags included)."""
    label_smoothing: float = _f(0.0, "epsilon for label smoothing, 0 means no label smoothing")
    report_accuracy: bool = _f(False, "report accuracy metric")
    ignore_prefix_size: int = _f(0, "Ignore first N tokens")
    ignore_eos: bool = _f(True, "Ignore eos token")
    sentence_avg: bool = _f(False, "optimization.sentence_avg")
    drop_worst_ratio: float = _f(0.0, "ratio for discarding bad samples")
    drop_worst_after: int = _f(0, "steps for discarding bad samples")
    use_rdrop: bool = _f(False, "use R-Drop")
    reg_alpha: float = _f(1.0, "weight for R-Drop")
    sample_patch_num: int = _f(196, "sample patches for v1")
    constraint_range: Optional[str] = _f(None, "constraint range")
    upscale_lprobs: str = _f("true", "true | false")
    unsupervised_segmentation: str = _f("true", "true | false")
    criterion_update_freq: int = _f(1, "update frequency used in this criterion")
    freeze_embedding_iter: int = _f(-1, "Freeze the token embedding after this iteration (ignored if -1)")
    full_context_alignment: str = _f("false", "whether to apply full attention in decoder")
    init_seg_with_text: str = _f("true", "whether to lazy initialize the segmentation with text embedding bags")
    resnet_topk: int = _f(3, "filtering with topk adjacent resnet features")
    resnet_prob_temperature: float = _f(1.0, "resnet softmax temperature")
    resnet_iters: int = _f(0, "resnet filtering iterations")


class _FusedSegLossFn(torch.autograd.Function):
    """loss = mean CE(bilinear_x16(logits[:, :P]), target); backward hands out the gradient the
    forward kernel already produced."""

    @staticmethod
    def forward(ctx, logits, logits_pad, target, hp, wp, H, W, nseg, seg0, bufs, eps=0.0):
        B = logits_pad.shape[0]
        dev = logits_pad.device
        n_tiles = B * hp * wp
        nstat = 2 + 3 * nseg
        key = (B, hp, wp, nseg, logits_pad.shape[2], dev)
        if bufs.get("key") != key:
            bufs.clear()
            bufs["key"] = key
            bufs["tile"] = torch.empty(n_tiles * 9 * nseg, dtype=torch.float32, device=dev)
            bufs["sp"] = torch.empty(n_tiles * nstat, dtype=torch.float32, device=dev)
            bufs["stats"] = torch.empty(nstat, dtype=torch.float32, device=dev)
            bufs["dl"] = torch.empty_like(logits_pad)
            bufs["loss"] = torch.empty(1, dtype=torch.float32, device=dev)
            bufs["bad"] = torch.zeros(1, dtype=torch.int32, device=dev)
        hip.seg_loss(logits_pad, target, hp, wp, H, W, nseg, seg0, bufs["tile"], bufs["sp"], bufs["stats"],
                     bufs["dl"], bufs["loss"], bad_label=bufs["bad"], label_smoothing=float(eps))
        ctx.dl = bufs["dl"]
        ctx.nseg = nseg
        # fresh tensors: callers keep them in logging outputs across calls (update_freq > 1, validation loops)
        return bufs["loss"][0].clone(), bufs["stats"].clone()

    @staticmethod
    def backward(ctx, gloss, gstats):
        g = ctx.dl[:, :, : ctx.nseg]
        return g * gloss.to(g.dtype), None, None, None, None, None, None, None, None, None, None

PAD, EOS = 1, 2
FUSED_MAX_CLASSES = 512      # csrc/loss.hip NS_MAX


@register_criterion("seg_criterion", dataclass=SegCriterionConfig)
class SegCriterion(CriterionBase):
    def __init__(self, task=None, sentence_avg=False, label_smoothing=0.0, ignore_prefix_size=0, ignore_eos=True,
                 report_accuracy=False, drop_worst_ratio=0, drop_worst_after=0, use_rdrop=False, reg_alpha=1.0,
                 sample_patch_num=196, constraint_range=None, upscale_lprobs="true", unsupervised_segmentation="true",
                 criterion_update_freq=1, freeze_embedding_iter=-1, full_context_alignment="false",
                 init_seg_with_text="true", resnet_topk=3, resnet_prob_temperature=1.0, resnet_iters=0,
                 num_seg_tokens=None, seg_id_offset=None):
        """Signature of the reference (seg_criterion.py:115-161; fairseq's `build_criterion` fills it from
        SegCriterionConfig by name).  `num_seg_tokens` / `seg_id_offset` stand in for `task.cfg.num_seg_tokens` and
        `task.target_dictionary.index("<seg_0>")` when the criterion is used without a task."""
        super().__init__(task)
        self.sentence_avg = sentence_avg
        self.eps = label_smoothing
        self.sample_patch_num = sample_patch_num
        self.iter = -1
        self.effective_iter = -1
        self.criterion_update_freq = criterion_update_freq
        self.upscale_lprobs = str_bool(upscale_lprobs)
        self.unsupervised_segmentation = str_bool(unsupervised_segmentation)
        self.full_context_alignment = str_bool(full_context_alignment)
        self.init_seg_with_text = str_bool(init_seg_with_text)
        # eval-time top-k neighbour smoothing on the trunk features (seg_criterion.py:93-101,197-213)
        self.resnet_topk, self.resnet_prob_temperature, self.resnet_iters = resnet_topk, resnet_prob_temperature, resnet_iters
        cfg = getattr(task, "cfg", None)
        self.num_seg = num_seg_tokens if num_seg_tokens is not None else cfg.num_seg_tokens
        self.seg_id_offset = seg_id_offset if seg_id_offset is not None else task.target_dictionary.index("<seg_0>")
        cats = getattr(cfg, "category_list", "") or ""
        self.id2rawtext = [x.strip() for x in cats.split(",")] if cats else []
        if self.id2rawtext and len(self.id2rawtext) != self.num_seg:
            raise AssertionError("category_list names %d classes, num_seg_tokens is %d" % (len(self.id2rawtext), self.num_seg))
        self.padding_idx = PAD

    def _lazy_initialization(self, sample, model, ema_model=None):
        """seg_criterion.py:373-407: every <seg_i> embedding := mean token embedding of its category name
        (EmbeddingBag over the frozen token table), written into encoder/decoder.seg_embed_tokens (and the untied
        projection).  On the device: `ifseg_embed_bag_mean` on the bf16 token table of the engine's arena."""
        if not self.init_seg_with_text:
            return
        task = self.task
        ids = getattr(task, "category_token_ids", None)
        if ids is None:
            if not self.id2rawtext:
                raise RuntimeError("init_seg_with_text: the task carries neither category_list (+ BPE) nor category_token_ids")
            ids = [task.encode_category(" %s" % x) for x in self.id2rawtext]
        ids = [torch.as_tensor(x, dtype=torch.long).reshape(-1) for x in ids]
        if len(ids) != self.num_seg:
            raise AssertionError("%d category names for %d seg tokens" % (len(ids), self.num_seg))
        avg = model.seg_tokens_from_text(ids)
        model.encoder.seg_embed_tokens.weight.data = avg
        model.decoder.seg_embed_tokens.weight.data = avg
        if ema_model is not None:
            ema_model.encoder.seg_embed_tokens.weight.data = avg
            ema_model.decoder.seg_embed_tokens.weight.data = avg
        if not model.decoder.tie_seg_projection:
            model.decoder.seg_projection.weight.data = avg
            if ema_model is not None:
                ema_model.decoder.seg_projection.weight.data = avg

    def forward(self, model, sample, update_num=0, reduce=True, ema_model=None):
        """seg_criterion.py:165-235 (lazy init on the first call, image-free / supervised train branches, eval branch)."""
        if self.iter == -1:
            self.iter = self.criterion_update_freq * update_num - 1
            self._lazy_initialization(sample, model, ema_model)
        self.iter += 1
        self.effective_iter = self.iter // self.criterion_update_freq
        if self.unsupervised_segmentation and model.training:
            # image-free training (seg_criterion.py:179-186): the loss comes from the artificial image; the real
            # images are only evaluated (no grad) for the logged metrics
            net_output = model(full_context_alignment=self.full_context_alignment, aux_input=sample["aux_input"])
            imfree_loss = self.compute_imfree_loss(model, net_output[1]["aux_output"], sample, update_num, reduce=reduce)
            loss = imfree_loss
            with torch.no_grad():
                seg_output = model(**sample["net_input"], full_context_alignment=self.full_context_alignment)
                seg_loss, metrics, ntokens = self.compute_loss(model, seg_output, sample, update_num, reduce=reduce,
                                                               bufs_name="_bufs_metrics")
        elif not model.training and sample.get("ori_semantic_seg") is not None:
            # evaluation at the original image resolution (seg_criterion.py:194-217,289-347); batch 1 like the
            # reference (`ori_semantic_seg[0]`)
            with torch.no_grad():
                net_output = model(**sample["net_input"], full_context_alignment=self.full_context_alignment)
                seg_loss, metrics, ntokens = self.compute_loss_eval(model, net_output, sample)
            loss, imfree_loss = seg_loss, seg_loss.data.new_zeros(1)
        else:
            net_output = model(**sample["net_input"], full_context_alignment=self.full_context_alignment)
            seg_loss, metrics, ntokens = self.compute_loss(model, net_output, sample, update_num, reduce=reduce)
            loss, imfree_loss = seg_loss, seg_loss.data.new_zeros(1)
        sample_size = sample["target"].size(0) if self.sentence_avg else ntokens
        logging_output = {"loss": loss.data, "imfree_loss": imfree_loss.data, "seg_loss": seg_loss.data,
                          "ntokens": sample["ntokens"], "nsentences": sample["nsentences"],
                          "sample_size": sample_size}
        logging_output.update({k: (v.data if isinstance(v, torch.Tensor) else v) for k, v in metrics.items()})
        return loss, sample_size, logging_output

    @staticmethod
    def upsample_logits(logits, hp=32, wp=32, h=512, w=512):
        """seg_criterion.py:237-244 (mmseg.ops.resize == F.interpolate)."""
        lo = logits[:, :-1]
        B, _, n = lo.shape
        lo = lo.transpose(1, 2).reshape(B, n, hp, wp)
        lo = F.interpolate(lo, size=(h, w), mode="bilinear", align_corners=False)
        lo = lo.reshape(B, n, h * w).transpose(1, 2)
        return torch.cat([lo, logits[:, -1:]], dim=1)

    def compute_imfree_loss(self, model, net_output, sample, update_num, reduce=True):
        """seg_criterion.py:246-267: CE of the bilinear-upsampled (32x32 -> 512x512 in the reference; here the
        model's grid x16) logits of the artificial image against ``text2seg_target``; eos / pad / ignore dropped."""
        scores_low, extra = net_output
        target = sample["text2seg_target"]
        hp, wp = extra["encoder_returns"]["image_embed_shape"][0]
        h, w = 16 * hp, 16 * wp
        pad = extra.get("logits_padded")
        if (pad is not None and self.num_seg <= FUSED_MAX_CLASSES and target.shape[1] == h * w + 1):
            if not hasattr(self, "_bufs_imfree"):
                self._bufs_imfree = {}
            loss, _ = _FusedSegLossFn.apply(scores_low, pad, target.contiguous(), hp, wp, h, w, self.num_seg,
                                            self.seg_id_offset, self._bufs_imfree, self.eps)
            return loss
        scores = self.upsample_logits(scores_low.float(), hp, wp, h, w)[:, :-1]
        tgt = target[:, :-1]
        scores = scores.reshape(-1, scores.shape[-1])
        tgt = tgt.reshape(-1)
        mask = (tgt != self.padding_idx) & (tgt != self.seg_id_offset + self.num_seg)
        return F.cross_entropy(scores[mask], tgt[mask] - self.seg_id_offset, label_smoothing=self.eps)

    def compute_loss_eval(self, model, net_output, sample):
        """compute_loss, ``not model.training`` (seg_criterion.py:289-347) + the top-k neighbour smoothing of
        :197-213: per-patch scores resized bilinearly to the ORIGINAL image shape, argmax, area histograms, display
        CE -- all on the device (csrc/evalops.hip), the [h*w, n] score tensor is never built."""
        scores_low, extra = net_output
        pad = extra["logits_padded"]
        hp, wp = extra["encoder_returns"]["image_embed_shape"][0]
        P, n, dev = hp * wp, self.num_seg, pad.device
        ori = sample["ori_semantic_seg"][0]
        ori = torch.as_tensor(ori)
        h, w = ori.shape[:2]
        target = ori.reshape(-1).long().to(dev) + self.seg_id_offset
        scores = hip.rows_to_f32(pad[:1], n, P)[0]
        loss, hist = hip.seg_eval(scores, hp, wp, target, h, w, self.seg_id_offset)
        f = lambda t: t.float()
        metrics = {"area_intersect": f(hist[0]), "area_pred_label": f(hist[1]), "area_label": f(hist[2]),
                   "area_union": f(hist[1] + hist[2] - hist[0]), "nll_loss": loss}
        if self.resnet_iters > 0:
            feat = extra["encoder_returns"]["image_embed_before_proj"][0]
            prob = hip.neighbour_smoothing(pad[:1], n, feat[:1], self.resnet_iters, self.resnet_topk,
                                           self.resnet_prob_temperature)
            extra["resnet_postprocess_probability"] = torch.cat([prob, prob.new_zeros(1, 1, n)], dim=1)
            _, hpp = hip.seg_eval(prob[0], hp, wp, target, h, w, self.seg_id_offset)
            metrics.update({"area_intersect_resnet_postprocess": f(hpp[0]), "area_pred_label_resnet_postprocess": f(hpp[1]),
                            "area_label_resnet_postprocess": f(hpp[2]),
                            "area_union_resnet_postprocess": f(hpp[1] + hpp[2] - hpp[0])})
        return loss, metrics, 1

    def compute_loss(self, model, net_output, sample, update_num, reduce=True, bufs_name="_bufs"):
        scores_low, extra = net_output
        target = sample["target"]
        hp, wp = extra["encoder_returns"]["image_embed_shape"][0]
        h, w = sample["net_input"]["patch_images"].shape[-2:]
        pad = extra.get("logits_padded")
        if (pad is not None and self.upscale_lprobs and h == 16 * hp and w == 16 * wp
                and self.num_seg <= FUSED_MAX_CLASSES and target.shape[1] == h * w + 1):
            if not hasattr(self, bufs_name):
                setattr(self, bufs_name, {})
            loss, stats = _FusedSegLossFn.apply(scores_low, pad, target.contiguous(), hp, wp, h, w, self.num_seg,
                                                self.seg_id_offset, getattr(self, bufs_name), self.eps)
            # the loss kernel flags labels that are neither a class nor pad / eos / ignore (read back without a sync)
            model.engine.deferred_check(
                getattr(self, bufs_name)["bad"], lambda t: ((t[0] != 0).clone(), t.zero_())[0],
                "seg_criterion: target label outside [<seg_0>, <seg_%d>] (F.cross_entropy: target out of bounds)" % self.num_seg,
                exc=IndexError, native=(hip.CHECK_FLAG, 0, 1))
            n = self.num_seg
            ai, ap, al = stats[2:2 + n], stats[2 + n:2 + 2 * n], stats[2 + 2 * n:2 + 3 * n]
            metrics = {"area_intersect": ai, "area_pred_label": ap, "area_label": al, "area_union": ap + al - ai,
                       "nll_loss": loss}
            return loss, metrics, 1
        return self.compute_loss_torch(model, net_output, sample, update_num, reduce)

    def compute_loss_torch(self, model, net_output, sample, update_num, reduce=True):
        scores_low, extra = net_output
        scores_low = scores_low.float()
        target = sample["target"]
        hp, wp = extra["encoder_returns"]["image_embed_shape"][0]
        h, w = sample["net_input"]["patch_images"].shape[-2:]
        scores = self.upsample_logits(scores_low, hp, wp, h, w)
        mask = (target == self.padding_idx) | (target == self.seg_id_offset + self.num_seg) | (target == EOS)
        t = target[~mask] - self.seg_id_offset
        s = scores[~mask]
        metrics = dict(zip(("area_intersect", "area_pred_label", "area_label", "area_union"),
                           self.compute_metric(s.detach(), t.detach())))
        loss = F.cross_entropy(s, t, label_smoothing=self.eps)
        metrics["nll_loss"] = loss
        return loss, metrics, 1

    @staticmethod
    def compute_metric(lprobs, target):
        """seg_criterion.py:349-362."""
        n = lprobs.size(-1)
        pred = lprobs.argmax(-1)
        inter = pred[pred == target]
        a_i = torch.histc(inter.float(), bins=n, min=0, max=n - 1)
        a_p = torch.histc(pred.float(), bins=n, min=0, max=n - 1)
        a_l = torch.histc(target.float(), bins=n, min=0, max=n - 1)
        return a_i, a_p, a_l, a_p + a_l - a_i

    @classmethod
    def reduce_metrics(cls, logging_outputs):
        """seg_criterion.py:414-572: sums over workers / micro-batches, losses per sample_size, aAcc / mIoU / mAcc from
        the summed area histograms (nanmean over classes).  Logged through fairseq's `metrics` when it is importable
        (same keys); the aggregate is also returned as a dict for the bundled harness."""
        tot = lambda k: sum(l.get(k, 0) for l in logging_outputs)
        ss = tot("sample_size")
        out = {k: float(tot(k)) / max(float(ss), 1e-9) for k in ("loss", "imfree_loss", "seg_loss", "nll_loss")}
        out.update(ntokens=tot("ntokens"), nsentences=tot("nsentences"), sample_size=ss)
        areas = {}
        for suf in ("", "_resnet_postprocess", "_lowres"):
            if "area_intersect" + suf not in logging_outputs[0]:
                continue
            ai, ap, al, au = (tot("area_%s%s" % (k, suf)) for k in ("intersect", "pred_label", "label", "union"))
            areas[suf] = (ai, ap, al, au)
            out["aAcc" + suf] = round(float(ai.sum() / ap.sum()), 4)
            out["mIoU" + suf] = round(float(torch.nanmean(ai / au)), 4)
            out["mAcc" + suf] = round(float(torch.nanmean(ai / al)), 4)
        try:
            from fairseq import metrics
        except ImportError:
            return out
        ntok = out["ntokens"]
        metrics.log_scalar("loss", out["loss"], ss, round=3)
        for k in ("imfree_loss", "seg_loss", "nll_loss"):
            metrics.log_scalar(k, out[k], ntok, round=3)
        for k in ("ntokens", "nsentences", "sample_size"):
            metrics.log_scalar(k, out[k], 1, round=3)
        for suf, (ai, ap, al, au) in areas.items():
            for k, v in (("intersect", ai), ("pred_label", ap), ("label", al), ("union", au)):
                metrics.log_scalar_sum("_area_%s%s" % (k, suf), v, 1)
            f = lambda a, b, s=suf: (lambda m: round(float(torch.nanmean(m["_area_%s%s" % (a, s)].sum / m["_area_%s%s" % (b, s)].sum)), 4))
            metrics.log_derived("aAcc" + suf, lambda m, s=suf: round(float(m["_area_intersect" + s].sum.sum() / m["_area_pred_label" + s].sum.sum()), 4))
            metrics.log_derived("mIoU" + suf, f("intersect", "union"))
            metrics.log_derived("mAcc" + suf, f("intersect", "label"))
        return out

    @staticmethod
    def logging_outputs_can_be_summed():
        return True
