"""segmentation task -- caller of the hot path (mirror of tasks/mm_tasks/segmentation.py).

Keeps the reference's task surface: ``@register_task("segmentation", dataclass=SegmentationConfig)`` (:100) on
``FairseqTask`` when fairseq is importable, the config dataclasses field for field (``OFAConfig``
tasks/ofa_task.py:24-87, ``SegmentationConfig`` segmentation.py:37-98, so `--bpe-dir`, `--selected-cols`,
`--prompt-prefix`, `--category-list`, ... of run_scripts/IFSeg/coco_unseen.sh parse), ``setup_task`` (:108-135:
dictionary = dict.txt + <mask> + <code_i> + <bin_i> + (nseg+1) <seg_i>), ``build_model`` (:166),
``train_step`` (:190-222), ``valid_step`` (:225-229) and the ``sample`` dict layout of
data/mm_data/segmentation_dataset.py:110-127.  Data loading (TSV / mmseg pipeline) is out of scope (SURVEY.md
section 2.1): ``load_dataset`` refuses; samples for the bundled harness are synthetic tensors of the right shapes
(SURVEY.md 8d).
"""
import os
from dataclasses import dataclass, field
from typing import Optional

import torch

from ...registry import HAVE_FAIRSEQ, DataclassBase, TaskBase, register_task

BOS, PAD, EOS = 0, 1, 2
# "what is the segmentation map of the image? object:" BPE ids
PROMPT_IDS = (99, 16, 5, 2835, 1258, 5456, 9, 5, 2274, 116, 7626, 35)


def _f(default, help=""):
    return field(default=default, metadata={"help": help})


@dataclass
class OFAConfig(DataclassBase):
    """tasks/ofa_task.py:24-87"""
    data: Optional[str] = _f(None, "comma separated path to data list; valid data are always in the last")
    selected_cols: Optional[str] = _f(None, "selected cols")
    train_selected_cols: Optional[str] = _f(None, "selected cols for train")
    eval_selected_cols: Optional[str] = _f(None, "selected cols for valid/eval")
    bpe: Optional[str] = _f("gpt2", "which bpe to use")
    bpe_dir: Optional[str] = _f(None, "bpe dir")
    max_source_positions: int = _f(1024, "max number of tokens in the source sequence")
    max_target_positions: int = _f(1024, "max number of tokens in the target sequence")
    max_src_length: int = _f(128, "the maximum src sequence length")
    max_tgt_length: int = _f(30, "the maximum target sequence length")
    code_dict_size: int = _f(8192, "code dict size")
    patch_image_size: int = _f(480, "patch image size")
    orig_patch_image_size: int = _f(256, "patch image size")
    num_bins: int = _f(1000, "number of quantization bins")
    imagenet_default_mean_and_std: bool = _f(False, "imagenet normalize")
    constraint_range: Optional[str] = _f(None, "constraint range")


@dataclass
class SegmentationConfig(OFAConfig):
    """tasks/mm_tasks/segmentation.py:37-98"""
    eval_acc: bool = _f(True, "evaluation with accuracy")
    eval_args: Optional[str] = _f("{}", "generation args as JSON string")
    eval_print_samples: bool = _f(False, "print sample generations during validation")
    uses_ema: Optional[bool] = _f(False, "whether to use ema")
    add_object: bool = _f(False, "add object to encoder")
    max_object_length: int = _f(30, "the maximum object sequence length")
    valid_batch_size: int = _f(1, "valid batch size per step")
    scst: bool = _f(False, "Self-critical sequence training")
    scst_args: str = _f("{}", "generation args for Self-critical sequence training, as JSON string")
    artificial_image_type: str = _f("random", "random | gt_seg | upsampling | none")
    prompt_prefix: str = _f("", 'could be "what is the segmentation map of the image? object:"')
    num_seg_tokens: int = _f(150, "number of seg tokens")
    category_list: str = _f("", "list of semantic category words (comma separated)")
    epoch_row_count: int = _f(-1, "if -1, disabled.")


class SizeDictionary:
    """Stand-in for the fairseq Dictionary where dict.txt is not available (the GPU box): sizes and the symbols the
    path asks for (dict.txt 50260 lines + 4 specials + <mask> + 8192 <code_i> + 1000 <bin_i> + (nseg+1) <seg_i>)."""

    def __init__(self, n_base, nseg):
        self.n_base, self.nseg = n_base, nseg

    def __len__(self):
        return self.n_base + self.nseg + 1

    def __contains__(self, sym):
        return sym == "<mask>" or sym.startswith(("<seg_", "<bin_", "<code_"))

    def pad(self):
        return PAD

    def bos(self):
        return BOS

    def eos(self):
        return EOS

    def index(self, sym):
        if sym.startswith("<seg_"):
            return self.n_base + int(sym[5:-1])
        raise KeyError(sym)


@register_task("segmentation", dataclass=SegmentationConfig)
class SegmentationTask(TaskBase):
    def __init__(self, cfg=None, src_dict=None, tgt_dict=None, num_seg_tokens=None, patch_image_size=None,
                 n_base_vocab=59457, arch="segofa_base", src_len=None, category_token_ids=None):
        """`cfg, src_dict, tgt_dict` as in the reference (segmentation.py:102-107); the keyword form
        (`num_seg_tokens=...`) builds the config of the bundled harness with a size-only dictionary."""
        if cfg is None:
            cfg = SegmentationConfig(num_seg_tokens=15 if num_seg_tokens is None else num_seg_tokens,
                                     patch_image_size=512 if patch_image_size is None else patch_image_size,
                                     orig_patch_image_size=512 if patch_image_size is None else patch_image_size)
        super().__init__(cfg)
        self.arch = arch
        nseg = cfg.num_seg_tokens
        if src_dict is None:
            src_dict = tgt_dict = SizeDictionary(n_base_vocab, nseg)
        self.src_dict, self.tgt_dict = src_dict, tgt_dict
        self.uses_ema = getattr(cfg, "uses_ema", False)
        self.num_seg_tokens = nseg
        self.category_list = cfg.category_list
        self.seg_id_offset = tgt_dict.index("<seg_0>")
        # token ids of every category name, for the criterion's lazy seg-token initialisation where no BPE encoder
        # exists (the reference encodes `category_list` with task.bpe: seg_criterion.py:373-388)
        self.category_token_ids = category_token_ids
        self.bpe = None
        # L = bos + 12 prompt ids + class-name ids + eos: 36 / 215 / 239 for 15 / 150 / 171 classes
        self.src_len = src_len or {15: 36, 150: 215, 171: 239}.get(nseg, 14 + 2 * nseg)

    @classmethod
    def setup_task(cls, cfg, **kwargs):
        """segmentation.py:108-135.  Needs fairseq's Dictionary + `<bpe-dir>/dict.txt`."""
        if not HAVE_FAIRSEQ:
            raise RuntimeError("SegmentationTask.setup_task loads dict.txt through fairseq's Dictionary; without fairseq "
                               "construct SegmentationTask(num_seg_tokens=...) directly")
        dicts = []
        for _ in range(2):
            d = cls.load_dictionary(os.path.join(cfg.bpe_dir, "dict.txt"))
            d.add_symbol("<mask>")
            for i in range(cfg.code_dict_size):
                d.add_symbol("<code_{}>".format(i))
            for i in range(cfg.num_bins):
                d.add_symbol("<bin_{}>".format(i))
            for i in range(cfg.num_seg_tokens + 1):
                d.add_symbol("<seg_{}>".format(i))
            dicts.append(d)
        return cls(cfg, dicts[0], dicts[1])

    @property
    def source_dictionary(self):
        return self.src_dict

    @property
    def target_dictionary(self):
        return self.tgt_dict

    def load_dataset(self, split, epoch=1, combine=False, **kwargs):
        raise NotImplementedError("ifseg_amd: the TSV / mmseg data pipeline (data/mm_data/segmentation_dataset.py) is out "
                                  "of scope -- import the reference's own `tasks` next to `ifseg_amd.models` "
                                  "(INTEGRATION.md section 1) or feed samples in the collater's layout")

    def build_model(self, cfg=None):
        """segmentation.py:166-174 / ofa_task.py:167-185 (the BPE encoder and the vestigial sequence generator of the
        reference's build_model are data / decode plumbing: `ifseg_amd.sequence_generator` is built on demand)."""
        from ...models.segofa.segofa import SegOFAModel, recipe_args
        if cfg is None or not hasattr(cfg, "encoder_normalize_before"):
            over = {k: getattr(cfg, k) for k in ("arch", "dropout", "encoder_drop_path_rate", "decoder_drop_path_rate")
                    if cfg is not None and getattr(cfg, k, None) is not None}
            cfg = recipe_args(over.pop("arch", self.arch), num_seg_tokens=self.cfg.num_seg_tokens,
                              patch_image_size=self.cfg.patch_image_size,
                              orig_patch_image_size=self.cfg.orig_patch_image_size, **over)
        model = SegOFAModel.build_model(cfg, self)
        self._build_bpe_from_dir()
        return model

    def _build_bpe_from_dir(self):
        """ofa_task.py:167-185: under fairseq the task owns the GPT-2 BPE encoder built from `--bpe-dir`
        (encoder.json / vocab.bpe); the criterion's lazy seg-token initialisation encodes the category names with it
        (seg_criterion.py:373-388).  Without fairseq, or when the files are absent, `category_token_ids` must be given
        (or `--init-seg-with-text=false`): `encode_category` says so."""
        if self.bpe is not None or not HAVE_FAIRSEQ:
            return
        bpe_dir = getattr(self.cfg, "bpe_dir", None)
        if not bpe_dir:
            return
        enc, voc = os.path.join(bpe_dir, "encoder.json"), os.path.join(bpe_dir, "vocab.bpe")
        if not (os.path.exists(enc) and os.path.exists(voc)):
            return
        from omegaconf import DictConfig
        self.bpe = self.build_bpe(DictConfig({"_name": "gpt2", "gpt2_encoder_json": enc, "gpt2_vocab_bpe": voc}))

    def build_generator(self, models, args=None, seq_gen_cls=None, extra_gen_cls_kwargs=None, prefix_allowed_tokens_fn=None):
        """tasks/ofa_task.py:187-260 with the task's eval_args ({"beam":5,"max_len":1024,"min_len":1024,...},
        coco_unseen.sh:111): the fixed-length decode of ifseg_amd/sequence_generator.py"""
        from ...sequence_generator import SequenceGenerator
        g = lambda k, d: getattr(args, k, d) if args is not None else d
        return SequenceGenerator(models, self.target_dictionary, beam_size=g("beam", 5), max_len=g("max_len", None),
                                 min_len=g("min_len", 1), temperature=g("temperature", 1.0))

    def inference_step(self, generator, models, sample, prefix_tokens=None, constraints=None):
        """fairseq_task.py `inference_step` -> [B, max_len] seg-class indices of the best beam (segmentation.py:266-268)"""
        with torch.no_grad():
            return generator.generate(models, sample)

    def encode_category(self, text):
        """token ids of one category name (seg_criterion.py:375-385: BPE of ' <word>' per word, dictionary lookup)"""
        if self.bpe is not None:
            line = " ".join(self.bpe.encode(" {}".format(w.strip())) for w in text.strip().split())
            return self.tgt_dict.encode_line(line=line, add_if_not_exist=False, append_eos=False).long()
        raise RuntimeError("no BPE encoder on this task (fairseq + <bpe-dir>/encoder.json, vocab.bpe build one in "
                           "build_model): pass category_token_ids=[...] to SegmentationTask or --init-seg-with-text=false")

    def synthetic_sample(self, batch, device, seed=1234, image_hw=None):
        """Synthetic batch with the collater's layout (segmentation_dataset.py:41-129)."""
        nseg, S = self.cfg.num_seg_tokens, self.cfg.patch_image_size
        hw = image_hw or (S, S)
        g = torch.Generator().manual_seed(seed)
        L = self.src_len
        body = list(PROMPT_IDS)[: max(0, L - 2)]
        body += torch.randint(4, min(50000, self.seg_id_offset - 1), (max(0, L - 2 - len(body)),), generator=g).tolist()
        src = torch.tensor([BOS] + body + [EOS]).repeat(batch, 1)
        img = torch.randn(batch, 3, hw[0], hw[1], generator=g)
        tgt = torch.randint(0, nseg, (batch, hw[0] * hw[1]), generator=g) + self.seg_id_offset
        tgt = torch.cat([tgt, torch.full((batch, 1), EOS, dtype=torch.long)], 1)
        return {
            "id": list(range(batch)), "nsentences": batch, "ntokens": int(batch * (hw[0] * hw[1] + 1)),
            "net_input": {"src_tokens": src.to(device), "src_lengths": torch.full((batch,), L).to(device),
                          "patch_images": img.to(device), "patch_masks": torch.ones(batch, dtype=torch.bool, device=device),
                          "prev_output_tokens": torch.zeros(batch, 1, dtype=torch.long, device=device)},
            "target": tgt.to(device),
        }

    def synthetic_aux_sample(self, batch, device, seed=4321):
        """Image-free half of a batch in the collater's layout (segmentation_dataset.py:303-345, :85-107):
        a random low-res class map (``rand_k-1-33``) nearest-resized to the patch grid (EmbeddingBag ids of the
        class names; every "name" here is 1-3 random BPE ids) and to the image (targets).
        -> {"aux_input": {...}, "text2seg_target": [B, S*S+1]}"""
        import torch.nn.functional as F
        nseg, S = self.cfg.num_seg_tokens, self.cfg.patch_image_size
        hp = S // 16
        g = torch.Generator().manual_seed(seed)
        L = self.src_len
        name_len = torch.randint(1, 4, (nseg,), generator=g)
        names = [torch.randint(4, min(50000, self.seg_id_offset - 1), (int(k),), generator=g) for k in name_len]
        body = list(PROMPT_IDS)[: max(0, L - 2)]
        body += torch.randint(4, min(50000, self.seg_id_offset - 1), (max(0, L - 2 - len(body)),), generator=g).tolist()
        src = torch.tensor([BOS] + body + [EOS]).repeat(batch, 1)
        ids, ends, tgts = [], [], []
        for _ in range(batch):
            sh, sw = (int(v) for v in torch.randint(1, 33, (2,), generator=g))
            coarse = torch.randint(0, nseg, (1, 1, sh, sw), generator=g).float()
            low = F.interpolate(coarse, size=(hp, hp), mode="nearest").long().reshape(-1)
            high = F.interpolate(coarse, size=(S, S), mode="nearest").long().reshape(-1)
            ids.append(torch.cat([names[int(c)] for c in low]))
            ends.append(name_len[low].cumsum(0))
            tgts.append(torch.cat([high + self.seg_id_offset, torch.tensor([EOS])]))
        maxlen = max(t.numel() for t in ids)
        padded = torch.full((batch, maxlen), PAD, dtype=torch.long)
        for b, t in enumerate(ids):
            padded[b, : t.numel()] = t
        return {"aux_input": {"src_tokens": src.to(device), "src_lengths": torch.full((batch,), L).to(device),
                              "patch_images": padded.to(device), "patch_masks": torch.cat(ends).to(device),
                              "prev_output_tokens": torch.zeros(batch, 1, dtype=torch.long, device=device)},
                "text2seg_target": torch.stack(tgts).to(device)}

    def train_step(self, sample, model, criterion, optimizer, update_num, ignore_grad=False, **extra_kwargs):
        """tasks/mm_tasks/segmentation.py:190-222."""
        if not model.training:      # nn.Module.train() walks every sub-module: only when the mode changes
            model.train()
        model.set_num_updates(update_num)
        loss, sample_size, logging_output = criterion(model, sample, update_num=update_num,
                                                      ema_model=extra_kwargs.get("ema_model"))
        if ignore_grad:
            loss = loss * 0
        if optimizer is not None:
            optimizer.backward(loss)
        else:
            loss.backward()
        return loss, sample_size, logging_output

    def valid_step(self, sample, model, criterion, **extra_kwargs):
        """tasks/mm_tasks/segmentation.py:225-229."""
        if model.training:
            model.eval()
        with torch.no_grad():
            loss, sample_size, logging_output = criterion(model, sample)
        return loss, sample_size, logging_output
