"""segmentation task -- caller of the hot path (mirror of tasks/mm_tasks/segmentation.py).

Keeps the reference's task surface (``@register_task("segmentation")`` :100,
``build_model`` :166, ``train_step`` :190-222, ``valid_step`` :225-229) and the
``sample`` dict layout of data/mm_data/segmentation_dataset.py:110-127.  Data loading
(TSV / mmseg pipeline) is out of scope: samples are synthetic
tensors of the right shapes, dictionary sizes follow
tasks/mm_tasks/segmentation.py:113-132.
"""
import torch

from ...registry import register_task

BOS, PAD, EOS = 0, 1, 2
# "what is the segmentation map of the image? object:" BPE ids
PROMPT_IDS = (99, 16, 5, 2835, 1258, 5456, 9, 5, 2274, 116, 7626, 35)


class _Dict:
    """Size-only stand-in for the fairseq Dictionary of the task (dict.txt 50260 lines + 4
    specials + <mask> + 8192 <code_i> + 1000 <bin_i> + (nseg+1) <seg_i>)."""

    def __init__(self, n_base, nseg):
        self.n_base, self.nseg = n_base, nseg

    def __len__(self):
        return self.n_base + self.nseg + 1

    def pad(self):
        return PAD

    def bos(self):
        return BOS

    def eos(self):
        return EOS

    def index(self, sym):
        if sym == "<seg_0>":
            return self.n_base
        raise KeyError(sym)


class _Cfg:
    def __init__(self, **kw):
        self.__dict__.update(kw)


@register_task("segmentation")
class SegmentationTask:
    def __init__(self, num_seg_tokens=15, patch_image_size=512, n_base_vocab=59457, arch="segofa_base",
                 src_len=None):
        self.cfg = _Cfg(num_seg_tokens=num_seg_tokens, patch_image_size=patch_image_size, arch=arch,
                        orig_patch_image_size=patch_image_size)
        self.src_dict = self.tgt_dict = _Dict(n_base_vocab, num_seg_tokens)
        self.seg_id_offset = n_base_vocab
        # L = bos + 12 prompt ids + class-name ids + eos: 36 / 215 / 239 for 15 / 150 / 171 classes
        self.src_len = src_len or {15: 36, 150: 215, 171: 239}.get(num_seg_tokens, 14 + 2 * num_seg_tokens)

    @property
    def source_dictionary(self):
        return self.src_dict

    @property
    def target_dictionary(self):
        return self.tgt_dict

    def build_model(self, args=None):
        from ...models.segofa import SegOFAModel
        a = args or _Cfg()
        for k in ("arch", "num_seg_tokens", "patch_image_size", "orig_patch_image_size"):
            if getattr(a, k, None) is None:
                setattr(a, k, getattr(self.cfg, k))
        return SegOFAModel.build_model(a, self)

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

    def train_step(self, sample, model, criterion, optimizer, update_num, ignore_grad=False):
        """tasks/mm_tasks/segmentation.py:190-222."""
        if not model.training:      # nn.Module.train() walks every sub-module: only when the mode changes
            model.train()
        loss, sample_size, logging_output = criterion(model, sample, update_num=update_num)
        if ignore_grad:
            loss = loss * 0
        if optimizer is not None:
            optimizer.backward(loss)
        else:
            loss.backward()
        return loss, sample_size, logging_output

    def valid_step(self, sample, model, criterion):
        """tasks/mm_tasks/segmentation.py:225-229."""
        if model.training:
            model.eval()
        with torch.no_grad():
            loss, sample_size, logging_output = criterion(model, sample)
        return loss, sample_size, logging_output
