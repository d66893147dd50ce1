"""Algorithmic FLOPs per image of the SegOFA fwd / fwd+bwd step, counted on the CPU oracle (oracle/segofa_ref.py) with
torch.utils.flop_counter.FlopCounterMode -- the method of BASELINE.md section 2 (2 x MAC of every GEMM / bmm / conv the
reference executes; softmax / LayerNorm / GELU / elementwise excluded; attention dense; frozen ResNet trunk: no backward
through it).  Run in the build container (CPU):

    python tools/count_flops.py base15 | base150 | large171

bench.py's GF_PER_IMG holds the printed figures (SURVEY 8d asks for the Large figure to be derived, not estimated).
"""
import os
import sys

import torch
from torch.utils.flop_counter import FlopCounterMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import segofa_ref as O  # noqa: E402

CASES = {"base15": (O.base_config, dict(num_seg_tokens=15), 36, 512),
         "base150": (O.base_config, dict(num_seg_tokens=150), 215, 512),
         "large171": (O.large_config, dict(num_seg_tokens=171), 239, 640)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "base15"
    mk, kw, L, S = CASES[name]
    cfg = mk(patch_image_size=S, orig_patch_image_size=S, **kw)
    torch.set_num_threads(os.cpu_count() or 1)
    sd = O.procedural_state_dict(cfg)
    spec = O.state_dict_spec(cfg)
    for k, v in sd.items():
        if v.dtype.is_floating_point and "embed_images" not in k and not spec[k][1].startswith("alias"):
            # the recipe's trainable set (coco_unseen.sh:31-33,76: token / seg embeddings, image_proj and the trunk frozen)
            v.requires_grad_("embed_tokens" not in k and "seg_embed_tokens" not in k and "image_proj" not in k)
    batch = O.synthetic_batch(cfg, 1, L, image_size=S)
    with FlopCounterMode(display=False) as fc:
        logits, extra = O.segofa_forward(sd, cfg, batch["src_tokens"], batch["patch_images"])
    fwd = fc.get_total_flops()
    hp = S // 16
    with FlopCounterMode(display=False) as fc2:
        logits, extra = O.segofa_forward(sd, cfg, batch["src_tokens"], batch["patch_images"])
        loss, _, _ = O.seg_loss(cfg, logits, batch["target"], hp, hp, S, S)
        loss.backward()
    tot = fc2.get_total_flops()
    print("%s: T_enc %d  fwd %.1f GF  fwd+bwd %.1f GF per image (frozen trunk)" % (name, hp * hp + L, fwd / 1e9, tot / 1e9))


if __name__ == "__main__":
    main()
