"""Where does the bf16 error of the HIP path come from?  Stage-wise rel-L2 vs the fp32 oracle (Base, B=1)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
torch.set_num_threads(32)
import segofa_ref as O
from ifseg_amd.models.segofa import SegOFAModel, make_config
rel = lambda a, b: ((a.float().cpu() - b.float().cpu()).norm() / b.float().norm()).item()
dev = torch.device("cuda:0")
cfg = O.base_config(); sd = O.procedural_state_dict(cfg); batch = O.synthetic_batch(cfg, 1, 36)
m = SegOFAModel(make_config("segofa_base")); torch.nn.Module.load_state_dict(m, sd, strict=False); m.to(dev).eval()
with torch.no_grad():
    feat = O.resnet_trunk(sd, "encoder.embed_images.", batch["patch_images"], cfg.resnet_layers)
    enc = O.encode(sd, cfg, batch["src_tokens"], batch["patch_images"], image_feat=feat)
    logits, _ = O.decode(sd, cfg, enc, batch["prev_output_tokens"])
    lg, extra = m(src_tokens=batch["src_tokens"].to(dev), patch_images=batch["patch_images"].to(dev),
                  prev_output_tokens=batch["prev_output_tokens"].to(dev))
    eng = m.engine
    print("resnet feat      ", rel(eng.ws["rn_feat"].view(1, 1024, 1024), feat.flatten(2).transpose(1, 2)))
    print("encoder_out      ", rel(eng.ctx["enc_out"], enc["encoder_out"]))
    print("logits           ", rel(lg, logits))
    # same, but feeding the HIP ResNet features to the oracle (isolates the transformer)
    f2 = eng.ws["rn_feat"].float().cpu().view(1, 32, 32, 1024).permute(0, 3, 1, 2)
    enc2 = O.encode(sd, cfg, batch["src_tokens"], batch["patch_images"], image_feat=f2)
    l2, _ = O.decode(sd, cfg, enc2, batch["prev_output_tokens"])
    print("encoder_out | same feat", rel(eng.ctx["enc_out"], enc2["encoder_out"]))
    print("logits      | same feat", rel(lg, l2))
