"""Error budget of the HIP path at Base (B=1): stage-wise rel-L2 against (a) the fp32 oracle and (b) the fp32 oracle
run on bf16-ROUNDED weights -- (b) isolates activation rounding from the unavoidable rounding of the parameters."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
torch.set_num_threads(min(32, os.cpu_count() or 1))
import segofa_ref as O
from ifseg_amd.models.segofa import SegOFAModel, make_config
rel = lambda a, b: ((a.float().cpu() - b.float().cpu()).norm() / b.float().norm()).item()
dev = torch.device("cuda:0")
cfg = O.base_config(); sd = O.procedural_state_dict(cfg); batch = O.synthetic_batch(cfg, 1, 36)
sd16 = {k: (v.to(torch.bfloat16).float() if v.dtype.is_floating_point and "embed_images" not in k else v) for k, v in sd.items()}
m = SegOFAModel(make_config("segofa_base")); torch.nn.Module.load_state_dict(m, sd, strict=False); m.to(dev).eval()


def oracle(w, feat=None):
    with torch.no_grad():
        f = O.resnet_trunk(w, "encoder.embed_images.", batch["patch_images"], cfg.resnet_layers) if feat is None else feat
        enc = O.encode(w, cfg, batch["src_tokens"], batch["patch_images"], image_feat=f)
        lg, _ = O.decode(w, cfg, enc, batch["prev_output_tokens"])
    return f, enc["encoder_out"], lg


with torch.no_grad():
    lg, extra = m(src_tokens=batch["src_tokens"].to(dev), patch_images=batch["patch_images"].to(dev),
                  prev_output_tokens=batch["prev_output_tokens"].to(dev))
eng = m.engine
hf = eng.ws["rn_feat"].float().cpu().view(1, 32, 32, 1024).permute(0, 3, 1, 2)
f, e, l = oracle(sd)
print("vs fp32 oracle               : feat %.4f enc %.4f logits %.4f" % (rel(hf, f), rel(eng.ctx["enc_out"], e), rel(lg, l)))
f2, e2, l2 = oracle(sd16)
print("fp32 oracle(bf16 w) vs fp32   : feat %.4f enc %.4f logits %.4f  (weight rounding alone)" % (rel(f2, f), rel(e2, e), rel(l2, l)))
print("vs oracle(bf16 w)            : feat %.4f enc %.4f logits %.4f" % (rel(hf, f2), rel(eng.ctx["enc_out"], e2), rel(lg, l2)))
_, e3, l3 = oracle(sd, feat=hf)
print("vs fp32 oracle | HIP feat    : enc %.4f logits %.4f" % (rel(eng.ctx["enc_out"], e3), rel(lg, l3)))
_, e4, l4 = oracle(sd16, feat=hf)
print("vs oracle(bf16 w) | HIP feat : enc %.4f logits %.4f" % (rel(eng.ctx["enc_out"], e4), rel(lg, l4)))
ag = lambda a, b: (a.float().cpu()[:, :-1].argmax(-1) == b[:, :-1].argmax(-1)).float().mean().item()
print("argmax agreement vs fp32 %.4f" % ag(lg, l))
