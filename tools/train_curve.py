"""loss curve of the learnable synthetic task (tests/test_configs_gpu.py) under an attention-path setting (debug)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import segofa_ref as O
import test_configs_gpu as T
from ifseg_amd.tasks.mm_tasks import SegmentationTask
from ifseg_amd.trainer import Trainer
dev = torch.device("cuda:0")
ocfg = O.base_config(num_seg_tokens=150, vocab_size=59458)
sd = O.round_weights_bf16(O.procedural_state_dict(ocfg))
m = T._base_model(ocfg, sd, dev)
task = SegmentationTask(num_seg_tokens=150, patch_image_size=512, n_base_vocab=ocfg.vocab_size - 1)
tr = Trainer(m, T._crit(ocfg), task, lr=float(os.environ.get("TW_LR", "5e-4")), max_update=900, device=dev)
samples = [T._sample(T._learnable_batch(ocfg, 4, s, dev), dev) for s in range(8)]
acc = []
for k in range(800):
    logs = tr.train_step([samples[k % 8]])
    acc.append(float(logs[0]["loss"]))
    if (k + 1) % 80 == 0:
        print("%4d %.3f |g| %.3f" % (k + 1, sum(acc[-8:]) / 8, tr.grad_norm()), flush=True)
