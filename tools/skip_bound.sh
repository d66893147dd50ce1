#!/bin/bash
# Upper bounds: the step with one kernel family LEFT OUT (wrong results, timing only) -- what optimising it could return at most.
# usage (GPU box): bash tools/skip_bound.sh
for v in "" lnwide dq dkv reduce dw "dq,dkv" ""; do
  IFSEG_EXP_SKIP=$v python bench.py --steps 60 --warmup 8 --no-cpu-baseline --steady-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip=%-10s %7.2f img/s  %7.3f ms' % ('$v', d['value'], d['ms_per_step']))"
done
# the trunk: features of the previous call are reused (no trunk pass at all)
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from ifseg_amd.criterions import SegCriterion
from ifseg_amd.tasks.mm_tasks import SegmentationTask
from ifseg_amd.trainer import Trainer
dev = torch.device("cuda:0")
task = SegmentationTask(num_seg_tokens=15, patch_image_size=512, arch="segofa_base")
torch.manual_seed(0)
tr = Trainer(task.build_model(), SegCriterion(task, unsupervised_segmentation=False, init_seg_with_text=False), task, device=dev)
sm = task.synthetic_sample(8, dev, seed=1); sm["net_input"]["patch_images"] = sm["net_input"]["patch_images"].to(torch.bfloat16)
eng = tr.eng
orig = eng._resnet
cache = {}
def fake(images, tag=""):
    if "f" not in cache: cache["f"] = orig(images, tag)
    return cache["f"]
for name, fn in (("with trunk", orig), ("no trunk", fake)):
    eng._resnet = fn
    for _ in range(6): tr.train_step([sm])
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(40): tr.train_step([sm])
    torch.cuda.synchronize()
    print("%-12s in line: %.3f ms/step" % (name, (time.time() - t0) / 40 * 1e3))
PY
