"""The data-parallel leg through RCCL's real stream semantics on ONE GPU (tests/test_configs_gpu.py): process group
"nccl" (= RCCL) with world_size 1, IFSEG_FORCE_GRAD_HOOK=1 -- `dist.all_reduce(bf16 slice, async_op=True)` is issued per
layer from the weight-gradient stream inside the backward, `finish()` waits on the works before clip + Adam, the logging
outputs go through one fp64 all-reduce.  Compared with the hook-less step of the same model: parameters bit-equal, step
time printed.
    python _rccl_worker.py PORT OUTFILE"""
import json
import os
import sys
import time

port, outfile = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from ifseg_amd.criterions import SegCriterion  # noqa: E402
from ifseg_amd.tasks.mm_tasks import SegmentationTask  # noqa: E402
from ifseg_amd.trainer import Trainer  # noqa: E402

task = SegmentationTask(num_seg_tokens=15, patch_image_size=512, arch="segofa_base")
ring = []
for j in range(2):
    sm = task.synthetic_sample(8, dev, seed=300 + j)
    sm["net_input"]["patch_images"] = sm["net_input"]["patch_images"].to(torch.bfloat16)
    ring.append(sm)


def run(forced, nsteps=6, timed=30):
    if forced:
        os.environ["IFSEG_LAB"] = "1"                 # (the gate of every laboratory switch: ifseg_amd/lab.py)
        os.environ["IFSEG_FORCE_GRAD_HOOK"] = "1"
    else:
        os.environ.pop("IFSEG_FORCE_GRAD_HOOK", None)
    torch.manual_seed(0)
    model = task.build_model()                       # recipe: dropout 0.1, drop-path 0.1
    tr = Trainer(model, SegCriterion(task, unsupervised_segmentation=False, init_seg_with_text=False), task, device=dev, lazy_logs=True)
    assert (tr.eng.grad_ready_hook is not None) == forced and tr.dist_on == forced
    losses = []
    for i in range(nsteps):
        logs = tr.train_step([ring[i % 2]], prefetch=[ring[(i + 1) % 2]])
        losses.append(float(logs[-1]["loss"]))
    tr.check_overflow(wait=True)
    torch.cuda.synchronize()
    state = (tr.eng.g16.clone(), tr.eng.p16.clone(), tr.p32.clone())
    t0 = time.time()
    for i in range(nsteps, nsteps + timed):
        tr.train_step([ring[i % 2]], prefetch=[ring[(i + 1) % 2]])
    torch.cuda.synchronize()
    ms = (time.time() - t0) / timed * 1e3
    nred = len(tr.reducer.slices)
    mode = "direct" if tr.reducer.direct is not None else tr.reducer.mode
    del tr, model
    torch.cuda.empty_cache()
    return losses, state, ms, nred, mode


l0, s0, ms0, _, _ = run(False)
l1, s1, ms1, nred, mode = run(True)
res = {"losses_equal": l0 == l1, "g16_equal": bool(torch.equal(s0[0], s1[0])), "p16_equal": bool(torch.equal(s0[1], s1[1])),
       "p32_equal": bool(torch.equal(s0[2], s1[2])), "ms_plain": ms0, "ms_rccl": ms1, "slices": nred,
       "backend": dist.get_backend(), "reduce_mode": mode}
with open(outfile, "w") as f:
    json.dump(res, f)
dist.destroy_process_group()
print("WORKER-OK", json.dumps(res))
