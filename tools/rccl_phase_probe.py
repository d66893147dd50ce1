"""GPU-side phase times of the hooked (RCCL world-1) step: events around task.train_step (fwd+loss+bwd), reducer.finish, clip+Adam.
IFSEG_REDUCE_MODE = direct | c10d | fake-extra-stream | fake-same-stream | none selects how the slices are "reduced"."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1] if len(sys.argv) > 1 else "29561", RANK="0", WORLD_SIZE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
if not os.environ.get("PROBE_PLAIN"):
    os.environ["IFSEG_FORCE_GRAD_HOOK"] = "1"
import torch, torch.distributed as dist
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from ifseg_amd.criterions import SegCriterion
from ifseg_amd.tasks.mm_tasks import SegmentationTask
from ifseg_amd.trainer import Trainer
task = SegmentationTask(num_seg_tokens=15, patch_image_size=512, arch="segofa_base")
ring = []
for j in range(2):
    sm = task.synthetic_sample(8, dev, seed=300 + j)
    sm["net_input"]["patch_images"] = sm["net_input"]["patch_images"].to(torch.bfloat16)
    ring.append(sm)
if os.environ.get("PROBE_DUMMY_AR"):
    t_ = torch.ones(1 << 20, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        dist.all_reduce(t_, async_op=True).wait()
    torch.cuda.synchronize()
torch.manual_seed(0)
model = task.build_model()
tr = Trainer(model, SegCriterion(task, unsupervised_segmentation=False, init_seg_with_text=False), task, device=dev, lazy_logs=True)
marks = []
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
orig_ts, orig_fin = task.train_step, tr.reducer.finish
if os.environ.get("PROBE_PLAIN"):
    marks_fin = lambda: marks.append(("finish_end", ev()))
def ts(*a, **k):
    marks.append(("step_begin", ev())); r = orig_ts(*a, **k); marks.append(("bwd_end", ev()))
    if os.environ.get("PROBE_PLAIN"):
        marks.append(("finish_end", ev()))
    return r
def fin():
    orig_fin(); marks.append(("finish_end", ev()))
task.train_step, tr.reducer.finish = ts, fin
for i in range(6):
    tr.train_step([ring[i % 2]], prefetch=[ring[(i + 1) % 2]])
torch.cuda.synchronize(); marks.clear()
t0 = time.time()
for i in range(6, 16):
    tr.train_step([ring[i % 2]], prefetch=[ring[(i + 1) % 2]])
    marks.append(("step_end", ev()))
torch.cuda.synchronize()
print("wall %.2f ms/step" % ((time.time() - t0) * 100))
names = [n for n, _ in marks[:4]]
acc = {}
for k in range(0, len(marks) - 4, 4):
    es = [e for _, e in marks[k:k + 5]]
    for a in range(4):
        acc.setdefault(marks[k + a][0] + " -> " + marks[k + a + 1][0], []).append(es[a].elapsed_time(es[a + 1]))
for k, v in acc.items():
    print("%-28s %.3f ms" % (k, sum(v) / len(v)))
dist.destroy_process_group()
