"""Plain (hook-less) Base step time before / after the process has issued its first RCCL collective.
env: PROBE_DESTROY=1 destroy the process group after the dummy collectives; PROBE_NO_AR=1 skip them"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1] if len(sys.argv) > 1 else "29581", RANK="0", WORLD_SIZE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
torch.manual_seed(0)
model = task.build_model()
tr = Trainer(model, SegCriterion(task, unsupervised_segmentation=False, init_seg_with_text=False), task, device=dev, lazy_logs=True)
n = [0]
def run(k, tag):
    for _ in range(4):
        tr.train_step([ring[n[0] % 2]], prefetch=[ring[(n[0] + 1) % 2]]); n[0] += 1
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(k):
        tr.train_step([ring[n[0] % 2]], prefetch=[ring[(n[0] + 1) % 2]]); n[0] += 1
    torch.cuda.synchronize()
    print("%-40s %.2f ms/step" % (tag, (time.time() - t0) / k * 1e3))
run(20, "before any collective")
if not os.environ.get("PROBE_NO_AR"):
    t_ = torch.ones(1 << 20, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        dist.all_reduce(t_, async_op=True).wait()
    torch.cuda.synchronize()
run(20, "after 3 all_reduce calls")
if os.environ.get("PROBE_DESTROY"):
    dist.destroy_process_group()
    run(20, "after destroy_process_group")
else:
    dist.destroy_process_group()
