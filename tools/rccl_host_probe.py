"""Where does the hooked (RCCL, world 1) step spend HOST time?  cProfile over 10 enqueued steps + per-step enqueue time."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1] if len(sys.argv) > 1 else "29533", RANK="0", WORLD_SIZE="1")
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
MODES = {"forced": (True,), "plain": (False,)}.get(sys.argv[2] if len(sys.argv) > 2 else "", (False, True))
for forced in MODES:
    if forced: os.environ["IFSEG_FORCE_GRAD_HOOK"] = "1"
    else: os.environ.pop("IFSEG_FORCE_GRAD_HOOK", None)
    torch.manual_seed(0)
    model = task.build_model()
    tr = Trainer(model, SegCriterion(task, unsupervised_segmentation=False, init_seg_with_text=False), task, device=dev, lazy_logs=True)
    for i in range(6):
        tr.train_step([ring[i % 2]], prefetch=[ring[(i + 1) % 2]])
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.time(); hs = []
    pr.enable()
    for i in range(6, 16):
        h0 = time.time()
        tr.train_step([ring[i % 2]], prefetch=[ring[(i + 1) % 2]])
        hs.append((time.time() - h0) * 1e3)
    pr.disable()
    torch.cuda.synchronize()
    print("forced=%s: %.2f ms/step wall, host enqueue per step: %s" % (forced, (time.time() - t0) * 100, " ".join("%.1f" % h for h in hs)))
    if forced:
        st = pstats.Stats(pr); st.sort_stats("cumulative"); st.print_stats(28)
    del tr, model
dist.destroy_process_group()
