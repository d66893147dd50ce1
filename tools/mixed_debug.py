"""Per-tensor gradient comparison of three attention-path settings over a few Trainer steps (debug)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    import torch, numpy as np
    import segofa_ref as O
    import test_configs_gpu as T
    from ifseg_amd.tasks.mm_tasks import SegmentationTask
    from ifseg_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    ocfg = O.base_config(num_seg_tokens=150, vocab_size=59458)
    sd = O.round_weights_bf16(O.procedural_state_dict(ocfg))
    m = T._base_model(ocfg, sd, dev)
    task = SegmentationTask(num_seg_tokens=150, patch_image_size=512, n_base_vocab=ocfg.vocab_size - 1)
    tr = Trainer(m, T._crit(ocfg), task, lr=5e-4, max_update=900, device=dev)
    samples = [T._sample(T._learnable_batch(ocfg, 4, s, dev), dev) for s in range(2)]
    out = {}
    for k in range(3):
        logs = tr.train_step([samples[k % 2]])
        torch.cuda.synchronize()
        out["loss%d" % k] = float(logs[0]["loss"])
        out["g%d" % k] = tr.eng.g16.float().cpu()
    out["offs"] = {n: (tr.eng.offs[n], tr.eng.shapes[n]) for n in tr.eng.trainable_names()}
    torch.save(out, sys.argv[2])
    sys.exit(0)
import torch, math
res = {}
for mode in ("1", "auto", "0"):
    f = "/tmp/mixed_%s.pt" % mode
    env = dict(os.environ, IFSEG_ATTN_BI=mode)
    subprocess.run([sys.executable, __file__, "worker", f], env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    res[mode] = torch.load(f)
    print(mode, [round(res[mode]["loss%d" % k], 5) for k in range(3)])
offs = res["1"]["offs"]
for k in range(3):
    print("step", k)
    for a, b in (("auto", "1"), ("0", "1")):
        ga, gb = res[a]["g%d" % k], res[b]["g%d" % k]
        worst = []
        for n, (o, sh) in offs.items():
            cnt = math.prod(sh)
            x, y = ga[o:o + cnt], gb[o:o + cnt]
            d = (x - y).norm().item() / (y.norm().item() + 1e-20)
            worst.append((d, n))
        worst.sort(reverse=True)
        print("  %s vs %s: total rel %.4f; worst: %s" % (a, b, ((ga - gb).norm() / gb.norm()).item(), [(round(d, 3), n) for d, n in worst[:6]]))
