"""Dense-CRF post-processing on the MI355X (mirror of the reference's crf.py:19-37, BASELINE config 5).

`rgb_dense_crf(image, output_probs, max_iter=10)` keeps the reference's name, arguments and result layout ([c, h, w]
class probabilities after `max_iter` mean-field iterations; numpy in -> numpy out, torch in -> torch out) and its
constants (crf.py:11-15).  The reference delegates to pydensecrf (permutohedral-lattice filtering, not available
here: parity unpinned); this implementation evaluates the same mean-field update with the exact dense kernels in
csrc/crf.hip.  No CPU fallback: it raises without a device."""
import math

import numpy as np
import torch

from . import hip

POS_W, POS_XY_STD, Bi_W, Bi_XY_STD, Bi_RGB_STD = 3, 1, 4, 67, 3         # crf.py:11-15
_POS_RADIUS = 5                                                         # exp(-25 / 2) = 3.7e-6


def rgb_dense_crf(image, output_probs, max_iter=10, device=None):
    was_numpy = isinstance(output_probs, np.ndarray)
    dev = torch.device(device) if device is not None else (output_probs.device if torch.is_tensor(output_probs) and output_probs.is_cuda
                                                            else torch.device("cuda", torch.cuda.current_device()))
    if dev.type != "cuda":
        raise RuntimeError("ifseg_amd.crf.rgb_dense_crf runs on the MI355X only")
    prob = torch.as_tensor(output_probs).to(dev, torch.float32).contiguous()
    C, H, W = prob.shape
    N, Cp, ldq = H * W, (C + 31) // 32 * 32, (H * W + 63) // 64 * 64
    img = torch.as_tensor(np.ascontiguousarray(image) if isinstance(image, np.ndarray) else image).to(dev, torch.float32).reshape(N, 3)
    ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.int32), torch.arange(W, device=dev, dtype=torch.int32), indexing="ij")
    feat = torch.empty(N, 4, dtype=torch.float32, device=dev)
    feat[:, :3] = img * math.sqrt(math.log2(math.e) / (2.0 * Bi_RGB_STD ** 2))
    feat[:, 3] = (xs.reshape(-1) | (ys.reshape(-1) << 16)).view(torch.float32)
    d = torch.arange(max(H, W), device=dev, dtype=torch.float32)
    gbi = torch.exp(-d * d / (2.0 * Bi_XY_STD ** 2)).contiguous()
    gpos = torch.exp(-d[: _POS_RADIUS + 1] ** 2 / (2.0 * POS_XY_STD ** 2)).contiguous()
    prev = hip.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        # normalisations n = 1 / sqrt(K 1)
        one_bi = torch.zeros(32, ldq, dtype=torch.bfloat16, device=dev)
        one_bi[0, :N] = 1
        k1 = torch.empty(32, N, dtype=torch.float32, device=dev)
        hip.crf_bilateral(feat, gbi, gbi, one_bi, k1, H, W)
        nbi = torch.empty(N, dtype=torch.float32, device=dev)
        hip.crf_norm(k1[0], nbi)
        kp = torch.empty(1, N, dtype=torch.float32, device=dev)
        hip.crf_spatial(torch.ones(1, N, dtype=torch.float32, device=dev), gpos, _POS_RADIUS, kp, H, W)
        npos = torch.empty(N, dtype=torch.float32, device=dev)
        hip.crf_norm(kp[0], npos)
        Q = torch.empty(C, N, dtype=torch.float32, device=dev)
        qpos = torch.empty(C, N, dtype=torch.float32, device=dev)
        qbi = torch.zeros(Cp, ldq, dtype=torch.bfloat16, device=dev)
        mpos = torch.empty(C, N, dtype=torch.float32, device=dev)
        mbi = torch.empty(Cp, N, dtype=torch.float32, device=dev)
        hip.crf_update(prob.view(C, N), None, None, npos, nbi, float(POS_W), float(Bi_W), Q, qpos, qbi)
        for _ in range(max_iter):
            hip.crf_spatial(qpos, gpos, _POS_RADIUS, mpos, H, W)
            hip.crf_bilateral(feat, gbi, gbi, qbi, mbi, H, W)
            hip.crf_update(prob.view(C, N), mpos, mbi, npos, nbi, float(POS_W), float(Bi_W), Q, qpos, qbi)
    finally:
        hip.set_stream(prev)
    out = Q.view(C, H, W)
    return out.cpu().numpy() if was_numpy else out
