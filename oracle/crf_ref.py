"""CPU restatement of the dense-CRF post-processing of the reference (crf.py:19-37) -- TEST INFRASTRUCTURE.

The reference calls the third-party `pydensecrf` (Krahenbuhl & Koltun's DenseCRF; not vendored, unpinned in
requirements.txt, NOT installed in this image):

    U = unary_from_softmax(probs)                      # -log(clip(p, 1e-5, 1))            pydensecrf/utils.py
    d = DenseCRF2D(w, h, c); d.setUnaryEnergy(U)
    d.addPairwiseGaussian(sxy=1, compat=3)             # k1 = exp(-|dp|^2 / (2 * 1^2)),       Potts weight 3
    d.addPairwiseBilateral(sxy=67, srgb=3, rgbim, 4)   # k2 = exp(-|dp|^2/(2*67^2) - |dI|^2/(2*3^2)), Potts weight 4
    Q = d.inference(10)

Published algorithm (densecrf.cpp `DenseCRF::inference`, pairwise.cpp `PairwisePotential::apply`, defaults
DIAG_KERNEL + NORMALIZE_SYMMETRIC): Q0 = softmax(-U); each iteration
    Q <- softmax( -U + sum_k w_k * n_k .* (K_k (n_k .* Q)) ),      n_k = 1 / sqrt(K_k 1 + 1e-20)
where K_k includes the i == j term.  pydensecrf evaluates K_k Q with the permutohedral lattice, an APPROXIMATE
high-dimensional Gaussian filter; this restatement (and the HIP kernel it checks) evaluates the same mean-field update
with the EXACT dense kernel.  PARITY UNPINNED: without pydensecrf the lattice's approximation error cannot be measured
here; the reference repository holds no test or golden vector for this step.
"""
import torch

POS_W, POS_XY_STD, BI_W, BI_XY_STD, BI_RGB_STD = 3.0, 1.0, 4.0, 67.0, 3.0     # crf.py:11-15


def _kernel_matrices(image, pos_xy=POS_XY_STD, bi_xy=BI_XY_STD, bi_rgb=BI_RGB_STD):
    """image uint8 / float [h, w, 3] -> dense K_pos, K_bi [N, N] (float64); small images only"""
    h, w = image.shape[:2]
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float64), torch.arange(w, dtype=torch.float64), indexing="ij")
    p = torch.stack([xs.reshape(-1), ys.reshape(-1)], 1)
    col = torch.as_tensor(image).reshape(-1, 3).to(torch.float64)
    d2p = ((p[:, None] - p[None]) ** 2).sum(-1)
    d2c = ((col[:, None] - col[None]) ** 2).sum(-1)
    return torch.exp(-d2p / (2 * pos_xy ** 2)), torch.exp(-d2p / (2 * bi_xy ** 2) - d2c / (2 * bi_rgb ** 2))


def rgb_dense_crf_exact(image, output_probs, max_iter=10):
    """image [h, w, 3] (RGB, 0..255), output_probs [c, h, w] -> Q [c, h, w] after `max_iter` mean-field iterations
    (exact dense kernels, O(N^2): for fixtures up to ~48 x 48 pixels)."""
    probs = torch.as_tensor(output_probs).to(torch.float64)
    c, h, w = probs.shape
    U = -torch.log(probs.clamp(1e-5, 1.0)).reshape(c, -1)          # [c, N]
    Kp, Kb = _kernel_matrices(image)
    npos, nbi = 1.0 / torch.sqrt(Kp.sum(1) + 1e-20), 1.0 / torch.sqrt(Kb.sum(1) + 1e-20)
    Q = torch.softmax(-U, 0)
    for _ in range(max_iter):
        msg = POS_W * npos * ((Q * npos) @ Kp) + BI_W * nbi * ((Q * nbi) @ Kb)       # K symmetric
        Q = torch.softmax(-U + msg, 0)
    return Q.reshape(c, h, w).float()
