"""round 6, VERDICT r5 item 2(b): "round filling for N = 768, measured, not argued".  Stand-alone times (idle MI355X, M = 8480) of the
forward products with N = 768 as 402 tiles of 128 x 128 (default), as 804 tiles of 128 x 64 (IFSEG_LAB=1 IFSEG_GEMM_NARROW_MAX=512 in a
child process; the switch exists in commit f42f41a only: it was removed with the experiment) and, for K = 3072, as a 2-way split-K (804 workgroups, fp32 slabs) + the reduction launch that sums them."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    from ifseg_amd import hip
    from tools.gemm_bench import bench
    dev = torch.device("cuda:0")
    r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
    M = 8480
    tag = "narrow(804)" if os.environ.get("IFSEG_GEMM_NARROW_MAX") else "default(402)"
    for (N, K) in [(768, 768), (768, 3072)]:
        x, w, b = r(M, K), r(N, K), r(N)
        y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        res = r(M, N)
        bench("%s NT N%d K%d +bias" % (tag, N, K), lambda: hip.linear_fwd(x, w, b, out=y), 2.0 * M * N * K, iters=50)
        bench("%s NT N%d K%d +bias+resid" % (tag, N, K), lambda: hip.linear_fwd(x, w, b, out=y, resid=res), 2.0 * M * N * K, iters=50)
        if K == 3072 and not os.environ.get("IFSEG_GEMM_NARROW_MAX"):
            slabs = torch.empty(2, M, N, dtype=torch.float32, device=dev)
            def split():
                hip.gemm(hip.GEMM_NT, x, w, slabs, M, N, K, K, K, N, flags=hip.GEMM_OUT_F32, splitk=2)
                hip.reduce_parts(slabs, y, 1, 2, M * N)
            bench("split-K 2 (804 wg) + reduce N%d K%d" % (N, K), split, 2.0 * M * N * K, iters=50)
            bench("  split-K 2 GEMM alone", lambda: hip.gemm(hip.GEMM_NT, x, w, slabs, M, N, K, K, K, N, flags=hip.GEMM_OUT_F32, splitk=2),
                  2.0 * M * N * K, iters=50)
            y2 = torch.empty_like(y)
            hip.linear_fwd(x, w, None, out=y2); split(); torch.cuda.synchronize()
            print("  split-K vs plain: max abs diff %.3g" % (y.float() - y2.float()).abs().max().item())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        run()
    else:
        run()
        env = dict(os.environ, IFSEG_LAB="1", IFSEG_GEMM_NARROW_MAX="512")
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
