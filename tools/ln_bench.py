"""Stand-alone timing of the LayerNorm kernels at the step's shapes (rows = 8 x 1060 / 8 x 1024, C = 768): what a launch costs with the
chip to itself, against its in-step duration (profiles/round5_kernel_stats.csv).  Buffers rotate through `nbuf` sets so that a
launch does not find its operands in the L2 / Infinity Cache unless nbuf = 1 says so.
usage: python tools/ln_bench.py [rows] [C] [nbuf]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ifseg_amd import hip

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8480
C = int(sys.argv[2]) if len(sys.argv) > 2 else 768
nbuf = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
def rnd(*s): return torch.randn(*s, device=dev, generator=g).bfloat16()
X = [rnd(rows, C) for _ in range(nbuf)]; DY = [rnd(rows, C) for _ in range(nbuf)]; ADD = [rnd(rows, C) for _ in range(nbuf)]
OUT = [torch.empty(rows, C, device=dev, dtype=torch.bfloat16) for _ in range(nbuf)]
OUT2 = [torch.empty(rows, C, device=dev, dtype=torch.bfloat16) for _ in range(nbuf)]
gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
mu, rs = torch.zeros(rows, device=dev), torch.ones(rows, device=dev)

def timeit(f, n=200):
    for i in range(10): f(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): f(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

def line(name, us, nbytes): print("%-52s %7.1f us  %6.0f GB/s" % (name, us, nbytes / us / 1e3), flush=True)
T = rows * C * 2
line("ln_fwd (read x, write y)", timeit(lambda i: hip.ln_fwd(X[i % nbuf], gam, bet, OUT[i % nbuf], mu, rs)), 2 * T)
line("ln_fwd + resid (read x, r, write y)", timeit(lambda i: hip.ln_fwd(X[i % nbuf], gam, bet, OUT[i % nbuf], mu, rs, resid=ADD[i % nbuf])), 3 * T)
mu2, rs2 = torch.zeros(rows, device=dev), torch.ones(rows, device=dev)
line("ln_fwd_pair + resid (read x, r, write y, y2)", timeit(lambda i: hip.ln_fwd_pair(X[i % nbuf], gam, bet, OUT[i % nbuf], mu, rs, gam, bet, OUT2[i % nbuf], mu2, rs2, resid=ADD[i % nbuf])), 4 * T)
hip.ln_fwd(X[0], gam, bet, OUT[0], mu, rs)
for nb in [int(v) for v in os.environ.get("BLOCKS", "256,512,768,1024,2048").split(",")]:
    hip.LN_BWD_BLOCKS = nb
    part = torch.empty(2, nb, C, device=dev)
    line("ln_bwd blocks=%d (read dy, x, write dx)" % nb, timeit(lambda i: hip.ln_bwd(DY[i % nbuf], X[i % nbuf], gam, mu, rs, OUT[i % nbuf], part[0], part[1])), 3 * T)
    line("ln_bwd blocks=%d + dx_add" % nb, timeit(lambda i: hip.ln_bwd(DY[i % nbuf], X[i % nbuf], gam, mu, rs, OUT[i % nbuf], part[0], part[1], dx_add=ADD[i % nbuf])), 4 * T)
    line("ln_bwd_drop blocks=%d + dx_add (2 outputs)" % nb, timeit(lambda i: hip.ln_bwd_drop(DY[i % nbuf], X[i % nbuf], gam, mu, rs, OUT[i % nbuf], part[0], part[1], OUT2[i % nbuf], dx_add=ADD[i % nbuf])), 5 * T)
cp = lambda i: OUT[i % nbuf].copy_(X[i % nbuf])
line("torch copy_ (read + write)", timeit(cp), 2 * T)
