"""Process-wide HIP streams of the engine, chosen so that they do NOT share a hardware queue.

HIP multiplexes the streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4; bench.py raises it to 8), and
which queue a stream gets depends on how many streams the process -- PyTorch's pool, RCCL, ... -- created before it.  Two
streams on one hardware queue execute in order: with the weight-gradient stream behind the main stream's queue the 4-stream
step loses all of its overlap (measured 18.2 -> 26-28 ms per step, bimodal from run to run; tools/stream_alias_probe.py
prints the aliasing map).  So the engine does not take "the next pool stream": it asks for candidates and MEASURES, once per
process, which of them run concurrently with the current stream and with each other -- a long kernel on one, a tiny one on
the other; if the tiny one finishes last the two are serialised.  ~30 ms at start-up.
"""
import os

import torch

_cache = {}


def _serialised(a, b, work, tiny):
    """True if a tiny kernel enqueued on b after `work` was enqueued on a completes only when a's work does"""
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(a):
        e0.record()
        work()
        e1.record()
    with torch.cuda.stream(b):
        tiny.add_(1)
        e2.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e2) > 0.5 * e0.elapsed_time(e1)


def engine_streams(device):
    """-> {"side": weight-gradient stream, "dq": dQ stream, "trunk": frozen-trunk stream (high priority)}: process-wide,
    mutually concurrent and concurrent with the stream that is current on `device` at the first call."""
    device = torch.device(device)
    key = (device.type, device.index)
    if key in _cache:
        return _cache[key]
    main = torch.cuda.current_stream(device)
    if torch.cuda.is_current_stream_capturing() or os.environ.get("IFSEG_NO_STREAM_PROBE"):
        out = {"side": torch.cuda.Stream(device), "dq": torch.cuda.Stream(device),
               "trunk": torch.cuda.Stream(device, priority=int(os.environ.get("IFSEG_TRUNK_PRIO", "-1"))), "probed": False}
        _cache[key] = out
        return out
    with torch.cuda.device(device):
        big = torch.randn(4096, 4096, device=device, dtype=torch.bfloat16)
        tiny = torch.zeros(64, device=device)

        def work():
            for _ in range(10):
                torch.mm(big, big)

        def pick(cands, against):
            for s in cands:
                with torch.cuda.stream(s):          # first use of a stream creates its queue: not part of the measurement
                    tiny.add_(1)
            torch.cuda.synchronize()
            for s in cands:
                if not any(_serialised(o, s, work, tiny) or _serialised(s, o, work, tiny) for o in against):
                    return s
            return None

        chosen, against = {}, [main]
        prio = int(os.environ.get("IFSEG_TRUNK_PRIO", "-1"))
        for name, pr in (("side", 0), ("dq", int(os.environ.get("IFSEG_DQ_PRIO", "0"))), ("trunk", prio)):
            cands = [torch.cuda.Stream(device, priority=pr) for _ in range(10)]
            s = pick(cands, against)
            chosen[name] = s if s is not None else cands[0]
            chosen.setdefault("fallback", []).append(name) if s is None else None
            against.append(chosen[name])
        torch.cuda.synchronize()
        del big, tiny
    chosen["probed"] = True
    _cache[key] = chosen
    return chosen
