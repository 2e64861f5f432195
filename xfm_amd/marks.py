"""Step marks (XFM_MARKS=1; off: one attribute test per call): host time and a GPU event on the current stream at named points of a
step -- tower forward / backward boundaries, the grouped weight-gradient launch, the optimizer.  Where the GPU reaches a mark long after
the host queued it the GPU is the bound; where the two coincide the stream was waiting for the host.  rocprofv3 cannot tell these
apart: its per-launch cost slows the host until every stream looks host-fed.  tools/step_marks.py prints the table."""
import os
import time

import torch

ON = os.environ.get("XFM_MARKS", "0") != "0"
_marks = []


def mark(label):
    if not ON or not torch.cuda.is_available():
        return
    ev = torch.cuda.Event(enable_timing=True)
    s = torch.cuda.current_stream()
    ev.record(s)
    _marks.append((label, time.perf_counter(), ev, s.cuda_stream))


def reset():
    _marks.clear()


def mean_table(step_label="step begin"):
    """The marks of several steps (each opened by `step_label`) averaged position by position; steps whose mark sequence differs
    from the first one's are left out.  -> ([(label, stream, host ms, gpu ms)], steps averaged)"""
    if not _marks:
        return [], 0
    torch.cuda.synchronize()
    steps, streams = [], {}
    for lab, h, ev, st in _marks:
        if lab == step_label:
            steps.append([])
        if steps:
            steps[-1].append((lab, streams.setdefault(st, len(streams)), h, ev))
    rows = [[(lab, st, (h - stp[0][2]) * 1e3, stp[0][3].elapsed_time(ev)) for lab, st, h, ev in stp] for stp in steps]
    shape = [(r[0], r[1]) for r in rows[0]]
    rows = [r for r in rows if [(x[0], x[1]) for x in r] == shape]
    n = len(rows)
    return [(shape[i][0], shape[i][1], sum(r[i][2] for r in rows) / n, sum(r[i][3] for r in rows) / n) for i in range(len(shape))], n


def table():
    """[(label, stream, host ms, gpu ms)] relative to the first mark (synchronizes)."""
    if not _marks:
        return []
    torch.cuda.synchronize()
    l0, h0, e0, _ = _marks[0]
    streams = {}
    return [(lab, streams.setdefault(st, len(streams)), (h - h0) * 1e3, e0.elapsed_time(ev)) for lab, h, ev, st in _marks]
