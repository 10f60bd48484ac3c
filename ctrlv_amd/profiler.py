"""Optional per-launch timing of the C-ABI kernels with HIP events on the launch stream (bench.py's roofline leg).

Disabled by default: `ops` pays one `is None` check per call.  When enabled, every wrapper records an event pair on
torch's current stream (the stream the kernel is launched on) plus the launch's algorithmic FLOPs / bytes."""
import torch

_active = None


class KernelTimer:
    def __init__(self):
        self.records = []          # (family, flops, bytes, start_event, end_event)
        self.details = []          # per record: a shape key (tools/shape_table.py) or None

    def __enter__(self):
        global _active
        _active = self
        return self

    def __exit__(self, *a):
        global _active
        _active = None
        return False

    def summary(self):
        """family -> dict(calls, ms, flops, bytes); call after torch.cuda.synchronize()."""
        out = {}
        for fam, fl, by, s, e in self.records:
            d = out.setdefault(fam, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
            d["calls"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += fl
            d["bytes"] += by
        return out


def begin():
    if _active is None:
        return None
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    return ev


def end(start, family, flops=0.0, nbytes=0.0, detail=None):
    if start is None:
        return
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    _active.records.append((family, flops, nbytes, start, ev))
    _active.details.append(detail)
