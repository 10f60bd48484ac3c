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


class PlanTimer:
    """The same summary from the C++ plan's OWN launches (ctrlv_plan_profile): what `bench.py` reports, so that the
    roofline describes the executor the timed region runs.  Usage: `with PlanTimer(unet, controlnet) as t: step()`, then
    `t.summary()` / `t.launches` (the models must have run one forward before, so that their plans exist)."""

    def __init__(self, *models):
        self.plans = [m._plan for m in models if m is not None and getattr(m, "_plan", None) is not None]
        self.launches = []         # (family, ms, flops, bytes, (M, N, K, flags)) in launch order, per plan

    def __enter__(self):
        for p in self.plans:
            p.profile(True)
        return self

    def __exit__(self, *a):
        torch.cuda.synchronize()
        for p in self.plans:
            self.launches += p.profile_read()
            p.profile(False)
        return False

    def summary(self):
        out = {}
        for fam, ms, fl, by, _ in self.launches:
            d = out.setdefault(fam, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
            d["calls"] += 1
            d["ms"] += ms
            d["flops"] += fl
            d["bytes"] += by
        return out
