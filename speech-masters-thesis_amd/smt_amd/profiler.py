"""Live per-kernel timing with HIP events on the launch stream (bench.py's ``roofline``).

Native ops wrap their library call in ``with profiler.region(name, bytes=..., flops=...)``.
Kernels are launched on torch's current stream and the events are recorded on that same
stream, so the elapsed time is the kernel's own duration.  Disabled (zero overhead) unless
``enable(True)`` was called.
"""
import contextlib
from collections import defaultdict

import torch

HBM_PEAK_GBS = 8000.0
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}
DOMINANT = "vq_forward"      # kernel group reported under "roofline" (updated as kernels land)

_enabled = False
DETAIL = bool(int(__import__("os").environ.get("SMT_PROFILE_DETAIL", "0")))  # per-geometry kernel names
_records = defaultdict(list)  # name -> [(start, end, bytes, flops, bound, dtype)]


def enable(flag):
    global _enabled
    _enabled = bool(flag)


def reset():
    _records.clear()


@contextlib.contextmanager
def region(name, nbytes=0, flops=0, bound="hbm", dtype="f32"):
    if not _enabled:
        yield
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    yield
    e.record()
    _records[name].append((s, e, nbytes, flops, bound, dtype))


def summary():
    torch.cuda.synchronize()
    out = []
    for name, recs in _records.items():
        us = [s.elapsed_time(e) * 1e3 for s, e, *_ in recs]
        avg_us = sum(us) / len(us)
        nbytes = sum(r[2] for r in recs) / len(recs)
        flops = sum(r[3] for r in recs) / len(recs)
        bound, dtype = recs[0][4], recs[0][5]
        if bound == "hbm":
            achieved, peak, unit = nbytes / avg_us * 1e-3, HBM_PEAK_GBS, "GB/s"
        else:
            achieved, peak, unit = flops / avg_us * 1e-6, PEAK_TFLOPS[dtype], "TFLOP/s"
        out.append({"name": name, "launches": len(recs), "avg_us": avg_us, "min_us": min(us), "max_us": max(us),
                    "total_ms": sum(us) * 1e-3,
                    "bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak,
                    "alg_bytes": nbytes, "alg_flops": flops,
                    "alt_tflops": flops / avg_us * 1e-6 if flops else None})
    out.sort(key=lambda r: -r["total_ms"])
    return out
