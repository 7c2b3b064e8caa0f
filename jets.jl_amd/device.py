"""Device context helpers over the C ABI: init, stream, events, tuning knobs.

The usual deployment is one process per GPU with the device's primary context (include/jetship.h conventions).
`init()` is called lazily by the first array factory; it raises JetsHipError when no gfx950 device is visible --
the block-operator path has no CPU fallback.

One process may also hold SEVERAL contexts (other GPUs, or further streams of one GPU): `context_create`,
`context_use`, `using_context`.  Array factories allocate in the CURRENT context; every operation on arrays /
operators runs in THEIR context, whatever is current (the ABI switches per call and refuses mixed handles).
"""
from __future__ import annotations

import ctypes as C
import os

from ._ffi import lib, check

__all__ = ["init", "is_initialized", "shutdown", "synchronize", "device_count", "device_info", "trim", "stream_handle",
           "set_stream", "Event", "tune", "tune_get", "local_device_from_env", "context_create", "context_use",
           "context_current", "context_destroy", "using_context", "context_of"]

_state = {"device": None, "devices": set(), "multi": False}


def device_count() -> int:
    n = C.c_int(0)
    check(lib.jh_device_count(C.byref(n)))
    return n.value


def local_device_from_env() -> int:
    """LOCAL_RANK -> device ordinal (one rank per GPU under torch.distributed.run)."""
    return int(os.environ.get("LOCAL_RANK", "0"))


def init(device: int | None = None) -> int:
    """The primary context of `device` (created on first use) becomes current.  Without an argument: nothing to do once any
    context exists, else the device of LOCAL_RANK."""
    if _state["device"] is not None and (device is None or (device == _state["device"] and not _state["multi"])):
        return _state["device"]
    dev = local_device_from_env() if device is None else int(device)
    check(lib.jh_init(dev))
    if _state["device"] is None:
        _state["device"] = dev
    _state["devices"].add(dev)
    if len(_state["devices"]) > 1:
        _state["multi"] = True
    return dev


def is_initialized() -> bool:
    return _state["device"] is not None


def shutdown() -> None:
    check(lib.jh_shutdown())
    _state.update(device=None, devices=set(), multi=False)


def context_create(device: int | None = None) -> int:
    """An ADDITIONAL context (own stream, workspaces, knobs) on `device` (default: the first device used); becomes current."""
    dev = (_state["device"] if _state["device"] is not None else init()) if device is None else int(device)
    if _state["device"] is None:
        init(dev)
    ctx = C.c_int(-1)
    check(lib.jh_context_create(dev, C.byref(ctx)))
    _state["multi"] = True
    return ctx.value


def context_use(ctx: int) -> None:
    check(lib.jh_context_use(int(ctx)))


def context_current():
    """(context id, device) of the calling thread's current context."""
    init()
    ctx, dev = C.c_int(-1), C.c_int(-1)
    check(lib.jh_context_current(C.byref(ctx), C.byref(dev)))
    return ctx.value, dev.value


def context_destroy(ctx: int) -> None:
    check(lib.jh_context_destroy(int(ctx)))


def context_of(x) -> int:
    """The context a device array lives in."""
    ctx = C.c_int(-1)
    check(lib.jh_bvec_context(x.handle, C.byref(ctx), None))
    return ctx.value


class using_context:
    """`with using_context(ctx): ...` -- array factories and handle-less calls inside act on `ctx`; the previous context is
    current again afterwards."""

    def __init__(self, ctx: int):
        self.ctx = int(ctx)

    def __enter__(self):
        self.prev = context_current()[0]
        context_use(self.ctx)
        return self.ctx

    def __exit__(self, *exc):
        context_use(self.prev)
        return False


def several_contexts() -> bool:
    return _state["multi"]


def synchronize() -> None:
    check(lib.jh_synchronize())


def device_info() -> dict:
    init()
    name = C.create_string_buffer(256)
    tot, fr, cu = C.c_int64(0), C.c_int64(0), C.c_int(0)
    check(lib.jh_device_info(name, 256, C.byref(tot), C.byref(fr), C.byref(cu)))
    return {"name": name.value.decode(), "total_mem": tot.value, "free_mem": fr.value, "cu_count": cu.value}


def trim() -> None:
    """Give the device memory the slab cache holds (destroyed vectors of 16 MiB or more, kept for the next vector of their size) back
    to the driver -- before another library of the process needs it.  include/jetship.h: jh_trim."""
    init()
    check(lib.jh_trim())


def stream_handle() -> int:
    """The hipStream_t (as an int) the library enqueues on; wrap it with torch.cuda.ExternalStream to
    order torch.distributed collectives against the kernels."""
    init()
    p = C.c_void_p()
    check(lib.jh_get_stream(C.byref(p)))
    return p.value or 0


def set_stream(handle: int | None) -> None:
    init()
    check(lib.jh_set_stream(C.c_void_p(handle) if handle else None))


class Event:
    """HIP event on the library stream (roofline timing, bench.py)."""

    def __init__(self):
        init()
        self._h = C.c_void_p()
        check(lib.jh_event_create(C.byref(self._h)))

    def record(self) -> "Event":
        check(lib.jh_event_record(self._h))
        return self

    def elapsed_ms(self, stop: "Event") -> float:
        ms = C.c_float(0)
        check(lib.jh_event_elapsed_ms(self._h, stop._h, C.byref(ms)))
        return float(ms.value)

    def __del__(self):
        try:
            if self._h:
                lib.jh_event_destroy(self._h)
                self._h = None
        except Exception:
            pass


def tune(**knobs) -> None:
    """Kernel-shape knobs (bench/tests): fwd_group, fwd_unroll, fwd_wg, adj_unroll, adj_depth, adj_wg (0 = pick from the
    problem size), fwd_order, nt, autotune, graphs, general_xcd (1 automatic / 0 never / 2 always) and -- the one with a
    numerical meaning -- adj_split: -1 (default) lets the kernels that sum over block rows / columns cut a long sum of
    SMALL blocks into parts (deterministic, tolerance parity), 0 keeps the reference's ordered sum
    always (bit-exact against the sequential CPU loop, up to 100x slower on such shapes), k > 1 forces k parts."""
    for k, v in knobs.items():
        check(lib.jh_tune_set(k.encode(), int(v)))


def tune_get(name: str) -> int:
    v = C.c_int64(0)
    check(lib.jh_tune_get(name.encode(), C.byref(v)))
    return v.value
