"""Device context helpers over the C ABI: init, stream, events, tuning knobs.

One process drives one GPU (include/jetship.h conventions).  `init()` is called lazily by the
first array factory; it raises JetsHipError when no gfx950 device is visible -- the block-operator
path has no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

from ._ffi import lib, check

__all__ = ["init", "is_initialized", "shutdown", "synchronize", "device_count", "device_info", "stream_handle",
           "set_stream", "Event", "tune", "tune_get", "local_device_from_env"]

_state = {"device": None}


def device_count() -> int:
    n = C.c_int(0)
    check(lib.jh_device_count(C.byref(n)))
    return n.value


def local_device_from_env() -> int:
    """LOCAL_RANK -> device ordinal (one rank per GPU under torch.distributed.run)."""
    return int(os.environ.get("LOCAL_RANK", "0"))


def init(device: int | None = None) -> int:
    if _state["device"] is not None and (device is None or device == _state["device"]):
        return _state["device"]
    dev = local_device_from_env() if device is None else int(device)
    check(lib.jh_init(dev))
    _state["device"] = dev
    return dev


def is_initialized() -> bool:
    return _state["device"] is not None


def shutdown() -> None:
    check(lib.jh_shutdown())
    _state["device"] = None


def synchronize() -> None:
    check(lib.jh_synchronize())


def device_info() -> dict:
    init()
    name = C.create_string_buffer(256)
    tot, fr, cu = C.c_int64(0), C.c_int64(0), C.c_int(0)
    check(lib.jh_device_info(name, 256, C.byref(tot), C.byref(fr), C.byref(cu)))
    return {"name": name.value.decode(), "total_mem": tot.value, "free_mem": fr.value, "cu_count": cu.value}


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
