"""jets.jl_amd -- MI355X-native block-operator mul! path behind the Jets.jl operator API.

The directory name carries a dot (it follows the reference's repository name), so it is imported
through the `jets_jl_amd` shim at the repository root:

    import jets_jl_amd as Jets
    A = Jets.blockop([[Jets.JopDiagonal(Jets.rand(R))] for _ in range(64)])
    d = A * m;  mt = A.H * d

Layers: `_ffi` (ctypes binding of include/jetship.h) -> `device`, `spaces`, `arrays` (BlockArray on
one HIP slab) -> `jets` (Jet/Jop/mul!/composition/sums) -> `jetblock` (JetBlock loops, fused launch).
There is no CPU fallback: importing needs libjetship.so, computing needs a gfx950 device.
"""
from ._ffi import JetsHipError, LIB_PATH  # noqa: F401
from . import device  # noqa: F401
from .device import init, synchronize, shutdown, Event, tune, tune_get, device_info, device_count, stream_handle, set_stream, trim  # noqa: F401
from .device import context_create, context_use, context_current, context_destroy, using_context, context_of  # noqa: F401
from .spaces import *  # noqa: F401,F403
from .arrays import *  # noqa: F401,F403
from .jets import *  # noqa: F401,F403
from .jets import range_ as range  # noqa: F401,A001  Jets.range(A)
from .jets import register_close, register_perfstat  # noqa: F401
from .jetblock import *  # noqa: F401,F403
from .broadcast import *  # noqa: F401,F403
from .symmetric import SymmetricArray, symspace  # noqa: F401
from .lsqr import lsqr, LsqrResult  # noqa: F401
from .cgls import cgls, cgnr  # noqa: F401
from . import rowpart  # noqa: F401
from .placement import stream_pair, probe_stream_direction  # noqa: F401
