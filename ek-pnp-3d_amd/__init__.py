"""ek-pnp-3d_amd — host side of the MI355X-native EK-PNP hot path.

The product is the C-ABI shared library `libekpnp.so` (include/ekpnp.h, sources in csrc/);
this package is a thin ctypes mirror of it whose method names follow the reference's host
functions (LBM.h:159-180): initialization, init_equilibrium, stream_collide_save,
fast_Poisson.  There is NO CPU fallback: importing works without a GPU (so that the symbol
table can be checked), but creating a Solver without the library or without a HIP device
raises.
"""
from .solver import (  # noqa: F401
    FIELDS,
    FIELD_ID,
    EkpnpError,
    Group,
    Params,
    STAGE_NAMES,
    Solver,
    TRANSPORT_AUTO,
    TRANSPORT_COPY,
    TRANSPORT_RCCL,
    comm_unique_id,
    compute_parameters,
    default_params,
    exported_symbols,
    library_path,
    load_library,
    rccl_available,
    slab_extent,
)
