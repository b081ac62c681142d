"""ark-mpc_amd -- MI355X-native engine for ark-mpc's batched authenticated-share hot path.

The product is the C-ABI shared library built from csrc/ (include/arkmpc.h).  This package is
plumbing: a ctypes binding of that ABI (engine.py) and the multi-GPU sharding helpers (sharding.py).  The host-side
mirror of the reference's fabric API is C++ (host/fabric.hpp, drivers host/mock_mpc_main.cpp and host/bench_main.cpp).  There is no CPU fallback: importing the
binding without the built library, or creating a context without a GPU, raises.

The directory name carries a hyphen, so import it with
    importlib.import_module("ark-mpc_amd")      or      import ark_mpc_amd   (root-level shim)
"""
from .engine import (  # noqa: F401
    ArkMpcError,
    Engine,
    Group,
    FIELD_IDS,
    FIELD_MODULI,
    lib_path,
    load_library,
    exported_symbols,
)

__all__ = ["ArkMpcError", "Engine", "Group", "FIELD_IDS", "FIELD_MODULI", "lib_path", "load_library", "exported_symbols"]
