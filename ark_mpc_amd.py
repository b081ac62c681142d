"""Import shim: the package directory is `ark-mpc_amd/` (hyphen), which `import` cannot spell."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("ark-mpc_amd")
