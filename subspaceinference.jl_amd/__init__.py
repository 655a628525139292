"""subspaceinference.jl_amd -- MI355X-native hot path of SubspaceInference.jl behind the reference's own API.

The directory name carries a dot (it mirrors the reference repository's name), which Python cannot import
directly; `import subspaceinference_jl_amd` (the shim module at the repository root) loads this package.

Layout: csrc/ (HIP kernels + the C ABI of include/subspace_hip.h), _capi.py (ctypes binding), api.py (the
reference's exported functions), flux.py (caller-side stand-ins for the Flux objects the API takes),
dist.py (thin caller of the library's own RCCL communicator, si_comm_*; torch.distributed only ships the id and
serves as the CPU test double), julia/ (the ccall wrapper).
"""
from . import flux  # noqa: F401
from ._capi import BoundsError, Context, SubspaceError, host_sym_eig, host_sym_eig_top, load  # noqa: F401
from .api import inference, sub_inference, subspace_construction, subspace_inference  # noqa: F401

__all__ = ["subspace_construction", "subspace_inference", "sub_inference", "inference", "Context",
           "SubspaceError", "BoundsError", "flux", "load", "host_sym_eig", "host_sym_eig_top"]
