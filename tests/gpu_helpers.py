"""Test-side helpers for the -m gpu tests (not product code)."""


def dev_view(ptr, shape, strides_elems=None):
    """torch view of library-owned device memory (fp64) through __cuda_array_interface__, for test-side reductions over
    buffers too large to pull to the host (cfg5: P is 26 GB).  shape / strides in elements, torch (row-major) order."""
    import torch
    iface = {"shape": tuple(int(v) for v in shape), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
    if strides_elems is not None:
        iface["strides"] = tuple(8 * int(v) for v in strides_elems)

    class _Buf:
        __cuda_array_interface__ = iface
    return torch.as_tensor(_Buf(), device="cuda")
