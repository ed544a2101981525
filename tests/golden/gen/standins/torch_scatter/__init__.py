"""Test-infrastructure stand-in for the absent third-party package `torch_scatter`.

Call sites in the reference: model.py:5,75 and data_loader.py:14,126. Published semantics of
scatter_add(src, index, dim, dim_size): zero-initialised output with `dim_size` entries along
`dim`, out[index[i]] += src[i], accumulation in index order on CPU. Not product code.
"""
import torch


def scatter_add(src, index, dim=-1, out=None, dim_size=None):
    if dim < 0:
        dim += src.dim()
    if out is None:
        if dim_size is None:
            dim_size = int(index.max()) + 1 if index.numel() else 0
        shape = list(src.shape)
        shape[dim] = dim_size
        out = torch.zeros(shape, dtype=src.dtype, device=src.device)
    return out.index_add_(dim, index, src)
