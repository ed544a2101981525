"""Test-infrastructure stand-in for the absent third-party package `torch_geometric` (see nn/conv, data)."""
