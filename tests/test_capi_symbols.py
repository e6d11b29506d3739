"""The C-ABI library loads and exports every symbol include/mzhip.h declares (no compute, no GPU)."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'mzhip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mz_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_exported_and_bound():
    from muzero_amd import _lib
    names = declared_symbols()
    assert len(names) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), f'libmzhip.so does not export {name}'
    assert sorted(_lib.PROTOTYPES) == names, 'muzero_amd/_lib.py prototypes out of sync with include/mzhip.h'
    bound = _lib.load()
    assert bound.mz_version() == 1
    assert bound.mz_last_error() is not None


def test_pool_create_fails_loudly_without_a_gpu():
    import pytest
    import torch
    from muzero_amd.exceptions import MuZeroError
    from muzero_amd.tree_pool import TreePool
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    with pytest.raises(MuZeroError):
        TreePool(num_trees=1, num_actions=2, max_simulations=1)
