# coding=utf-8
"""CPU: the C-ABI library builds/loads and exports every symbol include/dccf_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from conftest import REPO


def header_functions():
    src = open(os.path.join(REPO, 'include', 'dccf_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'^(?:int|int64_t|const char\*)\s+(\w+)\s*\(', src, flags=re.M)))


def test_library_exports_every_declared_symbol():
    from dccf_amd import build, _lib
    path = build.build(verbose=False)        # no-op when the in-tree .so is up to date
    lib = ctypes.CDLL(path)
    names = header_functions()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), 'libdccf_hip.so does not export %s' % n
    assert sorted(_lib.EXPORTS) == names
    loaded = _lib.load()
    assert loaded.dccf_abi_version() == 6      # 2: extra mlp layers in dccf_model_t / dccf_grads_t; 3: dccf_opt_t.lazy_*; 4: lazy_scal holds 4 floats
                                               # per step; 5: claims / lists by step parity, dccf_opt_t.lazy_list_cap; 6: dccf_opt_t.lazy_host,
                                               # dccf_ctx_set_deterministic
    assert loaded.dccf_last_error() is not None


def test_no_product_import_of_the_oracle():
    """The product path must never route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(REPO, 'dccf_amd')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                txt = open(os.path.join(root, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M), f
