"""The C-ABI library builds, loads, and exports every symbol the header
declares.  No device work is done here (runs on the CPU-only box)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_in(header):
    with open(os.path.join(ROOT, 'include', header)) as f:
        text = f.read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(wk_[a-z_0-9]+)\s*\(', text)))


def declared_symbols():
    """The product header and the measurement header together: what the
    library exports."""
    return sorted(set(declared_in('woltka_hip.h')) |
                  set(declared_in('woltka_hip_measure.h')))


def test_measurement_entry_points_live_in_their_own_header():
    """knobs, timers and the benchmark's resident-text passes are
    not part of the drop-in surface."""
    product = set(declared_in('woltka_hip.h'))
    measure = set(declared_in('woltka_hip_measure.h'))
    assert not product & measure
    for name in ('wk_tune', 'wk_timer_begin', 'wk_timer_end', 'wk_timer_ms',
                 'wk_profile_kernels', 'wk_last_kernel_ms'):
        assert name in measure and name not in product
    assert 'wk_create' in product and 'wk_classify_chunk' in product


def test_header_and_binding_agree():
    from woltka_amd import _native
    assert sorted(_native.SYMBOLS) == declared_symbols()


def test_library_exports_all_symbols():
    from woltka_amd import _native
    assert os.path.isfile(_native.LIB_PATH), (
        'libwoltka_hip.so not built; run __graft_entry__.build()')
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.wk_abi_version() == _native.ABI_VERSION


def test_no_device_is_a_loud_error():
    """Without a GPU the product path must fail, not fall back."""
    from woltka_amd import _native
    lib = _native.load_library()
    h = ctypes.c_void_p()
    rc = lib.wk_create(0, ctypes.byref(h))
    if rc == 0:       # a GPU is present (GPU box): clean up and stop here
        lib.wk_destroy(h)
        pytest.skip('HIP device present')
    assert rc == _native.E_HIP
    assert b'no HIP device' in lib.wk_last_error(None)
    with pytest.raises(RuntimeError):
        _native.Context(0)


def test_product_does_not_import_oracle():
    """woltka_amd must never reach into oracle/ (parity claims depend on it)."""
    pkg = os.path.join(ROOT, 'woltka_amd')
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.hpp', '.h', '.cpp')):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                assert 'oracle' not in src.replace('no CPU fallback', ''), (
                    os.path.join(dirpath, fn))


def test_product_does_not_import_torch():
    """north_star: host code over a ctypes C ABI, no PyTorch -- not even an
    optional launcher inside the package (tools/torch_world.py is the adapter
    for ranks torch.distributed.run started)."""
    import re
    pkg = os.path.join(ROOT, 'woltka_amd')
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith('.py'):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                assert not re.search(r'^\s*(import|from)\s+torch\b', src,
                                     re.M), os.path.join(dirpath, fn)


def test_library_carries_the_digest_of_its_sources():
    """build() must have compiled the sources as they are now: the digest in
    the .so (wk_build_id) equals the digest of csrc/ + the header."""
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    from woltka_amd import _native
    g.build_native()
    assert g.built_digest() == g.source_digest(g._sources())
    assert _native.build_id() == g.built_digest()
