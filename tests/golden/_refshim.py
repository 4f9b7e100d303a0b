"""Import shims that let the *real* reference package under /root/reference be
imported in the build container (never on the GPU box, never shipped).

The reference needs `numba` (absent here) only to JIT three functions; it ships
its own no-JIT branch (selected by ``numba.config.DISABLE_JIT``), which is what
its CI exercises.  ``biom`` (absent) is needed only for BIOM file I/O.  This
module installs minimal stand-in *modules* for those two third-party imports so
that the reference's own, unmodified code runs.  It is test infrastructure for
generating golden vectors; nothing in the product imports it.
"""
import sys
import types

REFERENCE_ROOT = '/root/reference'


def install():
    """Install the stubs and put the reference on sys.path. Returns True if the
    reference tree exists (False on the GPU box)."""
    import os
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, 'woltka')):
        return False

    if 'numba' not in sys.modules:
        nb = types.ModuleType('numba')

        def njit(*a, **k):
            if len(a) == 1 and callable(a[0]) and not k:
                return a[0]
            return lambda f: f
        nb.njit = njit
        cfg = types.SimpleNamespace(DISABLE_JIT=True)
        nb.config = cfg

        class _T:
            def __getitem__(self, _):
                return self

            def __call__(self, *a, **k):
                return self
        typed = types.ModuleType('numba.typed')
        typed.Dict = dict
        typed.List = list
        ntypes = types.ModuleType('numba.types')
        for name in ('uint32', 'int64', 'boolean', 'uint64', 'int32'):
            setattr(ntypes, name, _T())
        nb.typed = typed
        nb.types = ntypes
        sys.modules['numba'] = nb
        sys.modules['numba.typed'] = typed
        sys.modules['numba.types'] = ntypes

    if 'biom' not in sys.modules:
        try:
            import biom  # noqa: F401
        except ImportError:
            bm = types.ModuleType('biom')

            class Table:  # placeholder; BIOM I/O is never exercised
                def __init__(self, *a, **k):
                    raise RuntimeError('biom-format is not installed')
            bm.Table = Table
            bm.load_table = lambda *a, **k: (_ for _ in ()).throw(
                RuntimeError('biom-format is not installed'))
            util = types.ModuleType('biom.util')
            util.biom_open = None
            bm.util = util
            sys.modules['biom'] = bm
            sys.modules['biom.util'] = util

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    return True
