#!/bin/bash
# One-off check in the build container (needs /root/reference): run the
# reference's OWN unit tests against this package's host modules.  A scratch
# package named `woltka` re-exports woltka_amd.{file,tree,table,tools,align,
# workflow,ordinal,cli,ranges}; everything it lacks (the tests themselves,
# util, the biom glue) resolves to the reference.  Nothing is copied.
# Expected here (no GPU, no biom): test_file 13 passed, test_tools 5 passed,
# test_workflow 12 passed + the 4 device tests / 2 others failing for lack of
# a GPU, of biom, or for build_mapper's device default chunk.
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
REF=${REFERENCE_ROOT:-/root/reference}
[ -d "$REF/woltka" ] || { echo "no reference tree at $REF"; exit 0; }
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
mkdir -p "$T/woltka"
cat > "$T/woltka/__init__.py" <<PY
__version__ = '0.1.7'
__path__.append('$REF/woltka')
PY
for m in table tools file tree align cli workflow ordinal; do
  printf 'import woltka_amd.%s as _m\nglobals().update({k: getattr(_m, k) for k in dir(_m) if not k.startswith("__")})\n' $m > "$T/woltka/$m.py"
done
printf 'import woltka_amd.ranges as _m\nglobals().update({k: getattr(_m, k) for k in dir(_m) if not k.startswith("__")})\n' > "$T/woltka/range.py"
cat > "$T/stubplug.py" <<PY
import sys
sys.path.insert(0, '$REPO/tests/golden')
import _refshim
_refshim.install()                      # numba / biom stand-ins
sys.path[:] = [p for p in sys.path if p.rstrip('/') != _refshim.REFERENCE_ROOT]
for k in [k for k in sys.modules if k == 'woltka' or k.startswith('woltka.')]:
    del sys.modules[k]
sys.path.insert(0, '$T')
import woltka
assert woltka.__file__.startswith('$T'), woltka.__file__
PY
cd "$T"
for t in test_file test_tools test_workflow; do
  echo "== $t"
  PYTHONPATH="$T:$REPO" python -m pytest "$REF/woltka/tests/$t.py" --import-mode=importlib \
    -p no:cacheprovider -p stubplug -q --no-header 2>&1 | grep -E "passed|failed|FAILED" | cut -c1-160
done
