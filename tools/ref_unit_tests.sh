#!/bin/bash
# One-off check in the build container (needs /root/reference): run the
# reference's OWN unit tests against this package's host modules.  A scratch
# package named `woltka` re-exports woltka_amd.{file,tree,table,tools,align,
# workflow,ordinal,cli,ranges}; everything it lacks (the tests themselves,
# util, the biom glue) resolves to the reference.  Nothing is copied.
# Expected here (no GPU, no biom): test_file 13, test_tools 5, test_align 20,
# test_tree 10, test_classify 8 passed; test_workflow 12 passed + the 4 device
# tests / 2 others failing for lack of a GPU, of biom, or for build_mapper's
# device default chunk.  (find_rank / find_lca / the assigners / the counters
# have no host implementation in the product: for those two test modules the
# scratch package points at the ORACLE's restatements.)
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
# the hot-path functions exist only on the device and in the oracle: the
# reference's known-answer tests for them run against the ORACLE's restatements
cat >> "$T/woltka/tree.py" <<PY
import sys
sys.path.insert(0, '$REPO/oracle')
from woltka_oracle import ancestor_at_rank as find_rank, lowest_common_ancestor as find_lca
PY
cat > "$T/woltka/classify.py" <<PY
import sys
sys.path.insert(0, '$REPO/oracle')
from woltka_oracle import assign_none, assign_free, assign_rank, majority, count_float, count_sized
def counter(taxque):
    return count_float(taxque)
def counter_strat(qryque, taxque, strata):
    import woltka_oracle as o
    return {k: (int(v) if v.denominator == 1 else float(v)) for k, v in o.count_exact(taxque, qryque, strata).items()}
def counter_size(subque, taxque, sizes):
    return count_sized(subque, taxque, sizes)
def counter_size_strat(qryque, subque, taxque, sizes, strata):
    return count_sized(subque, taxque, sizes, qryque, strata)
PY
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
for t in test_file test_tools test_align test_tree test_classify test_workflow; do
  echo "== $t"
  PYTHONPATH="$T:$REPO" python -m pytest "$REF/woltka/tests/$t.py" --import-mode=importlib \
    -p no:cacheprovider -p stubplug -q --no-header 2>&1 | grep -E "passed|failed|FAILED" | cut -c1-160
done
