#!/bin/bash
# What a cold `woltka classify` process pays before its first alignment byte moves.
t() { local s=$(date +%s%N); "$@" > /dev/null 2>&1; local e=$(date +%s%N); printf "%-86s %6d ms\n" "$*" $(( (e - s) / 1000000 )); }
for i in 1 2; do
t python -c "pass"
t python -c "import numpy"
t python -c "import click"
t python -c "import woltka_amd.workflow"
t python -c "import woltka_amd.cli"
t python -c "from woltka_amd import _native as n; n.load_library()"
t python -c "from woltka_amd import _native as n; c = n.Context(0)"
t python -c "from woltka_amd import _native as n; c = n.Context(0); c.host_alloc(65<<20)"
done
