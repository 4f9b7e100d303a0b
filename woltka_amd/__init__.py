"""woltka_amd — MI355X-native implementation of the `woltka classify` hot
path (per-read subject->taxon assignment, multi-hit LCA, coord-match of reads
to genes, per-sample count reduction).  Python host layer over hand-written
HIP kernels for gfx950, reached through the C ABI in include/woltka_hip.h."""

__version__ = '0.1.0'
