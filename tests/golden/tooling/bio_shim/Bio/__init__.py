"""Minimal stand-in for Biopython, used ONLY by tests/golden/tooling/make_golden.py
in the build container to import the reference (Biopython is not installed and
there is no network).  It is golden-vector tooling: never imported by the
product (insilicoseq_amd/), never shipped to the GPU box as a dependency.
Covers exactly what the reference hot path touches."""
