"""CPU oracle for the BUSCA track-recovery hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement of the reference's algorithm
(/root/reference/busca, cited function by function) and is used solely as the *checker*:
only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it.
The product path (`busca_amd/`) never imports it and fails loudly without its HIP library.

How it is pinned: the reference has no tests and no golden vectors, so
`tests/golden/make_golden.py` imports the reference itself in the build container (with the import
shims in `oracle/ref_shims/`), runs it on seeded inputs, and commits the outputs under
`tests/golden/`.  `tests/test_oracle_golden.py` checks every oracle function against those vectors.
Third-party arithmetic that is not in /root/reference (OpenCV resize, positional_encodings 6.0.3,
cython_bbox) is restated from its published algorithm and is PARITY UNPINNED (stated per function).
"""
