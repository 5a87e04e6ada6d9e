"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A CPU restatement (torch-CPU fp32 / numpy) of the reference's algorithm for the tiled
panoramic denoising hot path.  Every function cites the reference
file:line it follows (paths relative to /root/reference).

Who may import this package: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` -- as the checker / the timed CPU baseline, never as the thing shipped.
`dynamicscaler_amd/` must not import it (tests/test_no_oracle_in_product.py enforces that).

Pinning: the reference holds no tests or golden vectors for this path,
so the restatement is pinned against outputs of the reference itself, run in the build
container by `tests/golden/make_golden.py` (committed) with the vectors committed under
`tests/golden/`.  `tests/test_oracle_golden.py` checks every function here against them.
"""
