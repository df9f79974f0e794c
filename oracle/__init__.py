"""oracle/ -- TEST INFRASTRUCTURE ONLY.

A CPU restatement (plain PyTorch CPU ops, fp32/fp64) of the LibContinual hot path named by
BASELINE.json `north_star`: ResNet forward/backward over a batch, the EWC / LwF / iCaRL / LUCIR
regularisation terms, Fisher-diagonal accumulation, herding, and SGD/Adam.  Every function cites
the reference file:line it follows (paths relative to /root/reference).

Pinning: the reference holds no tests or golden vectors for this path.  The
oracle is pinned against outputs of the reference itself, produced in the build container by
`oracle/gen_golden.py` (shimmed import of /root/reference, see `oracle/ref_shim.py`) and committed
as small fixtures under `tests/golden/`.  `tests/test_oracle_golden.py` re-checks the oracle against
those fixtures on every run.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this
package -- as the checker, never as the thing measured or shipped.  The product package
`libcontinual_amd` never imports it and has no CPU fallback.
"""
