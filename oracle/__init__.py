"""oracle — CPU restatement of the reference algorithm for the hot path.

TEST INFRASTRUCTURE ONLY. Nothing under `llm-inference-lab_amd/` imports this
package. The only importers are `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py`, and there it is the checker or the reported
baseline, never the thing measured or shipped.

Each function cites the reference file:line it restates. The restatements are
pinned against golden vectors produced by importing the reference itself in the
build container (tests/golden/make_golden.py; the reference never travels).
"""
