"""CPU oracle for the 3DTopia-XL DDIM / PrimX-DiT / VAE-decode hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``3dtopia-xl_amd/`` imports this package; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and only as the checker.

It is a from-scratch functional restatement (plain functions over a ``state_dict``; numpy float64
for the schedule bookkeeping, torch-CPU fp32 for the network arithmetic) of the reference's
algorithm; every function cites the reference file:line it follows.

Parity pinning: the reference ships no tests or golden vectors for this path.
The oracle is therefore pinned against OUTPUTS OF THE REFERENCE ITSELF: ``tests/golden/make_golden.py``
imports the unmodified reference modules from /root/reference (with an in-memory stand-in for the
absent third-party ``xformers.ops``: fp32 softmax(q k^T / sqrt(K)) v on [B,M,H,K] - xformers is
un-vendored and un-pinned, README.md:67) on deterministic synthetic weights/inputs and commits the
results as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this oracle against them.
"""
