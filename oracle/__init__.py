"""CPU oracle for the raw-audio VAE training step -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement of the arithmetic performed by the reference
hot path (`/root/reference/rawvae/model.py:5-47` plus `torch.optim.Adam` as
constructed at `/root/reference/train.py:163`).  It exists so that the HIP
kernels can be checked on a GPU box where the reference itself is absent.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it.  Nothing under `rawaudiovae_kelsey_amd/` or `rawvae/`
imports it; the product path raises when the HIP library is missing instead of
falling back to this code.

Parity pin: the reference carries no tests or golden vectors of its own
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, captured in this container by `tools/make_golden.py` and committed
under `tests/golden/` (`tests/test_oracle_golden.py` is the check).
"""
