"""oracle/ -- CPU restatement of the hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package,
and only as the checker / the reported CPU baseline.  Nothing under ``wheeledlab_amd/`` imports it; the product
path fails loudly when the HIP extension is missing instead of falling back to this code.

Parity status:
  * pinned   -- every function that restates arithmetic the reference OWNS (reward / termination / observation /
                action-term / reset / curriculum / traversability): checked against ``tests/golden/*.npz``, which
                were produced by importing the reference's own functions (``tests/golden/gen_golden.py``).
  * unpinned -- restatements of un-vendored IsaacLab 2.0.2 helpers (``mathlib``: euler/quaternion conversions,
                manager ordering, noise model, DC-motor clip) -- checked by self-consistency + known answers only.
  * designed -- the rigid-body / tyre integrator (``vehicle``): PhysX is closed source and the USD assets are
                missing, so there is nothing to be equal to; the oracle is the executable spec the HIP kernel must
                match to fp32 tolerance.
"""
