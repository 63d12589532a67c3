"""CPU oracle for the instance-field NeRF render/train hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  Nothing under ``instance_nerf_amd/`` imports it;
the product path fails loudly when the HIP library is missing instead of
falling back to anything in here.

PARITY UNPINNED.  The reference's implementation of this path lives in an
un-vendored git submodule (``/root/reference/.gitmodules:4-6`` ->
``zymk9/torch-ngp`` branch ``instance_nerf``, pinned only in prose at
``/root/reference/README.md:27,59`` to commit
``6be6af198f1092e8d75574727a030ae15e199fe8``).  The directory
``/root/reference/instance_nerf`` is empty, the reference ships no tests and no
golden vectors, and there is no network.  This oracle is
therefore a restatement of the *published* torch-ngp / Instant-NGP algorithms
as recorded in SURVEY.md Appendix A, and every parity statement made against
it reads "vs. this repository's CPU oracle", never "vs. the reference".

Where Appendix A leaves a choice open, the choice made here is canonical for
this repository and is documented on the function that makes it:

* sample slots of ``march_rays_train`` are assigned by an exclusive scan of the
  per-ray counts in ray order (upstream: racing ``atomicAdd``), so offsets are
  deterministic;
* per-level ``scale``/``resolution`` of the hash grid are computed once on the
  host in float64 and rounded to float32 (upstream: ``exp2f`` on the device),
  so both sides index from the same table;
* all ray-marching arithmetic is strict IEEE fp32 with no fused multiply-add,
  in the operation order written in ``march.py``.
"""

from . import rays, occupancy, march, hashgrid, sh, field, composite, render, consumers, roialign  # noqa: F401
