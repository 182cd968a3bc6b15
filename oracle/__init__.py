"""CPU oracle for the StereoScene hot path (TEST INFRASTRUCTURE -- never shipped, never measured).

This package is a CPU restatement of the reference algorithm for SURVEY.md section 8 rows a1-a18.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it;
the product package ``stereoscene_amd`` never does (tests/test_layout.py enforces that).

Pinning status
--------------
* The reference ships NO tests, golden vectors or fixtures for this path.
* The restatement is therefore pinned against outputs of the reference itself, imported in the
  build container with throw-away shims (``oracle/make_golden.py``); inputs/outputs are committed
  under ``tests/golden/*.npz`` and checked by ``tests/test_oracle_golden.py``.
* Three operators the path reaches live in third-party packages that are NOT under
  /root/reference (``mmdet3d.ops.bev_pool`` / ``voxel_pooling`` from the OpenOccupancy mmdet3d
  fork, ``mmcv-full==1.4.0`` DCN, mmdet ``BasicBlock``).  For those the oracle restates the
  published algorithm and is anchored on the reference's call sites only:
  **parity unpinned** for exactly those three operators (stated again in DESIGN.md).
"""
