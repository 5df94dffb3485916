"""CPU oracle for the Tramba hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``tramba_amd/`` may import this
package: it is the checker, never the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it.

Every function is a from-scratch CPU restatement (numpy / plain torch-CPU / C)
of the algorithm a reference file implements and cites that file:line
(paths relative to the upstream reference tree).

Parity status ("Oracle pinning"):
  * scan-order tables, gathers/merges, DCT split, module arithmetic and the
    assembled models are PINNED against golden vectors produced by importing
    the Python reference in the build container
    (tests/golden/make_golden.py -> tests/golden/*.npz / *.json).
  * the selective-scan recurrence itself lives in a third-party CUDA extension
    (MzeroMiko/VMamba ``selective_scan_cuda_oflex``, version unpinned by the
    reference, sources absent): for that one function parity is UNPINNED; the
    oracle follows the published recurrence and is anchored by closed-form
    known-answer tests and an fp64 cross-check only.
"""
