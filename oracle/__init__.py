"""CPU oracle for the style-transfer optimisation hot path.

TEST INFRASTRUCTURE ONLY.  This package restates, with plain ``torch`` CPU
fp32 ops, the algorithm of the reference's per-step path
(/root/reference/src/style_transfer_visualizer/core_model.py:29-350 and
optimization.py:274-327).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it, and only as the checker:
nothing under ``style_transfer_visualizer_amd/`` imports or falls back to it.

Pinning: the restatement is checked against golden vectors produced by the
*unmodified* reference modules imported from /root/reference in the build
container (``oracle/make_golden.py`` -> ``tests/golden/*.npz``); see
``tests/test_oracle_golden.py``.  The reference's own tests hold no numeric
fixtures for this path beyond the Gram known-answer values in SURVEY.md §8(c),
which are pinned too.
"""
