"""CPU oracle for the VQ-W-Net training hot path.  TEST INFRASTRUCTURE ONLY.

Everything under oracle/ is a checker: only tests/, __graft_entry__.smoke() and
bench.py's `cpu_baseline` leg may import it.  The shipped path
(medical-image-editing_amd/) never imports, links or executes anything here and
raises if its HIP library is missing.

Parity status: PINNED — vqwnet_ref.py is checked in tests/test_oracle_golden.py
against golden vectors produced by the upstream reference's own modules
(tests/golden/make_golden.py, run in the build container where
/root/reference is mounted).
"""
