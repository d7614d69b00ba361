"""CPU oracle for the Emu2 inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and there only as the checker (never as the
thing that is measured or shipped).  The product (``emu_amd``) fails loudly when
its HIP library is missing; it never falls back to this code.

Parity status
-------------
* ViT / projector / LLaMA / ``generate`` / ``generate_image`` restatements
  (``oracle/emu2_ref.py``) are PINNED: ``tests/test_oracle_golden.py`` checks them
  against outputs of the reference itself (``/root/reference/Emu2/emu`` imported
  with the two-symbol timm shim, ``oracle/ref_import.py``), frozen as fixtures in
  ``tests/golden/`` by ``oracle/make_golden.py``.
* UNet / Euler scheduler restatement: PARITY UNPINNED (diffusers==0.24.0 is a
  third-party dependency that is neither vendored in the reference nor installed
  here); see DESIGN.md.
"""
