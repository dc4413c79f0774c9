"""CPU oracle for the spoken-LID training hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain torch-CPU / numpy fp32 restatement of the reference's
algorithms on the hot path.  Every function cites the
reference file:line it follows.  It exists to CHECK the HIP path:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import it;
  * nothing under ``speech-lid_amd/`` imports it, and the product path raises
    when the HIP library is missing instead of falling back to this code.

Pinning status (see DESIGN.md "Oracle"):
  * Conformer encoder / heads / LangDiscriminator / CTC loss / Novograd /
    TriStage / Cavg / greedy CTC decode: PINNED against the reference itself,
    imported in the build container by ``oracle/gen_golden.py``; the vectors it
    wrote are committed under ``tests/golden/``.
  * log-mel front-end and SpecAugment: the arithmetic lives in torchaudio
    0.12.1 (pinned by lid/requirements/install.sh:4), which is absent from the
    reference tree and from this image -> PARITY UNPINNED against torchaudio;
    restated from its published algorithm and pinned by analytic known-answer
    tests plus an independent float64 numpy DFT restatement (tests/test_oracle_features.py).
"""
