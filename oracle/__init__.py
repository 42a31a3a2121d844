"""CPU oracle for the SSAK acoustic-model hot path.  TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement (numpy / eager torch-CPU fp32) of the
arithmetic the reference reaches through ``transformers`` / ``torch`` on its hot path
(SURVEY.md section 8a).  It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  The product (``ssak_amd``) never imports
this package and fails loudly when the HIP library is missing.

Pinning: the restatement is checked against golden vectors produced in the build container by
the third-party classes the reference calls (``transformers.Wav2Vec2ForCTC``,
``Wav2Vec2FeatureExtractor``, ``WhisperFeatureExtractor``, ``torch.nn.functional.ctc_loss``,
``torch.optim.AdamW``); the generator is ``oracle/gen_golden.py`` and the vectors live in
``tests/golden/``.  The reference's own end-to-end goldens (``tests/expected/...``) need
pretrained weights that cannot be fetched offline, so end-to-end parity on those is unpinned
(SURVEY.md section 8c); stage-level parity against the third-party oracle is pinned.
"""
