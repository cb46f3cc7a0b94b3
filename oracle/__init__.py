"""CPU oracle for the CLIP encode -> similarity -> top-k / Recall@K hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline, never as the thing shipped.  The product path
(``knowledge_enhanced_multimodal_retrieval_amd``) never imports this package
and fails loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):

* ``metrics_ref`` / ``fusion_ref`` / ``heads_ref`` are pinned by golden vectors
  generated in the build container from the reference's own importable modules
  (``/root/reference/src/clip/eval/metrics.py``, ``eval/fusion.py``,
  ``model/fusion_model.py``, ``src/retrieval.py`` logic) with
  ``tests/golden/make_golden.py``; the vectors are committed under
  ``tests/golden/``.
* ``clip_ref`` restates the encoder arithmetic of the third-party ``clip``
  package (github.com/openai/CLIP, unpinned in the reference's scripts,
  ``scripts/baselines/run_clip_base_l14.sh:10``), which is absent from
  ``/root/reference`` and from this image.  The reference holds no test or
  golden vector for the encoders, so against the reference itself the encoder
  parity is UNPINNED.  The restatement is cross-checked (<= 1e-5) against
  ``transformers.CLIPModel`` built from a local config object, the class the
  reference calls in ``src/clip/eval/evaluator_hf.py:115,130,144``; those
  cross-check vectors are committed under ``tests/golden/`` as well.
"""
