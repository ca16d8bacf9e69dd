"""CPU oracle: restatements of the reference's hot-path arithmetic. TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker. Nothing under ``audiotoken_amd/`` imports it.

Parity status:
* reference-pinned by import (golden vectors made from the reference's own files in the build
  container, see tests/golden/make_golden.py): fbank front-end (A5), rel-pos attention (A6),
  chunk/pad/trim/save harness (A9).
* dependency arithmetic pinned against HF `transformers` 5.15.0 restatements with seeded synthetic
  weights: SEANet encoder/decoder + RVQ (A2/A3/A11), conformer stack (A7), VQ assign (A8).
  The PyPI packages `encodec` and `vector_quantize_pytorch` are absent and no pretrained weights are
  available offline, so REAL-WEIGHT PARITY IS UNPINNED.
"""
