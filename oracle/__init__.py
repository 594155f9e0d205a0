"""CPU oracle for the audio-ssl upstream hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (numpy + torch-CPU
fp32/fp64) of the reference's `train_upstream.py` step.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it; nothing under `audio-ssl_amd/` (the product) does.

Parity status
-------------
* Everything that the reference computes with its own Python (augmentations,
  encoder, loss heads, optimisers, k-means maths) is pinned by the fixtures in
  `tests/golden/`, which were produced by running the reference's code itself
  in the build container (`tests/golden/make_goldens.py`).
* The log-mel front end follows the *published* algorithm of librosa 0.8.1
  (`requirements.txt:71`), which is absent from `/root/reference` and from the
  image: **parity unpinned vs librosa**; cross-checked against `torch.stft`.
"""
