"""CPU oracle for the JoliNeedle glimpse-rollout hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package, and only as the
checker.  The product (``jolineedle_amd``) never imports it.

Pure PyTorch fp32 on CPU.  Each module cites the reference file:line it restates
(paths relative to /root/reference).

Pinning status
--------------
* GPT / embeddings / recurrence / rollout / env / REINFORCE loss: pinned against
  the reference's own code, executed in the build container with name-only stub
  modules (``tests/golden/make_golden.py``) -> ``tests/golden/*.npz``.
* YOLOX (CSPDarknet/PAFPN/head/decode/postprocess), ``positional_encodings``,
  kornia ``Boxes.to_mask`` and torchvision ``nms`` are third-party packages that
  are absent from /root/reference (README.md:24-30, requirements.txt:9-11):
  **parity unpinned** for those; they are restated from the published
  algorithms and anchored on the reference's call sites, on published parameter
  counts / output shapes and on hand-computed known answers.
"""
