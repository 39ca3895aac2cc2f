"""TEST INFRASTRUCTURE ONLY — CPU oracle for the segmentation hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.

* ``oracle.seg_oracle``  – plain-PyTorch-CPU fp32/fp64 restatement of the reference's
  VNet/UNet forward, the seven reachable losses, the Dice/IoU metrics and one
  AdamW/Adam train step (each function cites the reference file:line it restates).
* ``oracle.prepost_oracle`` – numpy restatement of the pre/post-processing around ``predict`` (resample, the two
  normalisations, the patch loop); ``normalize`` is pinned by golden vectors from the reference's own function, the
  SimpleITK resampler is restated from ITK's published algorithm (**parity unpinned**: SimpleITK is not installed).
* ``oracle.ref_loader``  – loads the *real* reference modules from ``/root/reference``
  under alias names (only available in the build container, never on the GPU box).
* ``oracle.make_golden`` – regenerates ``tests/golden/*.npz`` from the real reference.

Parity status: the reference ships no tests or golden vectors, so the
restatement is pinned against outputs of the reference itself run in the build
container (``tests/golden``, produced by ``oracle/make_golden.py``) and, when
``/root/reference`` is present, against the live reference modules.
"""
