"""CPU oracle for the flickering-attack hot path.  TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement (numpy / torch-CPU fp32) of the
reference's algorithm (roiponytch/Flickering_Adversarial_Video).  It exists to CHECK the
HIP path; it is never the thing measured or shipped.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.  The
product package ``flickering_adversarial_video_amd`` never imports from here and fails
loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * torch-dialect attack math (Perturbation / Losses / Adversarial_metrics /
    Adam loop ordering): PINNED by golden vectors generated from the reference's own
    importable classes (tests/golden/make_golden.py -> tests/golden/*.npz).
  * TF-dialect attack math: the reference's TF graph cannot be imported here
    (tensorflow / sonnet absent).  Pinned where a formula is shared with the torch classes
    (regularisers, thickness/roughness, improve-loss prob mode, CE untargeted) by the same
    golden vectors, plus hand-derived known-answer tests.  Remaining TF-only branches:
    parity unpinned.
  * I3D network (i3d.py) and VideoResNet (torchvision 0.5.0, third party, not in the
    reference tree): no reference tests or golden vectors exist -> parity unpinned;
    torch-CPU ``conv3d`` autograd is the independent implementation.
"""
