#!/usr/bin/env python3
"""Universal flickering attack on I3D over shuffled uint8 Kinetics TFRecords, data-parallel over the GPUs of one node -- the
MI355X counterpart of the reference's i3d_adversarial_main_universal.py (section UNIVERSAL_ATTACK of run_config.yml; with
FLICKERING_ATTACK: False the dense "L12" baseline, with CYCLIC_PERTURBATION_ATTACK the time-invariant variant).

    python scripts/i3d_adversarial_main_universal.py [run_config.yml]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/i3d_adversarial_main_universal.py ...

The loop lives in flickering_adversarial_video_amd/i3d_dataset_attack.py (shared with i3d_adversarial_main_single_class_gen.py;
``--section CLASS_GEN_ATTACK`` still selects that mode from here).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flickering_adversarial_video_amd.i3d_dataset_attack import main  # noqa: E402

if __name__ == "__main__":
    main("UNIVERSAL_ATTACK")
