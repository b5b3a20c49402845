#!/usr/bin/env python3
"""Single-class generalisation flickering attack on I3D (one perturbation for every clip of a class, e.g. the 'triple jump'
TFRecords; BASELINE config 4) -- the MI355X counterpart of the reference's i3d_adversarial_main_single_class_gen.py (section
CLASS_GEN_ATTACK of run_config.yml): bs 8 per GPU, fooling rate + ``res.pkl`` + a TensorFlow ``model_step_%05d`` checkpoint after
every pass over the records, resume from the newest checkpoint.

    python scripts/i3d_adversarial_main_single_class_gen.py [run_config.yml] [--max-steps N]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/i3d_adversarial_main_single_class_gen.py ...

The loop lives in flickering_adversarial_video_amd/i3d_dataset_attack.py (shared with i3d_adversarial_main_universal.py).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flickering_adversarial_video_amd.i3d_dataset_attack import main  # noqa: E402

if __name__ == "__main__":
    main("CLASS_GEN_ATTACK")
