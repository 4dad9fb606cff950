#!/usr/bin/env python3
"""Drop-in entry point with the reference's name and flags (``experiments/diffusion_test.sh`` calls
``python train_diffusion_prior.py --is_test 1 ...``): see avi-talking_amd/host/cli.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from avi_talking_amd.host.cli import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
