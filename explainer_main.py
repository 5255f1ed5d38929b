#!/usr/bin/env python
"""Drop-in for the reference's top-level `explainer_main.py`: same flags, mask optimisation on the MI355X engine.

    python explainer_main.py --dataset=syn1 --explain-node=300 --epochs=300 --ckptdir=ckpt --logdir=log
(see gnn-model-explainer_amd/explainer_main.py; reference: explainer_main.py:23-313)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if __name__ == "__main__":
    import gnn_model_explainer_amd
    gnn_model_explainer_amd.tune_process()      # before torch is imported: this process exists to run the engine
from gnn_model_explainer_amd.explainer_main import main  # noqa: E402

if __name__ == "__main__":
    main()
