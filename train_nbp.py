#!/usr/bin/env python
"""Entry point, same name and flag as the reference's train_nbp.py (:13-33):
    python train_nbp.py [-c <config in configs/nbp/>]"""
import argparse
import os
import sys

dir_path = os.path.abspath(os.path.dirname(__file__))
sys.path.insert(0, dir_path)
from nextbestpath_amd.testers.nbp_planning import load_params  # noqa: E402
from nextbestpath_amd.trainers.train_nbp_model import run_training_nbp  # noqa: E402

if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Train the NBP model.")
    parser.add_argument("-c", "--config", type=str, help="name of the config file in configs/nbp/")
    args = parser.parse_args()
    params = load_params(os.path.join(dir_path, "configs/nbp", args.config or "nbp_default_training_config.json"))
    if not os.path.isabs(params.output_dir):
        params.output_dir = os.path.join(dir_path, params.output_dir)
    run_training_nbp(params)
