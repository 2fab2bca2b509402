#!/usr/bin/env python
"""Entry point, same name and flag as the reference's test_nbp_planning.py (:12-38):
    python test_nbp_planning.py [-c <config in configs/test/>]
    python -m torch.distributed.run --nproc-per-node 8 test_nbp_planning.py     # scene-parallel"""
import argparse
import os
import sys

import torch

dir_path = os.path.abspath(os.path.dirname(__file__))
sys.path.insert(0, dir_path)
from nextbestpath_amd.testers.nbp_planning import load_params, test_nbp_planning  # noqa: E402

test_configs_dir = os.path.join(dir_path, "configs/test/")

if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Test an NBP model: exploration rollouts on the listed scenes.")
    parser.add_argument("-c", "--config", type=str, help="name of the config file in configs/test/")
    parser.add_argument("--n-poses", type=int, default=101)
    parser.add_argument("--rollouts-per-gpu", type=int, default=None,
                        help="concurrent rollouts per GPU (batched NBP forward); default: the config's value or 24")
    args = parser.parse_args()
    p = load_params(os.path.join(test_configs_dir, args.config or "test_via_nbp_model.json"))
    ds = p.dataset_path if os.path.isabs(p.dataset_path) else os.path.join(dir_path, p.dataset_path)
    wt = p.nbp_weights if os.path.isabs(p.nbp_weights) else os.path.join(dir_path, p.nbp_weights)
    with torch.no_grad():
        test_nbp_planning(params_file=p.params_name, model_file=p.model_name, results_json_file=p.results_json_name,
                          numGPU=p.numGPU, test_scenes=p.test_scenes, test_resolution=p.test_resolution,
                          use_perfect_depth_map=p.use_perfect_depth_map, compute_collision=p.compute_collision,
                          load_json=p.load_json, dataset_path=ds, nbp_weights=wt,
                          configs_dir=os.path.join(dir_path, "configs/macarons"),
                          results_dir=os.path.join(dir_path, "data"), n_poses=args.n_poses,
                          seed=getattr(p, "random_seed", 8), torch_seed=getattr(p, "torch_seed", 9),
                          rollouts_per_gpu=args.rollouts_per_gpu or getattr(p, "rollouts_per_gpu", 48), grid_size=getattr(p, "grid_size", 256),
                          nbp_precision=getattr(p, "nbp_precision", None))     # None: the model default (fp32_split), what bench.py measures
