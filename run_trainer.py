"""CLI with the reference's flags (run_trainer.py:21-88): python run_trainer.py --config <name-or-path> [--seed N] [--device N]
`--config X` resolves ./config/**/X.yaml like the reference; multi-GPU runs are launched with torchrun."""
import argparse
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--device", type=int, default=None)
    a = ap.parse_args()
    path = a.config
    if not os.path.exists(path):
        hits = glob.glob(f"./config/**/{a.config}.yaml", recursive=True)
        if not hits:
            raise FileNotFoundError(a.config)
        path = hits[0]
    from libcontinual_amd.config import Config
    from libcontinual_amd.trainer import Trainer
    cfg = Config(path).get_config_dict()
    if a.seed is not None:
        cfg["seed"] = a.seed
    if a.device is not None:
        cfg["device_ids"] = a.device
    rank = int(os.environ.get("RANK", "0"))
    Trainer(rank, cfg).train_loop()


if __name__ == "__main__":
    main()
