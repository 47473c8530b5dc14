#!/usr/bin/env python3
"""Entry point with the reference's name and flags (animal2vec_train.py:27-90, README.md:203):

    python animal2vec_train.py --config-dir configs/MeerKAT --config-name a2v_large_pretrain_best

reads the hydra-style YAML and runs the pretraining hot loop natively on MI355X (synthetic waveforms of
the configured shape; the reference's data pipeline / checkpoint policy / loggers are out of scope)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def cli_main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-dir", required=True)
    ap.add_argument("--config-name", required=True)
    ap.add_argument("--updates", type=int, default=2)
    a = ap.parse_args()
    import nn
    from animal2vec_amd.train_routine import config_from_yaml
    path = os.path.join(a.config_dir, a.config_name if a.config_name.endswith(".yaml") else a.config_name + ".yaml")
    cfg, optim, run = config_from_yaml(path)
    clips = max(1, run["max_tokens"] // 80000)
    nn.animal2vec_audio_main(cfg, optim, updates=a.updates, clips_per_micro_batch=clips, update_freq=run["update_freq"])


if __name__ == "__main__":
    cli_main()
