#!/usr/bin/env python3
"""Entry point with the reference's name and flags (animal2vec_train.py:27-90, README.md:203):

    python animal2vec_train.py --config-dir configs/MeerKAT --config-name a2v_large_pretrain_best

reads the hydra-style YAML and runs the pretraining hot loop natively on MI355X -- on synthetic waveforms of the
configured shape, or with --data <manifest.tsv> on real clips; --finetune runs the finetune loop on a labelled
manifest from a --pretrained checkpoint (the reference's checkpoint policy / validation / loggers are out of scope)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def cli_main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-dir", required=True)
    ap.add_argument("--config-name", required=True)
    ap.add_argument("--updates", type=int, default=2)
    ap.add_argument("--data", default=None, help="manifest TSV (scripts/animal2vec_manifest.py layout)")
    ap.add_argument("--restore-file", default=None, help="fairseq-style checkpoint to resume pretraining from")
    ap.add_argument("--save", default=None, help="write a fairseq-style checkpoint here when done")
    ap.add_argument("--finetune", action="store_true", help="finetune on the labelled manifest (needs --data, --labels)")
    ap.add_argument("--pretrained", default=None, help="pretraining checkpoint for --finetune (model.w2v_path)")
    ap.add_argument("--labels", default=None, help="unique labels, e.g. \"['beep','cc','focal']\" (task.unique_labels)")
    a = ap.parse_args()
    import nn
    import yaml
    from animal2vec_amd.train_routine import config_from_yaml
    path = os.path.join(a.config_dir, a.config_name if a.config_name.endswith(".yaml") else a.config_name + ".yaml")
    is_ft_recipe = (yaml.safe_load(open(path)).get("model", {}) or {}).get("_name") == "wav2vec_ccas_finetune"
    if a.finetune or is_ft_recipe:
        import ast
        from animal2vec_amd.config import large_config
        from animal2vec_amd.train_routine import animal2vec_finetune_main, finetune_config_from_yaml
        assert a.data, "finetuning needs --data <labelled manifest.tsv>"
        if is_ft_recipe:
            # the recipe's model / optimization / optimizer / lr_scheduler blocks -> FinetuneConfig; the encoder's own
            # dimensions come with the pretrained checkpoint in the reference (w2v_args) -- here the large model
            fcfg, run = finetune_config_from_yaml(path)
            cfg = large_config()
            if run["average_top_k_layers"]:
                cfg.average_top_k_layers = int(run["average_top_k_layers"])
            labels = list(ast.literal_eval(a.labels)) if a.labels else run["unique_labels"]
            fcfg.n_labels = len(labels)
        else:
            cfg, _, run = config_from_yaml(path)
            assert a.labels, "--finetune on a pretraining recipe needs --labels"
            labels, fcfg = list(ast.literal_eval(a.labels)), None
        animal2vec_finetune_main(cfg, fcfg, data=a.data, unique_labels=labels, pretrained=a.pretrained, updates=a.updates,
                                 update_freq=run["update_freq"], max_tokens=run["max_tokens"], save_path=a.save)
        return
    cfg, optim, run = config_from_yaml(path)
    clips = max(1, run["max_tokens"] // 80000)
    nn.animal2vec_audio_main(cfg, optim, updates=a.updates, clips_per_micro_batch=clips, update_freq=run["update_freq"],
                             data=a.data, max_tokens=run["max_tokens"], restore_file=a.restore_file, save_path=a.save)


if __name__ == "__main__":
    cli_main()
