"""Generator of tests/golden/c1_sun_full.json -- the data side of BASELINE.json configs[0] ("ViT-B/16 SUP-imnet fine-tune on
the sun_full manifest, bs=8, fp32, CPU reference path: plumbing").

Runs the REFERENCE's own loaders in the build container (they import as-is):
    ssl4polyp.configs.manifests.load_split        data_packs/sun_full/{train,val,test}.csv
    ssl4polyp.configs.layered.load_layered_config config/exp/exp1.yaml (-> base.yaml defaults)
and derives what train_classification.py derives from them before its first step: per-class counts, the class weights
N / (n_class * count) (tc.py:5613-5630), pos_weight = neg / pos for the two-class packs (tc.py:6090-6102), the optimizer /
schedule settings (config/base.yaml), and -- with torch, as the reference does (nn.BCEWithLogitsLoss(pos_weight)) -- the loss
of a fixed logit vector on 8 training labels (every 577th row).  Only data is stored: counts, labels, numbers.

    PYTHONPATH=/root/reference/src python tests/golden/make_c1_fixture.py
"""
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

REF = Path(os.environ.get("SSL4POLYP_REFERENCE", "/root/reference"))
sys.path.insert(0, str(REF / "src"))


def build():
    from ssl4polyp.configs import layered as L
    from ssl4polyp.configs import manifests as M
    pack = REF / "data_packs" / "sun_full"
    out = {"pack": "sun_full", "splits": {}}
    labels = {}
    for split in ("train", "val", "test"):
        rows = M.load_split(pack / f"{split}.csv")
        lab = [int(r["label"]) for r in rows]
        labels[split] = lab
        counts = np.bincount(lab, minlength=2).tolist()
        out["splits"][split] = {"rows": len(rows), "class_counts": counts, "first_frame": rows[0]["frame_path"]}
    train = labels["train"]
    counts = np.bincount(train, minlength=2)
    n_class = len(set(train))
    out["n_class"] = n_class
    out["class_weights"] = [float(len(train) / (n_class * c)) for c in counts]   # tc.py:5621-5624
    out["pos_weight"] = float(counts[0]) / float(counts[1])                      # tc.py:6092-6096
    cfg = L.load_layered_config("exp/exp1.yaml")
    out["config"] = {k: cfg[k] for k in ("optimizer", "lr", "weight_decay", "batch_size", "epochs", "amp", "image_size")}
    out["config"]["scheduler"] = dict(cfg["scheduler"])
    out["config"]["seeds"] = list(cfg["seeds"])
    # C1: per-rank batch 8, fp32.  First 8 training labels + a fixed logit vector through the reference's loss construction.
    first8 = train[:: len(train) // 8][:8]  # (the CSV is ordered by case: a strided sample carries both classes)
    rng = np.random.Generator(np.random.PCG64(77))
    logits = rng.standard_normal((8, 2)).astype(np.float32)
    z = torch.from_numpy(logits)
    loss = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor(out["pos_weight"], dtype=torch.float32))(
        z[:, 1] - z[:, 0], torch.tensor(first8, dtype=torch.float32))
    out["c1_batch"] = {"labels": first8, "logits": logits.tolist(), "bce_loss": float(loss)}
    return out


if __name__ == "__main__":
    dst = Path(__file__).with_name("c1_sun_full.json")
    dst.write_text(json.dumps(build(), indent=1, sort_keys=True) + "\n")
    print("wrote", dst)
