"""Fixture generator (build container only).  Copies the reference's shipped DATA (not source) that
the GPU box cannot reach: the yelp test JSONs + vocabulary, and the SiameseNet / Predictor weights
(read with torch.load(weights_only=True), re-saved as safetensors)."""
import os
import shutil

import torch
from safetensors.torch import save_file

REF = "/root/reference"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

for task in ("yelp", "amazon"):
    os.makedirs(os.path.join(GOLD, task), exist_ok=True)
    for f in ("neg10_pos10_test0.json", "neg10_pos10_test1.json", "word_to_id.txt"):
        shutil.copyfile(os.path.join(REF, "data/%s/processed_files" % task, f), os.path.join(GOLD, task, f))
        os.chmod(os.path.join(GOLD, task, f), 0o644)
for name in ("siamese_model_params", "predictor_net_params"):
    sd = torch.load(os.path.join(REF, "method/yelp/save/origin", name + ".pkl"), map_location="cpu", weights_only=True)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(GOLD, name + ".safetensors"))
    print(name, {k: tuple(v.shape) for k, v in sd.items()})
