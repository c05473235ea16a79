"""Readers for tests/golden/layers_fwd_bwd.npz and models_step.npz (data produced
by the reference's shaDow/layers.py / models.py through oracle/gen_golden_layers.py)."""
import json
import os

import numpy as np

from tests._golden import GOLDEN


class LayerGolden:
    def __init__(self, fname="layers_fwd_bwd.npz"):
        self.z = np.load(os.path.join(GOLDEN, fname))
        self.cases = json.loads(bytes(self.z["cases"]).decode())

    def get(self, ci, name):
        return self.z[f"c{ci}_{name}"]

    def params(self, ci, prefix="p"):
        pre = f"c{ci}_{prefix}_"
        return {k[len(pre):]: self.z[k] for k in self.z.files if k.startswith(pre)}

    def grads(self, ci):
        return self.params(ci, "g")


class ModelGolden:
    def __init__(self, fname="models_step.npz"):
        self.z = np.load(os.path.join(GOLDEN, fname))
        self.cases = json.loads(bytes(self.z["cases"]).decode())

    def get(self, ci, name):
        key = f"m{ci}_{name}"
        return self.z[key] if key in self.z.files else None

    def group(self, ci, prefix):
        pre = f"m{ci}_{prefix}_"
        return {k[len(pre):]: self.z[k] for k in self.z.files if k.startswith(pre)}
