"""Regenerates tests/golden/ref_config_archs.json: the `architecture` and `sampler` sections and the dropout / dropedge / lr /
batch-size values of every training configuration the reference ships (config_train/<dataset>/<family>/<name>.yml) as plain DATA -- the
list the `-m gpu` test test_ref_configs_gpu.py builds and trains one model per entry from.  Reads /root/reference (not
present on the GPU box: only the JSON travels).

    python oracle/gen_ref_config_archs.py            # rewrite the fixture
    python oracle/gen_ref_config_archs.py --check    # exit 1 unless the committed fixture equals what this generates
"""
import glob
import json
import os
import sys

import yaml

REF = os.environ.get("SHADOW_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "ref_config_archs.json")


def _scalar(v):
    # PyYAML reads `2e-5` (no dot) as a string; the fixture keeps the file's own spelling
    return v


def collect():
    out = []
    base = os.path.join(REF, "config_train")
    for path in sorted(glob.glob(os.path.join(base, "*", "*", "*.yml"))):
        cfg = yaml.safe_load(open(path))
        arch = cfg["architecture"]
        hp = cfg["hyperparameter"]
        if isinstance(arch, list):
            arch = arch[0]
        if isinstance(hp, list):
            hp = hp[0]
        # the `sampler` section as it stands in the file (a list of {method, phase, per-ensemble parameter lists}) and the
        # batch size it is cut into: the sampler parameters test_ref_configs_gpu.py checks against the oracle
        out.append(dict(name=os.path.relpath(path, base), architecture=dict(arch), dropout=_scalar(hp["dropout"]),
                        dropedge=_scalar(hp["dropedge"]), lr=_scalar(hp["lr"]), batch_size=hp.get("batch_size"),
                        sampler=[dict(s) for s in cfg.get("sampler", [])]))
    return out


def main():
    got = collect()
    if "--check" in sys.argv:
        have = json.load(open(OUT))
        if have != got:
            print(f"{OUT} differs from the reference's config_train ({len(have)} vs {len(got)} entries)")
            sys.exit(1)
        print(f"ok: {len(got)} configurations")
        return
    with open(OUT, "w") as f:
        json.dump(got, f, indent=1)
        f.write("\n")
    print(f"wrote {len(got)} configurations to {OUT}")


if __name__ == "__main__":
    main()
