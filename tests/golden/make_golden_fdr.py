"""Golden vectors of the FDR stage, produced by RUNNING THE REFERENCE (alphadia/fdr/fdr.py,
alphadia/fdr/classifiers.py) in the build container:

    python tests/golden/make_golden_fdr.py

TEST INFRASTRUCTURE, same rules as make_golden.py: the reference is imported from /root/reference
through ``ref_shim`` (third-party stubs only), fed seeded synthetic inputs, and inputs + outputs are
stored in ``tests/golden/fdr.npz``.  The classifier runs with dropout = 0 so that its result does not
depend on torch's dropout stream (everything else - initial weights, train/test split, batch order,
Adam, BatchNorm - is deterministic for a given ``random_state``).
"""

from __future__ import annotations

import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
# fixtures go to tests/golden/ unless --out DIR is given (tests/test_golden_regenerate.py writes to a temp dir)
OUT_DIR = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else HERE
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402

ref_shim.install()

import torch  # noqa: E402
from alphadia.fdr import fdr as ref_fdr  # noqa: E402
from alphadia.fdr.classifiers import BinaryClassifierLegacyNewBatching, FeedForwardNN  # noqa: E402


def feature_table(rng, n_groups: int, d: int, separation: float):
    """Target/decoy pairs: decoys and 60 % of the targets from the null, the rest shifted."""
    n = 2 * n_groups
    decoy = np.tile([0, 1], n_groups).astype(np.int64)
    true_hit = (rng.random(n) < 0.4) & (decoy == 0)
    scale = rng.uniform(0.5, 20.0, size=d)
    offset = rng.uniform(-50.0, 500.0, size=d)
    shift = rng.uniform(0.2, 1.0, size=d) * separation
    z = rng.normal(size=(n, d)) + true_hit[:, None] * shift[None, :]
    x = (z * scale + offset).astype(np.float32)
    df = pd.DataFrame({f"f{i}": x[:, i] for i in range(d)})
    df["precursor_idx"] = np.arange(n, dtype=np.uint32)
    df["elution_group_idx"] = np.repeat(np.arange(n_groups, dtype=np.uint32), 2)
    df["channel"] = np.zeros(n, dtype=np.uint32)
    df["decoy"] = decoy.astype(np.uint8)
    df["rank"] = np.zeros(n, dtype=np.uint8)
    return df, [f"f{i}" for i in range(d)]


class FixedClassifier:
    """Stands in for the network where only the plumbing after it is under test."""

    def __init__(self, proba):
        self.proba = proba

    def fit(self, x, y):
        pass

    def predict_proba(self, x):
        assert len(x) == len(self.proba)
        return np.stack([1 - self.proba, self.proba], axis=1)


def main():
    d = {}
    rng = np.random.default_rng(20260928 + 77)

    # ---- get_q_values: ties in the score, in (score, decoy) and full ties
    n = 5000
    proba = np.round(rng.random(n), 3).astype(np.float32)  # many ties
    decoy = (rng.random(n) < 0.45 + 0.4 * proba).astype(np.float64)
    pidx = rng.integers(0, 1500, size=n).astype(np.uint32)
    df = pd.DataFrame({"proba": proba, "_decoy": decoy, "precursor_idx": pidx, "row": np.arange(n)})
    out = ref_fdr.get_q_values(df, "proba", "_decoy")
    d["q_proba"], d["q_decoy"], d["q_precursor_idx"] = proba, decoy, pidx
    d["q_out_qval"] = out["qval"].to_numpy()
    d["q_out_proba"] = out["proba"].to_numpy()
    d["q_out_decoy"] = out["_decoy"].to_numpy()
    d["q_out_precursor_idx"] = out["precursor_idx"].to_numpy()
    # decoys first (fdr = inf at the head)
    df2 = pd.DataFrame({"proba": np.array([0.1, 0.2, 0.3, 0.4, 0.5], np.float32),
                        "_decoy": np.array([1, 1, 0, 0, 1], np.float64), "precursor_idx": np.arange(5)})
    d["q2_out_qval"] = ref_fdr.get_q_values(df2, "proba", "_decoy")["qval"].to_numpy()

    # ---- keep_best: one and two group columns, ties inside a group
    n = 4000
    kb = pd.DataFrame({
        "proba": np.round(rng.random(n), 2).astype(np.float32),
        "elution_group_idx": rng.integers(0, 900, size=n).astype(np.uint32),
        "channel": rng.choice([0, 4, 8], size=n).astype(np.uint32),
        "precursor_idx": rng.integers(0, 1200, size=n).astype(np.uint32),
        "row": np.arange(n),
    })
    for k in ("proba", "elution_group_idx", "channel", "precursor_idx"):
        d["kb_" + k] = kb[k].to_numpy()
    d["kb_rows_eg_channel"] = ref_fdr.keep_best(kb, group_columns=["elution_group_idx", "channel"])["row"].to_numpy()
    d["kb_rows_eg"] = ref_fdr.keep_best(kb, group_columns=["elution_group_idx"])["row"].to_numpy()
    d["kb_rows_precursor"] = ref_fdr.keep_best(kb, group_columns=["precursor_idx"])["row"].to_numpy()

    # ---- perform_fdr with the probabilities fixed (competitive and not)
    tab, cols = feature_table(rng, 1500, 6, 2.0)
    p_fixed = np.clip(0.5 + 0.25 * rng.normal(size=len(tab)) - 0.35 * (tab["f0"].to_numpy() > tab["f0"].median()), 0, 1)
    p_fixed = np.round(p_fixed, 3).astype(np.float32)
    order = np.concatenate([np.flatnonzero(tab["decoy"] == 0), np.flatnonzero(tab["decoy"] == 1)])
    for competitive in (False, True):
        res = ref_fdr.perform_fdr(FixedClassifier(p_fixed[order]), cols, tab[tab["decoy"] == 0].copy(),
                                  tab[tab["decoy"] == 1].copy(), competitive=competitive, group_channels=True,
                                  random_state=5)
        tag = "pf_comp" if competitive else "pf_plain"
        d[tag + "_precursor_idx"] = res["precursor_idx"].to_numpy()
        d[tag + "_qval"] = res["qval"].to_numpy()
        d[tag + "_proba"] = res["proba"].to_numpy()
    d["pf_features"] = tab[cols].to_numpy()
    d["pf_decoy"] = tab["decoy"].to_numpy()
    d["pf_elution_group_idx"] = tab["elution_group_idx"].to_numpy()
    d["pf_fixed_proba"] = p_fixed

    # ---- the classifier: initial weights, training, probabilities
    tab, cols = feature_table(rng, 4000, 12, 1.5)
    x = tab[cols].to_numpy()
    y = tab["decoy"].to_numpy().astype(np.float64)
    hp = dict(test_size=0.2, batch_size=128, epochs=3, learning_rate=0.001, weight_decay=0.00001,
              layers=[100, 50, 20, 5], dropout=0.0, metric_interval=50, random_state=7)
    clf = BinaryClassifierLegacyNewBatching(**hp)
    # build the network exactly where fit() would (first torch draw after the constructor's manual_seed)
    clf.input_dim = x.shape[1]
    clf.network = FeedForwardNN(input_dim=x.shape[1], output_dim=2, layers=hp["layers"], dropout=0.0)
    for k, v in clf.network.state_dict().items():
        d["clf_init/" + k] = v.detach().numpy().copy()
    clf.fit(x, y)
    for k, v in clf.network.state_dict().items():
        d["clf_final/" + k] = v.detach().numpy().copy()
    d["clf_x"], d["clf_y"] = x, y
    d["clf_proba"] = clf.predict_proba(x)
    for k, v in clf.metrics.items():
        d["clf_metrics_" + k] = np.asarray(v, dtype=np.float64)
    for k, v in hp.items():
        d["clf_hp_" + k] = np.asarray(v)
    # the same schedule with dropout at the reference default, for a statistical comparison
    hp2 = dict(hp, dropout=0.001, random_state=11)
    clf2 = BinaryClassifierLegacyNewBatching(**hp2)
    clf2.fit(x, y)
    d["clf2_proba"] = clf2.predict_proba(x)

    # ---- perform_fdr end to end with the reference network (statistical comparison)
    clf3 = BinaryClassifierLegacyNewBatching(test_size=0.001, batch_size=128, epochs=5, learning_rate=0.001,
                                             dropout=0.0, random_state=3)
    res = ref_fdr.perform_fdr(clf3, cols, tab[tab["decoy"] == 0].copy(), tab[tab["decoy"] == 1].copy(),
                              competitive=True, group_channels=True, random_state=9)
    d["e2e_features"] = x
    d["e2e_decoy"] = tab["decoy"].to_numpy()
    d["e2e_elution_group_idx"] = tab["elution_group_idx"].to_numpy()
    d["e2e_precursor_idx"] = res["precursor_idx"].to_numpy()
    d["e2e_qval"] = res["qval"].to_numpy()
    d["e2e_proba"] = res["proba"].to_numpy()
    d["e2e_res_decoy"] = res["decoy"].to_numpy()

    d["caveat"] = np.asarray("reference fdr.py / classifiers.py executed with torch " + torch.__version__ +
                             ", numpy " + np.__version__ + ", pandas " + pd.__version__ + " on CPU")
    path = os.path.join(OUT_DIR, "fdr.npz")
    np.savez_compressed(path, **d)
    print(f"{path}: {os.path.getsize(path) / 1e6:.2f} MB; q<=0.01 targets end to end: "
          f"{int(((res['qval'] <= 0.01) & (res['decoy'] == 0)).sum())}")


if __name__ == "__main__":
    main()
