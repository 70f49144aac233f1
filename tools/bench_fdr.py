"""FDR stage at the size of the headline run: 3e6 candidate rows x 46 features (1e6 precursors x 3
candidates), the reference's production hyper-parameters
(peptidecentric.py:55-62 with enable_nn_hyperparameter_tuning: batch 4096, lr 1e-3, 10 epochs,
dropout 0.001).  Prints one JSON object: device times of training / inference / q-values / best
row per group, and the plain-PyTorch oracle (2 threads, as the reference trains) on a bounded sample.

    python tools/bench_fdr.py [--rows 3000000] [--features 46] [--epochs 10]
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=3_000_000)
    ap.add_argument("--features", type=int, default=46)
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--cpu-steps", type=int, default=150)
    args = ap.parse_args()

    from alphadia_amd import fdr, runtime

    n, d = args.rows, args.features
    rng = np.random.default_rng(20260928)
    decoy = (np.arange(n) % 2).astype(np.float64)
    hit = (rng.random(n) < 0.35) & (decoy == 0)
    x = rng.standard_normal((n, d), dtype=np.float32)
    x += hit[:, None] * rng.uniform(0.1, 0.8, d).astype(np.float32)
    x *= rng.uniform(0.5, 20, d).astype(np.float32)
    x += rng.uniform(-10, 100, d).astype(np.float32)

    clf = fdr.HipBinaryClassifier(test_size=0.001, batch_size=5000, learning_rate=0.001, epochs=args.epochs,
                                  experimental_hyperparameter_tuning=True, random_state=1)
    t0 = time.perf_counter()
    clf.fit(x, decoy)
    fit_wall = time.perf_counter() - t0
    n_steps = len(clf.metrics["batch_count"]) and (clf.epochs * (int(np.floor(0.999 * n)) // clf.batch_size - 1))
    t0 = time.perf_counter()
    proba = clf.predict_proba(x)[:, 1]
    predict_wall = time.perf_counter() - t0

    ctx = runtime.get_context(0)
    pidx = np.arange(n, dtype=np.int64)
    eg = pidx // 2
    t0 = time.perf_counter()
    order, qval = ctx.fdr_q_values(proba, decoy, pidx)
    q_wall = time.perf_counter() - t0
    t0 = time.perf_counter()
    keep = ctx.fdr_keep_best(proba, eg, np.zeros(n, np.int64))
    kb_wall = time.perf_counter() - t0
    ids = int(((qval <= 0.01) & (decoy[order] == 0)).sum())

    cpu = None
    if args.cpu_steps > 0:
        cpu = cpu_leg(args, fdr, clf, x, decoy, proba, pidx, n, d)

    print(json.dumps({
        "workload": f"{n} rows x {d} features, {clf.epochs} epochs, batch {clf.batch_size}, lr {clf.learning_rate:g}, "
                    f"dropout {clf.dropout}, layers {clf.layers}",
        "train_steps": int(n_steps),
        "fit_kernels_ms": clf.last_fit_ms,
        "fit_us_per_step": 1e3 * clf.last_fit_ms / max(n_steps, 1),
        "fit_rows_per_s": n_steps * clf.batch_size / (clf.last_fit_ms / 1e3) if clf.last_fit_ms else None,
        "fit_wall_s": fit_wall,
        "predict_kernel_ms": clf.last_predict_ms,
        "predict_rows_per_s": n / (clf.last_predict_ms / 1e3) if clf.last_predict_ms else None,
        "predict_wall_s": predict_wall,
        "q_values_wall_ms": 1e3 * q_wall,
        "keep_best_wall_ms": 1e3 * kb_wall,
        "targets_at_1pct": ids,
        "train_loss_first_last": [clf.metrics["train_loss"][0], clf.metrics["train_loss"][-1]],
        "cpu_oracle": cpu,
    }))


def cpu_leg(args, fdr, clf, x, decoy, proba, pidx, n, d):
    # CPU: the torch fp32 oracle, 2 threads (fdr/utils.py:55-94), a bounded number of steps
    from oracle import fdr_oracle

    clf_cpu = fdr.HipBinaryClassifier(test_size=0.001, batch_size=clf.batch_size, epochs=1, random_state=1)
    clf_cpu.input_dim = d
    clf_cpu._init_state()
    train_rows, _, schedule, _ = clf_cpu._plan(n)
    steps = min(args.cpu_steps, len(schedule))
    dims = [d, *clf_cpu.layers, 2]
    t0 = time.perf_counter()
    fdr_oracle.mlp_fit(dims, *clf_cpu._state[:3], x, decoy, train_rows, schedule[:steps], clf.batch_size,
                       clf.learning_rate, clf.weight_decay, dropout=0.001)
    cpu_fit = time.perf_counter() - t0
    # the oracle slices x[train_rows] once up front; charge only the steps
    t0 = time.perf_counter()
    _ = np.asarray(x, np.float32)[train_rows]
    cpu_fit -= time.perf_counter() - t0
    t0 = time.perf_counter()
    fdr_oracle.q_values(proba[:1_000_000], decoy[:1_000_000], pidx[:1_000_000])
    cpu_q = time.perf_counter() - t0
    return {
        "kind": "port (plain PyTorch fp32, 2 threads as the reference)",
        "fit_steps": steps,
        "fit_us_per_step": 1e6 * cpu_fit / max(steps, 1),
        "q_values_1e6_rows_ms": 1e3 * cpu_q,
    }


if __name__ == "__main__":
    main()
