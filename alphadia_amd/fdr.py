"""FDR stage on the device.

Host mirror of the reference's FDR interface, with the arithmetic on the GPU:

* ``HipBinaryClassifier``  <- ``BinaryClassifierLegacyNewBatching`` (alphadia/fdr/classifiers.py:145-495):
  same constructor, ``fit`` / ``predict`` / ``predict_proba`` / ``to_state_dict`` / ``from_state_dict``; the
  state dict is interchangeable with the reference's (``FeedForwardNN`` parameter names).  Training runs
  in ``adh_mlp_fit`` (alphadia_amd/csrc/adh_mlp.hip); torch is used only to draw the initial weights the
  way ``torch.nn.Linear`` does, so a seeded run starts from the reference's own initialisation.
* ``get_q_values`` / ``keep_best`` / ``perform_fdr``  <- alphadia/fdr/fdr.py:24-297 (sorts, scans and the
  per-group selection in ``adh_fdr_q_values`` / ``adh_fdr_keep_best``; fragment competition in
  ``adh_fragcomp``).
* ``HipFDRManager.fit_predict``  <- alphadia/workflow/managers/fdr_manager.py:105-232 (decoy strategies).

There is no CPU fallback: every call needs the built library and a GPU.
"""

from __future__ import annotations

import logging
import os
import warnings
from collections import defaultdict
from copy import deepcopy

import numpy as np
import pandas as pd

from alphadia_amd import runtime

logger = logging.getLogger(__name__)

MAX_DIA_CYCLE_SHAPE = 2  # fdr.py:19


_WARNED_NUMPY_INIT = False


class TooFewPSMError(ValueError):
    """alphadia.exceptions.TooFewPSMError: the train/test split is empty."""


# --------------------------------------------------------------------------------------------
# train / test split: sklearn.model_selection.train_test_split(X, y, indices, test_size, random_state)
# as used by alphadia/fdr/utils.py:16-52 (ShuffleSplit: one permutation, test rows first)
# --------------------------------------------------------------------------------------------
def train_test_indices(n_samples: int, test_size: float, random_state=None):
    n_test = int(np.ceil(test_size * n_samples))
    n_train = int(np.floor((1.0 - test_size) * n_samples))
    if n_samples == 0 or n_train == 0:
        raise TooFewPSMError(
            f"With n_samples={n_samples}, test_size={test_size} and train_size=None, the resulting train set "
            "will be empty. Adjust any of the aforementioned parameters."
        )
    rng = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(random_state)
    perm = rng.permutation(n_samples)
    return perm[n_test : n_test + n_train], perm[:n_test]


def scaled_training_params(n_samples: int, base_lr: float = 0.001, max_batch: int = 4096, min_batch: int = 128):
    """classifiers.py:102-142: batch size linear in the sample count, learning rate ~ sqrt(batch size)."""
    if n_samples >= 1_000_000:
        return max_batch, base_lr
    batch_size = int(np.clip((n_samples / 1_000_000) * max_batch, min_batch, max_batch))
    return batch_size, base_lr * np.sqrt(batch_size / max_batch)


def _bce(p: np.ndarray, t: np.ndarray) -> float:
    """torch.nn.BCELoss (mean, logs clamped at -100) in float32."""
    p = p.astype(np.float32)
    t = t.astype(np.float32)
    with np.errstate(divide="ignore"):
        lp = np.maximum(np.log(p), np.float32(-100.0))
        lq = np.maximum(np.log(np.float32(1.0) - p), np.float32(-100.0))
    return float(np.mean(-(t * lp + (np.float32(1.0) - t) * lq), dtype=np.float32))


class HipBinaryClassifier:
    """Feed-forward target/decoy classifier trained on the GPU."""

    def __init__(
        self,
        input_dim: int = 10,
        output_dim: int = 2,
        test_size: float = 0.2,
        batch_size: int = 1000,
        epochs: int = 10,
        learning_rate: float = 0.0002,
        weight_decay: float = 0.00001,
        layers: list[int] | None = None,
        dropout: float = 0.001,
        metric_interval: int = 1000,
        *,
        experimental_hyperparameter_tuning: bool = False,
        random_state: int | None = None,
        device: int | None = None,
        **kwargs,
    ):
        self.layers = [100, 50, 20, 5] if layers is None else list(layers)
        self.test_size = test_size
        self.batch_size = batch_size
        self.epochs = epochs
        self.learning_rate = learning_rate
        self.weight_decay = weight_decay
        self.dropout = dropout
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.metric_interval = metric_interval
        self.experimental_hyperparameter_tuning = experimental_hyperparameter_tuning
        self.device = device

        self._state = None  # (params, running_mean, running_var, num_batches_tracked)
        self._fitted = False
        self.metrics = {k: [] for k in ("epoch", "batch_count", "train_loss", "train_accuracy", "test_loss",
                                        "test_accuracy")}
        self.last_fit_ms = 0.0      # HIP-event time of the training kernels of the last fit
        self.last_predict_ms = 0.0  # the same for the last predict / predict_proba

        # classifiers.py:239-246: one numpy generator; torch (initial weights) is seeded from it
        self._np_rng = np.random.default_rng(seed=random_state)
        self._torch_seed = None
        self._dropout_seed = 0
        if random_state is not None:
            self._torch_seed = int(self._np_rng.integers(0, 1_000_000))
            self._dropout_seed = self._torch_seed
        if kwargs:
            warnings.warn(f"Unknown arguments: {kwargs}")

    # -- parameters ------------------------------------------------------------------------
    @property
    def fitted(self) -> bool:
        return self._fitted

    def _linear_shapes(self):
        dims = [self.input_dim, *self.layers, self.output_dim]
        return [(dims[i + 1], dims[i]) for i in range(len(dims) - 1)]

    def _init_state(self):
        """Initial parameters drawn like FeedForwardNN._build_model does (classifiers.py:514-528)."""
        d = self.input_dim
        parts = [np.ones(d, np.float32), np.zeros(d, np.float32)]
        torch = None
        if not os.environ.get("ADH_FDR_NUMPY_INIT"):
            try:
                import torch
            except ImportError:  # the draw below has torch's distribution, not its random stream
                torch = None
        if torch is not None:
            if self._torch_seed is not None:
                torch.manual_seed(self._torch_seed)
                self._torch_seed = None  # the reference seeds once, in the constructor
            for out_f, in_f in self._linear_shapes():
                lin = torch.nn.Linear(in_f, out_f)
                parts.append(lin.weight.detach().numpy().astype(np.float32).ravel())
                parts.append(lin.bias.detach().numpy().astype(np.float32).ravel())
        else:
            # nn.Linear.reset_parameters: weight and bias uniform in +-1 / sqrt(fan_in).  Used where torch is absent
            # or ADH_FDR_NUMPY_INIT is set (tools/bench_fdr.py in the bench line: an `import torch` costs minutes on
            # a cold box); same network and training, another random stream than the reference's.
            if not os.environ.get("ADH_FDR_NUMPY_INIT"):  # (asked for: the bench; torch missing: say so, once)
                global _WARNED_NUMPY_INIT
                if not _WARNED_NUMPY_INIT:
                    logger.warning("torch is not importable: the classifier's initial weights are drawn from a NumPy "
                                   "stream - same distribution and training, but q-values and PSM sets will differ from a "
                                   "torch-seeded run of the reference with the same random_state")
                    _WARNED_NUMPY_INIT = True
            self.init_stream = "numpy"  # (recorded on the classifier: to_state_dict keeps the reference's keys only)
            rng = np.random.default_rng(self._torch_seed)
            self._torch_seed = None
            for out_f, in_f in self._linear_shapes():
                bound = 1.0 / np.sqrt(in_f)
                parts.append(rng.uniform(-bound, bound, out_f * in_f).astype(np.float32))
                parts.append(rng.uniform(-bound, bound, out_f).astype(np.float32))
        self._state = (np.concatenate(parts), np.zeros(d, np.float32), np.ones(d, np.float32), 0)

    def _device_mlp(self) -> runtime.DeviceMlp:
        mlp = runtime.DeviceMlp(runtime.get_context(self.device), self.input_dim, self.layers, self.output_dim)
        mlp.set_state(*self._state)
        return mlp

    def _layer_indices(self):
        # nn.Sequential positions: 0 BatchNorm1d, then (Linear, ReLU, Dropout) x hidden, Linear, Softmax
        return [1 + 3 * i for i in range(len(self.layers))] + [1 + 3 * len(self.layers)]

    def to_state_dict(self) -> dict:
        sd = {
            "_fitted": self._fitted,
            "input_dim": self.input_dim,
            "output_dim": self.output_dim,
            "test_size": self.test_size,
            "batch_size": self.batch_size,
            "epochs": self.epochs,
            "learning_rate": self.learning_rate,
            "weight_decay": self.weight_decay,
            "layers": self.layers,
            "dropout": self.dropout,
            "metric_interval": self.metric_interval,
            "metrics": self.metrics,
        }
        if self._fitted:
            import torch

            params, rm, rv, nbt = self._state
            d = self.input_dim
            net = {
                "fc_layers.0.weight": torch.from_numpy(params[:d].copy()),
                "fc_layers.0.bias": torch.from_numpy(params[d : 2 * d].copy()),
                "fc_layers.0.running_mean": torch.from_numpy(rm.copy()),
                "fc_layers.0.running_var": torch.from_numpy(rv.copy()),
                "fc_layers.0.num_batches_tracked": torch.tensor(nbt, dtype=torch.long),
            }
            off = 2 * d
            for pos, (out_f, in_f) in zip(self._layer_indices(), self._linear_shapes()):
                net[f"fc_layers.{pos}.weight"] = torch.from_numpy(params[off : off + out_f * in_f].reshape(out_f, in_f).copy())
                off += out_f * in_f
                net[f"fc_layers.{pos}.bias"] = torch.from_numpy(params[off : off + out_f].copy())
                off += out_f
            sd["network_state_dict"] = net
        return sd

    def from_state_dict(self, state_dict: dict, *, load_hyperparameters: bool = False) -> None:
        sd = dict(state_dict)
        if "network_state_dict" in sd:
            net = sd.pop("network_state_dict")
            self.input_dim, self.output_dim = sd.pop("input_dim"), sd.pop("output_dim")
            self.layers, self.dropout = list(sd.pop("layers")), sd.pop("dropout")
            arr = lambda k: np.asarray(net[k].detach().cpu().numpy() if hasattr(net[k], "detach") else net[k])  # noqa: E731
            parts = [arr("fc_layers.0.weight"), arr("fc_layers.0.bias")]
            for pos in self._layer_indices():
                parts += [arr(f"fc_layers.{pos}.weight").ravel(), arr(f"fc_layers.{pos}.bias").ravel()]
            nbt = int(arr("fc_layers.0.num_batches_tracked")) if "fc_layers.0.num_batches_tracked" in net else 0
            self._state = (
                np.concatenate([p.astype(np.float32).ravel() for p in parts]),
                arr("fc_layers.0.running_mean").astype(np.float32),
                arr("fc_layers.0.running_var").astype(np.float32),
                nbt,
            )
            self._fitted = True
        if load_hyperparameters:
            self.__dict__.update(deepcopy(sd))

    # -- training --------------------------------------------------------------------------
    def _plan(self, n_samples: int):
        """Train/test rows and the batch of every optimiser step (classifiers.py:350-381); consumes
        the numpy generator exactly as the reference's fit does."""
        split_seed = int(self._np_rng.integers(0, 1_000_000))
        logger.info(f"Using random state {split_seed} for train-test-split")
        train_rows, test_rows = train_test_indices(n_samples, self.test_size, split_seed)
        num_batches = (len(train_rows) // self.batch_size) - 1
        starts = np.arange(max(num_batches, 0), dtype=np.int64) * self.batch_size
        schedule, epoch_of = [], []
        for epoch in range(self.epochs):
            if num_batches > 0:
                starts = starts[self._np_rng.permutation(num_batches)]  # :378-381, permuted cumulatively
                schedule.append(starts.copy())
                epoch_of.append(np.full(num_batches, epoch))
        schedule = np.concatenate(schedule) if schedule else np.zeros(0, np.int64)
        epoch_of = np.concatenate(epoch_of) if epoch_of else np.zeros(0, np.int64)
        return train_rows, test_rows, schedule, epoch_of

    def _prepare(self, n_features: int) -> None:
        if self._state is not None and self.input_dim != n_features:
            warnings.warn("Input dimension of network has changed. Network has been reinitialized.")
            self._state = None
        if self._state is None:
            self.input_dim = n_features
            self._init_state()
        if self.output_dim != 2:
            raise NotImplementedError("the device classifier is binary (output_dim = 2)")

    def _train(self, mlp, y1: np.ndarray, base_rows: np.ndarray | None = None) -> None:
        """The training loop of classifiers.py:350-433 over rows that are already staged in ``mlp``.
        ``y1`` holds the class-1 target of the rows the classifier sees; ``base_rows`` maps them to
        staged rows when the classifier is given a subset (perform_fdr trains on 80 %)."""
        at = (lambda r: r) if base_rows is None else (lambda r: base_rows[r])
        if self.experimental_hyperparameter_tuning:
            self.batch_size, self.learning_rate = scaled_training_params(len(y1))
            logger.info(f"Estimating optimal hyperparameters - samples: {len(y1):,}, batch_size: "
                        f"{self.batch_size:,}, learning_rate: {self.learning_rate:.2e}")
        train_rows, test_rows, schedule, epoch_of = self._plan(len(y1))
        self.last_fit_ms = 0.0
        t_test = np.stack([1 - y1[test_rows], y1[test_rows]], axis=1)
        done = 0
        n_steps = len(schedule)
        while done < n_steps:
            # run up to and including the next step that reports metrics (classifiers.py:395-428)
            stop = done
            while stop < n_steps and stop % self.metric_interval != 0:
                stop += 1
            stop = min(stop + 1, n_steps)
            loss = mlp.fit(at(train_rows), schedule[done:stop], self.batch_size, self.learning_rate,
                           self.weight_decay, self.dropout, seed=self._dropout_seed, first_step=done)
            self.last_fit_ms += mlp.time_ms()[0]
            last = stop - 1
            if last % self.metric_interval == 0:
                batch_rows = train_rows[schedule[last] : schedule[last] + self.batch_size]
                p_test = mlp.predict(at(test_rows))
                p_batch = mlp.predict(at(batch_rows))
                self.metrics["epoch"].append(int(epoch_of[last]))
                self.metrics["batch_count"].append(int(last))
                self.metrics["train_loss"].append(float(loss[-1]))
                self.metrics["test_loss"].append(_bce(p_test, t_test))
                self.metrics["train_accuracy"].append(float(np.mean(y1[batch_rows] == np.argmax(p_batch, axis=1))))
                self.metrics["test_accuracy"].append(float(np.mean(y1[test_rows] == np.argmax(p_test, axis=1)))
                                                     if len(test_rows) else float("nan"))
            done = stop
        self._state = mlp.get_state()
        self._fitted = True

    def fit(self, x: np.ndarray, y: np.ndarray) -> None:
        """classifiers.py:316-433, one optimiser step = three kernels on the device."""
        x = np.asarray(x)
        y = np.asarray(y)
        self._prepare(x.shape[1])
        y1 = y if y.ndim == 1 else y[:, 1]
        mlp = self._device_mlp()
        try:
            mlp.stage_rows(x, y1)
            self._train(mlp, y1)
        finally:
            mlp.close()

    def fit_resident(self, src_cols, decoy, extra_cols=(), subset=None):
        """Train on rows staged straight from the scoring tables in HBM (``adh_mlp_stage_rows_device``).
        ``subset(n_staged) -> rows`` picks the staged rows the classifier may see (perform_fdr's 80 %
        split).  Returns ``(mlp, table_rows)``: the live device network (the caller closes it) and the
        candidate row of every staged row."""
        self._prepare(len(src_cols))
        mlp = self._device_mlp()
        try:
            mlp.stage_rows_device(src_cols, decoy, extra_cols)
            table_rows = mlp.staged_rows()
            y_all = (np.asarray(decoy)[table_rows] != 0).astype(np.float64)
            try:
                base = None if subset is None else np.asarray(subset(len(table_rows)), dtype=np.int64)
            except TooFewPSMError as exc:
                exc.table_rows = table_rows  # the caller answers as perform_fdr does: every staged row, qval 1
                raise
            self._train(mlp, y_all if base is None else y_all[base], base_rows=base)
        except Exception:
            mlp.close()
            raise
        return mlp, table_rows

    def _forward(self, x: np.ndarray) -> np.ndarray:
        if not self.fitted:
            raise ValueError("Classifier has not been fitted yet.")
        x = np.asarray(x)
        assert x.ndim == 2, "Input data must have batch and feature dimension. (n_samples, n_features)"
        assert x.shape[1] == self.input_dim, \
            "Input data must have the same number of features as the fitted classifier."
        mlp = self._device_mlp()
        try:
            mlp.stage_rows(x)
            out = mlp.predict()
            self.last_predict_ms = mlp.time_ms()[1]
        finally:
            mlp.close()
        return out

    def predict(self, x: np.ndarray) -> np.ndarray:
        return np.argmax(self._forward(x), axis=1)

    def predict_proba(self, x: np.ndarray) -> np.ndarray:
        return self._forward(x)


# --------------------------------------------------------------------------------------------
# q-values, best row per group, perform_fdr
# --------------------------------------------------------------------------------------------
def _int_key(df: pd.DataFrame, columns: list[str]) -> np.ndarray:
    """One int64 key whose order is the lexicographic order of ``columns``."""
    if len(columns) == 1 and pd.api.types.is_integer_dtype(df[columns[0]].dtype):
        return df[columns[0]].to_numpy().astype(np.int64)
    order = np.lexsort([df[c].to_numpy() for c in reversed(columns)])
    vals = np.stack([df[c].to_numpy()[order] for c in columns], axis=1)
    new = np.ones(len(df), dtype=bool)
    if len(df) > 1:
        new[1:] = np.any(vals[1:] != vals[:-1], axis=1)
    key = np.empty(len(df), dtype=np.int64)
    key[order] = np.cumsum(new) - 1
    return key


def get_q_values(df: pd.DataFrame, score_column: str = "proba", decoy_column: str = "_decoy",
                 qval_column: str = "qval", extra_sort_columns: list[str] | None = None,
                 device: int | None = None) -> pd.DataFrame:
    """fdr.py:232-297 on the device: the frame comes back sorted by (score, decoy, extra columns)."""
    if extra_sort_columns is None:
        extra_sort_columns = ["precursor_idx"]
    ctx = runtime.get_context(device)
    tiebreak = _int_key(df, list(extra_sort_columns)) if len(extra_sort_columns) else None
    order, qval = ctx.fdr_q_values(df[score_column].to_numpy(), df[decoy_column].to_numpy(), tiebreak)
    df = df.iloc[order].copy()
    df[qval_column] = qval
    return df


def keep_best(df: pd.DataFrame, score_column: str = "proba", group_columns: list[str] | None = None,
              device: int | None = None) -> pd.DataFrame:
    """fdr.py:181-213 on the device: the lowest-score row of every group, input order kept."""
    if group_columns is None:
        group_columns = ["channel", "precursor_idx"]
    ctx = runtime.get_context(device)
    df = df.reset_index(drop=True)
    if len(group_columns) == 2 and all(pd.api.types.is_integer_dtype(df[c].dtype) for c in group_columns):
        a, b = (df[c].to_numpy().astype(np.int64) for c in group_columns)
    else:
        a, b = _int_key(df, list(group_columns)), None
    keep = ctx.fdr_keep_best(df[score_column].to_numpy(), a, b)
    return df[keep].reset_index(drop=True)


def perform_fdr(classifier, available_columns: list[str], df_target: pd.DataFrame, df_decoy: pd.DataFrame, *,
                competitive: bool = False, group_channels: bool = True, figure_path: str | None = None,
                df_fragments: pd.DataFrame | None = None, dia_cycle: np.ndarray | None = None,
                fdr_heuristic: float = 0.1, random_state: int | None = None,
                device: int | None = None) -> pd.DataFrame:
    """fdr.py:24-178: classifier on the feature columns -> q-values -> fragment competition ->
    best row per group -> q-values.  ``figure_path`` is accepted and ignored (no plotting here)."""
    from alphadia_amd.fragcomp import FragmentCompetition

    n_t, n_d = len(df_target), len(df_decoy)
    df_target = df_target.dropna(subset=available_columns)
    df_decoy = df_decoy.dropna(subset=available_columns)
    if n_t - len(df_target) > 0:
        logger.warning(f"dropped {n_t - len(df_target)} target PSMs due to missing features")
    if n_d - len(df_decoy) > 0:
        logger.warning(f"dropped {n_d - len(df_decoy)} decoy PSMs due to missing features")
    total = len(df_target) + len(df_decoy)
    if total > 0 and abs(len(df_target) - len(df_decoy)) / (total / 2) > 0.1:
        logger.warning("FDR calculation may be inaccurate as there is more than 10% difference in the number "
                       f"of target and decoy PSMs ({len(df_target)} / {len(df_decoy)})")

    X = np.concatenate([df_target[available_columns].to_numpy(), df_decoy[available_columns].to_numpy()])
    y = np.concatenate([np.zeros(len(df_target)), np.ones(len(df_decoy))])
    psm_df = pd.concat([df_target, df_decoy])
    try:
        train_rows, _ = train_test_indices(len(X), 0.2, random_state)
    except TooFewPSMError:
        logger.warning("Too few PSMs for FDR classification, assigning qval=1.0 and proba=1.0 to all PSMs.")
        psm_df["qval"] = 1.0
        psm_df["proba"] = 1.0
        return psm_df

    if device is not None and getattr(classifier, "device", device) is None:
        classifier.device = device  # one FDR call stays on one GPU
    classifier.fit(X[train_rows], y[train_rows])
    psm_df["_decoy"] = y
    if competitive:
        group_columns = ["elution_group_idx", "channel"] if group_channels else ["elution_group_idx"]
    else:
        group_columns = ["precursor_idx"]
    psm_df["proba"] = classifier.predict_proba(X)[:, 1]
    psm_df = get_q_values(psm_df, "proba", "_decoy", device=device)

    if dia_cycle is not None and dia_cycle.shape[2] <= MAX_DIA_CYCLE_SHAPE:
        # fdr.py:146-163: compete for fragments among the rows below the heuristic FDR
        start_idx = int(psm_df["qval"].searchsorted(fdr_heuristic, side="left"))
        if start_idx == 0:
            start_idx = len(psm_df)
        if df_fragments is not None:
            psm_df = FragmentCompetition(device=device)(psm_df.iloc[:start_idx].copy(), df_fragments, dia_cycle)

    psm_df = keep_best(psm_df, group_columns=group_columns, device=device)
    return get_q_values(psm_df, "proba", "_decoy", device=device)


def perform_fdr_resident(classifier, available_columns: list[str], candidates: pd.DataFrame, *, rt_column: str = "rt_library",
                         competitive: bool = False, group_channels: bool = True, dia_cycle: np.ndarray | None = None,
                         fdr_heuristic: float = 0.1, random_state: int | None = None,
                         device: int | None = None) -> pd.DataFrame:
    """:func:`perform_fdr` for tables that are still in HBM.

    ``candidates`` has ONE ROW PER ROW of the device tables of the last scoring call (the assembled
    candidate table: ``precursor_idx, rank, decoy, elution_group_idx, channel`` and the library columns
    a classifier column needs, i.e. ``rt_column`` for ``delta_rt`` and ``mz_library``).  The feature
    rows are staged for the classifier, scored, ranked, put through fragment competition (fragment
    m/z from the fragment table in HBM) and reduced to the best row per group on the device; what
    comes back is the surviving PSMs with ``proba`` and ``qval`` - the frame perform_fdr returns,
    restricted to the identifying columns."""
    from alphadia_amd.scoring import DEFAULT_FEATURE_COLUMNS

    ctx = runtime.get_context(device)
    extras, src_cols = [], []
    for c in available_columns:
        if c in DEFAULT_FEATURE_COLUMNS:
            src_cols.append(DEFAULT_FEATURE_COLUMNS.index(c))
        elif c == "delta_rt":  # rt_observed - library rt (scoring.py:458)
            extras.append(candidates[rt_column].to_numpy().astype(np.float32))
            src_cols.append(-len(extras))
        elif c in candidates.columns:
            extras.append(candidates[c].to_numpy().astype(np.float32))
            src_cols.append(len(DEFAULT_FEATURE_COLUMNS) + len(extras) - 1)
        else:
            raise KeyError(f"classifier column {c!r} is neither a scoring feature nor a column of `candidates`")
    decoy = candidates["decoy"].to_numpy()

    def subset(n_staged):
        return train_test_indices(n_staged, 0.2, random_state)[0]

    id_columns = [c for c in ("precursor_idx", "rank", "elution_group_idx", "channel", "decoy") if c in candidates.columns]
    if getattr(classifier, "device", device) is None:
        classifier.device = device
    try:
        mlp, table_rows = classifier.fit_resident(src_cols, decoy, extras, subset=subset)
    except TooFewPSMError as exc:
        # fdr.py:125-137: too few PSMs for a train / test split -> every usable PSM with qval = proba = 1
        logger.warning("Too few PSMs for FDR classification, assigning qval=1.0 and proba=1.0 to all PSMs.")
        rows = np.asarray(getattr(exc, "table_rows", np.zeros(0, np.int64)), dtype=np.int64)
        out = candidates.iloc[rows][id_columns].copy()
        out["_decoy"] = out["decoy"].to_numpy().astype(np.float64)
        out["proba"] = 1.0
        out["qval"] = 1.0
        out["table_row"] = rows
        return out
    try:
        mlp.predict_resident()
        if competitive:
            ga = candidates["elution_group_idx"].to_numpy()
            gb = candidates["channel"].to_numpy() if group_channels else None
        else:
            ga, gb = candidates["precursor_idx"].to_numpy(), None
        rows, proba, qval = ctx.fdr_resident(mlp, ga, gb, tiebreak=candidates["precursor_idx"].to_numpy(),
                                             cycle=dia_cycle, fdr_heuristic=fdr_heuristic)
    finally:
        mlp.close()
    out = candidates.iloc[rows][id_columns].copy()
    out["_decoy"] = out["decoy"].to_numpy().astype(np.float64)
    out["proba"] = proba
    out["qval"] = qval
    out["table_row"] = rows
    return out


class HipFDRManager:
    """The decoy strategies of ``FDRManager.fit_predict`` (fdr_manager.py:105-232) over
    :func:`perform_fdr`; classifiers are versioned per feature-column set in memory."""

    def __init__(self, feature_columns: list, classifier_base, *, compete_for_fragments: bool = True,
                 dia_cycle: np.ndarray | None = None, random_state: int | None = None, device: int | None = None):
        self.feature_columns = list(feature_columns)
        self.classifier_base = classifier_base
        self.classifier_store = defaultdict(list)
        self._compete_for_fragments = compete_for_fragments
        self._dia_cycle = dia_cycle
        self._np_rng = None if random_state is None else np.random.default_rng(random_state)
        self._current_version = -1
        self._device = device

    @property
    def current_version(self) -> int:
        return self._current_version

    def get_classifier(self, available_columns: list, version: int = -1):
        key = tuple(sorted(available_columns))
        store = self.classifier_store.get(key)
        return deepcopy(store[version] if store else self.classifier_base)

    def fit_predict(self, features_df: pd.DataFrame, decoy_strategy: str, competitive: bool,
                    df_fragments: pd.DataFrame | None = None, decoy_channel: int = -1, version: int = -1):
        available = [c for c in features_df.columns if c in set(self.feature_columns)]
        if not available:
            raise ValueError("No feature columns found in features_df")
        by_decoy_column = decoy_strategy in ("precursor", "precursor_channel_wise")
        if by_decoy_column and "decoy" not in features_df.columns:
            raise ValueError("Column 'decoy' not found in features_df")
        if decoy_strategy in ("precursor_channel_wise", "channel") and "channel" not in features_df.columns:
            raise ValueError("Column 'channel' not found in features_df")
        if decoy_strategy == "channel":
            if decoy_channel == -1:
                raise ValueError("decoy_channel must be set if decoy_type is channel")
            if decoy_channel not in features_df["channel"].unique():
                raise ValueError(f"decoy_channel {decoy_channel} not found in features_df")
        if by_decoy_column:
            decoy_channel = -1

        classifier = self.get_classifier(available, version)
        seed = None if self._np_rng is None else int(self._np_rng.integers(0, 1_000_000))
        frags = df_fragments if self._compete_for_fragments else None
        common = dict(competitive=competitive, random_state=seed, device=self._device)

        if decoy_strategy == "precursor":
            psm_df = perform_fdr(classifier, available, features_df[features_df["decoy"] == 0].copy(),
                                 features_df[features_df["decoy"] == 1].copy(), group_channels=True,
                                 df_fragments=frags, dia_cycle=self._dia_cycle, **common)
        elif decoy_strategy == "precursor_channel_wise":
            parts = []
            for channel in features_df["channel"].unique():
                sub = features_df[features_df["channel"].isin([channel, decoy_channel])]
                parts.append(perform_fdr(classifier, available, sub[sub["decoy"] == 0].copy(),
                                         sub[sub["decoy"] == 1].copy(), group_channels=True, df_fragments=frags,
                                         dia_cycle=self._dia_cycle, **common))
            psm_df = pd.concat(parts)
        elif decoy_strategy == "channel":
            parts = []
            for channel in sorted(set(features_df["channel"].unique()) - {decoy_channel}):
                sub = features_df[features_df["channel"].isin([channel, decoy_channel])]
                parts.append(perform_fdr(classifier, available, sub[sub["channel"] != decoy_channel].copy(),
                                         sub[sub["channel"] == decoy_channel].copy(), group_channels=False, **common))
            psm_df = pd.concat(parts)
            psm_df.loc[psm_df["channel"] == decoy_channel, "decoy"] = 1
        else:
            raise ValueError(f"Invalid decoy_strategy: {decoy_strategy}")

        self._current_version += 1
        self.classifier_store[tuple(sorted(available))].append(classifier)
        return psm_df
