"""Multiplex requantification on the HIP backend.

``HipMultiplexingRequantificationHandler`` has the constructor and the ``requantify`` method of the
reference's ``MultiplexingRequantificationHandler``
(alphadia/workflow/peptidecentric/multiplexing_requantification_handler.py:23-149): the best
candidate of every elution group is spread over all label channels, the channel copies are scored
as one score group with the reference channel's presence as the gate, and the reference's own
FDR manager assigns the channel-decoy q-values.  Only the scoring operator differs.
"""

from __future__ import annotations

import pandas as pd

from alphadia_amd.scoring import CandidateScoringConfig, HipCandidateScoring, multiplex_candidates

MULTIPLEXING_CHANNELS_DELIM = ","  # alphadia/workflow/config.py

CANDIDATE_COLUMNS = [
    "elution_group_idx", "precursor_idx", "rank", "scan_start", "scan_stop", "scan_center",
    "frame_start", "frame_stop", "frame_center",
]


def candidate_features_to_candidates(candidate_features_df: pd.DataFrame, optional_columns=None) -> pd.DataFrame:
    """scoring/utils.py:69-111: the candidate columns of a feature table (plus ``proba``)."""
    if optional_columns is None:
        optional_columns = ["proba"]
    return candidate_features_df[CANDIDATE_COLUMNS + optional_columns].copy()


class HipMultiplexingRequantificationHandler:
    def __init__(self, config, calibration_manager, fdr_manager, reporter, column_name_handler,
                 spectral_library, device: int | None = None):
        self._config = config
        self._calibration_manager = calibration_manager
        self._fdr_manager = fdr_manager
        self._reporter = reporter
        self._column_name_handler = column_name_handler
        self._spectral_library = spectral_library
        self._device = device

    def channels(self, psm_df: pd.DataFrame):
        """multiplexing_requantification_handler.py:60-93"""
        original = psm_df["channel"].unique().tolist()
        mp = self._config["multiplexing"]
        reference_channel = mp["reference_channel"]
        target = [int(c) for c in str(mp["target_channels"]).split(MULTIPLEXING_CHANNELS_DELIM)]
        decoy_channel = mp["decoy_channel"]
        return list(set(original + [reference_channel] + target + [decoy_channel])), reference_channel, decoy_channel

    def requantify(self, dia_data, psm_df: pd.DataFrame) -> pd.DataFrame:
        if self._calibration_manager is not None:
            # multiplexing_requantification_handler.py:45-50 (group names of CalibrationGroups)
            self._calibration_manager.predict(self._spectral_library.precursor_df_unfiltered, "precursor")
            self._calibration_manager.predict(self._spectral_library._fragment_df, "fragment")
        reference_candidates = candidate_features_to_candidates(psm_df)
        if "multiplexing" not in self._config:
            raise ValueError("no multiplexing config found")
        self._reporter.log_string(
            f"=== Multiplexing {len(reference_candidates):,} precursors ===", verbosity="progress"
        )
        channels, reference_channel, decoy_channel = self.channels(psm_df)
        multiplexed = multiplex_candidates(
            reference_candidates, self._spectral_library.precursor_df_unfiltered, channels=channels
        )
        self._reporter.log_string(
            f"=== Requantifying {len(multiplexed):,} precursors ===", verbosity="progress"
        )
        config = CandidateScoringConfig()
        config.score_grouped = True
        config.exclude_shared_ions = True
        config.reference_channel = reference_channel
        config.experimental_xic = self._config["search"]["experimental_xic"]
        scoring = HipCandidateScoring(
            dia_data=dia_data,
            precursors_flat=self._spectral_library.precursor_df_unfiltered,
            fragments_flat=self._spectral_library.fragment_df,
            config=config,
            rt_column=self._column_name_handler.get_rt_column(),
            mobility_column=self._column_name_handler.get_mobility_column(),
            precursor_mz_column=self._column_name_handler.get_precursor_mz_column(),
            fragment_mz_column=self._column_name_handler.get_fragment_mz_column(),
            device=self._device,
        )
        multiplexed["rank"] = 0
        features, _fragments = scoring(multiplexed)
        return self._fdr_manager.fit_predict(
            features,
            decoy_strategy="channel",
            competitive=self._config["multiplexing"]["competitive_scoring"],
            decoy_channel=decoy_channel,
        )
