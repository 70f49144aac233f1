"""Multiplex requantification on the HIP backend.

Plug-in counterpart of ``MultiplexingRequantificationHandler.requantify``
(alphadia/workflow/peptidecentric/multiplexing_requantification_handler.py:44-149).  The work is in
:func:`alphadia_amd.scoring.requantify_multiplexed` (candidate expansion over the label channels +
grouped scoring on the GPU); this class only reads the workflow's configuration and hands the
feature table to the workflow's own FDR manager.
"""

from __future__ import annotations

import pandas as pd

from alphadia_amd.scoring import requantify_multiplexed


class HipMultiplexingRequantificationHandler:
    def __init__(self, config, calibration_manager, fdr_manager, reporter, column_name_handler,
                 spectral_library, device: int | None = None):
        self._config, self._calibration, self._fdr, self._reporter = config, calibration_manager, fdr_manager, reporter
        self._names, self._library, self._device = column_name_handler, spectral_library, device

    def requantify(self, dia_data, psm_df: pd.DataFrame) -> pd.DataFrame:
        if "multiplexing" not in self._config:
            raise ValueError("no multiplexing config found")
        mp = self._config["multiplexing"]
        if self._calibration is not None:  # calibrated columns of the unfiltered library (handler :45-50)
            self._calibration.predict(self._library.precursor_df_unfiltered, "precursor")
            self._calibration.predict(self._library._fragment_df, "fragment")
        # every channel that occurs anywhere: identified, reference, targets, decoy (handler :60-93)
        channels = sorted({*psm_df["channel"].unique().tolist(), mp["reference_channel"], mp["decoy_channel"],
                           *(int(c) for c in str(mp["target_channels"]).split(","))})
        self._reporter.log_string(f"=== Multiplexing {len(psm_df):,} precursors over channels {channels} ===",
                                  verbosity="progress")
        features, _ = requantify_multiplexed(
            dia_data, psm_df, self._library.precursor_df_unfiltered, self._library.fragment_df, channels,
            mp["reference_channel"], self._config["search"]["experimental_xic"],
            dict(rt_column=self._names.get_rt_column(), mobility_column=self._names.get_mobility_column(),
                 precursor_mz_column=self._names.get_precursor_mz_column(),
                 fragment_mz_column=self._names.get_fragment_mz_column()),
            device=self._device,
        )
        return self._fdr.fit_predict(features, decoy_strategy="channel", competitive=mp["competitive_scoring"],
                                     decoy_channel=mp["decoy_channel"])
