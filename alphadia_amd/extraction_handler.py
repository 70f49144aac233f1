"""The "hip" extraction backend: the plug-in point of this repository.

The reference dispatches on ``config["search"]["extraction_backend"]`` in
``ExtractionHandler.create_handler``
(alphadia/workflow/peptidecentric/extraction_handler.py:70-119, with the
comment "add implementations for other backends here" at :114).
``HipExtractionHandler`` has the interface of ``ClassicExtractionHandler``
(extraction_handler.py:344-507): the same constructor arguments and the three
methods ``select_candidates`` / ``score_and_quantify_candidates`` /
``quantify_candidates``.  Candidate selection (``HipCandidateSelection``,
SURVEY.md section 8f-1), scoring and quantification all run on the GPU, for runs
with and without ion mobility; an optional ``selection_handler`` (e.g. the
reference's own ``ClassicExtractionHandler``) can take over the selection step.

INTEGRATION.md shows the three-line patch that registers the backend.
"""

from __future__ import annotations

import pandas as pd

from alphadia_amd.scoring import CandidateScoringConfig, HipCandidateScoring
from alphadia_amd.selection import CandidateSelectionConfig, HipCandidateSelection


class HipExtractionHandler:
    """MI355X backend with the ``ClassicExtractionHandler`` method surface."""

    # the values ClassicExtractionHandler passes to the two config classes
    # (extraction_handler.py:348-376)
    _base_selection_config = dict(
        peak_len_rt=10.0, sigma_scale_rt=0.5, peak_len_mobility=0.01, sigma_scale_mobility=1.0,
        top_k_precursors=3, kernel_size=30, f_mobility=1.0, f_rt=0.99, center_fraction=0.5,
        min_size_mobility=8, min_size_rt=3, max_size_mobility=20, max_size_rt=15, group_channels=False,
        use_weighted_score=True, join_close_candidates=False, join_close_candidates_scan_threshold=0.6,
        join_close_candidates_cycle_threshold=0.6,
    )
    _base_scoring_config = dict(score_grouped=False, top_k_isotopes=3, reference_channel=-1,
                                precursor_mz_tolerance=10, fragment_mz_tolerance=15)

    def __init__(self, config, optimization_manager, fdr_manager, reporter, column_name_handler,
                 selection_handler=None, device: int | None = None):
        self._config = config
        self._optimization_manager = optimization_manager
        self._fdr_manager = fdr_manager
        self._reporter = reporter
        self._column_name_handler = column_name_handler
        self._selection_handler = selection_handler
        self._device = device
        # extraction_handler.py:390-398
        self._selection_config = CandidateSelectionConfig()
        search = config["search"]
        self._selection_config.update(dict(
            self._base_selection_config, top_k_fragments=search["top_k_fragments_selection"],
            exclude_shared_ions=search["exclude_shared_ions"], min_size_rt=search["quant_window"]))
        # extraction_handler.py:400-409
        self._scoring_config = CandidateScoringConfig()
        self._scoring_config.update(dict(
            self._base_scoring_config, exclude_shared_ions=search["exclude_shared_ions"],
            quant_window=search["quant_window"], quant_all=search["quant_all"],
            experimental_xic=search["experimental_xic"]))

    def _select_candidates(self, dia_data, spectral_library) -> pd.DataFrame:
        """extraction_handler.py:411-446 with ``CandidateSelection`` replaced by the GPU operator."""
        om = self._optimization_manager
        self._selection_config.update(dict(
            rt_tolerance=om.rt_error, mobility_tolerance=om.mobility_error, candidate_count=om.num_candidates,
            precursor_mz_tolerance=om.ms1_error, fragment_mz_tolerance=om.ms2_error))
        selection = HipCandidateSelection(
            dia_data,
            spectral_library.precursor_df,
            spectral_library.fragment_df,
            self._selection_config,
            rt_column=self._column_name_handler.get_rt_column(),
            mobility_column=self._column_name_handler.get_mobility_column(),
            precursor_mz_column=self._column_name_handler.get_precursor_mz_column(),
            fragment_mz_column=self._column_name_handler.get_fragment_mz_column(),
            fwhm_rt=om.fwhm_rt,
            fwhm_mobility=om.fwhm_mobility,
            device=self._device,
        )
        return selection(thread_count=self._config["general"]["thread_count"])

    def select_candidates(self, dia_data, spectral_library, apply_cutoff: bool = False) -> pd.DataFrame:
        """extraction_handler.py:121-154; both run layouts are selected on the GPU.  A selection
        handler passed in by the integration (``selection_handler=``) takes precedence."""
        if self._selection_handler is not None:
            return self._selection_handler.select_candidates(dia_data, spectral_library, apply_cutoff)
        self._reporter.log_string(
            f"Extracting batch of {len(spectral_library.precursor_df)} precursors", verbosity="progress"
        )
        candidates_df = self._select_candidates(dia_data, spectral_library)
        if apply_cutoff:
            # extraction_handler.py:177-203 ("filter 1")
            num_before = len(candidates_df)
            candidates_df = candidates_df[candidates_df["score"] > self._optimization_manager.score_cutoff]
            num_after = len(candidates_df)
            num_removed = num_before - num_after
            self._reporter.log_string(
                f"Removed {num_removed} precursors with score below cutoff "
                f"{self._optimization_manager.score_cutoff}. {num_after} precursors remain.",
                verbosity="info",
            )
        return candidates_df

    def score_and_quantify_candidates(self, candidates_df, dia_data, spectral_library,
                                      top_k_fragments: int | None = None):
        """extraction_handler.py:449-486 with ``CandidateScoring`` replaced by the GPU operator."""
        om = self._optimization_manager
        k = top_k_fragments if top_k_fragments is not None else self._config["search"]["top_k_fragments_scoring"]
        self._scoring_config.update(dict(precursor_mz_tolerance=om.ms1_error, fragment_mz_tolerance=om.ms2_error,
                                         top_k_fragments=k))
        candidate_scoring = HipCandidateScoring(
            dia_data=dia_data,
            precursors_flat=spectral_library.precursor_df,
            fragments_flat=spectral_library.fragment_df,
            config=self._scoring_config,
            rt_column=self._column_name_handler.get_rt_column(),
            mobility_column=self._column_name_handler.get_mobility_column(),
            precursor_mz_column=self._column_name_handler.get_precursor_mz_column(),
            fragment_mz_column=self._column_name_handler.get_fragment_mz_column(),
            device=self._device,
        )
        return candidate_scoring(
            candidates_df,
            thread_count=self._config["general"]["thread_count"],
            include_decoy_fragment_features=True,
        )

    def quantify_candidates(self, candidates_df, precursor_fdr_df, dia_data, spectral_library,
                            top_k_fragments: int | None = None):
        """extraction_handler.py:488-507: scoring and quantification are one pass here too."""
        del precursor_fdr_df
        _features_df, fragments_df = self.score_and_quantify_candidates(
            candidates_df, dia_data, spectral_library, top_k_fragments
        )
        return None, fragments_df


def create_handler(config, optimization_manager, fdr_manager, reporter, column_name_handler,
                   selection_handler=None, device: int | None = None):
    """What ``ExtractionHandler.create_handler`` returns for ``extraction_backend: hip``."""
    backend = config["search"]["extraction_backend"].lower()
    if backend != "hip":
        raise ValueError(
            f"Invalid extraction backend '{backend}' for alphadia_amd. Supported backend: 'hip'"
        )
    reporter.log_string(f"Using {backend} extraction backend", verbosity="info")
    return HipExtractionHandler(
        config, optimization_manager, fdr_manager, reporter, column_name_handler,
        selection_handler=selection_handler, device=device,
    )
