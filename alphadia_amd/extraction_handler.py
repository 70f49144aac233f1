"""The "hip" extraction backend: the plug-in point of this repository.

The reference dispatches on ``config["search"]["extraction_backend"]`` in
``ExtractionHandler.create_handler``
(alphadia/workflow/peptidecentric/extraction_handler.py:70-119, with the
comment "add implementations for other backends here" at :114).
``HipExtractionHandler`` has the interface of ``ClassicExtractionHandler``
(extraction_handler.py:344-507): the same constructor arguments and the three
methods ``select_candidates`` / ``score_and_quantify_candidates`` /
``quantify_candidates``.  Scoring and quantification run on the GPU; candidate
selection is delegated to a selection handler passed in by the integration
(the reference's own ``ClassicExtractionHandler``), because selection is
outside this repository's scope.

INTEGRATION.md shows the three-line patch that registers the backend.
"""

from __future__ import annotations

import pandas as pd

from alphadia_amd.scoring import CandidateScoringConfig, HipCandidateScoring


class HipExtractionHandler:
    """MI355X backend with the ``ClassicExtractionHandler`` method surface."""

    # extraction_handler.py:370-376
    _base_scoring_config = {
        "score_grouped": False,
        "top_k_isotopes": 3,
        "reference_channel": -1,
        "precursor_mz_tolerance": 10,
        "fragment_mz_tolerance": 15,
    }

    def __init__(self, config, optimization_manager, fdr_manager, reporter, column_name_handler,
                 selection_handler=None, device: int | None = None):
        self._config = config
        self._optimization_manager = optimization_manager
        self._fdr_manager = fdr_manager
        self._reporter = reporter
        self._column_name_handler = column_name_handler
        self._selection_handler = selection_handler
        self._device = device
        # extraction_handler.py:400-409
        self._scoring_config = CandidateScoringConfig()
        self._scoring_config.update(
            {
                **self._base_scoring_config,
                "exclude_shared_ions": config["search"]["exclude_shared_ions"],
                "quant_window": config["search"]["quant_window"],
                "quant_all": config["search"]["quant_all"],
                "experimental_xic": config["search"]["experimental_xic"],
            }
        )

    def select_candidates(self, dia_data, spectral_library, apply_cutoff: bool = False) -> pd.DataFrame:
        """Candidate selection (extraction_handler.py:121-154) stays with the reference."""
        if self._selection_handler is None:
            raise NotImplementedError(
                "candidate selection is not part of the hip backend; construct the handler with "
                "selection_handler=ClassicExtractionHandler(...) (see INTEGRATION.md)"
            )
        return self._selection_handler.select_candidates(dia_data, spectral_library, apply_cutoff)

    def score_and_quantify_candidates(self, candidates_df, dia_data, spectral_library,
                                      top_k_fragments: int | None = None):
        """extraction_handler.py:449-486 with ``CandidateScoring`` replaced by the GPU operator."""
        self._scoring_config.update(
            {
                "precursor_mz_tolerance": self._optimization_manager.ms1_error,
                "fragment_mz_tolerance": self._optimization_manager.ms2_error,
                "top_k_fragments": top_k_fragments
                if top_k_fragments is not None
                else self._config["search"]["top_k_fragments_scoring"],
            }
        )
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
                   selection_handler=None):
    """What ``ExtractionHandler.create_handler`` returns for ``extraction_backend: hip``."""
    backend = config["search"]["extraction_backend"].lower()
    if backend != "hip":
        raise ValueError(
            f"Invalid extraction backend '{backend}' for alphadia_amd. Supported backend: 'hip'"
        )
    reporter.log_string(f"Using {backend} extraction backend", verbosity="info")
    return HipExtractionHandler(
        config, optimization_manager, fdr_manager, reporter, column_name_handler,
        selection_handler=selection_handler,
    )
