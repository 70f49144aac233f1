"""alphadia_amd: MI355X-native candidate scoring hot path for alphaDIA.

Only the peptide-centric candidate-scoring path (XIC extraction, the 46-feature
stack, fragment competition) lives here; see DESIGN.md for scope.
"""

__version__ = "0.1.0"
