#!/bin/bash
# GPU suite + stop-phase times of the current build
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
PHASES="${PHASES:-21 22 2 0}" bash tools/feature_phases.sh
