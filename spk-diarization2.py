#!/usr/bin/env python3
"""Drop-in for the reference orchestrator of the same name (plumbing only): runs
./generate_exp.py and feacat, then ./voice-detection2.py, ./spk-change-detection.py,
./spk-clustering.py and the exporters from the working directory, with the reference's
exact argv lists (spk-diarization2.py:89-138)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
orchestrator = importlib.import_module('speaker-diarization_amd.orchestrator')

if __name__ == '__main__':
    sys.exit(orchestrator.main())
