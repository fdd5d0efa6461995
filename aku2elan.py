#!/usr/bin/env python3
"""Drop-in for the reference script of the same name: same argv, same output.  Host-side only."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
ex = importlib.import_module('speaker-diarization_amd.exporters')

if __name__ == '__main__':
    sys.exit(ex.main_aku2elan())
