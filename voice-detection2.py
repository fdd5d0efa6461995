#!/usr/bin/env python3
"""Drop-in for the reference script of the same name (speech / non-speech turn recipe
from VAD .exp files): same argv, same recipe text, same stdout.  Host-side only."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
vd = importlib.import_module('speaker-diarization_amd.voice_detection')

if __name__ == '__main__':
    sys.exit(vd.main())
