#!/usr/bin/env python3
"""Drop-in for the reference script of the same name: same argv, same recipe
grammar, same stdout; the numerics run on the MI355X through libspkd_hip.so.
spk-diarization2.py invokes it as ./spk-clustering2.py from the working directory."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
cli = importlib.import_module('speaker-diarization_amd.cli')

if __name__ == '__main__':
    sys.exit(cli.main_clustering(variant=2))
