"""MI355X-native BIC/GLR/KL2 speaker-change detection and agglomerative
clustering (the hot path of spk-change-detection.py / spk-clustering(2).py).

The directory name carries a hyphen, so import it with
``importlib.import_module('speaker-diarization_amd')`` (the repo root on sys.path).
"""
__all__ = ['recipe', 'synth']
