"""Child process of test_library_before_torch_in_one_process: the library first, torch second."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hipabi = importlib.import_module('speaker-diarization_amd.hipabi')
engine = importlib.import_module('speaker-diarization_amd.engine')
synth = importlib.import_module('speaker-diarization_amd.synth')
eng = engine.HipEngine()
feats, _, truth = synth.make_session(4242, 300, 3)
eng.set_features(feats)
segs = [(a, b) for a, b, _ in truth]
r = eng.cluster_hi(segs, 1, 'BIC', 1.3, 0.0, 0)
print('library first: merges', len(r.merges))
import torch
print('then torch:', torch.cuda.is_available())
x = torch.ones(8, device='cuda'); print(float(x.sum()))
r = eng.cluster_hi(segs, 1, 'BIC', 1.3, 0.0, 0)
print('library again: merges', len(r.merges))
