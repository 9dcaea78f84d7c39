"""speechsplit_amd -- MI355X (gfx950) engine for the SpeechSplit Generator_3 / Generator_6 forward + training path.

The compute lives in ``lib/libspeechsplit_hip.so`` (hand-written HIP, C ABI in ``include/speechsplit_amd.h``);
this package is the host-side mirror of the reference's Python surface (model.py / solver.py / utils.py names).
"""
__all__ = ['engine', 'model', 'solver', 'hparams', 'utils', 'data_loader', 'dist']
