"""MI355X-native conformer line-OCR recognition path (drop-in for the reference's `PytorchRecognitionModel`)."""
import os as _os
import sys as _sys
import warnings as _warnings

# Callers that keep several batches in flight (`evaluate.recognize(streams=4)`, bench.py) need a hardware queue per stream: the HIP
# runtime multiplexes its streams onto GPU_MAX_HW_QUEUES (default 4) queues, and with 4 compute streams + copy streams on 4 queues
# independent batches serialise (30.2 k against 38.0 k lines/s).  The runtime reads the variable ONCE, when the process initialises HIP:
# a launcher should export GPU_MAX_HW_QUEUES=8 itself; it is set here in case the package is imported before that,
# and `hw_queues_note()` says so when it was not.
_HWQ_PRESET = 'GPU_MAX_HW_QUEUES' in _os.environ
_torch = _sys.modules.get('torch')
_HIP_UP_AT_IMPORT = bool(_torch is not None and _torch.cuda.is_initialized())
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
_warned = False


def hw_queues_note(streams: int = 2):
    """Called where a caller asks for several batches in flight: one warning if HIP was already initialised when this package set
    GPU_MAX_HW_QUEUES (the setting then has no effect: the streams share the runtime's default 4 hardware queues)."""
    global _warned
    if streams > 1 and not _HWQ_PRESET and _HIP_UP_AT_IMPORT and not _warned:
        _warned = True
        _warnings.warn('conformer_ocr_amd: HIP was initialised before this package was imported and GPU_MAX_HW_QUEUES was not set; '
                       f'{streams} batches in flight will share the default 4 hardware queues (about 20 % fewer lines/s). '
                       'Export GPU_MAX_HW_QUEUES=8 in the launcher.', RuntimeWarning, stacklevel=3)
    return {'gpu_max_hw_queues': _os.environ.get('GPU_MAX_HW_QUEUES'), 'preset_by_caller': _HWQ_PRESET,
            'hip_initialised_before_import': _HIP_UP_AT_IMPORT}
