"""MI355X-native conformer line-OCR recognition path (drop-in for the reference's `PytorchRecognitionModel`)."""
import os as _os

# Callers that keep several batches in flight (`evaluate.recognize(streams=4)`, bench.py) need a hardware queue per stream: the HIP
# runtime multiplexes its streams onto GPU_MAX_HW_QUEUES (default 4) queues, and with 4 compute streams + copy streams on 4 queues
# independent batches serialise.  Read once, when the process initialises HIP -- set here in case the package is imported before that.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
