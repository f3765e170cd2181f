"""MI355X-native inference path for the VT-CNN2-family modulation classifiers of
peteroh23/ModulationDetectionCNN.  Product code: formats + host surface +
libmdc.so (HIP).  The CPU oracle lives in /oracle and is never imported from here."""
from .topology import Topology, synthetic_weights, synthetic_frames  # noqa: F401
from .model import VTCNN2, Model  # noqa: F401
from .frontend import frames_from_iq_u8  # noqa: F401
from . import callbacks  # noqa: F401

__all__ = ["Topology", "VTCNN2", "Model", "callbacks", "synthetic_weights", "synthetic_frames", "frames_from_iq_u8"]
