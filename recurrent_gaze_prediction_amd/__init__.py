"""MI355X-native (gfx950) recurrent gaze-prediction path.

Host side mirrors the reference's Python interface (``models.base``,
``models.gaze_rnn``, ``models.gaze_grcn``, ``evaluation_metrics``); compute runs in
hand-written HIP kernels behind the C ABI of ``include/rgp.h`` (``librgp_hip.so``).
"""
__version__ = '0.1.0'
