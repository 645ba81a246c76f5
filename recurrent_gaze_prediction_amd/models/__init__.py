"""Host-side mirror of the reference's model interface (``models.base``,
``models.gaze_rnn``, ``models.gaze_grcn``, ``models.model_util``): same class / method /
config names and the same ``generate()`` dictionary, with the TF session replaced by
HIP kernels behind ``librgp_hip.so``."""
