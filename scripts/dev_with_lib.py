"""Dev: run a script of this repo (bench.py, a dev_* script) against ANOTHER build of librgp_hip.so -- an older one, or a
`make DEV=1` build that reads the RGP_* development knobs -- without touching the in-tree library:
    python scripts/dev_with_lib.py <path/to/lib.so> bench.py --workload finetune ...
The binding is narrowed to the symbols that build exports (an older build lacks the newer entry points)."""
import ctypes
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib, script = os.path.abspath(sys.argv[1]), sys.argv[2]
from recurrent_gaze_prediction_amd import _lib  # noqa: E402

_lib.LIB_PATH = lib
_probe = ctypes.CDLL(lib)
_lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if hasattr(_probe, k)}
sys.argv = [script] + sys.argv[3:]
runpy.run_path(os.path.join(ROOT, script) if not os.path.isabs(script) else script, run_name='__main__')
