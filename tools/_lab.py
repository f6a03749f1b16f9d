"""Import first in a tuning tool: selects the lab build of the library (libfrp_lab.so: the product library + the tuning
hooks of include/frp_lab.h) unless FRP_LIB already names one.  `make -C face-recognition-platform_amd/csrc lab` builds it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LAB = os.path.join(ROOT, "face-recognition-platform_amd", "libfrp_lab.so")
if "FRP_LIB" not in os.environ:
    if not os.path.exists(_LAB):
        sys.exit(f"{_LAB} not built: make -C face-recognition-platform_amd/csrc lab")
    os.environ["FRP_LIB"] = _LAB
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
