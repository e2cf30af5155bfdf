"""tools/step_overhead.py against a variant build: LIBV=abl/libbohip_xxx.so python tools/step_variant.py"""
import os, sys
sys.path.insert(0, ".")
from bohip import _lib
if os.environ.get("LIBV"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["LIBV"])
exec(open("tools/step_overhead.py").read())
