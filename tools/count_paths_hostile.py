"""Diagnostic (needs a -DTR_COUNT_PATHS build in TRHIP_LIB): how many wave-steps of tests/test_gpu_parity.py's hostile-operand
scene run on the fast / the exact arithmetic path of the cull kernel."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest
from toyrenderer_amd import rhi
L = rhi.load()
out = (C.c_ulonglong * 8)()
L.trhip_debug_read_stamps(out, 1)
rc = pytest.main(["-q", "-x", os.path.join(os.path.dirname(__file__), "..", "tests", "test_gpu_parity.py"), "-k", "hostile", "-p", "no:cacheprovider"])
L.trhip_debug_read_stamps(out, 1)
print("pytest rc", rc, "| wave-steps on the fast path", out[4], "on the exact path", out[5])
