"""Loader for the CPU oracle (oracle/_build/liboracle_rp.so).  Test-side only: the
product package never imports this module (tests/test_host_logic.py checks that)."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "liboracle_rp.so")

_api = None


def load_oracle():
    global _api
    if _api is None:
        if not os.path.exists(ORACLE_SO):
            # building the checker is allowed anywhere g++ exists (CPU only, seconds)
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        from calico_amd.capi import Api
        _api = Api(C.CDLL(ORACLE_SO), prefix="oracle_rp_")
        _api.lib.oracle_rp_set_threads.argtypes = [C.c_int]
        _api.lib.oracle_rp_set_threads.restype = C.c_int
    return _api
