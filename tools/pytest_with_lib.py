#!/usr/bin/env python
"""Run pytest against an alternative build of the library (kernel experiments on the GPU box):
    python tools/pytest_with_lib.py <path/to/lib.so> [pytest args...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from l2hmc_amd import _ffi
_ffi.LIB_PATH = os.path.abspath(sys.argv[1])
import pytest
sys.exit(pytest.main(sys.argv[2:]))
