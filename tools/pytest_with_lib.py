#!/usr/bin/env python
"""Run pytest against an alternative build of the library (kernel experiments on the GPU box):
    python tools/pytest_with_lib.py <path/to/lib.so> [pytest args...]
The path is exported as L2HMC_LIB (l2hmc_amd/_ffi.py honours it), so processes the tests spawn load the same build."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if __name__ == "__main__":          # (spawned test workers re-import this file as __mp_main__: they must not start pytest again)
    sys.path.insert(0, ROOT)
    os.chdir(ROOT)
    os.environ["L2HMC_LIB"] = os.path.abspath(sys.argv[1])
    import pytest
    sys.exit(pytest.main(sys.argv[2:]))
