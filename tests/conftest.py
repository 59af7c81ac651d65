"""pytest configuration: registers the `gpu` marker and builds the checker library on demand."""
import os
import subprocess
import sys

import pytest

try:                   # PyTorch (used by a few GPU tests for device arrays and torch.distributed) brings its own copy of the HIP
    import torch       # runtime; the copy initialised second in a process finds no GPU, so it is loaded before libcvr_amd.so
except Exception:      # noqa: BLE001 -- those tests skip themselves without it
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    lib = os.path.join(ROOT, "oracle", "liboracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("cvr_oracle.c", "cvr64_mirror.c", "cvr_oracle.h")]
    if not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in srcs):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all"], check=True,
                       stdout=subprocess.DEVNULL)
    # the product library (hipcc cross-compiles gfx950 without a GPU); normally built by __graft_entry__.build()
    prod = os.path.join(ROOT, "cvr_amd", "libcvr_amd.so")
    if not os.path.exists(prod) or not os.path.exists(os.path.join(ROOT, "spmv.cvr")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "cvr_amd", "csrc"), "all"], check=True,
                       stdout=subprocess.DEVNULL)
    yield
