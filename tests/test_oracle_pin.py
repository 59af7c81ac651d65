"""The checker every GPU test calls (orc_csr_spmv64, tests/oraclelib.csr_spmv64) is the pinned oracle (orc_csr_spmv = the reference's
CSR self-check loop, spmv.cpp:1843-1850, pinned by tests/golden/*.npz) bit for bit, and the product's own host loop (cvr_csr_spmv_host,
which spmv.cvr's verdict uses) equals the reference's y on the fixtures.  The same checks run on the CPU and -- marked gpu -- on the GPU
box, so that the box pins what its parity tests rely on."""
import glob
import os

import numpy as np
import pytest

import cvr_amd
import oraclelib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz")))


def _oracle64_is_the_pinned_oracle(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    numRows = int(z["dims"][1])
    rp, ci, va = z["csr_rowptr"][: numRows + 1].astype(np.int64), z["csr_col"], z["csr_val"]
    for mode in ("ones", "rand"):
        x = z[f"x_{mode}"]
        y64, absy = O.csr_spmv64(rp, ci, va, x)
        assert np.array_equal(y64.view(np.uint64), z[f"y_csr_{mode}"][:numRows].view(np.uint64)), (name, mode)      # == the reference's own y
        y32 = np.zeros(numRows, dtype=np.float64)
        rp32, ci32 = np.ascontiguousarray(z["csr_rowptr"], dtype=np.int32), np.ascontiguousarray(ci, dtype=np.int32)      # (kept alive across the call)
        va64, x64 = np.ascontiguousarray(va, dtype=np.float64), np.ascontiguousarray(x, dtype=np.float64)
        O.lib().orc_csr_spmv(numRows, rp32.ctypes.data, ci32.ctypes.data, va64.ctypes.data, x64.ctypes.data, y32.ctypes.data)
        assert np.array_equal(y64.view(np.uint64), y32.view(np.uint64)), (name, mode)
        assert np.all(absy >= np.abs(y64) * (1 - 1e-15))


def _host_loop_is_the_reference_loop(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    numRows = int(z["dims"][1])
    for mode, code in (("ones", 0), ("rand", 1)):
        x = z[f"x_{mode}"]
        assert np.array_equal(cvr_amd.fill_x(len(x), code), x)
        y = cvr_amd.csr_spmv_host(z["csr_rowptr"][: numRows + 1], z["csr_col"], z["csr_val"], x, nthreads=2)
        assert np.array_equal(y.view(np.uint64), z[f"y_csr_{mode}"].view(np.uint64))


@pytest.mark.parametrize("name", NAMES)
def test_oracle64_equals_pinned_oracle(name):
    _oracle64_is_the_pinned_oracle(name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_oracle64_equals_pinned_oracle_on_the_gpu_box(name):
    _oracle64_is_the_pinned_oracle(name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_host_csr_loop_equals_reference_on_the_gpu_box(name):
    _host_loop_is_the_reference_loop(name)
