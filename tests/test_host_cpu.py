"""CPU: the host side of the product through the C ABI -- loader against the golden outputs of the unmodified
reference, symbol table, error behaviour without a GPU.  No compute calls that need a device."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

import cvr_amd
from cvr_amd import capi
import oraclelib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz")))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "cvr_amd.h")).read()
    declared = set(re.findall(r"\b(cvr_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"cvr_csr_view", "cvr_handle"}
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    L = C.CDLL(capi.lib_path())
    for s in capi.SYMBOLS:
        assert hasattr(L, s), s


@pytest.mark.parametrize("name", NAMES)
def test_product_loader_equals_reference_loader(name):
    """cvr_mm_read(REFCOMPAT) == readMatrix (spmv.cpp:311-535) bit for bit, on the reference's own outputs"""
    z = np.load(os.path.join(GOLD, name + ".npz"))
    m = cvr_amd.load_mm(os.path.join(GOLD, "mtx", name + ".mtx"), capi.MM_REFCOMPAT)
    nItems, numRows, numCols = (int(v) for v in z["dims"])
    assert (m["ref_nItems"], m["ref_numRows"], m["ref_numCols"]) == (nItems, numRows, numCols)
    assert m["nrows"] == numRows + 1 and m["ncols"] >= numCols + 1
    assert np.array_equal(m["row_ptr"], z["csr_rowptr"].astype(np.int64))
    assert np.array_equal(m["col_idx"], z["csr_col"])
    assert np.array_equal(m["vals"].view(np.uint64), z["csr_val"].view(np.uint64))


@pytest.mark.parametrize("name", NAMES)
def test_host_csr_loop_equals_reference(name):
    """cvr_csr_spmv_host == the reference's self-check loop (spmv.cpp:1843-1850) bit for bit"""
    z = np.load(os.path.join(GOLD, name + ".npz"))
    numRows = int(z["dims"][1])
    for mode, code in (("ones", 0), ("rand", 1)):
        x = z[f"x_{mode}"]
        assert np.array_equal(cvr_amd.fill_x(len(x), code), x)
        y = cvr_amd.csr_spmv_host(z["csr_rowptr"][: numRows + 1], z["csr_col"], z["csr_val"], x, nthreads=2)
        assert np.array_equal(y.view(np.uint64), z[f"y_csr_{mode}"].view(np.uint64))


def test_strict_loader_semantics():
    m = cvr_amd.load_mm(os.path.join(GOLD, "mtx", "sym4_pattern.mtx"), capi.MM_STRICT)
    assert m["nrows"] == 4 and m["ncols"] == 4
    assert np.all(m["vals"] == 1.0) and m["row_ptr"][0] == 0 and m["col_idx"].min() >= 0
    # symmetric: A == A^T
    A = np.zeros((4, 4))
    for r in range(4):
        A[r, m["col_idx"][m["row_ptr"][r]:m["row_ptr"][r + 1]]] = 1
    assert np.array_equal(A, A.T)
    d = cvr_amd.load_mm(os.path.join(GOLD, "mtx", "dense4_nonl.mtx"), capi.MM_STRICT)
    assert d["nnz"] == 16                      # the reference drops the unterminated last line (Q5); strict keeps it
    assert d["vals"][0] == 1.03                # fp64, not through a float (Q2)


def test_loader_errors_are_codes_not_exits(tmp_path):
    with pytest.raises(cvr_amd.CvrError) as e:
        cvr_amd.load_mm(str(tmp_path / "missing.mtx"))
    assert e.value.code == capi.ERR_IO
    p = tmp_path / "array.mtx"
    p.write_text("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n")
    with pytest.raises(cvr_amd.CvrError) as e:
        cvr_amd.load_mm(str(p))
    assert e.value.code == capi.ERR_IO


def test_verdict_rule():
    y = np.array([0.0, 1.0, 2.0, 3.0])
    yr = np.array([0.0, 1.0 + 9e-4, 2.0 + 2e-3, 3.5])
    assert cvr_amd.verdict(y, yr, 4) == 2       # |d|^2 > 1e-6 (spmv.cpp:1924)
    assert cvr_amd.verdict(y, yr, 3) == 1       # only the rows asked for (spmv.cpp:1920)


def test_invalid_csr_is_rejected_before_any_device_work():
    rp = np.array([0, 2, 1], dtype=np.int64)
    with pytest.raises(cvr_amd.CvrError) as e:
        cvr_amd.CvrMatrix(2, 4, rp, np.zeros(2, dtype=np.int32), np.zeros(2))
    assert e.value.code == capi.ERR_INVALID
    rp = np.array([0, 1, 2], dtype=np.int64)
    with pytest.raises(cvr_amd.CvrError) as e:
        cvr_amd.CvrMatrix(2, 4, rp, np.array([0, 7], dtype=np.int32), np.zeros(2))
    assert e.value.code == capi.ERR_INVALID and "col_idx" in str(e.value)


def test_no_cpu_fallback_without_a_device():
    if cvr_amd.device_count() > 0:
        pytest.skip("a GPU is visible")
    rp = np.array([0, 1, 2], dtype=np.int64)
    with pytest.raises(cvr_amd.CvrError) as e:
        cvr_amd.CvrMatrix(2, 4, rp, np.array([0, 3], dtype=np.int32), np.ones(2))
    assert e.value.code == capi.ERR_NO_DEVICE


def test_synthetic_refcompat_view_matches_loader_convention(tmp_path):
    """synth.to_refcompat builds the arrays the reference loader would build from a row-major `pattern` file
    of the same entries: checked through the product loader, itself pinned to the reference's loader above"""
    from cvr_amd import synth
    n, nc, rp, ci, va = synth.web_google_like(scale=0.002)
    p = tmp_path / "wg.mtx"
    rows = np.repeat(np.arange(n), np.diff(rp))
    with open(p, "w") as f:
        f.write("%%MatrixMarket matrix coordinate pattern general\n")
        f.write(f"{n} {nc} {len(ci)}\n")
        f.write("".join(f"{r + 1} {c + 1}\n" for r, c in zip(rows, ci)))
    m = cvr_amd.load_mm(str(p), capi.MM_REFCOMPAT)
    rc = synth.to_refcompat(n, nc, rp, ci, va)
    assert (rc["nItems"], rc["nItemsRaw"], rc["numRows"]) == (m["ref_nItems"], m["ref_nItemsRaw"], m["ref_numRows"])
    assert np.array_equal(rc["rowptr"].astype(np.int64), m["row_ptr"])
    assert np.array_equal(rc["cols"], m["col_idx"])
    assert np.array_equal(rc["val"], m["vals"])


@pytest.mark.parametrize("mode", [capi.MM_REFCOMPAT, capi.MM_STRICT])
def test_binary_cache_roundtrip(tmp_path, mode):
    src = os.path.join(GOLD, "mtx", "sym250_real.mtx")
    cache = str(tmp_path / "m.cvrbin")
    a = cvr_amd.load_mm(src, mode, cache=cache)           # parses the text, writes the image
    assert os.path.exists(cache)
    b = cvr_amd.load_mm("/nonexistent/ignored.mtx", mode, cache=cache)   # served from the image alone
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    with open(cache, "r+b") as f:                          # a damaged image is an error code, not a crash
        f.write(b"garbage!")
    with pytest.raises(cvr_amd.CvrError):
        cvr_amd.load_mm(src, mode, cache=cache)


def test_keyed_cache_refuses_a_stale_image(tmp_path):
    """the cache beside a .mtx file is keyed to it (size, mtime, hash of the first and last MiB, loader mode): it is read while the
    file is what it was, and the text is parsed again -- and the cache rewritten -- once the file has changed (the reference
    parses on every run, spmv.cpp:411-451; an unkeyed cache silently returned the old arrays)"""
    import shutil
    import time
    src = str(tmp_path / "m.mtx")
    shutil.copy(os.path.join(GOLD, "mtx", "pl2000_pattern.mtx"), src)
    a = cvr_amd.load_mm(src, capi.MM_REFCOMPAT, cache=True)
    assert a["cache_hit"] is False and os.path.exists(src + ".ref.csrbin")
    b = cvr_amd.load_mm(src, capi.MM_REFCOMPAT, cache=True)
    assert b["cache_hit"] is True
    for k in ("row_ptr", "col_idx", "vals"):
        assert np.array_equal(a[k], b[k]), k
    assert cvr_amd.load_mm(src, capi.MM_STRICT, cache=True)["cache_hit"] is False      # the other mode has a cache of its own
    # the file changes (another matrix under the same name): the key no longer matches
    time.sleep(0.01)
    shutil.copy(os.path.join(GOLD, "mtx", "sym250_real.mtx"), src)
    c = cvr_amd.load_mm(src, capi.MM_REFCOMPAT, cache=True)
    fresh = cvr_amd.load_mm(src, capi.MM_REFCOMPAT)
    assert c["cache_hit"] is False and c["ref_numRows"] == fresh["ref_numRows"] != a["ref_numRows"]
    assert np.array_equal(c["vals"], fresh["vals"])
    assert cvr_amd.load_mm(src, capi.MM_REFCOMPAT, cache=True)["cache_hit"] is True      # rewritten under the new key
    # same size, same bytes, touched: mtime is part of the key
    os.utime(src, ns=(1, 1))
    assert cvr_amd.load_mm(src, capi.MM_REFCOMPAT, cache=True)["cache_hit"] is False
    # the keyed reader itself: a key that is not the image's is CVR_ERR_STATE, an unkeyed image too
    k = capi.source_key(src, capi.MM_REFCOMPAT)
    m = capi.MmMatrix()
    assert capi.lib().cvr_mm_read_bin_keyed(os.fsencode(src + ".ref.csrbin"), C.byref(k), C.byref(m)) == 0
    capi.lib().cvr_mm_free(C.byref(m))
    k.hash ^= 1
    assert capi.lib().cvr_mm_read_bin_keyed(os.fsencode(src + ".ref.csrbin"), C.byref(k), C.byref(m)) == capi.ERR_STATE
    with pytest.raises(cvr_amd.CvrError):
        capi.source_key(str(tmp_path / "missing.mtx"))


def test_parallel_parse_matches_on_a_large_file(tmp_path):
    """a file large enough to be cut into one text segment per thread: both loader modes against numpy"""
    from cvr_amd import synth
    n, nc, rp, ci, va = synth.web_google_like(scale=0.02)
    rows = np.repeat(np.arange(n), np.diff(rp))
    perm = np.random.default_rng(5).permutation(len(ci))  # file order shuffled: the loader has to sort
    p = tmp_path / "big.mtx"
    with open(p, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n% comment\n")
        f.write(f"{n} {nc} {len(ci)}\n")
        f.write("".join(f"{rows[k] + 1} {ci[k] + 1} {va[k] + 0.25:.17g}\n" for k in perm))
    assert os.path.getsize(p) > (1 << 16)
    s = cvr_amd.load_mm(str(p), capi.MM_STRICT)
    assert np.array_equal(s["row_ptr"], rp) and np.array_equal(s["col_idx"], ci) and np.array_equal(s["vals"], va + 0.25)
    r = cvr_amd.load_mm(str(p), capi.MM_REFCOMPAT)
    raw = len(ci)
    assert r["ref_nItemsRaw"] == raw and r["ref_nItems"] % 16 == 0
    # same entries 1-based, values through a float, plus zero-valued pad copies of the file's last entry (Q6)
    npad, lc = r["ref_nItems"], int(ci[perm[-1]]) + 1
    assert len(r["col_idx"]) == npad and r["row_ptr"][-1] == npad - 1      # Q9: tail row pointers = nItems - 1
    exp_c = np.sort(np.concatenate([ci + 1, np.full(npad - raw, lc, dtype=np.int32)]))
    exp_v = np.sort(np.concatenate([(va + 0.25).astype(np.float32).astype(np.float64), np.zeros(npad - raw)]))
    assert np.array_equal(np.sort(r["col_idx"]), exp_c) and np.array_equal(np.sort(r["vals"]), exp_v)
    # rows: the row pointers up to the last non-empty row reproduce the 1-based row counts
    cnt = np.bincount(rows + 1, minlength=n + 2).astype(np.int64)
    cnt[int(rows[perm[-1]]) + 1] += npad - raw
    lastrow = int(np.nonzero(cnt)[0].max())
    assert np.array_equal(r["row_ptr"][: lastrow + 1], np.concatenate([[0], np.cumsum(cnt)])[: lastrow + 1])


def test_host_sources_under_sanitizers():
    """loader, planner and CSR loop under AddressSanitizer + UBSan over the golden files (make asan-check);
    sanitizers run on the CPU build only -- the GPU pool offers none"""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "cvr_amd", "csrc"), "asan-check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 failure(s)" in r.stdout


def test_ring_kernel_isa_guard():
    """spmv_ilv_kernel and spmv_gang_kernel keep their in-flight loads in registers the compiler does not allocate and wait with counted vmcnt: the build
    checks the compiler's assembly for spills, for compiler-issued vector-memory instructions inside the ring region and for compiler
    use of the ring's registers (make isa-check, tools/isa_check.py; also run by __graft_entry__.build()).  The guard must pass on the
    kernel as it is and must FAIL the build -- not a parity test -- on one compiled with a register cap the compiler cannot keep."""
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "cvr_amd", "csrc"), "isa-check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "isa_check: ok" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    # every instantiation of both ring kernels was looked at (value type x dictionary x 16-bit tags x non-temporal stream loads, each)
    assert r.stdout.count("spmv_ilv_kernel<") == r.stdout.count("spmv_gang_kernel<") == 16 and r.stdout.count("vgpr spills 0") == 32
    env = dict(os.environ, HIPCC_EXTRA="-DCVR_RING_CAP=24")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "cvr_amd", "csrc"), "isa-check"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "isa_check: FAIL" in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("kind", ["pattern symmetric", "real general", "real symmetric", "integer general"])
def test_parallel_loader_equals_pinned_oracle_loader_on_large_files(tmp_path, kind):
    """files large enough to be parsed in several text segments: the product loader (REFCOMPAT) against the oracle's
    sequential restatement of readMatrix, which the golden fixtures pin to the unmodified reference"""
    rng = np.random.default_rng(99)
    field, sym = kind.split()
    n, m = 3000, 40000
    r = rng.integers(1, n + 1, size=m)
    c = rng.integers(1, n + 1, size=m)
    if sym == "symmetric":
        r, c = np.maximum(r, c), np.minimum(r, c)      # lower triangle, with diagonal entries and duplicates
    p = tmp_path / "big.mtx"
    with open(p, "w") as f:
        f.write(f"%%MatrixMarket matrix coordinate {field} {sym}\n% a comment line\n{n} {n} {m}\n")
        for k in range(m):
            if field == "pattern":
                f.write(f"{r[k]} {c[k]}\n")
            elif field == "integer":
                f.write(f"{r[k]} {c[k]} {int(rng.integers(-50, 50))}\n")
            else:
                f.write(f"{r[k]} {c[k]} {rng.normal():.9g}\n")
    assert os.path.getsize(p) > (1 << 16)
    a = cvr_amd.load_mm(str(p), capi.MM_REFCOMPAT)
    b = O.read_matrix(str(p))
    assert (a["ref_nItems"], a["ref_nItemsRaw"], a["ref_numRows"]) == (b["nItems"], b["nItemsRaw"], b["numRows"])
    assert np.array_equal(a["row_ptr"], b["rowptr"].astype(np.int64))
    assert np.array_equal(a["col_idx"], b["cols"])
    assert np.array_equal(a["vals"].view(np.uint64), b["val"].view(np.uint64))


def test_size_limits_are_rejected_with_a_message():
    """the format's limits (bit 31 of a column word is the segment-end flag; x is addressed through a 32-bit buffer
    descriptor) are error codes at cvr_create, before any device work"""
    rp = np.array([0, 1], dtype=np.int64)
    with pytest.raises(cvr_amd.CvrError) as e:
        cvr_amd.CvrMatrix(1, 2**31 - 1, rp, np.zeros(1, dtype=np.int32), np.ones(1))
    assert e.value.code == capi.ERR_INVALID and "2^31" in str(e.value)
    with pytest.raises(cvr_amd.CvrError) as e:
        cvr_amd.CvrMatrix(1, 600_000_000, rp, np.zeros(1, dtype=np.int32), np.ones(1))      # 4.8 GB of fp64 x
    assert e.value.code == capi.ERR_INVALID and "4 GiB" in str(e.value)
    with pytest.raises(cvr_amd.CvrError) as e:
        cvr_amd.CvrMatrix(1, 8, rp, np.zeros(1, dtype=np.int32), np.ones(1), steps_per_chunk=6)
    assert e.value.code in (capi.ERR_INVALID, capi.ERR_NO_DEVICE)


def test_auto_panel_rule_on_the_host():
    """cvr_auto_panels: x small -> 1; x large and banded (lines re-used) -> 1; x large and scattered -> panels: in rounds of
    eight with ~2.6 MB of x each (one panel per XCD at a time), or, with CVR_DEBUG=xcd_panels=0 (every panel over the whole chip), one
    per 1.8 MB of missing x; comm / tuning entry points fail with codes (no device here)"""
    rng = np.random.default_rng(5)
    n = 4_000_000                                                          # x = 32 MB of fp64
    rp = np.arange(n + 1, dtype=np.int64) * 2
    scattered = rng.integers(0, n, 2 * n).astype(np.int32)
    P, miss = capi.auto_panels(n, n, rp, scattered)
    assert miss > 0.6 and P == 8 * int(np.ceil(n * 8 / (8 * 2.6e6)))
    os.environ["CVR_DEBUG"] = "xcd_panels=0"
    try:
        P0, miss0 = capi.auto_panels(n, n, rp, scattered)
    finally:
        del os.environ["CVR_DEBUG"]
    assert miss0 == miss and P0 == int(n * 8 * miss / 1.8e6 + 0.5)
    band = (np.repeat(np.arange(n, dtype=np.int64), 2) + np.tile([0, 3], n)).clip(0, n - 1).astype(np.int32)
    P, miss = capi.auto_panels(n, n, rp, band)
    assert P == 1 and miss < 0.2                                           # first touches only: 2 of 32 gathers per line
    m = 1_000_000                                                          # x = 8 MB: never panelled
    P, miss = capi.auto_panels(m, m, rp[:m + 1], scattered[:2 * m] % m)
    assert P == 1 and miss == 0.0
    # x of 12 .. 24 MB: the rule runs only for matrices beyond the resident layout (here: more rows than 4 chunks per workgroup
    # accumulate in LDS in one pass); one that the resident layout holds stays whole
    m = 2_400_000                                                          # x = 19.2 MB, 2.4 M rows
    P, miss = capi.auto_panels(m, m, rp[:m + 1], scattered[:2 * m] % m)
    assert P == 8 and miss > 0.6
    m = 1_600_000                                                          # x = 12.8 MB, 1.6 M rows: resident (5 x 48)
    P, miss = capi.auto_panels(m, m, rp[:m + 1], scattered[:2 * m] % m)
    assert P == 1 and miss == 0.0
    assert capi.lib().cvr_auto_panels(None, None) == capi.ERR_INVALID


def test_exchange_and_tuning_entry_points_fail_with_codes_without_a_device():
    """cvr_comm_* / cvr_spmv_gather_repeat / cvr_tune_steps: argument checks and the no-device case return codes"""
    L = capi.lib()
    assert L.cvr_comm_unique_id(None) == capi.ERR_INVALID
    h = C.c_void_p()
    ident = C.create_string_buffer(capi.COMM_ID_BYTES)
    assert L.cvr_comm_create(C.byref(h), ident, 0, 0, 0) == capi.ERR_INVALID          # nranks < 1
    assert L.cvr_comm_create(C.byref(h), ident, 2, 2, 0) == capi.ERR_INVALID          # rank out of range
    assert L.cvr_comm_create(None, ident, 1, 0, 0) == capi.ERR_INVALID
    assert L.cvr_comm_destroy(None) == 0
    assert L.cvr_comm_all_gather(None, None, None, 0, 0, None) == capi.ERR_INVALID
    assert L.cvr_spmv_gather_repeat(None, None, None, None, None, 0, 1, 0, None, None) == capi.ERR_INVALID
    assert L.cvr_tune_steps(None, None, None, None, None) == capi.ERR_INVALID
    if cvr_amd.device_count() == 0:
        rp = np.array([0, 1], dtype=np.int64)
        ci = np.zeros(1, dtype=np.int32)
        va = np.ones(1)
        view = capi.CsrView(1, 1, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, 0)
        best = C.c_int32()
        assert L.cvr_tune_steps(C.byref(view), None, C.byref(best), None, None) == capi.ERR_NO_DEVICE
        rc = L.cvr_comm_create(C.byref(h), ident, 1, 0, 0)                             # RCCL may load, but there is no GPU
        assert rc < 0 and cvr_amd.last_error()


def test_header_is_plain_c99_and_links_against_the_library(tmp_path):
    """include/cvr_amd.h is the boundary a C (cgo / JNI / ctypes) binding compiles against: strict C99, and a C program
    linked with libcvr_amd.so runs the host-only entry points"""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "cvr_amd.h"
int main(void)
{
    cvr_options o;
    cvr_default_options(&o);
    cvr_csr_view v = {0};
    long long rp[3] = {0, 1, 2};
    int ci[2] = {0, 1};
    double va[2] = {1.0, 2.0};
    v.nrows = 2; v.ncols = 2; v.row_ptr = (const int64_t *)rp; v.col_idx = ci; v.vals = va;
    double miss = -1;
    int P = cvr_auto_panels(&v, &miss);
    cvr_handle *h = 0;
    int rc = cvr_device_count() > 0 ? 0 : cvr_create(&h, &v, &o);
    printf("%s|%d|%d|%d|%d|%d|%d|%d|%d|%d|%d\n", cvr_version(), P, rc, (int)sizeof(cvr_csr_view), (int)sizeof(cvr_options), (int)sizeof(cvr_info), (int)sizeof(cvr_timing),
           (int)offsetof(cvr_info, preprocess_fused), (int)offsetof(cvr_info, nsegments), (int)offsetof(cvr_options, piece_max), (int)offsetof(cvr_timing, gather_mean_s));
    return 0;
}
''')
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", f"-I{root}/include", str(src), "-o", str(exe),
                    f"-L{root}/cvr_amd", "-lcvr_amd", f"-Wl,-rpath,{root}/cvr_amd", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().split("|")
    assert out[0].startswith("cvr_amd") and out[1] == "1"
    assert int(out[2]) in (0, capi.ERR_NO_DEVICE)
    assert int(out[3]) == C.sizeof(capi.CsrView) and int(out[4]) == C.sizeof(capi.Options)
    assert int(out[5]) == C.sizeof(capi.Info) and int(out[6]) == C.sizeof(capi.Timing)
    assert int(out[7]) == capi.Info.preprocess_fused.offset and int(out[8]) == capi.Info.nsegments.offset          # (the fields a round added last)
    assert int(out[9]) == capi.Options.piece_max.offset and int(out[10]) == capi.Timing.gather_mean_s.offset


def test_loader_fuzz_against_the_pinned_oracle_loader(tmp_path):
    """random small Matrix-Market files (rectangular, duplicates, comments, every field / symmetry the reference accepts):
    cvr_mm_read(REFCOMPAT) == the oracle's restatement of readMatrix, array for array"""
    rng = np.random.default_rng(20261003)
    for case in range(80):
        field = ["pattern", "real", "integer"][case % 3]
        sym = ["general", "symmetric"][(case // 3) % 2]
        nr = int(rng.integers(1, 60))
        nc = nr if sym == "symmetric" else int(rng.integers(1, 60))
        m = int(rng.integers(1, 400))
        r = rng.integers(1, nr + 1, size=m)
        c = rng.integers(1, nc + 1, size=m)
        if sym == "symmetric":
            r, c = np.maximum(r, c), np.minimum(r, c)
        p = tmp_path / f"f{case}.mtx"
        with open(p, "w") as f:
            f.write(f"%%MatrixMarket matrix coordinate {field} {sym}\n")
            for _ in range(int(rng.integers(0, 3))):
                f.write("% comment\n")
            f.write(f"{nr} {nc} {m}\n")
            for k in range(m):
                if field == "pattern":
                    f.write(f"{r[k]} {c[k]}\n")
                elif field == "integer":
                    f.write(f"{r[k]} {c[k]} {int(rng.integers(-9, 10))}\n")
                else:
                    f.write(f"{r[k]}  {c[k]}\t{rng.normal():.7g}\n")
        a = cvr_amd.load_mm(str(p), capi.MM_REFCOMPAT)
        b = O.read_matrix(str(p))
        ctx = dict(case=case, field=field, sym=sym, nr=nr, nc=nc, m=m)
        assert (a["ref_nItems"], a["ref_nItemsRaw"], a["ref_numRows"], a["ref_numCols"]) == (b["nItems"], b["nItemsRaw"], b["numRows"], b["numCols"]), ctx
        assert np.array_equal(a["row_ptr"], b["rowptr"].astype(np.int64)), ctx
        assert np.array_equal(a["col_idx"], b["cols"]), ctx
        assert np.array_equal(a["vals"].view(np.uint64), b["val"].view(np.uint64)), ctx


def test_power_law_stand_ins_have_their_shapes():
    """the two further SuiteSparse-shaped stand-ins (cvr_amd/synth_dev.py), scaled down, on the CPU: com-Orkut's shape is symmetric
    without self-loops, wiki-Talk's has ~94 % empty rows and a few rows that hold a large share; rows sorted, no duplicates"""
    torch = pytest.importorskip("torch")
    from cvr_amd import synth_dev as D
    n, rp, ci, va = D.orkut_like(0.004)
    r = torch.repeat_interleave(torch.arange(n), rp[1:] - rp[:-1])
    assert torch.equal(torch.sort(r * n + ci).values, torch.sort(ci.to(torch.int64) * n + r).values)      # A == A^T
    assert not bool((r == ci).any()) and int(rp[-1]) == len(ci) == len(va)
    key = r * n + ci
    assert bool((key[1:] > key[:-1]).all())
    assert 20 < len(ci) / n < 90
    n, rp, ci, va = D.wikitalk_like(0.05)
    d = rp[1:] - rp[:-1]
    assert 0.92 < float((d == 0).float().mean()) < 0.96
    assert float(torch.sort(d, descending=True).values[:20].sum()) / len(ci) > 0.1
    key = torch.repeat_interleave(torch.arange(n), d) * n + ci
    assert bool((key[1:] > key[:-1]).all())
    assert set(va.unique().tolist()) <= set(float(v) for v in range(13))


def test_bench_cpu_baseline_respects_the_cgroup_quota(monkeypatch, tmp_path):
    """bench.py's CPU baseline never starts more reference threads than the container's CPU quota allows (round 5: the GPU boxes show 256 logical
    CPUs and `cpu.max` = 1600000 100000; 64 threads were throttled in 80 % of the scheduler periods and identical pinned runs differed by a
    factor of three).  _cpu_quota reads cgroup v2 / v1 files; no file or "max": no limit."""
    import builtins
    import importlib
    import io
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    real_open = builtins.open
    files = {}

    def fake_open(path, *a, **k):
        if isinstance(path, str) and path.startswith("/sys/fs/cgroup/"):
            if path in files:
                return io.StringIO(files[path])
            raise FileNotFoundError(path)
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    assert bench._cpu_quota() is None
    files["/sys/fs/cgroup/cpu.max"] = "1600000 100000\n"
    assert bench._cpu_quota() == 16
    files["/sys/fs/cgroup/cpu.max"] = "max 100000\n"
    assert bench._cpu_quota() is None
    del files["/sys/fs/cgroup/cpu.max"]
    files["/sys/fs/cgroup/cpu/cpu.cfs_quota_us"] = "250000\n"
    files["/sys/fs/cgroup/cpu/cpu.cfs_period_us"] = "100000\n"
    assert bench._cpu_quota() == 2
    files["/sys/fs/cgroup/cpu/cpu.cfs_quota_us"] = "-1\n"
    assert bench._cpu_quota() is None
