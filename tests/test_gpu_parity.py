"""GPU (MI355X): the HIP path through the C ABI against the oracle.

  * converter image  == CPU mirror of the format, bit for bit (integer/byte work: desc, target, column words,
    value bits, fix-up list)
  * y                within 1e-12 * sum_j |a_ij x_j| of the CSR oracle (spmv.cpp:1843-1850) for fp64, x == 1 and
                     seeded x; 1e-5 for the fp32 path against the fp64-accumulated CSR oracle
  * golden fixtures  the reference loader's arrays and the reference's CSR y (unmodified reference, run in the
                     build container by oracle/gen_fixtures.py)
  * full size        web-Google-shaped matrix (BASELINE.json configs[1]): oracle comparison + linearity +
                     run-to-run bit equality
"""
import glob
import os

import numpy as np
import pytest

import cases as K
import cvr_amd
from cvr_amd import capi, synth
import oraclelib as O

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz")))
CASES = K.cases()
CASES32 = K.cases(np.float32)
TOL64, TOL32 = 1e-12, 1e-5


def _assert_close(y, yref, absy, tol, ctx):
    bad, worst = O.tol_check(y, yref, absy, tol=tol)
    assert len(bad) == 0, (ctx, "rows", bad[:8], "worst rel", worst)


def test_library_is_the_hip_build_and_sees_the_gpu():
    assert os.path.exists(capi.lib_path())
    assert cvr_amd.device_count() >= 1


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("S", [4, 8, 32])
def test_converter_image_bit_exact_and_y_parity(name, S):
    nrows, ncols, rp, ci, va = CASES[name]
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=S)
    mir = O.Cvr64(nrows, ncols, rp, ci, va, S, use_dict=A.info.value_dict > 0, narrow=A.info.narrow_cols)
    assert (A.info.nchunks, A.info.nshared, A.info.value_dict) == (mir.nchunks, mir.nshared, mir.ndict)
    img = A.export_image()
    assert np.array_equal(img["desc"], mir.desc)
    assert np.array_equal(img["target"], mir.target)
    assert np.array_equal(img["shared"], mir.shared)
    assert np.array_equal(img["image"], mir.image)
    for mode in ("ones", "rand"):
        x = O.x_vec_fast(ncols, mode)
        yref, absy = O.csr_spmv64(rp, ci, va, x)
        y, _ = A.spmv(x)
        _assert_close(y, yref, absy, TOL64, (name, S, mode))
    A.close()


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("S,wpb,win,P,tags", [(8, 1, 0, 3, 0), (4, 8, 64, 2, 0), (32, 4, 1024, 5, 0), (16, 16, 0, 1, 0), (8, 2, 256, 1, 0),
                                              (8, 1, 0, 3, 1), (32, 4, 1024, 5, 1), (64, 14, 512, 7, 1)])
@pytest.mark.parametrize("pmax", [0, 4])
def test_workgroup_window_and_column_phases(name, S, wpb, win, P, tags, pmax):
    """Several chunks per workgroup sharing an LDS window of x, and column phases (a chunk feeds its rows' pieces column
    range by column range; their sums are added up in LDS): the image against the CPU mirror bit for bit, y against the CSR
    oracle, and -- when no row is cut over chunks -- y bitwise equal to the mirror's interpretation of the image (same
    order of additions)."""
    nrows, ncols, rp, ci, va = CASES[name]
    if P > 1 and ncols < 64 * P:
        pytest.skip("too few columns for phases: the library falls back to one phase")
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=S, waves_per_block=wpb, x_window=win, col_phases=P, row_tags16=tags, piece_max=pmax)
    i = A.info
    assert (i.col_phases, i.waves_per_block, i.row_tags16, i.piece_max) == (P, wpb, tags if P > 1 else 0, pmax if P > 1 else 0)
    mir = O.Cvr64(nrows, ncols, rp, ci, va, S, use_dict=i.value_dict > 0, phases=P, max_rows=i.chunk_row_cap, narrow=i.narrow_cols, tag16=tags, piece_max=pmax)
    img = A.export_image()
    assert (i.nchunks, i.nshared) == (mir.nchunks, mir.nshared)
    for key in ("desc", "target", "shared", "image"):
        assert np.array_equal(img[key], getattr(mir, key)), key
    for mode in ("ones", "rand"):
        x = O.x_vec_fast(ncols, mode)
        yref, absy = O.csr_spmv64(rp, ci, va, x)
        y, _ = A.spmv(x)
        _assert_close(y, yref, absy, TOL64, (name, S, wpb, win, P, tags, mode))
        if i.nshared == 0:          # (the fix-up of rows cut over chunks adds the carries as a tree, the mirror one by one)
            assert np.array_equal(y, mir.spmv(x)), (name, S, wpb, win, P, tags, mode)
        y2, _ = A.spmv(x)
        assert np.array_equal(y, y2)
    A.close()


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("S,wpb,tags", [(4, 1, 1), (16, 4, 1), (32, 2, 0), (64, 4, -1)])
def test_interleaved_chunks_image_bit_exact_and_y_parity(name, S, wpb, tags):
    """cvr_options.interleave: a chunk's non-zeros dealt to the lanes in column order, every slot with its row, the rows' sums in LDS
    (cvr_ilv.hip).  The image against the CPU mirror (orc_cvr64_build_ilv) bit for bit, y against the CSR oracle, y bitwise equal to
    the mirror's interpretation of the image when no row is cut over chunks, and bitwise equal from run to run."""
    nrows, ncols, rp, ci, va = CASES[name]
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=S, waves_per_block=wpb, row_tags16=tags, interleave=1, col_panels=1)
    i = A.info
    assert (i.interleave, i.waves_per_block, i.col_panels, i.piece_max) == (1, wpb, 1, 1)
    if tags >= 0:
        assert i.row_tags16 == tags
    mir = O.Cvr64(nrows, ncols, rp, ci, va, S, use_dict=i.value_dict > 0, max_rows=i.chunk_row_cap, tag16=i.row_tags16, interleave=True)
    img = A.export_image()
    assert (i.nchunks, i.nshared, i.value_dict) == (mir.nchunks, mir.nshared, mir.ndict)
    for key in ("desc", "shared", "image"):
        assert np.array_equal(img[key], getattr(mir, key)), key
    for mode in ("ones", "rand"):
        x = O.x_vec_fast(ncols, mode)
        yref, absy = O.csr_spmv64(rp, ci, va, x)
        y, _ = A.spmv(x)
        _assert_close(y, yref, absy, TOL64, (name, S, wpb, tags, mode))
        if i.nshared == 0:
            assert np.array_equal(y, mir.spmv(x)), (name, S, wpb, tags, mode)
        y2, _ = A.spmv(x)
        assert np.array_equal(y, y2)
    A.close()


def _gang_ystage(i, f32):
    """accumulators per chunk of a gang image, from the workgroup's LDS (cvr_spmv.hip: spmv_lds_bytes)"""
    vs = 4 if f32 else 8
    return (i.lds_bytes - 80 - (256 * vs if i.value_dict else 0)) // (i.waves_per_block * vs)


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("S,wpb,tags", [(4, 4, -1), (16, 4, 1), (32, 2, -1), (64, 4, -1), (16, 8, -1)])
def test_gang_chunks_image_bit_exact_and_y_parity(name, S, wpb, tags):
    """cvr_options.gang: the chunks of a workgroup sorted together and walked by its wavefronts in turn, the additions in the list's order by a token in
    LDS (spmv_gang_kernel).  The image, the groups' first columns and the gangs' group counts against the CPU mirror (orc_cvr64_build_gang) bit for bit, y
    against the CSR oracle, y bitwise equal to the mirror's walk of the list when no row is cut over chunks, and bitwise equal from run to run."""
    nrows, ncols, rp, ci, va = CASES[name]
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=S, waves_per_block=wpb, row_tags16=tags, interleave=1, gang=1, col_panels=1)
    i = A.info
    assert (i.interleave, i.gang, i.waves_per_block, i.col_panels) == (1, wpb, wpb, 1)
    ystage = _gang_ystage(i, False)
    mir = O.Cvr64(nrows, ncols, rp, ci, va, S, use_dict=i.value_dict > 0, max_rows=i.chunk_row_cap, tag16=i.row_tags16, gang=wpb, ystage=ystage)
    img = A.export_image()
    assert (i.nchunks, i.nshared, i.value_dict) == (mir.nchunks, mir.nshared, mir.ndict)
    for key in ("desc", "shared", "image", "gbase"):
        assert np.array_equal(img[key], getattr(mir, key)), key
    assert np.array_equal(img["desc2"][:, 0], mir.ggroups) and np.array_equal(img["desc2"][:, 1], mir.nrows_in)
    for mode in ("ones", "rand"):
        x = O.x_vec_fast(ncols, mode)
        yref, absy = O.csr_spmv64(rp, ci, va, x)
        y, _ = A.spmv(x)
        _assert_close(y, yref, absy, TOL64, (name, S, wpb, tags, mode))
        if i.nshared == 0:
            assert np.array_equal(y, mir.spmv(x)), (name, S, wpb, tags, mode)
        for _ in range(3):
            y2, _ = A.spmv(x)
            assert np.array_equal(y, y2)
    A.close()


def test_gang_chunks_fall_back_to_tags_and_fp32_panels():
    """(1) a matrix whose gangs hold few non-zeros over 600 000 columns: a group's 256 sorted columns span more than the 17 bits of an offset, the converter
    says so and cvr_preprocess converts again with 16-bit tags -- same y, the mirror's bits; (2) the automatic layout's configuration scaled down: sixteen
    fp32 / fp64 panels one per XCD, four wavefronts per workgroup, gang chunks by the rule; unsorted rows are fine (the converter sorts)"""
    rng = np.random.default_rng(5)
    nrows, ncols = 4000, 600_000
    deg = rng.integers(0, 6, nrows)
    rp = np.zeros(nrows + 1, dtype=np.int64)
    rp[1:] = np.cumsum(deg)
    ci = np.concatenate([np.sort(rng.choice(ncols, d, replace=False)) for d in deg]).astype(np.int32)
    va = rng.random(len(ci))
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=8, waves_per_block=4, interleave=1, gang=1, col_panels=1)
    i = A.info
    assert (i.gang, i.row_tags16) == (4, 1), (i.gang, i.row_tags16)
    mir = O.Cvr64(nrows, ncols, rp, ci, va, 8, use_dict=i.value_dict > 0, max_rows=i.chunk_row_cap, tag16=1, gang=4, ystage=_gang_ystage(i, False))
    img = A.export_image()
    assert np.array_equal(img["image"], mir.image) and np.array_equal(img["desc"], mir.desc) and not img["gbase"].any()
    x = O.x_vec_fast(ncols, "rand")
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    y, _ = A.spmv(x)
    _assert_close(y, yref, absy, TOL64, "gang, tags after the fall-back")
    if i.nshared == 0:
        assert np.array_equal(y, mir.spmv(x))
    A.close()
    for f32 in (False, True):
        n, nc, rp, ci, va = synth.livejournal_like(scale=0.03)
        if f32:
            va = (rng.random(len(ci)) - 0.5).astype(np.float32)
        for r in rng.integers(0, n, 200):
            a, b = int(rp[r]), int(rp[r + 1])
            p = rng.permutation(b - a)
            ci[a:b], va[a:b] = ci[a:b][p], va[a:b][p]
        A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, col_panels=16, interleave=1)
        i = A.info
        assert (i.interleave, i.gang, i.col_panels, i.waves_per_block, i.spmv_launches) == (1, 4, 16, 4, 1), (i.interleave, i.gang, i.col_panels, i.waves_per_block)
        for mode in ("ones", "rand"):
            x = O.x_vec_fast(nc, mode).astype(va.dtype)
            yref, absy = O.csr_spmv64(rp, ci, va, x)
            y, _ = A.spmv(x)
            _assert_close(y, yref, absy, TOL32 if f32 else TOL64, ("gang panels", f32, mode))
            y2, _ = A.spmv(x)
            assert np.array_equal(y, y2)
        B = cvr_amd.CvrMatrix(n, nc, rp, ci, va, col_panels=16, interleave=1, gang=0)          # the private chunks of round 4: the same sums, bit for bit (both add in column order)
        assert B.info.gang == 0
        yb, _ = B.spmv(x)
        assert np.array_equal(y, yb)
        A.close(); B.close()


@pytest.mark.parametrize("debug", ["ilv_helpers=2,ilv_flip=1,ilv_stream_nt=1,ilv_ahead=8", "ilv_helpers=1,ilv_per_line=1", "combine_mul=8,combine_batch=9", "combine_mul=8,combine_batch=12",
                                   "combine_batch=8", "combine_batch=16", "fuse", "combine_bits=0", "combine_bits=1"])
@pytest.mark.parametrize("gang", [0, 1])
def test_launch_parameter_paths_on_small_matrices(debug, gang, monkeypatch):
    """The launch parameters that the rules switch on for large handles only -- helper wavefronts, the alternating sweep direction, non-temporal stream loads,
    the combine pass's wide workgroups and batches, and the combine pass inside the gang kernel (CVR_DEBUG=fuse: measured, not adopted) -- forced onto a small
    matrix of interleaved panels, with private chunks and with gang chunks: the same y, bit for bit, as without them, several SpMVs in a row (the sweep alternates)."""
    n, nc, rp, ci, va = synth.livejournal_like(scale=0.02)
    x = O.x_vec_fast(nc, "rand")
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, col_panels=16, interleave=1, gang=gang)
    y0, _ = A.spmv(x)
    _assert_close(y0, yref, absy, TOL64, ("plain launch", gang))
    A.close()
    monkeypatch.setenv("CVR_DEBUG", debug)
    B = cvr_amd.CvrMatrix(n, nc, rp, ci, va, col_panels=16, interleave=1, gang=gang)
    assert (B.info.gang > 0) == bool(gang)
    for _ in range(4):
        y, _ = B.spmv(x)
        assert np.array_equal(y, y0), (debug, gang)
    B.close()


@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("kw", [{"col_panels": 8, "steps_per_chunk": 16}, {"col_panels": 16, "interleave": 1, "gang": 1}, {"col_panels": 3, "steps_per_chunk": 8},
                                {"col_panels": 16, "steps_per_chunk": 4}])
def test_combine_bitmap_form_is_the_row_number_form(kw, f32, monkeypatch):
    """The combine pass over a bitmap of the rows that have a partial sum in a panel (combine_bits_kernel: a thread owns four rows and adds their sums in panel
    order in registers) writes the bits of the pass over row numbers (combine_kernel: LDS accumulators, a barrier per panel) -- plain and interleaved panels,
    8 / 16 / an odd number of panels, fp32, rows cut over chunks (short chunks), a row count that is no multiple of the pass's 1 024-row blocks, empty rows."""
    n, nc, rp, ci, va = synth.livejournal_like(scale=0.021)
    va = (va.astype(np.float32) if f32 else va)
    x = O.x_vec_fast(nc, "rand").astype(np.float32 if f32 else np.float64)
    ys = []
    for form in ("0", "1"):
        monkeypatch.setenv("CVR_DEBUG", "combine_bits=" + form)
        A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, **kw)
        assert A.info.col_panels == kw["col_panels"]
        if kw.get("steps_per_chunk") == 4:
            assert A.info.nshared > 0          # (rows cut over chunks: the fix-up launch runs in front of either form)
        y, _ = A.spmv(x)
        y2, _ = A.spmv(x)
        assert np.array_equal(y, y2)
        ys.append(y)
        A.close()
    assert np.array_equal(ys[0].view(np.uint8), ys[1].view(np.uint8)), kw
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    _assert_close(ys[1], yref, absy, TOL32 if f32 else TOL64, ("bitmap combine", kw, f32))


@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("kw", [{"col_panels": 8, "steps_per_chunk": 32}, {"col_panels": 8, "interleave": 1, "gang": 1, "steps_per_chunk": 32, "waves_per_block": 4}])
def test_cut_rows_folded_into_the_bitmap_combine(kw, f32, monkeypatch, capfd):
    """A handful of rows cut over chunks (here three rows with 3 000-4 500 non-zeros inside ONE panel each, in a matrix of 12 per row; chunks of 2 048 slots) on a panelled handle whose
    combine pass runs in its bitmap form: the pass sums those rows' carries itself (CutEntry; no fix-up launch) -- the bits of the fix-up launch in front of the
    pass (CVR_DEBUG=no_cut_fold) and of the row-number form (combine_bits=0)."""
    rng = np.random.default_rng(99)
    n = 40_000
    deg = np.full(n, 12, dtype=np.int64)
    long_rows = {123: (4500, 0), 20_001: (4000, 20_000), 39_990: (3000, 35_000)}          # row: (non-zeros, first column of the panel they lie in; a panel is 5 000 columns wide)
    for r, (d, _) in long_rows.items():
        deg[r] = d
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    ci = np.empty(rp[-1], dtype=np.int32)
    for r in range(n):
        if r in long_rows:
            ci[rp[r]:rp[r + 1]] = long_rows[r][1] + np.sort(rng.choice(5000, size=deg[r], replace=False))
        else:
            ci[rp[r]:rp[r + 1]] = np.sort(rng.choice(n, size=deg[r], replace=False))
    va = (rng.random(rp[-1]) * 2 - 1).astype(np.float32 if f32 else np.float64)
    x = (rng.random(n) * 2 - 1).astype(va.dtype)
    ys, infos = [], []
    for dbg in ("fused_trace", "fused_trace,no_cut_fold", "fused_trace,combine_bits=0"):
        monkeypatch.setenv("CVR_DEBUG", dbg)
        capfd.readouterr()
        A = cvr_amd.CvrMatrix(n, n, rp, ci, va, **kw)
        err = capfd.readouterr().err
        assert A.info.col_panels == 8 and 0 < A.info.nshared <= 8, A.info.nshared
        assert ("3 cut rows folded" in err) == (dbg == "fused_trace") and ("bitmap form" in err) == ("combine_bits=0" not in dbg), err
        y, _ = A.spmv(x)
        y2, _ = A.spmv(x)
        assert np.array_equal(y, y2)
        ys.append(y)
        A.close()
    assert np.array_equal(ys[0].view(np.uint8), ys[1].view(np.uint8)) and np.array_equal(ys[0].view(np.uint8), ys[2].view(np.uint8)), kw
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    _assert_close(ys[0], yref, absy, TOL32 if f32 else TOL64, ("cut rows folded", kw, f32))


def test_panel_question_for_mid_size_matrices_without_a_diagonal(monkeypatch, capfd):
    """x of 8 .. 24 MB with every layout option left to the rules: the layout probe runs in front of the panel rule, and a matrix whose non-zeros are not near the
    diagonal (no use for the resident layout's window) is asked the panel question like one beyond the resident layout -- a wiki-Talk-like matrix of 14 MB of x becomes
    gang panels (60.8 -> 23.9 us: profiles/r06_thin_lists_rule.log); a web-Google-like one of 11 MB keeps the resident layout with its window.  y checked for both."""
    torch = pytest.importorskip("torch")
    from cvr_amd import synth_dev as D
    monkeypatch.setenv("CVR_DEBUG", "fused_trace")
    n, rp_t, ci_t, va_t = D.wikitalk_like(scale=0.75, device="cuda")
    rp, ci, va = rp_t.cpu().numpy(), ci_t.cpu().numpy(), va_t.cpu().numpy()
    del rp_t, ci_t, va_t
    for arrays_on_device in (False, True):
        capfd.readouterr()
        if arrays_on_device:
            t = [torch.from_numpy(a).cuda() for a in (rp, ci, va)]
            A = cvr_amd.CvrMatrix.from_device(n, n, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), is_f32=False, device=0)
        else:
            A = cvr_amd.CvrMatrix(n, n, rp, ci, va)
        err = capfd.readouterr().err
        assert "the panel question is asked" in err, err[-600:]
        assert A.info.col_panels >= 8 and A.info.interleave == 1 and A.info.gang > 0, (A.info.col_panels, A.info.interleave, A.info.gang)
        if not arrays_on_device:
            x = O.x_vec_fast(n, "rand")
            y, _ = A.spmv(x)
            yref, absy = O.csr_spmv64(rp, ci, va, x)
            _assert_close(y, yref, absy, TOL64, "wiki-Talk x 0.75")
        A.close()
    n, nc, rp, ci, va = synth.web_google_like(scale=1.5)
    capfd.readouterr()
    A = cvr_amd.CvrMatrix(n, nc, rp, ci, va)
    err = capfd.readouterr().err
    assert "not asked (resident layout)" in err, err[-600:]
    assert A.info.col_panels == 1 and A.info.x_window > 0, (A.info.col_panels, A.info.x_window)
    x = O.x_vec_fast(nc, "rand")
    y, _ = A.spmv(x)
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    _assert_close(y, yref, absy, TOL64, "web-Google x 1.5")
    A.close()


def test_interleaved_chunk_length_limits():
    """interleave = 1 with the longest chunks the converter sorts in one workgroup (S = 508: 32 pairs per thread; S = 576: 36) converts and
    runs, image and y the mirror's bits; one group more is refused by cvr_create with CVR_ERR_INVALID and a message, before any planning
    (it used to surface as hipErrorInvalidValue from the converter).  A matrix large enough for chunks of 36 864 slots."""
    nrows, ncols, rp, ci, va = synth.web_google_like(scale=0.06, seed=5)
    x = O.x_vec_fast(ncols, "rand")
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    for S in (508, 576):
        A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=S, waves_per_block=4, interleave=1, col_panels=1)
        assert A.info.interleave == 1 and A.info.steps_per_chunk == S
        y, _ = A.spmv(x)
        _assert_close(y, yref, absy, TOL64, ("interleaved", S))
        mir = O.Cvr64(nrows, ncols, rp, ci, va, S, use_dict=A.info.value_dict > 0, max_rows=A.info.chunk_row_cap, tag16=A.info.row_tags16, interleave=True)
        assert (A.info.nchunks, A.info.nshared) == (mir.nchunks, mir.nshared)
        img = A.export_image()
        assert np.array_equal(img["image"], mir.image) and np.array_equal(img["desc"], mir.desc)
        if A.info.nshared == 0:
            assert np.array_equal(y, mir.spmv(x))
        A.close()
    with pytest.raises(capi.CvrError) as e:
        cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=580, waves_per_block=4, interleave=1, col_panels=1)
    assert e.value.code == capi.ERR_INVALID and "interleave" in str(e.value)


@pytest.mark.parametrize("f32", [False, True])
def test_interleaved_column_panels_one_per_xcd(f32):
    """the configuration the automatic rule picks for the soc-LiveJournal1 shape, scaled down: column panels that run one per XCD, every
    panel's chunks interleaved, four chunks per workgroup; unsorted rows are fine (the converter sorts)"""
    n, nc, rp, ci, va = synth.livejournal_like(scale=0.03)
    rng = np.random.default_rng(11)
    if f32:
        va = (rng.random(len(ci)) - 0.5).astype(np.float32)
    for r in rng.integers(0, n, 200):              # a few rows with their entries out of order
        a, b = int(rp[r]), int(rp[r + 1])
        p = rng.permutation(b - a)
        ci[a:b], va[a:b] = ci[a:b][p], va[a:b][p]
    A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, col_panels=16, interleave=1)
    i = A.info
    assert (i.interleave, i.col_panels, i.waves_per_block, i.spmv_launches) == (1, 16, 4, 1), (i.interleave, i.col_panels, i.waves_per_block, i.spmv_launches)
    for mode in ("ones", "rand"):
        x = O.x_vec_fast(nc, mode).astype(va.dtype)
        yref, absy = O.csr_spmv64(rp, ci, va, x)
        y, _ = A.spmv(x)
        _assert_close(y, yref, absy, TOL32 if f32 else TOL64, ("interleaved panels", f32, mode))
        y2, _ = A.spmv(x)
        assert np.array_equal(y, y2)
    A.close()


def test_panel_rule_counts_lines_in_lds_or_with_atomics(monkeypatch, capfd):
    """the panel rule's L2-miss estimate on the device counts the touched lines of x per window of rows in LDS, one workgroup per range of lines
    (cvr_split.hip: est_count_lds_kernel), or with global atomics (CVR_DEBUG=est_atomics, and matrices of more than 32 ranges): the same counters,
    hence the same miss share to the last digit and the same panels -- on a scattered matrix (panels) and on a banded one (none)"""
    import torch
    n = 3_200_000
    g = torch.Generator(device="cuda")
    g.manual_seed(17)
    rows = torch.arange(n, device="cuda", dtype=torch.int64).repeat_interleave(5)
    for kind in ("scattered", "banded"):
        if kind == "scattered":
            cols = (n * torch.rand(len(rows), generator=g, device="cuda", dtype=torch.float64)).to(torch.int64).clamp_(0, n - 1)
        else:
            cols = (rows + torch.randint(-40, 41, (len(rows),), generator=g, device="cuda")).clamp_(0, n - 1)
        key = torch.unique(rows * n + cols)
        r = torch.div(key, n, rounding_mode="floor")
        ci = (key - r * n).to(torch.int32).cpu().numpy()
        rp = np.concatenate([[0], np.cumsum(torch.bincount(r, minlength=n).cpu().numpy())]).astype(np.int64)
        va = (np.arange(len(ci)) % 13).astype(np.float64)
        got = []
        for knob in ("fused_trace", "fused_trace,est_atomics"):
            monkeypatch.setenv("CVR_DEBUG", knob + ",panel_rule_trace")
            A = cvr_amd.CvrMatrix(n, n, rp, ci, va)
            err = capfd.readouterr().err
            miss = [ln for ln in err.splitlines() if "L2 miss share" in ln]
            got.append((A.info.col_panels, A.info.interleave, miss[-1] if miss else None))
            A.close()
        assert got[0] == got[1] and got[0][2] is not None, got
        assert (got[0][0] > 1) == (kind == "scattered"), got
    monkeypatch.delenv("CVR_DEBUG")


def test_interleaved_panels_planned_again_when_the_estimate_is_low(monkeypatch, capfd):
    """the chunk length of one-per-XCD panels is chosen for whole generations of workgroups from an ESTIMATE of the chunk counts; when the
    plan has a generation more, cvr_create plans once more with longer chunks (here the estimate is scaled down to 70 % to force that)"""
    n, nc, rp, ci, va = synth.livejournal_like(scale=0.2)
    x = O.x_vec_fast(nc, "rand")
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    got = []
    for pct in ("100", "70"):
        monkeypatch.setenv("CVR_DEBUG", f"fused_trace,ilv_est_percent={pct}")
        A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, col_panels=16, interleave=1)
        err = capfd.readouterr().err
        i = A.info
        got.append((i.steps_per_chunk, i.nchunks, "planned again" in err))
        assert i.interleave == 1 and "whole generations" in err
        y, _ = A.spmv(x)
        _assert_close(y, yref, absy, TOL64, ("planned again", pct))
        A.close()
    monkeypatch.delenv("CVR_DEBUG")
    assert got[1][2], got                                   # (the low estimate was corrected by a second plan ...)
    assert got[1][0] > 16 and got[1][1] > 0, got


@pytest.mark.parametrize("ncols", [129, 262, 1000, 4097])
def test_interleaved_panels_of_unequal_width(ncols):
    """the last column panel is narrower than the others (its column index needs fewer bits): the panels share a launch, so they share
    the width of the column and row fields (a fuzz case found them decoded with the first panel's width)"""
    rng = np.random.default_rng(5)
    nrows, nc, rp, ci, va = K.csr_from_lengths(np.full(3841, 46, dtype=np.int64), ncols, rng, sort=False)
    for P in (8, 5, 16):
        A = cvr_amd.CvrMatrix(nrows, nc, rp, ci, va, steps_per_chunk=128, col_panels=P, interleave=1)
        assert (A.info.interleave, A.info.col_panels) == (1, P)
        x = O.x_vec_fast(nc, "rand")
        yref, absy = O.csr_spmv64(rp, ci, va, x)
        y, _ = A.spmv(x)
        _assert_close(y, yref, absy, TOL64, ("panels of unequal width", ncols, P))
        A.close()


def test_column_phases_need_sorted_rows_and_fit_the_row_field():
    rng = np.random.default_rng(5)
    nrows, ncols, rp, ci, va = K.csr_from_lengths([7] * 300, 5000, rng, sort=False)
    with pytest.raises(cvr_amd.CvrError):
        cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=8, col_phases=4)
    # fp32 with phases, window and several waves: full pipeline against the fp64-accumulated oracle
    n, nc, rp, ci, va = synth.web_google_like(scale=0.03)
    va = (np.random.default_rng(2).random(len(ci)) - 0.5).astype(np.float32)
    A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, steps_per_chunk=12, waves_per_block=8, x_window=2048, col_phases=6)
    x = synth.x_rand(nc, np.float32)
    y, _ = A.spmv(x)
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    _assert_close(y, yref, absy, TOL32, "fp32 phases")
    mir = O.Cvr64(n, nc, rp, ci, va, 12, phases=6, max_rows=A.info.chunk_row_cap)
    assert np.array_equal(A.export_image()["image"], mir.image)
    assert np.array_equal(y, mir.spmv(x))
    A.close()


@pytest.mark.parametrize("name", sorted(CASES32))
def test_fp32_path(name):
    nrows, ncols, rp, ci, va = CASES32[name]
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=8)
    mir = O.Cvr64(nrows, ncols, rp, ci, va, 8, use_dict=A.info.value_dict > 0, narrow=A.info.narrow_cols)
    assert np.array_equal(A.export_image()["image"], mir.image)
    x = O.x_vec_fast(ncols, "rand").astype(np.float32)
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    y, _ = A.spmv(x)
    _assert_close(y, yref, absy + 1e-30, TOL32, name)
    A.close()


@pytest.mark.parametrize("thr", [1, 16, 100000])
@pytest.mark.parametrize("swz", [0, 1, 3, 4, 6])
def test_split_threshold_and_launch_options(thr, swz):
    for name, S in (("power_law_3000", 8), ("two_giants", 16), ("dense_row_plus_singletons", 4)):
        nrows, ncols, rp, ci, va = CASES[name]
        A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=S, split_threshold=thr, xcd_swizzle=swz)
        mir = O.Cvr64(nrows, ncols, rp, ci, va, S, thr, use_dict=A.info.value_dict > 0, narrow=A.info.narrow_cols)
        img = A.export_image()
        assert np.array_equal(img["image"], mir.image) and np.array_equal(img["shared"], mir.shared)
        x = O.x_vec_fast(ncols, "rand")
        yref, absy = O.csr_spmv64(rp, ci, va, x)
        y, _ = A.spmv(x)
        _assert_close(y, yref, absy, TOL64, (name, thr))
        A.close()


@pytest.mark.parametrize("name", NAMES)
def test_golden_reference_fixtures(name):
    """reference loader arrays (taken literally: 1-based, Q1/Q9) -> GPU y == the reference's CSR y"""
    z = np.load(os.path.join(GOLD, name + ".npz"))
    m = cvr_amd.load_mm(os.path.join(GOLD, "mtx", name + ".mtx"), capi.MM_REFCOMPAT)
    numRows = int(z["dims"][1])
    for S in (4, 16):
        A = cvr_amd.CvrMatrix(m["nrows"], m["ncols"], m["row_ptr"], m["col_idx"], m["vals"], steps_per_chunk=S)
        for mode in ("ones", "rand"):
            x = np.zeros(m["ncols"])
            xs = z[f"x_{mode}"]
            x[: min(len(xs), m["ncols"])] = xs[: m["ncols"]]
            y, _ = A.spmv(x)
            _, absy = O.csr_spmv64(m["row_ptr"], m["col_idx"], m["vals"], x)
            _assert_close(y[:numRows], z[f"y_csr_{mode}"], absy[:numRows], TOL64, (name, S, mode))
            assert cvr_amd.verdict(y, np.concatenate([z[f"y_csr_{mode}"], y[numRows:]]), numRows) == 0
        A.close()


def test_full_size_mtx_file_end_to_end(tmp_path):
    """run_sample.sh:5-10 at the size BASELINE.json quotes: the web-Google-shaped matrix as a `pattern` Matrix-Market file -> the product's
    REFCOMPAT loader == the pinned oracle loader bit for bit (values idx % 13 in file order, padded 1-based arrays) -> HIP SpMV == the
    oracle's CSR loop on those arrays -> ./spmv.cvr on the same file prints the reference's lines with 0 wrong rows"""
    import json
    import subprocess
    n, nc, rp, ci, _ = synth.web_google_like()
    mtx = str(tmp_path / "web-Google-shaped.mtx")
    O.write_mtx_pattern(mtx, n, nc, rp, ci)
    ref = O.read_matrix(mtx)
    m = cvr_amd.load_mm(mtx, capi.MM_REFCOMPAT)
    assert (m["ref_nItems"], m["ref_nItemsRaw"], m["ref_numRows"]) == (ref["nItems"], ref["nItemsRaw"], ref["numRows"]) and ref["nItemsRaw"] == len(ci)
    assert np.array_equal(m["row_ptr"], ref["rowptr"].astype(np.int64)) and np.array_equal(m["col_idx"], ref["cols"])
    assert np.array_equal(m["vals"].view(np.uint64), ref["val"].view(np.uint64))
    A = cvr_amd.CvrMatrix(m["nrows"], m["ncols"], m["row_ptr"], m["col_idx"], m["vals"])
    assert A.info.value_dict == 13 and A.info.col_phases > 1          # (the loader's 13 values; the resident layout)
    x = O.x_vec_fast(m["ncols"], "rand")
    yref, absy = O.csr_spmv64(m["row_ptr"], m["col_idx"], m["vals"], x)
    y, _ = A.spmv(x)
    _assert_close(y, yref, absy, TOL64, "full-size .mtx")
    A.close()
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spmv.cvr")
    r = subprocess.run([exe, mtx, "68", "100"], capture_output=True, text=True, timeout=600, env=dict(os.environ, CVR_X="rand"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Very Good! Your result is correct" in r.stdout
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"backend"')][0])
    assert line["wrong"] == 0 and line["nnz"] >= len(ci), line


def test_call_order_errors():
    nrows, ncols, rp, ci, va = CASES["uniform_2000"]
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va)
    with pytest.raises(ValueError):
        A.spmv(np.ones(3))
    y1, t = A.spmv(np.ones(ncols), iters=3)
    assert t.iters == 3 and t.mean_s > 0 and t.min_s <= t.mean_s <= t.max_s
    A.close()


@pytest.fixture(scope="module")
def web_google():
    nrows, ncols, rp, ci, va = synth.web_google_like()
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=32)
    yield nrows, ncols, rp, ci, va, A
    A.close()


def test_full_size_web_google_parity(web_google):
    nrows, ncols, rp, ci, va, A = web_google
    assert A.info.nshared == 0              # S = 32: no row of web-Google's shape (max 456 nnz) exceeds the threshold 512
    for mode in ("ones", "rand"):
        x = O.x_vec_fast(ncols, mode)
        yref, absy = O.csr_spmv64(rp, ci, va, x)
        y, _ = A.spmv(x, iters=2)
        _assert_close(y, yref, absy, TOL64, mode)
        assert np.all(y[np.diff(rp) == 0] == 0)         # rows without non-zeros are written as +0


# The layout cvr_create's automatic rule gives the full-size web-Google shape on an MI355X: what bench.py times (its JSON line
# repeats these numbers under config).  Change it together with the rule (cvr_layout.hip, choose_layout).
TIMED_LAYOUT = dict(steps_per_chunk=48, waves_per_block=7, x_window=12288, col_phases=16, value_dict=13)


@pytest.mark.parametrize("value_dict", [-1, 0])
def test_timed_configuration_full_size_strict(value_dict):
    """The configuration bench.py TIMES -- the web-Google shape with DEFAULT options (automatic layout: resident workgroups, LDS
    window of x, column phases, value dictionary) and the same with full fp64 values in the stream -- against the pinned CSR
    oracle at 1e-12 * sum |a x| per row (spmv.cpp:1843-1850), the converter image against the CPU mirror bit for bit, and
    bitwise reproducibility.  (The other full-size tests pass an explicit S, which switches the automatic layout off.)"""
    nrows, ncols, rp, ci, va = synth.web_google_like()
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, value_dict=value_dict)
    i = A.info
    got = dict(steps_per_chunk=i.steps_per_chunk, waves_per_block=i.waves_per_block, x_window=i.x_window, col_phases=i.col_phases,
               value_dict=i.value_dict)
    assert got == dict(TIMED_LAYOUT, value_dict=13 if value_dict else 0), got
    assert i.preprocess_fused == 1                     # analysis, plan, segment table and conversion as one submission (cvr_fused.hip)
    mir = O.Cvr64(nrows, ncols, rp, ci, va, i.steps_per_chunk, use_dict=i.value_dict > 0, phases=i.col_phases, max_rows=i.chunk_row_cap,
                  tag16=i.row_tags16, piece_max=i.piece_max)
    img = A.export_image()
    assert (i.nchunks, i.nshared) == (mir.nchunks, mir.nshared)
    for key in ("desc", "target", "shared", "image"):
        assert np.array_equal(img[key], getattr(mir, key)), key
    for mode in ("ones", "rand"):
        x = O.x_vec_fast(ncols, mode)
        yref, absy = O.csr_spmv64(rp, ci, va, x)
        y, _ = A.spmv(x, iters=2)
        _assert_close(y, yref, absy, TOL64, (value_dict, mode))
        y2, _ = A.spmv(x)
        assert np.array_equal(y.view(np.uint64), y2.view(np.uint64))
        assert cvr_amd.verdict(y, yref, nrows) == 0                       # the reference's own criterion (spmv.cpp:1916-1938)
        if i.nshared == 0:
            assert np.array_equal(y, mir.spmv(x)), (value_dict, mode)      # same order of additions as the mirror
    A.close()


@pytest.mark.parametrize("kind", ["pattern", "values_fp32", "long_rows"])
def test_fused_preprocessing_same_image(kind, monkeypatch):
    """cvr_create's one-submission preprocessing (cvr_fused.hip: planner -> per-chunk tables on the device -> segment table ->
    conversion without the host in between) gives the image, the descriptors and the y bits of the staged path (CVR_DEBUG=no_fused);
    with rows cut over chunks (their fix-up list comes from the device plan), fp32 values and no dictionary as well."""
    nrows, ncols, rp, ci, va = synth.web_google_like(1.0 if kind == "values_fp32" else 0.5)      # (x of more than 2.5 MB: column phases)
    if kind == "values_fp32":
        va = np.random.default_rng(5).standard_normal(len(ci)).astype(np.float32)
    if kind == "long_rows":          # three rows far beyond a chunk: cut over several chunks
        import scipy.sparse as sp
        rng = np.random.default_rng(11)
        M = sp.csr_matrix((va, ci, rp), shape=(nrows, ncols))
        rows = np.repeat(np.array([7, nrows // 2, nrows - 3]), 9000)
        cols = np.concatenate([rng.choice(ncols, 9000, replace=False) for _ in range(3)])
        M = (M + sp.csr_matrix((rng.standard_normal(27000), (rows, cols)), shape=(nrows, ncols))).tocsr()
        M.sort_indices()
        rp, ci, va = M.indptr.astype(np.int64), M.indices.astype(np.int32), M.data.astype(np.float64)
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va)
    monkeypatch.setenv("CVR_DEBUG", "no_fused")
    B = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va)
    monkeypatch.delenv("CVR_DEBUG")
    ia, ib = A.info, B.info
    assert ia.preprocess_fused == 1 and ib.preprocess_fused == 0
    for f in ("steps_per_chunk", "waves_per_block", "x_window", "col_phases", "value_dict", "nchunks", "nshared", "nsegments", "row_tags16", "piece_max",
              "chunk_row_cap", "lds_bytes", "image_bytes", "nslots"):
        assert getattr(ia, f) == getattr(ib, f), f
    if kind == "long_rows":
        assert ia.nshared > 0
    ea, eb = A.export_image(), B.export_image()
    for key in ("desc", "target", "shared", "image"):
        assert np.array_equal(ea[key], eb[key]), key
    x = O.x_vec_fast(ncols, "rand").astype(va.dtype)
    ya, _ = A.spmv(x)
    yb, _ = B.spmv(x)
    assert np.array_equal(ya.view(np.uint8), yb.view(np.uint8))
    yref, absy = O.csr_spmv64(rp, ci, va.astype(np.float64), x.astype(np.float64))
    _assert_close(ya.astype(np.float64), yref, absy, TOL64 if va.dtype == np.float64 else TOL32, kind)
    A.close(); B.close()


@pytest.mark.parametrize("kind", ["pattern", "values_fp32"])
def test_lds_staged_converter_same_image(kind, monkeypatch):
    """CVR_DEBUG=convert_lds: the converter that copies a chunk's columns and values (as dictionary codes) into LDS first and passes the feed
    table through a ring there (convert_lds_kernel; measured and not the default: DESIGN.md section 5.7) writes the default converter's image."""
    nrows, ncols, rp, ci, va = synth.web_google_like(1.0 if kind == "values_fp32" else 0.5)
    if kind == "values_fp32":
        va = np.random.default_rng(5).standard_normal(len(ci)).astype(np.float32)
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va)
    monkeypatch.setenv("CVR_DEBUG", "convert_lds")
    B = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va)
    monkeypatch.delenv("CVR_DEBUG")
    assert A.info.col_phases > 1 and A.info.nchunks == B.info.nchunks
    ea, eb = A.export_image(), B.export_image()
    for key in ("desc", "target", "shared", "image"):
        assert np.array_equal(ea[key], eb[key]), key
    A.close(); B.close()


@pytest.mark.parametrize("kind", ["banded", "unsorted_rows", "tiny_rows"])
def test_fused_preprocessing_falls_back(kind):
    """The one-submission path is an attempt: a probe that does not confirm the resident layout (a band: everything near the diagonal;
    rows with unsorted columns) or a plan with more chunks than workgroup slots (many rows of one non-zero: the cap on a chunk's rows
    ends the chunks, whatever their length -- such a matrix gets the plain layout, not ever longer chunks) leave the staged path to do
    the work; y is checked against the CSR oracle either way."""
    if kind == "banded":
        nrows, ncols, rp, ci, va = synth.banded_sym(400000, 13)
    elif kind == "unsorted_rows":
        nrows, ncols, rp, ci, va = synth.web_google_like(0.5)
        ci = ci.copy(); va = va.copy()
        for r in np.flatnonzero(np.diff(rp) >= 2)[::3]:          # every third row of two or more: columns descending
            a, z = rp[r], rp[r + 1]
            ci[a:z] = ci[a:z][::-1]; va[a:z] = va[a:z][::-1]
    else:
        nrows = ncols = 2_400_000
        rng = np.random.default_rng(3)
        rp = np.arange(nrows + 1, dtype=np.int64)
        near = rng.random(nrows) < 0.5
        ci = np.where(near, np.clip(np.arange(nrows) + rng.integers(-200, 200, nrows), 0, nrows - 1), rng.integers(0, nrows, nrows)).astype(np.int32)
        va = rng.standard_normal(nrows)
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va)
    i = A.info
    assert i.preprocess_fused == 0
    if kind == "tiny_rows":
        assert i.col_phases == 1 and i.waves_per_block == 1 and i.steps_per_chunk <= 64 and i.image_bytes < 100e6, (i.steps_per_chunk, i.image_bytes)
    x = O.x_vec_fast(ncols, "rand")
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    y, _ = A.spmv(x)
    _assert_close(y, yref, absy, TOL64, kind)
    A.close()


@pytest.mark.parametrize("arrays", ["host", "device"])
def test_matrices_beyond_the_resident_layout_get_panels_from_12_mb_of_x(arrays):
    """x of 12 .. 24 MB: a matrix with more rows than the resident layout's chunks accumulate in one pass (web-Google shape x 2.4: 2.2 M rows,
    x = 17.6 MB) runs as eight column panels, one per XCD (76 against 97 us as one plain image: profiles/r03_mid_size_panels.log); one that
    the resident layout holds (x 1.7: x = 12.5 MB) stays a single image with column phases.  From host arrays and from device arrays;
    y against the CSR oracle."""
    import torch
    for scale, panels in ((2.4, 8), (1.7, 1)):
        nrows, ncols, rp, ci, va = synth.web_google_like(scale)
        assert 12e6 <= ncols * 8 < 24e6
        if arrays == "device":
            trp, tci, tva = (torch.from_numpy(a).cuda() for a in (rp, ci, va))
            A = cvr_amd.CvrMatrix.from_device(nrows, ncols, trp.data_ptr(), tci.data_ptr(), tva.data_ptr(), is_f32=False)
        else:
            A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va)
        i = A.info
        assert i.col_panels == panels, (scale, i.col_panels)
        assert (i.col_phases > 1 and i.waves_per_block >= 4) if panels == 1 else i.col_phases == 1
        x = O.x_vec_fast(ncols, "rand")
        yref, absy = O.csr_spmv64(rp, ci, va, x)
        y, _ = A.spmv(x)
        _assert_close(y, yref, absy, TOL64, f"scale {scale}")
        A.close()


@pytest.mark.parametrize("kind", ["device_arrays", "rectangular", "row_shard", "reconvert"])
def test_fused_preprocessing_other_inputs(kind, monkeypatch):
    """The one-submission path with CSR arrays that are already on the device, a rectangular matrix (more columns than rows), a row
    shard whose row_ptr does not start at 0 (the view a caller hands over for one GPU's rows of a larger matrix), and a second
    cvr_preprocess on a handle that kept its CSR (the staged converter over the tables the fused path left): image and y equal the
    staged path's / the CSR oracle's."""
    nrows, ncols, rp, ci, va = synth.web_google_like(0.5)
    if kind == "rectangular":
        ncols = ncols + 300000
        ci = ci.copy()
        far = np.flatnonzero(np.arange(len(ci)) % 7 == 0)
        ci[far] = (ci[far].astype(np.int64) * 5 % ncols).astype(np.int32)
        import scipy.sparse as sp
        M = sp.csr_matrix((va, ci, rp), shape=(nrows, ncols)); M.sum_duplicates(); M.sort_indices()
        rp, ci, va = M.indptr.astype(np.int64), M.indices.astype(np.int32), M.data.astype(np.float64)
    x = O.x_vec_fast(ncols, "rand")
    if kind == "row_shard":
        r0 = nrows // 5
        view_rp = rp[r0:].copy()                                  # starts at rp[r0] > 0: col_idx / vals are indexed from 0
        lrows = nrows - r0
        A = cvr_amd.CvrMatrix(lrows, ncols, view_rp, ci, va)
        monkeypatch.setenv("CVR_DEBUG", "no_fused")
        B = cvr_amd.CvrMatrix(lrows, ncols, view_rp, ci, va)
        monkeypatch.delenv("CVR_DEBUG")
        yref, absy = O.csr_spmv64(view_rp - view_rp[0], ci[view_rp[0]:], va[view_rp[0]:], x)
    elif kind == "device_arrays":
        import torch
        keep = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (rp, ci, va)]
        torch.cuda.synchronize()
        A = cvr_amd.CvrMatrix.from_device(nrows, ncols, keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr())
        monkeypatch.setenv("CVR_DEBUG", "no_fused")
        B = cvr_amd.CvrMatrix.from_device(nrows, ncols, keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr())
        monkeypatch.delenv("CVR_DEBUG")
        yref, absy = O.csr_spmv64(rp, ci, va, x)
    else:
        A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, keep_csr=(kind == "reconvert"))
        monkeypatch.setenv("CVR_DEBUG", "no_fused")
        B = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va)
        monkeypatch.delenv("CVR_DEBUG")
        yref, absy = O.csr_spmv64(rp, ci, va, x)
    assert A.info.preprocess_fused == 1 and B.info.preprocess_fused == 0
    if kind == "reconvert":
        sec = capi.C.c_double()
        assert capi.lib().cvr_preprocess(A._h, 0, capi.C.byref(sec)) == 0 and sec.value > 0
    ea, eb = A.export_image(), B.export_image()
    for key in ("desc", "target", "shared", "image"):
        assert np.array_equal(ea[key], eb[key]), key
    ya, _ = A.spmv(x)
    yb, _ = B.spmv(x)
    assert np.array_equal(ya.view(np.uint64), yb.view(np.uint64))
    _assert_close(ya, yref, absy, TOL64, kind)
    A.close(); B.close()


def test_full_size_properties(web_google):
    nrows, ncols, rp, ci, va, A = web_google
    x1 = O.x_vec_fast(ncols, "rand")
    x2 = np.cos(np.arange(ncols) * 0.37)
    y1, _ = A.spmv(x1)
    y1b, _ = A.spmv(x1)
    assert np.array_equal(y1.view(np.uint64), y1b.view(np.uint64))     # bitwise reproducible (no atomics on y)
    y2, _ = A.spmv(x2)
    y12, _ = A.spmv(2.0 * x1 - 3.0 * x2)
    _, a1 = O.csr_spmv64(rp, ci, va, np.abs(x1))
    _, a2 = O.csr_spmv64(rp, ci, va, np.abs(x2))
    err = np.abs(y12 - (2.0 * y1 - 3.0 * y2))
    assert np.all(err <= 1e-12 * (2 * a1 + 3 * a2) + 1e-300)          # linearity
    # checksum of checksums: 1^T (A x) == (A^T 1)^T x
    colsum = np.bincount(ci, weights=va, minlength=ncols)
    assert abs(y1.sum() - colsum @ x1) <= 1e-9 * (np.abs(colsum) @ np.abs(x1))


def test_full_size_chunk_lengths_agree(web_google):
    """results are invariant to the chunk length within the tolerance (property 2)"""
    nrows, ncols, rp, ci, va, A = web_google
    x = O.x_vec_fast(ncols, "rand")
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    for S in (8, 64, 256):
        B = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=S)
        y, _ = B.spmv(x)
        _assert_close(y, yref, absy, TOL64, S)
        B.close()


def test_livejournal_shape_scaled_with_cut_rows():
    """soc-LiveJournal1's shape at 1/8 scale: rows longer than the split threshold exercise the carry fix-up"""
    nrows, ncols, rp, ci, va = synth.livejournal_like(scale=0.125)
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=16, split_threshold=64)
    assert A.info.nshared > 0
    x = O.x_vec_fast(ncols, "rand")
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    y, _ = A.spmv(x)
    _assert_close(y, yref, absy, TOL64, "lj")
    A.close()


@pytest.mark.parametrize("win", [0, 64, 1000, 8192, 16384])
def test_lds_window_is_bitwise_neutral(win):
    """the LDS window of x only changes where a value is read from: y must be bit-identical to the windowless kernel"""
    for name, S in (("power_law_3000", 8), ("uniform_2000", 16), ("leading_trailing_empty", 4), ("single_entry", 4)):
        nrows, ncols, rp, ci, va = CASES[name]
        x = O.x_vec_fast(ncols, "rand")
        A0 = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=S, x_window=0)
        A1 = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=S, x_window=win)
        y0, _ = A0.spmv(x)
        y1, _ = A1.spmv(x)
        assert np.array_equal(y0.view(np.uint64), y1.view(np.uint64)), (name, win)
        A0.close()
        A1.close()
    nrows, ncols, rp, ci, va = CASES32["power_law_3000"]
    x = O.x_vec_fast(ncols, "rand").astype(np.float32)
    y0, _ = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=8, x_window=0).spmv(x)
    y1, _ = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=8, x_window=win).spmv(x)
    assert np.array_equal(y0.view(np.uint32), y1.view(np.uint32))


def test_cli_prints_the_reference_lines():
    """./spmv.cvr [mtx] [nThreads] [nIters]: argv contract (spmv.cpp:1693, 1703, 1771) and the four greppable lines
    (spmv.cpp:1009, 1662, 1664, 1932; README.md:47-49) on fixtures the unmodified reference was run on"""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "spmv.cvr")
    assert os.path.exists(exe), "build the host program: make -C cvr_amd/csrc"
    for name, xmode in (("pl2000_pattern", "ones"), ("sym250_real", "rand"), ("skew12", "ones")):
        mtx = os.path.join(GOLD, "mtx", name + ".mtx")
        r = subprocess.run([exe, mtx, "2", "7"], capture_output=True, text=True, timeout=120, env=dict(os.environ, CVR_X=xmode))
        assert r.returncode == 0, r.stderr
        out = r.stdout
        f = re.escape(mtx)
        assert re.search(r"^The Pre-processing\(CSR->CVR\)   Time of CVR   is \S+ seconds\.   \[file: " + f + r"\] \[threads: 2\]$", out, re.M)
        assert re.search(r"^The SpMV Execution Time of CVR    is \S+ seconds\.   \[file: " + f + r"\] \[threads: 2\]$", out, re.M)
        assert re.search(r"^         The Throughput of CVR    is \S+ GFlops\.    \[file: " + f + r"\] \[threads: 2\]$", out, re.M)
        assert "     Very Good! Your result is correct  " in out
        assert '"wrong":0' in out
    # the sharded path on one GPU: three row shards, x replicated, slices gathered (D2D copies stand in for RCCL)
    mtx = os.path.join(GOLD, "mtx", "pl2000_pattern.mtx")
    r = subprocess.run([exe, mtx, "2", "5"], capture_output=True, text=True, timeout=120, env=dict(os.environ, CVR_DEVICES="0,0,0", CVR_X="rand"))
    assert r.returncode == 0, r.stderr
    assert "Very Good! Your result is correct" in r.stdout and '"gpus":3' in r.stdout and '"wrong":0' in r.stdout
    # CVR_POWER: the native iterative caller from the host program (square matrix, one GPU)
    import json
    r = subprocess.run([exe, mtx, "2", "3"], capture_output=True, text=True, timeout=120, env=dict(os.environ, CVR_POWER="25", CVR_MM="strict"))
    assert r.returncode == 0, r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{"power_iterations"')]
    assert len(line) == 1
    pw = json.loads(line[0])
    assert pw["power_iterations"] == 25 and pw["rayleigh_quotient"] > 0 and pw["seconds_per_iteration"] > 0
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "usage" in r.stderr
    r = subprocess.run([exe, "/nonexistent.mtx", "1", "1"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1                                   # loader errors exit 1 as the reference (spmv.cpp:325)


def test_multi_device_handle_one_call_all_gpus():
    """cvr_create_multi / cvr_spmv_multi: the library owns the row partition, the shard handles, the replicated x and the gather of y
    (the reference's one call drives all its threads, spmv.cpp:1857, 1882).  On a one-GPU box the device is listed three times:
    same sharding and gather layout, device-to-device copies in place of RCCL."""
    nrows, ncols, rp, ci, va = synth.web_google_like(scale=0.05)
    x = O.x_vec_fast(ncols, "rand")
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    for devices in ([0], [0, 0, 0]):
        M = cvr_amd.MultiMatrix(nrows, ncols, rp, ci, va, devices)
        assert M.shards == len(devices) and not M.uses_rccl
        b = cvr_amd.row_partition(rp, len(devices), capi.ROW_COST_MILLI_DEFAULT)      # the library cuts on predicted time (cvr_row_partition_cost)
        seen = 0
        for p in range(M.shards):
            info, r0, r1, dev = M.shard_info(p)
            assert (r0, r1, dev) == (b[p], b[p + 1], devices[p]) and info.nrows == r1 - r0 and info.nnz == rp[r1] - rp[r0]
            seen += info.nnz
        assert seen == len(ci)
        y, t = M.spmv(x, iters=5)
        _assert_close(y, yref, absy, TOL64, ("multi", devices))
        assert t.iters == 5 and 0 < t.min_s <= t.median_s <= t.max_s and t.step_mean_s >= 0.5 * t.mean_s
        assert (t.gather_mean_s == 0) == (len(devices) == 1)
        y2, _ = M.spmv(x)
        assert np.array_equal(y.view(np.uint64), y2.view(np.uint64))
        M.close()
    with pytest.raises(cvr_amd.CvrError):
        cvr_amd.MultiMatrix(nrows, ncols, rp, ci, va, [0, 99])
    # single-device timing carries the same fields
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va)
    _, t = A.spmv(x, iters=7)
    assert t.min_s <= t.median_s <= t.max_s and t.step_median_s == t.median_s and t.gather_mean_s == 0
    A.close()


def test_multi_device_shards_take_interleaved_panels():
    """row shards of a power-law matrix (soc-LiveJournal1 shape x 0.65, two shards on one device): every shard is large enough for the
    automatic rule to give it column panels with interleaved chunks; the gathered y against the oracle, bitwise reproducible"""
    nrows, ncols, rp, ci, va = synth.livejournal_like(scale=0.65)
    x = O.x_vec_fast(ncols, "rand")
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    M = cvr_amd.MultiMatrix(nrows, ncols, rp, ci, va, [0, 0])
    for p in range(M.shards):
        info, r0, r1, _ = M.shard_info(p)
        assert info.nnz >= 8 << 20 and info.col_panels >= 8 and info.interleave == 1, (p, info.nnz, info.col_panels, info.interleave)
    y, _ = M.spmv(x, iters=3)
    _assert_close(y, yref, absy, TOL64, "interleaved shards")
    y2, _ = M.spmv(x)
    assert np.array_equal(y.view(np.uint64), y2.view(np.uint64))
    M.close()


@pytest.mark.parametrize("kind", ["resident_phases", "plain", "panels", "hub", "interleaved", "interleaved_panels", "gang", "gang_tags"])
def test_image_cache_roundtrip_and_staleness(tmp_path, kind):
    """cvr_save_image / cvr_load_image: the converted image from disk gives the same y bit for bit without analysis, planner or
    converter; a file written for another source file, other options or a damaged file is refused with a code"""
    if kind == "hub":
        nrows, ncols, rp, ci, va = synth.rmat(16, dtype=np.float32)
        opts = dict(hub_table=300, steps_per_chunk=16)
    else:
        nrows, ncols, rp, ci, va = synth.web_google_like(scale=0.05)
        opts = dict(resident_phases=dict(steps_per_chunk=12, waves_per_block=8, x_window=2048, col_phases=6), plain=dict(steps_per_chunk=16),
                    panels=dict(col_panels=3, steps_per_chunk=16), interleaved=dict(col_panels=1, interleave=1, steps_per_chunk=32, waves_per_block=4),
                    interleaved_panels=dict(col_panels=8, interleave=1), gang=dict(col_panels=1, interleave=1, steps_per_chunk=32, waves_per_block=4, gang=1),
                    gang_tags=dict(col_panels=1, interleave=1, steps_per_chunk=16, waves_per_block=2, gang=1, row_tags16=1))[kind]
    x = O.x_vec_fast(ncols, "rand").astype(va.dtype)
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, **opts)
    y, _ = A.spmv(x)
    path = str(tmp_path / "m.cvrimg")
    key = capi.SourceKey(size=123, mtime_ns=456, hash=789, mode=0)
    A.save_image(path, key)
    B = cvr_amd.CvrMatrix.from_image(path, key, **opts)
    assert (B.info.nchunks, B.info.nshared, B.info.col_panels, B.info.col_phases, B.info.value_dict, B.info.hub_entries, B.info.interleave, B.info.spmv_launches) == \
           (A.info.nchunks, A.info.nshared, A.info.col_panels, A.info.col_phases, A.info.value_dict, A.info.hub_entries, A.info.interleave, A.info.spmv_launches)
    assert A.info.interleave == (1 if kind.startswith("interleaved") or kind.startswith("gang") else 0)
    assert (A.info.gang, B.info.gang) == ((A.info.waves_per_block,) * 2 if kind != "interleaved" and A.info.interleave else (0, 0))
    assert B.info.plan_s == 0 and B.info.convert_s == 0
    y2, _ = B.spmv(x)
    assert np.array_equal(y.view(np.uint8), y2.view(np.uint8))
    if kind in ("resident_phases", "plain", "interleaved", "gang", "gang_tags"):
        ia, ib = A.export_image(), B.export_image()
        for k in ("image", "desc", "target", "shared") + (("gbase", "desc2") if kind.startswith("gang") else ()):
            assert np.array_equal(ia[k], ib[k]), k
    B.close()
    A.close()
    other = capi.SourceKey(size=123, mtime_ns=457, hash=789, mode=0)              # the source file has changed
    with pytest.raises(cvr_amd.CvrError) as e:
        cvr_amd.CvrMatrix.from_image(path, other, **opts)
    assert e.value.code == capi.ERR_STATE
    with pytest.raises(cvr_amd.CvrError) as e:                                    # other options
        cvr_amd.CvrMatrix.from_image(path, key, **dict(opts, split_threshold=7))
    assert e.value.code == capi.ERR_STATE
    with open(path, "r+b") as f:                                                  # a truncated file
        f.truncate(os.path.getsize(path) - 100)
    with pytest.raises(cvr_amd.CvrError) as e:
        cvr_amd.CvrMatrix.from_image(path, key, **opts)
    assert e.value.code == capi.ERR_IO


def test_cli_caches_keyed_to_the_file(tmp_path):
    """CVR_CACHE=1: second run of spmv.cvr on the same file takes the parsed CSR and the converted image from the caches beside
    it (no text parse, no conversion); once the file changes both are refused and rebuilt"""
    import json
    import shutil
    import subprocess
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "spmv.cvr")
    mtx = str(tmp_path / "m.mtx")
    shutil.copy(os.path.join(GOLD, "mtx", "pl2000_pattern.mtx"), mtx)

    def run():
        r = subprocess.run([exe, mtx, "2", "5"], capture_output=True, text=True, timeout=120, env=dict(os.environ, CVR_CACHE="1", CVR_X="rand"))
        assert r.returncode == 0 and "Very Good! Your result is correct" in r.stdout, r.stdout + r.stderr
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"backend"')][0])
    a, b = run(), run()
    assert (a["csr_cache"], a["image_cache"]) == ("miss", "miss") and (b["csr_cache"], b["image_cache"]) == ("hit", "hit")
    assert b["wrong"] == 0 and b["chunks"] == a["chunks"]
    time.sleep(0.01)
    shutil.copy(os.path.join(GOLD, "mtx", "skew12.mtx"), mtx)                 # another matrix under the same name
    c = run()
    assert (c["csr_cache"], c["image_cache"]) == ("miss", "miss") and c["rows"] != a["rows"] and c["wrong"] == 0
    assert run()["image_cache"] == "hit"


@pytest.mark.parametrize("workload,rank", [("rmat26", "3/8"), ("banded28e6", "0/8")])
def test_one_rank_of_the_eight_gpu_configurations(workload, rank):
    """BASELINE.json configs[3] / [4] (nlpkkt240's shape and R-MAT scale 26 over 8 GPUs) cannot run here (one GPU per box): one
    rank's row shard of each, built on the device with the full replicated x (268 MB / 224 MB), timed and checked row by row
    against a torch fp64 segment sum inside bench.py (--emulate-rank; the eight per-rank times: profiles/r03_rank_emulation_*.json)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", workload, "--emulate-rank", rank, "--steps", "20", "--warmup", "3", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["verdict_wrong_rows"] == 0 and d["config"]["emulated_rank"] == [int(v) for v in rank.split("/")]
    assert d["config"]["rank_nnz"] > 9e7 and d["roofline"]["kernel_us"] > 0
    # bench.py's guard is a torch segment sum; the oracle's word on the same configuration: a million rows of that rank's shard, built the
    # same way (device-resident CSR, the full replicated x), every row against the pinned CSR loop
    import torch
    from cvr_amd import synth_dev as D
    r, n = (int(v) for v in rank.split("/"))
    if workload.startswith("rmat"):
        scale = int(workload[4:])
        ncols = 1 << scale
        lo = (ncols // n) * r + 12345
        rp_t, ci_t, va_t = D.rmat_rows(scale, lo, lo + 1_000_000, device="cuda")
    else:
        ncols = int(float(workload[6:]))
        lo = (ncols // n) * r
        rp_t, ci_t, va_t = D.banded_rows(ncols, lo, lo + 1_000_000, device="cuda")
    f32 = va_t.dtype == torch.float32
    A = cvr_amd.CvrMatrix.from_device(1_000_000, ncols, rp_t.data_ptr(), ci_t.data_ptr(), va_t.data_ptr(), is_f32=f32)
    x = synth.x_rand(ncols, np.float32 if f32 else np.float64)
    y, _ = A.spmv(x)
    yref, absy = O.csr_spmv64(rp_t.cpu().numpy(), ci_t.cpu().numpy(), va_t.cpu().numpy(), x)
    _assert_close(y, yref, absy + 1e-30, TOL32 if f32 else TOL64, (workload, rank, "slice against the oracle"))
    A.close()


def test_banded_and_rmat_shapes():
    """the other BASELINE.json shapes at reduced size: banded symmetric (nlpkkt240's shape) fp64, R-MAT fp32"""
    nrows, ncols, rp, ci, va = synth.banded_sym(300_000)
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va)
    x = O.x_vec_fast(ncols, "rand")
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    y, _ = A.spmv(x)
    _assert_close(y, yref, absy, TOL64, "banded")
    A.close()
    nrows, ncols, rp, ci, va = synth.rmat(17, dtype=np.float32)
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va)
    assert A.info.nshared > 0 or np.diff(rp).max() <= 16 * A.info.steps_per_chunk
    x = O.x_vec_fast(ncols, "rand").astype(np.float32)
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    y, _ = A.spmv(x)
    _assert_close(y, yref, absy + 1e-30, TOL32, "rmat17")
    y2, _ = A.spmv(x)
    assert np.array_equal(y.view(np.uint32), y2.view(np.uint32))
    A.close()


def test_degenerate_shapes():
    """no rows, no columns, no non-zeros: handled by the boundary without touching the kernels' limits"""
    A = cvr_amd.CvrMatrix(0, 5, np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int32), np.zeros(0))
    y, _ = A.spmv(np.ones(5))
    assert len(y) == 0 and A.info.nchunks == 0
    A.close()
    A = cvr_amd.CvrMatrix(3, 0, np.zeros(4, dtype=np.int64), np.zeros(0, dtype=np.int32), np.zeros(0))
    y, _ = A.spmv(np.zeros(0))
    assert np.array_equal(y, np.zeros(3))
    A.close()


def _power_iteration_numpy(rp, ci, va, iters=20):
    """the same loop in numpy: the checker of the device-resident caller"""
    n = len(rp) - 1
    rows = np.repeat(np.arange(n), np.diff(rp))
    x = np.ones(n) / np.sqrt(n)
    lam = 0.0
    for _ in range(iters):
        y = np.bincount(rows, weights=va * x[ci], minlength=n)
        lam = float(x @ y)
        x = y / np.linalg.norm(y)
    return lam, x


def test_power_iteration_device_resident():
    """the iterative caller (cvr_amd/power.py): y feeds x on the device; against the same loop in numpy"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    from cvr_amd import power
    nrows, ncols, rp, ci, va = synth.web_google_like(scale=0.02)
    va = np.abs(va) + 0.5                      # positive matrix: the dominant eigenpair is real and simple
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va)
    lam, x, sec = power.power_iteration(A, nrows, iters=30)
    lam_ref, x_ref = _power_iteration_numpy(rp, ci, va, iters=30)
    assert abs(lam - lam_ref) <= 1e-9 * abs(lam_ref) and sec > 0
    assert np.allclose(x.cpu().numpy(), x_ref, rtol=0, atol=1e-9)
    lam2, x2, _ = power.power_iteration(A, nrows, iters=30)              # fixed reduction trees: bit for bit again
    assert lam2 == lam and torch.equal(x2.view(torch.int64), x.view(torch.int64))
    # the sharded form of the loop with a 1-rank RCCL communicator (all-gather, un-padding, rebuilt x)
    comm = cvr_amd.Comm(cvr_amd.comm_unique_id(), 1, 0, 0)
    ranks, rank, version = comm.info()                                    # what RCCL itself reports (bench.py's "rccl" field at N > 1)
    assert (ranks, rank) == (1, 0) and version > 20000, (ranks, rank, version)
    lam3, x3, _ = power.power_iteration(A, nrows, bounds=[0, nrows], comm=comm, iters=30)
    assert lam3 == lam and torch.equal(x3.view(torch.int64), x.view(torch.int64))
    with pytest.raises(cvr_amd.CvrError):
        power.power_iteration(A, nrows, bounds=[0, nrows - 1], comm=comm, iters=1)
    comm.close()
    A.close()
    # fp32 matrix and column panels: the same loop, fp32 tolerance
    A32 = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va.astype(np.float32), col_panels=2)
    lam32, x32, _ = power.power_iteration(A32, nrows, iters=30)
    assert abs(lam32 - lam_ref) <= 1e-4 * abs(lam_ref)
    assert np.allclose(x32.cpu().numpy().astype(np.float64), x_ref, rtol=0, atol=1e-5)
    A32.close()
    # an fp32 matrix whose dominant eigenvalue (~1e21) squared is beyond fp32: the loop falls back to exact normalisation
    # after its first step instead of overflowing on lambda^2
    Abig = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, (va * 1e19).astype(np.float32))
    lamb, xb, _ = power.power_iteration(Abig, nrows, iters=30)
    assert np.isfinite(lamb) and abs(lamb - lam_ref * 1e19) <= 1e-4 * abs(lam_ref) * 1e19
    assert np.allclose(xb.cpu().numpy().astype(np.float64), x_ref, rtol=0, atol=1e-5)
    Abig.close()
    nr, nc, rp2, ci2, va2 = CASES["few_rows_lt_lanes"]
    if nr != nc:                                                         # not square: refused with a code
        B = cvr_amd.CvrMatrix(nr, nc, rp2, ci2, va2)
        xx = torch.ones(B.info.x_elems, dtype=torch.float64, device="cuda")
        with pytest.raises(cvr_amd.CvrError):
            B.power_iteration(xx.data_ptr(), 1)
        B.close()


def test_power_iteration_step_in_the_spmv_epilogue(monkeypatch):
    """On a matrix in the resident layout (column phases, no rows cut over chunks) the step's dot products and the next iterate come
    out of the SpMV kernel's write-out (IterEpilogue): against the numpy loop, bit for bit again on a second run, against the loop with
    the step as a pass of its own (CVR_DEBUG=iter_unfused), and fp32 with an eigenvalue whose square leaves fp32 (exact normalisation)."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    from cvr_amd import power
    nrows, ncols, rp, ci, va = synth.web_google_like()          # (full size: S = 48, no row is cut over chunks)
    va = np.abs(va) + 0.5
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va)
    assert A.info.col_phases > 1 and A.info.nshared == 0
    lam, x, sec = power.power_iteration(A, nrows, iters=25)
    lam_ref, x_ref = _power_iteration_numpy(rp, ci, va, iters=25)
    assert abs(lam - lam_ref) <= 1e-9 * abs(lam_ref) and sec > 0
    assert np.allclose(x.cpu().numpy(), x_ref, rtol=0, atol=1e-9)
    lam2, x2, sec2 = power.power_iteration(A, nrows, iters=25)
    assert lam2 == lam and torch.equal(x2.view(torch.int64), x.view(torch.int64))
    monkeypatch.setenv("CVR_DEBUG", "iter_unfused")
    lam3, x3, sec3 = power.power_iteration(A, nrows, iters=25)
    monkeypatch.delenv("CVR_DEBUG")
    assert abs(lam3 - lam) <= 1e-12 * abs(lam) and np.allclose(x3.cpu().numpy(), x.cpu().numpy(), rtol=0, atol=1e-12)
    assert min(sec, sec2) < sec3                     # one launch per iteration against two
    A.close()
    for scale_v, tol in ((1.0, 1e-4), (1e19, 1e-4)):
        A32 = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, (va * scale_v).astype(np.float32))
        assert A32.info.col_phases > 1
        lam32, x32, _ = power.power_iteration(A32, nrows, iters=25)
        assert np.isfinite(lam32) and abs(lam32 - lam_ref * scale_v) <= tol * abs(lam_ref) * scale_v
        assert np.allclose(x32.cpu().numpy().astype(np.float64), x_ref, rtol=0, atol=1e-5)
        A32.close()


@pytest.mark.parametrize("P", [2, 3, 7])
def test_column_panels_parity(P):
    """column panels (each panel's slice of x L2-resident, partial sums combined in panel order): same tolerance
    against the CSR oracle, bitwise reproducible, empty rows written as 0"""
    for name, S in (("power_law_3000", 8), ("leading_trailing_empty", 4), ("dense_row_plus_singletons", 8), ("single_entry", 4),
                    ("empty_matrix_rows_only", 4), ("two_giants", 16)):
        nrows, ncols, rp, ci, va = CASES[name]
        A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=S, col_panels=P)
        assert A.info.col_panels == P
        for mode in ("ones", "rand"):
            x = O.x_vec_fast(ncols, mode)
            yref, absy = O.csr_spmv64(rp, ci, va, x)
            y, _ = A.spmv(x)
            _assert_close(y, yref, absy, TOL64, (name, P, mode))
            assert np.all(y[np.diff(rp) == 0] == 0)
        y2, _ = A.spmv(x)
        assert np.array_equal(y.view(np.uint64), y2.view(np.uint64))
        with pytest.raises(cvr_amd.CvrError):
            A.export_image()
        A.close()
    nrows, ncols, rp, ci, va = CASES32["power_law_3000"]
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=8, col_panels=P)
    x = O.x_vec_fast(ncols, "rand").astype(np.float32)
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    y, _ = A.spmv(x)
    _assert_close(y, yref, absy + 1e-30, TOL32, ("f32", P))
    A.close()


def test_column_panels_livejournal_shape():
    nrows, ncols, rp, ci, va = synth.livejournal_like(scale=0.125)
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, col_panels=4)
    x = O.x_vec_fast(ncols, "rand")
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    y, _ = A.spmv(x)
    _assert_close(y, yref, absy, TOL64, "lj panels")
    A.close()


def test_panel_plans_as_one_submission_same_result(monkeypatch, capfd):
    """Column panels split on the device, without hub tables: all panels' chunk plans are enqueued together and write nzb / pad / desc /
    cut rows on the device (plan_panels_batched); against the panel-by-panel path (CVR_DEBUG=serial_panel_plans): same counts, same y bits; y
    against the CSR oracle.  (hub_table = 0 takes the matrix down that path whatever its columns' popularity.)"""
    nrows, ncols, rp, ci, va = synth.livejournal_like(scale=0.25)
    x = O.x_vec_fast(ncols, "rand")
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    got = []
    monkeypatch.setenv("CVR_DEBUG", "fused_trace")
    for serial in (False, True):
        if serial:
            monkeypatch.setenv("CVR_DEBUG", "fused_trace,serial_panel_plans")
        A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, col_panels=8, hub_table=0, split_threshold=64)      # (a low threshold: rows are cut over chunks)
        if serial:
            monkeypatch.setenv("CVR_DEBUG", "fused_trace")
        i = A.info
        y, _ = A.spmv(x)
        got.append(((i.col_panels, i.nchunks, i.nshared, i.steps_per_chunk, i.lds_bytes, i.spmv_launches), y))
        A.close()
    assert capfd.readouterr().err.count("chunk plans as one submission") == 1        # (the first handle took the path, the second did not)
    assert got[0][0] == got[1][0], (got[0][0], got[1][0])
    assert got[0][0][2] > 0                                           # rows cut over chunks: their list came from the device plan
    assert np.array_equal(got[0][1].view(np.uint64), got[1][1].view(np.uint64))
    _assert_close(got[0][1], yref, absy, TOL64, "batched panel plans")


def test_value_dictionary():
    """matrices with at most 256 distinct values (pattern matrices: the reference assigns index % 13, spmv.cpp:417)
    store one code byte per slot: image == mirror bit for bit, y bitwise identical to the plain layout (same values,
    same order of operations), general real matrices fall back to the plain layout"""
    nrows, ncols, rp, ci, va = synth.web_google_like(scale=0.05)
    assert len(np.unique(va)) == 13
    for dt in (np.float64, np.float32):
        v = va.astype(dt)
        x = O.x_vec_fast(ncols, "rand").astype(dt)
        A0 = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, v, steps_per_chunk=16, value_dict=0)
        A1 = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, v, steps_per_chunk=16)           # auto
        assert A0.info.value_dict == 0 and A1.info.value_dict == 13                   # 0 is among the 13 values
        assert A1.info.image_bytes < (0.45 if dt == np.float64 else 0.7) * A0.info.image_bytes    # 12 -> 5 and 8 -> 5 bytes per slot
        mir = O.Cvr64(nrows, ncols, rp, ci, v, 16, use_dict=True)
        img = A1.export_image()
        assert np.array_equal(img["image"], mir.image) and np.array_equal(img["desc"], mir.desc)
        y0, _ = A0.spmv(x)
        y1, _ = A1.spmv(x)
        assert np.array_equal(y0.view(np.uint8), y1.view(np.uint8))
        yref, absy = O.csr_spmv64(rp, ci, v, x)
        _assert_close(y1, yref, absy + 1e-30, TOL32 if dt == np.float32 else TOL64, ("dict", dt))
        A0.close()
        A1.close()
    # special bit patterns are values like any other: -0.0, inf, nan payloads, the all-ones pattern
    nrows, ncols, rp, ci, va = CASES["uniform_2000"]
    special = np.array([0.0, -0.0, 1.5, -2.25, np.inf], dtype=np.float64)
    v = special[np.arange(len(ci)) % len(special)].copy()
    v[::97] = np.frombuffer(np.uint64(0xFFFFFFFFFFFFFFFF).tobytes(), dtype=np.float64)[0]      # a NaN: all ones
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, v, steps_per_chunk=8)
    assert A.info.value_dict == 6
    mir = O.Cvr64(nrows, ncols, rp, ci, v, 8, use_dict=True)
    assert np.array_equal(A.export_image()["image"], mir.image)
    A.close()
    v32 = special[np.arange(len(ci)) % len(special)].astype(np.float32)
    v32[::89] = np.frombuffer(np.uint32(0xFFFFFFFF).tobytes(), dtype=np.float32)[0]
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, v32, steps_per_chunk=8)
    assert A.info.value_dict == 6
    mir = O.Cvr64(nrows, ncols, rp, ci, v32, 8, use_dict=True)
    assert np.array_equal(A.export_image()["image"], mir.image)
    A.close()
    # 300 distinct values: no dictionary
    nrows, ncols, rp, ci, va = CASES["power_law_3000"]
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, (np.arange(len(ci)) % 300).astype(np.float64), steps_per_chunk=8)
    assert A.info.value_dict == 0
    A.close()


def test_abi_call_order_and_null_handling():
    """the C ABI returns codes instead of crashing: null handles, preprocess twice, keep_csr, destroy(NULL)"""
    import ctypes as C
    L = capi.lib()
    assert L.cvr_destroy(None) == 0
    assert L.cvr_spmv(None, None, None, 1, None) == capi.ERR_INVALID
    assert L.cvr_preprocess(None, 0, None) == capi.ERR_INVALID
    assert L.cvr_get_info(None, None) == capi.ERR_INVALID
    nrows, ncols, rp, ci, va = CASES["uniform_2000"]
    view = capi.CsrView(nrows, ncols, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, 0)
    for keep in (0, 1):
        h = C.c_void_p()
        assert L.cvr_create(C.byref(h), C.byref(view), None) == 0, cvr_amd.last_error()
        x = np.ones(ncols)
        y = np.zeros(nrows)
        assert L.cvr_spmv(h, x.ctypes.data, y.ctypes.data, 1, None) == capi.ERR_STATE      # before cvr_preprocess
        sec = C.c_double()
        assert L.cvr_preprocess(h, keep, C.byref(sec)) == 0 and sec.value > 0
        rc2 = L.cvr_preprocess(h, keep, C.byref(sec))
        assert rc2 == (0 if keep else capi.ERR_STATE)                                       # the CSR is gone unless kept
        assert L.cvr_spmv(h, x.ctypes.data, y.ctypes.data, 2, None) == 0
        yref, absy = O.csr_spmv64(rp, ci, va, x)
        _assert_close(y, yref, absy, TOL64, ("abi", keep))
        assert L.cvr_spmv_device(h, None, None, None) == capi.ERR_INVALID
        assert L.cvr_destroy(h) == 0


def test_native_gather_loop_single_rank_rccl():
    """cvr_spmv_gather_repeat with a 1-rank RCCL communicator: the library's own pipelined SpMV + all-gather loop
    (the N > 1 path of bench.py; more ranks need more GPUs than this box has)"""
    import torch
    nrows, ncols, rp, ci, va = CASES["power_law_3000"]
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va)
    comm = cvr_amd.Comm(cvr_amd.comm_unique_id(), 1, 0, 0)
    dev = torch.device("cuda", 0)
    xh = O.x_vec_fast(ncols)
    x = torch.zeros(A.info.x_elems, dtype=torch.float64, device=dev)
    x[:ncols] = torch.from_numpy(xh).to(dev)
    max_rows = nrows + 5                                   # padded slices, as with uneven shards
    ny = max(A.info.yext_elems, max_rows)
    ys = [torch.full((ny,), float("nan"), dtype=torch.float64, device=dev) for _ in range(2)]
    yalls = [torch.full((max_rows,), float("nan"), dtype=torch.float64, device=dev) for _ in range(2)]
    st = torch.cuda.Stream(device=dev)
    yref, absy = O.csr_spmv64(rp, ci, va, xh)
    for n, ov in ((1, False), (2, True), (7, False), (7, True), (4, False)):
        for t in ys + yalls:
            t.fill_(float("nan"))
        torch.cuda.synchronize()
        last = A.spmv_gather(comm, x.data_ptr(), [t.data_ptr() for t in ys], [t.data_ptr() for t in yalls], max_rows, n, st.cuda_stream, overlap=ov)
        st.synchronize()
        assert last == (n - 1) & 1
        _assert_close(yalls[last][:nrows].cpu().numpy(), yref, absy, TOL64, ("native gather", n))
        assert torch.equal(yalls[last][:nrows].view(torch.int64), ys[last][:nrows].view(torch.int64))
    with pytest.raises(cvr_amd.CvrError):                  # a slice shorter than the shard
        A.spmv_gather(comm, x.data_ptr(), [t.data_ptr() for t in ys], [t.data_ptr() for t in yalls], nrows - 1, 1, st.cuda_stream)
    comm.close()
    A.close()


def test_lds_row_stage_sizes_and_direct_store_fallback():
    """the LDS stage of row sums is sized for the chunk with the most rows; beyond kYStageMax (4096) a chunk stores directly:
    rows of one non-zero (64*S rows per chunk) with S = 16 (1024 rows), 64 (4096: the cap), 128 (8192: fallback), mixed with long rows"""
    rng = np.random.default_rng(7)
    nrows = ncols = 30000
    lens = np.ones(nrows, dtype=np.int64)
    lens[rng.integers(0, nrows, 40)] = rng.integers(100, 3000, 40)
    lens[rng.integers(0, nrows, 500)] = 0
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ci = rng.integers(0, ncols, int(rp[-1])).astype(np.int32)
    va = rng.standard_normal(int(rp[-1]))
    x = O.x_vec_fast(ncols)
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    for S in (16, 64, 128):
        A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=S)
        y, _ = A.spmv(x)
        _assert_close(y, yref, absy, TOL64, ("ystage", S))
        A.close()


def test_tune_steps_returns_a_measured_choice():
    """cvr_tune_steps: S from measurement; the tuned matrix computes the same y"""
    nrows, ncols, rp, ci, va = CASES["power_law_3000"]
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, tune_steps=True)
    assert A.info.steps_per_chunk in range(8, 68, 4) and A.tuning_s > 0
    x = O.x_vec_fast(ncols)
    y, _ = A.spmv(x)
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    _assert_close(y, yref, absy, TOL64, "tuned")
    A.close()


@pytest.mark.parametrize("name", ["power_law_3000", "two_giants", "leading_trailing_empty", "empty_matrix_rows_only"])
def test_csr_arrays_already_on_the_device(name):
    """cvr_csr_view.arrays_on_device = 1 (a GPU-resident caller, here torch tensors): same image, bit for bit, and same y as
    with host arrays; a column out of range is still rejected (device-side range check)"""
    import torch
    nrows, ncols, rp, ci, va = CASES[name]
    dev = torch.device("cuda", 0)
    trp = torch.from_numpy(np.ascontiguousarray(rp, dtype=np.int64)).to(dev)
    tci = torch.from_numpy(np.ascontiguousarray(ci, dtype=np.int32)).to(dev) if len(ci) else torch.zeros(1, dtype=torch.int32, device=dev)
    tva = torch.from_numpy(np.ascontiguousarray(va, dtype=np.float64)).to(dev) if len(va) else torch.zeros(1, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=16)
    B = cvr_amd.CvrMatrix.from_device(nrows, ncols, trp.data_ptr(), tci.data_ptr(), tva.data_ptr(), steps_per_chunk=16)
    ia, ib = A.export_image(), B.export_image()
    for k in ("image", "desc", "target", "shared"):
        assert np.array_equal(ia[k], ib[k]), (name, k)
    x = O.x_vec_fast(ncols)
    ya, _ = A.spmv(x)
    yb, _ = B.spmv(x)
    assert np.array_equal(ya.view(np.uint8), yb.view(np.uint8))
    A.close()
    B.close()
    if len(ci):
        bad = tci.clone()
        bad[len(ci) // 2] = ncols                                  # one past the last column
        torch.cuda.synchronize()
        with pytest.raises(cvr_amd.CvrError) as e:
            cvr_amd.CvrMatrix.from_device(nrows, ncols, trp.data_ptr(), bad.data_ptr(), tva.data_ptr())
        assert e.value.code == capi.ERR_INVALID


def test_large_device_arrays_are_checked_and_planned_where_they_are():
    """row_ptr of a large matrix in device memory never comes to the host (the checks are a kernel, the planner reads the device copy):
    same image and y as from host arrays; a row pointer that decreases, a negative first entry and a column out of range are rejected
    with the codes of the host checks."""
    import torch
    nrows, ncols, rp, ci, va = synth.web_google_like(0.3)
    assert nrows >= 200000                                     # (cvr_create's threshold for the device planner)
    dev = torch.device("cuda", 0)
    trp, tci, tva = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (rp.astype(np.int64), ci.astype(np.int32), va.astype(np.float64)))
    torch.cuda.synchronize()
    for kw in (dict(), dict(steps_per_chunk=4), dict(col_panels=3)):
        A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, **kw)
        B = cvr_amd.CvrMatrix.from_device(nrows, ncols, trp.data_ptr(), tci.data_ptr(), tva.data_ptr(), **kw)
        assert (A.info.nchunks, A.info.nshared, A.info.steps_per_chunk, A.info.col_panels) == (B.info.nchunks, B.info.nshared, B.info.steps_per_chunk, B.info.col_panels)
        if not kw.get("col_panels"):
            ia, ib = A.export_image(), B.export_image()
            for k in ("image", "desc", "target", "shared"):
                assert np.array_equal(ia[k], ib[k]), (kw, k)
        x = O.x_vec_fast(ncols, "rand")
        ya, _ = A.spmv(x)
        yb, _ = B.spmv(x)
        assert np.array_equal(ya.view(np.uint64), yb.view(np.uint64)), kw
        A.close(); B.close()
    bad = trp.clone()
    bad[nrows // 2] = bad[nrows // 2 - 1] - 1                     # decreases
    with pytest.raises(cvr_amd.CvrError) as e:
        cvr_amd.CvrMatrix.from_device(nrows, ncols, bad.data_ptr(), tci.data_ptr(), tva.data_ptr())
    assert e.value.code == capi.ERR_INVALID and "decreases" in str(e.value)
    bad = trp.clone()
    bad[0] = -1
    with pytest.raises(cvr_amd.CvrError) as e:
        cvr_amd.CvrMatrix.from_device(nrows, ncols, bad.data_ptr(), tci.data_ptr(), tva.data_ptr())
    assert e.value.code == capi.ERR_INVALID
    badc = tci.clone()
    badc[len(ci) // 3] = ncols
    torch.cuda.synchronize()
    with pytest.raises(cvr_amd.CvrError) as e:
        cvr_amd.CvrMatrix.from_device(nrows, ncols, trp.data_ptr(), badc.data_ptr(), tva.data_ptr())
    assert e.value.code == capi.ERR_INVALID


def test_device_arrays_with_column_panels_are_split_on_the_device():
    """the column-panel split runs on the device (cvr_split.hip: one stable radix-sort pass by panel), for device-resident
    arrays and for host arrays (staged once); bit for bit the same y as the host split (CVR_DEBUG=host_split, the fallback) of
    the same matrix -- sorted and unsorted rows, empty rows, fp32, 2..64 panels"""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(31)
    todo = [(name, CASES[name], P) for name, P in (("power_law_3000", 3), ("two_giants", 2), ("leading_trailing_empty", 7), ("dense_row_plus_singletons", 64))]
    nrows, ncols, rp, ci, va = CASES["power_law_3000"]
    ci_shuffled = ci.copy()
    for r in range(nrows):                                     # rows whose columns are not sorted
        seg = ci_shuffled[rp[r]:rp[r + 1]]
        rng.shuffle(seg)
    todo.append(("unsorted rows", (nrows, ncols, rp, ci_shuffled, va), 5))
    todo.append(("fp32", CASES32["power_law_3000"], 4))
    # several tiles of the split (2 048 non-zeros each); one of them spans more rows than the workgroup's copy of the row pointers holds (a run
    # of 9 000 empty rows: that tile searches row_ptr itself), the others look their rows up in LDS; single-element rows and a 3 000-element row
    deg = np.concatenate([rng.integers(1, 40, 300), np.zeros(9000, dtype=np.int64), np.ones(2500, dtype=np.int64), [3000], rng.integers(0, 6, 4000)])
    n2 = len(deg)
    rp2 = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    ci2 = np.concatenate([np.sort(rng.choice(n2, int(k), replace=False)) for k in deg]).astype(np.int32)
    todo.append(("long run of empty rows", (n2, n2, rp2, ci2, rng.standard_normal(len(ci2))), 5))
    for name, (nrows, ncols, rp, ci, va), P in todo:
        trp = torch.from_numpy(np.ascontiguousarray(rp, dtype=np.int64)).to(dev)
        tci = torch.from_numpy(np.ascontiguousarray(ci, dtype=np.int32)).to(dev)
        tva = torch.from_numpy(np.ascontiguousarray(va)).to(dev)
        torch.cuda.synchronize()
        f32 = va.dtype == np.float32
        os.environ["CVR_DEBUG"] = "host_split"                       # the threaded counting sort on the host (the fallback path)
        try:
            A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, col_panels=P)
        finally:
            del os.environ["CVR_DEBUG"]
        B = cvr_amd.CvrMatrix.from_device(nrows, ncols, trp.data_ptr(), tci.data_ptr(), tva.data_ptr(), is_f32=f32, col_panels=P)
        C2 = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, col_panels=P)      # host arrays, staged to the device and split there
        assert A.info.col_panels == P and B.info.col_panels == P and C2.info.col_panels == P
        assert (A.info.nchunks, A.info.nslots, A.info.nshared) == (B.info.nchunks, B.info.nslots, B.info.nshared) == (C2.info.nchunks, C2.info.nslots, C2.info.nshared), name
        x = O.x_vec_fast(ncols, "rand").astype(va.dtype)
        ya, yb, yc = A.spmv(x)[0], B.spmv(x)[0], C2.spmv(x)[0]
        C2.close()
        assert np.array_equal(ya.view(np.uint8), yb.view(np.uint8)) and np.array_equal(ya.view(np.uint8), yc.view(np.uint8)), name
        yref, absy = O.csr_spmv64(rp, ci, va, x)
        _assert_close(yb, yref, absy + 1e-30, TOL32 if f32 else TOL64, ("device split", name))
        A.close()
        B.close()


def test_panels_keep_their_slices_of_the_split_until_the_csr_is_released():
    """Column panels split on the device: the panels' column indices and values are slices of the split's arrays, owned by the handle.  A handle
    that keeps its CSR converts a second time from them (same y bit for bit), releases them with the third conversion, and a further cvr_preprocess
    is refused with a code because the CSR is gone; device memory returns to its level after cvr_destroy."""
    import torch
    nrows, ncols, rp, ci, va = synth.web_google_like(0.25)
    x = O.x_vec_fast(ncols, "rand")
    yref, absy = O.csr_spmv64(rp, ci, va, x)
    cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, col_panels=4, keep_csr=True).close()      # (what a process allocates once: code objects, streams, the runtime's pools)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, col_panels=4, keep_csr=True)
    assert A.info.col_panels == 4
    y1 = A.spmv(x)[0].copy()
    _assert_close(y1, yref, absy, TOL64, "first conversion")
    sec = capi.C.c_double()
    assert capi.lib().cvr_preprocess(A._h, 1, capi.C.byref(sec)) == 0          # again, still keeping the CSR
    assert np.array_equal(A.spmv(x)[0].view(np.uint8), y1.view(np.uint8))
    assert capi.lib().cvr_preprocess(A._h, 0, capi.C.byref(sec)) == 0          # and once more, releasing it
    assert np.array_equal(A.spmv(x)[0].view(np.uint8), y1.view(np.uint8))
    assert capi.lib().cvr_preprocess(A._h, 0, capi.C.byref(sec)) != 0          # nothing left to convert from
    assert np.array_equal(A.spmv(x)[0].view(np.uint8), y1.view(np.uint8))
    A.close()
    torch.cuda.synchronize()
    assert abs(torch.cuda.mem_get_info()[0] - free0) < (8 << 20)


@pytest.mark.parametrize("kw", [{}, {"steps_per_chunk": 16, "col_panels": 4}, {"steps_per_chunk": 16, "col_panels": 4, "interleave": 1, "waves_per_block": 4, "gang": 1}])
def test_spmv_launches_can_be_captured_in_a_hip_graph(kw):
    """cvr_spmv_device makes no synchronising call: a caller can capture it (here with torch.cuda.graph) and replay.  With column panels the handle has been used
    on the default stream before: the capture on a side stream must neither fail nor touch that stream (run_spmv: the partial sums' event and stream capture)"""
    import torch
    nrows, ncols, rp, ci, va = CASES["power_law_3000"]
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, **kw)
    assert A.info.col_panels == kw.get("col_panels", 1)
    dev = torch.device("cuda", 0)
    xh = O.x_vec_fast(ncols)
    x = torch.zeros(A.info.x_elems, dtype=torch.float64, device=dev)
    x[:ncols] = torch.from_numpy(xh).to(dev)
    y = torch.full((A.info.yext_elems,), float("nan"), dtype=torch.float64, device=dev)
    A.spmv_device(x.data_ptr(), y.data_ptr(), torch.cuda.current_stream().cuda_stream)          # (an ordinary launch first)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        for _ in range(3):
            A.spmv_device(x.data_ptr(), y.data_ptr(), torch.cuda.current_stream().cuda_stream)
    y.fill_(float("nan"))
    g.replay()
    torch.cuda.synchronize()
    yref, absy = O.csr_spmv64(rp, ci, va, xh)
    _assert_close(y[:nrows].cpu().numpy(), yref, absy, TOL64, "graph replay")
    x[:ncols] *= 2.0                                    # the graph reads x at replay time
    g.replay()
    torch.cuda.synchronize()
    _assert_close(y[:nrows].cpu().numpy(), 2.0 * yref, 2.0 * absy, TOL64, "graph replay, new x")
    y.fill_(float("nan"))                               # and ordinary launches again, on the default stream and on a third one
    A.spmv_device(x.data_ptr(), y.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    _assert_close(y[:nrows].cpu().numpy(), 2.0 * yref, 2.0 * absy, TOL64, "after the graph")
    third = torch.cuda.Stream(device=dev)
    y.fill_(float("nan"))
    torch.cuda.synchronize()
    A.spmv_device(x.data_ptr(), y.data_ptr(), third.cuda_stream)
    torch.cuda.synchronize()
    _assert_close(y[:nrows].cpu().numpy(), 2.0 * yref, 2.0 * absy, TOL64, "after the graph, third stream")
    A.close()


def test_handles_release_their_device_memory():
    """create / preprocess / spmv / destroy in a loop (one image, column panels, tuning, device-resident input, a communicator):
    the free device memory comes back to where it was"""
    import torch
    nrows, ncols, rp, ci, va = CASES["power_law_3000"]
    dev = torch.device("cuda", 0)
    trp, tci, tva = (torch.from_numpy(a).to(dev) for a in (np.ascontiguousarray(rp, dtype=np.int64), np.ascontiguousarray(ci, dtype=np.int32), np.ascontiguousarray(va, dtype=np.float64)))
    x = O.x_vec_fast(ncols)

    def cycle():
        for kw in (dict(), dict(col_panels=3), dict(tune_steps=True), dict(value_dict=0, keep_csr=True)):
            A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, **kw)
            A.spmv(x)
            A.close()
        B = cvr_amd.CvrMatrix.from_device(nrows, ncols, trp.data_ptr(), tci.data_ptr(), tva.data_ptr())
        B.spmv(x)
        B.close()

    cycle()                                         # first use: code objects, RCCL, allocator pools
    comm = cvr_amd.Comm(cvr_amd.comm_unique_id(), 1, 0, 0)
    comm.close()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(dev)
    for _ in range(5):
        cycle()
    comm = cvr_amd.Comm(cvr_amd.comm_unique_id(), 1, 0, 0)
    comm.close()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(dev)
    assert free0 - free1 < (8 << 20), (free0, free1)          # 8 MiB of slack for driver-side pools


def test_independent_handles_on_concurrent_host_threads():
    """different handles may be used from different host threads at the same time (the error text is thread-local, every
    handle has its own stream and buffers); one handle is for one thread at a time, as documented in the header"""
    import threading
    names = ["power_law_3000", "uniform_2000", "two_giants", "dense_row_plus_singletons"]
    out, errs = {}, []

    def work(name):
        try:
            nrows, ncols, rp, ci, va = CASES[name]
            x = O.x_vec_fast(ncols)
            yref, absy = O.csr_spmv64(rp, ci, va, x)
            for _ in range(5):
                A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va)
                y, _ = A.spmv(x, iters=20)
                A.close()
                _assert_close(y, yref, absy, TOL64, ("threads", name))
            out[name] = True
        except Exception as e:          # noqa: BLE001 -- reported below with the case name
            errs.append((name, repr(e)))

    th = [threading.Thread(target=work, args=(n,)) for n in names]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs and len(out) == len(names), errs


def test_bench_line_keeps_the_driver_contract():
    """python bench.py prints ONE JSON line with the fields the driver and the judge read (metric .. config, roofline,
    cpu_baseline), the timed configuration passes its own verdict, and the roofline numbers are consistent"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "60", "--warmup", "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 60 and d["warmup"] == 5 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "GFLOP/s" and d["dtype"] == "f64" and "workload" in d["config"] and "model" not in d["config"]
    assert d["verdict_wrong_rows"] == 0
    assert abs(d["value"] - 2 * 5105039 / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.05 < rf["frac"] < 1.0
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["kernel_us"] * 1e-6) / 1e9) < 1e-6 * rf["achieved"]
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, (k, cb)
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0 and cb["cores"] >= 1


def test_minimal_c_example_builds_and_runs():
    """examples/minimal.c: a C99 program over the boundary, compiled with gcc on the GPU box, computes the 4 x 4 product"""
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("gcc"):
        pytest.skip("no gcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "minimal")
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", f"-I{root}/include", os.path.join(root, "examples", "minimal.c"),
                        f"-L{root}/cvr_amd", "-lcvr_amd", f"-Wl,-rpath,{root}/cvr_amd", "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert r.stdout.startswith("y = 201 0 5043 600"), r.stdout


def test_bench_starts_itself_for_two_ranks_on_one_device():
    """python bench.py --gpus 2 with no torch.distributed environment starts one rank per GPU itself (here both on cuda:0 over
    gloo, CVR_BENCH_ONE_DEVICE=1: the plumbing of the N > 1 path) and relays ONE line with n_gpus = 2 and a clean verdict"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CVR_BENCH_ONE_DEVICE="1", CVR_BENCH_NO_TUNE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3"], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["verdict_wrong_rows"] == 0 and d["scaling"] == "strong"
    assert len(d["config"]["rows_per_gpu"]) == 2 and sum(d["config"]["nnz_per_gpu"]) == 5105039


@pytest.mark.parametrize("workload", ["rmat20", "banded1e6"])
def test_bench_eight_ranks_on_one_device(workload, tmp_path):
    """The 8-rank path of configs[3] / [4] end to end on ONE device (CVR_BENCH_ONE_DEVICE=1: eight processes on cuda:0, gloo carries the
    exchange -- RCCL refuses two ranks on one GPU): device-built shards (R-MAT fp32, banded fp64), the cost-balanced cut, x replicated, y
    all-gathered.  Checks n_gpus, a clean in-run verdict on every rank's slice, the balance of the cut, and the gathered y of the last
    timed step (--dump-y) row by row against the pinned oracle on the matrix built in one piece.  (No 8-GPU box in this pool: the RCCL
    form of the same exchange has only ever run with one rank -- DESIGN.md section 6.)"""
    import json
    import subprocess
    import sys
    import torch
    from cvr_amd import synth_dev as D
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CVR_BENCH_ONE_DEVICE="1", CVR_BENCH_NO_TUNE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    ypath = str(tmp_path / "y.npy")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "10", "--warmup", "2", "--workload", workload, "--no-cpu-baseline",
                        "--dump-y", ypath], capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["verdict_wrong_rows"] == 0 and d["scaling"] == "strong"
    cfg = d["config"]
    assert len(cfg["rows_per_gpu"]) == 8 and len(cfg["nnz_per_gpu"]) == 8
    if workload == "rmat20":
        n, f32 = 1 << 20, True
        rp_t, ci_t, va_t = D.rmat_rows(20, 0, n, device="cuda")
        assert cfg["nnz_imbalance_max_over_mean"] < 1.35          # (the cut balances non-zeros + 1.25 per row, not non-zeros: R-MAT's last shard holds the short rows)
    else:
        n, f32 = 1_000_000, False
        rp_t, ci_t, va_t = D.banded_rows(n, 0, n, device="cuda")
        assert cfg["nnz_imbalance_max_over_mean"] < 1.01
    assert sum(cfg["rows_per_gpu"]) == n and sum(cfg["nnz_per_gpu"]) == int(rp_t[-1])
    assert d["gathered_slices_differing_between_ranks"] == 0          # every rank holds the same gathered vector (bench.py's own cross-rank check)
    rp, ci, va = rp_t.cpu().numpy(), ci_t.cpu().numpy(), va_t.cpu().numpy()
    # every rank built the slice of THIS matrix that the cut gives it (rows, non-zeros, column sum, value sum)
    b = np.concatenate([[0], np.cumsum(cfg["rows_per_gpu"])])
    for p in range(8):
        lo, hi = int(rp[b[p]]), int(rp[b[p + 1]])
        want = [int(b[p + 1] - b[p]), hi - lo, int(ci[lo:hi].astype(np.int64).sum()), int(np.round(va[lo:hi].astype(np.float64) * 1e6).astype(np.int64).sum())]
        assert d["shard_checksums"][p] == want, (p, d["shard_checksums"][p], want)
    del rp_t, ci_t, va_t
    torch.cuda.empty_cache()
    y = np.load(ypath)
    assert y.shape == (n,) and y.dtype == (np.float32 if f32 else np.float64)
    x = synth.x_rand(n, np.float32 if f32 else np.float64)
    yref, absy = O.csr_spmv64(rp, ci, va.astype(np.float64), x.astype(np.float64))
    bad, worst = O.tol_check(y.astype(np.float64), yref, absy, tol=1e-5 if f32 else 1e-12)
    per_slice = [int(np.count_nonzero((bad >= b[p]) & (bad < b[p + 1]))) for p in range(8)]
    assert len(bad) == 0, (len(bad), worst, "wrong rows per rank's slice", per_slice)


# the layout that came out best for each hold-out shape in the full sweep of tools/holdout.py (profiles/r06_holdout.log): what the automatic choice is held against
HOLDOUT_CONTENDER = {"webgoogle_seed7": "resident 7x48 window phases", "webgoogle_real": "measured (cvr_tune)", "lj_half": "16 panels interleaved", "lj_x2": "32 panels interleaved",
                     "road": "plain S=16", "citation": "1 image interleaved, gang", "rmat21b": "measured (cvr_tune)", "orkut_half": "8 panels interleaved",
                     "wikitalk_x2": "16 panels interleaved", "uniform16": "8 panels interleaved", "forum_sparse": "8 panels interleaved", "bipartite_sparse": "8 panels interleaved"}


@pytest.mark.parametrize("shape", sorted(HOLDOUT_CONTENDER))
def test_automatic_layout_on_held_out_shapes(shape):
    """The automatic layout on the twelve shapes its rules were not fitted on (tools/holdout.py: other seeds and other families -- road-network-like,
    citation-like, uniform random, a flatter R-MAT, real values, mostly-empty and bipartite matrices, the big stand-ins halved and doubled) against (1) the PLAIN
    layout -- never slower, up to timing noise -- and (2) the layout that won the full sweep of 15-25 explicit layouts for that shape (profiles/r06_holdout.log):
    within 8 % of it; the same y everywhere.  (Round 4: the citation-like shape ran 33 % slower than plain before the panel rule weighed the partial sums; round 5:
    the bipartite shape 3 % slower than plain and twice the time of 32 panels; round 6: the citation-like shape 21 % behind one interleaved image of gang chunks
    until the dropped-panels case took that layout.)"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import holdout as H
    n, nc, rp, ci, va = H.SHAPES[shape]()
    f32 = va.dtype == np.float32
    x = synth.x_rand(nc, va.dtype)
    yref, absy = O.csr_spmv64(rp, ci, va.astype(np.float64), x.astype(np.float64))
    cand = dict(H.candidates(n, nc, len(ci), 4 if f32 else 8))
    best_label = HOLDOUT_CONTENDER[shape]
    t, layout = {}, {}
    for label, kw in (("automatic", {}), ("plain", dict(H.PLAIN)), (best_label, cand[best_label])):
        A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, **kw)
        y, _ = A.spmv(x)
        _assert_close(y, yref, absy, TOL32 if f32 else TOL64, (shape, label))
        t[label] = min(A.bench(5, 50) for _ in range(3))
        i = A.info
        layout[label] = (i.steps_per_chunk, i.waves_per_block, i.col_panels, i.col_phases, i.x_window, i.hub_entries, i.interleave, i.gang, i.value_dict, i.nchunks)
        A.close()
    if layout["automatic"] != layout["plain"]:          # (else the rules chose the plain layout itself: two runs of one code path)
        assert t["automatic"] <= 1.03 * t["plain"], (shape, t, layout)
    if layout["automatic"] != layout[best_label]:
        assert t["automatic"] <= 1.08 * t[best_label], (shape, t, layout)


def test_bench_device_built_workload():
    """bench.py --workload rmat20: the matrix is built shard by shard on the GPU (cvr_amd/synth_dev.py), handed over as
    device-resident CSR, and the timed configuration is checked on the device against a torch fp64 segment sum"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "rmat20", "--steps", "20", "--warmup", "3", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert d["verdict_wrong_rows"] == 0 and d["dtype"] == "f32" and d["n_gpus"] == 1
    assert d["preprocess"]["workload_build_s"] < 20 and "R-MAT scale 20" in d["config"]["workload"]
    # the same device-built matrix against the oracle (bench.py itself may not use it)
    import torch
    from cvr_amd import synth_dev as D
    rp_t, ci_t, va_t = D.rmat_rows(20, 0, 1 << 20, device="cuda")
    A = cvr_amd.CvrMatrix.from_device(1 << 20, 1 << 20, rp_t.data_ptr(), ci_t.data_ptr(), va_t.data_ptr(), is_f32=True)
    x = synth.x_rand(1 << 20, np.float32)
    y, _ = A.spmv(x)
    yref, absy = O.csr_spmv64(rp_t.cpu().numpy(), ci_t.cpu().numpy(), va_t.cpu().numpy(), x)
    _assert_close(y, yref, absy + 1e-30, TOL32, "rmat20 device-built against the oracle")
    A.close()


@pytest.mark.parametrize("name", ["livejournal", "banded3.5e6", "rmat22", "orkut", "wikitalk"])
def test_full_size_shapes_every_row(name):
    """BASELINE.json configs[2] (soc-LiveJournal1 shape, full size, column panels), the nlpkkt240 shape at one GPU's share of
    8 (3.5 M rows, 94 M nnz), R-MAT-22 fp32 (67 M nnz) and the two further SuiteSparse-shaped power-law stand-ins (com-Orkut's shape,
    234 M nnz; wiki-Talk's, 5 M nnz with rows of 100 000): every row against the CSR oracle, bitwise equal reruns"""
    import time
    t0 = time.time()
    if name == "livejournal":
        nrows, ncols, rp, ci, va = synth.livejournal_like()
    elif name.startswith("banded"):
        nrows, ncols, rp, ci, va = synth.banded_sym(int(float(name[6:])))
    elif name in ("orkut", "wikitalk"):
        from cvr_amd import synth_dev as D
        nrows, rp_t, ci_t, va_t = (D.orkut_like if name == "orkut" else D.wikitalk_like)(device="cuda")
        ncols = nrows
        rp, ci, va = rp_t.cpu().numpy(), ci_t.cpu().numpy(), va_t.cpu().numpy()
        del rp_t, ci_t, va_t
    else:
        from cvr_amd import synth_dev as D
        import torch
        t = D.rmat_rows(22, 0, 1 << 22, device="cuda")
        nrows = ncols = 1 << 22
        rp, ci, va = (a.cpu().numpy() for a in t)
        del t
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va)
    f32 = va.dtype == np.float32
    x = synth.x_rand(ncols, va.dtype)
    y, _ = A.spmv(x)
    y2, _ = A.spmv(x)
    yref, absy = O.csr_spmv64(rp, ci, va, x)          # the pinned CSR oracle (spmv.cpp:1843-1850; fp32: accumulated in fp64), OpenMP over rows
    _assert_close(y, yref, absy, TOL32 if f32 else TOL64, name)
    assert np.array_equal(y.view(np.uint8), y2.view(np.uint8))
    if name == "livejournal":
        assert A.info.col_panels > 1
    if name == "wikitalk" and A.info.gang > 0:
        assert A.info.col_panels == 16          # (thin lists: fewer than two non-zeros of a gang per line of an eighth of x -- panels half as wide: cvr_panels.hip, thin_lists)
    A.close()
    assert time.time() - t0 < 180, "time box"


def test_amortisation_report_small():
    """paper Eq. 1 / Table 4 (reference: run_comparison.sh:20-45): CVR64 against the CSR comparators on the same GPU, every
    result checked against the oracle; I_pre with T_pre = planner + probe + dictionary scan + conversion, and with H2D"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import compare_csr as R
    nrows, ncols, rp, ci, va = synth.web_google_like(scale=0.1)
    out = R.report(nrows, ncols, rp, ci, va, iters=50, name="webgoogle/10")
    assert out["cvr"]["result_ok"]
    pre = out["cvr"]["preprocess_us"]
    assert pre["total"] >= pre["plan"] + pre["dict_scan"] + pre["convert_device_events"] and pre["total_with_h2d"] > pre["total"] - pre["dict_scan"]
    assert len(out["baselines"]) == 3
    for label, b in out["baselines"].items():
        assert b.get("result_ok"), (label, b)
        assert b["spmv_us"] > 0 and ("I_pre_iterations" in b)


def test_panelled_handle_on_two_streams_keeps_its_launches_apart():
    """a handle with column panels keeps the panels' partial sums in one buffer: launches on two streams with different x must
    not mix them (the library orders them with an event recorded on the stream it leaves, when the stream changes: run_spmv)"""
    torch = pytest.importorskip("torch")
    nrows, ncols, rp, ci, va = synth.livejournal_like(scale=0.05)
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=16, col_panels=4)
    assert A.info.col_panels == 4
    dev = torch.device("cuda", 0)
    xs, refs = [], []
    for seed in (1, 2):
        xh = np.random.default_rng(seed).random(ncols) * 2 - 1
        x = torch.zeros(A.info.x_elems, dtype=torch.float64, device=dev)
        x[:ncols] = torch.from_numpy(xh).to(dev)
        xs.append(x)
        refs.append(O.csr_spmv64(rp, ci, va, xh))
    ys = [torch.zeros(A.info.yext_elems, dtype=torch.float64, device=dev) for _ in range(2)]
    st = [torch.cuda.Stream(device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    for rep in range(30):
        for k in range(2):
            A.spmv_device(xs[k].data_ptr(), ys[k].data_ptr(), st[k].cuda_stream)
    torch.cuda.synchronize()
    for k in range(2):
        _assert_close(ys[k][:nrows].cpu().numpy(), refs[k][0], refs[k][1], TOL64, ("two streams", k))
    A.close()


@pytest.mark.parametrize("scale,S,wpb,win,hub,f32", [(12, 8, 1, 0, 500, False), (14, 32, 8, 0, 4000, True), (14, 16, 4, 256, 1000, False), (13, 32, 8, 1024, 100000, True)])
def test_hub_table(scale, S, wpb, win, hub, f32):
    """hub table (cvr_hub.hip): the columns with the most non-zeros staged in LDS, their slots carrying bit 30 and the table
    index -- image against the CPU mirror bit for bit (same ranking of the columns), y against the CSR oracle, bitwise reruns;
    with and without the value dictionary, with one chunk per workgroup and with persistent multi-chunk workgroups"""
    nrows, ncols, rp, ci, va = synth.rmat(scale, dtype=np.float32 if f32 else np.float64)
    for vd in (-1, 0):
        A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=S, waves_per_block=wpb, x_window=win, hub_table=hub, value_dict=vd, col_phases=0)
        i = A.info
        assert 0 < i.hub_entries <= hub and i.hub_share > 0.3
        mir = O.Cvr64(nrows, ncols, rp, ci, va, S, use_dict=i.value_dict > 0, hub_max=i.hub_entries, reorder=i.hub_reorder)
        assert mir.hub_n == i.hub_entries and i.hub_reorder == 0
        img = A.export_image()
        for key in ("desc", "target", "shared", "image"):
            assert np.array_equal(img[key], getattr(mir, key)), key
        x = synth.x_rand(ncols, va.dtype)
        yref, absy = O.csr_spmv64(rp, ci, va.astype(np.float64), x.astype(np.float64))
        y, _ = A.spmv(x)
        _assert_close(y, yref, absy, TOL32 if f32 else TOL64, (scale, S, wpb, win, hub, vd))
        y2, _ = A.spmv(x)
        assert np.array_equal(y.view(np.uint8), y2.view(np.uint8))
        A.close()


def test_hub_table_in_column_panels_and_automatic_choice():
    """device-resident CSR with column panels: every panel ranks the columns of its own range; and the automatic rule
    (a table only when the columns that fit the LDS hold at least half of the sampled non-zeros)"""
    torch = pytest.importorskip("torch")
    from cvr_amd import synth_dev as D
    scale = 18
    rp, ci, va = D.rmat_rows(scale, 0, 1 << scale, device="cuda")
    n = 1 << scale
    A = cvr_amd.CvrMatrix.from_device(n, n, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), is_f32=True, steps_per_chunk=16, col_panels=3, hub_table=2048)
    assert A.info.hub_entries == 2048
    A2 = cvr_amd.CvrMatrix(n, n, rp.cpu().numpy(), ci.cpu().numpy(), va.cpu().numpy(), steps_per_chunk=16, col_panels=1, hub_table=0)
    assert A.info.col_panels == 3
    x = synth.x_rand(n, np.float32)
    y, _ = A.spmv(x)
    y0, _ = A2.spmv(x)
    yref, absy = O.csr_spmv64(rp.cpu().numpy(), ci.cpu().numpy(), va.cpu().numpy().astype(np.float64), x.astype(np.float64))
    _assert_close(y, yref, absy, TOL32, "hub + panels")
    _assert_close(y0, yref, absy, TOL32, "no hub")
    # R-MAT 18 is too small for the automatic rule (4 M non-zeros, x of 1 MB): no table unless asked for
    A3 = cvr_amd.CvrMatrix.from_device(n, n, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), is_f32=True, steps_per_chunk=16, col_panels=3)
    assert A3.info.hub_entries == 0
    A3.close()
    A.close(); A2.close()


@pytest.mark.parametrize("f32", [False, True])
def test_narrow_chunks_store_16_bit_columns(f32):
    """banded matrix: every chunk spans fewer than 32 767 columns, so the image stores 16-bit column offsets (10 instead of 12
    bytes per fp64 slot): image against the mirror bit for bit, y against the CSR oracle; a matrix with one far column keeps
    32-bit column words; the value dictionary takes precedence over narrow columns"""
    nrows, ncols, rp, ci, va = synth.banded_sym(40_000, dtype=np.float32 if f32 else np.float64)
    A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=16)
    B = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=16, narrow_cols=0)
    assert A.info.narrow_cols == 1 and B.info.narrow_cols == 0 and A.info.image_bytes < 0.87 * B.info.image_bytes
    mir = O.Cvr64(nrows, ncols, rp, ci, va, 16, narrow=True)
    img = A.export_image()
    for key in ("desc", "target", "shared", "image"):
        assert np.array_equal(img[key], getattr(mir, key)), key
    x = synth.x_rand(ncols, va.dtype)
    yref, absy = O.csr_spmv64(rp, ci, va.astype(np.float64), x.astype(np.float64))
    ya, _ = A.spmv(x)
    yb, _ = B.spmv(x)
    _assert_close(ya, yref, absy, TOL32 if f32 else TOL64, "narrow")
    assert np.array_equal(ya.view(np.uint8), yb.view(np.uint8))            # same slots, same order of operations
    A.close(); B.close()
    ci2 = ci.copy()
    ci2[rp[100]] = 39_999 if ci2[rp[100]] != 39_999 else 0                  # one entry far from the band: that chunk is wide
    C2 = cvr_amd.CvrMatrix(nrows, ncols, rp, ci2, va, steps_per_chunk=16)
    assert C2.info.narrow_cols == 0
    y2, _ = C2.spmv(x)
    yref2, absy2 = O.csr_spmv64(rp, ci2, va.astype(np.float64), x.astype(np.float64))
    _assert_close(y2, yref2, absy2, TOL32 if f32 else TOL64, "wide chunk")
    C2.close()
    vd = np.where(np.arange(len(va)) % 2 == 0, 1.0, -0.5).astype(va.dtype)
    D2 = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, vd, steps_per_chunk=16)
    assert D2.info.value_dict > 0 and D2.info.narrow_cols == 0
    D2.close()


@pytest.mark.parametrize("f32", [False, True])
def test_hub_table_with_reordered_x(f32):
    """hub_reorder: every column index of the image is the column's popularity rank and the whole of x is re-ordered before
    every SpMV -- image against the mirror (same ranking), y against the CSR oracle, bitwise reruns, y equal to the table-only run"""
    nrows, ncols, rp, ci, va = synth.rmat(14, dtype=np.float32 if f32 else np.float64)
    x = synth.x_rand(ncols, va.dtype)
    yref, absy = O.csr_spmv64(rp, ci, va.astype(np.float64), x.astype(np.float64))
    ys = []
    for reo in (1, 0):
        A = cvr_amd.CvrMatrix(nrows, ncols, rp, ci, va, steps_per_chunk=16, waves_per_block=8, hub_table=3000, hub_reorder=reo, col_phases=0)
        i = A.info
        assert i.hub_reorder == reo and i.hub_entries == 3000
        mir = O.Cvr64(nrows, ncols, rp, ci, va, 16, use_dict=i.value_dict > 0, hub_max=3000, reorder=reo)
        img = A.export_image()
        for key in ("desc", "target", "shared", "image"):
            assert np.array_equal(img[key], getattr(mir, key)), (key, reo)
        y, _ = A.spmv(x)
        _assert_close(y, yref, absy, TOL32 if f32 else TOL64, ("reorder", reo))
        y2, _ = A.spmv(x)
        assert np.array_equal(y.view(np.uint8), y2.view(np.uint8))
        ys.append(y)
        A.close()
    assert np.array_equal(ys[0].view(np.uint8), ys[1].view(np.uint8))      # same slots, same order of operations
