#!/usr/bin/env python3
"""tools/holdout.py [shape ...] -- hold-out check of the AUTOMATIC layout (round-4 verdict, item 4).

The layout rules of cvr_create (resident layout with window and column phases, column panels one per XCD, interleaved chunks, hub tables,
chunk lengths that fill whole generations of workgroups) were fitted on the five shapes bench.py times.  This script builds shapes the
rules were NOT fitted on -- other seeds and other families -- and, for each, times the library's own choice against a small sweep of
explicit layouts (the plain layout, other chunk lengths, panels plain / interleaved, hub table, the measured layout of cvr_tune):

    regret = t(automatic) / min over the sweep (automatic included) - 1

Every candidate's y is checked against the library's host CSR loop (a layout that computes something else does not count).  Output: one
line per candidate and a table per shape -> profiles/r06_holdout.log (the bar: regret <= 8 % everywhere, and the automatic layout never
more than 10 % slower than the plain layout: tests/test_gpu_parity.py::test_automatic_layout_on_held_out_shapes asserts the latter).

  python tools/holdout.py                 all shapes
  python tools/holdout.py road citation   some
  HOLDOUT_SCALE=0.25 ...                  smaller versions (the GPU test)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (first: its wheel carries its own HIP runtime)

import cvr_amd  # noqa: E402
from cvr_amd import synth  # noqa: E402
from cvr_amd import synth_dev as D  # noqa: E402

SCALE = float(os.environ.get("HOLDOUT_SCALE", "1"))
DEV = "cuda" if torch.cuda.is_available() else "cpu"


def _gen(seed):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    return g


def _csr(rows, cols, n, vals=None):
    """COO on the device -> sorted, duplicate-free CSR on the host (values: pattern idx % 13 unless given as a function of nnz)"""
    key = torch.unique(rows.to(torch.int64) * n + cols.to(torch.int64))
    r = torch.div(key, n, rounding_mode="floor")
    c = (key - r * n).to(torch.int32)
    rp = torch.zeros(n + 1, dtype=torch.int64, device=key.device)
    rp[1:] = torch.cumsum(torch.bincount(r, minlength=n), 0)
    nnz = len(c)
    va = (np.arange(nnz, dtype=np.int64) % 13).astype(np.float64) if vals is None else vals(nnz)
    return n, n, rp.cpu().numpy(), c.cpu().numpy(), va


def road(seed=101):
    """road-network-like: a 2-D grid's neighbours (i +- 1, i +- w), each edge kept with probability 0.65: 2.6 non-zeros per row, all within
    +- w of the diagonal, real values (europe_osm / roadNet-CA are of this kind)"""
    n = int(6_000_000 * SCALE)
    w = int(np.sqrt(n))
    g = _gen(seed)
    i = torch.arange(n, device=DEV, dtype=torch.int64)
    rows, cols = [], []
    for off in (1, w):
        keep = torch.rand(n, generator=g, device=DEV) < 0.65
        a, b = i[keep], i[keep] + off
        ok = b < n
        a, b = a[ok], b[ok]
        rows += [a, b]
        cols += [b, a]
    rng = np.random.default_rng(seed)
    return _csr(torch.cat(rows), torch.cat(cols), n, vals=lambda m: rng.random(m) * 2 - 1)


def citation(seed=102):
    """citation-like (cit-Patents: 3.77 M x 3.77 M, 16.5 M non-zeros): document i cites ~4.4 EARLIER documents, half of them recent
    (exponential look-back), half by popularity among the earlier ones: strictly lower triangular, pattern values"""
    n = int(3_774_768 * SCALE)
    m = int(16_518_948 * SCALE * 1.03)
    g = _gen(seed)
    src = (n * torch.rand(m, generator=g, device=DEV, dtype=torch.float64) ** 0.8).to(torch.int64).clamp_(1, n - 1)
    back = (-torch.log1p(-torch.rand(m, generator=g, device=DEV, dtype=torch.float64) * 0.999999) * 30000.0).to(torch.int64) + 1
    recent = (src - back).clamp_(min=0)
    pop = (src.to(torch.float64) * torch.rand(m, generator=g, device=DEV, dtype=torch.float64) ** 2.2).to(torch.int64)
    dst = torch.where(torch.rand(m, generator=g, device=DEV) < 0.5, recent, pop).clamp_(min=0)
    dst = torch.minimum(dst, src - 1)
    return _csr(src, dst, n)


def uniform16(seed=103):
    """Erdos-Renyi-like: 2 M rows, 16 uniformly random columns each, real values (no locality, no hubs)"""
    n = int(2_000_000 * SCALE)
    g = _gen(seed)
    rows = torch.arange(n, device=DEV, dtype=torch.int64).repeat_interleave(16)
    cols = (n * torch.rand(len(rows), generator=g, device=DEV, dtype=torch.float64)).to(torch.int64).clamp_(0, n - 1)
    rng = np.random.default_rng(seed)
    return _csr(rows, cols, n, vals=lambda m: rng.random(m) * 2 - 1)


def rmat21b():
    """R-MAT scale 21 with (a, b, c, d) = (0.45, 0.22, 0.22, 0.11), fp32: flatter than the Graph500 parameters the hub-table rule saw"""
    scale = 21 if SCALE >= 1 else max(14, 21 + int(np.log2(SCALE)))
    rp, ci, va = D.rmat_rows(scale, 0, 1 << scale, a=0.45, b=0.22, c=0.22, seed=5, device=DEV)
    return 1 << scale, 1 << scale, rp.cpu().numpy(), ci.cpu().numpy(), va.cpu().numpy()


def webgoogle_seed7():
    return synth.web_google_like(scale=SCALE, seed=7)


def webgoogle_real():
    """the web-Google shape with real values: no value dictionary"""
    p = dict(synth.WEB_GOOGLE)
    n, nnz = max(64, int(p["n"] * SCALE)), max(64, int(p["nnz"] * SCALE))
    return synth.power_law_graph(n, nnz, p["empty_frac"], min(p["max_deg"], n // 2), 11, pattern_values=False)


def lj_half():
    return synth.livejournal_like(scale=0.5 * SCALE, seed=5)


def lj_x2():
    return synth.livejournal_like(scale=2.0 * SCALE, seed=6)


def orkut_half():
    old = dict(D.ORKUT)
    D.ORKUT["seed"] = 777
    try:
        n, rp, ci, va = D.orkut_like(scale=0.5 * SCALE, device=DEV)
    finally:
        D.ORKUT.update(old)
    return n, n, rp.cpu().numpy(), ci.cpu().numpy(), va.cpu().numpy()


def wikitalk_x2():
    old = dict(D.WIKITALK)
    D.WIKITALK["seed"] = 778
    try:
        n, rp, ci, va = D.wikitalk_like(scale=2.0 * SCALE, device=DEV)
    finally:
        D.WIKITALK.update(old)
    return n, n, rp.cpu().numpy(), ci.cpu().numpy(), va.cpu().numpy()


def forum_sparse(seed=104):
    """mostly-empty rows of another family than wiki-Talk: 3 M rows, 88 % empty, the others 8-40 uniformly random columns (no popular head)"""
    n = int(3_000_000 * SCALE)
    g = _gen(seed)
    act = torch.nonzero(torch.rand(n, generator=g, device=DEV) < 0.12).flatten()
    deg = (8 + 32 * torch.rand(len(act), generator=g, device=DEV)).to(torch.int64)
    rows = act.repeat_interleave(deg)
    cols = (n * torch.rand(len(rows), generator=g, device=DEV, dtype=torch.float64)).to(torch.int64).clamp_(0, n - 1)
    return _csr(rows, cols, n)


def bipartite_sparse(seed=105):
    """a bipartite rating-like matrix stored square: 4 M rows of which the first 8 % (the "users") hold everything, 25 entries each, columns among the
    other 92 % with a mild popularity skew (u ** 1.5), real values"""
    n = int(4_000_000 * SCALE)
    nu = max(1, int(0.08 * n))
    g = _gen(seed)
    rows = torch.arange(nu, device=DEV, dtype=torch.int64).repeat_interleave(25)
    cols = nu + ((n - nu) * torch.rand(len(rows), generator=g, device=DEV, dtype=torch.float64) ** 1.5).to(torch.int64).clamp_(0, n - nu - 1)
    rng = np.random.default_rng(seed)
    return _csr(rows, cols, n, vals=lambda m: rng.random(m) * 2 - 1)


SHAPES = dict(forum_sparse=forum_sparse, bipartite_sparse=bipartite_sparse, webgoogle_seed7=webgoogle_seed7, webgoogle_real=webgoogle_real, lj_half=lj_half, lj_x2=lj_x2, road=road, citation=citation, rmat21b=rmat21b,
              orkut_half=orkut_half, wikitalk_x2=wikitalk_x2, uniform16=uniform16)

PLAIN = dict(col_panels=1, col_phases=1, x_window=0, waves_per_block=1, hub_table=0, interleave=0)


def candidates(n, ncols, nnz, vbytes):
    xb = ncols * vbytes
    c = [("automatic", {}), ("plain", dict(PLAIN)), ("plain S=16", dict(PLAIN, steps_per_chunk=16)), ("plain S=32", dict(PLAIN, steps_per_chunk=32)),
         ("plain S=64", dict(PLAIN, steps_per_chunk=64)), ("hub table", dict(col_panels=1, hub_table=-1, waves_per_block=8, steps_per_chunk=32)),
         ("resident 7x48 window phases", dict(col_panels=1, waves_per_block=7, steps_per_chunk=48, x_window=12288, col_phases=16)),
         ("resident 8 phases no window", dict(col_panels=1, waves_per_block=8, col_phases=16, x_window=0)),
         ("measured (cvr_tune)", dict(tune_steps=True))]
    if xb >= 6e6:
        for P in (8, 16, 32):
            if xb / P >= 0.5e6:
                c.append((f"{P} panels plain", dict(col_panels=P, interleave=0)))
                c.append((f"{P} panels interleaved", dict(col_panels=P, interleave=1)))          # (gang chunks by the rule: four wavefronts on one sorted list)
                if P <= 16:
                    c.append((f"{P} panels interleaved, private", dict(col_panels=P, interleave=1, gang=0)))      # (round 4's chunks: a sorted list per wavefront)
                if P <= 16:
                    c.append((f"{P} panels interleaved, 2 wavefronts", dict(col_panels=P, interleave=1, waves_per_block=2)))
                    c.append((f"{P} panels interleaved, 4 wavefronts", dict(col_panels=P, interleave=1, waves_per_block=4)))
        c.append(("1 image interleaved", dict(col_panels=1, interleave=1)))
        c.append(("1 image interleaved, gang", dict(col_panels=1, interleave=1, gang=1)))
    return c


def run_shape(name, out):
    t0 = time.time()
    n, nc, rp, ci, va = SHAPES[name]()
    nnz = len(ci)
    f32 = va.dtype == np.float32
    vb = 4 if f32 else 8
    balg = synth.b_alg(n, nc, nnz, vb)
    x = synth.x_rand(nc, va.dtype)
    xh = x.astype(np.float64)
    yref = cvr_amd.csr_spmv_host(rp, ci, va.astype(np.float64), xh, nthreads=16)
    absy = cvr_amd.csr_spmv_host(rp, ci, np.abs(va).astype(np.float64), np.abs(xh), nthreads=16)
    tol = 1e-5 if f32 else 1e-12
    out(f"# {name}: {n} x {nc}, nnz {nnz} ({nnz / max(n, 1):.1f} per row), {'fp32' if f32 else 'fp64'}, x {nc * vb / 1e6:.1f} MB, B_alg {balg / 1e6:.1f} MB, built in {time.time() - t0:.1f} s")
    res = []
    for label, kw in candidates(n, nc, nnz, vb):
        try:
            A = cvr_amd.CvrMatrix(n, nc, rp, ci, va, **kw)
        except Exception as e:  # noqa: BLE001
            out(f"  {label:34s} refused: {str(e)[:120]}")
            continue
        y, _ = A.spmv(x)
        wrong = int(np.count_nonzero(np.abs(y.astype(np.float64) - yref) > tol * absy + (1e-30 if f32 else 1e-300)))
        s = A.bench(10, 100 if nnz > 50e6 else 200)
        i = A.info
        out(f"  {label:34s} {s * 1e6:9.2f} us  {balg / s / 8e12 * 100:5.1f} %  wrong {wrong}  S {i.steps_per_chunk} wpb {i.waves_per_block} panels {i.col_panels} ilv {i.interleave} "
            f"phases {i.col_phases} win {i.x_window} hub {i.hub_entries} ({i.hub_share:.2f}) dict {i.value_dict} chunks {i.nchunks} gang {i.gang}")
        A.close()
        if wrong == 0:
            res.append((label, s))
    auto = dict(res).get("automatic")
    plain = dict(res).get("plain")
    best = min(res, key=lambda r: r[1])
    rec = {"shape": name, "nrows": int(n), "nnz": int(nnz), "fp": 32 if f32 else 64, "automatic_us": auto * 1e6 if auto else None, "plain_us": plain * 1e6 if plain else None,
           "best": best[0], "best_us": best[1] * 1e6, "regret": auto / best[1] - 1 if auto else None, "automatic_over_plain": auto / plain if auto and plain else None,
           "frac_of_8TBs": balg / auto / 8e12 if auto else None}
    out("  -> " + json.dumps(rec))
    return rec


def main():
    names = [a for a in sys.argv[1:] if not a.startswith("-")] or list(SHAPES)
    log = open(os.environ.get("HOLDOUT_LOG", "/dev/null"), "a")

    def out(s):
        print(s, flush=True)
        log.write(s + "\n")
        log.flush()
    recs = [run_shape(nm, out) for nm in names]
    out("# shape                 automatic us   best us   regret   automatic / plain   best layout")
    for r in recs:
        out(f"# {r['shape']:20s} {r['automatic_us']:12.2f} {r['best_us']:9.2f} {100 * r['regret']:7.1f} % {r['automatic_over_plain']:12.3f}          {r['best']}")
    out(f"# max regret {100 * max(r['regret'] for r in recs):.1f} %, worst automatic / plain {max(r['automatic_over_plain'] for r in recs):.3f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
