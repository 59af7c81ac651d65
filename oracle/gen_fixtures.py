#!/usr/bin/env python3
"""oracle/gen_fixtures.py -- TEST INFRASTRUCTURE, container-only.

Generates the golden fixtures under tests/golden/ by running the UNMODIFIED reference
(/root/reference/spmv.cpp, built by `make -C oracle ref` into oracle/_ref/) on small
Matrix-Market files authored here.  Only inputs (our own .mtx text) and the reference's
OUTPUT arrays are committed; no reference source enters the repo.

What is captured per matrix:
  * the reference loader's CSR arrays (readMatrix, spmv.cpp:311-535) -- pins the loader quirks
    Q1-Q9 (1-based indices, fp32-rounded values, idx%13 pattern values, pad-to-16, tail rowptr)
  * y of the reference's CSR loop (spmv.cpp:1843-1850) for x == 1 and for the seeded x
  * for T in {1,2,4}: the reference's 8-lane CVR arrays (pre_processing, spmv.cpp:565-1014) and
    the y its CVR kernel produced (spmv_compute_kernel, spmv.cpp:1016-1667), plus whether that
    y agrees with the CSR loop (it does not when bugs K1/K2 fire)

Run:  make -C oracle ref && python oracle/gen_fixtures.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
GOLD = os.path.join(ROOT, "tests", "golden")
MTX = os.path.join(GOLD, "mtx")


def write(name, text):
    with open(os.path.join(MTX, name), "w", newline="") as f:
        f.write(text)


def coo_text(header, nr, nc, entries, trailing_newline=True, comments=()):
    lines = [header] + ["%" + c for c in comments] + [f"{nr} {nc} {len(entries)}"]
    for e in entries:
        lines.append(" ".join(str(v) for v in e))
    t = "\n".join(lines)
    return t + ("\n" if trailing_newline else "")


def author_matrices():
    os.makedirs(MTX, exist_ok=True)
    rng = np.random.default_rng(20261002)
    G = "%%MatrixMarket matrix coordinate real general"
    P = "%%MatrixMarket matrix coordinate pattern general"
    PS = "%%MatrixMarket matrix coordinate pattern symmetric"
    RS = "%%MatrixMarket matrix coordinate real symmetric"
    IG = "%%MatrixMarket matrix coordinate integer general"

    # 1/2: 4x4 dense, 16 entries (nnz == 0 mod 16 -> no padding, Q9), values that are not fp32-exact (Q2)
    ent = [(i + 1, j + 1, f"{1.03 * (i * 4 + j + 1):.6f}") for i in range(4) for j in range(4)]
    ent = [ent[k] for k in rng.permutation(16)]
    write("dense4.mtx", coo_text(G, 4, 4, ent, comments=[" 4x4 dense, file order shuffled"]))
    write("dense4_nonl.mtx", coo_text(G, 4, 4, ent, trailing_newline=False))  # Q5: last line dropped

    # 3: 4x4 pattern symmetric (Q3 idx%13 values incl. mirrored entries, Q6 padding)
    ent = [(1, 1), (2, 1), (3, 1), (3, 2), (4, 2), (4, 4), (3, 3)]
    write("sym4_pattern.mtx", coo_text(PS, 4, 4, ent))

    # 4: 12 rows, skewed, empty rows (feed + steal + tail with T = 1)
    ent = []
    deg = [9, 0, 1, 14, 0, 0, 2, 1, 30, 3, 0, 5]
    for r, d in enumerate(deg):
        cs = rng.choice(40, size=d, replace=False) + 1
        for c in cs:
            ent.append((r + 1, int(c), f"{rng.uniform(-2, 2):.5f}"))
    ent = [ent[k] for k in rng.permutation(len(ent))]
    write("skew12.mtx", coo_text(G, 12, 40, ent))

    # 5: 9-row K2 trigger (rows of similar short length, T = 1)
    ent = []
    for r in range(9):
        for c in rng.choice(9, size=3 + (r % 3), replace=False) + 1:
            ent.append((r + 1, int(c), f"{rng.uniform(0.5, 1.5):.4f}"))
    write("k2_9rows.mtx", coo_text(G, 9, 9, ent))

    # 6: 2000-row power law, one ~2000-nnz row (K1 trigger at higher T), pattern -> idx%13 values
    ent = []
    n = 2000
    deg = np.minimum((rng.pareto(1.6, size=n) * 1.5).astype(int), 60)
    deg[rng.random(n) < 0.25] = 0
    deg[700] = 1990
    for r in range(n):
        for c in rng.choice(n, size=int(deg[r]), replace=False) + 1:
            ent.append((r + 1, int(c)))
    ent = [ent[k] for k in rng.permutation(len(ent))]
    write("pl2000_pattern.mtx", coo_text(P, n, n, ent))

    # 7: 300x200 real symmetric?  (symmetric requires square) -> 250x250 real symmetric, lower triangle
    ent = []
    n = 250
    for r in range(n):
        k = int(rng.integers(0, 6))
        for c in rng.choice(r + 1, size=min(k, r + 1), replace=False):
            ent.append((r + 1, int(c) + 1, f"{rng.normal():.7e}"))
    write("sym250_real.mtx", coo_text(RS, n, n, ent, comments=[" lower triangle only"]))

    # 8: rectangular integer general with duplicate-free entries, 64 x 300, trailing empty rows
    ent = []
    for r in range(50):
        for c in rng.choice(300, size=int(rng.integers(0, 12)), replace=False) + 1:
            ent.append((r + 1, int(c), int(rng.integers(-9, 10))))
    write("rect64x300_int.mtx", coo_text(IG, 64, 300, ent))

    # 9: single row, 1 x 50 with 37 entries; and 10: 96x96 diagonal-ish (every row 1-2 nnz)
    ent = [(1, int(c) + 1, f"{rng.uniform(-1, 1):.6f}") for c in rng.choice(50, size=37, replace=False)]
    write("onerow.mtx", coo_text(G, 1, 50, ent))
    ent = []
    for r in range(96):
        ent.append((r + 1, r + 1, f"{2 + 0.01 * r:.4f}"))
        if r % 3 == 0 and r + 1 < 96:
            ent.append((r + 1, r + 2, "-1.0"))
    write("diag96.mtx", coo_text(G, 96, 96, ent))


def rd(d, name, dt):
    return np.fromfile(os.path.join(d, name), dtype=dt)


def run_harness(mtx, T, xmode):
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, OMP_NUM_THREADS=str(T))
        r = subprocess.run([HARNESS, mtx, d, str(T), xmode], stdout=subprocess.DEVNULL,
                           stderr=subprocess.PIPE, env=env, timeout=120)
        if r.returncode != 0:
            raise RuntimeError(f"harness failed on {mtx} T={T}: rc={r.returncode} {r.stderr[-300:]}")
        man = dict(l.split() for l in open(os.path.join(d, "manifest.txt")))
        out = {
            "nItems": int(man["nItems"]), "numRows": int(man["numRows"]), "numCols": int(man["numCols"]),
            "csr_val": rd(d, "csr_val.f64", "<f8"), "csr_col": rd(d, "csr_col.i32", "<i4"),
            "csr_rowptr": rd(d, "csr_rowptr.i32", "<i4"), "x": rd(d, "x.f64", "<f8"),
            "y_csr": rd(d, "y_csr.f64", "<f8"), "y_cvr": rd(d, "y_cvr.f64", "<f8"),
            "cvr_val": rd(d, "cvr_val.f64", "<f8"), "cvr_col": rd(d, "cvr_col.i32", "<i4"),
            "cvr_record": rd(d, "cvr_record.i32", "<i4"), "cvr_split": rd(d, "cvr_split.i32", "<i4"),
            "cvr_final2": rd(d, "cvr_final2.i32", "<i4"), "cvr_nnz_rows": rd(d, "cvr_nnz_rows.i32", "<i4"),
            "sentinel": int(man["record_sentinel"]),
        }
        return out


def main():
    if not os.path.exists(HARNESS):
        sys.exit("build the reference first: make -C oracle ref")
    author_matrices()
    names = sorted(f for f in os.listdir(MTX) if f.endswith(".mtx"))
    for fn in names:
        path = os.path.join(MTX, fn)
        pack = {}
        base = None
        for xmode in ("ones", "rand"):
            for T in (1, 2, 4):
                # the reference's CVR kernel can race (K1); CSR part is deterministic
                try:
                    o = run_harness(path, T, xmode)
                except (RuntimeError, subprocess.TimeoutExpired) as e:
                    # the reference itself crashes/hangs when nnz < 16*T (zero-length chunks)
                    assert T > 1, e
                    if xmode == "ones":
                        pack[f"T{T}_ok"] = np.array([0], dtype=np.int64)
                    continue
                if xmode == "ones":
                    pack[f"T{T}_ok"] = np.array([1], dtype=np.int64)
                if base is None:
                    base = o
                    for k in ("csr_val", "csr_col", "csr_rowptr"):
                        pack[k] = o[k]
                    pack["dims"] = np.array([o["nItems"], o["numRows"], o["numCols"]], dtype=np.int64)
                else:
                    for k in ("csr_val", "csr_col", "csr_rowptr"):
                        assert np.array_equal(base[k], o[k]), (fn, k)
                if T == 1:
                    pack[f"x_{xmode}"] = o["x"]
                    pack[f"y_csr_{xmode}"] = o["y_csr"]
                if xmode == "ones":
                    for k in ("cvr_val", "cvr_col", "cvr_record", "cvr_split", "cvr_final2", "cvr_nnz_rows"):
                        pack[f"T{T}_{k}"] = o[k]
                    pack[f"T{T}_y_cvr_ones"] = o["y_cvr"]
                    d = np.abs(o["y_cvr"] - o["y_csr"])
                    pack[f"T{T}_cvr_agrees"] = np.array([int(np.all(d * d <= 1e-6))], dtype=np.int64)
                    pack["record_sentinel"] = np.array([o["sentinel"]], dtype=np.int64)
        outp = os.path.join(GOLD, fn.replace(".mtx", ".npz"))
        np.savez_compressed(outp, **pack)
        ag = [int(pack[f"T{T}_cvr_agrees"][0]) if pack[f"T{T}_ok"][0] else None for T in (1, 2, 4)]
        print(f"{fn:24s} nItems={pack['dims'][0]:6d} rows={pack['dims'][1]:5d} cols={pack['dims'][2]:5d} "
              f"ref-CVR-agrees(T=1,2,4)={ag}  -> {os.path.getsize(outp)} B")


if __name__ == "__main__":
    main()
