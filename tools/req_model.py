import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from cvr_amd import synth

def model(rp, ci, n, vbytes, panels_list, R_list, name):
    nnz = len(ci)
    rows = np.repeat(np.arange(len(rp)-1, dtype=np.int64), np.diff(rp))
    per_line = 128 // vbytes
    line = ci.astype(np.int64) // per_line
    print(f"{name}: n={n} nnz={nnz} x={n*vbytes/1e6:.1f}MB lines={n//per_line}")
    for P in panels_list:
        pw = (n + P - 1) // P
        pw = (pw + per_line - 1)//per_line*per_line
        panel = ci.astype(np.int64) // pw
        # sub-row index inside panel: rank of (panel,row) among unique pairs of that panel
        pr = panel * (len(rp)) + rows
        order = np.argsort(pr, kind='stable')
        prs = pr[order]
        newsub = np.ones(nnz, dtype=bool); newsub[1:] = prs[1:] != prs[:-1]
        subid = np.cumsum(newsub) - 1                      # global sub-row id in (panel,row) order
        npairs = int(subid[-1]) + 1
        # first subid of each panel
        pan_s = panel[order]
        firstsub = np.zeros(P + 1, dtype=np.int64)
        # subid at panel starts
        starts = np.searchsorted(pan_s, np.arange(P))
        firstsub[:P] = subid[np.minimum(starts, nnz-1)]
        local = subid - firstsub[pan_s]
        ln = line[order]
        for R in R_list:
            blk = local // R
            key = (pan_s * (1 << 22) + blk) * (1 << 24) + ln
            u = np.unique(key)
            print(f"  P={P:3d} (slice {pw*vbytes/1e6:.2f}MB) pairs={npairs/1e6:.1f}M R={R:6d}: req/nnz={len(u)/nnz:.3f}  ({len(u)/1e6:.1f}M lines)")
which = sys.argv[1]
t=time.time()
if which == 'lj':
    sc = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
    n, _, rp, ci, va = synth.livejournal_like(scale=sc)
    print('gen', time.time()-t)
    model(rp, ci, n, 8, [1, int(max(1,round(16*sc)))], [1280, 2048, 4096, 8192, 16384], f'lj x{sc}')
elif which == 'orkut':
    import torch
    from cvr_amd import synth_dev as D
    sc = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
    n, rp_t, ci_t, _ = D.orkut_like(scale=sc, device='cpu')
    rp, ci = rp_t.numpy(), ci_t.numpy()
    print('gen', time.time()-t)
    model(rp, ci, n, 8, [int(max(1,round(8*sc)))], [2048, 3400, 4096, 8192, 16384, 32768], f'orkut x{sc}')
elif which == 'rmat':
    s = int(sys.argv[2])
    n, _, rp, ci, va = synth.rmat(s)
    print('gen', time.time()-t)
    # popularity renumbering
    cnt = np.bincount(ci, minlength=n)
    rank = np.empty(n, dtype=np.int64); rank[np.argsort(-cnt, kind='stable')] = np.arange(n)
    model(rp, ci, n, 4, [1, 8], [2560, 4096, 8192, 16384, 32768], f'rmat{s} natural cols')
    model(rp, rank[ci].astype(np.int32), n, 4, [1, 8], [2560, 4096, 8192, 16384, 32768], f'rmat{s} popularity-ranked cols')
