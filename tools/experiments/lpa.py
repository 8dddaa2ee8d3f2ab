import sys, time, torch
sys.path.insert(0, '.')
from gnndelete_amd.framework.synth import dcsbm_edges
n, m = 235868, 1179052
E, comm = dcsbm_edges(n, m, 42)
src = torch.cat([E[0], E[1]]); dst = torch.cat([E[1], E[0]])

def lpa(src, dst, n, iters=8, seed=0):
    g = torch.Generator().manual_seed(seed)
    labels = torch.arange(n)
    for it in range(iters):
        key = dst * n + labels[src]
        uk, cnt = torch.unique(key, return_counts=True)
        d, l = uk // n, uk % n
        score = cnt * n + (n - 1 - l)
        best = torch.zeros(n, dtype=torch.long).scatter_reduce(0, d, score, 'amax', include_self=True)
        new = torch.where(best > 0, n - 1 - best % n, labels)
        upd = torch.rand(n, generator=g) < 0.5 if it < iters - 1 else torch.ones(n, dtype=torch.bool)
        labels = torch.where(upd, new, labels)
        nl = torch.unique(labels).numel()
        print(it, 'labels', nl, 'changed', int((new != labels).sum()))
    return labels

def locality(E, new_id, win):
    a, b = new_id[E[0]], new_id[E[1]]
    return float(((a - b).abs() < win).float().mean())

t = time.time(); lab = lpa(src, dst, n); print('lpa time', time.time() - t)
sizes = torch.bincount(lab); sizes = sizes[sizes > 0]
print('n labels', sizes.numel(), 'max size', int(sizes.max()), 'mean', float(sizes.float().mean()))
order = torch.argsort(lab * n + torch.arange(n)); new_id = torch.empty(n, dtype=torch.long); new_id[order] = torch.arange(n)
for win in (128, 1024, 4096):
    print('win', win, 'lpa', locality(E, new_id, win), 'random', locality(E, torch.arange(n), win))
o2 = torch.argsort(comm * n + torch.arange(n)); nid2 = torch.empty(n, dtype=torch.long); nid2[o2] = torch.arange(n)
print('oracle community order', [locality(E, nid2, w) for w in (128, 1024, 4096)])
