"""Locality-preserving node renumbering for the SpMM kernels.

The gather side of a CSR SpMM re-reads a feature row once per incident edge; with arbitrary node
ids almost every one of those reads misses the 4 MB per-XCD L2 and goes out to the fabric
(measured on MI355X: 18 % L2 hit rate, 5.4x the algorithmic bytes).  Real graphs have community
structure, so renumbering nodes such that a community occupies a contiguous id range lets the
rows that one XCD processes at a time (gd_spmm_csr_balanced_f32 hands every XCD one contiguous
row range) share their neighbours' feature rows in that XCD's L2.

The order is found with a few rounds of semi-synchronous label propagation (every node adopts
the most frequent label among its neighbours; half of the nodes update per round to avoid
oscillation), all as sort/unique/scatter ops on the device.  It is an internal numbering of the
training engine only: inputs are permuted once at set-up, the Del weights it learns do not
depend on the numbering."""
import torch


def label_propagation(edge_index, num_nodes, iters=8, seed=0):
    n = int(num_nodes)
    dev = edge_index.device
    src = torch.cat([edge_index[0], edge_index[1]]).long()
    dst = torch.cat([edge_index[1], edge_index[0]]).long()
    gen = torch.Generator().manual_seed(seed)
    labels = torch.arange(n, device=dev)
    for it in range(iters):
        key = dst * n + labels[src]
        uniq, cnt = torch.unique(key, return_counts=True)
        node, lab = uniq // n, uniq % n
        # most frequent neighbour label, ties broken towards the smaller label
        score = cnt * n + (n - 1 - lab)
        best = torch.zeros(n, dtype=torch.long, device=dev).scatter_reduce(0, node, score, 'amax', include_self=True)
        proposal = torch.where(best > 0, n - 1 - best % n, labels)
        if it + 1 < iters:
            move = (torch.rand(n, generator=gen) < 0.5).to(dev)
            labels = torch.where(move, proposal, labels)
        else:
            labels = proposal
    return labels


def locality_order(edge_index, num_nodes, iters=8, seed=0):
    """-> (perm, inv): perm[new_id] = old_id, inv[old_id] = new_id; nodes with the same
    propagated label are contiguous, original order inside a label."""
    n = int(num_nodes)
    labels = label_propagation(edge_index, n, iters, seed)
    perm = torch.argsort(labels * n + torch.arange(n, device=labels.device))
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n, device=perm.device)
    return perm, inv
