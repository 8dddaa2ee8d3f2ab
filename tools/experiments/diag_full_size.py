"""Layer-by-layer HIP vs CPU oracle on the full-size requests (which stage carries a gap?)."""
import sys, torch
sys.path.insert(0, '.')
from types import SimpleNamespace
import bench
from oracle import gnndelete_ref as R
from oracle import pyg_semantics as pyg
rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())
what = sys.argv[1] if len(sys.argv) > 1 else 'gat'
dev = torch.device('cuda')
torch.set_num_threads(32)
if what == 'gat':
    args = SimpleNamespace(workload='synth-collab', gnn='gat', df='in', df_size=5.0, seed=42)
    data, model, neg, ni1, ni2 = bench.build_request(args, dev)
    E = data.train_pos_edge_index
    for name, ei in (('e_dr', E[:, data.dr_mask]), ('e_sdf', E[:, data.sdf_mask])):
        hip = model.to(dev)
        x = data.x
        c1, c2 = hip.conv1, hip.conv2
        with torch.no_grad():
            r1 = pyg.gat_conv(x, ei, c1.lin_src.weight.cpu(), c1.att_src.cpu(), c1.att_dst.cpu(), c1.bias.cpu())
            h1 = c1(x.to(dev), ei.to(dev).contiguous())
            print(name, 'conv1 out rel', rel(h1, r1), 'max abs', float((h1.cpu() - r1).abs().max()), 'ref norm', float(r1.norm()))
            inp = torch.relu(r1)
            r2 = pyg.gat_conv(inp, ei, c2.lin_src.weight.cpu(), c2.att_src.cpu(), c2.att_dst.cpu(), c2.bias.cpu())
            h2 = c2(inp.to(dev), ei.to(dev).contiguous())
            print(name, 'conv2 out rel', rel(h2, r2), 'max abs', float((h2.cpu() - r2).abs().max()), 'ref norm', float(r2.norm()))
            d = (h2.cpu() - r2).norm(dim=1) / r2.norm(dim=1).clamp_min(1e-20)
            worst = torch.topk(d, 5)
            print('  worst rows', worst.indices.tolist(), worst.values.tolist())
            n = x.shape[0]
            deg = torch.bincount(ei[1], minlength=n)
            print('  in-degree of the worst rows', deg[worst.indices].tolist(), 'row norms', r2[worst.indices].norm(dim=1).tolist())
            # pieces: linear, logits
            t_h = inp.to(dev) @ c2.lin_src.weight.t()
            t_r = inp @ c2.lin_src.weight.cpu().t()
            print('  linear rel', rel(t_h, t_r))
            m2 = data.sdf_node_2hop_mask
            print('  rel on S2 rows', rel(h2.cpu()[m2], r2[m2]), ' on the others', rel(h2.cpu()[~m2], r2[~m2]))
else:
    from gnndelete_amd.framework.models import RGCNDelete  # noqa
    from gnndelete_amd.framework.synth import make_kg_dataset
    data, _ = make_kg_dataset('synth-biokg', seed=42)
    n, nr = data.num_nodes, 51
    E, et = data.train_pos_edge_index, data.train_edge_type
    ei, ety = torch.cat([E, E.flip(0)], 1), torch.cat([et, et + nr])
    if what == 'rgcn_bwd':
        # the typed conv kernel alone, forward and transposed, against an fp64 per-relation loop on the device
        from gnndelete_amd.graph import TypedNodeCSR
        from gnndelete_amd import _lib
        from gnndelete_amd._lib import check, ptr, stream_ptr
        ei_d, et_d = ei.cuda(), ety.cuda()
        tg = TypedNodeCSR(ei_d, et_d, n, 2 * nr)
        torch.manual_seed(1)
        W = (torch.randn(2 * nr, 4, 32, 16, device='cuda') * 0.1).contiguous()      # conv2's block weights [R, nb, ib, ob]
        for trans in (0, 1):
            d_in, d_out = (128, 64) if not trans else (64, 128)
            xin = torch.randn(n, d_in, device='cuda')
            y = torch.zeros(n, d_out, device='cuda')
            node_ptr, seg_ptr, seg_rel, col, w = tg.bwd if trans else tg.fwd
            check(_lib.lib().gd_rgcn_conv_f32(ptr(node_ptr), ptr(seg_ptr), ptr(seg_rel), ptr(col), ptr(w), ptr(xin), xin.stride(0),
                                              d_in, ptr(W), 4, trans, ptr(y), y.stride(0), d_out, n, stream_ptr(xin.device)), 'k')
            src, dst = ei_d[0], ei_d[1]
            # forward weights: 1 / |N_r(i)|
            run = dst * (2 * nr) + et_d
            cnt = torch.bincount(run, minlength=n * 2 * nr).double()
            we = 1.0 / cnt[run]
            ref = torch.zeros(n, d_out, dtype=torch.float64, device='cuda')
            x64 = xin.double()
            for r in range(2 * nr):
                sel = (et_d == r).nonzero().flatten()
                if sel.numel() == 0:
                    continue
                Wr = torch.block_diag(*[W[r, b].double() for b in range(4)])         # [128, 64]
                if not trans:
                    ref.index_add_(0, dst[sel], (x64[src[sel]] * we[sel, None]) @ Wr)
                else:
                    ref.index_add_(0, src[sel], (x64[dst[sel]] * we[sel, None]) @ Wr.t())
            err = (y.double() - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-30)
            worst = torch.topk(err, 5)
            deg = torch.bincount(src if trans else dst, minlength=n)
            nruns = (node_ptr[1:] - node_ptr[:-1])
            print('trans', trans, 'rel', rel(y, ref), 'worst rows', worst.indices.tolist(), worst.values.tolist(),
                  'their degree', deg[worst.indices].tolist(), 'runs', nruns[worst.indices].tolist(), 'max degree', int(deg.max()))

        sys.exit(0)
    g = torch.Generator().manual_seed(5)
    m1, m2 = torch.rand(n, generator=g) < 0.3, torch.rand(n, generator=g) < 0.6
    torch.manual_seed(11)
    hip = RGCNDelete(SimpleNamespace(in_dim=128, hidden_dim=128, out_dim=64), n, nr, m1, m2)
    with torch.no_grad():
        for name, p in hip.named_parameters():
            if 'deletion_weight' in name:
                p.copy_(torch.eye(p.shape[0]) * 0.5 + torch.randn_like(p) * 0.05)
    ref = R.TwoLayerDelete('rgcn', 128, 128, 64, m1, m2, num_nodes=n, num_edge_type=nr)
    ref.load_state_dict(hip.state_dict(), strict=False)
    ref64 = R.TwoLayerDelete('rgcn', 128, 128, 64, m1, m2, num_nodes=n, num_edge_type=nr).double()
    ref64.load_state_dict({k: v.double() for k, v in hip.state_dict().items()}, strict=False)
    def loss_of(z1, z2, a, b):
        return (z1[a] ** 2).mean() + (z2[b] ** 2).mean()
    outs = {}
    for tag, m in (('f32', ref), ('f64', ref64)):
        z1, z2 = m(data.x, ei, ety, return_all_emb=True)
        z1.retain_grad()
        loss_of(z1, z2, m1, m2).backward()
        outs[tag] = (z1.detach(), z2.detach(), m.deletion1.deletion_weight.grad, m.deletion2.deletion_weight.grad, z1.grad)
    hip = hip.cuda()
    h1, h2 = hip(data.x.cuda(), ei.cuda(), ety.cuda(), return_all_emb=True)
    h1.retain_grad()
    loss_of(h1, h2, m1.cuda(), m2.cuda()).backward()
    got = (h1.detach(), h2.detach(), hip.deletion1.deletion_weight.grad, hip.deletion2.deletion_weight.grad, h1.grad)
    # the weight gradient recomputed in fp64 from the HIP path's own operands: is it the reduction kernel?
    with torch.no_grad():
        p1 = hip.conv1(hip.node_emb(data.x.cuda()), ei.cuda(), ety.cuda())
        idx = m1.cuda().nonzero().flatten()
        gw1_from_hip_ops = p1[idx].double().t() @ h1.grad[idx].double()
        print('gW1 kernel vs fp64 product of its own operands', rel(got[2], gw1_from_hip_ops))
        d = (got[4].double().cpu() - outs['f64'][4]).norm(dim=1)
        worst = torch.topk(d, 5)
        print('dz1 worst rows', worst.indices.tolist(), worst.values.tolist(), 'ref row norms', outs['f64'][4][worst.indices].norm(dim=1).tolist())
    for i, nm in enumerate(['z1', 'z2', 'gW1', 'gW2', 'dz1']):
        print(nm, 'hip vs f64', rel(got[i], outs['f64'][i]), ' cpu-f32 vs f64', rel(outs['f32'][i], outs['f64'][i]), ' hip vs cpu-f32', rel(got[i], outs['f32'][i]))
