#!/usr/bin/env python3
"""Benchmark of the GNNDelete hot path on MI355X: Del-operator training iterations per second.

One step = one pass of the reference's loop body (framework/trainer/gnndelete_nodeemb.py:188-299:
frozen-backbone forward on E[:, sdf_mask] -> Del -> DEC/NI losses -> Del-weight gradients ->
Adam) over the whole synthetic OGB-Collab-shaped graph (GCN, 5 % IN edge deletion, --loss_type
both_layerwise, mse_mean), all inputs resident in HBM.  Nothing is cached across steps: the
frozen X W1^T GEMM and both SpMMs are recomputed every iteration exactly as upstream does.

Prints ONE JSON line (see README / DESIGN.md for the field contract).  Launch for N > 1:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32-input matrix peak, v_mfma_f32_32x32x2_f32 (same guide)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def cpu_quota():
    """CPUs the container may actually use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown: the GPU boxes
    show 256 logical CPUs and run under a quota of 16 - what `cpu_baseline.cores` threads can really get."""
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        return None if q == 'max' else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=200)
    p.add_argument('--warmup', type=int, default=20)
    p.add_argument('--workload', default='synth-collab')
    p.add_argument('--gnn', default='gcn', choices=['gcn', 'gat', 'gin', 'sage', 'rgcn'],
                   help="'rgcn' = BASELINE config 4: --workload synth-biokg (51 relation types), full-graph fused Del step")
    p.add_argument('--df', default='in')
    p.add_argument('--df_size', type=float, default=5.0)
    p.add_argument('--loss_type', default='both_layerwise')
    p.add_argument('--seed', type=int, default=42)
    p.add_argument('--cpu_baseline_iters', type=int, default=20)
    p.add_argument('--pretrain_epochs', type=int, default=60,
                   help='original-model training epochs (HIP convs) before the request is served, so that the AUCs of '
                        'post_delete_auc are those of a trained backbone; never inside the timed region')
    p.add_argument('--no_cpu_baseline', action='store_true')
    p.add_argument('--no_graph', action='store_true')
    p.add_argument('--unroll', type=int, default=4, help='training iterations captured per hipGraph launch')
    p.add_argument('--no_cached_rate', action='store_true')
    p.add_argument('--repeats', type=int, default=0,
                   help='timed regions of exactly --steps steps each; the MEDIAN region is reported (0 = 5 when --steps < 100, '
                        'else 1): a 20-step region is 14 ms, shorter than the clock / power ramp of the part')
    p.add_argument('--stage_profile', default=None,
                   help='per-stage in-step kernel durations + PMC traffic from the committed rocprofv3 runs (tools/rocpd_stage_table.py); '
                        'default profiles/r06_final_stages.json (gcn) / profiles/r06_final_stages_<gnn>.json / ..._<workload>_<gnn>.json')
    p.add_argument('--probe_partition', action='store_true', help=argparse.SUPPRESS)   # child-process self test
    p.add_argument('--probe_overlap', action='store_true', help=argparse.SUPPRESS)     # ... of the overlapped exchanges
    p.add_argument('--parallel', default='auto', choices=['auto', 'partition', 'replicas'],
                   help='N>1: "auto" and "partition" = ONE request row-partitioned over the GPUs (RCCL halo all-to-all + '
                        'all-reduce per step) is the headline at every N, "scaling": "strong" - what north_star asks to be '
                        'measured; "auto" additionally reports the independent-replicas rate of the same GPUs and the '
                        "planner's estimate under extras / config.parallel_auto (it never changes the headline: a curve "
                        'must not mix strong- and weak-scaling points); "replicas" = every GPU serves its own unlearning '
                        'request (no data-path collective, "scaling": "weak") as the headline, on explicit request only')
    a = p.parse_args()
    if a.stage_profile is None:
        tag = ('' if a.gnn == 'gcn' else f'_{a.gnn}') if a.workload == 'synth-collab' else f"_{a.workload.replace('synth-', '').replace('-', '_')}_{a.gnn}"
        a.stage_profile = os.path.join(ROOT, 'profiles', f'r06_final_stages{tag}.json')
    return a


def build_request(args, device):
    """Synthetic dataset + unlearning request + randomly initialised frozen backbone."""
    from types import SimpleNamespace
    from gnndelete_amd.framework.data import prepare_edge_deletion, resolve_df_size
    from gnndelete_amd.framework.graph_utils import negative_sampling
    from gnndelete_amd.framework.models import GATDelete, GCNDelete, GINDelete, SAGEDelete
    from gnndelete_amd.framework.synth import make_linkpred_dataset
    from gnndelete_amd.framework.utils import seed_everything

    if args.gnn == 'rgcn':
        return build_kg_request(args)
    data, df_masks = make_linkpred_dataset(args.workload, seed=args.seed)
    seed_everything(args.seed)
    size = resolve_df_size(args.df_size, data.train_pos_edge_index.shape[1])
    prepare_edge_deletion(data, df_masks[args.df], size)
    margs = SimpleNamespace(in_dim=data.x.shape[1], hidden_dim=128, out_dim=64)
    cls = {'gcn': GCNDelete, 'gat': GATDelete, 'gin': GINDelete, 'sage': SAGEDelete}[args.gnn]
    model = cls(margs, data.sdf_node_1hop_mask, data.sdf_node_2hop_mask)
    neg = negative_sampling(data.train_pos_edge_index, data.num_nodes, int(data.df_mask.sum()))
    keep = torch.ones(data.num_nodes, dtype=torch.bool)
    keep[data.directed_df_edge_index.flatten().unique()] = False
    ni1, ni2 = data.sdf_node_1hop_mask & keep, data.sdf_node_2hop_mask & keep
    return data, model, neg, ni1, ni2


def build_kg_request(args):
    """BASELINE config 4: knowledge-graph unlearning request (delete_gnn.py:85-171, relational branch) on the synthetic
    ogbl-biokg stand-in, R-GCN with block-diagonal relation weights; Del masks = S_Df minus the Df endpoints, DEC on
    the forward-direction Df triples against head-shuffled negatives (gnndelete_nodeemb.py:749-768)."""
    from types import SimpleNamespace
    from gnndelete_amd.framework.data import prepare_edge_deletion, resolve_df_size
    from gnndelete_amd.framework.models import RGCNDelete
    from gnndelete_amd.framework.synth import KG_SHAPES, make_kg_dataset
    from gnndelete_amd.framework.utils import negative_sampling_kg, seed_everything
    data, df_masks = make_kg_dataset(args.workload, seed=args.seed)
    nr = KG_SHAPES[args.workload][1]
    seed_everything(args.seed)
    size = resolve_df_size(args.df_size, data.train_pos_edge_index.shape[1])
    prepare_edge_deletion(data, df_masks[args.df], size, True, nr)
    keep = torch.ones(data.num_nodes, dtype=torch.bool)
    keep[data.directed_df_edge_index.flatten().unique()] = False
    ni1, ni2 = data.sdf_node_1hop_mask & keep, data.sdf_node_2hop_mask & keep
    model = RGCNDelete(SimpleNamespace(in_dim=128, hidden_dim=128, out_dim=64), data.num_nodes, nr, ni1, ni2)
    pos, pt = data.edge_index[:, data.df_mask], data.edge_type[data.df_mask]
    fw = pt < nr
    data.kg_dec_edge, data.kg_num_edge_type = pos[:, fw].contiguous(), nr
    neg = negative_sampling_kg(data.kg_dec_edge, pt[fw])
    return data, model, neg, ni1, ni2


def gather_halo_bytes(eng, world, ctl):
    """[recv, send] bytes per step of every rank of a partitioned engine (all ranks call this; control group)."""
    import torch.distributed as dist
    rep = eng.halo_report()
    mine = torch.tensor([rep['recv_bytes_per_step'], rep['send_bytes_per_step']], dtype=torch.float64)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=ctl)
    return [[int(v) for v in t.tolist()] for t in every]


def layer1_share(stage_profile, default=0.29):
    """Share of the single-GPU step spent in layer 1 (the frozen product + its aggregation: the stages a partitioned rank
    recomputes on its halo rows), from the committed stage profile of THIS model (profiles/r*_final_stages[_gnn].json)."""
    try:
        with open(stage_profile) as f:
            st = json.load(f)['stages']
        tot = sum(v['in_step_us'] for v in st.values())
        names = list(st)
        first_del = next((i for i, k in enumerate(names) if k.startswith('del1')), None)      # ('del1' or the fused 'del1_loss_wgrad1')
        l1 = sum(st[k]['in_step_us'] for k in names[:first_del]) if first_del is not None else sum(st[k]['in_step_us'] for k in names[:2])
        return (l1 / tot, os.path.relpath(stage_profile, ROOT)) if tot > 0 else (default, 'default')
    except Exception:                                            # noqa: BLE001
        return default, 'default (no stage profile of this model)'


def partition_estimate(eng, world, ctl, single_us, link_gbs=100.0, l1_share=0.29, l1_source='default'):
    """Planner's prediction of the partitioned step (printed under config.parallel_auto, informational: the headline of
    `--parallel auto` is the partitioned step whatever it says): compute = the single-GPU step scaled by the rows this rank works on (layer 1 on own +
    halo rows, the rest on own rows) + ~60 us of segment launches; exchange = the heaviest rank's halo bytes over ONE xGMI
    link each way (a pair of GPUs shares one link; link_gbs = an assumed sustained rate, not a measurement: no multi-GPU
    run has been recorded) + ~30 us for the packed all-reduce.  MAX over the ranks."""
    import torch.distributed as dist
    rep = eng.halo_report()
    own, n = rep['own_rows'], eng.n
    compute = single_us * (l1_share * (own + rep['layer1_rows_recomputed']) / n + (1 - l1_share) * own / n) + 60.0
    # the exchanges move over one link per pair; the largest single transfer into this rank bounds it from below, all of it
    # over one link from above - take the per-rank total over (world - 1) links as the estimate, never below the largest pair
    recv = rep['recv_bytes_per_step']
    exchange = recv / max(1, world - 1) / (link_gbs * 1e3) * (1.0 if world > 2 else 1.0) + 30.0
    mine = torch.tensor([compute, exchange], dtype=torch.float64)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=ctl)
    comp, exch = max(float(t[0]) for t in every), max(float(t[1]) for t in every)
    return {'single_gpu_step_us': single_us, 'predicted_partitioned_step_us': comp + exch, 'compute_us': comp, 'exchange_us': exch,
            'assumed_link_gbs': link_gbs, 'layer1_share_of_step': l1_share, 'layer1_share_from': l1_source,
            'note': 'informational: the partitioned step is the headline at every N; link rate and launch overheads are assumptions, no multi-GPU run has been recorded'}


def planner_curve(eng, single_us, link_gbs=100.0, l1_share=0.29, worlds=(2, 4, 8)):
    """the planner's model of the row-partitioned step at EVERY world size of the scaling curve, from one
    run - per N the heaviest rank's halo bytes (both exchanges), the largest single pair, the layer-1 rows a rank recomputes
    for its halo, and the step the model of partition_estimate() predicts from them (compute = the measured single-GPU step
    scaled by rows worked on + launches, exchange = heaviest rank's bytes over its N - 1 links at an ASSUMED sustained
    link rate + the packed all-reduce).  Computed on the engine's replicated graph structure (the same halo_plan the engine
    itself uses, in the engine's internal node order); informational, never `value`."""
    from gnndelete_amd.collectives import halo_plan, row_blocks
    g, n = eng.graph, eng.n
    dev = g.rowptr.device
    m1 = torch.zeros(n, dtype=torch.bool, device=dev)
    sdf1 = eng.model.deletion1.mask.to(dev)
    m1[:] = sdf1[eng.perm] if getattr(eng, 'perm', None) is not None else sdf1
    row_f, row_b = 4 * getattr(eng, 'wf', eng.o), 4 * eng.o
    out = {}
    for world in worlds:
        chunk, _ = row_blocks(n, world)
        f = halo_plan(g.rowptr, g.col, n, 0, world, chunk)
        b = halo_plan(g.rowptr_t, g.col_t, n, 0, world, chunk, m1) if eng._mode != 'gat' else f
        pf, pb = torch.tensor(f.pair_counts), torch.tensor(b.pair_counts)          # [receiver, sender] rows
        recv = row_f * pf.sum(1) + row_b * pb.sum(1)
        pair = row_f * (pf + pf.t()) + row_b * (pb + pb.t())
        need1 = pf.sum(1)                                   # halo rows whose layer-1 output a rank recomputes
        own = torch.tensor([min(n, (q + 1) * chunk) - min(n, q * chunk) for q in range(world)])
        compute = (single_us * (l1_share * (own + need1).double() / n + (1 - l1_share) * own.double() / n) + 60.0).max()
        exchange = float(recv.max()) / max(1, world - 1) / (link_gbs * 1e3) + 30.0
        out[str(world)] = {'rows_per_rank': int(chunk), 'recv_bytes_per_step_max_rank': int(recv.max()),
                           'recv_bytes_per_step_every_rank': [int(v) for v in recv.tolist()], 'pair_bytes_per_step_max': int(pair.max()),
                           'layer1_rows_recomputed_max_rank': int(need1.max()), 'allreduce_bytes': 4 * (eng.h * eng.h + eng.o * eng.o + 4),
                           'predicted_step_us': float(compute) + exchange, 'predicted_compute_us': float(compute), 'predicted_exchange_us': exchange}
    return {'single_gpu_step_us': single_us, 'assumed_link_gbs': link_gbs, 'layer1_share_of_step': l1_share, 'per_world': out,
            'note': 'model only: the link rate and the launch overheads are assumptions - no multi-GPU run has been recorded; '
                    'compare predicted_step_us with the measured ms_per_step of the SCALE runs at the same N'}


def make_kg_engine(args, data, model, neg, ni1, ni2, device, rank=0, world=1, group=None, partition=False, **engine_opts):
    from gnndelete_amd.engine import NodeembEngine
    model = model.to(device)
    ei = data.edge_index[:, data.dr_mask].to(device).contiguous()
    et = data.edge_type[data.dr_mask].to(device).contiguous()
    x = data.x.to(device)
    with torch.no_grad():
        z1o, z2o = model.get_original_embeddings(x, ei, et, return_all_emb=True)
    common = (model, x, ei, z1o, z2o, data.kg_dec_edge.to(device), neg.to(device), ni1, ni2)
    if partition and world > 1:
        # ONE request, target rows partitioned over the ranks with typed halos (dist_engine, mode 'rgcn')
        from gnndelete_amd.dist_engine import PartitionedNodeembEngine
        return PartitionedNodeembEngine(*common, rank, world, loss_type=args.loss_type, alpha=0.5, lr=1e-3,
                                        use_graph=not args.no_graph, group=group, edge_type=et, overlap=getattr(args, 'dist_overlap', None))
    return NodeembEngine(*common, loss_type=args.loss_type, alpha=0.5, lr=1e-3, use_graph=not args.no_graph, edge_type=et, **engine_opts)


def time_typed_conv(eng):
    """The typed conv of layer 1 (128 -> 128, block-diagonal W_r) against BOTH roofs, SURVEY 8(d) conventions:
    flops = 2 * runs * d_in * d_out / n_blocks (one block-diagonal product per (node, relation) run - the mean aggregation
    itself is adds), against the dense fp32 MFMA peak; compulsory bytes = every operand once: x read (N d), y read-modify-
    written (2 N d: the root product is already in it), the re-sorted edge list (col + weight, 8 B/edge), the plan (8 B per
    piece + per-step scalars) and the packed relation weights - NOT the per-edge gathered rows (x is 48 MB and lives in the
    L2s / Infinity Cache; round 2 priced 4 nnz d gathered bytes here, which is not a roofline quantity)."""
    conv, tg = eng.model.conv1, eng.typed
    nnz, runs, n, d = int(tg.fwd[3].numel()), int(tg.fwd[1].numel()) - 1, eng.n, eng.h
    nb = conv.num_blocks or 1
    out = torch.empty(n, d, device=eng.x.device)
    from gnndelete_amd import ops
    out.zero_()
    dur = _avg_seconds(lambda: ops.rgcn_typed_accumulate(tg, eng.x, conv.weight.detach(), nb, 0, out), reps=10)
    dur_root = _avg_seconds(lambda: ops.rows_gemm(eng.x, None, conv.root.detach(), trans_w=False, bias=conv.bias.detach(), out=out), reps=10)
    tiled = os.environ.get('GD_RGCN_NODE_MAJOR') != '1'
    wave = tiled and ops.rgcn_wave_form(d, d, nb)
    plan = (tg.wave_plan(False) if wave else tg.tile_plan(False)) if tiled else {}
    flops = 2.0 * runs * d * d / nb
    n_pieces = plan.get('n_pieces', runs)
    if wave:      # x once, y read-modify-written, the unit plan (512 B of (source, weight) pairs + 64 B of slot words + 4 B per unit), weights
        nbytes = 4.0 * n * d * 3 + 580.0 * plan['n_units'] + 4.0 * conv.weight.numel()
    else:
        nbytes = 4.0 * n * d * 3 + 8.0 * nnz + 8.0 * n_pieces + 8.0 * plan.get('n_steps', 0) + 4.0 * conv.weight.numel()
    tf = flops / dur / 1e12
    traffic, traffic_file = None, 'profiles/r04_rgcn_wave_traffic.json' if wave else 'profiles/r03_rgcn_tile_traffic.json'
    try:            # PMC bytes per launch of the committed counter passes (same request, same kernel form only)
        with open(os.path.join(ROOT, traffic_file)) as f_:
            rec_ = json.load(f_)
        if (rec_['workload']['num_nodes'], rec_['workload']['typed_edges'], rec_['workload']['runs']) == (n, nnz, runs) and tiled:
            traffic = rec_['traffic_bytes_per_launch']
    except (OSError, KeyError, ValueError):
        pass
    if wave:
        kernel = ('rgcn_wave_kernel<32,32,64,1> (one wave per (64-node tile, diagonal block): units of 16 slots x 4 edges gathered a unit '
                  'ahead, typed mean aggregation into a wave-private LDS tile, block-diagonal transform on v_mfma_f32_16x16x4_f32, 128 -> 128)')
    elif tiled:
        kernel = ('rgcn_tile_kernel<128,32,32> ((64-node tile, relation) steps: typed mean aggregation into a compact LDS tile, '
                  'block-diagonal transform on v_mfma_f32_16x16x4_f32, 128 -> 128)')
    else:
        kernel = 'rgcn_conv_kernel (node-major typed mean aggregation + block-diagonal transform, 128 -> 128)'
    plan_keys = ('n_tiles', 'n_units', 'n_pieces', 'max_units') if wave else ('n_tiles', 'n_steps', 'n_pieces', 'n_hubs', 'n_slice_rows', 'max_steps')
    return {'kernel': kernel,
            'bound': 'mfma', 'achieved': tf, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf / MFMA_F32_PEAK_TFLOPS,
            'traffic': traffic, 'traffic_gbs': traffic / dur / 1e9 if traffic else None,
            'traffic_unit': f'bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE, {traffic_file}): the L2-miss side; '
                            'it equals the gathered-row volume - the fabric carries every gathered row',
            'algorithmic_flops': flops, 'avg_us': dur * 1e6,
            'hbm': {'compulsory_bytes': nbytes, 'gbs': nbytes / dur / 1e9, 'frac': nbytes / dur / 1e9 / HBM_PEAK_GBS},
            'gathered_rows_gbs': 4.0 * nnz * d / dur / 1e9,          # L2 / Infinity Cache side, informational
            'root_product_us': dur_root * 1e6, 'typed_edges': nnz, 'runs': runs,
            'note': (f'this launch: {tf / MFMA_F32_PEAK_TFLOPS:.2f} of the MFMA peak, {nbytes / dur / 1e9 / HBM_PEAK_GBS:.2f} of HBM on compulsory bytes'
                     + (f', fabric traffic (the gathered rows, all L2 misses) at {traffic / dur / 1e12:.2f} TB/s' if traffic else '')
                     + '; what bounds the kernel: DESIGN.md kernel table'),
            'plan': {k: plan[k] for k in plan_keys if k in plan}}


def kg_cpu_baseline(args, data, state, neg, ni1, ni2, iters):
    from oracle import gnndelete_ref as R
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    nr = data.kg_num_edge_type
    m = R.TwoLayerDelete('rgcn', 128, 128, 64, ni1, ni2, num_nodes=data.num_nodes, num_edge_type=nr)
    m.load_state_dict(state, strict=False)
    ei, et = data.edge_index[:, data.dr_mask], data.edge_type[data.dr_mask]
    with torch.no_grad():
        z1o, z2o = m.get_original_embeddings(data.x, ei, et, return_all_emb=True)
    targets = dict(z1_ori=z1o, z2_ori=z2o, pos_edge=data.kg_dec_edge, neg_edge=neg, ni_mask1=ni1, ni_mask2=ni2)
    opt = R.make_optimizer(m, args.loss_type, 1e-3)
    times = []
    for _ in range(iters):
        t0 = time.perf_counter()
        R.nodeemb_epoch(m, lambda: m(data.x, ei, et, return_all_emb=True), targets, opt, args.loss_type, 0.5, R.LOSSES['mse_mean'])
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {'value': 1.0 / med, 'unit': 'iters/s', 'cores': threads, 'cpu_quota': cpu_quota(), 'kind': 'port',
            'sample': f'{iters} full-graph R-GCN iterations of the same request, median ({med:.1f} s; torch CPU, {threads} host threads, '
                      f'cgroup quota {cpu_quota()} CPUs)'}, m


def kg_main(args, device, rank=0, world=1, group=None, barrier=lambda: None, ctl=None):
    """BASELINE config 4 (`--workload synth-biokg --gnn rgcn`): ONE JSON line with the same contract.  N > 1: the request's
    target rows are partitioned over the ranks (typed halo all-to-all + packed all-reduce per step, strong scaling) or, with
    --parallel replicas, every rank serves its own copy of the request."""
    data, model, neg, ni1, ni2 = build_kg_request(args)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    partitioned = world > 1 and args.parallel == 'partition'
    eng = make_kg_engine(args, data, model, neg, ni1, ni2, device, rank, world, group, partition=partitioned)
    run = getattr(eng, 'run', None)
    if run is not None and args.unroll > 1 and not args.no_graph:
        eng.prepare_unrolled(args.unroll)
    for _ in range(args.warmup):
        eng.step()

    def timed_region():
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if run is not None:
            run(args.steps, unroll=args.unroll)
        else:
            for _ in range(args.steps):
                eng.step()
        torch.cuda.synchronize()
        barrier()
        dt_ = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            tmax = torch.tensor([dt_], dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=ctl)
            dt_ = float(tmax)
        return dt_
    n_regions = args.repeats if args.repeats > 0 else (5 if args.steps < 100 else 1)
    region_s = [timed_region() for _ in range(n_regions)]
    dt = sorted(region_s)[len(region_s) // 2]
    units = args.steps if (partitioned or world == 1) else world * args.steps
    per_rank = gather_halo_bytes(eng, world, ctl) if partitioned else None
    if rank != 0:
        return
    if world > 1:
        out = {'metric': 'Del-op train iters/sec', 'value': units / dt, 'unit': 'iters/s', 'n_gpus': world, 'steps': args.steps,
               'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
               'scaling': 'strong' if partitioned else 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': f'{args.workload} R-GCN 2-layer, {args.df_size}% {args.df.upper()} triple deletion, full-graph fused Del step',
                          'num_nodes': data.num_nodes, 'hip_graph': not args.no_graph,
                          'parallelism': f'row-partition x{world} (typed halo all-to-all + all-reduce)' if partitioned else f'replicas x{world}',
                          'ranks_seen': world, 'halo': eng.halo_report() if partitioned else None,
                          'halo_recv_send_bytes_per_rank': per_rank},
               'timing': {'regions': n_regions, 'ms_per_step_each_region': [1e3 * t / args.steps for t in region_s]},
               'final_loss': float(eng.loss_history()[-1, 0])}
        print(json.dumps(out))
        return
    out = {'metric': 'Del-op train iters/sec', 'value': args.steps / dt, 'unit': 'iters/s', 'n_gpus': 1, 'steps': args.steps,
           'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak',
           'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
           'timing': {'regions': n_regions, 'ms_per_step_each_region': [1e3 * t / args.steps for t in region_s]},
           'config': {'workload': f'{args.workload} R-GCN 2-layer ({data.kg_num_edge_type} relation types, block-diagonal weights), '
                                  f'{args.df_size}% {args.df.upper()} triple deletion, full-graph fused Del step ({args.loss_type}, mse_mean)',
                      'num_nodes': data.num_nodes, 'typed_edges_dr': int(data.dr_mask.sum()), 'df_triples': int(data.directed_df_edge_index.shape[1]),
                      'S1': int(ni1.sum()), 'S2': int(ni2.sum()), 'hip_graph': not args.no_graph,
                      'matrix_products': matrix_products_label(), 'parallelism': 'single'},
           'roofline': time_typed_conv(eng), 'final_loss': float(eng.loss_history()[-1, 0])}
    if not args.no_cached_rate:
        # informational only (never `value`): the epoch as delete_gnn.py --fullgraph runs it by default - the frozen conv1 output
        # computed once, conv2's input gradient only on the Del-1 rows that read it (identical Del weights, tests/test_engine_gpu.py)
        out['extras'] = {}

        def after_ten(**opts):             # the Del weights ten iterations from the request's state
            model.load_state_dict(state)
            e_ = make_kg_engine(args, data, model, neg, ni1, ni2, device, **opts)
            for _ in range(10):
                e_.step()
            torch.cuda.synchronize()
            return model.deletion1.deletion_weight.detach().clone(), model.deletion2.deletion_weight.detach().clone()
        w_full = after_ten()
        rel_ = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
        for key, opts in (('iters_per_s_affected_rows_only', dict(affected_rows_only=True)),
                          ('iters_per_s_trainer_default', dict(affected_rows_only=True, cache_layer1=True))):
            w_opt = after_ten(**opts)
            out['extras'][key + '_W_rel_l2_vs_full_step_after_10_iterations'] = [rel_(w_opt[0], w_full[0]), rel_(w_opt[1], w_full[1])]
            model.load_state_dict(state)
            e2 = make_kg_engine(args, data, model, neg, ni1, ni2, device, **opts)
            out['extras'][key], out['extras'][key + '_ms_per_step_each_region'] = median_rate(e2, args)
            del e2
        model.load_state_dict(state)
    if not args.no_cpu_baseline:
        iters = max(1, min(args.cpu_baseline_iters, 2))
        cpu_data = data.clone().cpu() if hasattr(data, 'clone') else data
        out['cpu_baseline'], ref = kg_cpu_baseline(args, cpu_data, state, neg, ni1, ni2, iters)
        out['speedup_vs_cpu'] = out['value'] / out['cpu_baseline']['value']
        model.load_state_dict(state)
        eng2 = make_kg_engine(args, data, model, neg, ni1, ni2, device)
        for _ in range(iters):
            eng2.step()
        rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm())
        out['parity'] = {'iterations': iters, 'W_D1_rel_l2': rel(model.deletion1.deletion_weight.detach(), ref.deletion1.deletion_weight.detach()),
                         'W_D2_rel_l2': rel(model.deletion2.deletion_weight.detach(), ref.deletion2.deletion_weight.detach())}
    print(json.dumps(out))


def build_nodecls_request(args):
    """BASELINE config 5: delete_node.py's request (delete_node.py:77-142) - `df_size` % of the NODES deleted with every edge
    touching them, S_Df = 2-hop / 1-hop enclosing subgraph on the undirected edge_index - on the node-classification
    stand-in of the named shape (`synth-collab-nodecls`: 235,868 nodes, 4 classes; out_dim = #classes, delete_node.py:63-64)."""
    from types import SimpleNamespace
    from gnndelete_amd.framework.graph_utils import k_hop_subgraph, negative_sampling
    from gnndelete_amd.framework.models import GATDelete, GCNDelete, GINDelete, SAGEDelete
    from gnndelete_amd.framework.synth import make_nodecls_dataset
    from gnndelete_amd.framework.utils import seed_everything
    data = make_nodecls_dataset(args.workload[:-len('-nodecls')], seed=args.seed)
    n = data.num_nodes
    seed_everything(args.seed)
    df_nodes = torch.randperm(n)[:int(args.df_size / 100 * n)]
    gone = torch.zeros(n, dtype=torch.bool)
    gone[df_nodes] = True
    E = data.edge_index
    df_mask = gone[E[0]] | gone[E[1]]
    df_edge = E[:, df_mask]
    data.directed_df_edge_index = df_edge[:, df_edge[0] < df_edge[1]]
    seeds = df_edge.flatten().unique()
    _, e2, _, m2e = k_hop_subgraph(seeds, 2, E, num_nodes=n)
    _, e1, _, _ = k_hop_subgraph(seeds, 1, E, num_nodes=n)
    s1, s2 = torch.zeros(n, dtype=torch.bool), torch.zeros(n, dtype=torch.bool)
    s1[e1.flatten().unique()] = True
    s2[e2.flatten().unique()] = True
    data.sdf_node_1hop_mask, data.sdf_node_2hop_mask, data.sdf_mask, data.df_mask = s1, s2, m2e, df_mask
    data.dr_mask = data.dtrain_mask = ~df_mask
    data.train_pos_edge_index = E                      # (the oracle's loop reads the edges under this name)
    cls = {'gcn': GCNDelete, 'gat': GATDelete, 'gin': GINDelete, 'sage': SAGEDelete}[args.gnn]
    model = cls(SimpleNamespace(in_dim=data.x.shape[1], hidden_dim=128, out_dim=data.num_classes), s1, s2)
    neg = negative_sampling(E, n, int(df_mask.sum()))
    keep = torch.ones(n, dtype=torch.bool)
    keep[data.directed_df_edge_index.flatten().unique()] = False
    return data, model, neg, s1 & keep, s2 & keep


def time_gat_aggregation(eng):
    """The layer-1 attention-scored aggregation of the GAT step (d = 128: edge scores, softmax over a row's in-edges and the
    weighted gather in one kernel) back to back on the step's own graph.  Algorithmic bytes: rowptr + col + h read once + out
    written once + the two logit vectors + row max / row sum written (SURVEY 8d's SpMM formula without a value stream)."""
    from gnndelete_amd import ops
    g, n, h = eng.graph, eng.n, eng.h
    c = eng.model.conv1
    h1 = torch.randn(n, h, device=eng.x.device)
    a_src, a_dst = ops.row_dots(h1, c.att_src, c.att_dst)
    out = torch.empty_like(h1)
    dur = _avg_seconds(lambda: ops.gat_forward_raw(g, h1, a_src, a_dst, c.bias, c.negative_slope, out=out))
    nbytes = 4 * (n + 1) + 4 * g.nnz + 8 * n * h + 16 * n
    return {'kernel': 'gat_items_fwd_kernel (layer-1 attention-scored aggregation: scores, softmax, weighted gather; d=128)',
            'bound': 'hbm', 'achieved': nbytes / dur / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': nbytes / dur / 1e9 / HBM_PEAK_GBS,
            'traffic': None, 'algorithmic_bytes': nbytes, 'avg_us': dur * 1e6}


def nodecls_main(args, device, rank=0, world=1, barrier=lambda: None, ctl=None):
    """`--workload synth-collab-nodecls --gnn gat` (BASELINE config 5): steady-state Del epochs of the node-
    deletion request - one step = one epoch of GNNDeleteNodeClassificationTrainer's loop body (gnndelete_nodeemb.py:570-607),
    the whole graph, nothing cached across steps - with the same JSON contract.  N > 1: independent replicas (no collective)."""
    from gnndelete_amd.engine import NodeembEngine
    data, model, neg, ni1, ni2 = build_nodecls_request(args)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model = model.to(device)
    x, E = data.x.to(device), data.edge_index.to(device)
    e_sdf, e_dr = E[:, data.sdf_mask.to(device)].contiguous(), E[:, data.dr_mask.to(device)].contiguous()
    with torch.no_grad():
        z1o, z2o = model.get_original_embeddings(x, e_dr, return_all_emb=True)
    common = (model, x, e_sdf, z1o, z2o, E[:, data.df_mask.to(device)], neg.to(device), ni1, ni2)

    def engine(**opts):
        model.load_state_dict(state)
        return NodeembEngine(*common, loss_type='both_layerwise', alpha=0.5, lr=1e-3, use_graph=not args.no_graph, **opts)

    def rate(eng_, regions):
        if args.unroll > 1 and not args.no_graph:
            eng_.prepare_unrolled(args.unroll)
        for _ in range(args.warmup):
            eng_.step()
        out_ = []
        for _ in range(regions):
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng_.run(args.steps, unroll=args.unroll)
            torch.cuda.synchronize()
            barrier()
            dt_ = time.perf_counter() - t0
            if world > 1:
                import torch.distributed as dist
                tmax = torch.tensor([dt_], dtype=torch.float64)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=ctl)
                dt_ = float(tmax)
            out_.append(dt_)
        return out_
    n_regions = args.repeats if args.repeats > 0 else (5 if args.steps < 100 else 1)
    eng = engine()
    region_s = rate(eng, n_regions)
    dt = sorted(region_s)[len(region_s) // 2]
    if rank != 0:
        return
    out = {'metric': 'Del-op train iters/sec', 'value': world * args.steps / dt, 'unit': 'iters/s', 'n_gpus': world, 'steps': args.steps,
           'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak',
           'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
           'config': {'workload': f'{args.workload} {args.gnn.upper()} 2-layer, {args.df_size}% NODE deletion (delete_node.py), full-graph Del '
                                  f'epoch (both_layerwise, mse_mean)',
                      'num_nodes': data.num_nodes, 'in_dim': int(data.x.shape[1]), 'hidden_dim': 128, 'out_dim': int(data.num_classes),
                      'out_dim_padded_to': int(eng.o), 'df_nodes': int(args.df_size / 100 * data.num_nodes),
                      'df_edges_undirected': int(data.df_mask.sum()), 'sdf_edges': int(data.sdf_mask.sum()), 'spmm_nnz': eng.graph.nnz,
                      'S1': int(data.sdf_node_1hop_mask.sum()), 'S2': int(data.sdf_node_2hop_mask.sum()), 'backbone': 'random init',
                      'hip_graph': not args.no_graph, 'iterations_per_graph_launch': 1 if args.no_graph else args.unroll,
                      'matrix_products': matrix_products_label(), 'parallelism': 'single' if world == 1 else f'replicas x{world}',
                      'ranks_seen': world, 'chained_del1': bool(eng._chain1), 'fused_layer2': bool(eng._fuse_l2)},
           'timing': {'regions': n_regions, 'steps_per_region': args.steps, 'reported': 'median region',
                      'ms_per_step_each_region': [1e3 * t / args.steps for t in region_s]},
           'final_loss': float(eng.loss_history()[-1, 0])}
    if world > 1:
        print(json.dumps(out))
        return
    prof, prof_note = load_stage_profile(args.stage_profile, data.num_nodes, eng.graph.nnz)
    if args.gnn == 'gat':
        out['roofline'] = time_gat_aggregation(eng)
    else:
        kdur, kbytes = time_dominant_kernel(eng)
        out['roofline'] = {'kernel': 'spmm_persist_kernel<32,1,4,true,true> (layer-1 CSR SpMM, d=128)', 'bound': 'hbm', 'achieved': kbytes / kdur / 1e9,
                           'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': kbytes / kdur / 1e9 / HBM_PEAK_GBS, 'traffic': None,
                           'algorithmic_bytes': kbytes, 'avg_us': kdur * 1e6}
    first = next(iter(prof.get('stages', {}).values()), None) if prof else None
    out['roofline']['stage_profile'] = os.path.relpath(args.stage_profile, ROOT) if prof else prof_note
    out['extras'] = {'stage_rooflines': stage_table_from_profile(prof)}
    if not args.no_cached_rate:
        # informational only (never `value`): what the trainer runs by default (frozen layer-1 output computed once, only the rows
        # the request can influence), and the same full step WITHOUT the padded class dimension (generic-width layer-2 kernels)
        out['extras']['iters_per_s_trainer_default'] = median_rate(engine(cache_layer1=True, affected_rows_only=True), args)[0]
        if eng._user_wd2 is not None:
            os.environ['GD_PAD_OUT'] = '0'
            try:
                out['extras']['iters_per_s_unpadded_class_dimension'] = median_rate(engine(), args)[0]
            finally:
                os.environ.pop('GD_PAD_OUT')
    if not args.no_cpu_baseline:
        from oracle import gnndelete_ref as R
        threads = min(os.cpu_count() or 1, 32)
        torch.set_num_threads(threads)
        iters = max(2, min(args.cpu_baseline_iters, 6))
        ref = R.TwoLayerDelete(args.gnn, data.x.shape[1], 128, data.num_classes, data.sdf_node_1hop_mask, data.sdf_node_2hop_mask)
        ref.load_state_dict(state, strict=False)
        Ec = data.edge_index
        with torch.no_grad():
            r1o, r2o = ref.get_original_embeddings(data.x, Ec[:, data.dr_mask], return_all_emb=True)
        targets = dict(z1_ori=r1o, z2_ori=r2o, pos_edge=Ec[:, data.df_mask], neg_edge=neg, ni_mask1=ni1, ni_mask2=ni2)
        opt = R.make_optimizer(ref, 'both_layerwise', 1e-3)
        c_sdf = Ec[:, data.sdf_mask]
        times = []
        for _ in range(iters):
            t0 = time.perf_counter()
            R.nodeemb_epoch(ref, lambda: ref(data.x, c_sdf, return_all_emb=True), targets, opt, 'both_layerwise', 0.5, R.LOSSES['mse_mean'])
            times.append(time.perf_counter() - t0)
        med = sorted(times[1:])[len(times[1:]) // 2]
        out['cpu_baseline'] = {'value': 1.0 / med, 'unit': 'iters/s', 'cores': threads, 'cpu_quota': cpu_quota(), 'kind': 'port',
                               'sample': f'{iters - 1} full-graph epochs of the same request after 1 warm-up, median ({med:.2f} s; oracle '
                                         f'nodeemb_epoch = gnndelete_nodeemb.py:570-607 restated, torch CPU, {threads} of {os.cpu_count()} host threads, cgroup quota {cpu_quota()} CPUs)'}
        out['speedup_vs_cpu'] = out['value'] / out['cpu_baseline']['value']
        e2 = engine()
        for _ in range(iters):
            e2.step()
        torch.cuda.synchronize()
        rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm())
        with torch.no_grad():
            rr1, rr2 = ref(data.x, Ec[:, data.dr_mask], return_all_emb=True)
            hh1, hh2 = model(x, e_dr, return_all_emb=True)
        s1m, s2m = data.sdf_node_1hop_mask, data.sdf_node_2hop_mask
        acc = lambda z: float((z.argmax(1).cpu()[data.test_mask] == data.y[data.test_mask]).float().mean())
        out['parity'] = {'iterations': iters, 'W_D1_rel_l2': rel(model.deletion1.deletion_weight.detach(), ref.deletion1.deletion_weight.detach()),
                         'W_D2_rel_l2': rel(model.deletion2.deletion_weight.detach(), ref.deletion2.deletion_weight.detach()),
                         'z1_affected_rel_l2': rel(hh1[s1m.to(device)], rr1[s1m]), 'z2_affected_rel_l2': rel(hh2[s2m.to(device)], rr2[s2m]),
                         'test_accuracy_hip': acc(hh2), 'test_accuracy_cpu_oracle': acc(rr2),
                         'note': 'HIP engine vs CPU oracle from identical state, negatives and iteration count (random-init backbone)'}
    print(json.dumps(out))


def train_backbone(model, data, device, epochs, lr=0.01):
    """Original-model training (framework/trainer/base.py:75-142: BCE-with-logits link prediction on all training
    edges against fresh negatives every epoch, Adam) on the HIP convs - what train_gnn.py does before any unlearning
    request exists.  Untimed set-up: it gives the frozen backbone the request is served on."""
    import torch.nn.functional as F
    from gnndelete_amd.framework.graph_utils import negative_sampling
    model = model.to(device)
    x, E = data.x.to(device), data.train_pos_edge_index.to(device)
    params = [p for name, p in model.named_parameters() if 'deletion' not in name]
    opt = torch.optim.Adam(params, lr=lr)
    label = torch.cat([torch.ones(E.shape[1]), torch.zeros(E.shape[1])]).to(device)
    loss = None
    for _ in range(epochs):
        neg = negative_sampling(E, data.num_nodes, E.shape[1])
        z = model.get_original_embeddings(x, E)
        loss = F.binary_cross_entropy_with_logits(model.decode(z, E, neg), label)
        loss.backward()
        opt.step()
        opt.zero_grad()
    return float(loss.detach()) if loss is not None else None


def make_engine(args, data, model, neg, ni1, ni2, device, rank=0, world=1, group=None):
    from gnndelete_amd.engine import NodeembEngine
    model = model.to(device)
    x = data.x.to(device)
    E = data.train_pos_edge_index.to(device)
    e_sdf = E[:, data.sdf_mask.to(device)].contiguous()
    e_dr = E[:, data.dr_mask.to(device)].contiguous()
    with torch.no_grad():
        z1o, z2o = model.get_original_embeddings(x, e_dr, return_all_emb=True)
    common = (model, x, e_sdf, z1o, z2o, E[:, data.df_mask.to(device)], neg.to(device), ni1, ni2)
    global eng_args
    eng_args = common
    if world > 1 and args.parallel == 'partition':
        # ONE request, rows partitioned over the ranks (strong scaling)
        from gnndelete_amd.dist_engine import PartitionedNodeembEngine
        return PartitionedNodeembEngine(*common, rank, world, loss_type=args.loss_type, alpha=0.5, lr=1e-3,
                                        use_graph=not args.no_graph, group=group, overlap=getattr(args, 'dist_overlap', None))
    return NodeembEngine(*common, loss_type=args.loss_type, alpha=0.5, lr=1e-3, use_graph=not args.no_graph)


def replicas_rate(args, model, state, device, world, barrier):
    """Informational (extras): every rank serves the WHOLE request on its own GPU at the same time, no data-path
    collective - requests x iterations all ranks complete per second (max over ranks of the wall time)."""
    import torch.distributed as dist
    from gnndelete_amd.engine import NodeembEngine
    model.load_state_dict(state)
    eng = NodeembEngine(*eng_args, loss_type=args.loss_type, alpha=0.5, lr=1e-3, use_graph=not args.no_graph)
    if args.unroll > 1 and not args.no_graph:
        eng.prepare_unrolled(args.unroll)
    for _ in range(args.warmup):
        eng.step()
    barrier()
    t0 = time.perf_counter()
    eng.run(args.steps, unroll=args.unroll)
    torch.cuda.synchronize()
    barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return world * args.steps / float(t)


def matrix_products_label():
    """Which arithmetic `value` was measured with (gd_set_matrix_split / GD_MATRIX_SPLIT)."""
    from gnndelete_amd import ops
    return ('fp32 matrix instruction (v_mfma_f32_32x32x2_f32)' if ops.matrix_split() == 0 else
            'fp32 products from six exact bf16 partial products (v_mfma_f32_32x32x16_bf16), fp32 accumulation - GD_MATRIX_SPLIT=6')


def spmm_algorithmic_bytes(n, nnz, d):
    """rowptr + col + val + read X once + write Y once (fp32 / int32)."""
    return 4 * (n + 1) + 4 * nnz + 4 * nnz + 4 * n * d + 4 * n * d


def time_dominant_kernel(eng, reps=20):
    """Average duration of the layer-1 SpMM (d = 128), launched back to back on the current
    stream between two HIP events, with the same operands the step uses."""
    g = eng.graph
    h = eng.h
    t1 = torch.randn(eng.n, h, device=eng.x.device)
    y = torch.empty_like(t1)
    bias = eng.model.conv1.bias if hasattr(eng.model.conv1, 'bias') else None

    from gnndelete_amd import ops
    plan = getattr(eng, 'plan', None) or g.plan          # a rank's own rows when partitioned

    def launch():
        ops._spmm_raw(g.rowptr, g.col, g.val, t1, bias, 0.0, eng.n, plan, out=y)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    dur_s = e0.elapsed_time(e1) / 1e3 / reps
    rows = plan.n_items - plan.n_slots + plan.n_split      # rows this launch produces
    frac = rows / max(1, eng.n)
    return dur_s, int(spmm_algorithmic_bytes(eng.n, g.nnz, h) * frac) if frac < 0.999 else spmm_algorithmic_bytes(eng.n, g.nnz, h)


def time_dominant_kernel_in_context(eng, reps=20):
    """The same launch timed where it runs in the step - BEHIND the kernel that precedes it there (the frozen product x W1^T: a
    matrix-bound launch that leaves the part at its matrix-load clock and the caches holding t1's tail): HIP events around `reps`
    (x W1^T; SpMM) pairs minus the same loop of x W1^T alone.  Twenty back-to-back launches of the SpMM alone run 7-8 % faster than
    the launch does inside the replayed step (96 against 104 us: VERDICT r5 could not reconcile the bench line's fraction with the
    committed rocprofv3 table); this figure is the one `roofline.achieved` / `frac` are computed from.  GCN full step only (-> None
    otherwise: the caller keeps the back-to-back figure)."""
    from gnndelete_amd import ops
    if getattr(eng, '_mode', None) != 'gcn' or not hasattr(eng, '_linear') or getattr(eng, 'plan', None) is not None:
        return None
    g, c1 = eng.graph, eng.model.conv1
    y = torch.empty(eng.n, eng.h, device=eng.x.device)

    def first():
        return eng._linear(eng.x, c1.lin.weight)

    def pair():
        ops._spmm_raw(g.rowptr, g.col, g.val, first(), c1.bias, 0.0, eng.n, g.plan, out=y)
    t_pair, t_first = _avg_seconds(pair, reps), _avg_seconds(first, reps)
    return max(t_pair - t_first, 0.0)


def time_del_gemm(eng, reps=20):
    """Average duration of the layer-1 Del operator (row-subset GEMM over the S1 rows, d = 128) launched
    back to back between two HIP events with the step's own operands; flops = 2 S d^2."""
    from gnndelete_amd import ops
    if eng.s1 == 0:
        return None
    z = torch.randn(eng.n, eng.h, device=eng.x.device)
    out = torch.empty_like(z)
    w = eng.wd1.detach()

    def launch():
        ops.rows_gemm(z, eng.idx1, w, out=out)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    dur_s = e0.elapsed_time(e1) / 1e3 / reps
    flops = 2.0 * eng.s1 * eng.h * eng.h
    from gnndelete_amd import _lib
    ws = bool(_lib.lib().gd_rows_gemm_ws_covers(eng.s1, eng.h, eng.h)) and ops.matrix_split() == 0
    return {'kernel': ('rows_gemm_ws_kernel<128,128> (weight-stationary row GEMM)' if ws else 'rows_gemm_mfma_kernel<4,0>')
                      + ' - the STAND-ALONE Del-1 product on the S1 rows, d=128 (ops.rows_gemm, what DeletionLayer.forward runs outside '
                        'the engine); not a launch of the fused step, which forms Del-1 inside its Del-1 pass: see stage_rooflines.del1_loss_wgrad1',
            'bound': 'mfma',
            'achieved': flops / dur_s / 1e12, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': flops / dur_s / 1e12 / MFMA_F32_PEAK_TFLOPS, 'avg_us': dur_s * 1e6, 'rows': eng.s1,
            'hbm_gbs': 4.0 * (2 * eng.s1 * eng.h + eng.h * eng.h + eng.s1) / dur_s / 1e9}


def median_rate(eng, args, regions=None):
    """iters/s of an engine the way the headline is timed: warm-up, then `regions` regions of exactly
    --steps steps each (5 when --steps < 100: a 20-step region is shorter than the clock ramp of the part), the MEDIAN
    region.  -> (rate, [ms per step of every region])."""
    regions = regions or (args.repeats if args.repeats > 0 else (5 if args.steps < 100 else 1))
    if args.unroll > 1 and not args.no_graph and hasattr(eng, 'prepare_unrolled'):
        eng.prepare_unrolled(args.unroll)
    for _ in range(args.warmup):
        eng.step()
    ts = []
    for _ in range(regions):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if hasattr(eng, 'run') and not args.no_graph:
            eng.run(args.steps, unroll=args.unroll)
        else:
            for _ in range(args.steps):
                eng.step()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return args.steps / sorted(ts)[len(ts) // 2], [1e3 * t / args.steps for t in ts]


def _avg_seconds(launch, reps=20):
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 1e3 / reps


def time_wgrad(eng):
    """The W_D1 weight-gradient kernel (dW = pre1[S1]^T g[S1], reduction over the S1 rows, d = 128) with the step's own
    operands: flops 2 S d^2; bytes = the four [S1, d] operands it streams (conv output rows, Del output rows, folded
    targets, the layer-2 gradient) - it sits between the two roofs, so both fractions are reported."""
    from gnndelete_amd import ops
    if eng.s1 == 0 or not getattr(eng, '_fuse_loss1', False):
        return None
    if getattr(eng, '_fuse_del1', False):
        # round 5: the weight gradient comes out of the Del-1 pass itself (two products, four row streams: pre1 / targets / dh read, z1 written)
        dur = _avg_seconds(lambda: eng._del1_fused(eng.dh))
        flops = 4.0 * eng.s1 * eng.h * eng.h
        nbytes = 4.0 * 4 * eng.s1 * eng.h
        if getattr(eng, '_chain1', False):
            flops += 2.0 * eng.s1 * eng.o * eng.h
            nbytes = 4.0 * (3 * eng.s1 * eng.h + eng.s1 * eng.o)
        return {'kernel': ('del1_chain_ws_kernel (previous input gradient + ' if getattr(eng, '_chain1', False) else 'del1_loss_wgrad_ws_kernel<true> (')
                          + 'Del-1 forward + folded layer-1 loss + W_D1 gradient partial sums in one pass over the S1 rows, d=128)',
                'avg_us': dur * 1e6, 'tflops': flops / dur / 1e12, 'frac_mfma': flops / dur / 1e12 / MFMA_F32_PEAK_TFLOPS,
                'hbm_gbs': nbytes / dur / 1e9, 'frac_hbm': nbytes / dur / 1e9 / HBM_PEAK_GBS, 'rows': eng.s1}
    dur = _avg_seconds(lambda: eng._wgrad1(False, eng.dh))     # (steps the Del weights: the engine is discarded afterwards)
    flops = 2.0 * eng.s1 * eng.h * eng.h
    nbytes = 4.0 * 4 * eng.s1 * eng.h
    return {'kernel': 'rows_wgrad_mfma_kernel<4,4,true,8> (W_D1 gradient partial products, S1 rows, d=128; the reduction is part of the step tail)',
            'avg_us': dur * 1e6, 'tflops': flops / dur / 1e12, 'frac_mfma': flops / dur / 1e12 / MFMA_F32_PEAK_TFLOPS,
            'hbm_gbs': nbytes / dur / 1e9, 'frac_hbm': nbytes / dur / 1e9 / HBM_PEAK_GBS, 'rows': eng.s1}


def time_spmm_d64(eng):
    """The layer-2 aggregation (d = 64; the step runs it twice: forward and transposed) on the same algorithmic-bytes
    formula as the headline roofline entry."""
    from gnndelete_amd import ops
    g = eng.graph
    t2 = torch.randn(eng.n, eng.o, device=eng.x.device)
    y = torch.empty_like(t2)
    bias = getattr(eng.model.conv2, 'bias', None)
    plan = getattr(eng, 'plan', None) or g.plan
    dur = _avg_seconds(lambda: ops._spmm_raw(g.rowptr, g.col, g.val, t2, bias, 0.0, eng.n, plan, out=y))
    nbytes = spmm_algorithmic_bytes(eng.n, g.nnz, eng.o)
    return {'kernel': 'spmm_persist_kernel<16,1,4,true,true> (layer-2 CSR SpMM, d=64)', 'bound': 'hbm',
            'achieved': nbytes / dur / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': nbytes / dur / 1e9 / HBM_PEAK_GBS,
            'algorithmic_bytes': nbytes, 'avg_us': dur * 1e6}


def kernel_source_hash():
    """sha256 over the kernel sources and the C header (gnndelete_amd/source_hash.py): stamps a stage profile with the kernels it
    measured; the same hash is stamped into the library at build time (gd_build_source_hash)."""
    from gnndelete_amd.source_hash import source_hash
    return source_hash()


def load_stage_profile(path, n, nnz):
    """The committed per-stage table of the replayed step (tools/rocpd_stage_table.py over the rocprofv3 kernel-trace and
    --pmc passes of THIS command, profiles/): in-step durations and counter traffic.  -> (record, why_not): the record is
    only used when it was taken on this workload AND with the kernel sources of this tree (`csrc_sha`): numbers of other
    kernels are not attached to this run's entries."""
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return {}, 'no stage profile at ' + os.path.relpath(path, ROOT)
    w = rec.get('workload', {})
    if (w.get('num_nodes'), w.get('spmm_nnz')) != (n, nnz):
        return {}, 'stage profile is of another workload'
    if rec.get('csrc_sha') != kernel_source_hash():
        return {}, f"stage profile is stale: taken with kernel sources {rec.get('csrc_sha')}, this tree is {kernel_source_hash()}"
    return rec, None


def fabric_ceiling(n, nnz, d):
    """Upper bound of the HBM-roofline fraction ANY kernel with one contiguous row range per XCD can reach on this graph:
    algorithmic bytes / (traffic floor with 4 MiB LRU L2s / the fastest rate the XCD <-> fabric links were measured at).
    From the committed records (profiles/r02_spmm_traffic_floor.json: exact, CPU; profiles/r02_fabric_probe.txt: streams of
    >= 64 MB read by every XCD); None when they are of another graph."""
    import re
    try:
        with open(os.path.join(ROOT, 'profiles', 'r02_spmm_traffic_floor.json')) as f:
            fl = json.load(f)
        if (fl['n'], fl['nnz'], fl['d']) != (n, nnz, d):
            return None
        rates = [float(m.group(1)) for line in open(os.path.join(ROOT, 'profiles', 'r02_fabric_probe.txt'))
                 for m in [re.match(r'stream buf=(?:64|128|192) MB every XCD reads all of it.*?([0-9.]+) TB/s', line)] if m]
        return {'frac': fl['algorithmic_bytes'] / fl['lru_bytes'] * max(rates) * 1e3 / HBM_PEAK_GBS, 'fabric_tbs': max(rates),
                'floor_bytes_lru_4mib': fl['lru_bytes'], 'floor_bytes_infinite_l2': fl['floor_inf_l2_bytes'],
                'from': 'profiles/r02_spmm_traffic_floor.json, profiles/r02_fabric_probe.txt'}
    except (OSError, KeyError, ValueError):
        return None


def stage_table_from_profile(prof):
    """Models without a hand-written stage list (GAT, GraphSAGE, GIN): every kernel of the replayed step by position from the
    committed in-step profile of THIS model (tools/experiments/r05_profile.sh with GNN=...): duration, PMC traffic and the rate
    that traffic moved at against the HBM roof (a traffic-based fraction: these entries carry no algorithmic byte count)."""
    out = []
    for key, ps in (prof or {}).get('stages', {}).items():
        e = {'stage': key, 'kernel': ps.get('kernel'), 'in_step_us': ps['in_step_us'], 'bound': 'hbm (traffic-based)'}
        if ps.get('traffic_bytes'):
            gbs = ps['traffic_bytes'] / ps['in_step_us'] / 1e3
            e.update(traffic=ps['traffic_bytes'], traffic_gbs=gbs, frac_hbm_on_traffic=gbs / HBM_PEAK_GBS)
        out.append(e)
    return out or None


def stage_rooflines(eng, prof):
    """Every kernel of the GCN both_layerwise step against the roof that bounds it: algorithmic flops / bytes
    (each operand row counted once) over (a) the duration measured HERE with HIP events around 20 back-to-back launches on
    the step's own operands and (b) the in-step duration of the committed rocprofv3 trace (`in_step_us`; inside a graph
    replay a single kernel cannot be bracketed by events).  `traffic` = PMC bytes per launch from the committed --pmc passes.
    Mutates the engine's state (weight updates): call it last."""
    from gnndelete_amd import ops
    if eng._mode != 'gcn' or eng.loss_type != 'both_layerwise' or not (eng._split1 and eng._fuse_l2 and eng._fuse_loss1):
        return None
    n, f, h, o, s1, s2 = eng.n, eng.x.shape[1], eng.h, eng.o, eng.s1, eng.s2
    g, c1, c2 = eng.graph, eng.model.conv1, eng.model.conv2
    nnz = g.nnz
    dt2 = torch.randn(n, o, device=eng.x.device)
    y64 = torch.empty(n, o, device=eng.x.device)
    t1 = torch.randn(n, h, device=eng.x.device)
    t2 = torch.randn(n, o, device=eng.x.device)
    y128 = torch.empty(n, h, device=eng.x.device)
    spmm_b = lambda d: spmm_algorithmic_bytes(n, nnz, d)
    stages = [
        ('xw1', 'x W1^T (frozen, recomputed every step)', lambda: eng._linear(eng.x, c1.lin.weight), 2.0 * n * f * h, 4.0 * (n * f + f * h + n * h)),
        ('spmm1', 'pre1 = A t1 + b1 (d=128)', lambda: ops._spmm_raw(g.rowptr, g.col, g.val, t1, c1.bias, 0.0, n, g.plan, out=y128), 2.0 * nnz * h, spmm_b(h)),
    ] + ([
        # (round 5: Del-1, the layer-1 loss and the W_D1 weight gradient in ONE pass - two products, four row streams: pre1 and the
        #  targets and dh read, z1 written; the two launches below moved six)
        # (chained form: dh[S1] = (dt2[S1] W2) * [z1_prev[S1] > 0] of the previous iteration is formed in the pass too - three
        #  products; rows streamed: pre1, dt2 (64 wide) and the targets read, z1 written)
        ('del1_loss_wgrad1', 'dh[S1] = (dt2[S1] W2) * [z1_prev > 0] (previous iteration), z1[S1] = pre1[S1] W_D1 + sign bits, layer-1 loss sums, '
         'dW_D1 partial sums = pre1[S1]^T (coef (z1 - t) + dh)[S1]',
         lambda: eng._del1_fused(eng.dh), 4.0 * s1 * h * h + 2.0 * s1 * o * h, 4.0 * (3 * s1 * h + s1 * o + h * h + o * h) + 36.0 * s1),
    ] if getattr(eng, '_chain1', False) else [
        ('del1_loss_wgrad1', 'z1[S1] = pre1[S1] W_D1 + sign bits, layer-1 loss sums, dW_D1 partial sums = pre1[S1]^T (coef (z1 - t) + dh)[S1]',
         lambda: eng._del1_fused(eng.dh), 4.0 * s1 * h * h, 4.0 * (4 * s1 * h + h * h) + 20.0 * s1),
    ] if eng._fuse_del1 else [
        ('del1', 'z1[S1] = pre1[S1] W_D1 + sign bits', lambda: ops.rows_gemm(eng.pre1, eng.idx1, eng.wd1, out=eng.z1, sign_bits=eng.z1_pos),
         2.0 * s1 * h * h, 4.0 * (2 * s1 * h + h * h) + 20.0 * s1),
        ('wgrad1', 'dW_D1 partial products = pre1[S1]^T (coef (z1 - t) + dh)[S1] + layer-1 loss sums',
         lambda: eng._wgrad1(False, eng.dh), 2.0 * s1 * h * h, 4.0 * 4 * s1 * h),
    ]) + [
        ('t2', 't2 = relu(z1 | pre1) W2^T', lambda: eng._linear_relu_z1(c2.lin.weight), 2.0 * n * h * o, 4.0 * (n * h + n * o + h * o) + n),
        ('spmm2', 'p2 = A t2 + b2 (d=64)', lambda: ops._spmm_raw(g.rowptr, g.col, g.val, t2, c2.bias, 0.0, n, g.plan, out=y64), 2.0 * nnz * o, spmm_b(o)),
        # (three products and three row streams - p2 read, targets read, dp2 written - when the W_D2 weight gradient's partial
        #  sums come out of the same kernel; otherwise two products, four streams and a weight-gradient launch of its own)
        ('del2_loss_bwd', 'z2 = p2[S2] W_D2, layer-2 loss, dp2[S2] = dz2 W_D2^T' + (', dW_D2 partial sums = p2[S2]^T dz2' if eng._fuse_wg2 else ''),
         eng._del2_fused, (6.0 if eng._fuse_wg2 else 4.0) * s2 * o * o, 4.0 * (3 if eng._fuse_wg2 else 4) * s2 * o),
    ] + ([] if eng._fuse_wg2 else [
        ('wgrad2', 'dW_D2 partial products = p2[S2]^T dz2',
         lambda: eng._wgrad(eng.p2, eng.dz2c, None, s2, eng.g2, False, eng.ws2, a_idx=eng.idx2, adam=eng.adam2), 2.0 * s2 * o * o, 4.0 * 2 * s2 * o),
    ]) + [
        ('spmm2_t', 'dt2 = A^T dp2 (d=64)', lambda: ops._spmm_raw(g.rowptr_t, g.col_t, g.val_t, eng.dz2, None, 0.0, n, g.plan_t, out=y64), 2.0 * nnz * o, spmm_b(o)),
    ] + ([] if getattr(eng, '_chain1', False) else [
        ('dh', 'dh[S1] = (dt2[S1] W2) * [z1[S1] > 0]', lambda: ops.rows_gemm(dt2, eng.idx1, c2.lin.weight, trans_w=False, out=eng.dh, gate_bits=eng.z1_pos),
         2.0 * s1 * o * h, 4.0 * (s1 * o + s1 * h) + 16.0 * s1),
    ])
    ridge = MFMA_F32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)          # flop per byte where the two roofs meet
    pst = (prof or {}).get('stages', {})
    out = []
    for key, what, fn, flops, nbytes in stages:
        dur = _avg_seconds(fn)
        bound = 'mfma' if flops / nbytes > ridge and not key.startswith('spmm') else 'hbm'

        def fracs(us):
            tf, gb = flops / us / 1e6, nbytes / us / 1e3
            return {'tflops': tf, 'frac_mfma': tf / MFMA_F32_PEAK_TFLOPS, 'gbs': gb, 'frac_hbm': gb / HBM_PEAK_GBS,
                    'frac': tf / MFMA_F32_PEAK_TFLOPS if bound == 'mfma' else gb / HBM_PEAK_GBS}
        e = {'stage': key, 'what': what, 'bound': bound, 'algorithmic_flops': flops, 'algorithmic_bytes': nbytes,
             'avg_us': dur * 1e6, **fracs(dur * 1e6)}
        ps = pst.get(key)
        if ps:
            e['kernel'] = ps.get('kernel')
            e['in_step_us'] = ps['in_step_us']
            e['in_step'] = fracs(ps['in_step_us'])
            if ps.get('traffic_bytes'):
                e['traffic'] = ps['traffic_bytes']
                e['traffic_over_algorithmic'] = ps['traffic_bytes'] / nbytes
        out.append(e)
    return out


def cpu_baseline(args, data, model_state, neg, iters):
    """The CPU oracle (oracle/gnndelete_ref.py, the validated restatement of the reference's
    loop) timed on this box's host cores on the SAME request; a bounded sample of `iters` steps."""
    from oracle import gnndelete_ref as R
    # 32 threads is the fastest setting on the MI355X box's 256-thread host (measured 16/32/64/128/256:
    # 1.68 / 1.56 / 2.23 / 3.68 / 25.5 s per iteration - the scatter-adds do not scale further)
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    m = R.TwoLayerDelete(args.gnn, data.x.shape[1], 128, 64, data.sdf_node_1hop_mask, data.sdf_node_2hop_mask)
    m.load_state_dict(model_state, strict=False)
    d = {k: v for k, v in data.items()}
    ni1, ni2 = R.non_df_masks(data.num_nodes, data.directed_df_edge_index, data.sdf_node_1hop_mask,
                              data.sdf_node_2hop_mask)
    E = data.train_pos_edge_index
    with torch.no_grad():
        z1o, z2o = m.get_original_embeddings(data.x, E[:, data.dr_mask], return_all_emb=True)
    targets = dict(z1_ori=z1o, z2_ori=z2o, pos_edge=E[:, data.df_mask], neg_edge=neg, ni_mask1=ni1, ni_mask2=ni2)
    opt = R.make_optimizer(m, args.loss_type, 1e-3)
    e_sdf = E[:, data.sdf_mask]

    def fwd():
        return m(data.x, e_sdf, return_all_emb=True)
    R.nodeemb_epoch(m, fwd, targets, opt, args.loss_type, 0.5, R.LOSSES['mse_mean'])     # warm-up
    times = []
    for _ in range(iters):
        t0 = time.perf_counter()
        R.nodeemb_epoch(m, fwd, targets, opt, args.loss_type, 0.5, R.LOSSES['mse_mean'])
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    rec = {'value': 1.0 / med, 'unit': 'iters/s', 'cores': threads, 'cpu_quota': cpu_quota(), 'kind': 'port', 'mode': 'faithful',
           'sample': f'{iters} full-graph iterations of the same request after 1 warm-up, median '
                     f'({med:.2f} s; min {min(times):.2f}, max {max(times):.2f}; torch CPU, {threads} of {os.cpu_count()} '
                     f'host threads, cgroup quota {cpu_quota()} CPUs - the fastest setting, see thread_sweep). Faithful = the frozen layer 1 (x W1^T and its '
                     f'aggregation) recomputed every iteration as upstream does; the >= 10x target is quoted against this one'}
    # the *fair* mode next to it (loop-invariant layer-1 output computed once), and the thread
    # sweep that justifies `cores`; bounded samples on copies of the model so that `m` keeps the state parity is checked on
    import copy
    import torch.nn.functional as F

    def one_mode(nthreads, cached, n_it):
        torch.set_num_threads(nthreads)
        mm = copy.deepcopy(m)
        oo = R.make_optimizer(mm, args.loss_type, 1e-3)
        if cached:
            with torch.no_grad():
                p1 = mm.conv1(data.x, e_sdf)

            def f():
                x1 = mm.deletion1(p1)
                return x1, mm.deletion2(mm.conv2(F.relu(x1), e_sdf))
        else:
            def f():
                return mm(data.x, e_sdf, return_all_emb=True)
        ts = []
        for _ in range(n_it):
            t0 = time.perf_counter()
            R.nodeemb_epoch(mm, f, targets, oo, args.loss_type, 0.5, R.LOSSES['mse_mean'])
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]
    if iters >= 4:
        fair = one_mode(threads, True, 5)
        rec['fair'] = {'value': 1.0 / fair, 'unit': 'iters/s', 'cores': threads,
                       'sample': f'5 iterations, median ({fair:.2f} s), layer-1 output cached (what the trainer also does)'}
        sweep = {}
        for nt in (16, 64):
            if nt <= (os.cpu_count() or 1):
                sweep[str(nt)] = one_mode(nt, False, 2)
        sweep[str(threads)] = med
        rec['thread_sweep_s_per_iter'] = sweep
        torch.set_num_threads(threads)
    return rec, m, iters + 1


def post_delete_parity(args, data, model, state, neg, ni1, ni2, device, cpu_model, n_iters):
    """The metric's second half ("+ post-delete AUC"; north_star: AUC within +-0.002, affected-node embeddings
    within 1e-4 rel-L2 on identical seeds): the HIP engine runs the SAME `n_iters` iterations the CPU oracle just
    ran, from the same state with the same negatives; both models then embed the graph on the retained edges
    (evaluation semantics of framework/trainer/base.py:238-249) and score the test edges."""
    from gnndelete_amd.framework.metrics import batched_roc_auc
    model.load_state_dict(state)
    eng = make_engine(args, data, model, neg, ni1, ni2, device)
    for _ in range(n_iters):
        eng.step()
    E = data.train_pos_edge_index
    e_dr = E[:, data.dr_mask]
    with torch.no_grad():
        r1, r2 = cpu_model(data.x, e_dr, return_all_emb=True)
        h1, h2 = model(data.x.to(device), e_dr.to(device).contiguous(), return_all_emb=True)

    def rel(a, b):
        return float((a.double().cpu() - b.double()).norm() / b.double().norm())

    def auc(z, pos, neg_e):
        ei = torch.cat([pos, neg_e], 1).to(z.device)
        score = (z[ei[0]] * z[ei[1]]).sum(-1).sigmoid()
        label = torch.cat([torch.ones(pos.shape[1]), torch.zeros(neg_e.shape[1])]).to(z.device)
        return float(batched_roc_auc(score, label)[0])
    m1, m2 = data.sdf_node_1hop_mask, data.sdf_node_2hop_mask
    a_hip = auc(h2, data.test_pos_edge_index, data.test_neg_edge_index)
    a_cpu = auc(r2, data.test_pos_edge_index, data.test_neg_edge_index)
    # Df-vs-Dr AUC (base.py:263-277: deleted edges labelled 0, an equally large random subset of Dr labelled 1)
    g = torch.Generator().manual_seed(0)
    k = data.directed_df_edge_index.shape[1]
    dr_sub = e_dr[:, torch.randperm(e_dr.shape[1], generator=g)[:k]]
    f_hip, f_cpu = auc(h2, dr_sub, data.directed_df_edge_index), auc(r2, dr_sub, data.directed_df_edge_index)
    return {'iterations': n_iters, 'test_auc_hip': a_hip, 'test_auc_cpu_oracle': a_cpu, 'abs_diff': abs(a_hip - a_cpu),
            'df_auc_hip': f_hip, 'df_auc_cpu_oracle': f_cpu, 'df_abs_diff': abs(f_hip - f_cpu),
            'tolerance': 0.002, 'z1_affected_rel_l2': rel(h1[m1.to(device)], r1[m1]),
            'z2_affected_rel_l2': rel(h2[m2.to(device)], r2[m2]), 'embedding_tolerance': 1e-4,
            'note': 'backbone trained on the HIP convs before the request (pretrain_epochs, bench config); HIP engine vs '
                    'CPU oracle on identical state, negatives and iteration count'}


def probe_partition_in_child(args, rank, overlap=False):
    """Run a few partitioned steps of a small request in CHILD processes (one per rank, rendezvous on
    their own port) under a hard timeout, before this process touches the GPU or RCCL.  A collective
    that hangs inside RCCL cannot be caught as an exception - but a child can be killed.  -> (ok, note)
    overlap: the children run the program with the exchanges on the communication stream AND the synchronous one from the
    same state and compare the two (Del weights and loss history bit for bit): the parent only switches the overlap on
    when this second probe passes."""
    import subprocess
    env = dict(os.environ, MASTER_PORT=str(int(os.environ.get('MASTER_PORT', '29500')) + (41 if overlap else 23)))
    env.pop('TORCHELASTIC_USE_AGENT_STORE', None)      # the children's rank 0 hosts its own rendezvous store
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', str(args.gpus), '--probe_partition', '--workload',
           'synth-small', '--df', 'in', '--df_size', '5', '--gnn', args.gnn, '--loss_type', args.loss_type]
    if overlap:
        cmd.append('--probe_overlap')
    what = 'overlapped-exchange self-test' if overlap else 'partition self-test'
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        if r.returncode == 0:
            return 1, None
        tail = [l for l in r.stdout.strip().splitlines() if l.strip()][-1:] or ['']
        return 0, f'{what} failed on rank {rank} (exit {r.returncode}): {tail[0][:160]}'
    except subprocess.TimeoutExpired:
        return 0, f'{what} timed out on rank {rank} (300 s)'


def launch_ranks(n):
    """`python bench.py --gpus N` without a torch.distributed environment: start the N ranks ourselves, exactly as the
    driver does (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`), as a CHILD
    process started before this process has made any GPU call (never an exec of a process that has touched the GPU);
    its output is relayed and its exit code returned."""
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    rc = 1
    for attempt in range(3):
        # a port that is free NOW can be taken before torchrun binds it (two bench runs started together by a scale script):
        # a launch that dies on the rendezvous address before any rank printed a line is retried on a fresh port
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print(f'bench.py: --gpus {n} without WORLD_SIZE in the environment - launching {n} ranks: {" ".join(cmd[1:8])} ...', file=sys.stderr, flush=True)
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        out, err = proc.communicate()
        sys.stdout.write(out)
        sys.stdout.flush()
        sys.stderr.write(err[-4000:])
        rc = proc.returncode
        in_use = 'EADDRINUSE' in err or 'Address already in use' in err or 'address already in use' in err
        if rc == 0 or not in_use or out.strip():
            break
    return rc


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world} (start one rank per GPU, or drop WORLD_SIZE and let bench.py launch them)'
    # GD_BENCH_BACKEND=gloo lets the N>1 code path be exercised on a one-GPU box (all ranks on cuda:0)
    backend = os.environ.get('GD_BENCH_BACKEND', 'nccl')
    auto = args.parallel == 'auto'
    if auto:
        # north_star: the 1-D partitioned step is what the scaling curve measures - tried first; kept as the headline only
        # where the planner predicts it to beat the single-GPU step (partition_estimate), else independent replicas
        args.parallel = 'partition'
    mode, note, probe_ok, overlap_ok, overlap_note = (('single' if world == 1 else args.parallel), None, 1, 0,
                                                      'the default; GD_DIST_OVERLAP=1 asks for the overlapped program, which then has to pass a self-test')
    force_probe = os.environ.get('GD_BENCH_FORCE_PROBE') == '1'          # lets the gloo test exercise the probe
    if world > 1 and mode == 'partition' and (backend == 'nccl' or force_probe) and not args.probe_partition:
        probe_ok, note = probe_partition_in_child(args, rank)      # before any GPU / RCCL initialisation here
        # The exchanges run under compute only on request (GD_DIST_OVERLAP=1) AND when a child-process self-test reproduces the
        # synchronous program's results bit for bit.  Not by default: the overlapped program has never run on a second device, and
        # a self-test at a small size cannot rule out a timing-dependent hazard at the bench's size.
        if probe_ok and os.environ.get('GD_DIST_OVERLAP') == '1' and args.gnn != 'rgcn':      # (R-GCN: synchronous exchanges)
            overlap_ok, overlap_note = probe_partition_in_child(args, rank, overlap=True)
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    if backend != 'nccl' and torch.cuda.device_count() < world:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    ctl = group = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        # control plane (flags, barriers, max-over-ranks of the wall time): gloo on the host, so that it
        # keeps working whatever state the data-path communicator is in
        dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=600))
        ctl = dist.group.WORLD
        flag = torch.tensor([probe_ok, overlap_ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=ctl)
        if mode == 'partition' and not int(flag[0]):
            mode = args.parallel = 'replicas'
            note = note or 'partition self-test failed on another rank'
        # exchanges under compute only where every rank's probe reproduced the synchronous result with them
        args.dist_overlap = bool(int(flag[1])) if not args.probe_partition else None
        if mode == 'partition' and backend == 'nccl':
            group = dist.new_group(backend='nccl', timeout=datetime.timedelta(seconds=300), device_id=device)

    def barrier():
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier(group=ctl)

    if args.gnn == 'rgcn':
        if args.probe_partition:
            args.workload, args.steps, args.warmup, args.repeats = 'synth-kg-small', 3, 1, 1
        kg_main(args, device, rank, world, group, barrier, ctl)
        if world > 1:
            dist.destroy_process_group()
        return
    if args.workload.endswith('-nodecls'):
        nodecls_main(args, device, rank, world, barrier, ctl)
        if world > 1:
            dist.destroy_process_group()
        return
    data, model, neg, ni1, ni2 = build_request(args, device)
    pretrain_loss = None
    if args.pretrain_epochs > 0 and not args.probe_partition:
        pretrain_loss = train_backbone(model, data, device, args.pretrain_epochs)       # set-up, not timed
        torch.cuda.synchronize()
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    if mode == 'partition':
        # build the partitioned engine and take one step; if ANY rank fails (RCCL set-up, capture),
        # every rank falls back to independent replicas so that the run still reports a number
        ok = 1
        try:
            eng = make_engine(args, data, model, neg, ni1, ni2, device, rank, world, group)
            eng.step()
            torch.cuda.synchronize()
        except Exception as e:                                   # noqa: BLE001
            ok, note = 0, f'{type(e).__name__}: {str(e)[:160]}'
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=ctl)
        if not int(flag):
            mode = args.parallel = 'replicas'
            note = note or 'partitioned engine failed on another rank'
            model.load_state_dict(state)
    if args.probe_partition:
        for _ in range(3):
            eng.step()
        barrier()
        assert bool(torch.isfinite(eng.loss_history()).all())
        if args.probe_overlap:
            # the same four steps from the same state with the exchanges on the communication stream: identical results
            ref = (model.deletion1.deletion_weight.detach().clone(), model.deletion2.deletion_weight.detach().clone(),
                   eng.loss_history().clone())
            model.load_state_dict(state)
            args.dist_overlap = True
            if os.environ.get('GD_PROBE_POISON') == '1':
                # diagnostic: what the allocator hands the second engine for its torch.empty workspaces is NaN - a kernel that
                # reads a workspace element nobody wrote shows up as a difference on every run instead of one in forty
                junk = torch.full((1 << 28,), float('nan'), device=device)
                del junk
            eng2 = make_engine(args, data, model, neg, ni1, ni2, device, rank, world, group)
            assert eng2._async
            for _ in range(4):
                eng2.step()
            barrier()
            got = (model.deletion1.deletion_weight.detach(), model.deletion2.deletion_weight.detach(), eng2.loss_history())
            same = all(torch.equal(a.nan_to_num(), b.nan_to_num()) for a, b in zip(ref, got))      # Del weights AND loss log, bit for bit
            worst = max(float((a.nan_to_num().double() - b.nan_to_num().double()).norm() / a.nan_to_num().double().norm().clamp(min=1e-30))
                        for a, b in zip(ref, got))
            where = ''
            for name_, a, b in zip(('W_D1', 'W_D2', 'loss history'), ref, got):
                dif = (a.nan_to_num() != b.nan_to_num()).nonzero()
                if dif.numel():
                    i0 = tuple(int(v) for v in dif[0])
                    where += f' {name_}: {dif.shape[0]} elements, first at {i0}: {float(a[i0]):.9e} vs {float(b[i0]):.9e};'
            assert same, (f'overlapped exchanges changed the result (largest rel-L2 difference {worst:.2e} over W_D1, W_D2, loss history;'
                          f'{where})')
        dist.destroy_process_group()
        return
    auto_est = None
    if auto and world > 1 and mode == 'partition':
        # measure the single-GPU step on every rank (= the replicas rate reported under extras) and print the planner's estimate
        # next to the measured partitioned step.  The HEADLINE stays the partitioned step at every N (a curve whose
        # points switch between strong and weak scaling cannot be read); `--parallel replicas` asks for the other headline.
        rep_rate_auto = replicas_rate(args, model, state, device, world, barrier)
        share, share_src = layer1_share(args.stage_profile)
        auto_est = partition_estimate(eng, world, ctl, 1e6 * world / rep_rate_auto, l1_share=share, l1_source=share_src)
        auto_est['chosen'] = 'partition'
        model.load_state_dict(state)
        eng = make_engine(args, data, model, neg, ni1, ni2, device, rank, world, group)    # fresh state for the timed run
    if mode != 'partition':
        eng = make_engine(args, data, model, neg, ni1, ni2, device, rank, world)

    # several iterations per graph launch (a replay boundary costs ~8 us that a kernel boundary inside a graph
    # does not); captured before the warm-up so that the timed region only replays
    run = getattr(eng, 'run', None)
    if run is not None and args.unroll > 1:
        eng.prepare_unrolled(args.unroll)
    for _ in range(args.warmup):
        eng.step()
    def timed_region():
        """EXACTLY --steps steps between barrier + synchronize on both sides; max over ranks."""
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if run is not None and args.unroll > 1:
            run(args.steps, unroll=args.unroll)
        else:
            for _ in range(args.steps):
                eng.step()
        torch.cuda.synchronize()
        barrier()
        dt_ = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt_], dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=ctl)
            dt_ = float(tmax)
        return dt_
    n_regions = args.repeats if args.repeats > 0 else (5 if args.steps < 100 else 1)
    region_s = [timed_region() for _ in range(n_regions)]
    dt = sorted(region_s)[len(region_s) // 2]
    losses = eng.loss_history()

    partitioned = mode == 'partition'
    units = args.steps if partitioned else world * args.steps      # iterations of whole requests
    per_rank_halo = gather_halo_bytes(eng, world, ctl) if (partitioned and world > 1) else None
    rep_rate = (rep_rate_auto if auto_est is not None else replicas_rate(args, model, state, device, world, barrier)) \
        if (partitioned and world > 1) else None
    if rank == 0:
        kdur_b2b, kbytes = time_dominant_kernel(eng)
        kdur_ctx = time_dominant_kernel_in_context(eng) if world == 1 else None
        kdur = kdur_ctx if kdur_ctx else kdur_b2b           # (in the step's context where that can be measured: see the function)
        achieved = kbytes / kdur / 1e9
        prof, prof_note = load_stage_profile(args.stage_profile, data.num_nodes, eng.graph.nnz) if world == 1 else ({}, 'N > 1')
        traffic = prof.get('stages', {}).get('spmm1', {}).get('traffic_bytes')
        ceil_ = fabric_ceiling(data.num_nodes, eng.graph.nnz, 128)
        out = {
            'metric': 'Del-op train iters/sec', 'value': units / dt, 'unit': 'iters/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'strong' if partitioned else 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{args.workload} {args.gnn.upper()} 2-layer, {args.df_size}% {args.df.upper()} '
                                   f'edge deletion, full-graph Del step ({args.loss_type}, mse_mean)',
                       'num_nodes': data.num_nodes, 'in_dim': int(data.x.shape[1]), 'hidden_dim': 128, 'out_dim': 64,
                       'train_edges_undirected': int(data.train_pos_edge_index.shape[1]),
                       'df_edges': int(data.directed_df_edge_index.shape[1]),
                       'sdf_edges': int(data.sdf_mask.sum()), 'spmm_nnz': eng.graph.nnz,
                       'S1': int(data.sdf_node_1hop_mask.sum()), 'S2': int(data.sdf_node_2hop_mask.sum()),
                       'backbone': f'trained {args.pretrain_epochs} epochs on the HIP convs before the request '
                                   f'(BCE link prediction, final loss {pretrain_loss:.4f})' if pretrain_loss is not None
                                   else 'random init',
                       'hip_graph': not args.no_graph, 'iterations_per_graph_launch': 1 if args.no_graph else args.unroll,
                       'matrix_products': matrix_products_label(),
                       'parallelism': 'single' if world == 1 else (f'row-partition x{world} (RCCL halo all-to-all + all-reduce)'
                                                                  if partitioned else f'replicas x{world}')},
            'roofline': {'kernel': 'spmm_persist_kernel<32,1,4,true,true> (layer-1 CSR SpMM, d=128)', 'bound': 'hbm',
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic,
                         # the recorded L2-miss (fabric) bytes over this run's launch duration: how close the kernel
                         # runs to the ~6.3 TB/s a streaming copy achieves on this part (MI355X_MICROARCH.md)
                         'traffic_gbs': traffic / kdur / 1e9 if traffic else None,
                         'traffic_unit': ('bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes of this command: '
                                          + os.path.relpath(args.stage_profile, ROOT) + ')') if prof else None,
                         'stage_profile': os.path.relpath(args.stage_profile, ROOT) if prof else prof_note,
                         'algorithmic_bytes': kbytes, 'avg_us': kdur * 1e6,
                         'timing': ('HIP events around 20 (x W1^T; SpMM) pairs minus 20 x W1^T alone: the launch behind the kernel that '
                                    'precedes it in the step' if kdur_ctx else 'HIP events around 20 back-to-back launches'),
                         'back_to_back_us': kdur_b2b * 1e6, 'frac_back_to_back': kbytes / kdur_b2b / 1e9 / HBM_PEAK_GBS,
                         'in_step_us': prof.get('stages', {}).get('spmm1', {}).get('in_step_us'),
                         # what any kernel with one contiguous row range per XCD can reach on this graph (measurement):
                         # the private L2s make the fabric carry every x row once per XCD that gathers it
                         'ceiling_frac_on_this_graph': ceil_['frac'] if ceil_ else None, 'ceiling': ceil_},
            'final_loss': float(losses[-1, 0]) if len(losses) else None,
            'timing': {'regions': n_regions, 'steps_per_region': args.steps, 'reported': 'median region',
                       'ms_per_step_each_region': [1e3 * t / args.steps for t in region_s]},
        }
        if args.gnn == 'gat' and world == 1 and hasattr(eng, 'model'):
            # (round 6) the GAT step never runs the plain SpMM: its dominant HBM kernel is the attention-scored aggregation
            out['roofline'] = dict(time_gat_aggregation(eng), stage_profile=out['roofline']['stage_profile'])
        if note:
            out['config']['partition_fallback'] = note
        out['config']['ranks_seen'] = world          # WORLD_SIZE of the torch.distributed job this line was measured in
        if auto_est is not None:
            out['config']['parallel_auto'] = auto_est
            try:        # the whole curve's model from this one run (halo bytes and predicted step at N = 2 / 4 / 8)
                out['config']['planner_curve'] = planner_curve(eng, auto_est['single_gpu_step_us'], l1_share=auto_est['layer1_share_of_step'])
            except Exception as e:                                   # noqa: BLE001
                out['config']['planner_curve'] = f'unavailable: {type(e).__name__}: {str(e)[:120]}'
        if world > 1:
            out['config']['halo_exchanges'] = ('overlapped with compute (the probe reproduced the synchronous result)' if getattr(args, 'dist_overlap', None)
                                               else f'synchronous ({overlap_note})')
        if partitioned:
            # rank 0's own figures + every rank's bytes (gathered over the control group): xGMI is point-to-point, the
            # heaviest rank / pair bounds an exchange
            rep = eng.halo_report()
            out['config']['halo'] = rep
            out['config']['halo_recv_send_bytes_per_rank'] = per_rank_halo
            out['config']['collective_backend'] = backend
        if world == 1 and not args.no_cached_rate:
            # informational only (never `value`): the same step with the loop-invariant frozen layer-1
            # output computed once, which is how the trainer runs by default
            from gnndelete_amd.engine import NodeembEngine
            model.load_state_dict(state)
            ceng = NodeembEngine(*eng_args, loss_type=args.loss_type, alpha=0.5, lr=1e-3, cache_layer1=True)
            # (every informational rate below: the median of as many --steps regions as the headline, not ONE region)
            out['extras'] = {'timing': 'median region, as the headline', 'ms_per_step_each_region': {}}
            per_region = out['extras']['ms_per_step_each_region']
            out['extras']['iters_per_s_with_loop_invariant_layer1_cached'], per_region['layer1_cached'] = median_rate(ceng, args)
            del ceng
            # informational only: the step restricted to the rows the request can influence (identical results,
            # DESIGN.md section 2), without and with the layer-1 cache - the latter is what the trainer runs
            for key, cached in (('iters_per_s_affected_rows_only', False), ('iters_per_s_trainer_default', True)):
                model.load_state_dict(state)
                reng = NodeembEngine(*eng_args, loss_type=args.loss_type, alpha=0.5, lr=1e-3, cache_layer1=cached,
                                     affected_rows_only=True)
                if not getattr(reng, '_rows_only', False):
                    continue
                out['extras'][key], per_region[key.replace('iters_per_s_', '')] = median_rate(reng, args)
                del reng
            out['extras']['affected_rows'] = {'S2': int(data.sdf_node_2hop_mask.sum()), 'of': data.num_nodes}
            # informational only: the same full step with the 128-wide row GEMMs' fp32 products formed from bf16 partial
            # products (gd_set_matrix_split(6), opt-in) - `value` above uses the fp32 instruction
            from gnndelete_amd import ops as _ops
            if _ops.matrix_split() == 0:
                _ops.set_matrix_split(6)
                try:
                    model.load_state_dict(state)
                    seng = NodeembEngine(*eng_args, loss_type=args.loss_type, alpha=0.5, lr=1e-3)
                    out['extras']['iters_per_s_bf16x6_split_products'], per_region['bf16x6_split_products'] = median_rate(seng, args)
                    out['extras']['roofline_del_gemm_bf16x6_split'] = time_del_gemm(seng)
                finally:
                    _ops.set_matrix_split(0)
        if rep_rate is not None:
            out.setdefault('extras', {})['iters_per_s_independent_replicas'] = rep_rate
        if world == 1 and hasattr(eng, 'idx1'):
            out.setdefault('extras', {})['roofline_del_gemm'] = time_del_gemm(eng)
            out['extras']['roofline_wgrad'] = time_wgrad(eng)
            out['extras']['roofline_spmm_d64'] = time_spmm_d64(eng)
            out['extras']['stage_rooflines'] = stage_rooflines(eng, prof) or stage_table_from_profile(prof)
        if not args.no_cpu_baseline and world == 1:
            cpu_data = data.clone().cpu() if hasattr(data, 'clone') else data
            out['cpu_baseline'], cpu_model, n_cpu = cpu_baseline(args, cpu_data, state, neg, args.cpu_baseline_iters)
            out['speedup_vs_cpu'] = out['value'] / out['cpu_baseline']['value']
            out['post_delete_auc'] = post_delete_parity(args, cpu_data, model, state, neg, ni1, ni2, device, cpu_model, n_cpu)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
