"""Command-line contract of the reference (framework/training_args.py): every flag with the same
name, type, default and help string, and the same post-parse overrides that silently replace CLI
values (ogbl* -> eval_on_cpu; KG -> lr / epochs / valid_freq / batch halving / num_edge_type;
original|retrain -> 2000 epochs; gnndelete on ogbl* -> 600 epochs; ...).  Pinned against the
reference's own parse_args by tests/golden/parse_args.json.  Flags added by this build are
listed in EXTRA_FLAGS and never change a reference default."""
import argparse

num_edge_type_mapping = {'FB15k-237': 237, 'WordNet18': 18, 'WordNet18RR': 11, 'ogbl-biokg': 51,
                         # synthetic stand-ins written by prepare_dataset.py (same relation counts)
                         'synth-kg-tiny': 4, 'synth-wn18': 18, 'synth-biokg': 51}

# (name, type, default, help); type None = store_true switch
FLAGS = [
    # Model
    ('unlearning_model', str, 'retrain', 'unlearning method'),
    ('gnn', str, 'gcn', 'GNN architecture'),
    ('in_dim', int, 128, 'input dimension'),
    ('hidden_dim', int, 128, 'hidden dimension'),
    ('out_dim', int, 64, 'output dimension'),
    # Data
    ('data_dir', str, './data', 'data dir'),
    ('df', str, 'none', 'Df set to use'),
    ('df_idx', str, 'none', 'indices of data to be deleted'),
    ('df_size', float, 0.5, 'Df size'),
    ('dataset', str, 'Cora', 'dataset'),
    ('random_seed', int, 42, 'random seed'),
    ('batch_size', int, 8192, 'batch size for GraphSAINTRandomWalk sampler'),
    ('walk_length', int, 2, 'random walk length for GraphSAINTRandomWalk sampler'),
    ('num_steps', int, 32, 'number of steps for GraphSAINTRandomWalk sampler'),
    # Training
    ('lr', float, 1e-3, 'initial learning rate'),
    ('weight_decay', float, 0.0005, 'weight decay'),
    ('optimizer', str, 'Adam', 'optimizer to use'),
    ('epochs', int, 3000, 'number of epochs to train'),
    ('valid_freq', int, 100, '# of epochs to do validation'),
    ('checkpoint_dir', str, './checkpoint', 'checkpoint folder'),
    ('alpha', float, 0.5, 'alpha in loss function'),
    ('neg_sample_random', str, 'non_connected', 'type of negative samples for randomness'),
    ('loss_fct', str, 'mse_mean', 'loss function. one of {mse, kld, cosine}'),
    ('loss_type', str, 'both_layerwise',
     'type of loss. one of {both_all, both_layerwise, only2_layerwise, only2_all, only1}'),
    # GraphEraser
    ('num_clusters', int, 10, 'top k for evaluation'),
    ('kmeans_max_iters', int, 1, 'top k for evaluation'),
    ('shard_size_delta', float, 0.005, None),
    ('terminate_delta', int, 0, None),
    # GraphEditor
    ('eval_steps', int, 1, None),
    ('runs', int, 1, None),
    ('num_remove_links', int, 11, None),
    ('parallel_unlearning', int, 4, None),
    ('lam', float, 0, None),
    ('regen_feats', None, False, None),
    ('regen_neighbors', None, False, None),
    ('regen_links', None, False, None),
    ('regen_subgraphs', None, False, None),
    ('hop_neighbors', int, 20, None),
    # Evaluation
    ('topk', int, 500, 'top k for evaluation'),
    ('eval_on_cpu', bool, False, 'whether to evaluate on CPU'),
    # KG
    ('num_edge_type', int, None, 'number of edges types'),
]

EXTRA_FLAGS = [
    ('minibatch', None, False, 'train ogbl-* graphs on GraphSAINT mini-batches as upstream does (default: full graph)'),
    ('fullgraph', None, False, 'knowledge-graph unlearning (R-GCN): one fused full-graph step per epoch instead of the GraphSAINT batches upstream trains on'),
    ('no_fused_step', None, False, 'use the autograd path even where the fused hipGraph step applies'),
    ('no_layer1_cache', None, False, 'recompute the frozen layer-1 output every epoch as upstream does (identical results)'),
    ('all_rows', None, False, 'run every row of the graph through the unlearning step as upstream does, not only the rows the request can influence (identical results)'),
]


def build_parser(extra=True):
    parser = argparse.ArgumentParser()
    for name, typ, default, help_ in FLAGS + (EXTRA_FLAGS if extra else []):
        if typ is None:
            parser.add_argument('--' + name, action='store_true')
        else:
            parser.add_argument('--' + name, type=typ, default=default, help=help_)
    return parser


OGB_STANDINS = {'synth-collab': 'ogbl-collab', 'synth-biokg': 'ogbl-biokg'}


def is_large(dataset):
    """The reference keys its big-graph behaviour on 'ogbl' in the dataset name (training_args.py:103-157,
    base.py:314-317); the seeded synthetic stand-ins with the OGB shapes follow the same rules."""
    return 'ogbl' in dataset or dataset in OGB_STANDINS


def apply_overrides(args):
    relational = args.gnn in ['rgcn', 'rgat']
    large = is_large(args.dataset)
    if large:
        args.eval_on_cpu = True
    if relational:
        args.lr, args.epochs, args.valid_freq = 1e-3, 3000, 500
        args.batch_size //= 2
        args.num_edge_type = num_edge_type_mapping[args.dataset]
        args.eval_on_cpu = True
    if args.unlearning_model in ['original', 'retrain']:
        args.epochs, args.valid_freq = 2000, 500
        if large and not relational:
            args.epochs, args.valid_freq = 600, 200
        if large and relational:
            args.batch_size = 1024
    if 'gnndelete' in args.unlearning_model:
        if large and not relational:
            args.epochs, args.valid_freq = 600, 100
        if relational and args.dataset == 'WordNet18':
            args.epochs, args.valid_freq, args.batch_size = 50, 2, 1024
        if relational and OGB_STANDINS.get(args.dataset, args.dataset) == 'ogbl-biokg':
            args.epochs, args.valid_freq, args.batch_size = 50, 10, 64
    elif args.unlearning_model == 'gradient_ascent':
        args.epochs, args.valid_freq = 10, 1
    elif args.unlearning_model == 'descent_to_delete':
        args.epochs = 1
    elif args.unlearning_model == 'graph_editor':
        args.epochs, args.valid_freq = 400, 200
    if args.dataset == 'ogbg-molhiv':
        args.epochs, args.valid_freq = 100, 5
    return args


def _test_knobs(args):
    """GNNDELETE_FORCE_{EPOCHS,VALID_FREQ,NUM_STEPS}: applied AFTER the upstream overrides (which
    replace --epochs for original / KG runs); used by the end-to-end tests to keep runs short."""
    import os
    for env, attr in [('GNNDELETE_FORCE_EPOCHS', 'epochs'), ('GNNDELETE_FORCE_VALID_FREQ', 'valid_freq'),
                      ('GNNDELETE_FORCE_NUM_STEPS', 'num_steps')]:
        if os.environ.get(env):
            setattr(args, attr, int(os.environ[env]))
    return args


def parse_args(argv=None):
    return _test_knobs(apply_overrides(build_parser().parse_args(argv)))
