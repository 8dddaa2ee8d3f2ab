"""Host-side mirror of the reference's ``framework`` package for the Del hot path: the same
factories (get_model / get_trainer), registry names and class names (framework/__init__.py:14-67).
Registry entries whose upstream implementation is outside the hot path (competing baselines,
membership-inference attack, the unfinished graph_eraser / missing graph_editor) are not built;
asking for one raises NotImplementedError naming it."""
from .models import (GAT, GCN, GIN, RGAT, RGCN, SAGE, GATDelete, GCNDelete, GINDelete, RGATDelete,  # noqa: F401
                     RGCNDelete, SAGEDelete)
from .trainer.base import NodeClassificationTrainer, Trainer
from .trainer.gnndelete_nodeemb import GNNDeleteNodeClassificationTrainer, GNNDeleteNodeembTrainer

_OUT_OF_SCOPE = ['gradient_ascent', 'descent_to_delete', 'approx_retrain', 'graph_eraser', 'graph_editor',
                 'member_infer_all', 'member_infer_sub', 'member_infer_all_node', 'member_infer_sub_node']

trainer_mapping = {
    'original': Trainer,
    'original_node': NodeClassificationTrainer,
    'gnndelete_nodeemb': GNNDeleteNodeembTrainer,
}

kg_trainer_mapping = {}


def _lazy_trainers():
    from .trainer.gnndelete import GNNDeleteTrainer
    for key in ('gnndelete', 'gnndelete_mse', 'gnndelete_kld', 'gnndelete_cosine'):
        trainer_mapping.setdefault(key, GNNDeleteTrainer)
    from .trainer.retrain import KGRetrainTrainer, RetrainTrainer
    trainer_mapping.setdefault('retrain', RetrainTrainer)
    try:
        from .trainer.kg import KGGNNDeleteNodeembTrainer, KGTrainer
        kg_trainer_mapping.setdefault('original', KGTrainer)
        kg_trainer_mapping.setdefault('retrain', KGRetrainTrainer)
        kg_trainer_mapping.setdefault('gnndelete', KGGNNDeleteNodeembTrainer)
        kg_trainer_mapping.setdefault('gnndelete_nodeemb', KGGNNDeleteNodeembTrainer)
    except ImportError:
        pass


def get_model(args, mask_1hop=None, mask_2hop=None, num_nodes=None, num_edge_type=None):
    if 'gnndelete' in args.unlearning_model:
        model_mapping = {'gcn': GCNDelete, 'gat': GATDelete, 'gin': GINDelete, 'rgcn': RGCNDelete, 'rgat': RGATDelete,
                         'sage': SAGEDelete}
    else:
        model_mapping = {'gcn': GCN, 'gat': GAT, 'gin': GIN, 'rgcn': RGCN, 'rgat': RGAT, 'sage': SAGE}
    if args.gnn not in model_mapping:
        raise NotImplementedError(f"gnn '{args.gnn}' is outside the hot-path scope of this build (have: "
                                  f"{sorted(model_mapping)})")
    return model_mapping[args.gnn](args, mask_1hop=mask_1hop, mask_2hop=mask_2hop, num_nodes=num_nodes,
                                   num_edge_type=num_edge_type)


def get_trainer(args):
    _lazy_trainers()
    table = kg_trainer_mapping if args.gnn in ['rgcn', 'rgat'] else trainer_mapping
    if args.unlearning_model not in table:
        scope = 'out of scope (reference baseline)' if args.unlearning_model in _OUT_OF_SCOPE else 'unknown'
        raise NotImplementedError(f"unlearning_model '{args.unlearning_model}' is {scope}")
    return table[args.unlearning_model](args)
