"""Import-path shim: with the repository root on sys.path, code written against the reference
(`from framework import get_model, get_trainer`, `from framework.training_args import parse_args`,
`from framework.models.gcn import GCN`, ...) resolves to gnndelete_amd.framework unchanged."""
import importlib
import sys

_impl = importlib.import_module('gnndelete_amd.framework')
for _name in ('models', 'models.gcn', 'models.gat', 'models.gin', 'models.rgcn', 'models.deletion', 'trainer',
              'trainer.base', 'trainer.gnndelete', 'trainer.gnndelete_nodeemb', 'training_args', 'utils',
              'evaluation', 'data', 'graph_utils', 'synth'):
    sys.modules[f'framework.{_name}'] = importlib.import_module(f'gnndelete_amd.framework.{_name}')
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
