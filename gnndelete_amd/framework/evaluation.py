"""verification_error (reference: framework/evaluation.py:63-81): sum over the shared named
parameters of the L2 distance between an unlearned model and a retrained one."""
import torch


@torch.no_grad()
def verification_error(model1, model2):
    '''L2 distance between aproximate model and re-trained model'''
    p1 = {n: p.detach().cpu() for n, p in model1.named_parameters()}
    p2 = {n: p.detach().cpu() for n, p in model2.named_parameters()}
    diff = torch.tensor(0.0)
    for name in set(p1) & set(p2):
        diff += torch.norm(p1[name] - p2[name])
    return diff
