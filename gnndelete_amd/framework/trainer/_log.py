"""Logging glue: optional wandb (absent here -> silent shim), compact console lines."""
try:                                     # the reference logs through wandb (delete_gnn.py:83,243)
    import wandb as _wandb
except Exception:                        # noqa: BLE001 - any import problem means "no wandb"
    _wandb = None


def wandb_log(record):
    if _wandb is not None and getattr(_wandb, 'run', None) is not None:
        _wandb.log(record)


def wandb_init(args):
    if _wandb is not None:
        try:
            _wandb.init(config=args)
        except Exception:                # offline / not configured: carry on without it
            pass


def fmt(record):
    return ' | '.join(f'{k}: {v:>4d}' if isinstance(v, int) else f'{k}: {v:.4f}' for k, v in record.items())
