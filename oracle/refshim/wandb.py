"""No-op stub (reference imports wandb at runner level only)."""
run = None


def init(*a, **k):
    return None


def log(*a, **k):
    return None
