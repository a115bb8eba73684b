"""Import the upstream reference (marlbenchmark/off-policy, mounted read-only at /root/reference) on CPU.

TEST INFRASTRUCTURE ONLY. Used by `oracle/make_golden.py` (fixture generation, run in the build container)
and by the `-m "not gpu"` tests that pin `oracle/*.py` against the real reference when it is present.
Nothing in the product package, `bench.py`'s GPU leg or the `-m gpu` tests may import this module: the
GPU box has no /root/reference.

Recipe: stub gym / wandb / tensorboardX / absl / setproctitle, and pre-register a
bare `offpolicy` package so that `offpolicy/__init__.py:1` (which pulls in pysc2 through the envs) never
runs. The reference is never modified; bytecode writing is disabled so nothing lands in the mount.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("OPE_REFERENCE_ROOT", "/root/reference")
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refshim")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "offpolicy"))


def load_reference():
    """Make `import offpolicy.<hot-path module>` work. Returns the bare `offpolicy` package."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    if _SHIM not in sys.path:
        sys.path.insert(0, _SHIM)
    if "offpolicy" not in sys.modules:
        pkg = types.ModuleType("offpolicy")
        pkg.__path__ = [os.path.join(REFERENCE_ROOT, "offpolicy")]
        sys.modules["offpolicy"] = pkg
    return sys.modules["offpolicy"]


def reference_args(argv=(), **overrides):
    """Reference hyper-parameter namespace: offpolicy/config.py:4-194 defaults + the train-script extras
    (offpolicy/scripts/train/train_smac.py:52-64) that the trainers read (`use_same_share_obs`,
    `use_available_actions`)."""
    load_reference()
    from offpolicy.config import get_config
    args = get_config().parse_known_args(list(argv))[0]
    args.use_same_share_obs = True
    args.use_available_actions = True
    for k, v in overrides.items():
        setattr(args, k, v)
    return args
