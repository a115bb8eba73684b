"""The helpers of offpolicy/utils/util.py that sit on the update path, under their upstream names.

`soft_update` / `hard_update` (util.py:123-146) act on the flat parameter vectors through `ope_polyak`; the action
helpers are the host-side versions the policies' rollout code uses (the update path has them as HIP kernels,
csrc/ope_ddpg.hip `action_kernel`)."""
import numpy as np
import torch

from .. import _lib
from ..algorithms.maddpg.algorithm.MADDPGPolicy import gumbel_softmax_hard, onehot_from_logits, sample_gumbel_uniform  # noqa: F401
from ..algorithms.qmix.algorithm.QMixPolicy import DecayThenFlatSchedule  # noqa: F401
from .spaces import get_dim_from_space  # noqa: F401


def to_torch(x):
    return torch.from_numpy(x) if isinstance(x, np.ndarray) else x


def to_numpy(x):
    return x.detach().cpu().numpy()


def huber_loss(e, d):
    a = (abs(e) <= d).float()
    b = (abs(e) > d).float()
    return a * e ** 2 / 2 + b * d * (abs(e) - d / 2)


def mse_loss(e):
    return e ** 2


def _param_pairs(target, source):
    """zip(target.parameters(), source.parameters()) as the reference does (util.py:131-134); policy objects (anything
    with .parameters()) and FlatModule networks are both accepted."""
    tp, sp = list(target.parameters()), list(source.parameters())
    assert len(tp) == len(sp), "target and source hold different numbers of parameters"
    return zip(tp, sp)


def soft_update(target, source, tau):
    """target <- (1 - tau) target + tau source, parameter by parameter (util.py:123-134). Acts on the registered
    parameters only -- i.e. on the MODULE's own slices of a shared flat vector: the agent q-network's update leaves the
    mixer block that lives in the same trainer vector alone, and a MADDPG critic with frozen (unregistered, SURVEY A-4)
    Q heads keeps its target heads untouched, exactly like the reference's zip over .parameters(). One `ope_polyak`
    launch per contiguous run of parameters."""
    runs = []          # (target tensor slice start ptr, source ptr, numel) merged over adjacent parameters
    for t, s_ in _param_pairs(target, source):
        assert t.shape == s_.shape and t.is_contiguous() and s_.is_contiguous()
        n = t.numel()
        if n == 0:
            continue
        if runs and runs[-1][0].data_ptr() + 4 * runs[-1][2] == t.data_ptr() and runs[-1][1].data_ptr() + 4 * runs[-1][2] == s_.data_ptr():
            runs[-1][2] += n
        else:
            runs.append([t.data, s_.data, n])
    st = _lib.current_stream()
    import ctypes as C
    for t, s_, n in runs:
        _lib.check(_lib.lib.ope_polyak(n, C.c_void_p(s_.data_ptr()), C.c_void_p(t.data_ptr()), float(tau), st), "ope_polyak")


def hard_update(target, source):
    """target <- source, parameter by parameter (util.py:137-146)."""
    soft_update(target, source, 1.0)


def sample_gumbel(shape, eps=1e-20):
    u = sample_gumbel_uniform(tuple(shape))
    return -torch.log(-torch.log(u + eps) + eps)


def gumbel_softmax(logits, avail_logits=None, temperature=1.0, hard=False, device=None):
    if temperature != 1.0 or not hard:
        raise NotImplementedError("only the hard, temperature-1 form is used on the update path")
    return gumbel_softmax_hard(logits, avail_logits, sample_gumbel_uniform(tuple(logits.shape)))


def is_discrete(space):
    return space.__class__.__name__ in ("Discrete", "MultiDiscrete")


def is_multidiscrete(space):
    return space.__class__.__name__ == "MultiDiscrete"


def avail_choose(x, avail_x=None):
    x = to_torch(x)
    if avail_x is not None:
        x = x.clone()
        x[to_torch(np.asarray(avail_x)) == 0] = -1e10
    return x


def make_onehot(int_action, action_dim, seq_len=None):
    if seq_len is not None:
        return np.eye(action_dim)[np.asarray(int_action).reshape(seq_len, -1)].reshape(seq_len, -1, action_dim)
    return np.eye(action_dim)[np.asarray(int_action).reshape(-1)]


def get_cent_act_dim(action_space):
    return int(sum(int(np.sum(get_dim_from_space(s))) for s in action_space))
