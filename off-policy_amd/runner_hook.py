"""Make the reference's runners instantiate the engine's classes without editing the reference.

`offpolicy/runner/rnn/base_runner.py:110-130` and `offpolicy/runner/mlp/base_runner.py:99-118` pick buffers, policies and
trainers with `from offpolicy.<module> import <Class>` statements executed inside the runner's constructor. `install()`
registers the engine's mirror modules in `sys.modules` under those exact dotted names, so those import statements (and
`from offpolicy.utils.rec_buffer import ...` at the top of the runner module) resolve to the MI355X classes. Call it
before the runner is constructed:

    import offpolicy_amd.runner_hook as hook
    hook.install()                      # or hook.install(only=("qmix", "vdn"))
    from offpolicy.runner.rnn.smac_runner import SMACRunner

Everything not listed in MODULE_MAP (envs, runners, config, logging) stays the reference's own code. `uninstall()` puts
the previous `sys.modules` entries back."""
import importlib
import sys

# reference module -> engine module (relative to the offpolicy_amd package); family tags select subsets
MODULE_MAP = {
    "offpolicy.utils.rec_buffer": ("utils.rec_buffer", ("qmix", "vdn", "rmaddpg", "rmatd3")),
    "offpolicy.utils.mlp_buffer": ("utils.mlp_buffer", ("mqmix", "mvdn", "maddpg", "matd3")),
    "offpolicy.utils.segment_tree": ("utils.segment_tree", ("qmix", "vdn", "rmaddpg", "rmatd3", "mqmix", "mvdn", "maddpg", "matd3")),
    "offpolicy.algorithms.qmix.qmix": ("algorithms.qmix.qmix", ("qmix", "vdn")),
    "offpolicy.algorithms.qmix.algorithm.QMixPolicy": ("algorithms.qmix.algorithm.QMixPolicy", ("qmix", "vdn")),
    "offpolicy.algorithms.qmix.algorithm.agent_q_function": ("algorithms.qmix.algorithm.agent_q_function", ("qmix", "vdn")),
    "offpolicy.algorithms.qmix.algorithm.q_mixer": ("algorithms.qmix.algorithm.q_mixer", ("qmix",)),
    "offpolicy.algorithms.vdn.vdn": ("algorithms.vdn.vdn", ("vdn",)),
    "offpolicy.algorithms.vdn.algorithm.VDNPolicy": ("algorithms.vdn.algorithm.VDNPolicy", ("vdn",)),
    "offpolicy.algorithms.vdn.algorithm.vdn_mixer": ("algorithms.vdn.algorithm.vdn_mixer", ("vdn",)),
    "offpolicy.algorithms.mqmix.mqmix": ("algorithms.mqmix.mqmix", ("mqmix", "mvdn")),
    "offpolicy.algorithms.mqmix.algorithm.mQMixPolicy": ("algorithms.mqmix.algorithm.mQMixPolicy", ("mqmix", "mvdn")),
    "offpolicy.algorithms.mvdn.mvdn": ("algorithms.mvdn.mvdn", ("mvdn",)),
    "offpolicy.algorithms.mvdn.algorithm.mVDNPolicy": ("algorithms.mvdn.algorithm.mVDNPolicy", ("mvdn",)),
    "offpolicy.algorithms.maddpg.maddpg": ("algorithms.maddpg.maddpg", ("maddpg", "matd3")),
    "offpolicy.algorithms.maddpg.algorithm.MADDPGPolicy": ("algorithms.maddpg.algorithm.MADDPGPolicy", ("maddpg", "matd3")),
    "offpolicy.algorithms.matd3.matd3": ("algorithms.matd3.matd3", ("matd3",)),
    "offpolicy.algorithms.matd3.algorithm.MATD3Policy": ("algorithms.matd3.algorithm.MATD3Policy", ("matd3",)),
    "offpolicy.algorithms.r_maddpg.r_maddpg": ("algorithms.r_maddpg.r_maddpg", ("rmaddpg", "rmatd3")),
    "offpolicy.algorithms.r_maddpg.algorithm.rMADDPGPolicy": ("algorithms.r_maddpg.algorithm.rMADDPGPolicy", ("rmaddpg", "rmatd3")),
    "offpolicy.algorithms.r_matd3.r_matd3": ("algorithms.r_matd3.r_matd3", ("rmatd3",)),
    "offpolicy.algorithms.r_matd3.algorithm.rMATD3Policy": ("algorithms.r_matd3.algorithm.rMATD3Policy", ("rmatd3",)),
}

_saved = {}


def install(only=None):
    """Register the engine's modules under the reference's names. `only`: iterable of algorithm names
    (qmix, vdn, mqmix, mvdn, maddpg, matd3, rmaddpg, rmatd3) to restrict the swap to; default all. Returns the list of
    reference module names that now resolve to the engine."""
    want = set(only) if only is not None else None
    done = []
    for ref_name, (ours, families) in MODULE_MAP.items():
        if want is not None and not (want & set(families)):
            continue
        mod = importlib.import_module("offpolicy_amd." + ours)      # raises if libope.so is missing: no silent fallback
        if ref_name not in _saved:
            _saved[ref_name] = sys.modules.get(ref_name)
        sys.modules[ref_name] = mod
        # `from offpolicy.utils import rec_buffer`-style access goes through the parent package's attribute
        parent, _, leaf = ref_name.rpartition(".")
        if parent in sys.modules and sys.modules[parent] is not None:
            setattr(sys.modules[parent], leaf, mod)
        done.append(ref_name)
    return done


def uninstall():
    for ref_name, prev in _saved.items():
        if prev is None:
            sys.modules.pop(ref_name, None)
        else:
            sys.modules[ref_name] = prev
    _saved.clear()
