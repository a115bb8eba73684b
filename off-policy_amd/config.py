"""Hyper-parameter namespace with the reference's defaults for every flag the update path reads.

The reference's argparse surface (offpolicy/config.py:4-194 plus train_smac.py:52-64) is launch glue and stays
with the reference; its DEFAULT VALUES are the parity contract, reproduced here so the
trainers can be constructed stand-alone. Any object with these attributes (e.g. the reference's parsed
namespace) is accepted wherever `args` is taken.
"""
from types import SimpleNamespace

DEFAULTS = dict(
    algorithm_name="qmix", hidden_size=64, layer_N=1, use_ReLU=True, use_feature_normalization=True,
    use_orthogonal=True, gain=0.01, use_conv1d=False, stacked_frames=1, prev_act_inp=False, use_rnn_layer=True,
    recurrent_N=1, lr=5e-4, opti_eps=1e-5, weight_decay=0.0, batch_size=32, gamma=0.99, max_grad_norm=10.0,
    use_huber_loss=False, huber_delta=10.0, use_soft_update=True, tau=0.005, hard_update_interval_episode=200,
    use_double_q=True, hypernet_layers=2, mixer_hidden_dim=32, hypernet_hidden_dim=64, buffer_size=5000,
    episode_length=80, use_reward_normalization=False, use_popart=False, use_per=False, per_nu=0.9, per_alpha=0.6,
    per_eps=1e-6, per_beta_start=0.4, use_value_active_masks=False, epsilon_start=1.0, epsilon_finish=0.05,
    epsilon_anneal_time=50000, use_same_share_obs=True, use_available_actions=True, share_policy=True,
    train_interval_episode=1, actor_train_interval_step=2, target_action_noise_std=0.2, act_noise_std=0.1,
)


def default_args(**overrides):
    d = dict(DEFAULTS)
    for k, v in overrides.items():
        if k not in d:
            raise AttributeError("unknown hyper-parameter %r" % k)
        d[k] = v
    return SimpleNamespace(**d)


def require_reference_architecture(args, allow_prev_act_inp=False, allow_hypernet_layers_1=False, allow_layer_N_2=False,
                                   allow_no_feature_norm=False, allow_tanh=False):
    """The kernels are specialised to the reference's default network shape; refuse anything else loudly.
    `prev_act_inp` (previous action appended to the agent's observation, config.py:81) only changes the width of the
    network input; the recurrent QMIX / VDN policy supports it (`allow_prev_act_inp`), the other families do not yet.
    `hypernet_layers = 1` (one-layer hyper-networks, q_mixer.py:39-44) is supported by the recurrent QMIX trainer with one shared
    policy (`allow_hypernet_layers_1`; the fused chain kernels, csrc/ope_chain.hip); `layer_N = 2` (a second hidden block behind fc1,
    mlp.py:14-28) by the recurrent QMIX / VDN trainer and its policy (`allow_layer_N_2`; csrc/ope_block.hip);
    `use_feature_normalization = False` (no LayerNorm on the network input, mlp.py:60-62) by the same trainer and policy and by the MLP
    Q-learning family (`allow_no_feature_norm`; OPE_DIMS_NO_FEATURE_NORM in include/ope.h); `use_ReLU = False` (tanh in the MLP base,
    mlp.py:9-12) likewise (`allow_tanh`; OPE_DIMS_TANH: one hidden block, input width <= 384)."""
    want = dict(hidden_size=64, layer_N=1, use_ReLU=True, use_feature_normalization=True, use_conv1d=False,
                prev_act_inp=False, use_rnn_layer=True, recurrent_N=1, hypernet_layers=2, mixer_hidden_dim=32,
                hypernet_hidden_dim=64, use_popart=False)
    if allow_prev_act_inp:
        del want["prev_act_inp"]
    if allow_hypernet_layers_1 and getattr(args, "hypernet_layers", 2) == 1:
        del want["hypernet_layers"]
    if allow_layer_N_2 and getattr(args, "layer_N", 1) == 2:
        del want["layer_N"]
    if allow_no_feature_norm:
        del want["use_feature_normalization"]
    if allow_tanh:
        del want["use_ReLU"]
    for k, v in want.items():
        got = getattr(args, k, v)
        if got != v:
            raise NotImplementedError("ope kernels support %s=%r only (got %r)" % (k, v, got))
