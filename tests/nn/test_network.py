"""NeuralNetwork wrapper (mirrors the reference's tests/nn/test_network.py)."""

import numpy as np
import pytest
import torch

from alphatriangle_amd.env import GameState
from alphatriangle_amd.models import NetworkEvaluationError, NeuralNetwork


@pytest.fixture
def nn_interface(mock_model_config, mock_env_config, mock_train_config):
    torch.manual_seed(0)
    return NeuralNetwork(mock_model_config, mock_env_config, mock_train_config, torch.device("cpu"))


@pytest.fixture
def game(mock_env_config):
    return GameState(mock_env_config, initial_seed=123)


def test_init(nn_interface, action_dim):
    assert nn_interface.action_dim == action_dim and not nn_interface.model.training
    assert nn_interface.support.shape == (51,) and abs(nn_interface.delta_z - 0.4) < 1e-9 and nn_interface.engine is None


def test_state_to_tensors(nn_interface, game, mock_model_config):
    g, o = nn_interface._state_to_tensors(game)
    assert g.shape == (1, 1, 3, 3) and o.shape == (1, mock_model_config.OTHER_NN_INPUT_FEATURES_DIM)
    gb, ob = nn_interface._batch_states_to_tensors([game, game.copy(), game.copy()])
    assert gb.shape == (3, 1, 3, 3) and ob.shape == (3, 10)
    ge, oe = nn_interface._batch_states_to_tensors([])
    assert ge.shape == (0, 1, 3, 3) and oe.shape == (0, 10)


def test_evaluate_state_and_batch(nn_interface, game, action_dim):
    policy, value = nn_interface.evaluate_state(game)
    assert isinstance(policy, dict) and len(policy) == action_dim and isinstance(value, float)
    assert abs(sum(policy.values()) - 1.0) < 1e-5 and all(p >= 0 for p in policy.values())
    res = nn_interface.evaluate_batch([game, game.copy()])
    assert len(res) == 2 and abs(sum(res[1][0].values()) - 1.0) < 1e-5
    assert abs(res[0][1] - value) < 1e-5 and nn_interface.evaluate_batch([]) == []
    logits, values = nn_interface.evaluate_batch_arrays([game.core, game.copy()])
    assert logits.shape == (2, action_dim) and values.shape == (2,)
    probs = np.exp(logits[0] - logits[0].max())
    probs /= probs.sum()
    assert np.allclose(probs, [policy[i] for i in range(action_dim)], atol=1e-5) and abs(values[0] - value) < 1e-5


def test_get_set_weights_roundtrip(nn_interface, mock_model_config, mock_env_config, mock_train_config):
    w = nn_interface.get_weights()
    assert all(t.device.type == "cpu" for t in w.values()) and "conv_body.0.1.running_mean" in w
    other = NeuralNetwork(mock_model_config, mock_env_config, mock_train_config, torch.device("cpu"))
    changed = {k: (v + 0.5 if v.dtype.is_floating_point else v) for k, v in w.items()}
    other.set_weights(changed)
    back = other.get_weights()
    for k in w:
        assert torch.equal(back[k], changed[k])
    assert not other.model.training
    with pytest.raises(Exception):
        other.set_weights({"bogus": torch.zeros(1)})


def test_nan_features_raise(nn_interface, game, monkeypatch):
    def bad(*_a, **_k):
        return {"grid": np.full((1, 3, 3), np.nan, dtype=np.float32), "other_features": np.zeros(10, dtype=np.float32)}

    monkeypatch.setattr("alphatriangle_amd.models.network.extract_state_features", bad)
    with pytest.raises(NetworkEvaluationError):
        nn_interface.evaluate_state(game)
    with pytest.raises(NetworkEvaluationError):
        nn_interface.evaluate_batch([game])


def test_uniform_fallback_when_policy_mass_vanishes(nn_interface, game):
    probs = nn_interface._finalise_probs(np.zeros(9, dtype=np.float32), game, "t")
    valid = game.valid_actions()
    assert abs(probs.sum() - 1.0) < 1e-6 and all(probs[a] > 0 for a in valid)
    renorm = nn_interface._finalise_probs(np.full(9, 0.2, dtype=np.float32), game, "t")
    assert abs(renorm.sum() - 1.0) < 1e-6
