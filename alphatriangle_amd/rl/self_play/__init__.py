from .mcts_helpers import PolicyGenerationError, get_policy_target_from_visits, select_action_from_visits
from .nstep import NStepAccumulator
from .worker import SelfPlayWorker

__all__ = [
    "NStepAccumulator",
    "PolicyGenerationError",
    "SelfPlayWorker",
    "get_policy_target_from_visits",
    "select_action_from_visits",
]
