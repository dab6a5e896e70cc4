"""Headless training entry point (parity: alphatriangle/training/runner.py:28-319)."""

from __future__ import annotations

import logging
import time
from typing import Any

from ..config import AlphaTriangleMCTSConfig, EnvConfig, ModelConfig, SystemConfig, TrainConfig
from ..logging_config import setup_logging
from ..stats import RawMetricEvent, TrieyeConfig
from .components import TrainingComponents
from .logging_utils import log_configs_to_mlflow
from .loop import TrainingLoop
from .setup import count_parameters, setup_training_components

logger = logging.getLogger(__name__)


def _load_and_apply_initial_state(components: TrainingComponents) -> TrainingLoop:
    """Load checkpoint / buffer through the stats actor and apply them to the components."""
    loaded = components.trieye_actor.load_initial_state()
    loop: TrainingLoop
    if components.dist is not None and not components.dist.layout.colocated:
        from .split_loop import SplitRoleLoop

        loop = SplitRoleLoop(components, components.experience_source)
    else:
        loop = TrainingLoop(components)
    ckpt = loaded.checkpoint_data
    buf = loaded.buffer_data
    if ckpt is not None and ckpt.model_state_dict:
        # Auto-resume picks the latest run under the data root; when that run used another model or board configuration
        # its weights do not fit.  That is a fresh start (weights, optimizer, counters and replay), not a failed run.
        try:
            components.nn.set_weights(ckpt.model_state_dict)
        except Exception as exc:
            logger.warning(
                "Checkpoint of run %s (step %d) does not fit this model configuration (%s): starting fresh.",
                ckpt.run_name, ckpt.global_step, str(exc).splitlines()[0],
            )
            ckpt = None
            buf = None
    if ckpt is not None:
        logger.info("Applying loaded checkpoint data (Run: %s, Step: %d)", ckpt.run_name, ckpt.global_step)
        if ckpt.optimizer_state_dict:
            try:
                if components.trainer is not None:
                    components.trainer.load_optimizer_state(ckpt.optimizer_state_dict)
                logger.info("Optimizer state loaded and moved to device.")
            except Exception as exc:
                logger.error("Could not load optimizer state: %s. Optimizer might reset.", exc)
        loop.set_initial_state(ckpt.global_step, ckpt.episodes_played, ckpt.total_simulations_run)
        if components.gpu_actor is not None and ckpt.actor_state:
            components.gpu_actor.load_state(ckpt.actor_state)
    else:
        logger.info("No checkpoint data loaded. Starting fresh.")
        loop.set_initial_state(0, 0, 0)
    if buf is not None and buf.buffer_list:
        logger.info("Rebuilding replay buffer from %d saved experiences (priorities restart at max).", len(buf.buffer_list))
        components.buffer.load_list(buf.buffer_list)
        logger.info("Buffer size after rebuild: %d", len(components.buffer))
    else:
        logger.info("No buffer data loaded.")
    _align_learner_replicas(components, loop)
    components.nn.model.train()
    return loop


def _align_learner_replicas(components: TrainingComponents, loop: TrainingLoop) -> None:
    """Data-parallel learners all-reduce gradients and never exchange parameters afterwards, so they have to START from
    the same parameters, statistics and step count: the first learner's are sent to the others once (a different seed
    per rank, or a checkpoint only one rank found, would otherwise leave the replicas apart for the whole run)."""
    ctx = components.dist
    if ctx is None or components.trainer is None or not ctx.is_learner or ctx.num_learners <= 1:
        return
    import torch

    from ..parallel.weights import WeightWire

    wire = WeightWire(components.nn._orig_model)
    lead = ctx.layout.learner_ranks[0]
    if ctx.rank == lead:
        wire.pack()
    ctx.broadcast_learners(wire.buffer)
    counters = torch.tensor([loop.global_step, loop.episodes_played, loop.total_simulations_run], dtype=torch.float64,
                            device=wire.buffer.device)
    ctx.broadcast_learners(counters)
    if ctx.rank != lead:
        wire.unpack()
        step, episodes, sims = (int(v) for v in counters.tolist())
        if step != loop.global_step:
            logger.warning("rank %d resumes at the first learner's step %d (its own state was at %d).", ctx.rank, step, loop.global_step)
            loop.set_initial_state(step, episodes, sims)
        components.nn.sync_inference_weights()


def _save_final_state(loop: TrainingLoop) -> None:
    """Final checkpoint (with the replay when the persistence config says so), then wait for the writer thread."""
    actor = loop.trieye_actor
    if not actor:
        logger.warning("Cannot save final state: stats actor not available.")
        return
    logger.info("Saving final training state...")
    loop.save_state(is_best=False, save_buffer=bool(loop.trieye_config.persistence.SAVE_BUFFER))
    try:
        actor.wait_for_saves(timeout=120.0)
    except Exception as exc:
        logger.error("Waiting for the final save failed: %s", exc, exc_info=True)


def run_training(
    log_level_str: str,
    train_config_override: TrainConfig,
    trieye_config_override: TrieyeConfig,
    profile: bool,
    model_config_override: ModelConfig | None = None,
    mcts_config_override: AlphaTriangleMCTSConfig | None = None,
    env_config_override: EnvConfig | None = None,
    system_config_override: SystemConfig | None = None,
    dist_context: Any = None,
) -> int:
    """Run the pipeline; exit code 0 = completed or stopped cleanly, 1 = failed."""
    loop: TrainingLoop | None = None
    components: TrainingComponents | None = None
    exit_code = 1
    actor = None
    setup_logging(log_level_str)
    try:
        components, _ = setup_training_components(
            train_config_override,
            trieye_config_override,
            profile,
            model_config_override=model_config_override,
            mcts_config_override=mcts_config_override,
            env_config_override=env_config_override,
            system_config_override=system_config_override,
            dist_context=dist_context,
        )
        if not components:
            raise RuntimeError("Failed to initialize training components.")
        actor = components.trieye_actor
        log_configs_to_mlflow(components)
        total, trainable = count_parameters(components.nn.model)
        logger.info("Model Parameters: Total=%s, Trainable=%s", f"{total:,}", f"{trainable:,}")
        actor.log_params({"model_total_params": total, "model_trainable_params": trainable})

        loop = _load_and_apply_initial_state(components)
        loop.initialize_workers()
        loop.run()
        if loop.training_complete:
            exit_code = 0
        elif loop.training_exception:
            exit_code = 1
        else:
            exit_code = 0  # stopped early without an error
    except Exception as exc:
        logger.critical("An unhandled error occurred during training setup or execution: %s", exc, exc_info=True)
        if actor is not None:
            step = loop.global_step if loop else 0
            for name, value in (("training_status_code", -1.0), ("error_message_flag", 1.0)):
                try:
                    actor.log_event(RawMetricEvent(name=name, value=value, global_step=step, timestamp=time.time()))
                except Exception:  # pragma: no cover
                    pass
        exit_code = 1
    finally:
        if loop is not None:
            _save_final_state(loop)
            loop.cleanup_actors()
        if actor is not None:
            try:
                actor.force_process_and_log(loop.global_step if loop else 0)
                actor.shutdown(timeout=15.0)
            except Exception as exc:
                logger.error("Error shutting down the stats actor: %s", exc)
        logger.info("Training run finished with exit code %d.", exit_code)
    return exit_code
